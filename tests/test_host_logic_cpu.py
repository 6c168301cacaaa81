"""CPU tests of the host-side mirror of the reference interface (no kernels are launched)."""
import os
import numpy as np
import pytest

from model_predictive_control_amd import main as mpc_main
from model_predictive_control_amd.car_dynamics import (DEFAULT_PARAMS, PARAM_NAMES,
                                                       KinematicBicyclePacejka,
                                                       KinematicBicycleSimplified)
from model_predictive_control_amd.controller import _tiled
from model_predictive_control_amd.sharding import shard_bounds


def test_get_centerline_matches_main_py():
    c = mpc_main.get_centerline(100, True)                       # main.py:13
    assert c.shape == (100, 2) and np.allclose(c[:3], [[-0.1, 0], [0.0, 0], [0.1, 0]])
    flat = c.ravel(order="F")                                    # main.py:113
    assert np.array_equal(flat[:100], c[:, 0]) and np.array_equal(flat[100:], c[:, 1])
    k = mpc_main.get_centerline(100, False)                      # main.py:15-22
    assert np.allclose(np.hypot(k[:, 0], k[:, 1] - 5), 5)


def test_problem_container_mirrors_main_py():
    model = KinematicBicyclePacejka()
    prob = mpc_main.create_casadi_problem(model, 12, 100, 1.0, 1.0, 0.32)
    assert prob.n == 24 and prob.m == 72
    assert prob.param.shape == (6 + 200 + 22,)                   # main.py:30,:119
    assert np.array_equal(prob.C.lowerbound, np.tile([-1.0, -0.32], 12))   # main.py:55
    assert np.array_equal(prob.C.upperbound, np.tile([1.0, 0.32], 12))     # main.py:56
    assert np.all(np.isinf(prob.D.lowerbound)) and np.all(np.isinf(prob.D.upperbound))  # main.py:57
    assert np.array_equal(prob.param[206:], DEFAULT_PARAMS)
    assert len(PARAM_NAMES) == 22 and PARAM_NAMES[7] == "mass" and PARAM_NAMES[21] == "cr2"


def test_input_to_matrix_is_column_major():
    """car_dynamics.py:149-157: flat [d0, delta0, d1, delta1, ...] <-> (2, N)."""
    model = KinematicBicycleSimplified()
    U = np.arange(8.0)
    M = model.input_to_matrix(U)
    assert M.shape == (2, 4) and np.array_equal(M[:, 0], [0, 1]) and np.array_equal(M[1], [1, 3, 5, 7])


def test_wrap_to_pi_host_helper():
    model = KinematicBicyclePacejka()
    a = np.array([-4.0, -np.pi, 0.0, 3.0, np.pi, 7.0])
    w = model.wrap_to_pi(a)
    assert np.all(w >= -np.pi) and np.all(w < np.pi) and np.allclose(np.sin(w), np.sin(a))


def test_bounds_must_be_stage_periodic():
    assert np.array_equal(_tiled(np.tile([-1, -0.32], 5), 2, "lb"), [-1, -0.32])
    with pytest.raises(ValueError):
        _tiled([-1, -0.32, -1, -0.3], 2, "lb")


@pytest.mark.parametrize("B,world", [(65536, 8), (10, 3), (2, 4), (0, 2), (524288, 8)])
def test_shard_bounds_partition(B, world):
    spans = [shard_bounds(B, r, world) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == B
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 == b0 and a1 >= a0
    sizes = [hi - lo for lo, hi in spans]
    assert max(sizes) - min(sizes) <= 1


def test_bezier_mirror_matches_reference_fixture(ref_golden):
    """bezier_curves.py:19-48 for i = 1..10 (vectors recorded from the reference module)."""
    from model_predictive_control_amd import bezier_curves as bc
    for i in range(1, 11):
        P, tca = bc.get_bezier_control_points(i)
        assert np.array_equal(P, ref_golden["bez_P"][i - 1]) and tca == ref_golden["bez_tca"][i - 1]
        x, y = bc.bezier_curve(ref_golden["bez_j"], P)
        assert np.allclose(np.stack([x, y], 1), ref_golden["bez_xy"][i - 1], rtol=1e-13, atol=1e-12)
    x, y = bc.bezier_curve(0.25, bc.get_bezier_control_points(5)[0])
    assert np.isclose(x, 44.467364988500634) and np.isclose(y, 0.38818359375)
    assert bc.binomial_coefficient(5, 2) == 10
    tab = bc.lane_change_centerlines(S=100)
    assert tab.shape == (10, 200) and np.allclose(tab[:, 0], 0) and np.allclose(tab[:, 99], 10.0)
    assert np.allclose(tab[:, 199], 3.75 * 10.0 / 193.76417765201978)


def _wave_sum_order(v, n):
    """The order in which the wavefront reductions of csrc/mpc_solver.hpp add a vector with element j on lane j
    (row_sum16: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror inside every 16-lane row, then
    cross_rows<NROWS(n)>), replayed lane by lane in IEEE doubles."""
    x = np.zeros(64)
    x[:n] = v
    for partner in (lambda l: l ^ 1, lambda l: l ^ 2, lambda l: (l & ~7) | (7 - (l & 7)), lambda l: (l & ~15) | (15 - (l & 15))):
        x = np.array([x[l] + x[partner(l)] for l in range(64)])
    r = [x[0], x[16], x[32], x[48]]
    assert all(np.all(x[16 * i:16 * i + 16] == r[i]) for i in range(4))       # every lane of a row holds the row total
    return r[0] + r[1] if n <= 32 else (r[0] + r[1]) + r[2] if n <= 48 else (r[0] + r[1]) + (r[2] + r[3])


def _tree_sum_by_stages(v, n):
    """TreeSum of csrc/mpc_solver.hpp (the thread-per-agent PH_W_LS_G, chain_block): the same vector met two elements
    at a time -- stage k = elements 2k, 2k + 1 -- in DESCENDING stage order, with three holders."""
    h1 = h2 = h3 = 0.0
    r = [0.0, 0.0, 0.0, 0.0]
    for k in range(n // 2 - 1, -1, -1):
        t = v[2 * k] + v[2 * k + 1]
        if k & 1:
            h1 = t; continue
        q = t + h1; h1 = 0.0
        if k & 2:
            h2 = q; continue
        hf = q + h2; h2 = 0.0
        if k & 4:
            h3 = hf; continue
        r[k >> 3] = hf + h3; h3 = 0.0
    return r[0] + r[1] if n <= 32 else (r[0] + r[1]) + r[2] if n <= 48 else (r[0] + r[1]) + (r[2] + r[3])


def test_thread_serial_tree_sum_has_the_bits_of_the_wave_reduction():
    """The thread-per-agent form of PH_W_LS_G must give the wavefront form's ||p||^2, grad'p and ||x||^2 bit for bit:
    both are balanced trees over the lanes in lane order -- checked here on the host for every horizon 1 .. 32 with
    values of mixed sign and magnitude (where any other order of additions rounds differently)."""
    rng = np.random.default_rng(0)
    differs_from_sequential = 0
    for N in range(1, 33):
        n = 2 * N
        for _ in range(20):
            v = rng.standard_normal(n) * 10.0 ** rng.integers(-6, 7, n)
            a, b = _wave_sum_order(v, n), _tree_sum_by_stages(v, n)
            assert a == b and np.signbit(a) == np.signbit(b), (N, a, b)
            differs_from_sequential += a != float(np.add.reduce(v))
    assert differs_from_sequential > 100          # (the test can tell orders apart)


def test_bench_uses_only_a_counter_summary_of_the_running_build(tmp_path, monkeypatch):
    """bench.py's `roofline.traffic` comes from a committed rocprofv3 counter summary -- only from one taken on the
    RUNNING library (its `library_source_sha256` = mpc_source_hash, which _lib.build() compiles in): counters of an
    older build are another kernel's traffic, and the line then carries `traffic: null` (a stale summary
    passed the workload-string guard).  Among matching summaries the most recent file wins."""
    import json as _json
    import time as _time
    import bench
    from model_predictive_control_amd import _lib
    mine = _lib.library_hash()
    assert mine == _lib.source_hash() and len(mine) == 64         # the library in the tree is built from the tree
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.pmc_profile() == (None, None)                    # nothing committed
    (prof / "r09z_pmc_summary.json").write_text(_json.dumps({"library_source_sha256": "0" * 64, "hbm_bytes_per_solve": 1.0}))
    (prof / "r09y_pmc_summary.json").write_text(_json.dumps({"hbm_bytes_per_solve": 2.0}))          # no hash at all
    assert bench.pmc_profile() == (None, None)                    # another build's counters are not used
    (prof / "r04a_pmc_summary.json").write_text(_json.dumps({"library_source_sha256": mine, "hbm_bytes_per_solve": 3.0}))
    d, src = bench.pmc_profile()
    assert d["hbm_bytes_per_solve"] == 3.0 and src == os.path.join("profiles", "r04a_pmc_summary.json")
    _time.sleep(0.05)
    (prof / "r04b_pmc_summary.json").write_text(_json.dumps({"library_source_sha256": mine, "hbm_bytes_per_solve": 4.0}))
    assert bench.pmc_profile()[0]["hbm_bytes_per_solve"] == 4.0
