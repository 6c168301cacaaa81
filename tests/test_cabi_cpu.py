"""CPU tests of the drop-in boundary: the C-ABI library loads without a GPU, exports every symbol
declared in include/mpc_hip.h, its config struct matches the ctypes mirror byte for byte, and the
host-only entry points behave (no compute call is made here)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

import model_predictive_control_amd as mp
from model_predictive_control_amd import _lib


@pytest.fixture(scope="module")
def L():
    _lib.build()
    return _lib.load()


def _declared_functions():
    hdr = open(os.path.join(ROOT, "include", "mpc_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(mpc_[a-z_0-9]+)\s*\(", hdr)))


def test_every_declared_symbol_is_exported(L):
    names = _declared_functions()
    assert len(names) >= 17
    for n in names:
        assert hasattr(L, n), f"{n} is declared in include/mpc_hip.h but not exported"
    assert sorted(_lib.EXPORTS) == names


def test_config_struct_layout_matches_header(tmp_path):
    """sizeof/offsetof from gcc on the real header == the ctypes mirror."""
    fields = [f[0] for f in _lib.MpcConfig._fields_]
    src = tmp_path / "layout.c"
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{ROOT}/include/mpc_hip.h"',
             'int main(void){', 'printf("%zu\\n", sizeof(mpc_config));']
    lines += [f'printf("%zu\\n", offsetof(mpc_config, {f}));' for f in fields]
    lines += ['return 0;}']
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-o", str(exe), str(src)])
    out = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert out[0] == C.sizeof(_lib.MpcConfig)
    for f, off in zip(fields, out[1:]):
        assert getattr(_lib.MpcConfig, f).offset == off, f


def test_default_config_holds_reference_constants(L, O):
    cfg = mp.default_config(mp.MODEL_PACEJKA, 12)
    assert (cfg.N, cfg.S, cfg.nfe, cfg.lbfgs_memory) == (12, 100, 4, 12)       # main.py:68-70, controller.py:36
    assert (cfg.max_iter, cfg.max_outer, cfg.hess_heuristic) == (1000, 1000, 15)  # controller.py:31,32,45
    assert (cfg.alm_eps, cfg.alm_delta, cfg.Sigma0) == (1e-6, 1e-4, 1e5)        # controller.py:41-43
    assert list(cfg.cost_w) == [0.5, 1, 1, .5, 0.1, 0.01]                       # car_dynamics.py:230
    assert list(cfg.u_ub) == [1.0, 0.32] and list(cfg.u_lb) == [-1.0, -0.32]    # main.py:55-56,:82
    assert cfg.Ts == 0.05 and cfg.v_ref == 1.0
    # the oracle's independent default table says the same, field by field
    ocfg = O.default_config(O.MODEL_PACEJKA, 12)
    for name, _ in _lib.MpcConfig._fields_:
        a, b = getattr(cfg, name), getattr(ocfg, name)
        if hasattr(a, "__len__"):
            assert list(a) == list(b), name
        else:
            assert a == b, name
    assert L.mpc_nx(C.byref(cfg)) == 6 and L.mpc_m(C.byref(cfg)) == 0
    cfg.constr_mode = mp.CONSTR_STATE_SQ
    assert L.mpc_m(C.byref(cfg)) == 72                                          # controller.py:21


def test_bad_arguments_return_codes_not_exceptions(L):
    cfg = _lib.MpcConfig()
    assert L.mpc_default_config(C.byref(cfg), 0, 0) != 0          # horizon out of range
    assert L.mpc_default_config(C.byref(cfg), 7, 12) != 0         # unknown model
    assert b"bad model or horizon" in L.mpc_last_error()
    with pytest.raises(ValueError):
        mp.default_config(mp.MODEL_PACEJKA, 1000)


def test_config_is_validated_before_the_device_is_touched(L):
    """mpc_create refuses a configuration the kernels cannot run with MPC_E_ARG (-1) -- checked before any
    HIP call, so it holds without a GPU: non-positive iteration limits (a negative budget used to end as
    'round limit reached'), Sigma_0 = 0 (alpaqa's separate initial-penalty path is not restated: the
    kernels divide by Sigma), inverted Lipschitz or box bounds, non-positive Ts."""
    bad = [dict(max_total_inner=-5), dict(max_iter=0), dict(max_outer=0), dict(Sigma0=0.0), dict(Sigma0=-1.0),
           dict(L_min=1.0, L_max=0.5), dict(Ts=0.0), dict(Ts=float("nan")), dict(tau_min=0.0), dict(max_total_evals=-1),
           dict(u_lb=[1.0, 0.0], u_ub=[-1.0, 0.3]), dict(S=2), dict(nfe=0), dict(lbfgs_memory=0), dict(alm_eps=0.0)]
    for kw in bad:
        cfg = mp.default_config(mp.MODEL_KINEMATIC, 20, **kw)
        h = C.c_void_p()
        assert L.mpc_create(C.byref(cfg), 0, C.byref(h)) == -1, kw
        assert L.mpc_last_error()
    # a valid configuration gets past the check (and then fails on the missing device, or succeeds)
    cfg = mp.default_config(mp.MODEL_KINEMATIC, 20)
    h = C.c_void_p()
    rc = L.mpc_create(C.byref(cfg), 0, C.byref(h))
    assert rc in (0, -1, -2)
    if rc == 0:
        L.mpc_destroy(h)
    else:
        assert b"mpc_create" in L.mpc_last_error() or b"hip" in L.mpc_last_error().lower()


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mp.BatchedMPC(mp.default_config(mp.MODEL_KINEMATIC, 20))
    # the C-ABI itself reports the missing device through its error code
    L = _lib.load()
    h = C.c_void_p()
    cfg = mp.default_config(mp.MODEL_KINEMATIC, 20)
    assert L.mpc_create(C.byref(cfg), 0, C.byref(h)) != 0
    assert L.mpc_last_error()


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under model_predictive_control_amd/ may use it."""
    pkg = os.path.join(ROOT, "model_predictive_control_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("no CPU fallback", ""), os.path.join(dirpath, f)
    code = ("import sys; sys.path.insert(0, %r); import model_predictive_control_amd; "
            "assert not any('oracle' in m for m in sys.modules), 'oracle imported'" % ROOT)
    subprocess.check_call([sys.executable, "-c", code])


def test_debug_record_names_follow_the_record_layout(L):
    """mpc_debug_record_names(): one name per slot in use of the per-agent solver record (static_assert in
    mpc_api.hip ties the count to the enum), no duplicates, within MPC_NREC."""
    names = L.mpc_debug_record_names().decode().split(",")
    assert len(names) == len(set(names)) == 64 and len(names) <= _lib.NREC
    for must in ("L", "gamma", "tau", "nJ", "lidx", "lfull", "nevals", "k", "outer", "fallback", "phase"):
        assert must in names
