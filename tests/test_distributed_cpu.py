"""gloo tests (world_size 2 and 8) of the multi-GPU path's host logic: contiguous sharding of the agent
batch, independent per-rank work, and the final gather of the controls (the only collective)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, B, k, out_dir):
    sys.path.insert(0, ROOT)
    from model_predictive_control_amd.sharding import gather_controls, shard_bounds
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_bounds(B, rank, world)
    # stand-in for the per-rank solve: a deterministic function of the GLOBAL agent index, so the
    # gathered result can be checked against a single-process evaluation
    idx = torch.arange(lo, hi, dtype=torch.float64)
    local = torch.stack([idx * 10 + j for j in range(k)], 1) if hi > lo else torch.zeros(0, k, dtype=torch.float64)
    full = gather_controls(local, B, dst=0)
    dist.barrier()
    if rank == 0:
        np.save(os.path.join(out_dir, "full.npy"), full.numpy())
    else:
        assert full is None
    dist.destroy_process_group()


def _shard_worker(rank, world, port, B, out_dir):
    """What every rank of bench.py does before its solve: take its contiguous shard of the GLOBAL
    synthetic batch.  The rows must be the ones a single process generates for those agent indices, so
    that every sharding of the batch solves the same problems (and the per-N checksums are comparable)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_states", os.path.join(ROOT, "bench.py"))
    from model_predictive_control_amd.sharding import gather_controls, shard_bounds
    # bench.py imports the HIP package at module level; only its pure-NumPy generator is used here
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    lo, hi = shard_bounds(B, rank, world)
    for model in (0, 1):
        local = torch.from_numpy(bench.synthetic_states(model, lo, hi))
        full = gather_controls(local, B, dst=0)
        if rank == 0:
            np.save(os.path.join(out_dir, f"states{model}.npy"), full.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shards_are_the_rows_of_the_single_process_batch(tmp_path):
    B, world = 10000, 2          # crosses the generator's 4096-agent blocks at a ragged boundary
    port = _free_port()
    mp.spawn(_shard_worker, args=(world, port, B, str(tmp_path)), nprocs=world, join=True)
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_states", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    for model, nx in ((0, 4), (1, 6)):
        got = np.load(os.path.join(str(tmp_path), f"states{model}.npy"))
        ref = bench.synthetic_states(model, 0, B)
        assert got.shape == (B, nx) and np.array_equal(got, ref)
        # and a shard taken on its own equals the same slice (pure function of the global index)
        assert np.array_equal(bench.synthetic_states(model, 4000, 4200), ref[4000:4200])


def _run(B, k, tmp_path, world=2):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, B, k, str(tmp_path)), nprocs=world, join=True)
    full = np.load(os.path.join(str(tmp_path), "full.npy"))
    expect = np.stack([np.arange(B) * 10.0 + j for j in range(k)], 1)
    assert full.shape == (B, k) and np.array_equal(full, expect)


def test_gather_even_shards(tmp_path):
    _run(64, 40, tmp_path)


def test_gather_ragged_shards(tmp_path):
    _run(7, 4, tmp_path)


def test_gather_with_empty_shard(tmp_path):
    _run(1, 3, tmp_path)


def _config4_worker(rank, world, port, B, k, out_dir):
    """BASELINE config 4's row counts on eight ranks: every rank builds its shard as a pure function of the
    global agent index (what bench.py's generator guarantees), rank 0 gathers and checks the rows in place --
    524 288 x k doubles are 16 MB at k = 4, so nothing is written to disk but the verdict."""
    sys.path.insert(0, ROOT)
    from model_predictive_control_amd.sharding import gather_controls, shard_bounds
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_bounds(B, rank, world)
    idx = torch.arange(lo, hi, dtype=torch.float64)
    local = torch.stack([idx * 8 + j for j in range(k)], 1)
    full = gather_controls(local, B, dst=0)
    ok = True
    if rank == 0:
        expect = torch.stack([torch.arange(B, dtype=torch.float64) * 8 + j for j in range(k)], 1)
        ok = full.shape == (B, k) and torch.equal(full, expect)
    else:
        ok = full is None
    # shards tile [0, B) in rank order, sizes differ by at most one
    bounds = [shard_bounds(B, r, world) for r in range(world)]
    ok = ok and bounds[0][0] == 0 and bounds[-1][1] == B and all(bounds[r][1] == bounds[r + 1][0] for r in range(world - 1))
    ok = ok and max(h - l for l, h in bounds) - min(h - l for l, h in bounds) <= 1
    dist.barrier()
    with open(os.path.join(out_dir, f"ok{rank}"), "w") as f:
        f.write("ok" if ok else "bad")
    dist.destroy_process_group()


def test_eight_ranks_at_config4_row_counts(tmp_path):
    """BASELINE config 4: 524 288 agents sharded over 8 ranks (65 536 each), and a ragged total that no rank count
    divides -- shard_bounds / gather_controls with world_size 8 (gloo on the CPU; the 8-GPU run is the driver's)."""
    for sub, B in (("even", 524288), ("ragged", 524288 - 8 * 3 + 5)):
        d = tmp_path / sub
        d.mkdir()
        port = _free_port()
        mp.spawn(_config4_worker, args=(8, port, B, 4, str(d)), nprocs=8, join=True)
        for r in range(8):
            assert (d / f"ok{r}").read_text() == "ok", (sub, r)


def _probe_worker(rank, world, port, out_dir):
    """The collective capability probe: asked once, the same answer on every rank; with the answer forced to
    "no gather" every rank takes the all_gather path and the result is the same; an error inside the real
    collective is raised, never mapped to another collective."""
    sys.path.insert(0, ROOT)
    from model_predictive_control_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = sharding.gather_supported()
    assert ok is True and sharding.gather_supported() is True        # gloo implements gather; second call: cached
    assert len(sharding._GATHER_OK) == 1
    B, k = 11, 3
    lo, hi = sharding.shard_bounds(B, rank, world)
    local = torch.arange(lo * k, hi * k, dtype=torch.float64).reshape(hi - lo, k)
    a = sharding.gather_controls(local, B, dst=0)
    for key in list(sharding._GATHER_OK):
        sharding._GATHER_OK[key] = False                             # a backend build without gather
    b = sharding.gather_controls(local, B, dst=0)
    if rank == 0:
        assert torch.equal(a, b) and torch.equal(a, torch.arange(B * k, dtype=torch.float64).reshape(B, k))
    else:
        assert a is None and b is None
    # a failure of the collective itself is an error on the rank it happens on
    real = dist.all_gather
    def boom(*args, **kw):
        raise RuntimeError("gather: peer lost")                      # the text the old fallback matched on
    dist.all_gather = boom
    try:
        raised = False
        try:
            sharding.gather_controls(local, B, dst=0)
        except RuntimeError:
            raised = True
        assert raised
    finally:
        dist.all_gather = real
    dist.barrier()
    with open(os.path.join(out_dir, f"ok{rank}"), "w") as f:
        f.write("ok")
    dist.destroy_process_group()


def test_gather_capability_is_probed_once_and_errors_are_raised(tmp_path):
    port = _free_port()
    mp.spawn(_probe_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(os.path.join(str(tmp_path), "ok0")) and os.path.exists(os.path.join(str(tmp_path), "ok1"))
