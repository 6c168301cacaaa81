"""Development script (not a pytest test; needs the library built with -DMPC_DEV_K1A_TIMES, MPC_LIB_PATH):
when do the waves of one K1a launch start and end, and on which SIMDs?"""
import os, sys, ctypes as C, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import model_predictive_control_amd as mp
from model_predictive_control_amd import _lib

dev = torch.device("cuda:0")
N = 20
L = _lib.load()
sizes = [int(a) for a in sys.argv[1:]] or [16384, 32768, 65536, 98304]
Bm = max(sizes)
X0 = torch.tensor(bench.synthetic_states(0, 0, Bm), dtype=torch.float64, device=dev)
cl = torch.tensor(bench.straight_centerline(), dtype=torch.float64, device=dev)
rng = np.random.default_rng(1)
U = torch.tensor(np.tile([1.0, 0.0], (Bm, N)) + 0.05 * rng.standard_normal((Bm, 2 * N)), dtype=torch.float64, device=dev)
eng = mp.BatchedMPC(mp.default_config(0, N), dev)
buf = (C.c_longlong * (4 * 16384))()
for B in sizes:
    for rep in range(3):
        eng.eval_cost_grad(X0[:B], cl, U[:B])
    torch.cuda.synchronize()
    assert L.mpc_dev_k1a_times(buf) == 0
    a = np.frombuffer(buf, dtype=np.int64).reshape(-1, 4)[: B // 32]
    t0, t1, hw, xcc = a[:, 0], a[:, 1], a[:, 2], a[:, 3] & 15
    base = t0.min()
    us = lambda t: (t - base) / 100.0                     # 100 MHz
    simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    where = collections.Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist(), simd.tolist()))
    per_simd = collections.Counter(where.values())
    cus = collections.Counter((x, s, h, c) for (x, s, h, c, _) in where.elements())
    print("B %6d: %5d waves | start spread %.1f us (median %.1f, 90%% %.1f) | wave duration median %.1f max %.1f us | kernel %.1f us"
          % (B, len(a), us(t0).max(), np.median(us(t0)), np.percentile(us(t0), 90), np.median(t1 - t0) / 100.0,
             (t1 - t0).max() / 100.0, us(t1).max()))
    print("     distinct SIMDs %d, waves per used SIMD histogram %s; distinct CUs %d, waves per CU min/max %d/%d; per XCC %s"
          % (len(where), dict(sorted(per_simd.items())), len(cus), min(cus.values()), max(cus.values()),
             dict(sorted(collections.Counter(xcc.tolist()).items()))))
    # the waves in start order: start time of every 128th
    order = np.argsort(t0)
    print("     start of wave #k (us):", " ".join("%d:%.1f" % (k, us(t0[order[k]])) for k in range(0, len(a), max(1, len(a) // 12))))
    late = order[-max(1, len(a) // 20):]
    print("     the last 5%% to start: blocks %d..%d (median %d), durations median %.1f us"
          % (late.min(), late.max(), np.median(late), np.median((t1 - t0)[late]) / 100.0))
