"""Development script (not a pytest test; needs the library built with -DMPC_DEV_K1A_TIMES, MPC_LIB_PATH; run with
MPC_GROUPS=1): the waves of the K1a launch of round R of bench.py's solve (mpc_set_round_limit stops it there)."""
import os, sys, ctypes as C, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import model_predictive_control_amd as mp
from model_predictive_control_amd import _lib

dev = torch.device("cuda:0")
N, B = 20, int(os.environ.get("TRACE_B", 65536))
L = _lib.load()
X0 = torch.tensor(bench.synthetic_states(0, 0, B), dtype=torch.float64, device=dev)
cl = torch.tensor(bench.straight_centerline(), dtype=torch.float64, device=dev)
U0 = torch.tensor([1.0, 0.0], dtype=torch.float64, device=dev).repeat(B, N)
eng = mp.BatchedMPC(mp.default_config(0, N), dev)
buf = (C.c_longlong * (4 * 16384))()
for R in [int(a) for a in sys.argv[1:]] or [8, 48, 152, 304]:
    eng.set_round_limit(R)
    try:
        eng.solve(X0, cl, U0)
    except Exception as e:
        pass
    torch.cuda.synchronize()
    assert L.mpc_dev_k1a_times(buf) == 0
    a = np.frombuffer(buf, dtype=np.int64).reshape(-1, 4)
    # the last launch: the blocks whose stamps are the newest (within 1 ms of the newest end)
    newest = a[:, 1].max()
    a = a[(a[:, 1] > newest - 100000) & (a[:, 0] > 0)]
    t0, t1, hw, xcc = a[:, 0], a[:, 1], a[:, 2], a[:, 3] & 15
    base = t0.min()
    d = (t1 - t0) / 100.0
    live = d > 2.0                                         # (blocks beyond the requests leave at once)
    simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    where = collections.Counter(zip(xcc[live].tolist(), se[live].tolist(), sh[live].tolist(), cu[live].tolist(), simd[live].tolist()))
    print("round %4d: %5d blocks, %5d with requests | start spread %.1f us | duration of those: min %.1f median %.1f max %.1f us | launch %.1f us"
          % (R, len(a), live.sum(), (t0.max() - base) / 100.0, d[live].min(), np.median(d[live]), d[live].max(), (t1.max() - base) / 100.0))
    print("     waves per used SIMD histogram %s (distinct SIMDs %d)" % (dict(sorted(collections.Counter(where.values()).items())), len(where)))
    print("     duration percentiles 10/50/90/99: %s" % np.round(np.percentile(d[live], [10, 50, 90, 99]), 1))
    nfall, nmid, nslow = (a[:, 3] >> 8) & 255, (a[:, 3] >> 16) & 255, (a[:, 3] >> 24) & 255
    slow = live & (d > np.percentile(d[live], 97))
    print("     waves with a stage outside the fast range: %d (stages: %s); with |delta| > 0.75 somewhere: %d; with the library route: %d"
          % ((nfall[live] > 0).sum(), dict(sorted(collections.Counter(nfall[live & (nfall > 0)].tolist()).items())), (nmid[live] > 0).sum(), (nslow[live] > 0).sum()))
    print("     the slowest 3%% of the waves: stages outside the range median %.0f, |delta| > 0.75 turns median %.0f, library turns median %.0f; duration vs fallback stages corr %.2f"
          % (np.median(nfall[slow]), np.median(nmid[slow]), np.median(nslow[slow]), np.corrcoef(nfall[live], d[live])[0, 1]))
