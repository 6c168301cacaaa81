"""Development script (not a pytest test; needs the library built with -DMPC_DEV_STAMP=1|2|3 and MPC_LIB_PATH; run with
MPC_GROUPS=1): when do the waves of the chosen kernel's launch in round R of bench.py's solve start and end
(mpc_set_round_limit stops the solve there)?  1: K1a, 2: the fused K1b+K1c kernel, 3: the step kernel."""
import os, sys, ctypes as C, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
import model_predictive_control_amd as mp
from model_predictive_control_amd import _lib

dev = torch.device("cuda:0")
MODEL = int(os.environ.get("TRACE_MODEL", 0))
N, B = int(os.environ.get("TRACE_N", 12 if MODEL else 20)), int(os.environ.get("TRACE_B", 65536))
L = _lib.load()
X0 = torch.tensor(bench.synthetic_states(MODEL, 0, B), dtype=torch.float64, device=dev)
cl = torch.tensor(bench.straight_centerline(), dtype=torch.float64, device=dev)
U0 = torch.tensor([1.0, 0.0], dtype=torch.float64, device=dev).repeat(B, N)
eng = mp.BatchedMPC(mp.default_config(MODEL, N), dev)
NS = 65536
buf = (C.c_longlong * (4 * NS))()
for R in [int(a) for a in sys.argv[1:]] or [48, 152, 304]:
    eng.set_round_limit(R)
    try:
        eng.solve(X0, cl, U0)
    except Exception as e:
        pass
    torch.cuda.synchronize()
    assert L.mpc_dev_stamps(buf) == 0
    a = np.frombuffer(buf, dtype=np.int64).reshape(-1, 4)
    newest = a[:, 1].max()
    a = a[(a[:, 1] > newest - 100000) & (a[:, 0] > 0)]         # the last launch: stamps within 1 ms of the newest
    t0, t1, hw, misc = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
    base = t0.min()
    d = (t1 - t0) / 100.0
    live = d > 1.0
    end = (t1 - base) / 100.0
    start = (t0 - base) / 100.0
    n1, n2 = (misc >> 8) & 255, (misc >> 16) & 255
    print("round %4d: %6d waves stamped, %6d longer than 1 us | launch %.1f us | wave duration min %.1f median %.1f 90%% %.1f 99%% %.1f max %.1f us"
          % (R, len(a), live.sum(), end.max(), d[live].min(), np.median(d[live]), np.percentile(d[live], 90), np.percentile(d[live], 99), d[live].max()))
    print("     starts: median %.1f 90%% %.1f 99%% %.1f max %.1f us;  ends: 50%% %.1f 90%% %.1f 99%% %.1f us"
          % (np.median(start[live]), np.percentile(start[live], 90), np.percentile(start[live], 99), start[live].max(),
             np.median(end[live]), np.percentile(end[live], 90), np.percentile(end[live], 99)))
    # how many waves are resident over time
    ts = np.linspace(0, end.max(), 11)[1:-1]
    print("     waves resident at", " ".join("%.0f us:%d" % (t, ((start <= t) & (end > t) & live).sum()) for t in ts))
    if os.environ.get("TRACE_STEP"):
        n3 = (misc >> 24) & 255
        import collections as cc
        print("     longest agent-step per wave (x 10 ns): median %.0f, 90%% %.0f, 99%% %.0f, max %d (255 = capped); phase of the longest where it is >= 1.5 us: %s"
              % (np.median(n2[live]), np.percentile(n2[live], 90), np.percentile(n2[live], 99), n2[live].max(),
                 dict(sorted(cc.Counter(n3[live & (n2 >= 150)].tolist()).items()))))
        print("     phase of the longest agent-step, all waves: %s" % dict(sorted(cc.Counter(n3[live].tolist()).items())))
        late = live & (end > np.percentile(end[live], 95))
        print("     the 5%% of the waves that end last: agent-steps median %.0f (all: %.0f), longest agent-step median %.0f x 10 ns (all: %.0f)"
              % (np.median(n1[late]), np.median(n1[live]), np.median(n2[late]), np.median(n2[live])))
    if n1.max() > 0 and os.environ.get("TRACE_SLOW"):
        slow = live & (d > 2.0 * np.median(d[live]))
        print("     waves longer than twice the median: %d; of those with counter 1 > 0: %d, with counter 2 > 0: %d, neither: %d; among the others counter 1 > 0: %d, counter 2 > 0: %d"
              % (slow.sum(), (slow & (n1 > 0)).sum(), (slow & (n2 > 0)).sum(), (slow & (n1 == 0) & (n2 == 0)).sum(),
                 (live & ~slow & (n1 > 0)).sum(), (live & ~slow & (n2 > 0)).sum()))
        print("     counter 1 among the slow ones: %s" % dict(sorted(collections.Counter(n1[slow].tolist()).items())))
    if n1.max() > 0:
        print("     counter 1 per wave (K1a: stages out of range; step: agent-steps): median %.0f max %d; counter 2: median %.0f max %d; duration vs counter 1 corr %.2f"
              % (np.median(n1[live]), n1[live].max(), np.median(n2[live]), n2[live].max(), np.corrcoef(n1[live], d[live])[0, 1]))
