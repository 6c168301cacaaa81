"""Development script (not a pytest test; library built with -DMPC_DEV_STAMP=5, MPC_LIB_PATH; MPC_GROUPS=1): how long does
each agent of the persistent kernel's phase take, when is it claimed, how many evaluations does it walk?"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
import model_predictive_control_amd as mp
from model_predictive_control_amd import _lib

dev = torch.device("cuda:0")
MODEL = int(os.environ.get("TRACE_MODEL", 1))
N, B = int(os.environ.get("TRACE_N", 12 if MODEL else 20)), int(os.environ.get("TRACE_B", 65536))
L = _lib.load()
X0 = torch.tensor(bench.synthetic_states(MODEL, 0, B), dtype=torch.float64, device=dev)
cl = torch.tensor(bench.straight_centerline(), dtype=torch.float64, device=dev)
U0 = torch.tensor([1.0, 0.0], dtype=torch.float64, device=dev).repeat(B, N)
eng = mp.BatchedMPC(mp.default_config(MODEL, N), dev)
buf = (C.c_longlong * (4 * 65536))()
for rep in range(2):
    U, _, st = eng.solve(X0, cl, U0)
torch.cuda.synchronize()
info = eng.last_solve_info()
assert L.mpc_dev_stamps(buf) == 0
n = info["solo_agents"]
full = np.frombuffer(buf, dtype=np.int64).reshape(-1, 4)
a = full[:n]
t0, t1, trips, agent = a[:, 0], a[:, 1], a[:, 2] & 0xFFFFF, a[:, 3] & 0xFFFFF
t_adv, t_roll, t_recs, t_adj = a[:, 2] >> 20, a[:, 3] >> 20, full[32768:32768 + n, 0], full[32768:32768 + n, 1]
base = t0.min()
dur = (t1 - t0) / 100.0
print("solo agents %d, rounds %d; kernel span %.1f ms; per agent: duration median %.0f us, 90%% %.0f, 99%% %.0f, max %.0f us; trips median %d max %d; us per trip median %.1f"
      % (n, info["rounds"], (t1.max() - base) / 1e5, np.median(dur), np.percentile(dur, 90), np.percentile(dur, 99), dur.max(),
         np.median(trips), trips.max(), np.median(dur[trips > 0] / trips[trips > 0])))
order = np.argsort(-dur)[:8]
st = st.cpu().numpy()
for i in order:
    print("   claim #%4d agent %6d: start %.2f ms, duration %.2f ms, %d trips (%.1f us per trip); evaluations of the whole solve %d, inner iterations %d"
          % (i, agent[i], (t0[i] - base) / 1e5, dur[i] / 1e3, trips[i], dur[i] / max(1, trips[i]), st[agent[i], 7], st[agent[i], 2]))
i = order[0]
print("the longest agent, per trip: load + step of the state machine %.1f us, rollout %.1f us, stage records %.1f us, adjoint %.1f us (of %.1f us)"
      % (t_adv[i] / 100.0 / trips[i], t_roll[i] / 100.0 / trips[i], t_recs[i] / 100.0 / trips[i], t_adj[i] / 100.0 / trips[i], dur[i] / trips[i]))
late = (t0 - base) / 100.0 > 50.0
print("agents claimed later than 50 us after the start: %d; the latest claim at %.2f ms; work claimed late: %.1f ms of %.1f ms of wave time"
      % (late.sum(), (t0.max() - base) / 1e5, dur[late].sum() / 1e3, dur.sum() / 1e3))
# with one wave per SIMD on 1024 SIMDs: the waves busy over time
end = (t1 - base) / 100.0; start = (t0 - base) / 100.0
ts = np.linspace(0, end.max(), 9)[1:-1]
print("waves busy at", " ".join("%.1f ms:%d" % (t / 1e3, ((start <= t) & (end > t)).sum()) for t in ts))
