"""Development script (not a pytest test): timing of the solve at benchmark size."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import model_predictive_control_amd as mp

def cl_straight(S=100):
    c = np.array([[i / 10 - 0.1, 0] for i in range(S)]); return c.ravel(order='F')
def batch(model, B, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.uniform(0, 5, B); y = rng.uniform(-.3, .3, B); phi = rng.uniform(-.3, .3, B); vx = rng.uniform(.3, 1.5, B)
    if model == 1:
        vy = rng.uniform(-.05, .05, B); om = rng.uniform(-.5, .5, B)
        return np.stack([x, y, phi, vx, vy, om], 1)
    return np.stack([x, y, phi, vx], 1)
dev = torch.device('cuda:0')
T = lambda a, dt=torch.float64: torch.tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
cl = cl_straight()
model = int(sys.argv[1]) if len(sys.argv) > 1 else 0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
B = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
prof = int(sys.argv[4]) if len(sys.argv) > 4 else 1
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 2
cfg = mp.default_config(model, N, max_total_inner=600)
eng = mp.BatchedMPC(cfg, dev)
eng.set_profile(bool(prof))
if len(sys.argv) > 6: eng.set_groups(int(sys.argv[6]))
X0 = T(batch(model, B)); U0 = T(np.tile([1., 0.], (B, N))); clt = T(cl)
for rep in range(reps):
    torch.cuda.synchronize(); t = time.time()
    Us, lam, st = eng.solve(X0, clt, U0)
    torch.cuda.synchronize(); dt = time.time() - t
    info = eng.last_solve_info()
    st = st.cpu().numpy()
    print('B', B, 'model', model, 'N', N, 'time %.4f s -> %.0f solves/s' % (dt, B / dt), info)
    print(' status', np.unique(st[:, 0], return_counts=True), 'iters mean %.1f max %.0f evals mean %.1f max %.0f' % (st[:, 2].mean(), st[:, 2].max(), st[:, 7].mean(), st[:, 7].max()))
# single eval timing
U = U0.clone()
for want_grad in (True, False):
    torch.cuda.synchronize(); t = time.time()
    for _ in range(10): eng.eval_cost_grad(X0, clt, U, want_grad=want_grad)
    torch.cuda.synchronize(); dt = (time.time() - t) / 10
    print(' eval grad=%s: %.3f ms per launch (incl pack/unpack)' % (want_grad, dt * 1e3))
