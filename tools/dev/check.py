"""Development script (not a pytest test): first parity check of the HIP path vs the oracle."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import model_predictive_control_amd as mp
from oracle import oracle as O

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
for model, N in ((0, 20), (1, 12), (1, 20)):
    B = 256
    cfg = mp.default_config(model, N, max_total_inner=600)
    ocfg = O.default_config(model, N, max_total_inner=600)
    eng = mp.BatchedMPC(cfg, dev)
    X0 = batch(model, B); rng = np.random.default_rng(1)
    U = np.tile([0.5, 0.0], (B, N)) + rng.uniform(-.3, .3, (B, 2 * N)) * np.tile([1, .3], N)
    # rhs
    dx = eng.rhs(T(X0), T(U[:, :2])).cpu().numpy()
    dxo = np.stack([O.rhs(ocfg, X0[b], U[b, :2]) for b in range(B)])
    print('model', model, 'N', N, 'rhs maxabs', np.abs(dx - dxo).max())
    Xr = eng.rollout(T(X0), T(U)).cpu().numpy()
    Xo = np.stack([O.rollout(ocfg, X0[b], U[b]) for b in range(B)])
    print(' rollout maxabs', np.abs(Xr - Xo).max())
    psi, g, _ = eng.eval_cost_grad(T(X0), T(cl), T(U))
    po, go = O.psi_batch(ocfg, X0, cl, U)
    print(' psi relerr', (np.abs(psi.cpu().numpy() - po) / np.abs(po)).max(), 'grad relerr', np.abs(g.cpu().numpy() - go).max() / np.abs(go).max())
    psi2, _, _ = eng.eval_cost_grad(T(X0), T(cl), T(U), want_grad=False)
    print(' psi(cost-only) vs psi(grad) maxabs', (psi2 - psi).abs().max().item())
    U0 = np.tile([1., 0.], (B, N))
    torch.cuda.synchronize(); t = time.time()
    Us, lam, st = eng.solve(T(X0), T(cl), T(U0))
    torch.cuda.synchronize(); dt = time.time() - t
    Uo, lo, sto = O.solve_batch(ocfg, X0, cl, U0)
    st = st.cpu().numpy(); Us = Us.cpu().numpy()
    print(' solve time %.3f s; info' % dt, eng.last_solve_info())
    print(' status gpu', np.unique(st[:, 0], return_counts=True), 'orc', np.unique(sto[:, 0], return_counts=True))
    print(' iters gpu mean %.1f orc mean %.1f; same-iters frac %.3f' % (st[:, 2].mean(), sto[:, 2].mean(), (st[:, 2] == sto[:, 2]).mean()))
    dU = np.abs(Us - Uo).max(1)
    print(' |dU| max %.3e median %.3e; frac<1e-6: %.3f' % (dU.max(), np.median(dU), (dU < 1e-6).mean()))
    print(' psi diff max', np.abs(st[:, 6] - sto[:, 6]).max())
