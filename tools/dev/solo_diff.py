"""Development script: where do the persistent kernel and the round path differ?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import model_predictive_control_amd as mp
import bench
dev = torch.device("cuda:0")
model, N, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cl = torch.tensor(bench.straight_centerline(), dtype=torch.float64, device=dev)
X0 = torch.tensor(bench.synthetic_states(model, 0, B), dtype=torch.float64, device=dev)
U0 = torch.tensor([1.0, 0.0], dtype=torch.float64, device=dev).repeat(B, N)
rng = np.random.default_rng(1)
Ur = torch.tensor(np.tile([0.5, 0.0], (B, N)) + rng.uniform(-.3, .3, (B, 2 * N)) * np.tile([1, .3], N), dtype=torch.float64, device=dev)
eng = mp.BatchedMPC(mp.default_config(model, N), dev)
for U in (U0, Ur):
    for wg in (True, False):
        p1, g1, _ = eng.eval_cost_grad(X0, cl, U, want_grad=wg)
        p2, g2, _ = eng.eval_cost_grad(X0, cl, U, want_grad=wg, wave=True)
        print("eval want_grad", wg, "psi equal", bool(torch.equal(p1, p2)), "max dpsi", float((p1 - p2).abs().max()),
              "grad equal", None if g1 is None else bool(torch.equal(g1, g2)), None if g1 is None else float((g1 - g2).abs().max()))
res = []
for th in (0, 100000):
    e = mp.BatchedMPC(mp.default_config(model, N), dev); e.set_solo_max(th)
    U, _, st = e.solve(X0, cl, U0)
    res.append((U.cpu().numpy(), st.cpu().numpy()))
(Ua, sa), (Ub, sb) = res
d = np.abs(Ua - Ub).max(1)
print("agents differing", int((d > 0).sum()), "of", B, "max |dU|", d.max(), "iters equal", int((sa[:, 2] == sb[:, 2]).sum()), "evals equal", int((sa[:, 7] == sb[:, 7]).sum()))
bad = np.where(d > 0)[0][:8]
for a in bad:
    print(" agent", a, "dU %.3e" % d[a], "iters", sa[a, 2], sb[a, 2], "evals", sa[a, 7], sb[a, 7], "psi", sa[a, 6], sb[a, 6], "x0", X0[a].cpu().numpy())
# short solves: after how many inner iterations does the first difference appear?
for mti in (1, 2, 3, 5, 8, 12, 20, 40):
    res = []
    for th in (0, 100000):
        e = mp.BatchedMPC(mp.default_config(model, N, max_total_inner=mti), dev); e.set_solo_max(th)
        U, _, st = e.solve(X0, cl, U0)
        res.append((U.cpu().numpy(), st.cpu().numpy()))
    (Ua, sa), (Ub, sb) = res
    d = np.abs(Ua - Ub).max(1)
    print("max_total_inner", mti, "agents differing", int((d > 0).sum()), "max |dU| %.3e" % d.max(), "evals equal", int((sa[:, 7] == sb[:, 7]).sum()), "psi equal", int((sa[:, 6] == sb[:, 6]).sum()))
