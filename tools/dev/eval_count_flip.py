#!/usr/bin/env python3
"""Development script (GPU box): a Pacejka agent whose HIP and oracle solves take the SAME decisions but count
different numbers of evaluations in one iteration (tools/dev/first_divergence.py, kind "evaluation count only").
Every point the oracle evaluates in that iteration is evaluated by the HIP K1 too (mpc_eval_cost_grad) and the two
values are printed side by side: where one is finite and the other is not, a descent-lemma loop inside a REJECTED
line-search trial doubles L in one implementation and not in the other -- evaluations, not iterates.
    python tools/dev/eval_count_flip.py <agent> <iteration> [model N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
import model_predictive_control_amd as mp
from oracle import oracle as O

agent, it = int(sys.argv[1]), int(sys.argv[2])
model = int(sys.argv[3]) if len(sys.argv) > 3 else 1
N = int(sys.argv[4]) if len(sys.argv) > 4 else 12
dev = torch.device("cuda:0")
cl = bench.straight_centerline()
X0 = bench.synthetic_states(model, 0, agent + 1)[agent]
U0 = np.tile([1.0, 0.0], N)
ocfg = O.default_config(model, N)
rows = O.solve_dump_evals(ocfg, X0, cl, U0, it)
T = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
eng = mp.BatchedMPC(mp.default_config(model, N), dev)
pts = rows[:, 2:]
psi, g, _ = eng.eval_cost_grad(T(np.tile(X0, (len(pts), 1))), T(cl), T(pts))
psi = psi.cpu().numpy()
X = eng.rollout(T(np.tile(X0, (len(pts), 1))), T(pts)).cpu().numpy()
print(f"agent {agent}, iteration {it}: {len(rows)} evaluations in the oracle")
for i, r in enumerate(rows):
    Xo = O.rollout(ocfg, X0, pts[i])
    print(f"  {i:3d} {'grad' if r[0] else 'cost'}  oracle psi {r[1]: .6e}  HIP psi {psi[i]: .6e}   max|u| {np.abs(pts[i]).max():.3g}  "
          f"max|x| oracle {np.nanmax(np.abs(Xo)) if np.isfinite(Xo).any() else float('nan'):.3g} (non-finite states: {int((~np.isfinite(Xo)).sum())}) "
          f"HIP {np.nanmax(np.abs(X[i])) if np.isfinite(X[i]).any() else float('nan'):.3g} (non-finite: {int((~np.isfinite(X[i])).sum())})")
g = g.cpu().numpy()
for i, r in enumerate(rows):
    if r[0]:
        po, go = O.psi(ocfg, X0, cl, pts[i])
        bad_h, bad_o = int((~np.isfinite(g[i])).sum()), int((~np.isfinite(go)).sum())
        print(f"  gradient at evaluation {i}: oracle non-finite {bad_o}, HIP non-finite {bad_h}, max |g| oracle {np.nanmax(np.abs(go)):.4e} "
              f"HIP {np.nanmax(np.abs(g[i])):.4e}, max |dg| {np.nanmax(np.abs(g[i] - go)):.2e}")
# the descent-lemma test of the second trial, oracle's numbers
L = float(os.environ.get("L", "32.83460799"))
gam = 0.95 / L
for i in (1, 3):
    x = pts[i]; po, go = O.psi(ocfg, X0, cl, x)
    lb = np.tile([-1.0, -0.32], N); ub = -lb
    for src, gg in (("oracle", go), ("HIP", g[i])):
        p = np.minimum(np.maximum(-gam * gg, lb - x), ub - x)
        xh = x + p
        pxh = O.psi(ocfg, X0, cl, xh, want_grad=False)[0]
        print(f"  trial at evaluation {i} ({src} gradient): psi(xhat) - psi = {pxh - po:.6e}, g'p + L/2 ||p||^2 = {gg @ p + 0.5 * L * (p @ p):.6e}  (g'p {gg @ p:.6e}, ||p||^2 {p @ p:.6e})")
