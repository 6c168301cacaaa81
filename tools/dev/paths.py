"""Development script (not a pytest test): how many agents take IDENTICAL solver paths (same status, same inner
iterations, same evaluation count) in the HIP solver and in the CPU oracle, per model, with and without an
evaluation budget -- the numbers behind the floors asserted in tests/test_gpu_parity.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from conftest import synthetic_states, straight_centerline
import model_predictive_control_amd as mp
from oracle import oracle as O

O.build()
dev = torch.device("cuda:0")
T = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
cl = straight_centerline()
for model, N, B, budget in ((1, 12, 512, 400), (0, 20, 512, 300), (1, 12, 512, 0), (0, 20, 512, 0)):
    kw = dict(max_total_evals=budget) if budget else {}
    X0 = synthetic_states(model, B, seed=3)
    U0 = np.tile([1., 0.], (B, N))
    U, _, st = mp.BatchedMPC(mp.default_config(model, N, **kw), dev).solve(T(X0), T(cl), T(U0))
    st = st.cpu().numpy()
    _, _, so = O.solve_batch(O.default_config(model, N, **kw), X0, cl, U0)
    same = (st[:, 2] == so[:, 2]) & (st[:, 0] == so[:, 0])
    print("model", model, "N", N, "budget", budget, "status equal %.3f" % np.mean(st[:, 0] == so[:, 0]),
          "identical (status, inner iterations) %.3f" % same.mean(),
          "of those, same evaluation count %.3f" % np.mean(st[same, 7] == so[same, 7]),
          "MaxTime hip/oracle", int((st[:, 0] == 2).sum()), int((so[:, 0] == 2).sum()), flush=True)
