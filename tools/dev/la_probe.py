"""Development script: the persistent kernel's lookahead on the slowest Pacejka agents of bench.py's batch, one by one
(a lone agent = a lone wave: the chain the batch waits for), with and without it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench, model_predictive_control_amd as mp
dev = torch.device("cuda:0")
B, N = 65536, 12
X = bench.synthetic_states(1, 0, B)
T = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
cl = T(bench.straight_centerline())
eng = mp.BatchedMPC(mp.default_config(1, N), dev)
U, _, st = eng.solve(T(X), cl, T(np.tile([1.0, 0.0], (B, N))))
r = eng.debug_records(B)
ex = r["ngrad"] + r["ncost"] - (r["nspec"] - r["nspec_used"])
top = np.argsort(-ex)[:8]
print("slowest agents:", top, "requests", ex[top].astype(int), "lookahead hits", r["la_hits"][top].astype(int), "candidate evaluations", r["la_evals"][top].astype(int))
for a in top[:6]:
    x, u0 = T(X[a:a + 1]), T(np.tile([1.0, 0.0], (1, N)))
    out = []
    for env in ("", "1"):
        if env: os.environ["MPC_NO_LOOKAHEAD"] = "1"
        else: os.environ.pop("MPC_NO_LOOKAHEAD", None)
        e = mp.BatchedMPC(mp.default_config(1, N), dev)
        e.solve(x, cl, u0); torch.cuda.synchronize()
        t = time.perf_counter(); _, _, s1 = e.solve(x, cl, u0); torch.cuda.synchronize(); dt = time.perf_counter() - t
        rr = e.debug_records(1)
        req = rr["ngrad"][0] + rr["ncost"][0]
        out.append((dt * 1e3, req, rr["la_hits"][0], rr["la_evals"][0], s1[0, 7].item()))
    (t1, q1, h1, c1, n1), (t0, q0, h0, c0, n0) = out
    print("agent %5d: %7.1f ms with lookahead (%d requests, %d served from the cache -> %d trips, %.0f us per trip; %d candidates) | %7.1f ms without (%d trips, %.0f us per trip) | counted evaluations %d / %d"
          % (a, t1, q1, h1, q1 - h1, t1 * 1e3 / max(1, q1 - h1), c1, t0, q0, t0 * 1e3 / max(1, q0), n1, n0))
