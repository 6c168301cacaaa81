"""Development script (product build: mpc_debug_records): what in an agent's record at round R0 says that it will be among
the last to finish?  A full solve gives the evaluations each agent executes; the same solve stopped at R0
(mpc_set_round_limit, persistent kernel off) leaves the records as they stand there.
    TRACE_MODEL=1 python tools/dev/predict_stragglers.py 50 100 150 200 250"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
import model_predictive_control_amd as mp

dev = torch.device("cuda:0")
MODEL = int(os.environ.get("TRACE_MODEL", 0))
N, B = int(os.environ.get("TRACE_N", 12 if MODEL else 20)), 65536
X0 = torch.tensor(bench.synthetic_states(MODEL, 0, B), dtype=torch.float64, device=dev)
cl = torch.tensor(bench.straight_centerline(), dtype=torch.float64, device=dev)
U0 = torch.tensor([1.0, 0.0], dtype=torch.float64, device=dev).repeat(B, N)
eng = mp.BatchedMPC(mp.default_config(MODEL, N), dev)
eng.set_solo_max(0)
_, _, st = eng.solve(X0, cl, U0)
st = st.cpu().numpy()
info = eng.last_solve_info()
fin = eng.debug_records(B)
ex_final = fin["ngrad"] + fin["ncost"] - (fin["nspec"] - fin["nspec_used"])     # useful evaluations executed (a round each)
print("rounds of the rounds-only solve:", info["rounds"], "; executed evaluations per agent: mean %.0f, 99 %% %.0f, 99.9 %% %.0f, top 10:"
      % (ex_final.mean(), np.percentile(ex_final, 99), np.percentile(ex_final, 99.9)), np.sort(ex_final)[-10:].astype(int))
print("agents with a failed inner solve:", int((st[:, 3] > 0).sum()), "; of the 64 with the most executed evaluations:",
      int((st[np.argsort(-ex_final)[:64], 3] > 0).sum()), "; counted evaluations of the top 5:", st[np.argsort(-ex_final)[:5], 7].astype(int),
      "inner failures:", st[np.argsort(-ex_final)[:5], 3].astype(int))
for thr in (300, 400, 500, 600, 800):
    print("   agents that execute more than %d evaluations: %d" % (thr, int((ex_final > thr).sum())))
for R0 in [int(a) for a in sys.argv[1:]] or [50, 100, 150, 200, 250]:
    eng.set_round_limit(R0)
    try:
        eng.solve(X0, cl, U0)
    except Exception:
        pass
    torch.cuda.synchronize()
    r = eng.debug_records(B)
    phase = r["phase"].astype(np.int64) & 63
    active = phase != 0
    done_ex = r["ngrad"] + r["ncost"] - (r["nspec"] - r["nspec_used"])
    remaining = ex_final - done_ex
    it = r["inner_tot"] + r["k"]
    print("round %d: %d agents still running; remaining executed evaluations among them: median %.0f, 99%% %.0f, max %.0f"
          % (R0, active.sum(), np.median(remaining[active]), np.percentile(remaining[active], 99), remaining[active].max()))
    top = np.argsort(-np.where(active, remaining, -1))[:64]
    feats = {"outer iteration (low)": -r["outer"], "inner tolerance now (high)": r["eps"], "inner iterations so far (high)": it,
             "evaluations per inner iteration so far (high)": r["nevals"] / np.maximum(1, it),
             "failed inner solves + retries (high)": r["inner_fail"] + r["initred"] + r["penred"],
             "stop measure ||p||/gamma now (high)": np.sqrt(np.maximum(r["pp"], 0)) / np.maximum(r["gamma"], 1e-300),
             "stop measure / tolerance (high)": np.sqrt(np.maximum(r["pp"], 0)) / np.maximum(r["gamma"], 1e-300) / r["eps"],
             "Lipschitz estimate (high)": r["L"], "cost psi (high)": r["psi"],
             "line-search step tau now (low)": -r["tau"]}
    for name, f in feats.items():
        f = np.where(active & np.isfinite(f), f, -np.inf)
        line = "   %-48s" % name
        for M in (64, 256, 1024):
            pick = np.argsort(-f, kind="stable")[:M]
            line += " | top %4d: %2d of 64, latest %s, latest-5 %d" % (M, len(set(pick) & set(top)), "Y" if top[0] in pick else "n", len(set(pick) & set(top[:5])))
        print(line)
