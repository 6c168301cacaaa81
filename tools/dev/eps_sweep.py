"""Development script (not a pytest test): at which alm_eps do the GPU controls of bench.py's first 4 096
agents agree with the CPU oracle's within north_star's 1e-5 relative -- all of them -- and what does the
65 536-agent solve cost there?  (the value found is fixed in bench.py as PARITY_EPS.)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
import model_predictive_control_amd as mp
from oracle import oracle as O

dev = torch.device("cuda:0")
N, B, n_s = 20, 65536, 4096
cl_np = bench.straight_centerline()
X0 = torch.tensor(bench.synthetic_states(0, 0, B), dtype=torch.float64, device=dev)
cl = torch.tensor(cl_np, dtype=torch.float64, device=dev)
U0 = torch.tensor([1.0, 0.0], dtype=torch.float64, device=dev).repeat(B, N)
cores = min(O.lib().orc_max_threads(), len(os.sched_getaffinity(0)), 16)
rows = []
for eps in [float(a) for a in sys.argv[1:]] or [1e-6, 3e-7, 1e-7, 3e-8, 1e-8, 1e-9]:
    cfg = mp.default_config(0, N, alm_eps=eps)
    eng = mp.BatchedMPC(cfg, dev)
    eng.solve(X0, cl, U0)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(3):
        U, _, st = eng.solve(X0, cl, U0)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
    ocfg = O.default_config(0, N, alm_eps=eps)
    t = time.perf_counter()
    Uo, _, so = O.solve_batch(ocfg, X0[:n_s].cpu().numpy(), cl_np, np.tile([1.0, 0.0], (n_s, N)), nthreads=cores)
    dto = time.perf_counter() - t
    Ug, sg = U[:n_s].cpu().numpy(), st[:n_s].cpu().numpy()
    d = np.abs(Ug - Uo).max(1) / np.maximum(1.0, np.abs(Uo).max(1))
    row = {"alm_eps": eps, "ms_per_solve": dt * 1e3, "solves_per_s": B / dt, "converged": float((st[:, 0] == 1).double().mean()),
           "inner_mean": float(st[:, 2].mean()), "max_rel_dU": float(d.max()), "frac_le_1e-5": float((d <= 1e-5).mean()),
           "status_mismatch": int((sg[:, 0] != so[:, 0]).sum()), "oracle_s": dto, "rounds": eng.last_solve_info()["rounds"]}
    rows.append(row)
    print(json.dumps(row), flush=True)
    eng.close()
