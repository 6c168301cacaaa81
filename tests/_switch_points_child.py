"""Child process of test_switch_points_do_not_change_results (tests/test_gpu_parity.py): solves the given batch
sizes in a FRESH process -- GPU_MAX_HW_QUEUES is read by the HIP runtime once, when it initialises, so the
4-queue configuration cannot be had in the pytest process -- and prints one JSON line per size: SHA-256 of the
embedded agents' controls + statistics, the stream concurrency the library measured, the sub-batch groups used."""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import model_predictive_control_amd as mp
from switch_points_common import N, K_EMBED, batch, digest

dev = torch.device("cuda:0")
T = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
cl = T(np.array([[i / 10 - 0.1, 0] for i in range(100)]).ravel(order="F"))
eng = mp.BatchedMPC(mp.default_config(0, N), dev)
for B in [int(a) for a in sys.argv[1:]]:
    U, _, st = eng.solve(T(batch(B)), cl, T(np.tile([1.0, 0.0], (B, N))))
    streams, groups = eng.stream_concurrency()
    print(json.dumps({"B": B, "sha": digest(U, st), "streams": streams, "groups": groups,
                      "solo_agents": eng.last_solve_info()["solo_agents"], "rounds": eng.last_solve_info()["rounds"]}), flush=True)
