"""Development script: many consecutive solves on one engine (and a second engine created and destroyed meanwhile): the same
bits every time, no growth of device memory, no error -- kinematic headline batch and the Pacejka batch."""
import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench, model_predictive_control_amd as mp
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
cl = torch.tensor(bench.straight_centerline(), dtype=torch.float64, device=dev)
for model, N, B in ((0, 20, 65536), (1, 12, 65536), (0, 20, 3000), (1, 12, 700)):
    X = torch.tensor(bench.synthetic_states(model, 0, B), dtype=torch.float64, device=dev)
    U0 = torch.tensor([1.0, 0.0], dtype=torch.float64, device=dev).repeat(B, N)
    eng = mp.BatchedMPC(mp.default_config(model, N), dev)
    ref, mem0, t0 = None, None, time.perf_counter()
    n = reps if B > 4096 else 3 * reps
    for i in range(n):
        U, _, st = eng.solve(X, cl, U0)
        if i % 10 == 0:
            h = hashlib.sha256(np.ascontiguousarray(U.cpu().numpy()).tobytes() + np.ascontiguousarray(st.cpu().numpy()).tobytes()).hexdigest()
            ref = ref or h
            assert h == ref, (model, i)
            other = mp.BatchedMPC(mp.default_config(model, N), dev); other.solve(X[:256], cl, U0[:256]); other.close()
        if i == 5:
            mem0 = torch.cuda.mem_get_info(dev)[0]
    torch.cuda.synchronize()
    mem1 = torch.cuda.mem_get_info(dev)[0]
    print("model %d N %d B %6d: %4d solves, %.1f ms each, same bits %s, free memory change since solve 5: %+.1f MB"
          % (model, N, B, n, (time.perf_counter() - t0) / n * 1e3, ref[:12], (mem1 - mem0) / 1e6), flush=True)
    assert abs(mem1 - mem0) < 64e6
    eng.close()
print("soak ok")
