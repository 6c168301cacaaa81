"""Development script: consecutive batched solves pipelined over two handles (mpc_solve_batch_async) --
the tail of one batch (a few waves in the persistent kernel) overlaps the bulk of the next."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import model_predictive_control_amd as mp
import bench
dev = torch.device("cuda:0")
N, B, K = 20, 65536, 12
cl = torch.tensor(bench.straight_centerline(), dtype=torch.float64, device=dev)
X0 = torch.tensor(bench.synthetic_states(0, 0, B), dtype=torch.float64, device=dev)
U0 = torch.tensor([1.0, 0.0], dtype=torch.float64, device=dev).repeat(B, N)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
engs = []
for s in streams:
    with torch.cuda.stream(s):
        engs.append(mp.BatchedMPC(mp.default_config(0, N), dev))
ref, _, _ = engs[0].solve(X0, cl, U0); engs[1].solve(X0, cl, U0)
torch.cuda.synchronize(); t = time.perf_counter()
for k in range(K): engs[0].solve(X0, cl, U0)
torch.cuda.synchronize(); seq = (time.perf_counter() - t) / K
print("sequential: %.2f ms per batch -> %.0f solves/s" % (seq * 1e3, B / seq), flush=True)
torch.cuda.synchronize(); t = time.perf_counter()
pend = [None, None]
outs = []
for k in range(K):
    i = k & 1
    if pend[i] is not None: outs.append(pend[i]())
    with torch.cuda.stream(streams[i]):
        pend[i] = engs[i].solve_async(X0, cl, U0)
for i in ((K & 1), ((K + 1) & 1)):
    if pend[i] is not None: outs.append(pend[i]())
torch.cuda.synchronize(); pip = (time.perf_counter() - t) / K
print("two handles pipelined: %.2f ms per batch -> %.0f solves/s" % (pip * 1e3, B / pip), flush=True)
print("same bits:", all(torch.equal(o[0], ref) for o in outs), len(outs))
