#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric: MPC solves/s, bicycle model N=20 nx=4 nu=2, batch 65536
per GPU (weak scaling over 1/2/4/8 GPUs), on synthetic batched road-following problems.

A "step" is ONE batched MPC solve (ALM + structured PANOC to eps = 1e-6, controller.py:27-48) of
the rank's shard, inputs already resident in HBM, followed by the final gather of the controls
(the only collective).  Contract: W untimed warm-up steps, EXACTLY K timed steps bracketed by
barrier + synchronize, MAX over ranks, one JSON line from rank 0.

    python bench.py                       # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

# hardware queues for the solver's sub-batch streams (read by the HIP runtime at its initialisation;
# the package sets the same default when it is imported first)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import model_predictive_control_amd as mp  # noqa: E402
from model_predictive_control_amd.sharding import gather_controls, shard_bounds  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_VALU_PEAK_TF = 78.6   # vendor fp64 vector peak
# The loosest ALM/PANOC tolerance at which ALL of the first 4 096 agents of this batch have controls within
# north_star's 1e-5 relative of the CPU oracle's (tests/dev_gpu_eps_sweep.py, profiles/r03_eps_sweep.txt: 1e-6 ->
# 87.5 % of the agents, max 3.1e-5; 3e-7 -> 100 %, max 7.8e-6; 1e-7 -> max 3.3e-6; 1e-8 -> max 2.7e-7.  At the
# reference's own 1e-6, controller.py:41, both solvers stop inside the same 1e-6 ball of a flat problem).  The `parity_at_1e-5` leg reports the throughput there; tests/test_gpu_parity.py asserts the 100 %.
PARITY_EPS = 3e-7


def straight_centerline(S=100):
    """main.py:13,:113."""
    return np.array([[i / 10 - 0.1, 0] for i in range(S)]).ravel(order="F")


def synthetic_states(model, lo, hi, seed=0):
    """SURVEY 8(d) synthetic initial states, a pure function of the GLOBAL agent index so that
    every sharding of the batch solves the same problems."""
    n = hi - lo
    out = np.empty((n, 6))
    # counter-based: one generator per block of 4096 agents, keyed by the block id
    blk = 4096
    for b0 in range(lo - lo % blk, hi, blk):
        rng = np.random.default_rng([seed, b0 // blk])
        chunk = np.stack([rng.uniform(0, 5, blk), rng.uniform(-.3, .3, blk), rng.uniform(-.3, .3, blk),
                          rng.uniform(.3, 1.5, blk), rng.uniform(-.05, .05, blk), rng.uniform(-.5, .5, blk)], 1)
        s0, s1 = max(lo, b0), min(hi, b0 + blk)
        out[s0 - lo:s1 - lo] = chunk[s0 - b0:s1 - b0]
    return out if model == mp.MODEL_PACEJKA else out[:, :4].copy()


def fp64_flops_per_solve(model, N, e_grad, e_cost):
    """F = (4 E_g + E_f) 16 N C_ode per solve -- E_g gradient evaluations (forward + adjoint = 4
    forward-rollout equivalents), E_f cost evaluations, 16 RHS evaluations per stage, C_ode flop-equivalents per
    RHS evaluation (an fp64 transcendental counted as 20: a convention, not a count).  The nearest-point
    search is not in it (the grid search looks at ~10 points per stage, < 1 % of a stage's work)."""
    c_ode = 250.0 if model == mp.MODEL_PACEJKA else 120.0
    return (4.0 * e_grad + e_cost) * 16.0 * N * c_ode, c_ode


def timed_solves(eng, X0, cl, U0, dev, steps, warmup=1):
    """`steps` blocking solves after `warmup`: (seconds per solve, controls, stats, info of the last)."""
    for _ in range(warmup):
        eng.solve(X0, cl, U0)
    torch.cuda.synchronize(dev)
    t = time.perf_counter()
    for _ in range(steps):
        U, _, st = eng.solve(X0, cl, U0)
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t) / steps, U, st, eng.last_solve_info()


def cpu_baseline(args, cfg_kw, cl, U_gpu=None, st_gpu=None):
    """The CPU oracle (same algorithm, OpenMP over agents) on a bounded sample of the same
    workload.  A reported baseline, not the optimisation target.  The controls it produces also
    certify the GPU result of the same agents (parity_sample), outside the timed region."""
    from oracle import oracle as O
    ocfg = O.default_config(args.model, args.horizon, **cfg_kw)
    # the GPU box gives one GPU's job a share of the host: stay inside it (16 threads at most)
    cores = min(O.lib().orc_max_threads(), len(os.sched_getaffinity(0)), int(os.environ.get("MPC_CPU_THREADS", "16")))
    n = args.cpu_sample
    X0 = synthetic_states(args.model, 0, n)
    U0 = np.tile([1., 0.], (n, args.horizon))
    O.solve_batch(ocfg, X0[:64], cl, U0[:64], nthreads=cores)   # warm-up (page-in, thread pool)
    t = time.perf_counter()
    Uo, _, sto = O.solve_batch(ocfg, X0, cl, U0, nthreads=cores)
    dt = time.perf_counter() - t
    base = {"value": n / dt, "unit": "solves/s", "cores": int(cores), "kind": "port",
            "sample": f"first {n} agents of the same synthetic batch, {dt:.1f} s, "
                      f"oracle/mpc_oracle.c (-O3, OpenMP), {int((sto[:, 0] == 1).sum())}/{n} converged; "
                      "substitutes for alpaqa+CasADi, which are not installable offline"}
    parity = None
    if U_gpu is not None:
        Ug, sg = U_gpu[:n], st_gpu[:n]
        scale = np.maximum(1.0, np.abs(Uo).max(1))
        d = np.abs(Ug - Uo).max(1) / scale
        parity = {"agents": int(n), "oracle": "oracle/mpc_oracle.c (parity unpinned for the solver layer: DESIGN.md 3)",
                  "status_mismatches": int((sg[:, 0] != sto[:, 0]).sum()),
                  "max_abs_dpsi": float(np.abs(sg[:, 6] - sto[:, 6]).max()),
                  "max_rel_dU": float(d.max()), "median_rel_dU": float(np.median(d)),
                  "frac_dU_le_1e-5": float((d <= 1e-5).mean()), "frac_dU_le_2e-4": float((d <= 2e-4).mean()),
                  "outer_iterations_equal_frac": float((sg[:, 1] == sto[:, 1]).mean()),
                  "inner_iterations_mean": [float(sg[:, 2].mean()), float(sto[:, 2].mean())],
                  "tolerance": "eps = %g: both stop inside the same eps-ball of a flat problem, ||dU|| ~ eps/mu "
                               "(1e-5 relative is met with both converged to 1e-10: tests/test_gpu_parity.py)" % ocfg.alm_eps}
    return base, parity


def pmc_profile():
    """HBM traffic per kernel launch from the committed rocprofv3 counter passes (tools_profile.sh ->
    tools_pmc_summary.py): the newest profiles/r*_pmc_summary.json, else profiles/pmc_summary.json."""
    import glob
    # (tags run r03a .. r03z, r03aa ..: shorter names first, then alphabetically)
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")), key=lambda f: (len(os.path.basename(f)), f))
    path = cands[-1] if cands else os.path.join(ROOT, "profiles", "pmc_summary.json")
    if not os.path.exists(path):
        return None, None
    try:
        return json.load(open(path)), os.path.relpath(path, ROOT)
    except Exception:
        return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=65536, help="agents per GPU (weak scaling)")
    ap.add_argument("--horizon", type=int, default=20)
    ap.add_argument("--model", type=int, default=mp.MODEL_KINEMATIC, help="0 kinematic nx=4, 1 Pacejka nx=6")
    ap.add_argument("--max-total-inner", type=int, default=0,
                    help="inner-iteration budget per solve (stands in for controller.py:30,:44 wall-clock caps); "
                         "0 = the library default (5000)")
    ap.add_argument("--max-total-evals", type=int, default=0, help="evaluation budget per solve (0 = none)")
    ap.add_argument("--lbfgs-memory", type=int, default=0, help="L-BFGS memory (0 = the reference's: N_horiz, controller.py:36)")
    ap.add_argument("--cpu-sample", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline-pass", action="store_true",
                    help="skip the untimed extra pass that solves consecutive batches on two handles side by side")
    ap.add_argument("--no-parity-leg", action="store_true",
                    help="skip the extra leg at the tolerance where the controls are within 1e-5 of the CPU oracle's")
    ap.add_argument("--no-secondary", action="store_true", help="skip the Pacejka nx=6 N=12 secondary measurement")
    ap.add_argument("--no-kernel-pass", action="store_true",
                    help="skip the untimed single-group pass that measures per-kernel durations")
    ap.add_argument("--profile-timed", action="store_true", help="HIP-event sampling inside the timed steps too")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--same-gpu", action="store_true", help="rehearsal: every rank uses device 0")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    assert torch.cuda.is_available(), "bench.py needs HIP devices (there is no CPU path)"
    if args.same_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    if world > 1:
        from model_predictive_control_amd.sharding import gather_supported
        gather_supported(None, dev)     # the one-off collective capability probe, outside every timed region
    B_total = args.batch * world
    lo, hi = shard_bounds(B_total, rank, world)
    B = hi - lo
    N = args.horizon
    cfg_kw = {}
    if args.max_total_inner > 0:
        cfg_kw["max_total_inner"] = args.max_total_inner
    if args.max_total_evals > 0:
        cfg_kw["max_total_evals"] = args.max_total_evals
    if args.lbfgs_memory > 0:
        cfg_kw["lbfgs_memory"] = args.lbfgs_memory
    cfg = mp.default_config(args.model, N, **cfg_kw)
    eng = mp.BatchedMPC(cfg, dev)
    eng.set_profile(bool(args.profile_timed))
    cl_np = straight_centerline()
    X0 = torch.tensor(synthetic_states(args.model, lo, hi), dtype=torch.float64, device=dev)
    cl = torch.tensor(cl_np, dtype=torch.float64, device=dev)
    U0 = torch.tensor([1.0, 0.0], dtype=torch.float64, device=dev).repeat(B, N)   # controller.py:20

    def step():
        U, _, st = eng.solve(X0, cl, U0)
        full = gather_controls(U, B_total, dst=0)
        return U, st, full

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    # the timed region: wall clock between the fences (the contract's number) and, on the stream the solves
    # are launched on (torch's current stream: the sub-batch streams fork from it and join it), HIP events
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    infos = []
    for _ in range(args.steps):
        U, st, full = step()
        infos.append(eng.last_solve_info())
    ev1.record()
    fence()
    dt = time.perf_counter() - t0
    ev_ms_per_step = ev0.elapsed_time(ev1) / args.steps
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.item())

    conv = float((st[:, 0] == 1).double().mean().item())
    it_mean = float(st[:, 2].mean().item()); it_max = float(st[:, 2].max().item())
    ev_mean = float(st[:, 7].mean().item()); ev_max = float(st[:, 7].max().item())

    # ---- untimed: per-kernel durations WITHOUT stream overlap (one sub-batch group, HIP events on the
    # solve's stream around sampled launch sets), the inputs of the per-kernel roofline figures
    kinfo = None
    if rank == 0 and not args.no_kernel_pass:
        eng.set_groups(1); eng.set_profile(True)
        eng.solve(X0, cl, U0)
        kinfo = eng.last_solve_info()
        eng.set_groups(0); eng.set_profile(bool(args.profile_timed))
        torch.cuda.synchronize(dev)

    # ---- untimed: consecutive batches pipelined over TWO handles (mpc_solve_batch_async): what a caller
    # with a stream of batches gets -- the tail of one batch (a few waves in the persistent kernel) and its
    # idle issue slots are filled by the next.  Reported beside `value`, never as it: a timed step above is
    # one blocking solve of one batch.
    pipe = None
    if rank == 0 and not args.no_pipeline_pass:
        eng.set_profile(False)
        eng2 = mp.BatchedMPC(cfg, dev)
        side = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
        engs = [eng, eng2]
        with torch.cuda.stream(side[1]):
            eng2.solve(X0, cl, U0)
        torch.cuda.synchronize(dev)
        Kp = max(4, min(args.steps, 8))
        pend, outs = [None, None], []
        tp = time.perf_counter()
        for k in range(Kp):
            i = k & 1
            if pend[i] is not None:
                outs.append(pend[i]())
            with torch.cuda.stream(side[i]):
                pend[i] = engs[i].solve_async(X0, cl, U0)
        for i in (Kp & 1, (Kp + 1) & 1):
            if pend[i] is not None:
                outs.append(pend[i]())
        torch.cuda.synchronize(dev)
        tp = time.perf_counter() - tp
        pipe = {"value": B * Kp / tp, "unit": "solves/s", "ms_per_batch": tp / Kp * 1e3, "handles": 2, "batches": Kp,
                "same_controls_as_timed_steps": bool(all(torch.equal(o[0], U) for o in outs)),
                "note": "untimed extra pass on rank 0: %d consecutive batches of %d agents, two handles, each batch "
                        "one mpc_solve_batch_async on its own stream; `value` above is NOT this figure" % (Kp, B)}
        del eng2
        eng.set_profile(bool(args.profile_timed))

    if rank == 0:
        import hashlib
        K = args.steps
        nx, n, m = eng.nx, eng.n, eng.m
        rounds = float(np.mean([i["rounds"] for i in infos]))
        eg = float(np.mean([i["evals_grad"] for i in infos])); ec = float(np.mean([i["evals_cost"] for i in infos]))
        lb_rows = float(np.mean([i["lbfgs_rows"] for i in infos]))
        spec_i = float(np.mean([i["spec_issued"] for i in infos])); spec_u = float(np.mean([i["spec_used"] for i in infos]))
        step_s = dt / K
        # (1) SURVEY 8(d): algorithmic bytes per solve = 8 (nx + 2 nu N + 2 m_c + 4): read x0, read+write U,
        # read+write lambda, 4 stat words; the shared centerline amortises to 0
        alg_per_solve = 8 * (nx + 2 * 2 * N + 2 * m + 4)
        alg_bytes_step = alg_per_solve * B
        alg_gbps = alg_bytes_step / (ev_ms_per_step * 1e-3) / 1e9      # HIP-event duration of a step on the solve's stream
        # (2) moved bytes: what the implementation sends through HBM per solve, from the committed PMC
        # passes (FETCH_SIZE doubled per the gfx950 note, + WRITE_SIZE), per launch x launches per solve
        pmc, pmc_src = pmc_profile()
        workload = ("configs[1] shape at the metric's batch: %d agents/GPU, %s bicycle nx=%d nu=2, "
                    "N=%d, box input constraints, straight S=100 centerline, ALM+PANOC eps=1e-6"
                    % (args.batch, "Pacejka" if args.model else "kinematic", nx, N))
        if pmc and pmc.get("workload") != workload:
            pmc, pmc_src = None, None          # the committed counters describe another workload
        moved = None
        if pmc and pmc.get("hbm_bytes_per_solve"):
            mb = float(pmc["hbm_bytes_per_solve"])
            moved = {"bytes_per_step": mb, "GBps": mb / step_s / 1e9, "frac": mb / step_s / 1e9 / HBM_PEAK_GBS,
                     "frac_of_achievable_6300GBps": mb / step_s / 1e9 / 6300.0,
                     "x_algorithmic": mb / alg_bytes_step, "source": pmc_src,
                     "note": "bytes from the profiled run, time from this run"}
        # (3) per-kernel model on the single-group pass: bytes any implementation with this
        # round decomposition moves per launch / that kernel's average launch duration without overlap
        kernels, dominant = {}, None
        if kinfo:
            kms = kinfo["kernel_ms"]
            launches = kinfo["launches"]
            per = kinfo["evals_grad"] + kinfo["evals_cost"]
            egk = kinfo["evals_grad"]
            agent_steps = per - kinfo["spec_issued"]
            jac = nx * (nx + 1) + 2
            model = {
                "step": 8 * (2 * 64 + 6 * n) * agent_steps + kinfo["lbfgs_rows"] * 2 * n * 8,
                "rollout": 8 * (nx + n + (N + 1) * nx) * per,
                "stage": 8 * ((2 * nx + 2 + 1) * N * per + jac * N * egk),
                "adjoint": 8 * ((N + 1) * per + (jac * N + n) * egk),
                "solo": 8 * (nx + 2 * n + 2 * 64) * kinfo.get("solo_agents", 0),
            }
            tot_ms = sum(kms.values())
            for k, ms in kms.items():
                nl = max(1, launches.get(k, 0))
                if ms <= 0 and launches.get(k, 0) == 0:
                    continue
                ent = {"ms_per_solve": ms, "launches": launches.get(k, 0), "avg_launch_ms": ms / nl,
                       "share_of_kernel_time": ms / tot_ms if tot_ms > 0 else None,
                       "model_bytes_per_launch": model.get(k, 0) / nl,
                       "model_GBps": model.get(k, 0) / (ms * 1e-3) / 1e9 if ms > 0 else None}
                if pmc and (k + "_kernel") in pmc:
                    ent["pmc_hbm_bytes_per_launch"] = pmc[k + "_kernel"].get("hbm_bytes_per_launch_corrected")
                    ent["pmc_source"] = pmc_src
                kernels[k + "_kernel"] = ent
            dominant = max(kms, key=kms.get)
        # fp64 work, SURVEY 8(d): F = (4 E_g + E_f) 16 N C_ode with the measured evaluation counts (executed
        # evaluations of rank 0's shard, the speculative ones included) against the wall time of a timed step
        # (all kernels, overlap included) -- and, on the single-group pass, against the K1 kernels' own time
        f_step, c_ode = fp64_flops_per_solve(args.model, N, eg / B, ec / B)
        fp64 = {"achieved": f_step * B_total / step_s / 1e12, "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s",
                "what": "whole timed step: flops per solve x solves per step / wall time of a step",
                "flops_per_solve": f_step, "E_g": eg / B, "E_f": ec / B, "C_ode": c_ode,
                "flop_model": "SURVEY 8(d): F = (4 E_g + E_f) 16 N C_ode; E_g, E_f = executed gradient / cost "
                              "evaluations per solve (measured); a transcendental counts 20 (convention)"}
        fp64["frac"] = fp64["achieved"] / (FP64_VALU_PEAK_TF * world)
        if kinfo:
            kms = kinfo["kernel_ms"]
            per = kinfo["evals_grad"] + kinfo["evals_cost"]
            f_pass, _ = fp64_flops_per_solve(args.model, N, kinfo["evals_grad"] / B, kinfo["evals_cost"] / B)
            k1_ms = kms.get("rollout", 0) + kms.get("stage", 0) + kms.get("adjoint", 0)
            if k1_ms > 0:
                k1 = f_pass * B / (k1_ms * 1e-3) / 1e12
                fp64["k1_only"] = {"achieved": k1, "frac": k1 / FP64_VALU_PEAK_TF, "ms_per_solve": k1_ms,
                                   "kernels": "K1a+K1b+K1c on the single-group pass (their own time, no overlap "
                                              "with the step kernel), same F"}
        dk = kernels.get(dominant + "_kernel") if dominant else None
        out = {
            "metric": "MPC solves/sec, bicycle model N=20 nx=4 nu=2, batch=65536; 1/2/4/8 GPU",
            "value": B_total * K / dt, "unit": "solves/s", "n_gpus": world, "steps": K,
            "warmup": args.warmup, "ms_per_step": step_s * 1e3, "hip_event_ms_per_step": ev_ms_per_step,
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload,
                       "batch_per_gpu": args.batch, "horizon": N, "nx": nx, "nu": 2, "m_c": m,
                       "lbfgs_memory": int(cfg.lbfgs_memory), "tolerance": cfg.alm_eps,
                       "max_total_inner": int(cfg.max_total_inner), "max_total_evals": int(cfg.max_total_evals),
                       "hw_queues_env": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
                       "streams_side_by_side_measured": eng.stream_concurrency()[0],
                       "sub_batch_groups": int(infos[-1].get("groups", 0)),
                       "parallelism": f"agents sharded x{world}, no collective in the solve, final gather to rank 0"},
            "solver": {"converged_frac": conv, "inner_iters_mean": it_mean, "inner_iters_max": it_max,
                       "evals_per_solve_mean": ev_mean, "evals_per_solve_max": ev_max, "rounds": rounds,
                       "solo_agents": float(np.mean([i.get("solo_agents", 0) for i in infos])),
                       "speculative_gradients": {"issued": spec_i, "used": spec_u}},
            "roofline": {
                "bound": "hbm", "kernel": (dominant + "_kernel") if dominant else None,
                "achieved": alg_gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg_gbps / HBM_PEAK_GBS,
                "traffic": moved["bytes_per_step"] if moved else None, "traffic_source": pmc_src if moved else None,
                "definition": "SURVEY 8(d): algorithmic bytes per solve = 8 (nx + 2 nu N + 2 m_c + 4) = %d; achieved = "
                              "bytes per solve x solves per step / duration of a step (one batched solve = one pass of "
                              "the hot path; HIP events on the solve's stream over the timed region); traffic = HBM bytes "
                              "per step from the PMC passes of the named profile" % alg_per_solve,
                "algorithmic": {"bytes_per_solve": alg_per_solve, "bytes_per_step": alg_bytes_step, "GBps": alg_gbps,
                                "frac": alg_gbps / HBM_PEAK_GBS},
                "moved": moved,
                "dominant_kernel": ({"name": dominant + "_kernel", **dk,
                                     "model_frac_of_peak": (dk["model_GBps"] or 0) / HBM_PEAK_GBS,
                                     "timing": "HIP events on the solve's stream, one sub-batch group (no overlap "
                                               "with other launches), untimed extra solve"} if dk else None),
                "fp64_valu": fp64, "flops_fraction": fp64["frac"] if fp64 else None,
                "note": "the solve is fp64-issue / latency bound, not HBM bound and not MFMA (SURVEY 8d, DESIGN.md 5)"},
            "kernels": kernels,
            "controls_sha256_first_65536": hashlib.sha256(
                np.ascontiguousarray(full[:65536].cpu().numpy()).tobytes()).hexdigest(),
        }
        if pipe:
            out["pipelined_two_handles"] = pipe
        if args.profile_timed:
            out["kernels_overlapped_ms_per_step"] = {
                k: float(np.mean([i["kernel_ms"][k] for i in infos])) for k in infos[0]["kernel_ms"]}
        if not args.no_cpu_baseline and world == 1:
            base, parity = cpu_baseline(args, cfg_kw, cl_np, U.cpu().numpy(), st.cpu().numpy())
            out["cpu_baseline"] = base
            out["parity_sample"] = parity
        # ---- reported beside the headline, never as it: the SAME batch at the loosest tolerance where north_star's
        # parity bar (controls within 1e-5 relative of the CPU path) holds for every agent of the sample
        if not args.no_parity_leg and world == 1 and args.model == mp.MODEL_KINEMATIC:
            eng.set_profile(False)
            kw = dict(cfg_kw, alm_eps=PARITY_EPS)
            eng_p = mp.BatchedMPC(mp.default_config(args.model, N, **kw), dev)
            sec, Up, stp, infp = timed_solves(eng_p, X0, cl, U0, dev, steps=3)
            leg = {"alm_eps": PARITY_EPS, "value": B / sec, "unit": "solves/s", "ms_per_step": sec * 1e3, "steps": 3,
                   "converged_frac": float((stp[:, 0] == 1).double().mean().item()),
                   "inner_iters_mean": float(stp[:, 2].mean().item()), "rounds": infp["rounds"],
                   "note": "same 65 536-agent batch and kernels as `value`, tolerance alm_eps tightened from the "
                           "reference's 1e-6 (controller.py:41) to the loosest value at which every agent of the "
                           "4 096-agent sample is within 1e-5 relative of the CPU oracle (tests/dev_gpu_eps_sweep.py)"}
            if not args.no_cpu_baseline:
                pargs = argparse.Namespace(**vars(args))
                _, par = cpu_baseline(pargs, kw, cl_np, Up.cpu().numpy(), stp.cpu().numpy())
                leg["parity_sample"] = par
                leg["frac_dU_le_1e-5"] = par["frac_dU_le_1e-5"]
            out["parity_at_1e-5"] = leg
            del eng_p
        # ---- the reference's own model (car_dynamics.py:93-129, nx = 6; main.py:67-68 N = 12) at the same batch
        if not args.no_secondary and world == 1 and args.model == mp.MODEL_KINEMATIC:
            Np = 12
            eng_s = mp.BatchedMPC(mp.default_config(mp.MODEL_PACEJKA, Np), dev)
            Xs = torch.tensor(synthetic_states(mp.MODEL_PACEJKA, lo, hi), dtype=torch.float64, device=dev)
            Us0 = torch.tensor([1.0, 0.0], dtype=torch.float64, device=dev).repeat(B, Np)
            sec, Us, sts, infs = timed_solves(eng_s, Xs, cl, Us0, dev, steps=2)
            eng_s.set_profile(True)                      # one more solve with HIP events around the kernels
            eng_s.solve(Xs, cl, Us0)
            infk = eng_s.last_solve_info()
            out["secondary"] = {"pacejka_nx6_N12": {
                "workload": "%d agents/GPU, Pacejka bicycle nx=6 nu=2 (car_dynamics.py:93-129), N=12 (main.py:68), box input "
                            "constraints, straight S=100 centerline, ALM+PANOC eps=1e-6, no evaluation budget" % B,
                "value": B / sec, "unit": "solves/s", "ms_per_step": sec * 1e3, "steps": 2, "blocking": True,
                "converged_frac": float((sts[:, 0] == 1).double().mean().item()),
                "inner_iters_mean": float(sts[:, 2].mean().item()), "evals_per_solve_mean": float(sts[:, 7].mean().item()),
                "evals_per_solve_max": float(sts[:, 7].max().item()), "rounds": infs["rounds"],
                "solo_agents": infs["solo_agents"],
                "solo_kernel_ms_longest": infk["solo_longest_ms"], "solo_kernel_share": infk["solo_longest_ms"] / (sec * 1e3),
                "solo_kernel_ms_sum_over_groups": infk["kernel_ms"]["solo"],
                "controls_sha256": hashlib.sha256(np.ascontiguousarray(Us.cpu().numpy()).tobytes()).hexdigest()}}
            del eng_s
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
