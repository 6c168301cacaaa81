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
# The loosest ALM/PANOC tolerance at which ALL sampled agents of this batch have controls within north_star's 1e-5
# relative of the CPU oracle's.  Chosen on the first 4 096 agents (tools/dev/eps_sweep.py, profiles/r03_eps_sweep.txt:
# 1e-6 -> 87.5 % of the agents, max 3.1e-5; 3e-7 -> 100 %, max 7.8e-6; 1e-7 -> max 3.3e-6; 1e-8 -> max 2.7e-7.  At the
# reference's own 1e-6, controller.py:41, both solvers stop inside the same 1e-6 ball of a flat problem) and CERTIFIED on
# samples that are not the one it was chosen on: every 8th agent of the whole batch -- all 16 generator blocks -- in the
# `parity_at_1e-5` leg below, agents 5, 21, 37, ... in tests/test_gpu_parity.py.
PARITY_EPS = 3e-7
# The controls metric of every parity figure in this file and in the tests (stated in the JSON as `dU_metric`)
DU_METRIC = ("per agent: max_j |U_hip[j] - U_oracle[j]| / max(1, max_j |U_oracle[j]|) -- vector-relative with a floor of 1 "
             "on the scale; the drive input saturates at |d| = 1, so this is an ABSOLUTE 1e-5 on every component, the "
             "steering angle (|delta| <= 0.32) included")


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


def rel_dU(Ug, Uo):
    """DU_METRIC."""
    return np.abs(Ug - Uo).max(1) / np.maximum(1.0, np.abs(Uo).max(1))


def cpu_baseline(model, N, cfg_kw, cl, U_gpu=None, st_gpu=None, agents=None, n=4096, what="first %d agents"):
    """The CPU oracle (same algorithm, OpenMP over agents) on a bounded sample of the same
    workload.  A reported baseline, not the optimisation target.  The controls it produces also
    certify the GPU result of the same agents (parity_sample), outside the timed region.
    `agents`: global agent indices of the sample (default: the first n)."""
    from oracle import oracle as O
    ocfg = O.default_config(model, N, **cfg_kw)
    # the GPU box gives one GPU's job a share of the host: stay inside it (16 threads at most)
    cores = min(O.lib().orc_max_threads(), len(os.sched_getaffinity(0)), int(os.environ.get("MPC_CPU_THREADS", "16")))
    if agents is None:
        agents = np.arange(n)
        what = what % n
    n = len(agents)
    span = int(agents.max()) + 1
    X0 = synthetic_states(model, 0, span)[agents]
    U0 = np.tile([1., 0.], (n, N))
    O.solve_batch(ocfg, X0[:64], cl, U0[:64], nthreads=cores)   # warm-up (page-in, thread pool)
    t = time.perf_counter()
    Uo, _, sto = O.solve_batch(ocfg, X0, cl, U0, nthreads=cores)
    dt = time.perf_counter() - t
    base = {"value": n / dt, "unit": "solves/s", "cores": int(cores), "kind": "port",
            "sample": f"{what} of the same synthetic batch ({n} agents), {dt:.1f} s, "
                      f"oracle/mpc_oracle.c (-O3, OpenMP), {int((sto[:, 0] == 1).sum())}/{n} converged; "
                      "substitutes for alpaqa+CasADi, which are not installable offline"}
    parity = None
    if U_gpu is not None:
        Ug, sg = U_gpu[agents], st_gpu[agents]
        d = rel_dU(Ug, Uo)
        parity = {"agents": int(n), "which": what,
                  "oracle": "oracle/mpc_oracle.c (parity unpinned for the solver layer: DESIGN.md 3)",
                  "dU_metric": DU_METRIC,
                  "status_mismatches": int((sg[:, 0] != sto[:, 0]).sum()),
                  "max_abs_dpsi": float(np.abs(sg[:, 6] - sto[:, 6]).max()),
                  "max_rel_dU": float(d.max()), "median_rel_dU": float(np.median(d)),
                  "frac_dU_le_1e-5": float((d <= 1e-5).mean()), "frac_dU_le_2e-4": float((d <= 2e-4).mean()),
                  "outer_iterations_equal_frac": float((sg[:, 1] == sto[:, 1]).mean()),
                  "identical_path_frac": float(((sg[:, 0] == sto[:, 0]) & (sg[:, 2] == sto[:, 2])).mean()),
                  "inner_iterations_mean": [float(sg[:, 2].mean()), float(sto[:, 2].mean())],
                  "tolerance": "eps = %g: both stop inside the same eps-ball of a flat problem, ||dU|| ~ eps/mu "
                               "(1e-5 relative is met with both converged to 1e-10: tests/test_gpu_parity.py)" % ocfg.alm_eps}
    return base, parity


def pmc_profile():
    """HBM traffic per kernel from the committed rocprofv3 counter passes (tools/profile.sh -> tools/pmc_summary.py):
    the profiles/r*_pmc_summary.json whose `library_source_sha256` is the RUNNING library's (mpc_source_hash), or
    nothing -- counters of another build are not this run's traffic (`traffic: null` rather than a stale figure)."""
    import glob
    from model_predictive_control_amd import _lib
    mine = _lib.library_hash()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")), key=os.path.getmtime, reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("library_source_sha256") == mine and mine != "unknown":
            return d, os.path.relpath(path, ROOT)
    return None, None


def fp64_roofline(model, N, B, info, sec_per_solve, world=1):
    """SURVEY 8(d)'s flop model against the wall time of a solve, executed and useful (consumed evaluations only:
    speculative gradients that the next iteration did not take are work done for nothing)."""
    eg, ec = info["evals_grad"] / B, info["evals_cost"] / B
    wasted = (info["spec_issued"] - info["spec_used"]) / B
    f_exec, c_ode = fp64_flops_per_solve(model, N, eg, ec)
    f_use, _ = fp64_flops_per_solve(model, N, eg - wasted, ec)
    ach_exec = f_exec * B * world / sec_per_solve / 1e12
    ach_use = f_use * B * world / sec_per_solve / 1e12
    return {"achieved": ach_use, "achieved_executed": ach_exec, "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s",
            "frac": ach_use / (FP64_VALU_PEAK_TF * world), "frac_useful": ach_use / (FP64_VALU_PEAK_TF * world),
            "frac_executed": ach_exec / (FP64_VALU_PEAK_TF * world),
            "modelled": True,
            "what": "whole timed step: flops per solve x solves per step / wall time of a step",
            "flops_per_solve": f_use, "flops_per_solve_executed": f_exec,
            "E_g": eg - wasted, "E_g_executed": eg, "E_f": ec, "speculative_gradients_unused_per_solve": wasted, "C_ode": c_ode,
            "flop_model": "SURVEY 8(d): F = (4 E_g + E_f) 16 N C_ode -- a MODEL, not a count: E_g, E_f = gradient / cost "
                          "evaluations per solve (measured); a gradient is priced at four forward rollouts and an fp64 "
                          "transcendental at 20 flops (conventions).  `frac_useful` counts the evaluations the solver "
                          "consumed, `frac_executed` also the speculative gradients it issued and threw away; the "
                          "issue-based view of the same kernels is the SQ_INSTS_VALU pass in profiles/*_sq_counters.txt"}


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

    # Everything below is untimed and rank 0's alone: the process group is taken down first so that the other ranks
    # leave instead of sitting in a collective for the seconds rank 0 spends on its extra passes
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
        if rank != 0:
            return

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
                nl = launches.get(k, 0)
                if nl == 0 or ms <= 0:
                    continue               # a kernel that did not run in this pass (adjoint_kernel: K1c runs fused)
                ent = {"ms_per_solve": ms, "launches": nl, "avg_launch_ms": ms / nl,
                       "share_of_kernel_time": ms / tot_ms if tot_ms > 0 else None,
                       "model_bytes_per_solve": model.get(k, 0),
                       "model_GBps": model.get(k, 0) / (ms * 1e-3) / 1e9}
                # the kernels the counters name for this slot of the round (K1a has three forms, K1b runs fused with K1c
                # on the kinematic model)
                names = {"step": ["step_kernel"], "rollout": ["rollout_kernel", "rollout_pair_kernel", "rollout_quad_kernel", "rollout_wide_kernel"],
                         "stage": ["stage_kernel", "stage_adjoint_kernel"], "adjoint": ["adjoint_kernel"], "solo": ["solo_kernel"]}.get(k, [])
                got = [pmc[nm].get("hbm_bytes_per_solve_corrected", 0.0) for nm in names if pmc and isinstance(pmc.get(nm), dict)]
                if got:
                    # both byte figures per SOLVE (the counter passes run the default four groups, i.e. quarter-size
                    # launches; this pass one group: per launch they are not comparable, per solve they are)
                    ent["pmc_hbm_bytes_per_solve"] = float(sum(got))
                    ent["pmc_source"] = pmc_src
                kernels[k + "_kernel"] = ent
            dominant = max(kms, key=kms.get)
        # fp64 work, SURVEY 8(d): F = (4 E_g + E_f) 16 N C_ode with the measured evaluation counts (executed
        # evaluations of rank 0's shard, the speculative ones included) against the wall time of a timed step
        # (all kernels, overlap included) -- and, on the single-group pass, against the K1 kernels' own time
        mean_info = {"evals_grad": eg, "evals_cost": ec, "spec_issued": spec_i, "spec_used": spec_u}
        fp64 = fp64_roofline(args.model, N, B, mean_info, step_s, world)
        if kinfo:
            kms = kinfo["kernel_ms"]
            per = kinfo["evals_grad"] + kinfo["evals_cost"]
            f_pass, _ = fp64_flops_per_solve(args.model, N, (kinfo["evals_grad"] - kinfo["spec_issued"] + kinfo["spec_used"]) / B,
                                             kinfo["evals_cost"] / B)
            k1_ms = kms.get("rollout", 0) + kms.get("stage", 0) + kms.get("adjoint", 0)
            if k1_ms > 0:
                k1 = f_pass * B / (k1_ms * 1e-3) / 1e12
                fp64["k1_only"] = {"achieved": k1, "frac": k1 / FP64_VALU_PEAK_TF, "ms_per_solve": k1_ms,
                                   "kernels": "K1a+K1b+K1c on the single-group pass (their own time, no overlap "
                                              "with the step kernel), the useful F"}
        if pmc and pmc.get("sq_valu_wave_insts_per_solve_total"):
            # the counter-based sibling of the modelled fraction: vector instructions the kernels ISSUED per
            # solve (SQ_INSTS_VALU of the profiled run of this build) against the issue slots of the chip in a step, at the
            # 4 cycles an fp64 FMA holds a SIMD-32 (other vector instructions hold it for 2: an upper bound of the share)
            vi = float(pmc["sq_valu_wave_insts_per_solve_total"])
            fp64["issue_based"] = {"valu_wave_instructions_per_solve": vi, "source": pmc_src,
                                   "simd_issue_share_at_4_cycles": vi * 4.0 / (256 * 4 * 2.4e9 * step_s),
                                   "note": "counted instructions of the profiled run (one solve on one stream), time of this run; "
                                           "256 CUs x 4 SIMDs at 2.4 GHz"}
        dk = kernels.get(dominant + "_kernel") if dominant else None
        out = {
            "metric": "MPC solves/sec, bicycle model N=20 nx=4 nu=2, batch=65536; 1/2/4/8 GPU",
            "value": B_total * K / dt, "unit": "solves/s", "n_gpus": world, "steps": K,
            "warmup": args.warmup, "ms_per_step": step_s * 1e3, "hip_event_ms_per_step": ev_ms_per_step,
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "note": "`value` is measured at the reference's own tolerance eps = 1e-6 (controller.py:41); north_star's parity "
                    "bar (controls within 1e-5 of the CPU path) holds for EVERY sampled agent at eps = %g: that throughput is "
                    "`parity_at_1e-5.value`, and `parity_sample.frac_dU_le_1e-5` is the fraction within the bar at `value`'s "
                    "tolerance" % PARITY_EPS,
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
                "fp64_valu": fp64, "flops_fraction": fp64["frac_useful"], "flops_fraction_is": "modelled, useful evaluations only",
                "note": "the solve is fp64-issue / latency bound, not HBM bound and not MFMA (SURVEY 8d, DESIGN.md 5)"},
            "kernels": kernels,
            "library_source_sha256": mp._lib.library_hash(),
            "controls_sha256_first_65536": hashlib.sha256(
                np.ascontiguousarray(full[:65536].cpu().numpy()).tobytes()).hexdigest(),
        }
        if pipe:
            out["pipelined_two_handles"] = pipe
        if args.profile_timed:
            out["kernels_overlapped_ms_per_step"] = {
                k: float(np.mean([i["kernel_ms"][k] for i in infos])) for k in infos[0]["kernel_ms"]}
        Unp, stnp = U.cpu().numpy(), st.cpu().numpy()
        if not args.no_cpu_baseline and world == 1:
            base, parity = cpu_baseline(args.model, N, cfg_kw, cl_np, Unp, stnp, n=args.cpu_sample)
            out["cpu_baseline"] = base
            out["parity_sample"] = parity
            out["solver"]["frac_dU_le_1e-5_at_this_tolerance"] = parity["frac_dU_le_1e-5"]
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
                           "reference's 1e-6 (controller.py:41) to the loosest value at which every sampled agent is "
                           "within 1e-5 relative of the CPU oracle; certified here on every 8th agent of the batch (all 16 "
                           "blocks of the generator), which is not the sample the tolerance was chosen on"}
            if not args.no_cpu_baseline:
                stride = np.arange(0, B, 8)
                _, par = cpu_baseline(args.model, N, kw, cl_np, Up.cpu().numpy(), stp.cpu().numpy(), agents=stride,
                                      what="every 8th agent (0, 8, 16, ...)")
                leg["parity_sample"] = par
                leg["frac_dU_le_1e-5"] = par["frac_dU_le_1e-5"]
            out["parity_at_1e-5"] = leg
            del eng_p
        secondary = {}
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
            alg_p = 8 * (6 + 2 * 2 * Np + 4)
            pac = {
                "workload": "%d agents/GPU, Pacejka bicycle nx=6 nu=2 (car_dynamics.py:93-129), N=12 (main.py:68), box input "
                            "constraints, straight S=100 centerline, ALM+PANOC eps=1e-6, no evaluation budget" % B,
                "value": B / sec, "unit": "solves/s", "ms_per_step": sec * 1e3, "steps": 2, "blocking": True,
                "converged_frac": float((sts[:, 0] == 1).double().mean().item()),
                "inner_iters_mean": float(sts[:, 2].mean().item()), "evals_per_solve_mean": float(sts[:, 7].mean().item()),
                "evals_per_solve_max": float(sts[:, 7].max().item()), "rounds": infs["rounds"],
                "solo_agents": infs["solo_agents"],
                "solo_kernel_ms_longest": infk["solo_longest_ms"], "solo_kernel_share": infk["solo_longest_ms"] / (sec * 1e3),
                "solo_kernel_ms_sum_over_groups": infk["kernel_ms"]["solo"],
                "kernel_ms_one_profiled_solve": infk["kernel_ms"],
                "lookahead": {"candidate_evaluations": infs["lookahead_evals"], "requests_served_from_them": infs["lookahead_hits"],
                              "note": "persistent kernel only: points the state machine was going to ask for, evaluated in the idle "
                                      "lanes of a trip (not counted in E_g / E_f of the roofline: they cost no trip)"},
                "roofline": {"bound": "hbm", "achieved": alg_p * B / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": alg_p * B / sec / 1e9 / HBM_PEAK_GBS, "traffic": None,
                             "algorithmic_bytes_per_solve": alg_p,
                             "fp64_valu": fp64_roofline(mp.MODEL_PACEJKA, Np, B, infs, sec),
                             "note": "same definitions as the headline's roofline; no counter pass exists for this workload"},
                "controls_sha256": hashlib.sha256(np.ascontiguousarray(Us.cpu().numpy()).tobytes()).hexdigest()}
            pac["roofline"]["flops_fraction"] = pac["roofline"]["fp64_valu"]["frac_useful"]
            if not args.no_cpu_baseline:
                pb, pp = cpu_baseline(mp.MODEL_PACEJKA, Np, {}, cl_np, Us.cpu().numpy(), sts.cpu().numpy(), n=args.cpu_sample)
                pac["cpu_baseline"] = pb
                pac["parity_sample"] = pp
            secondary["pacejka_nx6_N12"] = pac
            del eng_s
            # ---- BASELINE configs[1]: 4 096 agents (the whole batch runs in the persistent wave-per-agent kernel)
            B2 = 4096
            eng_2 = mp.BatchedMPC(cfg, dev)
            sec2, U2, st2, inf2 = timed_solves(eng_2, X0[:B2].contiguous(), cl, U0[:B2].contiguous(), dev, steps=2)
            secondary["config2_b4096"] = {
                "workload": "BASELINE configs[1]: 4096 agents, kinematic bicycle nx=4 nu=2, N=%d, box input constraints, "
                            "straight S=100 centerline, eps=1e-6" % N,
                "value": B2 / sec2, "unit": "solves/s", "ms_per_step": sec2 * 1e3, "steps": 2, "blocking": True,
                "converged_frac": float((st2[:, 0] == 1).double().mean().item()), "rounds": inf2["rounds"],
                "solo_agents": inf2["solo_agents"],
                "same_controls_as_the_first_4096_of_the_65536_batch": bool(torch.equal(U2, U[:B2])),
                "fp64_valu": fp64_roofline(args.model, N, B2, inf2, sec2)}
            del eng_2
            # ---- BASELINE configs[2]: N = 40, lane band around per-agent Bezier lane-change centerlines
            from model_predictive_control_amd.bezier_curves import lane_change_centerlines
            N3 = 40
            cfg3 = mp.default_config(mp.MODEL_KINEMATIC, N3, constr_mode=mp.CONSTR_LANE, lane_halfwidth=0.05,
                                     max_total_inner=1000, max_total_evals=4000, Sigma0=10.0)
            eng_3 = mp.BatchedMPC(cfg3, dev)
            tabs = lane_change_centerlines(S=100)
            rng = np.random.default_rng(0)
            x3 = np.stack([rng.uniform(0, 2, B), rng.uniform(-.02, .02, B), rng.uniform(-.05, .05, B), rng.uniform(.5, 1.2, B)], 1)
            X3 = torch.tensor(x3, dtype=torch.float64, device=dev)
            cl3 = torch.tensor(tabs, dtype=torch.float64, device=dev)
            ci3 = torch.tensor(rng.integers(0, tabs.shape[0], B).astype(np.int32), device=dev)
            U30 = torch.tensor([1.0, 0.0], dtype=torch.float64, device=dev).repeat(B, N3)
            eng_3.solve(X3, cl3, U30, cl_index=ci3)
            torch.cuda.synchronize(dev)
            t3 = time.perf_counter()
            for _ in range(2):
                U3, lam3, st3 = eng_3.solve(X3, cl3, U30, cl_index=ci3)
            torch.cuda.synchronize(dev)
            sec3 = (time.perf_counter() - t3) / 2
            inf3 = eng_3.last_solve_info()
            secondary["config3_n40_lane"] = {
                "workload": "BASELINE configs[2]: %d agents, kinematic bicycle nx=4, N=40, lane band of half-width 0.05 around "
                            "per-agent Bezier lane-change centerlines (bezier_curves.py, 10 rows by cl_index), Sigma0=10, "
                            "budgets 1000 inner iterations / 4000 evaluations (tools/dev/config3.py)" % B,
                "value": B / sec3, "unit": "solves/s", "ms_per_step": sec3 * 1e3, "steps": 2, "blocking": True,
                "converged_frac": float((st3[:, 0] == 1).double().mean().item()),
                "inner_iters_mean": float(st3[:, 2].mean().item()), "evals_per_solve_mean": float(st3[:, 7].mean().item()),
                "rounds": inf3["rounds"], "solo_agents": inf3["solo_agents"],
                "roofline": {"bound": "hbm", "algorithmic_bytes_per_solve": 8 * (4 + 2 * 2 * N3 + 2 * N3 + 4),
                             "achieved": 8 * (4 + 2 * 2 * N3 + 2 * N3 + 4) * B / sec3 / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "fp64_valu": fp64_roofline(mp.MODEL_KINEMATIC, N3, B, inf3, sec3)}}
            del eng_3
        if secondary:
            out["secondary"] = secondary
        print(json.dumps(out))


if __name__ == "__main__":
    main()
