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
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
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
            # fp64 work estimate for K1: 16 RHS per stage (rollout), +16 RHS with partials and
            # NX tangent directions per stage (gradient requests), 98-point nearest scan per stage
            c_ode = 250.0 if args.model == mp.MODEL_PACEJKA else 120.0
            flop_k1 = per * N * (16 * c_ode + 98 * 8) + egk * N * 16 * c_ode * (1 + 0.5 * nx)
            k1_ms = kms.get("rollout", 0) + kms.get("stage", 0) + kms.get("adjoint", 0)
            fp64 = {"achieved": flop_k1 / (k1_ms * 1e-3) / 1e12 if k1_ms > 0 else None, "peak": FP64_VALU_PEAK_TF,
                    "unit": "TFLOP/s", "kernels": "K1a+K1b+K1c (single-group pass)",
                    "flop_model": "16 RHS/stage + 98-pt scan; gradient: +16 RHS with NX tangents; C_ode=%g" % c_ode}
            if fp64["achieved"]:
                fp64["frac"] = fp64["achieved"] / FP64_VALU_PEAK_TF
                # the same work against the wall time of a timed step (all kernels, overlap included)
                fp64["whole_step_TFLOPs"] = flop_k1 / step_s / 1e12
        else:
            fp64 = None
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
                       "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
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
                "fp64_valu": fp64,
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
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
