#!/usr/bin/env python3
"""Development script (GPU box; not a pytest test): WHERE do the HIP solver and the CPU oracle first part?

VERDICT r3 item 1: on the headline configuration only a few per cent of the agents take the oracle's (status,
inner-iteration) path.  This script finds, for agents whose final iteration counts differ, the FIRST inner iteration
at which the two implementations disagree, says which comparison of the algorithm went the other way and by what
margin the oracle decided it, and how far apart the two iterates were just before.

Method: `max_total_inner = k` stops both implementations after k inner iterations (the last inner solve hands back
its prox point under the `overwrite` rule, oracle/mpc_oracle.c orc_solve / mpc_solver.hpp PH_OUTER_BEGIN), so a solve
with budget k IS the first k iterations of the long solve.  For k = 1, 2, ... the HIP batch solve gives (U_k, stats_k)
and the solver records (mpc_debug_records: step sizes, accepted line-search step, |J|, history fill, counters); the
oracle gives the same from prefix solves, plus one trace row per iteration with the smallest margin by which each kind
of comparison was decided in that iteration (orc_solve_itertrace).

    python tools/dev/first_divergence.py [--agents 64] [--pool 1024] [--kmax 400] [--out profiles/r04_first_divergence.txt]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
import model_predictive_control_amd as mp
from oracle import oracle as O

KINDS = ["stop test (inner solve ends at another iteration)", "active-set membership", "line-search accept",
         "descent lemma (L doubling)", "L-BFGS pair / history", "step-size heuristic", "evaluation count only",
         "iterate only (no discrete difference seen)"]


def prefix_hip(dev, model, N, k, X0, cl, U0, kw):
    eng = mp.BatchedMPC(mp.default_config(model, N, max_total_inner=k, **kw), dev)
    U, _, st = eng.solve(X0, cl, U0)
    rec = eng.debug_records(X0.shape[0])
    eng.close()
    return U.cpu().numpy(), st.cpu().numpy(), rec


def l_kind(Lh, Lo):
    """The Lipschitz estimate is a CONTINUOUS quantity (a finite-difference quotient, or 1 / eta of the step-size
    heuristic every 15 iterations) times a power of two (descent-lemma doublings): 0 = the same up to drift, 1 = apart
    by a power of two (a doubling more or less), 2 = apart by something else (> 5 %)."""
    lg = np.log2(Lh / Lo)
    if abs(lg - round(lg)) < 0.02 and round(lg) != 0:
        return 1
    return 2 if abs(Lh / Lo - 1.0) > 0.05 else 0


def classify(rec, a, row, row_next, st_h, st_o, M):
    """Which decision of iteration k differs: HIP record of agent a after k iterations vs the oracle's trace row k."""
    cnt_h = M if rec["lfull"][a] else rec["lidx"][a]
    tau_o, safe_o = abs(row[4]) / 2.0, bool(np.signbit(row[4]))
    if st_h[1] != st_o[1] or rec["k"][a] != row[2]:
        # one implementation's inner solve ended on this iterate, the other's went on: the stop test on iterate k
        return 0, abs(row[18])
    if rec["nJ"][a] != row[8]:
        return 1, row[17]
    if rec["tau"][a] != tau_o or bool(rec["fallback"][a]) != safe_o:
        return 2, row[15]
    lk = l_kind(rec["L"][a], row[6])
    if st_h[7] != st_o[7] and row[2] % 15 == 0 and lk != 1:
        # the evaluations of the heuristic at the top of the next iteration differ (its margin is in the next row)
        return 5, (row_next[19] if row_next is not None else np.nan)
    if lk == 1:
        return 3, row[16]                  # step sizes a power of two apart: a descent-lemma test went the other way
    if lk == 2:
        return 5, row[19]                  # apart by another factor: the step-size heuristic taken by one of them only
    if cnt_h != row[9]:
        return 4, np.nan
    if st_h[7] != st_o[7]:
        return 6, min(row[15], row[16])
    return 7, np.nan


def study(dev, model, N, args, out):
    kw = {}
    cl_np = bench.straight_centerline()
    pool = args.pool
    X0p = bench.synthetic_states(model, 0, pool)
    U0p = np.tile([1.0, 0.0], (pool, N))
    T = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
    cfg, ocfg = mp.default_config(model, N), O.default_config(model, N)
    Uh, _, sh = mp.BatchedMPC(cfg, dev).solve(T(X0p), T(cl_np), T(U0p))
    Uh, sh = Uh.cpu().numpy(), sh.cpu().numpy()
    Uo, _, so = O.solve_batch(ocfg, X0p, cl_np, U0p)
    same_path = (sh[:, 0] == so[:, 0]) & (sh[:, 2] == so[:, 2])
    same_ev = same_path & (sh[:, 7] == so[:, 7])
    name = "Pacejka nx=6" if model else "kinematic nx=4"
    print(f"== {name}, N = {N}, eps = {ocfg.alm_eps:g}: first {pool} agents of bench.py's batch", file=out)
    print(f"   identical (status, inner iterations): {same_path.mean():.3f}; also the same evaluation count: {same_ev.mean():.3f};"
          f" inner iterations mean HIP {sh[:, 2].mean():.1f} / oracle {so[:, 2].mean():.1f}", file=out)
    d = np.abs(Uh - Uo).max(1) / np.maximum(1.0, np.abs(Uo).max(1))
    print(f"   controls: max rel dU {d.max():.2e}, within 1e-5: {(d <= 1e-5).mean():.4f}  (identical paths only: "
          f"max {d[same_path].max() if same_path.any() else float('nan'):.2e})", file=out)
    # the yardstick: the oracle against itself with 2 ulp of noise on every psi / gradient component it sees
    with O.eval_jitter(2, 1):
        Uj, _, sj = O.solve_batch(ocfg, X0p, cl_np, U0p)
    pj = (sj[:, 0] == so[:, 0]) & (sj[:, 2] == so[:, 2])
    dj = np.abs(Uj - Uo).max(1) / np.maximum(1.0, np.abs(Uo).max(1))
    print(f"   yardstick -- the ORACLE against itself with every evaluation moved by a random -2 .. 2 ulp: identical paths "
          f"{pj.mean():.3f}; same evaluation count too: {(pj & (sj[:, 7] == so[:, 7])).mean():.3f}; controls max rel dU {dj.max():.2e}, "
          f"within 1e-5: {(dj <= 1e-5).mean():.4f}", file=out)
    pick = np.flatnonzero(~same_path)[:args.agents]
    A = len(pick)
    X0, U0 = X0p[pick], U0p[pick]
    # the oracle's trace of every picked agent
    traces = []
    for a in range(A):
        _, _, st, tr = O.solve_itertrace(ocfg, X0[a], cl_np, U0[a])
        assert st[2] == so[pick[a], 2]
        traces.append(tr)
    kmax = int(min(args.kmax, max(min(sh[pick[a], 2], so[pick[a], 2]) for a in range(A)) + 1))
    first = np.full(A, -1)
    kind = np.full(A, -1)
    margin = np.full(A, np.nan)
    drift_before = np.full(A, np.nan)
    drift_prev = np.zeros(A)
    drift_table = {}
    examples = []
    hess_every = int(ocfg.hess_heuristic)
    X0d, cld, U0d = T(X0), T(cl_np), T(U0)
    for k in range(1, kmax + 1):
        if (first >= 0).all():
            break
        Uk, sk, rec = prefix_hip(dev, model, N, k, X0d, cld, U0d, kw)
        Uok, _, sok = O.solve_batch(O.default_config(model, N, max_total_inner=k), X0, cl_np, U0)
        dU = np.abs(Uk - Uok).max(1)
        live = first < 0
        if k in (1, 2, 3, 5, 10, 20, 40, 80, 160):
            drift_table[k] = (int(live.sum()), float(np.median(dU[live])) if live.any() else np.nan,
                              float(dU[live].max()) if live.any() else np.nan)
        for a in np.flatnonzero(live):
            tr = traces[a]
            if k > len(tr) or k > sh[pick[a], 2]:
                # one of the two solves has finished before k iterations: the count itself is the difference
                first[a], kind[a], margin[a], drift_before[a] = k, 0, abs(tr[min(k, len(tr)) - 1][18]), drift_prev[a]
                continue
            row = tr[k - 1]
            # (L is a continuous quantity times a power of two -- the finite-difference Lipschitz estimate differs in its
            # ninth digit between the two, a doubling by a factor of two: compared by ratio.  The iterates themselves drift
            # apart under IDENTICAL decisions -- see the drift line -- so no threshold on dU marks a divergence.)
            # At a multiple of 15 iterations of the inner solve the step-size heuristic runs at the TOP of the next
            # iteration, before the stop test that ends a budgeted solve: the HIP record then holds the state after it
            # (new L, history flushed), the oracle's row -- written at the end of the iteration -- the state before.
            # Whether both took it shows in the evaluation counts of the two prefix solves, which are compared.
            heur_top = row[2] % hess_every == 0 if hess_every > 0 else False
            discrete = (sk[a, 7] != sok[a, 7] or sk[a, 1] != sok[a, 1] or rec["nJ"][a] != row[8]
                        or rec["tau"][a] != abs(row[4]) / 2.0
                        or (not heur_top and (l_kind(rec["L"][a], row[6]) != 0
                                              or (model_M(ocfg) if rec["lfull"][a] else rec["lidx"][a]) != row[9])))
            if discrete:
                kd, mg = classify(rec, a, row, tr[k] if k < len(tr) else None, sk[a], sok[a], model_M(ocfg))
                first[a], kind[a], margin[a], drift_before[a] = k, kd, mg, drift_prev[a]
                if len(examples) < args.examples:
                    cnt = model_M(ocfg) if rec["lfull"][a] else rec["lidx"][a]
                    examples.append(
                        f"     agent {pick[a]} parts at iteration {k} ({KINDS[kd]}; oracle margin {mg:.2e}; drift before {drift_prev[a]:.1e}):\n"
                        f"        HIP    L {rec['L'][a]:.9e} gamma {rec['gamma'][a]:.6e} tau {2 * rec['tau'][a]:g}{' (safe step)' if rec['fallback'][a] else ''} |J| {int(rec['nJ'][a])} pairs {int(cnt)} "
                        f"psi {rec['psi'][a]:.12e} ||p||^2 {rec['pp'][a]:.4e} evals {int(sk[a, 7])} outer {int(sk[a, 1])} k {int(rec['k'][a])}\n"
                        f"        oracle L {row[6]:.9e} gamma {row[7]:.6e} tau {abs(row[4]):g}{' (safe step)' if np.signbit(row[4]) else ''} |J| {int(row[8])} pairs {int(row[9])} "
                        f"psi {row[11]:.12e} ||p||^2 {row[13]:.4e} evals {int(sok[a, 7])} outer {int(sok[a, 1])} k {int(row[2])}; trials {int(row[5])}; "
                        f"margins ls {row[15]:.1e} dl {row[16]:.1e} active {row[17]:.1e} stop {row[18]:.1e} heuristic {row[19]:.1e}")
        drift_prev = dU
    print(f"   {A} agents whose final (status, iterations) differ, scanned k = 1 .. {kmax}:", file=out)
    print("   iterate drift max|U_hip - U_oracle| after k iterations over the agents that have not parted yet: "
          + "; ".join(f"k={k}: n={v[0]} median {v[1]:.1e} max {v[2]:.1e}" for k, v in drift_table.items()), file=out)
    found = first >= 0
    print(f"   first divergence found for {int(found.sum())} of {A} (the others part after iteration {kmax})", file=out)
    fin = np.array([min(sh[pick[a], 2], so[pick[a], 2]) for a in range(A)])
    frac = first[found] / np.maximum(1, fin[found])
    print(f"   position of the first divergence in the solve (k_first / final iterations): median {np.median(frac):.2f}, "
          f"quartiles {np.percentile(frac, 25):.2f} .. {np.percentile(frac, 75):.2f}; k_first median {int(np.median(first[found]))}", file=out)
    print("   which comparison went the other way (the oracle's margin for that comparison in that iteration; relative to "
          "1 + |psi| or 1 + |phi| for the line-search and descent-lemma tests, absolute distance to the bound for the active "
          "set, eps_k / eps - 1 for the stop test):", file=out)
    for kd in range(len(KINDS)):
        sel = found & (kind == kd)
        if not sel.any():
            continue
        mg = margin[sel]
        mg = mg[np.isfinite(mg)]
        hist = ""
        if len(mg):
            edges = [0, 1e-15, 1e-14, 1e-13, 1e-12, 1e-11, 1e-10, 1e-9, 1e-8, 1e-6, 1e-4, 1e-2, np.inf]
            h, _ = np.histogram(mg, edges)
            hist = " margins: " + ", ".join(f"<{edges[i + 1]:.0e}: {h[i]}" for i in range(len(h)) if h[i])
            hist += f"; largest {mg.max():.2e}"
        print(f"     {int(sel.sum()):3d}  {KINDS[kd]}{hist}; iterate drift just before: median "
              f"{np.median(drift_before[sel]):.1e}, max {drift_before[sel].max():.1e}", file=out)
    print("   examples (the state each implementation is in after the iteration at which they part):", file=out)
    for e in examples:
        print(e, file=out)
    # a flip is explained by the drift when the oracle's margin is within what the drift accumulated under identical
    # decisions can move the compared quantity by (psi and phi_gamma have O(1) .. O(10) gradients in U here)
    unexplained = np.flatnonzero(found & np.isfinite(margin) & (kind >= 1) & (kind <= 3) & (margin > 30.0 * np.maximum(drift_before, 1e-15)))
    if len(unexplained):
        print("   flips whose oracle margin exceeds 30 x the iterate drift just before (to be looked at one by one):", file=out)
        for a in unexplained:
            print(f"     agent {pick[a]} k_first {first[a]} kind '{KINDS[kind[a]]}' margin {margin[a]:.3e} drift before "
                  f"{drift_before[a]:.2e}", file=out)
    else:
        print("   every line-search / descent-lemma / active-set flip has an oracle margin below 30 x the iterate drift "
              "accumulated just before it: decided by the last bits, amplified -- not by a different algorithm.", file=out)
    out.flush()


def model_M(ocfg):
    return int(ocfg.lbfgs_memory)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--agents", type=int, default=64)
    ap.add_argument("--pool", type=int, default=1024)
    ap.add_argument("--kmax", type=int, default=400)
    ap.add_argument("--out", default="")
    ap.add_argument("--examples", type=int, default=6)
    ap.add_argument("--models", default="0,1")
    args = ap.parse_args()
    O.build()
    dev = torch.device("cuda:0")
    out = open(args.out, "w") if args.out else sys.stdout
    print("# tools/dev/first_divergence.py: first inner iteration at which the HIP solver and the CPU oracle disagree\n"
          "# (VERDICT r3 item 1).  A 'margin' is the oracle's own slack on the comparison that went the other way in HIP.", file=out)
    for m in [int(v) for v in args.models.split(",")]:
        study(dev, m, 12 if m else 20, args, out)
    if args.out:
        out.close()
        print(open(args.out).read())


if __name__ == "__main__":
    main()
