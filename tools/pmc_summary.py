#!/usr/bin/env python3
"""profiles/<tag>_* from what tools/profile.sh collected under gpurun_out/<tag>/.

    python tools/pmc_summary.py r02a

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section) prescribes:
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read stream,
so the read side is doubled (other access widths are uncalibrated: both figures are kept).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", tag)
PROF = os.path.join(ROOT, "profiles")


def short(name):
    return name.split("(")[0].replace("void mpc::", "").replace("mpc::", "").split("<")[0]


def counter_rows(sub):
    for f in glob.glob(os.path.join(OUT, sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "mpc" in row["Kernel_Name"]:
                yield short(row["Kernel_Name"]), row["Counter_Name"], float(row["Counter_Value"])


# ---- 1. kernel stats
for f in glob.glob(os.path.join(OUT, "stats", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(PROF, tag + "_kernel_stats.csv"))
for f in glob.glob(os.path.join(OUT, "stats_groups1", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(PROF, tag + "_kernel_stats_groups1.csv"))
for name in ("bench_under_rocprof.json",):
    if os.path.exists(os.path.join(OUT, name)):
        shutil.copy(os.path.join(OUT, name), os.path.join(PROF, tag + "_" + name))

# ---- 2. HBM traffic
acc = {"FETCH_SIZE": collections.defaultdict(lambda: [0.0, 0]), "WRITE_SIZE": collections.defaultdict(lambda: [0.0, 0])}
for sub, key in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    for k, cname, v in counter_rows(sub):
        if cname == key:
            acc[key][k][0] += v; acc[key][k][1] += 1
summary = {
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), "
              "bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-pass (one batched solve of 65536 agents, "
              "the default sub-batch groups: 4 with GPU_MAX_HW_QUEUES >= 5, else 3), profile %s" % tag,
    "correction": "gfx950: FETCH_SIZE reports half of a wide coalesced read stream (MI355X_MICROARCH.md HBM "
                  "section): read side doubled; other access widths uncalibrated",
}
try:
    bj = json.load(open(os.path.join(OUT, "bench_pmc_fetch.json")))
    summary["workload"] = bj["config"]["workload"]
    # the build the counters were taken on: bench.py uses this summary only when it is the running library's
    summary["library_source_sha256"] = bj.get("library_source_sha256")
except Exception:
    summary["workload"] = None
    summary["library_source_sha256"] = None
total = total_unc = 0.0
for k in sorted(set(acc["FETCH_SIZE"]) | set(acc["WRITE_SIZE"])):
    f, nf = acc["FETCH_SIZE"].get(k, [0.0, 0]); w, nw = acc["WRITE_SIZE"].get(k, [0.0, 0])
    if not nf or not nw:
        continue
    fk, wk = f / nf, w / nw
    summary[k] = {"dispatches": nf, "fetch_kb_reported_per_launch": fk, "write_kb_per_launch": wk,
                  "hbm_bytes_per_launch_corrected": (2 * fk + wk) * 1024,
                  "hbm_bytes_per_launch_uncorrected": (fk + wk) * 1024,
                  "hbm_bytes_per_solve_corrected": (2 * f + w) * 1024}
    total += (2 * f + w) * 1024; total_unc += (f + w) * 1024
summary["hbm_bytes_per_solve"] = total
summary["hbm_bytes_per_solve_uncorrected"] = total_unc
json.dump(summary, open(os.path.join(PROF, tag + "_pmc_summary.json"), "w"), indent=1)
print("HBM bytes per solve (corrected): %.1f GB" % (total / 1e9))

# ---- 3. SQ counters (one group)
sq = collections.defaultdict(collections.Counter); nd = collections.Counter()
for k, cname, v in counter_rows("sq"):
    sq[k][cname] += v
    if cname == "SQ_WAVES":
        nd[k] += 1
# the VALU instruction count of ONE solve (the SQ pass runs bench.py --steps 1 --warmup 0 on one stream) goes into the
# counter summary too: bench.py's issue-based sibling of the modelled flop fraction
try:
    sj = json.load(open(os.path.join(PROF, tag + "_pmc_summary.json")))
    sj["sq_valu_wave_insts_per_solve"] = {k: v.get("SQ_INSTS_VALU", 0.0) for k, v in sq.items() if v.get("SQ_INSTS_VALU", 0.0) > 1e6}
    sj["sq_valu_wave_insts_per_solve_total"] = sum(sj["sq_valu_wave_insts_per_solve"].values())
    json.dump(sj, open(os.path.join(PROF, tag + "_pmc_summary.json"), "w"), indent=1)
except Exception as e:
    print("no SQ totals in the summary:", e)
with open(os.path.join(PROF, tag + "_sq_counters.txt"), "w") as fo:
    fo.write("rocprofv3 --kernel-trace --pmc SQ_* of bench.py --steps 1 --warmup 0, MPC_GROUPS=1 (no overlap), profile %s\n"
             "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves\n" % tag)
    for k, v in sq.items():
        fo.write("%s dispatches %d\n" % (k, nd[k]))
        for c, val in sorted(v.items()):
            fo.write("   %-22s %.4g  per dispatch %.4g\n" % (c, val, val / max(nd[k], 1)))
        wc = v.get("SQ_WAVE_CYCLES", 0)
        if wc:
            fo.write("   -> wait %.0f %%, issue-stall %.0f %%, active %.0f %% of wave cycles; VALU per wave %.0f\n"
                     % (100 * v["SQ_WAIT_ANY"] / wc, 100 * v["SQ_WAIT_INST_ANY"] / wc, 100 * v["SQ_ACTIVE_INST_ANY"] / wc,
                        v["SQ_INSTS_VALU"] / max(v["SQ_WAVES"], 1)))

# ---- 4. the tail: one group, the last solve of the trace, persistent kernel on / off
def tail(sub):
    rows = []
    for f in glob.glob(os.path.join(OUT, sub, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "mpc" in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]),
                             int(r.get("Grid_Size", 0) or 0)))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if r[2] == "init_kernel"]
    if not starts:
        return None
    rows = rows[starts[-1]:]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    # rollout launches whose grid was sized for <= 4096 requests mark the few-request phase
    small = [r for r in rows if r[2] in ("rollout_wide_kernel",)]
    first_small = small[0][0] if small else t1
    solo = [r for r in rows if r[2] == "solo_kernel"]
    return {"solve_ms": (t1 - t0) / 1e6, "few_request_phase_ms": (t1 - first_small) / 1e6,
            "few_request_share": (t1 - first_small) / (t1 - t0),
            "launches": len(rows), "solo_kernel_ms": sum(r[1] - r[0] for r in solo) / 1e6,
            "rollout_wide_launches": len(small)}

with open(os.path.join(PROF, tag + "_tail.txt"), "w") as fo:
    fo.write("Kernel trace of ONE batched solve (65536 agents, MPC_GROUPS=1: a single stream, no overlap), profile %s.\n"
             "few-request phase = from the first round served by the wave-per-request rollout (<= 4096 requests) to the end.\n" % tag)
    for sub, label in (("trace_rounds", "rounds only (MPC_SOLO_MAX=0)"), ("trace_solo", "persistent kernel from <= 1024 requests (default)")):
        t = tail(sub)
        fo.write("%s: %s\n" % (label, json.dumps(t)))
        print(label, t)
