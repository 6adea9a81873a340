#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ directory (scripts/profile_r.sh) into profiles/<tag>_*.
   python scripts/summarize_profile.py gpurun_out/prof_r01a profiles/r01a"""
import collections
import csv
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)


def short(name):
    name = name.replace("void ", "").replace("se3tn::", "")
    return name.split("(")[0][:70]


lines = []
# ---- kernel-trace stats -----------------------------------------------------------------------
stats = list(csv.DictReader(open(os.path.join(src, "trace", "trace_kernel_stats.csv"))))
ours = [r for r in stats if "se3tn::" in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in ours)
cmd = "python bench.py --steps 20 --warmup 3"
if os.path.exists(os.path.join(src, "cmd.txt")):
    cmd = open(os.path.join(src, "cmd.txt")).read().strip()
lines.append("## rocprofv3 --kernel-trace --stats (%s)\n" % cmd)
lines.append("| kernel | calls | avg us | min us | max us | % of se3tn time |\n|---|---|---|---|---|---|")
for r in ours:
    lines.append("| %s | %s | %.1f | %.1f | %.1f | %.2f |" % (
        short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3,
        float(r["MaxNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
with open(dst + "_kernel_stats.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs"])
    for r in ours:
        w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"]])
bj = os.path.join(src, "bench_traced.json")
if os.path.exists(bj):
    try:
        d = json.loads(open(bj).read().strip().splitlines()[-1])
        lines.append("\nbench.py line of the traced run: value=%s pairs/s, roofline=%s\n" % (d["value"], json.dumps(d["roofline"])))
    except Exception as e:  # noqa
        lines.append("\n(bench line unreadable: %s)\n" % e)


# ---- PMC passes ------------------------------------------------------------------------------
def pmc(sub):
    p = os.path.join(src, sub, "pmc_counter_collection.csv")
    if not os.path.exists(p):
        return {}
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(p)):
        if "se3tn::" not in r["Kernel_Name"]:
            continue
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


def pmc_totals(sub, counter):
    """Per-kernel SUM of a counter over every dispatch of the pass + the number of dispatches (for per-step figures)."""
    p = os.path.join(src, sub, "pmc_counter_collection.csv")
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    if os.path.exists(p):
        for r in csv.DictReader(open(p)):
            if "se3tn::" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                tot[short(r["Kernel_Name"])] += float(r["Counter_Value"])
                cnt[short(r["Kernel_Name"])] += 1
    return tot, cnt


sq, lds, fe, wr = pmc("pmc_sq"), pmc("pmc_lds"), pmc("pmc_fetch"), pmc("pmc_write")
# ---- HBM + Infinity-Cache bytes PER STEP: every dispatch of the pass summed, divided by the number of steps the pass ran (= the
# number of stem launches: one per step).  FETCH_SIZE x 2 (gfx950 wide-load correction) + WRITE_SIZE, counters in KB.
per_step = None
ft, fc = pmc_totals("pmc_fetch", "FETCH_SIZE")
wt, wc_ = pmc_totals("pmc_write", "WRITE_SIZE")
stem = [k for k in fc if k.startswith("stem7x7")]
if stem and all(k in wc_ for k in stem):
    nsteps = sum(fc[k] for k in stem)
    assert nsteps == sum(wc_[k] for k in stem), "the FETCH and WRITE passes ran different numbers of steps"
    by_kernel = {k: (2.0 * ft[k] + wt.get(k, 0.0)) * 1024.0 / nsteps for k in ft}
    conv = lambda k: k.startswith(("conv3x3", "conv_reduce", "wino", "tail_kernel", "fc_finish"))   # noqa: E731
    per_step = {"steps_counted": int(nsteps), "all_kernels_bytes": int(sum(by_kernel.values())),
                "conv_family_bytes": int(sum(v for k, v in by_kernel.items() if conv(k) and not k.startswith("wino_weight"))),
                "launches_per_step": {k: round(fc[k] / nsteps, 3) for k in fc if not k.startswith("wino_weight")},
                "bytes_per_step_by_kernel": {k: int(v) for k, v in by_kernel.items() if not k.startswith("wino_weight")},
                "note": "sum over every dispatch of the FETCH_SIZE (x2) and WRITE_SIZE passes / steps of the pass; Infinity-Cache hits included"}
dur = {short(r["Name"]): float(r["AverageNs"]) for r in ours}
if sq:
    lines.append("\n## PMC (separate passes, averages per dispatch)\n")
    lines.append("GRBM_GUI_ACTIVE is summed over the 8 XCDs (a 620 us kernel reads 11.8 M => 8 x 2.37 GHz), so "
                 "eff. clock = GRBM_GUI_ACTIVE / 8 / kernel time (trace run) and MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / "
                 "(GRBM_GUI_ACTIVE/8 x 1024 SIMDs); SQ_VALU_MFMA_BUSY_CYCLES = 64 x (number of v_mfma_f32_32x32x2_f32), "
                 "checked against the launch's FLOPs. HBM bytes: FETCH_SIZE x2 (gfx950 wide-load correction, "
                 "MI355X_MICROARCH.md section HBM; Infinity-Cache hits are included) and WRITE_SIZE, counters in KB.\n")
    lines.append("| kernel | GRBM_GUI_ACTIVE | MFMA_BUSY_CYCLES | mfma busy % | eff. clock GHz | WAVE_CYCLES | WAIT_ANY % | WAIT_INST_ANY % | FETCH MB (x2) | WRITE MB | LDS bank conflict % |\n|---|---|---|---|---|---|---|---|---|---|---|")
    for k, d in sq.items():
        gui = d.get("GRBM_GUI_ACTIVE", 0)
        mf = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
        wc = d.get("SQ_WAVE_CYCLES", 0)
        busy = 100 * mf / (gui / 8 * 1024) if gui else float("nan")
        clk = gui / 8 / dur[k] if k in dur else float("nan")
        f = fe.get(k, {}).get("FETCH_SIZE")
        w = wr.get(k, {}).get("WRITE_SIZE")
        l = lds.get(k, {})
        bc = 100 * l.get("SQ_LDS_BANK_CONFLICT", 0) / l["SQ_LDS_IDX_ACTIVE"] if l.get("SQ_LDS_IDX_ACTIVE") else float("nan")
        lines.append("| %s | %.0f | %.3e | %.1f | %.2f | %.3e | %.1f | %.1f | %s | %s | %.2f |" % (
            k, gui, mf, busy, clk, wc, 100 * d.get("SQ_WAIT_ANY", 0) / wc if wc else 0,
            100 * d.get("SQ_WAIT_INST_ANY", 0) / wc if wc else 0,
            "%.1f" % (2 * f / 1024) if f is not None else "-", "%.1f" % (w / 1024) if w is not None else "-", bc))
    with open(dst + "_pmc.json", "w") as f:
        json.dump({"sq": sq, "lds": lds, "fetch": fe, "write": wr, "per_step": per_step}, f, indent=1)
    if per_step:
        lines.append("\n## HBM + Infinity-Cache bytes per step (every dispatch of the FETCH x2 / WRITE passes summed, / %d steps)\n" % per_step["steps_counted"])
        lines.append("all kernels %.1f MB, conv family %.1f MB\n" % (per_step["all_kernels_bytes"] / 1e6, per_step["conv_family_bytes"] / 1e6))
        lines.append("| kernel | launches / step | MB / step |\n|---|---|---|")
        for k, v in sorted(per_step["bytes_per_step_by_kernel"].items(), key=lambda kv: -kv[1]):
            lines.append("| %s | %s | %.1f |" % (k, per_step["launches_per_step"].get(k), v / 1e6))
open(dst + "_summary.md", "w").write("# rocprofv3 summary %s\n\n" % os.path.basename(dst) + "\n".join(lines) + "\n")
print("\n".join(lines))
