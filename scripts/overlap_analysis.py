#!/usr/bin/env python3
"""Occupancy of the timeline of a two-lane (pipelined) bench run, from a rocprofv3 --kernel-trace csv:
for the last `--window-ms` of se3tn kernels: wall time with 0 / 1 / >=2 kernels in flight, with >=1 matrix-core
kernel in flight, and the busiest pairs of co-running kernels.
   python scripts/overlap_analysis.py gpurun_out/ovl/trace/trace_kernel_trace.csv"""
import collections
import csv
import sys

path = sys.argv[1]
window_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0
rows = [r for r in csv.DictReader(open(path)) if "se3tn::" in r["Kernel_Name"]]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("se3tn::", "").split("(")[0][:48])
      for r in rows]
ev.sort()
t_end = max(e[1] for e in ev)
t0 = t_end - int(window_ms * 1e6)
ev = [e for e in ev if e[0] >= t0]
MFMA = ("conv3x3_", "stem7x7", "wino_gemm")
pts = []
for s, e, n in ev:
    pts.append((s, 1, n)); pts.append((e, -1, n))
pts.sort(key=lambda p: (p[0], p[1]))
active = collections.Counter()
hist = collections.Counter(); mf = 0; pair = collections.Counter()
last = pts[0][0]
for t, d, n in pts:
    dt = t - last
    if dt > 0:
        k = sum(active.values())
        hist[min(k, 3)] += dt
        names = sorted(x for x in active.elements())
        if any(x.startswith(MFMA) for x in names):
            mf += dt
        if k >= 2:
            pair[tuple(names[:2])] += dt
    active[n] += d
    if active[n] == 0:
        del active[n]
    last = t
tot = sum(hist.values())
print("window %.1f ms, %d kernels" % (tot / 1e6, len(ev)))
for k in sorted(hist):
    print("  %s kernels in flight: %5.1f %%" % (("%d" % k) if k < 3 else ">=3", 100.0 * hist[k] / tot))
print("  >=1 matrix-core kernel (conv / stem / Winograd GEMM) in flight: %.1f %%" % (100.0 * mf / tot))
print("  top co-running pairs (%% of the window):")
for (a, b), dt in pair.most_common(8):
    print("    %5.1f %%  %s | %s" % (100.0 * dt / tot, a, b))
