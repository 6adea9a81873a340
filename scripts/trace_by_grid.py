#!/usr/bin/env python3
"""Per (kernel, grid size) durations of a rocprofv3 --kernel-trace CSV: separates the launches that share a kernel name (e.g. the
grouped A|B and the single-branch launches of wino64_fused_kernel<EPI>).   python scripts/trace_by_grid.py <trace_kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if "se3tn::" not in r["Kernel_Name"]:
        continue
    name = r["Kernel_Name"].replace("void ", "").replace("se3tn::", "").split("(")[0][:60]
    grid = (int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    acc[(name, grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("%-62s %-16s %6s %9s %9s %9s %9s" % ("kernel", "workgroups x,y,z", "calls", "median us", "min us", "max us", "max/min"))
for (name, grid), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    if len(v) < 3:
        continue
    core = v[:-2] if len(v) > 6 else v             # (the two slowest are the clock ramp of the first steps)
    print("%-62s %-16s %6d %9.1f %9.1f %9.1f %9.2f" % (name, "%d,%d,%d" % grid, len(v), core[len(core) // 2], core[0], core[-1], core[-1] / core[0]))
