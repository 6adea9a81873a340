#!/bin/bash
# Run ON the GPU box: rocprofv3 kernel-trace stats of a short bench run -> gpurun_out/ktrace_<tag>.csv (sorted by total time)
TAG=${1:-x}; shift
ROOT=$PWD
OUT=$ROOT/gpurun_out/ktrace_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
SE3TN_NOCHECK=1 SE3TN_NO_ALT=${NOALT:-1} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $ROOT/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-parity --track-frames 0 --exact-steps --streams 1 "$@" > $OUT/bench.json 2> $OUT/trace.err
F=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
python - "$F" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:32]:
    print("%9.1f us x %4d  %s" % (float(r["AverageNs"])/1e3, int(r["Calls"]), r["Name"][:110]))
PY
