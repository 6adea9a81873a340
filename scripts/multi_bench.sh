#!/bin/bash
# scripts/multi_bench.sh lib1.so lib2.so ... : one line per library: value, executed TF, per-launch ms (timing only)
for L in "$@"; do
  SE3TN_NOCHECK=1 SE3TN_NO_ALT=1 SE3TN_LIB=$L python bench.py --no-cpu-baseline --no-parity --track-frames 0 --steps ${STEPS:-100} --exact-steps --streams 1 $BENCH_ARGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-22s'%'$L'[-22:], d['value'], d['roofline']['achieved'], ' '.join('%.3f'%v for v in d['layers_ms'].values()))"
done
