#!/bin/bash
# scripts/multi_bench.sh lib1.so lib2.so ... : one bench line per library (parity not checked)
for L in "$@"; do
  SE3TN_LIB=$L python bench.py --no-cpu-baseline --steps 30 $BENCH_ARGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$L'[-14:], d['value'], d['roofline']['achieved'], ' '.join('%.3f'%v for v in d['layers_ms'].values()))"
done
