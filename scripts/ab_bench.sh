#!/bin/bash
# within-session A/B of two builds of the library: scripts/ab_bench.sh libA.so libB.so [rounds]
A=$1; B=$2; R=${3:-2}
for i in $(seq $R); do
  for L in $A $B; do
    SE3TN_NO_ALT=1 SE3TN_LIB=$L python bench.py --no-cpu-baseline --no-parity --track-frames 0 --exact-steps --streams 1 --steps 40 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$L'[-20:], d['value'], d['roofline']['achieved'], ' '.join('%.3f'%v for v in d['layers_ms'].values()))"
  done
done
