#!/bin/bash
# one bench line (value, conv TFLOP/s, per-launch ms) per library given: scripts/variants_bench.sh libA.so libB.so ...
for L in "$@"; do
  SE3TN_NO_ALT=1 SE3TN_NOCHECK=1 SE3TN_LIB=$L python bench.py --no-cpu-baseline --no-parity --track-frames 0 --exact-steps --streams 1 --steps 40 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$L'[-24:], d['value'], d['roofline']['achieved'], ' '.join('%.3f'%v for v in d['layers_ms'].values()))"
done
