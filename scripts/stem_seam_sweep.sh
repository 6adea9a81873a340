#!/bin/bash
# 1-8 pairs per step, single stream and two lanes, for the shipped library (stem_pool_small up to 5 pairs) and a variant built with
# -DSE3TN_STEM_SMALL_MAX_N=2 (round 5's rule: the batch-64 stem + pool pair from 3 pairs):  scripts/stem_seam_sweep.sh [variant.so]
for L in "" "$1"; do
  [ -z "$L" ] && echo "# shipped library (SE3TN_STEM_SMALL_MAX_N = 5)" || echo "# $L"
  for b in 1 2 3 4 5 6 8; do
    SE3TN_LIB=$L SE3TN_NO_ALT=1 python bench.py --no-parity --track-frames 0 --exact-steps --no-cpu-baseline --batch $b --steps 300 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f32    n=%-4d %9.1f pairs/s pipelined  %9.1f single-stream  %8.4f ms/step single' % ($b, d['value'], d['single_stream']['value'], d['single_stream']['ms_per_step']))"
  done
  [ -z "$1" ] && break
done
