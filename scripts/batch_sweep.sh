#!/bin/bash
# pairs/s vs pairs per step (float32 default algorithms and f16x3): two-lane pipelined throughput (the bench headline) and
# the strict single-stream figure, run ON the GPU box
for b in 1 2 4 6 8 16 32 64 128 256; do
  for p in f32 f16x3; do
    SE3TN_NO_ALT=1 python bench.py --no-parity --track-frames 0 --exact-steps --no-cpu-baseline --batch $b --steps 100 --precision $p 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-6s n=%-4d %9.1f pairs/s pipelined  %9.1f single-stream  %8.4f ms/step single' % ('$p', $b, d['value'], d['single_stream']['value'], d['single_stream']['ms_per_step']))"
  done
done
