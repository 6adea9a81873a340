#!/bin/bash
# pairs/s vs pairs per step (float32 default algorithms, f16x3, direct-only), run ON the GPU box
for b in 1 2 4 6 8 16 32 64 128 256; do
  SE3TN_NO_ALT=1 python bench.py --no-parity --track-frames 0 --exact-steps --no-cpu-baseline --batch $b --steps 100 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('f32    n=%-4d %9.1f pairs/s  %8.4f ms/step' % ($b, d['value'], d['ms_per_step']))"
  SE3TN_NO_ALT=1 python bench.py --no-parity --track-frames 0 --exact-steps --no-cpu-baseline --batch $b --steps 100 --precision f16x3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('f16x3  n=%-4d %9.1f pairs/s  %8.4f ms/step' % ($b, d['value'], d['ms_per_step']))"
done
