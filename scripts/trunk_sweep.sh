#!/bin/bash
# GPU box: per-launch times of the 64-channel trunk, direct kernels vs the fused Winograd F(2x2) kernel forced on at every batch size
# (SE3TN_TRUNK_WINOGRAD_FILL=0): the data behind SE3TN_TRUNK_WINOGRAD_DEFAULT_MIN_FILL.  Columns: the four trunk launches (ms).
for b in 8 12 16 20 24 28 32 40 48 56 64 80 96 112 128; do for w in 0 1; do
SE3TN_TRUNK_WINOGRAD=$w SE3TN_TRUNK_WINOGRAD_FILL=0 SE3TN_NO_ALT=1 python bench.py --no-parity --track-frames 0 --exact-steps --no-cpu-baseline --batch $b --steps 40 --streams 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n=%-4d %s %9.1f pairs/s  %8.4f ms/step ' % ($b, 'fused ' if $w else 'direct', d['value'], d['ms_per_step']), ' '.join('%.3f'%v for v in list(d['layers_ms'].values())[2:6]))"
done; done
