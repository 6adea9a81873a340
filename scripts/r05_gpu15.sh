#!/bin/bash
mkdir -p gpurun_out/r05k
echo "--- fused check"; SE3TN_SPLITK_FUSED=1 timeout 120 python scripts/small_kernels_check.py 2>&1 | tail -6
for f in 0 1; do
  echo "--- SE3TN_SPLITK_FUSED=$f"
  SE3TN_SPLITK_FUSED=$f SE3TN_NOCHECK=1 timeout 100 python bench.py --steps 3000 --warmup 50 --batch 1 --no-cpu-baseline --no-parity --track-frames 0 --exact-steps --streams 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench batch 1: ms_per_step', d['ms_per_step'], 'pairs/s', d['value'])"
  SE3TN_SPLITK_FUSED=$f SE3TN_NOCHECK=1 timeout 100 python bench.py --steps 2000 --warmup 50 --batch 4 --no-cpu-baseline --no-parity --track-frames 0 --exact-steps --streams 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench batch 4: ms_per_step', d['ms_per_step'], 'pairs/s', d['value'])"
  SE3TN_SPLITK_FUSED=$f timeout 200 python scripts/track_latency.py 2>&1 | grep on_track | head -2 | cut -c1-150
done
