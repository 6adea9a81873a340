#!/bin/bash
for v in "" /root/repo/variants/lib_c64n5.so; do
  echo "--- lib: ${v:-default}"
  for b in 3 4 5; do SE3TN_LIB=$v SE3TN_NOCHECK=1 timeout 100 python bench.py --steps 2000 --warmup 50 --batch $b --no-cpu-baseline --no-parity --track-frames 0 --exact-steps --streams 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench batch $b: ms_per_step', d['ms_per_step'], 'pairs/s', d['value'])"; done
done
SE3TN_LIB=/root/repo/variants/lib_c64n5.so timeout 100 python scripts/small_kernels_check.py 2>&1 | tail -5
