#!/bin/bash
timeout 100 python scripts/small_kernels_check.py 2>&1 | tail -5
timeout 100 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -2
SE3TN_LIB=/root/repo/variants/lib_trace.so timeout 100 python scripts/small_trace.py 2>&1 | tail -9
for b in 1 2; do SE3TN_NOCHECK=1 timeout 100 python bench.py --steps 3000 --warmup 50 --batch $b --no-cpu-baseline --no-parity --track-frames 0 --exact-steps --streams 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench batch $b: ms_per_step', d['ms_per_step'], 'pairs/s', d['value'])"; done
