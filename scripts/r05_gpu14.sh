#!/bin/bash
mkdir -p gpurun_out/r05j
timeout 120 python scripts/small_kernels_check.py 2>&1 | tail -6
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
NOALT=1 timeout 200 bash scripts/ktrace.sh r05j_b1 --batch 1 > gpurun_out/r05j/ktrace_b1.txt 2>&1; head -14 gpurun_out/r05j/ktrace_b1.txt; cat gpurun_out/ktrace_r05j_b1/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
timeout 200 python scripts/track_latency.py > gpurun_out/r05j/track_latency.txt 2>&1; grep on_track gpurun_out/r05j/track_latency.txt
