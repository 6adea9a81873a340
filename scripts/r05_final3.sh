#!/bin/bash
# round-5 evidence on the FINAL binary (after the stem rewrite): full GPU suite, small-batch soak, batch-1 kernel trace + timelines, small-batch sweep
mkdir -p gpurun_out/r05y
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q > gpurun_out/r05y/gpu_suite.txt 2>&1; tail -3 gpurun_out/r05y/gpu_suite.txt
( timeout 100 python scripts/soak_pipelined.py 12000 1; timeout 100 python scripts/soak_pipelined.py 6000 2; timeout 100 python scripts/soak_pipelined.py 4000 4 ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r05y/soak_small.txt; cat gpurun_out/r05y/soak_small.txt
NOALT=1 timeout 200 bash scripts/ktrace.sh r05y_b1 --batch 1 > gpurun_out/r05y/ktrace_b1.txt 2>&1; head -14 gpurun_out/r05y/ktrace_b1.txt
timeout 120 python scripts/batch1_breakdown.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05y/batch1_breakdown.txt; head -16 gpurun_out/r05y/batch1_breakdown.txt
timeout 200 python scripts/track_latency.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05y/track_latency.txt; grep on_track gpurun_out/r05y/track_latency.txt | cut -c1-120
for b in 1 2 3 4 5; do
  SE3TN_NO_ALT=1 python bench.py --no-parity --track-frames 0 --exact-steps --no-cpu-baseline --batch $b --steps 300 --precision f32 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-6s n=%-4d %9.1f pairs/s pipelined  %9.1f single-stream  %8.4f ms/step single' % ('f32', $b, d['value'], d['single_stream']['value'], d['single_stream']['ms_per_step']))"
done > gpurun_out/r05y/batch_sweep_small.txt 2>&1; cat gpurun_out/r05y/batch_sweep_small.txt
