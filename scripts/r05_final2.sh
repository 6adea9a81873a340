#!/bin/bash
# round-5 evidence on the FINAL binary (after the batch 1-5 kernel family): full GPU suite, soak, batch-1 kernel trace + timelines,
# small-batch sweep, default bench line
mkdir -p gpurun_out/r05z
export TMPDIR=/tmp
timeout 700 python -m pytest tests -m gpu -q > gpurun_out/r05z/gpu_suite.txt 2>&1; tail -3 gpurun_out/r05z/gpu_suite.txt
( timeout 200 python scripts/soak_pipelined.py 12000 64; echo "# batch 1 / 2 / 4, two lanes (the batch 1-5 kernel family: stem + pool tiles, conv64_small, conv_slices_small, the 16-workgroup tail):"; timeout 100 python scripts/soak_pipelined.py 12000 1; timeout 100 python scripts/soak_pipelined.py 6000 2; timeout 100 python scripts/soak_pipelined.py 6000 4 ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r05z/soak_pipelined.txt; cat gpurun_out/r05z/soak_pipelined.txt
NOALT=1 timeout 200 bash scripts/ktrace.sh r05z_b1 --batch 1 > gpurun_out/r05z/ktrace_b1.txt 2>&1; head -16 gpurun_out/r05z/ktrace_b1.txt
timeout 120 python scripts/batch1_breakdown.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05z/batch1_breakdown.txt
timeout 200 python scripts/track_latency.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05z/track_latency.txt; grep on_track gpurun_out/r05z/track_latency.txt | cut -c1-120
SE3TN_TRACK_TRACE=1 timeout 200 python scripts/track_latency.py 2>&1 | grep timeline | head -3 > gpurun_out/r05z/track_trace.txt; cat gpurun_out/r05z/track_trace.txt
for b in 1 2 3 4 5 6 8 64; do
  SE3TN_NO_ALT=1 python bench.py --no-parity --track-frames 0 --exact-steps --no-cpu-baseline --batch $b --steps 300 --precision f32 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-6s n=%-4d %9.1f pairs/s pipelined  %9.1f single-stream  %8.4f ms/step single' % ('f32', $b, d['value'], d['single_stream']['value'], d['single_stream']['ms_per_step']))"
done > gpurun_out/r05z/batch_sweep_small.txt 2>&1; cat gpurun_out/r05z/batch_sweep_small.txt
( time python bench.py ) > gpurun_out/r05z/bench_default.json 2> gpurun_out/r05z/bench_default.err; tail -4 gpurun_out/r05z/bench_default.err; cut -c1-400 gpurun_out/r05z/bench_default.json
