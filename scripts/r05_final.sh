#!/bin/bash
# round-5 evidence run on the FINAL binary: full GPU test suite, two-lane soak, kernel trace + PMC passes, default bench line,
# batch sweep, batch-1 kernel trace and per-launch breakdown, on_track latency
mkdir -p gpurun_out/r05f
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r05f/gpu_suite.txt 2>&1; tail -3 gpurun_out/r05f/gpu_suite.txt
timeout 600 python scripts/soak_pipelined.py 12000 64 > gpurun_out/r05f/soak_pipelined.txt 2>&1; cat gpurun_out/r05f/soak_pipelined.txt | tail -5
timeout 100 python scripts/soak_pipelined.py 4000 1 > gpurun_out/r05f/soak_pipelined_b1.txt 2>&1; tail -4 gpurun_out/r05f/soak_pipelined_b1.txt
timeout 900 bash scripts/profile_r.sh r05 > gpurun_out/r05f/profile.log 2>&1; tail -2 gpurun_out/r05f/profile.log
( time python bench.py ) > gpurun_out/r05f/bench_default.json 2> gpurun_out/r05f/bench_default.err; tail -3 gpurun_out/r05f/bench_default.err
timeout 600 bash scripts/batch_sweep.sh > gpurun_out/r05f/batch_sweep.txt 2>&1; cat gpurun_out/r05f/batch_sweep.txt
NOALT=1 timeout 200 bash scripts/ktrace.sh r05_b1 --batch 1 > gpurun_out/r05f/ktrace_b1.txt 2>&1; head -24 gpurun_out/r05f/ktrace_b1.txt
timeout 120 python scripts/batch1_breakdown.py > gpurun_out/r05f/batch1_breakdown.txt 2>&1
timeout 200 python scripts/track_latency.py > gpurun_out/r05f/track_latency.txt 2>&1; grep on_track gpurun_out/r05f/track_latency.txt
SE3TN_TRACK_TRACE=1 timeout 200 python scripts/track_latency.py 2>&1 | grep timeline | head -3 > gpurun_out/r05f/track_trace.txt; cat gpurun_out/r05f/track_trace.txt
