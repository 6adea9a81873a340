#!/bin/bash
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_batch_sizes.py tests/test_gpu_parity.py tests/test_tracker_surface.py tests/test_media_pair.py -m gpu -x -q > gpurun_out/r05/tests11.txt 2>&1; tail -4 gpurun_out/r05/tests11.txt
timeout 120 python scripts/batch1_breakdown.py > gpurun_out/r05/batch1_small.txt 2>&1; grep -A16 "n = 1" gpurun_out/r05/batch1_small.txt | head -18; grep total gpurun_out/r05/batch1_small.txt
timeout 200 python scripts/track_latency.py > gpurun_out/r05/track_latency_small.txt 2>&1; grep on_track gpurun_out/r05/track_latency_small.txt
