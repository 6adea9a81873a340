#!/bin/bash
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_tracker_surface.py tests/test_gl_swiftshader.py tests/test_renderer.py tests/test_gpu_parity.py tests/test_closed_loop.py -m gpu -x -q > gpurun_out/r05/tests7.txt 2>&1; tail -5 gpurun_out/r05/tests7.txt
timeout 120 python scripts/batch1_breakdown.py > gpurun_out/r05/batch1_r05.txt 2>&1; grep -A16 "n = 1" gpurun_out/r05/batch1_r05.txt | head -18; grep total gpurun_out/r05/batch1_r05.txt
timeout 200 python scripts/track_latency.py > gpurun_out/r05/track_latency_r05.txt 2>&1; grep on_track gpurun_out/r05/track_latency_r05.txt
timeout 100 python scripts/graph_latency.py 2>&1 | tail -4
