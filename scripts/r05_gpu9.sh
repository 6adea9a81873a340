#!/bin/bash
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_tracker_surface.py tests/test_closed_loop.py tests/test_predict_tracker_golden.py tests/test_media_pair.py -m gpu -x -q > gpurun_out/r05/tests9.txt 2>&1; tail -4 gpurun_out/r05/tests9.txt
timeout 200 python scripts/track_latency.py > gpurun_out/r05/track_latency_r05b.txt 2>&1; grep on_track gpurun_out/r05/track_latency_r05b.txt
