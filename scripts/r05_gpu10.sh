#!/bin/bash
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
SE3TN_TRACK_TRACE=1 timeout 200 python scripts/track_latency.py > gpurun_out/r05/track_trace.txt 2>&1; grep -E "timeline|on_track" gpurun_out/r05/track_trace.txt | head -12
