#!/bin/bash
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
timeout 300 python scripts/on_track_breakdown.py > gpurun_out/r05/on_track_breakdown.txt 2>&1; cat gpurun_out/r05/on_track_breakdown.txt | grep us
timeout 120 python scripts/graph_latency.py > gpurun_out/r05/graph_latency.txt 2>&1; tail -6 gpurun_out/r05/graph_latency.txt
