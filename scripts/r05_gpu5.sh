#!/bin/bash
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
timeout 600 python scripts/splitk_fused_check.py > gpurun_out/r05/splitk_fused_check.txt 2>&1; echo "check rc=$?"; tail -1 gpurun_out/r05/splitk_fused_check.txt
SE3TN_SPLITK_FUSED=1 timeout 120 python scripts/batch1_breakdown.py > gpurun_out/r05/batch1_fused.txt 2>&1; grep -A16 "n = 1" gpurun_out/r05/batch1_fused.txt | head -18; grep "total" gpurun_out/r05/batch1_fused.txt
SE3TN_SPLITK_FUSED=0 timeout 120 python scripts/batch1_breakdown.py > gpurun_out/r05/batch1_unfused.txt 2>&1; grep "total" gpurun_out/r05/batch1_unfused.txt
SE3TN_SPLITK_FUSED=1 timeout 200 python scripts/track_latency.py > gpurun_out/r05/track_latency_fused.txt 2>&1; cat gpurun_out/r05/track_latency_fused.txt | grep on_track
