#!/bin/bash
# round-6 final profiles: batch-64 kernel trace + PMC passes (profile_r.sh), per-grid breakdown, batch-1 trace + PMC, batch sweep, soak
scripts/profile_r.sh r06 > gpurun_out/r06_profile.log 2>&1
python scripts/trace_by_grid.py $(find gpurun_out/prof_r06/trace -name "*kernel_trace.csv" | head -1) > gpurun_out/r06_kernel_trace_by_grid.txt 2>&1
scripts/profile_b1_pmc.sh > gpurun_out/r06_b1_pmc.txt 2>&1
cp $(find gpurun_out/pmc_b1/trace -name "*kernel_stats.csv" | head -1) gpurun_out/r06_b1_kernel_stats.csv
python scripts/batch1_breakdown.py > gpurun_out/r06_batch1_breakdown.txt 2>/dev/null
python scripts/track_latency.py > gpurun_out/r06_track_latency.txt 2>/dev/null
scripts/batch_sweep.sh > gpurun_out/r06_batch_sweep.txt 2>&1
( python scripts/soak_pipelined.py 4000 64; python scripts/soak_pipelined.py 12000 1; python scripts/soak_pipelined.py 6000 4 ) > gpurun_out/r06_soak_pipelined.txt 2>&1
tail -3 gpurun_out/r06_profile.log; head -30 gpurun_out/r06_kernel_trace_by_grid.txt; tail -5 gpurun_out/r06_soak_pipelined.txt
rm -rf gpurun_out/prof_r06/pmc_*/*/*.db 2>/dev/null; du -sh gpurun_out
