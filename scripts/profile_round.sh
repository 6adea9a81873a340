#!/bin/bash
# Run ON the GPU box (through gpurun): everything profiles/<tag>_* is made of -- batch-64 kernel trace + PMC passes (profile_r.sh), the
# trace split by grid size, batch-1 trace + PMC, batch-1 breakdown, on_track latency, batch sweep, two-lane soak.
#   scripts/profile_round.sh r06   -> gpurun_out/<tag>_*, gpurun_out/prof_<tag>/ ; then scripts/summarize_profile.py gpurun_out/prof_<tag> profiles/<tag>
TAG=${1:-r06}
scripts/profile_r.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1
python scripts/trace_by_grid.py $(find gpurun_out/prof_$TAG/trace -name "*kernel_trace.csv" | head -1) > gpurun_out/${TAG}_kernel_trace_by_grid.txt 2>&1
scripts/profile_b1_pmc.sh > gpurun_out/${TAG}_b1_pmc.txt 2>&1
cp $(find gpurun_out/pmc_b1/trace -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_b1_kernel_stats.csv
python scripts/batch1_breakdown.py > gpurun_out/${TAG}_batch1_breakdown.txt 2>/dev/null
python scripts/track_latency.py > gpurun_out/${TAG}_track_latency.txt 2>/dev/null
scripts/batch_sweep.sh > gpurun_out/${TAG}_batch_sweep.txt 2>&1
( python scripts/soak_pipelined.py 4000 64; python scripts/soak_pipelined.py 12000 1; python scripts/soak_pipelined.py 6000 4 ) > gpurun_out/${TAG}_soak_pipelined.txt 2>&1
tail -3 gpurun_out/${TAG}_profile.log; head -30 gpurun_out/${TAG}_kernel_trace_by_grid.txt; tail -5 gpurun_out/${TAG}_soak_pipelined.txt
rm -rf gpurun_out/prof_$TAG/pmc_*/*/*.db 2>/dev/null; du -sh gpurun_out
