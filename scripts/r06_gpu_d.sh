#!/bin/bash
python -m pytest tests/test_free_run.py tests/test_gpu_edge_cases.py tests/test_c_host.py -m gpu -q 2>&1 | tail -6
python scripts/free_run_report.py 1000 3 > gpurun_out/r06_free_run.json 2> gpurun_out/r06_free_run.err; tail -2 gpurun_out/r06_free_run.err
for th in 1 4; do echo "# SE3TN_STAGE_THREADS=$th"; SE3TN_STAGE_THREADS=$th SE3TN_TRACK_TRACE=1 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys, os
sys.path.insert(0, os.getcwd())
import se3tracknet_amd as se3
from oracle import closed_loop
for n in (8, 21, 64):
    r = closed_loop.time_batch(se3, n, frames=100, check_frames=0)
    print(n, r["ms_per_step_median"], r["pairs_per_s"])
PY
done > gpurun_out/r06_tracker_batch_staging.txt 2>&1
cat gpurun_out/r06_tracker_batch_staging.txt
