#!/bin/bash
python -m pytest tests/test_free_run.py tests/test_synth_track.py tests/test_small_batch_kernels.py tests/test_c_host.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -15
python scripts/free_run_report.py 1000 3 > gpurun_out/r06_free_run.json 2> gpurun_out/r06_free_run.err; tail -2 gpurun_out/r06_free_run.err
for tp in 1 0; do echo "# SE3TN_TAIL_PARTS=$tp"; SE3TN_TAIL_PARTS=$tp python scripts/batch1_breakdown.py 2>/dev/null | head -18; SE3TN_TAIL_PARTS=$tp python scripts/track_latency.py 2>/dev/null | tail -3; done > gpurun_out/r06_tail_parts_ab.txt 2>&1
cat gpurun_out/r06_tail_parts_ab.txt
