#!/bin/bash
python -m pytest tests/test_free_run.py tests/test_small_batch_kernels.py tests/test_c_host.py tests/test_gpu_parity.py tests/test_tracker_surface.py -m gpu -q 2>&1 | tail -12
for tp in 1 2 0; do echo "# SE3TN_TAIL_PARTS=$tp"; SE3TN_TAIL_PARTS=$tp python scripts/batch1_breakdown.py 2>/dev/null | head -16 | tail -5; SE3TN_TAIL_PARTS=$tp python scripts/track_latency.py 2>/dev/null | tail -2 | head -1; done > gpurun_out/r06_tail_parts_ab2.txt 2>&1
cat gpurun_out/r06_tail_parts_ab2.txt
