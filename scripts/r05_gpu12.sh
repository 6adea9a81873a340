#!/bin/bash
mkdir -p gpurun_out/r05h
timeout 120 python scripts/small_kernels_check.py 2>&1 | tail -8
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -5
timeout 120 python scripts/batch1_breakdown.py > gpurun_out/r05h/batch1_breakdown.txt 2>&1; cat gpurun_out/r05h/batch1_breakdown.txt
