#!/bin/bash
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -s > gpurun_out/r05/gpu_suite.txt 2>&1
grep -E "passed|failed|FAILED|drop-in|predict_sequence|get_results|with the HIP|image A from" gpurun_out/r05/gpu_suite.txt | tail -30
