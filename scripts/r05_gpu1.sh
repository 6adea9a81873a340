#!/bin/bash
# round-5 GPU call 1: rasteriser byte-parity tests, batch-1 evidence, res-hoist variant A/B
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
python -m pytest tests/test_gl_swiftshader.py tests/test_renderer.py -m gpu -x -q > gpurun_out/r05/raster_tests.txt 2>&1
tail -5 gpurun_out/r05/raster_tests.txt
python scripts/track_latency.py > gpurun_out/r05/track_latency.txt 2>&1; cat gpurun_out/r05/track_latency.txt
python scripts/batch1_breakdown.py > gpurun_out/r05/batch1_breakdown.txt 2>&1; tail -60 gpurun_out/r05/batch1_breakdown.txt
NOALT=1 bash scripts/ktrace.sh r05_b1 --batch 1 > gpurun_out/r05/ktrace_b1.txt 2>&1; head -40 gpurun_out/r05/ktrace_b1.txt
bash scripts/variants_bench.sh iros20-6d-pose-tracking_amd/libse3tracknet.so variants/lib_reshoist.so iros20-6d-pose-tracking_amd/libse3tracknet.so variants/lib_reshoist.so > gpurun_out/r05/reshoist_ab.txt 2>&1; cat gpurun_out/r05/reshoist_ab.txt
