#!/bin/bash
# round-5 GPU call 2: renderer-inclusive closed loops, reference-driver goldens end to end, default bench line
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
python -m pytest tests/test_ycbv_drivers.py tests/test_predict_tracker_golden.py tests/test_driver_vs_reference.py tests/test_closed_loop.py tests/test_renderer.py tests/test_gl_swiftshader.py -m gpu -x -q -s > gpurun_out/r05/golden_tests.txt 2>&1
grep -v "^$" gpurun_out/r05/golden_tests.txt | tail -25
( time python bench.py ) > gpurun_out/r05/bench_default.json 2> gpurun_out/r05/bench_default.err; tail -3 gpurun_out/r05/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/bench_default.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'])
t=d['track']; print({k:t[k] for k in t if k!='regimes'})
PY
