#!/bin/bash
mkdir -p gpurun_out/r05h
NOALT=1 timeout 200 bash scripts/ktrace.sh r05h_b1 --batch 1 > gpurun_out/r05h/ktrace_b1.txt 2>&1; head -22 gpurun_out/r05h/ktrace_b1.txt; cat gpurun_out/ktrace_r05h_b1/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
NOALT=1 timeout 200 bash scripts/ktrace.sh r05h_b4 --batch 4 > gpurun_out/r05h/ktrace_b4.txt 2>&1; head -22 gpurun_out/r05h/ktrace_b4.txt; cat gpurun_out/ktrace_r05h_b4/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
timeout 200 python scripts/track_latency.py > gpurun_out/r05h/track_latency.txt 2>&1; grep on_track gpurun_out/r05h/track_latency.txt
