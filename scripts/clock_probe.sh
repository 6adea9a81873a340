#!/bin/bash
# What core clock does the chip run at in the batch-1 latency regime (a stream of 5-15 us kernels) vs batch 64?
ls /sys/class/drm/ | head; for c in /sys/class/drm/card*/device/pp_dpm_sclk; do echo $c; cat $c; done 2>/dev/null | head -30
rocm-smi --showclocks 2>/dev/null | head -20
rocm-smi --showperflevel 2>/dev/null | head -8
sample() { for i in $(seq 1 30); do for c in /sys/class/drm/card*/device/pp_dpm_sclk; do grep '\*' $c | tr '\n' ' '; done; echo; sleep 0.1; done; }
echo "--- batch 1"; (python bench.py --steps 20000 --warmup 50 --batch 1 --no-cpu-baseline --no-parity --track-frames 0 --streams 1 > /tmp/b1.json 2>/dev/null &) ; sleep 12; sample | sort | uniq -c; wait; tail -c 300 /tmp/b1.json
echo "--- batch 64"; (python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-parity --track-frames 0 > /tmp/b64.json 2>/dev/null &) ; sleep 12; sample | sort | uniq -c; wait
