#!/bin/bash
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
M=iros20-6d-pose-tracking_amd/libse3tracknet.so
for S in 1 2; do
  for R in 1 2; do
    for L in $M variants/lib_xcdpix.so variants/lib_nopool.so; do
      SE3TN_NO_ALT=1 SE3TN_NOCHECK=1 SE3TN_LIB=$L python bench.py --no-cpu-baseline --no-parity --track-frames 0 --exact-steps --streams $S --steps 60 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('streams $S', '%-22s'%'$L'[-22:], d['value'], d['roofline']['achieved'], ' '.join('%.3f'%v for v in d['layers_ms'].values()))"
    done
  done
done > gpurun_out/r05/ab_xcdpix_nopool.txt 2>&1
cat gpurun_out/r05/ab_xcdpix_nopool.txt
