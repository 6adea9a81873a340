#!/bin/bash
# scripts/env_ab.sh "NAME=VAL ..." "NAME=VAL ..." ... : one single-stream bench line per environment setting (developer switches of the
# library: SE3TN_WINOGRAD_FUSE, SE3TN_WINO_GEMMP, SE3TN_TRUNK_WINOGRAD, ...), timing only: value, executed TF, per-launch ms.
# Run ON the GPU box, all settings in ONE session (clocks / tenants differ between sessions).
for E in "$@"; do
  env $E SE3TN_NOCHECK=1 SE3TN_NO_ALT=1 python bench.py --no-cpu-baseline --no-parity --track-frames 0 --steps ${STEPS:-200} --exact-steps --streams 1 $BENCH_ARGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-44s'%'$E', d['value'], d['ms_per_step'], d['roofline']['achieved'], ' '.join('%s=%.4f'%(k.split(' [')[0][-22:],v) for k,v in d['layers_ms'].items()))"
done
