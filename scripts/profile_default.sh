#!/bin/bash
# Run ON the GPU box (through gpurun): rocprofv3 kernel-trace stats of the DRIVER's bench command
# (`python bench.py --gpus 1 --steps 20 --warmup 5`: single-stream region + pipelined region + the alt legs).
# usage: scripts/profile_default.sh <tag>      -> gpurun_out/prof_<tag>/{trace,bench_traced.json,cmd.txt}
TAG=${1:-r02c}
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5"
echo "python bench.py --gpus 1 --steps 20 --warmup 5" > $OUT/cmd.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/bench_traced.json 2> $OUT/trace.err
ls $OUT/trace | head
