#!/bin/bash
# Run ON the GPU box: L2 (TCC) hit / miss / DRAM-read counters per kernel for bench.py's step.
TAG=${1:-r01}
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_${TAG}_l2
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline"
cd /tmp
SE3TN_NO_ALT=1 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $OUT/tcc -o pmc -- $CMD > /dev/null 2> $OUT/tcc.err
SE3TN_NO_ALT=1 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $OUT/ea -o pmc -- $CMD > /dev/null 2> $OUT/ea.err
find $OUT -name "*.csv" | wc -l
