#!/bin/bash
# Run ON the GPU box (through gpurun): rocprofv3 kernel-trace stats + PMC passes of bench.py.
# usage: scripts/profile_r.sh <round-tag>      -> gpurun_out/prof_<tag>/...
TAG=${1:-r01}
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --track-frames 0 --exact-steps --streams 1"
cd /tmp
SE3TN_NO_ALT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/bench_traced.json 2> $OUT/trace.err
# the PMC passes run the default float32 configuration only (SE3TN_NO_ALT: no f16x3 / direct-kernel legs): per-step byte totals are
# then simply (sum over the dispatches) / (steps)
export SE3TN_NO_ALT=1
CMD5="python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-parity --track-frames 0 --exact-steps --streams 1"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -o pmc -- $CMD5 > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_lds -o pmc -- $CMD5 > /dev/null 2> $OUT/pmc_lds.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o pmc -- $CMD5 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o pmc -- $CMD5 > /dev/null 2> $OUT/pmc_write.err
find $OUT -name "*.csv" | wc -l
du -sh $OUT
