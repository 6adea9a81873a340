#!/bin/bash
# usage: build_variant.sh <name> "<extra hipcc flags>"   -> /root/repo/variants/lib_<name>.so
set -e
NAME=$1; FLAGS=$2
SRC=/root/repo/iros20-6d-pose-tracking_amd/csrc
OUT=/tmp/variants/$NAME
mkdir -p $OUT /root/repo/variants
cd $SRC
for f in api.cpp weights.cpp; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result $FLAGS -x hip -c $f -o $OUT/$f.o &
done
NOPK="-Xclang -target-feature -Xclang -packed-fp32-ops"   # as in the Makefile
for f in conv3x3_mfma.hip conv64_small.hip conv_slices_small.hip stem_pool_small.hip wino_mfma.hip wino64_fused.hip stem7x7_mfma.hip kernels_misc.hip raster.hip depth_fill.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result $NOPK $FLAGS -c $f -o $OUT/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o /root/repo/variants/lib_$NAME.so $OUT/*.o
ls -la /root/repo/variants/lib_$NAME.so
