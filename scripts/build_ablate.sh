#!/bin/bash
# Timing-only / tuning builds of the conv_split kernels for scripts/ablate_split.py: scripts/ablate/lib_<NAME>.so
#   scripts/build_ablate.sh "NAME:flags" ...      (run in the build container; the .so files travel with gpurun)
# The flags go to conv_split.hip and to the instantiation unit of the arithmetic under test (AB_NS = 2 | 3, default 2).
set -e
cd "$(dirname "$0")/../nndepth_amd/csrc"
make -j8 >/dev/null
mkdir -p ../../scripts/ablate
NS=${AB_NS:-2}
OTHER_NS=$((5 - NS))
OTHERS="conv_mfma.o conv_split_ns${OTHER_NS}.o corr1d.o agcl.o mask_upsample.o update_block.o encoder.o prepost.o cascade.o loftr.o conv3d.o thin3d.o slab3d.o calib.o error.o"
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  ( hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-slp-vectorize $flags -c conv_split.hip -o /tmp/cs_$name.o &&
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $flags -c conv_split_ns${NS}.hip -o /tmp/csn_$name.o &&
    hipcc -shared -fPIC --offload-arch=gfx950 -o ../../scripts/ablate/lib_$name.so /tmp/cs_$name.o /tmp/csn_$name.o $OTHERS && echo "$name ok" ) &
done
wait
