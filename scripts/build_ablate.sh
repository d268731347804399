#!/bin/bash
# Timing-only / tuning builds of conv_split.hip for scripts/ablate_split.py: scripts/ablate/lib_<NAME>.so
#   scripts/build_ablate.sh "NAME:flags" ...      (run in the build container; the .so files travel with gpurun)
set -e
cd "$(dirname "$0")/../nndepth_amd/csrc"
make -j8 >/dev/null
mkdir -p ../../scripts/ablate
OTHERS="conv_mfma.o corr1d.o agcl.o mask_upsample.o update_block.o encoder.o prepost.o cascade.o loftr.o conv3d.o thin3d.o error.o"
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  ( hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $flags -c conv_split.hip -o /tmp/cs_$name.o &&
    hipcc -shared -fPIC --offload-arch=gfx950 -o ../../scripts/ablate/lib_$name.so /tmp/cs_$name.o $OTHERS && echo "$name ok" ) &
done
wait
