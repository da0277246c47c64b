#!/bin/bash
# Build the gfx950 shared library in-tree: reid_gan_amd/libried_gan.so
set -e
cd "$(dirname "$0")/reid_gan_amd/csrc"
OUT=../libried_gan.so
FLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-result"
objs=()
for f in api gemm elementwise special ssim loader legacy; do
  if [ ! -f $f.o ] || [ $f.hip -nt $f.o ] || [ common.h -nt $f.o ]; then
    echo "hipcc $f.hip"
    hipcc $FLAGS -c $f.hip -o $f.o &
  fi
  objs+=($f.o)
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT "${objs[@]}"
echo "built $OUT"
