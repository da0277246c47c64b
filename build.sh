#!/bin/bash
# Build the gfx950 shared library in-tree: reid_gan_amd/libried_gan.so
set -e
cd "$(dirname "$0")/reid_gan_amd/csrc"
OUT=../libried_gan.so
FLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-result"
objs=()
pids=()
for f in api gemm elementwise special ssim loader legacy attn; do
  [ -f $f.hip ] || continue
  stale=0
  [ -f $f.o ] || stale=1
  for dep in $f.hip *.h; do
    [ $stale = 1 ] || { [ $dep -nt $f.o ] && stale=1; } || true
  done
  if [ $stale = 1 ]; then
    echo "hipcc $f.hip"
    rm -f $f.o
    hipcc $FLAGS -c $f.hip -o $f.o &
    pids+=($!)
  fi
  objs+=($f.o)
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT "${objs[@]}"
echo "built $OUT"
