#!/bin/bash
# A/B sweep of the engine's environment knobs on the training step (ms per step of each setting).
run() { timeout -k 10 200 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-infer 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed -n 1p; }
echo "default $(run)"
for v in 1048576 2097152 8388608 16777216; do export RIED_BN_FOLD_MAX_ELEMS=$v; echo "fold_max $v $(run)"; done; unset RIED_BN_FOLD_MAX_ELEMS
for v in 256 512 768; do export RIED_SPLITK_TARGET=$v; echo "splitk_target $v $(run)"; done; unset RIED_SPLITK_TARGET
for v in 128 512; do export RIED_SPLITK_MINPIX=$v; echo "splitk_minpix $v $(run)"; done; unset RIED_SPLITK_MINPIX
export RIED_NO_OVERLAP_VGG=1; echo "no_overlap_vgg $(run)"; unset RIED_NO_OVERLAP_VGG
export RIED_NO_OVERLAP_GBWD=1; echo "no_overlap_gbwd $(run)"; unset RIED_NO_OVERLAP_GBWD
echo "default $(run)"
