#!/bin/bash
# Inference batch (32 slices of 512x512) under the folded-statistics threshold and the branch-concurrency switch.
run() { timeout -k 10 150 python tools/infer_time.py 2>/dev/null | tail -n 1; }
export TAG=default; run
for v in 16777216 67108864 268435456; do export RIED_BN_FOLD_MAX_ELEMS=$v TAG="fold_max_$v"; run; done
unset RIED_BN_FOLD_MAX_ELEMS
export RIED_SERIAL_BRANCHES=1 TAG=serial_branches; run; unset RIED_SERIAL_BRANCHES
export RIED_NO_SIBLING_GROUPS=1 TAG=no_groups; run; unset RIED_NO_SIBLING_GROUPS
export TAG=default; run
