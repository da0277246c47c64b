#!/bin/bash
run() { timeout -k 10 200 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-infer 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed -n 1p; }
echo "default $(run)"
for t in 320 384 448; do for m in 512 768 1024 2048; do export RIED_SPLITK_TARGET=$t RIED_SPLITK_MINPIX=$m; echo "target $t minpix $m $(run)"; done; done
unset RIED_SPLITK_TARGET RIED_SPLITK_MINPIX
echo "default $(run)"
