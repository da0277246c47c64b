#!/bin/bash
# Two independent single-rank benches sharing the one card (no torch.distributed): does the step run side by side?
timeout -k 10 100 python bench.py --steps 10 --warmup 3 --no-infer --no-cpu-baseline --no-graph > gpurun_out/two_a.log 2>&1 &
A=$!
timeout -k 10 100 python bench.py --steps 10 --warmup 3 --no-infer --no-cpu-baseline --no-graph > gpurun_out/two_b.log 2>&1 &
B=$!
wait $A; ra=$?; wait $B; rb=$?
echo "exit codes $ra $rb"
grep -o '"ms_per_step": [0-9.]*' gpurun_out/two_a.log gpurun_out/two_b.log
