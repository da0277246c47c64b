"""Time the 512x512 inference batch (bench.infer_bench) - for A/B runs under environment switches."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
r = bench.infer_bench(steps=20, warmup=3)
m = r["metrics_kernel"]
print(os.environ.get("TAG", ""), round(r["value"], 1), "slices/s", round(r["ms_per_batch"], 3), "ms; metrics kernel",
      round(m["ms"] * 1e3, 1), "us =", round(m["frac"], 3), "of HBM peak")
