"""Print the headline fields of a bench.py JSON line.  usage: bench_summary.py <file with the line>"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling",
                         "vs_baseline", "dtype", "data")})
r = d.get("roofline")
if r:
    print({k: r[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch",
                             "launches_per_step")})
    print("families", r["families"])
    for t in r.get("templates", [])[:12]:
        print("  ", t)
print("cpu_baseline", d.get("cpu_baseline"))
if "infer" in d:
    print("infer", {k: v for k, v in d["infer"].items() if not isinstance(v, dict)})
    print("  metrics_kernel", d["infer"].get("metrics_kernel"), "per_slice", d["infer"].get("per_slice_statistics"))
print("step_mfma_frac", d.get("step_mfma_frac"))
for k in ("loader", "legacy", "prep", "vgg256"):
    if k in d:
        print(k, json.dumps(d[k])[:300])
