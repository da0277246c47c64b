"""Summarise a rocprofv3 kernel trace: per-kernel time per step and the slowest launches."""
import csv, sys, collections, glob
d = sys.argv[1]; steps = int(sys.argv[2])
trace = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(trace)))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:60]
    agg[k][0] += 1; agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"total kernel time {tot/1e3:.2f} ms over {steps} steps = {tot/1e3/steps:.2f} ms/step, {len(rows)/steps:.0f} launches/step")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{v[1]/steps:9.1f} us/step {v[0]/steps:7.1f} calls/step  avg {v[1]/v[0]:8.1f} us  {k}")
