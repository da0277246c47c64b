"""Per-kernel summary of the LAST repetition in a rocprofv3 kernel trace (CSV): the launches after the (per_rep+1)-th
last occurrence of a marker kernel up to and including the last occurrence.
usage: prof_last.py <trace dir> <marker substring> <marker launches per repetition> [rows=45]
  training step (tools/step_prof.py):   prof_last.py DIR adam_kernel 3
  inference batch (tools/infer_prof.py): prof_last.py DIR ssim_fwd_kernel 1"""
import csv, glob, sys, collections
d, marker, per = sys.argv[1], sys.argv[2], int(sys.argv[3])
nrows = int(sys.argv[4]) if len(sys.argv) > 4 else 45
rows = list(csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
seq = rows[idx[-per - 1] + 1:idx[-1] + 1]
t0 = int(rows[idx[-per - 1]]["End_Timestamp"])
span = (max(int(r["End_Timestamp"]) for r in seq) - t0) / 1e3
agg = collections.defaultdict(lambda: [0, 0.0])
for r in seq:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:64]
    agg[k][0] += 1; agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"last repetition: {len(seq)} launches, kernel time {tot / 1e3:.3f} ms, span {span / 1e3:.3f} ms (profiled)")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:nrows]:
    print(f"{v[1]:9.1f} us {v[0]:5d} launches  avg {v[1] / v[0]:8.1f} us  {k}")
