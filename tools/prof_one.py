"""Kernel-time table of ONE graph replay of a part (G / D / V) from a rocprofv3 kernel trace of tools/time_parts.py:
the last replay's kernels, in launch order, grouped by name."""
import csv, sys, glob, collections
d = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
try:
    nlast = next(p for p in range(8, len(names) // 3) if names[-p:] == names[-2 * p:-p] == names[-3 * p:-2 * p])
    rows = rows[-nlast:]                  # the last graph replay: the trace ends with identical replays
except StopIteration:                     # concurrent streams reorder launches: a full step ends with its third Adam
    adam = [i for i, n in enumerate(names) if n.startswith("adam_kernel")]
    rows = rows[adam[-4] + 1:adam[-1] + 1]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:50]
    agg[k][0] += 1; agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
print(f"{len(rows)} kernels, sum {sum(v[1] for v in agg.values()):.0f} us, span {span:.0f} us")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"{v[1]:8.1f} us  x{v[0]:<4d} avg {v[1]/v[0]:7.1f}  {k}")
if len(sys.argv) > 2:                     # also list every launch whose name contains the given text
    for r in rows:
        if sys.argv[2] in r["Kernel_Name"]:
            print(f'{(int(r["Start_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3:9.1f} us  '
                  f'{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:7.1f} us  grid {r.get("Grid_Size", "?")}')
