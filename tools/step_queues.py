"""Per-launch listing of the LAST training step in a rocprofv3 kernel trace (see step_timeline.py) with the hardware
queue and stream of every kernel.  usage: step_queues.py <trace dir>"""
import csv, glob, sys
d = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam_kernel")]
seq = rows[adam[-4] + 1:adam[-1] + 1]
t0 = int(rows[adam[-4]]["End_Timestamp"])
print("columns:", [k for k in rows[0].keys()])
for r in seq:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{s:8.1f} +{e - s:6.1f}  q{r.get('Queue_Id', '?'):>2} s{r.get('Stream_Id', '?'):>3}  "
          f"{r['Kernel_Name'].split('(')[0].replace('void ', '')[:60]}")
