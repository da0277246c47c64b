"""Every launch of the LAST repetition in a rocprofv3 kernel trace, in start order: start offset, duration, grid, name.
usage: prof_seq.py <trace dir> <marker substring> <marker launches per repetition>"""
import csv, glob, sys
d, marker, per = sys.argv[1], sys.argv[2], int(sys.argv[3])
rows = list(csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
seq = rows[idx[-per - 1] + 1:idx[-1] + 1]
t0 = int(rows[idx[-per - 1]]["End_Timestamp"])
for r in seq:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    grid = "x".join(str(int(r.get(k, 0)) // max(int(r.get(w, 1)), 1)) for k, w in
                    (("Grid_Size_X", "Workgroup_Size_X"), ("Grid_Size_Y", "Workgroup_Size_Y"), ("Grid_Size_Z", "Workgroup_Size_Z")))
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  q{r.get('Queue_Id', '?'):>2}  {grid:>14}  "
          f"{r['Kernel_Name'].split('(')[0].replace('void ', '')[:70]}")
