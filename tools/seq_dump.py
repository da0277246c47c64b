"""List the kernels of the LAST occurrence of a repeating launch sequence in a rocprofv3 kernel trace, in launch order:
usage: seq_dump.py <trace dir> <launches per repetition | 0 = guess from the first kernel name> [first kernel substring]"""
import csv, glob, sys
d = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
key = sys.argv[3] if len(sys.argv) > 3 else None
n = int(sys.argv[2])
if n == 0 and key:
    idx = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
    starts = idx[-2:]
    seq = rows[starts[0]:starts[1]]
else:
    seq = rows[-n:]
t0 = int(seq[0]["Start_Timestamp"])
tot = 0.0
for r in seq:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    tot += (e - s) / 1e3
    g = [int(r[k]) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")] if "Grid_Size_X" in r else []
    w = [int(r[k]) for k in ("Workgroup_Size_X",)] if "Workgroup_Size_X" in r else [1]
    nb = (g[0] // max(w[0], 1)) * g[1] * g[2] if g else 0
    print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:7.1f} us  blocks {nb:6d}  {r['Kernel_Name'].split('(')[0][:70]}")
print(f"span {(int(seq[-1]['End_Timestamp']) - t0) / 1e3:.1f} us, kernel sum {tot:.1f} us, {len(seq)} launches")
