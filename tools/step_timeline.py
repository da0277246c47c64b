"""Timeline of the LAST training step in a rocprofv3 kernel trace of tools/step_prof.py (graph replays).
usage: step_timeline.py <trace dir> [bucket_us=100] [marker=adam_kernel] [markers per repetition=3]
A step ends with the generator's Adam launch (the third adam_kernel of the step); for the inference batch of
tools/infer_prof.py the marker is ssim_fwd_kernel, once per repetition.  Prints, per time bucket, how much
of it had 0 / 1 / 2+ kernels running and which kernels took the most time in it; then the totals."""
import csv, glob, sys, collections
d = sys.argv[1]
bucket = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
rows = list(csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[3] if len(sys.argv) > 3 else "adam_kernel"
per = int(sys.argv[4]) if len(sys.argv) > 4 else 3
adam = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
lo, hi = adam[-per - 1] + 1, adam[-1] + 1
seq = rows[lo:hi]
t0 = int(rows[adam[-per - 1]]["End_Timestamp"])
ev = []
for r in seq:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    ev.append((s, e, r["Kernel_Name"].split("(")[0].replace("void ", "")[:44], r.get("Queue_Id", "?")))
span = max(e for _, e, _, _ in ev)
nb = int(span / bucket) + 1
conc = [[0.0, 0.0, 0.0] for _ in range(nb)]       # time with 0, 1, 2+ kernels per bucket
top = [collections.Counter() for _ in range(nb)]
pts = sorted([(s, 1) for s, _, _, _ in ev] + [(e, -1) for _, e, _, _ in ev])
cur, last = 0, 0.0
for t, dlt in pts:
    a = last
    while a < t:                                   # split [last, t) over the buckets
        b = min(t, (int(a / bucket) + 1) * bucket)
        conc[int(a / bucket)][min(cur, 2)] += b - a
        a = b
    cur += dlt; last = t
for s, e, name, _ in ev:
    a = s
    while a < e:
        b = min(e, (int(a / bucket) + 1) * bucket)
        top[int(a / bucket)][name] += b - a
        a = b
print(f"step span {span:.0f} us, {len(ev)} launches, kernel sum {sum(e - s for s, e, _, _ in ev):.0f} us, "
      f"queues {sorted(set(q for *_, q in ev))}")
tot = [0.0, 0.0, 0.0]
for i in range(nb):
    for k in range(3):
        tot[k] += conc[i][k]
    names = ", ".join(f"{n} {t:.0f}" for n, t in top[i].most_common(3))
    print(f"{i * bucket:7.0f} us  idle {conc[i][0]:5.1f}  one {conc[i][1]:5.1f}  2+ {conc[i][2]:5.1f}   {names}")
print(f"idle {tot[0]:.0f} us, one kernel {tot[1]:.0f} us, two or more {tot[2]:.0f} us")
