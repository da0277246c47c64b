"""Per-kernel counter sums from one rocprofv3 --pmc pass: for every kernel whose name contains one of the comma
separated patterns, launches and the per-launch mean of each collected counter.
usage: pmc_table.py DIR pattern[,pattern...]"""
import csv, sys, glob, json, collections
d, pat = sys.argv[1], sys.argv[2].split(",")
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
per = collections.defaultdict(lambda: collections.defaultdict(float))
ids = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if not any(q in k for q in pat):
        continue
    per[k][r["Counter_Name"]] += float(r["Counter_Value"])
    ids[k].add(r["Dispatch_Id"])
out = {}
for k, c in sorted(per.items()):
    n = len(ids[k])
    out[k] = {"launches": n, **{name: round(v / n, 1) for name, v in sorted(c.items())}}
print(json.dumps(out, indent=1))
