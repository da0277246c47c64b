"""Average per-launch HBM traffic of a kernel family from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE are
in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read, so it is doubled)."""
import csv, sys, glob, json, collections
fetch_dir, write_dir, pat = sys.argv[1], sys.argv[2], sys.argv[3]
def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    per = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            k = r["Kernel_Name"].split("(")[0]
            per[k][0] += 1; per[k][1] += float(r["Counter_Value"])
    return per
F, W = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
out = {}
for k in sorted(set(F) | set(W)):
    if any(q in k for q in pat.split(",")):
        nf, f = F.get(k, [0, 0.0]); nw, w = W.get(k, [0, 0.0])
        out[k] = {"launches": nf, "fetch_MB_per_launch_raw": f / max(nf, 1) / 1024, "write_MB_per_launch": w / max(nw, 1) / 1024}
tot_n = sum(v["launches"] for v in out.values())
fetch = sum(v["fetch_MB_per_launch_raw"] * v["launches"] for v in out.values()) / max(tot_n, 1)
write = sum(v["write_MB_per_launch"] * v["launches"] for v in out.values()) / max(tot_n, 1)
res = {"pattern": pat, "launches": tot_n, "fetch_MB_per_launch_raw": fetch, "fetch_MB_per_launch_corrected": 2 * fetch,
       "write_MB_per_launch": write, "traffic_MB_per_launch": 2 * fetch + write, "kernels": out}
print(json.dumps(res, indent=1))
