"""HBM traffic of one 512x512 inference batch from two rocprofv3 --pmc passes over tools/infer_prof.py (FETCH_SIZE and
WRITE_SIZE, one run each; both in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read, so
it is doubled).  usage: python tools/pmc_infer.py <fetch dir> <write dir> <passes in the run>"""
import collections, csv, glob, json, sys
fetch_dir, write_dir, passes = sys.argv[1], sys.argv[2], int(sys.argv[3])


def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    per = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            per[k][0] += 1
            per[k][1] += float(r["Counter_Value"])
    return per


F, W = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
kern, tot_f, tot_w = {}, 0.0, 0.0
for k in sorted(set(F) | set(W)):
    nf, f = F.get(k, [0, 0.0])
    nw, w = W.get(k, [0, 0.0])
    kern[k] = {"launches_per_batch": nf / passes, "fetch_MB_per_batch_corrected": 2 * f / 1024 / passes,
               "write_MB_per_batch": w / 1024 / passes}
    tot_f += 2 * f / 1024 / passes
    tot_w += w / 1024 / passes
top = dict(sorted(kern.items(), key=lambda kv: -(kv[1]["fetch_MB_per_batch_corrected"] + kv[1]["write_MB_per_batch"]))[:16])
print(json.dumps({"passes": passes, "fetch_MB_per_batch_corrected": tot_f, "write_MB_per_batch": tot_w,
                  "traffic_MB_per_batch": tot_f + tot_w, "kernels": top}, indent=1))
