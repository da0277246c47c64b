"""MFMA busy fraction per kernel from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE):
busy cycles summed over all SIMDs / (active cycles of the dispatch x 256 CUs x 4 SIMDs).  GRBM_GUI_ACTIVE is
reported summed over the 8 XCDs."""
import csv, sys, glob, json, collections
d, pat = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
per = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    per[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        cnt[k] += 1
out, tot_b, tot_a = {}, 0.0, 0.0
for k, c in sorted(per.items()):
    if not any(q in k for q in pat.split(",")):
        continue
    busy, act = c["SQ_VALU_MFMA_BUSY_CYCLES"], c["GRBM_GUI_ACTIVE"] / 8.0
    tot_b += busy; tot_a += act
    out[k] = {"launches": cnt[k], "mfma_busy_frac": busy / (act * 256 * 4) if act else 0.0}
print(json.dumps({"pattern": pat, "mfma_busy_frac": tot_b / (tot_a * 256 * 4), "kernels": out}, indent=1))
