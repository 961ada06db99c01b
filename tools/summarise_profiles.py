"""Turns the rocprofv3 outputs merged back under gpurun_out/<tag>_{stats,fetch,write,sq,xstats,n16}
into the committed summaries under profiles/.  Usage: python tools/summarise_profiles.py r01d"""
import collections, csv, json, os, shutil, sys
tag = sys.argv[1]
G = lambda n: os.path.join("gpurun_out", "%s_%s" % (tag, n))
shutil.copy(os.path.join(G("stats"), "s_kernel_stats.csv"), "profiles/r01_kernel_stats.csv")
shutil.copy(os.path.join(G("xstats"), "s_kernel_stats.csv"), "profiles/r01_xcorr_kernel_stats.csv")
if os.path.exists(os.path.join(G("n16"), "s_kernel_stats.csv")):
    shutil.copy(os.path.join(G("n16"), "s_kernel_stats.csv"), "profiles/r01_ncc16_kernel_stats.csv")
res = {}
for d, c in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    acc = collections.defaultdict(float); cnt = collections.Counter()
    for row in csv.DictReader(open(os.path.join(G(d), "p_counter_collection.csv"))):
        if row["Counter_Name"] != c: continue
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k] += float(row["Counter_Value"]); cnt[k] += 1
    for k in acc: res.setdefault(k, {})[c] = acc[k] / cnt[k]
with open("profiles/r01_pmc_hbm_traffic.csv", "w") as f:
    f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 0 --cpu-rows 0 (8192x8192 u8, W21, +-256)\n")
    f.write("# values are KiB per launch as reported; on gfx950 FETCH_SIZE under-reports wide (16 B/lane) streaming reads by 2x, these kernels read 4 B per lane (uncalibrated)\n")
    f.write("kernel,FETCH_SIZE_KiB,WRITE_SIZE_KiB\n")
    for k, v in sorted(res.items()):
        if k.startswith("ncc_"): f.write("%s,%.1f,%.1f\n" % (k.replace(",", ";"), v.get("FETCH_SIZE", 0), v.get("WRITE_SIZE", 0)))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for row in csv.DictReader(open(os.path.join(G("sq"), "p_counter_collection.csv"))):
    k = row["Kernel_Name"].split("(")[0].replace("void ", "")
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
names = sorted({c for k in acc for c in acc[k]})
with open("profiles/r01_pmc_sq_argmax.csv", "w") as f:
    f.write("# rocprofv3 --pmc " + " ".join(names) + " (one pass), bench.py --steps 1 --warmup 1 --cpu-rows 0\n")
    f.write("# per-launch sums over all shader engines; cycle counters are in units of 4 clocks; VALU busy = ACTIVE_INST_VALU / (BUSY_CYCLES/32 * 1024 SIMDs / 4)\n")
    f.write("kernel," + ",".join(names) + "\n")
    for k in sorted(acc):
        if k.startswith("ncc_"): f.write(k.replace(",", ";") + "," + ",".join("%.4g" % (acc[k][c] / max(cnt[(k, c)], 1)) for c in names) + "\n")
a = [k for k in acc if k.startswith("ncc_argmax")][0]
g = lambda c: acc[a][c] / cnt[(a, c)]
den = g("SQ_BUSY_CYCLES") / 32 * 1024 / 4
ka = [k for k in res if k.startswith("ncc_argmax")][0]
f_, w_ = res[ka]["FETCH_SIZE"] * 1024, res[ka]["WRITE_SIZE"] * 1024
j = {"ncc_argmax_kernel": {"bytes": f_ + w_, "fetch_bytes_raw": f_, "write_bytes": w_, "source": "profiles/r01_pmc_hbm_traffic.csv",
     "note": "FETCH_SIZE raw (4-byte-per-lane and scalar loads: the gfx950 x2 correction for wide streaming reads is not applied); the writes are the per-chunk key planes",
     "valu_busy": round(g("SQ_ACTIVE_INST_VALU") / den, 3), "lds_active": round(g("SQ_ACTIVE_INST_LDS") / den, 3),
     "valu_busy_source": "profiles/r01_pmc_sq_argmax.csv: SQ_ACTIVE_INST_VALU / (SQ_BUSY_CYCLES / 32 * 1024 / 4)"}}
json.dump(j, open("profiles/traffic.json", "w"), indent=1)
print(json.dumps(j["ncc_argmax_kernel"]))
for fn in ("profiles/r01_kernel_stats.csv", "profiles/r01_xcorr_kernel_stats.csv", "profiles/r01_ncc16_kernel_stats.csv"):
    if not os.path.exists(fn): continue
    print(fn)
    for row in csv.DictReader(open(fn)):
        if "ncc" in row["Name"] or "xc_" in row["Name"] or "tri_" in row["Name"]:
            print("   ", row["Name"][:44].ljust(46), row["Calls"].rjust(4), "%.3f ms" % (float(row["AverageNs"]) / 1e6))
