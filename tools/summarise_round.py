"""Turns the rocprofv3 outputs merged back under gpurun_out/<tag>_{stats,fetch,write,sq} (tools/profile_round.sh)
into the committed summaries under profiles/<round>_*.  Usage: python tools/summarise_round.py <tag> <round>"""
import collections, csv, json, os, re, shutil, sys
tag, rnd = sys.argv[1], sys.argv[2]
G = lambda n: os.path.join("gpurun_out", "%s_%s" % (tag, n))
OURS = ("ncc_", "ncc16_", "xc_", "xg_", "tri_")
def short(name):
    """Kernel name without its argument list; names the profiler leaves mangled (_Z<len><name>...) are cut out by length."""
    m = re.match(r"_Z(\d+)", name)
    if m:
        n = int(m.group(1))
        return name[m.end():m.end() + n]
    return name.split("(")[0].replace("void ", "")
shutil.copy(os.path.join(G("stats"), "s_kernel_stats.csv"), "profiles/%s_kernel_stats.csv" % rnd)
if os.path.exists(G("ubench.txt")):
    shutil.copy(G("ubench.txt"), "profiles/%s_ubench_valu_rates.txt" % rnd)
res = {}
for d, c in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    acc = collections.defaultdict(float); cnt = collections.Counter()
    for row in csv.DictReader(open(os.path.join(G(d), "p_counter_collection.csv"))):
        if row["Counter_Name"] != c: continue
        k = short(row["Kernel_Name"])
        acc[k] += float(row["Counter_Value"]); cnt[k] += 1
    for k in acc: res.setdefault(k, {})[c] = acc[k] / cnt[k]; res[k]["launches_" + c] = cnt[k]
with open("profiles/%s_pmc_hbm_traffic.csv" % rnd, "w") as f:
    f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --cpu-rows 0 --second-peak off --steps 1 --warmup 0\n")
    f.write("# (8192x8192 u8 W21 +-256 step, triangulation, 256 x 512x512 FFT tile batches, 16384x1024 u16 W31 +-512 band)\n")
    f.write("# average KiB per launch as reported, launches in the run; on gfx950 FETCH_SIZE under-reports wide (16 B/lane) streaming reads by 2x,\n")
    f.write("# other access widths are uncalibrated: the values are raw\n")
    f.write("kernel,FETCH_SIZE_KiB,WRITE_SIZE_KiB,launches\n")
    for k, v in sorted(res.items()):
        if k.startswith(OURS): f.write("%s,%.1f,%.1f,%d\n" % (k.replace(",", ";"), v.get("FETCH_SIZE", 0), v.get("WRITE_SIZE", 0), v.get("launches_FETCH_SIZE", 0)))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for row in csv.DictReader(open(os.path.join(G("sq"), "p_counter_collection.csv"))):
    k = short(row["Kernel_Name"])
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
names = sorted({c for k in acc for c in acc[k]})
def busy(k):
    g = lambda c: acc[k][c] / max(cnt[(k, c)], 1)
    den = g("SQ_BUSY_CYCLES") / 32 * 1024 / 4
    return (g("SQ_ACTIVE_INST_VALU") / den, g("SQ_ACTIVE_INST_LDS") / den) if den > 0 else (0.0, 0.0)
with open("profiles/%s_pmc_sq.csv" % rnd, "w") as f:
    f.write("# rocprofv3 --pmc " + " ".join(names) + " (one pass), bench.py --cpu-rows 0 --second-peak off --steps 1 --warmup 1\n")
    f.write("# per-launch averages summed over all shader engines; cycle counters are in units of 4 clocks;\n")
    f.write("# valu_busy = SQ_ACTIVE_INST_VALU / (SQ_BUSY_CYCLES / 32 * 1024 SIMDs / 4), lds_active likewise\n")
    f.write("kernel," + ",".join(names) + ",valu_busy,lds_active\n")
    for k in sorted(acc):
        if k.startswith(OURS):
            vb, la = busy(k)
            f.write(k.replace(",", ";") + "," + ",".join("%.4g" % (acc[k][c] / max(cnt[(k, c)], 1)) for c in names) + ",%.3f,%.3f\n" % (vb, la))
j = {}
def entry(prefix, label):
    # all template instances of the kernel that ran (the matrix-core argmax runs as an interior and an edge launch)
    ks = [k for k in res if k.startswith(prefix)]
    if not ks: return
    f_ = sum(res[k].get("FETCH_SIZE", 0) * 1024 for k in ks)
    w_ = sum(res[k].get("WRITE_SIZE", 0) * 1024 for k in ks)
    sq = sorted(q for q in acc if q.startswith(prefix))
    vb, la = busy(sq[0]) if sq else (None, None)
    j[label] = {"bytes": f_ + w_, "fetch_bytes_raw": f_, "write_bytes": w_, "source": "profiles/%s_pmc_hbm_traffic.csv" % rnd,
                "launches_summed": ks,
                "note": "FETCH_SIZE raw: the gfx950 x2 correction for 16-byte-per-lane streaming reads is not applied (the vector kernels read 4 B per lane or scalar; the matrix-core kernels stage their rows with 16-byte loads out of the L2, so their fetch figure may be low by up to 2x)",
                "valu_busy": round(vb, 3) if vb is not None else None, "lds_active": round(la, 3) if la is not None else None,
                "insts_valu": (acc[sq[0]]["SQ_INSTS_VALU"] / max(cnt[(sq[0], "SQ_INSTS_VALU")], 1)) if sq else None,
                "valu_busy_of": sq[0] if sq else None,
                "valu_busy_source": "profiles/%s_pmc_sq.csv" % rnd}
entry("ncc_mfma_argmax_kernel", "ncc_mfma_argmax_kernel")
entry("ncc_argmax_kernel", "ncc_argmax_kernel")
entry("ncc16_argmax_kernel", "ncc16_argmax_kernel")
# one batch of the FFT tile correlation: bench.py's path_f leg runs XC_BATCHES batches (2 warm-up + 10 timed), each as
# one or two launches of every xc_* kernel (a batch of 64 tiles or more runs as two halves on two streams)
XC_BATCHES = 12
xb = sum((v.get("FETCH_SIZE", 0) * v.get("launches_FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0) * v.get("launches_WRITE_SIZE", 0)) * 1024 / XC_BATCHES
         for k, v in res.items() if k.startswith("xc_"))
j["xcorr_batch"] = {"bytes": xb, "source": "profiles/%s_pmc_hbm_traffic.csv" % rnd,
                    "note": "all xc_* launches of the run (raw FETCH_SIZE + WRITE_SIZE) / %d batches of 256 tiles of 512x512" % XC_BATCHES}
# the whole 8-bit step: every ncc_* kernel of one step
st = {c: sum(v.get(c, 0) * 1024 * (v.get("launches_" + c, 1)) for k, v in res.items() if k.startswith("ncc_")) for c in ("FETCH_SIZE", "WRITE_SIZE")}
j["ncc_step"] = {"fetch_bytes_raw": st["FETCH_SIZE"], "write_bytes": st["WRITE_SIZE"], "bytes": st["FETCH_SIZE"] + st["WRITE_SIZE"],
                 "note": "all ncc_* kernel launches of the one timed step plus the parity-free extra calls of the run are NOT included: --steps 1 --warmup 0 runs exactly one step"}
json.dump(j, open("profiles/traffic.json", "w"), indent=1)
print(json.dumps(j, indent=1))
fn = "profiles/%s_kernel_stats.csv" % rnd
for row in csv.DictReader(open(fn)):
    if any(p in row["Name"] for p in OURS):
        print("   ", short(row["Name"])[:60].ljust(62), row["Calls"].rjust(4), "%.3f ms" % (float(row["AverageNs"]) / 1e6))
