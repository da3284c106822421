"""Trim rocprofv3 CSV output to the rows of this repo's kernels.  usage: prof_summarize.py <rocprof out dir> <tag> <dest dir>"""
import csv, glob, json, os, sys
src, tag, dest = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(dest, exist_ok=True)
OURS = ("gibbs_kernel", "gibbs_hot_kernel", "gibbs_single_kernel", "summary_kernel", "kmc_", "bloom_", "table_", "kmers_from", "nthash", "intercluster", "classify", "rocprim", "find_paths", "paths", "_kernel")
def ours(name): return any(k in name for k in OURS)
def short(name):
    for k in ("gibbs_hot_kernel", "gibbs_single_kernel", "gibbs_simple_kernel", "gibbs_noise_kernel", "gibbs_kernel", "summary_kernel", "kmc_scan_kernel<true>", "kmc_scan_kernel<false>", "kmc_scan_kernel", "kmc_route_kernel", "kmc_probe_kernel<true>", "kmc_probe_kernel<false>", "kmc_probe_kernel",
              "kmc_apply_kernel", "bloom_insert_kernel", "bloom_contains_kernel"):
        if k in name: return k
    if "rocprim" in name: return "rocprim " + ("onesweep" if "onesweep" in name else "histogram" if "histogram" in name else name[-40:])
    return name.replace("(anonymous namespace)::", "").split("(")[0][-50:]
def col(r, *subs):
    for k in r:
        if all(x.lower() in k.lower() for x in subs): return r[k]
    return None
for f in glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.reader(open(f)))
    keep = [rows[0]] + [r for r in rows[1:] if ours(r[0])]
    with open(os.path.join(dest, f"{tag}_kernel_stats.csv"), "w", newline="") as o:
        csv.writer(o).writerows(keep)
for f in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
    rd = csv.DictReader(open(f))
    out = []
    for r in rd:
        if ours(r["Kernel_Name"]):
            out.append({"kernel": short(r["Kernel_Name"]), "grid_x": col(r, "grid", "x") or col(r, "grid"), "wg_x": col(r, "workgroup", "x") or col(r, "workgroup"), "lds": r.get("LDS_Block_Size"),
                        "vgpr": r.get("VGPR_Count"), "accum_vgpr": r.get("Accum_VGPR_Count"), "sgpr": r.get("SGPR_Count"), "scratch": r.get("Scratch_Size"),
                        "start_ns": int(r["Start_Timestamp"]), "dur_ms": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6})
    if out:
        t0 = min(x["start_ns"] for x in out)
        for x in out: x["start_ms"] = (x.pop("start_ns") - t0) / 1e6
        # dispatches shorter than 0.1 ms (the thousands of per-iteration launches of a noise-driver chain, setup dispatches) are kept as one
        # aggregate row per kernel
        small = {}
        for x in out:
            if x["dur_ms"] < 0.1:
                a = small.setdefault(x["kernel"], {"kernel": x["kernel"], "aggregate_of_dispatches_shorter_than_0.1_ms": 0, "dur_ms": 0.0})
                a["aggregate_of_dispatches_shorter_than_0.1_ms"] += 1
                a["dur_ms"] += x["dur_ms"]
        out = [x for x in out if x["dur_ms"] >= 0.1] + list(small.values())
        json.dump(out, open(os.path.join(dest, f"{tag}_kernel_trace.json"), "w"), indent=0)
agg = {}
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if not ours(r["Kernel_Name"]): continue
        k = (short(r["Kernel_Name"]), col(r, "grid", "x") or col(r, "grid"), r["Counter_Name"])
        a = agg.setdefault(k, [0, 0.0])
        a[0] += 1; a[1] += float(r["Counter_Value"])
if agg:
    rows = [{"kernel": k[0], "grid": k[1], "counter": k[2], "dispatch_counter_rows": v[0], "sum": v[1]} for k, v in sorted(agg.items())]
    json.dump(rows, open(os.path.join(dest, f"{tag}_pmc.json"), "w"), indent=0)
print("summaries written to", dest, os.listdir(dest))
