"""HBM traffic per launch from the two PMC passes of the default bench command (tools/profile_round.sh) -> <dest>/<tag>_traffic.json, the
file bench.py's roofline.traffic is filled from (matched by the hash of the device sources).
usage: pmc_traffic.py <dir with <tag>_FETCH_SIZE_pmc.json / <tag>_WRITE_SIZE_pmc.json> <tag> <samples S of the pass>"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

d, tag, S = sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 3
tot = {"gibbs": {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}, "kmc": {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}}
detail = []
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for r in json.load(open(os.path.join(d, f"{tag}_{c}_pmc.json"))):
        k = r["kernel"]
        grp = "gibbs" if "gibbs" in k else "kmc" if (k.startswith("kmc_") or k.startswith("rocprim")) else None
        if grp is None or r["counter"] != c:
            continue
        tot[grp][c] += r["sum"] * 1024.0   # the counters are in KiB (MI355X_MICROARCH.md); no x2: most accesses here are narrow, not wide coalesced streams
        detail.append({"kernel": k, "grid": r["grid"], "counter": c, "bytes": r["sum"] * 1024.0})
out = {"source_hash": bench.source_hash(), "source_hash_gibbs": bench.source_hash("gibbs"), "source_hash_kmc": bench.source_hash("kmc"), "command": "python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-paths --no-pcie --no-extra (one rocprofv3 --pmc pass per counter)",
       "gibbs_bytes_per_schedule": tot["gibbs"]["FETCH_SIZE"] + tot["gibbs"]["WRITE_SIZE"], "gibbs_fetch_bytes": tot["gibbs"]["FETCH_SIZE"], "gibbs_write_bytes": tot["gibbs"]["WRITE_SIZE"],
       "kmc_bytes_per_scan": (tot["kmc"]["FETCH_SIZE"] + tot["kmc"]["WRITE_SIZE"]) / S, "kmc_fetch_bytes_per_scan": tot["kmc"]["FETCH_SIZE"] / S, "kmc_write_bytes_per_scan": tot["kmc"]["WRITE_SIZE"] / S,
       "samples": S, "detail": detail}
json.dump(out, open(os.path.join(d, f"{tag}_traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "detail"}))
