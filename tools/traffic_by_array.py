"""Where the HBM traffic of the default Gibbs schedule goes (3.28 TB counted against 179 GB algorithmic, no split by array).
Counter passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one schedule each, tools/perf_classes.py on the bench batch) of
  full        the product library, 20 x (100 + 250) sweeps
  starts      the product library, 20 x (0 + 1) sweeps: chain starts (k-mer subset selection and its compact copies, table fills, generator seeding) + one sweep
  full_nomt   a DIAGNOSTIC library whose generators draw from a counter hash instead of the mt19937 states (-DBT_DIAG_FAKE_MT, tools/build_variant_all.sh fakemt ...):
              the same control flow statistically, no generator state moved
  starts_nomt the same, chain starts only
-> per kernel: generator states in the sweeps = (full - starts) - (full_nomt - starts_nomt); in the chain starts = starts - starts_nomt; everything else in the
sweeps (tables of sums, candidate scratch, run logs, ring/state write-back at the end of a launch) = full_nomt - starts_nomt; chain starts without generators = starts_nomt.
usage (GPU box, repo root): python tools/traffic_by_array.py <tag> [S 3] [groups 600320] [fake-MT library bayestyper_amd/libbtgpu_fakemt.so]"""
import csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
S = sys.argv[2] if len(sys.argv) > 2 else "3"
G = sys.argv[3] if len(sys.argv) > 3 else "600320"
fake = sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "bayestyper_amd", "libbtgpu_fakemt.so")
out_dir = os.path.join(ROOT, "gpurun_out", "summ_" + tag)
os.makedirs(out_dir, exist_ok=True)
KERNELS = ("gibbs_simple_kernel", "gibbs_single_kernel", "gibbs_hot_kernel", "gibbs_kernel")


def one_pass(counter, env_extra):
    d = os.path.join("/tmp", "bt_tba")
    shutil.rmtree(d, ignore_errors=True)
    env = dict(os.environ, TMPDIR="/tmp", BT_PERF_RUNS="1", **env_extra)
    p = subprocess.run(["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "tools", "perf_classes.py"), S, G, "+"],
                       env=env, capture_output=True, text=True, timeout=900)
    agg = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = next((k for k in KERNELS if k in r["Kernel_Name"]), None)
            if name is None:
                continue
            agg[name] = agg.get(name, 0.0) + float(r["Counter_Value"]) * 1024.0   # KiB -> bytes (MI355X_MICROARCH.md); as counted, no width correction
    ms = None
    for line in p.stdout.splitlines():
        if line.startswith("{"):
            ms = json.loads(line)["ms"][0]
    shutil.rmtree(d, ignore_errors=True)
    return agg, ms


runs = {"full": {}, "starts": {"BT_PERF_BURN": "0", "BT_PERF_ITERS": "1"}, "full_nomt": {"BTGPU_LIB": fake}, "starts_nomt": {"BTGPU_LIB": fake, "BT_PERF_BURN": "0", "BT_PERF_ITERS": "1"}}
counted = {}
for name, env in runs.items():
    counted[name] = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        agg, ms = one_pass(counter, env)
        counted[name][counter] = agg
        counted[name]["launch_ms_under_pmc"] = ms
        print(name, counter, {k: "%.3g" % v for k, v in agg.items()}, ms, file=sys.stderr, flush=True)


def tot(run, kernel=None):
    return sum(v for c in ("FETCH_SIZE", "WRITE_SIZE") for k, v in counted[run][c].items() if kernel in (None, k))


split = {}
for k in (None,) + KERNELS:
    full, starts, full_n, starts_n = tot("full", k), tot("starts", k), tot("full_nomt", k), tot("starts_nomt", k)
    if full == 0:
        continue
    split[k or "all sampling kernels"] = {
        "total_bytes": full,
        "generator_states_in_sweeps (A_MT <-> A_RING refills)": (full - starts) - (full_n - starts_n),
        "generator_states_in_chain_starts (seeding, shuffle + one Bernoulli draw per k-mer)": starts - starts_n,
        "sweeps_without_generators (A_UCACHE / A_CUM / A_SCACHE reads, A_EVLOG stores, hot arrays in and out of LDS once per launch)": full_n - starts_n,
        "chain_starts_without_generators (A_UNIQ / A_USUB, A_M rows -> A_SUBM / A_SUBCNT / A_SUBIC, A_HVCOUNT, A_UCACHE fill, drain of the run logs -> A_ASTATS / A_DIPFREQ)": starts_n,
    }
import hashlib
sys.path.insert(0, ROOT)
import bench
out = {"source_hash_gibbs": bench.source_hash("gibbs"), "S": int(S), "groups": int(G), "command": "tools/traffic_by_array.py (see its docstring); bytes = counter KiB x 1024, as counted",
       "note": "the fake-generator build changes WHICH values are drawn, not how many: differences are statistical (same batch, same schedule), good to a few per cent",
       "split": split, "counted": counted}
dst = os.path.join(out_dir, f"{tag}_traffic_by_array.json")
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(split, indent=1))
