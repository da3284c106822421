"""Time the default Gibbs schedule of the bench's mixture batch as a whole and per shape class (one MI355X).
usage: python tools/perf_classes.py [S] [groups] [classes e.g. ABCD+]   ('+' = the whole mixture)
BT_PERF_RUNS = schedules per class (default 2: the second is the warm one; tools/sq_counters.sh sets 1 so that the counters of a pass cover ONE schedule)
BT_PERF_BURN / BT_PERF_ITERS = burn-in / collected sweeps per chain instead of 100 / 250 (tools/traffic_by_array.py: 0 / 1 isolates the chain starts)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bayestyper_amd import lib, shard, synth
from bayestyper_amd.host import count_model

S = int(sys.argv[1]) if len(sys.argv) > 1 else 3
G = int(sys.argv[2]) if len(sys.argv) > 2 else 150_000
which = sys.argv[3] if len(sys.argv) > 3 else "+ABCD"
ctx = lib.Ctx(0)
flat = synth.make_mixture(G, S, seed=1000, hetero=not os.environ.get("BT_HOMO"))
lut_g, lut_n = count_model.build_luts(S, mean=15.0, var=30.0, noise_rate=0.05)
bounds, at = {}, 0
for shape in ("D", "C", "B", "A"):
    n = flat["mixture"].get(shape, 0)
    bounds[shape] = (at, at + n)
    at += n
for w in which:
    f = flat if w == "+" else shard.take_groups(flat, np.arange(*bounds[w]))
    t_create = time.perf_counter()
    sched = {k: int(os.environ[e]) for k, e in (("burn", "BT_PERF_BURN"), ("iters", "BT_PERF_ITERS")) if e in os.environ}
    g = lib.Gibbs(ctx, f, lut_g, lut_n, seed=42, **sched)
    t_create = time.perf_counter() - t_create
    t = lib.Timer(ctx)
    ms = []
    for _ in range(int(os.environ.get("BT_PERF_RUNS", "2"))):
        t.start(); g.run(); t.stop(); ms.append(t.elapsed_ms())
    print(json.dumps({"class": w, "S": S, "groups": int(f["num_groups"]), "clusters": int(f["num_clusters"]), "ms": ms, "create_s": round(t_create, 2), "device_GB": g.device_bytes() / 1e9,
                      "cluster_sweeps_per_s": f["num_clusters"] * 7000 / (min(ms) * 1e-3)}), flush=True)
    g.close()
