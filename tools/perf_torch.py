"""The bench's Gibbs launch under the bench's process set-up, built up piece by piece: after every step of the set-up the schedule is timed again.
usage: perf_torch.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
import numpy as np
from bayestyper_amd import lib, synth
from bayestyper_amd.host import count_model
ctx = lib.Ctx(0)
flat = synth.make_mixture(600000, 3, seed=1000)
lg, ln = count_model.build_luts(3, mean=15.0, var=30.0, noise_rate=0.05)
g = lib.Gibbs(ctx, flat, lg, ln, seed=42)
t = lib.Timer(ctx)
def timeit(tag):
    ctx.sync(); torch.cuda.synchronize()
    t.start(); g.run(); t.stop()
    print(json.dumps({"after": tag, "ms": t.elapsed_ms()}), flush=True)
timeit("create")
K, KMC_P, REC, R = 55, 7, 13, 1_000_000_000
gen = torch.Generator(device=dev); gen.manual_seed(4)
records = torch.randint(0, 256, (R * REC + 16,), dtype=torch.uint8, device=dev, generator=gen)
timeit("records 13 GB (torch)")
lut = (np.arange(4 ** KMC_P + 1, dtype=np.float64) * (R / 4 ** KMC_P)).astype(np.uint64); lut[-1] = R
scan = lib.KmcScan(ctx, K, KMC_P, 1, R, lut)
timeit("KmcScan handle")
bloom = lib.Bloom.create(ctx, 51_000_000, 1e-4, K, threaded=True)
timeit("Bloom")
table = lib.Table(ctx, 30_000_000, 3, K)
timeit("Table")
d_hits = torch.zeros(1, dtype=torch.int64, device=dev)
ctx.sync(); torch.cuda.synchronize()
scan.run(bloom, table, 0, records.data_ptr(), 0, R, d_hits.data_ptr())
timeit("one KMC scan")
r = g.results()
timeit("results()")
tt = [lib.Timer(ctx) for _ in range(3)]
timeit("more timers")
