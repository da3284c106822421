"""Time per iteration of the noise drivers' loop (one sweep + noise counts with cache clearing) per shape class.  usage: perf_noise_classes.py [S] [groups] [classes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayestyper_amd import lib, shard, synth
from bayestyper_amd.host import count_model
S = int(sys.argv[1]) if len(sys.argv) > 1 else 30
G = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
which = sys.argv[3] if len(sys.argv) > 3 else "DCBA"
ctx = lib.Ctx(0)
flat = synth.make_mixture(G, S, seed=3030)
lg, ln = count_model.build_luts(S)
at = 0
for shape in ("D", "C", "B", "A"):
    n = flat["mixture"].get(shape, 0)
    ids = np.arange(at, at + n); at += n
    if shape not in which or n == 0: continue
    f = shard.take_groups(flat, ids)
    g = lib.Gibbs(ctx, f, lg, ln, seed=42, noise_seeding=1)
    g.init_chain(0)
    for phase, collect in (("burn", False), ("collect", True)):
        for _ in range(3): g.sweep(1, collect); g.noise_counts()
        ctx.sync(); t = time.perf_counter()
        for _ in range(20): g.sweep(1, collect); g.noise_counts()
        ctx.sync(); dt = (time.perf_counter() - t) / 20
        print(f"class {shape} S={S} {n} groups {phase}: {dt * 1e3:.2f} ms per iteration", flush=True)
    t = time.perf_counter(); g.sweep(20, False); ctx.sync(); print(f"   20 plain sweeps without cache clearing: {(time.perf_counter() - t) / 20 * 1e3:.2f} ms each")
    g.close()
