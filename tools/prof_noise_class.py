"""Per-phase cycle shares of one iteration of the noise drivers' loop (one sweep, noise counts, caches cleared) on the shape classes of a
mixture (needs the -DBT_PROF build: tools/build_prof.sh; run with BTGPU_LIB=bayestyper_amd/libbtgpu_prof.so).
usage: prof_noise_class.py [S] [groups] [classes]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayestyper_amd import lib, shard, synth
from bayestyper_amd.host import count_model
S = int(sys.argv[1]) if len(sys.argv) > 1 else 30
G = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
which = sys.argv[3] if len(sys.argv) > 3 else "DCBA"
ctx = lib.Ctx(0)
prof = lib._lib.bt_diag_prof
prof.argtypes = [C.c_void_p, C.c_int]
flat = synth.make_mixture(G, S, seed=3030)
lg, ln = count_model.build_luts(S)
names = {0: "nz scan", 1: "multi_refresh", 2: "candidates", 3: "draw+search", 4: "hfd(sets)", 5: "upd_multi", 6: "collect(stats)", 7: "frequencies", 11: "rng top-up", 12: "fill/invalidate table",
         13: "hot swap", 14: "prepare_nested", 24: "draw:u01", 25: "draw:exp+prefix", 26: "draw:search+margin", 28: "draw:exact chain"}
at = 0
for shape in ("D", "C", "B", "A"):
    n = flat["mixture"].get(shape, 0)
    ids = np.arange(at, at + n); at += n
    if shape not in which or n == 0:
        continue
    f = shard.take_groups(flat, ids)
    g = lib.Gibbs(ctx, f, lg, ln, seed=42, noise_seeding=1)
    g.init_chain(0)
    for _ in range(3): g.sweep(1, False); g.noise_counts()
    ctx.sync()
    buf = np.zeros(32, np.uint64)
    prof(buf.ctypes.data, 1)
    t = time.perf_counter()
    for _ in range(10): g.sweep(1, True); g.noise_counts()
    ctx.sync(); dt = (time.perf_counter() - t) / 10
    prof(buf.ctypes.data, 1)
    tot = max(1, sum(int(buf[i]) for i in names))
    print(f"class {shape} S={S} {n} groups: {dt * 1e3:.2f} ms per iteration; phase shares:", {nm: "%.1f%%" % (100 * int(buf[i]) / tot) for i, nm in names.items() if buf[i]}, flush=True)
    g.close()
