"""Per-phase cycle shares of the Gibbs kernel on one shape class of the bench's mixture (needs the -DBT_PROF build:
tools/build_prof.sh; run with BTGPU_LIB=bayestyper_amd/libbtgpu_prof.so).  usage: prof_class.py <class A|B|C|D> [S] [groups] [chains]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayestyper_amd import lib, shard, synth
from bayestyper_amd.host import count_model
w = sys.argv[1]
S = int(sys.argv[2]) if len(sys.argv) > 2 else 3
G = int(sys.argv[3]) if len(sys.argv) > 3 else 150_000
chains = int(sys.argv[4]) if len(sys.argv) > 4 else 4
ctx = lib.Ctx(0)
prof = lib._lib.bt_diag_prof
prof.argtypes = [C.c_void_p, C.c_int]
flat = synth.make_mixture(G, S, seed=1000, fractions={w: synth.MIXTURE[w]})
lg, ln = count_model.build_luts(S)
gg = lib.Gibbs(ctx, flat, lg, ln, chains=chains)
buf = np.zeros(32, np.uint64)
prof(buf.ctypes.data, 1)
t = lib.Timer(ctx); t.start(); gg.run(); t.stop(); ms = t.elapsed_ms()
prof(buf.ctypes.data, 1)
names = {0: "nz scan", 1: "multi_refresh", 2: "candidates", 3: "draw+search", 4: "hfd(sets)", 5: "upd_multi", 6: "collect(stats)", 7: "frequencies", 11: "rng top-up", 12: "fill/invalidate table",
         13: "hot swap", 14: "prepare_nested"}
sweep = max(1, sum(int(buf[i]) for i in list(names) + [21, 22, 23, 24, 25, 26, 28]))
init = int(buf[15])
print(f"class {w} S={S}: {flat['num_groups']} groups, {flat['num_clusters']} clusters, {chains} chains: {ms:.1f} ms")
print("  sweep phases (share of sweep cycles):", {n: "%.1f%%" % (100 * int(buf[i]) / sweep) for i, n in names.items()})
print("  chain init %.1f%% of (init + sweeps): shuffle %.1f%% select %.1f%% compact copies %.1f%% of init" % (100 * init / (init + sweep), 100 * int(buf[8]) / max(init, 1), 100 * int(buf[9]) / max(init, 1),
                                                                                                        100 * int(buf[10]) / max(init, 1)))
print("  collect detail: slow-path calls %d, rebuilds %d; cycles flush %.3g rebuild %.3g apply %.3g (collect total %.3g)" % (int(buf[19]), int(buf[20]), int(buf[16]), int(buf[17]), int(buf[18]), int(buf[6])))
xn = {21: "freq:simplex vec", 22: "freq:gamma(plus)", 23: "freq:add zeros", 24: "draw:u01", 25: "draw:exp+prefix", 26: "draw:search+margin", 28: "draw:exact chain"}
print("  sub-phases (share of sweep cycles; freq:rest and draw:decode stay in the parent rows):", {xn.get(i, i): "%.1f%%" % (100 * int(buf[i]) / sweep) for i in range(21, 32) if buf[i]})
nwaves = len(range(0, flat["num_groups"], 4 if w in "CD" else 64))
print("  cycles per wavefront-sweep (first lane's clock): %.3g" % (sweep / max(1, nwaves) / (chains * 350)))
gg.close()
