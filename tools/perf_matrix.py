"""GPU throughput of the Gibbs kernel over the synthetic shapes of SURVEY 8(d) x sample counts; short schedule (2 chains x (20 + 50)
sweeps), so chain starts weigh more than in the default schedule.  Run on the GPU box: python tools/perf_matrix.py"""
import sys
sys.path.insert(0, '.')
from bayestyper_amd import lib, synth
from bayestyper_amd.host import count_model

ctx = lib.Ctx(0)
kw = dict(seed=3, chains=2, burn=20, iters=50)
cases = [("A", 131072, 1), ("A", 65536, 10), ("A", 32768, 30), ("B", 16384, 1), ("B", 8192, 10), ("B", 4096, 30), ("C", 2048, 1), ("C", 1024, 10), ("C", 512, 30),
         ("D", 512, 4), ("D", 256, 30)]
for shape, n, S in cases:
    flat = synth.make_batch(shape, n, S, seed=1, templates=4)
    g, nz = count_model.build_luts(S)
    gg = lib.Gibbs(ctx, flat, g, nz, **kw)
    t = lib.Timer(ctx)
    t.start()
    gg.run()
    t.stop()
    ms = t.elapsed_ms()
    gg.close()
    print(f"{shape} S={S:2d} n={n:6d}: {ms:9.1f} ms = {flat['num_clusters'] * 140 / (ms * 1e-3):.3g} cluster-sweeps/s", flush=True)
