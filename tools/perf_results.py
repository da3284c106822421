"""Time the result path of a launch (bt_gibbs_result_sizes + bt_gibbs_result_fetch) by parts.  usage: perf_results.py [S] [groups]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from bayestyper_amd import lib, synth
from bayestyper_amd.host import count_model
S = int(sys.argv[1]) if len(sys.argv) > 1 else 3
G = int(sys.argv[2]) if len(sys.argv) > 2 else 600_000
ctx = lib.Ctx(0)
flat = synth.make_mixture(G, S, seed=1000)
lg, ln = count_model.build_luts(S)
g = lib.Gibbs(ctx, flat, lg, ln, seed=42)
g.run(); ctx.sync()
for rep in range(3):
    t0 = time.perf_counter()
    nd, nc = C.c_uint64(), C.c_uint64()
    lib.check(lib.bt_gibbs_result_sizes(g.h, C.byref(nd), C.byref(nc)))
    t1 = time.perf_counter()
    nd_, nc_ = nd.value, nc.value
    if rep < 2:
        bufs = [np.zeros(g.C + 1, np.uint64), np.zeros(max(nd_, 1), np.uint16), np.zeros(max(nd_, 1), np.uint16), np.zeros(max(nd_, 1) * S, np.uint32), np.zeros(g.C + 1, np.uint64), np.zeros(max(nc_, 1) * 12, np.float64)]
    t2 = time.perf_counter()
    lib.check(lib.bt_gibbs_result_fetch(g.h, *[lib._np_ptr(b) for b in bufs]))
    t3 = time.perf_counter()
    print(f"rep {rep}: sizes {1e3*(t1-t0):.1f} ms, host alloc {1e3*(t2-t1):.1f} ms, fetch {1e3*(t3-t2):.1f} ms ({'fresh' if rep < 2 else 'touched'} host buffers), bytes {sum(b.nbytes for b in bufs)/1e6:.0f} MB", flush=True)
