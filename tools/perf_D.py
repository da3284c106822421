import sys, time
sys.path.insert(0,'.')
import numpy as np
from bayestyper_amd import lib, synth
from bayestyper_amd.host import count_model
ctx=lib.Ctx(0)
n,S=int(sys.argv[1]),int(sys.argv[2])
t=time.perf_counter(); flat=synth.make_batch("D",n,S,seed=1); print("make_batch %.1f s"%(time.perf_counter()-t), "K", int((flat["kmer_off"][1:]-flat["kmer_off"][:-1]).max()))
g,nz=count_model.build_luts(S)
kw=dict(seed=9,chains=1,burn=5,iters=10)
t=time.perf_counter(); gg=lib.Gibbs(ctx,flat,g,nz,**kw); ctx.sync(); print("create %.1f s"%(time.perf_counter()-t), "device MB", gg.device_bytes()/1e6)
t=time.perf_counter(); gg.init_chain(0); ctx.sync(); print("init_chain(0) [construct+reset] %.2f s"%(time.perf_counter()-t))
t=time.perf_counter(); gg.sweep(5,False); ctx.sync(); print("5 burn-in sweeps %.2f s"%(time.perf_counter()-t))
t=time.perf_counter(); gg.sweep(10,True); ctx.sync(); print("10 collected sweeps %.2f s"%(time.perf_counter()-t))
