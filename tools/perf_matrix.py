"""GPU (whole batch) vs the oracle on all host cores (a sample of the same shape) for the synthetic shapes of SURVEY 8(d); short schedule
(2 chains x (20 + 50) sweeps), so chain starts weigh more than in the default schedule.  Run on the GPU box: python tools/perf_matrix.py"""
import sys, time, os
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from bayestyper_amd import lib, synth
from bayestyper_amd.host import count_model
import _oracle
ctx=lib.Ctx(0); orc=_oracle.load_oracle(); cores=os.cpu_count()
kw=dict(seed=3,chains=2,burn=20,iters=50)
cases=[("A",131072,1,8192),("A",65536,10,4096),("A",32768,30,2048),("B",16384,1,2048),("B",8192,10,1024),("B",4096,30,512),("C",2048,1,512),("C",1024,10,256),("C",512,30,256),("D",512,4,64),("D",256,30,32)]
for shape,n,S,ncpu in cases:
    flat=synth.make_batch(shape,n,S,seed=1,templates=4)
    g,nz=count_model.build_luts(S)
    gg=lib.Gibbs(ctx,flat,g,nz,**kw)
    t=lib.Timer(ctx); t.start(); gg.run(); t.stop(); gms=t.elapsed_ms(); gg.close()
    cf=synth.make_batch(shape,ncpu,S,seed=1,templates=4)
    og=_oracle.OrcGibbs(orc,cf,g,nz,**kw)
    t0=time.perf_counter(); og.run(cores); cs=time.perf_counter()-t0; og.close()
    sweeps=2*70
    gpu_rate=flat["num_clusters"]*sweeps/(gms*1e-3); cpu_rate=cf["num_clusters"]*sweeps/cs
    print(f"{shape} S={S:2d} n={n:6d}: GPU {gms:9.1f} ms = {gpu_rate:.3g} cl-sweeps/s | CPU({cores}t, n={ncpu}) {cs:6.2f} s = {cpu_rate:.3g} | ratio {gpu_rate/cpu_rate:6.1f}", flush=True)
