import sys, time
sys.path.insert(0,'.')
import numpy as np
from bayestyper_amd import lib, synth
from bayestyper_amd.host import count_model
from bayestyper_amd.host.inference_engine import InferenceEngine
ctx=lib.Ctx(0)
shape,n,S=sys.argv[1],int(sys.argv[2]),int(sys.argv[3])
flat=synth.make_batch(shape,n,S,seed=1, templates=4)
flat["group_index"]=np.arange(flat["num_groups"],dtype=np.uint32)
cd=count_model.CountDistribution(S,seed=42)
for s in range(S): cd.set_genomic(s,15.0,30.0)
eng=InferenceEngine(ctx,42,burn=10,samples=30,chains=2)
t=time.perf_counter(); g,tr=eng.estimate_noise_and_genotypes(flat,cd); ctx.sync(); dt=time.perf_counter()-t
its=2*40
print(shape,n,S,"noise-genotyping: %.1f ms per iteration (%d iterations, %.2f s), %.3g cluster-sweeps/s"%(dt/its*1e3,its,dt,flat['num_clusters']*its/dt), "rates", tr[-1][2:])
t=time.perf_counter(); g2=eng.estimate_genotypes(flat,cd); ctx.sync(); dt2=time.perf_counter()-t
print("  default mode same schedule: %.2f s, %.3g cluster-sweeps/s"%(dt2, flat['num_clusters']*its/dt2))
