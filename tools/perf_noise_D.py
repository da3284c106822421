import sys, time
sys.path.insert(0,'.')
import numpy as np
from bayestyper_amd import lib, synth
from bayestyper_amd.host import count_model
from bayestyper_amd.host.inference_engine import InferenceEngine
ctx=lib.Ctx(0)
n,S=int(sys.argv[1]),int(sys.argv[2])
flat=synth.make_batch("D",n,S,seed=1); flat["group_index"]=np.arange(n,dtype=np.uint32)
cd=count_model.CountDistribution(S,seed=42)
for s in range(S): cd.set_genomic(s,15.0,30.0)
eng=InferenceEngine(ctx,42,burn=5,samples=10,chains=1)
t=time.perf_counter(); g,tr=eng.estimate_noise_and_genotypes(flat,cd); ctx.sync(); dt=time.perf_counter()-t
print("D",n,S,"noise-genotyping 15 iterations: %.2f s"%dt)
