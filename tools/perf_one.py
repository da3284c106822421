import sys
sys.path.insert(0,'.')
import numpy as np
from bayestyper_amd import lib, synth
from bayestyper_amd.host import count_model
ctx=lib.Ctx(0)
shape,n,S=sys.argv[1],int(sys.argv[2]),int(sys.argv[3])
flat=synth.make_batch(shape,n,S,seed=1, templates=4)
g,nz=count_model.build_luts(S)
gg=lib.Gibbs(ctx,flat,g,nz,chains=2)
t=lib.Timer(ctx); t.start(); gg.run(); t.stop(); ms=t.elapsed_ms()
print(shape,n,S,ms,"ms")
