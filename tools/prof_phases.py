import sys, ctypes as C
sys.path.insert(0,'.')
import numpy as np
from bayestyper_amd import lib, synth
from bayestyper_amd.host import count_model
ctx=lib.Ctx(0)
prof=lib._lib.bt_diag_prof; prof.argtypes=[C.c_void_p,C.c_int]
names=["nzscan","multi_refresh","candidates","draw+search","hfd(sets)","upd_multi","collect","frequencies"]
for shape,n,S in [("A",262144,1),("B",16384,1)]:
    flat=synth.make_batch(shape,n,S,seed=1, templates=4)
    g,nz=count_model.build_luts(S)
    gg=lib.Gibbs(ctx,flat,g,nz,chains=2)
    buf=np.zeros(16,np.uint64); prof(buf.ctypes.data,1)
    t=lib.Timer(ctx); t.start(); gg.run(); t.stop(); ms=t.elapsed_ms()
    prof(buf.ctypes.data,1)
    tot=buf[:8].sum()
    print(shape,n,S,"%.1f ms"%ms, {nm:"%.1f%%"%(100*buf[i]/tot) for i,nm in enumerate(names)}, "cycles/wave-sweep %.3g"%(tot/(-(-flat['num_groups']//64))/700))
    print("   extra Mcycles: fill %.1f swap %.1f nested %.1f init %.1f (shuffle %.1f select %.1f compact+multi %.1f nsu_total %d) sections %.1f  wall %.1f"%(buf[12]/1e6,buf[13]/1e6,buf[14]/1e6,buf[15]/1e6,buf[8]/1e6,buf[9]/1e6,buf[10]/1e6,buf[11],tot/1e6, ms*2.4e3/1e3*1e3/1e3))
    #print("   lane0: visits %d cand/visit %.1f umiss/visit %.2f mmiss/visit %.2f multi-kmers/miss %.1f"%(buf[11],buf[8]/buf[11],buf[9]/buf[11],buf[10]/buf[11],buf[12]/max(1,buf[10])))
    gg.close()
