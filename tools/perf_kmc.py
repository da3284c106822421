"""Time the KMC scan alone (bench.py's stream shape).  usage: perf_kmc.py [records] [path_kmers] ["name:ENV=VAL,ENV=VAL;name2:..."]
(path_kmers 50 000 000: 1.9 KB sub-filters, the WGS shape; 1 000 000 000: 36 KB sub-filters, a ten-sample path filter)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayestyper_amd import lib
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000_000
K, KMC_P, REC = 55, 7, 13
dev = torch.device("cuda", 0)
ctx = lib.Ctx(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
gen = torch.Generator(device=dev); gen.manual_seed(4)
records = torch.randint(0, 256, (R * REC + 16,), dtype=torch.uint8, device=dev, generator=gen)
records.view(-1)[REC - 1: R * REC: REC] = torch.randint(1, 200, (R,), dtype=torch.uint8, device=dev, generator=gen)
lut = (np.arange(4 ** KMC_P + 1, dtype=np.float64) * (R / 4 ** KMC_P)).astype(np.uint64); lut[-1] = R
scan = lib.KmcScan(ctx, K, KMC_P, 1, R, lut)
bloom = lib.Bloom.create(ctx, P + 1_000_000, 1e-4, K, threaded=True)
n_hit = int(R * 0.02); stride = max(1, R // n_hit)
CH = 50_000_000 // stride * stride
kmers = torch.zeros((CH, 2), dtype=torch.int64, device=dev); cnts = torch.zeros(CH, dtype=torch.int32, device=dev)
ins = 0
for a in range(0, R, CH):
    m = min(CH, R - a)
    lib.check(lib.bt_kmc_scan_decode(scan.h, records.data_ptr() + a * REC, a, m, kmers.data_ptr(), cnts.data_ptr()))
    mem = kmers[:m:stride][: max(0, n_hit - ins)].contiguous()
    if mem.shape[0]: lib.check(lib.bt_bloom_insert_batch(bloom.h, mem.data_ptr(), mem.shape[0]))
    ins += mem.shape[0]; torch.cuda.synchronize()
absent = torch.randint(-(2 ** 62), 2 ** 62, (max(P - n_hit, 1), 2), dtype=torch.int64, device=dev, generator=gen); absent[:, 1] &= (1 << 46) - 1
lib.check(lib.bt_bloom_insert_batch(bloom.h, absent.data_ptr(), absent.shape[0])); torch.cuda.synchronize()
del kmers, cnts, absent
print("sub-filter bytes:", bloom.info()["num_bits"] // 8, flush=True)
d_hits = torch.zeros(1, dtype=torch.int64, device=dev)
t = lib.Timer(ctx)
# variants: "name:ENV=VAL,ENV=VAL;name2:..." (argv[3]); default: the partitioned form and the direct kernel
variants = sys.argv[3] if len(sys.argv) > 3 else "partitioned:BT_KMC_ROUTED=1;direct:BT_KMC_ROUTED=0"
for var in variants.split(";"):
    name, _, envs = var.partition(":")
    sets = [e.split("=", 1) for e in envs.split(",") if e]
    for k, v in sets: os.environ[k] = v
    table = lib.Table(ctx, int(n_hit * 1.5), 3, K)
    d_hits.zero_()
    ms = []
    for rep in range(2):
        table.clear(); torch.cuda.synchronize()
        for smp in range(3):
            t.start(); scan.run(bloom, table, smp, records.data_ptr(), 0, R, d_hits.data_ptr()); t.stop(); ms.append(t.elapsed_ms())
    print(f"{name}: scans {[round(x, 1) for x in ms]} ms -> {3 * R / (sum(ms[3:]) * 1e-3):.3e} records/s over the second round's three scans; hits {int(d_hits.item())} keys {table.status()['num_keys']}", flush=True)
    table.close()
    for k, v in sets: os.environ.pop(k, None)
