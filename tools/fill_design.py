"""Fill the measured sections of DESIGN.md (markers @@KERNEL_TABLE@@, @@GIBBS_ANALYSIS@@, @@MEASUREMENT@@ or the text between the
<!-- measured:NAME --> ... <!-- /measured:NAME --> pairs they become) from the summaries in profiles/ (tools/profile_round.sh,
tools/sq_counters.sh).  usage: python tools/fill_design.py [tag]"""
import json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
P = os.path.join(ROOT, "profiles")
bench = json.load(open(os.path.join(P, f"{tag}_bench_under_rocprof.json")))
trace = json.load(open(os.path.join(P, f"{tag}_bench_kernel_trace.json")))
fetch = json.load(open(os.path.join(P, f"{tag}_FETCH_SIZE_pmc.json")))
write = json.load(open(os.path.join(P, f"{tag}_WRITE_SIZE_pmc.json")))


def sq(cls):
    d = {}
    for line in open(os.path.join(P, f"{tag}_sq_{cls}.txt")):
        a = line.split()
        if len(a) == 2 and a[0].startswith("SQ_"):
            d[a[0]] = float(a[1])
        elif line.startswith("{"):
            d["ms"] = min(json.loads(line)["ms"])
            d["clusters"] = json.loads(line)["clusters"]
    return d


SQ = {c: sq(c) for c in "ABCD"}
g = [r for r in trace if r["kernel"].startswith("gibbs") and r["dur_ms"] > 1]
steps = {}
for r in g:
    steps.setdefault(round(r["start_ms"], -1), []).append(r)
last = steps[sorted(steps)[-1]]
classes = ", ".join(f"{int(r['grid_x']) // 64} tiles ({r['kernel']}): {r['dur_ms'] / 1e3:.2f} s" for r in sorted(last, key=lambda r: -r["dur_ms"]))
KiB = 1024.0
rd = sum(r["sum"] for r in fetch if r["kernel"].startswith("gibbs")) * KiB
wr = sum(r["sum"] for r in write if r["kernel"].startswith("gibbs")) * KiB
kmc_rd = sum(r["sum"] for r in fetch if r["kernel"].startswith("kmc_") or r["kernel"].startswith("rocprim")) * KiB
kmc_wr = sum(r["sum"] for r in write if r["kernel"].startswith("kmc_") or r["kernel"].startswith("rocprim")) * KiB
rf, rk, cpu = bench["roofline"], bench["roofline_kmer_match"], bench["cpu_baseline"]
alg = rf["algorithmic_bytes"]
sched_s = rf["avg_launch_ms"] / 1e3
valu = sum(SQ[c]["SQ_INSTS_VALU"] for c in "ABCD")
valu_s = valu * 4 / 1024 / 2.4e9


def e(x):
    m, ex = f"{x:.2e}".split("e")
    return f"{float(m):.2f}×10^{int(ex)}"


kernel_table = f"""| Kernel | Work per launch | Bound | Algorithmic bytes (SURVEY §8d) | Measured ({tag}, MI355X, `profiles/`) |
|---|---|---|---|---|
| KMC scan = `kmc_route_kernel` → rocPRIM radix sort (16 bits) → `kmc_probe_kernel` → `kmc_apply_kernel`, per chunk of 2^26 records | R records: decode + ntHash + route key (12 B + 2 B per record written) → sorted by sub-filter → one workgroup per sub-filter copies its ≤ 64 KB slice of the filter to LDS and probes from there, hits go through an LDS queue into a dense list → hits only: table find-or-insert + saturating count | HBM stream (13 B records in, 14 B route records out and back through the sort) | 15.9 B/record pure (13 B record + E[probes]·1 B + 2 % × 34 B table update); the routed form moves 13 + 3×14 B ≈ 55 B/record | {rk['launches_per_step']} scans per step into an emptied table: 2×10^8 records in {rk['insert_launch_ms']:.1f} ms (the inserting scan) / {rk['find_launch_ms']:.1f} ms (the two finding scans) → **{e(bench['kmer_matches_per_sec'])} records/s** = {rk['achieved']:.0f} GB/s algorithmic ({100 * rk['frac']:.1f} % of 8 TB/s); round 1's direct kernel: 121 ms cold (same-address atomics on the key counter, divergent hit path), 14–30 ms warm.  PMC per step (3 scans): {kmc_rd / 1e9:.0f} GB fetched + {kmc_wr / 1e9:.0f} GB written = {(kmc_rd + kmc_wr) / 3 / 2e8:.0f} B/record.  CPU oracle {e(cpu['kmer_matches_per_sec_1core'])} records/s/core |
| `gibbs_kernel` + `gibbs_simple_kernel` | G groups × 20 chains × 350 sweeps; one launch per LDS class, concurrent | wavefront slots × per-tile latency of a sequential sampler (below); no dense contraction → no MFMA | per (cluster, chain): `K·H + K·(S+4) + 0.1K·4 + 2(13H+4S) + 2·2·2496` B (inputs once, state in/out once): {alg / 1e9:.0f} GB for the bench batch | {bench['config']['groups_per_gpu']} groups / {bench['config']['clusters_per_gpu']} clusters, S = 3: {sched_s:.2f} s per schedule → **{e(bench['gibbs_kernel_cluster_sweeps_per_sec'])} cluster-sweeps/s**, {bench['gpu_over_cpu_allcores']:.0f}× the {cpu['cores']}-thread oracle run ({e(cpu['value'])}); launches of the last step: {classes}.  Algorithmic {rf['achieved']:.0f} GB/s = {100 * rf['frac']:.2f} % of HBM peak — the HBM fraction of this kernel is tiny by construction.  PMC traffic {(rd + wr) / 1e12:.1f} TB per schedule ({rd / 1e12:.1f} read, {wr / 1e12:.1f} written) = {(rd + wr) / alg:.0f}× the algorithmic floor (round 1: 69×) |
| `bt_paths_*` kernels | every k-mer window of every best path of every cluster of a unit | HBM random access (atomic find-or-insert into two open-addressing indexes) | per window: 1 B text + 17 B k-mer + 2×(20–28 B index entry) + ≈21+S B table probe | {bench['graph_stages']['clusters']} clusters, {e(bench['graph_stages']['kmer_windows'])} windows: enumerate {e(bench['graph_stages']['enumerate_windows_per_sec'])} windows/s, Bloom insert {e(bench['graph_stages']['bloom_insert_windows_per_sec'])}/s, classify {e(bench['graph_stages']['classify_windows_per_sec'])}/s, candidates {e(bench['graph_stages']['candidates_windows_per_sec'])}/s (host assembly included) |
| `mg_order_kernel` + the multigroup kernels | one lane per group replays the group's `unordered_set`; the rest one lane per k-mer | latency (sequential container replay per group) / HBM random access | ≈ 60 B per distinct (group, k-mer) | 50 000 single-cluster groups, 1.7×10^7 windows: {e(bench['graph_stages'].get('multigroup_windows_per_sec', 0))} windows/s (`bench.py`: `graph_stages.multigroup_windows_per_sec`, host wall-clock incl. the scratch allocations; 0.7–1.6×10^8 between runs); parity-tested with undersized filters and two units |
| `find_paths_kernel` | per sample: the best-path search of every cluster of a unit | latency of dependent accesses + random probes into the sample Bloom filter | per vertex nucleotide and live path: one Bloom probe chain | {e(bench['graph_stages']['find_sample_paths_clusters_per_sec'])} clusters/s per sample |
| `kmer_stats_kernel`, `summary_kernel`, `bloom_*`, `table_*`, `intercluster_kernel`, `classify_kernel`, `kmers_from_sequence_kernel` | one slot / k-mer / position per lane, grid-stride | HBM stream or random access | 4 + spad + 4 B per slot; 8 B per (cluster, sample); ≈3 B/position; 21+S B per path k-mer | parity-tested; `summary_kernel` {[r['dur_ms'] for r in trace if r['kernel'] == 'summary_kernel'][-1]:.0f} ms per step |"""

rows = []
for c in "ABCD":
    d = SQ[c]
    rows.append(f"| {c} | {d['ms'] / 1e3:.2f} | {d['SQ_WAIT_ANY'] / d['SQ_WAVE_CYCLES']:.2f} | {d['SQ_ACTIVE_INST_ANY'] / d['SQ_WAVE_CYCLES']:.2f} | {d['SQ_INSTS_VALU']:.3g} | {d['SQ_INSTS_VALU'] / (d['clusters'] * 7000):.0f} | {(d['SQ_INSTS_FLAT'] + d['SQ_INSTS_VMEM_RD'] + d['SQ_INSTS_VMEM_WR'] + d['SQ_INSTS_LDS']):.3g} |")
gibbs_analysis = f"""**What bounds the Gibbs launch: wavefront slots × tile latency.**  Both kernels need 256 VGPRs, i.e. two wavefronts per SIMD, 2 048 on the chip
(fewer registers cost more in spills than the third wavefront gives back, §4 end).  A tile is one wavefront running a strictly sequential
program — 7 000 sweeps — whose duration is set by dependent LDS / HBM accesses and by how many DIFFERENT groups it carries (their
control flow diverges in the data: rejection loops, set sizes, candidate counts), not by bandwidth.  The launch therefore behaves like
list scheduling of {sum(int(r['grid_x']) // 64 for r in last)} tiles on 2 048 slots: its time is (Σ tile durations) / 2 048 plus the tail, and the work of this round was shortening tile
durations.  SQ counters per shape class, each class of the bench batch run alone (`profiles/{tag}_sq_*.txt`, `tools/sq_counters.sh`; times under
`--pmc`, which serialises the launch classes):

| class | s (alone, under PMC) | SQ_WAIT_ANY / SQ_WAVE_CYCLES | SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES | VALU instructions | per cluster-sweep | memory instructions (flat + vmem + lds) |
|---|---|---|---|---|---|---|
{chr(10).join(rows)}

Round 1 measured 0.68 waiting for both A and C.  The two-haplotype class (90 % of the clusters) is close to issue-bound: its VALU
instructions alone, at four cycles per wave64 instruction on 1 024 SIMDs, are {SQ['A']['SQ_INSTS_VALU'] * 4 / 1024 / 2.4e9:.2f} s of its {SQ['A']['ms'] / 1e3:.2f} s.  Summed over the classes the VALU work
is {valu:.3g} instructions = {valu_s:.1f} s at full issue rate against the {sched_s:.2f} s the launch takes: the launch runs at {100 * valu_s / sched_s:.0f} % of the chip's VALU issue
rate, the rest is the latency of the narrow tiles that the two wavefronts per SIMD cannot hide.

**HBM traffic** is {(rd + wr) / 1e12:.1f} TB per schedule = {(rd + wr) / sched_s / 1e9:.0f} GB/s, {100 * (rd + wr) / sched_s / 8e12:.0f} % of peak — {(rd + wr) / (bench['config']['clusters_per_gpu'] * 7000):.0f} B per cluster-sweep, the same per cluster-sweep as in round 1 (2 960 B), so
the ratio to the algorithmic floor did not move ({(rd + wr) / alg:.0f}×; the VERDICT asked for < 10×).  The floor counts a cluster's generator states once
per chain (2 × 2 × 2 496 B / 350 sweeps = 28 B per sweep); the sampler consumes ≈ 30 raw words per two-haplotype sweep, and every generated word of
an HBM-resident mt19937 reads three state words at two places 397 words apart and writes one — ≈ 20 B of cache-line traffic per word even
with the draw-ahead rings (which removed the round TRIPS, not the bytes): {[r['sum'] for r in fetch if r['kernel'] == 'gibbs_simple_kernel'][0] * KiB / 1e12:.2f} TB read + {[r['sum'] for r in write if r['kernel'] == 'gibbs_simple_kernel'][0] * KiB / 1e12:.2f} TB written by `gibbs_simple_kernel` alone,
{([r['sum'] for r in fetch if r['kernel'] == 'gibbs_simple_kernel'][0] + [r['sum'] for r in write if r['kernel'] == 'gibbs_simple_kernel'][0]) * KiB / (SQ['A']['clusters'] * 7000):.0f} B per cluster-sweep.  The generator states of a 64-group tile are 320 KB, twice the LDS of a CU, so they cannot live on chip with one
group per lane; an exact mt19937 stream with less traffic needs fewer, wider-shared generators, which the reference's one-generator-per-cluster
seeding rules out.  The narrow tiles add their dense-table reads, the subset arrays and the statistics cells (≈ 25 KB per cluster-sweep of the
many-candidate and nested clusters).  At 18 % of peak the launch is not bandwidth-bound, but this traffic is what the waiting wavefronts wait for."""

S10 = None
s10p = os.path.join(P, f"{tag}_bench_samples10.json")
if os.path.exists(s10p):
    S10 = json.loads(open(s10p).read().strip().splitlines()[-1])
pc = bench["kmer_match_from_host_memory"]
measurement = f"""`bench.py` (contract of the task statement): a step = empty the count table + S KMC scans (one per sample: the first inserts, the others
find) + the full default Gibbs schedule + the posterior-summary gather.  N=1 workload = BASELINE `configs[2]`, the largest single-GPU
configuration ("GRCh38 whole genome, CEU trio"): S = 3 and one launch-sized slice of the unit — {bench['config']['groups_per_gpu']} variant-cluster groups in the WGS-like
mixture of BASELINE.md §3 (90 % two-haplotype SNV/indel groups "A", 8 % multi-variant clusters "B", 1.5 % nested SV groups "C", 0.5 %
many-candidate clusters "D"), every structure with its own dimensions and every group with its own truth genotypes and counts
(`bayestyper_amd/synth.py`) — and a 2×10^8-record KMC stream per sample; inputs resident in HBM.  `value` = cluster-sweeps/s over the whole
step; `roofline` for the dominant kernel (the Gibbs launch), `roofline_kmer_match` for one KMC scan; `roofline.traffic` is null in the line (not
measurable from inside the process) and comes from the separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of the same command in
`profiles/` (KiB → bytes, no ×2 correction: that factor is calibrated for wide coalesced streams, most accesses here are narrow).
`cpu_baseline` = the oracle on all host cores of the GPU box ({cpu['cores']} threads pulling groups from a shared queue) on a bounded sample of
the same mixture ({cpu['sample'].split(' groups')[0]} groups, full schedule, ≈12 s), kind "port".  Kernel times are HIP events on the stream the
kernels are launched on (`bt_timer_*`; the launch classes join that stream through events, so one interval spans all concurrent launches).

Final build, one MI355X (`profiles/{tag}_bench_under_rocprof.json`, i.e. under `rocprofv3 --kernel-trace --stats`; `{tag}_bench_kernel_trace.json` has one row
per dispatch): **{e(bench['value'])} cluster-sweeps/s** ({bench['ms_per_step'] / 1e3:.2f} s per step; the Gibbs launch {sched_s:.2f} s by HIP events, the trace's longest class
{max(r['dur_ms'] for r in last) / 1e3:.2f} s), {bench['gpu_over_cpu_allcores']:.0f}× the {cpu['cores']}-thread oracle run; {e(bench['kmer_matches_per_sec'])} KMC records/s.  The round-1 build on this workload: 1.5×10^8
cluster-sweeps/s (`gpurun_out/r02a`, first measurement of this round) — **{bench['value'] / 1.5e8:.1f}×**; round 1's own bench (configs[1]-like, S = 1, identical
structures) is not comparable.  {sum(int(r['grid_x']) // 64 for r in last)} tiles, {bench['gibbs_device_bytes'] / 1e9:.0f} GB of sampler state.
""" + (f"""`python bench.py --samples 10` (the same mixture, ten samples): {e(S10['value'])} cluster-sweeps/s, {S10['ms_per_step'] / 1e3:.1f} s per step, {S10.get('gpu_over_cpu_allcores', 0):.0f}× the
{S10['cpu_baseline']['cores']}-thread oracle run — the north star's 10-sample target is ≥ 20×.  Thirty samples (`--samples 30 --groups 100000`): 37.3 s per step, 1.9×10^7
cluster-sweeps/s, 20.5 GB.
""" if S10 else "") + f"""
PCIe: `bt_kmc_scan_run` takes device pointers; `bt_kmc_scan_run_host` streams a host-resident (memory-mapped) payload through two pinned staging
buffers and a copy stream, overlapping host copy, transfer and scan: **{e(pc['records_per_sec'])} records/s = {pc['host_gbytes_per_sec']:.0f} GB/s** from pageable host memory, i.e. the
PCIe Gen5 x16 link, against {e(bench['kmer_matches_per_sec'])} with the stream resident in HBM; reported beside `value`, never as `value`."""

p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
for name, text in (("KERNEL_TABLE", kernel_table), ("GIBBS_ANALYSIS", gibbs_analysis), ("MEASUREMENT", measurement)):
    block = f"<!-- measured:{name} -->\n{text}\n<!-- /measured:{name} -->"
    if f"@@{name}@@" in s:
        s = s.replace(f"@@{name}@@", block)
    else:
        s = re.sub(rf"<!-- measured:{name} -->.*?<!-- /measured:{name} -->", lambda m: block, s, flags=re.S)
open(p, "w").write(s)
print("DESIGN.md filled from", tag)
