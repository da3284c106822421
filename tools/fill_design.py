"""Fill the measured sections of DESIGN.md (the text between <!-- measured:NAME --> ... <!-- /measured:NAME -->) and the @@R3_*@@ markers of DESIGN.md / README.md
from the summaries in profiles/ (tools/profile_round.sh).  usage: python tools/fill_design.py [tag]"""
import json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
P = os.path.join(ROOT, "profiles")
bench = json.load(open(os.path.join(P, f"{tag}_bench_under_rocprof.json")))
trace = json.load(open(os.path.join(P, f"{tag}_bench_kernel_trace.json")))
traffic = json.load(open(os.path.join(P, f"{tag}_traffic.json")))


def sq(cls, S=3):
    """counters of one class's single-schedule passes (tools/sq_counters.sh: '<kernel> <counter> <value> dispatches <n>' per kernel, summed over the
    sampling kernels) + the launch time under --pmc"""
    d = {}
    p = os.path.join(P, f"{tag}_sq_{cls}_S{S}.txt")
    if not os.path.exists(p):
        return None
    for line in open(p):
        a = line.split()
        if len(a) == 5 and a[3] == "dispatches":
            d[a[1]] = d.get(a[1], 0.0) + float(a[2])
        elif line.startswith("{"):
            j = json.loads(line)
            d["ms"] = min(j["ms"]) if "ms" not in d else min(d["ms"], min(j["ms"]))
            d["clusters"] = j["clusters"]
    return d if "SQ_WAVE_CYCLES" in d else None


def e(x):
    m, ex = f"{x:.2e}".split("e")
    return f"{float(m):.2f}×10^{int(ex)}"


SQ = {c: sq(c) for c in "ABCD"}
g = sorted((r for r in trace if r["kernel"].startswith("gibbs") and r.get("start_ms") is not None and r["dur_ms"] > 500), key=lambda r: r["start_ms"])
# the launch classes of the main step: the grids of the run's first schedule (the sub-records launch other batches later); `last` = its last schedule
first_grids = []
for r in g:
    if r["grid_x"] in first_grids:
        break
    first_grids.append(r["grid_x"])
main = [r for r in g if r["grid_x"] in first_grids]
last = main[-len(first_grids):] if first_grids else g[-4:]
classes = ", ".join(f"{int(r['grid_x']) // 64} tiles ({r['kernel']}, scratch {r['scratch']} B/lane): {r['dur_ms'] / 1e3:.2f} s" for r in sorted(last, key=lambda r: -r["dur_ms"]))
rf, rk, cpu = bench["roofline"], bench["roofline_kmer_match"], bench["cpu_baseline"]
alg = rf["algorithmic_bytes"]
sched_s = rf["avg_launch_ms"] / 1e3
gb, gr, gw = traffic["gibbs_bytes_per_schedule"], traffic["gibbs_fetch_bytes"], traffic["gibbs_write_bytes"]
kb = traffic["kmc_bytes_per_scan"]
R = bench["config"]["kmc_records_per_gpu_per_sample"]
detail = {}
for r in traffic["detail"]:
    detail.setdefault(r["kernel"], 0.0)
    detail[r["kernel"]] += r["bytes"]
gs = bench["graph_stages"]
s10, ng, c4 = bench.get("samples10"), bench.get("noise_genotyping"), bench.get("kmer_match_c4_subfilters")
ng10 = bench.get("noise_genotyping_samples10")
pc = bench["kmer_match_from_host_memory"]
C = bench["config"]["clusters_per_gpu"]

kernel_table = f"""| Kernel | Work per launch | Bound | Algorithmic bytes (SURVEY §8d) | Measured ({tag}, MI355X, `profiles/`) |
|---|---|---|---|---|
| KMC scan = `kmc_partition_kernel` → `kmc_probe_bucket_kernel` → `kmc_apply_kernel`, per chunk of 2^26 records (every sub-filter size; rounds 2–3 sent sub-filters above 4 KB through a rocPRIM radix sort and LDS-staged sub-filters) | R records: a workgroup stages a slab of 4096 records in LDS, computes the ntHash straight from the raw record bytes (one 256-entry LDS table per suffix byte = four symbols, the prefix's part once per slab: 72 VALU instructions per record; round 3 assembled the k-mer first: ≈ 350), counting-sorts the slab's 12-byte route records by the upper 8 route bits and writes every bucket's run into its stripe of the bucket's region (stripe = workgroup mod 8 = XCD: one `atomicAdd` per bucket and slab, 2 048 per cursor and chunk instead of 16 384) → workgroups mapped so that an XCD works through one bucket at a time probe the bucket's 256 sub-filters (0.5 MB at the WGS shape: L2; 9 MB for a ten-sample path filter: Infinity Cache), four records per lane with the probes of a round issued together, two rounds per record in its lane and the survivors compacted through an LDS list (one per lane), hits go through an LDS queue into a dense list → hits only: the record as aligned words, table find-or-insert on the packed slots (§3: one sector, one burst) + saturating count | HBM stream (13 B records in, 12 B route records out and in once) | 15.9 B/record pure (13 B record + E[probes]·1 B + 2 % × 34 B table update); the partitioned form moves 13 + 2×12 B = 37 B/record + the hits' table traffic (round 2's sorted form: 13 + 3×14 B ≈ 55 B, measured 102 B) | SURVEY §8d's stream, {rk['launches_per_step']} scans per step into an emptied table: {e(R)} records in {rk['insert_launch_ms']:.0f} ms (the inserting scan, {e(rk['bloom_hits_per_scan'])} hits) / {rk['find_launch_ms']:.0f} ms (the finding scans) → **{e(bench['kmer_matches_per_sec'])} records/s** = {rk['achieved']:.0f} GB/s algorithmic ({100 * rk['frac']:.1f} % of 8 TB/s); round 3: 74 / 40 ms, 1.94×10^10 records/s.  Counter traffic {kb / 1e9:.0f} GB per scan = {kb / R:.0f} B/record ({kb / R / 15.9:.1f}× the pure floor; round 3: 52 B/record; the sorted form: 102 B/record).  Sub-filters of C4 size ({c4['path_filter'] if c4 else 'n/a'}): {e(c4['records_per_sec']) if c4 else 'n/a'} records/s (a finding scan).  CPU oracle: {e(cpu['kmer_matches_per_sec_single_producer'])} records/s with the reference's single producer, {e(cpu['kmer_matches_per_sec_parallel_decode'])} with every core decoding its own range |
| `gibbs_hot_kernel` (`gibbs_kernel` for tiles that do not keep every vertex in LDS) + `gibbs_simple_kernel` | G groups × 20 chains × 350 sweeps; one launch per LDS class, concurrent | wavefront slots × per-tile latency of a sequential sampler (below); no dense contraction → no MFMA | per (cluster, chain): `K·H + K·(S+4) + 0.1K·4 + 2(13H+4S) + 2·2·2496` B (inputs once, state in/out once): {alg / 1e9:.0f} GB for the bench batch | {bench['config']['groups_per_gpu']} groups / {C} clusters, S = 3: {sched_s:.2f} s per schedule → **{e(bench['gibbs_kernel_cluster_sweeps_per_sec'])} cluster-sweeps/s**, {bench['gpu_over_cpu_allcores']:.0f}× the {cpu['cores']}-thread oracle run ({e(cpu['value'])}; one core {e(cpu['one_core']['value'])}); launches of one step: {classes}.  Algorithmic {rf['achieved']:.0f} GB/s = {100 * rf['frac']:.2f} % of HBM peak — tiny by construction.  Counter traffic **{gb / 1e12:.2f} TB per schedule** ({gr / 1e12:.2f} read, {gw / 1e12:.2f} written) = {gb / alg:.0f}× the algorithmic floor (round 3: 3.58 TB, 20×; round 2: 12.1 TB, 68×), {gb / (C * 7000):.0f} B per cluster-sweep; round 3: 5.63 s per schedule |
| `build_tiles_kernel` (`bt_gibbs_create`) | one workgroup per cluster scatters the cluster's slices of the flat batch into its tile's rows | HBM stream | the batch once in, once out | 600 320 groups: 0.5 s for the whole `bt_gibbs_create` (planning on the host, upload, build) |
| `noise_update_kernel` (`bt_gibbs_noise_chain`) | per iteration of a noise driver: S × 256 histogram → S gamma draws (one thread: the stream is sequential) → S × 256 Poisson log-pmf entries | latency (a few µs per iteration; it replaces a host round trip) | 2 KB · S in, 2 KB · S out | thirty samples, 2 000 groups: {f"{ng['iterations_per_sec']:.0f} iterations/s, {ng['noise_over_default_time']:.1f}× the default mode's time on the same batch (the caches are cleared every iteration)" + (f", {ng['gpu_over_cpu_allcores']:.0f}× the oracle's estimateNoiseAndGenotypes on all cores" if 'gpu_over_cpu_allcores' in ng else '') if ng else 'n/a'}; ten samples, 100 000 groups: {f"{ng10['iterations_per_sec']:.0f} iterations/s, {ng10['noise_over_default_time']:.1f}× the default mode's time" if ng10 else 'n/a'} |
| `bt_paths_*` kernels | every k-mer window of every best path of every cluster of a unit | HBM random access (atomic find-or-insert into two open-addressing indexes) | per window: 1 B text + 17 B k-mer + 2×(20–28 B index entry) + ≈21+S B table probe | {gs['clusters']} clusters, {e(gs['kmer_windows'])} windows: enumerate {e(gs['enumerate_windows_per_sec'])} windows/s, Bloom insert {e(gs['bloom_insert_windows_per_sec'])}/s, classify {e(gs['classify_windows_per_sec'])}/s, candidates {e(gs['candidates_windows_per_sec'])}/s (host wall-clock, fetch to host arrays included) |
| `mg_order_kernel` + the multigroup kernels | one lane per group replays the group's `unordered_set`; the rest one lane per k-mer | latency (sequential container replay per group) / HBM random access | ≈ 60 B per distinct (group, k-mer) | {e(gs.get('multigroup_windows_per_sec', 0))} windows/s for 50 000 single-cluster groups; counters: @@MG_ORDER@@ |
| `find_paths_kernel` | per sample: the best-path search of every cluster of a unit, one lane per cluster | latency of dependent accesses + random probes into the sample Bloom filter | per vertex nucleotide and live path: one Bloom probe chain | {e(gs['find_sample_paths_clusters_per_sec'])} clusters/s per sample; counters: @@FIND_PATHS@@ |
| `kmer_stats_kernel`, `summary_kernel`, `bloom_*`, `table_*`, `intercluster_kernel`, `classify_kernel`, `kmers_from_sequence_kernel`, `export_count_rows_kernel`, `merge_count_rows_kernel` | one slot / k-mer / position / row per lane, grid-stride | HBM stream or random access | 4 + spad + 4 B per slot; 8 B per (cluster, sample); ≈3 B/position; 21+S B per path k-mer; 16 + spad B per count row | parity-tested |"""

issue_p = os.path.join(P, f"{tag}_issue.json")
issue = json.load(open(issue_p))["gibbs"] if os.path.exists(issue_p) else None
MIX = sq("mix")
S10 = {c: sq(c, 10) for c in "ABCD"}


def row(c, d):
    wc = d["SQ_WAVE_CYCLES"]
    f64 = sum(d.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"))
    return (f"| {c} | {d['ms'] / 1e3:.2f} | {d['SQ_WAIT_ANY'] / wc:.2f} | {d['SQ_ACTIVE_INST_ANY'] / wc:.2f} | {d['SQ_INSTS_VALU']:.3g} | {d['SQ_INSTS_VALU'] / (d['clusters'] * 7000):.0f} | "
            f"{100 * f64 / d['SQ_INSTS_VALU']:.0f} % | {d.get('SQ_INSTS_LDS', 0):.3g} | {(d.get('SQ_INSTS_FLAT', 0) + d.get('SQ_INSTS_VMEM_RD', 0) + d.get('SQ_INSTS_VMEM_WR', 0)):.3g} | "
            f"{(d.get('FETCH_SIZE', 0) + d.get('WRITE_SIZE', 0)) * 1024 / 1e9:.0f} |")


rows3 = [row(c, SQ[c]) for c in "ABCD" if SQ[c]] + ([row("mixture", MIX)] if MIX else [])
rows10 = [row(c, S10[c]) for c in "ABCD" if S10[c]]
simple = detail.get("gibbs_simple_kernel", 0.0)
general = detail.get("gibbs_kernel", 0.0) + detail.get("gibbs_hot_kernel", 0.0)
issue_txt = ""
if issue:
    cyc = issue["valu_issue_cycles_per_schedule"]
    issue_txt = (f"  Priced with the per-instruction issue cycles of MI355X_MICROARCH.md (2 for 32-bit wave64 VALU, 4 for f64 add / mul / fma and 64-bit integer, 16 / 8 for "
                 f"f64 / f32 transcendentals) the schedule's {issue['valu_insts_per_schedule']:.3g} VALU instructions are {cyc:.3g} SIMD cycles = {cyc / 1024 / 2.4e9:.2f} s of the chip's 1 024 SIMDs at 2.4 GHz "
                 f"against the {sched_s:.2f} s of the launch: `roofline.issue_frac` = {cyc / 1024 / 2.4e9 / sched_s:.2f} ({issue['valu_insts_per_cluster_sweep']:.0f} VALU instructions per cluster-sweep; "
                 f"MFMA busy cycles {issue['mfma_busy_cycles']:.0f}; `profiles/{tag}_issue.json`).")
gibbs_analysis = f"""**What bounds the Gibbs launch** (round 4, single-schedule counter passes: `profiles/{tag}_sq_*_S3.txt`, `tools/sq_counters.sh`; times under `--pmc`).  `gibbs_simple_kernel`
(two-haplotype tiles) holds three wavefronts per SIMD (168 VGPRs), `gibbs_hot_kernel` two (256 VGPRs, {max(int(r['scratch']) for r in last)} B of scratch per lane).  A tile is one wavefront running a
strictly sequential program — 7 000 sweeps — so a schedule's length is wavefront slots × per-tile latency: in the bench step the three `gibbs_hot_kernel` classes fill the chip
for {max(r['dur_ms'] for r in last if 'hot' in r['kernel']) / 1e3:.2f} s (≈ 3 rounds of ≈ 1.2 s tiles on 2 048 slots) and the two-haplotype class runs in what they leave and after them.

| class (S = 3) | s (alone, under PMC) | SQ_WAIT_ANY / SQ_WAVE_CYCLES | SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES | VALU instructions | per cluster-sweep | f64 share of VALU | LDS instructions | flat + vmem instructions | counter traffic GB |
|---|---|---|---|---|---|---|---|---|---|
{chr(10).join(rows3)}

(round 3, the same batch, corrected for that round's 2× double count: 127 / 1 020 / 2 710 / 5 710 VALU instructions per cluster-sweep and 1.74 / 3.10 / 4.00 / 2.34 s alone.){issue_txt}

Ten samples (100 000 groups; the classes alone):

| class (S = 10) | s | waiting | issuing | VALU instructions | per cluster-sweep | f64 share | LDS instructions | flat + vmem instructions | counter traffic GB |
|---|---|---|---|---|---|---|---|---|---|
{chr(10).join(rows10)}

**HBM traffic** (counter passes of the default bench command, `profiles/{tag}_traffic.json`, the file `bench.py` fills `roofline.traffic` from): {gb / 1e12:.2f} TB per schedule =
{gb / sched_s / 1e9:.0f} GB/s, {100 * gb / sched_s / 8e12:.0f} % of peak ({gb / alg:.0f}× the algorithmic {alg / 1e9:.0f} GB; round 3: 3.58 TB, 20×); `gibbs_hot_kernel` (+ `gibbs_kernel`'s non-sampling launches) {general / 1e12:.2f} TB,
`gibbs_simple_kernel` {simple / 1e12:.2f} TB = {simple / (SQ['A']['clusters'] * 7000) if SQ['A'] else 0:.0f} B per cluster-sweep (round 3: 1.91 TB, 506 B: the mt19937 states now move in aligned 16-byte chunks and the
two-haplotype sweep keeps its state in registers)."""
measurement = f"""`bench.py` (contract of the task statement): a step = empty the count table + S KMC scans (one per sample: the first inserts, the others
find) + the full default Gibbs schedule + the posterior-summary gather.  N=1 workload = BASELINE `configs[2]`, the largest single-GPU
configuration ("GRCh38 whole genome, CEU trio"): S = 3 and one launch-sized slice of the unit — {bench['config']['groups_per_gpu']} variant-cluster groups in the WGS-like
mixture of BASELINE.md §3 (90 % two-haplotype SNV/indel groups "A", 8 % multi-variant clusters "B", 1.5 % nested SV groups "C", 0.5 %
many-candidate clusters "D"), every structure with its own dimensions and every group with its own truth genotypes and counts
(`bayestyper_amd/synth.py`) — and SURVEY §8d's KMC stream per sample ({e(R)} records of 13 B, {e(bench['config'].get('path_kmers', 5e7))} path k-mers in a fpr-1e-4 ThreadedKmerBloom, 2 % hits); inputs
resident in HBM.  `value` = cluster-sweeps/s over the whole step; `roofline` for the dominant kernel (the Gibbs launch), `roofline_kmer_match` for one KMC scan;
`roofline.traffic` = FETCH_SIZE + WRITE_SIZE of the launch from the committed counter passes of the same command (`profiles/{tag}_traffic.json`: separate `--pmc` runs,
KiB → bytes, no ×2 correction — that factor is calibrated for wide coalesced streams, most accesses here are narrow), filled in when the library was built from the
sources the passes ran on (`source_hash`).  `cpu_baseline` = the oracle (kind "port") on all host cores of the GPU box on a sample sized from a probe for ≈ 25 s
({cpu['sample'].split(' groups')[0]} groups, {cpu['cores']} threads pulling groups from a shared queue), on ONE core ({cpu['one_core']['sample']}), and for the scan the reference's
single producer beside every core decoding its own range.  Kernel times are HIP events on the stream the kernels are launched on (`bt_timer_*`; the launch classes
join that stream through events, so one interval spans all concurrent launches).

Final build, one MI355X (`profiles/{tag}_bench_under_rocprof.json`, i.e. under `rocprofv3 --kernel-trace --stats`; `{tag}_bench_kernel_trace.json` has one row per dispatch
of at least 0.1 ms): **{e(bench['value'])} cluster-sweeps/s** ({bench['ms_per_step'] / 1e3:.2f} s per step, of which the Gibbs launch {sched_s:.2f} s by HIP events and the three scans
{rk['launches_per_step'] * rk['avg_launch_ms'] / 1e3:.2f} s), {bench['gpu_over_cpu_allcores']:.0f}× the {cpu['cores']}-thread oracle run and {bench['gibbs_kernel_cluster_sweeps_per_sec'] / cpu['one_core']['value']:.0f}× one core; {e(bench['kmer_matches_per_sec'])} KMC records/s
({bench['kmer_matches_per_sec'] / cpu['kmer_matches_per_sec_single_producer']:.0f}× the single-producer scan, {bench['kmer_matches_per_sec'] / cpu['kmer_matches_per_sec_parallel_decode']:.0f}× the parallel one).  Round 2 on this batch: 8.21 s per step.  {bench['gibbs_device_bytes'] / 1e9:.0f} GB of sampler state.
Sub-records of the same line: **ten samples** (`samples10`: {(str(s10['groups']) + ' groups of the mixture') if s10 else ''}, the north star's sample count): {e(s10['cluster_sweeps_per_sec']) if s10 else 'n/a'} cluster-sweeps/s, {f"{s10['gpu_over_cpu_allcores']:.0f}" if s10 else 'n/a'}× the
{cpu['cores']}-thread oracle run (target ≥ 20×); **thirty samples, `--noise-genotyping`** through the C++ engine (`noise_genotyping`): {f"{e(ng['noise_genotyping_cluster_sweeps_per_sec'])} cluster-sweeps/s against {e(ng['default_mode_cluster_sweeps_per_sec'])} in the default mode on the same batch" + (f" and {e(ng['cpu_allcores_cluster_sweeps_per_sec'])} for the oracle's estimateNoiseAndGenotypes on {cpu['cores']} threads" if 'cpu_allcores_cluster_sweeps_per_sec' in ng else '') if ng else 'n/a'}; the ten-sample batch in that mode (`noise_genotyping_samples10`): {f"{e(ng10['noise_genotyping_cluster_sweeps_per_sec'])} cluster-sweeps/s, {ng10['noise_over_default_time']:.1f}× the default mode's time" if ng10 else 'n/a'};
**C4-sized sub-filters** (`kmer_match_c4_subfilters`): {e(c4['records_per_sec']) if c4 else 'n/a'} records/s.

PCIe: `bt_kmc_scan_run` takes device pointers; `bt_kmc_scan_run_host` streams a host-resident (memory-mapped) payload through two pinned staging
buffers and a copy stream, overlapping host copy, transfer and scan: **{e(pc['records_per_sec'])} records/s = {pc['host_gbytes_per_sec']:.0f} GB/s** from pageable host memory
against {e(bench['kmer_matches_per_sec'])} with the stream resident in HBM; reported beside `value`, never as `value`."""

p = os.path.join(ROOT, "docs", "HISTORY.md")   # (the long-form design notes with the measured tables; DESIGN.md is the short current design)
s = open(p).read()
for name, text in (("KERNEL_TABLE", kernel_table), ("GIBBS_ANALYSIS", gibbs_analysis), ("MEASUREMENT", measurement)):
    block = f"<!-- measured:{name} -->\n{text}\n<!-- /measured:{name} -->"
    s = re.sub(rf"<!-- measured:{name} -->.*?<!-- /measured:{name} -->", lambda m: block, s, flags=re.S)
gq = os.path.join(P, f"{tag}_sq_graph_stages.txt")
if os.path.exists(gq):
    cnt = {}
    for line in open(gq):
        a = line.split()
        if len(a) == 3 and a[0] in ("find_paths_kernel", "mg_order_kernel"):
            cnt[(a[0], a[1])] = float(a[2])
    for k, mark in (("find_paths_kernel", "@@FIND_PATHS@@"), ("mg_order_kernel", "@@MG_ORDER@@")):
        if (k, "SQ_WAVE_CYCLES") in cnt:
            txt = (f"{cnt.get((k, 'duration_ms'), 0):.1f} ms, waiting {cnt[(k, 'SQ_WAIT_ANY')] / cnt[(k, 'SQ_WAVE_CYCLES')]:.2f} of its wave cycles, issuing {cnt[(k, 'SQ_ACTIVE_INST_ANY')] / cnt[(k, 'SQ_WAVE_CYCLES')]:.2f}, "
                   f"{cnt.get((k, 'SQ_INSTS_VALU'), 0):.3g} VALU / {cnt.get((k, 'SQ_INSTS_VMEM_RD'), 0) + cnt.get((k, 'SQ_INSTS_FLAT'), 0):.3g} memory-read instructions, "
                   f"{(cnt.get((k, 'FETCH_SIZE'), 0) + cnt.get((k, 'WRITE_SIZE'), 0)) * 1024 / 1e9:.2f} GB of counter traffic (`profiles/{tag}_sq_graph_stages.txt`): latency-bound, one lane per cluster / group")
            s = s.replace(mark, txt)
marks = {"@@R3_GK_TRAFFIC@@": f"{general / 1e12:.1f}"}
for k, v in marks.items():
    s = s.replace(k, v)
open(p, "w").write(s)
rp = os.path.join(ROOT, "README.md")
r = open(rp).read()
for k, v in {"@@R3_VALUE@@": e(bench["value"]), "@@R3_STEP@@": f"{bench['ms_per_step'] / 1e3:.1f}", "@@R3_RATIO@@": f"{bench['gpu_over_cpu_allcores']:.0f}",
             "@@R3_S10@@": e(s10["cluster_sweeps_per_sec"]) if s10 else "n/a", "@@R3_S10R@@": f"{s10['gpu_over_cpu_allcores']:.0f}" if s10 else "n/a",
             "@@R3_KMC@@": e(bench["kmer_matches_per_sec"]), "@@R3_TRAFFIC@@": f"{gb / 1e12:.1f}"}.items():
    r = r.replace(k, v)
open(rp, "w").write(r)
print("DESIGN.md / README.md filled from", tag)
