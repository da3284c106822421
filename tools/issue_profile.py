"""profiles/<tag>_issue.json from the single-schedule SQ counter passes of tools/sq_counters.sh (class '+': the bench's whole mixture).
usage: python tools/issue_profile.py <tag> [S]     (reads gpurun_out/summ_<tag>/<tag>_sq_mix_S<S>.txt)

Issue-side roofline of the Gibbs launch: every VALU instruction of the schedule priced with its issue cycles on a SIMD (MI355X_MICROARCH.md,
per-instruction table: 32-bit wave64 VALU 2 cycles; double-precision add / mul / fma 4; transcendental 4x that) against what the chip's 1024 SIMDs
issue in the launch's duration.  bench.py reads the file (matched by source hash) and divides by the launch time it measures live."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

tag = sys.argv[1]
S = int(sys.argv[2]) if len(sys.argv) > 2 else 3
path = os.path.join(ROOT, "gpurun_out", "summ_" + tag, f"{tag}_sq_mix_S{S}.txt")
cnt, runs = {}, []
for line in open(path):
    if line.startswith("{"):
        runs.append(json.loads(line))
        continue
    m = re.match(r"(\S+) (\S+) (\S+) dispatches (\d+)", line)
    if m and m.group(1) != "gibbs_kernel":   # (gibbs_kernel here = the set-up dispatch only)
        cnt[m.group(2)] = cnt.get(m.group(2), 0.0) + float(m.group(3))
CYC = {"SQ_INSTS_VALU_ADD_F64": 4, "SQ_INSTS_VALU_MUL_F64": 4, "SQ_INSTS_VALU_FMA_F64": 4, "SQ_INSTS_VALU_TRANS_F64": 16, "SQ_INSTS_VALU_INT64": 4,
       "SQ_INSTS_VALU_ADD_F32": 2, "SQ_INSTS_VALU_MUL_F32": 2, "SQ_INSTS_VALU_FMA_F32": 2, "SQ_INSTS_VALU_TRANS_F32": 8, "SQ_INSTS_VALU_INT32": 2, "SQ_INSTS_VALU_CVT": 2}
typed = sum(cnt.get(k, 0.0) for k in CYC)
other = cnt["SQ_INSTS_VALU"] - typed          # moves, selects, compares, bit operations, lane operations: 2 cycles
issue = sum(cnt.get(k, 0.0) * c for k, c in CYC.items()) + other * 2
clusters = runs[0]["clusters"]
out = {"source_hash": bench.source_hash(), "source_hash_gibbs": bench.source_hash("gibbs"), "source_hash_kmc": bench.source_hash("kmc"), "command": f"tools/sq_counters.sh {tag} + {S} {runs[0]['groups']}", "S": S, "groups": runs[0]["groups"], "clusters": clusters,
       "launch_ms_under_pmc": [r["ms"][0] for r in runs],
       "gibbs": {"valu_insts_per_schedule": cnt["SQ_INSTS_VALU"], "valu_issue_cycles_per_schedule": issue, "valu_insts_per_cluster_sweep": cnt["SQ_INSTS_VALU"] / (clusters * 7000.0),
                 "valu_by_type": {k.replace("SQ_INSTS_VALU_", "").lower(): cnt.get(k, 0.0) for k in CYC} | {"other": other},
                 "mfma_busy_cycles": cnt.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), "salu_insts": cnt.get("SQ_INSTS_SALU"), "lds_insts": cnt.get("SQ_INSTS_LDS"),
                 "vmem_rd_insts": cnt.get("SQ_INSTS_VMEM_RD"), "vmem_wr_insts": cnt.get("SQ_INSTS_VMEM_WR"),
                 "wave_quad_cycles": cnt.get("SQ_WAVE_CYCLES"), "active_quad_cycles": cnt.get("SQ_ACTIVE_INST_ANY"), "wait_any_quad_cycles": cnt.get("SQ_WAIT_ANY"),
                 "wait_inst_any_quad_cycles": cnt.get("SQ_WAIT_INST_ANY"), "waves": cnt.get("SQ_WAVES"),
                 "thread_quad_cycles_valu": cnt.get("SQ_THREAD_CYCLES_VALU"), "active_valu_quad_cycles": cnt.get("SQ_ACTIVE_INST_VALU"),
                 "icache_req": cnt.get("SQC_ICACHE_REQ"), "icache_misses": cnt.get("SQC_ICACHE_MISSES")}}
dst = os.path.join(ROOT, "gpurun_out", "summ_" + tag, f"{tag}_issue.json")   # (copy it to profiles/: gpurun_out is scratch)
json.dump(out, open(dst, "w"), indent=1)
print(dst, json.dumps(out["gibbs"])[:400])
