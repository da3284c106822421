#!/bin/bash
# usage: tools/mem_counters.sh <tag> <class A|B|C|D|+> [S] [groups]   (on the GPU box, from the repo root; extra environment is inherited: BT_GIBBS_NO_SINGLE_KERNEL ...)
# Memory-path counters of the Gibbs sampling launches on one shape class of the bench mixture — what a CU's wavefronts wait for when a third wavefront per
# SIMD buys nothing (round 6): texture-addresser / L1 busy and stall cycles, L1 -> L2 request latency, address translation (UTCL1), L2 hits and the fabric
# requests behind it, the wavefronts' in-flight instruction levels, the scalar data cache.  One --pmc pass per set, ONE schedule per pass (BT_PERF_RUNS=1).
# -> gpurun_out/summ_<tag>/<tag>_mem_<class>_S<S>.txt
tag=$1; cls=$2; S=${3:-3}; G=${4:-600320}
export TMPDIR=/tmp BT_PERF_RUNS=1
out=$PWD/gpurun_out
mkdir -p $out/summ_$tag $out/prof_$tag
dst=$out/summ_$tag/${tag}_mem_${cls/+/mix}_S$S.txt
: > $dst
for set in "GRBM_GUI_ACTIVE GRBM_TA_BUSY TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum TA_FLAT_WAVEFRONTS_sum" \
           "TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum GRBM_UTCL2_BUSY TCP_TCP_LATENCY_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_ATOMIC_sum" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_LEVEL_WAVES SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL SQC_DCACHE_BUSY_CYCLES SQC_ICACHE_BUSY_CYCLES SQ_INST_CYCLES_SMEM SQ_INSTS_SMEM" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum TD_LOAD_WAVEFRONT_sum TD_STORE_WAVEFRONT_sum SQ_LDS_BANK_CONFLICT"; do
  d=$out/prof_$tag/mem_$cls
  rm -rf $d
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $d -- python tools/perf_classes.py $S $G $cls > $d.log 2> $d.err
  python - "$d" >> $dst <<'PY'
import csv, glob, sys
agg, disp = {}, {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        name = next((n for n in ("gibbs_simple_kernel", "gibbs_single_kernel", "gibbs_hot_kernel", "gibbs_kernel") if n in k), None)
        if name is None or name == "gibbs_kernel": continue   # (gibbs_kernel here = the set-up dispatch)
        agg[(name, r["Counter_Name"])] = agg.get((name, r["Counter_Name"]), 0.0) + float(r["Counter_Value"])
        disp.setdefault((name, r["Counter_Name"]), set()).add(r["Dispatch_Id"])
for (k, c), v in sorted(agg.items()): print(k, c, "%.5g" % v, "dispatches", len(disp[(k, c)]))
PY
  grep -h '"class"' $d.log | cut -c1-240 >> $dst
  tail -2 $d.err | cut -c1-200 >> $dst.err
  rm -rf $d
done
cat $dst
