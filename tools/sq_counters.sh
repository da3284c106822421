#!/bin/bash
# usage: tools/sq_counters.sh <tag> <class A|B|C|D|+> [S] [groups]   (on the GPU box, from the repo root)
# SQ counters of the Gibbs sampling launches (gibbs_kernel / gibbs_hot_kernel / gibbs_single_kernel / gibbs_simple_kernel) on one shape class of the bench mixture: five --pmc passes
# (8 SQ slots each) + one each for FETCH_SIZE and WRITE_SIZE (KiB; TCC slots), every pass ONE schedule (BT_PERF_RUNS=1), counters summed over the dispatches of the sampling kernels and reported with the dispatch count,
# so that per-schedule figures follow without guessing -> gpurun_out/summ_<tag>/<tag>_sq_<class>_S<S>.txt
# (round 3's script summed two schedules + the set-up dispatch and the analysis divided by one: every per-sweep figure of that round is 2x too high.)
tag=$1; cls=$2; S=${3:-3}; G=${4:-600000}
export TMPDIR=/tmp BT_PERF_RUNS=1
out=$PWD/gpurun_out
mkdir -p $out/summ_$tag $out/prof_$tag
dst=$out/summ_$tag/${tag}_sq_${cls/+/mix}_S$S.txt
: > $dst
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_ACTIVE_INST_SCA" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_BUSY_CU_CYCLES" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  d=$out/prof_$tag/sq_$cls
  rm -rf $d
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $d -- python tools/perf_classes.py $S $G $cls > $d.log 2> $d.err
  python - "$d" >> $dst <<'PY'
import csv, glob, sys
agg, disp = {}, {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not any(n in k for n in ("gibbs_kernel", "gibbs_simple_kernel", "gibbs_hot_kernel", "gibbs_single_kernel")): continue
        if int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])) == 0: continue
        name = next((n for n in ("gibbs_simple_kernel", "gibbs_single_kernel", "gibbs_hot_kernel") if n in k), "gibbs_kernel")
        agg[(name, r["Counter_Name"])] = agg.get((name, r["Counter_Name"]), 0.0) + float(r["Counter_Value"])
        disp.setdefault((name, r["Counter_Name"]), set()).add(r["Dispatch_Id"])
for (k, c), v in sorted(agg.items()): print(k, c, "%.5g" % v, "dispatches", len(disp[(k, c)]))
PY
  grep -h '"class"' $d.log | cut -c1-240 >> $dst
  tail -3 $d.err | cut -c1-200 >> $dst.err
  rm -rf $d
done
cat $dst
