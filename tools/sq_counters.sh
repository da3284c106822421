#!/bin/bash
# usage: tools/sq_counters.sh <tag> <class A|B|C|D|+> [S] [groups]   (on the GPU box, from the repo root)
# SQ counters of gibbs_kernel / gibbs_hot_kernel + gibbs_simple_kernel on one shape class of the bench mixture, two --pmc passes (8 SQ slots each), summed over the
# dispatches of the sampling launches -> gpurun_out/summ_<tag>/<tag>_sq_<class>.txt
tag=$1; cls=$2; S=${3:-3}; G=${4:-600000}
export TMPDIR=/tmp
out=$PWD/gpurun_out
mkdir -p $out/summ_$tag $out/prof_$tag
dst=$out/summ_$tag/${tag}_sq_${cls/+/mix}.txt
: > $dst
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
  d=$out/prof_$tag/sq_$cls
  rm -rf $d
  timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d $d -- python tools/perf_classes.py $S $G $cls > $d.log 2> $d.err
  python - "$d" >> $dst <<'PY'
import csv, glob, sys
agg = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Kernel_Name"] for k in ("gibbs_kernel", "gibbs_simple_kernel", "gibbs_hot_kernel")):
            agg[r["Counter_Name"]] = agg.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k, v in agg.items(): print(k, "%.4g" % v)
PY
  grep -h '"class"' $d.log | cut -c1-200 >> $dst
  rm -rf $d
done
cat $dst
