#!/bin/bash
# usage: pmc_shape.sh <shape> <n> <S>   -> gpurun_out/pmc_<shape>.txt
export TMPDIR=/tmp
out=$PWD/gpurun_out
mkdir -p $out
: > $out/pmc_$1.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_FLAT" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT" "SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA"; do
  d=$out/pmc_tmp
  rm -rf $d
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -- python tools/perf_one.py $1 $2 $3 > /dev/null 2> $out/pmc_err.txt || { echo "set [$set] failed: $(tail -2 $out/pmc_err.txt)" >> $out/pmc_$1.txt; continue; }
  python - "$d" >> $out/pmc_$1.txt <<'PY'
import sys,glob,csv,collections
rows=collections.defaultdict(float)
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "gibbs_kernel" in r["Kernel_Name"] and int(r["Grid_Size"])>100000:
            rows[r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in rows.items(): print(k, "%.4g"%v)
PY
done
rm -rf $out/pmc_tmp
cat $out/pmc_$1.txt
