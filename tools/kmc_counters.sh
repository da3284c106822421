#!/bin/bash
# usage: tools/kmc_counters.sh <tag>   (on the GPU box, from the repo root): SQ counters + FETCH/WRITE_SIZE of the three kernels of the KMC scan on
# tools/perf_kmc.py's stream (4 x 10^8 records: 6 chunks x 6 scans), summed over the dispatches and reported with the dispatch count
# -> gpurun_out/summ_<tag>/<tag>_sq_kmc.txt
tag=${1:-r04}
export TMPDIR=/tmp
out=$PWD/gpurun_out/summ_$tag; mkdir -p $out
dst=$out/${tag}_sq_kmc.txt
: > $dst
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  d=/tmp/kmcpmc; rm -rf $d
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -- python tools/perf_kmc.py 400000000 50000000 "partitioned:BT_KMC_ROUTED=1" > $d.log 2> $d.err
  python - "$d" >> $dst <<'PY'
import csv, glob, sys
agg, disp, dur = {}, {}, {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        name = next((n for n in ("kmc_partition_kernel", "kmc_probe_bucket_kernel", "kmc_apply_kernel") if n in k), None)
        if not name: continue
        agg[(name, r["Counter_Name"])] = agg.get((name, r["Counter_Name"]), 0.0) + float(r["Counter_Value"])
        disp.setdefault((name, r["Counter_Name"]), set()).add(r["Dispatch_Id"])
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = next((n for n in ("kmc_partition_kernel", "kmc_probe_bucket_kernel", "kmc_apply_kernel") if n in r["Kernel_Name"]), None)
        if name: dur[name] = dur.get(name, 0.0) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for (k, c), v in sorted(agg.items()): print(k, c, "%.5g" % v, "dispatches", len(disp[(k, c)]))
for k, v in sorted(dur.items()): print(k, "total_ms_under_pmc", "%.2f" % v)
PY
  grep -h "scans" $d.log | cut -c1-200 >> $dst
  rm -rf $d
done
cat $dst
