#!/bin/bash
# usage: tools/profile_round.sh <tag>   (on the GPU box, from the repo root): rocprofv3 kernel trace + stats of the default bench command, then
# one PMC pass each for FETCH_SIZE and WRITE_SIZE; summaries -> gpurun_out/summ_<tag>/ (copy what should be judged into profiles/)
tag=${1:-r03}
export TMPDIR=/tmp
out=$PWD/gpurun_out
mkdir -p $out/prof_$tag $out/summ_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$tag/trace -- python bench.py --steps 2 --warmup 1 > $out/prof_$tag/bench_under_rocprof.json 2> $out/prof_$tag/trace.err
tail -1 $out/prof_$tag/bench_under_rocprof.json > $out/summ_$tag/${tag}_bench_under_rocprof.json
python tools/prof_summarize.py $out/prof_$tag/trace ${tag}_bench $out/summ_$tag
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/prof_$tag/pmc_$c -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-paths --no-pcie --no-extra > /dev/null 2> $out/prof_$tag/pmc_$c.err
  python tools/prof_summarize.py $out/prof_$tag/pmc_$c ${tag}_$c $out/summ_$tag
done
python tools/pmc_traffic.py $out/summ_$tag $tag 3
rm -rf $out/prof_$tag/trace $out/prof_$tag/pmc_*   # raw traces are large; summaries stay
if [ -n "$BT_PROFILE_QUICK" ]; then ls -la $out/summ_$tag; exit 0; fi   # the passes below do not depend on the Gibbs sources
: > $out/summ_$tag/${tag}_sq_graph_stages.txt
ls -la $out/summ_$tag
# SQ counters of the two kernels that run one lane per cluster / group (find_paths_kernel, mg_order_kernel): tools/graph_stages.py
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM FETCH_SIZE WRITE_SIZE"; do
  d=$out/prof_$tag/graph
  rm -rf $d
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -- python tools/graph_stages.py > $d.log 2> $d.err
  python - "$d" >> $out/summ_$tag/${tag}_sq_graph_stages.txt <<'PY'
import csv, glob, sys
agg, dur = {}, {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for k in ("find_paths_kernel", "mg_order_kernel"):
            if k in r["Kernel_Name"]:
                agg[(k, r["Counter_Name"])] = agg.get((k, r["Counter_Name"]), 0.0) + float(r["Counter_Value"])
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for k in ("find_paths_kernel", "mg_order_kernel"):
            if k in r["Kernel_Name"]:
                dur[k] = dur.get(k, 0.0) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for (k, c), v in sorted(agg.items()): print(k, c, "%.4g" % v)
for k, v in sorted(dur.items()): print(k, "duration_ms", "%.3f" % v)
PY
  grep -h clusters $d.log | cut -c1-300 >> $out/summ_$tag/${tag}_sq_graph_stages.txt
  rm -rf $d
done
# single-schedule SQ + TCC counters per shape class and for the whole mixture at S = 3 (the bench batch) and per class at S = 10 (the 100 352-group batch the
# ten-sample record replicates), then the issue-side profile bench.py reads (profiles/<tag>_issue.json)
for c in A B C D +; do bash tools/sq_counters.sh $tag $c 3 600000 > /dev/null 2>&1; done
python tools/issue_profile.py $tag 3
for c in A B C D; do bash tools/sq_counters.sh $tag $c 10 100000 > /dev/null 2>&1; done
ls -la $out/summ_$tag
