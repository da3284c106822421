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
ls -la $out/summ_$tag
