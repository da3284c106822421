#!/bin/bash
# End-to-end stage timings of the executables at BASELINE configs[1] size (on the GPU box, from the repo root):
#   tools/e2e_c2.sh <tag> [genome_len 64000000] [variants 200000] [threads, 0 = 2 x the container's CPU quota] [samples 1] [error k-mers per sample 140000000] [SVs per mille 0]
# (BASELINE configs[2]-shaped: 256000000 800000 0 3 560000000 10; configs[3]-shaped per GPU: 64000000 200000 0 10; BT_E2E_FREE_BYTES = pretend that much free
#  HBM so that the unit is genotyped in several launches)
# generates the synthetic data set (tools/make_c2_dataset.cpp), runs `bayesTyperTools makeBloom`, `bayesTyper cluster` and `bayesTyper genotype` with
# BT_STAGE_TIMES=1 and -p <threads>, and writes the stage table to gpurun_out/summ_<tag>/<tag>_e2e_c2.txt (copy it to profiles/).
set -uo pipefail
root=$PWD
# threads: two per core of the container's CPU quota (bayestyper_amd/hostinfo.py; nproc counts the box's logical CPUs, not what the container may use); 0 = that default
auto_threads=$(python3 -c 'import sys; sys.path.insert(0, sys.argv[1]); from bayestyper_amd import hostinfo as h; print(h.baseline_threads(h.host_facts()))' "$root")
tag=$1; L=${2:-64000000}; NV=${3:-200000}; T=${4:-0}; NS=${5:-1}; NE=${6:-140000000}; SV=${7:-0}
if [ "$T" = 0 ]; then T=$auto_threads; fi
out=$root/gpurun_out/summ_$tag; mkdir -p $out
d=/tmp/c2_$tag; rm -rf $d; mkdir -p $d
dst=$out/${tag}_e2e_c2.txt
exe=$root/bayestyper_amd/bayesTyper; tools_exe=$root/bayestyper_amd/bayesTyperTools
t() { local s=$(date +%s%N); "$@"; local rc=$?; local e=$(date +%s%N); echo "# wall $(( (e - s) / 1000000 )) ms (rc $rc)"; return $rc; }
{
echo "# e2e: genome $L nt, $NV candidate variants (80 % SNV, 10 % ins, 10 % del; $SV per mille long deletions with nested candidates), $NS sample(s), k=55, -p $T; $(date -u)"
g++ -O2 -std=c++17 -fopenmp $root/tools/make_c2_dataset.cpp -o $d/make_c2_dataset || exit 1
echo "## data set"; t $d/make_c2_dataset $d $L $NV $NS $NE $SV 2>&1
ls -l $d | awk '{print "#   " $5, $9}'
echo "## bayesTyperTools makeBloom"; ( cd $d && for s in $(seq 1 $NS); do t $tools_exe makeBloom -k sample$s -p $T 2>&1 | tail -3; done )
export BT_STAGE_TIMES=1
run_cluster() { $exe cluster -v $d/candidates.vcf -s $d/samples.tsv -g $d/genome.fa -o $d/bt -p $T -r 42 > $d/cluster.out 2> $d/cluster.err; }
[ -n "${BT_E2E_FREE_BYTES:-}" ] && export BT_GIBBS_FREE_BYTES=$BT_E2E_FREE_BYTES
run_genotype() { $exe genotype -v $d/bt_unit_1/variant_clusters.bin -c $d/bt_cluster_data -s $d/samples.tsv -g $d/genome.fa -o $d/bt -p $T -r 42 > $d/genotype.out 2> $d/genotype.err; }
echo "## bayesTyper cluster"; t run_cluster; tail -30 $d/cluster.err; grep -E "Parsed unit|kmers" $d/cluster.out | head -8
echo "## bayesTyper genotype"; t run_genotype; tail -${BT_E2E_TAIL:-40} $d/genotype.err; grep -E "Out of|genotyped|skipped|Estimated negative" $d/genotype.out | head -8
ls -l $d/bt.vcf 2>/dev/null | awk '{print "# output VCF bytes: " $5}'
if [ -n "${BT_E2E_SECOND:-}" ]; then   # a second `genotype` run with extra environment (e.g. "BT_GIBBS_DEBUG=1 BT_NOISE_CHAIN_PROF=1"): the lines BT_E2E_SECOND_GREP selects
  echo "## bayesTyper genotype again with $BT_E2E_SECOND"
  env $BT_E2E_SECOND $exe genotype -v $d/bt_unit_1/variant_clusters.bin -c $d/bt_cluster_data -s $d/samples.tsv -g $d/genome.fa -o $d/bt3 -p $T -r 42 > $d/genotype3.out 2> $d/genotype3.err
  grep -E "${BT_E2E_SECOND_GREP:-noise_chain|estimate noise|noise iterations}" $d/genotype3.err | cut -c1-1500 | head -${BT_E2E_SECOND_LINES:-12}
fi
if [ -n "${BT_E2E_TRACE:-}" ]; then   # kernel trace of a second `genotype` run: totals per kernel + a window of the noise driver's loop
  echo "## bayesTyper genotype under rocprofv3 --kernel-trace --stats"
  export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $d/trace -- $exe genotype -v $d/bt_unit_1/variant_clusters.bin -c $d/bt_cluster_data -s $d/samples.tsv -g $d/genome.fa -o $d/bt2 -p $T -r 42 > $d/genotype2.out 2> $d/genotype2.err
  python $root/tools/kstats.py $d/trace | sort -k9 -n -r | head -16
  python $root/tools/dispatch_timeline.py $d/trace > $d/timeline.txt
  n=$(grep -n gibbs_noise_kernel $d/timeline.txt | sed -n 2000p | cut -d: -f1)
  [ -n "$n" ] && sed -n "$((n - 30)),$((n + 10))p" $d/timeline.txt
fi
grep -vc '^#' $d/bt.vcf 2>/dev/null | awk '{print "# output VCF records: " $1}'
} > $dst 2>&1
cat $dst
rm -rf $d
