#!/bin/bash
# End-to-end stage timings of the executables at BASELINE configs[1] size (on the GPU box, from the repo root):
#   tools/e2e_c2.sh <tag> [genome_len 64000000] [variants 200000] [threads = nproc]
# generates the synthetic data set (tools/make_c2_dataset.cpp), runs `bayesTyperTools makeBloom`, `bayesTyper cluster` and `bayesTyper genotype` with
# BT_STAGE_TIMES=1 and -p <threads>, and writes the stage table to gpurun_out/summ_<tag>/<tag>_e2e_c2.txt (copy it to profiles/).
set -uo pipefail
tag=$1; L=${2:-64000000}; NV=${3:-200000}; T=${4:-$(nproc)}
root=$PWD
out=$root/gpurun_out/summ_$tag; mkdir -p $out
d=/tmp/c2_$tag; rm -rf $d; mkdir -p $d
dst=$out/${tag}_e2e_c2.txt
exe=$root/bayestyper_amd/bayesTyper; tools_exe=$root/bayestyper_amd/bayesTyperTools
t() { local s=$(date +%s%N); "$@"; local rc=$?; local e=$(date +%s%N); echo "# wall $(( (e - s) / 1000000 )) ms (rc $rc)"; return $rc; }
{
echo "# e2e C2: genome $L nt, $NV candidate variants (80 % SNV, 10 % ins, 10 % del), 1 sample, k=55, -p $T; $(date -u)"
g++ -O2 -std=c++17 -fopenmp $root/tools/make_c2_dataset.cpp -o $d/make_c2_dataset || exit 1
echo "## data set"; t $d/make_c2_dataset $d $L $NV 1 2>&1
ls -l $d | awk '{print "#   " $5, $9}'
echo "## bayesTyperTools makeBloom"; ( cd $d && t $tools_exe makeBloom -k sample1 -p $T 2>&1 | tail -4 )
export BT_STAGE_TIMES=1
run_cluster() { $exe cluster -v $d/candidates.vcf -s $d/samples.tsv -g $d/genome.fa -o $d/bt -p $T -r 42 > $d/cluster.out 2> $d/cluster.err; }
run_genotype() { $exe genotype -v $d/bt_unit_1/variant_clusters.bin -c $d/bt_cluster_data -s $d/samples.tsv -g $d/genome.fa -o $d/bt -p $T -r 42 > $d/genotype.out 2> $d/genotype.err; }
echo "## bayesTyper cluster"; t run_cluster; tail -30 $d/cluster.err; grep -E "Parsed unit|kmers" $d/cluster.out | head -8
echo "## bayesTyper genotype"; t run_genotype; tail -30 $d/genotype.err; grep -E "Out of|genotyped|skipped|Estimated negative" $d/genotype.out | head -8
ls -l $d/bt.vcf 2>/dev/null | awk '{print "# output VCF bytes: " $5}'
grep -vc '^#' $d/bt.vcf 2>/dev/null | awk '{print "# output VCF records: " $1}'
} > $dst 2>&1
cat $dst
rm -rf $d
