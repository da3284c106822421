#!/bin/bash
# usage: tools/kmc_profile.sh <tag>   (on the GPU box, from the repo root): the KMC scan alone (tools/perf_kmc.py, bench.py's stream) — scan times at the
# WGS sub-filter size and at a ten-sample path filter's, the per-kernel times of the former under rocprofv3 --kernel-trace --stats -> gpurun_out/summ_<tag>/<tag>_kmc_scan.txt
tag=${1:-r04}
export TMPDIR=/tmp
out=$PWD/gpurun_out/summ_$tag; mkdir -p $out
dst=$out/${tag}_kmc_scan.txt
{
echo "# python tools/perf_kmc.py 1000000000 50000000 (10^9 records of 13 B, 5x10^7 path k-mers: 1.9 KB sub-filters; two rounds of three scans into an emptied table)"
python tools/perf_kmc.py 1000000000 50000000 "partitioned:BT_KMC_ROUTED=1;release_publish:BT_KMC_ROUTED=1,BT_TABLE_RELEASE_PUBLISH=1" 2>&1 | grep -E "sub-filter|scans"
echo "# python tools/perf_kmc.py 400000000 1000000000 (4x10^8 records, 10^9 path k-mers: 36 KB sub-filters)"
python tools/perf_kmc.py 400000000 1000000000 "partitioned:BT_KMC_ROUTED=1" 2>&1 | grep -E "sub-filter|scans"
echo "# rocprofv3 --kernel-trace --stats -- python tools/perf_kmc.py 1000000000 50000000 partitioned:BT_KMC_ROUTED=1   (90 chunks of 2^26 records)"
rm -rf /tmp/kmcprof; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kmcprof -- python tools/perf_kmc.py 1000000000 50000000 "partitioned:BT_KMC_ROUTED=1" > /dev/null 2>&1
python tools/kstats.py /tmp/kmcprof | grep -E "kmc_|fillBuffer"
rm -rf /tmp/kmcprof
} > $dst 2>&1
cat $dst
