#!/bin/bash
# quick kernel-stats profile of the bench under the current environment: scripts/prof_quick.sh <tag> [grep pattern]
R=$GRAFT_REPO_ROOT; T=$1; P=${2:-.}
export BENCH_INDEX_CACHE=${BENCH_INDEX_CACHE:-/tmp/bmh_cache}
cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/pq_$T; mkdir -p $R/gpurun_out/pq_$T
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pq_$T -- python3 $R/bench.py --steps 5 --warmup 1 --no-pcie --cpu-sample 0 --no-next-rows > $R/gpurun_out/pq_$T/bench.json 2> $R/gpurun_out/pq_$T/err.log || exit 1
cd $R
find gpurun_out/pq_$T -name "*kernel_trace.csv" -delete
f=$(find gpurun_out/pq_$T -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/pq_$T.csv
grep -E "$P" gpurun_out/pq_$T.csv | cut -c1-160
