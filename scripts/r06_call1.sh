#!/bin/bash
# round 6, first GPU call: the chaining stage's ceiling.  Interleaved bench runs on one box (scripts/env_ab.sh) with the kernels of
# the seed-rich size classes skipped and their stored regions replayed (BENCH_CHAIN_REPLAY = class mask), and with the waves of the
# largest classes at a raised priority; first the classes' done-times of this build.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
mkdir -p gpurun_out
export BENCH_INDEX_CACHE=/tmp/bmh_cache
( while true; do sleep 60; echo "[$(date +%T)] running"; done ) &
HB=$!
export AB_ARGS="--no-pcie --cpu-sample 0 --no-next-rows --steps 20 --warmup 4 --distinct-batches 2"
O=gpurun_out/r06_chain_ablation.txt
echo "# chaining-stage ablation (scripts/r06_call1.sh): steps 20, --distinct-batches 2, two batches in flight" > $O
BMH_CHAIN_STATS=1 python bench.py $AB_ARGS --steps 4 --warmup 2 2>&1 >/dev/null | grep "^\[chain\]" | tail -4 >> $O
AB_ERR=gpurun_out/r06_call1.err bash scripts/env_ab.sh 3 - "BENCH_CHAIN_REPLAY=1023" "BENCH_CHAIN_REPLAY=1" "BENCH_CHAIN_REPLAY=3" "BENCH_CHAIN_REPLAY=1020" "BMH_CHAIN_WAVE_PRIO=480" "BMH_CHAIN_WAVE_PRIO=1022" >> $O
kill $HB
cat $O | cut -c1-400
