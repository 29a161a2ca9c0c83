#!/bin/bash
# round 6: the four-reads-per-wave chaining form (CHAIN_SUB) against the round-5 forms, interleaved on one box; class done-times first
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
mkdir -p gpurun_out
export BENCH_INDEX_CACHE=/tmp/bmh_cache
( while true; do sleep 60; echo "[$(date +%T)] running"; done ) &
HB=$!
export AB_ARGS="--no-pcie --cpu-sample 0 --no-next-rows --steps 20 --warmup 4 --distinct-batches 4"
O=gpurun_out/${1:-r06_chain_sub}.txt
shift
echo "# $0 $*" > $O
for cfg in "" "BMH_CHAIN_SUB=0"; do
  ( [ -n "$cfg" ] && export $cfg; echo "== $cfg concurrent"; BMH_CHAIN_STATS=1 python bench.py $AB_ARGS --steps 4 --warmup 2 2>&1 >/dev/null | grep "^\[chain\]" | tail -3
    echo "== $cfg serial"; BMH_CHAIN_SERIAL=1 BMH_CHAIN_STATS=1 python bench.py $AB_ARGS --steps 4 --warmup 2 2>&1 >/dev/null | grep "^\[chain\]" | tail -2 ) >> $O
done
AB_ERR=gpurun_out/r06_call2.err bash scripts/env_ab.sh ${ROUNDS:-3} "$@" >> $O
kill $HB
cat $O
