#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
mkdir -p gpurun_out
export BENCH_INDEX_CACHE=/tmp/bmh_cache
( while true; do sleep 60; echo "[$(date +%T)] a/b running"; done ) &
HB=$!
echo "== 300 bp" > gpurun_out/c15_ab.log
AB_ARGS="--no-pcie --cpu-sample 0 --no-next-rows --steps 6 --warmup 2 --read-len 300" bash scripts/ab.sh 3 base hyb r04ext >> gpurun_out/c15_ab.log 2>&1
kill $HB
cat gpurun_out/c15_ab.log
