#!/bin/bash
# round 5, call 8: same-box A/B of the extension kernels: round 4's (lib_r04ext) against round 5's (base) and its parameter variants, 150 bp and 300 bp
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
mkdir -p gpurun_out
export BENCH_INDEX_CACHE=/tmp/bmh_cache
( while true; do sleep 60; echo "[$(date +%T)] a/b running"; done ) &
HB=$!
echo "== 150 bp" > gpurun_out/c8_ab.log
bash scripts/ab.sh 3 base r04ext c16 c64 age1 age8 >> gpurun_out/c8_ab.log 2>&1
echo "== 300 bp" >> gpurun_out/c8_ab.log
AB_ARGS="--no-pcie --cpu-sample 0 --no-next-rows --steps 6 --warmup 2 --read-len 300" bash scripts/ab.sh 2 base r04ext c16 age8 >> gpurun_out/c8_ab.log 2>&1
kill $HB
cat gpurun_out/c8_ab.log
