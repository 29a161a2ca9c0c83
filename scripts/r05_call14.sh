#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
mkdir -p gpurun_out
export BENCH_INDEX_CACHE=/tmp/bmh_cache
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_pins.py -x -q -m gpu -k "extension or smoke or pins or reference_vectors or job_builder" > gpurun_out/c17_pytest.log 2>&1 || { tail -40 gpurun_out/c17_pytest.log; exit 1; }
tail -2 gpurun_out/c17_pytest.log
( while true; do sleep 60; echo "[$(date +%T)] a/b running"; done ) &
HB=$!
echo "== 300 bp" > gpurun_out/c17_ab.log
AB_ARGS="--no-pcie --cpu-sample 0 --no-next-rows --steps 6 --warmup 2 --read-len 300" bash scripts/ab.sh 3 base r04ext >> gpurun_out/c17_ab.log 2>&1
echo "== 150 bp" >> gpurun_out/c17_ab.log
bash scripts/ab.sh 3 base r04ext >> gpurun_out/c17_ab.log 2>&1
kill $HB
cat gpurun_out/c17_ab.log
