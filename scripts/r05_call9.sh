#!/bin/bash
# round 5, call 9: prefetch only in the four-lane classes, chunk 16: parity, then same-box A/B against round 4's extension kernels at 150 and 300 bp
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
mkdir -p gpurun_out
export BENCH_INDEX_CACHE=/tmp/bmh_cache
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_pins.py -x -q -m gpu -k "extension or smoke or pins or reference_vectors or job_builder" > gpurun_out/c9_pytest.log 2>&1 || { tail -40 gpurun_out/c9_pytest.log; exit 1; }
tail -2 gpurun_out/c9_pytest.log
( while true; do sleep 60; echo "[$(date +%T)] a/b running"; done ) &
HB=$!
echo "== 150 bp" > gpurun_out/c9_ab.log
bash scripts/ab.sh 3 base r04ext >> gpurun_out/c9_ab.log 2>&1
echo "== 300 bp" >> gpurun_out/c9_ab.log
AB_ARGS="--no-pcie --cpu-sample 0 --no-next-rows --steps 6 --warmup 2 --read-len 300" bash scripts/ab.sh 3 base r04ext >> gpurun_out/c9_ab.log 2>&1
echo "== paired" >> gpurun_out/c9_ab.log
AB_ARGS="--no-pcie --cpu-sample 0 --no-next-rows --steps 10 --warmup 2 --paired" bash scripts/ab.sh 2 base r04ext >> gpurun_out/c9_ab.log 2>&1
kill $HB
cat gpurun_out/c9_ab.log
