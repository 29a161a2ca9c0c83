#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
mkdir -p gpurun_out
export BENCH_INDEX_CACHE=/tmp/bmh_cache
BMH_CHAIN_B_SIDE=1 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_hg38_scale.py -x -q -m gpu -k "job_builder or smoke or hot_path or reads_to_sam or aligner_writes" > gpurun_out/c18_pytest.log 2>&1 || { tail -40 gpurun_out/c18_pytest.log; exit 1; }
tail -2 gpurun_out/c18_pytest.log
( while true; do sleep 60; echo "[$(date +%T)] a/b running"; done ) &
HB=$!
AB_ARGS="--no-pcie --cpu-sample 20000 --no-next-rows --steps 20 --warmup 5" AB_ERR=gpurun_out/c18_err.log bash scripts/env_ab.sh 3 - BMH_CHAIN_B_SIDE=1 > gpurun_out/c18_ab.log 2>&1
echo "== 300 bp" >> gpurun_out/c18_ab.log
AB_ARGS="--no-pcie --cpu-sample 0 --no-next-rows --steps 6 --warmup 2 --read-len 300" bash scripts/env_ab.sh 2 - BMH_CHAIN_B_SIDE=1 >> gpurun_out/c18_ab.log 2>&1
kill $HB
cat gpurun_out/c18_ab.log
