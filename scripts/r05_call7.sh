#!/bin/bash
# round 5, call 7: reads per wave of the lane-list chaining kernel
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
mkdir -p gpurun_out
export BENCH_INDEX_CACHE=/tmp/bmh_cache
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "job_builder or smoke" > gpurun_out/c7_pytest.log 2>&1 || { tail -40 gpurun_out/c7_pytest.log; exit 1; }
tail -2 gpurun_out/c7_pytest.log
BMH_CHAIN_LIST_LANES=16 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "job_builder" > gpurun_out/c7_pytest16.log 2>&1 || { tail -40 gpurun_out/c7_pytest16.log; exit 1; }
tail -2 gpurun_out/c7_pytest16.log
AB_ERR=gpurun_out/c7_bench_err.log timeout -k 10 1000 bash scripts/env_ab.sh 2 - BMH_CHAIN_LIST_LANES=32 BMH_CHAIN_LIST_LANES=16 BMH_CHAIN_LIST_LANES=8 > gpurun_out/c7_ab.log 2>&1
cat gpurun_out/c7_ab.log
for v in 64 32 16 8; do echo "== CHAIN_LIST_LANES=$v (classes one after the other)"; BMH_CHAIN_LIST_LANES=$v BMH_CHAIN_SERIAL=1 BMH_CHAIN_STATS=1 python bench.py --steps 2 --warmup 1 --no-pcie --cpu-sample 0 --no-next-rows 2>&1 >/dev/null | grep -i "class\|lane" | tail -14; done > gpurun_out/c7_chain_stats.log 2>&1
tail -60 gpurun_out/c7_chain_stats.log
