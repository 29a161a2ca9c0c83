#!/bin/bash
# round 5, call 3: forward kernel pulls its reads; seeding occupancy curve; co-run residency with the segmented trace; bench
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_pins.py -x -q -m gpu -k "seed or smoke or pins or reference_vectors or hot_path" > gpurun_out/c3_pytest.log 2>&1 || { tail -40 gpurun_out/c3_pytest.log; exit 1; }
tail -3 gpurun_out/c3_pytest.log
export BENCH_INDEX_CACHE=/tmp/bmh_cache
CORUN_TRACE=1 CORUN_PRIOS=0 CORUN_CONFIGS="-;EXT_PERSIST=2;EXT_PERSIST=1;SEED_LDS_PAD=80000;SEED_LDS_PAD=53000;SEED_LDS_PAD=40000;SEED_LDS_PAD=32000;SEED_LDS_PAD=26000" timeout -k 10 900 python scripts/corun_probe.py > gpurun_out/c3_corun.log 2>&1 || { tail -30 gpurun_out/c3_corun.log; exit 1; }
grep -v amdgpu.ids gpurun_out/c3_corun.log
AB_ERR=gpurun_out/c3_bench_err.log AB_ARGS="--no-pcie --cpu-sample 20000 --no-next-rows --steps 20 --warmup 5" timeout -k 10 900 bash scripts/env_ab.sh 2 - > gpurun_out/c3_ab.log 2>&1
cat gpurun_out/c3_ab.log; grep -v amdgpu.ids gpurun_out/c3_bench_err.log | tail -5
