#!/bin/bash
# round 5, call 2: the job pipeline of the packed kernels (records, chunk cache, one job staged ahead) -- parity, co-run sweep, bench A/B
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_pins.py -x -q -m gpu -k "extension or smoke or pins or reference_vectors or job_builder or reads_to_sam" > gpurun_out/c2_pytest.log 2>&1 || { tail -40 gpurun_out/c2_pytest.log; exit 1; }
tail -3 gpurun_out/c2_pytest.log
export BENCH_INDEX_CACHE=/tmp/bmh_cache
CORUN_PRIOS=0 CORUN_CONFIGS="-;EXT_PERSIST=3;EXT_PERSIST=2;EXT_PERSIST=2,SEED_LDS_PAD=20000" timeout -k 10 600 python scripts/corun_probe.py > gpurun_out/c2_corun.log 2>&1 || { tail -30 gpurun_out/c2_corun.log; exit 1; }
cat gpurun_out/c2_corun.log
df -h /tmp /dev/shm | tail -3
AB_ERR=gpurun_out/c2_bench_err.log timeout -k 10 900 bash scripts/env_ab.sh 2 - BMH_EXT_PERSIST=3 BMH_EXT_PERSIST=2 > gpurun_out/c2_ab.log 2>&1
cat gpurun_out/c2_ab.log; tail -5 gpurun_out/c2_bench_err.log
