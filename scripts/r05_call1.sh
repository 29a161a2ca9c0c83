#!/bin/bash
# round 5, call 1: the persistent packed kernel -- parity in both launch forms, co-run sweep (with wave residency), bench A/B
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "extension or smoke" > gpurun_out/c1_pytest.log 2>&1 || { tail -30 gpurun_out/c1_pytest.log; exit 1; }
tail -3 gpurun_out/c1_pytest.log
export BENCH_INDEX_CACHE=/tmp/bmh_cache
CORUN_TRACE=1 CORUN_PRIOS=0 CORUN_CONFIGS="-;EXT_PERSIST=3;EXT_PERSIST=2;EXT_PERSIST=1;EXT_PERSIST=2,SEED_SETPRIO=3;SEED_SETPRIO=3;EXT_PERSIST=2,SEED_LDS_PAD=20000" timeout -k 10 600 python scripts/corun_probe.py > gpurun_out/c1_corun.log 2>&1 || { tail -30 gpurun_out/c1_corun.log; exit 1; }
cat gpurun_out/c1_corun.log
timeout -k 10 900 bash scripts/env_ab.sh 2 - BMH_EXT_PERSIST=3 BMH_EXT_PERSIST=2 "BMH_EXT_PERSIST=2;BMH_SEED_SETPRIO=3" BMH_SEED_SETPRIO=3 > gpurun_out/c1_ab.log 2>&1
cat gpurun_out/c1_ab.log
