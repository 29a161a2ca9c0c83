#!/bin/bash
# round 5, call 5: the seeding stage confined to a subset of the compute units (BMH_SEED_CUS) beside the other batch's extension
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
mkdir -p gpurun_out
export BENCH_INDEX_CACHE=/tmp/bmh_cache
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "seeding_matches or smoke" > gpurun_out/c5_pytest.log 2>&1 || { tail -40 gpurun_out/c5_pytest.log; exit 1; }
tail -2 gpurun_out/c5_pytest.log
for q in 0 128 96 64; do
  echo "== BMH_SEED_CUS=$q"
  BMH_SEED_CUS=$q CORUN_TRACE=1 CORUN_PRIOS=0 CORUN_CONFIGS="-" timeout -k 10 600 python scripts/corun_probe.py 2>&1 | grep -v amdgpu.ids
done > gpurun_out/c5_corun.log 2>&1
cat gpurun_out/c5_corun.log | cut -c1-250
AB_ERR=gpurun_out/c5_bench_err.log timeout -k 10 900 bash scripts/env_ab.sh 2 - BMH_SEED_CUS=128 BMH_SEED_CUS=96 BMH_SEED_CUS=160 > gpurun_out/c5_ab.log 2>&1
cat gpurun_out/c5_ab.log
