#!/bin/bash
# round 5, call 4: is the seeding of one batch held back behind the other batch's extension by the hardware queues?  (co-run under GPU_MAX_HW_QUEUES)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
mkdir -p gpurun_out
export BENCH_INDEX_CACHE=/tmp/bmh_cache
for q in 4 8 16; do
  echo "== GPU_MAX_HW_QUEUES=$q"
  GPU_MAX_HW_QUEUES=$q CORUN_TRACE=1 CORUN_PRIOS=0 CORUN_CONFIGS="-;EXT_PERSIST=2" timeout -k 10 600 python scripts/corun_probe.py 2>&1 | grep -v amdgpu.ids
done > gpurun_out/c4_corun.log 2>&1
cat gpurun_out/c4_corun.log
AB_ERR=gpurun_out/c4_bench_err.log timeout -k 10 900 bash scripts/env_ab.sh 2 - GPU_MAX_HW_QUEUES=8 GPU_MAX_HW_QUEUES=16 "GPU_MAX_HW_QUEUES=16;BMH_EXT_PERSIST=2" > gpurun_out/c4_ab.log 2>&1
cat gpurun_out/c4_ab.log
