#!/bin/bash
# rounds 5-6: the extension fuzz (array jobs: packed and 32-bit kernels) and the WHOLE timed batch of each workload against the oracle (descriptor jobs: the job pipeline's fetch-ahead path)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
mkdir -p gpurun_out
export BENCH_INDEX_CACHE=/tmp/bmh_cache
( while true; do sleep 60; echo "[$(date +%T)] verifying"; done ) &
HB=$!
timeout -k 10 900 python scripts/fuzz_extend.py 200000 26 5 2>&1 | grep -v amdgpu > gpurun_out/r06_fuzz_extend.txt; tail -3 gpurun_out/r06_fuzz_extend.txt
: > gpurun_out/r06_full_batch_verified.jsonl
for extra in "" "--paired" "--read-len 300"; do
  python bench.py --steps 2 --warmup 1 --no-pcie --no-next-rows --cpu-sample 0 --verify-sample 1000000 $extra 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print(json.dumps({'workload_key': d['config']['workload_key'], 'verified': d['verified'], 'value': d['value']}))" >> gpurun_out/r06_full_batch_verified.jsonl
done
kill $HB
cat gpurun_out/r06_full_batch_verified.jsonl
