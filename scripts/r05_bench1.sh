#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
mkdir -p gpurun_out
export BENCH_INDEX_CACHE=/tmp/bmh_cache
python bench.py --gpus 1 --steps 20 --warmup 5 --no-next-rows > gpurun_out/b1.json 2> gpurun_out/b1.err || { tail -20 gpurun_out/b1.err; exit 1; }
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/b1.json') if l.startswith('{')][-1])
for k in ('value','ms_per_step','clock_mhz','incl_pcie_value','stage_ms_isolated','stage_ms','verified_identical','clock'):
    print(k, d.get(k))
print('roofline frac', d['roofline']['frac'], d['roofline'].get('avg_ms'))
print('cpu', d['cpu_baseline']['value'])
PY
