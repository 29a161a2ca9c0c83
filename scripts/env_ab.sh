#!/bin/bash
# A/B of environment knobs on one box: scripts/env_ab.sh <rounds> <config> <config> ...
# a config is "-" (nothing set) or VAR=value[;VAR=value...]; every round runs every config once (interleaved)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
rounds=$1; shift
export BENCH_INDEX_CACHE=${BENCH_INDEX_CACHE:-/tmp/bmh_cache}
ARGS=${AB_ARGS:---no-pcie --cpu-sample 0 --no-next-rows --steps 10 --warmup 2}
for r in $(seq 1 $rounds); do
  for c in "$@"; do
    ( if [ "$c" != "-" ]; then for kv in ${c//;/ }; do export "$kv"; done; fi
    python bench.py $ARGS 2>>${AB_ERR:-/dev/null} | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); i=d['stage_ms_isolated']; s=d['stage_ms']
print('%-44s value %.2f step %.2f ms verified %s | iso: seed %.2f (bwd %.2f) chain %.2f ext %.2f | piped: seed %.1f chain_light %.1f heavy %.1f ext_a %.1f ext_b %.1f' % ('$c', d['value'], d['ms_per_step'], d.get('verified_identical'), i['total'], i['backward'], i['chain'], i['extend'], s['total'], s['chain_light'], s['chain_heavy_beside'], s.get('extend_a',0), s.get('extend_b',0)))" )
  done
done
