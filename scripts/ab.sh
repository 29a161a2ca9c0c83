#!/bin/bash
# A/B of library builds on one box: scripts/ab.sh <rounds> <tag1> <tag2> ...   (tag "base" = the in-tree library; others = build/variants/lib_<tag>.so)
# Every round runs every variant once (interleaved, so that drifts of the box hit all of them alike); prints value, step and isolated stage times.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
rounds=$1; shift
export BENCH_INDEX_CACHE=${BENCH_INDEX_CACHE:-/tmp/bmh_cache}
ARGS=${AB_ARGS:---no-pcie --cpu-sample 0 --no-next-rows --steps 10 --warmup 2}
for r in $(seq 1 $rounds); do
  for t in "$@"; do
    if [ "$t" = base ]; then unset BMH_LIB; else export BMH_LIB=$R/build/variants/lib_$t.so; fi
    python bench.py $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); i=d['stage_ms_isolated']; s=d['stage_ms']
print('%-10s value %.2f step %.2f ms | iso: seed %.2f (bwd %.2f) chain %.2f ext %.2f | piped: seed %.1f chain_light %.1f heavy %.1f ext_a %.1f ext_b %.1f' % ('$t', d['value'], d['ms_per_step'], i['total'], i['backward'], i['chain'], i['extend'], s['total'], s['chain_light'], s['chain_heavy_beside'], s.get('extend_a',0), s.get('extend_b',0)))"
  done
done
