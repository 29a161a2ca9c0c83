cd $GRAFT_REPO_ROOT
export BENCH_INDEX_CACHE=/tmp/bmh_cache
ARGS="--no-pcie --cpu-sample 0 --no-next-rows --steps 10 --warmup 2"
for r in 1 2; do
for n in 0 192 128 96 64; do
  BMH_CHAIN_CUS=$n python bench.py $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']
print('cus=%-4s value %.2f step %.2f ms | piped: seed %.1f chain_light %.1f heavy %.1f ext_a %.1f ext_b %.1f cem %.1f' % ('$n', d['value'], d['ms_per_step'], s['total'], s['chain_light'], s['chain_heavy_beside'], s.get('extend_a',0), s.get('extend_b',0), s.get('chain_extend_merge',0)))"
done
done
