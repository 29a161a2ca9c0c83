#!/bin/bash
# round 6: library variants (build/variants/lib_<tag>.so) -- class done-times one after the other, then interleaved bench runs
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
mkdir -p gpurun_out
export BENCH_INDEX_CACHE=/tmp/bmh_cache
( while true; do sleep 60; echo "[$(date +%T)] running"; done ) &
HB=$!
export AB_ARGS="--no-pcie --cpu-sample 0 --no-next-rows --steps 20 --warmup 4 --distinct-batches 4"
O=gpurun_out/$1.txt; shift
echo "# $0 $*" > $O
if [ -n "$PYTEST_K" ]; then timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$PYTEST_K" 2>&1 | tail -3 >> $O || { kill $HB; cat $O; exit 1; }; fi
for t in "$@"; do
  ( if [ "$t" = base ]; then unset BMH_LIB; else export BMH_LIB=$R/build/variants/lib_$t.so; fi
    echo "== $t serial"; BMH_CHAIN_SERIAL=1 BMH_CHAIN_STATS=1 python bench.py $AB_ARGS --steps 2 --warmup 2 2>&1 >/dev/null | grep "^\[chain\]" | tail -1
    echo "== $t concurrent"; BMH_CHAIN_STATS=1 python bench.py $AB_ARGS --steps 2 --warmup 2 2>&1 >/dev/null | grep "^\[chain\]" | tail -2 ) >> $O
done
bash scripts/ab.sh ${ROUNDS:-2} "$@" >> $O
kill $HB
cat $O
