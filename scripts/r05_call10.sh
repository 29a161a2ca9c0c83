#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
export BENCH_INDEX_CACHE=/tmp/bmh_cache
for lib in base r04ext; do
  if [ "$lib" = base ]; then unset BMH_LIB; else export BMH_LIB=$R/build/variants/lib_$lib.so; fi
  echo "== $lib 300 bp"
  BMH_EXT_STATS=1 BMH_EXT_PHASES=1 python bench.py --steps 2 --warmup 1 --no-pcie --cpu-sample 0 --no-next-rows --inflight 1 --passes 1 --read-len 300 2>&1 >/dev/null | grep '^\[ext\]' | tail -6
done
