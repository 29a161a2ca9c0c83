#!/bin/bash
# the round's profile (bench lines of the three workloads, kernel stats, PMC passes), then the driver's command once more
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
( while true; do sleep 60; echo "[$(date +%T)] profiling"; done ) &
HB=$!
bash scripts/profile_round.sh ${TAG:-r06} variants > gpurun_out/${TAG:-r06}_profile.log 2>&1; rc=$?
kill $HB
tail -5 gpurun_out/${TAG:-r06}_profile.log
[ $rc -eq 0 ] || exit $rc
export BENCH_INDEX_CACHE=/tmp/bmh_cache
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG:-r06}_bench_driver_cmd.json 2> gpurun_out/${TAG:-r06}_bench_driver_cmd.err || { tail -5 gpurun_out/${TAG:-r06}_bench_driver_cmd.err; exit 1; }
head -c 1500 gpurun_out/${TAG:-r06}_bench_driver_cmd.json
