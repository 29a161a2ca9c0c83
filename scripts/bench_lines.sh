#!/bin/bash
# the bench lines of a round: bench_lines.sh <tag>  ->  gpurun_out/<tag>_bench{_driver_cmd,,_paired,_300bp}.json  (copy them to profiles/)
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
mkdir -p gpurun_out
export BENCH_INDEX_CACHE=${BENCH_INDEX_CACHE:-/tmp/bmh_cache}
( while true; do sleep 60; echo "[$(date +%T)] bench lines"; done ) &
HB=$!
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_cmd.json 2> gpurun_out/${TAG}_bench_driver_cmd.err || { kill $HB; tail -5 gpurun_out/${TAG}_bench_driver_cmd.err; exit 1; }
python3 bench.py > gpurun_out/${TAG}_bench.json 2>/dev/null || { kill $HB; exit 1; }
python3 bench.py --paired --cpu-sample 20000 > gpurun_out/${TAG}_bench_paired.json 2>/dev/null || { kill $HB; exit 1; }
python3 bench.py --read-len 300 --cpu-sample 20000 > gpurun_out/${TAG}_bench_300bp.json 2>/dev/null || { kill $HB; exit 1; }
kill $HB
head -c 600 gpurun_out/${TAG}_bench_driver_cmd.json
