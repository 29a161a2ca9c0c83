#!/bin/bash
# the whole GPU suite the way the driver runs it, then smoke
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
mkdir -p gpurun_out
( while true; do sleep 60; echo "[$(date +%T)] suite running"; done ) &
HB=$!
timeout -k 10 3000 python -m pytest tests/ -x -q -m gpu --durations=15 > gpurun_out/suite.log 2>&1; rc=$?
kill $HB
tail -40 gpurun_out/suite.log
[ $rc -eq 0 ] && python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
exit $rc
