# kernel trace of a bench.py run + scripts/timeline.py over its last 120 ms.  usage: bash scripts/timeline.sh <tag> [bench args]
TAG=$1; shift
R=$GRAFT_REPO_ROOT; cd $R
export BENCH_INDEX_CACHE=${BENCH_INDEX_CACHE:-/tmp/bmh_cache}
A="--cpu-sample 0 --no-next-rows --no-pcie $*"
# (the exact cache directory of THIS command: genome size, sample interval and generator version are in its name)
[ -f "$(python bench.py $A --print-cache-dir)/meta.json" ] || python bench.py $A --steps 1 --warmup 0 > /dev/null 2>&1
D=$R/gpurun_out/tl_$TAG; rm -rf $D; mkdir -p $D
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $R/bench.py --steps 4 --warmup 1 $A > $D/bench.json 2> $D/err.log || { tail -5 $D/err.log; exit 1; }
cd $R
T=$(find $D -name "*kernel_trace.csv" | head -1)
python scripts/timeline.py $T 130 > gpurun_out/timeline_$TAG.txt
find $D -name "*kernel_trace.csv" -delete
tail -c 400 $D/bench.json | head -c 0; python -c "
import json; b=json.loads([l for l in open('$D/bench.json') if l.startswith('{')][0]); print('value', b['value'], 'ms_per_step', b['ms_per_step'])"
