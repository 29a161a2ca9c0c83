# Kernel trace of one bench.py run on the GPU box, extension / chaining / seeding kernels listed with their durations.
# usage: bash scripts/quick_trace.sh <tag> [bench args]     (index cache under /tmp/bmh_cache; first run builds it)
TAG=$1; shift
R=$GRAFT_REPO_ROOT; cd $R
export BENCH_INDEX_CACHE=${BENCH_INDEX_CACHE:-/tmp/bmh_cache}
A="--cpu-sample 0 --no-next-rows --no-pcie $*"
# (the exact cache directory of THIS command: genome size, sample interval and generator version are in its name)
[ -f "$(python bench.py $A --print-cache-dir)/meta.json" ] || python bench.py $A --steps 1 --warmup 0 > /dev/null 2>&1
D=$R/gpurun_out/prof_$TAG; rm -rf $D; mkdir -p $D
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/bench.py --steps 4 --warmup 1 $A > $D/bench.json 2> $D/err.log || { tail -5 $D/err.log; exit 1; }
cd $R
find $D -name "*kernel_trace.csv" -delete
python - <<PY
import csv, glob, json
p = glob.glob("$D/**/*kernel_stats.csv", recursive=True)[0]
b = json.loads([l for l in open("$D/bench.json") if l.startswith("{")][0])
print("value", b["value"], "ms_per_step", b["ms_per_step"], "passes", b["passes"])
print({k: v for k, v in b["stage_ms"].items()})
for r in list(csv.reader(open(p)))[1:]:
    if any(k in r[0] for k in ("ext", "chain_", "smem_", "locate", "expand", "emit", "merge", "pack_reads", "cand_")):
        print(r[0].replace("void ", "")[:64].ljust(64), "calls", r[1].rjust(4), " total ms", str(round(float(r[2]) / 1e6, 1)).rjust(8), " avg ms", round(float(r[3]) / 1e6, 3))
PY
