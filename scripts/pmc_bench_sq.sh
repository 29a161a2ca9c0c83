# SQ counters of the bench's extension / chaining kernels (one rocprofv3 --pmc run of bench.py, index from the cache).
# usage: bash scripts/pmc_bench_sq.sh <tag> [bench args]
TAG=$1; shift
R=$GRAFT_REPO_ROOT; cd $R
export BENCH_INDEX_CACHE=${BENCH_INDEX_CACHE:-/tmp/bmh_cache}
A="--cpu-sample 0 --no-next-rows --no-pcie $*"
# (the exact cache directory of THIS command: genome size, sample interval and generator version are in its name)
[ -f "$(python bench.py $A --print-cache-dir)/meta.json" ] || python bench.py $A --steps 1 --warmup 0 > /dev/null 2>&1
D=$R/gpurun_out/pmcsq_$TAG; rm -rf $D; mkdir -p $D
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $D -- python3 $R/bench.py --steps 2 --warmup 1 $A > $D/bench.json 2> $D/err.log || { tail -5 $D/err.log; exit 1; }
cd $R
python - <<PY
import csv, glob, json, collections
p = glob.glob("$D/**/*counter_collection.csv", recursive=True)[0]
b = json.loads([l for l in open("$D/bench.json") if l.startswith("{")][0])
print("passes", b["passes"], "stage_ms_isolated", b["stage_ms_isolated"])
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(p)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if not any(x in k for x in ("ext", "chain_wave", "chain_lane")): continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_INSTS_VALU": n[k] += 1
tot = 0.0
for k in sorted(acc, key=lambda k: -acc[k]["SQ_INSTS_VALU"]):
    v = acc[k]
    if "ext" in k: tot += v["SQ_INSTS_VALU"]
    if v["SQ_INSTS_VALU"] / max(n[k], 1) > 2e7:
        print(k[:44].ljust(44), "launches", n[k], "VALU/launch %.3e" % (v["SQ_INSTS_VALU"] / n[k]), "busy_frac %.2f" % (4 * v["SQ_ACTIVE_INST_VALU"] / max(1024 * v["GRBM_GUI_ACTIVE"] / 8, 1)), "waves/launch %d" % (v["SQ_WAVES"] / n[k]))
print("extension family VALU wave-instr per hot-path pass: %.4e" % (tot / b["passes"]["extend"]))
PY
find $D -name "*.csv" -size +1M -delete
