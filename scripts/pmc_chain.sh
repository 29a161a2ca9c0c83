# SQ counters of the chaining kernels per size class (dispatches told apart by grid size): one or two rocprofv3 --pmc runs of bench.py, index from the cache.
# usage: bash scripts/pmc_chain.sh <tag> [bench args]      (BMH_* knobs in the environment apply)
TAG=$1; shift
R=$GRAFT_REPO_ROOT; cd $R
export BENCH_INDEX_CACHE=${BENCH_INDEX_CACHE:-/tmp/bmh_cache}
A="--cpu-sample 0 --no-next-rows --no-pcie --distinct-batches 2 $*"
[ -f "$(python bench.py $A --print-cache-dir)/meta.json" ] || python bench.py $A --steps 1 --warmup 0 > /dev/null 2>&1
for pass in a b; do
  D=$R/gpurun_out/pmcchain_${TAG}_$pass; rm -rf $D; mkdir -p $D
  # (PMC_MEM=1: the two passes are FETCH_SIZE and WRITE_SIZE instead -- KB per launch, with SQ_WAVES beside them for the launch count)
  if [ -n "$PMC_MEM" ]; then if [ $pass = a ]; then CTR="FETCH_SIZE SQ_WAVES"; else CTR="WRITE_SIZE SQ_WAVES"; fi
  elif [ $pass = a ]; then CTR="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"; else CTR="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"; fi
  ( cd /tmp; export TMPDIR=/tmp; rocprofv3 --pmc $CTR --output-format csv -d $D -- python3 $R/bench.py --steps 2 --warmup 1 $A > $D/bench.json 2> $D/err.log ) || { tail -5 $D/err.log; continue; }
  python - <<PY
import csv, glob, json, collections
p = glob.glob("$D/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
rows = list(csv.DictReader(open(p)))
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if not any(x in k for x in ("chain_", "emit_")): continue
    key = (k, r.get("Grid_Size", "?"), r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "?")))
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": n[key] += 1
print("# pass $pass: counters per launch (kernel, grid, lds)")
for key in sorted(acc):
    v = acc[key]; m = max(n[key], 1)
    print(("%-34s grid %-8s lds %-7s launches %-3d " % (key[0][:34], key[1], key[2], n[key])) + "  ".join("%s %.4g" % (c.replace("SQ_", ""), v[c] / m) for c in sorted(v)))
PY
  find $D -name "*.csv" -size +1M -delete
done
