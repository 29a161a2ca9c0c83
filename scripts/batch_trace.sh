# every kernel dispatch of the LAST batch of a one-batch-at-a-time bench run, in start order, with start / duration (ms) relative to the
# batch's first kernel: the critical path of a batch.  usage: bash scripts/batch_trace.sh [min_ms=0.05] [bench args]
MIN=${1:-0.05}; shift
R=$GRAFT_REPO_ROOT; cd $R
export BENCH_INDEX_CACHE=${BENCH_INDEX_CACHE:-/tmp/bmh_cache}
A="--cpu-sample 0 --no-next-rows --no-pcie --inflight 1 $*"
# (the exact cache directory of THIS command: genome size, sample interval and generator version are in its name)
[ -f "$(python bench.py $A --print-cache-dir)/meta.json" ] || python bench.py $A --steps 1 --warmup 0 > /dev/null 2>&1
D=$R/gpurun_out/bt; rm -rf $D; mkdir -p $D
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $R/bench.py --steps 2 --warmup 1 $A > $D/bench.json 2> $D/err.log || { tail -5 $D/err.log; exit 1; }
cd $R
python - $MIN <<PY
import csv, glob, sys
mn = float(sys.argv[1])
p = glob.glob("$D/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(p)), key=lambda r: int(r["Start_Timestamp"]))
# the last batch starts at the last pack_reads_kernel
starts = [i for i, r in enumerate(rows) if "pack_reads_kernel" in r["Kernel_Name"]]
rows = rows[starts[-1]:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s = (int(r["Start_Timestamp"]) - t0) / 1e6; d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if d >= mn:
        print(f"{s:8.3f} +{d:7.3f}  q{r.get('Queue_Id', '?'):>3}  {r['Kernel_Name'].split('(')[0].replace('void ', '')[:70]}  grid {r.get('Grid_Size', '')}")
PY
rm -rf $D
