# per-dispatch durations of the chaining kernels in one bench run (classes of chain_wave_kernel told apart by their grid size)
R=$GRAFT_REPO_ROOT; cd $R
export BENCH_INDEX_CACHE=${BENCH_INDEX_CACHE:-/tmp/bmh_cache}
A="--cpu-sample 0 --no-next-rows --no-pcie $*"
# (the exact cache directory of THIS command: genome size, sample interval and generator version are in its name)
[ -f "$(python bench.py $A --print-cache-dir)/meta.json" ] || python bench.py $A --steps 1 --warmup 0 > /dev/null 2>&1
D=$R/gpurun_out/prof_chain; rm -rf $D; mkdir -p $D
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $R/bench.py --steps 2 --warmup 1 $A > $D/bench.json 2> $D/err.log || { tail -5 $D/err.log; exit 1; }
cd $R
python - <<PY
import csv, glob, collections
p = glob.glob("$D/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(p)) if "chain_" in r["Kernel_Name"] or "extpk" in r["Kernel_Name"]]
t0 = min(int(r["Start_Timestamp"]) for r in rows)
acc = collections.defaultdict(list)
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    g = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1) if "Workgroup_Size" in r else r.get("Grid_Size_X", "?")
    acc[(k, g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for (k, g), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(k[:40].ljust(40), "blocks", str(g).rjust(6), "launches", len(v), "avg ms %.3f" % (sum(v) / len(v)), "max %.3f" % max(v))
PY
rm -rf $D
