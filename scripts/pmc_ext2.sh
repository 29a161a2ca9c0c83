# SQ counters of the extension / chaining kernels on the bench workload (two passes: the SQ block has 8 counter slots)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
TAG=${1:-r02}
for p in 1 2; do
  rm -rf $R/gpurun_out/pmc_sq${p}_$TAG
  if [ $p = 1 ]; then C="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"; else C="SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; fi
  rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_sq${p}_$TAG -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-pcie > /dev/null 2> $R/gpurun_out/pmc_err_$p.log; tail -3 $R/gpurun_out/pmc_err_$p.log
done
cd $R
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for p in (1, 2):
    for f in glob.glob("gpurun_out/pmc_sq%d_$TAG/**/*counter_collection.csv" % p, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if not any(x in k for x in ("extend16", "extend_wide", "ext_closed", "chain_wave", "chain_lane", "smem_backward", "smem_forward")): continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_INSTS_VALU", 0)):
    a = acc[k]; print(k[:60], {c: ("%.3g" % (v / max(n[k][c], 1))) for c, v in a.items()}, "launches", n[k].get("SQ_INSTS_VALU"))
PY
find gpurun_out/pmc_sq1_$TAG gpurun_out/pmc_sq2_$TAG -name "*.csv" -size +1M -delete
