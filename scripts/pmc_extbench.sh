# SQ counters of the extension microbenchmark (scripts/ext_bench.py), per kernel.  usage: bash scripts/pmc_extbench.sh [ext_bench args]
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for pass in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  rm -rf $R/gpurun_out/pmc_extb
  rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/pmc_extb -- python3 $R/scripts/ext_bench.py "$@" > $R/gpurun_out/pmc_extb.log 2>&1 || { tail -5 $R/gpurun_out/pmc_extb.log; exit 1; }
  python3 - <<PY
import csv, glob, collections
p = glob.glob("$R/gpurun_out/pmc_extb/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(p)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if not ("extpk" in k or "extend16" in k or "extend_wide" in k or "closed_form" in k): continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in sorted(acc, key=lambda k: -sum(acc[k].values()))[:6]:
    print(k[:40].ljust(40), {c: "%.4g" % (v / n[k][c]) for c, v in acc[k].items()})
PY
done
rm -rf $R/gpurun_out/pmc_extb
