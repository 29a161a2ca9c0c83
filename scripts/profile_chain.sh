R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_chain -- python3 $R/scripts/chain_probe.py > /dev/null 2>&1
cd $R; python - <<'PY'
import csv,glob
p=glob.glob('gpurun_out/prof_chain/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.reader(open(p)))[1:]:
    if any(k in r[0] for k in ('chain_','emit_','materialize','merge_kernel','cigar')): print(r[0][:70].ljust(70), r[1], r[3])
PY
find gpurun_out/prof_chain -name "*kernel_trace.csv" -delete
