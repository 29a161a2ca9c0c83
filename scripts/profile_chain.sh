# Per-kernel times of the stages around the hot path (device job builder, CIGAR stage) on the bench workload.
# usage (GPU box): bash scripts/profile_chain.sh [tag]   -> profiles/<tag>_chain_cigar_kernel_stats.csv
TAG=${1:-r01d}
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_chain
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_chain -- python3 $R/scripts/chain_probe.py > /dev/null 2>&1
cd $R; python - $TAG <<'PY'
import csv, glob, sys
p = glob.glob('gpurun_out/prof_chain/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.reader(open(p)))
keep = [rows[0]] + [r for r in rows[1:] if any(k in r[0] for k in ('chain_', 'emit_', 'materialize', 'merge_kernel', 'cigar'))]
with open(f'profiles/{sys.argv[1]}_chain_cigar_kernel_stats.csv', 'w', newline='') as f:
    csv.writer(f).writerows(keep)
for r in keep[1:]:
    print(r[0][:70].ljust(70), r[1], r[3])
PY
cp profiles/${TAG}_chain_cigar_kernel_stats.csv gpurun_out/
find gpurun_out/prof_chain -name "*kernel_trace.csv" -delete
