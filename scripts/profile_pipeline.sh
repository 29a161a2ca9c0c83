# rocprofv3 kernel stats of the native reads -> SAM pipeline (scripts/lanes_probe.py: bmh_aligner_run, 3 x 4 M reads on the hg38-scale index), single-end and paired.
# usage: bash scripts/profile_pipeline.sh <tag>     -> gpurun_out/<tag>_sam_{se,pe}_kernel_stats.csv (copy to profiles/)
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp; export TMPDIR=/tmp
export LANES_CFGS=2x4
rm -rf $R/gpurun_out/prof_sam_se $R/gpurun_out/prof_sam_pe
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_sam_se -- python3 $R/scripts/lanes_probe.py 3100 4000000 > $R/gpurun_out/prof_sam_se.log 2>&1 || exit 1
export LANES_CFGS=3x6
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_sam_pe -- python3 $R/scripts/lanes_probe.py 3100 4000000 pe > $R/gpurun_out/prof_sam_pe.log 2>&1 || exit 1
cd $R
for m in se pe; do
  f=$(find gpurun_out/prof_sam_$m -name "*kernel_stats.csv" | head -1)
  cp $f gpurun_out/${TAG}_sam_${m}_kernel_stats.csv
  find gpurun_out/prof_sam_$m -name "*kernel_trace.csv" -delete
  tail -2 gpurun_out/prof_sam_$m.log
done
