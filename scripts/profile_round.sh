# Round profile on the GPU box: bench lines + rocprofv3 kernel stats + PMC passes, condensed into profiles/<tag>_*.
# usage: bash scripts/profile_round.sh <tag>
set -x
TAG=${1:-r01c}
R=$GRAFT_REPO_ROOT; cd $R
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 300 gpurun_out/${TAG}_bench.json
python bench.py --host-jobs > gpurun_out/${TAG}_bench_hostjobs.json 2>/dev/null
python bench.py --paired > gpurun_out/${TAG}_bench_paired.json 2>/dev/null
python bench.py --read-len 300 > gpurun_out/${TAG}_bench_300bp.json 2>/dev/null
cd /tmp; export TMPDIR=/tmp
for d in prof pmc_fetch pmc_write pmc_sq; do rm -rf $R/gpurun_out/${d}_${TAG}; done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG} -- python3 $R/bench.py --steps 5 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_${TAG} -- python3 $R/bench.py --steps 2 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_${TAG} -- python3 $R/bench.py --steps 2 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_sq_${TAG} -- python3 $R/bench.py --steps 2 --warmup 1 > /dev/null 2>&1
cd $R
find gpurun_out/prof_${TAG} -name "*kernel_trace.csv" -delete
python scripts/summarize_profiles.py ${TAG} gpurun_out/prof_${TAG} gpurun_out/pmc_fetch_${TAG} gpurun_out/pmc_write_${TAG} gpurun_out/pmc_sq_${TAG}
cp profiles/${TAG}_kernel_stats_bench.csv profiles/${TAG}_pmc_fetch_write.json profiles/${TAG}_pmc_sq.json gpurun_out/
find gpurun_out/pmc_fetch_${TAG} gpurun_out/pmc_write_${TAG} gpurun_out/pmc_sq_${TAG} -name "*.csv" -size +1M -delete
