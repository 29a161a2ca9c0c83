# Round profile on the GPU box: bench lines + rocprofv3 kernel stats + PMC passes, condensed into profiles/<tag>_*.
# usage: bash scripts/profile_round.sh <tag> [variants]     (variants: also run --paired and --read-len 300 bench lines)
# Every rocprofv3 run is the same bench.py command (hg38-scale default workload, PCIe loop / CPU baseline / next rows off);
# counters are collected in their own runs, one counter group per run, without any trace option beside them.
set -x
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out
df -h /tmp /dev/shm | tail -2; free -g | head -2
export BENCH_INDEX_CACHE=${BENCH_INDEX_CACHE:-/tmp/bmh_cache}   # first run builds + saves genome and index, the rocprofv3 runs load them
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || { tail -5 gpurun_out/${TAG}_bench.err; exit 1; }
tail -c 300 gpurun_out/${TAG}_bench.json
if [ "$2" = "variants" ]; then
  python bench.py --paired --cpu-sample 0 > gpurun_out/${TAG}_bench_paired.json 2>/dev/null || exit 1
  python bench.py --read-len 300 --cpu-sample 0 > gpurun_out/${TAG}_bench_300bp.json 2>/dev/null || exit 1
fi
cd /tmp; export TMPDIR=/tmp
ARGS="--no-pcie --cpu-sample 0 --no-next-rows"
for d in prof pmc_fetch pmc_write pmc_sq; do rm -rf $R/gpurun_out/${d}_${TAG}; mkdir -p $R/gpurun_out/${d}_${TAG}; done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG} -- python3 $R/bench.py --steps 5 --warmup 1 $ARGS > $R/gpurun_out/prof_${TAG}/bench.json 2> $R/gpurun_out/prof_${TAG}/err.log || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_${TAG} -- python3 $R/bench.py --steps 2 --warmup 1 $ARGS > $R/gpurun_out/pmc_fetch_${TAG}/bench.json 2> $R/gpurun_out/pmc_fetch_${TAG}/err.log || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_${TAG} -- python3 $R/bench.py --steps 2 --warmup 1 $ARGS > $R/gpurun_out/pmc_write_${TAG}/bench.json 2> $R/gpurun_out/pmc_write_${TAG}/err.log || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_sq_${TAG} -- python3 $R/bench.py --steps 2 --warmup 1 $ARGS > $R/gpurun_out/pmc_sq_${TAG}/bench.json 2> $R/gpurun_out/pmc_sq_${TAG}/err.log || exit 1
cd $R
find gpurun_out/prof_${TAG} -name "*kernel_trace.csv" -delete
python scripts/summarize_profiles.py ${TAG} gpurun_out/prof_${TAG} gpurun_out/pmc_fetch_${TAG} gpurun_out/pmc_write_${TAG} gpurun_out/pmc_sq_${TAG}
cp profiles/${TAG}_kernel_stats_bench.csv profiles/${TAG}_pmc.json gpurun_out/
find gpurun_out/pmc_fetch_${TAG} gpurun_out/pmc_write_${TAG} gpurun_out/pmc_sq_${TAG} -name "*.csv" -size +1M -delete
