# Round profile on the GPU box: bench lines + rocprofv3 kernel stats + PMC passes, condensed into profiles/<tag>*_{kernel_stats_bench.csv,pmc.json}.
# usage: bash scripts/profile_round.sh <tag> [variants]     (variants: also profile the --paired and --read-len 300 workloads, tags <tag>_paired / <tag>_300bp)
# Every rocprofv3 run is the same bench.py command (hg38-scale default workload, PCIe loop / CPU baseline / next rows off);
# counters are collected in their own runs, one counter group per run, without any trace option beside them; the program itself
# (python3 bench.py ...) stands directly behind `--`.
set -x
TAG=${1:-r05}
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out
df -h /tmp /dev/shm | tail -2; free -g | head -2
export BENCH_INDEX_CACHE=${BENCH_INDEX_CACHE:-/tmp/bmh_cache}   # first run builds + saves genome and index, the rocprofv3 runs load them
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || { tail -5 gpurun_out/${TAG}_bench.err; exit 1; }
tail -c 300 gpurun_out/${TAG}_bench.json
if [ "$2" = "variants" ]; then
  python bench.py --paired --cpu-sample 20000 > gpurun_out/${TAG}_bench_paired.json 2>/dev/null || exit 1
  python bench.py --read-len 300 --cpu-sample 20000 > gpurun_out/${TAG}_bench_300bp.json 2>/dev/null || exit 1
fi
profile_one() {     # <tag> <extra bench args>
  local T=$1; shift
  local ARGS="--no-pcie --cpu-sample 0 --no-next-rows $*"
  cd /tmp; export TMPDIR=/tmp
  for d in prof pmc_fetch pmc_write pmc_sq; do rm -rf $R/gpurun_out/${d}_${T}; mkdir -p $R/gpurun_out/${d}_${T}; done
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${T} -- python3 $R/bench.py --steps 5 --warmup 1 $ARGS > $R/gpurun_out/prof_${T}/bench.json 2> $R/gpurun_out/prof_${T}/err.log || return 1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_${T} -- python3 $R/bench.py --steps 2 --warmup 1 $ARGS > $R/gpurun_out/pmc_fetch_${T}/bench.json 2> $R/gpurun_out/pmc_fetch_${T}/err.log || return 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_${T} -- python3 $R/bench.py --steps 2 --warmup 1 $ARGS > $R/gpurun_out/pmc_write_${T}/bench.json 2> $R/gpurun_out/pmc_write_${T}/err.log || return 1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_sq_${T} -- python3 $R/bench.py --steps 2 --warmup 1 $ARGS > $R/gpurun_out/pmc_sq_${T}/bench.json 2> $R/gpurun_out/pmc_sq_${T}/err.log || return 1
  cd $R
  find gpurun_out/prof_${T} -name "*kernel_trace.csv" -delete
  python scripts/summarize_profiles.py ${T} gpurun_out/prof_${T} gpurun_out/pmc_fetch_${T} gpurun_out/pmc_write_${T} gpurun_out/pmc_sq_${T} || return 1
  cp profiles/${T}_kernel_stats_bench.csv profiles/${T}_pmc.json gpurun_out/
  find gpurun_out/pmc_fetch_${T} gpurun_out/pmc_write_${T} gpurun_out/pmc_sq_${T} -name "*.csv" -size +1M -delete
}
profile_one ${TAG} || exit 1
if [ "$2" = "variants" ]; then
  profile_one ${TAG}_paired --paired || exit 1
  profile_one ${TAG}_300bp --read-len 300 || exit 1
fi
