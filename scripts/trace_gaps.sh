# rocprofv3 kernel trace of 44 bench steps -> scripts/trace_gaps.py: how many kernels are in flight over the timed steps (gpurun_out/trace_gaps.txt)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; export BENCH_INDEX_CACHE=/tmp/bmh_cache
cd $R && python bench.py --no-pcie --cpu-sample 0 --no-next-rows --verify-sample 0 --steps 2 > /dev/null 2>&1
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/trc
rocprofv3 --kernel-trace --output-format csv -d /tmp/trc -- python3 $R/bench.py --no-pcie --cpu-sample 0 --no-next-rows --verify-sample 0 --steps 40 --warmup 4 > /tmp/trc.json 2>/tmp/trc.err || exit 1
f=$(find /tmp/trc -name "*kernel_trace.csv" | head -1)
python3 $R/scripts/trace_gaps.py $f 0.25 0.85 > $R/gpurun_out/trace_gaps.txt
tail -c 400 /tmp/trc.json | head -c 200; cat $R/gpurun_out/trace_gaps.txt
