set -x
R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests -x -q -m gpu 2>&1 | tail -2
python bench.py > gpurun_out/bench_r01b.json 2> gpurun_out/bench_r01b.err; tail -c 600 gpurun_out/bench_r01b.json
python bench.py --paired > gpurun_out/bench_r01b_pe.json 2>/dev/null; cut -c1-200 gpurun_out/bench_r01b_pe.json
python bench.py --read-len 300 --genome-mbp 1000 > gpurun_out/bench_r01b_300.json 2>/dev/null; cut -c1-200 gpurun_out/bench_r01b_300.json
cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r01b $R/gpurun_out/pmc_fetch_r01b $R/gpurun_out/pmc_write_r01b
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01b -- python3 $R/bench.py --steps 5 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_r01b -- python3 $R/bench.py --steps 2 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_r01b -- python3 $R/bench.py --steps 2 --warmup 1 > /dev/null 2>&1
cd $R
find gpurun_out/prof_r01b -name "*kernel_trace.csv" -delete
python scripts/summarize_profiles.py r01b gpurun_out/prof_r01b gpurun_out/pmc_fetch_r01b gpurun_out/pmc_write_r01b
cp profiles/r01b_* gpurun_out/
find gpurun_out/pmc_fetch_r01b gpurun_out/pmc_write_r01b -name "*.csv" -size +20M -delete
