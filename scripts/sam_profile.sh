#!/bin/bash
# Where the device time of reads -> SAM goes: scripts/lanes_probe.py under rocprofv3 --kernel-trace --stats, interleaved pairs and 300 bp single-end reads
# (the two next-row rates VERDICT r04 item 7 names).  usage: scripts/sam_profile.sh <tag> [pe|se300|both]
R=$GRAFT_REPO_ROOT; T=${1:-r05}; W=${2:-both}
cd /tmp; export TMPDIR=/tmp
one() {     # <name> <reads> <lanes_probe args...>
	local N=$1; shift
	rm -rf $R/gpurun_out/sam_${T}_$N; mkdir -p $R/gpurun_out/sam_${T}_$N
	rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sam_${T}_$N -- python3 $R/scripts/lanes_probe.py "$@" > $R/gpurun_out/sam_${T}_$N/out.log 2> $R/gpurun_out/sam_${T}_$N/err.log || { tail -5 $R/gpurun_out/sam_${T}_$N/err.log; return 1; }
	find $R/gpurun_out/sam_${T}_$N -name "*kernel_trace.csv" -delete
	local f=$(find $R/gpurun_out/sam_${T}_$N -name "*kernel_stats.csv" | head -1)
	cp $f $R/gpurun_out/sam_${T}_$N.csv
	grep -E "lanes|pairs" $R/gpurun_out/sam_${T}_$N/out.log $R/gpurun_out/sam_${T}_$N/err.log | tail -12
	head -25 $R/gpurun_out/sam_${T}_$N.csv | cut -c1-150
}
export LANES_ITERS=4
if [ "$W" = pe ] || [ "$W" = both ]; then LANES_CFGS=${LANES_CFGS_PE:-4x8} BMH_PAIR_PROFILE=1 one pe 3100 4000000 pe || exit 1; fi
if [ "$W" = se300 ] || [ "$W" = both ]; then LANES_CFGS=2x4 LANES_READ_LEN=300 one se300 3100 4000000 || exit 1; fi
