#!/bin/bash
# Live comparisons of paired read sets (the reference's own host code in build/dropin/bwa-gasal2 against the device-resident path) with the mate rescue's
# windows found on the device (BMH_ALIGNER_RESCUE_DEV=1, the host's walk beside it: BMH_RESCUE_CHECK=1) -> gpurun_out/r06_e2e_rescue_dev.txt
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
export BMH_ALIGNER_RESCUE_DEV=1 BMH_RESCUE_CHECK=1
K=0; run() { K=$((K+1)); echo "== $1 [$2] ${3}"; env $3 E2E_TAG=rd$K timeout 400 python scripts/e2e_dropin.py /tmp/e2e_rd 20000000 60000 1 $1 "$2" 2>&1 | grep -a "differing\|IDENTICAL\|rc=\|Error\|error" | cut -c1-300; }
{
run pe ""
run pe_hard ""
run pe_hard "-k 21 -B 6 -O 8,9 -E 2,3 -T 50 -U 25 -m 20 -M -Y"
run pe_hard "-W 8 -a"
run pe_hard "" "E2E_CONTIGS=7 E2E_READLEN=250"
run pe_hard "" "E2E_CONTIGS=7 E2E_ALT=3"
} > gpurun_out/r06_e2e_rescue_dev.txt 2>&1
cat gpurun_out/r06_e2e_rescue_dev.txt
