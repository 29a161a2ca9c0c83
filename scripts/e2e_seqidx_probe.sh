#!/bin/bash
# Root cause of the reference's garbage records at -t >= 4, shown on the GPU box:
#   stock build of the reference host code (build/dropin/bwa-gasal2) vs the build with the four batch-relative seq[] indices of
#   mem_align1_core made absolute (bwa-gasal2-seqidx, scripts/build_dropin.sh), run three times each at -t 8 WITHOUT -K (several
#   chunks, work stealing) on a hard single-end set: run-to-run differences and records with an impossible AS.
# usage: bash scripts/e2e_seqidx_probe.sh [reads=200000]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
N=${1:-200000}
W=/tmp/seqidx_w; rm -rf $W
E2E_LONGDEL=7 python scripts/e2e_dropin.py $W 2000000 $N 1 se_hard 2>&1 | grep -E "differing records|SAM IDENTICAL|gase_aln rc" | sed "s/^/[setup, stock -t 1 with -K] /"
for exe in bwa-gasal2 bwa-gasal2-seqidx; do
  E2E_EXE=$exe python scripts/e2e_race_probe.py $W 8 3 se 2>&1 | sed "s/^/[$exe -t 8] /"
done
