# Live comparisons beyond the BASELINE shapes: tiny genome (reads at the text ends, many duplicates), 300 bp reads, N-rich reads, thousands of sequences.
# (Reads of mixed lengths are not compared: the reference's host code aborts on them -- bwa_gen_cigar2 rejections, then heap corruption -- while
# this path reports no rejected alignment on the same file; E2E_RAGGED=1 reproduces it.)
run() { echo "== $*"; env "$@" timeout 400 python scripts/e2e_dropin.py /tmp/e2e_w $G $N 1 $M 2>&1 | grep -a "differing\|IDENTICAL\|rc=\|Error\|error" | cut -c1-220; }
G=300000 N=100000 M=se_hard run E2E_TAG=w1
G=300000 N=100000 M=pe_hard run E2E_TAG=w2
G=20000000 N=60000 M=se_hard run E2E_TAG=w3 E2E_READLEN=300
G=20000000 N=60000 M=se_hard run E2E_TAG=w5 E2E_NRATE=0.03
G=20000000 N=60000 M=se_hard run E2E_TAG=w6 E2E_CONTIGS=3000 E2E_READLEN=300
