# Live comparisons beyond the BASELINE shapes (reference binary vs bwamem_hip.aligner, records compared as text; run on the GPU box):
# tiny genome (reads at the text ends, many duplicates), 300 bp reads single-end and paired, N-rich reads, thousands of sequences,
# option sets on top.  (Reads of mixed lengths are not compared: the reference's host code aborts on them -- bwa_gen_cigar2
# rejections, then heap corruption -- while this path reports no rejected alignment on the same file; E2E_RAGGED=1 reproduces it.)
run() { echo "== G=$G N=$N $M [$O] $*"; env "$@" timeout 400 python scripts/e2e_dropin.py /tmp/e2e_w $G $N 1 $M "$O" 2>&1 | grep -a "differing\|IDENTICAL\|rc=\|Error\|error" | cut -c1-220; }
O=""
G=300000 N=100000 M=se_hard run E2E_TAG=w1
G=300000 N=100000 M=pe_hard run E2E_TAG=w2
G=20000000 N=60000 M=se_hard run E2E_TAG=w3 E2E_READLEN=300
G=20000000 N=60000 M=se_hard run E2E_TAG=w5 E2E_NRATE=0.03
G=20000000 N=60000 M=se_hard run E2E_TAG=w6 E2E_CONTIGS=3000 E2E_READLEN=300
G=20000000 N=60000 M=pe_hard run E2E_TAG=x1 E2E_READLEN=300
O="-a -M -h 2"
G=20000000 N=60000 M=pe_hard run E2E_TAG=x2 E2E_CONTIGS=2000
O="-a -T 20 -k 15"
G=5000000 N=60000 M=se_hard run E2E_TAG=x3
O="-c 5 -D 0.9 -N 3"
G=5000000 N=60000 M=pe_hard run E2E_TAG=x4
# read names with read numbers and comments ("p7/1", "p7/2 x:y z"; "r7/2<TAB>comment"): both sides cut them (trim_readno, src/bwa.c:27-31)
O=""
G=5000000 N=40000 M=pe_hard run E2E_TAG=n1 E2E_READNO=1
G=5000000 N=40000 M=se_hard run E2E_TAG=n2 E2E_READNO=1
# read group: -R '@RG\tID:grp1\tSM:s1' (RG:Z:grp1 on every record, the unmapped ones too)
O='-R @RG\tID:grp1\tSM:s1'
G=5000000 N=40000 M=pe_hard run E2E_TAG=g1
G=5000000 N=40000 M=se_hard run E2E_TAG=g2
O='-a -R @RG\tID:x.y-7\tPL:ILLUMINA'
G=5000000 N=30000 M=se_hard run E2E_TAG=g3 E2E_CONTIGS=6 E2E_ALT=2
