# Live comparisons on tiny read sets (1, 3 single-end reads; 1, 6, 13, 15, 20 pairs: too few pairs for insert-size statistics, no rescue, no pairing), with and without ALT contigs
run() { echo "== G=$G N=$N $M [$O] $*"; env "$@" timeout 400 python scripts/e2e_dropin.py /tmp/e2e_t $G $N 1 $M "$O" 2>&1 | grep -a "differing\|IDENTICAL\|rc=\|Error\|error\|rror" | cut -c1-220; }
O=""
G=300000 N=2 M=pe_hard run E2E_TAG=t1
G=300000 N=12 M=pe_hard run E2E_TAG=t2
G=300000 N=30 M=pe_hard run E2E_TAG=t3
G=300000 N=1 M=se_hard run E2E_TAG=t4
G=300000 N=3 M=se_hard run E2E_TAG=t5
G=300000 N=40 M=pe_hard run E2E_TAG=t6 E2E_CONTIGS=5 E2E_ALT=2
O="-a"
G=300000 N=26 M=pe_hard run E2E_TAG=t7
