# Larger live comparisons (more reads: the rare paths -- pairs beyond the pairing kernel's 64 hits, CIGARs beyond the fixed slots, scores close to an integer): the
# reference's own host code (-t 1) against the device-resident path, 300 000 hard reads on a 30 Mbp repeat-rich genome of 24 sequences, with and without ALT contigs
run() { echo "== G=$G N=$N $M [$O] $*"; env "$@" timeout 1100 python scripts/e2e_dropin.py /tmp/e2e_l $G $N 1 $M "$O" 2>&1 | grep -a "differing\|IDENTICAL\|rc=\|Error\|error\|rror" | cut -c1-250; }
KW="E2E_GENOME_KW={'repeat_frac': 0.5, 'repeat_copies': (5, 400), 'repeat_len': (150, 3000), 'repeat_div': 0.02}"
O=""
G=30000000 N=300000 M=se_hard run E2E_TAG=l1 E2E_CONTIGS=24 "$KW"
G=30000000 N=300000 M=pe_hard run E2E_TAG=l2 E2E_CONTIGS=24 "$KW"
G=30000000 N=300000 M=pe_hard run E2E_TAG=l3 E2E_CONTIGS=24 E2E_ALT=5 "$KW"
O="-a"
G=30000000 N=200000 M=se_hard run E2E_TAG=l4 E2E_CONTIGS=24 E2E_ALT=5 "$KW"
