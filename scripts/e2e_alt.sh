# Live comparison with ALT contigs: the reference's own host code (build/dropin/bwa-gasal2, -t 1) against the device-resident path (device region tail with the ALT
# rules, pairing kernel, SAM writer) on repeat-rich genomes cut into sequences the last of which are named in <prefix>.alt; hard read sets, single-end and paired, default and -a.
run() { echo "== G=$G N=$N $M [$O] $*"; env "$@" timeout 900 python scripts/e2e_dropin.py /tmp/e2e_alt $G $N 1 $M "$O" 2>&1 | grep -a "differing\|IDENTICAL\|rc=\|Error\|error\|rror" | cut -c1-250; }
KW="E2E_GENOME_KW={'repeat_frac': 0.5, 'repeat_copies': (4, 40), 'repeat_len': (200, 1500), 'repeat_div': 0.02}"
O=""
G=3000000 N=40000 M=se_hard run E2E_TAG=a1 E2E_CONTIGS=7 E2E_ALT=3 "$KW"
G=3000000 N=40000 M=pe_hard run E2E_TAG=a2 E2E_CONTIGS=7 E2E_ALT=3 "$KW"
O="-a"
G=3000000 N=30000 M=se_hard run E2E_TAG=a3 E2E_CONTIGS=7 E2E_ALT=3 "$KW"
G=3000000 N=30000 M=pe_hard run E2E_TAG=a4 E2E_CONTIGS=7 E2E_ALT=3 "$KW"
O="-M -Y -h 3,50"
G=3000000 N=30000 M=se_hard run E2E_TAG=a5 E2E_CONTIGS=9 E2E_ALT=4 "$KW"
G=3000000 N=30000 M=pe_hard run E2E_TAG=a6 E2E_CONTIGS=9 E2E_ALT=4 "$KW"
