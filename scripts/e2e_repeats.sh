run() { echo "== G=$G N=$N $M [$O] $*"; env "$@" timeout 600 python scripts/e2e_dropin.py /tmp/e2e_r $G $N 1 $M "$O" 2>&1 | grep -a "differing\|IDENTICAL\|rc=\|Error\|error\|rror" | cut -c1-250; }
O=""
G=5000000 N=40000 M=se_hard run E2E_TAG=r1 "E2E_GENOME_KW={'repeat_frac': 0.6, 'repeat_copies': (2000, 5000), 'repeat_div': 0.02}"
G=5000000 N=40000 M=pe_hard run E2E_TAG=r2 "E2E_GENOME_KW={'repeat_frac': 0.6, 'repeat_copies': (2000, 5000), 'repeat_div': 0.02}"
O="-a -c 2000"
G=5000000 N=20000 M=se_hard run E2E_TAG=r3 "E2E_GENOME_KW={'repeat_frac': 0.5, 'repeat_copies': (300, 900), 'repeat_len': (300, 1500), 'repeat_div': 0.01}"
