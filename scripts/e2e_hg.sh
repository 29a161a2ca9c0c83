# Live comparisons beyond 2^32: a 2.2 Gbp genome (seq_len 4.4e9) of 24 sequences, index built by bmh_index_build and written in the reference's file layout; the reference's
# own host code (-t 1) against the device-resident path on 60 000 hard reads: paired, single-end, paired with five ALT contigs, single-end -a with ALT contigs
KW="E2E_GENOME_KW={'repeat_frac': 0.3, 'repeat_copies': (10, 3000), 'repeat_len': (300, 3000), 'repeat_div': 0.03}"
run() { echo "== G=2.2e9 N=$N $M [$O] $*"; env E2E_CONTIGS=24 E2E_NATIVE_BUILD=1 "$KW" "$@" timeout 1100 python scripts/e2e_dropin.py /tmp/e2e_hg 2200000000 $N 1 $M "$O" 2>&1 | grep -a "built\|differing\|IDENTICAL\|rc=\|Error\|error\|rror\|Killed" | cut -c1-250; }
O=""
N=60000 M=pe_hard run E2E_TAG=hg1
N=60000 M=se_hard run E2E_TAG=hg2
N=60000 M=pe_hard run E2E_TAG=hg3 E2E_ALT=5
O="-a"
N=40000 M=se_hard run E2E_TAG=hg4 E2E_ALT=5
rm -rf /tmp/e2e_hg
