# Live comparisons beyond 2^32: a 2.2 Gbp genome (seq_len 4.4e9) of 24 sequences, index built by bmh_index_build and written in the reference's file layout; the reference's
# own host code (-t 1) against the device-resident path on 60 000 hard reads: paired, single-end, paired with five ALT contigs, single-end -a with ALT contigs
KW="E2E_GENOME_KW={'repeat_frac': 0.3, 'repeat_copies': (10, 3000), 'repeat_len': (300, 3000), 'repeat_div': 0.03}"
run() { echo "== G=2.2e9 N=$N $M [$O] $*"; env E2E_CONTIGS=24 E2E_NATIVE_BUILD=1 "$KW" "$@" timeout 1100 python scripts/e2e_dropin.py /tmp/e2e_hg 2200000000 $N 1 $M "$O" 2>&1 | grep -a "built\|differing\|IDENTICAL\|rc=\|Error\|error\|rror\|Killed" | cut -c1-250; }
if [ -z "$1" ]; then
O=""
N=60000 M=pe_hard run E2E_TAG=hg1
N=60000 M=se_hard run E2E_TAG=hg2
N=60000 M=pe_hard run E2E_TAG=hg3 E2E_ALT=5
O="-a"
N=40000 M=se_hard run E2E_TAG=hg4 E2E_ALT=5
rm -rf /tmp/e2e_hg
fi
# ... and at BASELINE.json's scale: 3.1 Gbp (seq_len 6.2e9), 200 000 paired reads of 150 bp (configs[3]'s shape), 100 000 single-end reads of 300 bp (configs[4]'s)
if [ "$1" = "baseline" ]; then
run3() { echo "== G=3.1e9 N=$N $M [$O] $*"; env E2E_CONTIGS=24 E2E_NATIVE_BUILD=1 "$KW" "$@" timeout 1100 python scripts/e2e_dropin.py /tmp/e2e_hg 3100000000 $N 1 $M "$O" 2>&1 | grep -a "built\|differing\|IDENTICAL\|rc=\|Error\|error\|rror\|Killed" | cut -c1-250; }
O=""
N=200000 M=pe_hard run3 E2E_TAG=hg5
N=100000 M=se_hard run3 E2E_TAG=hg6 E2E_READLEN=300
rm -rf /tmp/e2e_hg
fi
# ... and BASELINE.json's configs at FULL size against the reference's own SAM: `full_se`: configs[1], 1 M single-end reads of 150 bp; `full_pe`: configs[3], 1 M pairs; `full_300`: configs[4], 1 M single-end reads of 300 bp; `full_10m`: configs[2], 10 M single-end reads
if [ "$1" = "full_se" ] || [ "$1" = "full_pe" ] || [ "$1" = "full_300" ] || [ "$1" = "full_10m" ]; then
run4() { echo "== G=3.1e9 N=$N $M [$O] $*"; env E2E_CONTIGS=24 E2E_NATIVE_BUILD=1 "$KW" "$@" timeout 1150 python scripts/e2e_dropin.py /tmp/e2e_hg 3100000000 $N 1 $M "$O" 2>&1 | grep -a "built\|differing\|IDENTICAL\|rc=\|Error\|error\|rror\|Killed" | cut -c1-250; }
O=""
if [ "$1" = "full_se" ]; then N=1000000 M=se_hard run4 E2E_TAG=hg7; elif [ "$1" = "full_300" ]; then N=1000000 M=se_hard run4 E2E_TAG=hg9 E2E_READLEN=300; elif [ "$1" = "full_10m" ]; then N=10000000 M=se_hard run4 E2E_TAG=hg10; else N=2000000 M=pe_hard run4 E2E_TAG=hg8; fi
rm -rf /tmp/e2e_hg
fi
