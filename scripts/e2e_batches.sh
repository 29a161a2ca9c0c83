# Live comparison with the reference's OWN batch cuts (no -K: 10 Mbases per batch at -t 1, the insert-size statistics are per batch): the build with the reference's four
# batch-relative seq[] subscripts corrected (bwa-gasal2-seqidx; the stock build reads another read's bases in its patch test once there is a second batch) against
# bwamem_hip.aligner's default cuts (bmh_aligner_run_fasta: bseq_read's rule), 300 000 paired reads = five batches
run() { echo "== G=$G N=$N $M [$O] $*"; env "$@" timeout 1100 python scripts/e2e_dropin.py /tmp/e2e_b $G $N 1 $M "$O" 2>&1 | grep -a "differing\|IDENTICAL\|rc=\|Error\|error\|rror" | cut -c1-250; }
KW="E2E_GENOME_KW={'repeat_frac': 0.5, 'repeat_copies': (5, 400), 'repeat_len': (150, 3000), 'repeat_div': 0.02}"
O=""
G=30000000 N=300000 M=pe_hard run E2E_TAG=k1 E2E_EXE=bwa-gasal2-seqidx E2E_DEFAULT_K=1 E2E_CONTIGS=24 "$KW"
G=30000000 N=300000 M=pe_hard run E2E_TAG=k2 E2E_EXE=bwa-gasal2-seqidx E2E_DEFAULT_K=1 E2E_CONTIGS=24 E2E_ALT=5 "$KW"
# ... and with four host threads on the reference's side (batches of 40 Mbases: three of them for 600 000 reads)
run4t() { echo "== G=$G N=$N $M -t 4 [$O] $*"; env "$@" timeout 1100 python scripts/e2e_dropin.py /tmp/e2e_b $G $N 4 $M "$O" 2>&1 | grep -a "differing\|IDENTICAL\|rc=\|Error\|error\|rror" | cut -c1-250; }
G=30000000 N=600000 M=pe_hard run4t E2E_TAG=k3 E2E_EXE=bwa-gasal2-seqidx E2E_DEFAULT_K=1 E2E_CONTIGS=24 E2E_ALT=5 "$KW"
