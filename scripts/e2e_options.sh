K=0; run() { K=$((K+1)); echo "== $1 [$2]"; E2E_TAG=opt$K timeout 400 python scripts/e2e_dropin.py /tmp/e2e_o 20000000 60000 1 $1 "$2" 2>&1 | grep -a "differing\|IDENTICAL\|rc=\|Error\|error" | cut -c1-300; }
run se_hard "-k 23 -c 300 -D 0.4 -B 5 -O 7 -E 2 -T 45 -a -h 3"
run pe_hard "-k 21 -B 6 -O 8,9 -E 2,3 -T 50 -U 25 -m 20 -M -Y"
run pe_hard "-S"
run pe_hard "-P"
run se_hard "-M -Y -G 500 -N 10 -X 0.3 -w 30 -Q 30"
run se_hard "-W 5"
run pe_hard "-W 8 -a"
