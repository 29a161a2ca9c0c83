#!/bin/bash
# Drop-in demonstration (build container only: needs /root/reference).
# Compiles the REFERENCE's own host sources from where they lie -- unchanged -- against this
# repository's two boundary headers (include/seed_gen.h, include/gasal2_root/GASAL2/include/*.h)
# and links them with libbwamem_hip.so instead of libseed.a + libgasal.a + CUDA:
#     build/dropin/bwa-gasal2     (git-ignored; travels to the GPU box with the snapshot)
# Same flags as the reference Makefile (:14,:18), no CUDA, no GASAL2 submodule.
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
REF=${REF:-/root/reference}
OUT=$ROOT/build/dropin
mkdir -p "$OUT/obj"
make -s -C "$ROOT/bwa-mem_gpu_amd/csrc"
CFLAGS="-Wall -Wno-unused-function -O3 -msse4.2 -std=c++11 -fpermissive -w -DHAVE_PTHREAD -DUSE_MALLOC_WRAPPERS"
INC="-include $ROOT/include/seed_gen.h -I$ROOT/include/gasal2_root/src -I$REF/src"
LOBJS="utils kthread kstring ksw bwt bntseq bwa bwamem bwamem_pair bwamem_extra malloc_wrap QSufSort bwt_gen rope rle is bwtindex"
AOBJS="bwashm bwase bwaseqio bwtgap bwtaln bamlite bwape kopen pemerge maxk bwtsw2_core bwtsw2_main bwtsw2_aux bwt_lite bwtsw2_chain fastmap bwtsw2_pair main"
SHD=""   # the optional SHD filter (-F) needs boost, absent from this image: its 1 entry point is left unresolved
pids=()
for o in $LOBJS $AOBJS; do g++ -c $CFLAGS $INC "$REF/src/$o.c" -o "$OUT/obj/$o.o" & pids+=($!); done
for o in $SHD; do g++ -c $CFLAGS $INC "$REF/src/$o.cpp" -o "$OUT/obj/$o.o" & pids+=($!); done
for p in "${pids[@]}"; do wait "$p"; done
g++ $CFLAGS "$OUT"/obj/*.o -o "$OUT/bwa-gasal2" -L"$ROOT/bwa-mem_gpu_amd" -lbwamem_hip -Wl,-rpath,'$ORIGIN/../../bwa-mem_gpu_amd' \
    -Wl,--unresolved-symbols=ignore-all -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -lm -lz -ldl -lpthread -lrt
echo "built $OUT/bwa-gasal2"
# Second binary for the root-cause demonstration of INTEGRATION.md section 2 (-t >= 4): identical, except that the four places of
# mem_align1_core that index seq[] with a batch-RELATIVE read index (src/bwamem.c:2228, 2251, 2295, 2313 -- the fill loop at :2046-2089
# and w_regs[j + batch_start_idx] at :2326 use the absolute one) get `+ batch_start_idx`.  The corrected text exists only in a
# temporary directory while this script runs; nothing of the reference's source is stored in the repository.
TMPD=$(mktemp -d)
sed -n '2228p;2251p' "$REF/src/bwamem.c" | grep -c 'seq\[r\]\.l_seq' | grep -qx 2 || { echo "src/bwamem.c is not the expected text"; exit 1; }
sed -n '2295p;2313p' "$REF/src/bwamem.c" | grep -c 'seq\[j\]\.' | grep -qx 2 || { echo "src/bwamem.c is not the expected text"; exit 1; }
sed -e '2228s/seq\[r\]/seq[r + batch_start_idx]/' -e '2251s/seq\[r\]/seq[r + batch_start_idx]/' \
    -e '2295s/seq\[j\]/seq[j + batch_start_idx]/' -e '2313s/seq\[j\]/seq[j + batch_start_idx]/' "$REF/src/bwamem.c" > "$TMPD/bwamem.c"
g++ -c $CFLAGS $INC -I"$REF/src" "$TMPD/bwamem.c" -o "$OUT/obj/bwamem.o"
g++ $CFLAGS "$OUT"/obj/*.o -o "$OUT/bwa-gasal2-seqidx" -L"$ROOT/bwa-mem_gpu_amd" -lbwamem_hip -Wl,-rpath,'$ORIGIN/../../bwa-mem_gpu_amd' \
    -Wl,--unresolved-symbols=ignore-all -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -lm -lz -ldl -lpthread -lrt
rm -rf "$TMPD" "$OUT/obj"
echo "built $OUT/bwa-gasal2-seqidx"
