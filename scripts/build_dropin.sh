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
rm -rf "$OUT/obj"
echo "built $OUT/bwa-gasal2"
