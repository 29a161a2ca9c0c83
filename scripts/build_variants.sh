#!/bin/bash
# A/B builds of one kernel file with different -D flags: build/variants/lib_<tag>.so (git-ignored, travels to the GPU box).
# usage: build_variants.sh <file.hip> <tag1>="<flags>" <tag2>="<flags>" ...   then run with BMH_LIB=build/variants/lib_<tag>.so
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd); SRC=$ROOT/bwa-mem_gpu_amd/csrc; OUT=$ROOT/build/variants
mkdir -p "$OUT"; make -s -C "$SRC" >/dev/null 2>&1
f=$1; shift
for spec in "$@"; do
  tag=${spec%%=*}; flags=${spec#*=}
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -I"$ROOT/include" $flags -c "$SRC/$f" -o "$OUT/$tag.o" 2>/dev/null
    objs=$(ls "$SRC"/*.o | grep -v "/${f%.hip}.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o "$OUT/lib_$tag.so" $objs "$OUT/$tag.o"; rm -f "$OUT/$tag.o"; echo "built lib_$tag.so" ) &
done
wait
