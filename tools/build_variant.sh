#!/bin/bash
# Build a variant of libbie_hip.so with ONE translation unit recompiled under extra -D flags (ablations / A-B arms for tools/*):
#   tools/build_variant.sh <name> <file.hip> "<-Dflags>"   ->  bitorch-engine_amd/variants/<name>/libbie_hip.so
# The other objects come from bitorch-engine_amd/build/ (run `make -C bitorch-engine_amd` first).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; PKG="$ROOT/bitorch-engine_amd"
name="$1"; file="$2"; flags="$3"
base="$(basename "$file" .hip)"
mkdir -p "$PKG/variants/$name"
extra=""
{ [ "$base" = "mpq_gemm" ] || [ "$base" = "mpq_prod" ]; } && extra="-fno-slp-vectorize"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function $extra $flags -c "$PKG/csrc/$base.hip" -o "$PKG/variants/$name/$base.o"
objs=$(ls "$PKG"/build/*.o | grep -v "/$base.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o "$PKG/variants/$name/libbie_hip.so" "$PKG/variants/$name/$base.o" $objs
echo "built variants/$name"
