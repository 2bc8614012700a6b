#!/bin/bash
# A/B tooling: build the product library from the same sources with extra compiler flags into
#   numericalnim_amd/csrc/variants/libnnhip_ode_<name>.so     (git-ignored; travels to the GPU box with gpurun)
# and select it at run time with NNHIP_LIB=<that path>.   usage: scripts/build_variant.sh <name> "<extra hipcc flags>"
set -eu
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME="$1"; EXTRA="${2:-}"; MAKEVARS="${3:-}"   # usage: build_variant.sh <name> "<extra hipcc flags>" ["VAR=value ..." for make]
TMP="$(mktemp -d /tmp/nnhip_variant_XXXX)"
mkdir -p "$TMP/numericalnim_amd" "$TMP/include"
cp -r "$ROOT/numericalnim_amd/csrc" "$TMP/numericalnim_amd/csrc"
cp "$ROOT"/include/*.h* "$TMP/include/"
rm -f "$TMP"/numericalnim_amd/csrc/*.o "$TMP"/numericalnim_amd/csrc/*.so "$TMP"/numericalnim_amd/csrc/embedded_headers.inc
rm -rf "$TMP/numericalnim_amd/csrc/variants"
make -C "$TMP/numericalnim_amd/csrc" -j"$(nproc)" HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function $EXTRA" $MAKEVARS > "$TMP/build.log" 2>&1 || { tail -30 "$TMP/build.log"; exit 1; }
mkdir -p "$ROOT/numericalnim_amd/csrc/variants"
cp "$TMP/numericalnim_amd/csrc/libnnhip_ode.so" "$ROOT/numericalnim_amd/csrc/variants/libnnhip_ode_$NAME.so"
rm -rf "$TMP"
echo "built numericalnim_amd/csrc/variants/libnnhip_ode_$NAME.so"
