#!/bin/bash
# A variant build of the library from a COPY of rspnet_amd/csrc with one sed expression applied (threshold sweeps without ablation
# macros in the product sources):  bash tools/build_variant.sh NAME FILE 'SED-EXPR'  ->  tools/librspnet_hip_NAME.so
# A/B on the GPU box: bash tools/ab_lib.sh tools/librspnet_hip_NAME.so [archs...]
set -e
NAME="$1"; FILE="$2"; EXPR="$3"
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
TMP="$(mktemp -d)"
cp -r "$HERE/../rspnet_amd/csrc" "$TMP/csrc"
sed -i "$EXPR" "$TMP/csrc/$FILE"
if cmp -s "$TMP/csrc/$FILE" "$HERE/../rspnet_amd/csrc/$FILE"; then echo "the expression changed nothing"; exit 1; fi
SRC="$TMP/csrc"
OUT="$HERE/librspnet_hip_$NAME.so"
OBJS=()
for f in errors conv_igemm conv_stem conv_wgrad bn_pool pool_gate head_loss glue augment; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -I"$HERE/../include" -I"$SRC" -c "$SRC/$f.hip" -o "$TMP/$f.o" &
done
wait
for f in errors conv_igemm conv_stem conv_wgrad bn_pool pool_gate head_loss glue augment; do OBJS+=("$TMP/$f.o"); done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" "${OBJS[@]}"
rm -rf "$TMP"
echo "built $OUT"
