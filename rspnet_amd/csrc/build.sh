#!/bin/bash
# Build librspnet_hip.so for gfx950 (in-tree, next to the Python package).
set -e
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../librspnet_hip.so"
INC="$HERE/../../include"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
OBJS=()
for f in errors conv_igemm conv_stem conv_wgrad bn_pool pool_gate head_loss glue augment; do
  "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -I"$INC" -I"$HERE" -c "$HERE/$f.hip" -o "$HERE/$f.o" &
done
wait
for f in errors conv_igemm conv_stem conv_wgrad bn_pool pool_gate head_loss glue augment; do OBJS+=("$HERE/$f.o"); done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT" "${OBJS[@]}"
echo "built $OUT"
