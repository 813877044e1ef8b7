#!/bin/bash
# Phase-timestamp build of the library for tools/phase_probe.py: a COPY of rspnet_amd/csrc with tools/phase_probe.patch applied
# (s_memtime stamps at the phase boundaries of igemm_body, written to a debug buffer) is compiled with -DRSP_PHASE_PROBE.  The
# product sources carry no measurement scaffolding.
set -e
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
TMP="$(mktemp -d)"
cp -r "$HERE/../rspnet_amd/csrc" "$TMP/csrc"
(cd "$TMP" && patch -p2 -d csrc < "$HERE/phase_probe.patch" >/dev/null) || (cd "$TMP/csrc" && patch -p3 < "$HERE/phase_probe.patch")
SRC="$TMP/csrc"
OUT="$HERE/librspnet_hip_probe.so"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DRSP_PHASE_PROBE=1 -Wno-unused-result -I"$HERE/../include" -I"$SRC" \
  "$SRC"/errors.hip "$SRC"/conv_igemm.hip "$SRC"/conv_stem.hip "$SRC"/conv_wgrad.hip "$SRC"/bn_pool.hip "$SRC"/pool_gate.hip "$SRC"/head_loss.hip "$SRC"/glue.hip "$SRC"/augment.hip -o "$OUT"
rm -rf "$TMP"
echo "built $OUT"
