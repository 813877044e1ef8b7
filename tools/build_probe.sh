#!/bin/bash
# Phase-timestamp build of the library (-DRSP_PHASE_PROBE): used only by tools/phase_probe.py.
set -e
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
SRC="$HERE/../rspnet_amd/csrc"
OUT="$HERE/librspnet_hip_probe.so"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DRSP_PHASE_PROBE=1 ${PROBE_DEFS:-} -Wno-unused-result -I"$HERE/../include" -I"$SRC" \
  "$SRC"/errors.hip "$SRC"/conv_igemm.hip "$SRC"/conv_stem.hip "$SRC"/conv_wgrad.hip "$SRC"/bn_pool.hip "$SRC"/pool_gate.hip "$SRC"/head_loss.hip "$SRC"/glue.hip "$SRC"/augment.hip -o "$OUT"
echo "built $OUT"
