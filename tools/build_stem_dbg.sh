#!/bin/bash
# Ablation build of the stem kernels for tools/stem_probe.py: a COPY of rspnet_amd/csrc whose conv_stem.hip reads the STEM_DBG
# environment variable into StemParams (bit 0: no output stores, 1: no statistics, 2: no next-halo copy).  The
# product sources carry no such switches.
set -e
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
TMP="$(mktemp -d)"
cp -r "$HERE/../rspnet_amd/csrc" "$TMP/csrc"
python3 - "$TMP/csrc/conv_stem.hip" <<'PY'
import sys
p = sys.argv[1]
s = open(p).read()
def rep(a, b, n=1):
    global s
    assert s.count(a) == n, (s.count(a), a[:60])
    s = s.replace(a, b)
rep("  int wide;   //", "  int dbg;\n  int wide;   //")
rep("  p.x_bytes = (unsigned)((unsigned long long)d->N", "  p.dbg = getenv(\"STEM_DBG\") ? atoi(getenv(\"STEM_DBG\")) : 0;\n  p.x_bytes = (unsigned)((unsigned long long)d->N")
# resident kernel only
rep("        if (addr >= 0 && col < p.Cout) *reinterpret_cast<floatx4*>(p.y + addr + col) = v + b.w[j];",
    "        if (addr >= 0 && col < p.Cout && !(p.dbg & 1)) *reinterpret_cast<floatx4*>(p.y + addr + col) = v + b.w[j];")
rep("      if (c == p.nchunks - 1 && next < p.tiles) issue_halo(next, cur ^ 1);",
    "      if (c == p.nchunks - 1 && next < p.tiles && !(p.dbg & 4)) issue_halo(next, cur ^ 1);")
rep("""    if (p.stat) {
      __syncthreads();
      if (t < 64 && t < p.Cout) {
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          s += red[(w * 64 + t) * 2 + 0];""", """    if (p.stat && !(p.dbg & 2)) {
      __syncthreads();
      if (t < 64 && t < p.Cout) {
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          s += red[(w * 64 + t) * 2 + 0];""")
open(p, "w").write(s)
PY
SRC="$TMP/csrc"
OUT="$HERE/librspnet_hip_stemdbg.so"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result -I"$HERE/../include" -I"$SRC" \
  "$SRC"/errors.hip "$SRC"/conv_igemm.hip "$SRC"/conv_stem.hip "$SRC"/conv_wgrad.hip "$SRC"/bn_pool.hip "$SRC"/pool_gate.hip "$SRC"/head_loss.hip "$SRC"/glue.hip "$SRC"/augment.hip -o "$OUT"
rm -rf "$TMP"
echo "built $OUT"
