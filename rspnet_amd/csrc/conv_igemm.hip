// Implicit-GEMM 3-D convolution for gfx950 on the exact-fp32 matrix pipe (v_mfma_f32_32x32x2_f32).
//
// One kernel serves nn.Conv3d forward (reference call sites: models/c3d.py:21-52 & friends) and, through
// per-stride-parity "affine gather" geometry, its input gradient (autograd conv backward-input).
//
//   out[row, co] = sum_{tap, ci} in[ gather(row, tap), ci ] * Wp[co, tap*Cin + ci]      (+ bias[co])
//   gather(row, tap): row -> (n, gd, gh, gw) on the output grid;  in_coord = g*S + off(tap)  per dim
//
// GEMM view: M = rows (positions), N = Cout, K = taps*Cin.  NDHWC activations make each (row, tap) slice a
// contiguous Cin run, the packed weight makes each co row a contiguous K run, so both LDS tiles are
// [row][k] with k contiguous and every MFMA operand fragment is one ds_read_b128.
//
// Tile: BM x BN x 32 per 256-thread workgroup (4 waves), register-staged double-buffered LDS, one barrier per
// K-chunk.  The fp32 MFMA issues every 64 cycles per SIMD, so a 32-deep chunk is 4096 MFMA-cycles per wave
// against 8 x 16-B global loads per thread: staging hides completely behind the matrix pipe; tap re-reads of the
// input are served by L2 (27 x re-read of conv2's 411 MB input = 1.6 TB/s of L2 traffic, L2 peak 34 TB/s).
#include "common.h"
#include <algorithm>
#include <atomic>
#include <type_traits>
#include "conv_stem.h"

namespace {

constexpr int BK = 32;         // K-chunk (floats)
constexpr int LDK = BK + 4;    // LDS row pitch: 36 floats => ds_read_b128 conflict-free (9r mod 16 distinct)
constexpr int MAX_TAPS = 343;  // 7x7x7

// 64 B of zeros: out-of-bounds taps / rows / K-tail lanes load from here instead of branching around the load
__device__ __attribute__((aligned(64))) float g_zero[16];

// Address of g_zero on the CURRENT device (a __device__ symbol has one instance per device; one process per GPU is the
// supported model, but a process that drives several devices must not reuse another device's address).
const float* igemm_zero_page() {
  static const float* cache[64] = {nullptr};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!cache[dev]) {
    void* z = nullptr;
    if (hipGetSymbolAddress(&z, HIP_SYMBOL(g_zero)) != hipSuccess || !z) return nullptr;
    cache[dev] = reinterpret_cast<const float*>(z);
  }
  return cache[dev];
}

// K order of the packed weights / of the chunks the kernels walk.  Tap-major (k = tap*C + c) makes a 256-channel row come back
// every 8 chunks per tap and every input row be fetched 27 times from beyond L2 (the 96 tiles an XCD runs at once stream tens
// of MB between two visits); channel-slice-major (all taps of a 32-channel slice back to back: k = slice*ntaps*32 + tap*32 + c%32)
// puts a row's w-neighbour taps in adjacent chunks.  Measured on C3D (PMC FETCH_SIZE x2 per launch): conv3b 4.5 -> 0.95 GB,
// conv4b 1.8 -> 0.67 GB, conv2 dgrad 9.7 -> 1.5 GB (conv2 fwd + dgrad average), at unchanged time.  Used when the GEMM's
// channel count is a multiple of 32 (every chunk is then one tap of one slice), there is more than one tap, and K is long
// (>= 48 chunks; 96 until round 5).  (Round 3, per-tile kernels: a threshold of 32 or 0 chunks measured the same step times on all
// four backbones, +-0.5 %; the short-K layers' 64..192-channel rows never thrashed, so they keep the tap-major kernel.)
// (threshold: 48 chunks since round 5 — C3D conv2's 54-chunk forward moved from the per-tile tap-major kernel to the persistent
//  slice-major one: 131 -> 137 TF, C3D +0.7 % per step on two boxes; R3D-18 / R(2+1)D / S3D-G within +-0.3 %; at 32 chunks R3D-18
//  loses 0.8 %: profiles/r05/experiments_r5.txt)
__host__ __device__ inline bool k_slice_major(int C, int ntaps) { return ntaps > 1 && (C & 31) == 0 && C * ntaps >= 48 * 32; }
__host__ __device__ inline int k_index(bool slice_major, int tap, int c, int C, int ntaps) {
  return slice_major ? (c >> 5) * (ntaps * 32) + tap * 32 + (c & 31) : tap * C + c;
}


struct IgemmParams {
  const float* __restrict__ x;
  const float* __restrict__ w;     // packed [Cout][Kld]
  const float* __restrict__ bias;  // nullable
  float* __restrict__ y;
  float* __restrict__ stat;     // nullable: [m_tiles][Cout][2]
  float* __restrict__ partial;  // split-K: [splitk][M][Cout]
  int M;                        // rows of the output grid
  int Gd, Gh, Gw;               // output grid dims (rows decode as n,gd,gh,gw)
  int oDm, oHm, oWm;            // output memory dims
  int oSd, oSh, oSw, oOd, oOh, oOw;  // memory coord = g*oS + oO
  int out_ld, Cout;
  int linear_out;                // output address = row * out_ld (dense output grid): set by fill_fastdiv()
  int stat_ld;                  // channels per stat-partial row (= Cout of the whole convolution; this launch may cover a column segment)
  int Di, Hi, Wi, in_ld, Cin;   // input tensor
  int sD, sH, sW;               // in coord = g*s + off(tap)
  int nTd, nTh, nTw;            // taps per dim (tap index = (a_d*nTh + a_h)*nTw + a_w)
  int off0d, off0h, off0w;      // first tap offset per dim
  int offstep;                  // +1 forward, -1 dgrad
  int K, Kld, nchunks;
  int splitk, chunks_per_split;   // K split of the TAIL tiles (tile id >= full_tiles); splitk == 1: none
  int full_tiles;                 // tiles [0, full_tiles) run whole-K and write y directly
  int tail_row0;                  // first GEMM row of the tail tiles (partial buffer is [splitk][M - tail_row0][Cout])
  int m_tiles, n_tiles;
  const float* __restrict__ zero;  // >= 64 B of zeros (g_zero)
  unsigned x_bytes, w_bytes;       // extents for the buffer descriptors of the DMA path (< 4 GiB)
  unsigned y_bytes, partial_bytes; // ... and of the persistent kernels' output stores (0: output not addressable with 32 bits)
  FastDiv dR, dNtl;                // tail tiles (m_tiles * n_tiles - full_tiles), n_tiles
  FastDiv dGw, dGh, dGd, dCin, dTw, dTh;   // filled by fill_fastdiv() from Gw, Gh, Gd, Cin, nTw, nTh
  int adv_tap, adv_ci;                     // BK / Cin, BK % Cin
  int nbuf;                                // LDS tile buffers: 2, or 1 (see igemm_body)
  int bn_narrow;                           // 64 / 32: this launch runs on the 64- / 32-wide tile whatever its column count (narrow_bn()); 0: tile_bn(Cout)
  int bm;                                  // rows per tile: 128, or 256 (the tall 64-wide instance of the persistent kernel, tall_tiles()); 0 = 128
  int kmajor;                              // 1: channel-slice-major K order (k_slice_major)
  int skip_pad;                            // slice-major walk: skip the chunks of taps that are padding for the whole tile
  int tm_skip, cpt;                        // the same in the tap-major DMA walk when a chunk is one tap for all lanes; cpt = Cin / BK
  FastDiv dThw;                            // ... or, any other channel count (cpt == 0): a chunk is dead when every kernel DEPTH of the
                                           //     taps it touches is (tap / (nTh * nTw) = depth index)
  int Nb;                                  // samples (M = Nb * Gd * Gh * Gw)
  int dmajor;                              // GEMM rows enumerate (part, depth, sample in part, h, w) instead of (sample, depth, h, w): row_decode
  int Np;                                  // samples per part (dmajor)
  FastDiv dNp, dGdNp;
  int dm_dense;                            // dmajor over a dense output grid (forward, stride-1 dgrad): piecewise-linear epilogue
  FastDiv dP;                              // Gh * Gw
  FastDiv dNt;                             // taps per slice (kmajor)
};

inline void fill_fastdiv(IgemmParams& p);

// GEMM row -> (sample, output-grid depth) from q2 = row / (Gh * Gw).  Depth-major enumeration (slice-major kernels on layers with
// depth padding): the samples are cut into parts of Np; within a part all its samples' positions of ONE output frame are
// consecutive rows, so a 128-row tile lies (mostly) in one frame and the depth taps that are padding for that frame are padding
// for the whole tile (skipped, igemm_body) — n-major tiles straddle frames (C3D conv4: 196 rows per frame and sample, conv5: 49).
// The frames of a part are walked 1, 2, ..., Gd-1, 0: the frames with fewer taps come last.  Eight parts when the batch allows:
// rsp_xcd_remap hands each XCD one contiguous eighth of the tiles = one part, so every XCD runs its full-length tiles first and
// its short ones at the end, where they fill the ragged last round instead of leaving full-length tiles to it.
__device__ __forceinline__ void row_decode(const IgemmParams& p, bool dmajor, int q2, int& n, int& gd) {
  if (dmajor) {
    const int part = fastdiv(q2, p.dGdNp);
    const int rem = q2 - part * (p.Gd * p.Np);
    const int g = fastdiv(rem, p.dNp);
    n = part * p.Np + (rem - g * p.Np);
    gd = g + 1 == p.Gd ? 0 : g + 1;
  } else {
    n = fastdiv(q2, p.dGd);
    gd = q2 - n * p.Gd;
  }
}

// `bid` / `nblk`: this workgroup's index and the number of workgroups of ITS problem (blockIdx.x / gridDim.x for a single
// problem; offsets into a shared grid when several problems run in one launch, igemm_multi_kernel).
// KS: the channel-slice-major K walk of the DMA path (k_slice_major) with wave-uniform tap counters — its own instantiation so
// that neither variant carries the other's state (the 128x128 tile sits 2 VGPRs under the limit for three workgroups per CU).
template <int BM, int BN, int WAVES_M, int WAVES_N, int VEC, bool KS = false>
__device__ __forceinline__ void igemm_body(const IgemmParams& p, const int bid, const int nblk) {
  static_assert(!KS || VEC == 4, "slice-major walk: DMA path only");
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int AR = BM / 32, BR = BN / 32;  // rows staged per thread
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  static_assert(TM >= 1 && TN >= 1, "tile");

  // VEC==4: tiles are filled by LDS-DMA (global_load_lds_dwordx4), which writes 64 lanes x 16 B linearly, so rows are
  // unpadded (32 floats) and bank conflicts are avoided by an XOR swizzle of the 16-byte slot applied on the SOURCE
  // address and on the fragment read:  physical slot = logical slot ^ ((row >> 1) & 7)
  // (ds_read_b128 lane groups then hit 16 distinct 16-byte slots of the 256-byte bank row).
  // VEC==1 (scalar gather, Cin % 4 != 0): register staging into rows padded to 36 floats.
  constexpr bool DMA = VEC == 4;
  constexpr int LDR = DMA ? BK : LDK;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // nbuf == 1 (chosen per launch, `single_buffer()`): ONE tile buffer, re-filled after a barrier that follows the fragment
  // reads.  The fragments of a whole chunk live in registers anyway, so the second buffer only saved that barrier; giving it
  // up halves the LDS footprint and lets a third workgroup share the CU (VGPRs allow 3), which fills the matrix pipe better
  // (C3D conv2 +3 %) and makes the round 768 tiles wide (conv4: 784 tiles, was 1.53 rounds of 512: +5-8 %).
  const int nbuf = DMA ? p.nbuf : 2;
  float* As = reinterpret_cast<float*>(smem_raw);             // [nbuf][BM][LDR]
  float* Bs = As + nbuf * BM * LDR;                           // [nbuf][BN][LDR]
  int4* taptab = reinterpret_cast<int4*>(Bs + nbuf * BN * LDR);  // [ntaps] {offd, offh, offw, delta}
  long long* rowaddr = reinterpret_cast<long long*>(taptab + p.nTd * p.nTh * p.nTw + 1);  // [BM]

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);   // scalar: LDS-DMA bases (M0) become SALU arithmetic
  const int l32 = lane & 31, h = lane >> 5;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // Blocks [0, full_tiles) own one whole-K tile each.  The remaining tiles -- the part of the grid that would otherwise
  // run as a mostly empty last round on the 256 CUs -- are cut along K into `splitk` units each (unit u: tile u % R,
  // K-slice u / R, so neighbouring units share the weight slice in L2) and leave fp32 partials for splitk_reduce_kernel.
  int tile, z = 0;
  bool is_partial = false;
  if (bid < p.full_tiles) {
    tile = rsp_xcd_remap(bid, p.full_tiles);
  } else {
    const int R = p.m_tiles * p.n_tiles - p.full_tiles;
    const int u = rsp_xcd_remap(bid - p.full_tiles, nblk - p.full_tiles);
    z = u / R;
    tile = p.full_tiles + (u - z * R);
    is_partial = p.splitk > 1;
  }
  const int m_tile = tile / p.n_tiles, n_tile = tile - m_tile * p.n_tiles;
  const int m0 = m_tile * BM, n0 = n_tile * BN;
  const int kc_begin = is_partial ? z * p.chunks_per_split : 0;
  const int kc_end = is_partial ? min(p.nchunks, kc_begin + p.chunks_per_split) : p.nchunks;

  // ---- tap table --------------------------------------------------------------------------------------
  const int ntaps = p.nTd * p.nTh * p.nTw;
  for (int i = t; i < ntaps; i += 256) {
    const int q = fastdiv(i, p.dTw), aw = i - q * p.nTw;
    const int ad = fastdiv(q, p.dTh), ah = q - ad * p.nTh;
    const int od = p.off0d + ad * p.offstep, oh = p.off0h + ah * p.offstep, ow = p.off0w + aw * p.offstep;
    // DMA path: .x = the three validity bits this tap needs from a row's bit set (see rbits below); scalar path: the offsets
    taptab[i] = DMA ? make_int4((1 << ad) | (1 << (8 + ah)) | (1 << (16 + aw)), 0, 0, ((od * p.Hi + oh) * p.Wi + ow) * p.in_ld)
                    : make_int4(od, oh, ow, ((od * p.Hi + oh) * p.Wi + ow) * p.in_ld);
  }

  // ---- per-thread row geometry for the A (im2col) tile ----------------------------------------------------
  const int arow = t >> 3;        // + 32*i
  // this thread's k offset inside the chunk: linear for register staging, de-swizzled for the DMA path (the LDS
  // position is fixed by the lane, the data it must fetch is not)
  const int kcol = (DMA ? ((t & 7) ^ ((arow >> 1) & 7)) : (t & 7)) * 4;
  long long abase[AR];
  int aid[AR], aih[AR], aiw[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int r = m0 + arow + 32 * i;
    if (r < p.M) {
      const int q1 = fastdiv(r, p.dGw), gw = r - q1 * p.Gw;
      const int q2 = fastdiv(q1, p.dGh), gh = q1 - q2 * p.Gh;
      int n, gd;
      row_decode(p, DMA && p.dmajor, q2, n, gd);
      aid[i] = gd * p.sD;
      aih[i] = gh * p.sH;
      aiw[i] = gw * p.sW;
      abase[i] = ((((long long)n * p.Di + aid[i]) * p.Hi + aih[i]) * p.Wi + aiw[i]) * p.in_ld;
    } else {
      aid[i] = -(1 << 20);  // every tap out of bounds
      aih[i] = 0;
      aiw[i] = 0;
      abase[i] = 0;
    }
  }
  const float* wrow[BR];
  bool wok[BR];
#pragma unroll
  for (int i = 0; i < BR; ++i) {
    const int co = n0 + arow + 32 * i;
    wok[i] = co < p.Cout;
    wrow[i] = p.w + (long long)(wok[i] ? co : 0) * p.Kld;
  }
  floatx4 areg[AR], breg[BR];

  // (tap, ci) of this thread's k position(s) in the NEXT chunk to load, advanced incrementally (no division in the loop);
  // the tap-table entry is fetched one chunk ahead so its LDS latency hides behind the MFMAs.
  constexpr int NE = VEC == 4 ? 1 : 4;
  const int adv_tap = p.adv_tap, adv_ci = p.adv_ci;
  int ntap[NE], nci[NE];
  int4 ntt[NE];
  const float* const zero = p.zero;
  // element offset of the zero page relative to p.x, so an invalid lane only swaps an offset (pure v_cndmask, no branch)
  const long long zoff = (reinterpret_cast<const char*>(p.zero) - reinterpret_cast<const char*>(p.x)) / 4;

  typedef __attribute__((address_space(3))) void* lptr_t;
  // DMA path addressing: raw buffer descriptors (wave-uniform SGPRs) + 32-bit byte offsets.  An out-of-range offset makes
  // the hardware return zeros, so padding taps / tail rows / K tail cost one v_cndmask and no zero page or 64-bit math.
  __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, (int)p.w_bytes, 0x00020000);
  unsigned abase32[AR], wrowoff[BR], rowoff[AR], rbits[AR];
  bool rowok[AR];
  int last_tap = -1;
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    abase32[i] = (unsigned)(abase[i] * 4);
    rowoff[i] = 0;
    rowok[i] = false;
    // per-row validity, one bit per tap index and dimension (bits 0-7: d, 8-15: h, 16-23: w): whether a tap is inside the
    // input for this row is then two VALU ops when the lane's tap changes, instead of three range checks per row
    // (the taps of one dimension that fall inside the input form an interval [lo, hi]: with c the coordinate of tap 0 —
    //  mirrored for the descending taps of dgrad — tap a is inside iff 0 <= c + a < D)
    unsigned m = 0;
    if (DMA) {
      auto span = [&](int base, int D, int nT) -> unsigned {
        const int c = p.offstep > 0 ? base : D - 1 - base;
        const int lo = min(max(0, -c), 8), hi = min(nT - 1, D - 1 - c);
        return hi >= lo ? (2u << hi) - (1u << lo) : 0u;
      };
      m = span(aid[i] + p.off0d, p.Di, p.nTd) | (span(aih[i] + p.off0h, p.Hi, p.nTh) << 8) | (span(aiw[i] + p.off0w, p.Wi, p.nTw) << 16);
    }
    rbits[i] = m;
  }
#pragma unroll
  for (int i = 0; i < BR; ++i) wrowoff[i] = (unsigned)(n0 + arow + 32 * i) * (unsigned)p.Kld * 4u;

  // Taps that fall into the zero padding for EVERY row of this tile contribute nothing: the slice-major walk below skips their
  // chunks outright (no copies, no MFMAs).  With 128 consecutive output positions per tile this is mostly the depth taps of the
  // first / last frame — a 3x3x3 convolution over T frames spends 2/(3T) of its products there: C3D conv3 8 %, conv4 17 %,
  // conv5 33 %, R3D-18 layer4 (T = 1) 67 %.  tmask = OR of the rows' validity bits (rbits), through the (still unused) row-address
  // table and the barrier that publishes the tap table.
  const bool tms = DMA && !KS && p.tm_skip;      // tap-major walk with one tap per chunk for all lanes (Cin % BK == 0): same skipping
  unsigned tmask = 0x00ffffffu;
  if (KS || tms) {
    unsigned m = 0;
#pragma unroll
    for (int i = 0; i < AR; ++i) m |= rbits[i];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m |= (unsigned)__shfl_xor((int)m, o);
    if (lane == 0) reinterpret_cast<unsigned*>(rowaddr)[wave] = m;
  }
  __syncthreads();  // tap table (and the waves' masks) ready
  if (KS || tms) {
    const unsigned* wmk = reinterpret_cast<const unsigned*>(rowaddr);
    tmask = (unsigned)__builtin_amdgcn_readfirstlane((int)(wmk[0] | wmk[1] | wmk[2] | wmk[3]));
    if (KS && !p.skip_pad) tmask = 0x00ffffffu;
  }
#pragma unroll
  for (int e = 0; e < NE && !KS; ++e) {
    const int k = kc_begin * BK + kcol + e;
    if (p.kmajor) {   // chunk kc = (slice kc / ntaps, tap kc % ntaps); this lane's channel = slice*32 + its offset in the chunk
      const int sl = fastdiv(kc_begin, p.dNt);
      ntap[e] = kc_begin - sl * ntaps;
      nci[e] = sl * BK + kcol + e;
    } else {
      ntap[e] = fastdiv(k, p.dCin);
      nci[e] = k - ntap[e] * p.Cin;
    }
    ntt[e] = taptab[min(ntap[e], ntaps - 1)];
  }

  // Channel-slice-major K order on the DMA path: every chunk is ONE tap of ONE 32-channel slice, the same for all lanes, so
  // the walk (kw fastest, then kh, kd, then the next slice) is kept in wave-uniform counters — scalar arithmetic beside the
  // matrix pipe — and a row costs and + compare + add + select per chunk; no tap table, no per-lane tap tracking.
  int ukw = 0, ukh = 0, ukd = 0, uslice = 0;      // (tap-major skipping: uslice counts the chunks inside the current tap)
  if (tms) {
    const int tap0 = fastdiv(kc_begin * BK, p.dCin);
    uslice = kc_begin - tap0 * p.cpt;
    const int q = fastdiv(tap0, p.dTw);
    ukw = tap0 - q * p.nTw;
    ukd = fastdiv(q, p.dTh);
    ukh = q - ukd * p.nTh;
  }
  if (KS) {
    uslice = fastdiv(kc_begin, p.dNt);
    const int tap0 = kc_begin - uslice * ntaps;
    const int q = fastdiv(tap0, p.dTw);
    ukw = tap0 - q * p.nTw;
    ukd = fastdiv(q, p.dTh);
    ukh = q - ukd * p.nTh;
  }

  // KS cursor: (ckc; ukw, ukh, ukd, uslice) is the next chunk to load.  ks_step() moves it on by one chunk, ks_seek() past the
  // chunks whose tap is padding for the whole tile (scalar arithmetic only: no LDS access may follow an LDS-DMA issue).
  int ckc = kc_begin;
  auto ks_step = [&]() {
    ++ckc;
    if (++ukw == p.nTw) {
      ukw = 0;
      if (++ukh == p.nTh) {
        ukh = 0;
        if (++ukd == p.nTd) {
          ukd = 0;
          ++uslice;
        }
      }
    }
  };
  auto ks_seek = [&]() {
    while (ckc < kc_end) {
      const unsigned need = (1u << ukd) | (1u << (8 + ukh)) | (1u << (16 + ukw));
      if ((tmask & need) == need) break;
      ks_step();
    }
  };

  // per-lane (tap, channel) state of the tap-major walk: on by one chunk (the next tap-table entry is fetched here, i.e. before any
  // LDS-DMA of the chunk being loaded is issued)
  auto lane_advance = [&]() {
#pragma unroll
    for (int e = 0; e < NE && !KS; ++e) {
      if (p.kmajor) {
        if (++ntap[e] >= ntaps) {
          ntap[e] = 0;
          nci[e] += BK;
        }
      } else {
        nci[e] += adv_ci;
        ntap[e] += adv_tap;
        if (nci[e] >= p.Cin) {
          nci[e] -= p.Cin;
          ++ntap[e];
        }
      }
      ntt[e] = taptab[min(ntap[e], ntaps - 1)];
    }
  };
  // tap-major skipping: scalar tap counters beside the per-lane state
  auto tm_step = [&]() {
    ++ckc;
    if (p.cpt > 0 && ++uslice == p.cpt) {
      uslice = 0;
      if (++ukw == p.nTw) {
        ukw = 0;
        if (++ukh == p.nTh) {
          ukh = 0;
          ++ukd;
        }
      }
    }
  };

  auto load_chunk = [&](int kc, int buf) {
    const int k = kc * BK + kcol;
    if (KS) {
      const unsigned k4 = (unsigned)k * 4u;
#pragma unroll
      for (int i = 0; i < BR; ++i) {
        const unsigned off = wok[i] ? wrowoff[i] + k4 : 0xffffffffu;      // K = ntaps * Cin is a whole number of chunks: no K tail
        float* dst = Bs + buf * BN * LDR + i * 32 * LDR + wave * 64 * 4;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lptr_t)dst, 16, off, 0, 0, 0);
      }
      const unsigned need = (1u << ukd) | (1u << (8 + ukh)) | (1u << (16 + ukw));
      const int od = p.off0d + ukd * p.offstep, oh = p.off0h + ukh * p.offstep, ow = p.off0w + ukw * p.offstep;
      const unsigned delta4 = (unsigned)((((od * p.Hi + oh) * p.Wi + ow) * p.in_ld + uslice * BK) * 4) + (unsigned)kcol * 4u;
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        const unsigned off = (rbits[i] & need) == need ? abase32[i] + delta4 : 0xffffffffu;
        float* dst = As + buf * BM * LDR + i * 32 * LDR + wave * 64 * 4;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lptr_t)dst, 16, off, 0, 0, 0);
      }
      return;
    }
    // current chunk's tap state; then advance and prefetch the following chunk's tap entry BEFORE any DMA is issued
    int ctap[NE], cci[NE];
    int4 ctt[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      ctap[e] = ntap[e];
      cci[e] = nci[e];
      ctt[e] = ntt[e];
    }
    lane_advance();
    if (DMA) {
      const unsigned k4 = (unsigned)k * 4u;
      const bool kin = k < p.Kld;
#pragma unroll
      for (int i = 0; i < BR; ++i) {
        const unsigned off = (wok[i] && kin) ? wrowoff[i] + k4 : 0xffffffffu;
        float* dst = Bs + buf * BN * LDR + i * 32 * LDR + wave * 64 * 4;   // wave-uniform base; hardware adds lane*16 B
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lptr_t)dst, 16, off, 0, 0, 0);
      }
      if (ctap[0] != last_tap) {   // row validity / offsets change only when this lane's tap does (every Cin/32 chunks)
        last_tap = ctap[0];
        const int4 tt = ctt[0];
        const bool kok = ctap[0] < ntaps;
        const unsigned need = (unsigned)tt.x;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
          rowok[i] = kok & ((rbits[i] & need) == need);
          rowoff[i] = abase32[i] + (unsigned)tt.w * 4u;
        }
      }
      const unsigned ci4 = (unsigned)cci[0] * 4u;
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        const unsigned off = rowok[i] ? rowoff[i] + ci4 : 0xffffffffu;
        float* dst = As + buf * BM * LDR + i * 32 * LDR + wave * 64 * 4;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lptr_t)dst, 16, off, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < BR; ++i) {
        const float* src = (wok[i] && k < p.Kld) ? wrow[i] + k : zero;
        breg[i] = *reinterpret_cast<const floatx4*>(src);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int4 tt = ctt[e];
        const bool kok = ctap[e] < ntaps;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
          const int id = aid[i] + tt.x, ih = aih[i] + tt.y, iw = aiw[i] + tt.z;
          const bool ok = kok && (unsigned)id < (unsigned)p.Di && (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi;
          const long long real = abase[i] + tt.w + cci[e];
          const long long m = ok ? -1ll : 0ll;
          areg[i][e] = p.x[(real & m) | (zoff & ~m)];
        }
      }
    }
  };
  auto store_chunk = [&](int buf) {
    if (DMA) return;   // the DMA already landed the tile in LDS
    float* a = As + buf * BM * LDR;
    float* b = Bs + buf * BN * LDR;
#pragma unroll
    for (int i = 0; i < AR; ++i) *reinterpret_cast<floatx4*>(a + (arow + 32 * i) * LDR + kcol) = areg[i];
#pragma unroll
    for (int i = 0; i < BR; ++i) *reinterpret_cast<floatx4*>(b + (arow + 32 * i) * LDR + kcol) = breg[i];
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  bool have = kc_begin < kc_end;
  if (KS) {
    ks_seek();
    have = ckc < kc_end;
    if (have) {
      load_chunk(ckc, 0);
      ks_step();
      ks_seek();
    }
  } else if (tms) {
    auto dead = [&]() {
      if (p.cpt > 0) {
        const unsigned need = (1u << ukd) | (1u << (8 + ukh)) | (1u << (16 + ukw));
        return (tmask & need) != need;
      }
      const int t0 = fastdiv(ckc * BK, p.dCin), t1 = min(fastdiv(ckc * BK + BK - 1, p.dCin), ntaps - 1);      // taps the chunk touches
      const unsigned span = (2u << fastdiv(t1, p.dThw)) - (1u << fastdiv(t0, p.dThw));      // depth indices [d(t0), d(t1)]
      return (tmask & span) == 0;
    };
    while (ckc < kc_end && dead()) {
      lane_advance();
      tm_step();
    }
    have = ckc < kc_end;
    if (have) {
      load_chunk(ckc, 0);
      tm_step();
    }
  } else if (have) {
    load_chunk(kc_begin, 0);
    store_chunk(0);
  }
  __syncthreads();   // (with DMA in flight hipcc emits s_waitcnt vmcnt(0) before the barrier: the tile has landed)

  int buf = 0;
  for (int kc = kc_begin; (KS || tms) ? have : kc < kc_end; ++kc) {
    if (tms) {      // on to the next chunk with a live tap (LDS reads of the walk: no LDS-DMA is pending here)
      auto dead = [&]() {
      if (p.cpt > 0) {
        const unsigned need = (1u << ukd) | (1u << (8 + ukh)) | (1u << (16 + ukw));
        return (tmask & need) != need;
      }
      const int t0 = fastdiv(ckc * BK, p.dCin), t1 = min(fastdiv(ckc * BK + BK - 1, p.dCin), ntaps - 1);      // taps the chunk touches
      const unsigned span = (2u << fastdiv(t1, p.dThw)) - (1u << fastdiv(t0, p.dThw));      // depth indices [d(t0), d(t1)]
      return (tmask & span) == 0;
    };
      while (ckc < kc_end && dead()) {
        lane_advance();
        tm_step();
      }
    }
    const bool more = (KS || tms) ? ckc < kc_end : kc + 1 < kc_end;
    // 1. this chunk's operand fragments -> registers.  They are read BEFORE the next chunk's LDS-DMA is issued: hipcc
    //    orders every ds_read behind all pending LDS-DMA (s_waitcnt vmcnt(0)), so a read issued after the DMA would
    //    serialise the copy with the MFMAs instead of overlapping it.
    const float* a = As + buf * BM * LDR + (wm * WM + l32) * LDR;
    const float* b = Bs + buf * BN * LDR + (wn * WN + l32) * LDR;
    const int swz = DMA ? ((l32 >> 1) & 7) : 0;   // tile row bases are multiples of 32, so (row >> 1) & 7 == (l32 >> 1) & 7
    floatx4 af[4][TM], bf[4][TN];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int slot = ((h * 4 + kk) ^ swz) * 4;
#pragma unroll
      for (int i = 0; i < TM; ++i) af[kk][i] = *reinterpret_cast<const floatx4*>(a + i * 32 * LDR + slot);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[kk][j] = *reinterpret_cast<const floatx4*>(b + j * 32 * LDR + slot);
    }
    // 2. next chunk: LDS-DMA (or global loads into registers) in flight under this chunk's MFMAs
    if (nbuf == 1) __syncthreads();   // every wave holds its fragments: the buffer may be overwritten
    if (more) load_chunk((KS || tms) ? ckc : kc + 1, nbuf == 1 ? 0 : buf ^ 1);
    if (KS && more) {
      ks_step();
      ks_seek();
    }
    if (tms && more) tm_step();
    if (KS || tms) have = more;
    // 3. 16 k-steps x TM x TN MFMAs
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk][i][e], bf[kk][j][e], acc[i][j], 0, 0, 0);
    if (more) store_chunk(buf ^ 1);
    // keep the wait-for-DMA + barrier BEHIND the MFMAs (hipcc otherwise sinks the register-only MFMAs below it, which
    // exposes the copy latency instead of hiding it under the matrix pipe)
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    buf = nbuf == 1 ? 0 : buf ^ 1;
  }

  // ---- epilogue -------------------------------------------------------------------------------------------
  // GEMM row -> output address.  Forward outputs, stride-1 input gradients and the K-split partials are LINEAR in the row
  // (address = base + row * pitch): plain arithmetic.  The strided parity classes of dgrad go through a per-tile table
  // of the general affine mapping.
  const bool linear = is_partial || p.linear_out;   // uniform
  // Depth-major rows over a dense output: a tile's rows are consecutive positions of one frame of one sample, then of the next
  // sample (or frame) — two linear runs unless a frame is shorter than the tile.  base A up to row `split`, base B from there on.
  long long pwA = 0, pwD = 0;      // base of the first run; what the second run adds to it
  int pw_split = 0;
  bool piecewise = false;
  if (DMA && !linear && p.dm_dense) {
    const int P = p.Gh * p.Gw;
    const int q2a = fastdiv(m0, p.dP), f0 = m0 - q2a * P;
    pw_split = __builtin_amdgcn_readfirstlane(P - f0);
    if (BM <= pw_split + P) {
      piecewise = true;
      int n, gd;
      row_decode(p, true, q2a, n, gd);
      const int ra = (n * p.oDm + gd) * P + f0;                               // output position (< 2^31) of row m0
      row_decode(p, true, min(q2a + 1, p.Nb * p.Gd - 1), n, gd);
      const int rb = (n * p.oDm + gd) * P - pw_split;                         // ... of row m0 + split, minus split
      pwA = (long long)__builtin_amdgcn_readfirstlane(ra) * p.out_ld;
      pwD = (long long)__builtin_amdgcn_readfirstlane(rb - ra) * p.out_ld;
    }
  }
  if (!linear && !piecewise) {
    if (t < BM) {
      const int r = m0 + t;
      long long addr = -1;
      if (r < p.M) {
        const int q1 = fastdiv(r, p.dGw), gw = r - q1 * p.Gw;
        const int q2 = fastdiv(q1, p.dGh), gh = q1 - q2 * p.Gh;
        int n, gd;
        row_decode(p, DMA && p.dmajor, q2, n, gd);
        addr = ((((long long)n * p.oDm + gd * p.oSd + p.oOd) * p.oHm + gh * p.oSh + p.oOh) * p.oWm + gw * p.oSw +
                p.oOw) * p.out_ld;
      }
      rowaddr[t] = addr;
    }
    __syncthreads();
  }

  float* dst = is_partial ? p.partial : p.y;
  const long long lin_ld = is_partial ? p.Cout : p.out_ld;
  const long long lin_base = is_partial ? ((long long)z * (p.M - p.tail_row0) - p.tail_row0) * p.Cout : 0ll;
  // A tile that lies wholly inside the M rows (all but the last row of tiles) stores without per-row checks: straight-line
  // code instead of one LDS round trip + branch in front of every store.
  auto store_tile = [&](auto checked, auto lin) {
    constexpr bool CHK = decltype(checked)::value, LIN = decltype(lin)::value;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * WN + j * 32 + l32;
      if (col < p.Cout) {
        const float bv = (p.bias && !is_partial) ? p.bias[col] : 0.f;
        if (LIN && piecewise) {      // two linear runs (depth-major rows)
          float* laneA = dst + pwA + (long long)(wm * WM + h * 4) * p.out_ld + col;
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int rl = wm * WM + i * 32 + (e >> 2) * 8 + h * 4 + (e & 3);
              if (!CHK || m0 + rl < p.M)
                laneA[(long long)(i * 32 + (e >> 2) * 8 + (e & 3)) * p.out_ld + (rl < pw_split ? 0ll : pwD)] = acc[i][j][e] + bv;
            }
          continue;
        }
        float* lane0 = dst + lin_base + (long long)(m0 + wm * WM + h * 4) * lin_ld + col;   // LIN: row (i, e) = lane0 + const * pitch
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int rl = wm * WM + i * 32 + (e >> 2) * 8 + h * 4 + (e & 3);
            if (LIN) {
              if (!CHK || m0 + rl < p.M) lane0[(long long)(i * 32 + (e >> 2) * 8 + (e & 3)) * lin_ld] = acc[i][j][e] + bv;
            } else {
              const long long addr = rowaddr[rl];
              if (!CHK || addr >= 0) dst[addr + col] = acc[i][j][e] + bv;
            }
          }
      }
    }
  };
  const bool whole = m0 + BM <= p.M;
  if (linear || piecewise) {
    if (whole) store_tile(std::false_type{}, std::true_type{});
    else store_tile(std::true_type{}, std::true_type{});
  } else {
    if (whole) store_tile(std::false_type{}, std::false_type{});
    else store_tile(std::true_type{}, std::false_type{});
  }

  if (p.stat && !is_partial) {
    // per-channel (sum, sumsq) of the bias-free conv output per 128-row block; rows >= M contributed zeros.
    constexpr int SB = BM / 128;             // stat blocks per tile
    constexpr int WPB = WAVES_M / SB;        // waves (along M) per stat block
    static_assert(SB >= 1 && WPB >= 1 && WPB * SB == WAVES_M, "stat blocks");
    float* red = As;  // [WAVES_M][BN][2]  (all LDS reads of the main loop are done: barrier above)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float s = 0.f, ss = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float v = acc[i][j][e];
          s += v;
          ss = fmaf(v, v, ss);
        }
      s += __shfl_xor(s, 32);
      ss += __shfl_xor(ss, 32);
      if (h == 0) {
        const int c = wn * WN + j * 32 + l32;
        red[(wm * BN + c) * 2 + 0] = s;
        red[(wm * BN + c) * 2 + 1] = ss;
      }
    }
    __syncthreads();
    for (int idx = t; idx < SB * BN; idx += 256) {
      const int sb = idx / BN, c = idx - sb * BN;
      if (n0 + c < p.Cout && (long long)(m_tile * SB + sb) * 128 < p.M) {
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int w = 0; w < WPB; ++w) {
          s += red[((sb * WPB + w) * BN + c) * 2 + 0];
          ss += red[((sb * WPB + w) * BN + c) * 2 + 1];
        }
        float* o = p.stat + ((long long)(m_tile * SB + sb) * p.stat_ld + n0 + c) * 2;
        o[0] = s;
        o[1] = ss;
      }
    }
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int VEC, int MINW = 2>
__global__ __launch_bounds__(256, MINW) void igemm_kernel(const IgemmParams p) {
  igemm_body<BM, BN, WAVES_M, WAVES_N, VEC>(p, (int)blockIdx.x, (int)gridDim.x);
}
// the same tile with the channel-slice-major K walk (LDS-DMA path)
template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256, 2) void igemm_ks_kernel(const IgemmParams p) {
  igemm_body<BM, BN, WAVES_M, WAVES_N, 4, true>(p, (int)blockIdx.x, (int)gridDim.x);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Persistent variant of the LDS-DMA implicit GEMM (round 4).  Same tiles, same K walks and the same arithmetic per tile as
// igemm_body; what changes is who pays the per-tile fixed cost.  Measured with per-wave timestamps (tools/phase_probe.py,
// profiles/r04/phase_*.txt): a short-K tile (K = 288 ... 1152) spends 15-30 us of a 70-250 us life between its launch and its
// first MFMA (kernel arguments, ~1 100 VALU + 700 SALU instructions of geometry — a third of them reloads of spilled scalars —,
// the tap table, the first copy's round trip) and another 8-10 us storing (64-bit address arithmetic per store, more reloads).
// Here a workgroup stays resident and walks a list of units (tile, K slice):
//   * launch constants — tap table, buffer descriptors, lane geometry — are set up once per workgroup;
//   * the NEXT unit's row geometry is computed and its first chunk is copied to LDS under the LAST chunk's MFMAs of the current
//     one, so a unit starts with its operands in LDS;
//   * the epilogue addresses its stores as raw-buffer offsets: 4 VGPR offsets per lane + scalar row-group offsets + an immediate
//     column offset, no per-store address arithmetic; rows / columns outside the tensor get an out-of-range offset, which the
//     hardware drops (no checked / unchecked code variants).
// ---------------------------------------------------------------------------------------------------------------------------
struct UnitPos {
  int m0, n0, m_tile, z, kc_begin, kc_end;
  bool is_partial;
};

// FOLD > 0: two-level summation over K (round 6).  An MFMA tile accumulates all K products of an output in ONE fp32 chain, whose
// rounding error grows with sqrt(K) (1.4e-6 relative at K = 13 824; the CPU reference's convolution sums K in panels: 3.4e-7 at every
// K — DESIGN.md section 2).  With FOLD, the running accumulators are added into a second set every FOLD chunks (FOLD * 32 products per
// panel) and cleared: two short chains instead of one long one, at the price of 64 more VGPRs (two waves per SIMD instead of three).
template <int BM, int BN, int WAVES_M, int WAVES_N, bool KS, int FOLD = 0>
__device__ __forceinline__ void igemm_persist(const IgemmParams& p, const int wg, const int nwg, const int n_units) {
  // HALF (BN = 32 * TN + 16, four waves stacked along M): the last 16 columns of the tile are two 16x16 blocks per wave on
  // v_mfma_f32_16x16x4_f32 — the same 64 flops per cycle and SIMD as the 32x32x2 form, so a 144-column segment (R(2+1)D's mid
  // channels) costs 4.5 column blocks of matrix-pipe time instead of the 160-wide tile's 5.  The B tile in LDS keeps whole 32-row
  // groups (BNL rows; the loader's rows beyond the segment read as zeros), the tile's N extent (unit_pos, statistics) is BN.
  constexpr bool HALF = BN % 32 == 16;
  constexpr int BNL = HALF ? BN + 16 : BN;
  constexpr int WM = BM / WAVES_M, WN = BNL / WAVES_N;
  constexpr int TM = WM / 32, TN = (BN / WAVES_N) / 32;
  constexpr int AR = BM / 32, BR = BNL / 32;
  static_assert(WAVES_M * WAVES_N == 4 && TM >= 1 && TN >= 1, "tile");
  static_assert(!HALF || (WAVES_N == 1 && TM == 1), "half column block: waves stacked along M, one row block each");
  constexpr int LDR = BK;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int nbuf = p.nbuf;
  const int ntaps = p.nTd * p.nTh * p.nTw;
  float* As = reinterpret_cast<float*>(smem_raw);                 // [nbuf][BM][LDR]
  float* Bs = As + nbuf * BM * LDR;                               // [nbuf][BNL][LDR]
  int2* taptab = reinterpret_cast<int2*>(Bs + nbuf * BNL * LDR);   // [ntaps + 1] {validity bits the tap needs, input offset}
  unsigned* rowaddr = reinterpret_cast<unsigned*>(taptab + ntaps + 1);   // [BM] byte offsets of the strided-output rows
  float* red = reinterpret_cast<float*>(rowaddr + BM);            // [WAVES_M][BNL][2] statistics scratch
  unsigned* xch = reinterpret_cast<unsigned*>(red + WAVES_M * BNL * 2);  // [4] the waves' tap masks
  unsigned char* pblk = reinterpret_cast<unsigned char*>(xch + 4);       // copy of the argument block (see below)

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l32 = lane & 31, h = lane >> 5;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  typedef __attribute__((address_space(3))) void* lptr_t;
  // Launch constants that only the per-unit code needs (row decoding, walk start, epilogue: ~80 scalars) are read from a copy of
  // the argument block in LDS where they are used, instead of living in — and being spilled from — the scalar registers across
  // the K loop (the compiler cannot carry an LDS value over a barrier, so nothing stays resident).
  {
    const int* src = reinterpret_cast<const int*>(&p);
    int* dstw = reinterpret_cast<int*>(pblk);
    for (int i = t; i < (int)(sizeof(IgemmParams) / 4); i += 256) dstw[i] = src[i];
  }
  const IgemmParams& vp = *reinterpret_cast<const IgemmParams*>(pblk);
  auto fdv = [](int n, const FastDiv& f) { return fastdiv(n, f); };
  auto row_decode_v = [&](const IgemmParams& v, bool dmajor, int q2, int& n, int& gd) { row_decode(v, dmajor, q2, n, gd); };
  // ---- once per workgroup ---------------------------------------------------------------------------------------------------
  for (int i = t; i < ntaps && !KS; i += 256) {
    const int q = fastdiv(i, p.dTw), aw = i - q * p.nTw;
    const int ad = fastdiv(q, p.dTh), ah = q - ad * p.nTh;
    const int od = p.off0d + ad * p.offstep, oh = p.off0h + ah * p.offstep, ow = p.off0w + aw * p.offstep;
    taptab[i] = make_int2((1 << ad) | (1 << (8 + ah)) | (1 << (16 + aw)), ((od * p.Hi + oh) * p.Wi + ow) * p.in_ld);
  }
  __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, (int)p.w_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)p.y_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsrc_p = __builtin_amdgcn_make_buffer_rsrc(p.partial ? p.partial : p.y, 0, (int)(p.partial ? p.partial_bytes : 4u), 0x00020000);
  const int arow = t >> 3;
  const int kcol = ((t & 7) ^ ((arow >> 1) & 7)) * 4;      // de-swizzled k offset of this thread inside a chunk
  const bool tms = !KS && p.tm_skip;
  const int R = p.m_tiles * p.n_tiles - p.full_tiles;

  auto unit_pos = [&](int u) {
    UnitPos q;
    int tile;
    q.z = 0;
    q.is_partial = false;
    if (u < vp.full_tiles) {
      tile = rsp_xcd_remap(u, vp.full_tiles);
    } else {
      const int uu = rsp_xcd_remap(u - vp.full_tiles, n_units - vp.full_tiles);
      q.z = fdv(uu, vp.dR);
      tile = vp.full_tiles + (uu - q.z * R);
      q.is_partial = vp.splitk > 1;
    }
    q.m_tile = fdv(tile, vp.dNtl);
    const int n_tile = tile - q.m_tile * vp.n_tiles;
    q.m0 = q.m_tile * BM;
    q.n0 = n_tile * BN;
    q.kc_begin = q.is_partial ? q.z * vp.chunks_per_split : 0;
    q.kc_end = q.is_partial ? min(vp.nchunks, q.kc_begin + vp.chunks_per_split) : vp.nchunks;
    // (values read from the LDS copy are wave-uniform but live in vector registers: back to scalars)
    q.m0 = __builtin_amdgcn_readfirstlane(q.m0);
    q.n0 = __builtin_amdgcn_readfirstlane(q.n0);
    q.m_tile = __builtin_amdgcn_readfirstlane(q.m_tile);
    q.z = __builtin_amdgcn_readfirstlane(q.z);
    q.kc_begin = __builtin_amdgcn_readfirstlane(q.kc_begin);
    q.kc_end = __builtin_amdgcn_readfirstlane(q.kc_end);
    q.is_partial = __builtin_amdgcn_readfirstlane((int)q.is_partial) != 0;
    return q;
  };

  // ---- per-unit state of the loader (always that of the unit whose chunks are being COPIED) --------------------------------
  unsigned abase32[AR], rbits[AR], wrowoff[BR];      // (a weight row beyond Cout carries an offset >= 2^31: out of range for any chunk)
  unsigned tmask = 0x00ffffffu;
  int ld_end = 0;                                        // kc_end of the loader's unit
  int ckc = 0;                                           // next chunk to copy
  int ukw = 0, ukh = 0, ukd = 0, uslice = 0;             // wave-uniform tap counters (KS; tap-major skipping)
  int ntap = 0, nci = 0;                                 // per-lane (tap, channel) of the tap-major walk
  int2 ntt = make_int2(0, 0);

  // rows of the unit -> lane geometry + this wave's OR of the validity bits (to xch)
  auto rows_of = [&](const UnitPos& q) {
    // (launch constants from the LDS copy, made scalar again: the vector registers are needed for the four rows' arithmetic)
    auto S = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    auto SF = [&](const FastDiv& f) {
      FastDiv r;
      r.mul = (unsigned)S((int)f.mul);
      r.shr = (unsigned)S((int)f.shr);
      r.d = 0;
      return r;
    };
    const int M = S(vp.M), Gw = S(vp.Gw), Gh = S(vp.Gh), Gd = S(vp.Gd), Np = S(vp.Np), dmaj = S(vp.dmajor);
    const int sD = S(vp.sD), sH = S(vp.sH), sW = S(vp.sW), Di = S(vp.Di), Hi = S(vp.Hi), Wi = S(vp.Wi), in_ld = S(vp.in_ld);
    const int o0d = S(vp.off0d), o0h = S(vp.off0h), o0w = S(vp.off0w), ostep = S(vp.offstep);
    const int nTd = S(vp.nTd), nTh = S(vp.nTh), nTw = S(vp.nTw);
    const FastDiv dGw = SF(vp.dGw), dGh = SF(vp.dGh), dGd = SF(vp.dGd), dNp = SF(vp.dNp), dGdNp = SF(vp.dGdNp);
    unsigned m_or = 0;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const int r = q.m0 + arow + 32 * i;
      unsigned m = 0, ab = 0;
      if (r < M) {
        const int q1 = fastdiv(r, dGw), gw = r - q1 * Gw;
        const int q2 = fastdiv(q1, dGh), gh = q1 - q2 * Gh;
        int n, gd;
        if (dmaj) {
          const int part = fastdiv(q2, dGdNp);
          const int rem = q2 - part * (Gd * Np);
          const int g = fastdiv(rem, dNp);
          n = part * Np + (rem - g * Np);
          gd = g + 1 == Gd ? 0 : g + 1;
        } else {
          n = fastdiv(q2, dGd);
          gd = q2 - n * Gd;
        }
        const int id = gd * sD, ih = gh * sH, iw = gw * sW;
        ab = (unsigned)(((((long long)n * Di + id) * Hi + ih) * Wi + iw) * in_ld * 4);
        auto span = [&](int base, int D, int nT) -> unsigned {
          const int c = ostep > 0 ? base : D - 1 - base;
          const int lo = min(max(0, -c), 8), hi = min(nT - 1, D - 1 - c);
          return hi >= lo ? (2u << hi) - (1u << lo) : 0u;
        };
        m = span(id + o0d, Di, nTd) | (span(ih + o0h, Hi, nTh) << 8) | (span(iw + o0w, Wi, nTw) << 16);
      }
      abase32[i] = ab;
      rbits[i] = m;
      m_or |= m;
    }
    const int Cout = S(vp.Cout), Kld = S(vp.Kld);
#pragma unroll
    for (int i = 0; i < BR; ++i) {
      const int co = q.n0 + arow + 32 * i;
      wrowoff[i] = co < Cout ? (unsigned)co * (unsigned)Kld * 4u : 0x80000000u;
    }
    if (KS || tms) {
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) m_or |= (unsigned)__shfl_xor((int)m_or, o);
      if (lane == 0) xch[wave] = m_or;
    }
  };
  // ... then, behind a barrier: the tile's tap mask and the walk's start position
  auto dead_tm = [&]() {      // tap-major skipping: is chunk ckc dead for the whole tile?
    if (p.cpt > 0) {
      const unsigned need = (1u << ukd) | (1u << (8 + ukh)) | (1u << (16 + ukw));
      return (tmask & need) != need;
    }
    const int t0 = fastdiv(ckc * BK, p.dCin), t1 = min(fastdiv(ckc * BK + BK - 1, p.dCin), ntaps - 1);
    const unsigned sp = (2u << fastdiv(t1, p.dThw)) - (1u << fastdiv(t0, p.dThw));
    return (tmask & sp) == 0;
  };
  auto lane_advance = [&]() {      // (the slice-major order always takes the KS instance: tap-major arithmetic only)
    nci += p.adv_ci;
    ntap += p.adv_tap;
    if (nci >= p.Cin) {
      nci -= p.Cin;
      ++ntap;
    }
    ntt = taptab[min(ntap, ntaps - 1)];
  };
  auto ks_step = [&]() {
    ++ckc;
    if (++ukw == p.nTw) {
      ukw = 0;
      if (++ukh == p.nTh) {
        ukh = 0;
        if (++ukd == p.nTd) {
          ukd = 0;
          ++uslice;
        }
      }
    }
  };
  auto tm_step = [&]() {
    ++ckc;
    if (p.cpt > 0 && ++uslice == p.cpt) {
      uslice = 0;
      if (++ukw == p.nTw) {
        ukw = 0;
        if (++ukh == p.nTh) {
          ukh = 0;
          ++ukd;
        }
      }
    }
  };
  // move the cursor to the next LIVE chunk of the loader's unit (no-op for the plain tap-major walk)
  auto seek = [&]() {
    if (KS) {
      while (ckc < ld_end) {
        const unsigned need = (1u << ukd) | (1u << (8 + ukh)) | (1u << (16 + ukw));
        if ((tmask & need) == need) break;
        ks_step();
      }
    } else if (tms) {
      while (ckc < ld_end && dead_tm()) {
        lane_advance();
        tm_step();
      }
    }
  };
  // start of a unit's walk, part 1 (before the barrier that publishes the waves' masks): cursor and tap state
  auto walk_pre = [&](const UnitPos& q) {
    ld_end = q.kc_end;
    ckc = q.kc_begin;
    if (KS) {
      uslice = fdv(q.kc_begin, vp.dNt);
      const int tap0 = q.kc_begin - uslice * ntaps;
      const int qq = fdv(tap0, vp.dTw);
      ukw = tap0 - qq * vp.nTw;
      ukd = fdv(qq, vp.dTh);
      ukh = qq - ukd * vp.nTh;
      uslice = __builtin_amdgcn_readfirstlane(uslice);
      ukw = __builtin_amdgcn_readfirstlane(ukw);
      ukd = __builtin_amdgcn_readfirstlane(ukd);
      ukh = __builtin_amdgcn_readfirstlane(ukh);
    } else {
      const int k = q.kc_begin * BK + kcol;
      ntap = fdv(k, vp.dCin);
      nci = k - ntap * vp.Cin;
      ntt = taptab[min(ntap, ntaps - 1)];
      if (tms) {
        const int tap0 = fdv(q.kc_begin * BK, vp.dCin);
        uslice = q.kc_begin - tap0 * vp.cpt;
        const int qq = fdv(tap0, vp.dTw);
        ukw = tap0 - qq * vp.nTw;
        ukd = fdv(qq, vp.dTh);
        ukh = qq - ukd * vp.nTh;
        uslice = __builtin_amdgcn_readfirstlane(uslice);
        ukw = __builtin_amdgcn_readfirstlane(ukw);
        ukd = __builtin_amdgcn_readfirstlane(ukd);
        ukh = __builtin_amdgcn_readfirstlane(ukh);
      }
    }
  };
  // ... part 2 (behind it): the tile's tap mask; on to the first live chunk
  auto walk_mask = [&]() {
    tmask = 0x00ffffffu;
    if ((KS && p.skip_pad) || tms) tmask = (unsigned)__builtin_amdgcn_readfirstlane((int)(xch[0] | xch[1] | xch[2] | xch[3]));
    seek();
  };
  // copy chunk ckc of the loader's unit into tile buffer `buf` and move the cursor on to the next live chunk.  (Every LDS read this
  // needs — the next tap-table entry — is issued before the first copy.)
  auto load_chunk = [&](int buf) {
    const int k = ckc * BK + kcol;
    const unsigned k4 = (unsigned)k * 4u;
    if (KS) {
#pragma unroll
      for (int i = 0; i < BR; ++i) {
        const unsigned off = wrowoff[i] + k4;
        float* dst = Bs + buf * BNL * LDR + i * 32 * LDR + wave * 64 * 4;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lptr_t)dst, 16, off, 0, 0, 0);
      }
      const unsigned need = (1u << ukd) | (1u << (8 + ukh)) | (1u << (16 + ukw));
      const int od = p.off0d + ukd * p.offstep, oh = p.off0h + ukh * p.offstep, ow = p.off0w + ukw * p.offstep;
      const unsigned delta4 = (unsigned)((((od * p.Hi + oh) * p.Wi + ow) * p.in_ld + uslice * BK) * 4) + (unsigned)kcol * 4u;
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        const unsigned off = (rbits[i] & need) == need ? abase32[i] + delta4 : 0xffffffffu;
        float* dst = As + buf * BM * LDR + i * 32 * LDR + wave * 64 * 4;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lptr_t)dst, 16, off, 0, 0, 0);
      }
      ks_step();
      seek();
      return;
    }
    const int ctap = ntap, cci = nci;
    const int2 ctt = ntt;
    lane_advance();
    const unsigned kout = k < p.Kld ? 0u : 0x80000000u;      // K tail: out of range
#pragma unroll
    for (int i = 0; i < BR; ++i) {
      const unsigned off = (wrowoff[i] + k4) | kout;
      float* dst = Bs + buf * BNL * LDR + i * 32 * LDR + wave * 64 * 4;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lptr_t)dst, 16, off, 0, 0, 0);
    }
    const unsigned need = ctap < ntaps ? (unsigned)ctt.x : 0xffffffffu;      // (a lane past the last tap: never satisfied)
    const unsigned delta4 = ((unsigned)ctt.y + (unsigned)cci) * 4u;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const unsigned off = (rbits[i] & need) == need ? abase32[i] + delta4 : 0xffffffffu;
      float* dst = As + buf * BM * LDR + i * 32 * LDR + wave * 64 * 4;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lptr_t)dst, 16, off, 0, 0, 0);
    }
    if (tms) {
      tm_step();
      seek();
    } else {
      ++ckc;
    }
  };

  floatx16 acc[TM][TN];
  floatx16 acc2[FOLD > 0 ? TM : 1][FOLD > 0 ? TN : 1];      // FOLD: the panels' sum
  int fold_n = 0;                                           // chunks in the running panel
  static_assert(FOLD == 0 || !HALF, "two-level summation: 32-wide column blocks only");
  floatx4 acch[2];      // HALF: rows 16*blk + 4*(lane / 16) + r of the wave's 32, column 32*TN + lane % 16
  auto fold_panel = [&]() {      // running accumulators -> panel sum
    if (FOLD > 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            acc2[i][j][e] += acc[i][j][e];
            acc[i][j][e] = 0.f;
          }
      fold_n = 0;
    }
  };
  auto zero_acc = [&]() {
    if (FOLD > 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc2[i][j][e] = 0.f;
      fold_n = 0;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#pragma unroll
    for (int bl = 0; bl < 2; ++bl)
#pragma unroll
      for (int e = 0; e < 4; ++e) acch[bl][e] = 0.f;
  };

  // ---- epilogue of one unit ---------------------------------------------------------------------------------------------------
  auto epilogue = [&](const UnitPos& q) {
    const bool linear = q.is_partial || vp.linear_out;
    const int lin_ld = q.is_partial ? vp.Cout : vp.out_ld;
    // byte offset of (local row 0, column 0 of the tile) for the linear forms; the piecewise form (depth-major rows over a dense
    // output) has base A up to local row pw_split and base A + D from there on
    unsigned base4 = 0, pwD4 = 0;
    int pw_split = BM;
    bool piecewise = false;
    if (linear) {
      const long long b = q.is_partial ? ((long long)q.z * (vp.M - vp.tail_row0) + (q.m0 - vp.tail_row0)) * vp.Cout : (long long)q.m0 * vp.out_ld;
      base4 = (unsigned)(b * 4);
    } else if (vp.dm_dense) {
      const int P = vp.Gh * vp.Gw;
      const int q2a = fdv(q.m0, vp.dP), f0 = q.m0 - q2a * P;
      pw_split = P - f0;
      if (BM <= pw_split + P) {
        piecewise = true;
        int n, gd;
        row_decode(p, true, q2a, n, gd);
        const int ra = (n * vp.oDm + gd) * P + f0;
        row_decode(p, true, min(q2a + 1, vp.Nb * vp.Gd - 1), n, gd);
        const int rb = (n * vp.oDm + gd) * P - pw_split;
        base4 = (unsigned)ra * (unsigned)vp.out_ld * 4u;
        pwD4 = (unsigned)(rb - ra) * (unsigned)vp.out_ld * 4u;
      }
    }
    const bool table = !linear && !piecewise;
    if (table) {
      if (t < BM) {
        const int r = q.m0 + t;
        unsigned addr = 0xffffffffu;
        if (r < vp.M) {
          const int q1 = fdv(r, vp.dGw), gw = r - q1 * vp.Gw;
          const int q2 = fdv(q1, vp.dGh), gh = q1 - q2 * vp.Gh;
          int n, gd;
          row_decode(p, vp.dmajor, q2, n, gd);
          addr = (unsigned)(((((long long)n * vp.oDm + gd * vp.oSd + vp.oOd) * vp.oHm + gh * vp.oSh + vp.oOh) * vp.oWm + gw * vp.oSw + vp.oOw) *
                            vp.out_ld * 4);
        }
        rowaddr[t] = addr;
      }
      __syncthreads();
    }
    const unsigned pitch4 = (unsigned)lin_ld * 4u;
    __amdgpu_buffer_rsrc_t rs = q.is_partial ? rsrc_p : rsrc_y;
    // rows of this lane: local row = wm*WM + h*4 + (group offset gb + k); rows at or beyond `lim` lie outside the M rows
    const int lim = vp.M - q.m0 - (wm * WM + h * 4);
    const int pw_lim = pw_split - (wm * WM + h * 4);      // piecewise: local rows from pw_split on belong to the second run
    // The bias of every column block is loaded — and waited for, once — before the first store.  Loaded block by block between the
    // stores, each load's `s_waitcnt vmcnt(0)` also waited for the acknowledgement of the 32 stores in front of it (loads and
    // stores share the counter): one drain per column block, four per tile in the 160-wide instance.
    float bvs[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = q.n0 + wn * WN + j * 32 + l32;
      bvs[j] = (vp.bias && !q.is_partial && col < vp.Cout) ? vp.bias[col] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(bvs[j]));
    const int r16 = lane & 15, g16 = lane >> 4;
    const int colh = q.n0 + TN * 32 + r16;      // HALF: this lane's column of the half block
    const bool cokh = HALF && colh < vp.Cout;
    float bvh = 0.f;
    if (HALF) {
      bvh = (vp.bias && !q.is_partial && cokh) ? vp.bias[colh] : 0.f;
      asm volatile("" : "+v"(bvh));
    }
    auto store_tile = [&](auto mode_tag) {
      constexpr int MODE = decltype(mode_tag)::value;      // 0 linear, 1 piecewise (two linear runs), 2 per-row table
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = q.n0 + wn * WN + j * 32 + l32;
        const bool cok = col < vp.Cout;
        const float bv = bvs[j];
        // offset of (local row wm*WM + h*4 + k, this column), k = 0..3; the row groups (i, g4) add a SCALAR offset
        unsigned vo[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) vo[k] = base4 + (unsigned)(wm * WM + h * 4 + k) * pitch4 + (unsigned)col * 4u;
        const int lim_c = cok ? lim : 0;      // a column beyond Cout: no row of it is stored
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const int gb = i * 32 + g4 * 8;
            const unsigned so = (unsigned)gb * pitch4;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float v = acc[i][j][g4 * 4 + k] + bv;
              if (MODE == 2) {
                const unsigned a = rowaddr[wm * WM + gb + h * 4 + k];
                const unsigned off = (cok && a != 0xffffffffu) ? a + (unsigned)col * 4u : 0x80000000u;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, off, 0, 0);
              } else {
                // (the hardware range-checks the VECTOR offset alone: in the piecewise form, where the second run's base may lie
                //  below the first one's, the whole offset goes into it; the linear form keeps the row group in the scalar offset)
                unsigned off = vo[k];
                if (MODE == 1) off += so + (gb + k < pw_lim ? 0u : pwD4);
                if (gb + k >= lim_c) off = 0x80000000u;      // outside the tensor: out of range (extents stay below 2 GiB), dropped by the hardware
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, off, MODE == 1 ? 0u : so, 0);
              }
            }
          }
      }
      if (HALF) {
#pragma unroll
        for (int bl = 0; bl < 2; ++bl)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int lrow = wm * WM + bl * 16 + g16 * 4 + k;      // local row of the tile
            const float v = acch[bl][k] + bvh;
            unsigned off;
            if (MODE == 2) {
              const unsigned a = rowaddr[lrow];
              off = (cokh && a != 0xffffffffu) ? a + (unsigned)colh * 4u : 0x80000000u;
            } else {
              off = base4 + (unsigned)lrow * pitch4 + (unsigned)colh * 4u;
              if (MODE == 1) off += lrow < pw_split ? 0u : pwD4;
              if (!cokh || q.m0 + lrow >= vp.M) off = 0x80000000u;
            }
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, off, 0, 0);
          }
      }
    };
    // statistics first: their barrier then waits for the (already landed) prefetch only, not for the tile's 64 stores per wave
    if (vp.stat && !q.is_partial) {
      constexpr int SB = BM / 128;
      constexpr int WPB = WAVES_M / SB;
      static_assert(SB >= 1 && WPB >= 1 && WPB * SB == WAVES_M, "stat blocks");
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const float v = acc[i][j][e];
            s += v;
            ss = fmaf(v, v, ss);
          }
        s += __shfl_xor(s, 32);
        ss += __shfl_xor(ss, 32);
        if (h == 0) {
          const int c = wn * WN + j * 32 + l32;
          red[(wm * BNL + c) * 2 + 0] = s;
          red[(wm * BNL + c) * 2 + 1] = ss;
        }
      }
      if (HALF) {
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int bl = 0; bl < 2; ++bl)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = acch[bl][e];
            s += v;
            ss = fmaf(v, v, ss);
          }
        s += __shfl_xor(s, 16);
        ss += __shfl_xor(ss, 16);
        s += __shfl_xor(s, 32);
        ss += __shfl_xor(ss, 32);
        if (g16 == 0) {
          red[(wm * BNL + TN * 32 + r16) * 2 + 0] = s;
          red[(wm * BNL + TN * 32 + r16) * 2 + 1] = ss;
        }
      }
      __syncthreads();
      for (int idx = t; idx < SB * BN; idx += 256) {
        const int sb = idx / BN, c = idx - sb * BN;
        if (q.n0 + c < vp.Cout && (long long)(q.m_tile * SB + sb) * 128 < vp.M) {
          float s = 0.f, ss = 0.f;
#pragma unroll
          for (int w = 0; w < WPB; ++w) {
            s += red[((sb * WPB + w) * BNL + c) * 2 + 0];
            ss += red[((sb * WPB + w) * BNL + c) * 2 + 1];
          }
          float* o = vp.stat + ((long long)(q.m_tile * SB + sb) * vp.stat_ld + q.n0 + c) * 2;
          o[0] = s;
          o[1] = ss;
        }
      }
      // (the next unit's statistics go through `red` again: its writers are behind that unit's K-loop barriers)
    }
    if (table) store_tile(std::integral_constant<int, 2>{});
    else if (piecewise) store_tile(std::integral_constant<int, 1>{});
    else store_tile(std::integral_constant<int, 0>{});
  };

  // ---- the unit loop ------------------------------------------------------------------------------------------------------------
  // Per unit: a tight inner loop over all chunks but the last (fragments -> registers, next chunk's copy, MFMAs, barrier: the loop
  // of igemm_body), then the BOUNDARY step, which computes the last chunk while the loader moves on to the next unit — its row
  // geometry is decoded before the fragment reads (the finished unit's loader registers are free by then), its first live chunk is
  // copied under this step's MFMAs — and the finished unit's epilogue behind the closing barrier.  A unit without a live chunk
  // (the very start; a K slice whose chunks are all padding) runs the boundary step without fragments or MFMAs.
  // HALF: operands of the 16x16x4 blocks.  Lane (r = lane % 16, g = lane / 16) holds the 16-byte slots 2g, 2g + 1 of rows r and
  // 16 + r of the wave's A rows and of row 32*TN + r of B: k-step (hh, e) multiplies the four k = 4 * (2g + hh) + e, g = 0..3.
  // (The chunk's 32 k are summed in another order than by the 32x32x2 blocks; A and B agree, which is all a block needs.)
  struct HalfFrag {
    floatx4 a[2][2], b[2];
  };
  auto read_frags = [&](int buf, floatx4 (&af)[4][TM], floatx4 (&bf)[4][TN], HalfFrag& hf) {
    const float* a = As + buf * BM * LDR + (wm * WM + l32) * LDR;
    const float* b = Bs + buf * BNL * LDR + (wn * WN + l32) * LDR;
    const int swz = (l32 >> 1) & 7;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int slot = ((h * 4 + kk) ^ swz) * 4;
#pragma unroll
      for (int i = 0; i < TM; ++i) af[kk][i] = *reinterpret_cast<const floatx4*>(a + i * 32 * LDR + slot);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[kk][j] = *reinterpret_cast<const floatx4*>(b + j * 32 * LDR + slot);
    }
    if (HALF) {
      const int r16 = lane & 15, g16 = lane >> 4;
      const int swh = (r16 >> 1) & 7;      // (rows 16 + r and 32*TN + r swizzle like row r)
      const float* ah = As + buf * BM * LDR + (wm * WM + r16) * LDR;
      const float* bh = Bs + buf * BNL * LDR + (TN * 32 + r16) * LDR;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int slot = ((2 * g16 + hh) ^ swh) * 4;
        hf.a[0][hh] = *reinterpret_cast<const floatx4*>(ah + slot);
        hf.a[1][hh] = *reinterpret_cast<const floatx4*>(ah + 16 * LDR + slot);
        hf.b[hh] = *reinterpret_cast<const floatx4*>(bh + slot);
      }
    }
  };
  auto mfmas = [&](const floatx4 (&af)[4][TM], const floatx4 (&bf)[4][TN], const HalfFrag& hf) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk][i][e], bf[kk][j][e], acc[i][j], 0, 0, 0);
        if (HALF && (kk & 1) == 0) {      // the half block's 8 k-steps, spread over the chunk's 16
          const int hh = kk >> 1;
#pragma unroll
          for (int bl = 0; bl < 2; ++bl)
            acch[bl] = __builtin_amdgcn_mfma_f32_16x16x4f32(hf.a[bl][hh][e], hf.b[hh][e], acch[bl], 0, 0, 0);
        }
      }
  };
  zero_acc();
  int u = wg - nwg;                      // the unit being computed (none yet)
  UnitPos cur = unit_pos(0);
  bool cur_valid = false, have = false;
  int buf = 0;
  __syncthreads();                       // tap table, argument block
  for (;;) {
    // ---- all chunks of the current unit but the last
    while (have && ckc < ld_end) {
      floatx4 af[4][TM], bf[4][TN];
      HalfFrag hf;
      read_frags(buf, af, bf, hf);
      const int nb = nbuf == 1 ? 0 : buf ^ 1;
      if (nbuf == 1) __syncthreads();    // every wave holds its fragments: the buffer may be overwritten
      load_chunk(nb);
      mfmas(af, bf, hf);
      if (FOLD > 0 && ++fold_n == FOLD) fold_panel();
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      buf = nb;
    }
    // ---- boundary step
    const bool has_next = u + nwg < n_units;
    UnitPos nxt = cur;
    if (has_next) {
      nxt = unit_pos(u + nwg);
      rows_of(nxt);
      walk_pre(nxt);
    }
    bool next_have = false;
    const int nb = nbuf == 1 ? 0 : buf ^ 1;
    if (have) {
      floatx4 af[4][TM], bf[4][TN];
      HalfFrag hf;
      read_frags(buf, af, bf, hf);
      __syncthreads();                   // fragments held by every wave (single buffer); the waves' masks
      if (has_next) {
        walk_mask();
        next_have = ckc < ld_end;
        if (next_have) load_chunk(nb);
      }
      mfmas(af, bf, hf);
      if (FOLD > 0) {      // the unit's last chunk: the tile is the sum of its panels
        fold_panel();
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = acc2[i][j];
      }
    } else {
      __syncthreads();
      if (has_next) {
        walk_mask();
        next_have = ckc < ld_end;
        if (next_have) load_chunk(nb);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    buf = nb;
    if (cur_valid) epilogue(cur);
    if (!has_next) break;
    u += nwg;
    cur = nxt;
    cur_valid = true;
    have = next_have;
    zero_acc();
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool KS, int MINW>
__global__ __launch_bounds__(256, MINW) void igemm_persist_kernel(const IgemmParams p, const int n_units) {
  igemm_persist<BM, BN, WAVES_M, WAVES_N, KS>(p, (int)blockIdx.x, (int)gridDim.x, n_units);
}
// ... with two-level summation over K (see igemm_persist, FOLD): two waves per SIMD
template <int BM, int BN, int WAVES_M, int WAVES_N, bool KS, int FOLD>
__global__ __launch_bounds__(256, 2) void igemm_persist_fold_kernel(const IgemmParams p, const int n_units) {
  igemm_persist<BM, BN, WAVES_M, WAVES_N, KS, FOLD>(p, (int)blockIdx.x, (int)gridDim.x, n_units);
}

// Several independent problems of one tile shape in a single launch: the stride-parity classes of a strided convolution's
// input gradient (8 for stride (2,2,2)) are small GEMMs — R3D-18's layer4.0: 8 x (512 rows x 256 columns) — that each left
// most of the machine idle as launches of their own (0.32 ms for 3.6 GFLOP).  Workgroup b belongs to class c with
// start[c] <= b < start[c+1] and runs that class's tile b - start[c]; no K split in this mode.
constexpr int MAX_MULTI = 8;
struct IgemmMulti {
  int n;
  int start[MAX_MULTI + 1];
  IgemmParams p[MAX_MULTI];
};
static_assert(sizeof(IgemmMulti) <= 4096, "kernel argument segment");

template <int BM, int BN, int WAVES_M, int WAVES_N, int VEC, int MINW = 2>
__global__ __launch_bounds__(256, MINW) void igemm_multi_kernel(const IgemmMulti m) {
  int c = 0;
  while (c + 1 < m.n && (int)blockIdx.x >= m.start[c + 1]) ++c;
  igemm_body<BM, BN, WAVES_M, WAVES_N, VEC>(m.p[c], (int)blockIdx.x - m.start[c], m.start[c + 1] - m.start[c]);
}


// ---------------------------------------------------------------------------------------------------------------------------
// Direct implicit GEMM for SMALL launches (round 5; measured, then superseded by narrow_tiles(): see direct_applies): 128 rows x 64
// columns per workgroup, operands straight from L2 into MFMA fragments — no LDS tiles, no tile pipeline to fill and drain.
//
// Why.  The LDS-DMA tile kernels above are built for launches of many rounds: a 128 x 128 tile lives 45-70 us even at K = 64-288
// (15 us before its first MFMA, 8 us storing: tools/phase_probe.py), and a launch of a few dozen to a few hundred tiles — the
// 14 x 14 / 7 x 7 stages of S3D-G and R3D-18 / R(2+1)D, every 1x1x1 stride-2 shortcut — is one ragged round of that life:
// 15-56 TFLOP/s where the big layers run at 105-140 (profiles/r04).  Here a wave owns 32 rows x 64 columns (two 32x32 accumulators)
// and per 32-deep K chunk each lane loads 64 contiguous bytes of ITS row of the im2col matrix (the half-wave h takes k 16h..16h+15 of
// the chunk) and 64 bytes of each of its two weight rows — whole 128-byte lines per row and chunk, so nothing relies on L1 reuse —
// one chunk ahead of the 32 MFMAs that consume them.  ~130 VGPRs: three waves per SIMD cover the L2 latency.  The K order is the
// packed weights' (tap-major, or channel-slice-major: fill_fastdiv), the row geometry the general affine one of IgemmParams, so
// forward convolutions and every stride-parity class of an input gradient run through it; a K split over blockIdx (unit = tile +
// tiles * slice) feeds the existing fixed-order splitk_reduce_vec_kernel when the tiles alone cannot fill the machine.
// Needs 16-byte rows with Cin % 16 == 0 (a 16-deep half-chunk never straddles a tap).
// ---------------------------------------------------------------------------------------------------------------------------
struct DirectFrag {
  floatx4 a[4];
  floatx4 b[2][4];
};

__global__ __launch_bounds__(256, 3) void igemm_direct_kernel(const IgemmParams p) {
  __shared__ long long rowaddr[128];
  __shared__ float red[4][64][2];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l32 = lane & 31, h = lane >> 5;
  const int tiles = p.m_tiles * p.n_tiles;
  const int z = (int)blockIdx.x / tiles;                      // K slice (slice-major units: neighbours share the A rows in L2)
  const int tile = (int)blockIdx.x - z * tiles;
  const int m_tile = tile / p.n_tiles, n_tile = tile - m_tile * p.n_tiles;
  const int m0 = m_tile * 128, n0 = n_tile * 64;
  const bool is_partial = p.splitk > 1;
  const int kc_begin = z * p.chunks_per_split;
  const int kc_end = min(p.nchunks, kc_begin + p.chunks_per_split);
  const int ntaps = p.nTd * p.nTh * p.nTw;

  // this lane's row of the im2col matrix
  const int r = m0 + wave * 32 + l32;
  const bool rv = r < p.M;
  int n = 0, gd = 0, gh = 0, gw = 0;
  {
    const int rr = rv ? r : 0;
    const int q1 = fastdiv(rr, p.dGw);
    gw = rr - q1 * p.Gw;
    const int q2 = fastdiv(q1, p.dGh);
    gh = q1 - q2 * p.Gh;
    row_decode(p, false, q2, n, gd);
  }
  const int id0 = gd * p.sD + p.off0d, ih0 = gh * p.sH + p.off0h, iw0 = gw * p.sW + p.off0w;
  const long long xn = (long long)n * p.Di;
  const float* wrow[2];
  bool cv[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + j * 32 + l32;
    cv[j] = col < p.Cout;
    wrow[j] = p.w + (long long)(cv[j] ? col : 0) * p.Kld + 16 * h;
  }

  auto load = [&](int kc, DirectFrag& f) {
    const int kpos = kc * BK + 16 * h;                         // this half-wave's 16 k of the chunk, in packed order
    int tap, ci;
    if (p.kmajor) {
      const int sl = fastdiv(kc, p.dNt);
      tap = kc - sl * ntaps;
      ci = sl * BK + 16 * h;
    } else {
      tap = fastdiv(kpos, p.dCin);
      ci = kpos - tap * p.Cin;
    }
    const bool kin = kpos < p.K;
    const int q = fastdiv(tap, p.dTw), aw = tap - q * p.nTw;
    const int ad = fastdiv(q, p.dTh), ah = q - ad * p.nTh;
    const int id = id0 + ad * p.offstep, ih = ih0 + ah * p.offstep, iw = iw0 + aw * p.offstep;
    const bool ok = rv && kin && (unsigned)id < (unsigned)p.Di && (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi;
    const floatx4* ap = reinterpret_cast<const floatx4*>(p.x + (((xn + id) * p.Hi + ih) * p.Wi + iw) * p.in_ld + ci);
    const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) f.a[i] = ok ? ap[i] : zero;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const floatx4* bp = reinterpret_cast<const floatx4*>(wrow[j] + (long long)kc * BK);
#pragma unroll
      for (int i = 0; i < 4; ++i) f.b[j][i] = (cv[j] && kin) ? bp[i] : zero;
    }
  };

  floatx16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  auto mma = [&](const DirectFrag& f) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][e], f.b[j][i][e], acc[j], 0, 0, 0);
  };

  DirectFrag f0, f1;
  if (kc_begin < kc_end) load(kc_begin, f0);
  for (int kc = kc_begin; kc < kc_end; kc += 2) {
    if (kc + 1 < kc_end) load(kc + 1, f1);
    mma(f0);
    if (kc + 1 < kc_end) {
      if (kc + 2 < kc_end) load(kc + 2, f0);
      mma(f1);
    }
  }

  // ---- epilogue: rows of the accumulators -> addresses (general affine mapping through a per-tile table) ---------------------
  if (!is_partial && !p.linear_out) {
    if (t < 128) {
      const int rr = m0 + t;
      long long addr = -1;
      if (rr < p.M) {
        const int q1 = fastdiv(rr, p.dGw), w_ = rr - q1 * p.Gw;
        const int q2 = fastdiv(q1, p.dGh), h_ = q1 - q2 * p.Gh;
        int n_, d_;
        row_decode(p, false, q2, n_, d_);
        addr = ((((long long)n_ * p.oDm + d_ * p.oSd + p.oOd) * p.oHm + h_ * p.oSh + p.oOh) * p.oWm + w_ * p.oSw + p.oOw) * p.out_ld;
      }
      rowaddr[t] = addr;
    }
    __syncthreads();
  }
  float* dst = is_partial ? p.partial + (long long)z * p.M * p.Cout : p.y;
  const long long ld = is_partial ? p.Cout : p.out_ld;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + j * 32 + l32;
    if (col < p.Cout) {
      const float bv = (p.bias && !is_partial) ? p.bias[col] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int rl = wave * 32 + (e >> 2) * 8 + h * 4 + (e & 3);
        if (m0 + rl < p.M) {
          const long long addr = (is_partial || p.linear_out) ? (long long)(m0 + rl) * ld : rowaddr[rl];
          dst[addr + col] = acc[j][e] + bv;
        }
      }
    }
  }
  if (p.stat && !is_partial) {
    // per-channel (sum, sumsq) of the bias-free output over this tile's 128 rows (rows >= M hold zeros): fixed order
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float sm = 0.f, ss = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float v = acc[j][e];
        sm += v;
        ss = fmaf(v, v, ss);
      }
      sm += __shfl_xor(sm, 32);
      ss += __shfl_xor(ss, 32);
      if (h == 0) {
        red[wave][j * 32 + l32][0] = sm;
        red[wave][j * 32 + l32][1] = ss;
      }
    }
    __syncthreads();
    if (t < 64 && n0 + t < p.Cout) {
      float* o = p.stat + ((long long)m_tile * p.stat_ld + n0 + t) * 2;
      o[0] = red[0][t][0] + red[1][t][0] + red[2][t][0] + red[3][t][0];
      o[1] = red[0][t][1] + red[1][t][1] + red[2][t][1] + red[3][t][1];
    }
  }
}

// split-K reduction: y[row] = sum_z partial[z][row] + bias, stats per 128-row tile.
struct ReduceParams {
  const float* __restrict__ partial;
  const float* __restrict__ bias;
  float* __restrict__ y;
  float* __restrict__ stat;
  int M, Cout, splitk, stat_ld;
  int row0;   // first row covered by the partials ([splitk][M - row0][Cout]); a multiple of 128
  int Gd, Gh, Gw, oDm, oHm, oWm, oSd, oSh, oSw, oOd, oOh, oOw, out_ld;
  int Nb, dmajor, Np;   // depth-major row enumeration (row_decode)
};

__device__ __forceinline__ long long reduce_row_addr(const ReduceParams& p, int r) {
  const int gw = r % p.Gw;
  int q = r / p.Gw;
  const int gh = q % p.Gh;
  q /= p.Gh;
  int gd, n;
  if (p.dmajor) {
    const int part = q / (p.Gd * p.Np), rem = q - part * (p.Gd * p.Np);
    const int g = rem / p.Np;
    n = part * p.Np + (rem - g * p.Np);
    gd = g + 1 == p.Gd ? 0 : g + 1;
  } else {
    gd = q % p.Gd;
    n = q / p.Gd;
  }
  return ((((long long)n * p.oDm + gd * p.oSd + p.oOd) * p.oHm + gh * p.oSh + p.oOh) * p.oWm + gw * p.oSw + p.oOw) * p.out_ld;
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ReduceParams p) {
  // block = one 128-row tile x 64 channels; thread = (row group of 4 lanes.., channel)
  __shared__ float red[4][64][2];
  const int m_tile = p.row0 / 128 + blockIdx.x, c0 = blockIdx.y * 64;
  const int c = c0 + (threadIdx.x & 63);
  const int rg = threadIdx.x >> 6;  // 0..3 -> rows rg, rg+4, ...
  float s = 0.f, ss = 0.f;
  if (c < p.Cout) {
    const float bv = p.bias ? p.bias[c] : 0.f;
    for (int rl = rg; rl < 128; rl += 4) {
      const int r = m_tile * 128 + rl;
      if (r >= p.M) break;
      float v = 0.f;
      for (int z = 0; z < p.splitk; ++z) v += p.partial[((long long)z * (p.M - p.row0) + (r - p.row0)) * p.Cout + c];
      const long long addr = reduce_row_addr(p, r);
      p.y[addr + c] = v + bv;
      s += v;
      ss = fmaf(v, v, ss);
    }
  }
  if (p.stat) {
    red[rg][threadIdx.x & 63][0] = s;
    red[rg][threadIdx.x & 63][1] = ss;
    __syncthreads();
    if (rg == 0 && c < p.Cout) {
      float a = 0.f, b = 0.f;
      for (int w = 0; w < 4; ++w) {
        a += red[w][threadIdx.x][0];
        b += red[w][threadIdx.x][1];
      }
      p.stat[((long long)m_tile * p.stat_ld + c) * 2 + 0] = a;
      p.stat[((long long)m_tile * p.stat_ld + c) * 2 + 1] = b;
    }
  }
}

// Vector variant (Cout % 4 == 0, out_ld % 4 == 0, y 16-byte aligned): thread = (4 channels, 8 rows); the 8 rows' loads are
// independent, so each thread keeps 8 x 16 B in flight per K-slice instead of one dependent 4-byte load at a time.
__device__ __forceinline__ void splitk_reduce_vec_body(const ReduceParams& p, const int bx, const int by) {
  __shared__ floatx4 red[2][16][16];
  const int m_tile = p.row0 / 128 + bx, c0 = by * 64;
  const int q4 = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = c0 + 4 * q4;
  const bool cok = c < p.Cout;
  const long long slab = (long long)(p.M - p.row0) * p.Cout;
  floatx4 v[8];
  long long addr[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = m_tile * 128 + rl + 16 * i;
    v[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    addr[i] = -1;
    if (r < p.M && cok) addr[i] = reduce_row_addr(p, r);
  }
  // K-slices are summed four at a time, ((z0 + z1) + (z2 + z3)) onto the running sum: 32 independent 16-byte loads in flight per
  // thread instead of 8 (the small late layers run S = 8..16 slices on a few dozen workgroups, where each dependent round of
  // loads is an exposed L2 round trip: 27 us -> ~10 us per reduce on R3D-18's layer 3/4 shapes); the order is fixed.
  const float* pbase[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) pbase[i] = p.partial + (long long)(m_tile * 128 + rl + 16 * i - p.row0) * p.Cout + c;
  int z = 0;
  for (; z + 4 <= p.splitk; z += 4) {
    floatx4 t[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i)
        t[u][i] = addr[i] >= 0 ? *reinterpret_cast<const floatx4*>(pbase[i] + (z + u) * slab) : floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += (t[0][i] + t[1][i]) + (t[2][i] + t[3][i]);
  }
  for (; z < p.splitk; ++z) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (addr[i] >= 0) v[i] += *reinterpret_cast<const floatx4*>(pbase[i] + z * slab);
  }
  floatx4 bv = floatx4{0.f, 0.f, 0.f, 0.f};
  if (p.bias && cok) bv = *reinterpret_cast<const floatx4*>(p.bias + c);
  floatx4 s = floatx4{0.f, 0.f, 0.f, 0.f}, ss = s;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (addr[i] >= 0) {
      *reinterpret_cast<floatx4*>(p.y + addr[i] + c) = v[i] + bv;
      s += v[i];
      ss += v[i] * v[i];
    }
  }
  if (p.stat) {
    red[0][rl][q4] = s;
    red[1][rl][q4] = ss;
    __syncthreads();
    if (threadIdx.x < 128) {
      const int which = threadIdx.x >> 6, cc = threadIdx.x & 63;   // 0: sum, 1: sum of squares
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < 16; ++w) a += red[which][w][cc >> 2][cc & 3];
      if (c0 + cc < p.Cout) p.stat[((long long)m_tile * p.stat_ld + c0 + cc) * 2 + which] = a;
    }
  }
}

__global__ __launch_bounds__(256) void splitk_reduce_vec_kernel(const ReduceParams p) {
  splitk_reduce_vec_body(p, (int)blockIdx.x, (int)blockIdx.y);
}

// ... of several problems in one launch (the K-split stride-parity classes of a few-tile input gradient, dgrad_run): blockIdx.z
// picks the problem; problems without a K split wrote their output themselves
struct ReduceMulti {
  int n;
  ReduceParams r[MAX_MULTI];
};
__global__ __launch_bounds__(256) void splitk_reduce_vec_multi_kernel(const ReduceMulti m) {
  const ReduceParams& p = m.r[blockIdx.z];
  if (p.splitk <= 1 || (long long)blockIdx.x * 128 >= p.M - p.row0 || (int)blockIdx.y * 64 >= p.Cout) return;
  splitk_reduce_vec_body(p, (int)blockIdx.x, (int)blockIdx.y);
}

// Weight re-pack: out[o][tapidx*C + c] (row pitch Kld, zero padded) from the reference layout (Cout,Cin,kT,kH,kW).
struct PackParams {
  const float* __restrict__ w;
  float* __restrict__ out;
  int Cout, Cin, kT, kH, kW;   // dims of the SOURCE tensor (reference layout); O / C below may be larger: zero padding
  int transpose;  // 0: o=co,c=ci (forward)   1: o=ci,c=co (dgrad)
  int O, C, Kld;
  int nTd, nTh, nTw;     // taps enumerated per dim
  int k0d, k0h, k0w;     // first kernel index per dim
  int kstepd, ksteph, kstepw;
};

__global__ void pack_weight_kernel(const PackParams p) {
  const long long total = (long long)p.O * p.Kld;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int o = (int)(i / p.Kld);
    const int k = (int)(i - (long long)o * p.Kld);
    float v = 0.f;
    const int ntaps = p.nTd * p.nTh * p.nTw;
    if (k < ntaps * p.C) {
      int tap, c;
      if (k_slice_major(p.C, ntaps)) {
        const int sl = k / (ntaps * 32), r = k - sl * (ntaps * 32);
        tap = r >> 5;
        c = sl * 32 + (r & 31);
      } else {
        tap = k / p.C;
        c = k - tap * p.C;
      }
      const int aw = tap % p.nTw, q = tap / p.nTw;
      const int ah = q % p.nTh, ad = q / p.nTh;
      const int kt = p.k0d + ad * p.kstepd, kh = p.k0h + ah * p.ksteph, kw = p.k0w + aw * p.kstepw;
      const int co = p.transpose ? c : o, ci = p.transpose ? o : c;
      if (co < p.Cout && ci < p.Cin) v = p.w[((((long long)co * p.Cin + ci) * p.kT + kt) * p.kH + kh) * p.kW + kw];
    }
    p.out[i] = v;
  }
}

// Many re-packs in one launch (blockIdx.y = job): a step re-packs every convolution weight of an encoder once (forward
// layout, and for the query encoder the dgrad layouts) — as single launches that was 80-270 four-microsecond kernels per step.
// blockIdx.x = packed row o (stem layout: a 4096-element slice); grid.x is sized for the job with the most rows, blocks past a
// job's end exit at once.  One workgroup walks its row tap by tap (tap decode is scalar), threads over the channel index: no
// per-element division (the single-conv kernel above spends ~150 integer instructions per element on index arithmetic).
__global__ __launch_bounds__(256) void pack_batch_kernel(const rsp_pack_job* __restrict__ jobs) {
  const rsp_pack_job j = jobs[blockIdx.y];
  const float* __restrict__ w = reinterpret_cast<const float*>(j.src);
  float* __restrict__ out = reinterpret_cast<float*>(j.dst);
  if (j.kind == 1) {   // stem layout [tap][64][4] (conv_stem.hip)
    const long long i0 = (long long)blockIdx.x * 4096;
    const long long i1 = i0 + 4096 < j.total ? i0 + 4096 : j.total;
    for (long long i = i0 + threadIdx.x; i < i1; i += 256) {
      const int ci = (int)(i & 3), n = (int)((i >> 2) & 63), tap = (int)(i >> 8);
      float v = 0.f;
      if (tap < j.ntaps && n < j.Cout_src && ci < j.Cin_src) {
        const int kw = tap % j.kW, r = tap / j.kW;
        const int kh = r % j.kH, kt = r / j.kH;
        v = w[((((long long)n * j.Cin_src + ci) * j.kT + kt) * j.kH + kh) * j.kW + kw];
      }
      out[i] = v;
    }
    return;
  }
  const int o = blockIdx.x;
  if (o >= j.O) return;
  // Whole filters (all T taps of one (co, ci) pair are T contiguous floats of the source) are staged through LDS, so every
  // source line is read once, by neighbouring lanes, and the packed row is written in order: tile[c][t] -> row[tap*C + c].
  __shared__ float tile[4096 + 64];
  __shared__ int tsrc_of[MAX_TAPS];
  float* __restrict__ row = out + (long long)o * j.Kld;
  for (int tap = threadIdx.x; tap < j.ntaps; tap += 256) {
    const int aw = tap % j.nTw, q = tap / j.nTw;
    const int ah = q % j.nTh, ad = q / j.nTh;
    tsrc_of[tap] = ((j.k0d + ad * j.kstepd) * j.kH + (j.k0h + ah * j.ksteph)) * j.kW + (j.k0w + aw * j.kstepw);
  }
  const int T = j.kT * j.kH * j.kW;
  const bool kmaj = k_slice_major(j.C, j.ntaps);
  const int CH = T >= 4096 ? 1 : (4096 / T > 128 ? 128 : 4096 / T);
  const bool o_in = o < (j.transpose ? j.Cin_src : j.Cout_src);
  const int c_in = j.transpose ? j.Cout_src : j.Cin_src;
  // filter of channel pair (o, c): forward o = co, c = ci -> consecutive c are consecutive filters; dgrad o = ci, c = co
  const long long fbase = j.transpose ? (long long)o * T : (long long)o * j.Cin_src * T;
  const long long fstride = j.transpose ? (long long)j.Cin_src * T : (long long)T;
  for (int c0 = 0; c0 < j.C; c0 += CH) {
    const int nc = min(CH, j.C - c0);
    __syncthreads();
    // (indices advance by 256 elements per trip: one division per thread and chunk instead of two per element — the kernel was
    //  bound by its index arithmetic at 2.3 TB/s)
    const int nel = nc * T;
    if (!j.transpose) {      // the filters of consecutive c follow each other in the source: one contiguous run
      const float* __restrict__ src = w + fbase + (long long)c0 * T;
      const int lim = o_in ? min(nel, max(0, c_in - c0) * T) : 0;      // elements of real (un-padded) channels
      for (int e = threadIdx.x; e < nel; e += 256) tile[e] = e < lim ? src[e] : 0.f;
    } else {
      int cl = (int)threadIdx.x / T, t = (int)threadIdx.x - cl * T;
      const int dcl = 256 / T, dt = 256 - dcl * T;
      for (int e = threadIdx.x; e < nel; e += 256) {
        const int c = c0 + cl;
        tile[e] = (o_in && c < c_in) ? w[fbase + c * fstride + t] : 0.f;
        cl += dcl;
        t += dt;
        if (t >= T) {
          t -= T;
          ++cl;
        }
      }
    }
    __syncthreads();
    {
      int tap = (int)threadIdx.x / nc, cl = (int)threadIdx.x - tap * nc;
      const int dtap = 256 / nc, dcl = 256 - dtap * nc;
      for (int e = threadIdx.x; e < j.ntaps * nc; e += 256) {
        row[k_index(kmaj, tap, c0 + cl, j.C, j.ntaps)] = tile[cl * T + tsrc_of[tap]];
        tap += dtap;
        cl += dcl;
        if (cl >= nc) {
          cl -= nc;
          ++tap;
        }
      }
    }
  }
  for (int k = j.ntaps * j.C + threadIdx.x; k < j.Kld; k += 256) row[k] = 0.f;
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
int launch_ks_cfg(const IgemmParams& p, hipStream_t s) {
  const size_t lds = (size_t)p.nbuf * (BM + BN) * BK * sizeof(float) + (size_t)(p.nTd * p.nTh * p.nTw + 1) * sizeof(int4) + BM * sizeof(long long);
  static bool attr_set = false;
  if (!attr_set) {
    const size_t lds_max = (size_t)2 * (BM + BN) * BK * sizeof(float) + (MAX_TAPS + 1) * sizeof(int4) + BM * sizeof(long long);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_ks_kernel<BM, BN, WAVES_M, WAVES_N>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
    attr_set = true;
  }
  dim3 grid(p.full_tiles + (p.m_tiles * p.n_tiles - p.full_tiles) * p.splitk);
  rsp_note_kernel("igemm_ks_kernel<%d, %d, %d, %d>", BM, BN, WAVES_M, WAVES_N);
  hipLaunchKernelGGL((igemm_ks_kernel<BM, BN, WAVES_M, WAVES_N>), grid, dim3(256), lds, s, p);
  return rsp_check_launch("igemm_ks_kernel");
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int VEC, int MINW = 2>
int launch_cfg(const IgemmParams& p, hipStream_t s) {
  // LDS: two tile buffers + tap table (sized by the actual tap count: short-K layers are latency-bound and want a
  // third workgroup per CU) + output-row address table
  const size_t nb = VEC == 4 ? (size_t)p.nbuf : 2;
  const size_t lds = nb * (BM + BN) * (VEC == 4 ? BK : LDK) * sizeof(float) +
                     (size_t)(p.nTd * p.nTh * p.nTw + 1) * sizeof(int4) + BM * sizeof(long long);
  static bool attr_set = false;
  if (!attr_set) {
    const size_t lds_max = (size_t)2 * (BM + BN) * (VEC == 4 ? BK : LDK) * sizeof(float) + (MAX_TAPS + 1) * sizeof(int4) +
                           BM * sizeof(long long);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<BM, BN, WAVES_M, WAVES_N, VEC, MINW>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
    attr_set = true;
  }
  dim3 grid(p.full_tiles + (p.m_tiles * p.n_tiles - p.full_tiles) * p.splitk);
  rsp_note_kernel("igemm_kernel<%d, %d, %d, %d, %d, %d>", BM, BN, WAVES_M, WAVES_N, VEC, MINW);
  hipLaunchKernelGGL((igemm_kernel<BM, BN, WAVES_M, WAVES_N, VEC, MINW>), grid, dim3(256), lds, s, p);
  return rsp_check_launch("igemm_kernel");
}

// N-tile width of a column segment.  Besides 128 / 64 / 32 there are two odd widths, run by 4 waves stacked along M (each wave
// 32 rows x the whole tile width): 160 for segments of 129..160 columns and 96 for 65..96 — one pass over the A operand with
// 90-100 % of the MFMA work useful, where 128 + a narrow second launch re-reads A for a few columns (R(2+1)D's 144-channel
// layers, S3D-G's 96 / 160) and a 128-wide tile for 96 columns wastes a quarter of it.
inline int tile_bn(int Cout) {
  if (Cout > 128 && Cout <= 160) return 160;
  if (Cout > 64 && Cout <= 96) return 96;
  return Cout > 64 ? 128 : (Cout > 32 ? 64 : 32);
}

// Launches of less than one round of 128-wide tiles run on the 64-WIDE tile instead (round 5): twice the units, so 392 tiles on 256 CUs
// are 2.7-3 short units per CU instead of "1.53 rounds" = two long ones on half of the CUs.  Measured per layer (tools/geom_bench.py,
// profiles/r05/experiments_r5.txt): S3D-G's 14 x 14 layers +2-5 % (528 -> 448: +16 %), R(2+1)D's 576 -> 256 temporal convolution +13 %,
// 7 x 7 / 4 x 4 layers +1-3 %; per step S3D-G 405.3 -> 412.5 (+1.8 %) and +0.25 % more with the 97..128-column launches included,
// R3D-18 +0.8 %, R(2+1)D +0.1 %, C3D -0.1 % (conv5: noise).  Launches of a full round or more lose up to 1 % (the 64-wide tile re-reads the
// A operand for half as many MFMAs: C3D -0.8 % with the limit at 1024 tiles) and keep the wide tile.
std::atomic<int> g_narrow_max_tiles{-1};      // >= 0: set through rsp_conv3d_set_option; -1: the environment / the default
static int narrow_max_tiles() {
  const int set = g_narrow_max_tiles.load(std::memory_order_relaxed);
  if (set >= 0) return set;
  static const int v = getenv("RSP_NARROW_MAX_TILES") ? atoi(getenv("RSP_NARROW_MAX_TILES")) : 512;      // (0: off; A/B switch, read once)
  return v;
}
// Launches of less than one UNIT per CU even on the 64-wide tile (round 6): S3D-G's 14 x 14 / 7 x 7 pointwise and (3,1,1) layers, R3D-18's
// layer3 / layer4 — 98 ... 256 tiles of 128 x 64.  Until round 5 they were cut along K (plan_split) to give every CU work, at the price
// of a partial round trip and a dependent reduce launch per convolution (121 per S3D-G step, 73 per R3D-18 step: +4.8 % / +4.2 % per step
// if they vanished, profiles/r05/experiments_r5.txt).  On the 32-wide tile the same launch has twice the units with the WHOLE K each:
// no partials, no second pass, one summation order per output.  (The 32-wide tile reads the A operand for a quarter of the 128-wide
// tile's MFMAs — irrelevant here: a launch this small is bound by the latency of its few chunks, not by L2 bandwidth.)
std::atomic<int> g_narrow32_max_units{-1};
static int narrow32_max_units() {
  const int set = g_narrow32_max_units.load(std::memory_order_relaxed);
  if (set >= 0) return set;
  static const int v = getenv("RSP_NARROW32_MAX_UNITS") ? atoi(getenv("RSP_NARROW32_MAX_UNITS")) : 256;      // (0: off; A/B switch, read once)
  return v;
}
// 0: the column count's own tile (tile_bn); 64 / 32: that tile width over all columns (no column segments)
inline int narrow_bn(long long M, int Cout, int nchunks) {
  const long long mt = rsp_cdiv(M, 128);
  if (Cout > 32 && nchunks >= 8 && mt * rsp_cdiv(Cout, 64) <= narrow32_max_units()) return 32;
  // (97..128 and > 160 columns: whole 64-wide tiles or nearly; 65..96 / 129..160 keep their 96- / 160-wide tile, whose last 64-wide
  //  tile would be half empty)
  const bool cols = Cout > 160 || (Cout > 96 && Cout <= 128);
  return (cols && mt * rsp_cdiv(Cout, 128) < narrow_max_tiles()) ? 64 : 0;
}
inline int tile_bn_of(const IgemmParams& p) { return p.bn_narrow ? p.bn_narrow : tile_bn(p.Cout); }

// 256 x 64 tiles for 33..64-column launches (round 6): four waves stacked along M, each 64 rows x 64 columns — the
// wave tile, accumulator count and LDS bytes per MFMA of the 128 x 128 instance (141 TF on C3D), where the 128 x 64 tile's waves
// (64 x 32) read 1.5 x the operand bytes per MFMA.  A 32 KB + B 8 KB per tile buffer: ONE buffer, three workgroups per CU; 168 VGPRs with
// 10-15 spilled (eight rows of loader state per thread instead of four).
// MEASURED, OFF BY DEFAULT (profiles/r06/experiments_r6.txt): back to back, R3D-18's layer1 (slice-major K) gains 4 % (116.7 -> 121.3 TF
// forward, 115.8 -> 120.5 input gradient), every tap-major candidate LOSES 1-2 % (the virtual-pixel stem 132.1 -> 131.0, R(2+1)D's
// 144 -> 64 temporal convolution 108.8 -> 107.2, S3D-G's (7,1,1) 140.8 -> 138.4) and the 56 x 56 pointwise layer 22 %; per step R3D-18
// 1325.7 -> 1320.4 clips/s, R(2+1)D 445.2 -> 442.0, S3D-G 418.5 -> 412.7, C3D 355.2 -> 354.1 (its 64-column input gradient of conv2:
// 136.6 -> 135.1 TF).  The 128 x 64 tile is not LDS-bound: what it loses against the 128 x 128 tile is the per-tile fixed cost, and a tile
// twice as tall pays the same cost per MFMA.  "tall_min_tiles" > 0 selects the instance (kernel tests, re-measurements).
std::atomic<int> g_tall_min_tiles{-1};
static int tall_min_tiles() {
  const int set = g_tall_min_tiles.load(std::memory_order_relaxed);
  if (set >= 0) return set;
  static const int v = getenv("RSP_TALL_MIN_TILES") ? atoi(getenv("RSP_TALL_MIN_TILES")) : 0;      // (0: off, the default; A/B switch, read once)
  return v;
}
inline bool tall_tiles(long long M, int bn, bool vec4) {
  const int lim = tall_min_tiles();
  return vec4 && bn == 64 && lim > 0 && M / 256 >= lim;
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int VEC, int MINW = 2>
int launch_multi_cfg(const IgemmMulti& m, int max_taps, hipStream_t s) {
  const size_t lds = (size_t)2 * (BM + BN) * BK * sizeof(float) + (size_t)(max_taps + 1) * sizeof(int4) + BM * sizeof(long long);
  static bool attr_set = false;
  if (!attr_set) {
    const size_t lds_max = (size_t)2 * (BM + BN) * BK * sizeof(float) + (MAX_TAPS + 1) * sizeof(int4) + BM * sizeof(long long);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_multi_kernel<BM, BN, WAVES_M, WAVES_N, VEC, MINW>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
    attr_set = true;
  }
  rsp_note_kernel("igemm_multi_kernel<%d, %d, %d, %d, %d, %d>", BM, BN, WAVES_M, WAVES_N, VEC, MINW);
  hipLaunchKernelGGL((igemm_multi_kernel<BM, BN, WAVES_M, WAVES_N, VEC, MINW>), dim3(m.start[m.n]), dim3(256), lds, s, m);
  return rsp_check_launch("igemm_multi_kernel");
}


// waves per SIMD the persistent instances are compiled for (launch bound) = workgroups per CU the grid is sized with
constexpr int persist_minw(int bn, int bm = 128) { return bm > 128 ? 3 : (bn > 128 ? 2 : (bn > 64 ? 3 : 4)); }

static bool persist_enabled() {
  static const bool off = getenv("RSP_NO_PERSIST") != nullptr;      // (A/B switch for measurements, read once)
  return !off;
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool KS>
int launch_persist_cfg(const IgemmParams& p, hipStream_t s) {
  constexpr int MINW = persist_minw(BN, BM);
  const int ntaps = p.nTd * p.nTh * p.nTw;
  constexpr int BNL = (BN + 31) / 32 * 32;      // (B tile rows in LDS: whole 32-row groups, igemm_persist)
  auto lds_of = [&](int nbuf, int taps) {
    return (size_t)nbuf * (BM + BNL) * BK * sizeof(float) + (size_t)(taps + 2) / 2 * sizeof(int4) + BM * sizeof(unsigned) +
           (size_t)WAVES_M * BNL * 2 * sizeof(float) + 4 * sizeof(unsigned) + sizeof(IgemmParams);
  };
  const size_t lds = lds_of(p.nbuf, ntaps);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_persist_kernel<BM, BN, WAVES_M, WAVES_N, KS, MINW>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_of(2, MAX_TAPS));
    attr_set = true;
  }
  const int n_units = p.full_tiles + (p.m_tiles * p.n_tiles - p.full_tiles) * p.splitk;
  // resident workgroups: LDS- and register-limited; every workgroup walks units wg, wg + G, wg + 2G, ... (G a multiple of 8 when
  // it is smaller than the unit count, so that a workgroup's units keep its XCD's contiguous share of the tiles: rsp_xcd_remap)
  int wpc = (int)(160 * 1024 / lds);
  wpc = wpc > MINW ? MINW : (wpc < 1 ? 1 : wpc);
  int G = 256 * wpc;
  if (n_units <= G) G = n_units;
  rsp_note_kernel("igemm_persist_kernel<%d, %d, %d, %d, %s, %d>", BM, BN, WAVES_M, WAVES_N, KS ? "true" : "false", MINW);
  hipLaunchKernelGGL((igemm_persist_kernel<BM, BN, WAVES_M, WAVES_N, KS, MINW>), dim3(G), dim3(256), lds, s, p, n_units);
  return rsp_check_launch("igemm_persist_kernel");
}

// Two-level summation over K on the long-K slice-major 128 x 128 launches (igemm_persist, FOLD): OFF by default — measured in round 6
// (profiles/r06/experiments_r6.txt), see DESIGN.md section 2.  "two_level_min_chunks" (rsp_conv3d_set_option / RSP_TWO_LEVEL_MIN_CHUNKS):
// launches of at least that many K chunks take the instance; 0: never.
constexpr int FOLD_CHUNKS = 16;      // 512 products per panel
std::atomic<int> g_two_level_min_chunks{-1};
static int two_level_min_chunks() {
  const int set = g_two_level_min_chunks.load(std::memory_order_relaxed);
  if (set >= 0) return set;
  static const int v = getenv("RSP_TWO_LEVEL_MIN_CHUNKS") ? atoi(getenv("RSP_TWO_LEVEL_MIN_CHUNKS")) : 0;
  return v;
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool KS, int FOLD>
int launch_persist_fold_cfg(const IgemmParams& p, hipStream_t s) {
  const int ntaps = p.nTd * p.nTh * p.nTw;
  auto lds_of = [&](int nbuf, int taps) {
    return (size_t)nbuf * (BM + BN) * BK * sizeof(float) + (size_t)(taps + 2) / 2 * sizeof(int4) + BM * sizeof(unsigned) +
           (size_t)WAVES_M * BN * 2 * sizeof(float) + 4 * sizeof(unsigned) + sizeof(IgemmParams);
  };
  const size_t lds = lds_of(p.nbuf, ntaps);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_persist_fold_kernel<BM, BN, WAVES_M, WAVES_N, KS, FOLD>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_of(2, MAX_TAPS));
    attr_set = true;
  }
  const int n_units = p.full_tiles + (p.m_tiles * p.n_tiles - p.full_tiles) * p.splitk;
  int wpc = (int)(160 * 1024 / lds);
  wpc = wpc > 2 ? 2 : (wpc < 1 ? 1 : wpc);
  int G = 256 * wpc;
  if (n_units <= G) G = n_units;
  rsp_note_kernel("igemm_persist_fold_kernel<%d, %d, %d, %d, %s, %d>", BM, BN, WAVES_M, WAVES_N, KS ? "true" : "false", FOLD);
  hipLaunchKernelGGL((igemm_persist_fold_kernel<BM, BN, WAVES_M, WAVES_N, KS, FOLD>), dim3(G), dim3(256), lds, s, p, n_units);
  return rsp_check_launch("igemm_persist_fold_kernel");
}

// Segments of 129..144 columns: the 160-wide tile's plan (one N tile, same splits) on the instance whose fifth column block is 16
// wide (igemm_persist, HALF).  RSP_NO_HALF_BLOCK=1: the 160-wide instance (A/B switch, read once).
static bool half_block_cols(int cols) {
  static const bool off = getenv("RSP_NO_HALF_BLOCK") != nullptr;
  return !off && cols > 128 && cols <= 144;
}
static bool half_block(const IgemmParams& p) { return half_block_cols(p.Cout); }

int launch_persist(IgemmParams& p, hipStream_t s) {
  const int bn = tile_bn_of(p);
  const int bm = p.bm > 128 ? 256 : 128;
  p.m_tiles = rsp_cdiv(p.M, bm);
  p.n_tiles = rsp_cdiv(p.Cout, bn);
  const int R = p.m_tiles * p.n_tiles - p.full_tiles;
  p.dR = fastdiv_make(R > 0 ? R : 1);
  p.dNtl = fastdiv_make(p.n_tiles);
  if (bm == 256) {      // the tall 64-wide instance (tall_tiles)
    if (bn != 64) {
      rsp_set_error("launch_persist: the 256-row tile is 64 columns wide");
      return RSP_EINVAL;
    }
    return p.kmajor ? launch_persist_cfg<256, 64, 4, 1, true>(p, s) : launch_persist_cfg<256, 64, 4, 1, false>(p, s);
  }
  if (p.kmajor) {
    if (bn == 128 && two_level_min_chunks() > 0 && p.nchunks >= two_level_min_chunks())
      return launch_persist_fold_cfg<128, 128, 2, 2, true, FOLD_CHUNKS>(p, s);
    switch (bn) {
      case 160: return half_block(p) ? launch_persist_cfg<128, 144, 4, 1, true>(p, s) : launch_persist_cfg<128, 160, 4, 1, true>(p, s);
      case 128: return launch_persist_cfg<128, 128, 2, 2, true>(p, s);
      case 96: return launch_persist_cfg<128, 96, 4, 1, true>(p, s);
      case 64: return launch_persist_cfg<128, 64, 2, 2, true>(p, s);
      default: return launch_persist_cfg<128, 32, 4, 1, true>(p, s);
    }
  }
  switch (bn) {
    case 160: return half_block(p) ? launch_persist_cfg<128, 144, 4, 1, false>(p, s) : launch_persist_cfg<128, 160, 4, 1, false>(p, s);
    case 128: return launch_persist_cfg<128, 128, 2, 2, false>(p, s);
    case 96: return launch_persist_cfg<128, 96, 4, 1, false>(p, s);
    case 64: return launch_persist_cfg<128, 64, 2, 2, false>(p, s);
    default: return launch_persist_cfg<128, 32, 4, 1, false>(p, s);
  }
}

// bytes reachable from the output base pointer (all positions of the output tensor, this segment's columns); 0 if >= 4 GiB
static unsigned out_extent_bytes(const IgemmParams& p) {
  const unsigned long long pos = (unsigned long long)p.Nb * p.oDm * p.oHm * p.oWm;
  const unsigned long long b = ((pos - 1) * (unsigned long long)p.out_ld + (unsigned long long)p.Cout) * 4ull;
  return b < 0x7ffffff0ull ? (unsigned)b : 0u;      // (< 2 GiB: an offset with bit 31 set is out of range whatever is added to it)
}

inline void fill_fastdiv_linear(IgemmParams& p) {
  const bool dense = p.oSd == 1 && p.oSh == 1 && p.oSw == 1 && p.oOd == 0 && p.oOh == 0 && p.oOw == 0 && p.oDm == p.Gd &&
                     p.oHm == p.Gh && p.oWm == p.Gw;
  p.linear_out = dense && !p.dmajor;
  p.dm_dense = dense && p.dmajor;
  p.dP = fastdiv_make(p.Gh * p.Gw);
}

inline void fill_fastdiv(IgemmParams& p) {
  p.dGw = fastdiv_make(p.Gw); p.dGh = fastdiv_make(p.Gh); p.dGd = fastdiv_make(p.Gd);
  p.dCin = fastdiv_make(p.Cin); p.dTw = fastdiv_make(p.nTw); p.dTh = fastdiv_make(p.nTh);
  p.adv_tap = BK / p.Cin;
  p.adv_ci = BK % p.Cin;
  p.kmajor = k_slice_major(p.Cin, p.nTd * p.nTh * p.nTw) ? 1 : 0;
  static const bool no_skip = getenv("RSP_NO_PAD_SKIP") != nullptr;      // (A/B switches for measurements, read once)
  static const bool no_dmajor = getenv("RSP_NO_DMAJOR") != nullptr;
  p.skip_pad = no_skip ? 0 : 1;
  // parts: eight (one per XCD) when each part's frame still spans several tiles, else one
  p.Np = (p.Nb % 8 == 0 && (long long)(p.Nb / 8) * p.Gh * p.Gw >= 4 * 128) ? p.Nb / 8 : (p.Nb > 0 ? p.Nb : 1);
  p.dNp = fastdiv_make(p.Np);
  p.dGdNp = fastdiv_make(p.Gd * p.Np);
  // depth-major rows: slice-major kernels whose depth taps reach into the padding for some output frame
  const int dlo = p.offstep > 0 ? p.off0d : p.off0d - (p.nTd - 1);
  const int dhi = (p.Gd - 1) * p.sD + (p.offstep > 0 ? p.off0d + p.nTd - 1 : p.off0d);
  const int ntaps = p.nTd * p.nTh * p.nTw;
  p.cpt = p.Cin % BK == 0 ? p.Cin / BK : 0;
  static const bool no_tm = getenv("RSP_NO_TM_SKIP") != nullptr;
  // (padding in any dimension can make a tap dead for a whole tile, but only depth does so often enough to pay for the walk)
  p.dThw = fastdiv_make(p.nTh * p.nTw);
  p.tm_skip = (p.skip_pad && !no_tm && !p.kmajor && p.nTd > 1 && (dlo < 0 || dhi >= p.Di)) ? 1 : 0;
  // depth-major rows where they buy something: frames of fewer than four tiles per sample (n-major tiles would straddle frames) or
  // launches of fewer than ~8 rounds (the short-frames-last order matters); big frames in long launches keep the n-major rows and
  // their linear epilogue (S3D-G's (7,1,1) convolution: 98 tiles per frame, 16 rounds — measured 0.25 % of the step slower depth-major)
  const long long tiles_est = ((long long)p.M + 127) / 128 * ((p.Cout + 127) / 128);
  const bool wants_dmajor = (long long)p.Gh * p.Gw < 4 * 128 || tiles_est < 8 * 768;
  p.dmajor = (p.skip_pad && !no_dmajor && (p.kmajor || p.tm_skip) && wants_dmajor && p.nTd > 1 && p.Gd > 1 && p.Nb > 1 &&
              (dlo < 0 || dhi >= p.Di)) ? 1 : 0;
  (void)ntaps;
  p.dNt = fastdiv_make(p.nTd * p.nTh * p.nTw);
  fill_fastdiv_linear(p);
}

int launch_igemm(IgemmParams& p, bool vec4, hipStream_t s) {
  // BM is fixed at 128 (stat partials are defined on 128-row tiles); BN follows Cout.
  int bn = tile_bn_of(p);
  if (!vec4 && (bn == 160 || bn == 96)) bn = 128;      // the scalar-gather fallback only has the power-of-two tiles
  p.m_tiles = rsp_cdiv(p.M, 128);
  p.n_tiles = rsp_cdiv(p.Cout, bn);
  if (vec4 && p.kmajor) {
    switch (bn) {
      case 160: return launch_ks_cfg<128, 160, 4, 1>(p, s);
      case 128: return launch_ks_cfg<128, 128, 2, 2>(p, s);
      case 96: return launch_ks_cfg<128, 96, 4, 1>(p, s);
      case 64: return launch_ks_cfg<128, 64, 2, 2>(p, s);
      default: return launch_ks_cfg<128, 32, 4, 1>(p, s);
    }
  }
  if (bn == 160) return launch_cfg<128, 160, 4, 1, 4>(p, s);
  if (bn == 96) return launch_cfg<128, 96, 4, 1, 4>(p, s);
  if (bn == 128) return vec4 ? launch_cfg<128, 128, 2, 2, 4>(p, s) : launch_cfg<128, 128, 2, 2, 1>(p, s);
  if (bn == 64) return vec4 ? launch_cfg<128, 64, 2, 2, 4>(p, s) : launch_cfg<128, 64, 2, 2, 1>(p, s);
  return vec4 ? launch_cfg<128, 32, 4, 1, 4>(p, s) : launch_cfg<128, 32, 4, 1, 1>(p, s);
}

// Tail split by a small cost model.  A CU holds `wpc` workgroups of this tile shape, so the machine has 256*wpc slots and
// equal-sized tiles run in ceil(tiles/slots) rounds: C3D conv3 (3136 tiles, 512 slots) spends 1/8 of its time in a last
// round that is 1/8 full, conv4 (784 tiles) half of it in a round that is half full.  The plan keeps the whole rounds as
// they are and cuts only the remainder R along K into S slices each (S*R units of ceil(nchunks/S) chunks), at the price
// of an fp32 partial round trip + one reduce launch for those tiles.
struct SplitPlan {
  int full_tiles;   // multiple of n_tiles
  int splitk;       // S (1: no split)
  int cps;          // chunks per slice
};

// One LDS buffer instead of two (igemm_body): only where it buys a third workgroup per CU — the 128- and 96-wide DMA tiles
// (166 / 148 VGPRs; the 160-wide tile's 215 VGPRs allow two, the 64- and 32-wide tiles already run 3+) — and only for launches
// of at least one full 768-tile round: below that the dispatcher packs three workgroups onto some CUs while others idle
// (392 tiles: +28 % time), and the 64-wide tile loses 9 % to the extra barrier without gaining a workgroup.
bool single_buffer(int m_tiles, int n_tiles, int bn, bool vec4) {
  return vec4 && (bn == 128 || bn == 96) && (long long)m_tiles * n_tiles >= 768;
}

SplitPlan plan_split(int m_tiles, int n_tiles, int bn, bool vec4, int nchunks, int Cout, int bm = 128) {
  const int tiles = m_tiles * n_tiles;
  // resident workgroups per CU of igemm_kernel<128, bn, .., VEC>: LDS-limited (DMA variants) or VGPR-limited (scalar gather)
  int wpc = vec4 ? (bn >= 96 ? 2 : 3) : (bn >= 64 ? 2 : 3);
  if (single_buffer(m_tiles, n_tiles, bn, vec4)) wpc = 3;
  if (bm == 256) {      // the tall 64-wide tile: the 128 x 128 tile's work per chunk, three workgroups per CU on one tile buffer
    wpc = 3;
    bn = 128;
  }
  const int slots = 256 * wpc;
  SplitPlan best = {tiles, 1, nchunks};
  if (nchunks < 8) return best;
  const double t_chunk = 2.35e-6 * wpc * bn / 128.0, t_chunk1 = 2.7e-6 * bn / 128.0;   // measured on the 128x128 tile
  const double ovh = 3.0;   // prologue + epilogue of a unit, in chunk times
  // The matrix pipe is the CU's: with more than one unit per CU the time is (units on the fullest CU) x (a unit's chunks), whatever
  // the number of co-resident workgroups — 294 tiles on 256 CUs take two tile times (38 CUs hold two), and a K split that leaves
  // every CU three or four shorter units is faster (S3D-G's 14x14 layers: tools/small_launch_probe.py).  (Until round 4 this was
  // quantised by SLOTS: 588 units counted as two rounds of 512.)
  auto round_time = [&](long long units, int chunks) {
    if (units <= 0) return 0.0;
    if (units <= 256) return (chunks + ovh) * t_chunk1;            // one workgroup per CU: no sharing of the matrix pipe
    return (double)rsp_cdiv(units, 256) * (chunks + ovh) * (t_chunk / wpc);
  };
  int full = tiles / slots * slots;
  full -= full % n_tiles;
  const int R = tiles - full;
  if (R == 0) return best;
  // only the tail differs between candidates (the whole rounds cost the same under every plan)
  double best_t = round_time(R, nchunks);
  for (int S = 2; S <= 16 && S * 4 <= nchunks; ++S) {
    const int cps = rsp_cdiv(nchunks, S), Se = rsp_cdiv(nchunks, cps);
    if (Se != S) continue;
    double t = round_time((long long)R * Se, cps);
    t += 4e-6 + (double)(Se + 1) * ((double)R / n_tiles * bm) * Cout * 4.0 / 3.0e12;
    if (t < best_t * 0.98) {
      best_t = t;
      best = {full, Se, cps};
    }
  }
  return best;
}

bool desc_ok(const rsp_conv3d_desc* d) {
  if (!d) return false;
  if (d->N <= 0 || d->Cin <= 0 || d->Cout <= 0) return false;
  if (d->kT * d->kH * d->kW > MAX_TAPS || d->kT <= 0 || d->kH <= 0 || d->kW <= 0) return false;
  if (d->sT <= 0 || d->sH <= 0 || d->sW <= 0 || d->pT < 0 || d->pH < 0 || d->pW < 0) return false;
  if (d->Do != (d->Di + 2 * d->pT - d->kT) / d->sT + 1) return false;
  if (d->Ho != (d->Hi + 2 * d->pH - d->kH) / d->sH + 1) return false;
  if (d->Wo != (d->Wi + 2 * d->pW - d->kW) / d->sW + 1) return false;
  if (d->Do <= 0 || d->Ho <= 0 || d->Wo <= 0) return false;
  if (d->in_ld < d->Cin || d->out_ld < d->Cout) return false;
  if ((long long)d->N * d->Do * d->Ho * d->Wo >= (1ll << 31)) return false;
  if ((long long)d->N * d->Di * d->Hi * d->Wi >= (1ll << 31)) return false;
  return true;
}

void fill_reduce(ReduceParams& r, const IgemmParams& p) {
  r.partial = p.partial;
  r.bias = p.bias;
  r.y = p.y;
  r.stat = p.stat;
  r.M = p.M;
  r.Cout = p.Cout;
  r.stat_ld = p.stat_ld;
  r.splitk = p.splitk;
  r.row0 = p.tail_row0;
  r.Gd = p.Gd; r.Gh = p.Gh; r.Gw = p.Gw;
  r.oDm = p.oDm; r.oHm = p.oHm; r.oWm = p.oWm;
  r.oSd = p.oSd; r.oSh = p.oSh; r.oSw = p.oSw;
  r.oOd = p.oOd; r.oOh = p.oOh; r.oOw = p.oOw;
  r.out_ld = p.out_ld;
  r.Nb = p.Nb;
  r.dmajor = p.dmajor;
  r.Np = p.Np;
}


size_t split_partial_bytes(const SplitPlan& sp, long long M, int n_tiles, int Cout, int bm = 128) {
  if (sp.splitk <= 1) return 0;
  const long long row0 = (long long)(sp.full_tiles / n_tiles) * bm;
  return (size_t)sp.splitk * (size_t)(M - row0) * Cout * sizeof(float);
}

int run_igemm_segment(IgemmParams& p, bool vec4, void* workspace, size_t ws_bytes, hipStream_t s) {
  const float* zero_page = igemm_zero_page();
  if (!zero_page) {
    rsp_set_error("hipGetSymbolAddress(g_zero) failed");
    return RSP_ELAUNCH;
  }
  p.zero = zero_page;
  p.nchunks = rsp_cdiv(p.K, BK);
  fill_fastdiv(p);
  if (!vec4) {      // only the LDS-DMA kernels know the depth-major enumeration and the skipping walks
    p.dmajor = 0;
    p.tm_skip = 0;
    fill_fastdiv_linear(p);
  }
  int bn = tile_bn_of(p);
  if (!vec4 && (bn == 160 || bn == 96)) bn = 128;
  p.y_bytes = out_extent_bytes(p);
  // rows per tile: 256 on the tall 64-wide instance of the persistent kernel (tall_tiles), which needs 32-bit output offsets
  const int bm = (tall_tiles(p.M, bn, vec4) && persist_enabled() && p.y_bytes != 0) ? 256 : 128;
  p.bm = bm;
  const int m_tiles = rsp_cdiv(p.M, bm), n_tiles = rsp_cdiv(p.Cout, bn);
  SplitPlan sp = plan_split(m_tiles, n_tiles, bn, vec4, p.nchunks, p.Cout, bm);
  p.nbuf = (bm == 256 || single_buffer(m_tiles, n_tiles, bn, vec4)) ? 1 : 2;
  if (sp.splitk > 1 && (!workspace || ws_bytes < split_partial_bytes(sp, p.M, n_tiles, p.Cout, bm)))
    sp = {m_tiles * n_tiles, 1, p.nchunks};   // degrade gracefully: still correct
  p.full_tiles = sp.full_tiles;
  p.splitk = sp.splitk;
  p.chunks_per_split = sp.cps;
  p.tail_row0 = sp.full_tiles / n_tiles * bm;
  p.partial = p.splitk > 1 ? reinterpret_cast<float*>(workspace) : nullptr;
  const size_t pbytes = split_partial_bytes(sp, p.M, n_tiles, p.Cout, bm);
  p.partial_bytes = pbytes < 0x7ffffff0ull ? (unsigned)pbytes : 0u;
  // (the long tap-major 128-wide launches — C3D conv2: 54 chunks per tile — measured 1 % faster on the per-tile kernel; everything
  //  else equal or better persistent: R3D-18 +2.9 %, R(2+1)D / S3D-G +0.3 % per step, profiles/r04/experiments_r4.txt)
  const bool long_tm128 = !p.kmajor && bn == 128 && p.nchunks >= 48;
  const bool persist = vec4 && persist_enabled() && !long_tm128 && p.y_bytes != 0 && (p.splitk == 1 || p.partial_bytes != 0);
  // the persistent 64-wide tile is compiled for four waves per SIMD (121-128 VGPRs): with ONE tile buffer (24.5 KB) four workgroups
  // share a CU on launches of at least one such round (R3D-18 +0.5 %, R(2+1)D +0.2 % per step); the per-tile 64-wide kernel, three
  // per CU either way, lost 9 % to the extra barrier (single_buffer)
  if (persist && bm == 128 && bn == 64 && (long long)m_tiles * n_tiles >= 1024) p.nbuf = 1;
  int rc = persist ? launch_persist(p, s) : launch_igemm(p, vec4, s);
  if (rc != RSP_OK) return rc;
  if (p.splitk > 1) {
    ReduceParams r;
    fill_reduce(r, p);
    dim3 grid(rsp_cdiv(p.M - p.tail_row0, 128), rsp_cdiv(p.Cout, 64));
    if (p.Cout % 4 == 0 && p.out_ld % 4 == 0 && rsp_aligned16(p.y) && (!p.bias || rsp_aligned16(p.bias)))
      hipLaunchKernelGGL(splitk_reduce_vec_kernel, grid, dim3(256), 0, s, r);
    else
      hipLaunchKernelGGL(splitk_reduce_kernel, grid, dim3(256), 0, s, r);
    rc = rsp_check_launch("splitk_reduce_kernel");
  }
  return rc;
}

// Column segments of one GEMM.  The N tiles are 128 wide (64 / 32 for narrow outputs); a channel count such as 144, 160 or 288
// would run its last 128-wide tile almost empty (144 -> 2 tiles, 56 % of the MFMA work useful).  When the part beyond the
// last multiple of 128 fits a 64- or 32-wide tile it is launched on its own with that tile shape (144 = 128 + 16 -> 90 %):
// same kernel, operands addressed through offset base pointers; the stat-partial rows keep the full convolution's pitch.
struct Segments {
  int n;
  int c0[2], width[2];
};

Segments plan_segments(int Cout) {
  Segments g;
  const int full = Cout / 128 * 128, r = Cout - full;
  if (full == 0 || r == 0 || r > 64 || tile_bn(Cout) == 160) {
    g.n = 1; g.c0[0] = 0; g.width[0] = Cout; g.c0[1] = 0; g.width[1] = 0;
  } else {
    g.n = 2; g.c0[0] = 0; g.width[0] = full; g.c0[1] = full; g.width[1] = r;
  }
  return g;
}

// ---- the direct kernel (igemm_direct_kernel): OFF by default since the 64-wide tiles of narrow_tiles().  History (profiles/r05/
// direct_vs_tile_r5h.txt, experiments_r5.txt): against 128-wide tiles it won the 7 x 7 / 4 x 4 stages (13-30 % per launch) and the
// 1x1x1 stride-2 shortcuts (15-25 %) and lost everything above ~1.4 GFLOP (it tops out at 45-50 TFLOP/s: L2 latency under three waves
// per SIMD); against 64-wide tiles it loses those too (eleven small geometries back to back: 361 vs 326 us, tools/geom_bench.py small).
// RSP_DIRECT_MAX_TILES=N selects it for launches of at most N 128 x 128 tiles with Cin % 16 == 0 (re-measurements, and the GPU tests run
// the small convolution cases through it in a child interpreter).
static int direct_max_tiles() {
  static const int v = getenv("RSP_DIRECT_MAX_TILES") ? atoi(getenv("RSP_DIRECT_MAX_TILES")) : 0;
  return v;
}
bool direct_applies(long long M, int Cout, int Cin, int K, bool vec4) {
  (void)K;
  if (!vec4 || direct_max_tiles() <= 0 || Cin % 16 != 0 || M <= 0) return false;
  return (long long)rsp_cdiv(M, 128) * rsp_cdiv(Cout, 128) <= direct_max_tiles();
}
// K split of a direct launch: enough units for ~3 workgroups per CU, at least 4 chunks per slice
struct DirectPlan {
  int splitk, cps;
};
DirectPlan direct_plan(long long M, int Cout, int nchunks) {
  const long long tiles = (long long)rsp_cdiv(M, 128) * rsp_cdiv(Cout, 64);
  int S = (int)((768 + tiles - 1) / tiles);
  if (S > nchunks / 4) S = nchunks / 4;
  if (S > 32) S = 32;
  if (S < 1) S = 1;
  const int cps = rsp_cdiv(nchunks, S);
  return {rsp_cdiv(nchunks, cps), cps};
}
size_t direct_partial_bytes(long long M, int Cout, int K) {
  const DirectPlan dp = direct_plan(M, Cout, rsp_cdiv(K, BK));
  return dp.splitk > 1 ? (size_t)dp.splitk * (size_t)M * Cout * sizeof(float) : 0;
}

// K split of the classes of a FEW-TILE strided input gradient that share one igemm_multi_kernel launch (dgrad_run): slices of about
// equal length across the classes (a class has 1 ... kT*kH*kW/(sT*sH*sW) taps), enough units for ~3 workgroups per CU, at least
// 4 chunks per slice; class c's partials follow class c-1's in the workspace.
struct MultiSplit {
  int S[MAX_MULTI], cps[MAX_MULTI];
  size_t off[MAX_MULTI], bytes;
};
static MultiSplit plan_multi_split(int n, const long long* M, const int* tiles, const int* nchunks, int Cout) {
  MultiSplit ms;
  long long total = 0;
  for (int i = 0; i < n; ++i) total += (long long)tiles[i] * nchunks[i];
  long long target = (total + 767) / 768;
  if (target < 4) target = 4;
  ms.bytes = 0;
  for (int i = 0; i < n; ++i) {
    int S = (int)((nchunks[i] + target - 1) / target);
    if (S > nchunks[i] / 4) S = nchunks[i] / 4;
    if (S > 16) S = 16;
    if (S < 1) S = 1;
    const int cps = rsp_cdiv(nchunks[i], S);
    ms.cps[i] = cps;
    ms.S[i] = rsp_cdiv(nchunks[i], cps);
    ms.off[i] = ms.bytes;
    if (ms.S[i] > 1) ms.bytes += rsp_align_up((size_t)ms.S[i] * (size_t)M[i] * Cout * sizeof(float), 256);
  }
  return ms;
}
static bool multi_split_enabled() {
  static const bool off = getenv("RSP_NO_MULTI_SPLIT") != nullptr;      // (A/B switch for measurements, read once)
  return !off;
}
int run_direct(IgemmParams& p, void* workspace, size_t ws_bytes, hipStream_t s) {
  p.zero = nullptr;
  p.nchunks = rsp_cdiv(p.K, BK);
  fill_fastdiv(p);
  p.dmajor = 0;                      // plain (sample, depth, h, w) rows
  p.tm_skip = 0;
  fill_fastdiv_linear(p);
  p.m_tiles = rsp_cdiv(p.M, 128);
  p.n_tiles = rsp_cdiv(p.Cout, 64);
  DirectPlan dp = direct_plan(p.M, p.Cout, p.nchunks);
  if (dp.splitk > 1 && (!workspace || ws_bytes < (size_t)dp.splitk * (size_t)p.M * p.Cout * sizeof(float))) dp = {1, p.nchunks};
  p.splitk = dp.splitk;
  p.chunks_per_split = dp.cps;
  p.full_tiles = 0;
  p.tail_row0 = 0;
  p.partial = p.splitk > 1 ? reinterpret_cast<float*>(workspace) : nullptr;
  rsp_note_kernel("igemm_direct_kernel");
  hipLaunchKernelGGL(igemm_direct_kernel, dim3((unsigned)(p.m_tiles * p.n_tiles * p.splitk)), dim3(256), 0, s, p);
  int rc = rsp_check_launch("igemm_direct_kernel");
  if (rc != RSP_OK) return rc;
  if (p.splitk > 1) {
    ReduceParams r;
    fill_reduce(r, p);
    dim3 grid(rsp_cdiv(p.M, 128), rsp_cdiv(p.Cout, 64));
    if (p.Cout % 4 == 0 && p.out_ld % 4 == 0 && rsp_aligned16(p.y) && (!p.bias || rsp_aligned16(p.bias)))
      hipLaunchKernelGGL(splitk_reduce_vec_kernel, grid, dim3(256), 0, s, r);
    else
      hipLaunchKernelGGL(splitk_reduce_kernel, grid, dim3(256), 0, s, r);
    rc = rsp_check_launch("splitk_reduce_kernel");
  }
  return rc;
}

int run_igemm(IgemmParams& p, bool vec4, void* workspace, size_t ws_bytes, hipStream_t s) {
  p.stat_ld = p.Cout;
  if (direct_applies(p.M, p.Cout, p.Cin, p.K, vec4)) return run_direct(p, workspace, ws_bytes, s);
  if (const int nb = vec4 ? narrow_bn(p.M, p.Cout, rsp_cdiv(p.K, BK)) : 0) {      // one launch of 64- / 32-wide tiles over all columns (no column segments)
    p.bn_narrow = nb;
    return run_igemm_segment(p, vec4, workspace, ws_bytes, s);
  }
  const Segments g = plan_segments(p.Cout);
  if (g.n == 1) return run_igemm_segment(p, vec4, workspace, ws_bytes, s);
  for (int i = 0; i < g.n; ++i) {
    IgemmParams q = p;
    const int c0 = g.c0[i];
    q.Cout = g.width[i];
    q.w = p.w + (long long)c0 * p.Kld;
    q.w_bytes = (unsigned)((unsigned long long)q.Cout * p.Kld * 4ull);
    q.bias = p.bias ? p.bias + c0 : nullptr;
    q.y = p.y + c0;
    q.stat = p.stat ? p.stat + 2ll * c0 : nullptr;
    const int rc = run_igemm_segment(q, vec4, workspace, ws_bytes, s);   // same stream: the workspace is free again when it starts
    if (rc != RSP_OK) return rc;
  }
  return RSP_OK;
}

size_t igemm_partial_bytes_segment(long long M, int Cout, int K);

size_t igemm_partial_bytes(long long M, int Cout, int K) {
  const Segments g = plan_segments(Cout);
  size_t best = direct_partial_bytes(M, Cout, K);      // (whether the direct kernel runs depends on alignment, unknown here)
  if (const int nb = narrow_bn(M, Cout, rsp_cdiv(K, BK))) {
    const int n_tiles = rsp_cdiv(Cout, nb);
    const size_t b = split_partial_bytes(plan_split(rsp_cdiv(M, 128), n_tiles, nb, true, rsp_cdiv(K, BK), Cout), M, n_tiles, Cout);
    best = b > best ? b : best;
  }
  for (int i = 0; i < g.n; ++i) {
    const size_t b = igemm_partial_bytes_segment(M, g.width[i], K);
    best = b > best ? b : best;
  }
  return best;
}

size_t igemm_partial_bytes_segment(long long M, int Cout, int K) {
  const int bn = tile_bn(Cout), bn_s = (bn == 160 || bn == 96) ? 128 : bn;     // scalar-gather fallback: power-of-two tiles only
  const int m_tiles = rsp_cdiv(M, 128), n_tiles = rsp_cdiv(Cout, bn), n_tiles_s = rsp_cdiv(Cout, bn_s);
  // the gather variant (hence the plan) depends on pointer alignment, unknown here: size for the larger of the two
  size_t a = split_partial_bytes(plan_split(m_tiles, n_tiles, bn, true, rsp_cdiv(K, BK), Cout), M, n_tiles, Cout);
  const size_t b = split_partial_bytes(plan_split(m_tiles, n_tiles_s, bn_s, false, rsp_cdiv(K, BK), Cout), M, n_tiles_s, Cout);
  if (tall_tiles(M, bn, true)) {      // ... or on 256-row tiles (whether they run depends on the output extent, unknown here)
    const size_t t = split_partial_bytes(plan_split(rsp_cdiv(M, 256), n_tiles, bn, true, rsp_cdiv(K, BK), Cout, 256), M, n_tiles, Cout, 256);
    a = t > a ? t : a;
  }
  return a > b ? a : b;
}

static inline int mod_pos(int a, int m) { return ((a % m) + m) % m; }

// One stride-parity class of the input-gradient problem.
struct DgradClass {
  int rt, rh, rw;     // residues
  int k0t, k0h, k0w;  // first kernel index per dim
  int nt, nh, nw;     // taps per dim
  int Gd, Gh, Gw;     // class grid
};

DgradClass dgrad_class(const rsp_conv3d_desc* d, int c) {
  DgradClass g;
  g.rw = c % d->sW; g.rh = (c / d->sW) % d->sH; g.rt = c / (d->sW * d->sH);
  g.k0t = mod_pos(g.rt + d->pT, d->sT); g.k0h = mod_pos(g.rh + d->pH, d->sH); g.k0w = mod_pos(g.rw + d->pW, d->sW);
  g.nt = g.k0t < d->kT ? (d->kT - 1 - g.k0t) / d->sT + 1 : 0;
  g.nh = g.k0h < d->kH ? (d->kH - 1 - g.k0h) / d->sH + 1 : 0;
  g.nw = g.k0w < d->kW ? (d->kW - 1 - g.k0w) / d->sW + 1 : 0;
  g.Gd = g.rt < d->Di ? (d->Di - 1 - g.rt) / d->sT + 1 : 0;
  g.Gh = g.rh < d->Hi ? (d->Hi - 1 - g.rh) / d->sH + 1 : 0;
  g.Gw = g.rw < d->Wi ? (d->Wi - 1 - g.rw) / d->sW + 1 : 0;
  return g;
}

// Host replica of the kernels' walk over K: the share of a problem's (tile, chunk) pairs that igemm_body executes once the chunks
// whose taps are padding for a whole tile are skipped (1.0 when nothing is skipped).  Same row enumeration (row_decode), same
// validity bits (span), same liveness tests as the device code; the column segments of one GEMM share the rows, hence the share.
bool direct_applies(long long M, int Cout, int Cin, int K, bool vec4);

double igemm_live_fraction(IgemmParams p, bool vec4, bool may_direct = true) {
  if (may_direct && direct_applies(p.M, p.Cout, p.Cin, p.K, vec4)) return 1.0;      // igemm_direct_kernel walks every chunk
  p.nchunks = rsp_cdiv(p.K, BK);
  fill_fastdiv(p);
  if (!vec4) return 1.0;
  const bool KS = p.kmajor != 0;
  const bool tms = !KS && p.tm_skip;
  if (!(KS && p.skip_pad) && !tms) return 1.0;
  auto span = [&](int base, int D, int nT) -> unsigned {
    const int c = p.offstep > 0 ? base : D - 1 - base;
    const int lo = std::min(std::max(0, -c), 8), hi = std::min(nT - 1, D - 1 - c);
    return hi >= lo ? (2u << hi) - (1u << lo) : 0u;
  };
  // (rows per tile as run_igemm / run_igemm_segment choose them; 32-bit output offsets assumed)
  const int nbw = narrow_bn(p.M, p.Cout, p.nchunks);
  const int bm = (persist_enabled() && tall_tiles(p.M, nbw ? nbw : tile_bn(plan_segments(p.Cout).width[0]), vec4)) ? 256 : 128;
  const int ntaps = p.nTd * p.nTh * p.nTw, m_tiles = rsp_cdiv(p.M, bm);
  long long live = 0;
  for (int mt = 0; mt < m_tiles; ++mt) {
    unsigned tmask = 0;
    for (int r = mt * bm; r < std::min(p.M, mt * bm + bm); ++r) {
      const int gw = r % p.Gw, q1 = r / p.Gw, gh = q1 % p.Gh, q2 = q1 / p.Gh;
      int gd;
      if (p.dmajor) {
        const int rem = q2 % (p.Gd * p.Np), g = rem / p.Np;
        gd = g + 1 == p.Gd ? 0 : g + 1;
      } else {
        gd = q2 % p.Gd;
      }
      tmask |= span(gd * p.sD + p.off0d, p.Di, p.nTd) | (span(gh * p.sH + p.off0h, p.Hi, p.nTh) << 8) |
               (span(gw * p.sW + p.off0w, p.Wi, p.nTw) << 16);
    }
    if (KS || p.cpt > 0) {
      int live_taps = 0;
      for (int ad = 0; ad < p.nTd; ++ad)
        for (int ah = 0; ah < p.nTh; ++ah)
          for (int aw = 0; aw < p.nTw; ++aw) {
            const unsigned need = (1u << ad) | (1u << (8 + ah)) | (1u << (16 + aw));
            live_taps += (tmask & need) == need ? 1 : 0;
          }
      live += (long long)live_taps * (p.Cin / BK);
    } else {
      for (int kc = 0; kc < p.nchunks; ++kc) {
        const int t0 = kc * BK / p.Cin, t1 = std::min((kc * BK + BK - 1) / p.Cin, ntaps - 1);
        const unsigned sp = (2u << (t1 / (p.nTh * p.nTw))) - (1u << (t0 / (p.nTh * p.nTw)));
        live += (tmask & sp) != 0 ? 1 : 0;
      }
    }
  }
  return (double)live / ((double)m_tiles * p.nchunks);
}

size_t dgrad_wpack_bytes(const rsp_conv3d_desc* d) {
  const int nclass = d->sT * d->sH * d->sW;
  return rsp_align_up(((size_t)d->kT * d->kH * d->kW * d->Cout + 4 * (size_t)nclass) * d->Cin * sizeof(float), 256);
}

// The GEMM description of a forward convolution; returns whether the LDS-DMA kernels apply (16-byte rows, 32-bit offsets).
static bool fill_fwd_params(const rsp_conv3d_desc* d, const float* x, const float* w_packed, const float* bias, float* y,
                            float* stat_partials, IgemmParams& p) {
  memset(&p, 0, sizeof p);
  p.x = x; p.w = w_packed; p.bias = bias; p.y = y; p.stat = stat_partials;
  p.M = d->N * d->Do * d->Ho * d->Wo;
  p.Nb = d->N;
  p.Gd = d->Do; p.Gh = d->Ho; p.Gw = d->Wo;
  p.oDm = d->Do; p.oHm = d->Ho; p.oWm = d->Wo;
  p.oSd = p.oSh = p.oSw = 1;
  p.oOd = p.oOh = p.oOw = 0;
  p.out_ld = d->out_ld; p.Cout = d->Cout;
  p.Di = d->Di; p.Hi = d->Hi; p.Wi = d->Wi; p.in_ld = d->in_ld; p.Cin = d->Cin;
  p.sD = d->sT; p.sH = d->sH; p.sW = d->sW;
  p.nTd = d->kT; p.nTh = d->kH; p.nTw = d->kW;
  p.off0d = -d->pT; p.off0h = -d->pH; p.off0w = -d->pW;
  p.offstep = 1;
  p.K = d->kT * d->kH * d->kW * d->Cin;
  p.Kld = (int)rsp_align_up((size_t)p.K, 4);
  const unsigned long long xb = (unsigned long long)d->N * d->Di * d->Hi * d->Wi * d->in_ld * 4ull;
  const unsigned long long wb = (unsigned long long)d->Cout * p.Kld * 4ull;
  p.x_bytes = (unsigned)xb;
  p.w_bytes = (unsigned)wb;
  // the LDS-DMA path addresses with 32-bit byte offsets; bigger tensors take the (slower) scalar-gather path
  return (d->Cin % 4 == 0) && (d->in_ld % 4 == 0) && rsp_aligned16(x) && xb < (1ull << 32) && wb < (1ull << 32) &&
                    d->kT <= 8 && d->kH <= 8 && d->kW <= 8;   // per-dimension validity bits of the DMA path
}

// ... and of one stride-parity class of its input gradient (w: this class's packed weights).
static bool fill_dgrad_class_params(const rsp_conv3d_desc* d, const DgradClass& g, const float* dy, const float* w, float* dx,
                                    IgemmParams& p) {
  const int Kld = (int)rsp_align_up((size_t)g.nt * g.nh * g.nw * d->Cout, 4);
    memset(&p, 0, sizeof p);
    p.x = dy; p.w = w; p.bias = nullptr; p.y = dx; p.stat = nullptr;
    p.M = d->N * g.Gd * g.Gh * g.Gw;
    p.Nb = d->N;
    p.Gd = g.Gd; p.Gh = g.Gh; p.Gw = g.Gw;
    p.oDm = d->Di; p.oHm = d->Hi; p.oWm = d->Wi;
    p.oSd = d->sT; p.oSh = d->sH; p.oSw = d->sW;
    p.oOd = g.rt; p.oOh = g.rh; p.oOw = g.rw;
    p.out_ld = d->in_ld; p.Cout = d->Cin;
    p.Di = d->Do; p.Hi = d->Ho; p.Wi = d->Wo; p.in_ld = d->out_ld; p.Cin = d->Cout;
    p.sD = p.sH = p.sW = 1;
    p.nTd = g.nt; p.nTh = g.nh; p.nTw = g.nw;
    // tap a (kernel index k = k0 + a*s) reads dy at  g + (r + p - k)/s  =  g + off0 - a
    p.off0d = (g.rt + d->pT - g.k0t) / d->sT;
    p.off0h = (g.rh + d->pH - g.k0h) / d->sH;
    p.off0w = (g.rw + d->pW - g.k0w) / d->sW;
    p.offstep = -1;
    p.K = g.nt * g.nh * g.nw * d->Cout;
    p.Kld = Kld;
    const unsigned long long xb = (unsigned long long)d->N * d->Do * d->Ho * d->Wo * d->out_ld * 4ull;
    const unsigned long long wb = (unsigned long long)d->Cin * Kld * 4ull;
    p.x_bytes = (unsigned)xb;
    p.w_bytes = (unsigned)wb;
    return (d->Cout % 4 == 0) && (d->out_ld % 4 == 0) && rsp_aligned16(dy) && rsp_aligned16(p.w) && xb < (1ull << 32) &&
                 wb < (1ull << 32) && g.nt <= 8 && g.nh <= 8 && g.nw <= 8;
}

}  // namespace

extern "C" {


size_t rsp_conv3d_packed_fwd_elems(const rsp_conv3d_desc* d) {
  if (!desc_ok(d)) return 0;
  if (rsp_stem_applicable(d)) return rsp_stem_packed_elems(d);
  const size_t K = (size_t)d->kT * d->kH * d->kW * d->Cin;
  return (size_t)d->Cout * rsp_align_up(K, 4);
}

int rsp_conv3d_pack_fwd(const rsp_conv3d_desc* d, const float* w_ref, float* w_packed, void* stream) {
  RSP_REQUIRE(desc_ok(d), "rsp_conv3d_pack_fwd: bad descriptor");
  RSP_REQUIRE(w_ref && w_packed, "rsp_conv3d_pack_fwd: null pointer");
  if (rsp_stem_applicable(d)) return rsp_stem_pack(d, w_ref, w_packed, (hipStream_t)stream);
  PackParams p;
  p.w = w_ref; p.out = w_packed;
  p.Cout = d->Cout; p.Cin = d->Cin; p.kT = d->kT; p.kH = d->kH; p.kW = d->kW;
  p.transpose = 0; p.O = d->Cout; p.C = d->Cin;
  p.Kld = (int)rsp_align_up((size_t)d->kT * d->kH * d->kW * d->Cin, 4);
  p.nTd = d->kT; p.nTh = d->kH; p.nTw = d->kW;
  p.k0d = p.k0h = p.k0w = 0;
  p.kstepd = p.ksteph = p.kstepw = 1;
  const long long total = (long long)p.O * p.Kld;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
  return rsp_check_launch("pack_weight_kernel");
}

int32_t rsp_conv3d_stat_tiles(const rsp_conv3d_desc* d) {
  if (!desc_ok(d)) return 0;
  if (rsp_stem_applicable(d)) return rsp_stem_tiles(d);
  return rsp_cdiv((long long)d->N * d->Do * d->Ho * d->Wo, 128);
}

size_t rsp_conv3d_fwd_workspace(const rsp_conv3d_desc* d) {
  if (!desc_ok(d)) return 0;
  if (rsp_stem_applicable(d)) return 0;
  return igemm_partial_bytes((long long)d->N * d->Do * d->Ho * d->Wo, d->Cout, d->kT * d->kH * d->kW * d->Cin);
}

int rsp_conv3d_fwd(const rsp_conv3d_desc* d, const float* x, const float* w_packed, const float* bias, float* y,
                   float* stat_partials, void* workspace, size_t workspace_bytes, void* stream) {
  RSP_REQUIRE(desc_ok(d), "rsp_conv3d_fwd: bad descriptor");
  RSP_REQUIRE(x && w_packed && y, "rsp_conv3d_fwd: null pointer");
  RSP_REQUIRE(rsp_aligned16(w_packed), "rsp_conv3d_fwd: packed weight must be 16-byte aligned");
  rsp_note_reset();
  if (rsp_stem_applicable(d)) return rsp_stem_fwd(d, x, w_packed, bias, y, stat_partials, (hipStream_t)stream);
  IgemmParams p;
  const bool vec4 = fill_fwd_params(d, x, w_packed, bias, y, stat_partials, p);
  return run_igemm(p, vec4, workspace, workspace_bytes, (hipStream_t)stream);
}

// Share of the algorithmic multiply-adds (padded taps included) that the kernels launched for this descriptor execute
// after skipping the K chunks / row chunks that are zero padding for a whole tile.  which: 0 forward, 1 dgrad, 2 wgrad.
double rsp_conv3d_executed_fraction(const rsp_conv3d_desc* d, int which) {
  if (!desc_ok(d)) return 1.0;
  float* const al = reinterpret_cast<float*>(uintptr_t(1) << 20);      // stands for any 16-byte aligned tensor
  if (which == 2) return rsp_wgrad_executed_fraction(d);
  if (which == 0) {
    if (rsp_stem_applicable(d)) return 1.0;
    IgemmParams p;
    const bool vec4 = fill_fwd_params(d, al, al, nullptr, al, nullptr, p);
    return igemm_live_fraction(p, vec4);
  }
  double live = 0.0, tot = 0.0;
  // (the tap-major classes that share one igemm_multi_kernel launch run the tile kernels' walk, whatever their size: dgrad_run)
  const bool multi = strstr(rsp_conv3d_kernel_name(d, 1), "igemm_multi_kernel") != nullptr;
  for (int c = 0; c < d->sT * d->sH * d->sW; ++c) {
    const DgradClass g = dgrad_class(d, c);
    if (g.nt * g.nh * g.nw == 0 || g.Gd * g.Gh * g.Gw == 0) continue;
    IgemmParams p;
    const bool vec4 = fill_dgrad_class_params(d, g, al, al, al, p);
    const double wgt = (double)p.M * p.K;
    live += wgt * igemm_live_fraction(p, vec4, !(multi && !k_slice_major(d->Cout, g.nt * g.nh * g.nw)));
    tot += wgt;
  }
  return tot > 0 ? live / tot : 1.0;
}

// Host evaluation of the kernels' constant division (same magic numbers, same multiply-high + shift): lets the CPU test suite
// check the derivation over the whole 31-bit range without a GPU.  Returns n / d as the device code would compute it.
int rsp_fastdiv_check(int d, int n) {
  const FastDiv f = fastdiv_make(d);
  return f.mul ? (int)((unsigned)(((unsigned long long)(unsigned)n * f.mul) >> 32) >> f.shr) : n;
}

// Name of the kernel template the library dispatches for this descriptor (16-byte aligned tensors assumed), so that a
// profiler summary row can be matched to a launch without parsing mangled names.  which: 0 forward, 1 dgrad, 2 wgrad.
const char* rsp_conv3d_kernel_name(const rsp_conv3d_desc* d, int which) {
  if (!desc_ok(d)) return "invalid";
  if (which == 2) return rsp_wgrad_kernel_name(d);
  if (which == 0 && rsp_stem_applicable(d)) return rsp_stem_kernel_name(d);
  const int cin = which == 0 ? d->Cin : d->Cout, ld = which == 0 ? d->in_ld : d->out_ld, cols = which == 0 ? d->Cout : d->Cin;
  const bool vec4 = cin % 4 == 0 && ld % 4 == 0 && d->kT <= 8 && d->kH <= 8 && d->kW <= 8;
  if (which == 1 && vec4 && d->sT * d->sH * d->sW > 1 && plan_segments(cols).n == 1) {
    // the tap-major stride-parity classes of a small or mid-sized strided dgrad share one launch, which runs first (dgrad_run)
    int nm = 0;
    long long tiles = 0;
    for (int c = 0; c < d->sT * d->sH * d->sW; ++c) {
      const DgradClass g = dgrad_class(d, c);
      if (g.nt * g.nh * g.nw == 0 || g.Gd * g.Gh * g.Gw == 0 || k_slice_major(d->Cout, g.nt * g.nh * g.nw)) continue;
      ++nm;
      tiles += (long long)rsp_cdiv(d->N * g.Gd * g.Gh * g.Gw, 128) * rsp_cdiv(cols, tile_bn(cols));
    }
    if (nm >= 2 && nm <= MAX_MULTI && ((tiles >= 512 && tiles <= 4096) || (tiles < 512 && multi_split_enabled() && cols % 4 == 0 &&
                                                                         (which == 0 ? d->out_ld : d->in_ld) % 4 == 0))) {
      switch (tile_bn(cols)) {
        case 160: return "igemm_multi_kernel<128, 160, 4, 1, 4, 2>";
        case 128: return "igemm_multi_kernel<128, 128, 2, 2, 4, 2>";
        case 96: return "igemm_multi_kernel<128, 96, 4, 1, 4, 2>";
        case 64: return "igemm_multi_kernel<128, 64, 2, 2, 4, 2>";
        default: return "igemm_multi_kernel<128, 32, 4, 1, 4, 2>";
      }
    }
  }
  {
    // small launches: the direct kernel (run_igemm).  dgrad: the first non-empty stride-parity class decides, as below
    long long Mrows = (long long)d->N * d->Do * d->Ho * d->Wo;
    int Kk = d->kT * d->kH * d->kW * d->Cin;
    if (which == 1) {
      for (int c = 0; c < d->sT * d->sH * d->sW; ++c) {
        const DgradClass g = dgrad_class(d, c);
        if (g.nt * g.nh * g.nw == 0 || g.Gd * g.Gh * g.Gw == 0) continue;
        Mrows = (long long)d->N * g.Gd * g.Gh * g.Gw;
        Kk = g.nt * g.nh * g.nw * d->Cout;
        break;
      }
    }
    if (direct_applies(Mrows, cols, cin, Kk, vec4)) return "igemm_direct_kernel";
  }
  // (a convolution that runs as two column segments is named after the first, wider one)
  int bn = tile_bn(plan_segments(cols).width[0]);
  if (!vec4 && (bn == 160 || bn == 96)) bn = 128;
  long long Mrows = (long long)d->N * d->Do * d->Ho * d->Wo;
  int kchunks = rsp_cdiv((long long)d->kT * d->kH * d->kW * d->Cin, BK);
  if (which == 1) {      // the first non-empty stride-parity class (its launch is the one rsp_last_conv_kernel reports)
    for (int c = 0; c < d->sT * d->sH * d->sW; ++c) {
      const DgradClass g = dgrad_class(d, c);
      if (g.nt * g.nh * g.nw == 0 || g.Gd * g.Gh * g.Gw == 0) continue;
      Mrows = (long long)d->N * g.Gd * g.Gh * g.Gw;
      kchunks = rsp_cdiv((long long)g.nt * g.nh * g.nw * d->Cout, BK);
      break;
    }
  }
  if (vec4) {
    const int nb = narrow_bn(Mrows, cols, kchunks);
    if (nb) bn = nb;
  }
  // spelled as rocprofv3 prints the demangled instance (minus namespace and argument list)
  const bool ks = vec4 && ((which == 0 && k_slice_major(d->Cin, d->kT * d->kH * d->kW)) ||
                           (which == 1 && d->sT * d->sH * d->sW == 1 && k_slice_major(d->Cout, d->kT * d->kH * d->kW)));
  // output addressable with 32-bit offsets: the persistent instances (launch_persist)
  const unsigned long long out_b = which == 0 ? (unsigned long long)d->N * d->Do * d->Ho * d->Wo * d->out_ld * 4ull
                                              : (unsigned long long)d->N * d->Di * d->Hi * d->Wi * d->in_ld * 4ull;
  if (vec4 && persist_enabled() && out_b < 0x7ffffff0ull && tall_tiles(Mrows, bn, vec4))
    return ks ? "igemm_persist_kernel<256, 64, 4, 1, true, 3>" : "igemm_persist_kernel<256, 64, 4, 1, false, 3>";
  const bool long_tm128 = !ks && bn == 128 && kchunks >= 48;      // as in run_igemm_segment
  if (vec4 && persist_enabled() && !long_tm128 && out_b < 0x7ffffff0ull) {
    if (ks) {
      if (bn == 128 && two_level_min_chunks() > 0 && kchunks >= two_level_min_chunks()) return "igemm_persist_fold_kernel<128, 128, 2, 2, true, 16>";
      switch (bn) {
        case 160: return half_block_cols(cols) ? "igemm_persist_kernel<128, 144, 4, 1, true, 2>" : "igemm_persist_kernel<128, 160, 4, 1, true, 2>";
        case 128: return "igemm_persist_kernel<128, 128, 2, 2, true, 3>";
        case 96: return "igemm_persist_kernel<128, 96, 4, 1, true, 3>";
        case 64: return "igemm_persist_kernel<128, 64, 2, 2, true, 4>";
        default: return "igemm_persist_kernel<128, 32, 4, 1, true, 4>";
      }
    }
    switch (bn) {
      case 160: return half_block_cols(cols) ? "igemm_persist_kernel<128, 144, 4, 1, false, 2>" : "igemm_persist_kernel<128, 160, 4, 1, false, 2>";
      case 128: return "igemm_persist_kernel<128, 128, 2, 2, false, 3>";
      case 96: return "igemm_persist_kernel<128, 96, 4, 1, false, 3>";
      case 64: return "igemm_persist_kernel<128, 64, 2, 2, false, 4>";
      default: return "igemm_persist_kernel<128, 32, 4, 1, false, 4>";
    }
  }
  if (ks) {
    switch (bn) {
      case 160: return "igemm_ks_kernel<128, 160, 4, 1>";
      case 128: return "igemm_ks_kernel<128, 128, 2, 2>";
      case 96: return "igemm_ks_kernel<128, 96, 4, 1>";
      case 64: return "igemm_ks_kernel<128, 64, 2, 2>";
      default: return "igemm_ks_kernel<128, 32, 4, 1>";
    }
  }
  if (vec4) {
    switch (bn) {
      case 160: return "igemm_kernel<128, 160, 4, 1, 4, 2>";
      case 128: return "igemm_kernel<128, 128, 2, 2, 4, 2>";
      case 96: return "igemm_kernel<128, 96, 4, 1, 4, 2>";
      case 64: return "igemm_kernel<128, 64, 2, 2, 4, 2>";
      default: return "igemm_kernel<128, 32, 4, 1, 4, 2>";
    }
  }
  return bn == 128 ? "igemm_kernel<128, 128, 2, 2, 1, 2>" : (bn == 64 ? "igemm_kernel<128, 64, 2, 2, 1, 2>" : "igemm_kernel<128, 32, 4, 1, 1, 2>");
}

// ---- dgrad ----------------------------------------------------------------------------------------------------
// dx[i] = sum_{k : (i + p - k) % s == 0} dy[(i + p - k)/s] * w[k]; positions are split into s^3 parity classes,
// each a dense stride-1 "conv" over dy with its own tap subset; class c writes the strided sub-grid i = s*g + r.
// partial bytes a descriptor's dgrad may need on the shared K-split launch (alignment-dependent conditions assumed true)
static size_t dgrad_multi_split_bytes(const rsp_conv3d_desc* d) {
  const int nclass = d->sT * d->sH * d->sW;
  if (nclass < 2 || plan_segments(d->Cin).n != 1) return 0;
  long long Ms[MAX_MULTI], tiles_all = 0;
  int tl[MAX_MULTI], nc[MAX_MULTI], n = 0;
  const int bn = tile_bn(d->Cin);
  for (int c = 0; c < nclass; ++c) {
    const DgradClass g = dgrad_class(d, c);
    if (g.nt * g.nh * g.nw == 0 || g.Gd * g.Gh * g.Gw == 0 || k_slice_major(d->Cout, g.nt * g.nh * g.nw)) continue;
    if (n >= MAX_MULTI) return 0;
    Ms[n] = (long long)d->N * g.Gd * g.Gh * g.Gw;
    tl[n] = rsp_cdiv(Ms[n], 128) * rsp_cdiv(d->Cin, bn);
    nc[n] = rsp_cdiv((long long)g.nt * g.nh * g.nw * d->Cout, BK);
    tiles_all += tl[n];
    ++n;
  }
  if (n < 2 || tiles_all >= 512) return 0;
  return plan_multi_split(n, Ms, tl, nc, d->Cin).bytes;
}

size_t rsp_conv3d_dgrad_workspace(const rsp_conv3d_desc* d) {
  if (!desc_ok(d)) return 0;
  size_t part = 0;
  const int nclass = d->sT * d->sH * d->sW;
  for (int c = 0; c < nclass; ++c) {
    const DgradClass g = dgrad_class(d, c);
    if (g.nt * g.nh * g.nw == 0 || g.Gd * g.Gh * g.Gw == 0) continue;
    const size_t b = igemm_partial_bytes((long long)d->N * g.Gd * g.Gh * g.Gw, d->Cin, g.nt * g.nh * g.nw * d->Cout);
    part = b > part ? b : part;
  }
  const size_t mb = dgrad_multi_split_bytes(d);
  part = mb > part ? mb : part;
  return dgrad_wpack_bytes(d) + part;
}


// Fill the re-pack description of one dgrad stride-parity class (source dims = the descriptor's: no padding).
static void dgrad_pack_params(const rsp_conv3d_desc* d, const DgradClass& g, const float* w_ref, float* out, PackParams& pk) {
  pk.w = w_ref; pk.out = out;
  pk.Cout = d->Cout; pk.Cin = d->Cin; pk.kT = d->kT; pk.kH = d->kH; pk.kW = d->kW;
  pk.transpose = 1; pk.O = d->Cin; pk.C = d->Cout;
  pk.Kld = (int)rsp_align_up((size_t)g.nt * g.nh * g.nw * d->Cout, 4);
  pk.nTd = g.nt; pk.nTh = g.nh; pk.nTw = g.nw;
  pk.k0d = g.k0t; pk.k0h = g.k0h; pk.k0w = g.k0w;
  pk.kstepd = d->sT; pk.ksteph = d->sH; pk.kstepw = d->sW;
}

// rows x C floats at a pitch of ld floats := 0.  A KERNEL, not hipMemsetAsync: captured into a linear HIP graph (rspnet_amd/graph_step.py)
// the memset node was not reliably ordered against the kernel nodes around it on this stack — R3D-18's shortcut input gradients (1x1x1,
// stride 2: the one place that needs the fill) came out wrong in a few replays out of a hundred, the loss of the step untouched and every
// gradient below the block off (profiles/r06/experiments_r6.txt r6race; eager issue and graphs with forks inside were never affected).
__global__ __launch_bounds__(256) void zero_rows_kernel(float* __restrict__ p, long long total, int cw, int ld, int vec) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += 256ll * gridDim.x) {
    const long long row = i / cw;
    const int c = (int)(i - row * cw);
    if (vec) *reinterpret_cast<float4*>(p + row * ld + c * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    else p[row * ld + c] = 0.f;
  }
}
static int zero_rows(float* p, size_t rows, int C, int ld, hipStream_t s) {
  const int vec = (C % 4 == 0 && ld % 4 == 0 && rsp_aligned16(p)) ? 1 : 0;
  const int cw = vec ? C / 4 : C;
  const long long total = (long long)rows * cw;
  if (total == 0) return RSP_OK;
  const long long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(zero_rows_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0, s, p, total, cw, ld, vec);
  return rsp_check_launch("zero_rows_kernel");
}

// The GEMM launches of dgrad over already packed per-class weights (class c's block follows class c-1's, O x Kld floats each).
static int dgrad_run(const rsp_conv3d_desc* d, const float* dy, const float* wpk, float* dx, void* part, size_t part_bytes,
                     hipStream_t s) {
  const int nclass = d->sT * d->sH * d->sW;
  bool any_empty = false;
  for (int c = 0; c < nclass; ++c) {
    const DgradClass g = dgrad_class(d, c);
    if (g.Gd * g.Gh * g.Gw != 0 && g.nt * g.nh * g.nw == 0) any_empty = true;
  }
  if (any_empty) {
    // some input positions receive no gradient at all (kernel smaller than stride, e.g. 1x1x1 stride 2)
    const size_t rows = (size_t)d->N * d->Di * d->Hi * d->Wi;
    const int rc = zero_rows(dx, rows, d->Cin, d->in_ld, s);
    if (rc != RSP_OK) return rc;
  }
  // per-class GEMM descriptions
  IgemmParams cls[64];
  bool cvec[64];
  int ncls = 0, max_taps = 0;
  size_t woff = 0;
  for (int c = 0; c < nclass; ++c) {
    const DgradClass g = dgrad_class(d, c);
    if (g.nt * g.nh * g.nw == 0 || g.Gd * g.Gh * g.Gw == 0) continue;
    const int Kld = (int)rsp_align_up((size_t)g.nt * g.nh * g.nw * d->Cout, 4);
    IgemmParams& p = cls[ncls];
    cvec[ncls] = fill_dgrad_class_params(d, g, dy, wpk + woff, dx, p);
    max_taps = max_taps > g.nt * g.nh * g.nw ? max_taps : g.nt * g.nh * g.nw;
    woff += (size_t)d->Cin * Kld;
    ++ncls;
  }
  // Small classes of a strided convolution share ONE launch (igemm_multi_kernel): the classes on the LDS-DMA path with the
  // tap-major K walk (one walk per launch; a class long enough for the slice-major walk — R3D-18 layer4.0's 8-tap class — runs on
  // its own behind them), a single column segment, and few enough tiles that separate launches would each leave most of the
  // machine idle.
  int mi[MAX_MULTI + 1], nm = 0;      // the classes that share the launch
  bool in_multi[64] = {false};
  long long tiles_all = 0;
  const int bn = tile_bn(d->Cin);
  if (ncls >= 2 && plan_segments(d->Cin).n == 1) {
    for (int i = 0; i < ncls; ++i)
      if (cvec[i] && !k_slice_major(cls[i].Cin, cls[i].nTd * cls[i].nTh * cls[i].nTw) && nm <= MAX_MULTI) mi[nm++] = i;
    if (nm > MAX_MULTI) nm = 0;
    for (int j = 0; j < nm; ++j) tiles_all += (long long)rsp_cdiv(cls[mi[j]].M, 128) * rsp_cdiv(d->Cin, bn);
  }
  const bool multi = nm >= 2;
  // (above a few thousand tiles each class fills the machine on its own)
  // Few tiles (R3D-18 layer3.0 / layer4.0: 8 classes of 25 / 8 tiles): the classes still share ONE launch, each cut along K by one
  // common plan, and ONE batched reduce sums all of them (round 5; until then each class ran its own K-split launch + reduce: 16
  // dependent launches of 10-15 us for 3.6 GFLOP)
  bool split_multi = multi && multi_split_enabled() && tiles_all < 512 && d->Cin % 4 == 0 && d->in_ld % 4 == 0 && rsp_aligned16(dx);
  MultiSplit msp;
  if (split_multi) {
    long long Ms[MAX_MULTI];
    int tl[MAX_MULTI], nc[MAX_MULTI];
    for (int j = 0; j < nm; ++j) {
      Ms[j] = cls[mi[j]].M;
      tl[j] = rsp_cdiv(cls[mi[j]].M, 128) * rsp_cdiv(d->Cin, bn);
      nc[j] = rsp_cdiv(cls[mi[j]].K, BK);
    }
    msp = plan_multi_split(nm, Ms, tl, nc, d->Cin);
    if (msp.bytes > part_bytes || (msp.bytes && !part)) split_multi = false;
  }
  if (multi && ((tiles_all >= 512 && tiles_all <= 4096) || split_multi)) {
    const float* zero_page = igemm_zero_page();
    if (!zero_page) {
      rsp_set_error("hipGetSymbolAddress(g_zero) failed");
      return RSP_ELAUNCH;
    }
    IgemmMulti m;
    memset(&m, 0, sizeof m);
    m.n = nm;
    int multi_taps = 0;
    for (int j = 0; j < nm; ++j) {
      IgemmParams& p = cls[mi[j]];
      in_multi[mi[j]] = true;
      const int taps = p.nTd * p.nTh * p.nTw;
      multi_taps = multi_taps > taps ? multi_taps : taps;
      p.zero = zero_page;
      p.stat_ld = p.Cout;
      p.nchunks = rsp_cdiv(p.K, BK);
      fill_fastdiv(p);
      p.m_tiles = rsp_cdiv(p.M, 128);
      p.n_tiles = rsp_cdiv(p.Cout, bn);
      p.full_tiles = p.m_tiles * p.n_tiles;
      p.splitk = 1;
      p.nbuf = 2;
      p.chunks_per_split = p.nchunks;
      p.tail_row0 = p.m_tiles * 128;
      int units = p.full_tiles;
      if (split_multi && msp.S[j] > 1) {      // every tile of this class is a "tail" tile cut into S slices (igemm_body)
        p.full_tiles = 0;
        p.splitk = msp.S[j];
        p.chunks_per_split = msp.cps[j];
        p.tail_row0 = 0;
        p.partial = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(part) + msp.off[j]);
        units = p.m_tiles * p.n_tiles * p.splitk;
      }
      m.p[j] = p;
      m.start[j + 1] = m.start[j] + units;
    }
    int rc;
    switch (bn) {
      case 160: rc = launch_multi_cfg<128, 160, 4, 1, 4>(m, multi_taps, s); break;
      case 128: rc = launch_multi_cfg<128, 128, 2, 2, 4>(m, multi_taps, s); break;
      case 96: rc = launch_multi_cfg<128, 96, 4, 1, 4>(m, multi_taps, s); break;
      case 64: rc = launch_multi_cfg<128, 64, 2, 2, 4>(m, multi_taps, s); break;
      default: rc = launch_multi_cfg<128, 32, 4, 1, 4>(m, multi_taps, s); break;
    }
    if (rc != RSP_OK) return rc;
    if (split_multi && msp.bytes != 0) {
      ReduceMulti rm;
      memset(&rm, 0, sizeof rm);
      rm.n = nm;
      int max_mt = 1;
      for (int j = 0; j < nm; ++j) {
        fill_reduce(rm.r[j], m.p[j]);
        if (m.p[j].splitk > 1 && m.p[j].m_tiles > max_mt) max_mt = m.p[j].m_tiles;
      }
      hipLaunchKernelGGL(splitk_reduce_vec_multi_kernel, dim3(max_mt, rsp_cdiv(d->Cin, 64), nm), dim3(256), 0, s, rm);
      rc = rsp_check_launch("splitk_reduce_vec_multi_kernel");
      if (rc != RSP_OK) return rc;
    }
  }
  (void)max_taps;
  // the rest one by one (same stream: the workspace is free again when each starts)
  for (int i = 0; i < ncls; ++i) {
    if (in_multi[i]) continue;
    int rc = run_igemm(cls[i], cvec[i], part, part_bytes, s);
    if (rc != RSP_OK) return rc;
  }
  return RSP_OK;
}

int rsp_conv3d_dgrad(const rsp_conv3d_desc* d, const float* dy, const float* w_ref, float* dx, void* workspace,
                     size_t workspace_bytes, void* stream) {
  RSP_REQUIRE(desc_ok(d), "rsp_conv3d_dgrad: bad descriptor");
  RSP_REQUIRE(dy && w_ref && dx && workspace, "rsp_conv3d_dgrad: null pointer");
  RSP_REQUIRE(rsp_aligned16(workspace), "rsp_conv3d_dgrad: workspace must be 16-byte aligned");
  rsp_note_reset();
  hipStream_t s = (hipStream_t)stream;
  const int nclass = d->sT * d->sH * d->sW;
  unsigned char* wsp = reinterpret_cast<unsigned char*>(workspace);
  const size_t wp_bytes = dgrad_wpack_bytes(d);
  if (workspace_bytes < wp_bytes) {
    rsp_set_error("rsp_conv3d_dgrad: workspace too small");
    return RSP_EWORKSPACE;
  }
  float* wpk = reinterpret_cast<float*>(wsp);
  size_t woff = 0;
  for (int c = 0; c < nclass; ++c) {
    const DgradClass g = dgrad_class(d, c);
    if (g.nt * g.nh * g.nw == 0 || g.Gd * g.Gh * g.Gw == 0) continue;
    PackParams pk;
    dgrad_pack_params(d, g, w_ref, wpk + woff, pk);
    const long long total = (long long)pk.O * pk.Kld;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, s, pk);
    int rc = rsp_check_launch("pack_weight_kernel(dgrad)");
    if (rc != RSP_OK) return rc;
    woff += (size_t)pk.O * pk.Kld;
  }
  return dgrad_run(d, dy, wpk, dx, wsp + wp_bytes, workspace_bytes - wp_bytes, s);
}

size_t rsp_conv3d_packed_dgrad_elems(const rsp_conv3d_desc* d) {
  if (!desc_ok(d)) return 0;
  return dgrad_wpack_bytes(d) / sizeof(float);
}

int rsp_conv3d_dgrad_packed(const rsp_conv3d_desc* d, const float* dy, const float* w_packed, float* dx, void* workspace,
                            size_t workspace_bytes, void* stream) {
  RSP_REQUIRE(desc_ok(d), "rsp_conv3d_dgrad_packed: bad descriptor");
  RSP_REQUIRE(dy && w_packed && dx, "rsp_conv3d_dgrad_packed: null pointer");
  RSP_REQUIRE(rsp_aligned16(w_packed) && (!workspace || rsp_aligned16(workspace)),
              "rsp_conv3d_dgrad_packed: packed weight / workspace must be 16-byte aligned");
  rsp_note_reset();
  return dgrad_run(d, dy, w_packed, dx, workspace, workspace_bytes, (hipStream_t)stream);
}

// ---- batched re-pack -------------------------------------------------------------------------------------------------
int32_t rsp_conv3d_pack_jobs(const rsp_conv3d_desc* d, int32_t which, int32_t Cout_src, int32_t Cin_src, const float* w_ref,
                             float* w_packed, rsp_pack_job* jobs, int32_t max_jobs) {
  if (!desc_ok(d) || !w_ref || !w_packed || !jobs || Cout_src <= 0 || Cin_src <= 0 || Cout_src > d->Cout || Cin_src > d->Cin) {
    rsp_set_error("rsp_conv3d_pack_jobs: bad argument");
    return RSP_EINVAL;
  }
  auto fill = [&](rsp_pack_job& j, const PackParams& pk) {
    memset(&j, 0, sizeof j);
    j.src = pk.w; j.dst = pk.out; j.kind = 0;
    j.Cout_src = Cout_src; j.Cin_src = Cin_src; j.kT = d->kT; j.kH = d->kH; j.kW = d->kW;
    j.transpose = pk.transpose; j.O = pk.O; j.C = pk.C; j.Kld = pk.Kld;
    j.nTd = pk.nTd; j.nTh = pk.nTh; j.nTw = pk.nTw; j.k0d = pk.k0d; j.k0h = pk.k0h; j.k0w = pk.k0w;
    j.kstepd = pk.kstepd; j.ksteph = pk.ksteph; j.kstepw = pk.kstepw;
    j.ntaps = pk.nTd * pk.nTh * pk.nTw;
    j.total = (long long)pk.O * pk.Kld;
    j.blocks = pk.O;
  };
  if (which == 0) {
    if (max_jobs < 1) return RSP_EINVAL;
    if (rsp_stem_applicable(d)) {
      rsp_pack_job& j = jobs[0];
      memset(&j, 0, sizeof j);
      j.src = w_ref; j.dst = w_packed; j.kind = 1;
      j.Cout_src = Cout_src; j.Cin_src = Cin_src; j.kT = d->kT; j.kH = d->kH; j.kW = d->kW;
      j.ntaps = d->kT * d->kH * d->kW;
      j.total = (long long)rsp_stem_packed_elems(d);
      j.blocks = (int32_t)((j.total + 4095) / 4096);
      rsp_stem_note_packed(w_packed, Cin_src <= 3);     // (the re-pack zero-fills channel 3: its k-step is skipped)
      return 1;
    }
    PackParams pk;
    pk.w = w_ref; pk.out = w_packed;
    pk.transpose = 0; pk.O = d->Cout; pk.C = d->Cin;
    pk.Kld = (int)rsp_align_up((size_t)d->kT * d->kH * d->kW * d->Cin, 4);
    pk.nTd = d->kT; pk.nTh = d->kH; pk.nTw = d->kW;
    pk.k0d = pk.k0h = pk.k0w = 0;
    pk.kstepd = pk.ksteph = pk.kstepw = 1;
    fill(jobs[0], pk);
    return 1;
  }
  int n = 0;
  size_t woff = 0;
  const int nclass = d->sT * d->sH * d->sW;
  for (int c = 0; c < nclass; ++c) {
    const DgradClass g = dgrad_class(d, c);
    if (g.nt * g.nh * g.nw == 0 || g.Gd * g.Gh * g.Gw == 0) continue;
    if (n >= max_jobs) return RSP_EINVAL;
    PackParams pk;
    dgrad_pack_params(d, g, w_ref, w_packed + woff, pk);
    fill(jobs[n++], pk);
    woff += (size_t)pk.O * pk.Kld;
  }
  return n;
}

void rsp_conv3d_pack_forget(const void* w_packed) { rsp_stem_forget_packed(w_packed); }

int rsp_conv3d_set_option(const char* name, int32_t value) {
  if (name && !strcmp(name, "narrow_max_tiles")) {
    const int prev = narrow_max_tiles();
    g_narrow_max_tiles.store(value < 0 ? -1 : value, std::memory_order_relaxed);
    return prev;
  }
  if (name && !strcmp(name, "narrow32_max_units")) {
    const int prev = narrow32_max_units();
    g_narrow32_max_units.store(value < 0 ? -1 : value, std::memory_order_relaxed);
    return prev;
  }
  if (name && !strcmp(name, "two_level_min_chunks")) {
    const int prev = two_level_min_chunks();
    g_two_level_min_chunks.store(value < 0 ? -1 : value, std::memory_order_relaxed);
    return prev;
  }
  if (name && !strcmp(name, "tall_min_tiles")) {
    const int prev = tall_min_tiles();
    g_tall_min_tiles.store(value < 0 ? -1 : value, std::memory_order_relaxed);
    return prev;
  }
  rsp_set_error("rsp_conv3d_set_option: unknown option");
  return RSP_EINVAL;
}

int rsp_pack_run(const rsp_pack_job* jobs_device, int32_t n_jobs, int32_t max_blocks, void* stream) {
  RSP_REQUIRE(jobs_device && n_jobs > 0 && n_jobs <= 65535 && max_blocks > 0, "rsp_pack_run: bad argument");
  hipLaunchKernelGGL(pack_batch_kernel, dim3((unsigned)max_blocks, n_jobs), dim3(256), 0, (hipStream_t)stream, jobs_device);
  return rsp_check_launch("pack_batch_kernel");
}

}  // extern "C"
