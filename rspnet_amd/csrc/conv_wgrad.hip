// Weight gradient of nn.Conv3d on the exact-fp32 matrix pipe (autograd conv backward-weight for every conv of
// the query encoder; reference trigger: loss.backward(), pretrain.py:164).
//
//   dW[co, k] = sum_rows dy[row, co] * im2col(x)[row, k],   k = tap*Cin + ci
//
// GEMM view: M' = Cout, N' = K = taps*Cin, reduction over rows = N*Do*Ho*Wo output positions.  Both operands have
// the reduction index as their slow (row) dimension, so LDS tiles are [row][co] and [row][k]; an MFMA fragment is
// a ds_read_b32 with 32 consecutive lanes on 32 consecutive floats (conflict-free).
// The row range is split over grid.y; slabs [split][Cout][Kld] are summed in a fixed order by a second kernel
// that also transposes to the reference layout (Cout,Cin,kT,kH,kW) => bitwise run-to-run reproducible.
#include "common.h"
#include "conv_stem.h"

namespace {

constexpr int RK = 32;  // rows per chunk
constexpr int MAX_TAPS = 343;

__device__ __attribute__((aligned(64))) float g_wzero[16];

struct WgradParams {
  const float* __restrict__ x;
  const float* __restrict__ dy;
  float* __restrict__ partial;  // [splitm][Cout][Kld]
  int M, Gd, Gh, Gw;            // output grid (rows)
  int Di, Hi, Wi, in_ld, Cin;
  int sD, sH, sW;
  int kT, kH, kW, pT, pH, pW;
  int dy_ld, Cout;
  int K, Kld;
  int rows_per_split, splitm;
  int co_tiles, k_tiles;
  const float* __restrict__ zero;  // >= 64 B of zeros
  unsigned x_bytes, dy_bytes;      // extents for the buffer descriptors of the DMA variant
  const uint2* __restrict__ rowgeom;  // [M] row geometry (DMA variant)
  int skip_pad;                       // DMA variant: skip the row chunks whose frames make this k tile's depth taps padding
  FastDiv dP, dGd;                    // rows per frame (Gh * Gw), frames per sample
  int kh0, kh1;                       // k tiles [kh0, kh1) never skip ("full"); the others skip some frames ("short"): wgrad_unit_lpt
  int lpt;                            // full units first (launches of two or more rounds; a single round gains nothing and loses L2 locality)
};

// Workgroup -> (slab, tile).  Every (co, k) tile of a row slab reads the slab's dy rows, and tiles of neighbouring taps read the
// same x rows; launched as a (tiles, slabs) grid the tiles of one slab were dealt round-robin over the 8 XCDs, so each XCD's L2
// fetched the whole slab for itself and the launch left 7-9x its algorithmic bytes at the L2s (conv2: 11.7 GB for 1.23 GB).
// The grid is now 1-D over the slab-major unit list u = slab * tiles + tile (tile co-major: same dy columns, neighbouring taps)
// and every XCD walks ONE contiguous chunk of that list (rsp_xcd_remap: workgroup L runs on XCD L % 8): the workgroups an XCD
// has in flight at any time are consecutive tiles of the same slab, advancing chunk by chunk together, so the slab's dy and x
// rows come through that L2 once.  Every XCD gets the same number of units (+-1) whatever the tile count — a first version that
// dealt whole per-slab teams to the XCDs cost up to 25 % on layers whose tile count does not divide the XCD's slots.
struct WUnit {
  int z, tile;
};
__device__ __forceinline__ WUnit wgrad_unit(int L, int nwg, int tiles) {
  int u = rsp_xcd_remap(L, nwg);
  WUnit w;
  w.z = u / tiles;
  w.tile = u - w.z * tiles;
  return w;
}

// With skipped frames the units are no longer equally long: the k tiles of the kernel's edge depths run 1 - 1/T of the row
// chunks.  A launch is only ~4 units deep per workgroup slot, so dealt in list order the slots that happen to draw four full
// units set the finish time and the skipped work buys nothing (measured: C3D conv3b 5.29 -> 5.27 ms).  Every XCD therefore walks
// its contiguous share of the FULL units first and its contiguous share of the SHORT ones after it (both slab-major, so the
// L2 argument above still holds within each phase): the short units fill the ragged end of the launch.
__device__ __forceinline__ WUnit wgrad_unit_lpt(int L, int nwg, int slabs, int co_tiles, int k_tiles, int kh0, int kh1) {
  const int Hk = kh1 - kh0, Lk = k_tiles - Hk;
  const int UH = slabs * co_tiles * Hk;
  const int x = L & 7, j = L >> 3;
  const int q = nwg >> 3, r = nwg & 7, qh = UH >> 3, rh = UH & 7;
  const int hq = qh + (x < rh ? 1 : 0);
  const int base_h = x * qh + min(x, rh);
  const int base_l = x * q + min(x, r) - base_h;
  WUnit w;
  int co, k;
  if (j < hq) {
    const int i = base_h + j, per = co_tiles * Hk;
    w.z = i / per;
    const int rem = i - w.z * per;
    co = rem / Hk;
    k = kh0 + (rem - co * Hk);
  } else {
    const int i = base_l + (j - hq), per = co_tiles * Lk;
    w.z = i / per;
    const int rem = i - w.z * per;
    co = rem / Lk;
    const int kk = rem - co * Lk;
    k = kk < kh0 ? kk : kk + Hk;
  }
  w.tile = co * k_tiles + k;
  return w;
}

struct RowPos {
  int n, gd, gh, gw;
};

__device__ __forceinline__ void advance(RowPos& r, int step, int Gd, int Gh, int Gw) {
  r.gw += step;
  while (r.gw >= Gw) {
    r.gw -= Gw;
    if (++r.gh >= Gh) {
      r.gh = 0;
      if (++r.gd >= Gd) {
        r.gd = 0;
        ++r.n;
      }
    }
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool VECA, bool VECB>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradParams p) {
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int ACOLS = BM / 4, BCOLS = BN / 4;            // float4 columns per tile row
  constexpr int AROWS_PER_PASS = 256 / ACOLS, BROWS_PER_PASS = 256 / BCOLS;
  constexpr int AR = RK / AROWS_PER_PASS, BR = RK / BROWS_PER_PASS;
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* At = reinterpret_cast<float*>(smem_raw);  // [2][RK][BM]
  float* Bt = At + 2 * RK * BM;                    // [2][RK][BN]

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);   // scalar: LDS-DMA bases (M0) become SALU arithmetic
  const int l32 = lane & 31, h = lane >> 5;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  const WUnit unit = wgrad_unit(blockIdx.x, gridDim.x, p.co_tiles * p.k_tiles);
  const int tile = unit.tile;
  const int co_tile = tile / p.k_tiles, k_tile = tile - co_tile * p.k_tiles;
  const int co0 = co_tile * BM, k0 = k_tile * BN;
  const int z = unit.z;
  const int row_begin = z * p.rows_per_split;
  const int row_end = min(p.M, row_begin + p.rows_per_split);

  // ---- A side (dy): thread -> 4 consecutive co, AR rows --------------------------------------------------------
  const int acol = (t % ACOLS) * 4, arow = t / ACOLS;
  const int aco = co0 + acol;
  // ---- B side (im2col x): thread -> 4 consecutive k, BR rows ----------------------------------------------------
  const int bcol = (t % BCOLS) * 4, brow = t / BCOLS;
  int boffd[VECB ? 1 : 4], boffh[VECB ? 1 : 4], boffw[VECB ? 1 : 4], bdelta[VECB ? 1 : 4];
  bool bok[VECB ? 1 : 4];
#pragma unroll
  for (int e = 0; e < (VECB ? 1 : 4); ++e) {
    const int k = k0 + bcol + e;
    bok[e] = k < p.K;
    const int kk = bok[e] ? k : 0;
    const int tap = kk / p.Cin, ci = kk - tap * p.Cin;
    const int kw = tap % p.kW, q = tap / p.kW;
    const int kh = q % p.kH, kt = q / p.kH;
    boffd[e] = kt - p.pT;
    boffh[e] = kh - p.pH;
    boffw[e] = kw - p.pW;
    bdelta[e] = ((boffd[e] * p.Hi + boffh[e]) * p.Wi + boffw[e]) * p.in_ld + ci;
  }
  RowPos bpos[BR];
  {
    const int r = row_begin + brow;
    RowPos r0;
    r0.gw = r % p.Gw;
    int q = r / p.Gw;
    r0.gh = q % p.Gh;
    q /= p.Gh;
    r0.gd = q % p.Gd;
    r0.n = q / p.Gd;
#pragma unroll
    for (int i = 0; i < BR; ++i) {
      bpos[i] = r0;
      advance(r0, BROWS_PER_PASS, p.Gd, p.Gh, p.Gw);
    }
  }

  floatx4 areg[AR], breg[BR];
  const int st_w = RK % p.Gw, st_h = (RK / p.Gw) % p.Gh, st_d = (RK / (p.Gw * p.Gh)) % p.Gd,
            st_n = RK / (p.Gw * p.Gh * p.Gd);

  // invalid rows / taps / channels read a zero page instead of branching around the load (pure v_cndmask)
  const long long zoff_a = (reinterpret_cast<const char*>(p.zero) - reinterpret_cast<const char*>(p.dy)) / 4;
  const long long zoff_b = (reinterpret_cast<const char*>(p.zero) - reinterpret_cast<const char*>(p.x)) / 4;

  auto load_chunk = [&](int rbase) {
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const int r = rbase + arow + i * AROWS_PER_PASS;
      const long long real = (long long)r * p.dy_ld + aco;
      if (VECA) {
        const long long m = (r < row_end && aco < p.Cout) ? -1ll : 0ll;
        areg[i] = *reinterpret_cast<const floatx4*>(p.dy + ((real & m) | (zoff_a & ~m)));
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const long long m = (r < row_end && aco + e < p.Cout) ? -1ll : 0ll;
          areg[i][e] = p.dy[((real + e) & m) | (zoff_a & ~m)];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < BR; ++i) {
      const int r = rbase + brow + i * BROWS_PER_PASS;
      const RowPos& rp = bpos[i];
      const int id0 = rp.gd * p.sD, ih0 = rp.gh * p.sH, iw0 = rp.gw * p.sW;
      const long long base = ((((long long)rp.n * p.Di + id0) * p.Hi + ih0) * p.Wi + iw0) * p.in_ld;
      if (VECB) {
        const int id = id0 + boffd[0], ih = ih0 + boffh[0], iw = iw0 + boffw[0];
        const bool ok = r < row_end && bok[0] && (unsigned)id < (unsigned)p.Di && (unsigned)ih < (unsigned)p.Hi &&
                        (unsigned)iw < (unsigned)p.Wi;
        const long long m = ok ? -1ll : 0ll;
        breg[i] = *reinterpret_cast<const floatx4*>(p.x + (((base + bdelta[0]) & m) | (zoff_b & ~m)));
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int id = id0 + boffd[e], ih = ih0 + boffh[e], iw = iw0 + boffw[e];
          const bool ok = r < row_end && bok[e] && (unsigned)id < (unsigned)p.Di && (unsigned)ih < (unsigned)p.Hi &&
                          (unsigned)iw < (unsigned)p.Wi;
          const long long m = ok ? -1ll : 0ll;
          breg[i][e] = p.x[((base + bdelta[e]) & m) | (zoff_b & ~m)];
        }
      }
    }
    // next chunk: +RK rows, as a branch-free mixed-radix add (each digit step < its radix, so carries are 0/1)
#pragma unroll
    for (int i = 0; i < BR; ++i) {
      RowPos& r = bpos[i];
      r.gw += st_w;
      int c = r.gw >= p.Gw;
      r.gw -= c ? p.Gw : 0;
      r.gh += st_h + c;
      c = r.gh >= p.Gh;
      r.gh -= c ? p.Gh : 0;
      r.gd += st_d + c;
      c = r.gd >= p.Gd;
      r.gd -= c ? p.Gd : 0;
      r.n += st_n + c;
    }
  };
  auto store_chunk = [&](int buf) {
    float* a = At + buf * RK * BM;
    float* b = Bt + buf * RK * BN;
#pragma unroll
    for (int i = 0; i < AR; ++i)
      *reinterpret_cast<floatx4*>(a + (arow + i * AROWS_PER_PASS) * BM + acol) = areg[i];
#pragma unroll
    for (int i = 0; i < BR; ++i)
      *reinterpret_cast<floatx4*>(b + (brow + i * BROWS_PER_PASS) * BN + bcol) = breg[i];
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  if (row_begin < row_end) {
    load_chunk(row_begin);
    store_chunk(0);
  }
  __syncthreads();

  int buf = 0;
  for (int rb = row_begin; rb < row_end; rb += RK) {
    const bool more = rb + RK < row_end;
    if (more) load_chunk(rb + RK);
    const float* a = At + buf * RK * BM + wm * WM + l32;
    const float* b = Bt + buf * RK * BN + wn * WN + l32;
#pragma unroll
    for (int s = 0; s < RK / 2; ++s) {
      float af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = a[(2 * s + h) * BM + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = b[(2 * s + h) * BN + j * 32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (more) store_chunk(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  float* dst = p.partial + (long long)z * p.Cout * p.Kld;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int k = k0 + wn * WN + j * 32 + l32;
    if (k < p.Kld) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int co = co0 + wm * WM + i * 32 + (e >> 2) * 8 + h * 4 + (e & 3);
          if (co < p.Cout) dst[(long long)co * p.Kld + k] = acc[i][j][e];
        }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// LDS-DMA variant (Cin % 4 == 0, Cout % 4 == 0, tensors < 4 GiB, kernel dims <= 8): both tiles are filled by
// buffer_load ... lds with 32-bit offsets (out-of-range => hardware zero fill).  The per-row geometry -- input byte
// offset of the row's window origin and one validity bit per kernel index and dimension -- does not depend on the k-tile,
// so a tiny pre-pass (rowgeom_kernel) writes it once per call and every workgroup streams it through a 3-deep LDS ring
// with one 256-byte copy per chunk: the copy threads then spend ~4 VALU per 16-byte copy instead of ~40, and no wave
// carries row arithmetic (on this fp32 MFMA loop every VALU issue slot comes straight out of the matrix pipe's).
// ------------------------------------------------------------------------------------------------------------------
typedef float floatx2 __attribute__((ext_vector_type(2)));
template <int T> struct FragVec;
template <> struct FragVec<1> {
  typedef float type;
  static __device__ __forceinline__ float get(const type& v, int) { return v; }
  static __device__ __forceinline__ void set(type& v, int, float x) { v = x; }
};
template <> struct FragVec<2> {
  typedef floatx2 type;
  static __device__ __forceinline__ float get(const type& v, int i) { return v[i]; }
  static __device__ __forceinline__ void set(type& v, int i, float x) { v[i] = x; }
};
template <> struct FragVec<4> {
  typedef floatx4 type;
  static __device__ __forceinline__ float get(const type& v, int i) { return v[i]; }
  static __device__ __forceinline__ void set(type& v, int i, float x) { v[i] = x; }
};

struct RowGeomParams {
  uint2* __restrict__ out;   // [M] {byte offset of x[n, gd*sD, gh*sH, gw*sW, 0] (mod 2^32), validity bits}
  int M, Gd, Gh, Gw, Di, Hi, Wi, in_ld, sD, sH, sW, kT, kH, kW, pT, pH, pW;
};

__global__ void rowgeom_kernel(const RowGeomParams p) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= p.M) return;
  const int gw = r % p.Gw;
  int q = r / p.Gw;
  const int gh = q % p.Gh;
  q /= p.Gh;
  const int gd = q % p.Gd;
  const int n = q / p.Gd;
  const int id0 = gd * p.sD, ih0 = gh * p.sH, iw0 = gw * p.sW;
  unsigned m = 0;
  for (int k = 0; k < p.kT; ++k) m |= (unsigned)((unsigned)(id0 + k - p.pT) < (unsigned)p.Di) << k;
  for (int k = 0; k < p.kH; ++k) m |= (unsigned)((unsigned)(ih0 + k - p.pH) < (unsigned)p.Hi) << (8 + k);
  for (int k = 0; k < p.kW; ++k) m |= (unsigned)((unsigned)(iw0 + k - p.pW) < (unsigned)p.Wi) << (16 + k);
  p.out[r] = make_uint2((unsigned)(((((long long)n * p.Di + id0) * p.Hi + ih0) * p.Wi + iw0) * p.in_ld * 4), m);
}

// H16 (the 32-row tile only): at most 16 output channels are live — the 16-channel remainder of R(2+1)D's 144 mid channels, whose
// launch walks all 1.6 M rows for them — and each wave's 32 x 32 block becomes two 16 x 16 blocks on v_mfma_f32_16x16x4_f32: half the
// matrix-pipe time for the same operand tiles.  Lane (r = lane % 16, g = lane / 16): channel r, GEMM-k rows 4*s + g of the chunk.
template <int BM, int BN, int WAVES_M, int WAVES_N, bool H16>
__device__ __forceinline__ void wgrad_dma_body(const WgradParams& p) {
  static_assert(!H16 || (BM == 32 && WAVES_M == 1 && BN / WAVES_N == 32), "H16: the 32-row tile, one 32-column block per wave");
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int ACOLS = BM / 4, BCOLS = BN / 4;
  constexpr int ARP = 256 / ACOLS, BRP = 256 / BCOLS;
  constexpr int AR = RK / ARP, BR = RK / BRP;
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  static_assert(RK == 32, "row-geometry ring: 32 rows x 8 B = 16 lanes x 16 B per chunk");
  typedef __attribute__((address_space(3))) void* lptr_t;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* At = reinterpret_cast<float*>(smem_raw);          // [2][RK][BM]
  float* Bt = At + 2 * RK * BM;                            // [2][RK][BN]
  uint2* rowtab = reinterpret_cast<uint2*>(Bt + 2 * RK * BN);   // [3][RK] ring of row geometry

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);   // scalar: LDS-DMA bases (M0) become SALU arithmetic
  const int l32 = lane & 31, h = lane >> 5;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  const WUnit unit = (p.lpt && p.kh1 > p.kh0 && p.kh1 - p.kh0 < p.k_tiles)
                         ? wgrad_unit_lpt(blockIdx.x, gridDim.x, p.splitm, p.co_tiles, p.k_tiles, p.kh0, p.kh1)
                         : wgrad_unit(blockIdx.x, gridDim.x, p.co_tiles * p.k_tiles);
  const int tile = unit.tile;
  const int co_tile = tile / p.k_tiles, k_tile = tile - co_tile * p.k_tiles;
  const int co0 = co_tile * BM, k0 = k_tile * BN;
  const int z = unit.z;
  const int row_begin = z * p.rows_per_split;
  const int row_end = min(p.M, row_begin + p.rows_per_split);

  __amdgpu_buffer_rsrc_t rsrc_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, (int)p.dy_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsrc_rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint2*>(p.rowgeom), 0, (int)((unsigned)p.M * 8u), 0x00020000);

  // A side (dy)
  const int acol = (t % ACOLS) * 4, arow = t / ACOLS;
  const bool acol_ok = co0 + acol < p.Cout;
  unsigned aoff0[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i)
    aoff0[i] = ((unsigned)(row_begin + arow + i * ARP) * (unsigned)p.dy_ld + (unsigned)(co0 + acol)) * 4u;
  const unsigned astep = (unsigned)RK * (unsigned)p.dy_ld * 4u;
  // B side (im2col x): this thread's k -> the three validity bits it needs and the byte delta of its (tap, ci)
  const int bcol = (t % BCOLS) * 4, brow = t / BCOLS;
  const int kk = k0 + bcol;
  const bool bok = kk < p.K;
  unsigned mymask = 0, bdelta4 = 0;
  {
    const int k1 = bok ? kk : k0;
    const int tap = k1 / p.Cin, ci = k1 - tap * p.Cin;
    const int kw = tap % p.kW, q = tap / p.kW;
    const int kh = q % p.kH, kt = q / p.kH;
    mymask = (1u << kt) | (1u << (8 + kh)) | (1u << (16 + kw));
    bdelta4 = (unsigned)(((((kt - p.pT) * p.Hi + (kh - p.pH)) * p.Wi + (kw - p.pW)) * p.in_ld + ci) * 4);
  }

  // row geometry of chunk `chunk` -> ring slot: 16 lanes of wave 0 copy 32 rows x 8 B (rows >= M read as zeros = invalid;
  // rows >= row_end are cancelled on the dy side)
  auto fetch_rows = [&](int chunk, int slot) {
    if (t < 16) {
      const unsigned off = ((unsigned)(row_begin + chunk * RK) + 2u * (unsigned)t) * 8u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_rg, (lptr_t)(rowtab + slot * RK), 16, off, 0, 0, 0);
    }
  };

  uint2 rt[BR];
  auto read_rowtab = [&](int slot) {
#pragma unroll
    for (int i = 0; i < BR; ++i) rt[i] = rowtab[slot * RK + brow + i * BRP];
  };
  auto issue = [&](int chunk, int buf) {   // uses rt[] read earlier (no LDS access after the first DMA of a chunk)
    const int rbase = row_begin + chunk * RK;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const bool ok = acol_ok & (rbase + arow + i * ARP < row_end);
      const unsigned off = ok ? aoff0[i] + (unsigned)chunk * astep : 0xffffffffu;
      float* dst = At + buf * RK * BM + i * ARP * BM + wave * 256;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_dy, (lptr_t)dst, 16, off, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < BR; ++i) {
      const bool ok = bok & ((rt[i].y & mymask) == mymask);   // '&': no short-circuit branch
      const unsigned off = ok ? rt[i].x + bdelta4 : 0xffffffffu;
      float* dst = Bt + buf * RK * BN + i * BRP * BN + wave * 256;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lptr_t)dst, 16, off, 0, 0, 0);
    }
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  floatx4 acc16[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};      // H16: columns 16 j + r of the wave's 32
  const int r16 = lane & 15, g16 = lane >> 4;

  const int nchunk = (row_end - row_begin + RK - 1) / RK;
  // Row chunks that cannot contribute to THIS k tile are skipped: the tile's taps span the kernel depths [kt_lo, kt_hi]; a chunk of
  // 32 consecutive rows lies in one or two output frames, and for a frame at the volume's edge those depths may all fall into
  // the zero padding (3x3x3 over T frames: a third of the k tiles skip 1/T of the rows each — C3D conv3 8 %, conv4 17 %, conv5 33 %
  // of the launch).  Scalar arithmetic per chunk; the walk below goes from live chunk to live chunk.
  const int khw = p.kH * p.kW;
  const int kt_lo = (k0 / p.Cin) / khw, kt_hi = ((min(k0 + BN, p.K) - 1) / p.Cin) / khw;
  auto frame_live = [&](int q) {
    const int g = q - fastdiv(q, p.dGd) * p.Gd;
    const int lo = g * p.sD - p.pT + kt_lo, hi = g * p.sD - p.pT + kt_hi;
    return hi >= 0 && lo < p.Di;
  };
  auto next_live = [&](int c) {
    if (!p.skip_pad) return c < nchunk ? c : nchunk;
    for (; c < nchunk; ++c) {
      const int r0 = row_begin + c * RK, r1 = min(r0 + RK, row_end) - 1;
      const int q0 = fastdiv(r0, p.dP), q1 = fastdiv(r1, p.dP);
      if (q1 - q0 > 1 || frame_live(q0) || frame_live(q1)) break;
    }
    return c;
  };
  int buf = 0;
  // one chunk: fragments of the current buffer -> registers, the row geometry of chunk c_fetch -> ring slot slot_fetch, the copies
  // of chunk c_issue (its geometry is in ring slot slot_next) -> the other buffer, `beside` (scalar work), MFMAs, barrier
  auto chunk = [&](int slot_next, int slot_fetch, int c_fetch, int c_issue, bool more, auto beside) {
    // 1. every LDS access of this iteration first (hipcc orders LDS accesses behind pending LDS-DMA); the row table goes
    //    first so that the copies' address math does not wait for the whole fragment burst
    if (more) read_rowtab(slot_next);
    const float* a = At + buf * RK * BM + wm * WM + TM * l32;
    const float* b = Bt + buf * RK * BN + wn * WN + TN * l32;
    // MFMA row m of sub-tile i is channel TM*m + i (columns likewise), so a lane's TM (TN) operands for one k-step are
    // adjacent floats: one ds_read_b64 instead of two ds_read_b32 (the epilogue undoes the interleave)
    typename FragVec<TM>::type af[RK / 2];
    typename FragVec<TN>::type bf[RK / 2];
    float a16[RK / 4], b16[2][RK / 4];
    if (H16) {
      const float* ah = At + buf * RK * BM + r16;
      const float* bh = Bt + buf * RK * BN + wn * WN + r16;
#pragma unroll
      for (int s4 = 0; s4 < RK / 4; ++s4) {
        a16[s4] = ah[(4 * s4 + g16) * BM];
        b16[0][s4] = bh[(4 * s4 + g16) * BN];
        b16[1][s4] = bh[(4 * s4 + g16) * BN + 16];
      }
    } else {
#pragma unroll
      for (int s2 = 0; s2 < RK / 2; ++s2) {
        af[s2] = *reinterpret_cast<const typename FragVec<TM>::type*>(a + (2 * s2 + h) * BM);
        bf[s2] = *reinterpret_cast<const typename FragVec<TN>::type*>(b + (2 * s2 + h) * BN);
      }
    }
    fetch_rows(c_fetch, slot_fetch);
    // 2. next chunk's copies in flight under this chunk's MFMAs
    if (more) issue(c_issue, buf ^ 1);
    beside();
    // 3. MFMAs
    if (H16) {
#pragma unroll
      for (int s4 = 0; s4 < RK / 4; ++s4)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc16[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a16[s4], b16[j][s4], acc16[j], 0, 0, 0);
    } else {
#pragma unroll
      for (int s2 = 0; s2 < RK / 2; ++s2)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(FragVec<TM>::get(af[s2], i), FragVec<TN>::get(bf[s2], j), acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    buf ^= 1;
  };
  if (!p.skip_pad) {      // every chunk in turn
    fetch_rows(0, 0);
    fetch_rows(1, 1);
    __syncthreads();
    if (nchunk > 0) {
      read_rowtab(0);
      issue(0, 0);
    }
    __syncthreads();   // vmcnt(0) + barrier: chunk 0 landed
    for (int c = 0; c < nchunk; ++c) chunk((c + 1) % 3, (c + 2) % 3, c + 2, c + 1, c + 1 < nchunk, [] {});
  } else {                // from live chunk to live chunk: c0 = current, c1 = next, c2 = the one after
    int c0 = next_live(0);
    int c1 = next_live(c0 + 1);
    int c2 = next_live(c1 + 1);
    fetch_rows(c0, 0);
    fetch_rows(c1, 1);
    __syncthreads();
    if (c0 < nchunk) {
      read_rowtab(0);
      issue(c0, 0);
    }
    __syncthreads();   // vmcnt(0) + barrier: the first chunk landed
    int slot = 0;
    for (int c3 = 0; c0 < nchunk; c0 = c1, c1 = c2, c2 = c3, slot = slot == 2 ? 0 : slot + 1)
      // (the seek is scalar: issued in front of the MFMAs it runs beside them instead of behind the barrier)
      chunk(slot == 2 ? 0 : slot + 1, slot == 0 ? 2 : slot - 1, c2, c1, c1 < nchunk, [&] { c3 = next_live(c2 + 1); });
  }

  float* dst = p.partial + (long long)z * p.Cout * p.Kld;
  if (H16) {      // block j: element r of lane (r16, g16) is channel 4 * g16 + r, column 16 j + r16
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int kc = k0 + wn * WN + 16 * j + r16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + 4 * g16 + r;
        if (kc < p.Kld && co < p.Cout) dst[(long long)co * p.Kld + kc] = acc16[j][r];
      }
    }
    return;
  }
  const int k = k0 + wn * WN + TN * l32;   // TN adjacent columns per lane; Kld % 4 == 0 and k % TN == 0
  if (k < p.Kld) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = co0 + wm * WM + TM * ((e >> 2) * 8 + h * 4 + (e & 3)) + i;
        if (co < p.Cout) {
          typename FragVec<TN>::type v;
#pragma unroll
          for (int j = 0; j < TN; ++j) FragVec<TN>::set(v, j, acc[i][j][e]);
          *reinterpret_cast<typename FragVec<TN>::type*>(dst + (long long)co * p.Kld + k) = v;
        }
      }
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256, 2) void wgrad_dma_kernel(const WgradParams p) {
  wgrad_dma_body<BM, BN, WAVES_M, WAVES_N, false>(p);
}
// (a kernel of its own name: the profiler's rows of the other instances keep their spelling)
__global__ __launch_bounds__(256, 2) void wgrad_dma_h16_kernel(const WgradParams p) { wgrad_dma_body<32, 128, 1, 4, true>(p); }

// dw_ref[co][ci][tap] = sum_z partial[z][co][tap*Cin + ci]   (fixed summation order)
// Block = one output channel x up to 32 input channels: partials are read along k (contiguous ci runs per tap), summed
// over the slabs with 4 loads in flight, transposed through LDS and written as one contiguous run of dw_ref.
// Cin_v (<= Cin): input channels of the PARAMETER — channels beyond are zero padding of the activations whose gradients are dropped
// (the output-channel padding is cut by the grid: blockIdx.y < valid output channels).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                           int splitm, int Cout, int Cin, int taps, int Kld, int Cin_v) {
  extern __shared__ float wr_tile[];   // [cit][taps]
  const int cit = Cin < 32 ? Cin : 32;
  const int ci0 = blockIdx.x * cit, co = blockIdx.y;
  const int nci = min(cit, Cin_v - ci0);
  if (nci <= 0) return;
  const long long slab = (long long)Cout * Kld;
  const float* base = partial + (long long)co * Kld;
  for (int e = threadIdx.x; e < nci * taps; e += 256) {
    const int tap = e / nci, cil = e - tap * nci;
    const float* src = base + tap * Cin + ci0 + cil;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int z = 0;
    for (; z + 4 <= splitm; z += 4) {
      s0 += src[(long long)z * slab];
      s1 += src[(long long)(z + 1) * slab];
      s2 += src[(long long)(z + 2) * slab];
      s3 += src[(long long)(z + 3) * slab];
    }
    for (; z < splitm; ++z) s0 += src[(long long)z * slab];
    wr_tile[cil * taps + tap] = (s0 + s1) + (s2 + s3);
  }
  __syncthreads();
  float* out = dw + ((long long)co * Cin_v + ci0) * taps;
  for (int e = threadIdx.x; e < nci * taps; e += 256) out[e] = wr_tile[e];
}

// Column sums of a [rows][C] (pitch ld) matrix, two deterministic stages.
__global__ __launch_bounds__(256) void colsum_stage1(const float* __restrict__ a, long long rows, int C, int ld,
                                                     float* __restrict__ part, int rows_per_block) {
  // block (bx = row slab, by = 64-channel group); thread = (row lane 0..3, channel 0..63)
  __shared__ float red[4][64];
  const int c = blockIdx.y * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  float s = 0.f;
  if (c < C)
    for (long long r = r0 + rl; r < r1; r += 4) s += a[r * ld + c];
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < C) part[(long long)blockIdx.x * C + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
__global__ void colsum_stage2(const float* __restrict__ part, int nblk, int C, float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0;
  for (int b = 0; b < nblk; ++b) s += (double)part[(long long)b * C + c];
  out[c] = (float)s;
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool VECA, bool VECB>
int launch_w(const WgradParams& p, hipStream_t s) {
  const size_t lds = (size_t)2 * RK * (BM + BN) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<BM, BN, WAVES_M, WAVES_N, VECA, VECB>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  dim3 grid((unsigned)(p.co_tiles * p.k_tiles) * (unsigned)p.splitm);
  rsp_note_kernel("wgrad_kernel<%d, %d, %d, %d, *>", BM, BN, WAVES_M, WAVES_N);
  hipLaunchKernelGGL((wgrad_kernel<BM, BN, WAVES_M, WAVES_N, VECA, VECB>), grid, dim3(256), lds, s, p);
  return rsp_check_launch("wgrad_kernel");
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
int launch_w_dma(const WgradParams& p, hipStream_t s) {
  const size_t lds = (size_t)2 * RK * (BM + BN) * sizeof(float) + 3 * RK * sizeof(uint2);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_dma_kernel<BM, BN, WAVES_M, WAVES_N>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  dim3 grid((unsigned)(p.co_tiles * p.k_tiles) * (unsigned)p.splitm);
  rsp_note_kernel("wgrad_dma_kernel<%d, %d, %d, %d>", BM, BN, WAVES_M, WAVES_N);
  hipLaunchKernelGGL((wgrad_dma_kernel<BM, BN, WAVES_M, WAVES_N>), grid, dim3(256), lds, s, p);
  return rsp_check_launch("wgrad_dma_kernel");
}

int launch_w_dma_h16(const WgradParams& p, hipStream_t s) {
  const size_t lds = (size_t)2 * RK * (32 + 128) * sizeof(float) + 3 * RK * sizeof(uint2);
  dim3 grid((unsigned)(p.co_tiles * p.k_tiles) * (unsigned)p.splitm);
  rsp_note_kernel("wgrad_dma_h16_kernel");
  hipLaunchKernelGGL(wgrad_dma_h16_kernel, grid, dim3(256), lds, s, p);
  return rsp_check_launch("wgrad_dma_h16_kernel");
}

// the 32-row tile with at most 16 live channels: two 16 x 16 blocks per wave (RSP_NO_HALF_BLOCK=1: the 32 x 32 block; read once)
static bool wgrad_h16(int cout) {
  static const bool off = getenv("RSP_NO_HALF_BLOCK") != nullptr;
  return !off && cout <= 16;
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
int launch_w_vec(const WgradParams& p, bool va, bool vb, hipStream_t s) {
  if (va && vb) return launch_w<BM, BN, WAVES_M, WAVES_N, true, true>(p, s);
  if (va) return launch_w<BM, BN, WAVES_M, WAVES_N, true, false>(p, s);
  if (vb) return launch_w<BM, BN, WAVES_M, WAVES_N, false, true>(p, s);
  return launch_w<BM, BN, WAVES_M, WAVES_N, false, false>(p, s);
}

bool wdesc_ok(const rsp_conv3d_desc* d) {
  if (!d) return false;
  if (d->N <= 0 || d->Cin <= 0 || d->Cout <= 0) return false;
  if (d->kT <= 0 || d->kH <= 0 || d->kW <= 0 || d->kT * d->kH * d->kW > MAX_TAPS) return false;
  if (d->sT <= 0 || d->sH <= 0 || d->sW <= 0 || d->pT < 0 || d->pH < 0 || d->pW < 0) return false;
  if (d->Do != (d->Di + 2 * d->pT - d->kT) / d->sT + 1) return false;
  if (d->Ho != (d->Hi + 2 * d->pH - d->kH) / d->sH + 1) return false;
  if (d->Wo != (d->Wi + 2 * d->pW - d->kW) / d->sW + 1) return false;
  if (d->Do <= 0 || d->Ho <= 0 || d->Wo <= 0) return false;
  if (d->in_ld < d->Cin || d->out_ld < d->Cout) return false;
  if ((long long)d->N * d->Do * d->Ho * d->Wo >= (1ll << 31)) return false;
  return true;
}

struct WPlan {
  int bm, bn, co_tiles, k_tiles, splitm, rows_per_split, Kld;
  size_t partial_bytes, colsum_bytes, rowgeom_bytes;
};

WPlan wplan(const rsp_conv3d_desc* d) {
  WPlan w;
  const long long M = (long long)d->N * d->Do * d->Ho * d->Wo;
  const int K = d->kT * d->kH * d->kW * d->Cin;
  w.Kld = (int)rsp_align_up((size_t)K, 4);
  w.bm = d->Cout > 64 ? 128 : (d->Cout > 32 ? 64 : 32);
  w.bn = K > 64 ? 128 : 64;
  if (w.bm == 32 && w.bn == 64) w.bm = 64;   // (the 32-row tile stacks its four waves along k: 4 x 32 columns)
  w.co_tiles = rsp_cdiv(d->Cout, w.bm);
  w.k_tiles = rsp_cdiv(K, w.bn);
  const int tiles = w.co_tiles * w.k_tiles;
  // Split the rows into `sp` slabs so that the tiles*sp units fill whole rounds of the resident workgroups (512 = 2 per CU for 128x128):
  // a few stragglers spilling into an extra round cost ~15% (measured: conv2 6.4 -> 5.4 ms), a last round that is 84% full
  // costs the missing 16%.  Utilisation first, then fewer rounds (longer units amortise the 64 KB partial write better).
  const long long max_sp = M / 256 > 0 ? M / 256 : 1;
  long long sp = 1;
  {
    const long long lds = 2ll * RK * (w.bm + w.bn) * 4 + 1024;
    const long long slots = 256 * (160 * 1024 / lds > 4 ? 4 : 160 * 1024 / lds);   // 128x128: 2 per CU, 64x128: 3, 64x64: 4
    double best = -1.0;
    for (long long c = 1; c <= max_sp && c * tiles <= 8192; ++c) {
      const long long units = c * tiles, rounds = (units + slots - 1) / slots;
      const double score = (double)units / (double)(rounds * slots) - 0.004 * (double)rounds;
      if (score > best) {
        best = score;
        sp = c;
      }
    }
    if (sp > max_sp) sp = max_sp;
  }
  long long rps = (M + sp - 1) / sp;
  rps = (rps + RK - 1) / RK * RK;
  w.rows_per_split = (int)rps;
  w.splitm = (int)((M + rps - 1) / rps);
  w.partial_bytes = rsp_align_up((size_t)w.splitm * d->Cout * w.Kld * sizeof(float), 256);
  w.colsum_bytes = rsp_align_up((size_t)rsp_cdiv(M, 1024) * d->Cout * sizeof(float), 256);
  w.rowgeom_bytes = rsp_align_up((size_t)M * sizeof(uint2), 256);
  return w;
}

}  // namespace

struct WSegs { int n, at[2], width[2]; };
WSegs wgrad_segments(const rsp_conv3d_desc* d);

const char* rsp_wgrad_kernel_name(const rsp_conv3d_desc* d0) {
  if (!wdesc_ok(d0)) return "invalid";
  rsp_conv3d_desc first = *d0;
  first.Cout = wgrad_segments(d0).width[0];     // a convolution run as several output-channel segments is named after the first
  const rsp_conv3d_desc* d = &first;
  const WPlan w = wplan(d);
  const bool dma = d->Cout % 4 == 0 && d->out_ld % 4 == 0 && d->Cin % 4 == 0 && d->in_ld % 4 == 0 && d->kT <= 8 && d->kH <= 8 &&
                   d->kW <= 8;
  if (w.bm == 32) return dma ? (wgrad_h16(d->Cout) ? "wgrad_dma_h16_kernel" : "wgrad_dma_kernel<32, 128, 1, 4>") : "wgrad_kernel<32, 128, 1, 4, *>";
  if (dma) return w.bm == 128 ? (w.bn == 128 ? "wgrad_dma_kernel<128, 128, 2, 2>" : "wgrad_dma_kernel<128, 64, 2, 2>")
                              : (w.bn == 128 ? "wgrad_dma_kernel<64, 128, 2, 2>" : "wgrad_dma_kernel<64, 64, 2, 2>");
  return w.bm == 128 ? (w.bn == 128 ? "wgrad_kernel<128, 128, 2, 2, *>" : "wgrad_kernel<128, 64, 2, 2, *>")
                     : (w.bn == 128 ? "wgrad_kernel<64, 128, 2, 2, *>" : "wgrad_kernel<64, 64, 2, 2, *>");
}

// Output-channel segments (same idea as the forward kernel's column segments, conv_igemm.hip).  The Cout tiles are 128, 64 or 32
// wide; a channel count is cut into the multiple of 128 below it plus a remainder r that runs on the narrowest tile that holds
// it: r <= 32 -> the 32-wide tile, r <= 64 -> the 64-wide one, else one more 128-wide tile (a 64 + 32 cut of 65..96 was measured:
// no gain on R(2+1)D's 84-channel stem, +41 % time on S3D-G's 96-channel pointwise layers — every segment is a launch triple of
// its own).  144 = 128 + 16: 90 % of the MFMA work useful instead of 56 % (R(2+1)D conv2 spatial: 3.29 -> 2.85 ms).  The
// segments of a call share the row-geometry table.
WSegs wgrad_segments(const rsp_conv3d_desc* d) {
  WSegs g;
  memset(&g, 0, sizeof g);
  const int C = d->Cout;
  int full = C / 128 * 128, r = C - full;
  if (r > 64) { full = C; r = 0; }     // the remainder rides in one more 128-wide tile
  if (full > 0) { g.at[g.n] = 0; g.width[g.n++] = full; }
  if (r > 0) { g.at[g.n] = full; g.width[g.n++] = r; }
  return g;
}

namespace {

size_t wgrad_ws_one(const rsp_conv3d_desc* d) {
  const WPlan w = wplan(d);
  return w.partial_bytes + w.colsum_bytes + w.rowgeom_bytes;
}

// Which (k tile, output frame) pairs the row walk of wgrad_dma_kernel skips, and whether it is switched on for this geometry.
struct WSkip {
  int skip_pad, kh0, kh1, lpt;
  long long dead_frames;      // (k tile, frame) pairs whose kernel depths all fall into the padding
};

WSkip wgrad_skip_plan(const rsp_conv3d_desc* d, const WPlan& w) {
  WSkip k = {0, 0, 0, 0, 0};
  static const bool no_skip = getenv("RSP_NO_PAD_SKIP") != nullptr;      // (A/B switch for measurements, read once)
  k.skip_pad = (!no_skip && d->kT > 1 && d->pT > 0 && d->Ho * d->Wo >= RK) ? 1 : 0;
  const int K = d->kT * d->kH * d->kW * d->Cin;
  // k tiles that skip no frame: those whose kernel depths [kt_lo, kt_hi] reach inside the input for every output frame
  const int khw = d->kH * d->kW;
  bool open = false;
  for (int j = 0; j < w.k_tiles; ++j) {
    const int k0 = j * w.bn, k1 = (k0 + w.bn < K ? k0 + w.bn : K) - 1;
    const int kt_lo = (k0 / d->Cin) / khw, kt_hi = (k1 / d->Cin) / khw;
    int dead = 0;
    for (int g = 0; g < d->Do; ++g) dead += (g * d->sT - d->pT + kt_hi < 0 || g * d->sT - d->pT + kt_lo >= d->Di) ? 1 : 0;
    k.dead_frames += dead;
    if (!dead && !open) { k.kh0 = j; open = true; }
    if (!dead) k.kh1 = j + 1;
  }
  // worth the walk and the re-ordered units (which cost some L2 locality) from ~6 % of the (k tile, frame) pairs on:
  // C3D conv2 4 % (measured 5.52 -> 5.66 ms with it), conv3 8 %, conv4 17 %, conv5 33 %
  if (k.dead_frames * 100 < 6ll * w.k_tiles * d->Do) k.skip_pad = 0;
  if (k.kh0 == 0 && k.kh1 == w.k_tiles) k.skip_pad = 0;      // nothing to skip (e.g. 4-channel stems: every k tile spans all depths)
  const long long lds = 2ll * RK * (w.bm + w.bn) * 4 + 1024;
  const long long slots = 256 * (160 * 1024 / lds > 4 ? 4 : 160 * 1024 / lds);      // as in wplan()
  k.lpt = (k.skip_pad && (long long)w.splitm * w.co_tiles * w.k_tiles * 2 > slots * 3) ? 1 : 0;
  k.skip_pad = k.lpt;      // (one round: every unit runs at once and the full-length ones set the time — C3D conv2: 5.52 -> 5.65 ms with the walk)
  return k;
}

int wgrad_one(const rsp_conv3d_desc* d, const float* x, const float* dy, float* dw_ref, float* dbias, void* workspace,
              size_t workspace_bytes, void* stream, bool rowgeom_ready, int cout_valid, int cin_valid, const uint2* ext_rowgeom);
int rowgeom_fill(const rsp_conv3d_desc* d, void* table, hipStream_t s);

}  // namespace

double rsp_wgrad_executed_fraction(const rsp_conv3d_desc* d0) {
  if (!wdesc_ok(d0)) return 1.0;
  const WSegs g = wgrad_segments(d0);
  double live = 0.0, tot = 0.0;
  for (int i = 0; i < g.n; ++i) {
    rsp_conv3d_desc seg = *d0;
    seg.Cout = g.width[i];
    const WPlan w = wplan(&seg);
    const bool dma = seg.Cout % 4 == 0 && seg.out_ld % 4 == 0 && seg.Cin % 4 == 0 && seg.in_ld % 4 == 0 && seg.kT <= 8 && seg.kH <= 8 &&
                     seg.kW <= 8;
    const WSkip k = wgrad_skip_plan(&seg, w);
    // (frame-granular: a 32-row chunk that straddles a live and a dead frame still runs)
    const double f = (dma && k.skip_pad) ? 1.0 - (double)k.dead_frames / ((double)w.k_tiles * seg.Do) : 1.0;
    live += f * seg.Cout;
    tot += seg.Cout;
  }
  return tot > 0 ? live / tot : 1.0;
}


extern "C" {

size_t rsp_conv3d_wgrad_workspace(const rsp_conv3d_desc* d) {
  if (!wdesc_ok(d)) return 0;
  const WSegs g = wgrad_segments(d);
  size_t best = 0;
  for (int i = 0; i < g.n; ++i) {
    rsp_conv3d_desc a = *d;
    a.Cout = g.width[i];
    const size_t b = wgrad_ws_one(&a);
    best = b > best ? b : best;
  }
  return best;
}

static int wgrad_all(const rsp_conv3d_desc* d, const float* x, const float* dy, float* dw_ref, float* dbias, int cout_valid,
                     int cin_valid, void* workspace, size_t workspace_bytes, void* stream, const uint2* ext_rowgeom = nullptr) {
  rsp_note_reset();
  // problems over disjoint output-channel ranges; dy keeps its row pitch (out_ld), dw / dbias are contiguous per channel
  const WSegs g = wgrad_segments(d);
  const long long per_co = (long long)cin_valid * d->kT * d->kH * d->kW;
  for (int i = 0; i < g.n; ++i) {
    rsp_conv3d_desc a = *d;
    a.Cout = g.width[i];
    const int at = g.at[i];
    int cov = cout_valid - at;              // valid output channels inside this segment
    cov = cov < 0 ? 0 : (cov > a.Cout ? a.Cout : cov);
    if (cov == 0) continue;
    const int rc = wgrad_one(&a, x, dy + at, dw_ref + at * per_co, dbias ? dbias + at : nullptr, workspace, workspace_bytes, stream,
                             i > 0, cov, cin_valid, ext_rowgeom);
    if (rc != RSP_OK) return rc;
  }
  return RSP_OK;
}

int rsp_conv3d_wgrad(const rsp_conv3d_desc* d, const float* x, const float* dy, float* dw_ref, float* dbias,
                     void* workspace, size_t workspace_bytes, void* stream) {
  RSP_REQUIRE(wdesc_ok(d), "rsp_conv3d_wgrad: bad descriptor");
  RSP_REQUIRE(x && dy && dw_ref && workspace, "rsp_conv3d_wgrad: null pointer");
  RSP_REQUIRE(rsp_aligned16(workspace), "rsp_conv3d_wgrad: workspace must be 16-byte aligned");
  return wgrad_all(d, x, dy, dw_ref, dbias, d->Cout, d->Cin, workspace, workspace_bytes, stream);
}

int rsp_conv3d_wgrad_v(const rsp_conv3d_desc* d, const float* x, const float* dy, float* dw_ref, int32_t cout_valid,
                       int32_t cin_valid, void* workspace, size_t workspace_bytes, void* stream) {
  RSP_REQUIRE(wdesc_ok(d), "rsp_conv3d_wgrad_v: bad descriptor");
  RSP_REQUIRE(x && dy && dw_ref && workspace, "rsp_conv3d_wgrad_v: null pointer");
  RSP_REQUIRE(rsp_aligned16(workspace), "rsp_conv3d_wgrad_v: workspace must be 16-byte aligned");
  RSP_REQUIRE(cout_valid > 0 && cout_valid <= d->Cout && cin_valid > 0 && cin_valid <= d->Cin, "rsp_conv3d_wgrad_v: bad valid channel counts");
  return wgrad_all(d, x, dy, dw_ref, nullptr, cout_valid, cin_valid, workspace, workspace_bytes, stream);
}

// Row-geometry table of a convolution (input byte offset + validity bits per output position; depends on the geometry only, not on
// channels or data): a caller that keeps it per geometry saves the pre-pass launch of every weight-gradient call.
size_t rsp_conv3d_rowgeom_bytes(const rsp_conv3d_desc* d) {
  if (!wdesc_ok(d)) return 0;
  return (size_t)d->N * d->Do * d->Ho * d->Wo * sizeof(uint2);
}

int rsp_conv3d_rowgeom(const rsp_conv3d_desc* d, void* table, void* stream) {
  RSP_REQUIRE(wdesc_ok(d) && table && rsp_aligned16(table), "rsp_conv3d_rowgeom: bad argument");
  RSP_REQUIRE(d->kT <= 8 && d->kH <= 8 && d->kW <= 8, "rsp_conv3d_rowgeom: kernel dims above 8 take the table-free path");
  return rowgeom_fill(d, table, (hipStream_t)stream);
}

int rsp_conv3d_wgrad_t(const rsp_conv3d_desc* d, const float* x, const float* dy, float* dw_ref, float* dbias, int32_t cout_valid,
                       int32_t cin_valid, const void* rowgeom_table, void* workspace, size_t workspace_bytes, void* stream) {
  RSP_REQUIRE(wdesc_ok(d), "rsp_conv3d_wgrad_t: bad descriptor");
  RSP_REQUIRE(x && dy && dw_ref && workspace, "rsp_conv3d_wgrad_t: null pointer");
  RSP_REQUIRE(rsp_aligned16(workspace) && (!rowgeom_table || rsp_aligned16(rowgeom_table)), "rsp_conv3d_wgrad_t: workspace / table must be 16-byte aligned");
  RSP_REQUIRE(cout_valid > 0 && cout_valid <= d->Cout && cin_valid > 0 && cin_valid <= d->Cin, "rsp_conv3d_wgrad_t: bad valid channel counts");
  RSP_REQUIRE(!dbias || (cout_valid == d->Cout && cin_valid == d->Cin), "rsp_conv3d_wgrad_t: a bias gradient needs unpadded channels");
  return wgrad_all(d, x, dy, dw_ref, dbias, cout_valid, cin_valid, workspace, workspace_bytes, stream,
                   reinterpret_cast<const uint2*>(rowgeom_table));
}

}  // extern "C"

namespace {

int rowgeom_fill(const rsp_conv3d_desc* d, void* table, hipStream_t s) {
  RowGeomParams g;
  g.out = reinterpret_cast<uint2*>(table);
  g.M = d->N * d->Do * d->Ho * d->Wo; g.Gd = d->Do; g.Gh = d->Ho; g.Gw = d->Wo; g.Di = d->Di; g.Hi = d->Hi; g.Wi = d->Wi; g.in_ld = d->in_ld;
  g.sD = d->sT; g.sH = d->sH; g.sW = d->sW; g.kT = d->kT; g.kH = d->kH; g.kW = d->kW; g.pT = d->pT; g.pH = d->pH; g.pW = d->pW;
  hipLaunchKernelGGL(rowgeom_kernel, dim3(rsp_cdiv(g.M, 256)), dim3(256), 0, s, g);
  return rsp_check_launch("rowgeom_kernel");
}

int wgrad_one(const rsp_conv3d_desc* d, const float* x, const float* dy, float* dw_ref, float* dbias, void* workspace,
              size_t workspace_bytes, void* stream, bool rowgeom_ready, int cout_valid, int cin_valid, const uint2* ext_rowgeom) {
  const WPlan w = wplan(d);
  if (workspace_bytes < w.partial_bytes + w.colsum_bytes + w.rowgeom_bytes) {
    rsp_set_error("rsp_conv3d_wgrad: workspace too small");
    return RSP_EWORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  static const float* zero_cache[64] = {nullptr};   // per device: a __device__ symbol has one instance per device
  int dev_id = 0;
  if (hipGetDevice(&dev_id) != hipSuccess || dev_id < 0 || dev_id >= 64) dev_id = 0;
  if (!zero_cache[dev_id]) {
    void* z = nullptr;
    if (hipGetSymbolAddress(&z, HIP_SYMBOL(g_wzero)) != hipSuccess || !z) {
      rsp_set_error("hipGetSymbolAddress(g_wzero) failed");
      return RSP_ELAUNCH;
    }
    zero_cache[dev_id] = reinterpret_cast<const float*>(z);
  }
  const float* zero_page = zero_cache[dev_id];
  WgradParams p;
  memset(&p, 0, sizeof p);
  p.zero = zero_page;
  p.x = x; p.dy = dy; p.partial = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(workspace) + w.rowgeom_bytes);
  p.M = d->N * d->Do * d->Ho * d->Wo;
  p.Gd = d->Do; p.Gh = d->Ho; p.Gw = d->Wo;
  p.Di = d->Di; p.Hi = d->Hi; p.Wi = d->Wi; p.in_ld = d->in_ld; p.Cin = d->Cin;
  p.sD = d->sT; p.sH = d->sH; p.sW = d->sW;
  p.kT = d->kT; p.kH = d->kH; p.kW = d->kW; p.pT = d->pT; p.pH = d->pH; p.pW = d->pW;
  p.dy_ld = d->out_ld; p.Cout = d->Cout;
  p.K = d->kT * d->kH * d->kW * d->Cin;
  p.Kld = w.Kld;
  p.rows_per_split = w.rows_per_split; p.splitm = w.splitm;
  p.co_tiles = w.co_tiles; p.k_tiles = w.k_tiles;
  p.dP = fastdiv_make(d->Ho * d->Wo);
  p.dGd = fastdiv_make(d->Do);
  {
    const WSkip sk = wgrad_skip_plan(d, w);
    p.skip_pad = sk.skip_pad; p.kh0 = sk.kh0; p.kh1 = sk.kh1; p.lpt = sk.lpt;
  }
  const bool va = (d->Cout % 4 == 0) && (d->out_ld % 4 == 0) && rsp_aligned16(dy);
  const bool vb = (d->Cin % 4 == 0) && (d->in_ld % 4 == 0) && rsp_aligned16(x);
  const unsigned long long xb = (unsigned long long)d->N * d->Di * d->Hi * d->Wi * d->in_ld * 4ull;
  const unsigned long long dyb = (unsigned long long)p.M * d->out_ld * 4ull;
  p.x_bytes = (unsigned)xb;
  p.dy_bytes = (unsigned)dyb;
  bool dma = va && vb && xb < (1ull << 32) && dyb < (1ull << 32) && d->kT <= 8 && d->kH <= 8 && d->kW <= 8 &&
             (unsigned long long)p.M * sizeof(uint2) < (1ull << 32);   // row-geometry table addressed with 32-bit offsets
  int rc;
  if (dma) {
    RowGeomParams g;
    g.out = reinterpret_cast<uint2*>(workspace);
    g.M = p.M; g.Gd = p.Gd; g.Gh = p.Gh; g.Gw = p.Gw; g.Di = p.Di; g.Hi = p.Hi; g.Wi = p.Wi; g.in_ld = p.in_ld;
    g.sD = p.sD; g.sH = p.sH; g.sW = p.sW; g.kT = p.kT; g.kH = p.kH; g.kW = p.kW; g.pT = p.pT; g.pH = p.pH; g.pW = p.pW;
    if (ext_rowgeom) {       // the caller keeps the table of this geometry (rsp_conv3d_rowgeom): nothing to compute per call
      g.out = const_cast<uint2*>(ext_rowgeom);
    } else if (!rowgeom_ready) {    // the table depends on the rows only: the output-channel segments of one call share it
      hipLaunchKernelGGL(rowgeom_kernel, dim3(rsp_cdiv(p.M, 256)), dim3(256), 0, s, g);
      rc = rsp_check_launch("rowgeom_kernel");
      if (rc != RSP_OK) return rc;
    }
    p.rowgeom = g.out;
  }
  if (dma && w.bm == 32) rc = wgrad_h16(p.Cout) ? launch_w_dma_h16(p, s) : launch_w_dma<32, 128, 1, 4>(p, s);
  else if (w.bm == 32) rc = launch_w_vec<32, 128, 1, 4>(p, va, vb, s);
  else if (dma && w.bm == 128 && w.bn == 128) rc = launch_w_dma<128, 128, 2, 2>(p, s);
  else if (dma && w.bm == 128) rc = launch_w_dma<128, 64, 2, 2>(p, s);
  else if (dma && w.bn == 128) rc = launch_w_dma<64, 128, 2, 2>(p, s);
  else if (dma) rc = launch_w_dma<64, 64, 2, 2>(p, s);
  else if (w.bm == 128 && w.bn == 128) rc = launch_w_vec<128, 128, 2, 2>(p, va, vb, s);
  else if (w.bm == 128) rc = launch_w_vec<128, 64, 2, 2>(p, va, vb, s);
  else if (w.bn == 128) rc = launch_w_vec<64, 128, 2, 2>(p, va, vb, s);
  else rc = launch_w_vec<64, 64, 2, 2>(p, va, vb, s);
  if (rc != RSP_OK) return rc;
  {
    const int taps = d->kT * d->kH * d->kW;
    const int cit = d->Cin < 32 ? d->Cin : 32;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rsp_cdiv(cin_valid, cit), cout_valid), dim3(256), (size_t)cit * taps * sizeof(float), s,
                       p.partial, dw_ref, p.splitm, d->Cout, d->Cin, taps, p.Kld, cin_valid);
    rc = rsp_check_launch("wgrad_reduce_kernel");
    if (rc != RSP_OK) return rc;
  }
  if (dbias) {
    float* part = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(workspace) + w.rowgeom_bytes + w.partial_bytes);
    const int nblk = rsp_cdiv(p.M, 1024);
    hipLaunchKernelGGL(colsum_stage1, dim3(nblk, rsp_cdiv(d->Cout, 64)), dim3(256), 0, s, dy, (long long)p.M, d->Cout,
                       d->out_ld, part, 1024);
    rc = rsp_check_launch("colsum_stage1");
    if (rc != RSP_OK) return rc;
    hipLaunchKernelGGL(colsum_stage2, dim3(rsp_cdiv(d->Cout, 256)), dim3(256), 0, s, part, nblk, d->Cout, dbias);
    rc = rsp_check_launch("colsum_stage2");
  }
  return rc;
}

}  // namespace

