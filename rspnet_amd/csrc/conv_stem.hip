// Direct convolution for the network stems (3 input channels padded to 4: C3D conv1 models/c3d.py:21, R3D conv1
// models/resnet.py:124, the (1,7,7) stems of S3D-G models/s3dg.py:207 and R(2+1)D models/r2plus1d_vcop.py:70).
//
// With Cin = 4 the implicit-GEMM gather moves one 16-byte pixel per lane per tap: every K-chunk re-derives 4 bounds
// checks per copy and the texture path sees 64 unrelated addresses per instruction; on the short-K stems (27..49 taps)
// that, plus the per-tile prologue/epilogue, holds the generic kernel at 57-82 TFLOP/s (profiles/r01/stem_kernel.txt).
// Here the im2col expansion happens at LDS-read time instead:
//
//   * a workgroup owns a TH x TW (8x16 or 16x8) patch of one output frame = 128 GEMM rows, all (<= 64) output channels;
//   * the input halo of that patch is copied ONCE per kernel-time-slice into LDS by LDS-DMA, as whole pixel rows
//     (coalesced, zero-filled outside the image by the buffer descriptor's range check) -- for stride 2 the even and odd
//     columns are stored de-interleaved so that a tap's 16 neighbouring output columns are 16 adjacent pixels;
//   * the A operand of the fp32 MFMA (32x32x2) for row m and tap (kt,kh,kw) is then the pixel at
//     rowbase(m) + tapoffset(kt,kh,kw): one ds_read_b128 (4 channels) feeds 4 MFMA k-steps.  K is ordered so that the
//     two k-lanes of a step hold the SAME channel of two adjacent taps: lanes 0-31 read tap 2g, lanes 32-63 tap 2g+1;
//   * weights are packed [tap][cout 64][ci 4], streamed through LDS in chunks of TCH taps (double buffered LDS-DMA).
//
// Output, bias and the per-tile BatchNorm partial sums follow the implicit-GEMM kernel's conventions (one stat tile per
// workgroup; rows outside the image contribute zeros).
#include "conv_stem.h"

#include <mutex>
#include <unordered_map>

namespace {

struct StemParams {
  const float* __restrict__ x;
  const float* __restrict__ w;     // [nchunks*TCH][64][4]
  const float* __restrict__ bias;  // nullable
  float* __restrict__ y;
  float* __restrict__ stat;        // nullable: [tiles][Cout][2]
  int N, Di, Hi, Wi, Do, Ho, Wo;
  int kT, kH, kW, sT, sH, sW, pT, pH, pW;
  int Cout, out_ld;
  int tw_shift;                    // patch width 16 (4) or 8 (3); patch height = 128 >> tw_shift
  int HT, WT, WTh, WTL;            // halo rows, halo columns, ceil(WT/2), LDS row length in pixels
  int npix, npix_r;                // pixels per halo frame, rounded up to a multiple of 64
  int FR;                          // frame ring size
  int ntaps, nchunks;
  int tiles_h, tiles_w, tiles;
  unsigned x_bytes, w_bytes;
  int wide;   // epilogue through LDS + 16-byte stores (Cout % 4 == 0, 16-byte aligned output rows and bias)
};

// Wide epilogue store.  A wave's 32 x 64 accumulator tile is column-per-lane (lane = output channel), so a direct store is 32
// four-byte stores per lane, each wave-instruction touching two 128-byte half rows.  Here the tile goes through a 2 KB
// wave-private LDS stage ([16 rows][32 cols], conflict-free both ways) in four rounds and leaves as two 16-byte stores per lane
// and round — eight rows x 128 contiguous bytes per wave-instruction, a quarter of the store instructions.  Needs Cout % 4 == 0
// and 16-byte aligned rows.
//
// The bias comes in REGISTERS, loaded and waited for once per kernel (StemBias): a bias load whose first use sits inside the
// `if (row valid)` around a store makes hipcc put `s_waitcnt vmcnt(0)` in front of EVERY store (the skip path has not waited, so
// after each merge the load counts as pending again) — and vmcnt(0) also waits for the previous store's acknowledgement: the
// stores went out one at a time, ~0.2 us each, 6 of the 10 us a C3D conv1 patch took (ISA: /tmp listing in experiments_r4.txt).
struct StemBias {
  floatx4 w[2];   // wide form: channels 32 j + 4 (lane & 7) .. + 3
  float n[2];     // narrow form: channel 32 j + (lane & 31)
};

__device__ __forceinline__ StemBias stem_bias(const float* __restrict__ bias, int Cout, bool wide, int lane) {
  StemBias b;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    b.w[j] = floatx4{0.f, 0.f, 0.f, 0.f};
    b.n[j] = 0.f;
    const int cw = 32 * j + 4 * (lane & 7), cn = 32 * j + (lane & 31);
    if (bias && wide && cw < Cout) b.w[j] = *reinterpret_cast<const floatx4*>(bias + cw);
    if (bias && !wide && cn < Cout) b.n[j] = bias[cn];
  }
  // the loads are consumed HERE, outside any branch
  asm volatile("" : "+v"(b.w[0]), "+v"(b.w[1]), "+v"(b.n[0]), "+v"(b.n[1]));
  return b;
}

__device__ __forceinline__ void stem_store_wide(const StemParams& p, const floatx16 (&acc)[2], float* stage /* wave-private 2 KB */,
                                                const long long* rowaddr /* [128] */, int wave, int lane, const StemBias& b) {
  const int l32 = lane & 31, h = lane >> 5;
  const int q = lane & 7, rbase = lane >> 3;   // read side: 8 lanes per row, 4 channels each
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = 32 * j + 4 * q;
#pragma unroll
    for (int r = 0; r < 2; ++r) {              // accumulator elements 8r .. 8r+7 = tile rows 16r .. 16r+15
#pragma unroll
      for (int e = 0; e < 8; ++e) stage[((e >> 2) * 8 + h * 4 + (e & 3)) * 32 + l32] = acc[j][8 * r + e];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int rl = rbase + 8 * i;
        const floatx4 v = *reinterpret_cast<const floatx4*>(stage + rl * 32 + 4 * q);
        const long long addr = rowaddr[wave * 32 + 16 * r + rl];
        if (addr >= 0 && col < p.Cout) *reinterpret_cast<floatx4*>(p.y + addr + col) = v + b.w[j];
      }
    }
  }
}

// NS = k-steps (input channels) per tap: 4, or 3 when the filters were packed from three real input channels (RGB padded to 4 for
// 16-byte pixels: channel 3 is zero on both operands, so its k-step — a quarter of the MFMAs — is not issued; rsp_stem_note_packed).
template <int G, int NS>
__global__ __launch_bounds__(256, 2) void stem_kernel(const StemParams p) {
  constexpr int TCH = 2 * G;            // taps per weight chunk
  constexpr int BU = TCH * 64;          // 16-byte units per weight chunk
  constexpr int BI = (BU + 255) / 256;  // copy rounds per thread
  typedef __attribute__((address_space(3))) void* lptr_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* halo = reinterpret_cast<float*>(smem_raw);                            // [FR][npix_r][4]
  float* Bs = halo + p.FR * p.npix_r * 4;                                      // [2][TCH][64][4]
  int* taptab = reinterpret_cast<int*>(Bs + 2 * BU * 4);                       // [nchunks*TCH] byte offset into halo
  long long* rowaddr = reinterpret_cast<long long*>(taptab + p.nchunks * TCH);  // [128]

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l32 = lane & 31, h = lane >> 5;

  const int tile = rsp_xcd_remap(blockIdx.x, gridDim.x);
  const int wb = tile % p.tiles_w;
  int q0 = tile / p.tiles_w;
  const int hb = q0 % p.tiles_h;
  q0 /= p.tiles_h;
  const int to = q0 % p.Do;
  const int n = q0 / p.Do;
  const int TW = 1 << p.tw_shift, TH = 128 >> p.tw_shift;
  const int h0 = hb * TH, w0 = wb * TW;
  const StemBias bias = stem_bias(p.bias, p.Cout, p.wide != 0, lane);

  const int khw = p.kH * p.kW;
  for (int i = t; i < p.nchunks * TCH; i += 256) {
    int off = 0;   // padding taps carry zero weights: any in-range pixel will do
    if (i < p.ntaps) {
      const int kw = i % p.kW, r = i / p.kW;
      const int kh = r % p.kH, kt = r / p.kH;
      const int col = p.sW == 2 ? (kw & 1) * p.WTh + (kw >> 1) : kw;
      off = (((kt % p.FR) * p.npix_r) + kh * p.WTL + col) * 16;
    }
    taptab[i] = off;
  }

  // this thread's halo pixels (the same for every frame): byte offset inside an input frame, or "outside"
  unsigned hrel[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = i * 256 + t;
    unsigned r = 0xffffffffu;
    if (q < p.npix) {
      const int hyy = q / p.WTL, cc = q - hyy * p.WTL;
      const int wxp = p.sW == 2 ? (cc < p.WTh ? 2 * cc : 2 * (cc - p.WTh) + 1) : cc;
      const int hi = h0 * p.sH - p.pH + hyy, wi = w0 * p.sW - p.pW + wxp;
      if (wxp < p.WT && (unsigned)hi < (unsigned)p.Hi && (unsigned)wi < (unsigned)p.Wi) r = (unsigned)((hi * p.Wi + wi) * 16);
    }
    hrel[i] = r;
  }

  __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, (int)p.w_bytes, 0x00020000);

  auto load_frame = [&](int f) {   // kernel time-slice f of this patch -> ring slot f % FR
    const int ti = to * p.sT - p.pT + f;
    const bool okf = (unsigned)ti < (unsigned)p.Di;
    const unsigned fbase = (unsigned)(((long long)(n * p.Di + (okf ? ti : 0)) * p.Hi * p.Wi) * 16);
    const int slot = f % p.FR;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i * 256 + wave * 64 < p.npix_r) {   // wave-uniform
        const unsigned off = (okf & (hrel[i] != 0xffffffffu)) ? fbase + hrel[i] : 0xffffffffu;
        float* dst = halo + (slot * p.npix_r + i * 256 + wave * 64) * 4;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lptr_t)dst, 16, off, 0, 0, 0);
      }
    }
  };
  auto load_weights = [&](int c, int buf) {
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      if (i * 256 + wave * 64 < BU) {
        const unsigned off = ((unsigned)c * BU + i * 256 + t) * 16u;
        float* dst = Bs + (buf * BU + i * 256 + wave * 64) * 4;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lptr_t)dst, 16, off, 0, 0, 0);
      }
    }
  };
  auto frame_hi = [&](int c) { return min(p.kT - 1, (c * TCH + TCH - 1) / khw); };

  int loaded = frame_hi(0);
  for (int f = 0; f <= loaded; ++f) load_frame(f);
  load_weights(0, 0);
  __syncthreads();   // tap table written, copies landed (hipcc waits vmcnt(0) ahead of the barrier)

  // GEMM row of this lane's A operand: m = 32*wave + l32 -> (hy, wx) inside the patch
  const int m = wave * 32 + l32;
  const int hy = m >> p.tw_shift, wx = m & (TW - 1);
  const char* arow = reinterpret_cast<const char*>(halo) + ((hy * p.sH) * p.WTL + wx) * 16;
  int toff[G];
#pragma unroll
  for (int g = 0; g < G; ++g) toff[g] = taptab[2 * g + h];

  floatx16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

  int buf = 0;
  for (int c = 0; c < p.nchunks; ++c) {
    const bool more = c + 1 < p.nchunks;
    // 1. all LDS reads of the chunk (hipcc orders LDS reads behind pending LDS-DMA, so they go before the copies)
    floatx4 af[G], bf[G][2];
#pragma unroll
    for (int g = 0; g < G; ++g) af[g] = *reinterpret_cast<const floatx4*>(arow + toff[g]);
    const float* bb = Bs + buf * BU * 4 + (h * 64 + l32) * 4;
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[g][j] = *reinterpret_cast<const floatx4*>(bb + (2 * g * 64 + 32 * j) * 4);
    int toffn[G];
#pragma unroll
    for (int g = 0; g < G; ++g) toffn[g] = more ? taptab[(c + 1) * TCH + 2 * g + h] : 0;
    // 2. next chunk's weights (and a new time-slice of the halo when the taps move on to it) under this chunk's MFMAs
    if (more) {
      load_weights(c + 1, buf ^ 1);
      const int fh = frame_hi(c + 1);
      while (loaded < fh) load_frame(++loaded);
    }
    // 3. NS k-steps (channels) x 2 column tiles per tap pair
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g][s], bf[g][j][s], acc[j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    buf ^= 1;
#pragma unroll
    for (int g = 0; g < G; ++g) toff[g] = toffn[g];
  }

  // ---- epilogue ---------------------------------------------------------------------------------------------------
  if (t < 128) {
    const int ry = t >> p.tw_shift, rx = t & (TW - 1);
    const int ho = h0 + ry, wo = w0 + rx;
    rowaddr[t] = (ho < p.Ho && wo < p.Wo) ? ((((long long)n * p.Do + to) * p.Ho + ho) * p.Wo + wo) * p.out_ld : -1;
  }
  __syncthreads();
  float* red = Bs;   // [4 waves][64][2]; the weight buffers are idle now
  if (p.wide) stem_store_wide(p, acc, Bs + 512 + wave * 512, rowaddr, wave, lane, bias);   // behind the 2 KB of `red`
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = 32 * j + l32;
    const float bv = bias.n[j];
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const long long addr = rowaddr[wave * 32 + (e >> 2) * 8 + h * 4 + (e & 3)];
      const float v = addr >= 0 ? acc[j][e] : 0.f;
      if (!p.wide && addr >= 0 && col < p.Cout) p.y[addr + col] = v + bv;
      s += v;
      ss = fmaf(v, v, ss);
    }
    s += __shfl_xor(s, 32);
    ss += __shfl_xor(ss, 32);
    if (h == 0) {
      red[(wave * 64 + col) * 2 + 0] = s;
      red[(wave * 64 + col) * 2 + 1] = ss;
    }
  }
  if (p.stat) {
    __syncthreads();
    if (t < 64 && t < p.Cout) {
      float s = 0.f, ss = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        s += red[(w * 64 + t) * 2 + 0];
        ss += red[(w * 64 + t) * 2 + 1];
      }
      float* o = p.stat + ((long long)tile * p.Cout + t) * 2;
      o[0] = s;
      o[1] = ss;
    }
  }
}

// Persistent variant for stems whose whole weight tensor and all kernel time-slices of two halos fit in LDS (C3D conv1:
// 28 KB + 2 x 9 KB; the (1,7,7) stems: 50 KB + 2 x 13 KB).  A workgroup loads the weights and the tap table once and then walks
// the patches with stride gridDim.x; the halo of the NEXT patch is copied (LDS-DMA) into the other halo buffer while the last
// tap chunk of the current patch is on the matrix pipe, so a patch costs neither a weight copy nor an exposed copy latency
// (PMC before: 46 % of the wave time parked on s_waitcnt / s_barrier).
template <int G, bool WIDE, int NS>
__global__ __launch_bounds__(256, 2) void stem_resident_kernel(const StemParams p) {
  constexpr int TCH = 2 * G;
  constexpr int BU = TCH * 64;
  constexpr int BI = (BU + 255) / 256;
  typedef __attribute__((address_space(3))) void* lptr_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int hsz = p.FR * p.npix_r * 4;                                          // floats per halo buffer
  float* halo = reinterpret_cast<float*>(smem_raw);                            // [2][FR][npix_r][4]
  float* Bs = halo + 2 * hsz;                                                  // [nchunks][TCH][64][4]
  int* taptab = reinterpret_cast<int*>(Bs + p.nchunks * BU * 4);               // [nchunks*TCH]
  long long* rowaddr = reinterpret_cast<long long*>(taptab + p.nchunks * TCH);  // [128]
  float* red = reinterpret_cast<float*>(rowaddr + 128);                        // [4 waves][64][2]

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l32 = lane & 31, h = lane >> 5;
  const int TW = 1 << p.tw_shift, TH = 128 >> p.tw_shift;

  for (int i = t; i < p.nchunks * TCH; i += 256) {
    int off = 0;
    if (i < p.ntaps) {
      const int kw = i % p.kW, r = i / p.kW;
      const int kh = r % p.kH, kt = r / p.kH;
      const int col = p.sW == 2 ? (kw & 1) * p.WTh + (kw >> 1) : kw;
      off = ((kt * p.npix_r) + kh * p.WTL + col) * 16;      // FR == kT here: slot = kt
    }
    taptab[i] = off;
  }
  __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, (int)p.w_bytes, 0x00020000);
  for (int c = 0; c < p.nchunks; ++c) {
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      if (i * 256 + wave * 64 < BU) {
        const unsigned off = ((unsigned)c * BU + i * 256 + t) * 16u;
        float* dst = Bs + (c * BU + i * 256 + wave * 64) * 4;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lptr_t)dst, 16, off, 0, 0, 0);
      }
    }
  }
  // this thread's (up to 4) halo pixels in patch-relative coordinates: the same for every patch
  int hyy[4], wxp[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = i * 256 + t;
    hyy[i] = -(1 << 20);   // "no pixel": fails every range check below
    wxp[i] = 0;
    if (q < p.npix) {
      const int r = q / p.WTL, cc = q - r * p.WTL;
      const int wx_ = p.sW == 2 ? (cc < p.WTh ? 2 * cc : 2 * (cc - p.WTh) + 1) : cc;
      if (wx_ < p.WT) {
        hyy[i] = r;
        wxp[i] = wx_;
      }
    }
  }
  auto issue_halo = [&](int tile, int buf) {   // every kernel time-slice of patch `tile` -> halo buffer `buf`
    const int wb = tile % p.tiles_w;
    int q0 = tile / p.tiles_w;
    const int hb = q0 % p.tiles_h;
    q0 /= p.tiles_h;
    const int to = q0 % p.Do, n = q0 / p.Do;
    const int hi0 = hb * TH * p.sH - p.pH, wi0 = wb * TW * p.sW - p.pW;
    unsigned hrel[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int hi = hi0 + hyy[i], wi = wi0 + wxp[i];
      hrel[i] = ((unsigned)hi < (unsigned)p.Hi && (unsigned)wi < (unsigned)p.Wi) ? (unsigned)((hi * p.Wi + wi) * 16) : 0xffffffffu;
    }
    for (int f = 0; f < p.kT; ++f) {
      const int ti = to * p.sT - p.pT + f;
      const bool okf = (unsigned)ti < (unsigned)p.Di;
      const unsigned fbase = (unsigned)(((long long)(n * p.Di + (okf ? ti : 0)) * p.Hi * p.Wi) * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i * 256 + wave * 64 < p.npix_r) {
          const unsigned off = (okf & (hrel[i] != 0xffffffffu)) ? fbase + hrel[i] : 0xffffffffu;
          float* dst = halo + buf * hsz + (f * p.npix_r + i * 256 + wave * 64) * 4;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lptr_t)dst, 16, off, 0, 0, 0);
        }
      }
    }
  };

  // GEMM row of this lane's A operand: m = 32*wave + l32 -> (hy, wx) inside the patch
  const int m = wave * 32 + l32;
  const int rowoff = (((m >> p.tw_shift) * p.sH) * p.WTL + (m & (TW - 1))) * 16;
  const StemBias bias = stem_bias(p.bias, p.Cout, WIDE, lane);

  int cur = 0;
  if ((int)blockIdx.x < p.tiles) issue_halo(blockIdx.x, 0);
  for (int tile = blockIdx.x; tile < p.tiles; tile += gridDim.x) {
    __syncthreads();   // halo[cur] (and, first time, weights + tap table) landed; the previous patch's epilogue is over
    const int next = tile + gridDim.x;
    const char* arow = reinterpret_cast<const char*>(halo + cur * hsz) + rowoff;
    floatx16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    for (int c = 0; c < p.nchunks; ++c) {
      floatx4 af[G], bf[G][2];
#pragma unroll
      for (int g = 0; g < G; ++g) af[g] = *reinterpret_cast<const floatx4*>(arow + taptab[c * TCH + 2 * g + h]);
      const float* bb = Bs + c * BU * 4 + (h * 64 + l32) * 4;
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[g][j] = *reinterpret_cast<const floatx4*>(bb + (2 * g * 64 + 32 * j) * 4);
      // all LDS reads of the patch are issued: start the next patch's halo under this (last) chunk's MFMAs
      if (c == p.nchunks - 1 && next < p.tiles) issue_halo(next, cur ^ 1);
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g][s], bf[g][j][s], acc[j], 0, 0, 0);
    }
    // ---- epilogue (as stem_kernel) ----
    const int wb = tile % p.tiles_w;
    int q0 = tile / p.tiles_w;
    const int hb = q0 % p.tiles_h;
    q0 /= p.tiles_h;
    const int to = q0 % p.Do, n = q0 / p.Do;
    if (t < 128) {
      const int ho = hb * TH + (t >> p.tw_shift), wo = wb * TW + (t & (TW - 1));
      rowaddr[t] = (ho < p.Ho && wo < p.Wo) ? ((((long long)n * p.Do + to) * p.Ho + ho) * p.Wo + wo) * p.out_ld : -1;
    }
    __syncthreads();
    // wide form: staged through this patch's halo buffer — every wave is past its last operand read (the barrier above), and
    // the next copy into THIS buffer is issued behind the next patch's top barrier
    if (WIDE) stem_store_wide(p, acc, halo + cur * hsz + wave * 512, rowaddr, wave, lane, bias);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = 32 * j + l32;
      const float bv = bias.n[j];
      float s = 0.f, ss = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const long long addr = rowaddr[wave * 32 + (e >> 2) * 8 + h * 4 + (e & 3)];
        const float v = addr >= 0 ? acc[j][e] : 0.f;
        if (!WIDE && addr >= 0 && col < p.Cout) p.y[addr + col] = v + bv;
        s += v;
        ss = fmaf(v, v, ss);
      }
      s += __shfl_xor(s, 32);
      ss += __shfl_xor(ss, 32);
      if (h == 0) {
        red[(wave * 64 + col) * 2 + 0] = s;
        red[(wave * 64 + col) * 2 + 1] = ss;
      }
    }
    if (p.stat) {
      __syncthreads();
      if (t < 64 && t < p.Cout) {
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          s += red[(w * 64 + t) * 2 + 0];
          ss += red[(w * 64 + t) * 2 + 1];
        }
        float* o = p.stat + ((long long)tile * p.Cout + t) * 2;
        o[0] = s;
        o[1] = ss;
      }
    }
    cur ^= 1;
  }
}

__global__ void stem_pack_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int kT, int kH, int kW,
                                 int ntaps, long long total) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int ci = (int)(i & 3), n = (int)((i >> 2) & 63), tap = (int)(i >> 8);
    float v = 0.f;
    if (tap < ntaps && n < Cout) {
      const int kw = tap % kW, r = tap / kW;
      const int kh = r % kH, kt = r / kH;
      v = w[((((long long)n * 4 + ci) * kT + kt) * kH + kh) * kW + kw];
    }
    out[i] = v;
  }
}

struct StemPlan {
  bool ok;
  int G, TCH, nchunks, tw_shift, HT, WT, WTh, WTL, npix, npix_r, FR, tiles_h, tiles_w;
  bool resident;      // stem_resident_kernel: weights + two halos fit in LDS
  bool wide_ok;       // 16-byte epilogue stores possible (channel count / pitch; pointer alignment is checked at launch)
  long long tiles;
  size_t lds;
};

StemPlan stem_plan(const rsp_conv3d_desc* d) {
  StemPlan s;
  memset(&s, 0, sizeof s);
  if (d->Cin != 4 || d->in_ld != 4 || d->Cout > 64 || d->out_ld < d->Cout) return s;
  if (d->sW < 1 || d->sW > 2 || d->sH < 1 || d->sH > 2) return s;
  const int ntaps = d->kT * d->kH * d->kW;
  if (ntaps > 343) return s;
  if ((unsigned long long)d->N * d->Di * d->Hi * d->Wi * 16ull >= 0xF0000000ull) return s;
  // patch shape: the one that wastes fewer rows on the frame borders
  const long long waste16 = (long long)rsp_cdiv(d->Ho, 8) * 8 * rsp_cdiv(d->Wo, 16) * 16;
  const long long waste8 = (long long)rsp_cdiv(d->Ho, 16) * 16 * rsp_cdiv(d->Wo, 8) * 8;
  s.tw_shift = waste8 < waste16 ? 3 : 4;
  // Long-K stems (R3D 7x7x7: 343 taps) already run the implicit-GEMM kernel at ~105 TFLOP/s; the patch kernel only wins
  // there when the frame tiles without border waste (measured: profiles/r01/stem_kernel.txt).
  const long long covered = waste8 < waste16 ? waste8 : waste16;
  if (ntaps > 64 && covered * 100 > (long long)d->Ho * d->Wo * 104) return s;
  const int TW = 1 << s.tw_shift, TH = 128 >> s.tw_shift;
  s.HT = (TH - 1) * d->sH + d->kH;
  s.WT = (TW - 1) * d->sW + d->kW;
  s.WTh = (s.WT + 1) / 2;
  s.WTL = d->sW == 2 ? 2 * s.WTh : s.WT;
  s.npix = s.HT * s.WTL;
  if (s.npix > 1024) return s;
  s.npix_r = (s.npix + 63) / 64 * 64;
  s.FR = d->kT < 3 ? d->kT : 3;
  // taps per weight chunk: least zero padding, then the longest chunk.  With a ring of 3 time-slices a chunk and its
  // successor must not span more than two slices unless every slice is resident.
  int best = 0, best_pad = 1 << 30;
  const int cand[3] = {8, 7, 5};
  for (int i = 0; i < 3; ++i) {
    const int tch = 2 * cand[i];
    if (d->kT > 3 && d->kH * d->kW < 2 * tch) continue;
    const int pad = rsp_cdiv(ntaps, tch) * tch;
    if (pad < best_pad) {
      best_pad = pad;
      best = cand[i];
    }
  }
  if (!best) return s;
  s.G = best;
  s.TCH = 2 * best;
  s.nchunks = rsp_cdiv(ntaps, s.TCH);
  s.tiles_h = rsp_cdiv(d->Ho, TH);
  s.tiles_w = rsp_cdiv(d->Wo, TW);
  s.tiles = (long long)d->N * d->Do * s.tiles_h * s.tiles_w;
  if (s.tiles >= (1ll << 31)) return s;
  const size_t lds_res = (size_t)2 * d->kT * s.npix_r * 16 + (size_t)s.nchunks * s.TCH * 1024 + (size_t)s.nchunks * s.TCH * 4 +
                         128 * 8 + 2048;
  s.resident = d->kT <= 3 && lds_res <= 78 * 1024;      // two workgroups per CU at least
  s.lds = s.resident ? lds_res
                     : (size_t)s.FR * s.npix_r * 16 + (size_t)2 * s.TCH * 1024 + (size_t)s.nchunks * s.TCH * 4 + 128 * 8;
  if (s.lds > 150 * 1024) return s;
  // (the resident kernel stages the wide stores through a halo buffer: 4 waves x 2 KB)
  // ... and with 16 taps per chunk its wide form needs 178 VGPRs — two waves per SIMD instead of three)
  s.wide_ok = d->Cout % 4 == 0 && d->out_ld % 4 == 0 && (!s.resident || ((size_t)d->kT * s.npix_r * 16 >= 8192 && s.G != 8));
  s.ok = true;
  return s;
}

// Packed stem weights whose source filters had three input channels (noted by the re-pack entry points: the layout is private to
// the library, so every producer of such a buffer passes through one of them).  Keyed by the packed buffer's address; the owner
// of the buffer withdraws the mark before it releases the memory (rsp_conv3d_pack_forget), so a recycled address starts unmarked.
std::mutex g_three_mu;
std::unordered_map<const void*, bool> g_three;

bool stem_three_channels(const void* w_packed) {
  std::lock_guard<std::mutex> lk(g_three_mu);
  auto it = g_three.find(w_packed);
  return it != g_three.end() && it->second;
}

template <int G, int NS>
int launch_stem(const StemParams& p, const StemPlan& pl, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_kernel<G, NS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              150 * 1024);
    attr_set = true;
  }
  if (pl.resident) {
    static bool attr_set_r = false;
    if (!attr_set_r) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_resident_kernel<G, true, NS>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_resident_kernel<G, false, NS>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
      attr_set_r = true;
    }
    long long per_cu = (160 * 1024) / (long long)pl.lds;
    per_cu = per_cu > 3 ? 3 : per_cu;                       // <= 170 VGPRs: three waves per SIMD
    const long long grid = pl.tiles < 256 * per_cu ? pl.tiles : 256 * per_cu;
    rsp_note_kernel(p.wide ? "stem_resident_kernel<%d, true, %d>" : "stem_resident_kernel<%d, false, %d>", G, NS);
    if (p.wide) hipLaunchKernelGGL((stem_resident_kernel<G, true, NS>), dim3((unsigned)grid), dim3(256), pl.lds, s, p);
    else hipLaunchKernelGGL((stem_resident_kernel<G, false, NS>), dim3((unsigned)grid), dim3(256), pl.lds, s, p);
    return rsp_check_launch("stem_resident_kernel");
  }
  rsp_note_kernel("stem_kernel<%d, %d>", G, NS);
  hipLaunchKernelGGL((stem_kernel<G, NS>), dim3((unsigned)pl.tiles), dim3(256), pl.lds, s, p);
  return rsp_check_launch("stem_kernel");
}

}  // namespace

bool rsp_stem_applicable(const rsp_conv3d_desc* d) { return stem_plan(d).ok; }

int rsp_stem_tiles(const rsp_conv3d_desc* d) { return (int)stem_plan(d).tiles; }

// (the three-channel instance: what the engine's re-pack produces for every RGB stem; a four-channel source runs <..., 4>)
const char* rsp_stem_kernel_name(const rsp_conv3d_desc* d) {
  const StemPlan pl = stem_plan(d);
  if (pl.resident && pl.wide_ok)
    return pl.G == 8 ? "stem_resident_kernel<8, true, 3>" : (pl.G == 7 ? "stem_resident_kernel<7, true, 3>" : "stem_resident_kernel<5, true, 3>");
  if (pl.resident)
    return pl.G == 8 ? "stem_resident_kernel<8, false, 3>" : (pl.G == 7 ? "stem_resident_kernel<7, false, 3>" : "stem_resident_kernel<5, false, 3>");
  return pl.G == 8 ? "stem_kernel<8, 3>" : (pl.G == 7 ? "stem_kernel<7, 3>" : "stem_kernel<5, 3>");
}

void rsp_stem_note_packed(const void* w_packed, bool three_channels) {
  std::lock_guard<std::mutex> lk(g_three_mu);
  if (three_channels) g_three[w_packed] = true;
  else g_three.erase(w_packed);                 // (unmarked = four channel steps: only marks are kept)
}

void rsp_stem_forget_packed(const void* w_packed) {
  std::lock_guard<std::mutex> lk(g_three_mu);
  g_three.erase(w_packed);
}

size_t rsp_stem_packed_elems(const rsp_conv3d_desc* d) {
  const StemPlan pl = stem_plan(d);
  return (size_t)pl.nchunks * pl.TCH * 256;
}

int rsp_stem_pack(const rsp_conv3d_desc* d, const float* w_ref, float* w_packed, hipStream_t s) {
  const StemPlan pl = stem_plan(d);
  const long long total = (long long)pl.nchunks * pl.TCH * 256;
  const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
  rsp_stem_note_packed(w_packed, false);      // this entry point is handed four-channel filters
  hipLaunchKernelGGL(stem_pack_kernel, dim3(blocks), dim3(256), 0, s, w_ref, w_packed, d->Cout, d->kT, d->kH, d->kW,
                     d->kT * d->kH * d->kW, total);
  return rsp_check_launch("stem_pack_kernel");
}

int rsp_stem_fwd(const rsp_conv3d_desc* d, const float* x, const float* w_packed, const float* bias, float* y,
                 float* stat_partials, hipStream_t s) {
  const StemPlan pl = stem_plan(d);
  RSP_REQUIRE(pl.ok, "rsp_stem_fwd: descriptor not on the stem path");
  RSP_REQUIRE(rsp_aligned16(x) && rsp_aligned16(w_packed), "rsp_conv3d_fwd: 4-channel input and packed weight must be 16-byte aligned");
  StemParams p;
  memset(&p, 0, sizeof p);
  p.x = x; p.w = w_packed; p.bias = bias; p.y = y; p.stat = stat_partials;
  p.N = d->N; p.Di = d->Di; p.Hi = d->Hi; p.Wi = d->Wi; p.Do = d->Do; p.Ho = d->Ho; p.Wo = d->Wo;
  p.kT = d->kT; p.kH = d->kH; p.kW = d->kW; p.sT = d->sT; p.sH = d->sH; p.sW = d->sW;
  p.pT = d->pT; p.pH = d->pH; p.pW = d->pW;
  p.Cout = d->Cout; p.out_ld = d->out_ld;
  p.tw_shift = pl.tw_shift;
  p.HT = pl.HT; p.WT = pl.WT; p.WTh = pl.WTh; p.WTL = pl.WTL;
  p.npix = pl.npix; p.npix_r = pl.npix_r; p.FR = pl.FR;
  p.ntaps = d->kT * d->kH * d->kW; p.nchunks = pl.nchunks;
  p.tiles_h = pl.tiles_h; p.tiles_w = pl.tiles_w; p.tiles = (int)pl.tiles;
  // wide epilogue: staged through the idle weight ring (streaming variant) or through the patch's own halo buffer (resident
  // variant — a stage of its own cost the third workgroup per CU: C3D conv1 1.02 -> 1.11 ms)
  p.wide = pl.wide_ok && rsp_aligned16(y) && (!bias || rsp_aligned16(bias));
  p.x_bytes = (unsigned)((unsigned long long)d->N * d->Di * d->Hi * d->Wi * 16ull);
  p.w_bytes = (unsigned)((size_t)pl.nchunks * pl.TCH * 1024);
  if (stem_three_channels(w_packed)) {
    switch (pl.G) {
      case 8: return launch_stem<8, 3>(p, pl, s);
      case 7: return launch_stem<7, 3>(p, pl, s);
      default: return launch_stem<5, 3>(p, pl, s);
    }
  }
  switch (pl.G) {
    case 8: return launch_stem<8, 4>(p, pl, s);
    case 7: return launch_stem<7, 4>(p, pl, s);
    default: return launch_stem<5, 4>(p, pl, s);
  }
}
