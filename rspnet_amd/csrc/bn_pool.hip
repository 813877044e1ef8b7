// BatchNorm3d (train mode) fused with ReLU / residual add / non-overlapping MaxPool3d — HBM-bound kernels.
// Reference call sites: models/c3d.py:22-24 (bn → relu → pool), models/resnet.py:61-77, models/s3dg.py:23,28-33,
// models/r2plus1d_vcop.py:59-60,116-123; semantics per SURVEY.md App. C (biased var for y, unbiased for running_var).
//
// The conv epilogue already produced per-128-row-tile (sum, sumsq) partials, so forward costs one read of y and one
// write of the (pooled) activation; backward recomputes xhat / ReLU mask / pool arg-max from the saved y instead of
// storing masks or indices.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------------------------
// statistics
// ------------------------------------------------------------------------------------------------------------------
// stage 1: [tiles][C][2] float partials -> [S][C][2] double partials
__global__ __launch_bounds__(256) void bn_finalize_stage1(const float* __restrict__ part, int tiles, int C, int ld, int S,
                                                          double* __restrict__ out) {
  __shared__ double red[4][64][2];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const int per = (tiles + S - 1) / S;
  const int t0 = blockIdx.y * per, t1 = min(tiles, t0 + per);
  double s = 0.0, ss = 0.0;
  if (c < C)
    for (int t = t0 + rl; t < t1; t += 4) {
      const float2 v = *reinterpret_cast<const float2*>(part + ((long long)t * ld + c) * 2);
      s += (double)v.x;
      ss += (double)v.y;
    }
  red[rl][threadIdx.x & 63][0] = s;
  red[rl][threadIdx.x & 63][1] = ss;
  __syncthreads();
  if (rl == 0 && c < C) {
    const int l = threadIdx.x;
    out[((long long)blockIdx.y * C + c) * 2 + 0] = red[0][l][0] + red[1][l][0] + red[2][l][0] + red[3][l][0];
    out[((long long)blockIdx.y * C + c) * 2 + 1] = red[0][l][1] + red[1][l][1] + red[2][l][1] + red[3][l][1];
  }
}

// stage 2: one workgroup of 1024 threads per 64 channels — 16 slice-lanes per channel add the S fp64 slice sums (lane-strided,
// then a 16-way LDS tree read in index order: fixed order), the first 64 threads finalize.  (The first version ran one thread per
// channel over all S = 256 slices on one or two CUs: 50-63 us per BatchNorm of the big layers — C3D conv1 / conv2, R(2+1)D's
// 144-channel layers — for a few KB of data.)
__global__ __launch_bounds__(1024) void bn_finalize_stage2(const double* __restrict__ part, int S, int C, int Cv, long long count,
                                                           const float* __restrict__ conv_bias, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, float momentum,
                                                           float* __restrict__ running_mean, float* __restrict__ running_var,
                                                           float* __restrict__ mean_invstd, float* __restrict__ scale_shift,
                                                           float* __restrict__ bstat) {
  __shared__ double red[16][64][2];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  double s = 0.0, ss = 0.0;
  if (c < C)
    for (int i = rl; i < S; i += 16) {
      s += part[((long long)i * C + c) * 2 + 0];
      ss += part[((long long)i * C + c) * 2 + 1];
    }
  red[rl][cl][0] = s;
  red[rl][cl][1] = ss;
  __syncthreads();
  if (rl != 0 || c >= C) return;
  s = 0.0;
  ss = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    s += red[i][cl][0];
    ss += red[i][cl][1];
  }
  const double n = (double)count;
  const double mean0 = s / n;                   // mean of the bias-free conv output
  double var = ss / n - mean0 * mean0;          // biased
  var = var > 0.0 ? var : 0.0;
  // channels [Cv, C) are zero padding of the convolution (R(2+1)D's odd mid-channel counts run padded to a multiple of 4): the
  // parameter vectors hold Cv entries, the pad channels get gamma = beta = 0 and move no running statistics
  const bool valid = c < Cv;
  const double mean = mean0 + ((conv_bias && valid) ? (double)conv_bias[c] : 0.0);
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  mean_invstd[c] = (float)mean;
  mean_invstd[C + c] = invstd;
  const float g = valid ? (gamma ? gamma[c] : 1.f) : 0.f, b = (valid && beta) ? beta[c] : 0.f;
  const float sc = g * invstd;
  scale_shift[c] = sc;
  scale_shift[C + c] = b - (float)mean * sc;
  if (!valid) return;
  const double unbiased = count > 1 ? var * n / (n - 1.0) : var;
  if (bstat) {
    // deferred running-statistics update (rsp_bn_running_update): this pass only reports its batch moments — two passes through
    // the same BatchNorm can then run side by side and have their moving averages applied afterwards, in order
    bstat[c] = (float)mean;
    bstat[Cv + c] = (float)unbiased;
    return;
  }
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
  if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
}

// Single-launch variant for up to a few thousand tiles: one workgroup of 1024 threads per FIN_CH = 16 channels — 64 tile-lanes per
// channel reduce the [tiles][C][2] partials in fp64 (fixed order: lane-strided, then an 8 x 8 LDS tree read in index order), then
// the first 16 threads finalize their channel exactly as bn_finalize_stage2 does.  Replaces two ~6 us launches per BatchNorm by
// one (S3D-G runs 231 BatchNorm forwards per step).  (A first version gave a workgroup 64 channels x 16 lanes: a 64-channel layer
// with 2 048 tiles was then read by ONE compute unit — 17-29 us per BatchNorm on C3D / R(2+1)D, forward and backward.)
constexpr int FIN_CH = 16, FIN_LANES = 1024 / FIN_CH;

// Sum of the [n][C][2] fp32 partials (row pitch ld channels) of channel blockIdx.x * FIN_CH + (threadIdx.x % FIN_CH), in double.
// Returns true in the ONE thread per channel that holds the totals.
__device__ __forceinline__ bool fin_sum_partials(const float* __restrict__ part, int n, int C, int ld, double& s, double& ss) {
  __shared__ double red[FIN_LANES][FIN_CH][2];
  const int cl = threadIdx.x & (FIN_CH - 1), rl = threadIdx.x / FIN_CH;
  const int c = blockIdx.x * FIN_CH + cl;
  s = 0.0;
  ss = 0.0;
  if (c < C)
    for (int t = rl; t < n; t += FIN_LANES) {
      const float2 v = *reinterpret_cast<const float2*>(part + ((long long)t * ld + c) * 2);
      s += (double)v.x;
      ss += (double)v.y;
    }
  red[rl][cl][0] = s;
  red[rl][cl][1] = ss;
  __syncthreads();
  if (rl < 8) {
    s = 0.0;
    ss = 0.0;
#pragma unroll
    for (int i = 0; i < FIN_LANES / 8; ++i) {
      s += red[rl * (FIN_LANES / 8) + i][cl][0];
      ss += red[rl * (FIN_LANES / 8) + i][cl][1];
    }
  }
  __syncthreads();
  if (rl < 8) {
    red[rl][cl][0] = s;
    red[rl][cl][1] = ss;
  }
  __syncthreads();
  if (rl != 0 || c >= C) return false;
  s = 0.0;
  ss = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    s += red[i][cl][0];
    ss += red[i][cl][1];
  }
  return true;
}

__global__ __launch_bounds__(1024) void bn_finalize_one_kernel(const float* __restrict__ part, int tiles, int C, int Cv, int ld, long long count,
                                                               const float* __restrict__ conv_bias, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps, float momentum,
                                                               float* __restrict__ running_mean, float* __restrict__ running_var,
                                                               float* __restrict__ mean_invstd, float* __restrict__ scale_shift,
                                                               float* __restrict__ bstat) {
  const int c = blockIdx.x * FIN_CH + (threadIdx.x & (FIN_CH - 1));
  double s, ss;
  if (!fin_sum_partials(part, tiles, C, ld, s, ss)) return;
  const double n = (double)count;
  const double mean0 = s / n;                   // mean of the bias-free conv output
  double var = ss / n - mean0 * mean0;          // biased
  var = var > 0.0 ? var : 0.0;
  // channels [Cv, C) are zero padding of the convolution (R(2+1)D's odd mid-channel counts run padded to a multiple of 4): the
  // parameter vectors hold Cv entries, the pad channels get gamma = beta = 0 and move no running statistics
  const bool valid = c < Cv;
  const double mean = mean0 + ((conv_bias && valid) ? (double)conv_bias[c] : 0.0);
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  mean_invstd[c] = (float)mean;
  mean_invstd[C + c] = invstd;
  const float g = valid ? (gamma ? gamma[c] : 1.f) : 0.f, b = (valid && beta) ? beta[c] : 0.f;
  const float sc = g * invstd;
  scale_shift[c] = sc;
  scale_shift[C + c] = b - (float)mean * sc;
  if (!valid) return;
  const double unbiased = count > 1 ? var * n / (n - 1.0) : var;
  if (bstat) {
    // deferred running-statistics update (rsp_bn_running_update): this pass only reports its batch moments — two passes through
    // the same BatchNorm can then run side by side and have their moving averages applied afterwards, in order
    bstat[c] = (float)mean;
    bstat[Cv + c] = (float)unbiased;
    return;
  }
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
  if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
}

// standalone stats over 128-row tiles (same partial layout as the conv epilogue)
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ y, long long rows, int C, int ld,
                                                       float* __restrict__ part) {
  __shared__ float red[4][64][2];
  const int tile = blockIdx.x;
  const int c = blockIdx.y * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  float s = 0.f, ss = 0.f;
  if (c < C)
    for (int r = rl; r < 128; r += 4) {
      const long long row = (long long)tile * 128 + r;
      if (row >= rows) break;
      const float v = y[row * ld + c];
      s += v;
      ss = fmaf(v, v, ss);
    }
  red[rl][threadIdx.x & 63][0] = s;
  red[rl][threadIdx.x & 63][1] = ss;
  __syncthreads();
  if (rl == 0 && c < C) {
    const int l = threadIdx.x;
    part[((long long)tile * C + c) * 2 + 0] = red[0][l][0] + red[1][l][0] + red[2][l][0] + red[3][l][0];
    part[((long long)tile * C + c) * 2 + 1] = red[0][l][1] + red[1][l][1] + red[2][l][1] + red[3][l][1];
  }
}

// ------------------------------------------------------------------------------------------------------------------
// forward: out = maxpool(act(scale*y + shift + res))
// ------------------------------------------------------------------------------------------------------------------
struct PoolParams {
  rsp_pool3d_desc d;
  const float* __restrict__ y;
  const float* __restrict__ ss;   // [2][C] scale, shift
  const float* __restrict__ res;  // nullable
  const float* __restrict__ gate; // nullable: [N][C] per-sample channel gates applied after the activation (S3D-G self-gating)
  float* __restrict__ out;
  int relu;
  int cg;  // channel groups = C / VEC
  // thread layout: a block covers cgc (<= 256) channel groups x ppi output positions; blockIdx.y walks the channel chunks
  int cgc, ppi;
  long long npos;            // output positions (< 2^31)
  FastDiv dcgc, dWo, dHo, dDo;
};

// Every thread keeps ONE channel group for its whole life (per-channel constants are loaded once) and walks output positions;
// a position index is decoded by multiply-high constant division (the earlier kernels did five 64-bit divisions per element).
struct Layout {
  int cgc, ppi;
  FastDiv dcgc, dWo, dHo, dDo;
  long long npos;
  dim3 grid;
};

static Layout make_layout(const rsp_pool3d_desc* d, int cg, int per_thread) {
  Layout L;
  L.cgc = cg < 256 ? cg : 256;
  L.ppi = 256 / L.cgc;
  L.dcgc = fastdiv_make(L.cgc);
  L.dWo = fastdiv_make(d->Wo); L.dHo = fastdiv_make(d->Ho); L.dDo = fastdiv_make(d->Do);
  L.npos = (long long)d->N * d->Do * d->Ho * d->Wo;
  long long b = (L.npos + (long long)L.ppi * per_thread - 1) / ((long long)L.ppi * per_thread);
  b = b > 16384 ? 16384 : (b < 1 ? 1 : b);
  L.grid = dim3((unsigned)b, (unsigned)rsp_cdiv(cg, 256));
  return L;
}

template <int VEC>
__device__ __forceinline__ void load_vec(const float* p, float (&v)[VEC]) {
  if (VEC == 4) {
    const floatx4 t = *reinterpret_cast<const floatx4*>(p);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  } else {
    v[0] = p[0];
  }
}
template <int VEC>
__device__ __forceinline__ void store_vec(float* p, const float (&v)[VEC]) {
  if (VEC == 4) {
    floatx4 t = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<floatx4*>(p) = t;
  } else {
    p[0] = v[0];
  }
}

// offsets (in positions) of the NW window positions from the window's first one, scan order (kt, kh, kw)
template <int NW>
__device__ __forceinline__ void window_offsets(const rsp_pool3d_desc& d, long long (&off)[NW]) {
  int kw = 0, kh = 0, kt = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    off[w] = ((long long)kt * d.Hi + kh) * d.Wi + kw;
    if (++kw == d.kW) { kw = 0; if (++kh == d.kH) { kh = 0; ++kt; } }
  }
}

// MODE picks the body at compile time so that each keeps its own register allocation (one kernel holding all of them ran the streaming
// body at the 8-window body's occupancy): 0 generic windows (padding, overlap, gates), 1 no pooling and no gate (pure streaming),
// 2 / 3 disjoint un-padded windows of 4 / 8 positions.  bn_mode() on the host applies the same conditions.
template <int VEC, int MODE>
__global__ __launch_bounds__(256) void bn_act_pool_fwd_kernel(const PoolParams p) {
  const rsp_pool3d_desc& d = p.d;
  const int t = threadIdx.x;
  const int pl = fastdiv(t, p.dcgc);
  const int cgi = blockIdx.y * 256 + (t - pl * p.cgc);
  if (pl >= p.ppi || cgi >= p.cg) return;
  const int c = cgi * VEC;
  float sc[VEC], sh[VEC];
  load_vec<VEC>(p.ss + c, sc);
  load_vec<VEC>(p.ss + d.C + c, sh);
  const bool unit = d.kT * d.kH * d.kW == 1 && d.sT == 1 && d.sH == 1 && d.sW == 1 && !(d.pT | d.pH | d.pW);   // no pooling
  if constexpr (MODE == 1) {
    // plain streaming form, four positions per trip: the loads of a trip are independent, so a thread keeps 64-128 B in flight
    // instead of 16-32 (at full occupancy one float4 per thread is 8 MB in flight on the chip — under the ~12 MB that 6 TB/s x 2 us
    // need), and the four are CONSECUTIVE position groups: a block touches 16 KB contiguous per trip (4.9-5.2 -> 5.4-6.0 TB/s forward, 4.4-4.9 -> 5.1-5.5 backward on the big tensors, tools/bn_bw_probe.py; torch.mul streams 6.0 here)
    const long long step = p.ppi;                              // positions between a thread's four loads of one trip
    const long long stride = (long long)gridDim.x * p.ppi * 4;  // ... and between trips
    long long o = (long long)blockIdx.x * p.ppi * 4 + pl;
    for (; o + 3 * step < p.npos; o += stride) {
      float v[4][VEC], r[4][VEC];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        load_vec<VEC>(p.y + (o + u * step) * d.in_ld + c, v[u]);
        if (p.res) load_vec<VEC>(p.res + (o + u * step) * d.res_ld + c, r[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float best[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          float z = fmaf(v[u][e], sc[e], sh[e]);
          if (p.res) z += r[u][e];
          best[e] = p.relu ? fmaxf(z, 0.f) : z;
        }
        store_vec<VEC>(p.out + (o + u * step) * d.out_ld + c, best);
      }
    }
    for (int u = 0; u < 4 && o < p.npos; ++u, o += step) {      // the last, partial group of four
      float v[VEC], r[VEC], best[VEC];
      load_vec<VEC>(p.y + o * d.in_ld + c, v);
      if (p.res) load_vec<VEC>(p.res + o * d.res_ld + c, r);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float z = fmaf(v[e], sc[e], sh[e]);
        if (p.res) z += r[e];
        best[e] = p.relu ? fmaxf(z, 0.f) : z;
      }
      store_vec<VEC>(p.out + o * d.out_ld + c, best);
    }
    return;
  }
  if constexpr (MODE == 2 || MODE == 3) {
    // disjoint un-padded windows of 4 or 8 positions (C3D's pools): every window load of an output is issued before the first
    // compare — 64-128 B (x2 with a residual) in flight per thread instead of one float4 behind each bounds check.  Measured
    // (tools/bn_bw_probe.py, C3D conv1 / conv2 + pool): the backward pair 4.2-4.7 -> 5.2-5.3 TB/s, this forward kernel unchanged at 5.0-5.1
    {
      constexpr int NW = MODE == 2 ? 4 : 8;
      long long off[NW];                                       // wave-uniform offsets of the window positions, scan order
      window_offsets<NW>(d, off);
      for (long long o = (long long)blockIdx.x * p.ppi + pl; o < p.npos; o += (long long)gridDim.x * p.ppi) {
        const int op = (int)o;
        const int q1 = fastdiv(op, p.dWo), ow = op - q1 * d.Wo;
        const int q2 = fastdiv(q1, p.dHo), oh = q1 - q2 * d.Ho;
        const int n = fastdiv(q2, p.dDo), od = q2 - n * d.Do;
        const long long base = (((long long)n * d.Di + od * d.sT) * d.Hi + oh * d.sH) * d.Wi + ow * d.sW;
        float v[NW][VEC], r[NW][VEC], g[VEC], best[VEC];
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          load_vec<VEC>(p.y + (base + off[w]) * d.in_ld + c, v[w]);
          if (p.res) load_vec<VEC>(p.res + (base + off[w]) * d.res_ld + c, r[w]);
        }
        if (p.gate) load_vec<VEC>(p.gate + (long long)n * d.C + c, g);
#pragma unroll
        for (int e = 0; e < VEC; ++e) best[e] = -INFINITY;
#pragma unroll
        for (int w = 0; w < NW; ++w)
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            float z = fmaf(v[w][e], sc[e], sh[e]);
            if (p.res) z += r[w][e];
            if (p.relu) z = fmaxf(z, 0.f);
            if (p.gate) z *= g[e];
            best[e] = fmaxf(best[e], z);
          }
        store_vec<VEC>(p.out + o * d.out_ld + c, best);
      }
    }
    return;
  }
  for (long long o = (long long)blockIdx.x * p.ppi + pl; o < p.npos; o += (long long)gridDim.x * p.ppi) {
    float best[VEC];
    if (unit) {
      float v[VEC], r[VEC];
      load_vec<VEC>(p.y + o * d.in_ld + c, v);
      if (p.res) load_vec<VEC>(p.res + o * d.res_ld + c, r);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float z = fmaf(v[e], sc[e], sh[e]);
        if (p.res) z += r[e];
        best[e] = p.relu ? fmaxf(z, 0.f) : z;
      }
      if (p.gate) {
        const int n = fastdiv(fastdiv(fastdiv((int)o, p.dWo), p.dHo), p.dDo);
        float g[VEC];
        load_vec<VEC>(p.gate + (long long)n * d.C + c, g);
#pragma unroll
        for (int e = 0; e < VEC; ++e) best[e] *= g[e];
      }
    } else {
      const int op = (int)o;
      const int q1 = fastdiv(op, p.dWo), ow = op - q1 * d.Wo;
      const int q2 = fastdiv(q1, p.dHo), oh = q1 - q2 * d.Ho;
      const int n = fastdiv(q2, p.dDo), od = q2 - n * d.Do;
      float g[VEC];
      if (p.gate) load_vec<VEC>(p.gate + (long long)n * d.C + c, g);
#pragma unroll
      for (int e = 0; e < VEC; ++e) best[e] = -INFINITY;
      for (int kt = 0; kt < d.kT; ++kt) {
        const int id = od * d.sT - d.pT + kt;
        if ((unsigned)id >= (unsigned)d.Di) continue;
        for (int kh = 0; kh < d.kH; ++kh) {
          const int ih = oh * d.sH - d.pH + kh;
          if ((unsigned)ih >= (unsigned)d.Hi) continue;
          for (int kw = 0; kw < d.kW; ++kw) {
            const int iw = ow * d.sW - d.pW + kw;
            if ((unsigned)iw >= (unsigned)d.Wi) continue;
            const long long pos = (((long long)n * d.Di + id) * d.Hi + ih) * d.Wi + iw;
            float v[VEC];
            load_vec<VEC>(p.y + pos * d.in_ld + c, v);
            float r[VEC];
            if (p.res) load_vec<VEC>(p.res + pos * d.res_ld + c, r);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
              float z = fmaf(v[e], sc[e], sh[e]);
              if (p.res) z += r[e];
              if (p.relu) z = fmaxf(z, 0.f);
              if (p.gate) z *= g[e];      // (the gated activation is what the reference pools: models/s3dg.py:105-108)
              best[e] = fmaxf(best[e], z);
            }
          }
        }
      }
    }
    store_vec<VEC>(p.out + o * d.out_ld + c, best);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------------------------
struct BwdParams {
  rsp_pool3d_desc d;
  const float* __restrict__ y;
  const float* __restrict__ res;
  const float* __restrict__ dout;  // [N,Do,Ho,Wo,C] pitch out_ld
  const float* __restrict__ gamma;
  const float* __restrict__ mi;    // mean, invstd [2][C]
  const float* __restrict__ ss;    // scale, shift [2][C]
  const double* __restrict__ sums; // [C][2] (sum dz, sum dz*xhat)  (apply only)
  float* __restrict__ partial;     // reduce: [blocks][C][2]
  float* __restrict__ dy;
  float* __restrict__ dres;
  int relu;
  int cg;
  int nblocks;
  long long count;  // positions per channel of y
  int cgc, ppi;     // thread layout of the apply kernels (see Layout)
  long long npos;
  FastDiv dcgc, dWo, dHo, dDo;
  int Cv;           // valid channels: gamma / dgamma / dbeta hold Cv entries, channels [Cv, C) are zero padding (gamma = 0)
  // S3D-G self-gating behind this BatchNorm (unit windows only): the incoming gradient is that of the GATED output; the gradient
  // at the activation is dout*gate[n][c] + dmean[n][c]/P (rsp_gate_bwd_params), formed on the fly instead of by a kernel of its own
  const float* __restrict__ gate;   // nullable [N][C]
  const float* __restrict__ dmean;  // [N][C]
  float invP;
};

template <int VEC>
__device__ __forceinline__ void gated_grad(const BwdParams& p, int n, int c, float (&g)[VEC]) {
  float gt[VEC], dm[VEC];
  load_vec<VEC>(p.gate + (long long)n * p.d.C + c, gt);
  load_vec<VEC>(p.dmean + (long long)n * p.d.C + c, dm);
#pragma unroll
  for (int e = 0; e < VEC; ++e) g[e] = fmaf(g[e], gt[e], dm[e] * p.invP);
}

// gamma of VEC channels starting at c (1 when there is no affine weight, 0 for padding channels)
template <int VEC>
__device__ __forceinline__ void load_gamma(const BwdParams& p, int c, float (&gam)[VEC]) {
  if (p.gamma && p.Cv == p.d.C) {
    load_vec<VEC>(p.gamma + c, gam);
  } else {
#pragma unroll
    for (int e = 0; e < VEC; ++e) gam[e] = c + e < p.Cv ? (p.gamma ? p.gamma[c + e] : 1.f) : 0.f;
  }
}

// z of one input position (post affine + residual), VEC channels
template <int VEC>
__device__ __forceinline__ void zval(const BwdParams& p, long long pos, int c, const float (&sc)[VEC],
                                     const float (&sh)[VEC], float (&yv)[VEC], float (&z)[VEC]) {
  load_vec<VEC>(p.y + pos * p.d.in_ld + c, yv);
  float r[VEC];
  if (p.res) load_vec<VEC>(p.res + pos * p.d.res_ld + c, r);
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    z[e] = fmaf(yv[e], sc[e], sh[e]);
    if (p.res) z[e] += r[e];
  }
}

// pass 1: output-centric.  For each pooled output: arg-max (first max in scan order, like max_pool3d), dz = dout*mask.
template <int VEC, int MODE>      // MODE: see bn_act_pool_fwd_kernel
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const BwdParams p) {
  const rsp_pool3d_desc& d = p.d;
  __shared__ float red[256][2 * VEC];
  const int cg0 = blockIdx.y * 256;
  const int cgc = min(p.cg - cg0, 256);
  const int ppi = p.ppi;
  const int t = threadIdx.x;
  const int pl = fastdiv(t, p.dcgc);
  const int cgi = cg0 + (t - pl * p.cgc);
  const bool active = pl < ppi && (t - pl * p.cgc) < cgc;
  const int c = cgi * VEC;
  float s1[VEC], s2[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) s1[e] = s2[e] = 0.f;
  if (active) {
    float sc[VEC], sh[VEC], mean[VEC], invstd[VEC];
    load_vec<VEC>(p.ss + c, sc);
    load_vec<VEC>(p.ss + d.C + c, sh);
    load_vec<VEC>(p.mi + c, mean);
    load_vec<VEC>(p.mi + d.C + c, invstd);
    const long long npos = (long long)d.N * d.Do * d.Ho * d.Wo;
    long long op = (long long)blockIdx.x * ppi + pl;
    long long stride = (long long)gridDim.x * ppi;
    if constexpr (MODE == 1) {
      // no pooling, no gate: the position is its own window.  Four consecutive position groups per trip with all their loads issued
      // first (see bn_act_pool_fwd_kernel: 16 KB contiguous per block and trip); fixed accumulation order per thread.
      const long long step = ppi;
      stride *= 4;
      op = (long long)blockIdx.x * ppi * 4 + pl;
      for (; op + 3 * step < npos; op += stride) {
        float yv[4][VEC], r[4][VEC], g[4][VEC];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          load_vec<VEC>(p.y + (op + u * step) * d.in_ld + c, yv[u]);
          if (p.res) load_vec<VEC>(p.res + (op + u * step) * d.res_ld + c, r[u]);
          load_vec<VEC>(p.dout + (op + u * step) * d.out_ld + c, g[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            float z = fmaf(yv[u][e], sc[e], sh[e]);
            if (p.res) z += r[u][e];
            const float zz = p.relu ? fmaxf(z, 0.f) : z;
            const float dz = (p.relu && !(zz > 0.f)) ? 0.f : g[u][e];
            s1[e] += dz;
            s2[e] = fmaf(dz, (yv[u][e] - mean[e]) * invstd[e], s2[e]);
          }
      }
      stride = step;                          // the last, partial group of four: one position group at a time
    }
    long long op_end = MODE == 1 ? min(npos, op + 4 * (long long)ppi) : npos;
    if constexpr (MODE == 2 || MODE == 3) {
      // disjoint un-padded windows of 4 or 8 positions: all loads of an output first (see bn_act_pool_fwd_kernel), same scan order
      {
        constexpr int NW = MODE == 2 ? 4 : 8;
        long long off[NW];
        window_offsets<NW>(d, off);
        for (; op < npos; op += stride) {
          const int q1 = fastdiv((int)op, p.dWo), ow = (int)op - q1 * d.Wo;
          const int q2 = fastdiv(q1, p.dHo), oh = q1 - q2 * d.Ho;
          const int n = fastdiv(q2, p.dDo), od = q2 - n * d.Do;
          const long long base = (((long long)n * d.Di + od * d.sT) * d.Hi + oh * d.sH) * d.Wi + ow * d.sW;
          float yv[NW][VEC], r[NW][VEC], g[VEC], best[VEC], by[VEC];
#pragma unroll
          for (int w = 0; w < NW; ++w) {
            load_vec<VEC>(p.y + (base + off[w]) * d.in_ld + c, yv[w]);
            if (p.res) load_vec<VEC>(p.res + (base + off[w]) * d.res_ld + c, r[w]);
          }
          load_vec<VEC>(p.dout + op * d.out_ld + c, g);           // (a gated unit never pools: rsp_bn_act_pool_bwd)
#pragma unroll
          for (int e = 0; e < VEC; ++e) { best[e] = -INFINITY; by[e] = 0.f; }
#pragma unroll
          for (int w = 0; w < NW; ++w)
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
              float z = fmaf(yv[w][e], sc[e], sh[e]);
              if (p.res) z += r[w][e];
              const float zz = p.relu ? fmaxf(z, 0.f) : z;
              if (zz > best[e]) { best[e] = zz; by[e] = yv[w][e]; }
            }
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            const float dz = (p.relu && !(best[e] > 0.f)) ? 0.f : g[e];
            s1[e] += dz;
            s2[e] = fmaf(dz, (by[e] - mean[e]) * invstd[e], s2[e]);
          }
        }
      }
      op_end = 0;
    }
    for (; op < op_end; op += stride) {
      const int q1 = fastdiv((int)op, p.dWo), ow = (int)op - q1 * d.Wo;
      const int q2 = fastdiv(q1, p.dHo), oh = q1 - q2 * d.Ho;
      const int n = fastdiv(q2, p.dDo), od = q2 - n * d.Do;
      float best[VEC], by[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) { best[e] = -INFINITY; by[e] = 0.f; }
      for (int kt = 0; kt < d.kT; ++kt) {
        const int id = od * d.sT - d.pT + kt;
        if ((unsigned)id >= (unsigned)d.Di) continue;
        for (int kh = 0; kh < d.kH; ++kh) {
          const int ih = oh * d.sH - d.pH + kh;
          if ((unsigned)ih >= (unsigned)d.Hi) continue;
          for (int kw = 0; kw < d.kW; ++kw) {
            const int iw = ow * d.sW - d.pW + kw;
            if ((unsigned)iw >= (unsigned)d.Wi) continue;
            const long long pos = (((long long)n * d.Di + id) * d.Hi + ih) * d.Wi + iw;
            float yv[VEC], z[VEC];
            zval<VEC>(p, pos, c, sc, sh, yv, z);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
              const float zz = p.relu ? fmaxf(z[e], 0.f) : z[e];
              if (zz > best[e]) { best[e] = zz; by[e] = yv[e]; }
            }
          }
        }
      }
      float g[VEC];
      load_vec<VEC>(p.dout + op * d.out_ld + c, g);
      if (p.gate) gated_grad<VEC>(p, n, c, g);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float dz = (p.relu && !(best[e] > 0.f)) ? 0.f : g[e];
        s1[e] += dz;
        s2[e] = fmaf(dz, (by[e] - mean[e]) * invstd[e], s2[e]);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) { red[t][e] = s1[e]; red[t][VEC + e] = s2[e]; }
  __syncthreads();
  if (t < cgc) {
    float a1[VEC], a2[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) a1[e] = a2[e] = 0.f;
    for (int l = 0; l < ppi; ++l)
#pragma unroll
      for (int e = 0; e < VEC; ++e) { a1[e] += red[l * cgc + t][e]; a2[e] += red[l * cgc + t][VEC + e]; }
    float* o = p.partial + ((long long)blockIdx.x * d.C + (cg0 + t) * VEC) * 2;
#pragma unroll
    for (int e = 0; e < VEC; ++e) { o[2 * e] = a1[e]; o[2 * e + 1] = a2[e]; }
  }
}

// sums[c] = (sum dz, sum dz*xhat) in double; also dgamma / dbeta.  Block = FIN_CH channels x 64 partial-row lanes.
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nblocks, int C, int Cv,
                                                               double* __restrict__ sums, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta) {
  const int c = blockIdx.x * FIN_CH + (threadIdx.x & (FIN_CH - 1));
  double a, b;
  if (!fin_sum_partials(partial, nblocks, C, C, a, b)) return;
  sums[2 * c] = a;
  sums[2 * c + 1] = b;
  if (dbeta && c < Cv) dbeta[c] = (float)a;
  if (dgamma && c < Cv) dgamma[c] = (float)b;
}

// pass 2: input-centric.  dy = gamma*invstd*(dz_in - mean(dz) - xhat*mean(dz*xhat)); dz_in = dout*mask if this position
// is its window's arg-max else 0 (windows are disjoint: kernel == stride, no padding).
template <int VEC>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const BwdParams p) {
  const rsp_pool3d_desc& d = p.d;
  const long long total = (long long)d.N * d.Di * d.Hi * d.Wi * p.cg;
  const double invn = 1.0 / (double)p.count;
  for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += 256ll * gridDim.x) {
    const int cgi = (int)(idx % p.cg);
    long long q = idx / p.cg;
    const long long pos = q;
    const int iw = (int)(q % d.Wi); q /= d.Wi;
    const int ih = (int)(q % d.Hi); q /= d.Hi;
    const int id = (int)(q % d.Di);
    const int n = (int)(q / d.Di);
    const int c = cgi * VEC;
    float sc[VEC], sh[VEC], mean[VEC], invstd[VEC], gam[VEC];
    load_vec<VEC>(p.ss + c, sc);
    load_vec<VEC>(p.ss + d.C + c, sh);
    load_vec<VEC>(p.mi + c, mean);
    load_vec<VEC>(p.mi + d.C + c, invstd);
    load_gamma<VEC>(p, c, gam);
    float yv[VEC], z[VEC], dz[VEC];
    zval<VEC>(p, pos, c, sc, sh, yv, z);
#pragma unroll
    for (int e = 0; e < VEC; ++e) dz[e] = 0.f;
    const int od = id / d.sT, oh = ih / d.sH, ow = iw / d.sW;
    if (od < d.Do && oh < d.Ho && ow < d.Wo) {
      // is this position the first maximum of its window?
      bool win[VEC];
      float mine[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) { mine[e] = p.relu ? fmaxf(z[e], 0.f) : z[e]; win[e] = true; }
      const int my = ((id - od * d.sT) * d.kH + (ih - oh * d.sH)) * d.kW + (iw - ow * d.sW);
      if (d.kT * d.kH * d.kW > 1) {
        for (int kt = 0; kt < d.kT; ++kt)
          for (int kh = 0; kh < d.kH; ++kh)
            for (int kw = 0; kw < d.kW; ++kw) {
              const int o = (kt * d.kH + kh) * d.kW + kw;
              if (o == my) continue;
              const long long pp = (((long long)n * d.Di + od * d.sT + kt) * d.Hi + oh * d.sH + kh) * d.Wi + ow * d.sW + kw;
              float y2[VEC], z2[VEC];
              zval<VEC>(p, pp, c, sc, sh, y2, z2);
#pragma unroll
              for (int e = 0; e < VEC; ++e) {
                const float v = p.relu ? fmaxf(z2[e], 0.f) : z2[e];
                // earlier positions win ties; later ones must be strictly greater to displace us
                if (o < my ? (v >= mine[e]) : (v > mine[e])) win[e] = false;
              }
            }
      }
      const long long op = (((long long)n * d.Do + od) * d.Ho + oh) * d.Wo + ow;
      float g[VEC];
      load_vec<VEC>(p.dout + op * d.out_ld + c, g);
#pragma unroll
      for (int e = 0; e < VEC; ++e) dz[e] = (win[e] && !(p.relu && !(mine[e] > 0.f))) ? g[e] : 0.f;
    }
    float o[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float m1 = (float)(p.sums[2 * (c + e)] * invn);
      const float m2 = (float)(p.sums[2 * (c + e) + 1] * invn);
      const float xhat = (yv[e] - mean[e]) * invstd[e];
      o[e] = gam[e] * invstd[e] * (dz[e] - m1 - xhat * m2);
    }
    store_vec<VEC>(p.dy + pos * d.in_ld + c, o);
    if (p.dres) store_vec<VEC>(p.dres + pos * d.res_ld + c, dz);
  }
}

// pass 2, output-centric fast path: windows tile the input exactly (Di % sT == 0 ...) and hold <= 8 positions, so one
// thread owns a whole window: every y element is read once and every dy element written once (the input-centric
// kernel above re-reads the window for every element).
template <int VEC, int MODE>      // MODE: see bn_act_pool_fwd_kernel (0 here: unit windows with a gate, or 2-7 positions)
__global__ __launch_bounds__(256) void bn_bwd_apply_win_kernel(const BwdParams p) {
  const rsp_pool3d_desc& d = p.d;
  const int t = threadIdx.x;
  const int pl = fastdiv(t, p.dcgc);
  const int cgi = blockIdx.y * 256 + (t - pl * p.cgc);
  if (pl >= p.ppi || cgi >= p.cg) return;
  const int c = cgi * VEC;
  const int nwin = d.kT * d.kH * d.kW;
  // per-channel constants, once per thread: dy = k1 * dz + k2 * y + k3 with
  //   k1 = gamma*invstd, k2 = -k1*invstd*m2, k3 = -k1*(m1 - mean*invstd*m2)       (m1 = mean dz, m2 = mean dz*xhat)
  float sc[VEC], sh[VEC], mean[VEC], invstd[VEC], gam[VEC], m1[VEC], m2[VEC];
  load_vec<VEC>(p.ss + c, sc);
  load_vec<VEC>(p.ss + d.C + c, sh);
  load_vec<VEC>(p.mi + c, mean);
  load_vec<VEC>(p.mi + d.C + c, invstd);
  load_gamma<VEC>(p, c, gam);
  const double invn = 1.0 / (double)p.count;
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    m1[e] = (float)(p.sums[2 * (c + e)] * invn);
    m2[e] = (float)(p.sums[2 * (c + e) + 1] * invn);
  }
  long long o = (long long)blockIdx.x * p.ppi + pl;
  long long stride = (long long)gridDim.x * p.ppi;
  long long o_end = p.npos;
  if constexpr (MODE == 1) {
    // no pooling, no gate: four consecutive position groups per trip, loads first (see bn_act_pool_fwd_kernel)
    const long long step = p.ppi;
    stride *= 4;
    o = (long long)blockIdx.x * p.ppi * 4 + pl;
    for (; o + 3 * step < p.npos; o += stride) {
      float yv[4][VEC], r[4][VEC], g[4][VEC];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        load_vec<VEC>(p.y + (o + u * step) * d.in_ld + c, yv[u]);
        if (p.res) load_vec<VEC>(p.res + (o + u * step) * d.res_ld + c, r[u]);
        load_vec<VEC>(p.dout + (o + u * step) * d.out_ld + c, g[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float ov[VEC], dz[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          float z = fmaf(yv[u][e], sc[e], sh[e]);
          if (p.res) z += r[u][e];
          dz[e] = (p.relu && !(z > 0.f)) ? 0.f : g[u][e];
          const float xhat = (yv[u][e] - mean[e]) * invstd[e];
          ov[e] = gam[e] * invstd[e] * (dz[e] - m1[e] - xhat * m2[e]);
        }
        store_vec<VEC>(p.dy + (o + u * step) * d.in_ld + c, ov);
        if (p.dres) store_vec<VEC>(p.dres + (o + u * step) * d.res_ld + c, dz);
      }
    }
    stride = step;                            // the last, partial group of four: one position group at a time
    o_end = min(p.npos, o + 4 * step);
  }
  if constexpr (MODE == 2 || MODE == 3) {
    // whole windows of 4 or 8 positions (the launcher guarantees exact tiling): loads first, then arg-max, then the stores
    {
      constexpr int NW = MODE == 2 ? 4 : 8;
      long long off[NW];
      window_offsets<NW>(d, off);
      for (; o < o_end; o += stride) {
        const int op = (int)o;
        const int q1 = fastdiv(op, p.dWo), ow = op - q1 * d.Wo;
        const int q2 = fastdiv(q1, p.dHo), oh = q1 - q2 * d.Ho;
        const int n = fastdiv(q2, p.dDo), od = q2 - n * d.Do;
        const long long base = (((long long)n * d.Di + od * d.sT) * d.Hi + oh * d.sH) * d.Wi + ow * d.sW;
        float yv[NW][VEC], r[NW][VEC], g[VEC], best[VEC];
        int bi[VEC];
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          load_vec<VEC>(p.y + (base + off[w]) * d.in_ld + c, yv[w]);
          if (p.res) load_vec<VEC>(p.res + (base + off[w]) * d.res_ld + c, r[w]);
        }
        load_vec<VEC>(p.dout + o * d.out_ld + c, g);
#pragma unroll
        for (int e = 0; e < VEC; ++e) { best[e] = -INFINITY; bi[e] = 0; }
#pragma unroll
        for (int w = 0; w < NW; ++w)
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            float z = fmaf(yv[w][e], sc[e], sh[e]);
            if (p.res) z += r[w][e];
            const float v = p.relu ? fmaxf(z, 0.f) : z;
            if (v > best[e]) { best[e] = v; bi[e] = w; }        // first maximum in scan order
          }
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          float ov[VEC], dz[VEC];
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            dz[e] = (bi[e] == w && !(p.relu && !(best[e] > 0.f))) ? g[e] : 0.f;
            const float xhat = (yv[w][e] - mean[e]) * invstd[e];
            ov[e] = gam[e] * invstd[e] * (dz[e] - m1[e] - xhat * m2[e]);
          }
          store_vec<VEC>(p.dy + (base + off[w]) * d.in_ld + c, ov);
          if (p.dres) store_vec<VEC>(p.dres + (base + off[w]) * d.res_ld + c, dz);
        }
      }
    }
    return;
  }
  for (; o < o_end; o += stride) {
    float g[VEC];
    load_vec<VEC>(p.dout + o * d.out_ld + c, g);
    if (MODE == 1 || nwin == 1) {
      // no pooling: the position is its own window
      if (p.gate) gated_grad<VEC>(p, fastdiv(fastdiv(fastdiv((int)o, p.dWo), p.dHo), p.dDo), c, g);
      float yv[VEC], z[VEC], ov[VEC], dz[VEC];
      zval<VEC>(p, o, c, sc, sh, yv, z);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        dz[e] = (p.relu && !(z[e] > 0.f)) ? 0.f : g[e];
        const float xhat = (yv[e] - mean[e]) * invstd[e];
        ov[e] = gam[e] * invstd[e] * (dz[e] - m1[e] - xhat * m2[e]);
      }
      store_vec<VEC>(p.dy + o * d.in_ld + c, ov);
      if (p.dres) store_vec<VEC>(p.dres + o * d.res_ld + c, dz);
      continue;
    }
    const int op = (int)o;
    const int q1 = fastdiv(op, p.dWo), ow = op - q1 * d.Wo;
    const int q2 = fastdiv(q1, p.dHo), oh = q1 - q2 * d.Ho;
    const int n = fastdiv(q2, p.dDo), od = q2 - n * d.Do;
    const long long base = (((long long)n * d.Di + od * d.sT) * d.Hi + oh * d.sH) * d.Wi + ow * d.sW;
    float yv[8][VEC], zv[8][VEC], best[VEC];
    int bi[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { best[e] = -INFINITY; bi[e] = 0; }
    // window offsets in scan order (kt, kh, kw), advanced incrementally
    int kw = 0, kh = 0, kt = 0;
#pragma unroll
    for (int wdx = 0; wdx < 8; ++wdx) {
      if (wdx < nwin) {
        const long long pos = base + ((long long)kt * d.Hi + kh) * d.Wi + kw;
        zval<VEC>(p, pos, c, sc, sh, yv[wdx], zv[wdx]);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const float v = p.relu ? fmaxf(zv[wdx][e], 0.f) : zv[wdx][e];
          if (v > best[e]) { best[e] = v; bi[e] = wdx; }     // first maximum in scan order
        }
        if (++kw == d.kW) { kw = 0; if (++kh == d.kH) { kh = 0; ++kt; } }
      }
    }
    kw = 0; kh = 0; kt = 0;
#pragma unroll
    for (int wdx = 0; wdx < 8; ++wdx) {
      if (wdx < nwin) {
        const long long pos = base + ((long long)kt * d.Hi + kh) * d.Wi + kw;
        float ov[VEC], dz[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          dz[e] = (bi[e] == wdx && !(p.relu && !(best[e] > 0.f))) ? g[e] : 0.f;
          const float xhat = (yv[wdx][e] - mean[e]) * invstd[e];
          ov[e] = gam[e] * invstd[e] * (dz[e] - m1[e] - xhat * m2[e]);
        }
        store_vec<VEC>(p.dy + pos * d.in_ld + c, ov);
        if (p.dres) store_vec<VEC>(p.dres + pos * d.res_ld + c, dz);
        if (++kw == d.kW) { kw = 0; if (++kh == d.kH) { kh = 0; ++kt; } }
      }
    }
  }
}

bool pool_ok(const rsp_pool3d_desc* d, bool need_disjoint) {
  if (!d) return false;
  if (d->N <= 0 || d->C <= 0 || d->kT <= 0 || d->kH <= 0 || d->kW <= 0) return false;
  if (d->sT <= 0 || d->sH <= 0 || d->sW <= 0 || d->pT < 0 || d->pH < 0 || d->pW < 0) return false;
  if (d->Do != (d->Di + 2 * d->pT - d->kT) / d->sT + 1) return false;
  if (d->Ho != (d->Hi + 2 * d->pH - d->kH) / d->sH + 1) return false;
  if (d->Wo != (d->Wi + 2 * d->pW - d->kW) / d->sW + 1) return false;
  if (d->in_ld < d->C || d->out_ld < d->C) return false;
  if (need_disjoint) {
    if (d->kT != d->sT || d->kH != d->sH || d->kW != d->sW || d->pT || d->pH || d->pW) return false;
  }
  return true;
}

// body of the three pooling-aware kernels for this window shape (their MODE template argument)
int bn_mode(const rsp_pool3d_desc* d, bool gated) {
  const int nwin = d->kT * d->kH * d->kW;
  const bool unit = nwin == 1 && d->sT == 1 && d->sH == 1 && d->sW == 1 && !(d->pT | d->pH | d->pW);
  if (unit) return gated ? 0 : 1;
  const bool tile = d->kT == d->sT && d->kH == d->sH && d->kW == d->sW && !(d->pT | d->pH | d->pW);
  return tile && nwin == 4 ? 2 : (tile && nwin == 8 ? 3 : 0);
}

#define RSP_BN_LAUNCH_V(K, V, mode, grid, stream, p)                                                   \
  do {                                                                                                 \
    switch (mode) {                                                                                    \
      case 1: hipLaunchKernelGGL((K<V, 1>), grid, dim3(256), 0, stream, p); break;                     \
      case 2: hipLaunchKernelGGL((K<V, 2>), grid, dim3(256), 0, stream, p); break;                     \
      case 3: hipLaunchKernelGGL((K<V, 3>), grid, dim3(256), 0, stream, p); break;                     \
      default: hipLaunchKernelGGL((K<V, 0>), grid, dim3(256), 0, stream, p); break;                    \
    }                                                                                                  \
  } while (0)
#define RSP_BN_LAUNCH(K, vec, mode, grid, stream, p)                                                   \
  do {                                                                                                 \
    if (vec) RSP_BN_LAUNCH_V(K, 4, mode, grid, stream, p);                                             \
    else RSP_BN_LAUNCH_V(K, 1, mode, grid, stream, p);                                                 \
  } while (0)

int grid_for(long long total) {
  long long b = (total + 255) / 256;
  return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

int reduce_blocks(const rsp_pool3d_desc* d, int cg) {
  const int cgc = cg < 256 ? cg : 256;
  const int ppi = 256 / cgc;
  const long long npos = (long long)d->N * d->Do * d->Ho * d->Wo;
  long long b = (npos + (long long)ppi * 8 - 1) / ((long long)ppi * 8);  // >= 8 positions per thread
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" {

int32_t rsp_bn_stat_tiles(int64_t rows) { return rsp_cdiv(rows, 128); }

int rsp_bn_stats(const float* y, int64_t rows, int32_t C, int32_t ld, float* stat_partials, void* stream) {
  RSP_REQUIRE(y && stat_partials && rows > 0 && C > 0 && ld >= C, "rsp_bn_stats: bad argument");
  dim3 grid(rsp_cdiv(rows, 128), rsp_cdiv(C, 64));
  hipLaunchKernelGGL(bn_stats_kernel, grid, dim3(256), 0, (hipStream_t)stream, y, (long long)rows, C, ld, stat_partials);
  return rsp_check_launch("bn_stats_kernel");
}

// slices of the two-stage reduction (tiles > 2048): enough (channel-block, slice) workgroups to cover the machine — C3D conv1 has
// 50 176 tiles of ONE 64-channel block, and 64 slices left three quarters of the CUs idle (91 us per finalize)
static int finalize_slices(int tiles) { return tiles >= 8192 ? 256 : (tiles >= 4096 ? 64 : (tiles >= 64 ? 16 : 1)); }

size_t rsp_bn_finalize_workspace(int32_t tiles, int32_t C) {
  const int S = finalize_slices(tiles);
  return (size_t)S * C * 2 * sizeof(double);
}

int rsp_bn_finalize(const float* stat_partials, int32_t tiles, int32_t C, int32_t stat_ld, int64_t count, const float* conv_bias,
                    const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                    float* running_var, float* mean_invstd, float* scale_shift, void* workspace,
                    size_t workspace_bytes, void* stream) {
  return rsp_bn_finalize_v(stat_partials, tiles, C, C, stat_ld, count, conv_bias, gamma, beta, eps, momentum, running_mean, running_var,
                           mean_invstd, scale_shift, workspace, workspace_bytes, stream);
}

int rsp_bn_finalize_v(const float* stat_partials, int32_t tiles, int32_t C, int32_t c_valid, int32_t stat_ld, int64_t count,
                      const float* conv_bias, const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                      float* running_var, float* mean_invstd, float* scale_shift, void* workspace,
                      size_t workspace_bytes, void* stream) {
  return rsp_bn_finalize_x(stat_partials, tiles, C, c_valid, stat_ld, count, conv_bias, gamma, beta, eps, momentum, running_mean,
                           running_var, nullptr, mean_invstd, scale_shift, workspace, workspace_bytes, stream);
}

// EMA of the running statistics for a list of BatchNorm layers in one launch (jobs in device memory): what rsp_bn_finalize does
// per layer when it is not told to defer (batch_stats_out).  grid = (channel blocks of the widest layer, jobs).
__global__ __launch_bounds__(256) void bn_ema_kernel(const rsp_bn_ema_job* __restrict__ jobs) {
  const rsp_bn_ema_job j = jobs[blockIdx.y];
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= j.C) return;
  j.running_mean[c] = (1.f - j.momentum) * j.running_mean[c] + j.momentum * j.batch_stats[c];
  j.running_var[c] = (1.f - j.momentum) * j.running_var[c] + j.momentum * j.batch_stats[j.C + c];
}

int rsp_bn_running_update(const rsp_bn_ema_job* jobs_device, int32_t n_jobs, int32_t max_c, void* stream) {
  RSP_REQUIRE(jobs_device && n_jobs > 0 && n_jobs <= 65535 && max_c > 0, "rsp_bn_running_update: bad argument");
  hipLaunchKernelGGL(bn_ema_kernel, dim3(rsp_cdiv(max_c, 256), n_jobs), dim3(256), 0, (hipStream_t)stream, jobs_device);
  return rsp_check_launch("bn_ema_kernel");
}

int rsp_bn_finalize_x(const float* stat_partials, int32_t tiles, int32_t C, int32_t c_valid, int32_t stat_ld, int64_t count,
                      const float* conv_bias, const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                      float* running_var, float* batch_stats_out, float* mean_invstd, float* scale_shift, void* workspace,
                      size_t workspace_bytes, void* stream) {
  RSP_REQUIRE(stat_partials && mean_invstd && scale_shift && workspace, "rsp_bn_finalize: null pointer");
  RSP_REQUIRE(tiles > 0 && C > 0 && count > 0 && stat_ld >= C && c_valid > 0 && c_valid <= C, "rsp_bn_finalize: bad size");
  const int S = finalize_slices(tiles);
  if (workspace_bytes < (size_t)S * C * 2 * sizeof(double)) {
    rsp_set_error("rsp_bn_finalize: workspace too small");
    return RSP_EWORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  if (tiles <= 2048) {
    hipLaunchKernelGGL(bn_finalize_one_kernel, dim3(rsp_cdiv(C, FIN_CH)), dim3(1024), 0, s, stat_partials, tiles, C, c_valid, stat_ld, (long long)count,
                       conv_bias, gamma, beta, eps, momentum, running_mean, running_var, mean_invstd, scale_shift, batch_stats_out);
    return rsp_check_launch("bn_finalize_one_kernel");
  }
  double* part = reinterpret_cast<double*>(workspace);
  hipLaunchKernelGGL(bn_finalize_stage1, dim3(rsp_cdiv(C, 64), S), dim3(256), 0, s, stat_partials, tiles, C, stat_ld, S, part);
  int rc = rsp_check_launch("bn_finalize_stage1");
  if (rc != RSP_OK) return rc;
  hipLaunchKernelGGL(bn_finalize_stage2, dim3(rsp_cdiv(C, 64)), dim3(1024), 0, s, part, S, C, c_valid, (long long)count, conv_bias,
                     gamma, beta, eps, momentum, running_mean, running_var, mean_invstd, scale_shift, batch_stats_out);
  return rsp_check_launch("bn_finalize_stage2");
}

int rsp_bn_act_pool_fwd(const rsp_pool3d_desc* d, const float* y, const float* scale_shift, const float* residual,
                        int relu, float* out, void* stream) {
  return rsp_bn_act_pool_gate_fwd(d, y, scale_shift, residual, relu, nullptr, out, stream);
}

int rsp_bn_act_pool_gate_fwd(const rsp_pool3d_desc* d, const float* y, const float* scale_shift, const float* residual,
                             int relu, const float* gate, float* out, void* stream) {
  RSP_REQUIRE(pool_ok(d, false), "rsp_bn_act_pool_fwd: bad descriptor");
  RSP_REQUIRE(y && scale_shift && out, "rsp_bn_act_pool_fwd: null pointer");
  // overlapping 3x3x3 / 1x3x3 windows without residual (the ResNet stems; S3D-G's gated front-end units): the pooling body with the apply folded
  // into its loads — all 27 loads of an output issued before the first compare — instead of the generic window loop below
  if (!residual && (d->kT != d->sT || d->kH != d->sH || d->kW != d->sW || d->pT || d->pH || d->pW) &&
      rsp_bn_act_maxpool_applicable(d) && rsp_aligned16(y) && rsp_aligned16(out) && rsp_aligned16(scale_shift) &&
      (!gate || rsp_aligned16(gate)))
    return rsp_bn_act_maxpool_gate_fwd(d, y, scale_shift, relu, gate, out, nullptr, stream);
  PoolParams p;
  p.d = *d; p.y = y; p.ss = scale_shift; p.res = residual; p.gate = gate; p.out = out; p.relu = relu;
  const bool vec = d->C % 4 == 0 && d->in_ld % 4 == 0 && d->out_ld % 4 == 0 && rsp_aligned16(y) && rsp_aligned16(out) &&
                   rsp_aligned16(scale_shift) && (!residual || (d->res_ld % 4 == 0 && rsp_aligned16(residual))) &&
                   (!gate || rsp_aligned16(gate));
  p.cg = vec ? d->C / 4 : d->C;
  const Layout L = make_layout(d, p.cg, 4);
  RSP_REQUIRE(L.npos < (1ll << 31), "rsp_bn_act_pool_fwd: more than 2^31 - 1 output positions");
  p.cgc = L.cgc; p.ppi = L.ppi; p.npos = L.npos; p.dcgc = L.dcgc; p.dWo = L.dWo; p.dHo = L.dHo; p.dDo = L.dDo;
  RSP_BN_LAUNCH(bn_act_pool_fwd_kernel, vec, bn_mode(d, gate != nullptr), L.grid, (hipStream_t)stream, p);
  return rsp_check_launch("bn_act_pool_fwd_kernel");
}

size_t rsp_bn_bwd_workspace(const rsp_pool3d_desc* d) {
  if (!pool_ok(d, true)) return 0;
  return rsp_align_up((size_t)2048 * d->C * 2 * sizeof(float), 256) + (size_t)d->C * 2 * sizeof(double);
}

int rsp_bn_act_pool_bwd(const rsp_pool3d_desc* d, const float* y, const float* residual, const float* dout,
                        const float* gamma, const float* mean_invstd, const float* scale_shift, int relu, float* dy,
                        float* dres, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                        void* stream) {
  return rsp_bn_act_pool_bwd_v(d, y, residual, dout, gamma, mean_invstd, scale_shift, relu, dy, dres, dgamma, dbeta, d ? d->C : 0,
                               workspace, workspace_bytes, stream);
}

int rsp_bn_act_pool_bwd_v(const rsp_pool3d_desc* d, const float* y, const float* residual, const float* dout,
                          const float* gamma, const float* mean_invstd, const float* scale_shift, int relu, float* dy,
                          float* dres, float* dgamma, float* dbeta, int32_t c_valid, void* workspace, size_t workspace_bytes,
                          void* stream) {
  return rsp_bn_act_pool_bwd_g(d, y, residual, dout, gamma, mean_invstd, scale_shift, relu, dy, dres, dgamma, dbeta, c_valid, nullptr,
                               nullptr, workspace, workspace_bytes, stream);
}

int rsp_bn_act_pool_bwd_g(const rsp_pool3d_desc* d, const float* y, const float* residual, const float* dout,
                          const float* gamma, const float* mean_invstd, const float* scale_shift, int relu, float* dy,
                          float* dres, float* dgamma, float* dbeta, int32_t c_valid, const float* gate, const float* dmean,
                          void* workspace, size_t workspace_bytes, void* stream) {
  RSP_REQUIRE(pool_ok(d, true), "rsp_bn_act_pool_bwd: needs disjoint windows (kernel == stride, no padding)");
  RSP_REQUIRE(!gate || (dmean && d->kT * d->kH * d->kW == 1 && d->sT * d->sH * d->sW == 1 && !residual),
              "rsp_bn_act_pool_bwd: a gated unit has a unit window, no residual, and needs dmean");
  RSP_REQUIRE(c_valid > 0 && c_valid <= d->C, "rsp_bn_act_pool_bwd: bad valid channel count");
  RSP_REQUIRE(y && dout && mean_invstd && scale_shift && dy && workspace, "rsp_bn_act_pool_bwd: null pointer");
  if (workspace_bytes < rsp_bn_bwd_workspace(d)) {
    rsp_set_error("rsp_bn_act_pool_bwd: workspace too small");
    return RSP_EWORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  BwdParams p;
  memset(&p, 0, sizeof p);
  p.d = *d; p.y = y; p.res = residual; p.dout = dout; p.gamma = gamma; p.mi = mean_invstd; p.ss = scale_shift;
  p.dy = dy; p.dres = dres; p.relu = relu;
  p.Cv = c_valid;
  p.gate = gate; p.dmean = dmean;
  p.invP = 1.f / (float)((long long)d->Di * d->Hi * d->Wi);
  p.count = (long long)d->N * d->Di * d->Hi * d->Wi;
  const bool vec = d->C % 4 == 0 && d->in_ld % 4 == 0 && d->out_ld % 4 == 0 && rsp_aligned16(y) && rsp_aligned16(dout) &&
                   rsp_aligned16(dy) && rsp_aligned16(scale_shift) && rsp_aligned16(mean_invstd) &&
                   (!gamma || c_valid != d->C || rsp_aligned16(gamma)) &&
                   (!residual || (d->res_ld % 4 == 0 && rsp_aligned16(residual) && (!dres || rsp_aligned16(dres)))) &&
                   (!gate || (rsp_aligned16(gate) && rsp_aligned16(dmean)));
  p.cg = vec ? d->C / 4 : d->C;
  p.partial = reinterpret_cast<float*>(workspace);
  double* sums = reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(workspace) +
                                           rsp_align_up((size_t)2048 * d->C * 2 * sizeof(float), 256));
  p.sums = sums;
  const Layout L = make_layout(d, p.cg, 4);
  RSP_REQUIRE(L.npos < (1ll << 31), "rsp_bn_act_pool_bwd: more than 2^31 - 1 output positions");
  p.cgc = L.cgc; p.ppi = L.ppi; p.npos = L.npos; p.dcgc = L.dcgc; p.dWo = L.dWo; p.dHo = L.dHo; p.dDo = L.dDo;
  p.nblocks = reduce_blocks(d, p.cg);
  dim3 rgrid(p.nblocks, rsp_cdiv(p.cg, 256));
  const int mode = bn_mode(d, gate != nullptr);
  RSP_BN_LAUNCH(bn_bwd_reduce_kernel, vec, mode, rgrid, s, p);
  int rc = rsp_check_launch("bn_bwd_reduce_kernel");
  if (rc != RSP_OK) return rc;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(rsp_cdiv(d->C, FIN_CH)), dim3(1024), 0, s, p.partial, p.nblocks, d->C, c_valid, sums,
                     dgamma, dbeta);
  rc = rsp_check_launch("bn_bwd_finalize_kernel");
  if (rc != RSP_OK) return rc;
  const bool exact = d->Di % d->sT == 0 && d->Hi % d->sH == 0 && d->Wi % d->sW == 0 && d->kT * d->kH * d->kW <= 8;
  if (exact) {
    RSP_BN_LAUNCH(bn_bwd_apply_win_kernel, vec, mode, L.grid, s, p);
    return rsp_check_launch("bn_bwd_apply_win_kernel");
  }
  const long long total = (long long)d->N * d->Di * d->Hi * d->Wi * p.cg;
  if (vec) hipLaunchKernelGGL(bn_bwd_apply_kernel<4>, dim3(grid_for(total)), dim3(256), 0, s, p);
  else hipLaunchKernelGGL(bn_bwd_apply_kernel<1>, dim3(grid_for(total)), dim3(256), 0, s, p);
  return rsp_check_launch("bn_bwd_apply_kernel");
}

}  // extern "C"
