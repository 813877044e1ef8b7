// Stand-alone MaxPool3d with arbitrary (overlapping / padded) windows, and S3D-G self-gating.
// Reference call sites: models/resnet.py:139 (MaxPool3d(3, stride 2, pad 1)), models/s3dg.py:90 (3,1,1), :107-119
// ((1,3,3)/(1,2,2), (3,3,3)/2, (2,2,2)/2); gating: models/s3dg.py:63-72  x * sigmoid(Conv1x1x1(mean_{T,H,W}(x))).
// HBM-bound streaming kernels; backward is input-centric over saved arg-max indices (deterministic, no atomics).
#include "common.h"

namespace {

struct MPParams {
  rsp_pool3d_desc d;
  const float* __restrict__ x;
  float* __restrict__ out;
  int* __restrict__ idx;         // arg-max as linear input position (d*Hi+h)*Wi+w within the sample
  const float* __restrict__ dout;
  float* __restrict__ dx;
};

__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const MPParams p) {
  const rsp_pool3d_desc& d = p.d;
  const long long total = (long long)d.N * d.Do * d.Ho * d.Wo * d.C;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += 256ll * gridDim.x) {
    const int c = (int)(i % d.C);
    long long q = i / d.C;
    const int ow = (int)(q % d.Wo); q /= d.Wo;
    const int oh = (int)(q % d.Ho); q /= d.Ho;
    const int od = (int)(q % d.Do);
    const int n = (int)(q / d.Do);
    float best = -INFINITY;
    int bi = -1;
    for (int kt = 0; kt < d.kT; ++kt) {
      const int id = od * d.sT - d.pT + kt;
      if ((unsigned)id >= (unsigned)d.Di) continue;
      for (int kh = 0; kh < d.kH; ++kh) {
        const int ih = oh * d.sH - d.pH + kh;
        if ((unsigned)ih >= (unsigned)d.Hi) continue;
        for (int kw = 0; kw < d.kW; ++kw) {
          const int iw = ow * d.sW - d.pW + kw;
          if ((unsigned)iw >= (unsigned)d.Wi) continue;
          const int lin = (id * d.Hi + ih) * d.Wi + iw;
          const float v = p.x[((long long)n * d.Di * d.Hi * d.Wi + lin) * d.in_ld + c];
          if (v > best || bi < 0) { best = v; bi = lin; }   // first maximum in scan order, like max_pool3d
        }
      }
    }
    const long long o = (((long long)n * d.Do + od) * d.Ho + oh) * d.Wo + ow;
    p.out[o * d.out_ld + c] = best;
    if (p.idx) p.idx[o * d.C + c] = bi;
  }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const MPParams p) {
  const rsp_pool3d_desc& d = p.d;
  const long long total = (long long)d.N * d.Di * d.Hi * d.Wi * d.C;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += 256ll * gridDim.x) {
    const int c = (int)(i % d.C);
    long long q = i / d.C;
    const int iw = (int)(q % d.Wi); q /= d.Wi;
    const int ih = (int)(q % d.Hi); q /= d.Hi;
    const int id = (int)(q % d.Di);
    const int n = (int)(q / d.Di);
    const int lin = (id * d.Hi + ih) * d.Wi + iw;
    float g = 0.f;
    // output positions whose window contains this input position: o*s - p <= i <= o*s - p + k - 1
    const int od0 = max(0, (id + d.pT - d.kT + d.sT) / d.sT), od1 = min(d.Do - 1, (id + d.pT) / d.sT);
    const int oh0 = max(0, (ih + d.pH - d.kH + d.sH) / d.sH), oh1 = min(d.Ho - 1, (ih + d.pH) / d.sH);
    const int ow0 = max(0, (iw + d.pW - d.kW + d.sW) / d.sW), ow1 = min(d.Wo - 1, (iw + d.pW) / d.sW);
    for (int od = od0; od <= od1; ++od)
      for (int oh = oh0; oh <= oh1; ++oh)
        for (int ow = ow0; ow <= ow1; ++ow) {
          const long long o = (((long long)n * d.Do + od) * d.Ho + oh) * d.Wo + ow;
          if (p.idx[o * d.C + c] == lin) g += p.dout[o * d.out_ld + c];
        }
    p.dx[((long long)n * d.Di * d.Hi * d.Wi + lin) * d.in_ld + c] = g;
  }
}

// float4 variants (C % 4 == 0, pitches % 4 == 0, 16-byte aligned bases): one thread owns 4 channels of one position, so the
// position decode and the window bounds are computed once per 16 bytes and every access is a 16-byte transaction.
typedef int intx4 __attribute__((ext_vector_type(4)));

// KT/KH/KW > 0: window dims known at compile time — the tap loops unroll into straight-line code whose loads are all issued
// before the first compare (an out-of-range tap reads a clamped, valid address and is masked to -inf), instead of one
// load-compare round trip per tap behind three run-time loops with `continue`s.  KEEP = false (key-encoder passes): value only.
// BN: the input is a convolution output and every loaded value first goes through the BatchNorm apply a = act(x * scale + shift)
// (bn_maxpool_fwd_vec_kernel: the ResNet stems' bn1 -> relu -> MaxPool3d(3, 2, 1) in one pass, engine._pool_fusion).
template <bool KEEP, int KT, int KH, int KW, bool BN>
__device__ __forceinline__ void maxpool_fwd_vec_body(const MPParams& p, const float* __restrict__ ss, const int relu,
                                                     const float* __restrict__ gate = nullptr) {
  const rsp_pool3d_desc& d = p.d;
  const int C4 = d.C >> 2;
  const int kT = KT > 0 ? KT : d.kT, kH = KH > 0 ? KH : d.kH, kW = KW > 0 ? KW : d.kW;
  const long long total = (long long)d.N * d.Do * d.Ho * d.Wo * C4;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += 256ll * gridDim.x) {
    const int c = (int)(i % C4) * 4;
    int q = (int)(i / C4);
    const int ow = q % d.Wo; q /= d.Wo;
    const int oh = q % d.Ho; q /= d.Ho;
    const int od = q % d.Do;
    const int n = q / d.Do;
    floatx4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    intx4 bi = {-1, -1, -1, -1};
    const float* xn = p.x + (long long)n * d.Di * d.Hi * d.Wi * d.in_ld + c;
    if (KT > 0) {
      floatx4 v[KT * KH * KW > 0 ? KT * KH * KW : 1];
      int lin[KT * KH * KW > 0 ? KT * KH * KW : 1];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int kh = 0; kh < KH; ++kh)
#pragma unroll
          for (int kw = 0; kw < KW; ++kw) {
            const int id = od * d.sT - d.pT + kt, ih = oh * d.sH - d.pH + kh, iw = ow * d.sW - d.pW + kw;
            const bool ok = (unsigned)id < (unsigned)d.Di && (unsigned)ih < (unsigned)d.Hi && (unsigned)iw < (unsigned)d.Wi;
            const int l = (min(max(id, 0), d.Di - 1) * d.Hi + min(max(ih, 0), d.Hi - 1)) * d.Wi + min(max(iw, 0), d.Wi - 1);
            const int t = (kt * KH + kh) * KW + kw;
            lin[t] = ok ? l : -1;
            v[t] = *reinterpret_cast<const floatx4*>(xn + (long long)l * d.in_ld);
          }
      if (BN) {
        const floatx4 sc = *reinterpret_cast<const floatx4*>(ss + c), sh = *reinterpret_cast<const floatx4*>(ss + d.C + c);
#pragma unroll
        for (int t = 0; t < KT * KH * KW; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float z = fmaf(v[t][e], sc[e], sh[e]);
            v[t][e] = relu ? fmaxf(z, 0.f) : z;
          }
        if (gate) {      // S3D-G's self-gating between the activation and the pool (models/s3dg.py:105-108): per (sample, channel)
          const floatx4 gt = *reinterpret_cast<const floatx4*>(gate + (long long)n * d.C + c);
#pragma unroll
          for (int t = 0; t < KT * KH * KW; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[t][e] *= gt[e];
        }
      }
#pragma unroll
      for (int t = 0; t < KT * KH * KW; ++t) {
        if (KEEP) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (lin[t] >= 0 && (v[t][e] > best[e] || bi[e] < 0)) { best[e] = v[t][e]; bi[e] = lin[t]; }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) best[e] = lin[t] >= 0 ? fmaxf(best[e], v[t][e]) : best[e];
        }
      }
    } else {
      for (int kt = 0; kt < kT; ++kt) {
        const int id = od * d.sT - d.pT + kt;
        if ((unsigned)id >= (unsigned)d.Di) continue;
        for (int kh = 0; kh < kH; ++kh) {
          const int ih = oh * d.sH - d.pH + kh;
          if ((unsigned)ih >= (unsigned)d.Hi) continue;
          for (int kw = 0; kw < kW; ++kw) {
            const int iw = ow * d.sW - d.pW + kw;
            if ((unsigned)iw >= (unsigned)d.Wi) continue;
            const int lin = (id * d.Hi + ih) * d.Wi + iw;
            const floatx4 v = *reinterpret_cast<const floatx4*>(xn + (long long)lin * d.in_ld);
            if (KEEP) {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (v[e] > best[e] || bi[e] < 0) { best[e] = v[e]; bi[e] = lin; }   // first maximum in scan order
            } else {     // value only: same result as the tracked scan for finite inputs
#pragma unroll
              for (int e = 0; e < 4; ++e) best[e] = fmaxf(best[e], v[e]);
            }
          }
        }
      }
    }
    const long long o = (((long long)n * d.Do + od) * d.Ho + oh) * d.Wo + ow;
    *reinterpret_cast<floatx4*>(p.out + o * d.out_ld + c) = best;
    if (KEEP) *reinterpret_cast<intx4*>(p.idx + o * d.C + c) = bi;
  }
}
template <bool KEEP, int KT, int KH, int KW>
__global__ __launch_bounds__(256) void maxpool_fwd_vec_kernel(const MPParams p) {
  maxpool_fwd_vec_body<KEEP, KT, KH, KW, false>(p, nullptr, 0);
}
template <bool KEEP, int KT, int KH, int KW>
__global__ __launch_bounds__(256) void bn_maxpool_fwd_vec_kernel(const MPParams p, const float* __restrict__ ss, const int relu,
                                                                 const float* __restrict__ gate) {
  static_assert(KT > 0, "compile-time windows only");
  maxpool_fwd_vec_body<KEEP, KT, KH, KW, true>(p, ss, relu, gate);
}

__global__ __launch_bounds__(256) void maxpool_bwd_vec_kernel(const MPParams p) {
  const rsp_pool3d_desc& d = p.d;
  const int C4 = d.C >> 2;
  const long long total = (long long)d.N * d.Di * d.Hi * d.Wi * C4;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += 256ll * gridDim.x) {
    const int c = (int)(i % C4) * 4;
    int q = (int)(i / C4);
    const int iw = q % d.Wi; q /= d.Wi;
    const int ih = q % d.Hi; q /= d.Hi;
    const int id = q % d.Di;
    const int n = q / d.Di;
    const int lin = (id * d.Hi + ih) * d.Wi + iw;
    floatx4 g = {0.f, 0.f, 0.f, 0.f};
    const int od0 = max(0, (id + d.pT - d.kT + d.sT) / d.sT), od1 = min(d.Do - 1, (id + d.pT) / d.sT);
    const int oh0 = max(0, (ih + d.pH - d.kH + d.sH) / d.sH), oh1 = min(d.Ho - 1, (ih + d.pH) / d.sH);
    const int ow0 = max(0, (iw + d.pW - d.kW + d.sW) / d.sW), ow1 = min(d.Wo - 1, (iw + d.pW) / d.sW);
    for (int od = od0; od <= od1; ++od)
      for (int oh = oh0; oh <= oh1; ++oh)
        for (int ow = ow0; ow <= ow1; ++ow) {
          const long long o = (((long long)n * d.Do + od) * d.Ho + oh) * d.Wo + ow;
          const intx4 a = *reinterpret_cast<const intx4*>(p.idx + o * d.C + c);
          const floatx4 v = *reinterpret_cast<const floatx4*>(p.dout + o * d.out_ld + c);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (a[e] == lin) g[e] += v[e];
        }
    *reinterpret_cast<floatx4*>(p.dx + ((long long)n * d.Di * d.Hi * d.Wi + lin) * d.in_ld + c) = g;
  }
}

// ---- 3x3x3 / stride 1 / pad 1 ("same") pooling: S3D-G's inception branch 3 (models/s3dg.py:90) -----------------------------
// The generic kernels above load all 27 taps per output (forward) and 27 arg-max + 27 gradient quads per input (backward): at
// 1.0-1.5 TB/s algorithmic they are bound by L1 traffic, not HBM.  Here a thread owns a run of `seg` positions along W of one
// (sample, t, h, channel quad) row and slides along it: per step it loads ONE new column — the 9 (kt, kh) rows at w + 1 — and
// keeps the previous two, so every input quad is loaded 9 times instead of 27.
struct ColMax {
  floatx4 v;
  intx4 r;      // (kt * 3 + kh) of the first maximum in scan order, -1: none (KEEP only)
};

template <bool KEEP>
__device__ __forceinline__ ColMax mp333_column(const float* __restrict__ xn, const int (&rowoff)[9], unsigned rowmask, int w, int ld) {
  ColMax m;
  m.v = floatx4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  m.r = intx4{-1, -1, -1, -1};
  floatx4 v[9];
#pragma unroll
  for (int r = 0; r < 9; ++r) v[r] = *reinterpret_cast<const floatx4*>(xn + (long long)(rowoff[r] + w) * ld);
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    if (!((rowmask >> r) & 1u)) continue;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (KEEP) {
        if (v[r][e] > m.v[e] || m.r[e] < 0) { m.v[e] = v[r][e]; m.r[e] = r; }
      } else {
        m.v[e] = fmaxf(m.v[e], v[r][e]);
      }
    }
  }
  return m;
}

template <bool KEEP>
__global__ __launch_bounds__(256) void maxpool333_fwd_kernel(const MPParams p, int seg, int nseg) {
  const rsp_pool3d_desc& d = p.d;
  const int C4 = d.C >> 2;
  const long long total = (long long)d.N * d.Di * d.Hi * nseg * C4;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += 256ll * gridDim.x) {
    const int c = (int)(i % C4) * 4;
    int q = (int)(i / C4);
    const int sg = q % nseg; q /= nseg;
    const int h = q % d.Hi; q /= d.Hi;
    const int t = q % d.Di;
    const int n = q / d.Di;
    const float* xn = p.x + (long long)n * d.Di * d.Hi * d.Wi * d.in_ld + c;
    int rowoff[9];
    unsigned rowmask = 0;
#pragma unroll
    for (int kt = 0; kt < 3; ++kt)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int id = t - 1 + kt, ih = h - 1 + kh;
        const bool ok = (unsigned)id < (unsigned)d.Di && (unsigned)ih < (unsigned)d.Hi;
        rowoff[kt * 3 + kh] = (min(max(id, 0), d.Di - 1) * d.Hi + min(max(ih, 0), d.Hi - 1)) * d.Wi;
        rowmask |= (ok ? 1u : 0u) << (kt * 3 + kh);
      }
    const int w0 = sg * seg, w1 = min(d.Wi, w0 + seg);
    ColMax none;
    none.v = floatx4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    none.r = intx4{-1, -1, -1, -1};
    ColMax prev = w0 > 0 ? mp333_column<KEEP>(xn, rowoff, rowmask, w0 - 1, d.in_ld) : none;
    ColMax cur = mp333_column<KEEP>(xn, rowoff, rowmask, w0, d.in_ld);
    const long long orow = (((long long)n * d.Do + t) * d.Ho + h) * d.Wo;
    for (int w = w0; w < w1; ++w) {
      const ColMax next = w + 1 < d.Wi ? mp333_column<KEEP>(xn, rowoff, rowmask, w + 1, d.in_ld) : none;
      floatx4 best;
      if (KEEP) {
        // first maximum in (kt, kh, kw) scan order, like max_pool3d: among equal values the smaller (kt, kh), then the smaller kw
        intx4 bi;
        const ColMax* col[3] = {&prev, &cur, &next};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float bv = -INFINITY;
          int br = -1, bk = 0;
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const float v = col[kw]->v[e];
            const int r = col[kw]->r[e];
            if (r >= 0 && (br < 0 || v > bv || (v == bv && r < br))) { bv = v; br = r; bk = kw; }
          }
          best[e] = bv;
          const int kt = br / 3, kh = br - kt * 3;
          bi[e] = ((t - 1 + kt) * d.Hi + (h - 1 + kh)) * d.Wi + (w - 1 + bk);
        }
        *reinterpret_cast<intx4*>(p.idx + (orow + w) * d.C + c) = bi;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) best[e] = fmaxf(fmaxf(prev.v[e], cur.v[e]), next.v[e]);
      }
      *reinterpret_cast<floatx4*>(p.out + (orow + w) * d.out_ld + c) = best;
      prev = cur;
      cur = next;
    }
  }
}

// Backward of the same pooling, input-centric and sliding the same way: a thread owns a run of input positions of one row; per
// step it loads the arg-max and gradient quads of ONE output column (the 9 (od, oh) rows at ow) and adds them into the three
// inputs iw = ow - 1 .. ow + 1 of its row they may point at.  Fixed order (ow ascending, then od, oh) -> deterministic.
__global__ __launch_bounds__(256) void maxpool333_bwd_kernel(const MPParams p, int seg, int nseg) {
  const rsp_pool3d_desc& d = p.d;
  const int C4 = d.C >> 2;
  const long long total = (long long)d.N * d.Di * d.Hi * nseg * C4;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += 256ll * gridDim.x) {
    const int c = (int)(i % C4) * 4;
    int q = (int)(i / C4);
    const int sg = q % nseg; q /= nseg;
    const int h = q % d.Hi; q /= d.Hi;
    const int t = q % d.Di;
    const int n = q / d.Di;
    int rowoff[9];
    unsigned rowmask = 0;
#pragma unroll
    for (int kt = 0; kt < 3; ++kt)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int od = t - 1 + kt, oh = h - 1 + kh;
        const bool ok = (unsigned)od < (unsigned)d.Do && (unsigned)oh < (unsigned)d.Ho;
        rowoff[kt * 3 + kh] = (min(max(od, 0), d.Do - 1) * d.Ho + min(max(oh, 0), d.Ho - 1)) * d.Wo;
        rowmask |= (ok ? 1u : 0u) << (kt * 3 + kh);
      }
    const long long obase = (long long)n * d.Do * d.Ho * d.Wo;
    const int lin0 = (t * d.Hi + h) * d.Wi;
    const int w0 = sg * seg, w1 = min(d.Wi, w0 + seg);
    floatx4 acc[3];      // gradients of iw = ow - 1, ow, ow + 1 while column ow is being added
#pragma unroll
    for (int k = 0; k < 3; ++k) acc[k] = floatx4{0.f, 0.f, 0.f, 0.f};
    for (int ow = max(w0 - 1, 0); ow <= min(w1, d.Wo - 1); ++ow) {
      intx4 a[9];
      floatx4 g[9];
#pragma unroll
      for (int r = 0; r < 9; ++r) {
        a[r] = *reinterpret_cast<const intx4*>(p.idx + (obase + rowoff[r] + ow) * d.C + c);
        g[r] = *reinterpret_cast<const floatx4*>(p.dout + (obase + rowoff[r] + ow) * d.out_ld + c);
      }
#pragma unroll
      for (int r = 0; r < 9; ++r) {
        if (!((rowmask >> r) & 1u)) continue;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int lin = lin0 + ow - 1 + k;
          const bool inrow = (unsigned)(ow - 1 + k) < (unsigned)d.Wi;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (inrow && a[r][e] == lin) acc[k][e] += g[r][e];
        }
      }
      // input ow - 1 has now seen its last window (ow): store it if it belongs to this thread's run
      const int done = ow - 1;
      if (done >= w0 && done < w1) *reinterpret_cast<floatx4*>(p.dx + ((long long)n * d.Di * d.Hi * d.Wi + lin0 + done) * d.in_ld + c) = acc[0];
      acc[0] = acc[1];
      acc[1] = acc[2];
      acc[2] = floatx4{0.f, 0.f, 0.f, 0.f};
    }
    // the run's last input when the row ends with it (no window ow = Wi)
    if (w1 == d.Wi) *reinterpret_cast<floatx4*>(p.dx + ((long long)n * d.Di * d.Hi * d.Wi + lin0 + w1 - 1) * d.in_ld + c) = acc[0];
  }
}

// ---- S3D-G gating ---------------------------------------------------------------------------------------------------
// sum over a slice of positions per (sample, channel): block = (sample*S + slice, 64-channel group), 4 position lanes.
// Two deterministic stages so that a (16, 8x112x112, 64) tensor is reduced by thousands of workgroups, not sixteen.
__global__ __launch_bounds__(256) void spatial_sum_kernel(const float* __restrict__ x, const float* __restrict__ x2, int P,
                                                          int C, int ld, int ld2, int S, float* __restrict__ part) {
  __shared__ float red[4][64];
  const int n = blockIdx.x / S, sl = blockIdx.x % S;
  const int c = blockIdx.y * 64 + (threadIdx.x & 63), pl = threadIdx.x >> 6;
  const int per = (P + S - 1) / S;
  const int p0 = sl * per, p1 = min(P, p0 + per);
  float s = 0.f;
  if (c < C) {
    if (x2) {
      for (int pp = p0 + pl; pp < p1; pp += 4) {
        const long long row = (long long)n * P + pp;
        s = fmaf(x[row * ld + c], x2[row * ld2 + c], s);
      }
    } else {
      for (int pp = p0 + pl; pp < p1; pp += 4) s += x[((long long)n * P + pp) * ld + c];
    }
  }
  red[pl][threadIdx.x & 63] = s;
  __syncthreads();
  if (pl == 0 && c < C) {
    const int l = threadIdx.x;
    part[((long long)n * S + sl) * C + c] = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
  }
}
// 16-byte variant (C, pitches multiples of 4, aligned bases): block = (sample*S + slice, 64-channel group) as above, but
// 16 channel quads x 16 position lanes, one float4 load per lane and position (the scalar kernel moved 256 B per wave
// instruction and ran at 1.8 TB/s over S3D-G's 80 gate reductions per step).
__global__ __launch_bounds__(256) void spatial_sum_vec_kernel(const float* __restrict__ x, const float* __restrict__ x2, int P,
                                                              int C, int ld, int ld2, int S, float* __restrict__ part) {
  __shared__ floatx4 red[16][16];
  const int n = blockIdx.x / S, sl = blockIdx.x - n * S;
  const int q = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int c = blockIdx.y * 64 + q * 4;
  const int per = (P + S - 1) / S;
  const int p0 = sl * per, p1 = min(P, p0 + per);
  floatx4 s = {0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    const float* xp = x + ((long long)n * P) * ld + c;
    if (x2) {
      const float* yp = x2 + ((long long)n * P) * ld2 + c;
      for (int pp = p0 + pl; pp < p1; pp += 16) {
        const floatx4 a = *reinterpret_cast<const floatx4*>(xp + (long long)pp * ld);
        const floatx4 b = *reinterpret_cast<const floatx4*>(yp + (long long)pp * ld2);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] = fmaf(a[e], b[e], s[e]);
      }
    } else {
      for (int pp = p0 + pl; pp < p1; pp += 16) s += *reinterpret_cast<const floatx4*>(xp + (long long)pp * ld);
    }
  }
  red[pl][q] = s;
  __syncthreads();
  if (pl == 0 && c < C) {
    floatx4 a = red[0][q];
#pragma unroll
    for (int l = 1; l < 16; ++l) a += red[l][q];
    *reinterpret_cast<floatx4*>(part + ((long long)n * S + sl) * C + c) = a;
  }
}

static void launch_spatial_sum(const float* x, const float* x2, int N, int P, int C, int ld, int ld2, int S, float* part,
                               hipStream_t s) {
  const bool vec = C % 4 == 0 && ld % 4 == 0 && rsp_aligned16(x) && rsp_aligned16(part) && (!x2 || (ld2 % 4 == 0 && rsp_aligned16(x2)));
  if (vec) hipLaunchKernelGGL(spatial_sum_vec_kernel, dim3(N * S, rsp_cdiv(C, 64)), dim3(256), 0, s, x, x2, P, C, ld, ld2, S, part);
  else hipLaunchKernelGGL(spatial_sum_kernel, dim3(N * S, rsp_cdiv(C, 64)), dim3(256), 0, s, x, x2, P, C, ld, ld2, S, part);
}

// The gate's spatial sums taken where its input is PRODUCED: a = relu(y*scale + shift) (the BatchNorm-apply in front of every
// self-gating unit, models/s3dg.py:52-72), summed per (sample, channel) over the same slices, lanes and LDS tree as
// spatial_sum_kernel / spatial_sum_vec_kernel — bit-identical partials — and stored to `act` only when a backward will need it.
// Saves the reduction's own read of the activation; with act == null (key passes) also its write.
__global__ __launch_bounds__(256) void bn_act_sum_kernel(const float* __restrict__ y, const float* __restrict__ ss, int relu,
                                                         float* __restrict__ act, int P, int C, int ld, int act_ld, int S,
                                                         float* __restrict__ part) {
  __shared__ float red[4][64];
  const int n = blockIdx.x / S, sl = blockIdx.x % S;
  const int c = blockIdx.y * 64 + (threadIdx.x & 63), pl = threadIdx.x >> 6;
  const int per = (P + S - 1) / S;
  const int p0 = sl * per, p1 = min(P, p0 + per);
  float s = 0.f;
  if (c < C) {
    const float sc = ss[c], sh = ss[C + c];
    for (int pp = p0 + pl; pp < p1; pp += 4) {
      const long long row = (long long)n * P + pp;
      float z = fmaf(y[row * ld + c], sc, sh);
      if (relu) z = fmaxf(z, 0.f);
      if (act) act[row * act_ld + c] = z;
      s += z;
    }
  }
  red[pl][threadIdx.x & 63] = s;
  __syncthreads();
  if (pl == 0 && c < C) {
    const int l = threadIdx.x;
    part[((long long)n * S + sl) * C + c] = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
  }
}
__global__ __launch_bounds__(256) void bn_act_sum_vec_kernel(const float* __restrict__ y, const float* __restrict__ ss, int relu,
                                                             float* __restrict__ act, int P, int C, int ld, int act_ld, int S,
                                                             float* __restrict__ part) {
  __shared__ floatx4 red[16][16];
  const int n = blockIdx.x / S, sl = blockIdx.x - n * S;
  const int q = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int c = blockIdx.y * 64 + q * 4;
  const int per = (P + S - 1) / S;
  const int p0 = sl * per, p1 = min(P, p0 + per);
  floatx4 s = {0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    const floatx4 sc = *reinterpret_cast<const floatx4*>(ss + c), sh = *reinterpret_cast<const floatx4*>(ss + C + c);
    const float* yp = y + ((long long)n * P) * ld + c;
    float* ap = act ? act + ((long long)n * P) * act_ld + c : nullptr;
    for (int pp = p0 + pl; pp < p1; pp += 16) {
      const floatx4 v = *reinterpret_cast<const floatx4*>(yp + (long long)pp * ld);
      floatx4 z;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        z[e] = fmaf(v[e], sc[e], sh[e]);
        if (relu) z[e] = fmaxf(z[e], 0.f);
      }
      if (ap) *reinterpret_cast<floatx4*>(ap + (long long)pp * act_ld) = z;
      s += z;
    }
  }
  red[pl][q] = s;
  __syncthreads();
  if (pl == 0 && c < C) {
    floatx4 a = red[0][q];
#pragma unroll
    for (int l = 1; l < 16; ++l) a += red[l][q];
    *reinterpret_cast<floatx4*>(part + ((long long)n * S + sl) * C + c) = a;
  }
}

// Backward twin: sum_p dout * a with a = relu(y*scale + shift) recomputed from y (the forward kept no activation) — the lanes,
// slices and tree of spatial_sum_kernel / spatial_sum_vec_kernel with x2 = a.
__global__ __launch_bounds__(256) void dout_act_sum_kernel(const float* __restrict__ dout, const float* __restrict__ y,
                                                           const float* __restrict__ ss, int relu, int P, int C, int dout_ld, int y_ld,
                                                           int S, float* __restrict__ part) {
  __shared__ float red[4][64];
  const int n = blockIdx.x / S, sl = blockIdx.x % S;
  const int c = blockIdx.y * 64 + (threadIdx.x & 63), pl = threadIdx.x >> 6;
  const int per = (P + S - 1) / S;
  const int p0 = sl * per, p1 = min(P, p0 + per);
  float s = 0.f;
  if (c < C) {
    const float sc = ss[c], sh = ss[C + c];
    for (int pp = p0 + pl; pp < p1; pp += 4) {
      const long long row = (long long)n * P + pp;
      float z = fmaf(y[row * y_ld + c], sc, sh);
      if (relu) z = fmaxf(z, 0.f);
      s = fmaf(dout[row * dout_ld + c], z, s);
    }
  }
  red[pl][threadIdx.x & 63] = s;
  __syncthreads();
  if (pl == 0 && c < C) {
    const int l = threadIdx.x;
    part[((long long)n * S + sl) * C + c] = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
  }
}
__global__ __launch_bounds__(256) void dout_act_sum_vec_kernel(const float* __restrict__ dout, const float* __restrict__ y,
                                                               const float* __restrict__ ss, int relu, int P, int C, int dout_ld,
                                                               int y_ld, int S, float* __restrict__ part) {
  __shared__ floatx4 red[16][16];
  const int n = blockIdx.x / S, sl = blockIdx.x - n * S;
  const int q = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int c = blockIdx.y * 64 + q * 4;
  const int per = (P + S - 1) / S;
  const int p0 = sl * per, p1 = min(P, p0 + per);
  floatx4 s = {0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    const floatx4 sc = *reinterpret_cast<const floatx4*>(ss + c), sh = *reinterpret_cast<const floatx4*>(ss + C + c);
    const float* dp = dout + ((long long)n * P) * dout_ld + c;
    const float* yp = y + ((long long)n * P) * y_ld + c;
    for (int pp = p0 + pl; pp < p1; pp += 16) {
      const floatx4 a = *reinterpret_cast<const floatx4*>(dp + (long long)pp * dout_ld);
      const floatx4 v = *reinterpret_cast<const floatx4*>(yp + (long long)pp * y_ld);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float z = fmaf(v[e], sc[e], sh[e]);
        if (relu) z = fmaxf(z, 0.f);
        s[e] = fmaf(a[e], z, s[e]);
      }
    }
  }
  red[pl][q] = s;
  __syncthreads();
  if (pl == 0 && c < C) {
    floatx4 a = red[0][q];
#pragma unroll
    for (int l = 1; l < 16; ++l) a += red[l][q];
    *reinterpret_cast<floatx4*>(part + ((long long)n * S + sl) * C + c) = a;
  }
}

// out[n][c] = scale * sum_s part[n][s][c]  (* g*(1-g) when gate != null: the sigmoid derivative of the gating backward)
__global__ void spatial_sum_final_kernel(const float* __restrict__ part, int N, int C, int S, float scale,
                                         const float* __restrict__ gate, float* __restrict__ out) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= (long long)N * C) return;
  const int n = (int)(i / C), c = (int)(i % C);
  float s = 0.f;
  for (int k = 0; k < S; ++k) s += part[((long long)n * S + k) * C + c];
  s *= scale;
  if (gate) {
    const float g = gate[i];
    s *= g * (1.f - g);
  }
  out[i] = s;
}

// gate[n][co] = sigmoid(b[co] + sum_ci w[co][ci] * mean[n][ci]) ; one wave per output
__global__ __launch_bounds__(256) void gate_fc_kernel(const float* __restrict__ mean, const float* __restrict__ w,
                                                      const float* __restrict__ b, int N, int C,
                                                      float* __restrict__ gate) {
  const int lane = threadIdx.x & 63;
  const long long o = blockIdx.x * 4ll + (threadIdx.x >> 6);
  if (o >= (long long)N * C) return;
  const int n = (int)(o / C), co = (int)(o % C);
  float s = 0.f;
  for (int ci = lane; ci < C; ci += 64) s = fmaf(w[(long long)co * C + ci], mean[(long long)n * C + ci], s);
  s = rsp_wave_sum(s);
  if (lane == 0) gate[o] = 1.f / (1.f + expf(-(s + b[co])));
}

// out = x * gate[n][c]
__global__ __launch_bounds__(256) void gate_apply_kernel(const float* __restrict__ x, const float* __restrict__ gate,
                                                         int P, int C, int in_ld, int out_ld, long long total,
                                                         float* __restrict__ out) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += 256ll * gridDim.x) {
    const int c = (int)(i % C);
    const long long row = i / C;
    const int n = (int)(row / P);
    out[row * out_ld + c] = x[row * in_ld + c] * gate[(long long)n * C + c];
  }
}

// backward: dx = dout*gate + (dmean/P); dpre[n][c] = (sum_p dout*x) * gate*(1-gate)  (spatial_sum kernels above)
// dw[co][ci] = sum_n dpre[n][co]*mean[n][ci]; db[co] = sum_n dpre[n][co]; dmean[n][ci] = sum_co dpre[n][co] w[co][ci]
__global__ void gate_bwd_param_kernel(const float* __restrict__ dpre, const float* __restrict__ mean, int N, int C,
                                      float* __restrict__ dw, float* __restrict__ db) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= (long long)C * C) return;
  const int co = (int)(i / C), ci = (int)(i % C);
  float s = 0.f, sb = 0.f;
  for (int n = 0; n < N; ++n) {
    const float d = dpre[(long long)n * C + co];
    s = fmaf(d, mean[(long long)n * C + ci], s);
    sb += d;
  }
  dw[i] = s;
  if (ci == 0) db[co] = sb;
}
// dmean[n][ci] = sum_co dpre[n][co] * w[co][ci].  Block = (64 input channels, sample) x 16 lanes over co (one thread per
// output walked all C rows of w serially: 56 us for a 16 x 384 result).
__global__ __launch_bounds__(1024) void gate_bwd_dmean_kernel(const float* __restrict__ dpre, const float* __restrict__ w, int N, int C,
                                                              float* __restrict__ dmean) {
  __shared__ float red[16][64];
  const int cl = threadIdx.x & 63, l = threadIdx.x >> 6;
  const int ci = blockIdx.x * 64 + cl, n = blockIdx.y;
  float s = 0.f;
  if (ci < C)
    for (int co = l; co < C; co += 16) s = fmaf(dpre[(long long)n * C + co], w[(long long)co * C + ci], s);
  red[l][cl] = s;
  __syncthreads();
  if (l == 0 && ci < C) {
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) a += red[k][cl];
    dmean[(long long)n * C + ci] = a;
  }
}
__global__ __launch_bounds__(256) void gate_bwd_apply_kernel(const float* __restrict__ dout, const float* __restrict__ gate,
                                                             const float* __restrict__ dmean, int P, int C, int dout_ld,
                                                             int dx_ld, long long total, float* __restrict__ dx) {
  const float invP = 1.f / (float)P;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += 256ll * gridDim.x) {
    const int c = (int)(i % C);
    const long long row = i / C;
    const int n = (int)(row / P);
    dx[row * dx_ld + c] = fmaf(dout[row * dout_ld + c], gate[(long long)n * C + c], dmean[(long long)n * C + c] * invP);
  }
}

// 16-byte variants (C, pitches multiples of 4, aligned pointers): blockIdx.y = sample, threads over (position, channel quad) —
// no per-element 64-bit division, a quarter of the memory instructions.  dmean == nullptr: forward (out = x * gate).
__global__ __launch_bounds__(256) void gate_apply_vec_kernel(const float* __restrict__ x, const float* __restrict__ gate,
                                                             const float* __restrict__ dmean, int P, int C, int in_ld, int out_ld,
                                                             float* __restrict__ out) {
  const int n = blockIdx.y, cq = C >> 2;
  const long long rows0 = (long long)n * P;
  const float invP = 1.f / (float)P;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < P * cq; i += 256 * gridDim.x) {
    const int pp = i / cq, c = (i - pp * cq) * 4;
    const floatx4 v = *reinterpret_cast<const floatx4*>(x + (rows0 + pp) * in_ld + c);
    const floatx4 g = *reinterpret_cast<const floatx4*>(gate + (long long)n * C + c);
    floatx4 o = v * g;
    if (dmean) o += *reinterpret_cast<const floatx4*>(dmean + (long long)n * C + c) * invP;
    *reinterpret_cast<floatx4*>(out + (rows0 + pp) * out_ld + c) = o;
  }
}

bool gate_vec_ok(int P, int C, int ld_a, int ld_b, const void* a, const void* b, const void* g, const void* m) {
  return C % 4 == 0 && ld_a % 4 == 0 && ld_b % 4 == 0 && rsp_aligned16(a) && rsp_aligned16(b) && rsp_aligned16(g) &&
         (!m || rsp_aligned16(m)) && (long long)P * (C / 4) < (1ll << 31);
}

int gate_vec_blocks(int P, int C, int N) {
  long long b = ((long long)P * (C / 4) + 255) / 256;
  const long long cap = 4096 / (N > 0 ? N : 1) + 1;      // ~4096 workgroups over all samples
  return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}

bool mp_ok(const rsp_pool3d_desc* d) {
  if (!d) return false;
  if (d->N <= 0 || d->C <= 0 || d->kT <= 0 || d->kH <= 0 || d->kW <= 0) return false;
  if (d->sT <= 0 || d->sH <= 0 || d->sW <= 0 || d->pT < 0 || d->pH < 0 || d->pW < 0) return false;
  if (d->Do != (d->Di + 2 * d->pT - d->kT) / d->sT + 1) return false;
  if (d->Ho != (d->Hi + 2 * d->pH - d->kH) / d->sH + 1) return false;
  if (d->Wo != (d->Wi + 2 * d->pW - d->kW) / d->sW + 1) return false;
  if (d->in_ld < d->C || d->out_ld < d->C) return false;
  if ((long long)d->Di * d->Hi * d->Wi >= (1ll << 31)) return false;
  return true;
}
int gate_splits(int P) {
  const int s = (P + 255) / 256;
  return s > 64 ? 64 : (s < 1 ? 1 : s);
}
bool mp_same333(const rsp_pool3d_desc* d) {
  return d->kT == 3 && d->kH == 3 && d->kW == 3 && d->sT == 1 && d->sH == 1 && d->sW == 1 && d->pT == 1 && d->pH == 1 && d->pW == 1;
}

// positions along W per thread: whole rows.  Shorter runs (more threads, one re-read column on either side of each run) were swept
// on S3D-G's nine branch pools — 1 / 2 / 3 / 4 / 7 / 14 / 28 — and lose everywhere: the kernels are bound by L1 traffic, not by
// occupancy (28x28x192 forward 106 / 145 / 125 / 113 / 99 / 72 / 56 us; 14x14x480: 38 / 40 / 38 / 32 / 33 / 20 / 20 us).
int mp333_seg(const rsp_pool3d_desc* d) {
  return d->Wi;
}

int grid_for(long long total) {
  long long b = (total + 255) / 256;
  return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" {

int rsp_maxpool3d_fwd(const rsp_pool3d_desc* d, const float* x, float* out, int32_t* argmax, void* stream) {
  RSP_REQUIRE(mp_ok(d), "rsp_maxpool3d_fwd: bad descriptor");
  RSP_REQUIRE(x && out, "rsp_maxpool3d_fwd: null pointer");
  MPParams p;
  memset(&p, 0, sizeof p);
  p.d = *d; p.x = x; p.out = out; p.idx = argmax;
  const bool vec = d->C % 4 == 0 && d->in_ld % 4 == 0 && d->out_ld % 4 == 0 && rsp_aligned16(x) && rsp_aligned16(out) &&
                   (!argmax || rsp_aligned16(argmax)) && (long long)d->N * d->Do * d->Ho * d->Wo * (d->C / 4) < (1ll << 31);
  if (vec) {
    const dim3 grid(grid_for((long long)d->N * d->Do * d->Ho * d->Wo * (d->C / 4)));
    hipStream_t st = (hipStream_t)stream;
    const int kkk = d->kT * 100 + d->kH * 10 + d->kW;
    if (mp_same333(d)) {
      const int seg = mp333_seg(d), nseg = rsp_cdiv(d->Wi, seg);
      const dim3 g3(grid_for((long long)d->N * d->Di * d->Hi * nseg * (d->C / 4)));
      if (argmax) hipLaunchKernelGGL((maxpool333_fwd_kernel<true>), g3, dim3(256), 0, st, p, seg, nseg);
      else hipLaunchKernelGGL((maxpool333_fwd_kernel<false>), g3, dim3(256), 0, st, p, seg, nseg);
    } else if (kkk == 333) {
      if (argmax) hipLaunchKernelGGL((maxpool_fwd_vec_kernel<true, 3, 3, 3>), grid, dim3(256), 0, st, p);
      else hipLaunchKernelGGL((maxpool_fwd_vec_kernel<false, 3, 3, 3>), grid, dim3(256), 0, st, p);
    } else if (kkk == 133) {
      if (argmax) hipLaunchKernelGGL((maxpool_fwd_vec_kernel<true, 1, 3, 3>), grid, dim3(256), 0, st, p);
      else hipLaunchKernelGGL((maxpool_fwd_vec_kernel<false, 1, 3, 3>), grid, dim3(256), 0, st, p);
    } else {
      if (argmax) hipLaunchKernelGGL((maxpool_fwd_vec_kernel<true, 0, 0, 0>), grid, dim3(256), 0, st, p);
      else hipLaunchKernelGGL((maxpool_fwd_vec_kernel<false, 0, 0, 0>), grid, dim3(256), 0, st, p);
    }
  } else
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for((long long)d->N * d->Do * d->Ho * d->Wo * d->C)), dim3(256), 0,
                       (hipStream_t)stream, p);
  return rsp_check_launch("maxpool_fwd_kernel");
}

int rsp_bn_act_maxpool_applicable(const rsp_pool3d_desc* d) {
  if (!mp_ok(d)) return 0;
  const int kkk = d->kT * 100 + d->kH * 10 + d->kW;
  return (kkk == 333 || kkk == 133) && d->C % 4 == 0 && d->in_ld % 4 == 0 && d->out_ld % 4 == 0 &&
         (long long)d->N * d->Do * d->Ho * d->Wo * (d->C / 4) < (1ll << 31);
}

int rsp_bn_act_maxpool_gate_fwd(const rsp_pool3d_desc* d, const float* y, const float* scale_shift, int relu, const float* gate, float* out,
                                int32_t* argmax, void* stream) {
  RSP_REQUIRE(rsp_bn_act_maxpool_applicable(d), "rsp_bn_act_maxpool_fwd: 3x3x3 / 1x3x3 windows, channels and pitches multiples of 4");
  RSP_REQUIRE(y && scale_shift && out, "rsp_bn_act_maxpool_fwd: null pointer");
  RSP_REQUIRE(rsp_aligned16(y) && rsp_aligned16(out) && rsp_aligned16(scale_shift) && (!argmax || rsp_aligned16(argmax)) &&
                  (!gate || rsp_aligned16(gate)),
              "rsp_bn_act_maxpool_fwd: pointers must be 16-byte aligned");
  MPParams p;
  memset(&p, 0, sizeof p);
  p.d = *d; p.x = y; p.out = out; p.idx = argmax;
  const dim3 grid(grid_for((long long)d->N * d->Do * d->Ho * d->Wo * (d->C / 4)));
  hipStream_t st = (hipStream_t)stream;
  if (d->kT == 3) {
    if (argmax) hipLaunchKernelGGL((bn_maxpool_fwd_vec_kernel<true, 3, 3, 3>), grid, dim3(256), 0, st, p, scale_shift, relu, gate);
    else hipLaunchKernelGGL((bn_maxpool_fwd_vec_kernel<false, 3, 3, 3>), grid, dim3(256), 0, st, p, scale_shift, relu, gate);
  } else {
    if (argmax) hipLaunchKernelGGL((bn_maxpool_fwd_vec_kernel<true, 1, 3, 3>), grid, dim3(256), 0, st, p, scale_shift, relu, gate);
    else hipLaunchKernelGGL((bn_maxpool_fwd_vec_kernel<false, 1, 3, 3>), grid, dim3(256), 0, st, p, scale_shift, relu, gate);
  }
  return rsp_check_launch("bn_maxpool_fwd_vec_kernel");
}

int rsp_bn_act_maxpool_fwd(const rsp_pool3d_desc* d, const float* y, const float* scale_shift, int relu, float* out, int32_t* argmax,
                           void* stream) {
  return rsp_bn_act_maxpool_gate_fwd(d, y, scale_shift, relu, nullptr, out, argmax, stream);
}

int rsp_maxpool3d_bwd(const rsp_pool3d_desc* d, const float* dout, const int32_t* argmax, float* dx, void* stream) {
  RSP_REQUIRE(mp_ok(d), "rsp_maxpool3d_bwd: bad descriptor");
  RSP_REQUIRE(dout && argmax && dx, "rsp_maxpool3d_bwd: null pointer");
  MPParams p;
  memset(&p, 0, sizeof p);
  p.d = *d; p.dout = dout; p.idx = const_cast<int*>(argmax); p.dx = dx;
  const bool vec = d->C % 4 == 0 && d->in_ld % 4 == 0 && d->out_ld % 4 == 0 && rsp_aligned16(dout) && rsp_aligned16(dx) &&
                   rsp_aligned16(argmax) && (long long)d->N * d->Di * d->Hi * d->Wi * (d->C / 4) < (1ll << 31);
  if (vec && mp_same333(d)) {
    const int seg = mp333_seg(d), nseg = rsp_cdiv(d->Wi, seg);
    hipLaunchKernelGGL(maxpool333_bwd_kernel, dim3(grid_for((long long)d->N * d->Di * d->Hi * nseg * (d->C / 4))), dim3(256), 0,
                       (hipStream_t)stream, p, seg, nseg);
  } else if (vec)
    hipLaunchKernelGGL(maxpool_bwd_vec_kernel, dim3(grid_for((long long)d->N * d->Di * d->Hi * d->Wi * (d->C / 4))), dim3(256), 0,
                       (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for((long long)d->N * d->Di * d->Hi * d->Wi * d->C)), dim3(256), 0,
                       (hipStream_t)stream, p);
  return rsp_check_launch("maxpool_bwd_kernel");
}

size_t rsp_gate_fwd_workspace(int32_t N, int32_t P, int32_t C) { return (size_t)N * gate_splits(P) * C * sizeof(float); }

int rsp_gate_fwd(const float* x, int32_t N, int32_t P, int32_t C, int32_t in_ld, const float* w, const float* b,
                 float* out, int32_t out_ld, float* mean, float* gate, void* workspace, size_t workspace_bytes,
                 void* stream) {
  RSP_REQUIRE(x && w && b && out && mean && gate && workspace, "rsp_gate_fwd: null pointer");
  RSP_REQUIRE(N > 0 && P > 0 && C > 0 && in_ld >= C && out_ld >= C, "rsp_gate_fwd: bad size");
  if (workspace_bytes < rsp_gate_fwd_workspace(N, P, C)) {
    rsp_set_error("rsp_gate_fwd: workspace too small");
    return RSP_EWORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  const int S = gate_splits(P);
  float* part = reinterpret_cast<float*>(workspace);
  launch_spatial_sum(x, nullptr, N, P, C, in_ld, 0, S, part, s);
  int rc = rsp_check_launch("spatial_sum_kernel");
  if (rc != RSP_OK) return rc;
  hipLaunchKernelGGL(spatial_sum_final_kernel, dim3(rsp_cdiv((long long)N * C, 256)), dim3(256), 0, s, part, N, C, S, 1.f / (float)P,
                     (const float*)nullptr, mean);
  rc = rsp_check_launch("spatial_sum_final_kernel");
  if (rc != RSP_OK) return rc;
  hipLaunchKernelGGL(gate_fc_kernel, dim3(rsp_cdiv((long long)N * C, 4)), dim3(256), 0, s, mean, w, b, N, C, gate);
  rc = rsp_check_launch("gate_fc_kernel");
  if (rc != RSP_OK) return rc;
  const long long total = (long long)N * P * C;
  if (gate_vec_ok(P, C, in_ld, out_ld, x, out, gate, nullptr))
    hipLaunchKernelGGL(gate_apply_vec_kernel, dim3(gate_vec_blocks(P, C, N), N), dim3(256), 0, s, x, gate, (const float*)nullptr, P, C,
                       in_ld, out_ld, out);
  else
    hipLaunchKernelGGL(gate_apply_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, gate, P, C, in_ld, out_ld, total, out);
  return rsp_check_launch("gate_apply_kernel");
}

int rsp_bn_gate_sums(const float* y, int32_t N, int32_t P, int32_t C, int32_t y_ld, const float* scale_shift, int relu,
                     float* act, int32_t act_ld, const float* w, const float* b, float* mean, float* gate, void* workspace,
                     size_t workspace_bytes, void* stream) {
  RSP_REQUIRE(y && scale_shift && w && b && mean && gate && workspace, "rsp_bn_gate_sums: null pointer");
  RSP_REQUIRE(N > 0 && P > 0 && C > 0 && y_ld >= C && (!act || act_ld >= C), "rsp_bn_gate_sums: bad size");
  if (workspace_bytes < rsp_gate_fwd_workspace(N, P, C)) {
    rsp_set_error("rsp_bn_gate_sums: workspace too small");
    return RSP_EWORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  const int S = gate_splits(P);
  float* part = reinterpret_cast<float*>(workspace);
  const bool vec = C % 4 == 0 && y_ld % 4 == 0 && rsp_aligned16(y) && rsp_aligned16(part) && rsp_aligned16(scale_shift) &&
                   (!act || (act_ld % 4 == 0 && rsp_aligned16(act)));
  if (vec)
    hipLaunchKernelGGL(bn_act_sum_vec_kernel, dim3(N * S, rsp_cdiv(C, 64)), dim3(256), 0, s, y, scale_shift, relu, act, P, C, y_ld, act_ld,
                       S, part);
  else
    hipLaunchKernelGGL(bn_act_sum_kernel, dim3(N * S, rsp_cdiv(C, 64)), dim3(256), 0, s, y, scale_shift, relu, act, P, C, y_ld, act_ld, S,
                       part);
  int rc = rsp_check_launch("bn_act_sum_kernel");
  if (rc != RSP_OK) return rc;
  hipLaunchKernelGGL(spatial_sum_final_kernel, dim3(rsp_cdiv((long long)N * C, 256)), dim3(256), 0, s, part, N, C, S, 1.f / (float)P,
                     (const float*)nullptr, mean);
  rc = rsp_check_launch("spatial_sum_final_kernel");
  if (rc != RSP_OK) return rc;
  hipLaunchKernelGGL(gate_fc_kernel, dim3(rsp_cdiv((long long)N * C, 4)), dim3(256), 0, s, mean, w, b, N, C, gate);
  return rsp_check_launch("gate_fc_kernel");
}

int rsp_gate_apply(const float* x, int32_t N, int32_t P, int32_t C, int32_t in_ld, const float* gate, float* out, int32_t out_ld,
                   void* stream) {
  RSP_REQUIRE(x && gate && out, "rsp_gate_apply: null pointer");
  RSP_REQUIRE(N > 0 && P > 0 && C > 0 && in_ld >= C && out_ld >= C, "rsp_gate_apply: bad size");
  hipStream_t s = (hipStream_t)stream;
  const long long total = (long long)N * P * C;
  if (gate_vec_ok(P, C, in_ld, out_ld, x, out, gate, nullptr))
    hipLaunchKernelGGL(gate_apply_vec_kernel, dim3(gate_vec_blocks(P, C, N), N), dim3(256), 0, s, x, gate, (const float*)nullptr, P, C,
                       in_ld, out_ld, out);
  else
    hipLaunchKernelGGL(gate_apply_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, gate, P, C, in_ld, out_ld, total, out);
  return rsp_check_launch("gate_apply_kernel");
}

size_t rsp_gate_bwd_workspace(int32_t N, int32_t P, int32_t C) {
  return (size_t)(2 + gate_splits(P)) * N * C * sizeof(float);
}

int rsp_gate_bwd(const float* x, const float* dout, int32_t N, int32_t P, int32_t C, int32_t x_ld, int32_t dout_ld,
                 const float* w, const float* mean, const float* gate, float* dx, int32_t dx_ld, float* dw, float* db,
                 void* workspace, size_t workspace_bytes, void* stream) {
  RSP_REQUIRE(x && dout && w && mean && gate && dx && dw && db && workspace, "rsp_gate_bwd: null pointer");
  RSP_REQUIRE(N > 0 && P > 0 && C > 0 && x_ld >= C && dout_ld >= C && dx_ld >= C, "rsp_gate_bwd: bad size");
  if (workspace_bytes < rsp_gate_bwd_workspace(N, P, C)) {
    rsp_set_error("rsp_gate_bwd: workspace too small");
    return RSP_EWORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  float* dpre = reinterpret_cast<float*>(workspace);
  float* dmean = dpre + (size_t)N * C;
  const int S = gate_splits(P);
  float* part = dmean + (size_t)N * C;
  launch_spatial_sum(dout, x, N, P, C, dout_ld, x_ld, S, part, s);
  int rc = rsp_check_launch("spatial_sum_kernel(bwd)");
  if (rc != RSP_OK) return rc;
  hipLaunchKernelGGL(spatial_sum_final_kernel, dim3(rsp_cdiv((long long)N * C, 256)), dim3(256), 0, s, part, N, C, S, 1.f, gate, dpre);
  rc = rsp_check_launch("spatial_sum_final_kernel(bwd)");
  if (rc != RSP_OK) return rc;
  hipLaunchKernelGGL(gate_bwd_param_kernel, dim3(rsp_cdiv((long long)C * C, 256)), dim3(256), 0, s, dpre, mean, N, C, dw, db);
  rc = rsp_check_launch("gate_bwd_param_kernel");
  if (rc != RSP_OK) return rc;
  hipLaunchKernelGGL(gate_bwd_dmean_kernel, dim3(rsp_cdiv(C, 64), N), dim3(1024), 0, s, dpre, w, N, C, dmean);
  rc = rsp_check_launch("gate_bwd_dmean_kernel");
  if (rc != RSP_OK) return rc;
  const long long total = (long long)N * P * C;
  if (gate_vec_ok(P, C, dout_ld, dx_ld, dout, dx, gate, dmean))
    hipLaunchKernelGGL(gate_apply_vec_kernel, dim3(gate_vec_blocks(P, C, N), N), dim3(256), 0, s, dout, gate, dmean, P, C, dout_ld,
                       dx_ld, dx);
  else
    hipLaunchKernelGGL(gate_bwd_apply_kernel, dim3(grid_for(total)), dim3(256), 0, s, dout, gate, dmean, P, C, dout_ld, dx_ld,
                       total, dx);
  return rsp_check_launch("gate_bwd_apply_kernel");
}

// Parameter half of the gating backward for a unit whose forward kept no activation (rsp_bn_gate_sums with act == NULL): the
// activation is recomputed from y.  Writes dw, db and dmean ([N][C], = d loss / d sum_p a before the 1/P) for
// rsp_bn_act_pool_bwd_g, which folds the data half (dx = dout*gate + dmean/P) into the BatchNorm backward in front of it.
int rsp_gate_bwd_params(const float* y, const float* scale_shift, int relu, const float* dout, int32_t N, int32_t P, int32_t C,
                        int32_t y_ld, int32_t dout_ld, const float* w, const float* mean, const float* gate, float* dw, float* db,
                        float* dmean, void* workspace, size_t workspace_bytes, void* stream) {
  RSP_REQUIRE(y && scale_shift && dout && w && mean && gate && dw && db && dmean && workspace, "rsp_gate_bwd_params: null pointer");
  RSP_REQUIRE(N > 0 && P > 0 && C > 0 && y_ld >= C && dout_ld >= C, "rsp_gate_bwd_params: bad size");
  if (workspace_bytes < rsp_gate_bwd_workspace(N, P, C)) {
    rsp_set_error("rsp_gate_bwd_params: workspace too small");
    return RSP_EWORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  float* dpre = reinterpret_cast<float*>(workspace);
  const int S = gate_splits(P);
  float* part = dpre + (size_t)2 * N * C;
  const bool vec = C % 4 == 0 && y_ld % 4 == 0 && dout_ld % 4 == 0 && rsp_aligned16(y) && rsp_aligned16(dout) && rsp_aligned16(part) &&
                   rsp_aligned16(scale_shift);
  if (vec)
    hipLaunchKernelGGL(dout_act_sum_vec_kernel, dim3(N * S, rsp_cdiv(C, 64)), dim3(256), 0, s, dout, y, scale_shift, relu, P, C, dout_ld,
                       y_ld, S, part);
  else
    hipLaunchKernelGGL(dout_act_sum_kernel, dim3(N * S, rsp_cdiv(C, 64)), dim3(256), 0, s, dout, y, scale_shift, relu, P, C, dout_ld, y_ld,
                       S, part);
  int rc = rsp_check_launch("dout_act_sum_kernel");
  if (rc != RSP_OK) return rc;
  hipLaunchKernelGGL(spatial_sum_final_kernel, dim3(rsp_cdiv((long long)N * C, 256)), dim3(256), 0, s, part, N, C, S, 1.f, gate, dpre);
  rc = rsp_check_launch("spatial_sum_final_kernel(bwd)");
  if (rc != RSP_OK) return rc;
  hipLaunchKernelGGL(gate_bwd_param_kernel, dim3(rsp_cdiv((long long)C * C, 256)), dim3(256), 0, s, dpre, mean, N, C, dw, db);
  rc = rsp_check_launch("gate_bwd_param_kernel");
  if (rc != RSP_OK) return rc;
  hipLaunchKernelGGL(gate_bwd_dmean_kernel, dim3(rsp_cdiv(C, 64), N), dim3(1024), 0, s, dpre, w, N, C, dmean);
  return rsp_check_launch("gate_bwd_dmean_kernel");
}

}  // extern "C"
