// Projection heads, contrastive logits, InfoNCE + ranking loss, queue update.
// Reference call sites: moco/split_wrapper.py:138-152,163-169 (heads), moco/builder_diffspeed_diffloss.py:521-538
// (logits), :263-283 (Loss), :345-359 (queue).  All of this is <0.1 % of the step; kernels are written for
// determinism (fixed summation orders) and exactness, the q·queue contraction runs on the fp32 MFMA pipe.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------------------------
// heads
// ------------------------------------------------------------------------------------------------------------------
// raw[h][b][o] = b_h[o] + sum_c W_h[o][c] * pooled[b][c]: workgroup = (sample, 32 output rows of the stacked [W1; W2]), one
// wave per 8 rows, every weight load of a wave independent (8 rows x C/64 columns in flight).  (Round 1 ran the whole head
// in one workgroup per sample: ~64 dependent L2 round trips on only B CUs, 205-414 us measured.)
// Summation order per output: lane-strided partial sums over c, then the wave butterfly.
__global__ __launch_bounds__(256) void head_raw_kernel(const float* __restrict__ pooled, int B, int C,
                                                       const float* __restrict__ w1, const float* __restrict__ b1,
                                                       const float* __restrict__ w2, const float* __restrict__ b2, int dim,
                                                       float* __restrict__ raw) {
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int o0 = blockIdx.y * 32 + wave * 8;
  const float* x = pooled + (long long)b * C;
  const float* rows[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int o = min(o0 + r, 2 * dim - 1);
    rows[r] = (o >= dim ? w2 + (long long)(o - dim) * C : w1 + (long long)o * C);
  }
  float acc[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) acc[r] = 0.f;
#pragma unroll 4
  for (int c = lane; c < C; c += 64) {
    const float xv = x[c];
#pragma unroll
    for (int r = 0; r < 8; ++r) acc[r] = fmaf(rows[r][c], xv, acc[r]);
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int o = o0 + r;
    const float sum = rsp_wave_sum(acc[r]);
    if (lane == 0 && o < 2 * dim) {
      const int hsel = o >= dim, oo = o - hsel * dim;
      raw[((long long)hsel * B + b) * dim + oo] = sum + (hsel ? b2 : b1)[oo];
    }
  }
}

// out_h[b][:] = raw[h][b][:] / max(|raw[h][b][:]|, 1e-12); one wave per (sample, head)
__global__ __launch_bounds__(64) void head_norm_kernel(const float* __restrict__ raw, int B, int dim, float* __restrict__ out1,
                                                       float* __restrict__ out2) {
  const int b = blockIdx.x, hsel = blockIdx.y, lane = threadIdx.x;
  const float* r = raw + ((long long)hsel * B + b) * dim;
  float ss = 0.f;
  for (int o = lane; o < dim; o += 64) ss = fmaf(r[o], r[o], ss);
  ss = rsp_wave_sum(ss);
  const float nrm = fmaxf(sqrtf(ss), 1e-12f);
  float* out = (hsel ? out2 : out1) + (long long)b * dim;
  for (int o = lane; o < dim; o += 64) out[o] = r[o] / nrm;
}

// draw[h][b][:] = (dout - y*(y.dout)) / n   with y = raw/n, n = max(|raw|, eps)
__global__ __launch_bounds__(64) void head_bwd_norm_kernel(const float* __restrict__ dout1, const float* __restrict__ dout2,
                                                           const float* __restrict__ raw, int B, int dim,
                                                           float* __restrict__ draw) {
  const int b = blockIdx.x, hsel = blockIdx.y, lane = threadIdx.x;
  const float* r = raw + ((long long)hsel * B + b) * dim;
  const float* g = (hsel ? dout2 : dout1) + (long long)b * dim;
  float ss = 0.f, dot = 0.f;
  for (int o = lane; o < dim; o += 64) {
    ss = fmaf(r[o], r[o], ss);
    dot = fmaf(r[o], g[o], dot);
  }
  ss = rsp_wave_sum(ss);
  dot = rsp_wave_sum(dot);
  const float nrm = sqrtf(ss);
  float* d = draw + ((long long)hsel * B + b) * dim;
  if (nrm > 1e-12f) {
    const float inv = 1.f / nrm;
    const float k = dot * inv * inv;  // (y.dout)/n with y = r/n  ->  r * dot / n^2 ... applied below
    for (int o = lane; o < dim; o += 64) d[o] = (g[o] - r[o] * k) * inv;
  } else {
    for (int o = lane; o < dim; o += 64) d[o] = g[o] * 1e12f;  // clamp branch: y = x/eps, norm term has zero grad
  }
}

// dW_h[o][c] = sum_b draw[h][b][o]*pooled[b][c];  db_h[o] = sum_b draw[h][b][o]
__global__ __launch_bounds__(256) void head_bwd_w_kernel(const float* __restrict__ draw, const float* __restrict__ pooled,
                                                         int B, int C, int dim, float* __restrict__ dw1,
                                                         float* __restrict__ db1, float* __restrict__ dw2,
                                                         float* __restrict__ db2) {
  const long long total = 2ll * dim * C;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += 256ll * gridDim.x) {
    const int c = (int)(i % C);
    const int o2 = (int)(i / C);
    const int hsel = o2 / dim, o = o2 - hsel * dim;
    float s = 0.f, sb = 0.f;
    for (int b = 0; b < B; ++b) {
      const float d = draw[((long long)hsel * B + b) * dim + o];
      s = fmaf(d, pooled[(long long)b * C + c], s);
      sb += d;
    }
    (hsel ? dw2 : dw1)[(long long)o * C + c] = s;
    if (c == 0) (hsel ? db2 : db1)[o] = sb;
  }
}

// dfeat[b][p][c] = (sum_h sum_o draw[h][b][o] * W_h[o][c]) / P
__global__ __launch_bounds__(256) void head_bwd_x_kernel(const float* __restrict__ draw, const float* __restrict__ w1,
                                                         const float* __restrict__ w2, int B, int P, int C, int ld,
                                                         int dim, float* __restrict__ dfeat) {
  extern __shared__ __attribute__((aligned(16))) float sd[];  // [2*dim] draw of this sample
  const int b = blockIdx.x, t = threadIdx.x;
  for (int o = t; o < 2 * dim; o += 256) sd[o] = draw[((long long)(o / dim) * B + b) * dim + (o % dim)];
  __syncthreads();
  const float invP = 1.f / (float)P;
  for (int c = blockIdx.y * 256 + t; c < C; c += 256 * gridDim.y) {
    float s = 0.f;
    for (int o = 0; o < dim; ++o) s = fmaf(sd[o], w1[(long long)o * C + c], s);
    for (int o = 0; o < dim; ++o) s = fmaf(sd[dim + o], w2[(long long)o * C + c], s);
    s *= invP;
    float* f = dfeat + (long long)b * P * ld + c;
    for (int p = 0; p < P; ++p) f[(long long)p * ld] = s;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// logits
// ------------------------------------------------------------------------------------------------------------------
// l_neg = qA @ queue on the matrix pipe: wave = 32 queue columns x 32 query rows, K = dim (<= 256, even).
__global__ __launch_bounds__(256) void logits_neg_kernel(const float* __restrict__ qA, const float* __restrict__ queue,
                                                         int B, int dim, int K, float inv_T, float* __restrict__ logits1,
                                                         float* __restrict__ logits2) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l32 = lane & 31, h = lane >> 5;
  const int col0 = (blockIdx.x * 4 + wave) * 32;
  if (col0 >= K) return;
  const int col = col0 + l32;
  const bool cok = col < K;
  const int K1 = K + 1;
  for (int rb = 0; rb < B; rb += 32) {
    const int row = rb + l32;
    const bool rok = row < B;
    floatx16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int s = 0; s < dim / 2; ++s) {
      const int k = 2 * s + h;
      const float a = rok ? qA[(long long)row * dim + k] : 0.f;
      const float b = cok ? queue[(long long)k * K + col] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    if (cok) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int r = rb + (e >> 2) * 8 + h * 4 + (e & 3);
        if (r < B) {
          const float v = acc[e] * inv_T;
          logits1[(long long)r * K1 + 1 + col] = v;
          logits2[(long long)r * K1 + 1 + col] = v;
        }
      }
    }
  }
}

// the four row-wise dots (l_pos_A1, l_pos_A2, l_pos_M, l_neg_M), one wave each
__global__ __launch_bounds__(256) void logits_pos_kernel(const float* __restrict__ qA, const float* __restrict__ qM,
                                                         const float* __restrict__ kA, const float* __restrict__ kM,
                                                         const float* __restrict__ knegA, const float* __restrict__ knegM,
                                                         int dimA, int dimM, int K1, float inv_T, float* __restrict__ logits1,
                                                         float* __restrict__ logits2, float* __restrict__ lposM,
                                                         float* __restrict__ lnegM) {
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int dim = wave < 2 ? dimA : dimM;
  const float* q = (wave < 2 ? qA : qM) + (long long)b * dim;
  const float* k = (wave == 0 ? kA : wave == 1 ? knegA : wave == 2 ? kM : knegM) + (long long)b * dim;
  float s = 0.f;
  for (int c = lane; c < dim; c += 64) s = fmaf(q[c], k[c], s);
  s = rsp_wave_sum(s) * inv_T;
  if (lane == 0) {
    if (wave == 0) logits1[(long long)b * K1] = s;
    else if (wave == 1) logits2[(long long)b * K1] = s;
    else if (wave == 2) lposM[b] = s;
    else lnegM[b] = s;
  }
}

// backward, negatives: partial[slice][b][c] = sum_{k in slice} (g1+g2)[b][1+k] * queue[c][k]
constexpr int LB_KS = 128;  // K-slice per block
__global__ __launch_bounds__(256) void logits_bwd_neg_kernel(const float* __restrict__ g1, const float* __restrict__ g2,
                                                             const float* __restrict__ queue, int B, int dim, int K,
                                                             float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Gs = sm;                  // [32][LB_KS]
  float* Qs = sm + 32 * LB_KS;     // [dim][LB_KS+1]
  const int t = threadIdx.x;
  const int k0 = blockIdx.x * LB_KS;
  const int rb = blockIdx.y * 32;
  const int K1 = K + 1;
  for (int i = t; i < 32 * LB_KS; i += 256) {
    const int b = i / LB_KS, kk = i - b * LB_KS;
    float v = 0.f;
    if (rb + b < B && k0 + kk < K) {
      const long long o = (long long)(rb + b) * K1 + 1 + k0 + kk;
      v = g1[o] + g2[o];
    }
    Gs[i] = v;
  }
  for (int i = t; i < dim * LB_KS; i += 256) {
    const int c = i / LB_KS, kk = i - c * LB_KS;
    Qs[c * (LB_KS + 1) + kk] = (k0 + kk < K) ? queue[(long long)c * K + k0 + kk] : 0.f;
  }
  __syncthreads();
  // thread -> (c, half of the 32 rows)
  for (int c = t % 128; c < dim; c += 128) {
    const int bh = t / 128;
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    for (int kk = 0; kk < LB_KS; ++kk) {
      const float q = Qs[c * (LB_KS + 1) + kk];
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] = fmaf(Gs[(bh * 16 + j) * LB_KS + kk], q, acc[j]);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int b = rb + bh * 16 + j;
      if (b < B) partial[((long long)blockIdx.x * B + b) * dim + c] = acc[j];
    }
  }
}

__global__ void logits_bwd_final_kernel(const float* __restrict__ partial, int nslices, const float* __restrict__ g1,
                                        const float* __restrict__ g2, const float* __restrict__ gp,
                                        const float* __restrict__ gn, const float* __restrict__ kA,
                                        const float* __restrict__ kM, const float* __restrict__ knegA,
                                        const float* __restrict__ knegM, int B, int dim, int dimM, int K1, float inv_T,
                                        float* __restrict__ dqA, float* __restrict__ dqM) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * dimM) {
    const int b = i / dimM;
    dqM[i] = (gp[b] * kM[i] + gn[b] * knegM[i]) * inv_T;
  }
  if (i >= B * dim) return;
  const int b = i / dim;
  float s = 0.f;
  for (int z = 0; z < nslices; ++z) s += partial[(long long)z * B * dim + i];
  s = fmaf(g1[(long long)b * K1], kA[i], s);
  s = fmaf(g2[(long long)b * K1], knegA[i], s);
  dqA[i] = s * inv_T;
}

// ------------------------------------------------------------------------------------------------------------------
// loss
// ------------------------------------------------------------------------------------------------------------------
// block = one row of logits1 (blockIdx.y = 0) or logits2 (1): CE with target 0 and its gradient scaled by gscale.
__global__ __launch_bounds__(256) void ce_row_kernel(const float* __restrict__ l1, const float* __restrict__ l2, int K1,
                                                     float gscale, float* __restrict__ d1, float* __restrict__ d2,
                                                     float* __restrict__ row_loss, int B) {
  __shared__ float red[4];
  __shared__ float bc;
  const int b = blockIdx.x, which = blockIdx.y, t = threadIdx.x;
  const float* z = (which ? l2 : l1) + (long long)b * K1;
  float* d = (which ? d2 : d1) + (long long)b * K1;
  float m = -INFINITY;
  for (int j = t; j < K1; j += 256) m = fmaxf(m, z[j]);
  m = rsp_wave_max(m);
  if ((t & 63) == 0) red[t >> 6] = m;
  __syncthreads();
  if (t == 0) bc = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  m = bc;
  float s = 0.f;
  for (int j = t; j < K1; j += 256) s += expf(z[j] - m);
  s = rsp_wave_sum(s);
  __syncthreads();
  if ((t & 63) == 0) red[t >> 6] = s;
  __syncthreads();
  if (t == 0) bc = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  const float lse = m + logf(bc);
  if (t == 0) row_loss[which * B + b] = lse - z[0];
  for (int j = t; j < K1; j += 256) d[j] = (expf(z[j] - lse) - (j == 0 ? 1.f : 0.f)) * gscale;
}

__global__ void loss_final_kernel(const float* __restrict__ row_loss, const float* __restrict__ lp,
                                  const float* __restrict__ ln, int B, float margin, float A, float M,
                                  float* __restrict__ losses, float* __restrict__ dlp, float* __restrict__ dln) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  double ce1 = 0.0, ce2 = 0.0, rk = 0.0;
  for (int b = 0; b < B; ++b) {
    ce1 += (double)row_loss[b];
    ce2 += (double)row_loss[B + b];
    const float v = margin - (lp[b] - ln[b]);
    const bool act = v >= 0.f;  // clamp_min backward passes grad where input >= min
    rk += act ? (double)v : 0.0;
    dlp[b] = act ? -M / (float)B : 0.f;
    dln[b] = act ? M / (float)B : 0.f;
  }
  const float ce = (float)(ce1 / B) + (float)(ce2 / B);
  const float r = (float)(rk / B);
  losses[0] = A * ce + M * r;
  losses[1] = ce;
  losses[2] = r;
}

__global__ void enqueue_kernel(float* __restrict__ queue, int dim, int K, int ptr, const float* __restrict__ keys, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * dim) return;
  const int c = i / n, j = i - c * n;
  queue[(long long)c * K + ptr + j] = keys[(long long)j * dim + c];
}

// The same with the write pointer read from (and advanced in) device memory — queue_ptr is a device buffer of the module
// (builder_diffspeed_diffloss.py:332) — so that a step captured in a HIP graph needs no host-side pointer.  Two launches: every
// workgroup of the first reads the pointer, the second moves it.
__global__ void enqueue_dev_kernel(float* __restrict__ queue, int dim, int K, const long long* __restrict__ ptr_dev,
                                   const float* __restrict__ keys, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * dim) return;
  const int ptr = (int)*ptr_dev;
  const int c = i / n, j = i - c * n;
  if (ptr >= 0 && ptr + n <= K) queue[(long long)c * K + ptr + j] = keys[(long long)j * dim + c];
}
__global__ void enqueue_advance_kernel(long long* __restrict__ ptr_dev, int K, int n) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *ptr_dev = (*ptr_dev + n) % K;
}

// ------------------------------------------------------------------------------------------------------------------
// generic small pieces for the 'mlp' projection head (split_wrapper.py:171-179): mean -> Linear -> ReLU -> Linear -> L2
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void smean_kernel(const float* __restrict__ x, int P, int C, int ld, float* __restrict__ mean) {
  __shared__ float red[4][64];
  const int n = blockIdx.x, c = blockIdx.y * 64 + (threadIdx.x & 63), pl = threadIdx.x >> 6;
  float s = 0.f;
  if (c < C)
    for (int pp = pl; pp < P; pp += 4) s += x[((long long)n * P + pp) * ld + c];
  red[pl][threadIdx.x & 63] = s;
  __syncthreads();
  if (pl == 0 && c < C) {
    const int l = threadIdx.x;
    mean[(long long)n * C + c] = ((red[0][l] + red[1][l]) + (red[2][l] + red[3][l])) / (float)P;
  }
}
__global__ void smean_bwd_kernel(const float* __restrict__ dmean, int P, int C, int ld, long long total, float* __restrict__ dx) {
  const float invP = 1.f / (float)P;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long row = i / C;
    dx[row * ld + c] = dmean[(row / P) * C + c] * invP;
  }
}
// y[b][o] = act(b[o] + sum_c w[o][c] x[b][c]); one wave per output
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, int B, int Cin, int Cout, int relu,
                                                         float* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const long long o = blockIdx.x * 4ll + (threadIdx.x >> 6);
  if (o >= (long long)B * Cout) return;
  const int b = (int)(o / Cout), co = (int)(o % Cout);
  float s = 0.f;
  for (int c = lane; c < Cin; c += 64) s = fmaf(w[(long long)co * Cin + c], x[(long long)b * Cin + c], s);
  s = rsp_wave_sum(s);
  if (lane == 0) {
    s += bias ? bias[co] : 0.f;
    y[o] = relu ? fmaxf(s, 0.f) : s;
  }
}
__global__ void linear_bwd_dz_kernel(const float* __restrict__ y, const float* __restrict__ dy, long long n, int relu,
                                     float* __restrict__ dz) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i < n) dz[i] = (relu && !(y[i] > 0.f)) ? 0.f : dy[i];
}
__global__ void linear_bwd_w_kernel(const float* __restrict__ dz, const float* __restrict__ x, int B, int Cin, int Cout,
                                    float* __restrict__ dw, float* __restrict__ db) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= (long long)Cout * Cin) return;
  const int o = (int)(i / Cin), c = (int)(i % Cin);
  float s = 0.f, sb = 0.f;
  for (int b = 0; b < B; ++b) {
    const float d = dz[(long long)b * Cout + o];
    s = fmaf(d, x[(long long)b * Cin + c], s);
    sb += d;
  }
  dw[i] = s;
  if (c == 0 && db) db[o] = sb;
}
__global__ void linear_bwd_x_kernel(const float* __restrict__ dz, const float* __restrict__ w, int B, int Cin, int Cout,
                                    float* __restrict__ dx) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= (long long)B * Cin) return;
  const int b = (int)(i / Cin), c = (int)(i % Cin);
  float s = 0.f;
  for (int o = 0; o < Cout; ++o) s = fmaf(dz[(long long)b * Cout + o], w[(long long)o * Cin + c], s);
  dx[i] = s;
}
// y = x / max(|x|, 1e-12) per row; one wave per row
__global__ __launch_bounds__(64) void l2norm_fwd_kernel(const float* __restrict__ x, int dim, float* __restrict__ y) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float ss = 0.f;
  for (int o = lane; o < dim; o += 64) ss = fmaf(x[(long long)b * dim + o], x[(long long)b * dim + o], ss);
  ss = rsp_wave_sum(ss);
  const float nrm = fmaxf(sqrtf(ss), 1e-12f);
  for (int o = lane; o < dim; o += 64) y[(long long)b * dim + o] = x[(long long)b * dim + o] / nrm;
}
__global__ __launch_bounds__(64) void l2norm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, int dim,
                                                        float* __restrict__ dx) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* r = x + (long long)b * dim;
  const float* g = dy + (long long)b * dim;
  float ss = 0.f, dot = 0.f;
  for (int o = lane; o < dim; o += 64) {
    ss = fmaf(r[o], r[o], ss);
    dot = fmaf(r[o], g[o], dot);
  }
  ss = rsp_wave_sum(ss);
  dot = rsp_wave_sum(dot);
  const float nrm = sqrtf(ss);
  float* d = dx + (long long)b * dim;
  if (nrm > 1e-12f) {
    const float inv = 1.f / nrm, k = dot * inv * inv;
    for (int o = lane; o < dim; o += 64) d[o] = (g[o] - r[o] * k) * inv;
  } else {
    for (int o = lane; o < dim; o += 64) d[o] = g[o] * 1e12f;
  }
}

}  // namespace

extern "C" {

int rsp_head_fwd(const float* feat, int32_t B, int32_t P, int32_t C, int32_t feat_ld, const float* w1,
                 const float* b1, const float* w2, const float* b2, int32_t dim, float* out1, float* out2,
                 float* pooled, float* raw, void* stream) {
  RSP_REQUIRE(feat && w1 && b1 && w2 && b2 && out1 && out2 && pooled && raw, "rsp_head_fwd: null pointer");
  RSP_REQUIRE(B > 0 && P > 0 && C > 0 && dim > 0 && feat_ld >= C,
              "rsp_head_fwd: bad size");
  hipStream_t s = (hipStream_t)stream;
  // pool with B x C/64 workgroups, then the stacked 2*dim x C mat-vec on B x 2*dim/32 workgroups, then the two l2-norms
  hipLaunchKernelGGL(smean_kernel, dim3(B, rsp_cdiv(C, 64)), dim3(256), 0, s, feat, P, C, feat_ld, pooled);
  int rc = rsp_check_launch("smean_kernel");
  if (rc != RSP_OK) return rc;
  hipLaunchKernelGGL(head_raw_kernel, dim3(B, rsp_cdiv(2 * dim, 32)), dim3(256), 0, s, pooled, B, C, w1, b1, w2, b2, dim, raw);
  rc = rsp_check_launch("head_raw_kernel");
  if (rc != RSP_OK) return rc;
  hipLaunchKernelGGL(head_norm_kernel, dim3(B, 2), dim3(64), 0, s, raw, B, dim, out1, out2);
  return rsp_check_launch("head_norm_kernel");
}

size_t rsp_head_bwd_workspace(int32_t B, int32_t dim) { return (size_t)2 * B * dim * sizeof(float); }

int rsp_head_bwd(const float* dout1, const float* dout2, const float* pooled, const float* raw, const float* w1,
                 const float* w2, int32_t B, int32_t P, int32_t C, int32_t feat_ld, int32_t dim, float* dw1,
                 float* db1, float* dw2, float* db2, float* dfeat, void* workspace, size_t workspace_bytes,
                 void* stream) {
  RSP_REQUIRE(dout1 && dout2 && pooled && raw && w1 && w2 && dw1 && db1 && dw2 && db2 && dfeat && workspace,
              "rsp_head_bwd: null pointer");
  RSP_REQUIRE(B > 0 && P > 0 && C > 0 && dim > 0 && feat_ld >= C && dim <= 4096, "rsp_head_bwd: bad size");
  if (workspace_bytes < rsp_head_bwd_workspace(B, dim)) {
    rsp_set_error("rsp_head_bwd: workspace too small");
    return RSP_EWORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  float* draw = reinterpret_cast<float*>(workspace);
  hipLaunchKernelGGL(head_bwd_norm_kernel, dim3(B, 2), dim3(64), 0, s, dout1, dout2, raw, B, dim, draw);
  int rc = rsp_check_launch("head_bwd_norm_kernel");
  if (rc != RSP_OK) return rc;
  const long long total = 2ll * dim * C;
  hipLaunchKernelGGL(head_bwd_w_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, draw, pooled, B, C, dim,
                     dw1, db1, dw2, db2);
  rc = rsp_check_launch("head_bwd_w_kernel");
  if (rc != RSP_OK) return rc;
  hipLaunchKernelGGL(head_bwd_x_kernel, dim3(B, rsp_cdiv(C, 256)), dim3(256), (size_t)2 * dim * 4, s, draw, w1, w2, B, P, C,
                     feat_ld, dim, dfeat);
  return rsp_check_launch("head_bwd_x_kernel");
}

int rsp_logits_fwd(const float* qA, const float* qM, const float* kA, const float* kM, const float* knegA,
                   const float* knegM, const float* queue, int32_t B, int32_t dim, int32_t dim_m, int32_t K, float inv_T,
                   float* logits1, float* logits2, float* lposM, float* lnegM, void* stream) {
  RSP_REQUIRE(qA && qM && kA && kM && knegA && knegM && queue && logits1 && logits2 && lposM && lnegM,
              "rsp_logits_fwd: null pointer");
  RSP_REQUIRE(B > 0 && dim > 0 && dim % 2 == 0 && dim_m > 0 && K > 0, "rsp_logits_fwd: bad size (dim must be even)");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(logits_neg_kernel, dim3(rsp_cdiv(K, 128)), dim3(256), 0, s, qA, queue, B, dim, K, inv_T, logits1,
                     logits2);
  int rc = rsp_check_launch("logits_neg_kernel");
  if (rc != RSP_OK) return rc;
  hipLaunchKernelGGL(logits_pos_kernel, dim3(B), dim3(256), 0, s, qA, qM, kA, kM, knegA, knegM, dim, dim_m, K + 1, inv_T, logits1,
                     logits2, lposM, lnegM);
  return rsp_check_launch("logits_pos_kernel");
}

size_t rsp_logits_bwd_workspace(int32_t B, int32_t dim, int32_t K) {
  return (size_t)rsp_cdiv(K, LB_KS) * B * dim * sizeof(float);
}

int rsp_logits_bwd(const float* dlogits1, const float* dlogits2, const float* dlposM, const float* dlnegM,
                   const float* kA, const float* kM, const float* knegA, const float* knegM, const float* queue,
                   int32_t B, int32_t dim, int32_t dim_m, int32_t K, float inv_T, float* dqA, float* dqM, void* workspace,
                   size_t workspace_bytes, void* stream) {
  RSP_REQUIRE(dlogits1 && dlogits2 && dlposM && dlnegM && kA && kM && knegA && knegM && queue && dqA && dqM && workspace,
              "rsp_logits_bwd: null pointer");
  RSP_REQUIRE(B > 0 && dim > 0 && dim_m > 0 && dim_m <= dim && K > 0, "rsp_logits_bwd: bad size");
  if (workspace_bytes < rsp_logits_bwd_workspace(B, dim, K)) {
    rsp_set_error("rsp_logits_bwd: workspace too small");
    return RSP_EWORKSPACE;
  }
  const size_t lds = (size_t)(32 * LB_KS + dim * (LB_KS + 1)) * sizeof(float);
  RSP_REQUIRE(lds <= 150 * 1024, "rsp_logits_bwd: dim too large");
  hipStream_t s = (hipStream_t)stream;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&logits_bwd_neg_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr_set = true;
  }
  float* partial = reinterpret_cast<float*>(workspace);
  const int nslices = rsp_cdiv(K, LB_KS);
  hipLaunchKernelGGL(logits_bwd_neg_kernel, dim3(nslices, rsp_cdiv(B, 32)), dim3(256), lds, s, dlogits1, dlogits2, queue, B,
                     dim, K, partial);
  int rc = rsp_check_launch("logits_bwd_neg_kernel");
  if (rc != RSP_OK) return rc;
  hipLaunchKernelGGL(logits_bwd_final_kernel, dim3(rsp_cdiv((long long)B * dim, 256)), dim3(256), 0, s, partial, nslices,
                     dlogits1, dlogits2, dlposM, dlnegM, kA, kM, knegA, knegM, B, dim, dim_m, K + 1, inv_T, dqA, dqM);
  return rsp_check_launch("logits_bwd_final_kernel");
}

int rsp_loss_fwd_bwd(const float* logits1, const float* logits2, const float* lposM, const float* lnegM, int32_t B,
                     int32_t K1, float margin, float A, float M, float* losses, float* dlogits1, float* dlogits2,
                     float* dlposM, float* dlnegM, float* row_scratch, void* stream) {
  RSP_REQUIRE(logits1 && logits2 && lposM && lnegM && losses && dlogits1 && dlogits2 && dlposM && dlnegM && row_scratch,
              "rsp_loss_fwd_bwd: null pointer");
  RSP_REQUIRE(B > 0 && K1 > 0, "rsp_loss_fwd_bwd: bad size");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(ce_row_kernel, dim3(B, 2), dim3(256), 0, s, logits1, logits2, K1, A / (float)B, dlogits1, dlogits2,
                     row_scratch, B);
  int rc = rsp_check_launch("ce_row_kernel");
  if (rc != RSP_OK) return rc;
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, s, row_scratch, lposM, lnegM, B, margin, A, M, losses, dlposM,
                     dlnegM);
  return rsp_check_launch("loss_final_kernel");
}

int rsp_queue_enqueue(float* queue, int32_t dim, int32_t K, int32_t ptr, const float* keys, int32_t n, void* stream) {
  RSP_REQUIRE(queue && keys, "rsp_queue_enqueue: null pointer");
  RSP_REQUIRE(dim > 0 && K > 0 && n > 0 && ptr >= 0 && ptr + n <= K, "rsp_queue_enqueue: slab out of range");
  hipLaunchKernelGGL(enqueue_kernel, dim3(rsp_cdiv((long long)n * dim, 256)), dim3(256), 0, (hipStream_t)stream, queue, dim, K,
                     ptr, keys, n);
  return rsp_check_launch("enqueue_kernel");
}

int rsp_queue_enqueue_dev(float* queue, int32_t dim, int32_t K, int64_t* ptr_dev, const float* keys, int32_t n, void* stream) {
  RSP_REQUIRE(queue && keys && ptr_dev, "rsp_queue_enqueue_dev: null pointer");
  RSP_REQUIRE(dim > 0 && K > 0 && n > 0 && n <= K && K % n == 0, "rsp_queue_enqueue_dev: K must be a multiple of the batch");
  hipLaunchKernelGGL(enqueue_dev_kernel, dim3(rsp_cdiv((long long)n * dim, 256)), dim3(256), 0, (hipStream_t)stream, queue, dim, K,
                     reinterpret_cast<const long long*>(ptr_dev), keys, n);
  int rc = rsp_check_launch("enqueue_dev_kernel");
  if (rc != RSP_OK) return rc;
  hipLaunchKernelGGL(enqueue_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, reinterpret_cast<long long*>(ptr_dev), K, n);
  return rsp_check_launch("enqueue_advance_kernel");
}

int rsp_spatial_mean_fwd(const float* x, int32_t N, int32_t P, int32_t C, int32_t ld, float* mean, void* stream) {
  RSP_REQUIRE(x && mean && N > 0 && P > 0 && C > 0 && ld >= C, "rsp_spatial_mean_fwd: bad argument");
  hipLaunchKernelGGL(smean_kernel, dim3(N, rsp_cdiv(C, 64)), dim3(256), 0, (hipStream_t)stream, x, P, C, ld, mean);
  return rsp_check_launch("smean_kernel");
}

int rsp_spatial_mean_bwd(const float* dmean, int32_t N, int32_t P, int32_t C, int32_t ld, float* dx, void* stream) {
  RSP_REQUIRE(dmean && dx && N > 0 && P > 0 && C > 0 && ld >= C, "rsp_spatial_mean_bwd: bad argument");
  const long long total = (long long)N * P * C;
  const long long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(smean_bwd_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, (hipStream_t)stream, dmean, P,
                     C, ld, total, dx);
  return rsp_check_launch("smean_bwd_kernel");
}

int rsp_linear_fwd(const float* x, int32_t B, int32_t Cin, const float* w, const float* bias, int32_t Cout, int relu,
                   float* y, void* stream) {
  RSP_REQUIRE(x && w && y && B > 0 && Cin > 0 && Cout > 0, "rsp_linear_fwd: bad argument");
  hipLaunchKernelGGL(linear_fwd_kernel, dim3(rsp_cdiv((long long)B * Cout, 4)), dim3(256), 0, (hipStream_t)stream, x, w, bias, B,
                     Cin, Cout, relu, y);
  return rsp_check_launch("linear_fwd_kernel");
}

size_t rsp_linear_bwd_workspace(int32_t B, int32_t Cout) { return (size_t)B * Cout * sizeof(float); }

int rsp_linear_bwd(const float* x, const float* y, const float* dy, const float* w, int32_t B, int32_t Cin, int32_t Cout,
                   int relu, float* dx, float* dw, float* db, void* workspace, size_t workspace_bytes, void* stream) {
  RSP_REQUIRE(x && y && dy && w && dw && workspace && B > 0 && Cin > 0 && Cout > 0, "rsp_linear_bwd: bad argument");
  if (workspace_bytes < rsp_linear_bwd_workspace(B, Cout)) {
    rsp_set_error("rsp_linear_bwd: workspace too small");
    return RSP_EWORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  float* dz = reinterpret_cast<float*>(workspace);
  const long long n = (long long)B * Cout;
  hipLaunchKernelGGL(linear_bwd_dz_kernel, dim3(rsp_cdiv(n, 256)), dim3(256), 0, s, y, dy, n, relu, dz);
  int rc = rsp_check_launch("linear_bwd_dz_kernel");
  if (rc != RSP_OK) return rc;
  hipLaunchKernelGGL(linear_bwd_w_kernel, dim3(rsp_cdiv((long long)Cout * Cin, 256)), dim3(256), 0, s, dz, x, B, Cin, Cout, dw, db);
  rc = rsp_check_launch("linear_bwd_w_kernel");
  if (rc != RSP_OK || !dx) return rc;
  hipLaunchKernelGGL(linear_bwd_x_kernel, dim3(rsp_cdiv((long long)B * Cin, 256)), dim3(256), 0, s, dz, w, B, Cin, Cout, dx);
  return rsp_check_launch("linear_bwd_x_kernel");
}

int rsp_l2norm_fwd(const float* x, int32_t B, int32_t dim, float* y, void* stream) {
  RSP_REQUIRE(x && y && B > 0 && dim > 0, "rsp_l2norm_fwd: bad argument");
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, x, dim, y);
  return rsp_check_launch("l2norm_fwd_kernel");
}

int rsp_l2norm_bwd(const float* x, const float* dy, int32_t B, int32_t dim, float* dx, void* stream) {
  RSP_REQUIRE(x && dy && dx && B > 0 && dim > 0, "rsp_l2norm_bwd: bad argument");
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, x, dy, dim, dx);
  return rsp_check_launch("l2norm_bwd_kernel");
}

}  // extern "C"
