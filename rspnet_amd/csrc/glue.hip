// Step glue: clip speed sub-sampling + layout change, momentum update, SGD, row gather.
// Reference call sites: moco/builder_diffspeed_diffloss.py:421-443 (_diff_speed), :384-387 (shuffle select),
// :337-343 (momentum), :389-406 (un-shuffle), pretrain.py:65-72,165 (torch.optim.SGD).  All HBM-bound streaming kernels.
#include "common.h"

namespace {

// out[j][t][h][w][c] = im[src[j]][c][t*step[j]][h][w]      (NCDHW in, NDHWC out)
__global__ __launch_bounds__(256) void clip_gather_kernel(const float* __restrict__ im, int C, int T_in, int H, int W,
                                                          const int* __restrict__ src, const int* __restrict__ step,
                                                          int B_out, int T_out, int C_out, float* __restrict__ out) {
  const long long hw = (long long)H * W;
  const long long total = (long long)B_out * T_out * hw;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += 256ll * gridDim.x) {
    const long long p = i % hw;
    const long long q = i / hw;
    const int t = (int)(q % T_out);
    const int j = (int)(q / T_out);
    const int b = src[j];
    const int tin = t * step[j];
    const float* s = im + (((long long)b * C) * T_in + tin) * hw + p;
    float* o = out + i * C_out;
    for (int c = 0; c < C; ++c) o[c] = s[(long long)c * T_in * hw];
    for (int c = C; c < C_out; ++c) o[c] = 0.f;   // zero channel padding (lets a Cin=3 stem use 16-byte gathers)
  }
}

// The layout every backbone's clips take: 3 real channels padded to 16-byte pixels, H*W a multiple of 4.  A workgroup owns 512
// consecutive float4 columns of one (clip, frame) row — the frame's source planes are fixed per workgroup (no per-element division)
// — and a thread turns two groups of FOUR consecutive pixels: six 16-byte plane loads issued up front, then eight 16-byte pixels
// (2 x 64 contiguous bytes) out; a wave writes 4 KB contiguous per group.  (The general kernel stores 4 bytes per lane at a 16-byte
// stride, four times: 3.4 TB/s of algorithmic bytes; r4 VERDICT item 7.)
__global__ __launch_bounds__(256) void clip_gather_rgb4_kernel(const float* __restrict__ im, int T_in, int hw4, int chunks, int cw,
                                                               const int* __restrict__ src, const int* __restrict__ step,
                                                               int T_out, float* __restrict__ out) {
  const long long plane4 = (long long)T_in * hw4;      // one channel of one clip, in float4 units
  const int row = blockIdx.x / chunks, chunk = blockIdx.x - row * chunks;
  const int j = row / T_out, t = row - j * T_out;
  const floatx4* s = reinterpret_cast<const floatx4*>(im) + ((long long)src[j] * 3 * T_in + (long long)t * step[j]) * hw4;
  floatx4* o = reinterpret_cast<floatx4*>(out) + (long long)row * hw4 * 4;
  // (cw <= 512 columns per workgroup, equal shares of the row: 112 x 112 = 7 x 448, no nearly empty last workgroup)
  const int q0 = chunk * cw + threadIdx.x, q1 = q0 + 256, qe = min(hw4, (chunk + 1) * cw);
  floatx4 v[2][3];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int q = u ? q1 : q0;
    if (q < qe) {
#pragma unroll
      for (int c = 0; c < 3; ++c) v[u][c] = __builtin_nontemporal_load(s + c * plane4 + q);
    }
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int q = u ? q1 : q0;
    if (q < qe) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const floatx4 px = {v[u][0][e], v[u][1][e], v[u][2][e], 0.f};
        o[(long long)q * 4 + e] = px;
      }
    }
  }
}

// ... the same for up to four gathers of one geometry in one launch (blockIdx.y picks the job)
struct GatherJobs {
  const float* im[4];
  const int* src[4];
  const int* step[4];
  float* out[4];
};
__global__ __launch_bounds__(256) void clip_gather_rgb4_multi_kernel(const GatherJobs jobs, int T_in, int hw4, int chunks, int cw, int T_out) {
  const int jb = blockIdx.y;
  const float* __restrict__ im = jobs.im[jb];
  const int* __restrict__ src = jobs.src[jb];
  const int* __restrict__ step = jobs.step[jb];
  float* __restrict__ out = jobs.out[jb];
  const long long plane4 = (long long)T_in * hw4;
  const int row = blockIdx.x / chunks, chunk = blockIdx.x - row * chunks;
  const int j = row / T_out, t = row - j * T_out;
  const floatx4* s = reinterpret_cast<const floatx4*>(im) + ((long long)src[j] * 3 * T_in + (long long)t * step[j]) * hw4;
  floatx4* o = reinterpret_cast<floatx4*>(out) + (long long)row * hw4 * 4;
  // (cw <= 512 columns per workgroup, equal shares of the row: 112 x 112 = 7 x 448, no nearly empty last workgroup)
  const int q0 = chunk * cw + threadIdx.x, q1 = q0 + 256, qe = min(hw4, (chunk + 1) * cw);
  floatx4 v[2][3];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int q = u ? q1 : q0;
    if (q < qe) {
#pragma unroll
      for (int c = 0; c < 3; ++c) v[u][c] = __builtin_nontemporal_load(s + c * plane4 + q);
    }
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int q = u ? q1 : q0;
    if (q < qe) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const floatx4 px = {v[u][0][e], v[u][1][e], v[u][2][e], 0.f};
        o[(long long)q * 4 + e] = px;
      }
    }
  }
}

__global__ __launch_bounds__(256) void momentum_kernel(float* __restrict__ k, const float* __restrict__ q, long long n,
                                                       float m) {
  const float om = 1.f - m;
  const long long n4 = n >> 2;
  floatx4* k4 = reinterpret_cast<floatx4*>(k);
  const floatx4* q4 = reinterpret_cast<const floatx4*>(q);
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += 256ll * gridDim.x) {
    floatx4 a = k4[i];
    const floatx4 b = q4[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] = a[e] * m + b[e] * om;   // param_k*m + param_q*(1-m): two roundings, as the reference
    k4[i] = a;
  }
  for (long long i = (n4 << 2) + blockIdx.x * 256ll + threadIdx.x; i < n; i += 256ll * gridDim.x)
    k[i] = k[i] * m + q[i] * om;
}

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                  float* __restrict__ buf, long long n, float lr, float mu, float wd,
                                                  float gscale, int first) {
  const long long n4 = n >> 2;
  floatx4* p4 = reinterpret_cast<floatx4*>(p);
  const floatx4* g4 = reinterpret_cast<const floatx4*>(g);
  floatx4* b4 = reinterpret_cast<floatx4*>(buf);
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += 256ll * gridDim.x) {
    floatx4 pp = p4[i];
    const floatx4 gg = g4[i];
    floatx4 bb = first ? floatx4{0.f, 0.f, 0.f, 0.f} : b4[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = gg[e] * gscale + wd * pp[e];
      bb[e] = first ? d : bb[e] * mu + d;
      pp[e] = pp[e] - lr * bb[e];
    }
    b4[i] = bb;
    p4[i] = pp;
  }
  for (long long i = (n4 << 2) + blockIdx.x * 256ll + threadIdx.x; i < n; i += 256ll * gridDim.x) {
    const float d = g[i] * gscale + wd * p[i];
    const float b = first ? d : buf[i] * mu + d;
    buf[i] = b;
    p[i] = p[i] - lr * b;
  }
}

__global__ void rows_gather_kernel(const float* __restrict__ in, const int* __restrict__ idx, int n, int width,
                                   float* __restrict__ out) {
  const long long total = (long long)n * width;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i / width), c = (int)(i - (long long)j * width);
    out[i] = in[(long long)idx[j] * width + c];
  }
}

// Small element-wise pieces of the 'conv' / 'speednet' projection heads (moco/split_wrapper.py:18-39,146-149) and the
// gradient accumulation at fan-out points of the layer graph (residual / inception branches share an input).
// op: 0 relu fwd (y = max(a,0)), 1 relu bwd (y = a > 0 ? b : 0), 2 sigmoid fwd, 3 sigmoid bwd (y = b*a*(1-a), a = sigmoid out),
//     4 accumulate (y = a + b)
// `y` may alias `a` or `b` at identical indices (the engine accumulates gradients over the fresh operand, ReLU runs in place):
// no __restrict__ on any of the three — every thread reads index i of the inputs before it writes index i of the output.
template <int OP>
__global__ __launch_bounds__(256) void eltwise_kernel(const float* a, const float* b, float* y, long long n) {
  auto f = [](float u, float v) -> float {
    if (OP == 0) return fmaxf(u, 0.f);
    if (OP == 1) return u > 0.f ? v : 0.f;
    if (OP == 2) return 1.f / (1.f + expf(-u));
    if (OP == 3) return v * u * (1.f - u);
    return u + v;
  };
  const bool vec = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(y) | (b ? reinterpret_cast<uintptr_t>(b) : 0)) & 15) == 0;
  const long long n4 = vec ? n >> 2 : 0;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += 256ll * gridDim.x) {
    const floatx4 u = reinterpret_cast<const floatx4*>(a)[i];
    const floatx4 v = b ? reinterpret_cast<const floatx4*>(b)[i] : floatx4{0.f, 0.f, 0.f, 0.f};
    floatx4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = f(u[e], v[e]);
    reinterpret_cast<floatx4*>(y)[i] = o;
  }
  for (long long i = (n4 << 2) + blockIdx.x * 256ll + threadIdx.x; i < n; i += 256ll * gridDim.x) y[i] = f(a[i], b ? b[i] : 0.f);
}

int grid_for(long long total) {
  long long b = (total + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" {

int rsp_clip_gather(const float* im, int32_t B_in, int32_t C, int32_t T_in, int32_t H, int32_t W, const int32_t* src,
                    const int32_t* step, int32_t B_out, int32_t T_out, int32_t C_out, float* out, void* stream) {
  RSP_REQUIRE(im && src && step && out, "rsp_clip_gather: null pointer");
  RSP_REQUIRE(B_in > 0 && C > 0 && T_in > 0 && H > 0 && W > 0 && B_out > 0 && T_out > 0 && C_out >= C,
              "rsp_clip_gather: bad size");
  const long long total = (long long)B_out * T_out * H * W;
  if (C == 3 && C_out == 4 && ((long long)H * W) % 4 == 0 && rsp_aligned16(im) && rsp_aligned16(out)) {
    const int hw4 = (int)(((long long)H * W) / 4), chunks = (hw4 + 511) / 512, cw = ((hw4 + chunks - 1) / chunks + 63) / 64 * 64;
    const long long grid = (long long)B_out * T_out * chunks;
    if (grid < (1ll << 31)) {
      hipLaunchKernelGGL(clip_gather_rgb4_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, im, T_in, hw4, chunks, cw, src,
                         step, T_out, out);
      return rsp_check_launch("clip_gather_rgb4_kernel");
    }
  }
  hipLaunchKernelGGL(clip_gather_kernel, dim3(grid_for(total) * 2), dim3(256), 0, (hipStream_t)stream, im, C, T_in, H, W, src,
                     step, B_out, T_out, C_out, out);
  return rsp_check_launch("clip_gather_kernel");
}

int rsp_clip_gather_multi(int32_t n_jobs, const float* const* ims, const int32_t* const* srcs, const int32_t* const* steps,
                          float* const* outs, int32_t B_in, int32_t C, int32_t T_in, int32_t H, int32_t W, int32_t B_out, int32_t T_out,
                          int32_t C_out, void* stream) {
  RSP_REQUIRE(n_jobs >= 1 && n_jobs <= 4 && ims && srcs && steps && outs, "rsp_clip_gather_multi: bad argument");
  bool fast = C == 3 && C_out == 4 && ((long long)H * W) % 4 == 0;
  for (int i = 0; i < n_jobs; ++i) {
    RSP_REQUIRE(ims[i] && srcs[i] && steps[i] && outs[i], "rsp_clip_gather_multi: null pointer");
    fast = fast && rsp_aligned16(ims[i]) && rsp_aligned16(outs[i]);
  }
  const int hw4 = (int)(((long long)H * W) / 4), chunks = (hw4 + 511) / 512, cw = ((hw4 + chunks - 1) / chunks + 63) / 64 * 64;
  const long long grid = (long long)B_out * T_out * chunks;
  if (fast && grid < (1ll << 31) && B_in > 0 && T_in > 0 && B_out > 0 && T_out > 0) {
    GatherJobs j;
    memset(&j, 0, sizeof j);
    for (int i = 0; i < n_jobs; ++i) {
      j.im[i] = ims[i]; j.src[i] = srcs[i]; j.step[i] = steps[i]; j.out[i] = outs[i];
    }
    hipLaunchKernelGGL(clip_gather_rgb4_multi_kernel, dim3((unsigned)grid, (unsigned)n_jobs), dim3(256), 0, (hipStream_t)stream, j, T_in,
                       hw4, chunks, cw, T_out);
    return rsp_check_launch("clip_gather_rgb4_multi_kernel");
  }
  for (int i = 0; i < n_jobs; ++i) {
    const int rc = rsp_clip_gather(ims[i], B_in, C, T_in, H, W, srcs[i], steps[i], B_out, T_out, C_out, outs[i], stream);
    if (rc != RSP_OK) return rc;
  }
  return RSP_OK;
}

int rsp_momentum_update(float* k, const float* q, int64_t n, float m, void* stream) {
  RSP_REQUIRE(k && q && n > 0, "rsp_momentum_update: bad argument");
  RSP_REQUIRE(rsp_aligned16(k) && rsp_aligned16(q), "rsp_momentum_update: buffers must be 16-byte aligned");
  hipLaunchKernelGGL(momentum_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, k, q, (long long)n, m);
  return rsp_check_launch("momentum_kernel");
}

int rsp_sgd_step(float* p, const float* g, float* buf, int64_t n, float lr, float mu, float wd, float gscale, int first,
                 void* stream) {
  RSP_REQUIRE(p && g && buf && n > 0, "rsp_sgd_step: bad argument");
  RSP_REQUIRE(rsp_aligned16(p) && rsp_aligned16(g) && rsp_aligned16(buf), "rsp_sgd_step: buffers must be 16-byte aligned");
  hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, p, g, buf, (long long)n, lr, mu,
                     wd, gscale, first);
  return rsp_check_launch("sgd_kernel");
}

int rsp_eltwise(int32_t op, const float* a, const float* b, float* y, int64_t n, void* stream) {
  RSP_REQUIRE(a && y && n > 0, "rsp_eltwise: bad argument");
  RSP_REQUIRE(op >= 0 && op <= 4, "rsp_eltwise: unknown op");
  RSP_REQUIRE(b || op == 0 || op == 2, "rsp_eltwise: this op needs a second operand");
  const dim3 g(grid_for(n / 4 + 1)), t(256);
  hipStream_t s = (hipStream_t)stream;
  switch (op) {
    case 0: hipLaunchKernelGGL(eltwise_kernel<0>, g, t, 0, s, a, b, y, (long long)n); break;
    case 1: hipLaunchKernelGGL(eltwise_kernel<1>, g, t, 0, s, a, b, y, (long long)n); break;
    case 2: hipLaunchKernelGGL(eltwise_kernel<2>, g, t, 0, s, a, b, y, (long long)n); break;
    case 3: hipLaunchKernelGGL(eltwise_kernel<3>, g, t, 0, s, a, b, y, (long long)n); break;
    default: hipLaunchKernelGGL(eltwise_kernel<4>, g, t, 0, s, a, b, y, (long long)n); break;
  }
  return rsp_check_launch("eltwise_kernel");
}

int rsp_rows_gather(const float* in, const int32_t* idx, int32_t n, int32_t width, float* out, void* stream) {
  RSP_REQUIRE(in && idx && out && n > 0 && width > 0, "rsp_rows_gather: bad argument");
  hipLaunchKernelGGL(rows_gather_kernel, dim3(grid_for((long long)n * width)), dim3(256), 0, (hipStream_t)stream, in, idx, n,
                     width, out);
  return rsp_check_launch("rows_gather_kernel");
}

}  // extern "C"
