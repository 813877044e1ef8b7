// Fused per-clip augmentation: uint8 (T,h,w,3) crop -> float32 (3,T,S,S) model input in one pass.
// Reference: datasets/classification/__init__.py:189-202 applied clip by clip (transforms_tensor.py:207-233) as ~25 small
// ATen launches per clip; here the whole batch is two launches (+ a tiny one) regardless of the number of clips.
//
// HBM-bound streaming kernel: per clip it reads the crop (<= h*w*3*T bytes, twice when a contrast op needs the mean of
// the intermediate image) and writes 3*T*S*S*4 bytes.  The arithmetic follows the reference's evaluation order operation by
// operation (fp contraction is disabled for this file), so every step but the contrast mean (a sum whose order differs) is
// bit-identical to the CPU restatement.
#include "common.h"

#pragma clang fp contract(off)

namespace {

constexpr int PIX_PER_BLOCK = 1024;   // 256 threads x 4 pixels

struct Rgb {
  float r, g, b;
};

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }
__device__ __forceinline__ float gray_of(const Rgb& p) { return (0.2989f * p.r + 0.5870f * p.g) + 0.1140f * p.b; }

// torch.remainder(a, 1.0) = fmod(a, 1) (+1 when negative).  fmod(a, 1) = a - trunc(a) is exact, so both forms round the same real
// number a - floor(a) exactly once: a - floorf(a) is bit-identical and needs no fmod sequence.
__device__ __forceinline__ float mod1(float a) { return a - floorf(a); }

// ToTensorVideo + Resize: bilinear sample of output pixel (y, x) of frame t (upsample_bilinear2d, align_corners=False)
__device__ __forceinline__ Rgb sample(const rsp_augment_clip_desc& d, const float* __restrict__ lut, int t, int y, int x, int S) {
  const float sh = (float)d.h / (float)S, sw = (float)d.w / (float)S;
  float sy = sh * ((float)y + 0.5f) - 0.5f;
  if (sy < 0.f) sy = 0.f;
  float sx = sw * ((float)x + 0.5f) - 0.5f;
  if (sx < 0.f) sx = 0.f;
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = y0 + (y0 < d.h - 1 ? 1 : 0), x1 = x0 + (x0 < d.w - 1 ? 1 : 0);
  const float ly1 = sy - (float)y0, lx1 = sx - (float)x0;
  const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
  const uint8_t* f = d.src + (long long)t * d.frame_pitch;
  const uint8_t* r0 = f + (long long)y0 * d.row_pitch;
  const uint8_t* r1 = f + (long long)y1 * d.row_pitch;
  float v[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v00 = lut[r0[x0 * 3 + c]], v01 = lut[r0[x1 * 3 + c]];   // lut[u] = (float)u / 255.0f (ToTensorVideo)
    const float v10 = lut[r1[x0 * 3 + c]], v11 = lut[r1[x1 * 3 + c]];
    v[c] = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
  }
  return Rgb{v[0], v[1], v[2]};
}

__device__ __forceinline__ Rgb hue_shift(Rgb p, float shift) {   // functional_tensor.py:376-417 (rgb_to_hsv :304, hsv_to_rgb :254)
  const float maxc = fmaxf(p.r, fmaxf(p.g, p.b)), minc = fminf(p.r, fminf(p.g, p.b));
  const int idx = (p.r >= p.g && p.r >= p.b) ? 0 : (p.g >= p.b ? 1 : 2);   // first maximum, as torch.max returns
  const float delta = maxc - minc;
  const float s = maxc == 0.f ? 0.f : delta / maxc;
  float h;
  if (delta == 0.f) h = 0.f;
  else if (idx == 0) h = (p.g - p.b) / delta;
  else if (idx == 1) h = (p.b - p.r) / delta + 2.0f;
  else h = (p.r - p.g) / delta + 4.0f;
  h = mod1(h / 6.0f);
  h = mod1(h + shift);
  const float h6 = h * 6.f;
  const float hi = floorf(h6);
  const float f = h6 - hi;
  const float v = maxc;
  const float tt = v * (1.f - (1.f - f) * s), pp = v * (1.f - s), qq = v * (1.f - f * s);
  switch (((int)hi) % 6) {
    case 0: return Rgb{v, tt, pp};
    case 1: return Rgb{qq, v, pp};
    case 2: return Rgb{pp, v, tt};
    case 3: return Rgb{pp, qq, v};
    case 4: return Rgb{tt, pp, v};
    default: return Rgb{v, pp, qq};
  }
}

// ops [first, last) of the clip's colour program; `mean` is the contrast mean (only read by a contrast op)
__device__ __forceinline__ Rgb run_ops(const rsp_augment_clip_desc& d, Rgb p, int first, int last, float mean) {
  for (int i = first; i < last; ++i) {
    const float r = d.factor[i], q = d.one_minus[i];
    switch (d.op[i]) {
      case RSP_AUG_BRIGHTNESS:   // _blend(img, zeros, ratio)
        p = Rgb{clamp01(r * p.r + q * 0.f), clamp01(r * p.g + q * 0.f), clamp01(r * p.b + q * 0.f)};
        break;
      case RSP_AUG_CONTRAST: {   // _blend(img, mean(gray(img)), ratio)
        const float m = q * mean;
        p = Rgb{clamp01(r * p.r + m), clamp01(r * p.g + m), clamp01(r * p.b + m)};
        break;
      }
      case RSP_AUG_SATURATION: {   // _blend(img, gray(img), ratio)
        const float g = q * gray_of(p);
        p = Rgb{clamp01(r * p.r + g), clamp01(r * p.g + g), clamp01(r * p.b + g)};
        break;
      }
      default:
        p = hue_shift(p, r);
    }
  }
  return p;
}

__device__ __forceinline__ int contrast_index(const rsp_augment_clip_desc& d) {
  for (int i = 0; i < d.n_ops; ++i)
    if (d.op[i] == RSP_AUG_CONTRAST) return i;
  return -1;
}

struct Blur9 {
  float k[9];
};

// the per-pixel chain up to (not including) blur / flip / normalise
__device__ __forceinline__ Rgb pixel(const rsp_augment_clip_desc& d, const float* __restrict__ lut, int t, int y, int x, int S,
                                     float mean) {
  Rgb v = sample(d, lut, t, y, x, S);
  if ((d.gray & 3) == 1) {
    const float g = gray_of(v);
    v = Rgb{g, g, g};
  }
  v = run_ops(d, v, 0, d.n_ops, mean);
  if ((d.gray & 3) == 2) {
    const float g = gray_of(v);
    v = Rgb{g, g, g};
  }
  return v;
}

// pass 1 (clips with a contrast op): sum of gray(intermediate image just before the contrast op) per block
__global__ __launch_bounds__(256) void augment_mean_kernel(const rsp_augment_clip_desc* __restrict__ descs, int T, int S,
                                                           float* __restrict__ partial, int nblk) {
  __shared__ float red[4];
  __shared__ float lut[256];
  const rsp_augment_clip_desc d = descs[blockIdx.y];
  const int ci = contrast_index(d);
  if (ci < 0) return;
  lut[threadIdx.x] = (float)threadIdx.x / 255.0f;
  __syncthreads();
  const int npix = T * S * S;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int p = blockIdx.x * PIX_PER_BLOCK + k * 256 + threadIdx.x;
    if (p < npix) {
      const int x = p % S, q = p / S;
      const int y = q % S, t = q / S;
      Rgb v = sample(d, lut, t, y, x, S);   // the mean is flip-invariant: no need to mirror here
      if ((d.gray & 3) == 1) {
        const float g = gray_of(v);
        v = Rgb{g, g, g};
      }
      v = run_ops(d, v, 0, ci, 0.f);
      s += gray_of(v);
    }
  }
  s = rsp_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[(long long)blockIdx.y * nblk + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void augment_mean_final_kernel(const float* __restrict__ partial, int nblk, int npix,
                                                                float* __restrict__ mean) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 256) s += (double)partial[(long long)blockIdx.x * nblk + i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) mean[blockIdx.x] = (float)(red[0] / (double)npix);
}

// pass 2: the whole pipeline, written as three (T,S,S) planes per clip
__global__ __launch_bounds__(256) void augment_apply_kernel(const rsp_augment_clip_desc* __restrict__ descs, int T, int S,
                                                            const float* __restrict__ mean, float m0, float m1, float m2,
                                                            float s0, float s1, float s2, const Blur9 blur,
                                                            float* __restrict__ out, long long clip_stride) {
  __shared__ float lut[256];
  lut[threadIdx.x] = (float)threadIdx.x / 255.0f;
  __syncthreads();
  const rsp_augment_clip_desc d = descs[blockIdx.y];
  const int npix = T * S * S;
  const float cm = mean[blockIdx.y];
  float* o = out + (long long)blockIdx.y * clip_stride;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int p = blockIdx.x * PIX_PER_BLOCK + k * 256 + threadIdx.x;
    if (p < npix) {
      const int x = p % S, q = p / S;
      const int y = q % S, t = q / S;
      const int xs = d.flip ? S - 1 - x : x;
      Rgb v;
      if (d.gray & 4) {   // GaussianBlur: conv2d(3x3, zero padding) of the intermediate image = the pipeline at the 9 neighbours
        v = Rgb{0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const int yy = y + i - 1, xx = xs + j - 1;
            if ((unsigned)yy < (unsigned)S && (unsigned)xx < (unsigned)S) {
              const Rgb n = pixel(d, lut, t, yy, xx, S, cm);
              const float k = blur.k[i * 3 + j];
              v = Rgb{v.r + k * n.r, v.g + k * n.g, v.b + k * n.b};
            }
          }
      } else {
        v = pixel(d, lut, t, y, xs, S, cm);
      }
      o[p] = (v.r - m0) / s0;
      o[(long long)npix + p] = (v.g - m1) / s1;
      o[2ll * npix + p] = (v.b - m2) / s2;
    }
  }
}

}  // namespace

extern "C" {

size_t rsp_augment_workspace(int32_t n_clips, int32_t T, int32_t size) {
  if (n_clips <= 0 || T <= 0 || size <= 0) return 0;
  const long long nblk = ((long long)T * size * size + PIX_PER_BLOCK - 1) / PIX_PER_BLOCK;
  return rsp_align_up((size_t)n_clips * (size_t)(nblk + 1) * sizeof(float), 256);
}

int rsp_augment_batch(const rsp_augment_clip_desc* descs, int32_t n_clips, int32_t T, int32_t size, const float* mean3,
                      const float* std3, const float* blur9, float* out, int64_t out_clip_stride, void* workspace,
                      size_t workspace_bytes, void* stream) {
  RSP_REQUIRE(descs && mean3 && std3 && out && workspace, "rsp_augment_batch: null pointer");
  RSP_REQUIRE(n_clips > 0 && n_clips <= 65535 && T > 0 && size > 0, "rsp_augment_batch: bad sizes");
  RSP_REQUIRE((long long)T * size * size < (1ll << 30), "rsp_augment_batch: clip too large");
  RSP_REQUIRE(out_clip_stride >= 3ll * T * size * size, "rsp_augment_batch: out_clip_stride smaller than one clip");
  RSP_REQUIRE(std3[0] != 0.f && std3[1] != 0.f && std3[2] != 0.f, "rsp_augment_batch: zero std");
  if (workspace_bytes < rsp_augment_workspace(n_clips, T, size)) {
    rsp_set_error("rsp_augment_batch: workspace too small");
    return RSP_EWORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  const int npix = T * size * size;
  const int nblk = (npix + PIX_PER_BLOCK - 1) / PIX_PER_BLOCK;
  float* partial = reinterpret_cast<float*>(workspace);
  float* mean = partial + (size_t)n_clips * nblk;
  // clips without a contrast op leave their partials untouched: zero them so the (unused) mean is finite
  (void)hipMemsetAsync(workspace, 0, (size_t)n_clips * (nblk + 1) * sizeof(float), s);
  hipLaunchKernelGGL(augment_mean_kernel, dim3(nblk, n_clips), dim3(256), 0, s, descs, T, size, partial, nblk);
  int rc = rsp_check_launch("augment_mean_kernel");
  if (rc != RSP_OK) return rc;
  hipLaunchKernelGGL(augment_mean_final_kernel, dim3(n_clips), dim3(256), 0, s, partial, nblk, npix, mean);
  rc = rsp_check_launch("augment_mean_final_kernel");
  if (rc != RSP_OK) return rc;
  Blur9 blur;
  for (int i = 0; i < 9; ++i) blur.k[i] = blur9 ? blur9[i] : (i == 4 ? 1.f : 0.f);
  hipLaunchKernelGGL(augment_apply_kernel, dim3(nblk, n_clips), dim3(256), 0, s, descs, T, size, mean, mean3[0], mean3[1],
                     mean3[2], std3[0], std3[1], std3[2], blur, out, (long long)out_clip_stride);
  return rsp_check_launch("augment_apply_kernel");
}

}  // extern "C"
