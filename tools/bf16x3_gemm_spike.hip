// Research spike (NOT a product path, nothing links it): how fast is an fp32 GEMM emulated on the bf16 matrix pipe with a
// 2-way split a = hi + lo (hi = bf16(a), lo = bf16(a - hi)) and three MFMAs per product (hi.hi, hi.lo, lo.hi)?
// DESIGN.md §8 has the accuracy side (tools/bf16_split_accuracy.py); this measures the throughput side on the GEMM shape of
// C3D conv3b (M = 200704 rows, N = 256, K = 6912) with the simplest structure the product kernels also use: 128x128 tile,
// 4 waves x (64x64), K-chunks of 32, register-staged double-buffered LDS, one barrier per chunk.
//
//   hipcc --offload-arch=gfx950 -O3 tools/bf16x3_gemm_spike.hip -o /tmp/bf16x3_spike && /tmp/bf16x3_spike
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// fp32 [rows][K] -> bf16 hi / lo planes [rows][K]
__global__ void split_kernel(const float* __restrict__ x, __bf16* __restrict__ hi, __bf16* __restrict__ lo, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float v = x[i];
    const __bf16 h = (__bf16)v;
    hi[i] = h;
    lo[i] = (__bf16)(v - (float)h);
  }
}

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int PITCH = BK + 8;   // bf16 elements per LDS row: 80 bytes, keeps ds_read_b128 lane groups on distinct banks

// C[M][N] = A[M][K] . B[N][K]^T  from split planes
template <int NBUF>
__global__ __launch_bounds__(256, 2) void gemm_bf16x3(const __bf16* __restrict__ Ah, const __bf16* __restrict__ Al,
                                                      const __bf16* __restrict__ Bh, const __bf16* __restrict__ Bl,
                                                      float* __restrict__ C, int M, int N, int K) {
  __shared__ __attribute__((aligned(16))) __bf16 sA[NBUF][2][BM][PITCH];   // [buf][plane][row][k]
  __shared__ __attribute__((aligned(16))) __bf16 sB[NBUF][2][BN][PITCH];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l32 = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int n_tiles = N / BN;
  const int m0 = (blockIdx.x / n_tiles) * BM, n0 = (blockIdx.x % n_tiles) * BN;
  // staging: 128 rows x 32 k bf16 = 128 x 4 x 16 B per plane; thread -> (row = t/4 + 64*i, 16-byte slot t%4)
  const int srow = t >> 2, sslot = (t & 3) * 8;
  uint4 ra[2][2], rb[2][2];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const long long ao = (long long)(m0 + srow + 64 * i) * K + k0 + sslot;
      const long long bo = (long long)(n0 + srow + 64 * i) * K + k0 + sslot;
      ra[0][i] = *reinterpret_cast<const uint4*>(Ah + ao);
      ra[1][i] = *reinterpret_cast<const uint4*>(Al + ao);
      rb[0][i] = *reinterpret_cast<const uint4*>(Bh + bo);
      rb[1][i] = *reinterpret_cast<const uint4*>(Bl + bo);
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        *reinterpret_cast<uint4*>(&sA[buf][p][srow + 64 * i][sslot]) = ra[p][i];
        *reinterpret_cast<uint4*>(&sB[buf][p][srow + 64 * i][sslot]) = rb[p][i];
      }
  };
  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  gload(0);
  sstore(0);
  __syncthreads();
  int buf = 0;
  for (int k0 = 0; k0 < K; k0 += BK) {
    const bool more = k0 + BK < K;
    if (more) gload(k0 + BK);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {   // two K=16 steps per chunk
      bf16x8 a[2][2], b[2][2];         // [plane][tile]
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a[p][i] = *reinterpret_cast<const bf16x8*>(&sA[buf][p][wm * 64 + i * 32 + l32][ks * 16 + h * 8]);
          b[p][i] = *reinterpret_cast<const bf16x8*>(&sB[buf][p][wn * 64 + i * 32 + l32][ks * 16 + h * 8]);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], acc[i][j], 0, 0, 0);   // lo.hi
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], acc[i][j], 0, 0, 0);   // hi.lo
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], acc[i][j], 0, 0, 0);   // hi.hi
        }
    }
    if (NBUF == 2) {
      if (more) sstore(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    } else {   // single buffer, three workgroups per CU cover the two barriers
      __syncthreads();
      if (more) sstore(0);
      __syncthreads();
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 64 + i * 32 + (e >> 2) * 8 + h * 4 + (e & 3), col = n0 + wn * 64 + j * 32 + l32;
        C[(long long)row * N + col] = acc[i][j][e];
      }
}

int main() {
  const int M = 200704, N = 256, K = 6912;
  std::vector<float> hA((size_t)4096 * K), hB((size_t)N * K);
  srand(1);
  for (auto& v : hA) v = fmaxf(0.f, (rand() / (float)RAND_MAX - 0.4f) * 2.6f);        // post-ReLU like
  for (auto& v : hB) v = (rand() / (float)RAND_MAX - 0.5f) * 0.06f;
  float *dA, *dB, *dC;
  __bf16 *Ah, *Al, *Bh, *Bl;
  CK(hipMalloc(&dA, (size_t)M * K * 4)); CK(hipMalloc(&dB, (size_t)N * K * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
  CK(hipMalloc(&Ah, (size_t)M * K * 2)); CK(hipMalloc(&Al, (size_t)M * K * 2));
  CK(hipMalloc(&Bh, (size_t)N * K * 2)); CK(hipMalloc(&Bl, (size_t)N * K * 2));
  for (int r = 0; r < M; r += 4096) CK(hipMemcpy(dA + (size_t)r * K, hA.data(), (size_t)4096 * K * 4, hipMemcpyHostToDevice));   // 49 copies
  CK(hipMemcpy(dB, hB.data(), (size_t)N * K * 4, hipMemcpyHostToDevice));
  split_kernel<<<8192, 256>>>(dA, Ah, Al, (long long)M * K);
  split_kernel<<<2048, 256>>>(dB, Bh, Bl, (long long)N * K);
  CK(hipDeviceSynchronize());
  const int grid = (M / BM) * (N / BN);
  for (int nb = 1; nb <= 2; ++nb) {
  if (nb == 1) gemm_bf16x3<1><<<grid, 256>>>(Ah, Al, Bh, Bl, dC, M, N, K); else gemm_bf16x3<2><<<grid, 256>>>(Ah, Al, Bh, Bl, dC, M, N, K);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int i = 0; i < 5; ++i) { if (nb == 1) gemm_bf16x3<1><<<grid, 256>>>(Ah, Al, Bh, Bl, dC, M, N, K); else gemm_bf16x3<2><<<grid, 256>>>(Ah, Al, Bh, Bl, dC, M, N, K); }
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 5;
  const double fl = 2.0 * M * N * K;
  printf("bf16x3 GEMM %dx%dx%d, %d LDS buffer(s): %.3f ms = %.1f TFLOP/s fp32-equivalent (%.2f x the 157.3 TF fp32-MFMA peak; grid %d)\n", M, N, K, nb, ms,
         fl / ms / 1e9, fl / ms / 1e9 / 157.3, grid);
  }
  // accuracy on sampled outputs vs float64
  std::vector<float> hC((size_t)256 * N);
  CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0;
  for (int r = 0; r < 256; r += 17)
    for (int c = 0; c < N; c += 13) {
      double s = 0, sa = 0;
      for (int k = 0; k < K; ++k) {
        s += (double)hA[(size_t)r * K + k] * hB[(size_t)c * K + k];
        sa += fabs((double)hA[(size_t)r * K + k] * hB[(size_t)c * K + k]);
      }
      worst = fmax(worst, fabs(hC[(size_t)r * N + c] - s) / sa);
    }
  printf("max |err| / sum|a||b| over sampled outputs: %.2e\n", worst);
  return 0;
}
