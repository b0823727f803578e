// Probe for the TN-GEMM inner loop: what limits a wave that streams its MFMA operands straight from
// global memory.  MODE 0: constant operands (no loads); 1: real streaming loads; 2: loads that always hit
// the same small (L2-resident) window.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int U>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                             int64_t M, int64_t m_per_slice, float* out) {
  const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i0 = blockIdx.x * 128;
  const int64_t slice = (int64_t)blockIdx.y * 4 + wave;
  const int64_t m_begin = slice * m_per_slice;
  int64_t m_end = m_begin + m_per_slice;
  if (m_end > M) m_end = M;
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float a0[U][4], b0[U], a1[U][4], b1[U];
  const int offa = i0 + 4 * l31 + half * lda, offb = l31 + half * ldb;
  auto load = [&](float (&av)[U][4], float (&bv)[U], int64_t m) {
    int64_t mb = m < M - 2 * U ? m : M - 2 * U;
    if (MODE == 2) mb = (mb & 63);
    const float* as = A + mb * lda;
    const float* bs = B + mb * ldb;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (MODE == 0) { av[u][0] = av[u][1] = av[u][2] = av[u][3] = (float)m; bv[u] = 1.f; continue; }
      const float4 t = *reinterpret_cast<const float4*>(as + 2 * u * lda + offa);
      av[u][0] = t.x; av[u][1] = t.y; av[u][2] = t.z; av[u][3] = t.w;
      bv[u] = bs[2 * u * ldb + offb];
    }
  };
  float sink = 0.f;
  auto mma = [&](float (&av)[U][4], float (&bv)[U]) {
    if (MODE == 3) {          // wait for the loads (one dependent VALU op per register), MFMAs on constants
      float t = 0.f;
#pragma unroll
      for (int u = 0; u < U; ++u) t += (av[u][0] + av[u][1]) + (av[u][2] + av[u][3]) + bv[u];
      sink += t;
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(1.0f, 2.0f, acc[a], 0, 0, 0);
      return;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][a], bv[u], acc[a], 0, 0, 0);
  };
  if (m_begin < m_end) {
    load(a0, b0, m_begin);
    for (int64_t m = m_begin; m < m_end; m += 4 * U) {
      load(a1, b1, m + 2 * U);
      mma(a0, b0);
      load(a0, b0, m + 4 * U);
      mma(a1, b1);
    }
  }
  float s = sink;
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  if (s == 123.456f) out[0] = s;
}

template <int MODE, int U>
void run(const float* A, const float* B, int64_t M, int I, int J, int by, float* out) {
  int64_t mps = (M + 4 * by - 1) / (4 * by);
  mps = (mps + 4 * U - 1) / (4 * U) * (4 * U);
  dim3 grid(I / 128, by);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) probe<MODE, U><<<grid, 256>>>(A, I, B, J, M, mps, out);
  (void)hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) probe<MODE, U><<<grid, 256>>>(A, I, B, J, M, mps, out);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  double us = ms * 100.0;
  printf("mode=%d U=%d by=%3d (blocks %4d, %5ld nodes/wave)  %7.1f us  %6.1f TF/s  %6.0f GB/s\n", MODE, U, by, grid.x * by,
         (long)mps, us, 2.0 * M * I * J / us / 1e6, 4.0 * M * (I + J) / us / 1e3);
}

int main() {
  const int64_t M = 81920; const int I = 640, J = 32;
  float *A, *B, *out;
  (void)hipMalloc(&A, M * I * sizeof(float)); (void)hipMalloc(&B, M * J * sizeof(float)); (void)hipMalloc(&out, 1024);
  (void)hipMemset(A, 0, M * I * sizeof(float)); (void)hipMemset(B, 0, M * J * sizeof(float));
  for (int by : {51, 102, 153}) {
    run<0, 4>(A, B, M, I, J, by, out);
    run<3, 4>(A, B, M, I, J, by, out);
    run<2, 4>(A, B, M, I, J, by, out);
    run<1, 4>(A, B, M, I, J, by, out);
    run<1, 2>(A, B, M, I, J, by, out);
  }
  return 0;
}
