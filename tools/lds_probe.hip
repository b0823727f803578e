// Probe for the NN-GEMM inner loop: MFMA operands read from LDS (tiles staged once, no global traffic).
// VARIANT 0: ds_read_b32 per operand per MFMA (k-major tiles, what gemm_nn_kernel does);
// VARIANT 1: ds_read_b128: lane holds 4 consecutive k of its row/column (row-major [row][k] tiles).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VARIANT, int TM, int TN>
__global__ __launch_bounds__(256) void probe(int ksteps, float* out) {
  constexpr int BK = 32;
  __shared__ __attribute__((aligned(16))) float As[BK * (64 * TM + 4)];
  __shared__ __attribute__((aligned(16))) float Bs[BK * (64 * TN + 4)];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
  for (int i = threadIdx.x; i < BK * (64 * TM + 4); i += 256) As[i] = i * 1e-4f;
  for (int i = threadIdx.x; i < BK * (64 * TN + 4); i += 256) Bs[i] = 1.f + i * 1e-5f;
  __syncthreads();
  f32x16 acc[TM][TN];
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int ks = 0; ks < ksteps; ++ks) {
    if (VARIANT == 0) {
      constexpr int LDA = 64 * TM + 1, LDB = 64 * TN + 4;
#pragma unroll
      for (int kk = 0; kk < BK; kk += 2) {
        float a[TM], b[TN];
        const int kr = kk + (lane >> 5);
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = As[kr * LDA + (wm * TM + i) * 32 + (lane & 31)];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = Bs[kr * LDB + (wn * TN + j) * 32 + (lane & 31)];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    } else {
      // [row][k] tiles with a 36-float pitch: lane (row = l & 31, half = l >> 5) reads k = 4 (2q + half) .. + 3
      constexpr int P = BK + 4;
#pragma unroll
      for (int q = 0; q < BK / 8; ++q) {
        float4 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          a[i] = *reinterpret_cast<const float4*>(&As[((wm * TM + i) * 32 + (lane & 31)) * P + 4 * (2 * q + (lane >> 5))]);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          b[j] = *reinterpret_cast<const float4*>(&Bs[((wn * TN + j) * 32 + (lane & 31)) * P + 4 * (2 * q + (lane >> 5))]);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              const float av = c == 0 ? a[i].x : c == 1 ? a[i].y : c == 2 ? a[i].z : a[i].w;
              const float bv = c == 0 ? b[j].x : c == 1 ? b[j].y : c == 2 ? b[j].z : b[j].w;
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
            }
      }
    }
    __builtin_amdgcn_s_barrier();      // the k-loop barrier of the real kernel
  }
  float s = 0.f;
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  if (s == 123.456f) out[0] = s;
}

template <int VARIANT, int TM, int TN>
void run(int blocks, int ksteps, float* out) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) probe<VARIANT, TM, TN><<<blocks, 256>>>(ksteps, out);
  (void)hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) probe<VARIANT, TM, TN><<<blocks, 256>>>(ksteps, out);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  double us = ms * 100.0;
  double flop = (double)blocks * 4 * ksteps * 16.0 * TM * TN * 4096.0;
  printf("variant=%d tile=%dx%d blocks=%4d  %7.1f us  %6.1f TF/s\n", VARIANT, TM, TN, blocks, us, flop / us / 1e6);
}

int main() {
  float* out; (void)hipMalloc(&out, 1024);
  for (int blocks : {256, 512, 768, 1024}) {
    run<0, 1, 1>(blocks, 36, out);
    run<1, 1, 1>(blocks, 36, out);
    run<0, 2, 1>(blocks, 36, out);
    run<1, 2, 1>(blocks, 36, out);
    run<0, 2, 2>(blocks, 36, out);
    run<1, 2, 2>(blocks, 36, out);
  }
  return 0;
}
