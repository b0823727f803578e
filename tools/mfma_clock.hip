// Measures what the fp32 MFMA pipe sustains on this device: register-only loops of
// v_mfma_f32_32x32x2_f32 on every SIMD, shader clock (s_memtime) against the constant 100 MHz wall
// clock.  Prints achieved TFLOP/s and the clock held under that load -- the practical ceiling the GEMM
// numbers in DESIGN.md are read against.  Variants: NACC independent accumulator chains per wave,
// operands rotating over NOP distinct registers.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/mfma_clock tools/mfma_clock.hip && /tmp/mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int NOP>
__global__ __launch_bounds__(256) void mfma_loop(int iters, float* out, unsigned long long* clk) {
  f32x16 acc[NACC];
  for (int t = 0; t < NACC; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float a[NOP], b[NOP];
  for (int q = 0; q < NOP; ++q) { a[q] = threadIdx.x * 1e-3f + q; b[q] = 1.0f + blockIdx.x * 1e-6f + q; }
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int q = 0; q < NOP; ++q)
#pragma unroll
      for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(q + t) % NOP], b[q], acc[t], 0, 0, 0);
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
  float s = 0.f;
  for (int t = 0; t < NACC; ++t)
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int NACC, int NOP>
void run(int blocks_per_cu) {
  const int iters = 40000 / (NACC * NOP);
  const int blocks = 256 * blocks_per_cu;
  float* out; unsigned long long* clk;
  (void)hipMalloc(&out, blocks * 256 * sizeof(float));
  (void)hipMalloc(&clk, blocks * 2 * sizeof(unsigned long long));
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  mfma_loop<NACC, NOP><<<blocks, 256>>>(10, out, clk);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  mfma_loop<NACC, NOP><<<blocks, 256>>>(iters, out, clk);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2]; (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  double n_mfma = (double)iters * NACC * NOP;
  double flop = (double)blocks * 4 * n_mfma * 4096.0;
  printf("acc chains=%d operands=%d waves/SIMD=%d  %6.1f TFLOP/s  clock %.3f GHz  cycles/MFMA/wave %.1f\n", NACC, NOP,
         blocks_per_cu, flop / (ms * 1e-3) / 1e12, h[0] / (h[1] / 100e6) / 1e9, (double)h[0] / n_mfma);
  (void)hipFree(out); (void)hipFree(clk);
}

int main() {
  run<4, 1>(1); run<4, 1>(2);
  run<1, 1>(1); run<1, 1>(2); run<1, 1>(4);
  run<1, 8>(1); run<1, 8>(2); run<1, 8>(4);
  run<2, 8>(1); run<2, 8>(2);
  run<4, 8>(1); run<4, 8>(2);
  return 0;
}
