// Do fp32 MFMA chains and VALU work of DIFFERENT waves on one SIMD overlap?  Workgroups of 512 threads, one per CU =
// two waves per SIMD.  Roles by wave index (waves w and w + 4 share a SIMD):
//   mode 0  both waves of a SIMD run a dependent v_mfma_f32_32x32x2_f32 chain
//   mode 1  both run a VALU loop (v_fma_f32, or v_fmac_f32_dpp with VDPP = 1)
//   mode 2  wave w runs the MFMA chain, wave w + 4 the VALU loop       (max(t_mfma, t_valu) if they overlap, the sum if not)
//   mode 3  every wave alternates 16 MFMAs and NV VALU instructions (the shape of the fused heads' chunk loop)
//   mode 6  every wave INTERLEAVES them: one MFMA, then nv16 VALU instructions that do not depend on it, 16 times
// Times are per workgroup in shader cycles (s_memtime), register-only loops.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/overlap_probe tools/overlap_probe.hip && /tmp/overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int VDPP>
__device__ __forceinline__ void valu_block(float (&v)[16], float m) {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if constexpr (VDPP) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(m), "v"(v[(i + 5) & 15]));
    else asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[i]) : "v"(m), "v"(v[(i + 5) & 15]));
  }
}

template <int MODE, int VDPP, int SHAPE16, int PRIO = 0, int NVI = 16>
__global__ __launch_bounds__(512) void probe(int iters, int nv16, float* out, unsigned long long* clk) {
  const int wave = threadIdx.x >> 6;
  f32x16 acc; f32x4 acc4a = {0, 0, 0, 0}, acc4b = {0, 0, 0, 0};
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-3f + i;
  const float a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f, m = 1.0001f;
  if constexpr (MODE == 6) {
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        if constexpr (SHAPE16) {
          acc4a = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4a, 0, 0, 0);
          acc4b = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4b, 0, 0, 0);
        } else {
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NVI; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[i]) : "v"(m), "v"(v[(i + 5) & 15]));
      }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r] + v[r];
    s += acc4a[0] + acc4b[1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = c1 - c0;
    if (threadIdx.x == 256) clk[gridDim.x + blockIdx.x] = c1 - c0;
    return;
  }
  const bool do_mfma = MODE == 0 || MODE == 3 || (MODE == 2 && wave < 4);
  const bool do_valu = MODE == 1 || MODE == 3 || (MODE == 2 && wave >= 4);
  __syncthreads();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (do_mfma) {
      if constexpr (SHAPE16) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {     // two interleaved 16x16x4 chains (the fused FeaSt kernels' matrix phase)
          acc4a = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4a, 0, 0, 0);
          acc4b = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4b, 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      }
    }
    if (do_valu) {
      if constexpr (PRIO) __builtin_amdgcn_s_setprio(3);       // the VALU stretch outranks the other wave's MFMA chain
      for (int q = 0; q < nv16; ++q) valu_block<VDPP>(v, m);
      if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc[r] + v[r];
  s += acc4a[0] + acc4b[1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) clk[blockIdx.x] = c1 - c0;
  if (threadIdx.x == 256) clk[gridDim.x + blockIdx.x] = c1 - c0;
}

template <int MODE, int VDPP, int SHAPE16, int PRIO = 0, int NVI = 16>
double run(int nv16, const char* what) {
  const int blocks = 256, iters = 2000;
  float* out; unsigned long long* clk;
  (void)hipMalloc(&out, blocks * 512 * sizeof(float));
  (void)hipMalloc(&clk, 2 * blocks * sizeof(unsigned long long));
  probe<MODE, VDPP, SHAPE16, PRIO, NVI><<<blocks, 512>>>(10, nv16, out, clk);
  (void)hipDeviceSynchronize();
  probe<MODE, VDPP, SHAPE16, PRIO, NVI><<<blocks, 512>>>(iters, nv16, out, clk);
  (void)hipDeviceSynchronize();
  unsigned long long h[512]; (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  double a = 0, b = 0;
  for (int i = 0; i < blocks; ++i) { a += h[i]; b += h[blocks + i]; }
  a /= blocks * (double)iters; b /= blocks * (double)iters;
  printf("%-64s wave 0: %7.1f  wave 4: %7.1f cycles per iteration\n", what, a, b);
  (void)hipFree(out); (void)hipFree(clk);
  return a;
}

int main() {
  // one iteration = 16 MFMAs 32x32x2 (1024 pipe cycles) and / or nv16 x 16 VALU instructions (4 issue cycles each)
  printf("--- v_mfma_f32_32x32x2_f32 chain of 16 against 256 v_fmac_f32 (1024 issue cycles)\n");
  run<0, 0, 0>(16, "both waves of a SIMD: MFMA chain");
  run<1, 0, 0>(16, "both waves of a SIMD: VALU");
  run<2, 0, 0>(16, "wave w MFMA chain, wave w + 4 VALU");
  run<3, 0, 0>(16, "every wave: chain, then VALU");
  printf("--- the same with v_fmac_f32_dpp row_newbcast\n");
  run<1, 1, 0>(16, "both waves of a SIMD: VALU (DPP)");
  run<2, 1, 0>(16, "wave w MFMA chain, wave w + 4 VALU (DPP)");
  printf("--- two interleaved v_mfma_f32_16x16x4_f32 chains of 16 (1024 pipe cycles) against 256 v_fmac_f32\n");
  run<0, 0, 1>(16, "both waves of a SIMD: MFMA chains");
  run<2, 0, 1>(16, "wave w MFMA chains, wave w + 4 VALU");
  run<3, 0, 1>(16, "every wave: chains, then VALU");
  printf("--- light VALU share: 16 MFMAs against 96 v_fmac_f32 (the heads' forward chunk)\n");
  run<3, 0, 0>(6, "every wave: chain, then 96 VALU");
  run<2, 0, 0>(6, "wave w MFMA chain, wave w + 4 96 VALU");
  printf("--- interleaved inside every wave: 16 x (one MFMA, then n independent v_fmac_f32)\n");
  run<6, 0, 0, 0, 16>(16, "32x32x2, n = 16 (256 VALU per 16 MFMAs)");
  run<6, 0, 0, 0, 6>(6, "32x32x2, n = 6 (96 VALU per 16 MFMAs)");
  run<6, 0, 1, 0, 16>(16, "2 x 16x16x4, n = 16");
  run<6, 0, 1, 0, 6>(6, "2 x 16x16x4, n = 6");
  printf("--- s_setprio 3 around the VALU stretch\n");
  run<2, 0, 0, 1>(16, "wave w MFMA chain, wave w + 4 VALU at priority 3");
  run<3, 0, 0, 1>(16, "every wave: chain, then VALU at priority 3");
  run<3, 0, 0, 1>(6, "every wave: chain, then 96 VALU at priority 3");
  run<3, 0, 1, 1>(16, "every wave: 16x16x4 chains, then VALU at priority 3");
  return 0;
}
