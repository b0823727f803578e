// Fused regression heads: Linear(32 -> 1024) + leaky_relu(0.2) + Linear(1024 -> 1|3) + finish, with
// the [N, 1024] hidden activation kept in MFMA accumulators -- it never touches HBM, forward or
// backward (/root/reference/code/network.py:324-343 materialises it twice: 126 MB per 20 k-face
// mesh, plus the autograd copies).
//
// MFMA 32x32x2 fp32 bookkeeping used throughout (lane l: l31 = l & 31, half = l >> 5):
//   A operand  : A[i = l31][k(s, half)]        B operand : B[k(s, half)][j = l31]
//   C/D        : column j = l31, row(r, half) = (r & 3) + 8 (r >> 2) + 4 half,  r = 0..15
// Any pairing k(s, half) works as long as A and B agree.  Two pairings are used:
//   "contiguous"  k(s, half) = 16 half + s      -- both operands are 16 contiguous floats per lane
//   "accumulator" k(r, half) = row(r, half)     -- a C/D tile is fed back as the B operand as is
#include "common.h"

#include <cstdlib>

namespace geobi {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CIN = 32;      // head input channels (GNNModule output width)
constexpr int HID = 1024;    // hidden width
constexpr int NCHUNK = HID / 32;
constexpr float kEps = 1e-12f;

__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

__device__ __forceinline__ void load16(const float* __restrict__ p, float (&v)[16]) {
  const float4* q = reinterpret_cast<const float4*>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float4 t = q[i];
    v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
  }
}

// ------------------------------------------------------------------------------ forward
// One block = 32 nodes, its four waves take every fourth hidden chunk of 32 units (8 chunks each) and their [32, NOUT]
// partial outputs meet in 1.5 KB of LDS in wave order (fixed: deterministic).  Until round 3 one WAVE walked all 32
// chunks of its 32 nodes: the vertex head of the bench batch was 320 blocks on 256 CUs -- one wave per SIMD, nothing to
// run under a dependent MFMA chain (16 x 64 cycles per chunk, then ~100 VALU instructions that wait for it) and a second,
// quarter-full round of blocks: 0.33 of the fp32 MFMA peak.
template <int NOUT>
__global__ __launch_bounds__(256) void head_fwd_fused_kernel(
    const float* __restrict__ x, int N, const float* __restrict__ w1, const float* __restrict__ b1,
    const float* __restrict__ w2, const float* __restrict__ b2, float slope, int mode,
    const float* __restrict__ dd, const float* __restrict__ resid, int ld_resid, float* __restrict__ raw,
    float* __restrict__ out) {
  __shared__ float s_part[4][32][NOUT + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int n0 = blockIdx.x * 32;
  const int row = min(n0 + l31, N - 1);
  float ax[16];
  load16(x + (size_t)row * CIN + 16 * half, ax);

  float part[16][NOUT];
#pragma unroll
  for (int r = 0; r < 16; ++r)
#pragma unroll
    for (int o = 0; o < NOUT; ++o) part[r][o] = 0.f;

  auto chunk = [&](int c, const float (&bw)[16]) {
    const int j = c * 32 + l31;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[s], bw[s], acc, 0, 0, 0);
    const float b1v = b1[j];
    float w2v[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) w2v[o] = w2[(size_t)o * HID + j];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float h = acc[r] + b1v;
      h = h > 0.f ? h : h * slope;
#pragma unroll
      for (int o = 0; o < NOUT; ++o) part[r][o] = fmaf(h, w2v[o], part[r][o]);
    }
  };
  // the next chunk's W1 rows are in flight while the current chunk multiplies; wave w owns chunks w, w + 4, ...
  const float* w1l = w1 + (size_t)l31 * CIN + 16 * half;
  float bwa[16], bwb[16];
  load16(w1l + (size_t)wave * 32 * CIN, bwa);
  for (int c = wave; c < NCHUNK; c += 8) {
    load16(w1l + (size_t)(c + 4) * 32 * CIN, bwb);
    chunk(c, bwa);
    if (c + 8 < NCHUNK) load16(w1l + (size_t)(c + 8) * 32 * CIN, bwa);
    chunk(c + 4, bwb);
  }
  // sum over the 32 hidden units held by the lanes of each half
#pragma unroll
  for (int r = 0; r < 16; ++r)
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
      float v = part[r][o];
#pragma unroll
      for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
      part[r][o] = v;
    }
  // lane (l31 == r) holds row(r, half) of this wave's partial output
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    if (l31 != r) continue;
#pragma unroll
    for (int o = 0; o < NOUT; ++o) s_part[wave][acc_row(r, half)][o] = part[r][o];
  }
  __syncthreads();
  // thread t < 32 finishes node n0 + t: the four waves' shares in wave order, bias, finish
  if (threadIdx.x < 32) {
    const int node = n0 + threadIdx.x;
    if (node >= N) return;
    float v[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
      v[o] = ((s_part[0][threadIdx.x][o] + s_part[1][threadIdx.x][o]) + s_part[2][threadIdx.x][o]) +
             s_part[3][threadIdx.x][o] + b2[o];
      raw[(size_t)node * NOUT + o] = v[o];
    }
    float res[3];
    if (mode == 0) {
#pragma unroll
      for (int cidx = 0; cidx < 3; ++cidx) {
        float t = (NOUT == 3) ? v[cidx % NOUT] : v[0] * dd[(size_t)node * 3 + cidx];
        res[cidx] = t + resid[(size_t)node * ld_resid + cidx];
      }
    } else {
      float len = fmaxf(sqrtf(v[0] * v[0] + v[1 % NOUT] * v[1 % NOUT] + v[2 % NOUT] * v[2 % NOUT]), kEps);
#pragma unroll
      for (int cidx = 0; cidx < 3; ++cidx) res[cidx] = v[cidx % NOUT] / len;
    }
    out[(size_t)node * 3] = res[0]; out[(size_t)node * 3 + 1] = res[1]; out[(size_t)node * 3 + 2] = res[2];
  }
}

// ------------------------------------------------------------------------------ forward, split precision
// The first GEMM (x W1^T: the head's 2 N 32 1024 flops) as SIX bf16 products with fp32 accumulation: every fp32 operand is
// cut into three bf16 pieces a = a1 + a2 + a3 (each the round-to-nearest bf16 of what is left: 3 x 8 = 24 significand bits),
// and a b = a1 b1 + a1 b2 + a2 b1 + a1 b3 + a3 b1 + a2 b2 (+ terms below 2^-32 |a b|), every partial product exact in the
// fp32 accumulator.  On the path's real operands this form is 0.5-0.9 x plain fp32's own distance to fp64
// (tools/bf16_split_study.py, profiles/r04_bf16_split_study.txt).  v_mfma_f32_32x32x16_bf16 runs at 16 x the fp32 MFMA rate
// (32 cycles for K = 16 against 64 for K = 2): a 32-unit chunk is 12 instructions = 384 matrix cycles instead of 16 = 1 024,
// and -- unlike the fp32 shapes -- it leaves the SIMD's vector issue slots to the other waves' VALU work.
// Behind geobi_set_head_precision(1) / GEOBI_HEAD_BF16X3=1; the default path stays exact fp32.
// Operand maps (cdna_hip_programming: lane l, r = l & 31, h = l >> 5): A[row r][k = 8 h + j], B[k = 8 h + j][col r], j = 0..7
// per K = 16 block; C/D as the fp32 shape (column on the lane, acc_row(reg, h)), so the epilogue is the fp32 kernel's.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Split3 { bf16x8 p[3]; };

__device__ __forceinline__ Split3 split3(const float (&v)[8]) {
  Split3 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 a = (__bf16)v[j];
    const float r1 = v[j] - (float)a;
    const __bf16 b = (__bf16)r1;
    const float r2 = r1 - (float)b;
    o.p[0][j] = a; o.p[1][j] = b; o.p[2][j] = (__bf16)r2;
  }
  return o;
}

__device__ __forceinline__ void load8(const float* __restrict__ p, float (&v)[8]) {
  const float4* q = reinterpret_cast<const float4*>(p);
  const float4 a = q[0], b = q[1];
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// acc += a b over one K = 16 block, smallest terms first
__device__ __forceinline__ f32x16 mfma6(const Split3& a, const Split3& b, f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[2], b.p[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], b.p[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[1], b.p[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[1], b.p[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], b.p[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], b.p[0], acc, 0, 0, 0);
  return acc;
}

// W1 [HID, CIN] fp32 -> its three bf16 pieces, [3][HID][CIN] (once per call: 32 K elements)
__global__ void head_split_w1_kernel(const float* __restrict__ w1, __bf16* __restrict__ pk) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= HID * CIN) return;
  const float v = w1[i];
  const __bf16 a = (__bf16)v;
  const float r1 = v - (float)a;
  const __bf16 b = (__bf16)r1;
  pk[i] = a; pk[HID * CIN + i] = b; pk[2 * HID * CIN + i] = (__bf16)(r1 - (float)b);
}

__device__ __forceinline__ Split3 load_split(const __bf16* __restrict__ pk, size_t off) {
  Split3 o;
#pragma unroll
  for (int p = 0; p < 3; ++p) o.p[p] = *reinterpret_cast<const bf16x8*>(pk + (size_t)p * HID * CIN + off);
  return o;
}

template <int NOUT>
__global__ __launch_bounds__(256) void head_fwd_fused_bf16x3_kernel(
    const float* __restrict__ x, int N, const __bf16* __restrict__ w1p, const float* __restrict__ b1,
    const float* __restrict__ w2, const float* __restrict__ b2, float slope, int mode,
    const float* __restrict__ dd, const float* __restrict__ resid, int ld_resid, float* __restrict__ raw,
    float* __restrict__ out) {
  __shared__ float s_part[4][32][NOUT + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int n0 = blockIdx.x * 32;
  const int row = min(n0 + l31, N - 1);
  // the lane's share of its node row: k = 8 half + j of both K = 16 blocks, in three bf16 pieces
  Split3 ax[2];
  {
    float v[8];
    load8(x + (size_t)row * CIN + 8 * half, v);      ax[0] = split3(v);
    load8(x + (size_t)row * CIN + 16 + 8 * half, v); ax[1] = split3(v);
  }
  float part[16][NOUT];
#pragma unroll
  for (int r = 0; r < 16; ++r)
#pragma unroll
    for (int o = 0; o < NOUT; ++o) part[r][o] = 0.f;

  auto chunk = [&](int c, const Split3& bw0, const Split3& bw1) {
    const int j = c * 32 + l31;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = mfma6(ax[0], bw0, acc);
    acc = mfma6(ax[1], bw1, acc);
    const float b1v = b1[j];
    float w2v[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) w2v[o] = w2[(size_t)o * HID + j];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float h = acc[r] + b1v;
      h = h > 0.f ? h : h * slope;
#pragma unroll
      for (int o = 0; o < NOUT; ++o) part[r][o] = fmaf(h, w2v[o], part[r][o]);
    }
  };
  // wave w owns chunks w, w + 4, ...; the next chunk's W1 pieces are in flight while the current chunk multiplies
  const size_t wl = (size_t)l31 * CIN + 8 * half;
  Split3 wa0 = load_split(w1p, wl + (size_t)wave * 32 * CIN), wa1 = load_split(w1p, wl + (size_t)wave * 32 * CIN + 16);
  for (int c = wave; c < NCHUNK; c += 8) {
    const Split3 wb0 = load_split(w1p, wl + (size_t)(c + 4) * 32 * CIN), wb1 = load_split(w1p, wl + (size_t)(c + 4) * 32 * CIN + 16);
    chunk(c, wa0, wa1);
    if (c + 8 < NCHUNK) { wa0 = load_split(w1p, wl + (size_t)(c + 8) * 32 * CIN); wa1 = load_split(w1p, wl + (size_t)(c + 8) * 32 * CIN + 16); }
    chunk(c + 4, wb0, wb1);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r)
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
      float v = part[r][o];
#pragma unroll
      for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
      part[r][o] = v;
    }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    if (l31 != r) continue;
#pragma unroll
    for (int o = 0; o < NOUT; ++o) s_part[wave][acc_row(r, half)][o] = part[r][o];
  }
  __syncthreads();
  if (threadIdx.x < 32) {
    const int node = n0 + threadIdx.x;
    if (node >= N) return;
    float v[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
      v[o] = ((s_part[0][threadIdx.x][o] + s_part[1][threadIdx.x][o]) + s_part[2][threadIdx.x][o]) +
             s_part[3][threadIdx.x][o] + b2[o];
      raw[(size_t)node * NOUT + o] = v[o];
    }
    float res[3];
    if (mode == 0) {
#pragma unroll
      for (int cidx = 0; cidx < 3; ++cidx) {
        float t = (NOUT == 3) ? v[cidx % NOUT] : v[0] * dd[(size_t)node * 3 + cidx];
        res[cidx] = t + resid[(size_t)node * ld_resid + cidx];
      }
    } else {
      float len = fmaxf(sqrtf(v[0] * v[0] + v[1 % NOUT] * v[1 % NOUT] + v[2 % NOUT] * v[2 % NOUT]), kEps);
#pragma unroll
      for (int cidx = 0; cidx < 3; ++cidx) res[cidx] = v[cidx % NOUT] / len;
    }
    out[(size_t)node * 3] = res[0]; out[(size_t)node * 3 + 1] = res[1]; out[(size_t)node * 3 + 2] = res[2];
  }
}

// ----------------------------------------------------------------------------- backward
// Persistent blocks of 8 waves, one block per CU = two waves per SIMD: a dependent MFMA chain blocks its wave's
// instruction stream for the chain's whole duration (in-order issue), so only ANOTHER wave on the SIMD can run the
// VALU / LDS stretches under it.  All waves of a block work on the same 32-node tile, wave w owning the hidden
// chunks {w, w+8, w+16, w+24}.  Per chunk: recompute h (16 MFMA), dh in registers, then
//   dW1^T[k, j] += x^T dh      -- dh (a C/D tile) is the B operand as it stands ("accumulator" pairing)
//   dx[n, k]    += dh W1       -- dh transposed through a 2 KB per-wave LDS tile, 16 node rows at a time
// dW1 / dW2 / db1 partial sums stay in registers across tiles and are written once per block; a
// second kernel adds the per-block slabs in a fixed order (deterministic, no atomics).
#ifdef GEOBI_HEAD_STAMPS
// Diagnostic build only: shader-clock time wave 0 of every block spends between the phase boundaries of a chunk,
// summed over the block's chunks (tools/head_stamps.py).
__device__ unsigned long long g_head_stamps[1024][8];
#define GEOBI_HS(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); hs_acc[i] += now_ - hs_t; hs_t = now_; } while (0)
#else
#define GEOBI_HS(i) do { } while (0)
#endif

constexpr int HBW = 8;                 // waves per block of the backward kernel
constexpr int HCPW = NCHUNK / HBW;     // hidden chunks per wave and tile

template <int NOUT>
__global__ __launch_bounds__(512, 1) void head_bwd_fused_kernel(
    const float* __restrict__ x, int N, int ntiles, const float* __restrict__ w1, const float* __restrict__ b1,
    const float* __restrict__ w2, float slope, const float* __restrict__ graw, float* __restrict__ dx,
    float* __restrict__ p_dw1, float* __restrict__ p_dw2, float* __restrict__ p_db1, float* __restrict__ p_db2) {
  // The block's running sums live in LDS (all 160 KiB of it: 128 KiB dW1^T tiles, 12 KiB dW2, 4 KiB
  // db1, 16 KiB transposition / fold scratch: 2 KiB per wave); every LDS word has exactly one writer lane, so the
  // accumulation order is fixed.
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* sW1 = lds;                                  // [NCHUNK][16][64]
  float* sW2 = sW1 + NCHUNK * 16 * 64;               // [3][HID]
  float* sB1 = sW2 + 3 * HID;                        // [HID]
  float* sT = sB1 + HID;                             // [8][16][32] dh transposition, reused as [8][8][64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  for (int i = threadIdx.x; i < NCHUNK * 16 * 64 + 3 * HID + HID; i += 64 * HBW) lds[i] = 0.f;
  __syncthreads();

  float db2a[NOUT];
#pragma unroll
  for (int o = 0; o < NOUT; ++o) db2a[o] = 0.f;
  float* tw = sT + wave * 512;
#ifdef GEOBI_HEAD_STAMPS
  unsigned long long hs_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long hs_t = __builtin_amdgcn_s_memtime();
#endif

  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int n0 = t * 32;
    float ax[16], ax2[16], gr[16][NOUT];
    load16(x + (size_t)min(n0 + l31, N - 1) * CIN + 16 * half, ax);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int node = n0 + acc_row(r, half);
      const bool ok = node < N;
      ax2[r] = x[(size_t)(ok ? node : N - 1) * CIN + l31];
#pragma unroll
      for (int o = 0; o < NOUT; ++o) {
        gr[r][o] = ok ? graw[(size_t)node * NOUT + o] : 0.f;
        db2a[o] += gr[r][o];                                     // every lane of a half holds the same 16 rows
      }
    }
    f32x16 dxacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) dxacc[r] = 0.f;
    GEOBI_HS(0);                                   // tile prologue (x rows, output gradients)

    // Order inside a chunk (s_memtime stamps, tools/head_stamps.py): the W1 columns of the dx product and the next
    // chunk's W1 rows are requested before h is waited for; the dW1 chain is issued as soon as dh exists, the dx chain
    // right behind the transposition, the db1 / dW2 sums after them; the dW1 tile is added to its LDS copy last.
    float bw[16];
    load16(w1 + (size_t)(wave * 32 + l31) * CIN + 16 * half, bw);
    for (int cc = 0; cc < HCPW; ++cc) {
      const int c = wave + HBW * cc;
      const int j = c * 32 + l31;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[s], bw[s], acc, 0, 0, 0);
      GEOBI_HS(1);                                 // recompute chain issued
      float bq[16];
      const float* wq = w1 + (size_t)(c * 32 + 16 * half) * CIN + l31;
#pragma unroll
      for (int s = 0; s < 16; ++s) bq[s] = wq[s * CIN];
      const float b1v = b1[j];
      float w2v[NOUT], dw2c[NOUT];
#pragma unroll
      for (int o = 0; o < NOUT; ++o) { w2v[o] = w2[(size_t)o * HID + j]; dw2c[o] = 0.f; }
      load16(w1 + (size_t)((wave + HBW * ((cc + 1) % HCPW)) * 32 + l31) * CIN + 16 * half, bw);   // next chunk's rows
      float dh[16];
      float db1c = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float hpre = acc[r] + b1v;
        float hval = hpre > 0.f ? hpre : hpre * slope;
        float gh = 0.f;
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
          gh = fmaf(gr[r][o], w2v[o], gh);
          dw2c[o] = fmaf(gr[r][o], hval, dw2c[o]);
        }
        dh[r] = hpre > 0.f ? gh : gh * slope;
        db1c += dh[r];
      }
      GEOBI_HS(2);                                 // waited for h, dh formed
      // dW1^T chunk tile [k x j]: A = x^T (ax2), B = dh as it stands
      f32x16 tacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) tacc[r] = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) tacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ax2[r], dh[r], tacc, 0, 0, 0);
      // dx: transpose dh through LDS (XOR-swizzled 16-B slots) -> A operand [node][j]; B = W1 rows.  The scratch
      // holds 16 node rows: rows 0..15 (registers r < 8) serve lanes l31 < 16, rows 16..31 the others.
      float ad[16];
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int r8 = 0; r8 < 8; ++r8) {
          const int r = 8 * pass + r8;
          const int row = acc_row(r, half) - 16 * pass;
          tw[row * 32 + (l31 ^ ((row & 7) << 2))] = dh[r];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if ((l31 >> 4) == pass) {
          const int row = l31 & 15;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float4 v = *reinterpret_cast<const float4*>(&tw[row * 32 + ((16 * half + 4 * q) ^ ((row & 7) << 2))]);
            ad[4 * q] = v.x; ad[4 * q + 1] = v.y; ad[4 * q + 2] = v.z; ad[4 * q + 3] = v.w;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      GEOBI_HS(3);                                 // dW1 chain issued, dh transposed
#pragma unroll
      for (int s = 0; s < 16; ++s) dxacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ad[s], bq[s], dxacc, 0, 0, 0);
      GEOBI_HS(4);                                 // dx chain issued
      // the two halves of a column hold different node rows: fold them, lane < 32 owns the LDS word
      db1c += __shfl_xor(db1c, 32, 64);
      if (half == 0) sB1[j] += db1c;
#pragma unroll
      for (int o = 0; o < NOUT; ++o) {
        float d = dw2c[o] + __shfl_xor(dw2c[o], 32, 64);
        if (half == 0) sW2[o * HID + j] += d;
      }
      // the dW1 tile, added to the block's LDS copy
      float* w1c = sW1 + (size_t)c * 16 * 64 + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) w1c[r * 64] += tacc[r];
      GEOBI_HS(5);                                 // sums, dW1 tile added
    }
    // fold the eight waves' dx tiles in a fixed order, eight accumulator registers at a time (scratch as [8][8][64])
#pragma unroll
    for (int hr = 0; hr < 2; ++hr) {
      __syncthreads();
#pragma unroll
      for (int r8 = 0; r8 < 8; ++r8) sT[(wave * 8 + r8) * 64 + lane] = dxacc[8 * hr + r8];
      __syncthreads();
      {
        const int r8 = wave;                       // wave w folds register 8 hr + w
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < HBW; ++w) v += sT[(w * 8 + r8) * 64 + lane];
        const int node = n0 + acc_row(8 * hr + r8, half);
        if (node < N) dx[(size_t)node * CIN + l31] = v;
      }
    }
    __syncthreads();
    GEOBI_HS(6);                                   // dx fold over the block's waves, stores
  }
#ifdef GEOBI_HEAD_STAMPS
  if (threadIdx.x == 0 && blockIdx.x < 1024)
    for (int i = 0; i < 8; ++i) g_head_stamps[blockIdx.x][i] = hs_acc[i];
#endif

  // per-block partial sums -> slabs
  __syncthreads();
  float* pw = p_dw1 + (size_t)blockIdx.x * (NCHUNK * 16 * 64);
  for (int i = threadIdx.x; i < NCHUNK * 16 * 64; i += 64 * HBW) pw[i] = sW1[i];
  for (int i = threadIdx.x; i < NOUT * HID; i += 64 * HBW) p_dw2[(size_t)blockIdx.x * NOUT * HID + i] = sW2[i];
  for (int i = threadIdx.x; i < HID; i += 64 * HBW) p_db1[(size_t)blockIdx.x * HID + i] = sB1[i];
  if (wave == 0) {
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
      float d = db2a[o] + __shfl_xor(db2a[o], 32, 64);
      if (lane == 0) p_db2[(size_t)blockIdx.x * 4 + o] = d;
    }
  }
}

// Sum of the per-block slabs in a fixed order.  64 outputs per block, four threads per output: thread (o, q) adds
// slabs q, q + 4, ... (the loads of different slabs are independent: 16 in flight per thread), the four partial sums
// meet in LDS in the order q = 0..3.  (One thread per output walking all 256 slabs in dw1's order -- reads 256 B apart --
// took 38 us for 37 MB.)
__global__ __launch_bounds__(256) void head_bwd_reduce_kernel(const float* __restrict__ p_dw1, const float* __restrict__ p_dw2,
                                       const float* __restrict__ p_db1, const float* __restrict__ p_db2,
                                       int blocks, int nout, int accumulate, float* __restrict__ dw1,
                                       float* __restrict__ db1, float* __restrict__ dw2, float* __restrict__ db2) {
  __shared__ float part[4][64];
  const int o = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + o;
  const int n_dw1 = HID * CIN;
  const int total = n_dw1 + HID + nout * HID + nout;
  const float* src = nullptr;
  size_t stride = 0, off = 0;
  float* dst = nullptr;
  if (idx < n_dw1) {
    // threads walk the slab in ITS order (coalesced reads of 37 MB; the 128 KB of writes scatter instead):
    // slab element (chunk c, reg r, lane)  ->  dw1[j, k] with j = 32 c + (lane & 31), k = row(r, lane >> 5)
    const int c = idx >> 10, r = (idx >> 6) & 15, lane = idx & 63;
    const int j = c * 32 + (lane & 31), k = acc_row(r, lane >> 5);
    src = p_dw1; stride = (size_t)NCHUNK * 16 * 64; off = idx; dst = dw1 + (size_t)j * CIN + k;
  } else if (idx < n_dw1 + HID) {
    src = p_db1; stride = HID; off = idx - n_dw1; dst = db1 + off;
  } else if (idx < n_dw1 + HID + nout * HID) {
    src = p_dw2; stride = (size_t)nout * HID; off = idx - n_dw1 - HID; dst = dw2 + off;
  } else if (idx < total) {
    src = p_db2; stride = 4; off = idx - n_dw1 - HID - nout * HID; dst = db2 + off;
  }
  float s = 0.f;
  if (src != nullptr) {
#pragma unroll 16
    for (int b = q; b < blocks; b += 4) s += src[(size_t)b * stride + off];
  }
  part[q][o] = s;
  __syncthreads();
  if (q == 0 && dst != nullptr) {
    const float t = ((part[0][o] + part[1][o]) + part[2][o]) + part[3][o];
    *dst = accumulate ? *dst + t : t;
  }
}

int bwd_blocks(int64_t N) {
  int ntiles = cdiv(N, 32);
  return ntiles < 256 ? ntiles : 256;
}

}  // namespace

#ifdef GEOBI_HEAD_STAMPS
extern "C" int geobi_debug_head_stamps(void* host_dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_head_stamps), bytes, 0, hipMemcpyDeviceToHost);
}
#endif

bool head_fused_supported(int Cin, int K, int nout) { return Cin == CIN && K == HID && (nout == 1 || nout == 3); }

// 0: exact fp32 MFMA (default); 1: the first GEMM of the heads as six bf16 products (head_fwd_fused_bf16x3_kernel)
static std::atomic<int> g_head_precision{[] { const char* e = getenv("GEOBI_HEAD_BF16X3"); return (e && atoi(e) != 0) ? 1 : 0; }()};
int set_head_precision(int mode) {
  GEOBI_REQUIRE(mode == 0 || mode == 1, "head precision: 0 (fp32) or 1 (3 x bf16 split, six products)");
  g_head_precision = mode;
  return 0;
}

int head_fwd_fused(const float* x, int64_t N, const float* w1, const float* b1, const float* w2, const float* b2,
                   int nout, float slope, int mode, const float* dd, const float* resid, int ld_resid, float* raw,
                   float* out, hipStream_t s) {
  int blocks = cdiv(N, 32);
  if (g_head_precision.load(std::memory_order_relaxed) == 1) {
    // the pieces of W1 (192 KB) live in a per-context buffer the library owns (the one allocation of this mode); they are
    // re-formed on every call, on the call's stream: the weights move with every optimiser step
    static thread_local __bf16* pack = nullptr;
    if (pack == nullptr) GEOBI_HIP(hipMalloc((void**)&pack, (size_t)3 * HID * CIN * sizeof(__bf16)));
    head_split_w1_kernel<<<cdiv(HID * CIN, 256), 256, 0, s>>>(w1, pack);
    if (nout == 3)
      head_fwd_fused_bf16x3_kernel<3><<<blocks, 256, 0, s>>>(x, (int)N, pack, b1, w2, b2, slope, mode, dd, resid, ld_resid,
                                                             raw, out);
    else
      head_fwd_fused_bf16x3_kernel<1><<<blocks, 256, 0, s>>>(x, (int)N, pack, b1, w2, b2, slope, mode, dd, resid, ld_resid,
                                                             raw, out);
    GEOBI_LAUNCH_OK();
    return 0;
  }
  if (nout == 3)
    head_fwd_fused_kernel<3><<<blocks, 256, 0, s>>>(x, (int)N, w1, b1, w2, b2, slope, mode, dd, resid, ld_resid, raw,
                                                    out);
  else
    head_fwd_fused_kernel<1><<<blocks, 256, 0, s>>>(x, (int)N, w1, b1, w2, b2, slope, mode, dd, resid, ld_resid, raw,
                                                    out);
  GEOBI_LAUNCH_OK();
  return 0;
}

size_t head_bwd_fused_ws_bytes(int64_t N) {
  size_t b = bwd_blocks(N);
  return align_up(b * (size_t)NCHUNK * 16 * 64 * sizeof(float)) + align_up(b * 3 * HID * sizeof(float)) +
         align_up(b * HID * sizeof(float)) + align_up(b * 4 * sizeof(float)) + 1024;
}

// graw [N, nout] is the gradient w.r.t. the pre-finish head output (head_finish_bwd in geom.hip).
int head_bwd_fused(const float* x, int64_t N, const float* w1, const float* b1, const float* w2, int nout,
                   float slope, const float* graw, float* dx, float* dw1, float* db1, float* dw2, float* db2,
                   int accumulate, void* ws, size_t ws_bytes, hipStream_t s) {
  const int blocks = bwd_blocks(N);
  Arena a(ws, ws_bytes);
  float* p_dw1 = a.take<float>((size_t)blocks * NCHUNK * 16 * 64);
  float* p_dw2 = a.take<float>((size_t)blocks * 3 * HID);
  float* p_db1 = a.take<float>((size_t)blocks * HID);
  float* p_db2 = a.take<float>((size_t)blocks * 4);
  GEOBI_REQUIRE(a.ok() && ws, "head_bwd_fused: workspace too small (%zu < %zu)", ws_bytes, a.off);
  const int ntiles = cdiv(N, 32);
  constexpr size_t kLds = (size_t)(NCHUNK * 16 * 64 + 3 * HID + HID + HBW * 16 * 32) * sizeof(float);   // 160 KiB
  static_assert(kLds == 163840, "the backward head kernel uses the whole LDS of a CU");
  static std::atomic<bool> attr_set{false};   // several host threads may launch (one per mesh group)
  if (!attr_set) {
    GEOBI_HIP(hipFuncSetAttribute((const void*)head_bwd_fused_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)kLds));
    GEOBI_HIP(hipFuncSetAttribute((const void*)head_bwd_fused_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)kLds));
    attr_set = true;
  }
  if (nout == 3)
    head_bwd_fused_kernel<3><<<blocks, 64 * HBW, kLds, s>>>(x, (int)N, ntiles, w1, b1, w2, slope, graw, dx, p_dw1, p_dw2,
                                                       p_db1, p_db2);
  else
    head_bwd_fused_kernel<1><<<blocks, 64 * HBW, kLds, s>>>(x, (int)N, ntiles, w1, b1, w2, slope, graw, dx, p_dw1, p_dw2,
                                                       p_db1, p_db2);
  GEOBI_LAUNCH_OK();
  const int total = HID * CIN + HID + nout * HID + nout;
  head_bwd_reduce_kernel<<<cdiv(total, 64), 256, 0, s>>>(p_dw1, p_dw2, p_db1, p_db2, blocks, nout, accumulate, dw1, db1,
                                                          dw2, db2);
  GEOBI_LAUNCH_OK();
  return 0;
}

}  // namespace geobi
