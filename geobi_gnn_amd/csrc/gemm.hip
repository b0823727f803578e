// fp32 MFMA GEMMs for the dense per-node transforms of the FeaSt layers and the two heads.
//
// v_mfma_f32_32x32x2_f32 is exact fp32 (a k-ordered fma chain), which the 1e-5 parity bar
// against the fp32 reference needs; there is no reduced-precision path.
//   gemm_nn : C[M,N] = A[M,K] * B[K,N] (+bias, leaky-relu, optional column-split output)
//             B may be given transposed ([N,K] row-major, e.g. nn.Linear.weight).
//   gemm_tn : C[I,J] = sum_m A[m,I] * B[m,J]   (weight gradients; the reduction runs over the
//             node dimension, split across workgroups into partial slabs that a second kernel
//             adds in a fixed order -> deterministic, no float atomics).
//   colsum  : column sums (bias / c gradients), same two-stage scheme.
#include "common.h"

namespace geobi {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));


// A fragment for 32x32x2: lane l holds A[i = l & 31][k = l >> 5]; B fragment: B[k = l >> 5][j = l & 31].
// C/D: col = l & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5).
// FAST: every operand is 16-B aligned with leading dimensions, K (and N for a non-transposed B)
// multiples of 4 -- true for every GEMM the model issues.  The staging loads are then branch-free
// (clamped address + select), which lets the compiler keep them in flight across the MFMA section
// instead of waiting at the end of each bounds-check branch.
template <int WAVES_M, int WAVES_N, int TM, int TN, int BK, bool FAST>
__global__ __launch_bounds__(256) void gemm_nn_kernel(const float* __restrict__ A, int lda,
                                                      const float* __restrict__ B, int ldb, int transB,
                                                      float* __restrict__ C, int ldc, int M, int N, int K,
                                                      const float* __restrict__ bias, float slope,
                                                      float* __restrict__ C1, int split, int ldc1, int k_chunk,
                                                      float* __restrict__ partial, int wide_store) {
  // split-K (partial != nullptr): blockIdx.z owns k in [z * k_chunk, (z+1) * k_chunk) and stores the raw
  // accumulators to partial[z][M][N]; splitk_reduce_kernel adds the slices in order + epilogue.
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per block");
  constexpr int BM = WAVES_M * TM * 32;
  constexpr int BN = WAVES_N * TN * 32;
  constexpr int LDA_S = BM + 1;
  constexpr int LDB_S = BN + 4;
  // one LDS block: the staging tiles, re-used by the wide-store epilogue once the k loop is over
  constexpr int TILE_FLOATS = BK * LDA_S + BK * LDB_S;
  constexpr int EPI_FLOATS = 4 * 32 * 36;
  __shared__ __attribute__((aligned(16))) float smem[TILE_FLOATS > EPI_FLOATS ? TILE_FLOATS : EPI_FLOATS];
  float (*As)[LDA_S] = reinterpret_cast<float (*)[LDA_S]>(smem);
  float (*Bs)[LDB_S] = reinterpret_cast<float (*)[LDB_S]>(smem + BK * LDA_S);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const bool a_vec = ((lda & 3) == 0) && ((((uintptr_t)A) & 15) == 0);
  const bool b_vec = ((ldb & 3) == 0) && ((((uintptr_t)B) & 15) == 0);

  // Software pipeline: the global loads of k-tile t+1 are issued into registers before the MFMAs
  // of tile t run out of LDS, so HBM/L2 latency overlaps the matrix pipe (single LDS buffer,
  // two barriers per tile).
  constexpr int KQ = BK / 4;                               // float4 per A-tile row
  constexpr int A_ROWS = 256 / KQ;                         // A rows staged per pass
  constexpr int A_PASSES = (BM + A_ROWS - 1) / A_ROWS;
  constexpr int B_Q = (BK * BN / 4 + 255) / 256;          // float4 per thread for the B tile
  float ra[A_PASSES][4];
  float rb[B_Q][4];

  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int pass = 0; pass < A_PASSES; ++pass) {
      int row = tid / KQ + pass * A_ROWS;
      int kq = (tid % KQ) * 4;
      int gm = m0 + row, gk = k0 + kq;
#pragma unroll
      for (int i = 0; i < 4; ++i) ra[pass][i] = 0.f;
      if constexpr (FAST) {
        // raw load from a clamped (always valid) address; the out-of-range select happens in
        // store_tiles, i.e. after the MFMA section, so nothing waits on this load before the MFMAs
        const float* src = A + (size_t)min(gm, M - 1) * lda + min(gk, K - 4);
        float4 t = *reinterpret_cast<const float4*>(src);
        ra[pass][0] = t.x; ra[pass][1] = t.y; ra[pass][2] = t.z; ra[pass][3] = t.w;
      } else if (row < BM && gm < M) {
        const float* src = A + (size_t)gm * lda + gk;
        if (a_vec && gk + 3 < K) {
          float4 t = *reinterpret_cast<const float4*>(src);
          ra[pass][0] = t.x; ra[pass][1] = t.y; ra[pass][2] = t.z; ra[pass][3] = t.w;
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (gk + i < K) ra[pass][i] = src[i];
        }
      }
    }
#pragma unroll
    for (int qi = 0; qi < B_Q; ++qi) {
      int q = tid + qi * 256;
#pragma unroll
      for (int i = 0; i < 4; ++i) rb[qi][i] = 0.f;
      if (q >= BK * BN / 4) continue;
      if (!transB) {
        constexpr int QPR = BN / 4;
        int kk = q / QPR, nq = (q % QPR) * 4;
        int gk = k0 + kk, gn = n0 + nq;
        if constexpr (FAST) {
          const float* src = B + (size_t)min(gk, K - 1) * ldb + min(gn, N - 4);
          float4 t = *reinterpret_cast<const float4*>(src);
          rb[qi][0] = t.x; rb[qi][1] = t.y; rb[qi][2] = t.z; rb[qi][3] = t.w;
        } else if (gk < K) {
          const float* src = B + (size_t)gk * ldb + gn;
          if (b_vec && gn + 3 < N) {
            float4 t = *reinterpret_cast<const float4*>(src);
            rb[qi][0] = t.x; rb[qi][1] = t.y; rb[qi][2] = t.z; rb[qi][3] = t.w;
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (gn + i < N) rb[qi][i] = src[i];
          }
        }
      } else {
        int nn = q / KQ, kq = (q % KQ) * 4;
        int gn = n0 + nn, gk = k0 + kq;
        if constexpr (FAST) {
          const float* src = B + (size_t)min(gn, N - 1) * ldb + min(gk, K - 4);
          float4 t = *reinterpret_cast<const float4*>(src);
          rb[qi][0] = t.x; rb[qi][1] = t.y; rb[qi][2] = t.z; rb[qi][3] = t.w;
        } else if (gn < N) {
          const float* src = B + (size_t)gn * ldb + gk;
          if (b_vec && gk + 3 < K) {
            float4 t = *reinterpret_cast<const float4*>(src);
            rb[qi][0] = t.x; rb[qi][1] = t.y; rb[qi][2] = t.z; rb[qi][3] = t.w;
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (gk + i < K) rb[qi][i] = src[i];
          }
        }
      }
    }
  };

  auto store_tiles = [&](int k0) {      // k0: the k-tile the registers were loaded for
#pragma unroll
    for (int pass = 0; pass < A_PASSES; ++pass) {
      int row = tid / KQ + pass * A_ROWS;
      int kq = (tid % KQ) * 4;
      if (row < BM) {
        const bool ok = !FAST || (m0 + row < M && k0 + kq < K);
#pragma unroll
        for (int i = 0; i < 4; ++i) As[kq + i][row] = ok ? ra[pass][i] : 0.f;
      }
    }
#pragma unroll
    for (int qi = 0; qi < B_Q; ++qi) {
      int q = tid + qi * 256;
      if (q >= BK * BN / 4) continue;
      if (!transB) {
        constexpr int QPR = BN / 4;
        int kk = q / QPR, nq = (q % QPR) * 4;
        const bool ok = !FAST || (k0 + kk < K && n0 + nq < N);
#pragma unroll
        for (int i = 0; i < 4; ++i) Bs[kk][nq + i] = ok ? rb[qi][i] : 0.f;
      } else {
        int nn = q / KQ, kq = (q % KQ) * 4;
        const bool ok = !FAST || (n0 + nn < N && k0 + kq < K);
#pragma unroll
        for (int i = 0; i < 4; ++i) Bs[kq + i][nn] = ok ? rb[qi][i] : 0.f;
      }
    }
  };

  const int k_lo = partial ? blockIdx.z * k_chunk : 0;
  const int k_hi = partial ? min(K, k_lo + k_chunk) : K;
  load_tiles(k_lo);
  for (int k0 = k_lo; k0 < k_hi; k0 += BK) {
    store_tiles(k0);
    __syncthreads();
    if (k0 + BK < k_hi) load_tiles(k0 + BK);
    // ---- MFMA over the k-tile
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float a[TM], b[TN];
      const int kr = kk + (lane >> 5);
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[kr][(wm * TM + i) * 32 + (lane & 31)];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Bs[kr][(wn * TN + j) * 32 + (lane & 31)];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- epilogue
  if (wide_store) {
    // Row-major 16-B stores: each accumulator tile (column on the lane, rows in registers) is turned
    // through a per-wave LDS tile so that a store instruction writes 8 rows x 128 contiguous bytes
    // (dwordx4 per lane) instead of 2 x 128 B with one dword per lane -- 4x fewer store instructions
    // on the output-bound GEMMs (dz = g Wf^T writes 9*Cin floats per node).
    float (*Es)[32][36] = reinterpret_cast<float (*)[32][36]>(smem);     // the k loop ended on a barrier
    const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) Es[wave][(r & 3) + 8 * (r >> 2) + 4 * half][l31] = acc[i][j][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int c4 = (lane & 7) * 4;
        const int col = n0 + (wn * TN + j) * 32 + c4;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias && col < N) bv = *reinterpret_cast<const float4*>(bias + col);
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
          const int rr = pass * 8 + (lane >> 3);
          const int row = m0 + (wm * TM + i) * 32 + rr;
          float4 v = *reinterpret_cast<const float4*>(&Es[wave][rr][c4]);
          v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
          v.x = v.x > 0.f ? v.x : v.x * slope; v.y = v.y > 0.f ? v.y : v.y * slope;
          v.z = v.z > 0.f ? v.z : v.z * slope; v.w = v.w > 0.f ? v.w : v.w * slope;
          if (row < M && col < N) *reinterpret_cast<float4*>(C + (size_t)row * ldc + col) = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      int col = n0 + (wn * TN + j) * 32 + (lane & 31);
      if (col >= N) continue;
      float bv = bias ? bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row >= M) continue;
        if (partial) {
          partial[((size_t)blockIdx.z * M + row) * N + col] = acc[i][j][r];
          continue;
        }
        float v = acc[i][j][r] + bv;
        v = v > 0.f ? v : v * slope;
        if (C1 != nullptr && col >= split)
          C1[(size_t)row * ldc1 + (col - split)] = v;
        else
          C[(size_t)row * ldc + col] = v;
      }
    }
}

__global__ void splitk_reduce_kernel(const float* __restrict__ partial, int slices, int M, int N,
                                     float* __restrict__ C, int ldc, const float* __restrict__ bias, float slope,
                                     float* __restrict__ C1, int split, int ldc1) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)M * N) return;
  float v = 0.f;
  for (int z = 0; z < slices; ++z) v += partial[(size_t)z * M * N + idx];
  int row = (int)(idx / N), col = (int)(idx % N);
  if (bias) v += bias[col];
  v = v > 0.f ? v : v * slope;
  if (C1 != nullptr && col >= split)
    C1[(size_t)row * ldc1 + (col - split)] = v;
  else
    C[(size_t)row * ldc + col] = v;
}

// ------------------------------------------------------------------------------ TN
// Where output element (i, j) of a TN product goes (the unpacking that used to be a kernel of its own).
__device__ __forceinline__ void tn_emit(const TnOutput& o, int i, int j, int I, int J, float s) {
  float* dst = nullptr;
  if (o.mode == TN_PLAIN) {
    // an implicit ones row / column (always the last one) lands in C2
    if (o.extra_row && i == I - 1) { if (!(o.extra_col && j == J - 1)) dst = o.C2 + j; }
    else if (o.extra_col && j == J - 1) dst = o.C2 + i;
    else dst = o.C + (size_t)i * o.ldc + j;
  } else if (o.mode == TN_LIN_UNPACK) {
    // rows are (head h, in-channel k) of the packed weight, columns are out-channels:
    // lin.weight[h * Cout + o, k]  (FeaStConv `lin.weight [H*Cout, Cin]`); the implicit ones row
    // (i == I - 1) carries the bias gradient
    if (i == I - 1) dst = o.C2 + j;
    else {
      int h = i / o.Cin, k = i % o.Cin;
      if (h < GEOBI_H) dst = o.C + ((size_t)h * o.Cout + j) * o.Cin + k;
    }
  } else if (o.mode == TN_RPRIME) {
    // rows: input channels (+ the column-sums row when extra_row), columns: r' = [r (9 Cout) | dp 9 + 3 | dcs 9 + 3]
    //   data row i, j < 9 Cout            -> lin.weight.grad[j, col0 + i]        (j = h Cout + o)
    //   data row i, 9 Cout <= j < +9      -> u.weight.grad[j - 9 Cout, col0 + i]
    //   sums row,   j in the dcs columns  -> c.grad[h]
    //   sums row,   j < 9 Cout            -> bias.grad[c] = sum_h (sums row)[h Cout + c]: the extra block of
    //                                        tn_reduce_kernel (sum_h sum_j q_ijh / deg_i = 1: the r columns of a channel
    //                                        add up to the column sum of g)
    const int HC = GEOBI_H * o.Cout;
    const bool sums = o.extra_row && i == I - 1;
    if (!sums) {
      if (j < HC) dst = o.C + ((size_t)j * o.Cin + o.col0 + i);
      else if (j < HC + GEOBI_H && o.C3 != nullptr) dst = o.C3 + ((size_t)(j - HC) * o.Cin + o.col0 + i);
    } else if (j >= HC + GEOBI_HP && j < HC + GEOBI_HP + GEOBI_H) {
      if (o.C4 != nullptr) dst = o.C4 + (j - HC - GEOBI_HP);
    }
  } else {
    // TN_DU_DC: A = [dp (rows 0..8) | pad | dcs (rows 12..20) | pad], B = [x | 1]
    //   du[h, col0 + j] = row h, j < J-1 ;   dc[h] = row 12+h, j == J-1
    if (j < J - 1) { if (i < GEOBI_H) dst = o.C + (size_t)i * o.ldc + j; }
    else if (o.C2 != nullptr && i >= GEOBI_HP && i < GEOBI_HP + GEOBI_H) dst = o.C2 + (i - GEOBI_HP);
  }
  if (dst != nullptr) *dst = o.accumulate ? *dst + s : s;
}

// Each wave owns a (32*TI) x (32*TJ) output tile and a slice of the reduction (node) range.
// Operands are read straight from global memory in MFMA fragment order: for a k-step of two
// consecutive nodes, lanes 0-31 read 32 consecutive floats of node m, lanes 32-63 of node m+1.
// TN kernel.  Every load is unconditional from a clamped (always valid) address and nothing is selected
// at load time, so the compiler can count the loads of the NEXT stage as outstanding (s_waitcnt vmcnt(n))
// while the MFMAs of the current one run; masks (node tail, implicit ones row / column) are applied to the
// registers right before the MFMAs.  Garbage in lanes whose column lies beyond I (or J) only reaches
// accumulator rows (columns) that are never stored.
//   VA: the lane loads A[m][i0 + 4 l .. + 3] as ONE 16-B load and MFMA tile a takes component a, i.e.
//       tile a owns output rows i0 + 4 r + a (undone at the write-out).  Needs lda, the data column count
//       and the base address to be multiples of 4 floats.  Otherwise tile a owns rows i0 + 32 a + r and
//       loads one dword per tile.
template <int TI, int TJ, int U, bool VA>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const float* __restrict__ A, int lda,
                                                      const float* __restrict__ B, int ldb, int64_t M, int I,
                                                      int J, int ones_row, int ones_col, int tiles_j,
                                                      int64_t m_per_slice, float* __restrict__ slabs, int sum_row) {
  // I, J are the LOGICAL output sizes; row `ones_row` of A^T / column `ones_col` of B read as 1
  // (bias-style column sums ride along in the same pass); pass -1 to disable.  They are always the
  // LAST logical row / column.
  // The 4 waves of a block own 4 consecutive node slices of the SAME output tile and fold their
  // accumulators through LDS, so one slab is written per block (4x fewer slabs to reduce).
  // sum_row >= 0 (== I - 1): logical row I - 1 of the product is the COLUMN SUMS of B -- what an implicit ones row of
  // A^T would give, but formed on the VALU from the B values the waves stream anyway (TJ adds per TI x TJ MFMAs; as
  // an MFMA row it cost a whole padded 32-row tile).  Only the blocks of the first row tile form them.
  // (Tried in round 3: the block that arrives LAST at its tile -- a ticket behind an agent-scope release -- adds the
  // tile's slabs itself, no tn_reduce launch.  3 x slower: 5-20 tiles x 1 block then do the work of the I J / 64 blocks of
  // tn_reduce_kernel, 150 us against 46 + 16 us per product.  profiles/r03_side_stream.txt)
  static_assert(!VA || TI == 4, "vector A loads feed 4 row tiles");
  __shared__ float red[2][TI * TJ * 16 * 64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // scalar: node rows stay in SGPRs
  const int tile = blockIdx.x;
  const int ti = tile / tiles_j, tj = tile % tiles_j;
  const int slice = blockIdx.y * 4 + wave;
  const int64_t m_begin = (int64_t)slice * m_per_slice;
  int64_t m_end = m_begin + m_per_slice;
  if (m_end > M) m_end = M;
  const int i0 = ti * 32 * TI, j0 = tj * 32 * TJ;
  const int Id = I - ((ones_row >= 0 || sum_row >= 0) ? 1 : 0), Jd = J - (ones_col >= 0 ? 1 : 0);   // columns that exist in memory
  const bool do_sum = sum_row >= 0 && ti == 0;                    // block-uniform
  float csum[TJ];
#pragma unroll
  for (int b = 0; b < TJ; ++b) csum[b] = 0.f;
  const int half = lane >> 5, l31 = lane & 31;

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int a = 0; a < TI; ++a)
#pragma unroll
    for (int b = 0; b < TJ; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // per-lane column offsets (clamped into the matrix) and the lanes that carry the implicit ones
  int ia[VA ? 1 : TI], jb[TJ];
  bool a_one[TI], b_one[TJ];
#pragma unroll
  for (int a = 0; a < TI; ++a) {
    const int ii = VA ? i0 + 4 * l31 + a : i0 + 32 * a + l31;
    a_one[a] = ii == ones_row;
    if (VA) { if (a == 0) ia[0] = min(ii, Id - 4); }
    else ia[a] = min(ii, Id - 1);
  }
#pragma unroll
  for (int b = 0; b < TJ; ++b) {
    const int jj = j0 + 32 * b + l31;
    b_one[b] = jj == ones_col;
    jb[b] = min(jj, Jd - 1);
  }
  const bool ones_a_here = ones_row >= i0 && ones_row < i0 + 32 * TI;     // wave-uniform
  const bool ones_b_here = ones_col >= j0 && ones_col < j0 + 32 * TJ;

  float a0[U][TI], b0[U][TJ], a1[U][TI], b1[U][TJ];

  // Addresses: the node row m + 2u is wave-uniform (scalar base pointer, clamped to the last row); the
  // lane adds its column and, for the odd node of the pair, one row -- 32-bit offsets, ~1 VALU op per load.
  const int half_lda = half ? lda : 0, half_ldb = half ? ldb : 0;
  auto load = [&](float (&av)[U][TI], float (&bv)[U][TJ], int64_t m) {
    const int64_t mb = m < M ? m : M - 1;                      // stage base row, 64-bit once per stage
    const int64_t left = M - 1 - mb;
    const int last = left > 4 * U ? 4 * U : (int)left;         // rows available past the base, 32-bit
    const float* as = A + mb * lda;
    const float* bs = B + mb * ldb;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int rel = 2 * u < last ? 2 * u : last;             // clamped to the last row
      const bool odd_ok = 2 * u + 1 <= last;                   // uniform: the pair's odd node exists
      const float* ap = as + rel * lda;
      const float* bp = bs + rel * ldb;
      const int da = odd_ok ? half_lda : 0, db = odd_ok ? half_ldb : 0;
      if constexpr (VA) {
        const float4 t = *reinterpret_cast<const float4*>(ap + (ia[0] + da));
        av[u][0] = t.x; av[u][1] = t.y; av[u][2] = t.z; av[u][3] = t.w;
      } else {
#pragma unroll
        for (int a = 0; a < TI; ++a) av[u][a] = ap[ia[a] + da];
      }
#pragma unroll
      for (int b = 0; b < TJ; ++b) bv[u][b] = bp[jb[b] + db];
    }
  };
  auto mma = [&](float (&av)[U][TI], float (&bv)[U][TJ], int64_t m) {
    const bool full = m + 2 * U <= m_end;                     // wave-uniform: no node mask needed
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float x[TI], y[TJ];
#pragma unroll
      for (int a = 0; a < TI; ++a) x[a] = av[u][a];
#pragma unroll
      for (int b = 0; b < TJ; ++b) y[b] = bv[u][b];
      if (ones_a_here) {
#pragma unroll
        for (int a = 0; a < TI; ++a) x[a] = a_one[a] ? 1.0f : x[a];
      }
      if (ones_b_here) {
#pragma unroll
        for (int b = 0; b < TJ; ++b) y[b] = b_one[b] ? 1.0f : y[b];
      }
      if (!full) {
        const bool ok = m + 2 * u + half < m_end;
#pragma unroll
        for (int a = 0; a < TI; ++a) x[a] = ok ? x[a] : 0.f;
#pragma unroll
        for (int b = 0; b < TJ; ++b) y[b] = ok ? y[b] : 0.f;
      }
      if (do_sum) {
#pragma unroll
        for (int b = 0; b < TJ; ++b) csum[b] += y[b];
      }
#pragma unroll
      for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int b = 0; b < TJ; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[a], y[b], acc[a][b], 0, 0, 0);
    }
  };

  // two register stages of 2U nodes: the loads of the next stage are in flight while this one multiplies
  if (m_begin < m_end) {
    load(a0, b0, m_begin);
    for (int64_t m = m_begin; m < m_end; m += 4 * U) {
      load(a1, b1, m + 2 * U);
      mma(a0, b0, m);
      load(a0, b0, m + 4 * U);
      if (m + 2 * U < m_end) mma(a1, b1, m + 2 * U);
    }
  }

  // ---- fold the 4 waves: (2,3) -> LDS -> (0,1) add; 1 -> LDS -> 0 adds and writes the slab
  if (wave >= 2) {
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
      for (int b = 0; b < TJ; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave - 2][((a * TJ + b) * 16 + r) * 64 + lane] = acc[a][b][r];
  }
  __syncthreads();
  if (wave < 2) {
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
      for (int b = 0; b < TJ; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] += red[wave][((a * TJ + b) * 16 + r) * 64 + lane];
  }
  __syncthreads();
  if (wave == 1) {
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
      for (int b = 0; b < TJ; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[0][((a * TJ + b) * 16 + r) * 64 + lane] = acc[a][b][r];
  }
  __syncthreads();
  float* out = slabs + (size_t)blockIdx.y * I * J;
  if (wave == 0) {
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
      for (int b = 0; b < TJ; ++b) {
        int jj = j0 + b * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[a][b][r] + red[0][((a * TJ + b) * 16 + r) * 64 + lane];
          const int rr = (r & 3) + 8 * (r >> 2) + 4 * half;
          const int ii = VA ? i0 + 4 * rr + a : i0 + a * 32 + rr;
          if (jj < J && ii < Id + (ones_row >= 0 ? 1 : 0)) out[(size_t)ii * J + jj] = v;
        }
      }
  }
  // ---- column sums of B: the two nodes of a pair (lane halves), then the four waves in wave order
  if (do_sum) {
    __syncthreads();                                   // red[] is free again
#pragma unroll
    for (int b = 0; b < TJ; ++b) {
      const float t = csum[b] + __shfl_xor(csum[b], 32, 64);
      if (half == 0) red[0][(wave * TJ + b) * 32 + l31] = t;
    }
    __syncthreads();
    if (wave == 0 && half == 0) {
#pragma unroll
      for (int b = 0; b < TJ; ++b) {
        const int jj = j0 + b * 32 + l31;
        float t = 0.f;
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) t += red[0][(w4 * TJ + b) * 32 + l31];
        if (jj < J) out[(size_t)sum_row * J + jj] = t;
      }
    }
  }
}

// 64 outputs per block, four threads per output: thread (e, q) adds slabs q, q + 4, ... (independent loads, eight in
// flight), the four partial sums meet in LDS in the order q = 0..3 (fixed: deterministic).  One thread per output
// walking all slabs left a 64 x 312 product with 78 blocks of serial 4-byte loads: 18 us per launch.
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ slabs, int slices, int I, int J, TnOutput o) {
  __shared__ float part[4][64];
  const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
  int idx = blockIdx.x * 64 + e;
  const int nbias = (o.mode == TN_RPRIME && o.extra_row && o.C2 != nullptr) ? (o.Cout + 7) / 8 : 0;
  if (nbias > 0 && (int)blockIdx.x >= (int)gridDim.x - nbias) {
    // extra blocks: bias.grad[c] = sum_h (sums row)[h Cout + c], eight channels per block.  72 (head, channel) sums,
    // three threads each (slabs k = part, part + 3, ...: three chains of independent loads; one thread per sum walking
    // every slab in ONE block for all channels made this the longest block of the launch), folded in the fixed order
    // part 0..2, then heads 0..8 per channel.
    __shared__ float psum[3][GEOBI_H * 8];
    const int c0 = ((int)blockIdx.x - ((int)gridDim.x - nbias)) * 8;
    const int pair = threadIdx.x / 3, part = threadIdx.x % 3;          // pair = h * 8 + cl
    if (pair < GEOBI_H * 8) {
      const int h = pair >> 3, c = c0 + (pair & 7);
      float t = 0.f;
      if (c < o.Cout) {
        const float* base = slabs + (size_t)(I - 1) * J + h * o.Cout + c;
#pragma unroll 8
        for (int k = part; k < slices; k += 3) t += base[(size_t)k * I * J];
      }
      psum[part][pair] = t;
    }
    __syncthreads();
    if (threadIdx.x < 8 && c0 + (int)threadIdx.x < o.Cout) {
      const int c = c0 + threadIdx.x;
      float t = 0.f;
      for (int h = 0; h < GEOBI_H; ++h) t += (psum[0][h * 8 + threadIdx.x] + psum[1][h * 8 + threadIdx.x]) + psum[2][h * 8 + threadIdx.x];
      o.C2[c] = o.accumulate ? o.C2[c] + t : t;
    }
    return;
  }
  float sp = 0.f;
  if (idx < I * J) {
#pragma unroll 8
    for (int k = q; k < slices; k += 4) sp += slabs[(size_t)k * I * J + idx];
  }
  part[q][e] = sp;
  __syncthreads();
  if (q != 0 || idx >= I * J) return;
  const float s = ((part[0][e] + part[1][e]) + part[2][e]) + part[3][e];
  tn_emit(o, idx / J, idx % J, I, J, s);
}

__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ A, int lda, int64_t M, int J,
                                                             int64_t rows_per_block, float* __restrict__ partial) {
  // block (bx, by) sums rows [bx*rpb, (bx+1)*rpb) of the column block by*cols .. ; the 256 threads
  // form a (rows_par x cols) grid so narrow matrices (J = 9..32) still use the whole block.
  __shared__ float red[256];
  const int cols = J < 256 ? J : 256;
  const int rows_par = 256 / cols;
  const int rr = threadIdx.x / cols, jc = threadIdx.x % cols;
  const int j = blockIdx.y * 256 + jc;
  int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > M) r1 = M;
  float s = 0.f;
  if (rr < rows_par && j < J)
    for (int64_t r = r0 + rr; r < r1; r += rows_par) s += A[r * lda + j];
  red[threadIdx.x] = s;
  __syncthreads();
  if (rr == 0 && j < J) {
    float t = 0.f;
    for (int q = 0; q < rows_par; ++q) t += red[q * cols + jc];
    partial[(size_t)blockIdx.x * J + j] = t;
  }
}

__global__ void colsum_final_kernel(const float* __restrict__ partial, int blocks, int J, float* __restrict__ out) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= J) return;
  float s = 0.f;
  for (int b = 0; b < blocks; ++b) s += partial[(size_t)b * J + j];
  out[j] = s;
}

struct TnPlan {
  int ti, tj, va, u, tiles_i, tiles_j, slices, blocks_y;
  int64_t m_per_slice;
};

TnPlan plan_tn(const float* A, int lda, int I, int J, int64_t M, int ones_row) {
  TnPlan p;
  // per-wave tile = (32 TI) x (32 TJ): as large as the shape fills (MFMA work is padded to tiles)
  int ci = cdiv(I, 32), cj = cdiv(J, 32);
  const int Id = I - (ones_row >= 0 ? 1 : 0);
  p.va = ci >= 3 && (lda & 3) == 0 && (Id & 3) == 0 && Id >= 4 && (((uintptr_t)A) & 15) == 0;
  if (p.va) {
    p.ti = 4;                           // one 16-B load feeds 4 row tiles; wider column tiles did not pay
    p.tj = 1;                           // (1 wave/SIMD at 4x2 accumulators), see tools/tn_probe.hip
  } else {
    p.ti = ci >= 2 ? 2 : 1;
    p.tj = cj >= 4 ? 2 : cj;            // 1, 2 or 3 column tiles per wave; >= 4 -> pairs
  }
  p.u = 4;
  p.tiles_i = cdiv(I, 32 * p.ti);
  p.tiles_j = cdiv(J, 32 * p.tj);
  int tiles = p.tiles_i * p.tiles_j;
  // One to two blocks per CU in ONE resident round (a second, partial round would idle most CUs for a
  // whole block time; more, shorter slices only add slab traffic), each wave taking >= 64 nodes.
  static const int64_t cap_env = [] { const char* f = getenv("GEOBI_TN_CAP"); return f ? atoll(f) : 0ll; }();
  // GEOBI_TN_CAP: tuning knob (tools/tn_cap_sweep.py).  Few output tiles (du/dc, narrow dWf): the slab
  // reduction is a chain of dependent loads per output, so half the slabs measured ~15 % faster overall.
  const int64_t capacity = cap_env > 0 ? cap_env : (tiles <= 3 ? 256 : 512);
  int64_t by = capacity / (tiles > 0 ? tiles : 1);
  if (by < 1) by = 1;
  int64_t max_by = (M + 255) / 256;                  // 4 waves x 64 nodes per block at least
  if (by > max_by) by = max_by;
  if (by < 1) by = 1;
  const int stage = 2 * p.u;
  int64_t mps = (M + 4 * by - 1) / (4 * by);
  p.m_per_slice = (mps + stage - 1) / stage * stage;  // whole stages
  p.blocks_y = (int)((M + 4 * p.m_per_slice - 1) / (4 * p.m_per_slice));
  if (p.blocks_y < 1) p.blocks_y = 1;
  p.slices = 4 * p.blocks_y;
  return p;
}

}  // namespace

int gemm_nn(const float* A, int lda, const float* B, int ldb, int transB, float* C, int ldc, int M, int N, int K,
            const GemmEpilogue& ep, hipStream_t s) {
  if (M <= 0 || N <= 0) return 0;
  GEOBI_REQUIRE(K > 0, "gemm_nn: K must be positive");
  prof_begin(PROF_GEMM, s, 2.0 * M * N * K, 1);          // "bytes" carries the flop count here
  dim3 block(256);
  const bool fast = ((lda & 3) == 0) && ((ldb & 3) == 0) && ((K & 3) == 0) && K >= 4 &&
                    ((((uintptr_t)A) | ((uintptr_t)B)) & 15) == 0 && (transB || ((N & 3) == 0 && N >= 4));
#define GEOBI_GEMM_LAUNCH(WM, WN, TM_, TN_, BK_)                                                                    \
  do {                                                                                                          \
    if (fast) { GEOBI_GEMM_LAUNCH_F(WM, WN, TM_, TN_, BK_, true); } else { GEOBI_GEMM_LAUNCH_F(WM, WN, TM_, TN_, BK_, false); } \
  } while (0)
#define GEOBI_GEMM_LAUNCH_F(WM, WN, TM_, TN_, BK_, FAST_)                                                          \
  do {                                                                                                          \
    constexpr int BM_ = WM * TM_ * 32;                                                                          \
    constexpr int BN_ = WN * TN_ * 32;                                                                          \
    dim3 grid(cdiv(M, BM_), cdiv(N, BN_), slices);                                                              \
    gemm_nn_kernel<WM, WN, TM_, TN_, BK_, FAST_><<<grid, block, 0, s>>>(A, lda, B, ldb, transB, C, ldc, M, N, K, \
                                                                ep.bias,                                        \
                                                           ep.slope, ep.C1, ep.split, ep.ldc1, k_chunk, partial,  \
                                                           wide);                                                \
  } while (0)
  // Few rows (coarse graph levels): shrink the block tile so the grid still covers the 256 CUs, and
  // split K across blockIdx.z when even the small tiles leave CUs idle.
  const int64_t big_blocks = (int64_t)cdiv(M, 128) * cdiv(N, N > 64 ? 128 : (N > 32 ? 64 : 32));
  int slices = 1, k_chunk = K;
  float* partial = nullptr;
  const int64_t blocks_small = N <= 32 ? (int64_t)cdiv(M, 128) : (int64_t)cdiv(M, 64) * cdiv(N, 64);   // tiles below
  if (ep.fixed_slices > 0) {
    if (ep.fixed_slices > 1 && ep.ws != nullptr) {
      k_chunk = ((K + ep.fixed_slices - 1) / ep.fixed_slices + 63) / 64 * 64;
      slices = (K + k_chunk - 1) / k_chunk;
      if (slices > 1 && (size_t)slices * M * N * sizeof(float) <= ep.ws_bytes) partial = (float*)ep.ws;
      else { slices = 1; k_chunk = K; }
    }
  } else if (big_blocks < 384 && blocks_small < 512 && K >= 256 && ep.ws != nullptr) {
    int want = (int)((768 + blocks_small - 1) / blocks_small);
    int maxs = K / 128;
    slices = want < maxs ? want : maxs;
    if (slices > 8) slices = 8;
    if (slices < 1) slices = 1;
    k_chunk = ((K + slices - 1) / slices + 63) / 64 * 64;       // whole k-tiles per slice
    slices = (K + k_chunk - 1) / k_chunk;
    if (slices > 1 && (size_t)slices * M * N * sizeof(float) <= ep.ws_bytes) partial = (float*)ep.ws;
    else { slices = 1; k_chunk = K; }
  }
  // 16-B row stores need a plain, aligned, 4-float-granular output
  const int wide = (partial == nullptr && ep.C1 == nullptr && (ldc & 3) == 0 && (N & 3) == 0 &&
                    (((uintptr_t)C) & 15) == 0 && (ep.bias == nullptr || (((uintptr_t)ep.bias) & 15) == 0)) ? 1 : 0;
  // one-tile-per-wave shapes: 64-deep k-tiles when there is k to cover (half the barriers per MFMA and
  // twice the prefetch distance)
  const bool deep = k_chunk >= 128;
  // Short reductions (K <= 64: dz = g Wf^T) are bound by writing the output; big register tiles buy no
  // reuse there and only lower the number of blocks in flight: 64 x 64 tiles measured 10-20 % faster.
  // Tile shape by measurement on this path's shapes (config sweeps of tools/gemm_bench.py, section 6 of
  // DESIGN.md): one 32 x 32 tile per wave almost everywhere -- the grids here are a few blocks per CU, so
  // more, smaller blocks beat register-tile reuse; the 128-row tiles only pay from ~10^5 rows on.
  //   N <= 32           128 x 32   (four row tiles, the narrow output uses every wave)
  //   N <= 64           64 x 64, or 128 x 64 from 98 304 rows
  //   N  > 64           64 x 64, or 128 x 128 from 81 920 rows when the reduction is long (K > 64: short
  //                     reductions -- dz = g Wf^T -- are bound by writing the output)
  // 64-deep k-tiles from K >= 512 (half the barriers per MFMA); the k order is the same for every shape,
  // so results do not depend on the choice.
  static const int forced = [] { const char* f = getenv("GEOBI_NN_CFG"); return f ? atoi(f) : 0; }();
  if (forced) {                 // tuning knob for tools/nn_cfg_sweep.py: force one tile shape for every call
    switch (forced) {
      case 1: GEOBI_GEMM_LAUNCH(2, 2, 1, 1, 32); break;
      case 2: GEOBI_GEMM_LAUNCH(2, 2, 2, 1, 32); break;
      case 4: GEOBI_GEMM_LAUNCH(2, 2, 2, 2, 16); break;
      case 5: GEOBI_GEMM_LAUNCH(4, 1, 1, 1, 32); break;
      default: GEOBI_GEMM_LAUNCH(2, 2, 1, 1, 64); break;
    }
  } else if (N <= 32) {
    GEOBI_GEMM_LAUNCH(4, 1, 1, 1, 32);
  } else if (N <= 64 && M >= 98304) {
    GEOBI_GEMM_LAUNCH(2, 2, 2, 1, 32);
  } else if (N > 64 && K > 64 && M >= 81920) {
    GEOBI_GEMM_LAUNCH(2, 2, 2, 2, 16);
  } else if (deep && K >= 512) {
    GEOBI_GEMM_LAUNCH(2, 2, 1, 1, 64);
  } else {
    GEOBI_GEMM_LAUNCH(2, 2, 1, 1, 32);
  }
#undef GEOBI_GEMM_LAUNCH
#undef GEOBI_GEMM_LAUNCH_F
  GEOBI_LAUNCH_OK();
  if (partial) {
    splitk_reduce_kernel<<<cdiv((int64_t)M * N, 256), 256, 0, s>>>(partial, slices, M, N, C, ldc, ep.bias, ep.slope,
                                                                  ep.C1, ep.split, ep.ldc1);
    GEOBI_LAUNCH_OK();
  }
  prof_end(PROF_GEMM, s);
  return 0;
}

size_t gemm_tn_ws_bytes(int I, int J, int64_t M) {
  // the slab count depends on the tile shape, which depends on the alignment of A: take the larger
  TnPlan p0 = plan_tn(nullptr, 4, I, J, M, (I & 3) == 1 ? I - 1 : -1);
  TnPlan p1 = plan_tn(nullptr, 1, I, J, M, -1);
  int by = p0.blocks_y > p1.blocks_y ? p0.blocks_y : p1.blocks_y;
  // the caller may ask for the column sums of B as one more row (I + 1 rows per slab) + a [J] scratch for them
  TnPlan p2 = plan_tn(nullptr, 4, I > 1 ? I - 1 : I, J, M, -1), p3 = plan_tn(nullptr, 1, I > 1 ? I - 1 : I, J, M, -1);
  if (p2.blocks_y > by) by = p2.blocks_y;
  if (p3.blocks_y > by) by = p3.blocks_y;
  return align_up((size_t)by * (I + 1) * J * sizeof(float)) + align_up((size_t)J * sizeof(float)) + 512;
}

// Upper bound over every column width J' <= J the caller may pass for the same (I, M): the conv backward
// issues du = dp^T [x | 1] once per input half when the layer reads a split (skip, up) input.
size_t gemm_tn_ws_bytes_any_width(int I, int J, int64_t M) {
  size_t best = 0;
  for (int j = 1; j <= J; ++j) {
    size_t b = gemm_tn_ws_bytes(I, j, M);
    if (b > best) best = b;
  }
  return best;
}

int gemm_tn(const float* A, int lda, const float* B, int ldb, int64_t M, int I, int J, int ones_row, int ones_col,
            const TnOutput& o, void* ws, size_t ws_bytes, hipStream_t s, int sum_row) {
  if (I <= 0 || J <= 0) return 0;
  const bool extra = ones_row >= 0 || sum_row >= 0;
  GEOBI_REQUIRE(M >= 0 && I - extra >= 1 && J - (ones_col >= 0) >= 1, "gemm_tn: empty operand");
  GEOBI_REQUIRE(ones_row < 0 || ones_row == I - 1, "gemm_tn: the ones row is the last row");
  GEOBI_REQUIRE(sum_row < 0 || (sum_row == I - 1 && ones_row < 0), "gemm_tn: the column-sums row is the last row");
  GEOBI_REQUIRE(ones_col < 0 || ones_col == J - 1, "gemm_tn: the ones column is the last column");
  // the column-sums row is not an MFMA row: tiles cover the data rows only
  const int I_mma = sum_row >= 0 ? I - 1 : I;
  TnPlan p = plan_tn(A, lda, I_mma, J, M > 0 ? M : 1, ones_row);
  Arena a(ws, ws_bytes);
  float* slabs = a.take<float>((size_t)p.blocks_y * I * J);
  GEOBI_REQUIRE(a.ok() && slabs, "gemm_tn: workspace too small (%zu < %zu)", ws_bytes, a.off);
  prof_begin(PROF_GEMM, s, 2.0 * (double)M * I_mma * J, 2);
  dim3 grid(p.tiles_i * p.tiles_j, p.blocks_y);
#define GEOBI_TN(TI_, TJ_, U_, VA_)                                                                         \
  gemm_tn_kernel<TI_, TJ_, U_, VA_><<<grid, 256, 0, s>>>(A, lda, B, ldb, M, I, J, ones_row, ones_col,         \
                                                          p.tiles_j, p.m_per_slice, slabs, sum_row)
  switch (p.va * 100 + p.ti * 10 + p.tj) {
    case 141: GEOBI_TN(4, 1, 4, true); break;
    case 11: GEOBI_TN(1, 1, 4, false); break;
    case 12: GEOBI_TN(1, 2, 4, false); break;
    case 13: GEOBI_TN(1, 3, 4, false); break;
    case 21: GEOBI_TN(2, 1, 4, false); break;
    case 22: GEOBI_TN(2, 2, 4, false); break;
    default: GEOBI_TN(2, 3, 4, false); break;
  }
#undef GEOBI_TN
  GEOBI_LAUNCH_OK();
  const int bias_blocks = (o.mode == TN_RPRIME && o.extra_row && o.C2 != nullptr) ? cdiv(o.Cout, 8) : 0;
  tn_reduce_kernel<<<cdiv((int64_t)I * J, 64) + bias_blocks, 256, 0, s>>>(slabs, p.blocks_y, I, J, o);
  GEOBI_LAUNCH_OK();
  prof_end(PROF_GEMM, s);
  return 0;
}

static int colsum_blocks(int64_t M) {
  int64_t b = (M + 511) / 512;
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return (int)b;
}

size_t colsum_ws_bytes(int64_t M, int J) { return align_up((size_t)colsum_blocks(M) * J * sizeof(float)) + 256; }

int colsum(const float* A, int lda, int64_t M, int J, float* out, void* ws, size_t ws_bytes, hipStream_t s) {
  if (J <= 0) return 0;
  int blocks = colsum_blocks(M);
  Arena a(ws, ws_bytes);
  float* partial = a.take<float>((size_t)blocks * J);
  GEOBI_REQUIRE(a.ok() && partial, "colsum: workspace too small");
  int64_t rpb = (M + blocks - 1) / blocks;
  if (rpb < 1) rpb = 1;
  colsum_partial_kernel<<<dim3(blocks, cdiv(J, 256)), 256, 0, s>>>(A, lda, M, J, rpb, partial);
  GEOBI_LAUNCH_OK();
  colsum_final_kernel<<<cdiv(J, 256), 256, 0, s>>>(partial, blocks, J, out);
  GEOBI_LAUNCH_OK();
  return 0;
}

}  // namespace geobi
