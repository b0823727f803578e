// FeaSt layer, fused form: aggregation + node-level transform in ONE kernel -- the aggregated rows
// z_i = [1/deg sum_j q_ijh x_j]_h (9*C floats per node) never reach HBM.
//
// A workgroup owns a tile of consecutive nodes -- 16 nodes / 4 waves (default, four workgroups per CU) or 32 nodes /
// 8 waves (round-2 geometry; `Shape`, GEOBI_TILE16, geobi_set_tile_rows):
//   phase A  the gather of feast.hip's aggregation kernel (16 lanes per node, per-edge softmax evaluated once and
//            parked, neighbour rows gathered as one contiguous segment per row), but the 9 x VEC register
//            accumulators of a node are stored into an LDS tile z[rows][LD] instead of global memory.  The per-edge
//            parking slots live in the first 3C floats of the node's own (not yet written) tile row, so the tile is
//            the kernel's only LDS.
//   phase B  out[rows, NOUT] = z[rows, K] * Bp[K, NOUT] on the matrix cores (v_mfma_f32_16x16x4_f32 / 32x32x2, exact
//            fp32).  The waves split the column tiles and the K range; the K-split partial tiles are folded through
//            LDS in a fixed order (deterministic), bias + leaky-relu applied, rows written with 8-B stores (one
//            128-B segment per 16 lanes).
// Operand delivery: the weights are packed so that ONE 16-B load per lane feeds four consecutive MFMAs in either shape,
//   Bp[k / 4][col][k % 4] = B[k][col]           (K padded to 16),
// and the A operand is the matching 16 B of the lane's tile row (ds_read_b128; the row stride keeps it conflict-free).
// Any bijection between k-slots and lanes works as long as A and B agree; the order of the fp32 sum is fixed by it and
// does not depend on N or on the batch.
//
// MODE 0  forward:   rows = layer input (xa | xb), K = 9 Cin, Bp from lin.weight, epilogue bias + leaky-relu.
// MODE 1  backward:  dx = [r | dp | dcs] [lin.weight ; u.weight ; 0]: rows = g (gradient w.r.t. the
//                    pre-activation output), transposed CSR, weights q_ij / deg_i; the tile gets 24 extra
//                    columns [dp | dcs] -- dp summed in the kernel from the row pass' dl rows, dcs read from
//                    `dpd`; output split into (dxa | dxb) for split inputs; the tile rows r' are written out
//                    once for the weight-gradient product.
// The backward's first half (g, dz = g Wf^T on the matrix cores, per-edge softmax backward) is
// feast_rowpass_fused_kernel / feast_rowpass_fused128_kernel below.
#include "common.h"
#include <hip/hip_ext.h>
#include <type_traits>
#ifdef GEOBI_FUSED_STAMPS
// diagnostic build: the staged row pass (feast_dev.h) stamps its own steps into the backward kernel's buffer
namespace geobi { namespace { __device__ unsigned long long g_stamps_bwd[16384][8]; } }
#define GEOBI_RP_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 16384) ::geobi::g_stamps_bwd[blockIdx.x][i] = __builtin_amdgcn_s_memtime(); } while (0)
#endif
#include "feast_dev.h"

namespace geobi {

namespace {

using namespace feast_dev;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Two tile geometries (template parameter ROWS of the kernels below):
//   32 rows  8 waves, v_mfma_f32_32x32x2_f32, 74-78 KB of LDS at 64 channels: TWO workgroups per CU
//   16 rows  4 waves, v_mfma_f32_16x16x4_f32 (same flops per cycle, 32 cycles per instruction, 4-register
//            accumulators), 37-40 KB: FOUR workgroups per CU -- the same 16 waves, but four independent phase
//            sequences instead of two, so one tile's matrix phase covers another's gather latency.
// The packed weights serve both: element (k, col) sits at Bp[k / 4][col][k % 4] (one 16-B load per lane = the lane's
// k-slot of four consecutive MFMAs); only the padding of K differs (16 covers both).
constexpr int TN = 32;        // rows of the 32-row geometry (128-channel backward kernel)
constexpr int NW = 8;
constexpr int RED_LD = 36;    // row stride of a partial output tile in LDS
constexpr int G = 16;         // lanes per node in the gather phase: a wave owns 4 nodes
constexpr int NPW = 64 / G;
static_assert(NW * NPW == TN, "one gather pass covers the tile");
constexpr int KPAD = 16;      // the packed weights' K is padded to whole 16-deep blocks (zero rows)

// Shape of one instantiation.  C = width of the gathered rows; MODE 0: K = 9 C (padded to 8), MODE 1: 9 C + 24.
// The 16 lanes of a node split the C channels: VEC floats per lane in NP pieces of PV floats (C = 128: two
// 16-B pieces, channels [4k, 4k+4) and [64 + 4k, ..), so every load instruction still covers one contiguous
// 256-B segment of the row); C = 6 / 12: one float on the first C lanes.
// C = 128 does not fit a [32][9 C] tile twice into a CU's LDS: the tile then holds HC = 4 heads at a time
// (3 chunks: 4 + 4 + 1 heads), the node's 9 x VEC accumulators wait in registers, and the MFMA accumulators run
// across the chunks.
template <int C, int MODE, int ROWS>
struct Shape {
  static constexpr int NWV = ROWS / NPW;                          // waves per workgroup
  static constexpr int KB = ROWS == 32 ? 8 : 16;                  // depth of one weight block (4 MFMAs)
  static constexpr int VEC = C >= 16 ? C / 16 : 1;
  static constexpr int NP = VEC > 4 ? VEC / 4 : 1;
  static constexpr int PV = VEC / NP;
  static constexpr int ACTIVE = C >= 16 ? 16 : C;                 // lanes of the group that own channels
  static constexpr int HC = C >= 128 ? 4 : H;                     // heads per LDS chunk
  static constexpr int NCHUNK = (H + HC - 1) / HC;
  static constexpr int KR = MODE == 0 ? H * C : H * C + 2 * HP;   // real columns of a tile row (MODE 1: the r' row)
  static constexpr int KD = (KR + KB - 1) / KB * KB;              // K of the whole product: whole weight blocks
  static constexpr int KC_FULL = HC * C;                          // chunk width (all but the last chunk)
  static constexpr int KC_LAST = KD - (NCHUNK - 1) * KC_FULL;
  static constexpr int KR_LAST = KR - (NCHUNK - 1) * KC_FULL;     // real columns of the last chunk
  static constexpr int KC_MAX = NCHUNK == 1 ? KD : (KC_FULL > KC_LAST ? KC_FULL : KC_LAST);
  // row stride: 32 rows: = 4 mod 64, conflict-free ds_read_b128 of A; 16 rows: LD / 4 odd is all 16 rows need (one
  // 2-way quad remains between the k-slots of a lane group: 36 A reads per wave and tile, immaterial) and keeps the
  // 64-channel tiles under 40 KB
  static constexpr int LD = ROWS == 32 ? (KC_MAX - 4 + 63) / 64 * 64 + 4 : KC_MAX + 4;
  static constexpr bool SLOT_IN_ROW = G * HP <= (NCHUNK == 1 ? H * C : KC_FULL);
  static constexpr int RED_FLOATS = ROWS == 32 ? NW * 32 * RED_LD : 4 * 16 * 36;     // K-split partial tiles
  static constexpr int TILE_FLOATS = ROWS * LD > RED_FLOATS ? ROWS * LD : RED_FLOATS;
  static constexpr int SLOT_FLOATS = SLOT_IN_ROW ? 0 : NWV * 64 * HP;
  // MODE 1: the node's [dp | dcs] columns are formed while its edges are walked -- in their place at the end of the
  // tile row when the row is one chunk, in a side array when the last chunk's columns overlap the parking slots
  static constexpr int DP_FLOATS = (MODE == 1 && NCHUNK > 1) ? ROWS * 2 * HP : 0;
  static_assert(VEC * ACTIVE == C || C < 16, "channel split");
  static_assert(KD % KB == 0 && (NCHUNK == 1 || KC_FULL % KB == 0) && KC_LAST % KB == 0 && KC_LAST > 0, "whole k-blocks per chunk");
  static_assert(KC_MAX % 16 == 0 || ROWS == 32, "LD / 4 odd");
};
template <int C, int MODE, int LC, int ROWS>
constexpr int fused_lds_floats() {
  return Shape<C, MODE, ROWS>::TILE_FLOATS + Shape<C, MODE, ROWS>::SLOT_FLOATS + (LC > 0 ? LC * HP : 0) +
         Shape<C, MODE, ROWS>::DP_FLOATS + 64 * Shape<C, MODE, ROWS>::NWV;      // + the landing zone of warm_l2
}
// K of the PACKED weights (both geometries read the same pack): the r' / z row rounded up to 16
constexpr int fused_k(int C, int MODE) { return ((MODE == 0 ? H * C : H * C + 2 * HP) + KPAD - 1) / KPAD * KPAD; }

#ifdef GEOBI_FUSED_STAMPS
// Diagnostic build only (tools/build_variant.sh ... -DGEOBI_FUSED_STAMPS): shader-clock stamps of wave 0 of each
// workgroup at the phase boundaries, written to a buffer nothing else reads.
__device__ unsigned long long g_stamps[16384][8];
#define GEOBI_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 16384) g_stamps[blockIdx.x][i] = __builtin_amdgcn_s_memtime(); } while (0)
#define GEOBI_STAMP_BWD(i) do { if (threadIdx.x == 0 && blockIdx.x < 16384) g_stamps_bwd[blockIdx.x][i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define GEOBI_STAMP(i) do { } while (0)
#define GEOBI_STAMP_BWD(i) do { } while (0)
#endif

// (Five waves per SIMD for the 32-channel instantiations -- 96 registers, LDS allows seven workgroups -- measured
// 56.9 against 57.3 us for the dx kernel of the 64 -> 32 layers and slower elsewhere: not occupancy-bound.)
// CS = 2 (16-row tiles only): the launch's grid has a second dimension and workgroup (tile, part) produces, of every
// wave's 32-column group, the 16 columns [16 part, 16 part + 16) -- each wave runs ONE accumulator chain over the same
// k range in the same order as the unsplit kernel (bit-identical columns), with half the matrix work and half the
// weight traffic per workgroup; the gather phase is done by both parts.  For launches with fewer tiles than the chip has
// workgroup slots (the coarse graph levels): see column_parts().
template <int C, int MODE, int LC, int NT, int ROWS, int CS = 1>
__global__ __launch_bounds__(16 * ROWS, 4) void feast_fused_kernel(
    const float* __restrict__ xa, const float* __restrict__ xb, int Ca, const float* __restrict__ p,
    const float* __restrict__ cvec, const int* __restrict__ rowptr, const int* __restrict__ col,
    const int* __restrict__ deg_rowptr, int N, const float* __restrict__ xl, const float* __restrict__ ul,
    const float* __restrict__ dpd, const float* __restrict__ dl, const int* __restrict__ pos,
    const float* __restrict__ dpn, const float* __restrict__ Bp, int NOUT, const float* __restrict__ bias,
    float slope, float* __restrict__ out, int ldo, float* __restrict__ out1, int split, int ldo1,
    float* __restrict__ tile_out, int warm_on) {
  using S = Shape<C, MODE, ROWS>;
  constexpr int VEC = S::VEC, NP = S::NP, PV = S::PV, LD = S::LD, HC = S::HC, NCHUNK = S::NCHUNK;
  constexpr int TN = ROWS, NW = S::NWV, NTHREADS = 64 * NW;       // shadow the 32-row constants
  static_assert(NW % NT == 0, "column tiles divide the waves");
  static_assert(CS == 1 || (CS == 2 && ROWS == 16), "column parts: two halves of the 16-row geometry's column groups");
  const int cpart = CS > 1 ? blockIdx.y : 0;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_slots = smem + S::TILE_FLOATS;
  float* s_u = s_slots + S::SLOT_FLOATS;
  float* s_dp = s_u + (LC > 0 ? LC * HP : 0);
  if constexpr (LC > 0) stage_u<LC>(ul, s_u);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = xcd_block(blockIdx.x, gridDim.x);
  if (tile * TN >= N) return;          // whole workgroup: surplus tile of the XCD-padded grid

  GEOBI_STAMP(0);
  warm_l2<NW>(Bp, fused_k(C, MODE) * 32 * NT * (int)sizeof(float), warm_on, s_dp + S::DP_FLOATS);
  // ------------------------------------------------------------------ phase A: aggregate into registers
  const int g = lane / G, k = lane % G;
  const int nl = wave * NPW + g;                 // node within the tile
  const int node = tile * TN + nl;
  const bool valid = node < N;
  float* zrow = smem + nl * LD;
  float acc[H][VEC];
  {
    const bool act = k < S::ACTIVE;              // lanes that own channels (C = 6 / 12: the first C)
    const float* fb[NP];
    int fs[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const int c0 = act ? q * (C / NP) + k * PV : 0;
      if (c0 < Ca) { fb[q] = xa + c0; fs[q] = Ca; } else { fb[q] = xb + (c0 - Ca); fs[q] = C - Ca; }
    }
    float cc[H], qs[H];
    // the self edge: u(x_i - x_i) + c = c exactly; its softmax is the same for every node: packed behind the weights
    const float* qself = Bp + (size_t)fused_k(C, MODE) * 32 * NT;
#pragma unroll
    for (int h = 0; h < H; ++h) { cc[h] = cvec[h]; qs[h] = qself[h]; }

    const int ns = valid ? node : N - 1;
    const int rs = rowptr[ns];
    const int re = valid ? rowptr[ns + 1] : rs;
    float(*slot)[HP] = S::SLOT_IN_ROW ? reinterpret_cast<float(*)[HP]>(zrow)
                                      : reinterpret_cast<float(*)[HP]>(s_slots + (wave * 64 + g * G) * HP);
    // the centre's logit row is re-read (L1) at the top of every chunk instead of living in 9 registers across the
    // row-gather loop: that is what lets four neighbour rows be in flight at once within the 128-register budget
    float xc[LC > 0 ? LC : 1];
    if constexpr (LC > 0) load_row<LC>(xl + (size_t)ns * LC, xc);
    // MODE 1: dp_j = sum over the out-edges (j -> i) of dl_ij - dpn_j is summed right here, over the edges this phase
    // walks anyway (dl row = one more 48-B read per edge next to the logit row; until round 3 a kernel of its own per
    // layer): lane k < 3 owns float4 k of dp, lanes 3..5 fetch dcs
    float* dprow = nullptr;
    if constexpr (MODE == 1) {
      dprow = NCHUNK == 1 ? zrow + H * C : s_dp + nl * (2 * HP);
      if (k < 6) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) {
          if (k < 3) {
            t = reinterpret_cast<const float4*>(dpn + (size_t)node * HP)[k];
            t = make_float4(-t.x, -t.y, -t.z, -t.w);
          } else {
            t = reinterpret_cast<const float4*>(dpd + (size_t)node * (2 * HP))[k];
          }
        }
        reinterpret_cast<float4*>(dprow)[k] = t;
      }
    }
    {
      float xs[VEC];
#pragma unroll
      for (int q = 0; q < NP; ++q) load_piece<PV>(fb[q] + (size_t)ns * fs[q], xs + q * PV);
      float sscale = 1.0f;
      if constexpr (MODE == 1) sscale = 1.0f / (float)(deg_rowptr[ns + 1] - deg_rowptr[ns] + 1);
#pragma unroll
      for (int h = 0; h < H; ++h)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[h][v] = qs[h] * sscale * xs[v];
    }
#ifdef GEOBI_FUSED_UNR
    constexpr int UNR = GEOBI_FUSED_UNR;          // tuning builds (tools/build_variant.sh)
#else
    constexpr int UNR = VEC <= 4 ? 4 : 2;
#endif
    // (Tried in round 3: the first UNR rows requested right behind col -- lane t's id handed to the channel lanes by a
    // DPP row broadcast -- i.e. beside the logit rows instead of behind the softmax: 48.3 against 47.9 us, neutral.)
    for (int base = rs; base < re; base += G) {
      // ---- lane k of the group handles edge base + k: logits, softmax, park q and the neighbour id
      {
        const int e = base + k;
        float q[H];
        int j = ns;
        float dsum[MODE == 1 ? H : 1];
        if constexpr (MODE == 1) {
#pragma unroll
          for (int h = 0; h < H; ++h) dsum[h] = 0.f;
        }
        if (e < re) {
          j = col[e];
          if constexpr (MODE == 1) load_hp(dl + (size_t)pos[e] * HP, dsum);
          float pc[H];
          if constexpr (LC == 0) {
            const float* prow = p + (size_t)ns * HP;
            asm volatile("" : "+v"(prow));            // keep the load inside the loop
            load_hp(prow, pc);
          }
          if constexpr (LC > 0) {
            float d[LC];
            load_row<LC>(xl + (size_t)j * LC, d);
#pragma unroll
            for (int i = 0; i < LC; ++i) d[i] = (MODE == 0) ? (d[i] - xc[i]) : (xc[i] - d[i]);
            edge_logits<LC>(d, s_u, cc, q);
          } else {
            float pn[H];
            load_hp(p + (size_t)j * HP, pn);
#pragma unroll
            for (int h = 0; h < H; ++h) q[h] = (MODE == 0) ? (pn[h] - pc[h] + cc[h]) : (pc[h] - pn[h] + cc[h]);
          }
          softmax9(q);
          if constexpr (MODE == 1) {
            float w = 1.0f / (float)(deg_rowptr[j + 1] - deg_rowptr[j] + 1);
#pragma unroll
            for (int h = 0; h < H; ++h) q[h] *= w;
          }
        } else {
#pragma unroll
          for (int h = 0; h < H; ++h) q[h] = 0.f;
        }
        float4* dst = reinterpret_cast<float4*>(slot[k]);
        dst[0] = make_float4(q[0], q[1], q[2], q[3]);
        dst[1] = make_float4(q[4], q[5], q[6], q[7]);
        dst[2] = make_float4(q[8], __int_as_float(j), 0.f, 0.f);
        if constexpr (MODE == 1) {
          // the group's 16 dl rows -> every lane holds the nine sums (fixed butterfly order); lane k < 3 adds its float4
#pragma unroll
          for (int h = 0; h < H; ++h) dsum[h] = group_allreduce<G>(dsum[h]);
          if (k < 3) {
            float4 t = reinterpret_cast<float4*>(dprow)[k];
            const float4 a = k == 0 ? make_float4(dsum[0], dsum[1], dsum[2], dsum[3])
                           : k == 1 ? make_float4(dsum[4], dsum[5], dsum[6], dsum[7])
                                    : make_float4(dsum[8], 0.f, 0.f, 0.f);
            t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w;
            reinterpret_cast<float4*>(dprow)[k] = t;
          }
        }
      }
      wave_lds_sync();
      GEOBI_STAMP(1);
      // ---- the lanes that own channels walk the parked edges, UNR at a time (surplus slots hold q = 0, j = ns)
      const int cnt = min(G, re - base);
      if (act) {
        for (int t = 0; t < cnt; t += UNR) {
          // UNR neighbour rows in flight; the parked q of an edge is read (LDS broadcast) right before its FMAs
          float xv[UNR][VEC];
#pragma unroll
          for (int u = 0; u < UNR; ++u) {
            const int jn = __float_as_int(slot[t + u][H]);
#pragma unroll
            for (int q = 0; q < NP; ++q) load_piece<PV>(fb[q] + (size_t)jn * fs[q], xv[u] + q * PV);
          }
#pragma unroll
          for (int u = 0; u < UNR; ++u) {
            const float4* sp = reinterpret_cast<const float4*>(slot[t + u]);
            const float4 a = sp[0], b = sp[1];
            const float q8 = slot[t + u][8];
            const float qq[H] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, q8};
#pragma unroll
            for (int h = 0; h < H; ++h)
#pragma unroll
              for (int v = 0; v < VEC; ++v) acc[h][v] = fmaf(qq[h], xv[u][v], acc[h][v]);
          }
        }
      }
      wave_lds_sync();
    }
    GEOBI_STAMP(2);
    float scale = valid ? 1.0f : 0.0f;
    if constexpr (MODE == 0) scale = valid ? 1.0f / (float)(re - rs + 1) : 0.0f;
#pragma unroll
    for (int h = 0; h < H; ++h)
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[h][v] *= scale;
  }

  // ------------------------------------------------------------------ chunks: registers -> LDS tile -> MFMA
  // Matrix-phase roles.  32 rows: wave = (column tile ct of 32, k-range ks), one 32 x 32 accumulator.  16 rows: wave =
  // (column group cg of 32 = two 16-column tiles, k-range ks): two independent 4-register accumulators whose MFMAs
  // alternate (dependent latency 40 cycles > the 32-cycle issue interval of v_mfma_f32_16x16x4_f32).
  constexpr int KS = NW / NT;
  const int ct = wave % NT, ks = wave / NT;
  const int hf = lane >> 5, l31 = lane & 31;
  const int kq = lane >> 4, l15 = lane & 15;                      // 16 rows: k-slot and row / column of the lane
  f32x16 macc;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  if constexpr (ROWS == 32) {
#pragma unroll
    for (int r = 0; r < 16; ++r) macc[r] = 0.f;
  }

#pragma unroll
  for (int c = 0; c < NCHUNK; ++c) {
    const int nheads = (c == NCHUNK - 1) ? H - c * HC : HC;
    // ---- this lane's slices of heads [c HC, c HC + nheads) -> its node's tile row
    if (k < S::ACTIVE) {
#pragma unroll
      for (int h = 0; h < H; ++h) {
        if (h / HC != c) continue;
#pragma unroll
        for (int q = 0; q < NP; ++q)
          store_piece<PV>(zrow + (h - c * HC) * C + q * (C / NP) + k * PV, &acc[h][q * PV]);
      }
    }
    if (c == NCHUNK - 1) {
      if constexpr (MODE == 0) {
        for (int i = nheads * C + k; i < S::KC_LAST; i += G) zrow[i] = 0.f;          // K padding
      } else {
        if constexpr (NCHUNK > 1) {                                                 // [dp | dcs]: 24 floats
          if (k < 6)
            reinterpret_cast<float4*>(zrow + nheads * C)[k] = reinterpret_cast<const float4*>(s_dp + nl * (2 * HP))[k];
        }
        for (int i = S::KR_LAST + k; i < S::KC_LAST; i += G) zrow[i] = 0.f;          // K padding behind them
      }
    }
    // ---- tile chunk x packed weights on the matrix cores.  The weights of a wave's k-range come from L2, one 16-B load
    // per lane, k-block and column tile (= 4 MFMAs per tile), always requested BEFORE the barrier that publishes the tile
    // and then ahead of their use in one of two ways that hold the same number of registers:
    //   two batches  "current" and "next" sets of BATCH blocks, the next batch requested at the top of the current one:
    //                a lead of BATCH blocks.  The short k-ranges (K split over the waves) and the launches with four waves
    //                per SIMD run best on it (the level-0 layers: 58.3 against 61.5 us for the ring, same box).
    //   ring         RING = 2 BATCH blocks in flight, a slot requested again right behind the MFMAs that read it: twice
    //                the lead.  For the long k-ranges of the coarse levels (128 channels read, or one wave per column
    //                group), where a launch has one or two waves per SIMD and nothing else hides the weight latency
    //                (43.7 -> 37.1 us for the 128 -> 128 dx at 264 tiles; profiles/r04_weight_ring.txt).
    // A is read from LDS one block ahead.
    {
      constexpr int KB = S::KB;
      constexpr int WPB = (ROWS == 32 || CS == 2) ? 1 : 2;         // weight loads per block (column tiles per wave)
      constexpr int BATCH = (VEC >= 8 ? 4 : 8) / WPB;
      constexpr bool USE_RING = VEC >= 8 || NT == 4;
      constexpr int RING = 2 * BATCH;
      constexpr int BSTR = (KB / 4) * 32 * NT;                     // float4 per weight block
      const int nkb = ((c == NCHUNK - 1) ? S::KC_LAST : S::KC_FULL) / KB;
      const int kb_base = c * (S::KC_FULL / KB);
      const int kb_per = (nkb + KS - 1) / KS;
      const int kb0 = ks * kb_per;
      const int kb1 = min(nkb, kb0 + kb_per);
      const bool work = kb0 < kb1;
      const float* arow = ROWS == 32 ? smem + l31 * LD + 4 * hf : smem + l15 * LD + 4 * kq;
      const float4* bcol = reinterpret_cast<const float4*>(Bp) + (size_t)kb_base * BSTR +
                           (ROWS == 32 ? (size_t)hf * (32 * NT) + ct * 32 + l31
                                       : (size_t)kq * (32 * NT) + ct * 32 + 16 * cpart + l15);
      float4 w[RING][WPB];                                          // two batches: [0, BATCH) current, [BATCH, RING) next
      auto load_w = [&](float4 (&wv)[WPB], int b) {                // clamped: always a valid address
#pragma unroll
        for (int j = 0; j < WPB; ++j) wv[j] = bcol[(size_t)min(b, kb1 - 1) * BSTR + 16 * j];
      };
      auto mma = [&](const float4& a, const float4 (&wv)[WPB]) {
        if constexpr (ROWS == 32) {
          macc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, wv[0].x, macc, 0, 0, 0);
          macc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, wv[0].y, macc, 0, 0, 0);
          macc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, wv[0].z, macc, 0, 0, 0);
          macc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, wv[0].w, macc, 0, 0, 0);
        } else {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, wv[0].x, acc0, 0, 0, 0);
          if constexpr (WPB == 2) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, wv[WPB - 1].x, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, wv[0].y, acc0, 0, 0, 0);
          if constexpr (WPB == 2) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, wv[WPB - 1].y, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, wv[0].z, acc0, 0, 0, 0);
          if constexpr (WPB == 2) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, wv[WPB - 1].z, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, wv[0].w, acc0, 0, 0, 0);
          if constexpr (WPB == 2) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, wv[WPB - 1].w, acc1, 0, 0, 0);
        }
      };
      if (work) {
#pragma unroll
        for (int u = 0; u < (USE_RING ? RING : BATCH); ++u) load_w(w[u], kb0 + u);
      }
      GEOBI_STAMP(3);
      __syncthreads();
      GEOBI_STAMP(4);
      if (tile_out != nullptr && cpart == 0) {
        // the tile rows themselves (MODE 1: r' = [r | dp | dcs]) for the weight-gradient GEMM [x | 1]^T r'
        const int kc = (c == NCHUNK - 1) ? S::KR_LAST : S::KC_FULL;
        const int q4 = kc >> 2;                                    // float4 per row of this chunk
        for (int i = threadIdx.x; i < TN * q4; i += NTHREADS) {
          const int r = i / q4, c4 = (i - r * q4) * 4;
          const int gn = tile * TN + r;
          if (gn < N)
            *reinterpret_cast<float4*>(tile_out + (size_t)gn * S::KR + c * S::KC_FULL + c4) =
                *reinterpret_cast<const float4*>(smem + r * LD + c4);
        }
      }
      if (work) {
        float4 a = *reinterpret_cast<const float4*>(arow + KB * kb0);
        if constexpr (USE_RING) {
          for (int b = kb0; b < kb1; b += RING) {
#pragma unroll
            for (int u = 0; u < RING; ++u) {
              const float4 an = *reinterpret_cast<const float4*>(arow + KB * min(b + u + 1, kb1 - 1));
              if (b + u < kb1) mma(a, w[u]);
              // unconditional (clamped): behind a branch the compiler's vmcnt waits assume the load was skipped.  The
              // scheduling barrier keeps the request HERE: left alone, the scheduler gathers an iteration's refills behind
              // its last MFMA, which leaves the first blocks of the next iteration no lead at all
              load_w(w[u], b + u + RING);
              __builtin_amdgcn_sched_barrier(0);
              a = an;
            }
          }
        } else {
          for (int b = kb0; b < kb1; b += BATCH) {
#pragma unroll
            for (int u = 0; u < BATCH; ++u) load_w(w[BATCH + u], b + BATCH + u);
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
              const float4 an = *reinterpret_cast<const float4*>(arow + KB * min(b + u + 1, kb1 - 1));
              if (b + u < kb1) mma(a, w[u]);
              a = an;
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u)
#pragma unroll
              for (int j = 0; j < WPB; ++j) w[u][j] = w[BATCH + u][j];
          }
        }
      }
    }
    GEOBI_STAMP(5);
    __syncthreads();                               // every wave is done reading this chunk of the tile
    GEOBI_STAMP(6);
  }
  // ---- partial tiles of the K splits -> LDS.  C/D layouts: 32 x 32: row (r & 3) + 8 (r >> 2) + 4 hf, column l31;
  // 16 x 16: row 4 kq + r, column l15
  constexpr int RLD = ROWS == 32 ? RED_LD : 36;
  float(*red)[TN][RLD] = reinterpret_cast<float(*)[TN][RLD]>(smem);
  if constexpr (ROWS == 32) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * hf][l31] = macc[r];
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      red[wave][4 * kq + r][l15] = acc0[r];
      if constexpr (CS == 1) red[wave][4 * kq + r][16 + l15] = acc1[r];
    }
  }
  __syncthreads();

  // ------------------------------------------------------------------ epilogue: fold the K splits, store
  const int row = threadIdx.x >> 4, c2 = (threadIdx.x & 15) * 2;
  const int onode = tile * TN + row;
  GEOBI_STAMP(7);
  if (onode >= N) return;
  if (CS == 2 && c2 >= 16) return;                 // a part holds 16 columns of every group
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    float2 s = make_float2(0.f, 0.f);
#pragma unroll
    for (int q = 0; q < KS; ++q) {
      const float2 v = *reinterpret_cast<const float2*>(&red[q * NT + t][row][c2]);
      s.x += v.x; s.y += v.y;
    }
    const int cidx = t * 32 + 16 * cpart + c2;
    if (cidx >= NOUT) continue;
    if constexpr (MODE == 0) {
      s.x += bias[cidx]; s.y += bias[cidx + 1];
      s.x = s.x > 0.f ? s.x : s.x * slope;
      s.y = s.y > 0.f ? s.y : s.y * slope;
      *reinterpret_cast<float2*>(out + (size_t)onode * ldo + cidx) = s;
    } else {
      const float v2[2] = {s.x, s.y};
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int cc_ = cidx + i;
        if (cc_ >= NOUT) continue;
        if (out1 != nullptr && cc_ >= split) out1[(size_t)onode * ldo1 + (cc_ - split)] = v2[i];
        else out[(size_t)onode * ldo + cc_] = v2[i];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Backward, first half, fused:  g = d(leaky-relu) gout,  dz = g Wf^T  (MFMA, into an LDS tile -- dz [N, 9 Cin] never
// reaches HBM),  then the row pass over the tile's 32 target nodes: per in-edge recompute q, s_h = dz_i[h,:] . x_j,
// softmax backward dl_h = q_h (s_h - sum q s) / deg_i  ->  dl [E, 12] and the per-node sums dpn (what flows to -p_i)
// and dcs (dpn + the self edge's share).  The mirror image of the forward kernel: matrix phase first, gather second;
// the node's dz row is read into registers and its LDS row then serves as the parking slots of its edges.
// Shapes: C = Cin <= 64 (one chunk); Cout in {32, 64, 128} (the reduction length of the matrix phase).
// Rows (target nodes) per tile of the fused backward row pass: 32 = one full MFMA tile, 8 waves, two workgroups per CU
// (78 KB of LDS each); 16 = half-used MFMA tiles, 4 waves, FOUR workgroups per CU (39 KB each) whose phases interleave
// better -- but the matrix phase then does twice the MFMA work per node, and that costs more than the interleaving
// gains: 262 against 265 M-edges/s in a same-box A/B (tools/ab_lib.sh).  32 is the product build.
// (Round 2 tried 16 rows on HALF-USED 32 x 32 MFMA tiles: twice the matrix work per node, no gain.  The 16-row form below
// runs on v_mfma_f32_16x16x4_f32: same matrix work per node as the 32-row form.)
// RP = 1: the row pass stages the neighbour rows through LDS (feast_dev.h: rowpass_edge_node_staged; 64 channels, node-level
// logits, both input parts whole 32-channel halves).
template <int C, int LC, int COUT, int RT, int RP = 0>
__global__ __launch_bounds__(16 * RT, 4) void feast_rowpass_fused_kernel(
    const float* __restrict__ xa, const float* __restrict__ xb, int Ca, const float* __restrict__ p,
    const float* __restrict__ cvec, const int* __restrict__ rowptr, const int* __restrict__ col, int N,
    const float* __restrict__ ul, const float* __restrict__ gout, const float* __restrict__ out_act, float slope,
    const float* __restrict__ Wf, int Kp, float* __restrict__ g_out, float* __restrict__ dl, float* __restrict__ dpn,
    float* __restrict__ dcs, int ld_dcs) {
  constexpr int K = H * C;
  constexpr int KW = RT / NPW, KT = 64 * KW;     // waves / threads per workgroup
  constexpr int NCT = (K + 31) / 32;             // 32-column tiles of dz
  constexpr int LDZ = NCT * 32 + 4;              // + 4: the four node groups of a wave read distinct bank quads
  constexpr int GL = COUT + 4;                   // g tile row stride: conflict-free ds_read_b128 of the A operand
  static_assert(C <= 64, "one chunk");
  static_assert(RT == 32 || RT == 16, "a tile is one 32 x 32 MFMA tile or a row of 16 x 16 tiles");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_g = smem;                             // [RT][GL]
  float* s_z = smem + RT * GL;                   // [RT][LDZ]
  float* s_u = s_z + RT * LDZ;
  if constexpr (LC > 0) stage_u<LC>(ul, s_u);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = xcd_block(blockIdx.x, gridDim.x);
  if (tile * RT >= N) return;
  GEOBI_STAMP_BWD(0);

  // ---- g tile: gradient through the fused leaky-relu, kept for the MFMAs and written out for the dx kernel
  auto g_tile = [&]() {
    constexpr int Q = COUT / 4;
    for (int i = threadIdx.x; i < RT * Q; i += KT) {
      const int r = i / Q, c4 = (i - r * Q) * 4;
      const int gn = tile * RT + r;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gn < N) {
        v = *reinterpret_cast<const float4*>(gout + (size_t)gn * COUT + c4);
        if (out_act != nullptr) {
          const float4 o = *reinterpret_cast<const float4*>(out_act + (size_t)gn * COUT + c4);
          v.x = o.x > 0.f ? v.x : v.x * slope; v.y = o.y > 0.f ? v.y : v.y * slope;
          v.z = o.z > 0.f ? v.z : v.z * slope; v.w = o.w > 0.f ? v.w : v.w * slope;
          *reinterpret_cast<float4*>(g_out + (size_t)gn * COUT + c4) = v;
        }
      }
      *reinterpret_cast<float4*>(s_g + r * GL + c4) = v;
    }
    __syncthreads();
    GEOBI_STAMP_BWD(1);
  };

  // ---- matrix phase: dz[RT, K] = g[RT, COUT] Wf^T; wave w owns column tiles w, w + KW, w + 2 KW, ...
  if constexpr (RT == 32) {
    // The weights of up to three of a wave's tiles are requested together, four k-blocks at a time: one exposed load
    // latency per batch instead of one per tile.
    g_tile();
    const int hf = lane >> 5, l31 = lane & 31;
    constexpr int NKB = COUT / 8;
    constexpr int MAXT = (NCT + KW - 1) / KW;                      // column tiles per wave
    constexpr int TB = MAXT < 3 ? MAXT : 3;                        // tiles per round (48 accumulator registers)
    constexpr int HB = 4;                                          // k-blocks per batch of weight loads
    static_assert(NKB % HB == 0, "whole batches");
    const float* arow = s_g + l31 * GL + 4 * hf;
#pragma unroll
    for (int t0 = 0; t0 < MAXT; t0 += TB) {
      const float* brow[TB];
      f32x16 acc[TB];
#pragma unroll
      for (int t = 0; t < TB; ++t) {
        const int ct = wave + (t0 + t) * KW;
        const int krow = min(ct * 32 + l31, Kp - 1);               // rows past K: clamped, their columns are never read
        brow[t] = Wf + (size_t)krow * COUT + 4 * hf;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
      }
#pragma unroll
      for (int b0 = 0; b0 < NKB; b0 += HB) {
        float4 w[TB][HB];
#pragma unroll
        for (int t = 0; t < TB; ++t)
#pragma unroll
          for (int u = 0; u < HB; ++u) w[t][u] = *reinterpret_cast<const float4*>(brow[t] + 8 * (b0 + u));
#pragma unroll
        for (int u = 0; u < HB; ++u) {
          const float4 a = *reinterpret_cast<const float4*>(arow + 8 * (b0 + u));
#pragma unroll
          for (int t = 0; t < TB; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, w[t][u].x, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, w[t][u].y, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, w[t][u].z, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, w[t][u].w, acc[t], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int t = 0; t < TB; ++t) {
        const int ct = wave + (t0 + t) * KW;
        if (ct < NCT) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            s_z[((r & 3) + 8 * (r >> 2) + 4 * hf) * LDZ + ct * 32 + l31] = acc[t][r];
        }
      }
    }
  } else {
    // 16 rows: 16-column tiles on v_mfma_f32_16x16x4_f32, lane = (row / column l15, k-slot kq); a lane's 16-B load of
    // Wf row (ct 16 + l15) at k = 16 kb + 4 kq feeds four MFMAs, the matching A operand is the g tile's
    // [l15][16 kb + 4 kq ..].  A wave's tiles are independent accumulator chains (all of them in one round, their MFMAs
    // interleaved: dependent latency 40 cycles against a 32-cycle issue interval).
    // The weights depend on nothing the tile computes: the first TWO batches are requested at the very top of the
    // kernel, ahead of the g tile (a Cout = 32 layer's whole 72 KB), so their L2 round trip -- 3-5 k cycles with three
    // other workgroups' row gathers queued on the CU's memory pipeline -- runs under the g tile's own loads and barrier
    // instead of in front of the first MFMA; every further batch is requested two batches ahead into the registers the
    // batch just issued has freed.
    const int kq = lane >> 4, l15 = lane & 15;
    constexpr int NCT16 = (K + 15) / 16;
    constexpr int NKB = COUT / 16;                                 // 16-deep k-blocks
    constexpr int TB = (NCT16 + KW - 1) / KW;                      // column tiles per wave: one round
    constexpr int HB = TB > 5 ? 1 : (NKB < 2 ? NKB : 2);           // k-blocks per batch of weight loads
    constexpr int NB = NKB / HB;
    static_assert(TB <= 9, "accumulators of all of a wave's tiles at once");
    static_assert(NKB % HB == 0, "whole batches");
    const float* brow[TB];
#pragma unroll
    for (int t = 0; t < TB; ++t) {
      const int ct = wave + t * KW;
      const int krow = min(ct * 16 + l15, Kp - 1);                 // rows past K: clamped, their columns are never read
      brow[t] = Wf + (size_t)krow * COUT + 4 * kq;
    }
    float4 w[2][TB][HB];
    auto load_batch = [&](int b) {
#pragma unroll
      for (int t = 0; t < TB; ++t)
#pragma unroll
        for (int u = 0; u < HB; ++u) w[b & 1][t][u] = *reinterpret_cast<const float4*>(brow[t] + 16 * (b * HB + u));
    };
    load_batch(0);
    if constexpr (NB > 1) load_batch(1);
    g_tile();
    const float* arow = s_g + l15 * GL + 4 * kq;
    f32x4 acc[TB];
#pragma unroll
    for (int t = 0; t < TB; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
      for (int u = 0; u < HB; ++u) {
        const float4 a = *reinterpret_cast<const float4*>(arow + 16 * (b * HB + u));
#pragma unroll
        for (int t = 0; t < TB; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w[b & 1][t][u].x, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < TB; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w[b & 1][t][u].y, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < TB; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w[b & 1][t][u].z, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < TB; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w[b & 1][t][u].w, acc[t], 0, 0, 0);
      }
      if (b + 2 < NB) load_batch(b + 2);
    }
#pragma unroll
    for (int t = 0; t < TB; ++t) {
      const int ct = wave + t * KW;
      if (ct < NCT16) {
#pragma unroll
        for (int r = 0; r < 4; ++r) s_z[(4 * kq + r) * LDZ + ct * 16 + l15] = acc[t][r];
      }
    }
  }
  GEOBI_STAMP_BWD(2);
  __syncthreads();
  GEOBI_STAMP_BWD(3);

  // ---- row pass over the tile's nodes, lane = edge (feast_dev.h): the dz rows come from the LDS tile
  if constexpr (RP == 1) {
    static_assert(C == 64 && LC == 0, "staged row pass: 64 channels, node-level logits");
    const int wv = __builtin_amdgcn_readfirstlane(wave);      // the landing zone's address is wave-uniform (M0)
    rowpass_edge_node_staged<C, LDZ>(s_z + wv * NPW * LDZ, xa, xb, Ca, p, cvec, rowptr, col, N,
                                     tile * RT + wave * NPW + lane / G, lane / G, lane % G, dl, dpn, dcs, ld_dcs);
  } else {
    rowpass_edge_node<C, LC>(s_z + (wave * NPW + lane / G) * LDZ, xa, xb, Ca, p, cvec, s_u, rowptr, col, N,
                             tile * RT + wave * NPW + lane / G, lane % G, dl, dpn, dcs, ld_dcs);
  }
  GEOBI_STAMP_BWD(4);
}

// ------------------------------------------------------------------------------------------------------------------
// The same first half of the backward for layers reading 128 channels.  A [32][9 x 128] dz tile does not fit, and the
// dz GEMM of these layers (gemm_nn) was writing 4.6 KB per node to HBM for the row pass to read back.  Here the
// channels go in four chunks of 32: per chunk the matrix phase forms dz[32 nodes][9 heads x 32 channels] in LDS (one
// 32-column MFMA tile per head), the row pass adds that chunk's share to the nine dot products of every item (in-edge
// or self loop), and the softmax backward runs once all four chunks are in.  The partial dot products of a node's
// first 16 items wait in registers; nodes with more items park the rest in the rows they overwrite at the end anyway
// (dl rows of the edges, the node's dcs row for the self loop).
// Nine waves per workgroup: one dz tile (head) per wave in the matrix phase -- with eight, one wave had two tiles per
// chunk and the other seven waited at the barrier.  The ninth wave takes no part in the row pass.  100 registers per
// lane: five waves per SIMD, so two such workgroups still share a CU.
constexpr int KT9 = 64 * H;
template <int COUT, int C = 128>
__global__ __launch_bounds__(KT9, 5) void feast_rowpass_fused128_kernel(
    const float* __restrict__ xa, const float* __restrict__ xb, int Ca, const float* __restrict__ p,
    const float* __restrict__ cvec, const int* __restrict__ rowptr, const int* __restrict__ col, int N,
    const float* __restrict__ gout, const float* __restrict__ out_act, float slope, const float* __restrict__ Wf,
    float* __restrict__ g_out, float* __restrict__ dl, float* __restrict__ dpn, float* __restrict__ dcs, int ld_dcs) {
  constexpr int CCH = 32, NCC = C / CCH;
  static_assert(C % CCH == 0, "whole channel chunks");
  constexpr int KC = H * CCH;                    // dz columns per chunk: head h at [32 h, 32 h + 32)
  constexpr int LDZ = KC + 4;                    // + 4: the four node groups of a wave read distinct bank quads
  constexpr int GL = COUT + 4;
  constexpr int NQ = CCH / 4;                    // 16-B pieces per head and chunk
  constexpr int NPIECE = H * NQ, NSLOT = (NPIECE + G - 1) / G;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_g = smem;                             // [32][GL]
  float* s_z = smem + TN * GL;                   // [32][LDZ]

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = xcd_block(blockIdx.x, gridDim.x);
  if (tile * TN >= N) return;
#ifdef GEOBI_FUSED_STAMPS
  // diagnostic build: cycles of thread 0 per phase, summed over the channel chunks (tools/rp128_stamps.py)
  unsigned long long t_last = __builtin_amdgcn_s_memtime(), t_acc[4] = {0, 0, 0, 0};
#define GEOBI_RP128_MARK(i) do { if (threadIdx.x == 0) { const unsigned long long now = __builtin_amdgcn_s_memtime(); t_acc[i] += now - t_last; t_last = now; } } while (0)
#else
#define GEOBI_RP128_MARK(i) do { } while (0)
#endif

  // ---- g tile (as in the kernel above)
  {
    constexpr int Q = COUT / 4;
    for (int i = threadIdx.x; i < TN * Q; i += KT9) {
      const int r = i / Q, c4 = (i - r * Q) * 4;
      const int gn = tile * TN + r;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gn < N) {
        v = *reinterpret_cast<const float4*>(gout + (size_t)gn * COUT + c4);
        if (out_act != nullptr) {
          const float4 o = *reinterpret_cast<const float4*>(out_act + (size_t)gn * COUT + c4);
          v.x = o.x > 0.f ? v.x : v.x * slope; v.y = o.y > 0.f ? v.y : v.y * slope;
          v.z = o.z > 0.f ? v.z : v.z * slope; v.w = o.w > 0.f ? v.w : v.w * slope;
          *reinterpret_cast<float4*>(g_out + (size_t)gn * COUT + c4) = v;
        }
      }
      *reinterpret_cast<float4*>(s_g + r * GL + c4) = v;
    }
  }

  // ---- row-pass roles: group of 16 lanes <-> node, lane k <-> one item per chunk of 16 items
  const int g = lane / G, k = lane % G;
  const int nl = (wave < NW ? wave : 0) * NPW + g;            // wave 8: matrix phase only
  const int node = tile * TN + nl;
  const bool valid = node < N && wave < NW;
  const int ns = valid ? node : N - 1;
  const int rs = rowptr[ns];
  const int deg = valid ? rowptr[ns + 1] - rs : -1;            // items = deg edges + the self loop; none if invalid
  const int Cb = C - Ca;
  float* selfrow = dcs + (size_t)ns * ld_dcs;                  // where a self loop outside the first 16 items parks
  float sv0[H];                                                // partial dot products of the node's first 16 items
#pragma unroll
  for (int h = 0; h < H; ++h) sv0[h] = 0.f;
  const int j0 = k < deg ? col[rs + k] : ns;                   // the first 16 items' neighbour, kept across the chunks

  const int hf = lane >> 5, l31 = lane & 31;
  // Matrix-phase weights: head t's rows of Wf for this channel chunk, COUT / 8 float4 per lane.  Cout <= 64: all of a
  // chunk's (<= 8) are requested at once BEFORE the barrier that ends the previous chunk's row pass (before the g-tile
  // barrier for chunk 0) -- one round trip per chunk, mostly under the barrier: 111.3 -> 108.6 us at 712 tiles.  Cout = 128
  // (16 per chunk): batches of four as before; a ring of 8 with pinned refills measured 57 -> 86 us there (the scheduler
  // already moves a batch's loads above the previous batch's MFMAs where it pays; profiles/r04_weight_ring.txt).
  // tools/rp128_stamps.py: the matrix phases are 26-32 k of a tile's 58 k cycles at 128 -> 64 -- the nine dz tiles of a
  // chunk are 18 k wave-cycles of MFMA per SIMD and tile, shared with the other resident workgroup's row pass.
  constexpr int NKB = COUT / 8, HB = 4;
  constexpr bool AHEAD = NKB <= 8;
  static_assert(NKB % HB == 0, "whole weight batches");
  float4 wa[AHEAD ? NKB : 1];
  auto issue_w = [&](int cc) {
    if constexpr (AHEAD) {
      const float* brow = Wf + (size_t)(wave * C + cc * CCH + l31) * COUT + 4 * hf;
#pragma unroll
      for (int u = 0; u < NKB; ++u) wa[u] = *reinterpret_cast<const float4*>(brow + 8 * u);
    }
  };
  issue_w(0);
  __syncthreads();
  GEOBI_RP128_MARK(0);

  static_for<0, NCC>([&](auto cci) {
    constexpr int cc = decltype(cci)::value;
    // ---- matrix phase: head h's 32 columns of this channel chunk = dz tile h = wave h's (nine waves, nine heads)
    {
      const float* arow = s_g + l31 * GL + 4 * hf;
      const float* brow = Wf + (size_t)(wave * C + cc * CCH + l31) * COUT + 4 * hf;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int b0 = 0; b0 < NKB; b0 += HB) {
        float4 wv[HB];
#pragma unroll
        for (int u = 0; u < HB; ++u) {
          if constexpr (AHEAD) wv[u] = wa[b0 + u];
          else wv[u] = *reinterpret_cast<const float4*>(brow + 8 * (b0 + u));
        }
#pragma unroll
        for (int u = 0; u < HB; ++u) {
          const float4 a = *reinterpret_cast<const float4*>(arow + 8 * (b0 + u));
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, wv[u].x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, wv[u].y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, wv[u].z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, wv[u].w, acc, 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) s_z[((r & 3) + 8 * (r >> 2) + 4 * hf) * LDZ + wave * 32 + l31] = acc[r];
    }
    __syncthreads();
    GEOBI_RP128_MARK(1);
    // ---- this chunk's share of the dot products, lane = item; every lane of a group runs the FMAs (row broadcasts)
    {
      float dzr[NSLOT][4];
      const float* zrow = s_z + nl * LDZ;
#pragma unroll
      for (int sl = 0; sl < NSLOT; ++sl) {
        const int pidx = sl * G + k;
        if (pidx < NPIECE) load_piece<4>(zrow + pidx * 4, dzr[sl]);
        else { dzr[sl][0] = 0.f; dzr[sl][1] = 0.f; dzr[sl][2] = 0.f; dzr[sl][3] = 0.f; }
      }
      for (int base = 0; base <= deg; base += G) {
        const int idx = base + k;
        const bool real = idx < deg, self = idx == deg;
        const int e = rs + idx;
        const int j = base == 0 ? j0 : (real ? col[e] : ns);
        const float* src = (cc * CCH < Ca) ? xa + (size_t)j * Ca + cc * CCH : xb + (size_t)j * Cb + (cc * CCH - Ca);
        float* park = real ? dl + (size_t)e * HP : selfrow;
        float sv[H];
        if (base == 0) {
#pragma unroll
          for (int h = 0; h < H; ++h) sv[h] = sv0[h];
        } else {
          float t[HP];
#pragma unroll
          for (int h = 0; h < HP; ++h) t[h] = 0.f;
          if (cc > 0 && (real || self)) load_row<HP>(park, t);
#pragma unroll
          for (int h = 0; h < H; ++h) sv[h] = t[h];
        }
        static_for<0, NQ / 4>([&](auto bi) {
          constexpr int q0 = decltype(bi)::value * 4;
          float xj[4][4];
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) load_piece<4>(src + (q0 + qd) * 4, xj[qd]);
          static_for<0, 4>([&](auto qi) {
            constexpr int qd = decltype(qi)::value;
            static_for<0, H / 3>([&](auto gi) {
              constexpr int h = decltype(gi)::value * 3;
              constexpr int p0 = h * NQ + q0 + qd, p1 = p0 + NQ, p2 = p1 + NQ;
              fmac_bcast3<p0 % G, p1 % G, p2 % G>(sv[h], sv[h + 1], sv[h + 2], dzr[p0 / G], dzr[p1 / G], dzr[p2 / G], xj[qd]);
            });
          });
        });
        if (base == 0) {
#pragma unroll
          for (int h = 0; h < H; ++h) sv0[h] = sv[h];
        } else if (real || self) {
          float4* pr = reinterpret_cast<float4*>(park);
          pr[0] = make_float4(sv[0], sv[1], sv[2], sv[3]);
          pr[1] = make_float4(sv[4], sv[5], sv[6], sv[7]);
          pr[2] = make_float4(sv[8], 0.f, 0.f, 0.f);
        }
      }
    }
    if constexpr (cc + 1 < NCC) {
      if constexpr (AHEAD) __builtin_amdgcn_sched_barrier(0);  // not earlier: the row pass needs its registers
      issue_w(cc + 1);                                         // in flight across the barrier
      __syncthreads();                                         // the next chunk overwrites the dz tile
    }
    GEOBI_RP128_MARK(2);
  });

  // ---- softmax backward with the complete dot products
  float cc9[H];
#pragma unroll
  for (int h = 0; h < H; ++h) cc9[h] = cvec[h];
  const float invd = 1.0f / (float)(deg + 1);
  float dsum[H], d[H];
#pragma unroll
  for (int h = 0; h < H; ++h) { dsum[h] = 0.f; d[h] = 0.f; }
  bool self = false;
  for (int base = 0; base <= deg; base += G) {
    const int idx = base + k;
    const bool real = idx < deg;
    self = idx == deg;
    const int e = rs + idx;
    float sv[H];
    if (base == 0) {
#pragma unroll
      for (int h = 0; h < H; ++h) sv[h] = sv0[h];
    } else {
      float t[HP];
#pragma unroll
      for (int h = 0; h < HP; ++h) t[h] = 0.f;
      if (real || self) load_row<HP>(real ? dl + (size_t)e * HP : selfrow, t);
#pragma unroll
      for (int h = 0; h < H; ++h) sv[h] = t[h];
    }
    float q[H];
    if (real) {
      const int j = base == 0 ? j0 : col[e];
      float pc[H], pn[H];
      load_hp(p + (size_t)ns * HP, pc);
      load_hp(p + (size_t)j * HP, pn);
#pragma unroll
      for (int h = 0; h < H; ++h) q[h] = pn[h] - pc[h] + cc9[h];
    } else {
#pragma unroll
      for (int h = 0; h < H; ++h) q[h] = cc9[h];
    }
    softmax9(q);
    float tq = 0.f;
#pragma unroll
    for (int h = 0; h < H; ++h) tq = fmaf(q[h], sv[h], tq);
#pragma unroll
    for (int h = 0; h < H; ++h) d[h] = q[h] * (sv[h] - tq) * invd;
    if (real) {
      float4* drow = reinterpret_cast<float4*>(dl + (size_t)e * HP);
      drow[0] = make_float4(d[0], d[1], d[2], d[3]);
      drow[1] = make_float4(d[4], d[5], d[6], d[7]);
      drow[2] = make_float4(d[8], 0.f, 0.f, 0.f);
#pragma unroll
      for (int h = 0; h < H; ++h) dsum[h] += d[h];
    }
  }
  float dself[H];
#pragma unroll
  for (int h = 0; h < H; ++h) {
    dsum[h] = group_allreduce<G>(dsum[h]);
    dself[h] = group_allreduce<G>(self ? d[h] : 0.f);
  }
#ifdef GEOBI_FUSED_STAMPS
  GEOBI_RP128_MARK(3);
  if (threadIdx.x == 0 && blockIdx.x < 16384)
    for (int i = 0; i < 4; ++i) g_stamps_bwd[blockIdx.x][i] = t_acc[i];
#endif
  if (!valid || k != 0) return;
  float4* a4 = reinterpret_cast<float4*>(dpn + (size_t)node * HP);
  a4[0] = make_float4(dsum[0], dsum[1], dsum[2], dsum[3]);
  a4[1] = make_float4(dsum[4], dsum[5], dsum[6], dsum[7]);
  a4[2] = make_float4(dsum[8], 0.f, 0.f, 0.f);
  float4* bq = reinterpret_cast<float4*>(dcs + (size_t)node * ld_dcs);
  bq[0] = make_float4(dsum[0] + dself[0], dsum[1] + dself[1], dsum[2] + dself[2], dsum[3] + dself[3]);
  bq[1] = make_float4(dsum[4] + dself[4], dsum[5] + dself[5], dsum[6] + dself[6], dsum[7] + dself[7]);
  bq[2] = make_float4(dsum[8] + dself[8], 0.f, 0.f, 0.f);
}

// Behind the packed weights of either kernel mode: QSELF floats = softmax(c), the attention weights of the self loop
// (u (x_i - x_i) + c = c exactly) -- the same nine numbers for every node of a layer, which every wave of the fused kernels
// used to recompute at its start (nine exps + the softmax around them: ~85 of a wave's ~730 vector instructions per tile).
constexpr int QSELF = 12;
__device__ __forceinline__ float pack_qself(const float* __restrict__ c, int i) {
  if (i >= H) return 0.f;
  float q[H];
#pragma unroll
  for (int h = 0; h < H; ++h) q[h] = c[h];
  softmax9(q);
  float v = 0.f;
#pragma unroll
  for (int h = 0; h < H; ++h) v = h == i ? q[h] : v;
  return v;
}

// Packed weights of the forward:  Bp[kb][half][col][s] = lin.weight[h * Cout + col, kin] for k = 8 kb + 4 half + s
// = h * Cin + kin (zero for k >= 9 Cin or col >= Cout), NP = padded column count (multiple of 32).
__global__ void pack_fused_fwd_kernel(const float* __restrict__ lin_w, const float* __restrict__ c, int Cin, int Cout,
                                      int KD, int NP, float* __restrict__ bp) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= KD * NP) {
    if (idx < KD * NP + QSELF) bp[idx] = pack_qself(c, idx - KD * NP);
    return;
  }
  const int s = idx & 3, colx = (idx >> 2) % NP, rest = (idx >> 2) / NP;
  const int hf = rest & 1, kb = rest >> 1;
  const int k = 8 * kb + 4 * hf + s;
  float v = 0.f;
  if (k < H * Cin && colx < Cout) v = lin_w[((size_t)(k / Cin) * Cout + colx) * Cin + (k % Cin)];
  bp[idx] = v;
}

// Packed weights of dx = r' W':  rows k < 9 Cout: lin.weight[k, col]; the next 9: u.weight[k - 9 Cout, col];
// the remaining 15 rows ([dp] padding and the dcs columns of r') are zero.
__global__ void pack_fused_dx_kernel(const float* __restrict__ lin_w, const float* __restrict__ u_w,
                                     const float* __restrict__ c, int Cin, int Cout, int KD, int NP,
                                     float* __restrict__ bp) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= KD * NP) {
    if (idx < KD * NP + QSELF) bp[idx] = pack_qself(c, idx - KD * NP);
    return;
  }
  const int s = idx & 3, colx = (idx >> 2) % NP, rest = (idx >> 2) / NP;
  const int hf = rest & 1, kb = rest >> 1;
  const int k = 8 * kb + 4 * hf + s;
  float v = 0.f;
  if (colx < Cin) {
    if (k < H * Cout) v = lin_w[(size_t)k * Cin + colx];
    else if (k < H * Cout + H) v = u_w[(size_t)(k - H * Cout) * Cin + colx];
  }
  bp[idx] = v;
}

// Training forward: every packed form the layer's forward AND backward need, in one launch --
//   wf  [Kp, Cout]      row h Cin + k = lin.weight[h Cout + o, k]      (dz = g Wf^T in the backward)
//   bf  fragment-ordered forward weights (pack_fused_fwd_kernel),  bdx  fragment-ordered dx weights.
__global__ void pack_fused_all_kernel(const float* __restrict__ lin_w, const float* __restrict__ u_w,
                                      const float* __restrict__ c, int Cin, int Cout, int Kp, int KDf, int NPf, int KDx,
                                      int NPx, float* __restrict__ wf, float* __restrict__ bf, float* __restrict__ bdx) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int n_wf = Kp * Cout, n_bf = KDf * NPf, n_bx = KDx * NPx;
  if (idx >= n_wf + n_bf + n_bx) {           // the two softmax(c) tails
    const int t = idx - (n_wf + n_bf + n_bx);
    if (t < QSELF) bf[n_bf + t] = pack_qself(c, t);
    else if (t < 2 * QSELF) bdx[n_bx + t - QSELF] = pack_qself(c, t - QSELF);
    return;
  }
  if (idx < n_wf) {
    const int kk = idx / Cout, o = idx % Cout;
    const int h = kk / Cin, k = kk % Cin;
    wf[idx] = (h < H) ? lin_w[((size_t)h * Cout + o) * Cin + k] : 0.f;
    return;
  }
  idx -= n_wf;
  if (idx < n_bf) {
    const int s = idx & 3, colx = (idx >> 2) % NPf, rest = (idx >> 2) / NPf;
    const int k = 8 * (rest >> 1) + 4 * (rest & 1) + s;
    float v = 0.f;
    if (k < H * Cin && colx < Cout) v = lin_w[((size_t)(k / Cin) * Cout + colx) * Cin + (k % Cin)];
    bf[idx] = v;
    return;
  }
  idx -= n_bf;
  if (idx < n_bx) {
    const int s = idx & 3, colx = (idx >> 2) % NPx, rest = (idx >> 2) / NPx;
    const int k = 8 * (rest >> 1) + 4 * (rest & 1) + s;
    float v = 0.f;
    if (colx < Cin) {
      if (k < H * Cout) v = lin_w[(size_t)k * Cin + colx];
      else if (k < H * Cout + H) v = u_w[(size_t)(k - H * Cout) * Cin + colx];
    }
    bdx[idx] = v;
  }
}

// Every packed form of SEVERAL layers in one launch (a branch's eight FeaSt layers: 8 launches of ~4 us -> 1).
// all != 0: wf | bf | bdx per layer (training);  all == 0: bf only (inference).
struct PackBatch {
  FusedPackItem it[kMaxPackBatch];
  int64_t start[kMaxPackBatch + 1];      // element offsets of the layers in the launch's index space
  int n, all;
};

__global__ void pack_fused_batch_kernel(PackBatch pb) {
  int64_t gidx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gidx >= pb.start[pb.n]) return;
  int l = 0;
  while (gidx >= pb.start[l + 1]) ++l;
  const FusedPackItem& d = pb.it[l];
  int idx = (int)(gidx - pb.start[l]);
  const int Cin = d.Cin, Cout = d.Cout;
  const int Kp = (H * Cin + 3) / 4 * 4;          // feast_ldz
  const int KDf = fused_k(Cin, 0), NPf = 32 * (Cout <= 32 ? 1 : (Cout <= 64 ? 2 : 4));
  const int KDx = fused_k(Cout, 1), NPx = 32 * (Cin <= 32 ? 1 : (Cin <= 64 ? 2 : 4));
  const int n_wf = pb.all ? Kp * Cout : 0, n_bf = KDf * NPf, n_bx = pb.all ? KDx * NPx : 0;
  if (idx >= n_wf + n_bf + n_bx) {           // the softmax(c) tails behind bf (and bdx)
    const int t = idx - (n_wf + n_bf + n_bx);
    if (t < QSELF) d.bf[n_bf + t] = pack_qself(d.c, t);
    else if (pb.all && t < 2 * QSELF) d.bdx[n_bx + t - QSELF] = pack_qself(d.c, t - QSELF);
    return;
  }
  if (idx < n_wf) {
    const int kk = idx / Cout, o = idx % Cout;
    const int h = kk / Cin, k = kk % Cin;
    d.wf[idx] = (h < H) ? d.lin_w[((size_t)h * Cout + o) * Cin + k] : 0.f;
    return;
  }
  idx -= n_wf;
  if (idx < n_bf) {
    const int sidx = idx & 3, colx = (idx >> 2) % NPf, rest = (idx >> 2) / NPf;
    const int k = 8 * (rest >> 1) + 4 * (rest & 1) + sidx;
    float v = 0.f;
    if (k < H * Cin && colx < Cout) v = d.lin_w[((size_t)(k / Cin) * Cout + colx) * Cin + (k % Cin)];
    d.bf[idx] = v;
    return;
  }
  idx -= n_bf;
  if (idx < n_bx) {
    const int sidx = idx & 3, colx = (idx >> 2) % NPx, rest = (idx >> 2) / NPx;
    const int k = 8 * (rest >> 1) + 4 * (rest & 1) + sidx;
    float v = 0.f;
    if (colx < Cin) {
      if (k < H * Cout) v = d.lin_w[(size_t)k * Cin + colx];
      else if (k < H * Cout + H) v = d.u_w[(size_t)(k - H * Cout) * Cin + colx];
    }
    d.bdx[idx] = v;
  }
}

// Tile geometry of the fused kernels: GEOBI_TILE16 (read once): 1 = 16-row tiles / four workgroups per CU (default),
// 0 = 32-row tiles / two workgroups per CU.  Same-box A/B knob; both forms stay under the parity tests.
int g_tile_rows = 0;          // 0: not set yet -> GEOBI_TILE16 decides; 16 / 32 once geobi_set_tile_rows was called
// Forms of the fused backward row pass at 64 input channels (-1: environment / built-in default; set_rowpass_form):
//   staged   1 (default; GEOBI_ROWPASS_STAGED): neighbour rows staged through LDS, 0: lane-private row reads
//   chunked  1: the channel-chunked kernel (32-node tiles, two chunks of 32 channels) for every 64-channel layer,
//            0: for none; default (GEOBI_ROWPASS_CHUNKED64 unset): for Cout = 128 only -- measured 31 against 36 us
//            there, 102 against 86 us at Cout = 32
int g_rp_staged = -1, g_rp_chunked = -1;
bool rp_staged() {
  static const bool env_on = [] { const char* f = getenv("GEOBI_ROWPASS_STAGED"); return !f || atoi(f) != 0; }();
  return g_rp_staged < 0 ? env_on : g_rp_staged != 0;
}
bool rp_chunked64(int Cout) {
  static const int env = [] { const char* f = getenv("GEOBI_ROWPASS_CHUNKED64"); return f ? (atoi(f) != 0 ? 1 : 0) : -1; }();
  const int v = g_rp_chunked < 0 ? env : g_rp_chunked;
  return v < 0 ? Cout == 128 : v != 0;
}
bool tile16() {
  static const bool env_on = [] { const char* f = getenv("GEOBI_TILE16"); return !f || atoi(f) != 0; }();
  return g_tile_rows ? g_tile_rows == 16 : env_on;
}

// Column parts of a launch (feast_fused_kernel's CS): 2 when the layer reads 128 channels and the 16-row tiles alone would
// fill less than half of the chip's workgroup slots -- level 2 of the bench batch, 264-424 tiles on 1 024 slots, where a
// tile's matrix phase (one wave per SIMD, 0.3-0.6 MB of weights streamed per tile) is what the kernel's time follows:
// 128 -> 128 dx 43.7 -> 33.0 us at 264 tiles, 65.2 -> 54.2 at 424.  Measured slower for 64-channel layers and from 768 tiles
// up (profiles/r04_weight_ring.txt).  Same bits either way, so the choice may depend on N.
std::atomic<int> g_col_parts{-1};          // -1: from the environment on first use; 0: per launch; 1 / 2: forced
int g_col_parts_max_tiles = 512;
int column_parts(int tiles) {
  int v = g_col_parts.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* f = getenv("GEOBI_COLUMN_PARTS");
    v = f ? atoi(f) : 0;
    if (v < 0 || v > 2) v = 0;
    if (const char* t = getenv("GEOBI_COLUMN_PARTS_MAX_TILES")) g_col_parts_max_tiles = atoi(t);
    g_col_parts.store(v, std::memory_order_relaxed);
  }
  if (v != 0) return v;
  return tiles <= g_col_parts_max_tiles ? 2 : 1;
}

// GEOBI_WARM_L2=0: without the up-front request of the packed weights (feast_dev.h: warm_l2) -- same-box A/B
bool warm_l2_enabled() {
  static const bool on = [] { const char* f = getenv("GEOBI_WARM_L2"); return !f || atoi(f) != 0; }();
  return on;
}

template <int C, int MODE, int LC, int NT, int ROWS, int CS = 1>
int launch_one(const float* xa, const float* xb, int Ca, const float* p, const float* cvec, const int* rowptr,
               const int* col, const int* deg_rowptr, int N, const float* xl, const float* ul, const float* dpd,
               const float* dl, const int* pos, const float* dpn, const float* Bp, int NOUT, const float* bias, float slope, float* out, int ldo, float* out1, int split,
               int ldo1, float* tile_out, hipStream_t s) {
  constexpr size_t lds = (size_t)fused_lds_floats<C, MODE, LC, ROWS>() * sizeof(float);
  static_assert(lds <= 163840, "tile exceeds the LDS of a CU");
  static_assert(ROWS == 32 || lds <= 40960, "16-row tiles: four workgroups per CU");
  static std::atomic<bool> attr_set{false};   // several host threads may launch (one per mesh group)
  if (!attr_set) {
    GEOBI_HIP(hipFuncSetAttribute((const void*)feast_fused_kernel<C, MODE, LC, NT, ROWS, CS>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipEvent_t ev_a, ev_b;
  if (MODE == 0 && prof_take_launch_events(&ev_a, &ev_b)) {
    // profiling pass of the bench (geobi_prof_enable): the two events are bound to THIS dispatch
    hipExtLaunchKernelGGL((feast_fused_kernel<C, MODE, LC, NT, ROWS, CS>), dim3(xcd_grid(cdiv(N, ROWS)), CS), dim3(16 * ROWS),
                          lds, s, ev_a, ev_b, 0, xa, xb, Ca, p, cvec, rowptr, col, deg_rowptr, N, xl, ul, dpd, dl, pos, dpn,
                          Bp, NOUT, bias, slope, out, ldo, out1, split, ldo1, tile_out, warm_l2_enabled() ? 1 : 0);
    GEOBI_LAUNCH_OK();
    return 0;
  }
  feast_fused_kernel<C, MODE, LC, NT, ROWS, CS><<<dim3(xcd_grid(cdiv(N, ROWS)), CS), 16 * ROWS, lds, s>>>(
      xa, xb, Ca, p, cvec, rowptr, col, deg_rowptr, N, xl, ul, dpd, dl, pos, dpn, Bp, NOUT, bias, slope, out, ldo, out1,
      split, ldo1, tile_out, warm_l2_enabled() ? 1 : 0);
  GEOBI_LAUNCH_OK();
  return 0;
}

#define GEOBI_FUSED_ARGS                                                                                            \
  xa, xb, Ca, p, cvec, rowptr, col, deg_rowptr, N, xl, ul, dpd, dl, pos, dpn, Bp, NOUT, bias, slope, out, ldo, out1, \
      split, ldo1, tile_out, s

template <int C, int MODE, int LC>
int launch_nt(int NT, const float* xa, const float* xb, int Ca, const float* p, const float* cvec, const int* rowptr,
              const int* col, const int* deg_rowptr, int N, const float* xl, const float* ul, const float* dpd,
              const float* dl, const int* pos, const float* dpn, const float* Bp, int NOUT, const float* bias, float slope, float* out, int ldo, float* out1, int split,
              int ldo1, float* tile_out, hipStream_t s) {
  // dx with per-edge logits writes a 6- or 12-channel input gradient: one column tile
  constexpr bool kNarrowOnly = MODE == 1 && LC > 0;
  if (tile16()) {
    // two column parts: 128 channels read, node-level logits (0.3-0.6 MB of weights per tile)
    if constexpr (C >= 128 && LC == 0) {
      if (column_parts(cdiv(N, 16)) == 2) {
        if (NT == 1) return launch_one<C, MODE, LC, 1, 16, 2>(GEOBI_FUSED_ARGS);
        if (NT == 2) return launch_one<C, MODE, LC, 2, 16, 2>(GEOBI_FUSED_ARGS);
        if (NT == 4) return launch_one<C, MODE, LC, 4, 16, 2>(GEOBI_FUSED_ARGS);
      }
    }
    if (NT == 1) return launch_one<C, MODE, LC, 1, 16>(GEOBI_FUSED_ARGS);
    if constexpr (!kNarrowOnly) {
      if (NT == 2) return launch_one<C, MODE, LC, 2, 16>(GEOBI_FUSED_ARGS);
      if (NT == 4) return launch_one<C, MODE, LC, 4, 16>(GEOBI_FUSED_ARGS);
    }
  } else {
    if (NT == 1) return launch_one<C, MODE, LC, 1, 32>(GEOBI_FUSED_ARGS);
    if constexpr (!kNarrowOnly) {
      if (NT == 2) return launch_one<C, MODE, LC, 2, 32>(GEOBI_FUSED_ARGS);
      if (NT == 4) return launch_one<C, MODE, LC, 4, 32>(GEOBI_FUSED_ARGS);
    }
  }
  return set_error("feast fused: unsupported output width %d", NOUT);
}

}  // namespace

#ifdef GEOBI_FUSED_STAMPS
extern "C" int geobi_debug_stamps(void* host_dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_stamps), bytes, 0, hipMemcpyDeviceToHost);
}
extern "C" int geobi_debug_stamps_bwd(void* host_dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_stamps_bwd), bytes, 0, hipMemcpyDeviceToHost);
}
#endif

namespace {
template <int C, int LC, int COUT, int RT, int RP = 0>
int launch_rowpass_fused_rt(const float* xa, const float* xb, int Ca, const float* p, const float* cvec, const int* rowptr,
                         const int* col, int N, const float* ul, const float* gout, const float* out_act, float slope,
                         const float* Wf, int Kp, float* g_out, float* dl, float* dpn, float* dcs, int ld_dcs,
                         hipStream_t s) {
  constexpr int NCT = (H * C + 31) / 32;
  constexpr int LDZ = NCT * 32 + 4;
  constexpr size_t lds = ((size_t)RT * (COUT + 4) + (size_t)RT * LDZ + (LC > 0 ? LC * HP : 0)) * sizeof(float);
  static_assert(lds <= 163840, "tiles exceed the LDS of a CU");
  static std::atomic<bool> attr_set{false};   // several host threads may launch (one per mesh group)
  if (!attr_set) {
    GEOBI_HIP(hipFuncSetAttribute((const void*)feast_rowpass_fused_kernel<C, LC, COUT, RT, RP>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  feast_rowpass_fused_kernel<C, LC, COUT, RT, RP><<<xcd_grid(cdiv(N, RT)), 16 * RT, lds, s>>>(
      xa, xb, Ca, p, cvec, rowptr, col, N, ul, gout, out_act, slope, Wf, Kp, g_out, dl, dpn, dcs, ld_dcs);
  GEOBI_LAUNCH_OK();
  return 0;
}

template <int C, int LC, int COUT>
int launch_rowpass_fused(const float* xa, const float* xb, int Ca, const float* p, const float* cvec, const int* rowptr,
                         const int* col, int N, const float* ul, const float* gout, const float* out_act, float slope,
                         const float* Wf, int Kp, float* g_out, float* dl, float* dpn, float* dcs, int ld_dcs,
                         hipStream_t s) {
  if (tile16()) {
    if constexpr (C == 64 && LC == 0) {
      // GEOBI_ROWPASS_STAGED=0 / set_rowpass_form: the lane-private row reads (same-box A/B; results are bit-identical)
      if (rp_staged() && Ca % 32 == 0)
        return launch_rowpass_fused_rt<C, LC, COUT, 16, 1>(xa, xb, Ca, p, cvec, rowptr, col, N, ul, gout, out_act, slope, Wf,
                                                           Kp, g_out, dl, dpn, dcs, ld_dcs, s);
    }
    return launch_rowpass_fused_rt<C, LC, COUT, 16>(xa, xb, Ca, p, cvec, rowptr, col, N, ul, gout, out_act, slope, Wf, Kp,
                                                    g_out, dl, dpn, dcs, ld_dcs, s);
  }
  return launch_rowpass_fused_rt<C, LC, COUT, 32>(xa, xb, Ca, p, cvec, rowptr, col, N, ul, gout, out_act, slope, Wf, Kp,
                                                  g_out, dl, dpn, dcs, ld_dcs, s);
}
}  // namespace

namespace {
template <int COUT, int C = 128>
int launch_rowpass_fused128(const float* xa, const float* xb, int Ca, const float* p, const float* cvec,
                            const int* rowptr, const int* col, int N, const float* gout, const float* out_act,
                            float slope, const float* Wf, float* g_out, float* dl, float* dpn, float* dcs, int ld_dcs,
                            hipStream_t s) {
  constexpr size_t lds = ((size_t)TN * (COUT + 4) + (size_t)TN * (H * 32 + 4)) * sizeof(float);
  static_assert(lds <= 81920, "two workgroups per CU");
  static std::atomic<bool> attr_set{false};   // several host threads may launch (one per mesh group)
  if (!attr_set) {
    GEOBI_HIP(hipFuncSetAttribute((const void*)feast_rowpass_fused128_kernel<COUT, C>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  feast_rowpass_fused128_kernel<COUT, C><<<xcd_grid(cdiv(N, TN)), KT9, lds, s>>>(
      xa, xb, Ca, p, cvec, rowptr, col, N, gout, out_act, slope, Wf, g_out, dl, dpn, dcs, ld_dcs);
  GEOBI_LAUNCH_OK();
  return 0;
}
}  // namespace

// column parts of the fused kernel from one process (parity tests, A/B timing): 1 / 2, 0 = chosen per launch
int set_column_parts(int parts) {
  if (parts < 0 || parts > 2) return set_error("fused kernel column parts: 1, 2 or 0 (per launch), got %d", parts);
  column_parts(1);                           // the environment's tile limit is read once
  g_col_parts.store(parts, std::memory_order_relaxed);
  return 0;
}

// forms of the 64-channel backward row pass from one process (parity tests, A/B timing): 1 / 0, -1 = default
int set_rowpass_form(int staged, int chunked64) {
  if (staged < -1 || staged > 1 || chunked64 < -1 || chunked64 > 1)
    return set_error("row-pass form: staged and chunked64 are 1, 0 or -1 (default), got %d, %d", staged, chunked64);
  g_rp_staged = staged;
  g_rp_chunked = chunked64;
  return 0;
}

// both tile geometries from one process (parity tests, A/B timing): rows = 16 / 32, 0 = back to GEOBI_TILE16
int set_tile_rows(int rows) {
  if (rows != 0 && rows != 16 && rows != 32) return set_error("tile rows: 16, 32 or 0 (environment default), got %d", rows);
  g_tile_rows = rows;
  return 0;
}

// split inputs: the row is read in batches of 16 channels (32 at 128 channels), so the first part must end on such a
// boundary; 128 channels: Cout 64 or 128 (GEOBI_ROWPASS_FUSED128=0: the GEMM + standalone row pass)
bool feast_rowpass_fused_supported(int Cin, int Cb, int Cout) {
  if (Cin == 128) {
    static const bool on = [] { const char* f = getenv("GEOBI_ROWPASS_FUSED128"); return !f || atoi(f) != 0; }();   // A/B knob
    return on && (Cout == 64 || Cout == 128) && (Cb == 0 || (Cin - Cb) % 32 == 0);
  }
  return Cin <= 64 && (Cb == 0 || (Cin >= 32 && (Cin - Cb) % 16 == 0));
}

// g (written to g_out when slope != 1), dl, dpn, dcs of one layer in one launch; LC: per-edge logit channels (0 / 6 / 12)
int feast_rowpass_fused(const float* xa, const float* xb, int Ca, int Cin, const float* p, const float* cvec,
                        const int* rowptr, const int* col, int N, int LC, const float* ul, const float* gout,
                        const float* out_act, float slope, int Cout, const float* Wf, int Kp, float* g_out, float* dl,
                        float* dpn, float* dcs, int ld_dcs, hipStream_t s) {
#define GEOBI_RP_ARGS xa, xb, Ca, p, cvec, rowptr, col, N, ul, gout, out_act, slope, Wf, Kp, g_out, dl, dpn, dcs, ld_dcs, s
#define GEOBI_RP_COUT(C_, L_)                                                            \
  switch (Cout) {                                                                        \
    case 32: return launch_rowpass_fused<C_, L_, 32>(GEOBI_RP_ARGS);                     \
    case 64: return launch_rowpass_fused<C_, L_, 64>(GEOBI_RP_ARGS);                     \
    case 128: return launch_rowpass_fused<C_, L_, 128>(GEOBI_RP_ARGS);                   \
    default: return set_error("feast fused row pass: unsupported Cout=%d", Cout);        \
  }
  if (Cin == 128) {
    GEOBI_REQUIRE(LC == 0, "feast fused row pass: per-edge logits at 128 channels");
    if (Cout == 64)
      return launch_rowpass_fused128<64>(xa, xb, Ca, p, cvec, rowptr, col, N, gout, out_act, slope, Wf, g_out, dl, dpn,
                                         dcs, ld_dcs, s);
    if (Cout == 128)
      return launch_rowpass_fused128<128>(xa, xb, Ca, p, cvec, rowptr, col, N, gout, out_act, slope, Wf, g_out, dl, dpn,
                                          dcs, ld_dcs, s);
    return set_error("feast fused row pass: unsupported Cout=%d at 128 input channels", Cout);
  }
  {
    // the channel-chunked kernel (32-node tiles, half the weight bytes per node) at 64 input channels: see rp_chunked64
    if (Cin == 64 && LC == 0 && (Ca == Cin || Ca % 32 == 0) && rp_chunked64(Cout)) {
      if (Cout == 32)
        return launch_rowpass_fused128<32, 64>(xa, xb, Ca, p, cvec, rowptr, col, N, gout, out_act, slope, Wf, g_out, dl,
                                               dpn, dcs, ld_dcs, s);
      if (Cout == 64)
        return launch_rowpass_fused128<64, 64>(xa, xb, Ca, p, cvec, rowptr, col, N, gout, out_act, slope, Wf, g_out, dl,
                                               dpn, dcs, ld_dcs, s);
      if (Cout == 128)
        return launch_rowpass_fused128<128, 64>(xa, xb, Ca, p, cvec, rowptr, col, N, gout, out_act, slope, Wf, g_out, dl,
                                                dpn, dcs, ld_dcs, s);
    }
  }
  switch (Cin * 100 + LC) {
    case 600: GEOBI_RP_COUT(6, 0)
    case 606: GEOBI_RP_COUT(6, 6)
    case 1200: GEOBI_RP_COUT(12, 0)
    case 1212: GEOBI_RP_COUT(12, 12)
    case 3200: GEOBI_RP_COUT(32, 0)
    case 6400: GEOBI_RP_COUT(64, 0)
    default: return set_error("feast fused row pass: unsupported Cin=%d (per-edge logit channels %d)", Cin, LC);
  }
#undef GEOBI_RP_COUT
#undef GEOBI_RP_ARGS
}

int feast_fused_nt(int nout) { return nout <= 32 ? 1 : (nout <= 64 ? 2 : 4); }
// packed weights + the QSELF floats softmax(c) behind them
size_t feast_fused_fwd_pack_floats(int Cin, int Cout) { return (size_t)fused_k(Cin, 0) * 32 * feast_fused_nt(Cout) + QSELF; }
size_t feast_fused_dx_pack_floats(int Cin, int Cout) { return (size_t)fused_k(Cout, 1) * 32 * feast_fused_nt(Cin) + QSELF; }

int feast_fused_pack_fwd(const float* lin_w, const float* c, int Cin, int Cout, float* bp, hipStream_t s) {
  const int KD = fused_k(Cin, 0), NP = 32 * feast_fused_nt(Cout);
  pack_fused_fwd_kernel<<<cdiv((int64_t)KD * NP + QSELF, 256), 256, 0, s>>>(lin_w, c, Cin, Cout, KD, NP, bp);
  GEOBI_LAUNCH_OK();
  return 0;
}

int feast_fused_pack_all(const float* lin_w, const float* u_w, const float* c, int Cin, int Cout, int Kp, float* wf,
                         float* bf, float* bdx, hipStream_t s) {
  const int KDf = fused_k(Cin, 0), NPf = 32 * feast_fused_nt(Cout), KDx = fused_k(Cout, 1), NPx = 32 * feast_fused_nt(Cin);
  const int64_t total = (int64_t)Kp * Cout + (int64_t)KDf * NPf + (int64_t)KDx * NPx + 2 * QSELF;
  pack_fused_all_kernel<<<cdiv(total, 256), 256, 0, s>>>(lin_w, u_w, c, Cin, Cout, Kp, KDf, NPf, KDx, NPx, wf, bf, bdx);
  GEOBI_LAUNCH_OK();
  return 0;
}

// items[i].wf != NULL for every i (all forms: wf, then bf and bdx behind it as feast_wpack_floats lays them out) or for
// none (items[i].bf only)
int feast_fused_pack_batch(const FusedPackItem* items, int n, hipStream_t s) {
  GEOBI_REQUIRE(n > 0 && n <= kMaxPackBatch, "pack batch: %d layers (max %d)", n, kMaxPackBatch);
  PackBatch pb;
  pb.n = n;
  pb.all = items[0].wf != nullptr;
  pb.start[0] = 0;
  for (int i = 0; i < n; ++i) {
    pb.it[i] = items[i];
    const int Cin = items[i].Cin, Cout = items[i].Cout;
    GEOBI_REQUIRE((items[i].wf != nullptr) == (pb.all != 0), "pack batch: mixed modes");
    int64_t cnt = (int64_t)fused_k(Cin, 0) * 32 * feast_fused_nt(Cout) + QSELF;
    if (pb.all) cnt += (int64_t)feast_ldz(Cin) * Cout + (int64_t)fused_k(Cout, 1) * 32 * feast_fused_nt(Cin) + QSELF;
    pb.start[i + 1] = pb.start[i] + cnt;
  }
  pack_fused_batch_kernel<<<cdiv(pb.start[n], 256), 256, 0, s>>>(pb);
  GEOBI_LAUNCH_OK();
  return 0;
}

int feast_fused_pack_dx(const float* lin_w, const float* u_w, const float* c, int Cin, int Cout, float* bp,
                        hipStream_t s) {
  const int KD = fused_k(Cout, 1), NP = 32 * feast_fused_nt(Cin);
  pack_fused_dx_kernel<<<cdiv((int64_t)KD * NP + QSELF, 256), 256, 0, s>>>(lin_w, u_w, c, Cin, Cout, KD, NP, bp);
  GEOBI_LAUNCH_OK();
  return 0;
}

// algorithmic bytes of one fused launch (B_agg with the node transform fused: W = 4 N NOUT)
double feast_fused_bytes(int64_t N, int64_t E, int C, int nout) {
  return (double)E * (4.0 + 4.0 * C + 4.0 * H) + 4.0 * (double)N * H + 4.0 * (double)(N + 1) + 4.0 * (double)N * nout;
}

// forward: out = lrelu(aggregate(x) Wf + bias); LC > 0: per-edge logits from the unsplit 6 / 12-channel input
int feast_fused_fwd(const float* xa, const float* xb, int Ca, int Cin, const float* p, const float* cvec,
                    const int* rowptr, const int* col, int N, int LC, const float* ul, const float* Bp, int Cout,
                    const float* bias, float slope, float* out, hipStream_t s) {
  const int* deg_rowptr = nullptr;
  const float* xl = xa;
  const float *dpd = nullptr, *dl = nullptr, *dpn = nullptr;
  const int* pos = nullptr;
  const int NOUT = Cout, ldo = Cout, split = 0, ldo1 = 0;
  float* out1 = nullptr;
  float* tile_out = nullptr;
  const int NT = feast_fused_nt(Cout);
  switch (Cin * 100 + LC) {
    case 600: return launch_nt<6, 0, 0>(NT, GEOBI_FUSED_ARGS);
    case 606: return launch_nt<6, 0, 6>(NT, GEOBI_FUSED_ARGS);
    case 1200: return launch_nt<12, 0, 0>(NT, GEOBI_FUSED_ARGS);
    case 1212: return launch_nt<12, 0, 12>(NT, GEOBI_FUSED_ARGS);
    case 3200: return launch_nt<32, 0, 0>(NT, GEOBI_FUSED_ARGS);
    case 6400: return launch_nt<64, 0, 0>(NT, GEOBI_FUSED_ARGS);
    case 12800: return launch_nt<128, 0, 0>(NT, GEOBI_FUSED_ARGS);
    default: return set_error("feast fused forward: unsupported Cin=%d (per-edge logit channels %d)", Cin, LC);
  }
}

// backward: (dxa | dxb) = [r | dp | dcs] W', r aggregated over the transposed CSR from g [N, Cout]; dp is formed in
// the kernel from the row pass's dl [E, 12] (in-CSR order, reached through pos_in) and dpn [N, 12], dcs is the second
// half of dpd's rows (the first half is not read)
int feast_fused_dx(const float* g, int Cout, const float* p, const float* cvec, const int* rowptr_out,
                   const int* col_out, const int* rowptr_in, const int* pos, const float* dl, const float* dpn, int N,
                   int LC, const float* xl, const float* ul, const float* dpd, const float* Bp, int Cin, float* dxa, int Ca, float* dxb, int Cb, float* tile_out,
                   hipStream_t s) {
  const float *xa = g, *xb = g;
  const int* rowptr = rowptr_out;
  const int* col = col_out;
  const int* deg_rowptr = rowptr_in;
  const float* bias = nullptr;
  const float slope = 1.0f;
  const int NOUT = Cin;
  float* out = dxa;
  float* out1 = Cb ? dxb : nullptr;
  const int ldo = Cb ? Ca : Cin, split = Cb ? Ca : Cin, ldo1 = Cb;
  const int NT = feast_fused_nt(Cin);
  Ca = Cout;                      // the gathered rows are the unsplit g
  switch (Cout * 100 + LC) {
    case 3200: return launch_nt<32, 1, 0>(NT, GEOBI_FUSED_ARGS);
    case 3206: return launch_nt<32, 1, 6>(NT, GEOBI_FUSED_ARGS);
    case 3212: return launch_nt<32, 1, 12>(NT, GEOBI_FUSED_ARGS);
    case 6400: return launch_nt<64, 1, 0>(NT, GEOBI_FUSED_ARGS);
    case 6406: return launch_nt<64, 1, 6>(NT, GEOBI_FUSED_ARGS);
    case 6412: return launch_nt<64, 1, 12>(NT, GEOBI_FUSED_ARGS);
    case 12800: return launch_nt<128, 1, 0>(NT, GEOBI_FUSED_ARGS);
    case 12806: return launch_nt<128, 1, 6>(NT, GEOBI_FUSED_ARGS);
    case 12812: return launch_nt<128, 1, 12>(NT, GEOBI_FUSED_ARGS);
    default: return set_error("feast fused dx: unsupported Cout=%d (per-edge logit channels %d)", Cout, LC);
  }
}

}  // namespace geobi
