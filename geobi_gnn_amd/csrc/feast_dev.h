// Device helpers shared by the FeaSt kernels (feast.hip, feast_fused.hip).  gfx950 only.
#pragma once
#include "common.h"
#include <type_traits>

namespace geobi {
namespace feast_dev {

constexpr int H = GEOBI_H;
constexpr int HP = GEOBI_HP;

__device__ __forceinline__ void wave_lds_sync() {
  // LDS ops of one wave execute in order; this only stops the compiler moving them across.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// exp(x) for the softmax terms (x <= 0 once the maximum is subtracted; finite logits).  ocml's expf spends 13
// instructions per call, 9 calls per edge in three kernels: product, residual of the product and the low part of log2 e,
// integer / fraction split, v_exp_f32, ldexp, two range selects.  v_exp_f32 takes the whole argument at 1 ulp (the
// fraction of a float is exact), so the split and the selects are not needed: exp2(t) with t = fl(x log2 e) and the
// residual e = x log2 e - t applied to first order, 2^e = 1 + e ln 2 (|e| <= 2^-18: second order 2^-37).  6 instructions,
// the same <= 2 ulp; -3 % on the forward kernel (GEOBI_EXP_OCML=1 in a variant build restores expf).
__device__ __forceinline__ float exp_le0(float x) {
#ifdef GEOBI_EXP_OCML
  return expf(x);
#else
  const float t = x * 1.44269502e+00f;                       // 0x3fb8aa3b
  float e = fmaf(x, 1.44269502e+00f, -t);
  e = fmaf(x, 1.92596299e-08f, e);                            // 0x32a5705f: log2 e - fl(log2 e)
  const float r = __builtin_amdgcn_exp2f(t);                  // v_exp_f32; 0 below 2^-126
  return fmaf(r, e * 6.93147182e-01f, r);
#endif
}

__device__ __forceinline__ void softmax9(float (&l)[H]) {
  float m = l[0];
#pragma unroll
  for (int h = 1; h < H; ++h) m = fmaxf(m, l[h]);
  float s = 0.f;
#pragma unroll
  for (int h = 0; h < H; ++h) {
    l[h] = exp_le0(l[h] - m);
    s += l[h];
  }
  float inv = 1.0f / s;
#pragma unroll
  for (int h = 0; h < H; ++h) l[h] *= inv;
}

__device__ __forceinline__ void load_hp(const float* __restrict__ row, float (&v)[H]) {
  const float4* r4 = reinterpret_cast<const float4*>(row);
  float4 a = r4[0], b = r4[1], c = r4[2];
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  v[8] = c.x;
}

// XCD-aware block order.  Workgroups are dealt round-robin to the 8 XCDs (block b -> XCD b % 8), each with its
// own L2.  A node's neighbours are mostly nearby nodes, so a contiguous eighth of the node range per XCD
// keeps every gathered row in ONE L2 instead of eight: launched block b works on virtual block
// (b % 8) * (grid / 8) + b / 8.  The grid is padded to a multiple of 8; the surplus blocks land beyond N and exit.
__host__ __device__ __forceinline__ int xcd_grid(int blocks) { return (blocks + 7) / 8 * 8; }
__device__ __forceinline__ int xcd_block(int b, int grid) { return (b & 7) * (grid >> 3) + (b >> 3); }

// L2 warm-up of a read-only operand that EVERY workgroup of the launch streams (a layer's packed weights, 0.15-0.6 MB).
// All tiles walk the same weight addresses at about the same time, so in a launch of one round or less nobody finds a line
// in L2 that somebody else fetched earlier: each wave's 2-4 loads in flight then see the full fabric latency, and a tile's
// matrix phase runs at a third of the matrix rate (stamps: tools/fused_stamps_small.py).  Here the first <= 32 workgroups
// of every XCD request the whole operand up front -- one dword per 128-B line and lane, 64 lines per wave instruction, two
// instructions per wave -- while their gather phase runs.  The loads go to LDS (global_load_lds_dword: no destination
// register, nothing waits for the value; an inline-assembly load into a register was tried first and is unsafe -- the
// compiler spills or copies a register it believes defined while the load is still in flight); their landing zone is 256 B
// per wave that nothing reads.
typedef __attribute__((address_space(3))) void* warm_lds_ptr;
template <int NWAVES>
__device__ __forceinline__ void warm_l2(const float* __restrict__ base, int bytes, bool on, float* scratch) {
  const int nw = min((int)(gridDim.x >> 3), 32);
  const int bi = blockIdx.x >> 3;
  if (on && bi < nw) {
    const int lines = bytes >> 7;
    const int stride = nw * NWAVES * 64;
    const int line = bi * (NWAVES * 64) + threadIdx.x;
    float* dst = scratch + (threadIdx.x >> 6) * 64;
#pragma unroll
    for (int r = 0; r < 2; ++r)
      __builtin_amdgcn_global_load_lds(base + (size_t)min(line + r * stride, lines - 1) * 32, (warm_lds_ptr)dst, 4, 0, 0);
  }
}

template <int VEC>
__device__ __forceinline__ void load_vec(const float* __restrict__ ptr, float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    float4 t = *reinterpret_cast<const float4*>(ptr);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) v[i] = ptr[i];
  }
}

template <int VEC>
__device__ __forceinline__ void store_vec(float* __restrict__ ptr, const float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    *reinterpret_cast<float4*>(ptr) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) ptr[i] = v[i];
  }
}

// Per-edge logits of the layers that read raw mesh coordinates (Cin = 6 / 12, level 0): there the features are
// positions scaled by 1 / mean edge length, |x| ~ 27 at n = 32 and ~ 72 at n = 87, and the node-level form
// p_j - p_i (p = x u^T) loses |x| / |x_j - x_i| in the subtraction.  These layers therefore evaluate
// u (x_j - x_i) per edge exactly as the reference does: the difference of two nearby coordinates is (nearly)
// exact in fp32, and a level-0 row is only 24 / 48 B -- no more than the 48-B logit row it replaces.
// LDS image of u: [LC][HP] (head-minor, 3 x float4 per input channel).
template <int LC>
__device__ __forceinline__ void stage_u(const float* __restrict__ u, float* s_u) {
  for (int i = threadIdx.x; i < LC * HP; i += blockDim.x) {
    const int k = i / HP, h = i % HP;
    s_u[i] = h < H ? u[h * LC + k] : 0.f;
  }
  __syncthreads();
}

template <int LC>
__device__ __forceinline__ void load_row(const float* __restrict__ row, float (&v)[LC]) {
  if constexpr ((LC & 3) == 0) {
#pragma unroll
    for (int i = 0; i < LC; i += 4) {
      float4 t = *reinterpret_cast<const float4*>(row + i);
      v[i] = t.x; v[i + 1] = t.y; v[i + 2] = t.z; v[i + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < LC; i += 2) {
      float2 t = *reinterpret_cast<const float2*>(row + i);
      v[i] = t.x; v[i + 1] = t.y;
    }
  }
}

// l_h = c_h + sum_k u[h,k] d[k]
template <int LC>
__device__ __forceinline__ void edge_logits(const float (&d)[LC], const float* s_u, const float (&cc)[H], float (&l)[H]) {
#pragma unroll
  for (int h = 0; h < H; ++h) l[h] = cc[h];
#pragma unroll
  for (int k = 0; k < LC; ++k) {
    const float4* r = reinterpret_cast<const float4*>(s_u + k * HP);
    const float4 a = r[0], b = r[1], c = r[2];
    l[0] = fmaf(a.x, d[k], l[0]); l[1] = fmaf(a.y, d[k], l[1]); l[2] = fmaf(a.z, d[k], l[2]);
    l[3] = fmaf(a.w, d[k], l[3]); l[4] = fmaf(b.x, d[k], l[4]); l[5] = fmaf(b.y, d[k], l[5]);
    l[6] = fmaf(b.z, d[k], l[6]); l[7] = fmaf(b.w, d[k], l[7]); l[8] = fmaf(c.x, d[k], l[8]);
  }
}


// All-reduce over the G consecutive lanes of a group with DPP lane permutes (one v_add with a DPP
// source modifier per step, no LDS round trip as ds_bpermute/__shfl would take):
//   xor 1 / xor 2 inside a quad (quad_perm), quad <-> quad inside 8 lanes (row_half_mirror),
//   8 <-> 8 inside a 16-lane DPP row (row_mirror); only G = 32 needs one cross-row step.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true);
  return v + __int_as_float(t);
}

template <int G>
__device__ __forceinline__ float group_allreduce(float v) {
  v = dpp_add<0xB1>(v);                              // quad_perm [1,0,3,2]
  if constexpr (G >= 4) v = dpp_add<0x4E>(v);        // quad_perm [2,3,0,1]
  if constexpr (G >= 8) v = dpp_add<0x141>(v);       // row_half_mirror
  if constexpr (G >= 16) v = dpp_add<0x140>(v);      // row_mirror
  if constexpr (G >= 32) v += __shfl_xor(v, 16, 64); // across the two 16-lane rows of the group
  return v;
}


// One lane's share of a row: PV floats per piece

template <int PV>
__device__ __forceinline__ void load_piece(const float* __restrict__ ptr, float* v) {
  if constexpr (PV == 4) {
    const float4 t = *reinterpret_cast<const float4*>(ptr);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else if constexpr (PV == 2) {
    const float2 t = *reinterpret_cast<const float2*>(ptr);
    v[0] = t.x; v[1] = t.y;
  } else {
    v[0] = ptr[0];
  }
}
template <int PV>
__device__ __forceinline__ void store_piece(float* ptr, const float* v) {
  if constexpr (PV == 4) *reinterpret_cast<float4*>(ptr) = make_float4(v[0], v[1], v[2], v[3]);
  else if constexpr (PV == 2) *reinterpret_cast<float2*>(ptr) = make_float2(v[0], v[1]);
  else ptr[0] = v[0];
}

// acc += sum_v  (lane OWNER's dz[v]) * x[v]  for the PV floats of one piece: the row broadcast rides on the FMA's first
// operand (v_fmac_f32_dpp row_newbcast, gfx90a+), no separate move.  Every lane of the 16-lane row must be active: a
// source lane EXEC has switched off does not deliver.  The leading s_nop covers the two wait states a DPP read needs
// after a VALU write of the same register -- the compiler's hazard pass does not look inside the statement.
template <int OWNER, int PV>
__device__ __forceinline__ void fmac_bcast(float& acc, const float* dz, const float* x) {
  static_assert(OWNER >= 0 && OWNER < 16 && (PV == 4 || PV == 2), "a DPP row has 16 lanes");
#define GEOBI_FB(M)                                                                                                   \
  if constexpr (OWNER == M) {                                                                                         \
    if constexpr (PV == 4)                                                                                            \
      asm("s_nop 1\n\t"                                                                                               \
          "v_fmac_f32_dpp %0, %1, %5 row_newbcast:" #M " row_mask:0xf bank_mask:0xf\n\t"                              \
          "v_fmac_f32_dpp %0, %2, %6 row_newbcast:" #M " row_mask:0xf bank_mask:0xf\n\t"                              \
          "v_fmac_f32_dpp %0, %3, %7 row_newbcast:" #M " row_mask:0xf bank_mask:0xf\n\t"                              \
          "v_fmac_f32_dpp %0, %4, %8 row_newbcast:" #M " row_mask:0xf bank_mask:0xf"                                  \
          : "+v"(acc)                                                                                                 \
          : "v"(dz[0]), "v"(dz[1]), "v"(dz[2]), "v"(dz[3]), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]));              \
    else                                                                                                              \
      asm("s_nop 1\n\t"                                                                                               \
          "v_fmac_f32_dpp %0, %1, %3 row_newbcast:" #M " row_mask:0xf bank_mask:0xf\n\t"                              \
          "v_fmac_f32_dpp %0, %2, %4 row_newbcast:" #M " row_mask:0xf bank_mask:0xf"                                  \
          : "+v"(acc)                                                                                                 \
          : "v"(dz[0]), "v"(dz[1]), "v"(x[0]), "v"(x[1]));                                                            \
  }
  GEOBI_FB(0) GEOBI_FB(1) GEOBI_FB(2) GEOBI_FB(3) GEOBI_FB(4) GEOBI_FB(5) GEOBI_FB(6) GEOBI_FB(7)
  GEOBI_FB(8) GEOBI_FB(9) GEOBI_FB(10) GEOBI_FB(11) GEOBI_FB(12) GEOBI_FB(13) GEOBI_FB(14) GEOBI_FB(15)
#undef GEOBI_FB
}

// Three heads at a time: acc_t += sum_v (lane O_t's dz_t[v]) * x[v] for one 16-B piece, the three accumulators taking turns
// (no FMA waits on the one before it) behind ONE pair of wait states -- a third of the s_nops of the per-head form above
// (144 -> 48 per 16 items of a 64-channel row), same FMAs in the same order per accumulator.
template <int O0, int O1, int O2>
__device__ __forceinline__ void fmac_bcast3(float& a0, float& a1, float& a2, const float* d0, const float* d1, const float* d2,
                                            const float* x) {
  static_assert(O0 >= 0 && O0 < 16 && O1 >= 0 && O1 < 16 && O2 >= 0 && O2 < 16, "a DPP row has 16 lanes");
  asm("s_nop 1\n\t"
      "v_fmac_f32_dpp %0, %3, %15 row_newbcast:%19 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %1, %7, %15 row_newbcast:%20 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %2, %11, %15 row_newbcast:%21 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %0, %4, %16 row_newbcast:%19 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %1, %8, %16 row_newbcast:%20 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %2, %12, %16 row_newbcast:%21 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %0, %5, %17 row_newbcast:%19 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %1, %9, %17 row_newbcast:%20 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %2, %13, %17 row_newbcast:%21 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %0, %6, %18 row_newbcast:%19 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %1, %10, %18 row_newbcast:%20 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %2, %14, %18 row_newbcast:%21 row_mask:0xf bank_mask:0xf"
      : "+v"(a0), "+v"(a1), "+v"(a2)
      : "v"(d0[0]), "v"(d0[1]), "v"(d0[2]), "v"(d0[3]), "v"(d1[0]), "v"(d1[1]), "v"(d1[2]), "v"(d1[3]), "v"(d2[0]),
        "v"(d2[1]), "v"(d2[2]), "v"(d2[3]), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "n"(O0), "n"(O1), "n"(O2));
}

// f(integral_constant<int, I>) for I in [0, N): loop indices that must be constants (DPP controls, register slots)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}


// ------------------------------------------------------------------------------------------------------------------
// Backward row pass of one target node by its 16-lane group, lane = edge: each lane takes one in-edge (plus one
// pseudo-edge for the self loop) and forms all nine dot products s_h = dz_i[h,:] . x_j on its own -- the neighbour
// row is the lane's private read, the node's dz row is spread over the group (piece p -> slot p / 16, lane p % 16) and each piece reaches the FMAs by a row broadcast riding on the operand.  The softmax backward
//   dl_h = q_h (s_h - sum q s) / (deg_i + 1)
// then needs no cross-lane step, dl rows leave as 48 contiguous bytes per lane, and only the per-node sums
// dpn (what flows to -p_i) and dcs (dpn + the self edge's share) are reduced over the group, once per node.
// (The earlier form -- lanes = channel slices, nine 16-lane reductions per edge -- spent most of its instructions in
// those reductions; re-reading the dz row from LDS per edge was LDS-bandwidth bound.)
//   zrow: the node's dz row [9][C] (LDS or global, 16-B aligned).  At 128 channels the dz pieces are taken in two
//   halves of 64 channels (the dot products add up over channels), so the neighbour row is still read once.
// Every lane of the group runs the dot products: a row broadcast does not deliver from a lane EXEC has switched off.
template <int C, int LC>
__device__ __forceinline__ void rowpass_edge_node(
    const float* zrow, const float* __restrict__ xa, const float* __restrict__ xb, int Ca,
    const float* __restrict__ p, const float* __restrict__ cvec, const float* s_u, const int* __restrict__ rowptr,
    const int* __restrict__ col, int N, int node, int k, float* __restrict__ dl, float* __restrict__ dpn,
    float* __restrict__ dcs, int ld_dcs) {
  constexpr int G = 16;
  constexpr int VW = (C % 4 == 0) ? 4 : 2;                   // floats per piece
  constexpr int CCH = C > 64 ? 64 : C;                       // channels whose dz pieces are in registers at once
  constexpr int NCC = C / CCH;                               // channel chunks (2 at 128 channels)
  constexpr int NQ = CCH / VW;                               // pieces per head and chunk
  constexpr int XB = NQ % 4 == 0 ? 4 : NQ;                   // row pieces per batch (16 channels, or the whole short row)
  constexpr int NPIECE = H * NQ, NSLOT = (NPIECE + G - 1) / G;
  static_assert(NQ % XB == 0 && XB <= 4 && C % CCH == 0, "whole batches, whole channel chunks");
  const bool valid = node < N;
  const int ns = valid ? node : N - 1;
  const int rs = rowptr[ns];
  const int deg = valid ? rowptr[ns + 1] - rs : -1;          // items = deg real edges + the self loop; none if invalid
  const int Cb = C - Ca;

  float dzr[NSLOT][VW];
  auto load_dz = [&](int c) {                                // piece pidx of chunk c = head pidx / NQ, piece pidx % NQ
#pragma unroll
    for (int sl = 0; sl < NSLOT; ++sl) {
      const int pidx = sl * G + k;
      if (pidx < NPIECE) load_piece<VW>(zrow + (pidx / NQ) * C + c * CCH + (pidx % NQ) * VW, dzr[sl]);
      else {
#pragma unroll
        for (int v = 0; v < VW; ++v) dzr[sl][v] = 0.f;
      }
    }
  };
  if constexpr (NCC == 1) load_dz(0);

  float cc[H];
#pragma unroll
  for (int h = 0; h < H; ++h) cc[h] = cvec[h];
  float xc[LC > 0 ? LC : 1];
  if constexpr (LC > 0) load_row<LC>(xa + (size_t)ns * LC, xc);
  const float invd = 1.0f / (float)(deg + 1);
  // dsum: the lane's real edges.  The self loop is item `deg`, i.e. always in the node's LAST chunk: its share is
  // picked from that iteration's d after the loop (d and self are dead across the dot products of later chunks,
  // so they cost no registers there).
  float dsum[H], d[H];
#pragma unroll
  for (int h = 0; h < H; ++h) { dsum[h] = 0.f; d[h] = 0.f; }
  bool self = false;

  for (int base = 0; base <= deg; base += G) {
    const int idx = base + k;
    const bool real = idx < deg;
    self = idx == deg;
    const int e = rs + idx;
    const int j = real ? col[e] : ns;                        // lanes past the node's items work on its own row
    float q[H];
    if (real) {
      if constexpr (LC > 0) {
        float d[LC];
        load_row<LC>(xa + (size_t)j * LC, d);
#pragma unroll
        for (int i = 0; i < LC; ++i) d[i] -= xc[i];
        edge_logits<LC>(d, s_u, cc, q);
      } else {
        float pc[H], pn[H];
        load_hp(p + (size_t)ns * HP, pc);
        load_hp(p + (size_t)j * HP, pn);
#pragma unroll
        for (int h = 0; h < H; ++h) q[h] = pn[h] - pc[h] + cc[h];
      }
      softmax9(q);
    } else {                                                 // the self loop: u (x_i - x_i) + c = c exactly
#pragma unroll
      for (int h = 0; h < H; ++h) q[h] = cc[h];
      softmax9(q);
    }
    // nine dot products over the row, channels in ascending order; the row arrives in batches of XB pieces, every
    // piece of a batch requested before any is used
    float sv[H];
#pragma unroll
    for (int h = 0; h < H; ++h) sv[h] = 0.f;
    const float* ra = xa + (size_t)j * Ca;
    const float* rb = xb + (size_t)j * Cb - Ca;              // indexed by the channel of the whole row
    static_for<0, NCC>([&](auto ci) {
      constexpr int c = decltype(ci)::value;
      if constexpr (NCC > 1) load_dz(c);
      static_for<0, NQ / XB>([&](auto bi) {
        constexpr int q0 = decltype(bi)::value * XB;
        constexpr int ch0 = c * CCH + q0 * VW;               // first channel of the batch
        const float* src = ch0 < Ca ? ra : rb;               // a batch never straddles the two inputs
        float xj[XB][VW];
#pragma unroll
        for (int qd = 0; qd < XB; ++qd) load_piece<VW>(src + ch0 + qd * VW, xj[qd]);
        static_for<0, XB>([&](auto qi) {
          constexpr int qd = decltype(qi)::value;
          if constexpr (VW == 4) {
            static_for<0, H / 3>([&](auto gi) {
              constexpr int h = decltype(gi)::value * 3;
              constexpr int p0 = h * NQ + q0 + qd, p1 = p0 + NQ, p2 = p1 + NQ;
              fmac_bcast3<p0 % G, p1 % G, p2 % G>(sv[h], sv[h + 1], sv[h + 2], dzr[p0 / G], dzr[p1 / G], dzr[p2 / G], xj[qd]);
            });
          } else {
            static_for<0, H>([&](auto hi) {
              constexpr int h = decltype(hi)::value;
              constexpr int pidx = h * NQ + q0 + qd, sl = pidx / G, owner = pidx % G;
              fmac_bcast<owner, VW>(sv[h], dzr[sl], xj[qd]);
            });
          }
        });
      });
    });
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
  // per-node sums over the group (fixed butterfly: deterministic)
  float dself[H];
#pragma unroll
  for (int h = 0; h < H; ++h) {
    dsum[h] = group_allreduce<G>(dsum[h]);
    dself[h] = group_allreduce<G>(self ? d[h] : 0.f);
  }
  if (!valid || k != 0) return;
  float4* a = reinterpret_cast<float4*>(dpn + (size_t)node * HP);
  a[0] = make_float4(dsum[0], dsum[1], dsum[2], dsum[3]);
  a[1] = make_float4(dsum[4], dsum[5], dsum[6], dsum[7]);
  a[2] = make_float4(dsum[8], 0.f, 0.f, 0.f);
  float4* bq = reinterpret_cast<float4*>(dcs + (size_t)node * ld_dcs);
  bq[0] = make_float4(dsum[0] + dself[0], dsum[1] + dself[1], dsum[2] + dself[2], dsum[3] + dself[3]);
  bq[1] = make_float4(dsum[4] + dself[4], dsum[5] + dself[5], dsum[6] + dself[6], dsum[7] + dself[7]);
  bq[2] = make_float4(dsum[8] + dself[8], 0.f, 0.f, 0.f);
}

// ------------------------------------------------------------------------------------------------------------------
// The same row pass with the neighbour rows STAGED through LDS (64-channel inputs, node-level logits).  In the form
// above a lane reads its neighbour's row as sixteen private 16-B pieces, four in flight: every load instruction of a
// wave touches 64 different cache lines (one per lane, an eighth of each used) and a chunk of items costs four
// dependent L2 round trips.  Here the row arrives the way the forward kernel reads it -- half a row (32 channels =
// one 128-B line) by eight adjacent lanes -- but lands in LDS, not in registers: `global_load_lds_dwordx4`, eight
// instructions per half for the wave's 4 nodes x 16 items, all in flight at once at no register cost.  Landing zone =
// the wave's own four dz rows (dead once their pieces sit in registers): instruction i covers items 2 i and 2 i + 1
// and writes its 1 KiB lane-linearly (node g -> +256 B, odd item -> +128 B) at i x 1040 B -- the 16-B skew makes the
// lane = item reads (ds_read_b128, lane k at (k >> 1) x 1040 + (k & 1) x 128) conflict-free.  Lane k of a group then
// runs the same FMAs in the same order as above (channels ascending): the results are bit-identical.
// Per chunk of 16 items: two staged halves = two round trips instead of four, an eighth of the cache-line lookups.
#ifndef GEOBI_RP_STAMP
#define GEOBI_RP_STAMP(i) do { } while (0)
#endif
template <int L>
__device__ __forceinline__ int row_bcast(int v) {             // lane L of every 16-lane row -> the whole row
  return __builtin_amdgcn_update_dpp(0, v, 0x150 + L, 0xf, 0xf, false);   // row_newbcast:L
}

template <int C, int LDZ>
__device__ __forceinline__ void rowpass_edge_node_staged(
    float* wave_rows, const float* __restrict__ xa, const float* __restrict__ xb, int Ca, const float* __restrict__ p,
    const float* __restrict__ cvec, const int* __restrict__ rowptr, const int* __restrict__ col, int N, int node, int g,
    int k, float* __restrict__ dl, float* __restrict__ dpn, float* __restrict__ dcs, int ld_dcs) {
  // (Row pointers and neighbour ids requested by the kernel ahead of its matrix phase -- two dependent loads off this
  // function's chain -- measured 101 against 99 us: the workgroups of a CU cover each other's waits already.)
  constexpr int G = 16, HALF = 32, HQ = HALF / 4;             // 16 pieces per head (= G), 8 per half row
  constexpr int SLOT = 260;                                   // floats between two staging instructions (1 KiB + 16 B)
  static_assert(C == 64, "a row is two 128-B halves");
  static_assert(8 * SLOT <= 4 * LDZ, "the landing zone fits the wave's four dz rows");
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const bool valid = node < N;
  const int ns = valid ? node : N - 1;
  const int rs = rowptr[ns];
  const int deg = valid ? rowptr[ns + 1] - rs : -1;
  const int Cb = C - Ca;

  float dzr[H][4];                                            // piece (head h, 16-B piece k) of the node's dz row
  const float* zrow = wave_rows + g * LDZ;
#pragma unroll
  for (int h = 0; h < H; ++h) load_piece<4>(zrow + h * C + k * 4, dzr[h]);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // every lane's pieces have arrived: the rows may be overwritten

  float cc[H];
#pragma unroll
  for (int h = 0; h < H; ++h) cc[h] = cvec[h];
  const float invd = 1.0f / (float)(deg + 1);
  float dsum[H], d[H];
#pragma unroll
  for (int h = 0; h < H; ++h) { dsum[h] = 0.f; d[h] = 0.f; }
  bool self = false;
  const float* my = wave_rows + (k >> 1) * SLOT + g * 64 + (k & 1) * HALF;

  for (int base = 0; base <= deg; base += G) {
    const int idx = base + k;
    const bool real = idx < deg;
    self = idx == deg;
    const int e = rs + idx;
    const int j = real ? col[e] : ns;                          // lanes past the node's items work on its own row
    auto stage = [&](auto halfc) {
      constexpr int half = decltype(halfc)::value;
      const float* rbase = half * HALF < Ca ? xa + half * HALF : xb + (half * HALF - Ca);
      const int rstride = half * HALF < Ca ? Ca : Cb;
      static_for<0, 8>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        const int jlo = row_bcast<2 * i>(j), jhi = row_bcast<2 * i + 1>(j);
        const int jj = (k & 8) ? jhi : jlo;
        __builtin_amdgcn_global_load_lds(rbase + (size_t)jj * rstride + 4 * (k & 7), (lds_ptr)(wave_rows + i * SLOT), 16, 0,
                                         0);
      });
    };
    stage(std::integral_constant<int, 0>{});
    float q[H];
    if (real) {
      float pc[H], pn[H];
      load_hp(p + (size_t)ns * HP, pc);
      load_hp(p + (size_t)j * HP, pn);
#pragma unroll
      for (int h = 0; h < H; ++h) q[h] = pn[h] - pc[h] + cc[h];
    } else {                                                   // the self loop: u (x_i - x_i) + c = c exactly
#pragma unroll
      for (int h = 0; h < H; ++h) q[h] = cc[h];
    }
    softmax9(q);
    float sv[H];
#pragma unroll
    for (int h = 0; h < H; ++h) sv[h] = 0.f;
    if (base == 0) GEOBI_RP_STAMP(5);
    static_for<0, 2>([&](auto halfc) {
      constexpr int half = decltype(halfc)::value;
      if constexpr (half == 1) stage(halfc);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the half rows have landed
      if (base == 0) GEOBI_RP_STAMP(6 + half);
      static_for<0, HQ / 4>([&](auto bi) {
        constexpr int q0 = decltype(bi)::value * 4;
        float xj[4][4];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) load_piece<4>(my + (q0 + qd) * 4, xj[qd]);
        static_for<0, 4>([&](auto qi) {
          constexpr int qd = decltype(qi)::value;
          static_for<0, H / 3>([&](auto gi) {
            constexpr int h = decltype(gi)::value * 3, o = half * HQ + q0 + qd;
            fmac_bcast3<o, o, o>(sv[h], sv[h + 1], sv[h + 2], dzr[h], dzr[h + 1], dzr[h + 2], xj[qd]);
          });
        });
      });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // ... and have been read: the zone may be refilled
    });
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
  if (!valid || k != 0) return;
  float4* a = reinterpret_cast<float4*>(dpn + (size_t)node * HP);
  a[0] = make_float4(dsum[0], dsum[1], dsum[2], dsum[3]);
  a[1] = make_float4(dsum[4], dsum[5], dsum[6], dsum[7]);
  a[2] = make_float4(dsum[8], 0.f, 0.f, 0.f);
  float4* bq = reinterpret_cast<float4*>(dcs + (size_t)node * ld_dcs);
  bq[0] = make_float4(dsum[0] + dself[0], dsum[1] + dself[1], dsum[2] + dself[2], dsum[3] + dself[3]);
  bq[1] = make_float4(dsum[4] + dself[4], dsum[5] + dself[5], dsum[6] + dself[6], dsum[7] + dself[7]);
  bq[2] = make_float4(dsum[8] + dself[8], 0.f, 0.f, 0.f);
}

}  // namespace feast_dev
}  // namespace geobi
