// FeaSt layer, fused form: aggregation + node-level transform in ONE kernel -- the aggregated rows
// z_i = [1/deg sum_j q_ijh x_j]_h (9*C floats per node) never reach HBM.
//
// A workgroup (8 waves) owns a tile of 32 consecutive nodes:
//   phase A  the gather of feast.hip's aggregation kernel (G = C/VEC lanes per node, per-edge softmax
//            evaluated once and parked, neighbour rows gathered as one contiguous segment per row), but
//            the 9 x VEC register accumulators of a node are stored into an LDS tile z[32][LD] instead of
//            global memory.  The per-edge parking slots live in the first 3C floats of the node's own
//            (not yet written) tile row, so the tile is the kernel's only LDS.
//   phase B  out[32, NOUT] = z[32, K] * Bp[K, NOUT] on the matrix cores (v_mfma_f32_32x32x2_f32, exact
//            fp32).  The 8 waves split the NOUT/32 column tiles and the K range; the K-split partial tiles
//            are folded through LDS in a fixed order (deterministic), bias + leaky-relu applied, rows
//            written with 8-B stores (one 128-B segment per 16 lanes).
// Operand delivery: the weights are packed so that ONE 16-B load per lane feeds four consecutive MFMAs,
//   Bp[kb][half][col][s] = B[k = 8 kb + 4 half + s][col],
// and the A operand is the matching 16 B of the lane's tile row, z[lane & 31][8 kb + 4 half .. + 3]
// (ds_read_b128; LD = 4 mod 64 floats makes it conflict-free).  The MFMA's two k-slots of step s are thus
// k = 8 kb + s (lanes 0-31) and k = 8 kb + 4 + s (lanes 32-63): any bijection works as long as A and B
// agree; the order of the fp32 sum is fixed by it and does not depend on N or on the batch.
//
// MODE 0  forward:   rows = layer input (xa | xb), K = 9 Cin padded to 8, Bp from lin.weight, epilogue
//                    bias + leaky-relu.
// MODE 1  backward:  dx = [r | dp | dcs] [lin.weight ; u.weight ; 0]: rows = g (gradient w.r.t. the
//                    pre-activation output), transposed CSR, weights q_ij / deg_i; the tile gets 24 extra
//                    columns [dp | dcs] read from `dpd`; output split into (dxa | dxb) for split inputs.
#include "common.h"
#include "feast_dev.h"

namespace geobi {

namespace {

using namespace feast_dev;

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TN = 32;        // nodes per tile = rows of one MFMA tile
constexpr int NW = 8;         // waves per workgroup
constexpr int RED_LD = 36;    // row stride of a partial output tile in LDS

__host__ __device__ constexpr int tile_ld(int K) { return (K - 4 + 63) / 64 * 64 + 4; }
__host__ __device__ constexpr int fused_k(int C, int MODE) { return MODE == 0 ? (H * C + 7) / 8 * 8 : H * C + 2 * HP; }
__host__ __device__ constexpr int fused_lds_floats(int C, int MODE, int LC) {
  const int tile = TN * tile_ld(fused_k(C, MODE)), red = NW * 32 * RED_LD;
  return (tile > red ? tile : red) + (LC > 0 ? LC * HP : 0);
}

template <int C, int VEC, int MODE, int LC, int NT>
__global__ __launch_bounds__(512) void feast_fused_kernel(
    const float* __restrict__ xa, const float* __restrict__ xb, int Ca, const float* __restrict__ p,
    const float* __restrict__ cvec, const int* __restrict__ rowptr, const int* __restrict__ col,
    const int* __restrict__ deg_rowptr, int N, const float* __restrict__ xl, const float* __restrict__ ul,
    const float* __restrict__ dpd, const float* __restrict__ Bp, int NOUT, const float* __restrict__ bias,
    float slope, float* __restrict__ out, int ldo, float* __restrict__ out1, int split, int ldo1) {
  constexpr int G = C / VEC;
  constexpr int NPW = 64 / G;
  constexpr int KD = fused_k(C, MODE);
  constexpr int LD = tile_ld(KD);
  static_assert(G * VEC == C && G >= 2 && (64 % G) == 0, "group shape");
  static_assert(G * HP <= H * C, "the parking slots of a node fit its own tile row");
  static_assert(MODE == 0 || G >= 6, "the [dp | dcs] columns are copied by 6 lanes of the group");
  static_assert(NW % NT == 0, "column tiles divide the waves");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int TILE_FLOATS = TN * LD > NW * 32 * RED_LD ? TN * LD : NW * 32 * RED_LD;
  float* s_u = smem + TILE_FLOATS;
  if constexpr (LC > 0) stage_u<LC>(ul, s_u);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = xcd_block(blockIdx.x, gridDim.x);
  if (tile * TN >= N) return;          // whole workgroup: surplus tile of the XCD-padded grid

  // ------------------------------------------------------------------ phase A: aggregate into the LDS tile
  {
    const int g = lane / G, k = lane % G;
    const int c0 = k * VEC;
    const float* fbase;
    int fstride;
    if (c0 < Ca) { fbase = xa + c0; fstride = Ca; } else { fbase = xb + (c0 - Ca); fstride = C - Ca; }
    float cc[H], qs[H];
#pragma unroll
    for (int h = 0; h < H; ++h) { cc[h] = cvec[h]; qs[h] = cc[h]; }
    softmax9(qs);   // the self edge: u(x_i - x_i) + c = c exactly

    for (int nb = wave * NPW; nb < TN; nb += NW * NPW) {
      const int nl = nb + g;
      const int node = tile * TN + nl;
      const bool valid = node < N;
      const int ns = valid ? node : N - 1;
      const int rs = rowptr[ns];
      const int re = valid ? rowptr[ns + 1] : rs;
      float* zrow = smem + nl * LD;
      float(*slot)[HP] = reinterpret_cast<float(*)[HP]>(zrow);

      float pc[H];
      float xc[LC > 0 ? LC : 1];
      if constexpr (LC > 0) load_row<LC>(xl + (size_t)ns * LC, xc);
      else load_hp(p + (size_t)ns * HP, pc);

      float acc[H][VEC];
      {
        float xs[VEC];
        load_vec<VEC>(fbase + (size_t)ns * fstride, xs);
        float sscale = 1.0f;
        if constexpr (MODE == 1) sscale = 1.0f / (float)(deg_rowptr[ns + 1] - deg_rowptr[ns] + 1);
#pragma unroll
        for (int h = 0; h < H; ++h)
#pragma unroll
          for (int v = 0; v < VEC; ++v) acc[h][v] = qs[h] * sscale * xs[v];
      }

      for (int base = rs; base < re; base += G) {
        // ---- lane k of the group handles edge base + k: logits, softmax, park q and the neighbour id
        {
          const int e = base + k;
          float q[H];
          int j = ns;
          if (e < re) {
            j = col[e];
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
        }
        wave_lds_sync();
        // ---- all G lanes of the group walk the parked edges, two at a time
        const int cnt = min(G, re - base);
        for (int t = 0; t < cnt; t += 2) {
          const float4* s0 = reinterpret_cast<const float4*>(slot[t]);
          const float4* s1 = reinterpret_cast<const float4*>(slot[t + 1]);
          float4 a0 = s0[0], b0 = s0[1], d0 = s0[2];
          float4 a1 = s1[0], b1 = s1[1], d1 = s1[2];
          float x0[VEC], x1[VEC];
          load_vec<VEC>(fbase + (size_t)__float_as_int(d0.y) * fstride, x0);
          load_vec<VEC>(fbase + (size_t)__float_as_int(d1.y) * fstride, x1);
          const float q0[H] = {a0.x, a0.y, a0.z, a0.w, b0.x, b0.y, b0.z, b0.w, d0.x};
          const float q1[H] = {a1.x, a1.y, a1.z, a1.w, b1.x, b1.y, b1.z, b1.w, d1.x};
#pragma unroll
          for (int h = 0; h < H; ++h)
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[h][v] = fmaf(q0[h], x0[v], acc[h][v]);
#pragma unroll
          for (int h = 0; h < H; ++h)
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[h][v] = fmaf(q1[h], x1[v], acc[h][v]);
        }
        wave_lds_sync();
      }

      // ---- the node's aggregated row replaces its parking slots
      float scale = valid ? 1.0f : 0.0f;
      if constexpr (MODE == 0) scale = valid ? 1.0f / (float)(re - rs + 1) : 0.0f;
#pragma unroll
      for (int h = 0; h < H; ++h) {
        float v[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[i] = acc[h][i] * scale;
        store_vec<VEC>(zrow + h * C + c0, v);
      }
      if constexpr (MODE == 0) {
        for (int i = H * C + k; i < KD; i += G) zrow[i] = 0.f;          // K padding
      } else {
        if (k < 6) {                                                    // [dp | dcs]: 24 floats
          float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
          if (valid) t = reinterpret_cast<const float4*>(dpd + (size_t)node * (2 * HP))[k];
          reinterpret_cast<float4*>(zrow + H * C)[k] = t;
        }
      }
    }
  }
  __syncthreads();

  // ------------------------------------------------------------------ phase B: tile x packed weights on MFMA
  constexpr int NKB = KD / 8;
  constexpr int KS = NW / NT;
  constexpr int KB_PER = (NKB + KS - 1) / KS;
  const int ct = wave % NT, ks = wave / NT;
  const int kb0 = ks * KB_PER;
  const int kb1 = min(NKB, kb0 + KB_PER);
  const int hf = lane >> 5, l31 = lane & 31;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (kb0 < kb1) {
    // two k-blocks (8 MFMAs, 512 pipe cycles) per stage; the next stage's LDS and L2 loads are issued before
    // this stage's MFMAs (clamped block index: always a valid address, surplus blocks are skipped below)
    constexpr int U = 2;
    const float* arow = smem + l31 * LD + 4 * hf;
    const float4* bcol = reinterpret_cast<const float4*>(Bp) + (size_t)hf * (32 * NT) + ct * 32 + l31;
    float4 a_cur[U], w_cur[U], a_nxt[U], w_nxt[U];
    auto load = [&](float4 (&av)[U], float4 (&wv)[U], int b) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int bb = min(b + u, kb1 - 1);
        av[u] = *reinterpret_cast<const float4*>(arow + 8 * bb);
        wv[u] = bcol[(size_t)bb * (2 * 32 * NT)];
      }
    };
    load(a_cur, w_cur, kb0);
    for (int b = kb0; b < kb1; b += U) {
      load(a_nxt, w_nxt, b + U);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (b + u < kb1) {
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[u].x, w_cur[u].x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[u].y, w_cur[u].y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[u].z, w_cur[u].z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[u].w, w_cur[u].w, acc, 0, 0, 0);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) { a_cur[u] = a_nxt[u]; w_cur[u] = w_nxt[u]; }
    }
  }
  __syncthreads();                                 // every wave is done reading the z tile
  float(*red)[32][RED_LD] = reinterpret_cast<float(*)[32][RED_LD]>(smem);
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * hf][l31] = acc[r];
  __syncthreads();

  // ------------------------------------------------------------------ epilogue: fold the K splits, store
  const int row = threadIdx.x >> 4, c2 = (threadIdx.x & 15) * 2;
  const int node = tile * TN + row;
  if (node >= N) return;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    float2 s = make_float2(0.f, 0.f);
#pragma unroll
    for (int q = 0; q < KS; ++q) {
      const float2 v = *reinterpret_cast<const float2*>(&red[q * NT + t][row][c2]);
      s.x += v.x; s.y += v.y;
    }
    const int cidx = t * 32 + c2;
    if (cidx >= NOUT) continue;
    if constexpr (MODE == 0) {
      s.x += bias[cidx]; s.y += bias[cidx + 1];
      s.x = s.x > 0.f ? s.x : s.x * slope;
      s.y = s.y > 0.f ? s.y : s.y * slope;
      *reinterpret_cast<float2*>(out + (size_t)node * ldo + cidx) = s;
    } else {
      const float v2[2] = {s.x, s.y};
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int cc_ = cidx + i;
        if (cc_ >= NOUT) continue;
        if (out1 != nullptr && cc_ >= split) out1[(size_t)node * ldo1 + (cc_ - split)] = v2[i];
        else out[(size_t)node * ldo + cc_] = v2[i];
      }
    }
  }
}

// Packed weights of the forward:  Bp[kb][half][col][s] = lin.weight[h * Cout + col, kin] for k = 8 kb + 4 half + s
// = h * Cin + kin (zero for k >= 9 Cin or col >= Cout), NP = padded column count (multiple of 32).
__global__ void pack_fused_fwd_kernel(const float* __restrict__ lin_w, int Cin, int Cout, int KD, int NP,
                                      float* __restrict__ bp) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= KD * NP) return;
  const int s = idx & 3, colx = (idx >> 2) % NP, rest = (idx >> 2) / NP;
  const int hf = rest & 1, kb = rest >> 1;
  const int k = 8 * kb + 4 * hf + s;
  float v = 0.f;
  if (k < H * Cin && colx < Cout) v = lin_w[((size_t)(k / Cin) * Cout + colx) * Cin + (k % Cin)];
  bp[idx] = v;
}

// Packed weights of dx = r' W':  rows k < 9 Cout: lin.weight[k, col]; the next 9: u.weight[k - 9 Cout, col];
// the remaining 15 rows ([dp] padding and the dcs columns of r') are zero.
__global__ void pack_fused_dx_kernel(const float* __restrict__ lin_w, const float* __restrict__ u_w, int Cin, int Cout,
                                     int KD, int NP, float* __restrict__ bp) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= KD * NP) return;
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

template <int C, int VEC, int MODE, int LC, int NT>
int launch_one(const float* xa, const float* xb, int Ca, const float* p, const float* cvec, const int* rowptr,
               const int* col, const int* deg_rowptr, int N, const float* xl, const float* ul, const float* dpd,
               const float* Bp, int NOUT, const float* bias, float slope, float* out, int ldo, float* out1, int split,
               int ldo1, hipStream_t s) {
  constexpr size_t lds = (size_t)fused_lds_floats(C, MODE, LC) * sizeof(float);
  static_assert(lds <= 163840, "tile exceeds the LDS of a CU");
  static bool attr_set = false;
  if (!attr_set) {
    GEOBI_HIP(hipFuncSetAttribute((const void*)feast_fused_kernel<C, VEC, MODE, LC, NT>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  feast_fused_kernel<C, VEC, MODE, LC, NT><<<xcd_grid(cdiv(N, TN)), 512, lds, s>>>(
      xa, xb, Ca, p, cvec, rowptr, col, deg_rowptr, N, xl, ul, dpd, Bp, NOUT, bias, slope, out, ldo, out1, split, ldo1);
  GEOBI_LAUNCH_OK();
  return 0;
}

#define GEOBI_FUSED_ARGS                                                                                            \
  xa, xb, Ca, p, cvec, rowptr, col, deg_rowptr, N, xl, ul, dpd, Bp, NOUT, bias, slope, out, ldo, out1, split, ldo1, s

template <int C, int VEC, int MODE, int LC>
int launch_nt(int NT, const float* xa, const float* xb, int Ca, const float* p, const float* cvec, const int* rowptr,
              const int* col, const int* deg_rowptr, int N, const float* xl, const float* ul, const float* dpd,
              const float* Bp, int NOUT, const float* bias, float slope, float* out, int ldo, float* out1, int split,
              int ldo1, hipStream_t s) {
  // dx with per-edge logits writes a 6- or 12-channel input gradient: one column tile
  constexpr bool kNarrowOnly = MODE == 1 && LC > 0;
  if (NT == 1) return launch_one<C, VEC, MODE, LC, 1>(GEOBI_FUSED_ARGS);
  if constexpr (!kNarrowOnly) {
    if (NT == 2) return launch_one<C, VEC, MODE, LC, 2>(GEOBI_FUSED_ARGS);
    if (NT == 4) return launch_one<C, VEC, MODE, LC, 4>(GEOBI_FUSED_ARGS);
  }
  return set_error("feast fused: unsupported output width %d", NOUT);
}

}  // namespace

int feast_fused_nt(int nout) { return nout <= 32 ? 1 : (nout <= 64 ? 2 : 4); }
size_t feast_fused_fwd_pack_floats(int Cin, int Cout) { return (size_t)fused_k(Cin, 0) * 32 * feast_fused_nt(Cout); }
size_t feast_fused_dx_pack_floats(int Cin, int Cout) { return (size_t)fused_k(Cout, 1) * 32 * feast_fused_nt(Cin); }

int feast_fused_pack_fwd(const float* lin_w, int Cin, int Cout, float* bp, hipStream_t s) {
  const int KD = fused_k(Cin, 0), NP = 32 * feast_fused_nt(Cout);
  pack_fused_fwd_kernel<<<cdiv((int64_t)KD * NP, 256), 256, 0, s>>>(lin_w, Cin, Cout, KD, NP, bp);
  GEOBI_LAUNCH_OK();
  return 0;
}

int feast_fused_pack_dx(const float* lin_w, const float* u_w, int Cin, int Cout, float* bp, hipStream_t s) {
  const int KD = fused_k(Cout, 1), NP = 32 * feast_fused_nt(Cin);
  pack_fused_dx_kernel<<<cdiv((int64_t)KD * NP, 256), 256, 0, s>>>(lin_w, u_w, Cin, Cout, KD, NP, bp);
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
  const float* dpd = nullptr;
  const int NOUT = Cout, ldo = Cout, split = 0, ldo1 = 0;
  float* out1 = nullptr;
  const int NT = feast_fused_nt(Cout);
  switch (Cin * 100 + LC) {
    case 600: return launch_nt<6, 3, 0, 0>(NT, GEOBI_FUSED_ARGS);
    case 606: return launch_nt<6, 3, 0, 6>(NT, GEOBI_FUSED_ARGS);
    case 1200: return launch_nt<12, 3, 0, 0>(NT, GEOBI_FUSED_ARGS);
    case 1212: return launch_nt<12, 3, 0, 12>(NT, GEOBI_FUSED_ARGS);
    case 3200: return launch_nt<32, 4, 0, 0>(NT, GEOBI_FUSED_ARGS);
    case 6400: return launch_nt<64, 4, 0, 0>(NT, GEOBI_FUSED_ARGS);
    case 12800: return launch_nt<128, 4, 0, 0>(NT, GEOBI_FUSED_ARGS);
    default: return set_error("feast fused forward: unsupported Cin=%d (per-edge logit channels %d)", Cin, LC);
  }
}

// backward: (dxa | dxb) = [r | dp | dcs] W', r aggregated over the transposed CSR from g [N, Cout]
int feast_fused_dx(const float* g, int Cout, const float* p, const float* cvec, const int* rowptr_out,
                   const int* col_out, const int* rowptr_in, int N, int LC, const float* xl, const float* ul,
                   const float* dpd, const float* Bp, int Cin, float* dxa, int Ca, float* dxb, int Cb, hipStream_t s) {
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
    case 3200: return launch_nt<32, 4, 1, 0>(NT, GEOBI_FUSED_ARGS);
    case 3206: return launch_nt<32, 4, 1, 6>(NT, GEOBI_FUSED_ARGS);
    case 3212: return launch_nt<32, 4, 1, 12>(NT, GEOBI_FUSED_ARGS);
    case 6400: return launch_nt<64, 4, 1, 0>(NT, GEOBI_FUSED_ARGS);
    case 6406: return launch_nt<64, 4, 1, 6>(NT, GEOBI_FUSED_ARGS);
    case 6412: return launch_nt<64, 4, 1, 12>(NT, GEOBI_FUSED_ARGS);
    case 12800: return launch_nt<128, 4, 1, 0>(NT, GEOBI_FUSED_ARGS);
    case 12806: return launch_nt<128, 4, 1, 6>(NT, GEOBI_FUSED_ARGS);
    case 12812: return launch_nt<128, 4, 1, 12>(NT, GEOBI_FUSED_ARGS);
    default: return set_error("feast fused dx: unsupported Cout=%d (per-edge logit channels %d)", Cout, LC);
  }
}

}  // namespace geobi
