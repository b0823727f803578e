// Whole-network forward as one library call (inference): native host code walks the op sequence of
// DualGNN.forward (/root/reference/code/network.py:318-343), GNNModule.forward (:270-300) and PoolingLayer.forward
// (/root/reference/code/net_util.py:76-158) over the launchers of this library.  The module-by-module Python path
// issues the same launches in the same order; below ~10 k faces it is bound by the ~10 us of interpreter work
// around each of its ~180 launches, which this removes.
#include "common.h"
#include "../../include/geobi_hip.h"

namespace geobi {

namespace {

constexpr float kLeak = 0.2f;
constexpr int kRounds = 8;             // net_util.MATCH_ROUNDS
constexpr int kArenaFull = GEOBI_NET_ARENA;
constexpr int kFallback = GEOBI_NET_FALLBACK;

struct Level {
  int64_t N = 0, E = 0;
  const int32_t *rowptr = nullptr, *col = nullptr, *row = nullptr;
  const float* w = nullptr;
};

// Bump allocator over the caller's arena.  Results are taken first, per-call scratch after a mark that is released
// once the call is enqueued: everything runs on one stream, so a later kernel may reuse the bytes.
struct Bump {
  char* base;
  size_t off = 0, cap, peak = 0;
  bool ok = true;
  Bump(void* p, size_t bytes) : base((char*)p), cap(bytes) {}
  template <typename T>
  T* take(size_t n) {
    const size_t o = align_up(off);
    off = o + (n ? n : 1) * sizeof(T);
    if (off > peak) peak = off;
    if (off > cap) { ok = false; return nullptr; }
    return (T*)(base + o);
  }
  size_t mark() const { return off; }
  void release(size_t m) { off = m; }
  int64_t offset_of(const void* p) const { return (const char*)p - base; }
};

__global__ void compose_index_kernel(const int32_t* __restrict__ first, const int32_t* __restrict__ second, int64_t n,
                                     int32_t* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = second[first[i]];
}

// FeaStConv forward (no gradient): `out` is a result, logits and workspace are scratch
int conv_fwd(Bump& b, const Level& g, const float* xa, const float* xb, int Ca, int Cb, const geobi_conv_params_t& p,
             int Cout, float slope, float** out, hipStream_t s) {
  const int64_t N = g.N;
  float* o = b.take<float>((size_t)N * Cout);
  const size_t m = b.mark();
  float* logits = b.take<float>((size_t)N * GEOBI_HP);
  const size_t wsb = feast_fwd_ws_bytes(N, Ca + Cb, Cout);
  void* ws = b.take<char>(wsb);
  if (!b.ok) return kArenaFull;
  GEOBI_TRY(feast_fwd(xa, Cb ? xb : nullptr, Ca, Cb, N, g.E, g.rowptr, g.col, p.lin_w, p.u_w, p.c, p.bias, Cout, slope, o,
                      logits, nullptr, nullptr, ws, wsb, s));
  b.release(m);
  *out = o;
  return 0;
}

struct PoolResult {
  Level coarse;
  float* x = nullptr;            // pooled features [coarse.N, C]
  int32_t* unpool = nullptr;     // composed fine -> coarse index [fine.N]
  int32_t* raw[2] = {nullptr, nullptr};
  int64_t raw_len[2] = {0, 0};
};

// PoolingLayer.forward with edge_weight_type 10 and two matching steps: the integer pipeline of both steps is
// enqueued back to back (the second on the first's coarse graph padded to the fine node count, see
// net_util._coarsen_chain), ONE read-back returns the sizes, the features follow with exact sizes.
int pool_layer(Bump& b, const Level& g, const float* x, int C, int pool_mean, PoolResult& r, hipStream_t s) {
  const int64_t P = g.N, E = g.E;
  if (E <= 0 || P <= 0) return kFallback;
  // ---- results of the integer pipeline (kept: graphs, lists, cluster vectors)
  float* w10 = b.take<float>(E);
  int32_t* counters = b.take<int32_t>(8);
  int32_t *state[2], *craw[2], *cnew[2], *segptr[2], *members[2], *rowptr_c[2], *row_c[2], *col_c[2];
  float* w_c[2];
  for (int t = 0; t < 2; ++t) {
    state[t] = b.take<int32_t>(P); craw[t] = b.take<int32_t>(P); cnew[t] = b.take<int32_t>(P);
    segptr[t] = b.take<int32_t>(P + 1); members[t] = b.take<int32_t>(P);
    rowptr_c[t] = b.take<int32_t>(P + 1); row_c[t] = b.take<int32_t>(E); col_c[t] = b.take<int32_t>(E);
    w_c[t] = b.take<float>(E);
  }
  const size_t m = b.mark();
  const size_t mws = match_coarsen_ws_bytes(P), pws = pool_edge_rows_ws_bytes(P);
  void* ws_m = b.take<char>(mws);
  void* ws_p = b.take<char>(pws);
  if (!b.ok) return kArenaFull;
  GEOBI_HIP(hipMemsetAsync(counters, 0, 8 * sizeof(int32_t), s));
  GEOBI_TRY(edge_weight_t10(x, C, g.row, g.col, g.w, E, w10, s));
  // inputs of step t: the level itself (t = 0) or step 0's coarse graph padded to P rows (t = 1)
  const int32_t* in_rp[2] = {g.rowptr, rowptr_c[0]};
  const int32_t* in_cl[2] = {g.col, col_c[0]};
  const int32_t* in_rw[2] = {g.row, row_c[0]};
  const float* in_w[2] = {w10, w_c[0]};
  int rounds[2] = {kRounds, kRounds};
  auto run_match = [&](int t, int init) {
    return match_coarsen(in_rp[t], in_cl[t], in_w[t], P, rounds[t], init, state[t], craw[t], cnew[t], segptr[t], members[t],
                         counters + 4 * t, ws_m, mws, s);
  };
  auto run_edges = [&](int t) {
    int32_t* ctr = counters + 4 * t;
    return pool_edge_rows(cnew[t], segptr[t], members[t], in_rp[t], in_cl[t], in_w[t], ctr + 1, P, rowptr_c[t], row_c[t],
                          col_c[t], w_c[t], ctr + 2, ctr + 3, ws_p, pws, s);
  };
  for (int t = 0; t < 2; ++t) {
    GEOBI_TRY(run_match(t, 1));
    GEOBI_TRY(run_edges(t));
  }
  int32_t h[8];
  GEOBI_TRY(geobi_read_i32(counters, 8, h, (void*)s));
  // Rare repairs, each followed by a fresh read (the reference syncs after every step anyway): a matching whose
  // proposal chains outlast the rounds run so far is resumed from its saved state with twice the rounds; a coarse
  // row wider than the sort-free 64 entries sends that step through the radix-sort edge coarsening.  A repaired
  // first step invalidates the second, which is then redone from scratch.
  for (int repair = 0; repair < 12 && (h[0] || h[3] || h[4] || h[7]); ++repair) {
    const int t = (h[0] || h[3]) ? 0 : 1;
    if (h[4 * t]) {                                   // undecided nodes: resume
      if (rounds[t] >= 2048) break;
      rounds[t] *= 2;
      GEOBI_TRY(run_match(t, 0));
      GEOBI_HIP(hipMemsetAsync(counters + 4 * t + 2, 0, 2 * sizeof(int32_t), s));
      GEOBI_TRY(run_edges(t));
    } else {                                          // sort-free width exceeded: general path for this step
      const int64_t Et = t == 0 ? E : (int64_t)h[2];
      const size_t gws = pool_edge_ws_bytes(Et);
      const size_t mk = b.mark();
      void* ws_g = b.take<char>(gws);
      if (!b.ok) return kArenaFull;
      GEOBI_HIP(hipMemsetAsync(counters + 4 * t + 2, 0, 2 * sizeof(int32_t), s));
      GEOBI_TRY(pool_edge(cnew[t], in_rw[t], in_cl[t], in_w[t], Et, P, rowptr_c[t], row_c[t], col_c[t], w_c[t],
                          counters + 4 * t + 2, ws_g, gws, s));
      b.release(mk);
    }
    if (t == 0) {
      rounds[1] = kRounds;
      GEOBI_HIP(hipMemsetAsync(counters + 4, 0, 4 * sizeof(int32_t), s));
      GEOBI_TRY(run_match(1, 1));
      GEOBI_TRY(run_edges(1));
    }
    GEOBI_TRY(geobi_read_i32(counters, 8, h, (void*)s));
  }
  b.release(m);
  if (h[0] || h[3] || h[4] || h[7]) return kFallback;         // not repaired within the budget
  const int64_t R1 = h[1], E1 = h[2];
  const int64_t R2 = (int64_t)h[5] - (P - R1), E2 = h[6];     // the padding nodes came back as singletons
  if (E1 <= 0 || E2 <= 0 || R1 <= 0 || R2 <= 0) return kFallback;
  // ---- features with exact sizes
  float* x1 = b.take<float>((size_t)R1 * C);
  float* x2 = b.take<float>((size_t)R2 * C);
  int32_t* unpool = b.take<int32_t>(P);
  const size_t m2 = b.mark();
  int32_t* arg = b.take<int32_t>((size_t)R1 * C);
  if (!b.ok) return kArenaFull;
  if (pool_mean) {
    GEOBI_TRY(segment_sum(x, C, segptr[0], members[0], R1, 1, x1, s));
    GEOBI_TRY(segment_sum(x1, C, segptr[1], members[1], R2, 1, x2, s));
  } else {
    GEOBI_TRY(segment_max_fwd(x, C, segptr[0], members[0], R1, x1, arg, s));
    GEOBI_TRY(segment_max_fwd(x1, C, segptr[1], members[1], R2, x2, arg, s));
  }
  compose_index_kernel<<<cdiv(P, 256), 256, 0, s>>>(cnew[0], cnew[1], P, unpool);
  GEOBI_LAUNCH_OK();
  b.release(m2);
  r.coarse.N = R2; r.coarse.E = E2;
  r.coarse.rowptr = rowptr_c[1]; r.coarse.col = col_c[1]; r.coarse.row = row_c[1]; r.coarse.w = w_c[1];
  r.x = x2; r.unpool = unpool;
  r.raw[0] = craw[0]; r.raw_len[0] = P;
  r.raw[1] = craw[1]; r.raw_len[1] = R1;
  return 0;
}

int unpool_rows(Bump& b, const float* x, const int32_t* idx, int C, int64_t n, float** out, hipStream_t s) {
  float* o = b.take<float>((size_t)n * C);
  if (!b.ok) return kArenaFull;
  GEOBI_TRY(gather_rows(x, idx, C, n, o, s));
  *out = o;
  return 0;
}

// GNNModule.forward (network.py:270-300): channel plan Cin -> 32 | 64 | 128, 128 | 64, 64 | 32, 32
int gnn_forward(Bump& b, const Level& L0, const float* x_in, int Cin, const geobi_gnn_params_t& p, int pool_mean,
                float** feat, geobi_branch_out_t& bo, hipStream_t s) {
  float *x0, *x1, *x2a, *x2, *up2, *r1, *x1b, *up1, *r3, *out;
  GEOBI_TRY(conv_fwd(b, L0, x_in, nullptr, Cin, 0, p.conv[0], 32, kLeak, &x0, s));
  PoolResult p1, p2;
  GEOBI_TRY(pool_layer(b, L0, x0, 32, pool_mean, p1, s));
  const Level& L1 = p1.coarse;
  GEOBI_TRY(conv_fwd(b, L1, p1.x, nullptr, 32, 0, p.conv[1], 64, kLeak, &x1, s));
  GEOBI_TRY(pool_layer(b, L1, x1, 64, pool_mean, p2, s));
  const Level& L2 = p2.coarse;
  GEOBI_TRY(conv_fwd(b, L2, p2.x, nullptr, 64, 0, p.conv[2], 128, kLeak, &x2a, s));
  GEOBI_TRY(conv_fwd(b, L2, x2a, nullptr, 128, 0, p.conv[3], 128, kLeak, &x2, s));
  GEOBI_TRY(unpool_rows(b, x2, p2.unpool, 128, L1.N, &up2, s));
  GEOBI_TRY(conv_fwd(b, L1, up2, nullptr, 128, 0, p.conv[4], 64, 1.0f, &r1, s));
  GEOBI_TRY(conv_fwd(b, L1, x1, r1, 64, 64, p.conv[5], 64, kLeak, &x1b, s));
  GEOBI_TRY(unpool_rows(b, x1b, p1.unpool, 64, L0.N, &up1, s));
  GEOBI_TRY(conv_fwd(b, L0, up1, nullptr, 64, 0, p.conv[6], 32, 1.0f, &r3, s));
  GEOBI_TRY(conv_fwd(b, L0, x0, r3, 32, 32, p.conv[7], 32, kLeak, &out, s));
  *feat = out;
  bo.nodes[0] = L0.N; bo.nodes[1] = L1.N; bo.nodes[2] = L2.N;
  const PoolResult* pr[2] = {&p1, &p2};
  for (int l = 0; l < 2; ++l) {
    bo.unpool_off[l] = b.offset_of(pr[l]->unpool);
    for (int t = 0; t < 2; ++t) {
      bo.cluster_off[l][t] = b.offset_of(pr[l]->raw[t]);
      bo.cluster_len[l][t] = pr[l]->raw_len[t];
    }
  }
  return 0;
}

}  // namespace

}  // namespace geobi

using namespace geobi;

extern "C" size_t geobi_net_forward_arena_bytes(int64_t V, int64_t Ev, int64_t F, int64_t Ef) {
  // per branch: level-0 features (<= 32 + 64 + 32 + 32 floats per node live at once plus the widest scratch), the
  // integer arrays of two pooling layers (bounded by the level-0 sizes) and their workspaces
  auto branch = [](int64_t N, int64_t E) {
    return (size_t)N * (4 * 512 + 4 * 2 * 16) + (size_t)E * (4 * 2 * 8) + match_coarsen_ws_bytes(N) +
           pool_edge_rows_ws_bytes(N) + feast_fwd_ws_bytes(N, 128, 128) + ((size_t)8 << 20);
  };
  return branch(V, Ev) + branch(F, Ef) + (size_t)F * 12 * 4 + ((size_t)16 << 20);
}

extern "C" int geobi_net_forward(const geobi_net_params_t* prm, const geobi_level0_t* gv, const geobi_level0_t* gf,
                                 const float* x_v, const float* x_f, const int32_t* fv, const float* depth_direction,
                                 void* arena, size_t arena_bytes, geobi_net_out_t* out, void* stream) {
  if (!prm || !gv || !gf || !x_v || !x_f || !fv || !arena || !out) return set_error("geobi_net_forward: null argument");
  if (prm->force_depth && !depth_direction) return set_error("geobi_net_forward: force_depth needs depth_direction");
  hipStream_t s = (hipStream_t)stream;
  Bump b(arena, arena_bytes);
  Level Lv, Lf;
  Lv.N = gv->N; Lv.E = gv->E; Lv.rowptr = gv->rowptr; Lv.col = gv->col; Lv.row = gv->row; Lv.w = gv->weight;
  Lf.N = gf->N; Lf.E = gf->E; Lf.rowptr = gf->rowptr; Lf.col = gf->col; Lf.row = gf->row; Lf.w = gf->weight;
  const int64_t V = Lv.N, F = Lf.N;
  auto fail = [&](int rc) {
    out->used_bytes = (int64_t)b.peak;
    if (rc == kArenaFull) set_error("geobi_net_forward: arena too small (%zu bytes needed so far, %zu given)", b.peak, arena_bytes);
    return rc;
  };
  float *feat_v = nullptr, *feat_f = nullptr;
  int rc = gnn_forward(b, Lv, x_v, 6, prm->gnn_v, prm->pool_mean, &feat_v, out->v, s);
  if (rc) return fail(rc);
  // vertex head: fc_v2(leaky(fc_v1 .)) -> displacement (or depth along depth_direction) + xyz   (network.py:324-332)
  const int nout_v = prm->force_depth ? 1 : 3;
  float* verts = b.take<float>((size_t)V * 3);
  float* raw_v = b.take<float>((size_t)V * nout_v);
  float* xf12 = b.take<float>((size_t)F * 12);
  if (!b.ok) return fail(kArenaFull);
  rc = head_fwd(feat_v, 32, V, prm->fc_v1_w, prm->fc_v1_b, 1024, prm->fc_v2_w, prm->fc_v2_b, nout_v, kLeak, 0,
                prm->force_depth ? depth_direction : nullptr, x_v, 6, nullptr, raw_v, verts, s);
  if (rc) return fail(rc);
  // geometry coupling: x_f = [x_f | centroid | unit normal] of the predicted geometry   (network.py:335-337)
  rc = face_geom_fwd(verts, fv, x_f, 6, F, xf12, s);
  if (rc) return fail(rc);
  rc = gnn_forward(b, Lf, xf12, 12, prm->gnn_f, prm->pool_mean, &feat_f, out->f, s);
  if (rc) return fail(rc);
  float* normals = b.take<float>((size_t)F * 3);
  float* raw_f = b.take<float>((size_t)F * 3);
  if (!b.ok) return fail(kArenaFull);
  rc = head_fwd(feat_f, 32, F, prm->fc_f1_w, prm->fc_f1_b, 1024, prm->fc_f2_w, prm->fc_f2_b, 3, kLeak, 1, nullptr, nullptr,
                0, nullptr, raw_f, normals, s);
  if (rc) return fail(rc);
  out->verts_off = b.offset_of(verts);
  out->normals_off = b.offset_of(normals);
  out->xf_off = b.offset_of(xf12);
  out->used_bytes = (int64_t)b.peak;
  return 0;
}
