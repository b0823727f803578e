// Whole-network forward as one library call (inference): native host code walks the op sequence of
// DualGNN.forward (/root/reference/code/network.py:318-343), GNNModule.forward (:270-300) and PoolingLayer.forward
// (/root/reference/code/net_util.py:76-158) over the launchers of this library.  The module-by-module Python path
// issues the same launches in the same order; below ~10 k faces it is bound by the ~10 us of interpreter work
// around each of its ~180 launches, which this removes.
#include "common.h"
#include "../../include/geobi_hip.h"

#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace geobi {

namespace {

// one step of a bounded busy wait (host code: the pause hint exists on x86 only)
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#else
  std::this_thread::yield();
#endif
}
std::atomic<long> g_spin_cap_hits{0};       // size spins that ran into their cap and fell back to the blocking read

constexpr float kLeak = 0.2f;
constexpr int kRounds = 8;             // net_util.MATCH_ROUNDS
constexpr int kArenaFull = GEOBI_NET_ARENA;
constexpr int kFallback = GEOBI_NET_FALLBACK;

struct Level {
  int64_t N = 0, E = 0;
  const int32_t *rowptr = nullptr, *col = nullptr, *row = nullptr;
  const int32_t* pos_rev = nullptr;      // position of the reverse edge (training: the backward's transposition)
  const float* w = nullptr;
};

// What the backward of one op needs (host-side record of a training forward; device data lives in the arena)
struct ConvSave {
  const Level* g = nullptr;
  const float *xa = nullptr, *xb = nullptr;
  int Ca = 0, Cb = 0, Cout = 0;
  float slope = 1.f;
  const geobi_conv_params_t* p = nullptr;
  float *out = nullptr, *logits = nullptr, *wf = nullptr;
};
struct PoolSave {
  int C = 0, pool_mean = 0;
  int64_t P = 0, R1 = 0, R2 = 0;
  const int32_t *seg[2] = {nullptr, nullptr}, *segptr[2] = {nullptr, nullptr};
  const int32_t* arg[2] = {nullptr, nullptr};
  const int32_t* members[2] = {nullptr, nullptr};              // with segptr: the inverse lists of the two steps
  const int32_t* unpool = nullptr;                             // composed fine -> coarse index (max-pool backward)
};
struct BranchTape {
  Level L[3];
  ConvSave conv[8];
  PoolSave pool[2];
  int Cin = 0;
};
struct NetTape {
  geobi_net_params_t prm;
  BranchTape v, f;
  int64_t V = 0, F = 0;
  const float *x_v = nullptr, *dd = nullptr;
  const int32_t* fv = nullptr;
  float *feat_v = nullptr, *feat_f = nullptr, *raw_v = nullptr, *raw_f = nullptr, *verts = nullptr;
  char* arena = nullptr;
  size_t arena_bytes = 0, fwd_peak = 0;
  int side_low = 0;                       // weight-gradient side stream at the lowest priority whatever the batch size
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

// FeaStConv forward.  Inference: logits and workspace are scratch.  Training (`save`): logits and the packed weights
// stay for the backward.
// `packed`: the layer's weights as pack_branch left them (training: every form, the buffer the backward reads;
// inference: the forward form).
int conv_fwd(Bump& b, const Level& g, const float* xa, const float* xb, int Ca, int Cb, const geobi_conv_params_t& p,
             int Cout, float slope, float** out, ConvSave* save, float* packed, hipStream_t s) {
  const int64_t N = g.N;
  float* o = b.take<float>((size_t)N * Cout);
  float *logits = nullptr, *wf = nullptr;
  if (save) {
    logits = b.take<float>((size_t)N * GEOBI_HP);
    wf = packed;
  }
  const float* bf = save ? packed + feast_wpack_plain_floats(Ca + Cb, Cout) : packed;
  const size_t m = b.mark();
  if (!save) logits = b.take<float>((size_t)N * GEOBI_HP);
  const size_t wsb = feast_fwd_ws_bytes(N, Ca + Cb, Cout);
  void* ws = b.take<char>(wsb);
  if (!b.ok) return kArenaFull;
  GEOBI_TRY(feast_fwd(xa, Cb ? xb : nullptr, Ca, Cb, N, g.E, g.rowptr, g.col, p.lin_w, p.u_w, p.c, p.bias, Cout, slope, o,
                      logits, nullptr, wf, ws, wsb, s, bf));
  b.release(m);
  if (save) {
    save->g = &g; save->xa = xa; save->xb = Cb ? xb : nullptr; save->Ca = Ca; save->Cb = Cb; save->Cout = Cout;
    save->slope = slope; save->p = &p; save->out = o; save->logits = logits; save->wf = wf;
  }
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
int pool_layer(Bump& b, const Level& g, const float* x, int C, int pool_mean, PoolResult& r, PoolSave* save,
               hipStream_t s) {
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
  int32_t* unpool = b.take<int32_t>(P);          // composed fine -> coarse index (its size does not wait for the counts)
  const size_t m = b.mark();
  const size_t mws = match_coarsen_ws_bytes(P), pws = pool_edge_rows_ws_bytes_onepass(P, E);
  void* ws_m = b.take<char>(mws);
  void* ws_p = b.take<char>(pws);
  if (!b.ok) return kArenaFull;
  GEOBI_TRY(edge_weight_t10(x, C, g.row, g.col, g.w, E, w10, s, counters));      // also clears the 8 counters
  // inputs of step t: the level itself (t = 0) or step 0's coarse graph padded to P rows (t = 1)
  const int32_t* in_rp[2] = {g.rowptr, rowptr_c[0]};
  const int32_t* in_cl[2] = {g.col, col_c[0]};
  const int32_t* in_rw[2] = {g.row, row_c[0]};
  const float* in_w[2] = {w10, w_c[0]};
  int rounds[2] = {kRounds, kRounds};
  // the matching leaves each coarse node's member rows for the edge coarsening right behind it
  void* rowinfo = b.take<char>((size_t)P * 16);
  if (!b.ok) return kArenaFull;
  bool rowinfo_ok = false;
  auto run_match = [&](int t, int init) {
    return match_coarsen(in_rp[t], in_cl[t], in_w[t], P, rounds[t], init, state[t], craw[t], cnew[t], segptr[t], members[t],
                         counters + 4 * t, ws_m, mws, s, rowinfo, &rowinfo_ok);
  };
  // The sizes come back through mapped host memory, written by the last kernel of the pipeline as soon as it starts
  // (pool_edge_compact_kernel), not by a copy behind it.
  static thread_local int32_t* pub = nullptr;
  static thread_local int pub_seq = 0;
  if (pub == nullptr) {
    // portable: visible to every device context of the process, whichever was current at the first call
    GEOBI_HIP(hipHostMalloc((void**)&pub, 16 * sizeof(int32_t), hipHostMallocMapped | hipHostMallocPortable));
    for (int i = 0; i < 16; ++i) pub[i] = 0;
  }
  auto run_edges = [&](int t, bool publish) {
    int32_t* ctr = counters + 4 * t;
    return pool_edge_rows(cnew[t], segptr[t], members[t], in_rp[t], in_cl[t], in_w[t], ctr + 1, P, rowptr_c[t], row_c[t],
                          col_c[t], w_c[t], ctr + 2, ctr + 3, ws_p, pws, s, E,      // E bounds both steps' edge counts
                          rowinfo_ok ? rowinfo : nullptr, publish ? counters : nullptr, publish ? pub : nullptr,
                          pub_seq);
  };
  pub_seq = pub_seq >= (1 << 30) ? 1 : pub_seq + 1;
  for (int t = 0; t < 2; ++t) {
    GEOBI_TRY(run_match(t, 1));
    GEOBI_TRY(run_edges(t, t == 1));
  }
  // something for the device to do while the host reads the sizes and prepares the next launches
  compose_index_kernel<<<cdiv(P, 256), 256, 0, s>>>(cnew[0], cnew[1], P, unpool);
  GEOBI_LAUNCH_OK();
  int32_t h[8];
  {
    volatile int32_t* vp = pub;
    long spins = 0;
    bool got = false;
    // bounded: the stream is polled every 2^14 spins (a fault or an idle stream is noticed within microseconds), and
    // after ~2^26 spins (seconds) the blocking read below takes over whatever happened to the mapped word
    while (!(got = (__atomic_load_n(&pub[8], __ATOMIC_ACQUIRE) == pub_seq))) {
      ++spins;
      if ((spins & 0x3f) == 0) cpu_relax();
      if ((spins & 0x3fff) == 0 && hipStreamQuery(s) != hipErrorNotReady) {        // the stream ran dry without a word
        got = __atomic_load_n(&pub[8], __ATOMIC_ACQUIRE) == pub_seq;
        break;
      }
      if (spins > (1l << 26)) break;
    }
    if (got) {
      for (int i = 0; i < 8; ++i) h[i] = vp[i];
    } else {
      // the mapped word never showed up (or the stream ran dry first): counted, so that a device that stops
      // publishing degrades to the blocking read VISIBLY (geobi_net_spin_cap_hits) instead of silently stalling
      if (spins > (1l << 26)) g_spin_cap_hits.fetch_add(1, std::memory_order_relaxed);
      GEOBI_TRY(geobi_read_i32(counters, 8, h, (void*)s));
    }
  }
  // Rare repairs, each followed by a fresh read (the reference syncs after every step anyway): a matching whose
  // proposal chains outlast the rounds run so far is resumed from its saved state with twice the rounds; a coarse
  // row wider than the sort-free 64 entries sends that step through the radix-sort edge coarsening.  A repaired
  // first step invalidates the second, which is then redone from scratch.
  bool repaired = false;
  for (int repair = 0; repair < 12 && (h[0] || h[3] || h[4] || h[7]); ++repair) {
    repaired = true;
    const int t = (h[0] || h[3]) ? 0 : 1;
    if (h[4 * t]) {                                   // undecided nodes: resume
      if (rounds[t] >= 2048) break;
      rounds[t] *= 2;
      GEOBI_TRY(run_match(t, 0));
      GEOBI_HIP(hipMemsetAsync(counters + 4 * t + 2, 0, 2 * sizeof(int32_t), s));
      GEOBI_TRY(run_edges(t, false));
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
      GEOBI_TRY(run_edges(1, false));
    }
    GEOBI_TRY(geobi_read_i32(counters, 8, h, (void*)s));
  }
  if (repaired) {                                   // the cluster vectors changed under the early composition
    compose_index_kernel<<<cdiv(P, 256), 256, 0, s>>>(cnew[0], cnew[1], P, unpool);
    GEOBI_LAUNCH_OK();
  }
  b.release(m);
  if (h[0] || h[3] || h[4] || h[7]) return kFallback;         // not repaired within the budget
  const int64_t R1 = h[1], E1 = h[2];
  const int64_t R2 = (int64_t)h[5] - (P - R1), E2 = h[6];     // the padding nodes came back as singletons
  if (E1 <= 0 || E2 <= 0 || R1 <= 0 || R2 <= 0) return kFallback;
  // ---- features with exact sizes
  float* x2 = b.take<float>((size_t)R2 * C);
  int32_t *arg2 = nullptr, *pos_rev = nullptr;
  if (save) {       // training: arg-max rows and the coarse level's reverse-edge index stay
    arg2 = b.take<int32_t>((size_t)R2 * C);
    pos_rev = b.take<int32_t>(E2);
  }
  const size_t m2 = b.mark();
  if (!save) arg2 = b.take<int32_t>((size_t)R2 * C);
  float* x1 = pool_mean ? b.take<float>((size_t)R1 * C) : nullptr;      // the mean needs the step-one rows
  if (!b.ok) return kArenaFull;
  if (pool_mean) {
    GEOBI_TRY(segment_sum(x, C, segptr[0], members[0], R1, 1, x1, s));
    GEOBI_TRY(segment_sum(x1, C, segptr[1], members[1], R2, 1, x2, s));
  } else {
    // both matching steps in one pass over the composed segments: same values and the same arg-max routing as two
    // segment_max passes (pool.hip), the step-one maxima are never written
    GEOBI_TRY(segment_max2_fwd(x, C, segptr[0], members[0], segptr[1], members[1], R2, x2, arg2, s));
  }
  if (save) {
    GEOBI_TRY(csr_reverse_index(rowptr_c[1], row_c[1], col_c[1], E2, pos_rev, nullptr, s));
    save->C = C; save->pool_mean = pool_mean; save->P = P; save->R1 = R1; save->R2 = R2;
    save->seg[0] = cnew[0]; save->seg[1] = cnew[1]; save->segptr[0] = segptr[0]; save->segptr[1] = segptr[1];
    save->arg[0] = nullptr; save->arg[1] = arg2; save->members[0] = members[0]; save->members[1] = members[1];
    save->unpool = unpool;
  }
  b.release(m2);
  r.coarse.pos_rev = pos_rev;
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
                float** feat, geobi_branch_out_t& bo, BranchTape* tape, hipStream_t s) {
  float *x0, *x1, *x2a, *x2, *up2, *r1, *x1b, *up1, *r3, *out;
  PoolResult p1, p2;
  Level L1s, L2s;
  // levels referenced by the saved records must outlive this call: they live in the tape when there is one
  Level& L0r = tape ? tape->L[0] : const_cast<Level&>(L0);
  if (tape) { tape->L[0] = L0; tape->Cin = Cin; }
  auto cs = [&](int i) { return tape ? &tape->conv[i] : nullptr; };
  // the packed weights of the branch's eight layers, one launch (training: every form, kept for the backward;
  // inference: the forward forms)
  const int plan[8][2] = {{Cin, 32}, {32, 64}, {64, 128}, {128, 128}, {128, 64}, {128, 64}, {64, 32}, {64, 32}};
  float* pk[8];
  {
    FusedPackItem items[8];
    for (int i = 0; i < 8; ++i) {
      const int ci = plan[i][0], co = plan[i][1];
      pk[i] = b.take<float>(tape ? feast_wpack_floats(ci, co) : feast_fused_fwd_pack_floats(ci, co));
      if (!b.ok) return kArenaFull;
      items[i].lin_w = p.conv[i].lin_w; items[i].u_w = p.conv[i].u_w; items[i].c = p.conv[i].c;
      items[i].Cin = ci; items[i].Cout = co;
      if (tape) {
        items[i].wf = pk[i];
        items[i].bf = pk[i] + feast_wpack_plain_floats(ci, co);
        items[i].bdx = items[i].bf + feast_fused_fwd_pack_floats(ci, co);
      } else {
        items[i].wf = nullptr; items[i].bf = pk[i]; items[i].bdx = nullptr;
      }
    }
    GEOBI_TRY(feast_fused_pack_batch(items, 8, s));
  }
  GEOBI_TRY(conv_fwd(b, L0r, x_in, nullptr, Cin, 0, p.conv[0], 32, kLeak, &x0, cs(0), pk[0], s));
  GEOBI_TRY(pool_layer(b, L0r, x0, 32, pool_mean, p1, tape ? &tape->pool[0] : nullptr, s));
  Level& L1 = tape ? tape->L[1] : L1s;
  L1 = p1.coarse;
  GEOBI_TRY(conv_fwd(b, L1, p1.x, nullptr, 32, 0, p.conv[1], 64, kLeak, &x1, cs(1), pk[1], s));
  GEOBI_TRY(pool_layer(b, L1, x1, 64, pool_mean, p2, tape ? &tape->pool[1] : nullptr, s));
  Level& L2 = tape ? tape->L[2] : L2s;
  L2 = p2.coarse;
  GEOBI_TRY(conv_fwd(b, L2, p2.x, nullptr, 64, 0, p.conv[2], 128, kLeak, &x2a, cs(2), pk[2], s));
  GEOBI_TRY(conv_fwd(b, L2, x2a, nullptr, 128, 0, p.conv[3], 128, kLeak, &x2, cs(3), pk[3], s));
  GEOBI_TRY(unpool_rows(b, x2, p2.unpool, 128, L1.N, &up2, s));
  GEOBI_TRY(conv_fwd(b, L1, up2, nullptr, 128, 0, p.conv[4], 64, 1.0f, &r1, cs(4), pk[4], s));
  GEOBI_TRY(conv_fwd(b, L1, x1, r1, 64, 64, p.conv[5], 64, kLeak, &x1b, cs(5), pk[5], s));
  GEOBI_TRY(unpool_rows(b, x1b, p1.unpool, 64, L0.N, &up1, s));
  GEOBI_TRY(conv_fwd(b, L0r, up1, nullptr, 64, 0, p.conv[6], 32, 1.0f, &r3, cs(6), pk[6], s));
  GEOBI_TRY(conv_fwd(b, L0r, x0, r3, 32, 32, p.conv[7], 32, kLeak, &out, cs(7), pk[7], s));
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

__global__ void add_inplace_kernel(float* __restrict__ a, const float* __restrict__ bsrc, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] += bsrc[i];
}

// Backward of one FeaStConv from its saved record; gradients w.r.t. the input halves go to dxa / dxb (nullable)
int conv_bwd(Bump& b, const ConvSave& c, const geobi_conv_params_t& grad, const float* gout, float* dxa, float* dxb,
             int accumulate, hipStream_t s) {
  const Level& g = *c.g;
  const size_t m = b.mark();
  const size_t wsb = feast_bwd_ws_bytes_for(g.N, g.E, c.Ca + c.Cb, c.Cb, c.Cout, dxa != nullptr);
  void* ws = b.take<char>(wsb);
  if (!b.ok) return kArenaFull;
  GEOBI_TRY(feast_bwd(c.xa, c.xb, c.Ca, c.Cb, g.N, g.E, g.rowptr, g.col, g.rowptr, g.col, g.pos_rev, c.p->lin_w, c.p->u_w,
                      c.p->c, c.Cout, c.slope, c.out, gout, c.logits, nullptr, c.wf, dxa, dxb, (float*)grad.lin_w,
                      (float*)grad.u_w, (float*)grad.c, (float*)grad.bias, accumulate, ws, wsb, s));
  // the side stream may still read the workspace (weight-gradient GEMMs, joined once at the end of the backward):
  // scratch of consecutive layers must not alias -> no release here; the backward region is bump-only
  (void)m;
  return 0;
}

// pooling backward: gradient of the pooled features [R2, C], ADDED to `acc` [P, C] (the gradient the layer input
// already has from its skip connection)
int pool_bwd(Bump& b, const PoolSave& p, const float* g2, float* acc, hipStream_t s) {
  if (!p.pool_mean) return segment_max2_bwd(g2, p.arg[1], p.unpool, p.C, p.R2, p.P, acc, 1, s);
  float* g1 = b.take<float>((size_t)p.R1 * p.C);
  float* g0 = b.take<float>((size_t)p.P * p.C);
  if (!b.ok) return kArenaFull;
  GEOBI_TRY(segment_mean_bwd(g2, p.seg[1], p.segptr[1], p.C, p.R1, g1, s));
  GEOBI_TRY(segment_mean_bwd(g1, p.seg[0], p.segptr[0], p.C, p.P, g0, s));
  add_inplace_kernel<<<cdiv(p.P * p.C, 256), 256, 0, s>>>(acc, g0, p.P * p.C);
  GEOBI_LAUNCH_OK();
  return 0;
}

// GNNModule backward: g_out = gradient of the branch output [N0, 32]; dx_in (nullable) = gradient of its input
int gnn_backward(Bump& b, const BranchTape& t, const geobi_gnn_params_t& grad, const float* g_out, float* dx_in,
                 int accumulate, hipStream_t s) {
  const int64_t N0 = t.L[0].N, N1 = t.L[1].N, N2 = t.L[2].N;
  float* g_x0 = b.take<float>((size_t)N0 * 32);
  float* g_r3 = b.take<float>((size_t)N0 * 32);
  float* g_up1 = b.take<float>((size_t)N0 * 64);
  float* g_x1b = b.take<float>((size_t)N1 * 64);
  float* g_x1 = b.take<float>((size_t)N1 * 64);
  float* g_r1 = b.take<float>((size_t)N1 * 64);
  float* g_up2 = b.take<float>((size_t)N1 * 128);
  float* g_x2 = b.take<float>((size_t)N2 * 128);
  float* g_x2a = b.take<float>((size_t)N2 * 128);
  float* g_x2p = b.take<float>((size_t)N2 * 64);
  float* g_x1p = b.take<float>((size_t)N1 * 32);
  if (!b.ok) return kArenaFull;
  GEOBI_TRY(conv_bwd(b, t.conv[7], grad.conv[7], g_out, g_x0, g_r3, accumulate, s));
  GEOBI_TRY(conv_bwd(b, t.conv[6], grad.conv[6], g_r3, g_up1, nullptr, accumulate, s));
  // unpool backward: sums over the composed segments, walked through the two steps' lists (no composed list is built)
  GEOBI_TRY(segment_sum2(g_up1, 64, t.pool[0].segptr[0], t.pool[0].members[0], t.pool[0].segptr[1], t.pool[0].members[1],
                         N1, g_x1b, s));
  GEOBI_TRY(conv_bwd(b, t.conv[5], grad.conv[5], g_x1b, g_x1, g_r1, accumulate, s));
  GEOBI_TRY(conv_bwd(b, t.conv[4], grad.conv[4], g_r1, g_up2, nullptr, accumulate, s));
  GEOBI_TRY(segment_sum2(g_up2, 128, t.pool[1].segptr[0], t.pool[1].members[0], t.pool[1].segptr[1], t.pool[1].members[1],
                         N2, g_x2, s));
  GEOBI_TRY(conv_bwd(b, t.conv[3], grad.conv[3], g_x2, g_x2a, nullptr, accumulate, s));
  GEOBI_TRY(conv_bwd(b, t.conv[2], grad.conv[2], g_x2a, g_x2p, nullptr, accumulate, s));
  GEOBI_TRY(pool_bwd(b, t.pool[1], g_x2p, g_x1, s));                  // skip (r_conv2) + pooling2 paths
  GEOBI_TRY(conv_bwd(b, t.conv[1], grad.conv[1], g_x1, g_x1p, nullptr, accumulate, s));
  GEOBI_TRY(pool_bwd(b, t.pool[0], g_x1p, g_x0, s));                  // skip (r_conv4) + pooling1 paths
  GEOBI_TRY(conv_bwd(b, t.conv[0], grad.conv[0], g_x0, dx_in, nullptr, accumulate, s));
  return 0;
}

// Exact arena need of geobi_net_backward for a recorded forward (mirrors its allocations; the level sizes are known
// once the forward has run, so the training forward can refuse an arena the backward would overflow)
size_t branch_backward_bytes(const BranchTape& t) {
  const int64_t N0 = t.L[0].N, N1 = t.L[1].N, N2 = t.L[2].N;
  auto f = [](int64_t n) { return align_up((size_t)(n ? n : 1) * sizeof(float)); };
  size_t b = f(N0 * 32) * 2 + f(N0 * 64) + f(N1 * 64) * 3 + f(N1 * 128) + f(N2 * 128) * 2 + f(N2 * 64) + f(N1 * 32);
  for (int i = 0; i < 8; ++i) {
    const ConvSave& c = t.conv[i];
    // (the first layer may or may not owe an input gradient: without one its workspace is the larger)
    b += align_up(feast_bwd_ws_bytes_for(c.g->N, c.g->E, c.Ca + c.Cb, c.Cb, c.Cout, i != 0));
  }
  for (int l = 0; l < 2; ++l) b += f(t.pool[l].R1 * t.pool[l].C) + f(t.pool[l].P * t.pool[l].C);
  return b;
}
size_t net_backward_bytes(const NetTape& t) {
  auto f = [](int64_t n) { return align_up((size_t)(n ? n : 1) * sizeof(float)); };
  const int64_t V = t.V, F = t.F;
  return f(F * 32) + f(F * 12) + f(F * 9) + f(V * 3) + f(V * 32) + f((F > V ? F : V) * 3) +
         align_up(head_bwd_ws_bytes(F, 32, 1024)) + align_up(head_bwd_ws_bytes(V, 32, 1024)) + branch_backward_bytes(t.v) +
         branch_backward_bytes(t.f) + 4096;
}

int net_forward_impl(const geobi_net_params_t* prm, const geobi_level0_t* gv, const geobi_level0_t* gf, const float* x_v,
                     const float* x_f, const int32_t* fv, const float* depth_direction, const int32_t* pos_rev_v,
                     const int32_t* pos_rev_f, Bump& b, geobi_net_out_t* out, NetTape* tape, hipStream_t s) {
  Level Lv, Lf;
  Lv.N = gv->N; Lv.E = gv->E; Lv.rowptr = gv->rowptr; Lv.col = gv->col; Lv.row = gv->row; Lv.w = gv->weight;
  Lf.N = gf->N; Lf.E = gf->E; Lf.rowptr = gf->rowptr; Lf.col = gf->col; Lf.row = gf->row; Lf.w = gf->weight;
  Lv.pos_rev = pos_rev_v; Lf.pos_rev = pos_rev_f;
  const int64_t V = Lv.N, F = Lf.N;
  const geobi_net_params_t& P = tape ? tape->prm : *prm;       // saved records point into the tape's copy
  float *feat_v = nullptr, *feat_f = nullptr;
  GEOBI_TRY(gnn_forward(b, Lv, x_v, 6, P.gnn_v, P.pool_mean, &feat_v, out->v, tape ? &tape->v : nullptr, s));
  // vertex head: fc_v2(leaky(fc_v1 .)) -> displacement (or depth along depth_direction) + xyz   (network.py:324-332)
  const int nout_v = P.force_depth ? 1 : 3;
  float* verts = b.take<float>((size_t)V * 3);
  float* normals = b.take<float>((size_t)F * 3);               // right behind verts: the binding copies both out at once
  float* raw_v = b.take<float>((size_t)V * nout_v);
  float* xf12 = b.take<float>((size_t)F * 12);
  if (!b.ok) return kArenaFull;
  GEOBI_TRY(head_fwd(feat_v, 32, V, P.fc_v1_w, P.fc_v1_b, 1024, P.fc_v2_w, P.fc_v2_b, nout_v, kLeak, 0,
                     P.force_depth ? depth_direction : nullptr, x_v, 6, nullptr, raw_v, verts, s));
  // geometry coupling: x_f = [x_f | centroid | unit normal] of the predicted geometry   (network.py:335-337)
  GEOBI_TRY(face_geom_fwd(verts, fv, x_f, 6, F, xf12, s));
  GEOBI_TRY(gnn_forward(b, Lf, xf12, 12, P.gnn_f, P.pool_mean, &feat_f, out->f, tape ? &tape->f : nullptr, s));
  float* raw_f = b.take<float>((size_t)F * 3);
  if (!b.ok) return kArenaFull;
  GEOBI_TRY(head_fwd(feat_f, 32, F, P.fc_f1_w, P.fc_f1_b, 1024, P.fc_f2_w, P.fc_f2_b, 3, kLeak, 1, nullptr, nullptr, 0,
                     nullptr, raw_f, normals, s));
  out->verts_off = b.offset_of(verts);
  out->normals_off = b.offset_of(normals);
  out->xf_off = b.offset_of(xf12);
  if (tape) {
    tape->V = V; tape->F = F; tape->x_v = x_v; tape->dd = P.force_depth ? depth_direction : nullptr; tape->fv = fv;
    tape->feat_v = feat_v; tape->feat_f = feat_f; tape->raw_v = raw_v; tape->raw_f = raw_f; tape->verts = verts;
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
           pool_edge_rows_ws_bytes_onepass(N, E) + feast_fwd_ws_bytes(N, 128, 128) + ((size_t)8 << 20) +
           ((size_t)4 << 20);                     // the branch's packed forward weights (< 2 MB)
  };
  return branch(V, Ev) + branch(F, Ef) + (size_t)F * 12 * 4 + ((size_t)16 << 20);
}

extern "C" int geobi_net_forward(const geobi_net_params_t* prm, const geobi_level0_t* gv, const geobi_level0_t* gf,
                                 const float* x_v, const float* x_f, const int32_t* fv, const float* depth_direction,
                                 void* arena, size_t arena_bytes, geobi_net_out_t* out, void* stream) {
  if (!prm || !gv || !gf || !x_v || !x_f || !fv || !arena || !out) return set_error("geobi_net_forward: null argument");
  if (prm->force_depth && !depth_direction) return set_error("geobi_net_forward: force_depth needs depth_direction");
  if (gv->N > GEOBI_MAX_NODES || gf->N > GEOBI_MAX_NODES || gv->E > GEOBI_MAX_EDGES || gf->E > GEOBI_MAX_EDGES || gv->N < 0 ||
      gf->N < 0 || gv->E < 0 || gf->E < 0)
    return set_error("geobi_net_forward: level-0 sizes outside [0, GEOBI_MAX_NODES = %d] nodes / [0, GEOBI_MAX_EDGES = %d] edges (split the mesh into patches)",
                     GEOBI_MAX_NODES, GEOBI_MAX_EDGES);
  Bump b(arena, arena_bytes);
  int rc = net_forward_impl(prm, gv, gf, x_v, x_f, fv, depth_direction, nullptr, nullptr, b, out, nullptr, (hipStream_t)stream);
  out->used_bytes = (int64_t)b.peak;
  if (rc == kArenaFull) set_error("geobi_net_forward: arena too small (%zu bytes needed so far, %zu given)", b.peak, arena_bytes);
  return rc;
}

extern "C" size_t geobi_net_train_arena_bytes(int64_t V, int64_t Ev, int64_t F, int64_t Ef) {
  // forward records + the backward's scratch (one FeaSt backward workspace per layer: bump-only while the side stream
  // may still read it), bounded by the level-0 sizes
  auto branch = [](int64_t N, int64_t E) {
    size_t w = 0;
    const int cin[8] = {12, 32, 64, 128, 128, 128, 64, 64}, cout[8] = {32, 64, 128, 128, 64, 64, 32, 32};
    const int lvl[8] = {0, 1, 2, 2, 1, 1, 0, 0};
    for (int i = 0; i < 8; ++i) {
      const int64_t n = lvl[i] == 0 ? N : (lvl[i] == 1 ? (N * 2) / 5 : N / 8);          // typical level sizes + margin
      const int64_t e = lvl[i] == 0 ? E : (lvl[i] == 1 ? (E * 2) / 5 : E / 8);
      w += feast_bwd_ws_bytes(n, e, cin[i], cout[i]);
    }
    return w + (size_t)N * 4 * 1024;
  };
  return geobi_net_forward_arena_bytes(V, Ev, F, Ef) + branch(V, Ev) + branch(F, Ef) + head_bwd_ws_bytes(V, 32, 1024) +
         head_bwd_ws_bytes(F, 32, 1024) + ((size_t)64 << 20);
}

// Training forward: as geobi_net_forward, but everything the backward needs stays in the arena and a host-side
// record of it is returned as `*handle` (release with geobi_net_release; the arena must outlive the backward).
extern "C" int geobi_net_forward_train(const geobi_net_params_t* prm, const geobi_level0_t* gv, const geobi_level0_t* gf,
                                       const int32_t* pos_rev_v, const int32_t* pos_rev_f, const float* x_v,
                                       const float* x_f, const int32_t* fv, const float* depth_direction, void* arena,
                                       size_t arena_bytes, geobi_net_out_t* out, int64_t* handle, void* stream) {
  if (!prm || !gv || !gf || !pos_rev_v || !pos_rev_f || !x_v || !x_f || !fv || !arena || !out || !handle)
    return set_error("geobi_net_forward_train: null argument");
  if (prm->force_depth && !depth_direction) return set_error("geobi_net_forward_train: force_depth needs depth_direction");
  if (gv->N > GEOBI_MAX_NODES || gf->N > GEOBI_MAX_NODES || gv->E > GEOBI_MAX_EDGES || gf->E > GEOBI_MAX_EDGES || gv->N < 0 ||
      gf->N < 0 || gv->E < 0 || gf->E < 0)
    return set_error("geobi_net_forward_train: level-0 sizes outside [0, GEOBI_MAX_NODES = %d] nodes / [0, GEOBI_MAX_EDGES = %d] edges (split the mesh into patches)",
                     GEOBI_MAX_NODES, GEOBI_MAX_EDGES);
  NetTape* tape = new NetTape();
  tape->prm = *prm;
  tape->arena = (char*)arena; tape->arena_bytes = arena_bytes;
  Bump b(arena, arena_bytes);
  int rc = net_forward_impl(prm, gv, gf, x_v, x_f, fv, depth_direction, pos_rev_v, pos_rev_f, b, out, tape, (hipStream_t)stream);
  out->used_bytes = (int64_t)b.peak;
  if (rc != 0) {
    if (rc == kArenaFull) set_error("geobi_net_forward_train: arena too small (%zu bytes needed so far, %zu given)", b.peak, arena_bytes);
    delete tape;
    *handle = 0;
    return rc;
  }
  tape->fwd_peak = b.off;
  const size_t need = align_up(b.off) + net_backward_bytes(*tape);
  out->used_bytes = (int64_t)need;              // forward + backward: what an arena for this mesh must hold
  if (need > arena_bytes) {
    set_error("geobi_net_forward_train: arena too small for the backward (%zu bytes needed, %zu given)", need, arena_bytes);
    delete tape;
    *handle = 0;
    return kArenaFull;
  }
  *handle = (int64_t)(intptr_t)tape;
  return 0;
}

extern "C" int geobi_net_release(int64_t handle) {
  delete (NetTape*)(intptr_t)handle;
  return 0;
}

namespace geobi {
namespace {

// geobi_net_backward_facet_events: where the NEXT backward of this host thread marks "the facet branch's gradients are final"
thread_local hipEvent_t g_facet_ev_main = nullptr, g_facet_ev_side = nullptr;

// Backward of a recorded forward from arena offset `start` on (geobi_net_backward: right behind the forward's records;
// a mesh group of geobi_net_train_groups: behind its loss buffers as well).
int net_backward_impl(NetTape* t, size_t start, const float* g_verts, const float* g_normals, const geobi_net_params_t* grads,
                      int accumulate, const int32_t* corner_segptr, const int32_t* corner_members, hipStream_t s) {
  void* stream = (void*)s;
  Bump b(t->arena, t->arena_bytes);
  b.off = b.peak = start;
  const geobi_net_params_t& P = t->prm;
  const geobi_net_params_t& G = *grads;
  const int64_t V = t->V, F = t->F;
  const int nout_v = P.force_depth ? 1 : 3;
  float* g_feat_f = b.take<float>((size_t)F * 32);
  float* g_xf12 = b.take<float>((size_t)F * 12);
  float* corner = b.take<float>((size_t)F * 9);
  float* g_vsum = b.take<float>((size_t)V * 3);
  float* g_feat_v = b.take<float>((size_t)V * 32);
  float* zeros = nullptr;
  if (!g_normals || !g_verts) zeros = b.take<float>((size_t)(F > V ? F : V) * 3);
  const size_t hws_f = head_bwd_ws_bytes(F, 32, 1024), hws_v = head_bwd_ws_bytes(V, 32, 1024);
  void* ws_hf = b.take<char>(hws_f);
  void* ws_hv = b.take<char>(hws_v);
  auto fail = [&](int rc) {
    (void)geobi_side_defer(0);
    (void)geobi_side_join(stream);
    side_select(0);
    if (rc == kArenaFull) set_error("geobi_net_backward: arena too small (%zu bytes needed so far, %zu given)", b.peak, t->arena_bytes);
    return rc;
  };
  if (!b.ok) return fail(kArenaFull);
  if (zeros) GEOBI_HIP(hipMemsetAsync(zeros, 0, (size_t)(F > V ? F : V) * 3 * sizeof(float), s));
  // weight-gradient GEMMs of every layer run on the side stream and are joined once, at the end; from ~80 k level-0 nodes
  // on (device-bound batches) on the lowest-priority one
  side_select(V + F >= 80000 || t->side_low);
  (void)geobi_side_defer(1);
  int rc = head_bwd(t->feat_f, 32, F, P.fc_f1_w, P.fc_f1_b, 1024, P.fc_f2_w, 3, kLeak, 1, nullptr, nullptr, t->raw_f,
                    g_normals ? g_normals : zeros, g_feat_f, (float*)G.fc_f1_w, (float*)G.fc_f1_b, (float*)G.fc_f2_w,
                    (float*)G.fc_f2_b, accumulate, ws_hf, hws_f, s);
  if (rc) return fail(rc);
  rc = gnn_backward(b, t->f, G.gnn_f, g_feat_f, g_xf12, accumulate, s);
  if (rc) return fail(rc);
  if (g_facet_ev_main) {
    // every gradient of gnn_f / fc_f1 / fc_f2 is enqueued by now: the main stream's share up to here, the weight-gradient
    // products on the side stream.  A data-parallel caller starts the all-reduce of that half of the bucket behind these
    // two events, i.e. UNDER the vertex branch's backward ("overlap with backward tail").
    hipEvent_t em = g_facet_ev_main, es = g_facet_ev_side;
    g_facet_ev_main = g_facet_ev_side = nullptr;
    if (hipEventRecord(em, s) != hipSuccess) return fail(set_error("geobi_net_backward: facet event"));
    hipStream_t side = side_current();
    if (es && hipEventRecord(es, side ? side : s) != hipSuccess) return fail(set_error("geobi_net_backward: facet event"));
  }
  // geometry coupling: d[x_f | centroid | normal] -> per-corner gradients -> vertices (fixed-order segment sum)
  rc = face_geom_bwd(t->verts, t->fv, g_xf12, F, corner, s);
  if (rc) return fail(rc);
  rc = segment_sum(corner, 3, corner_segptr, corner_members, V, 0, g_vsum, s);
  if (rc) return fail(rc);
  if (g_verts) {
    add_inplace_kernel<<<cdiv(V * 3, 256), 256, 0, s>>>(g_vsum, g_verts, V * 3);
    if (hipGetLastError() != hipSuccess) return fail(set_error("geobi_net_backward: launch failed"));
  }
  rc = head_bwd(t->feat_v, 32, V, P.fc_v1_w, P.fc_v1_b, 1024, P.fc_v2_w, nout_v, kLeak, 0, t->dd, nullptr, t->raw_v, g_vsum,
                g_feat_v, (float*)G.fc_v1_w, (float*)G.fc_v1_b, (float*)G.fc_v2_w, (float*)G.fc_v2_b, accumulate, ws_hv,
                hws_v, s);
  if (rc) return fail(rc);
  rc = gnn_backward(b, t->v, G.gnn_v, g_feat_v, nullptr, accumulate, s);
  if (rc) return fail(rc);
  (void)geobi_side_defer(0);
  rc = geobi_side_join(stream);
  side_select(0);
  return rc;
}

// ---------------------------------------------------------------------------------------- mesh groups in flight together
// The reference walks the meshes of a batch one after another (train_dual.py:199-218: forward, loss / batch_size, backward,
// optimiser step every batch_size meshes).  Meshes are independent units, so the iterations of that loop may run at the same
// time: each GROUP of meshes (one disjoint-union graph) is a complete forward -> loss -> backward pipeline on its own
// stream, driven by its own host thread (= its own library context: side streams, events, scan state, size mailbox), with
// its own arena and its own gradient bucket.  One group's pooling chains (dozens of dependent launches of < 1 workgroup
// per CU, four host reads) then run under the other groups' FeaSt kernels instead of leaving the chip idle.
struct Worker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::atomic<int> posted{0}, done{0};
  std::function<void()> job;
  bool sleeping = false;
  void loop() {
    int seen = 0;
    for (;;) {
      long spins = 0;
      while (posted.load(std::memory_order_acquire) == seen) {
        if (++spins < 200000) { cpu_relax(); continue; }             // a few ms: the next step of a training loop finds it awake
        std::unique_lock<std::mutex> lk(mu);
        sleeping = true;
        cv.wait(lk, [&] { return posted.load(std::memory_order_acquire) != seen; });
        sleeping = false;
      }
      seen = posted.load(std::memory_order_acquire);
      job();
      done.store(seen, std::memory_order_release);
    }
  }
  int post(std::function<void()> j) {
    job = std::move(j);
    const int seq = posted.fetch_add(1, std::memory_order_acq_rel) + 1;
    std::lock_guard<std::mutex> lk(mu);
    if (sleeping) cv.notify_one();
    return seq;
  }
  void wait(int seq) {
    long spins = 0;
    while (done.load(std::memory_order_acquire) != seq)
      if ((++spins & 0xff) == 0) std::this_thread::yield(); else cpu_relax();
  }
};
// workers live as long as the process (never joined: a static destructor would meet threads parked in the wait)
std::mutex g_workers_mu;
std::vector<Worker*>* g_workers = nullptr;
hipEvent_t g_group_start = nullptr;
std::vector<hipEvent_t>* g_group_done = nullptr;

struct SrcList { const float* p[GEOBI_MAX_GROUPS]; };
__global__ void sum_buckets_list_kernel(float* __restrict__ dst, SrcList src, int n_src, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float a = src.p[0][i];
  for (int k = 1; k < n_src; ++k) a += src.p[k][i];        // fixed order: the step is bit-reproducible
  dst[i] = a;
}

struct SideOverride {          // scoped: the weight-gradient products of a group stay on the group's own stream
  explicit SideOverride(int mode) { side_override(mode); }
  ~SideOverride() { side_override(-1); }
};

int run_group(const geobi_net_params_t* prm, geobi_train_group_t* g, int kind_v, int kind_n, int device, hipEvent_t start,
              hipEvent_t done, int n_groups) {
  GEOBI_HIP(hipSetDevice(device));
  // With several groups in flight the other groups ARE the concurrent work: a side stream per group on top of that measured
  // slower (same box, alternating, bench batch as 2 groups: 4.13-4.25 ms per step without, 4.29-4.56 with side streams at
  // the default priority, 5.6-5.9 at the lowest -- the join at the end of a backward then waits for starved products).
  static const int side_in_groups = [] { const char* e = getenv("GEOBI_GROUP_SIDE"); return e ? atoi(e) : 0; }();
  SideOverride scoped(n_groups > 1 && !side_in_groups ? 0 : -1);
  hipStream_t s = (hipStream_t)g->stream;
  if (start) GEOBI_HIP(hipStreamWaitEvent(s, start, 0));
  if (g->grad_flat && g->grad_count > 0) GEOBI_HIP(hipMemsetAsync(g->grad_flat, 0, (size_t)g->grad_count * sizeof(float), s));
  std::unique_ptr<NetTape> tape(new NetTape());
  tape->prm = *prm;
  tape->arena = (char*)g->arena; tape->arena_bytes = g->arena_bytes;
  Bump b(g->arena, g->arena_bytes);
  int rc = net_forward_impl(prm, g->gv, g->gf, g->x_v, g->x_f, g->fv, g->depth_direction, g->pos_rev_v, g->pos_rev_f, b,
                            &g->out, tape.get(), s);
  g->out.used_bytes = (int64_t)b.peak;
  if (rc != 0) {
    if (rc == kArenaFull) set_error("geobi_net_train_groups: arena too small (%zu bytes needed so far, %zu given)", b.peak, g->arena_bytes);
    return rc;
  }
  const int64_t V = tape->V, F = tape->F;
  // loss (network.py:364-389, dual_loss :392-396) and its gradient with respect to the two predictions
  float* g_verts = b.take<float>((size_t)V * 3);
  float* g_normals = b.take<float>((size_t)F * 3);
  const size_t lws_v = row_loss_ws_bytes(V), lws_f = row_loss_ws_bytes(F);
  void* ws_lv = b.take<char>(lws_v);
  void* ws_lf = b.take<char>(lws_f);
  tape->fwd_peak = b.off;
  const size_t need = align_up(b.off) + net_backward_bytes(*tape);
  g->out.used_bytes = (int64_t)need;
  if (!b.ok || need > g->arena_bytes) {
    set_error("geobi_net_train_groups: arena too small (%zu bytes needed, %zu given)", need, g->arena_bytes);
    return kArenaFull;
  }
  const float* verts = (const float*)((const char*)g->arena + g->out.verts_off);
  const float* normals = (const float*)((const char*)g->arena + g->out.normals_off);
  const float sv = g->w_v ? g->scale_v : g->scale_v / (float)V, sn = g->w_f ? g->scale_n : g->scale_n / (float)F;
  GEOBI_TRY(row_loss_fwd(verts, g->y_v, g->w_v, V, kind_v, sv, g->losses, ws_lv, lws_v, s));
  GEOBI_TRY(row_loss_fwd(normals, g->y_f, g->w_f, F, kind_n, sn, g->losses + 1, ws_lf, lws_f, s));
  GEOBI_TRY(row_loss_bwd(verts, g->y_v, g->w_v, nullptr, V, kind_v, sv, g_verts, s));
  GEOBI_TRY(row_loss_bwd(normals, g->y_f, g->w_f, nullptr, F, kind_n, sn, g_normals, s));
  GEOBI_TRY(net_backward_impl(tape.get(), tape->fwd_peak, g_verts, g_normals, &g->grads, 1, g->corner_segptr,
                              g->corner_members, s));
  if (done) GEOBI_HIP(hipEventRecord(done, s));
  return 0;
}

}  // namespace

}  // namespace geobi

using namespace geobi;

// Backward of a recorded forward.  g_verts [V,3] / g_normals [F,3] (either may be NULL = zero); `grads` mirrors the
// parameter struct (same order) and receives (accumulate != 0: is added) the parameter gradients.  corner_segptr /
// corner_members: vertex -> corner inverse lists of the face table (the geometry coupling's gradient is summed through
// them in a fixed order).  Scratch is taken from the rest of the forward's arena.
extern "C" int geobi_net_backward(int64_t handle, const float* g_verts, const float* g_normals,
                                  const geobi_net_params_t* grads, int accumulate, const int32_t* corner_segptr,
                                  const int32_t* corner_members, void* stream) {
  NetTape* t = (NetTape*)(intptr_t)handle;
  if (!t || !grads || !corner_segptr || !corner_members) return set_error("geobi_net_backward: null argument");
  return net_backward_impl(t, t->fwd_peak, g_verts, g_normals, grads, accumulate, corner_segptr, corner_members,
                           (hipStream_t)stream);
}

extern "C" int geobi_net_backward_facet_events(void* ev_main, void* ev_side) {
  g_facet_ev_main = (hipEvent_t)ev_main;
  g_facet_ev_side = (hipEvent_t)ev_side;
  return 0;
}

extern "C" size_t geobi_abi_sizeof(int which) {
  switch (which) {
    case 0: return sizeof(geobi_net_params_t);
    case 1: return sizeof(geobi_level0_t);
    case 2: return sizeof(geobi_net_out_t);
    case 3: return sizeof(geobi_train_group_t);
    case 4: return sizeof(geobi_copy_seg_t);
    default: return 0;
  }
}

extern "C" int geobi_net_spin_cap_hits(void) { return (int)g_spin_cap_hits.load(std::memory_order_relaxed); }

// n_groups complete training pipelines (forward, loss, backward) in flight together, one host thread and one stream each;
// see the block comment above run_group.  Returns once every group has ENQUEUED all of its work and `main_stream` has
// been made to wait for all of it (and, with sum_into, has the buckets' fixed-order sum enqueued behind that).
extern "C" int geobi_net_train_groups(const geobi_net_params_t* prm, geobi_train_group_t* groups, int n_groups,
                                      int loss_kind_v, int loss_kind_n, float* sum_into, int64_t sum_count,
                                      void* main_stream) {
  if (!prm || !groups) return set_error("geobi_net_train_groups: null argument");
  if (n_groups < 1 || n_groups > GEOBI_MAX_GROUPS) return set_error("geobi_net_train_groups: 1..%d groups", GEOBI_MAX_GROUPS);
  if ((loss_kind_v != 0 && loss_kind_v != 1) || (loss_kind_n != 0 && loss_kind_n != 1))
    return set_error("geobi_net_train_groups: loss kinds are 0 (L1) or 1 (L2)");
  if (prm->force_depth) for (int k = 0; k < n_groups; ++k)
    if (!groups[k].depth_direction) return set_error("geobi_net_train_groups: force_depth needs depth_direction");
  for (int k = 0; k < n_groups; ++k) {
    geobi_train_group_t& g = groups[k];
    if (!g.gv || !g.gf || !g.pos_rev_v || !g.pos_rev_f || !g.x_v || !g.x_f || !g.fv || !g.y_v || !g.y_f ||
        !g.corner_segptr || !g.corner_members || !g.arena || !g.losses || !g.stream)
      return set_error("geobi_net_train_groups: null field in group %d (a group needs its OWN stream, not the null stream)", k);
    if (g.gv->N > GEOBI_MAX_NODES || g.gf->N > GEOBI_MAX_NODES || g.gv->E > GEOBI_MAX_EDGES || g.gf->E > GEOBI_MAX_EDGES ||
        g.gv->N < 0 || g.gf->N < 0 || g.gv->E < 0 || g.gf->E < 0)
      return set_error("geobi_net_train_groups: group %d: level-0 sizes outside GEOBI_MAX_NODES / GEOBI_MAX_EDGES", k);
    if (sum_into && (!g.grad_flat || g.grad_count != sum_count))
      return set_error("geobi_net_train_groups: sum_into needs every group's grad_flat with %lld floats", (long long)sum_count);
    for (int j = 0; j < k; ++j)
      if (groups[j].stream == g.stream || groups[j].arena == g.arena || (g.grad_flat && groups[j].grad_flat == g.grad_flat))
        return set_error("geobi_net_train_groups: groups %d and %d share a stream, an arena or a gradient bucket", j, k);
    g.rc = 0; g.error[0] = 0;
  }
  int device = 0;
  GEOBI_HIP(hipGetDevice(&device));
  hipStream_t ms = (hipStream_t)main_stream;
  // one call at a time drives the workers (they are a process-wide resource)
  std::lock_guard<std::mutex> call_lock(g_workers_mu);
  if (g_workers == nullptr) { g_workers = new std::vector<Worker*>(); g_group_done = new std::vector<hipEvent_t>(); }
  while ((int)g_workers->size() < n_groups - 1) {
    Worker* w = new Worker();
    w->th = std::thread([w] { w->loop(); });
    w->th.detach();
    g_workers->push_back(w);
  }
  while ((int)g_group_done->size() < n_groups) {
    hipEvent_t e;
    GEOBI_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    g_group_done->push_back(e);
  }
  if (g_group_start == nullptr) GEOBI_HIP(hipEventCreateWithFlags(&g_group_start, hipEventDisableTiming));
  // the groups read what `main_stream` has produced so far (parameters of the last optimiser step) and must not zero
  // their buckets before the previous step's sum has read them
  GEOBI_HIP(hipEventRecord(g_group_start, ms));
  static const int skew_us = [] { const char* e = getenv("GEOBI_GROUP_SKEW_US"); return e ? atoi(e) : 0; }();
  const auto call_t0 = std::chrono::steady_clock::now();
  auto body = [&](int k) {
    geobi_train_group_t& g = groups[k];
    if (skew_us > 0 && k > 0) {           // experiment: group k starts k * skew later, so that the groups' phases interleave
      const auto until = call_t0 + std::chrono::microseconds((long)skew_us * k);
      while (std::chrono::steady_clock::now() < until) cpu_relax();
    }
    g.rc = run_group(prm, &g, loss_kind_v, loss_kind_n, device, g_group_start, (*g_group_done)[k], n_groups);
    if (g.rc != 0) { strncpy(g.error, geobi_last_error(), sizeof(g.error) - 1); g.error[sizeof(g.error) - 1] = 0; }
  };
  int seq[GEOBI_MAX_GROUPS];
  for (int k = 1; k < n_groups; ++k) seq[k] = (*g_workers)[k - 1]->post([&body, k] { body(k); });
  body(0);
  for (int k = 1; k < n_groups; ++k) (*g_workers)[k - 1]->wait(seq[k]);
  int bad = -1;
  for (int k = 0; k < n_groups; ++k) {
    if (groups[k].rc == 0) GEOBI_HIP(hipStreamWaitEvent(ms, (*g_group_done)[k], 0));
    else if (bad < 0) bad = k;
  }
  if (bad >= 0) {
    // a failed group may have left work in flight: the caller's stream waits for all of it before anything is reused
    for (int k = 0; k < n_groups; ++k)
      if (groups[k].rc != 0) {
        hipEvent_t e = (*g_group_done)[k];
        if (hipEventRecord(e, (hipStream_t)groups[k].stream) == hipSuccess) (void)hipStreamWaitEvent(ms, e, 0);
      }
    set_error("geobi_net_train_groups: group %d: %s", bad, groups[bad].error);
    return groups[bad].rc;
  }
  if (sum_into) {
    SrcList src;
    for (int k = 0; k < n_groups; ++k) src.p[k] = groups[k].grad_flat;
    sum_buckets_list_kernel<<<cdiv(sum_count, 256), 256, 0, ms>>>(sum_into, src, n_groups, sum_count);
    GEOBI_LAUNCH_OK();
  }
  return 0;
}
