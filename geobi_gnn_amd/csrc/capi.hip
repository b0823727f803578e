// extern "C" surface of libgeobi_hip.so (declared in include/geobi_hip.h): argument checks,
// error reporting and the optional per-kernel event timing used by bench.py.
#include "../../include/geobi_hip.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include <mutex>
#include <vector>

#include "common.h"

namespace geobi {

static thread_local char g_err[512] = "";

int set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return 1;
}

// ------------------------------------------------------------------------- profiling
struct ProfRec {
  hipEvent_t a, b;
  double bytes;
  int tag;
  hipStream_t stream = nullptr;
  bool closed = false;
  bool bound = false;      // events bound to a kernel's dispatch by the extended launch (no hipEventRecord)
};
static std::atomic<int> g_prof_kernel{PROF_NONE};
static std::mutex g_prof_mu;             // the record list is shared by the host threads of a concurrent step
static std::vector<ProfRec> g_prof;
static std::vector<hipEvent_t> g_event_pool;

static hipEvent_t get_event() {
  if (!g_event_pool.empty()) {
    hipEvent_t e = g_event_pool.back();
    g_event_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

void prof_begin(int kernel, hipStream_t s, double alg_bytes, int tag) {
  if (kernel != g_prof_kernel.load(std::memory_order_relaxed)) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfRec r;
  r.a = get_event();
  r.b = get_event();
  r.bytes = alg_bytes;
  r.tag = tag;
  if (!r.a || !r.b) return;
  r.stream = s;
  (void)hipEventRecord(r.a, s);
  g_prof.push_back(r);
}

static thread_local hipEvent_t g_launch_ev[2] = {nullptr, nullptr};      // handed from prof_begin_launch to the launch site

void prof_begin_launch(int kernel, hipStream_t s, double alg_bytes, int tag) {
  if (kernel != g_prof_kernel.load(std::memory_order_relaxed)) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfRec r;
  r.a = get_event();
  r.b = get_event();
  r.bytes = alg_bytes;
  r.tag = tag;
  if (!r.a || !r.b) return;
  r.stream = s;
  g_launch_ev[0] = r.a;
  g_launch_ev[1] = r.b;
  r.bound = true;
  g_prof.push_back(r);
}

bool prof_take_launch_events(hipEvent_t* start, hipEvent_t* stop) {
  if (g_launch_ev[0] == nullptr) return false;
  *start = g_launch_ev[0];
  *stop = g_launch_ev[1];
  g_launch_ev[0] = g_launch_ev[1] = nullptr;
  return true;
}

void prof_end(int kernel, hipStream_t s) {
  if (kernel != g_prof_kernel.load(std::memory_order_relaxed)) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (g_launch_ev[0] != nullptr) {
    // prof_begin_launch's events were not taken (a launch path without the extended launch): bracket nothing, drop the record
    g_launch_ev[0] = g_launch_ev[1] = nullptr;
    for (size_t i = g_prof.size(); i-- > 0;)
      if (g_prof[i].stream == s && !g_prof[i].closed) { g_prof.erase(g_prof.begin() + (long)i); break; }
    return;
  }
  // the most recent open record of THIS stream (another thread's record may sit behind it)
  for (size_t i = g_prof.size(); i-- > 0;) {
    if (g_prof[i].stream == s && !g_prof[i].closed) {
      g_prof[i].closed = true;
      if (!g_prof[i].bound) (void)hipEventRecord(g_prof[i].b, s);
      return;
    }
  }
}

// ------------------------------------------------------------------------- side stream
// Context = host thread.  Everything below the overlap switch is per host thread: a thread drives one stream at a time
// (the caller's stream of a per-op call; the stream of ONE mesh group in geobi_net_train_groups), so every context has
// its own side streams and fork / join events and two groups in flight never meet on a shared event.
static std::atomic<int> g_overlap{[] { const char* e = getenv("GEOBI_OVERLAP"); return (e && atoi(e) == 0) ? 0 : 1; }()};
static thread_local int g_overlap_here = -1;   // this context's override of g_overlap (-1: none); side_override()
static thread_local int g_defer = 0;     // 1: backward calls fork but do not join; the caller joins once (geobi_side_join)
// Two side streams: [0] at the default priority, [1] at the lowest.  On a big batch the weight-gradient products
// compete with the backward's own kernels for workgroup slots: at the lowest priority they fill what the main stream
// leaves free (4.37-4.43 against 4.48-4.57 ms per step on the bench batch).  On small batches -- the host-bound regime,
// the device has idle gaps anyway -- the low-priority queue is served late and the join at the end waits for it (n = 16:
// 2.85-2.90 against 2.77-2.79 ms).  side_select() picks per backward; per-op callers get the default one.
static thread_local hipStream_t g_sides[2] = {nullptr, nullptr};
static thread_local int g_side_sel = 0;
static thread_local hipStream_t g_side = nullptr;           // the stream the current / last fork used
static thread_local hipEvent_t g_fork_ev = nullptr, g_join_ev = nullptr;

void side_select(int low_priority) {
  static const bool allow = [] { const char* e = getenv("GEOBI_SIDE_PRIORITY"); return !e || atoi(e) != 0; }();
  g_side_sel = (low_priority && allow) ? 1 : 0;
}

void side_override(int mode) { g_overlap_here = mode; }
hipStream_t side_current() { return g_side; }

Fork fork_side_stream(hipStream_t main) {
  Fork f;
  if (g_overlap_here == 0 || (g_overlap_here < 0 && !g_overlap)) return f;
  if (g_sides[g_side_sel] == nullptr) {
    int lo = 0, hi = 0;
    if (g_side_sel == 0 || hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) lo = 0;
    const hipError_t ce = hipStreamCreateWithPriority(&g_sides[g_side_sel], hipStreamNonBlocking, lo);
    if (ce != hipSuccess) {
      g_sides[g_side_sel] = nullptr;
      return f;
    }
    if (g_fork_ev == nullptr &&
        (hipEventCreateWithFlags(&g_fork_ev, hipEventDisableTiming) != hipSuccess ||
         hipEventCreateWithFlags(&g_join_ev, hipEventDisableTiming) != hipSuccess)) { g_overlap = 0; return f; }
  }
  g_side = g_sides[g_side_sel];
  if (hipEventRecord(g_fork_ev, main) != hipSuccess) return f;
  if (hipStreamWaitEvent(g_side, g_fork_ev, 0) != hipSuccess) return f;
  f.side = g_side;
  f.join = g_join_ev;
  return f;
}

int side_wait_main(const Fork& f, hipStream_t main) {
  if (f.side == nullptr) return 0;
  GEOBI_HIP(hipEventRecord(g_fork_ev, main));
  GEOBI_HIP(hipStreamWaitEvent(f.side, g_fork_ev, 0));
  return 0;
}

int join_side_stream(const Fork& f, hipStream_t main) {
  if (f.side == nullptr || g_defer) return 0;
  GEOBI_HIP(hipEventRecord(f.join, f.side));
  GEOBI_HIP(hipStreamWaitEvent(main, f.join, 0));
  return 0;
}

}  // namespace geobi

using namespace geobi;

#define S(stream) ((hipStream_t)(stream))
#define NOTNULL(p)                                                       \
  do {                                                                   \
    if ((p) == nullptr) return set_error("%s: %s is NULL", __func__, #p); \
  } while (0)

// Size limits of the header (GEOBI_MAX_NODES / GEOBI_MAX_EDGES): every entry point that takes a node or edge count
// rejects what is above them, so that no kernel ever forms a 32-bit element index beyond its range.
static int sizes_ok(const char* fn, int64_t nodes, int64_t edges) {
  if (nodes < 0 || edges < 0) return set_error("%s: negative size (%lld nodes, %lld edges)", fn, (long long)nodes, (long long)edges);
  if (nodes > GEOBI_MAX_NODES)
    return set_error("%s: %lld nodes exceed GEOBI_MAX_NODES = %d per call (split the mesh into patches)", fn, (long long)nodes, GEOBI_MAX_NODES);
  if (edges > GEOBI_MAX_EDGES)
    return set_error("%s: %lld edges exceed GEOBI_MAX_EDGES = %d per call (split the mesh into patches)", fn, (long long)edges, GEOBI_MAX_EDGES);
  return 0;
}
#define SIZES(nodes, edges) GEOBI_TRY(sizes_ok(__func__, (int64_t)(nodes), (int64_t)(edges)))

extern "C" {

int geobi_version(void) { return 100; }
const char* geobi_last_error(void) { return g_err; }

size_t geobi_csr_ws_bytes(int64_t E, int64_t N) { return csr_ws_bytes(E, N); }

int geobi_csr_from_coo(const int64_t* seg, const int64_t* nbr, int64_t E, int64_t N, int drop_self,
                       int32_t* rowptr, int32_t* col, int32_t* eid, int32_t* bad, void* ws, size_t ws_bytes,
                       void* stream) {
  SIZES(N, E);
  NOTNULL(rowptr);
  if (E > 0) { NOTNULL(seg); NOTNULL(nbr); NOTNULL(col); NOTNULL(eid); }
  return csr_from_coo(seg, nbr, E, N, drop_self, rowptr, col, eid, bad, ws, ws_bytes, S(stream));
}

int geobi_csr_transpose(const int32_t* rowptr, const int32_t* col, int64_t N, int64_t Ecap, int32_t* rowptr_t,
                        int32_t* col_t, int32_t* pos_t, int32_t* inv_pos, void* ws, size_t ws_bytes, void* stream) {
  SIZES(N, Ecap);
  NOTNULL(rowptr); NOTNULL(rowptr_t);
  if (Ecap > 0) { NOTNULL(col); NOTNULL(col_t); NOTNULL(pos_t); }
  return csr_transpose(rowptr, col, N, Ecap, rowptr_t, col_t, pos_t, inv_pos, ws, ws_bytes, S(stream));
}

int geobi_csr_reverse_index(const int32_t* rowptr, const int32_t* row, const int32_t* col, int64_t E,
                            int32_t* pos_rev, int32_t* flag, void* stream) {
  SIZES(0, E);
  NOTNULL(rowptr);                         // flag may be NULL: the caller knows the graph is symmetric
  if (E > 0) { NOTNULL(row); NOTNULL(col); NOTNULL(pos_rev); }
  return csr_reverse_index(rowptr, row, col, E, pos_rev, flag, S(stream));
}
int geobi_expand_rowptr(const int32_t* rowptr, int64_t N, int32_t* row, void* stream) {
  SIZES(N, 0);
  return expand_rowptr(rowptr, N, row, S(stream));
}
int geobi_concat32(const geobi_copy_seg_t* segs, int n_segs, int is_float, void* stream) {
  if (n_segs <= 0) return 0;
  NOTNULL(segs);
  static_assert(sizeof(geobi_copy_seg_t) == sizeof(CopySeg) && offsetof(geobi_copy_seg_t, value) == offsetof(CopySeg, value),
                "the internal and the public segment struct are one layout");
  return concat32(reinterpret_cast<const CopySeg*>(segs), n_segs, is_float, S(stream));
}
int geobi_gather_f32(const float* src, const int32_t* idx, int64_t n, float* dst, void* stream) {
  SIZES(0, n);
  return gather_f32(src, idx, n, dst, S(stream));
}

int geobi_debug_exp_le0(const float* x, float* y, int64_t n, void* stream) {
  NOTNULL(x); NOTNULL(y);
  return exp_le0_probe(x, y, n, S(stream));
}

size_t geobi_feast_wpack_floats(int Cin, int Cout) { return feast_wpack_floats(Cin, Cout); }

int geobi_feast_ldz(int Cin) { return feast_ldz(Cin); }
size_t geobi_feast_fwd_ws_bytes(int64_t N, int Cin, int Cout) { return feast_fwd_ws_bytes(N, Cin, Cout); }

static int check_channels(const char* fn, int Cin, int Cout) {
  bool in_ok = Cin == 6 || Cin == 12 || Cin == 32 || Cin == 64 || Cin == 128;
  bool out_ok = Cout == 32 || Cout == 64 || Cout == 128;
  if (!in_ok || !out_ok) return set_error("%s: unsupported channels Cin=%d Cout=%d", fn, Cin, Cout);
  return 0;
}

int geobi_feast_fwd(const float* xa, const float* xb, int Ca, int Cb, int64_t N, int64_t E,
                    const int32_t* rowptr_in, const int32_t* col_in, const float* lin_w, const float* u_w,
                    const float* c, const float* bias, int Cout, float slope, float* out, float* p, float* z,
                    float* wf, void* ws, size_t ws_bytes, void* stream) {
  SIZES(N, E);
  NOTNULL(xa); NOTNULL(rowptr_in); NOTNULL(lin_w); NOTNULL(u_w); NOTNULL(c); NOTNULL(bias);
  NOTNULL(out); NOTNULL(p);
  if (Cb > 0) NOTNULL(xb);
  if (E > 0) NOTNULL(col_in);
  GEOBI_TRY(check_channels(__func__, Ca + Cb, Cout));
  return feast_fwd(xa, Cb > 0 ? xb : nullptr, Ca, Cb, N, E, rowptr_in, col_in, lin_w, u_w, c, bias, Cout, slope, out,
                   p, z, wf, ws, ws_bytes, S(stream));
}

size_t geobi_feast_bwd_ws_bytes(int64_t N, int64_t E, int Cin, int Cout) { return feast_bwd_ws_bytes(N, E, Cin, Cout); }

int geobi_feast_bwd(const float* xa, const float* xb, int Ca, int Cb, int64_t N, int64_t E,
                    const int32_t* rowptr_in, const int32_t* col_in, const int32_t* rowptr_out,
                    const int32_t* col_out, const int32_t* pos_in, const float* lin_w, const float* u_w,
                    const float* c, int Cout, float slope, const float* out, const float* gout, const float* p,
                    const float* z, const float* wf, float* dxa, float* dxb, float* dlin_w, float* du_w, float* dc,
                    float* dbias, int accumulate, void* ws, size_t ws_bytes, void* stream) {
  SIZES(N, E);
  NOTNULL(xa); NOTNULL(rowptr_in); NOTNULL(rowptr_out); NOTNULL(lin_w); NOTNULL(u_w); NOTNULL(c);
  NOTNULL(gout); NOTNULL(p); NOTNULL(dlin_w); NOTNULL(du_w); NOTNULL(dc); NOTNULL(dbias);
  if (slope != 1.0f) NOTNULL(out);
  if (Cb > 0) { NOTNULL(xb); if (dxa) NOTNULL(dxb); }
  if (E > 0) { NOTNULL(col_in); NOTNULL(col_out); NOTNULL(pos_in); }
  GEOBI_TRY(check_channels(__func__, Ca + Cb, Cout));
  return feast_bwd(xa, Cb > 0 ? xb : nullptr, Ca, Cb, N, E, rowptr_in, col_in, rowptr_out, col_out, pos_in, lin_w,
                   u_w, c, Cout, slope, out, gout, p, z, wf, dxa, dxb, dlin_w, du_w, dc, dbias, accumulate, ws, ws_bytes,
                   S(stream));
}

int geobi_edge_weight_t10(const float* x, int C, const int32_t* row, const int32_t* col, const float* w_in,
                          int64_t E, float* w_out, void* stream) {
  SIZES(0, E);
  if (E > 0) { NOTNULL(x); NOTNULL(row); NOTNULL(col); NOTNULL(w_out); }
  return edge_weight_t10(x, C, row, col, w_in, E, w_out, S(stream));
}

int geobi_edge_weight_att(const float* x, int C, const float* att_l, const float* att_r, const int32_t* row,
                          const int32_t* col, const float* w_in, int64_t N, int64_t E, float* node_ws, float* w_out,
                          void* stream) {
  SIZES(N, E);
  if (E > 0 && N > 0) { NOTNULL(x); NOTNULL(att_l); NOTNULL(att_r); NOTNULL(row); NOTNULL(col); NOTNULL(node_ws); NOTNULL(w_out); }
  return edge_weight_att(x, C, att_l, att_r, row, col, w_in, N, E, node_ws, w_out, S(stream));
}

size_t geobi_match_ws_bytes(int64_t N) { return match_ws_bytes(N); }
int geobi_match_heavy_edge(const int32_t* rowptr, const int32_t* col, const float* w, int64_t N, int rounds,
                           int init, int32_t* cluster, int32_t* cluster_final, int32_t* status, void* ws, size_t ws_bytes, void* stream) {
  SIZES(N, 0);
  NOTNULL(rowptr); NOTNULL(cluster); NOTNULL(status);
  return match_heavy_edge(rowptr, col, w, N, rounds, init, cluster, cluster_final, status, ws, ws_bytes, S(stream));
}

int geobi_set_match_round_cap(int cap) { set_match_round_cap(cap); return 0; }
int geobi_set_tile_rows(int rows) { return set_tile_rows(rows); }
int geobi_set_head_precision(int mode) { return set_head_precision(mode); }
int geobi_set_rowpass_form(int staged, int chunked64) { return set_rowpass_form(staged, chunked64); }
int geobi_set_column_parts(int parts) { return set_column_parts(parts); }

size_t geobi_match_coarsen_ws_bytes(int64_t N) { return match_coarsen_ws_bytes(N); }
int geobi_match_coarsen(const int32_t* rowptr, const int32_t* col, const float* w, int64_t N, int rounds, int init,
                        int32_t* state, int32_t* cluster_final, int32_t* cnew, int32_t* segptr, int32_t* members,
                        int32_t* counters, void* ws, size_t ws_bytes, void* stream) {
  SIZES(N, 0);
  NOTNULL(rowptr); NOTNULL(state); NOTNULL(cluster_final); NOTNULL(cnew); NOTNULL(segptr);     // col: NULL when E = 0
  NOTNULL(members); NOTNULL(counters); NOTNULL(ws);
  return match_coarsen(rowptr, col, w, N, rounds, init, state, cluster_final, cnew, segptr, members, counters, ws,
                       ws_bytes, S(stream));
}

size_t geobi_relabel_ws_bytes(int64_t N) { return relabel_ws_bytes(N); }
int geobi_relabel_compact(const int32_t* cluster, int64_t N, int rep_is_self, int32_t* cnew, int32_t* count,
                          void* ws, size_t ws_bytes, void* stream) {
  SIZES(N, 0);
  NOTNULL(cluster); NOTNULL(cnew); NOTNULL(count);
  return relabel_compact(cluster, N, rep_is_self, cnew, count, ws, ws_bytes, S(stream));
}

size_t geobi_segment_csr_ws_bytes(int64_t n) { return segment_csr_ws_bytes(n); }
int geobi_segment_csr(const int32_t* seg, int64_t n, int64_t nseg, int32_t* segptr, int32_t* members, void* ws,
                      size_t ws_bytes, void* stream) {
  SIZES(nseg, n);          // n counts list entries (3 F corners of a face table): an edge-like count
  NOTNULL(segptr);
  return segment_csr(seg, n, nseg, segptr, members, ws, ws_bytes, S(stream));
}
size_t geobi_segment_pairs_ws_bytes(int64_t nseg) { return segment_pairs_ws_bytes(nseg); }
int geobi_segment_csr_pairs(const int32_t* cnew, const int32_t* raw, int64_t N, int64_t nseg, int32_t* segptr,
                            int32_t* members, void* ws, size_t ws_bytes, void* stream) {
  SIZES(N, 0);
  NOTNULL(cnew); NOTNULL(raw); NOTNULL(segptr); NOTNULL(members);
  return segment_csr_pairs(cnew, raw, N, nseg, segptr, members, ws, ws_bytes, S(stream));
}
int geobi_segment_csr_compose(const int32_t* segptr1, const int32_t* members1, const int32_t* segptr2,
                              const int32_t* members2, int64_t nseg2, int64_t n_fine, int32_t* segptr12,
                              int32_t* members12, void* ws, size_t ws_bytes, void* stream) {
  SIZES(n_fine, 0);
  NOTNULL(segptr1); NOTNULL(members1); NOTNULL(segptr2); NOTNULL(members2); NOTNULL(segptr12); NOTNULL(members12);
  return segment_csr_compose(segptr1, members1, segptr2, members2, nseg2, n_fine, segptr12, members12, ws, ws_bytes,
                             S(stream));
}
int geobi_segment_max_fwd(const float* x, int C, const int32_t* segptr, const int32_t* members, int64_t nseg,
                          float* out, int32_t* arg, void* stream) {
  SIZES(nseg, 0);
  return segment_max_fwd(x, C, segptr, members, nseg, out, arg, S(stream));
}
int geobi_segment_max_bwd(const float* gout, const int32_t* arg, const int32_t* seg, int C, int64_t nseg,
                          int64_t n_fine, float* gx, void* stream) {
  SIZES(n_fine, 0);
  return segment_max_bwd(gout, arg, seg, C, nseg, n_fine, gx, S(stream));
}
int geobi_segment_sum(const float* x, int C, const int32_t* segptr, const int32_t* members, int64_t nseg, int mean,
                      float* out, void* stream) {
  SIZES(nseg, 0);
  return segment_sum(x, C, segptr, members, nseg, mean, out, S(stream));
}
int geobi_segment_mean_bwd(const float* gout, const int32_t* seg, const int32_t* segptr, int C, int64_t n_fine,
                           float* gx, void* stream) {
  SIZES(n_fine, 0);
  return segment_mean_bwd(gout, seg, segptr, C, n_fine, gx, S(stream));
}
int geobi_gather_rows(const float* x, const int32_t* idx, int C, int64_t n_out, float* out, void* stream) {
  SIZES(n_out, 0);
  return gather_rows(x, idx, C, n_out, out, S(stream));
}
size_t geobi_pool_edge_rows_ws_bytes(int64_t nbound) { return pool_edge_rows_ws_bytes(nbound); }
int geobi_pool_edge_rows(const int32_t* cnew, const int32_t* segptr, const int32_t* members, const int32_t* rowptr,
                         const int32_t* col, const float* w, const int32_t* ncount, int64_t nbound,
                         int32_t* rowptr_c, int32_t* row_c, int32_t* col_c, float* w_c, int32_t* count,
                         int32_t* overflow, void* ws, size_t ws_bytes, void* stream) {
  SIZES(nbound, 0);
  NOTNULL(cnew); NOTNULL(segptr); NOTNULL(members); NOTNULL(rowptr); NOTNULL(ncount); NOTNULL(rowptr_c);
  NOTNULL(row_c); NOTNULL(col_c); NOTNULL(count); NOTNULL(overflow);
  return pool_edge_rows(cnew, segptr, members, rowptr, col, w, ncount, nbound, rowptr_c, row_c, col_c, w_c, count,
                        overflow, ws, ws_bytes, S(stream));
}
size_t geobi_pool_edge_ws_bytes(int64_t E) { return pool_edge_ws_bytes(E); }
int geobi_pool_edge(const int32_t* cnew, const int32_t* row, const int32_t* col, const float* w, int64_t E,
                    int64_t nmax, int32_t* rowptr_c, int32_t* row_c, int32_t* col_c, float* w_c, int32_t* count,
                    void* ws, size_t ws_bytes, void* stream) {
  SIZES(nmax, E);
  NOTNULL(cnew); NOTNULL(rowptr_c); NOTNULL(count);
  return pool_edge(cnew, row, col, w, E, nmax, rowptr_c, row_c, col_c, w_c, count, ws, ws_bytes, S(stream));
}

int geobi_face_geom_fwd(const float* verts, const int32_t* fv, const float* xf, int ldxf, int64_t F, float* out,
                        void* stream) {
  SIZES(F, 0);
  return face_geom_fwd(verts, fv, xf, ldxf, F, out, S(stream));
}
int geobi_face_geom_bwd(const float* verts, const int32_t* fv, const float* gout, int64_t F, float* corner_grad,
                        void* stream) {
  SIZES(F, 0);
  return face_geom_bwd(verts, fv, gout, F, corner_grad, S(stream));
}

int geobi_head_fwd(const float* x, int Cin, int64_t N, const float* w1, const float* b1, int K, const float* w2,
                   const float* b2, int nout, float slope, int mode, const float* dd, const float* resid,
                   int ld_resid, float* h, float* raw, float* out, void* stream) {
  SIZES(N, 0);
  NOTNULL(x); NOTNULL(w1); NOTNULL(b1); NOTNULL(w2); NOTNULL(b2); NOTNULL(raw); NOTNULL(out);
  if (mode == 0) NOTNULL(resid);
  return head_fwd(x, Cin, N, w1, b1, K, w2, b2, nout, slope, mode, dd, resid, ld_resid, h, raw, out, S(stream));
}
size_t geobi_head_bwd_ws_bytes(int64_t N, int Cin, int K) { return head_bwd_ws_bytes(N, Cin, K); }
int geobi_head_bwd(const float* x, int Cin, int64_t N, const float* w1, const float* b1, int K, const float* w2,
                   int nout, float slope, int mode, const float* dd, const float* h, const float* raw, const float* gout,
                   float* dx, float* dw1, float* db1, float* dw2, float* db2, int accumulate, void* ws,
                   size_t ws_bytes, void* stream) {
  SIZES(N, 0);
  NOTNULL(x); NOTNULL(w1); NOTNULL(w2); NOTNULL(raw); NOTNULL(gout);
  NOTNULL(dw1); NOTNULL(db1); NOTNULL(dw2); NOTNULL(db2);
  return head_bwd(x, Cin, N, w1, b1, K, w2, nout, slope, mode, dd, h, raw, gout, dx, dw1, db1, dw2, db2, accumulate, ws,
                  ws_bytes, S(stream));
}

size_t geobi_row_loss_ws_bytes(int64_t n) { return row_loss_ws_bytes(n); }
int geobi_row_loss_fwd(const float* a, const float* b, const float* w, int64_t n, int kind, float scale, float* out,
                       void* ws, size_t ws_bytes, void* stream) {
  SIZES(n, 0);
  NOTNULL(a); NOTNULL(b); NOTNULL(out);
  return row_loss_fwd(a, b, w, n, kind, scale, out, ws, ws_bytes, S(stream));
}
int geobi_row_loss_bwd(const float* a, const float* b, const float* w, const float* gout, int64_t n, int kind,
                       float scale, float* ga, void* stream) {
  SIZES(n, 0);
  NOTNULL(a); NOTNULL(b); NOTNULL(gout); NOTNULL(ga);
  return row_loss_bwd(a, b, w, gout, n, kind, scale, ga, S(stream));
}

int geobi_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                    float weight_decay, float bias_corr1, float bias_corr2, void* stream) {
  NOTNULL(p); NOTNULL(g); NOTNULL(m); NOTNULL(v);
  return adam_flat(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, bias_corr1, bias_corr2, S(stream));
}

size_t geobi_update_position_ws_bytes(int64_t V, int64_t F) { return update_position_ws_bytes(V, F); }
int geobi_update_position2(const float* points, const int32_t* fv, const int32_t* vf, int maxval,
                           const float* normals, const float* dd, int64_t V, int64_t F, int n_iter, float* out,
                           void* ws, size_t ws_bytes, void* stream) {
  SIZES(V > F ? V : F, 0);
  NOTNULL(points); NOTNULL(fv); NOTNULL(vf); NOTNULL(normals); NOTNULL(out);
  return update_position2(points, fv, vf, maxval, normals, dd, V, F, n_iter, out, ws, ws_bytes, S(stream));
}

size_t geobi_vertex_faces_ws_bytes(int64_t F, int64_t V) { return vertex_faces_ws_bytes(F, V); }

int geobi_vertex_faces(const int32_t* fv, int64_t F, int64_t V, int32_t* rowptr, int32_t* list, void* ws,
                       size_t ws_bytes, void* stream) {
  SIZES(F > V ? F : V, 0);
  NOTNULL(fv); NOTNULL(rowptr); NOTNULL(list); NOTNULL(ws);
  return vertex_faces(fv, F, V, rowptr, list, ws, ws_bytes, S(stream));
}

int geobi_vf_padded(const int32_t* rowptr, const int32_t* list, int64_t V, int maxval, int32_t* vf, void* stream) {
  SIZES(V, 0);
  NOTNULL(rowptr); NOTNULL(list); NOTNULL(vf);
  return vf_padded(rowptr, list, V, maxval, vf, S(stream));
}

int geobi_max_degree(const int32_t* rowptr, int64_t N, int32_t* out, void* stream) {
  SIZES(N, 0);
  NOTNULL(rowptr); NOTNULL(out);
  return max_degree(rowptr, N, out, S(stream));
}

int geobi_mesh_normals(const float* points, const int32_t* fv, int64_t F, int64_t V, const int32_t* rowptr,
                       const int32_t* list, float* fnormal, float* centroid, float* vnormal, void* stream) {
  SIZES(F > V ? F : V, 0);
  NOTNULL(points); NOTNULL(fv); NOTNULL(fnormal); NOTNULL(centroid);
  if (vnormal != nullptr) { NOTNULL(rowptr); NOTNULL(list); }
  return mesh_normals(points, fv, F, V, rowptr, list, fnormal, centroid, vnormal, S(stream));
}

size_t geobi_ring_graph_ws_bytes(int64_t n_nodes) { return ring_graph_ws_bytes(n_nodes); }

int geobi_ring_graph_count(int kind, const int32_t* fv, const int32_t* rowptr_vf, const int32_t* list,
                           int64_t n_nodes, int32_t* rowptr_g, void* ws, size_t ws_bytes, void* stream) {
  SIZES(n_nodes, 0);
  NOTNULL(fv); NOTNULL(rowptr_vf); NOTNULL(list); NOTNULL(rowptr_g); NOTNULL(ws);
  return ring_graph_count(kind, fv, rowptr_vf, list, n_nodes, rowptr_g, ws, ws_bytes, S(stream));
}

int geobi_ring_graph_fill(int kind, const int32_t* fv, const int32_t* rowptr_vf, const int32_t* list,
                          int64_t n_nodes, const int32_t* rowptr_g, int32_t* col, void* stream) {
  SIZES(n_nodes, 0);
  NOTNULL(fv); NOTNULL(rowptr_vf); NOTNULL(list); NOTNULL(rowptr_g); NOTNULL(col);
  return ring_graph_fill(kind, fv, rowptr_vf, list, n_nodes, rowptr_g, col, S(stream));
}

size_t geobi_calc_weight_parts_ws_bytes(int n_parts) { return calc_weight_parts_ws_bytes(n_parts); }

int geobi_calc_weight_parts(const float* pos, const float* normal, const int32_t* rowptr, const int32_t* row,
                            const int32_t* col, int64_t E, const int32_t* node_ptr, int n_parts, float* w, void* ws,
                            size_t ws_bytes, void* stream) {
  SIZES(0, E);
  NOTNULL(pos); NOTNULL(normal); NOTNULL(rowptr); NOTNULL(node_ptr); NOTNULL(ws);
  if (E > 0) { NOTNULL(row); NOTNULL(col); NOTNULL(w); }
  return calc_weight_parts(pos, normal, rowptr, row, col, E, node_ptr, n_parts, w, ws, ws_bytes, S(stream));
}

size_t geobi_calc_weight_ws_bytes(void) { return calc_weight_ws_bytes(); }

int geobi_calc_weight(const float* pos, const float* normal, const int32_t* row, const int32_t* col, int64_t E,
                      int64_t extra_zero_edges, float* w, float* mean_len, void* ws, size_t ws_bytes, void* stream) {
  SIZES(0, E);
  NOTNULL(pos); NOTNULL(ws);
  if (E > 0) { NOTNULL(row); NOTNULL(col); }
  if (w != nullptr) NOTNULL(normal);
  return calc_weight(pos, normal, row, col, E, extra_zero_edges, w, mean_len, ws, ws_bytes, S(stream));
}

size_t geobi_patch_grow_state_ints(int64_t F, int64_t V) { return patch_grow_state_ints(F, V); }

int geobi_patch_grow_init(int32_t* state, int64_t F, int64_t V, const float* d2, void* stream) {
  SIZES(F > V ? F : V, 0);
  NOTNULL(state); NOTNULL(d2);
  return patch_grow_init(state, F, V, d2, S(stream));
}

int geobi_patch_grow(const int32_t* fv, const int32_t* vf, int maxval, int64_t F, int64_t V, const float* d2, int64_t seed,
                     int64_t neighbor_count, int64_t ring_count, int patch_id, int32_t* state, int32_t* sel_out,
                     int32_t* n_out, int32_t* mailbox, int pick_next, void* stream) {
  SIZES(F > V ? F : V, 0);
  NOTNULL(fv); NOTNULL(vf); NOTNULL(state); NOTNULL(sel_out);
  return patch_grow(fv, vf, maxval, F, V, d2, seed, neighbor_count, ring_count, patch_id, state, sel_out, n_out, mailbox,
                    pick_next, S(stream));
}

// Spin (without the caller's interpreter lock: ctypes releases it for the call) until a mailbox word is non-zero or `stream`
// has drained without writing it; returns the word (0: the stream ran dry).
int geobi_host_mailbox_wait(const int32_t* word, void* stream) {
  if (word == nullptr) return 0;
  long spins = 0;
  int v;
  while ((v = __atomic_load_n(word, __ATOMIC_ACQUIRE)) == 0) {
    if ((++spins & 0x3fff) == 0 && hipStreamQuery(S(stream)) != hipErrorNotReady) return __atomic_load_n(word, __ATOMIC_ACQUIRE);
  }
  return v;
}

// Mapped host memory a kernel can hand small results over through (patch sizes): `n` int32 slots, zeroed, valid until the
// next call from the same host thread asks for more.
int geobi_host_mailbox(int n, int32_t** host_ptr) {
  NOTNULL(host_ptr);
  GEOBI_REQUIRE(n > 0 && n <= (1 << 20), "geobi_host_mailbox: 1 .. 2^20 slots");
  static thread_local int32_t* box = nullptr;
  static thread_local int cap = 0;
  if (n > cap) {
    if (box) (void)hipHostFree(box);
    box = nullptr; cap = 0;
    const int want = n < 1024 ? 1024 : n;
    GEOBI_HIP(hipHostMalloc((void**)&box, (size_t)want * sizeof(int32_t), hipHostMallocMapped | hipHostMallocPortable));
    cap = want;
  }
  for (int i = 0; i < n; ++i) box[i] = 0;
  *host_ptr = box;
  return 0;
}

size_t geobi_submesh_ws_bytes(int64_t n_sel, int64_t V) { return submesh_ws_bytes(n_sel, V); }

int geobi_submesh(const int32_t* fv, const int32_t* sel, int64_t n_sel, int64_t V, int32_t* v_idx, int32_t* f_sub,
                  int32_t* count, void* ws, size_t ws_bytes, void* stream) {
  SIZES(V > n_sel ? V : n_sel, 0);
  NOTNULL(fv); NOTNULL(sel); NOTNULL(v_idx); NOTNULL(f_sub); NOTNULL(count); NOTNULL(ws);
  return submesh(fv, sel, n_sel, V, v_idx, f_sub, count, ws, ws_bytes, S(stream));
}

int geobi_patch_accumulate(const float* vert_p, const float* norm_p, const int32_t* v_idx, const int32_t* f_idx,
                           int64_t nv, int64_t nf, float* Vp, float* Np, int32_t* sum_v, void* stream) {
  SIZES(nv > nf ? nv : nf, 0);
  NOTNULL(vert_p); NOTNULL(norm_p); NOTNULL(v_idx); NOTNULL(f_idx); NOTNULL(Vp); NOTNULL(Np); NOTNULL(sum_v);
  return patch_accumulate(vert_p, norm_p, v_idx, f_idx, nv, nf, Vp, Np, sum_v, S(stream));
}

int geobi_patch_finalize(float* Vp, float* Np, const int32_t* sum_v, int64_t V, int64_t F, float scale, float cx,
                         float cy, float cz, void* stream) {
  SIZES(V > F ? V : F, 0);
  NOTNULL(Vp); NOTNULL(Np); NOTNULL(sum_v);
  return patch_finalize(Vp, Np, sum_v, V, F, scale, cx, cy, cz, S(stream));
}

int geobi_gemm_nn(const float* A, int lda, const float* B, int ldb, int transB, float* C, int ldc, int M, int N,
                  int K, const float* bias, float slope, void* stream) {
  NOTNULL(A); NOTNULL(B); NOTNULL(C);
  GemmEpilogue ep;
  ep.bias = bias;
  ep.slope = slope;
  return gemm_nn(A, lda, B, ldb, transB, C, ldc, M, N, K, ep, S(stream));
}
size_t geobi_gemm_tn_ws_bytes(int I, int J, int64_t M) { return gemm_tn_ws_bytes(I, J, M); }
int geobi_gemm_tn(const float* A, int lda, const float* B, int ldb, int64_t M, int I, int J, float* C, int ldc,
                  void* ws, size_t ws_bytes, void* stream) {
  NOTNULL(A); NOTNULL(B); NOTNULL(C);
  TnOutput o;
  o.C = C;
  o.ldc = ldc;
  return gemm_tn(A, lda, B, ldb, M, I, J, -1, -1, o, ws, ws_bytes, S(stream));
}

int geobi_side_defer(int on) {
  g_defer = on ? 1 : 0;
  return 0;
}

int geobi_side_join(void* stream) {
  if (g_side == nullptr) return 0;          // nothing was ever forked
  GEOBI_HIP(hipEventRecord(g_join_ev, g_side));
  GEOBI_HIP(hipStreamWaitEvent(S(stream), g_join_ev, 0));
  return 0;
}

// Device -> host read of a few int32 (sizes a pooling step computed) with a spinning wait: the copy lands in a
// pinned staging buffer and the calling thread polls the event instead of sleeping on an interrupt, which
// takes ~10 us off every size read-back on this platform.
int geobi_read_i32(const int32_t* dev, int n, int32_t* host, void* stream) {
  NOTNULL(dev); NOTNULL(host);
  GEOBI_REQUIRE(n > 0 && n <= 64, "geobi_read_i32: 1..64 values");
  static thread_local int32_t* pinned = nullptr;
  static thread_local hipEvent_t ev = nullptr;
  if (pinned == nullptr) {
    GEOBI_HIP(hipHostMalloc((void**)&pinned, 64 * sizeof(int32_t), hipHostMallocDefault));
    GEOBI_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  }
  GEOBI_HIP(hipMemcpyAsync(pinned, dev, n * sizeof(int32_t), hipMemcpyDeviceToHost, S(stream)));
  GEOBI_HIP(hipEventRecord(ev, S(stream)));
  hipError_t st;
  while ((st = hipEventQuery(ev)) == hipErrorNotReady) {
  }
  GEOBI_HIP(st);
  for (int i = 0; i < n; ++i) host[i] = pinned[i];
  return 0;
}

int geobi_set_overlap(int enable) {
  g_overlap = enable ? 1 : 0;
  return 0;
}

int geobi_prof_enable(int kernel) {
  // returns recorded events to the pool; callers collect before re-enabling
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& r : g_prof) { g_event_pool.push_back(r.a); g_event_pool.push_back(r.b); }
  g_prof.clear();
  g_prof_kernel = kernel;
  return 0;
}

int geobi_prof_collect(int tag, int64_t* launches, double* total_ms, double* total_bytes) {
  NOTNULL(launches); NOTNULL(total_ms); NOTNULL(total_bytes);
  *launches = 0; *total_ms = 0.0; *total_bytes = 0.0;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& r : g_prof) {
    if (!r.closed) continue;
    GEOBI_HIP(hipEventSynchronize(r.b));
    if (tag != 0 && r.tag != tag) continue;
    float ms = 0.f;
    GEOBI_HIP(hipEventElapsedTime(&ms, r.a, r.b));
    *launches += 1;
    *total_ms += (double)ms;
    *total_bytes += r.bytes;
  }
  return 0;
}

}  // extern "C"
