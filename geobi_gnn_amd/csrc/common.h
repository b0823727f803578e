// Internal host/device helpers shared by the HIP translation units of libgeobi_hip.so.
// gfx950 (MI355X) only: 64-lane wavefronts are assumed throughout.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <atomic>

#define GEOBI_H 9    // FeaSt heads on the hot path (network.py:258-268 always passes 9)
#define GEOBI_HP 12  // row stride (floats) of per-node / per-edge head vectors: 9 padded to 3 x float4

namespace geobi {

int set_error(const char* fmt, ...);

#define GEOBI_HIP(expr)                                                                   \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return ::geobi::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
  } while (0)
#define GEOBI_LAUNCH_OK() GEOBI_HIP(hipGetLastError())
#define GEOBI_TRY(expr)        \
  do {                         \
    int r_ = (expr);           \
    if (r_ != 0) return r_;    \
  } while (0)
#define GEOBI_REQUIRE(cond, ...)                          \
  do {                                                    \
    if (!(cond)) return ::geobi::set_error(__VA_ARGS__);  \
  } while (0)

static inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// bump allocator over a caller-provided workspace (no hipMalloc on the hot path)
struct Arena {
  char* base;
  size_t off, cap;
  bool dry;  // dry run: only measure
  Arena(void* p, size_t bytes) : base((char*)p), off(0), cap(bytes), dry(p == nullptr) {}
  template <typename T>
  T* take(size_t n) {
    size_t o = align_up(off);
    off = o + n * sizeof(T);
    if (dry) return nullptr;
    return (off <= cap) ? (T*)(base + o) : nullptr;
  }
  bool ok() const { return dry || off <= cap; }
};

// ---- optional per-kernel timing for bench.py's roofline object (events on the launch stream)
enum ProfKernel { PROF_NONE = 0, PROF_AGG_FWD = 1, PROF_AGG_BWD = 2, PROF_ROWPASS = 3, PROF_GEMM = 4 };
void prof_begin(int kernel, hipStream_t s, double alg_bytes, int tag);
void prof_end(int kernel, hipStream_t s);
// The same bracket for ONE kernel launched through hipExtLaunchKernelGGL: the two events are bound to the kernel's own
// dispatch (their elapsed time is the kernel's execution time, as a tracer reports it -- a pair of hipEventRecord packets
// around a launch adds ~4.7 us of its own).  prof_begin_launch hands the events to the launch site of this host thread,
// which takes them with prof_take_launch_events (false: not profiling, launch as usual).
void prof_begin_launch(int kernel, hipStream_t s, double alg_bytes, int tag);
bool prof_take_launch_events(hipEvent_t* start, hipEvent_t* stop);

// ---- side stream for work that is off the critical path of a backward call (weight gradients)
struct Fork {
  hipStream_t side = nullptr;   // nullptr: overlap disabled, run on the caller's stream
  hipEvent_t join = nullptr;
};
void side_override(int mode);                  // this host thread: 0 = no side stream, 1 = side stream, -1 = the process-wide setting
hipStream_t side_current();                    // the side stream this thread's last fork used (nullptr: none yet)
void side_select(int low_priority);            // which side stream the next forks use (default / lowest priority)
Fork fork_side_stream(hipStream_t main);       // side stream waits for everything enqueued on `main` so far
int join_side_stream(const Fork& f, hipStream_t main);   // `main` waits for the side stream
int side_wait_main(const Fork& f, hipStream_t main);     // side stream waits for `main` as of now

// ---- launchers implemented across the translation units (all enqueue on `s`, never sync)
// graph.hip
size_t csr_ws_bytes(int64_t E, int64_t N);
int csr_from_coo(const int64_t* seg, const int64_t* nbr, int64_t E, int64_t N, int drop_self, int32_t* rowptr,
                 int32_t* col, int32_t* eid, int32_t* bad, void* ws, size_t ws_bytes, hipStream_t s);
int csr_transpose(const int32_t* rowptr, const int32_t* col, int64_t N, int64_t Ecap, int32_t* rowptr_t,
                  int32_t* col_t, int32_t* pos_t, int32_t* inv_pos, void* ws, size_t ws_bytes, hipStream_t s);
int csr_reverse_index(const int32_t* rowptr, const int32_t* row, const int32_t* col, int64_t E, int32_t* pos_rev,
                      int32_t* flag, hipStream_t s);
// gemm.hip
struct GemmEpilogue {
  const float* bias = nullptr;  // [N] added per column
  float slope = 1.0f;           // leaky-relu negative slope (1 = identity)
  float* C1 = nullptr;          // optional split output: columns >= split go to C1 (ld = ldc1)
  int split = 0, ldc1 = 0;
  void* ws = nullptr;           // optional scratch for split-K partial sums (small-M GEMMs)
  size_t ws_bytes = 0;
  int fixed_slices = 0;         // > 0: split K exactly this way whatever M is (forward pass: keeps a
                                // node's result independent of how many other nodes are batched with it)
};
// split-K only engages for grids of < 512 small tiles, i.e. M * N < ~2.1 M elements
static inline size_t gemm_nn_ws_bytes(int64_t M, int N) {
  return (M * N < (int64_t)2200000) ? align_up((size_t)8 * M * N * sizeof(float)) + 256 : 256;
}
// forward GEMM of a FeaSt layer: the split is a function of the layer shape only
// packed-weight buffer shared by forward and backward: Wf [ldz(Cin), Cout] for z Wf / g Wf^T, then
// W' [ldr(Cout), Cin] = [lin.weight ; u.weight ; 0] for dx = r' W'
// feast_fused.hip: the fused (aggregate -> LDS tile -> MFMA) kernels and their packed weights
int feast_fused_nt(int nout);
size_t feast_fused_fwd_pack_floats(int Cin, int Cout);
size_t feast_fused_dx_pack_floats(int Cin, int Cout);
int feast_fused_pack_fwd(const float* lin_w, const float* c, int Cin, int Cout, float* bp, hipStream_t s);
// one launch for the packed forms of several layers (feast_fused.hip)
constexpr int kMaxPackBatch = 8;
struct FusedPackItem {
  const float* lin_w;
  const float* u_w;
  const float* c;
  int Cin, Cout;
  float* wf;     // [Kp, Cout] plain form (NULL: forward form only)
  float* bf;     // fragment-ordered forward weights
  float* bdx;    // fragment-ordered dx weights (with wf)
};
int feast_fused_pack_batch(const FusedPackItem* items, int n, hipStream_t s);
int feast_fused_pack_dx(const float* lin_w, const float* u_w, const float* c, int Cin, int Cout, float* bp, hipStream_t s);
int feast_fused_pack_all(const float* lin_w, const float* u_w, const float* c, int Cin, int Cout, int Kp, float* wf,
                         float* bf, float* bdx, hipStream_t s);
double feast_fused_bytes(int64_t N, int64_t E, int C, int nout);
int feast_fused_fwd(const float* xa, const float* xb, int Ca, int Cin, const float* p, const float* cvec,
                    const int* rowptr, const int* col, int N, int LC, const float* ul, const float* Bp, int Cout,
                    const float* bias, float slope, float* out, hipStream_t s);
int feast_fused_dx(const float* g, int Cout, const float* p, const float* cvec, const int* rowptr_out,
                   const int* col_out, const int* rowptr_in, const int* pos, const float* dl, const float* dpn, int N,
                   int LC, const float* xl, const float* ul, const float* dpd, const float* Bp, int Cin, float* dxa,
                   int Ca, float* dxb, int Cb, float* tile_out, hipStream_t s);
bool feast_rowpass_fused_supported(int Cin, int Cb, int Cout);
int set_tile_rows(int rows);
int set_rowpass_form(int staged, int chunked64);
int set_column_parts(int parts);
int feast_rowpass_fused(const float* xa, const float* xb, int Ca, int Cin, const float* p, const float* cvec,
                        const int* rowptr, const int* col, int N, int LC, const float* ul, const float* gout,
                        const float* out_act, float slope, int Cout, const float* Wf, int Kp, float* g_out, float* dl,
                        float* dpn, float* dcs, int ld_dcs, hipStream_t s);
// packed-weight buffer layout: [Wf | W' | Bf (fused forward) | Bdx (fused dx)]
static inline size_t feast_wpack_plain_floats(int Cin, int Cout) {
  return (size_t)((GEOBI_H * Cin + 3) / 4 * 4) * Cout + (size_t)(GEOBI_H * Cout + 2 * GEOBI_HP) * Cin;
}
static inline size_t feast_wpack_floats(int Cin, int Cout) {
  return feast_wpack_plain_floats(Cin, Cout) + feast_fused_fwd_pack_floats(Cin, Cout) +
         feast_fused_dx_pack_floats(Cin, Cout);
}
static inline int feast_fwd_slices(int Kp, int Cout) { return Kp >= 1024 ? 4 : ((Kp >= 512 && Cout >= 64) ? 2 : 1); }
static inline size_t gemm_nn_fixed_ws_bytes(int64_t M, int N, int slices) {
  size_t b = (size_t)slices * M * N * sizeof(float);
  return (slices > 1 && b <= ((size_t)256 << 20)) ? align_up(b) + 256 : 256;
}
int gemm_nn(const float* A, int lda, const float* B, int ldb, int transB, float* C, int ldc, int M, int N, int K,
            const GemmEpilogue& ep, hipStream_t s);
size_t gemm_tn_ws_bytes(int I, int J, int64_t M);
size_t gemm_tn_ws_bytes_any_width(int I, int J, int64_t M);   // max over column widths 1..J
enum TnOut { TN_PLAIN = 0, TN_LIN_UNPACK = 1, TN_DU_DC = 2, TN_RPRIME = 3 };
struct TnOutput {
  int mode = TN_PLAIN;
  float* C = nullptr;   // primary output
  int ldc = 0;
  float* C2 = nullptr;  // secondary output (bias gradient / c gradient)
  int Cin = 0, Cout = 0;
  int extra_row = 0, extra_col = 0;  // TN_PLAIN: last row / column is the implicit-ones one -> C2
  int accumulate = 0;   // != 0: add to what C / C2 hold (gradient accumulation) instead of overwriting
  // TN_RPRIME ([x | 1]^T [r | dp | dcs]): C = dlin, C2 = dbias, C3 = du, C4 = dc; col0 = first input channel of
  // this A operand (split inputs issue one GEMM per half)
  float* C3 = nullptr;
  float* C4 = nullptr;
  int col0 = 0;
};
// I, J: logical output sizes INCLUDING an implicit ones row / column when ones_row / ones_col >= 0, or the column-sums
// row (sum_row == I - 1: the sums of B's columns over the M rows, formed on the VALU beside the MFMAs; TnOutput.extra_row
// routes it).
int gemm_tn(const float* A, int lda, const float* B, int ldb, int64_t M, int I, int J, int ones_row, int ones_col,
            const TnOutput& o, void* ws, size_t ws_bytes, hipStream_t s, int sum_row = -1);
size_t colsum_ws_bytes(int64_t M, int J);
int colsum(const float* A, int lda, int64_t M, int J, float* out, void* ws, size_t ws_bytes, hipStream_t s);

// feast.hip
int exp_le0_probe(const float* x, float* y, int64_t n, hipStream_t s);
int feast_ldz(int Cin);
size_t feast_fwd_ws_bytes(int64_t N, int Cin, int Cout);
int feast_fwd(const float* xa, const float* xb, int Ca, int Cb, int64_t N, int64_t E, const int32_t* rowptr_in,
              const int32_t* col_in, const float* lin_w, const float* u_w, const float* cvec, const float* bias,
              int Cout, float slope, float* out, float* p, float* z, float* wf_out, void* ws, size_t ws_bytes,
              hipStream_t s, const float* bf_packed = nullptr);
size_t feast_bwd_ws_bytes(int64_t N, int64_t E, int Cin, int Cout);
size_t feast_bwd_ws_bytes_for(int64_t N, int64_t E, int Cin, int Cb, int Cout, bool need_dx);
int feast_bwd(const float* xa, const float* xb, int Ca, int Cb, int64_t N, int64_t E, const int32_t* rowptr_in,
              const int32_t* col_in, const int32_t* rowptr_out, const int32_t* col_out, const int32_t* pos_in,
              const float* lin_w, const float* u_w, const float* cvec, int Cout, float slope, const float* out,
              const float* gout, const float* p, const float* z, const float* wf_saved, float* dxa, float* dxb,
              float* dlin_w, float* du_w, float* dc, float* dbias, int accumulate, void* ws, size_t ws_bytes,
              hipStream_t s);
struct CopySeg { const void* src; void* dst; int64_t n; int32_t add; float value; };     // = geobi_copy_seg_t
int concat32(const CopySeg* segs, int n_segs, int is_float, hipStream_t s);
// pool.hip
int edge_weight_t10(const float* x, int C, const int32_t* row, const int32_t* col, const float* w_in, int64_t E,
                    float* w_out, hipStream_t s, int32_t* zero8 = nullptr);
int edge_weight_att(const float* x, int C, const float* att_l, const float* att_r, const int32_t* row, const int32_t* col,
                    const float* w_in, int64_t N, int64_t E, float* node_ws, float* w_out, hipStream_t s);
int gather_f32(const float* src, const int32_t* idx, int64_t n, float* dst, hipStream_t s);
int expand_rowptr(const int32_t* rowptr, int64_t N, int32_t* row, hipStream_t s);
size_t match_ws_bytes(int64_t N);
size_t match_coarsen_ws_bytes(int64_t N);
int match_coarsen(const int32_t* rowptr, const int32_t* col, const float* w, int64_t N, int rounds, int init,
                  int32_t* state, int32_t* cluster_final, int32_t* cnew, int32_t* segptr, int32_t* members,
                  int32_t* counters, void* ws, size_t ws_bytes, hipStream_t s, void* rowinfo_out = nullptr,
                  bool* rowinfo_made = nullptr);
int match_heavy_edge(const int32_t* rowptr, const int32_t* col, const float* w, int64_t N, int rounds, int init,
                     int32_t* cluster, int32_t* cluster_final, int32_t* status, void* ws, size_t ws_bytes, hipStream_t s);
void set_match_round_cap(int cap);
size_t relabel_ws_bytes(int64_t N);
int relabel_compact(const int32_t* cluster, int64_t N, int rep_is_self, int32_t* cnew, int32_t* count, void* ws,
                    size_t ws_bytes,
                    hipStream_t s);
size_t segment_csr_ws_bytes(int64_t n);
int segment_csr(const int32_t* seg, int64_t n, int64_t nseg, int32_t* segptr, int32_t* members, void* ws,
                size_t ws_bytes, hipStream_t s);
size_t segment_pairs_ws_bytes(int64_t nseg);
int segment_csr_pairs(const int32_t* cnew, const int32_t* raw, int64_t N, int64_t nseg, int32_t* segptr,
                      int32_t* members, void* ws, size_t ws_bytes, hipStream_t s);
int segment_csr_compose(const int32_t* segptr1, const int32_t* members1, const int32_t* segptr2,
                        const int32_t* members2, int64_t nseg2, int64_t n_fine, int32_t* segptr12,
                        int32_t* members12, void* ws, size_t ws_bytes, hipStream_t s);
int segment_max_fwd(const float* x, int C, const int32_t* segptr, const int32_t* members, int64_t nseg, float* out,
                    int32_t* arg, hipStream_t s);
int segment_sum2(const float* x, int C, const int32_t* segptr1, const int32_t* members1, const int32_t* segptr2,
                 const int32_t* members2, int64_t nseg2, float* out, hipStream_t s);
int segment_max2_fwd(const float* x, int C, const int32_t* segptr1, const int32_t* members1, const int32_t* segptr2,
                     const int32_t* members2, int64_t nseg2, float* out, int32_t* arg12, hipStream_t s);
int segment_max2_bwd(const float* gout, const int32_t* arg12, const int32_t* seg12, int C, int64_t nseg2, int64_t n_fine,
                     float* gx, int add, hipStream_t s);
int segment_max_bwd(const float* gout, const int32_t* arg, const int32_t* seg, int C, int64_t nseg, int64_t n_fine,
                    float* gx, hipStream_t s);
int segment_sum(const float* x, int C, const int32_t* segptr, const int32_t* members, int64_t nseg, int mean,
                float* out, hipStream_t s);
int segment_mean_bwd(const float* gout, const int32_t* seg, const int32_t* segptr, int C, int64_t n_fine, float* gx,
                     hipStream_t s);
int gather_rows(const float* x, const int32_t* idx, int C, int64_t n_out, float* out, hipStream_t s);
size_t pool_edge_rows_ws_bytes(int64_t nbound);
size_t pool_edge_rows_ws_bytes_onepass(int64_t nbound, int64_t E);
int pool_edge_rows(const int32_t* cnew, const int32_t* segptr, const int32_t* members, const int32_t* rowptr,
                   const int32_t* col, const float* w, const int32_t* ncount, int64_t nbound, int32_t* rowptr_c,
                   int32_t* row_c, int32_t* col_c, float* w_c, int32_t* count, int32_t* overflow, void* ws,
                   size_t ws_bytes, hipStream_t s, int64_t E_fine = 0, const void* rowinfo_in = nullptr,
                   const int32_t* publish_src = nullptr, int32_t* publish_host = nullptr, int publish_seq = 0);
size_t pool_edge_ws_bytes(int64_t E);
int pool_edge(const int32_t* cnew, const int32_t* row, const int32_t* col, const float* w, int64_t E, int64_t nmax,
              int32_t* rowptr_c, int32_t* row_c, int32_t* col_c, float* w_c, int32_t* count, void* ws,
              size_t ws_bytes, hipStream_t s);
// meshprep.hip (mesh -> graphs / normals / bilateral weights on the device)
size_t vertex_faces_ws_bytes(int64_t F, int64_t V);
int vertex_faces(const int32_t* fv, int64_t F, int64_t V, int32_t* rowptr, int32_t* list, void* ws, size_t ws_bytes,
                 hipStream_t s);
int vf_padded(const int32_t* rowptr, const int32_t* list, int64_t V, int maxval, int32_t* vf, hipStream_t s);
int max_degree(const int32_t* rowptr, int64_t N, int32_t* out, hipStream_t s);
int mesh_normals(const float* points, const int32_t* fv, int64_t F, int64_t V, const int32_t* rowptr,
                 const int32_t* list, float* fnormal, float* centroid, float* vnormal, hipStream_t s);
size_t ring_graph_ws_bytes(int64_t n_nodes);
int ring_graph_count(int kind, const int32_t* fv, const int32_t* rowptr_vf, const int32_t* list, int64_t n_nodes,
                     int32_t* rowptr_g, void* ws, size_t ws_bytes, hipStream_t s);
int ring_graph_fill(int kind, const int32_t* fv, const int32_t* rowptr_vf, const int32_t* list, int64_t n_nodes,
                    const int32_t* rowptr_g, int32_t* col, hipStream_t s);
size_t calc_weight_parts_ws_bytes(int n_parts);
int calc_weight_parts(const float* pos, const float* normal, const int32_t* rowptr, const int32_t* row, const int32_t* col,
                      int64_t E, const int32_t* node_ptr, int n_parts, float* w, void* ws, size_t ws_bytes, hipStream_t s);
size_t calc_weight_ws_bytes();
int calc_weight(const float* pos, const float* normal, const int32_t* row, const int32_t* col, int64_t E,
                int64_t extra_zero_edges, float* w, float* mean_len, void* ws, size_t ws_bytes, hipStream_t s);
// patch.hip (patch split / merge for large meshes)
size_t patch_grow_state_ints(int64_t F, int64_t V);
int patch_grow_init(int32_t* state, int64_t F, int64_t V, const float* d2, hipStream_t s);
int patch_grow(const int32_t* fv, const int32_t* vf, int maxval, int64_t F, int64_t V, const float* d2, int64_t seed,
               int64_t neighbor_count, int64_t ring_count, int patch_id, int32_t* state, int32_t* sel_out, int32_t* n_out,
               int32_t* mailbox, int pick_next, hipStream_t s);
size_t submesh_ws_bytes(int64_t n_sel, int64_t V);
int submesh(const int32_t* fv, const int32_t* sel, int64_t n_sel, int64_t V, int32_t* v_idx, int32_t* f_sub,
            int32_t* count, void* ws, size_t ws_bytes, hipStream_t s);
int patch_accumulate(const float* vert_p, const float* norm_p, const int32_t* v_idx, const int32_t* f_idx, int64_t nv,
                     int64_t nf, float* Vp, float* Np, int32_t* sum_v, hipStream_t s);
int patch_finalize(float* Vp, float* Np, const int32_t* sum_v, int64_t V, int64_t F, float scale, float cx, float cy,
                   float cz, hipStream_t s);
// pool.hip: exclusive int scan shared with meshprep.hip (single launch below 2^18 elements)
size_t scan_ws_bytes(int64_t n);
int scan_exclusive_i32(void* temp, size_t temp_bytes, const int* in, int* out, int64_t n, hipStream_t s);
// geom.hip
int face_geom_fwd(const float* verts, const int32_t* fv, const float* xf, int ldxf, int64_t F, float* out,
                  hipStream_t s);
int face_geom_bwd(const float* verts, const int32_t* fv, const float* g, int64_t F, float* corner_grad,
                  hipStream_t s);
int head_fwd(const float* x, int Cin, int64_t N, const float* w1, const float* b1, int K, const float* w2,
             const float* b2, int nout, float slope, int mode, const float* dd, const float* resid, int ld_resid,
             float* h, float* raw, float* out, hipStream_t s);
size_t row_loss_ws_bytes(int64_t n);
int row_loss_fwd(const float* a, const float* b, const float* w, int64_t n, int kind, float scale, float* out,
                 void* ws, size_t ws_bytes, hipStream_t s);
int row_loss_bwd(const float* a, const float* b, const float* w, const float* gout, int64_t n, int kind,
                 float scale, float* ga, hipStream_t s);
int adam_flat(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps, float wd,
              float bias_corr1, float bias_corr2, hipStream_t s);
size_t update_position_ws_bytes(int64_t V, int64_t F);
int update_position2(const float* points, const int32_t* fv, const int32_t* vf, int maxval, const float* normals,
                     const float* dd, int64_t V, int64_t F, int n_iter, float* out, void* ws, size_t ws_bytes,
                     hipStream_t s);
size_t head_bwd_ws_bytes(int64_t N, int Cin, int K);
// head_fused.hip
bool head_fused_supported(int Cin, int K, int nout);
int set_head_precision(int mode);
int head_fwd_fused(const float* x, int64_t N, const float* w1, const float* b1, const float* w2, const float* b2,
                   int nout, float slope, int mode, const float* dd, const float* resid, int ld_resid, float* raw,
                   float* out, hipStream_t s);
size_t head_bwd_fused_ws_bytes(int64_t N);
int head_bwd_fused(const float* x, int64_t N, const float* w1, const float* b1, const float* w2, int nout,
                   float slope, const float* graw, float* dx, float* dw1, float* db1, float* dw2, float* db2,
                   int accumulate, void* ws, size_t ws_bytes, hipStream_t s);
int head_bwd(const float* x, int Cin, int64_t N, const float* w1, const float* b1, int K, const float* w2, int nout,
             float slope, int mode, const float* dd, const float* h, const float* raw, const float* gout, float* dx, float* dw1,
             float* db1, float* dw2, float* db2, int accumulate, void* ws, size_t ws_bytes, hipStream_t s);

}  // namespace geobi
