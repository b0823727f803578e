/* libgeobi_hip.so -- C ABI of the MI355X-native bi-domain mesh-graph convolution path.
 *
 * The reference (zhangyk18/GeoBi-GNN) has no FFI layer: its "operator API" for this path is the
 * Python nn.Module surface (code/network.py:254-343, code/net_util.py:56-380) over third-party
 * PyTorch extensions.  Each entry point below replaces the native kernel(s) one of those call
 * sites dispatches to; the replaced call site is cited per function.  INTEGRATION.md shows the
 * ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a BORROWED device pointer (HBM) unless marked `host`; the library never
 *     allocates, frees or synchronises: scratch comes in through `ws` / `ws_bytes` (query the size
 *     with the matching *_ws_bytes function, which is a pure host computation + rocPRIM size query).
 *     Two documented exceptions (plus one helper): geobi_read_i32 (the size read-back: waits for `stream`), the whole-network
 *     entry points geobi_net_forward / geobi_net_forward_train / geobi_net_train_groups (four reads of pooling sizes
 *     per pass, through mapped host memory) -- and, for its own purpose, geobi_host_mailbox (hands out mapped HOST memory)
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*)
 *   - return value 0 = ok, non-zero = error; the message is in geobi_last_error() (thread-local)
 *   - node features are row-major fp32; indices inside the library are int32.  Per call and graph level at most
 *     GEOBI_MAX_NODES nodes (rows of any per-node array, faces and vertices alike) and GEOBI_MAX_EDGES edges: below
 *     these every 32-bit element index a kernel forms -- node x channels (<= 128), node x 16 lanes, edge x 16 lanes --
 *     stays under 2^31 / 2^32; byte and float offsets into per-node / per-edge rows are 64-bit throughout.  Larger
 *     sizes are REJECTED by the entry points (error code, message), never truncated.  The path's largest
 *     configuration (a 150 k-face scan: 1.97 M edges) is 1 % of the limits; bigger meshes go through the patch split.
 *     The reference's int64 COO `edge_index` is accepted by geobi_csr_from_coo
 *   - FeaSt heads are fixed at 9 (every FeaStConv on the path is built with heads=9,
 *     code/network.py:258-268); per-node / per-edge head vectors use a padded row stride of
 *     GEOBI_HEAD_STRIDE floats
 */
#ifndef GEOBI_HIP_H_
#define GEOBI_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GEOBI_HEADS 9
#define GEOBI_HEAD_STRIDE 12
#define GEOBI_MAX_NODES ((1 << 24) - 1)  /* 16 777 215 */
#define GEOBI_MAX_EDGES ((1 << 28) - 1)  /* 268 435 455 */

int geobi_version(void);
const char* geobi_last_error(void);

/* ---------------------------------------------------------------- adjacency (CSR) ----------
 * Replaces the per-call COO handling inside torch_geometric.MessagePassing / remove_self_loops /
 * add_self_loops (FeaStConv.forward, called at code/network.py:271-299) and the CSR build inside
 * torch_cluster.graclus (code/net_util.py:127).
 *
 * geobi_csr_from_coo: sort E (segment, neighbour) int64 pairs by (segment, neighbour); self loops
 * are dropped when drop_self != 0.  rowptr[N] is the number of kept edges; col / eid have E slots
 * (eid[k] = position of sorted edge k in the input COO).  Pairs with an id outside [0, N) are dropped
 * and counted in bad[0] (device int32, optional): they never reach a kernel as an index.
 * geobi_csr_transpose: CSR of the reversed edges; pos_t[e'] = index of that edge in the input CSR,
 * inv_pos[e] = index of input edge e in the transposed CSR (either may be NULL... pos_t may not). */
size_t geobi_csr_ws_bytes(int64_t E, int64_t N);
int geobi_csr_from_coo(const int64_t* seg, const int64_t* nbr, int64_t E, int64_t N, int drop_self,
                       int32_t* rowptr, int32_t* col, int32_t* eid, int32_t* bad, void* ws, size_t ws_bytes,
                       void* stream);
int geobi_csr_transpose(const int32_t* rowptr, const int32_t* col, int64_t N, int64_t Ecap, int32_t* rowptr_t,
                        int32_t* col_t, int32_t* pos_t, int32_t* inv_pos, void* ws, size_t ws_bytes, void* stream);
/* geobi_csr_reverse_index: for a (row, col)-sorted CSR, pos_rev[e] = position of the reverse of edge e
 * (or -1, with flag[0] |= 1, when it is missing; flag is zeroed first and may be NULL when the caller
 * already knows the graph is symmetric).  For a symmetric graph (every mesh graph of the
 * path, and every pooled graph derived from one) the transposed CSR equals the CSR and pos_rev is the
 * edge correspondence -- no second sort. */
int geobi_csr_reverse_index(const int32_t* rowptr, const int32_t* row, const int32_t* col, int64_t E,
                            int32_t* pos_rev, int32_t* flag, void* stream);
int geobi_expand_rowptr(const int32_t* rowptr, int64_t N, int32_t* row, void* stream);
int geobi_gather_f32(const float* src, const int32_t* idx, int64_t n, float* dst, void* stream);
/* geobi_concat32: the disjoint-union batch of several meshes (the data loader's collate step: code/train_dual.py:199-201
 * hands the meshes over one by one; a batch is their union) as ONE launch per element type instead of a cat + add per
 * array and mesh.  `segs` is a HOST array of n_segs copy jobs over 4-byte elements:
 *   dst[i] = src[i] + add      (is_float == 0: int32 -- row pointers shifted by the edge offset, neighbour / vertex ids by
 *                               the node offset, reverse-edge and corner positions by theirs)
 *   dst[i] = src[i]            (is_float != 0: features, targets, weights; `add` ignored)
 *   src == NULL: dst[i] = add  (int32) / value (float): closing row pointers, constant per-mesh loss weights.          */
typedef struct {
  const void* src;
  void* dst;
  int64_t n;
  int32_t add;
  float value;
} geobi_copy_seg_t;
int geobi_concat32(const geobi_copy_seg_t* segs, int n_segs, int is_float, void* stream);

/* ---------------------------------------------------------------- FeaSt convolution --------
 * Replaces torch_geometric.nn.FeaStConv.forward / its autograd (16 call sites,
 * code/network.py:271,279,286,287,290,293,296,299), optionally fused with the leaky_relu that
 * follows most of them (slope = 1 disables it) and with the skip concatenation of
 * code/network.py:292,298 (the input may be given as two equal halves xa | xb; Cb = 0 otherwise).
 *
 *   in-CSR  : rowptr_in/col_in   -- for every TARGET node its SOURCE nodes (edge_index[0] -> [1])
 *   out-CSR : rowptr_out/col_out -- for every SOURCE node its TARGET nodes; pos_in[e] = position of
 *             out-edge e in the in-CSR.  Both without self loops (one per node is implied).
 *   lin_w [9*Cout, Cin], u_w [9, Cin], c [9], bias [Cout]   (PyG >= 2.0 state-dict layout)
 *   forward saves p [N, 12] (x u^T; not written for unsplit 6- / 12-channel inputs, whose logits are formed
 *   per edge from the rows themselves) for the backward; `out` after the activation is needed by the backward
 *   when slope != 1.
 *   z == NULL (the default of the Python layer): FUSED path -- aggregation and node transform in one kernel, the
 *   aggregated features [N, 9 Cin] stay in LDS and never reach HBM; the backward (z == NULL as well) recomputes
 *   them and forms dx in a second fused kernel.  z != NULL: [N, geobi_feast_ldz(Cin)] receives the aggregated
 *   features (separate aggregation + GEMM kernels) and must be handed to the backward.
 *   wf (optional, geobi_feast_wpack_floats(Cin, Cout) floats) receives the packed weights -- Wf [ldz, Cout],
 *   W' = [lin.weight ; u.weight ; 0] [9 Cout + 24, Cin], then the MFMA-fragment-ordered forms of both that the
 *   fused kernels read -- in the forward and, handed back to the backward, saves repacking them there (NULL:
 *   packed into the workspace on both sides).
 *   E = number of edges in the CSR (used for scratch sizing and byte accounting only).
 *   Supported channel counts: Cin, Cout in {6, 12, 32, 64, 128} (Cout: 32, 64, 128).           */
int geobi_feast_ldz(int Cin);
size_t geobi_feast_wpack_floats(int Cin, int Cout);
size_t geobi_feast_fwd_ws_bytes(int64_t N, int Cin, int Cout);
int geobi_feast_fwd(const float* xa, const float* xb, int Ca, int Cb, int64_t N, int64_t E,
                    const int32_t* rowptr_in, const int32_t* col_in, const float* lin_w, const float* u_w,
                    const float* c, const float* bias, int Cout, float slope, float* out, float* p, float* z,
                    float* wf, void* ws, size_t ws_bytes, void* stream);
size_t geobi_feast_bwd_ws_bytes(int64_t N, int64_t E, int Cin, int Cout);
/* dxa/dxb may be NULL (input needs no gradient); dlin_w/du_w/dc/dbias are overwritten, or added to when
 * accumulate != 0 (a caller that lets the kernels write into persistent .grad storage: gradient accumulation
 * over several backward passes, code/train_dual.py:211-218). */
int geobi_feast_bwd(const float* xa, const float* xb, int Ca, int Cb, int64_t N, int64_t E,
                    const int32_t* rowptr_in, const int32_t* col_in, const int32_t* rowptr_out,
                    const int32_t* col_out, const int32_t* pos_in, const float* lin_w, const float* u_w,
                    const float* c, int Cout, float slope, const float* out, const float* gout, const float* p,
                    const float* z, const float* wf, float* dxa, float* dxb, float* dlin_w, float* du_w, float* dc,
                    float* dbias, int accumulate, void* ws, size_t ws_bytes, void* stream);

/* Tile geometry of the fused FeaSt kernels: 16 (default: 16-node tiles on v_mfma_f32_16x16x4_f32, four workgroups per
 * CU) or 32 (round-2 geometry: v_mfma_f32_32x32x2_f32, two per CU); 0 returns to the GEOBI_TILE16 environment default.
 * Process-wide; for parity tests and same-process A/B timing.                                                       */
int geobi_set_tile_rows(int rows);

/* Form of the fused backward row pass of layers reading 64 channels; each argument 1, 0 or -1 (= default).
 * staged (default 1, GEOBI_ROWPASS_STAGED): the neighbour rows reach the lane = edge dot products through LDS
 *   (global_load_lds_dwordx4, half a row per eight lanes) instead of sixteen private 16-B reads per lane; bit-identical.
 * chunked64 (default: Cout = 128 only, GEOBI_ROWPASS_CHUNKED64): the channel-chunked kernel of the 128-channel layers
 *   (32-node tiles, 32 channels at a time) instead of the 16-node-tile kernel.
 * Process-wide; for parity tests and same-process A/B timing.                                                       */
int geobi_set_rowpass_form(int staged, int chunked64);
/* column parts of the fused FeaSt kernel (16-row tiles, layers reading 128 channels): 2 = every tile is worked on by two
 * workgroups, each producing half of the output columns; 1 = off; 0 = chosen per launch from the tile count (default: two parts
 * up to 512 tiles; GEOBI_COLUMN_PARTS, GEOBI_COLUMN_PARTS_MAX_TILES).  Bit-identical results. */
int geobi_set_column_parts(int parts);

/* ---------------------------------------------------------------- pooling ------------------
 * geobi_edge_weight_t10 : PoolingLayer._get_edge_weight, edge_weight_type 10
 *                         (code/net_util.py:226-230):  w_out = w_in + exp(-|x_row - x_col|^2 / 2)
 * geobi_match_heavy_edge: torch_cluster.graclus (code/net_util.py:127,325,353): heavy-edge matching,
 *                         cluster[u] = cluster[v] = min(u, v), unmatched -> own id.  Deterministic
 *                         (greedy in descending edge order).  init != 0 starts from scratch, init == 0
 *                         continues from `cluster` (the resumable state: -1 undecided, u closed as a
 *                         singleton, v matched with partner v); status[0] = nodes still undecided
 *                         after `rounds` proposal rounds (0 = converged); cluster_final (optional)
 *                         receives a copy with the undecided nodes closed as singletons.
 * geobi_relabel_compact : torch_geometric consecutive_cluster (code/net_util.py:128): dense ids by
 *                         ascending cluster id; count[0] = number of clusters (device int32).
 *                         rep_is_self != 0: the caller guarantees that every cluster id is the index of
 *                         one of its own members with cluster[id] == id (true for graclus / the matching
 *                         above: id = min member) -- saves the occupancy pass.
 * geobi_segment_csr     : inverse lists segment -> members (ascending), the sorted-segment form of
 *                         torch_scatter's index argument.
 * geobi_segment_max_*   : torch_scatter.scatter(reduce='max') + backward (code/net_util.py:134);
 *                         first maximum wins, empty segments give 0.
 * geobi_segment_sum     : scatter(reduce='sum'|'mean') forward (code/net_util.py:132) and the
 *                         backward of the unpool gather `x[unpooling_indices]` (:242-245).
 * geobi_gather_rows     : PoolingLayer.unpooling forward (code/net_util.py:242-245).
 * geobi_pool_edge       : pool_edge (code/net_util.py:289-295): relabel endpoints, drop loops,
 *                         sort by (row, col), merge duplicates (mean of weights).  Outputs sized for
 *                         E entries / nmax+1 row pointers; count[0] = kept edges (device int32).   */
int geobi_edge_weight_t10(const float* x, int C, const int32_t* row, const int32_t* col, const float* w_in,
                          int64_t E, float* w_out, void* stream);
/* geobi_edge_weight_att: PoolingLayer._get_edge_weight, edge_weight_type 3 / 4 / 5 (code/net_util.py:182-206): the
 * GAT-style learned weight  sigmoid((al[row] + ar[col]) + (al[col] + ar[row])),  al = x . att_l,  ar = x . att_r  per node
 * (x [N, C]: the features for type 3, leaky_relu(lin(x), 0.2) for types 4 / 5 -- geobi_gemm_nn with bias and slope);
 * w_in != NULL: averaged with the given weight, (sigmoid + w_in) / 2 (type 5).  node_ws: 2 N floats of scratch.          */
int geobi_edge_weight_att(const float* x, int C, const float* att_l, const float* att_r, const int32_t* row,
                          const int32_t* col, const float* w_in, int64_t N, int64_t E, float* node_ws, float* w_out,
                          void* stream);
size_t geobi_match_ws_bytes(int64_t N);
int geobi_match_heavy_edge(const int32_t* rowptr, const int32_t* col, const float* w, int64_t N, int rounds,
                           int init, int32_t* cluster, int32_t* cluster_final, int32_t* status, void* ws, size_t ws_bytes, void* stream);
/* test hook: at most cap x (rounds / 8) proposal rounds per call (<= 0: no cap).  With cap = 1 the default 8 rounds
 * become 1 and every resume of a caller's repair loop (rounds doubled each time) 2, 4, ...: the resume paths of
 * PoolingLayer.forward / geobi_net_forward run on graphs that otherwise converge at once. */
int geobi_set_match_round_cap(int cap);
/* geobi_match_coarsen: the integer front end of one pooling step (code/net_util.py:127-128 plus the inverse
 * lists the feature pooling needs) in one call: the rounds of geobi_match_heavy_edge, then dense ids and
 * segment lists of the matching -- the results of geobi_relabel_compact and geobi_segment_csr_pairs, with
 * 6 launches fewer.  state [N]: resumable (-1 undecided, u singleton, v partner); cluster_final [N]: graclus
 * ids (min member; undecided nodes closed as singletons); cnew [N]; segptr [N+1], members [N] (sized by the
 * fine node count); counters[0] = undecided nodes, counters[1] = coarse node count (device int32).        */
size_t geobi_match_coarsen_ws_bytes(int64_t N);
int geobi_match_coarsen(const int32_t* rowptr, const int32_t* col, const float* w, int64_t N, int rounds, int init,
                        int32_t* state, int32_t* cluster_final, int32_t* cnew, int32_t* segptr, int32_t* members,
                        int32_t* counters, void* ws, size_t ws_bytes, void* stream);
size_t geobi_relabel_ws_bytes(int64_t N);
int geobi_relabel_compact(const int32_t* cluster, int64_t N, int rep_is_self, int32_t* cnew, int32_t* count,
                          void* ws, size_t ws_bytes, void* stream);
size_t geobi_segment_csr_ws_bytes(int64_t n);
int geobi_segment_csr(const int32_t* seg, int64_t n, int64_t nseg, int32_t* segptr, int32_t* members, void* ws,
                      size_t ws_bytes, void* stream);
/* Sort-free inverse lists: _pairs for a matching (clusters of <= 2 nodes, raw id = smaller member, as
 * graclus / geobi_match_heavy_edge emit, cnew = its dense relabelling); _compose for the lists of a
 * composed index fine -> mid -> coarse (the `clust2[clust1]` of code/net_util.py:152-156). */
size_t geobi_segment_pairs_ws_bytes(int64_t nseg);
int geobi_segment_csr_pairs(const int32_t* cnew, const int32_t* raw, int64_t N, int64_t nseg, int32_t* segptr,
                            int32_t* members, void* ws, size_t ws_bytes, void* stream);
int geobi_segment_csr_compose(const int32_t* segptr1, const int32_t* members1, const int32_t* segptr2,
                              const int32_t* members2, int64_t nseg2, int64_t n_fine, int32_t* segptr12,
                              int32_t* members12, void* ws, size_t ws_bytes, void* stream);
int geobi_segment_max_fwd(const float* x, int C, const int32_t* segptr, const int32_t* members, int64_t nseg,
                          float* out, int32_t* arg, void* stream);
int geobi_segment_max_bwd(const float* gout, const int32_t* arg, const int32_t* seg, int C, int64_t nseg,
                          int64_t n_fine, float* gx, void* stream);
int geobi_segment_sum(const float* x, int C, const int32_t* segptr, const int32_t* members, int64_t nseg, int mean,
                      float* out, void* stream);
int geobi_segment_mean_bwd(const float* gout, const int32_t* seg, const int32_t* segptr, int C, int64_t n_fine,
                           float* gx, void* stream);
int geobi_gather_rows(const float* x, const int32_t* idx, int C, int64_t n_out, float* out, void* stream);
/* Sort-free pool_edge for a MATCHING: one wave per coarse node merges the (at most two) fine rows of its
 * members with a 64-lane bitonic network.  (segptr, members) = geobi_segment_csr_pairs built with
 * nseg = nbound = fine node count; ncount = device count of coarse nodes (geobi_relabel_compact).
 * overflow[0] |= 1 when a coarse node gathers more than 64 fine entries: use geobi_pool_edge then. */
size_t geobi_pool_edge_rows_ws_bytes(int64_t nbound);
int geobi_pool_edge_rows(const int32_t* cnew, const int32_t* segptr, const int32_t* members, const int32_t* rowptr,
                         const int32_t* col, const float* w, const int32_t* ncount, int64_t nbound,
                         int32_t* rowptr_c, int32_t* row_c, int32_t* col_c, float* w_c, int32_t* count,
                         int32_t* overflow, void* ws, size_t ws_bytes, void* stream);
size_t geobi_pool_edge_ws_bytes(int64_t E);
int geobi_pool_edge(const int32_t* cnew, const int32_t* row, const int32_t* col, const float* w, int64_t E,
                    int64_t nmax, int32_t* rowptr_c, int32_t* row_c, int32_t* col_c, float* w_c, int32_t* count,
                    void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------- geometry coupling --------
 * DualGNN.forward, code/network.py:335-337 + data_util.computer_face_normal (code/data_util.py:182-198):
 * out[f] = (xf[f, 0:6], centroid of the predicted triangle, its unit normal).  The backward emits
 * per-corner gradients [3F, 3] which geobi_segment_sum folds into the vertices through the
 * vertex -> corner inverse lists.                                                               */
int geobi_face_geom_fwd(const float* verts, const int32_t* fv, const float* xf, int ldxf, int64_t F, float* out,
                        void* stream);
int geobi_face_geom_bwd(const float* verts, const int32_t* fv, const float* gout, int64_t F, float* corner_grad,
                        void* stream);

/* ---------------------------------------------------------------- heads --------------------
 * code/network.py:324-332 (vertex head, mode 0: fc2(lrelu(fc1 x)) (* depth_direction) + xyz) and
 * :340-343 (face head, mode 1: F.normalize(fc2(lrelu(fc1 x)), dim=1)).  raw [N,nout] is saved for the
 * backward.  h = NULL selects the fused kernels (Cin = 32, K = 1024): the [N, K] hidden activation
 * stays in the matrix-core accumulators, forward and backward (recomputed), and never reaches HBM;
 * with a non-NULL h [N,K] the hidden activation is materialised and the generic GEMMs are used.   */
int geobi_head_fwd(const float* x, int Cin, int64_t N, const float* w1, const float* b1, int K, const float* w2,
                   const float* b2, int nout, float slope, int mode, const float* dd, const float* resid,
                   int ld_resid, float* h, float* raw, float* out, void* stream);
/* Precision of the heads' 32 -> 1024 product (process-wide switch, default 0; GEOBI_HEAD_BF16X3=1 starts with 1):
 *   0  exact fp32 on v_mfma_f32_32x32x2_f32 -- what every number of the bench line and every parity test uses
 *   1  the same product as SIX bf16 products with fp32 accumulation (each operand cut into three bf16 pieces: 24 significand
 *      bits; v_mfma_f32_32x32x16_bf16, 16 x the fp32 matrix rate): not bit-identical to mode 0, but no farther from fp64 than
 *      mode 0 is on the path's operands (profiles/r04_bf16_split_study.txt, test_head_split_precision_*).  A labelled variant:
 *      the reference computes in fp32 (code/network.py:324-343).                                                          */
int geobi_set_head_precision(int mode);
size_t geobi_head_bwd_ws_bytes(int64_t N, int Cin, int K);
int geobi_head_bwd(const float* x, int Cin, int64_t N, const float* w1, const float* b1, int K, const float* w2,
                   int nout, float slope, int mode, const float* dd, const float* h, const float* raw, const float* gout,
                   float* dx, float* dw1, float* db1, float* dw2, float* db2, int accumulate, void* ws,
                   size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------- losses / metrics ---------
 * code/network.py:364-413 on [n, 3] rows:  out[0] = scale * sum_i w_i * term_i  (w = NULL: w_i = 1)
 *   kind 0  L1   sum_c |a - b|          (loss_v / loss_n 'L1')       kind 1  L2  sum_c (a - b)^2
 *   kind 2  Euclidean distance          (error_v)                    kind 3  angle in degrees (error_n)
 * scale = 1/n gives the reference's `.mean()`; per-row weights give per-mesh means of a batch.
 * Deterministic two-stage reduction.  _bwd (kinds 0, 1): ga = gout[0] * scale * w_i * d term / d a.  */
size_t geobi_row_loss_ws_bytes(int64_t n);
int geobi_row_loss_fwd(const float* a, const float* b, const float* w, int64_t n, int kind, float scale, float* out,
                       void* ws, size_t ws_bytes, void* stream);
int geobi_row_loss_bwd(const float* a, const float* b, const float* w, const float* gout, int64_t n, int kind,
                       float scale, float* ga, void* stream);

/* ---------------------------------------------------------------- optimiser step ----
 * torch.optim.Adam's update rule (code/train_dual.py:162, the reference's default optimiser; no amsgrad) over one flat
 * fp32 vector of n parameters, its gradient and the two moment vectors (16-byte aligned), one launch:
 *   g += weight_decay p;  m += (1 - beta1)(g - m);  v = beta2 v + (1 - beta2) g g;
 *   p -= lr / bias_corr1 * m / (sqrt(v) / sqrt(bias_corr2) + eps),      bias_corr_k = 1 - beta_k^step (host-side).  */
int geobi_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                    float weight_decay, float bias_corr1, float bias_corr2, void* stream);

/* ---------------------------------------------------------------- vertex update ----
 * data_util.update_position2 (code/data_util.py:529-556; called at code/test_dual.py:63-72 after the
 * network): n_iter Jacobi sweeps  p_v += mean_{f adj v} n_f (n_f . (c_f - p_v)), c_f = face centroid,
 * vf = padded vertex->face table [V, maxval] with -1 fill, optional projection on depth_direction.   */
size_t geobi_update_position_ws_bytes(int64_t V, int64_t F);
int geobi_update_position2(const float* points, const int32_t* fv, const int32_t* vf, int maxval,
                           const float* normals, const float* dd, int64_t V, int64_t F, int n_iter, float* out,
                           void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------- mesh -> graphs ----
 * What the reference's loader computes on the CPU with openmesh / PyG before the network runs
 * (code/dataset.py:197-233 process_one_submesh).  fv = face->vertex table [F,3] int32.
 *   geobi_vertex_faces   vertex->face incidence as CSR (rowptr [V+1], list [3F], face ids ascending):
 *                        openmesh vf_indices (dataset.py:203); geobi_vf_padded emits the [V, maxval]
 *                        -1 padded table update_position2 takes; geobi_max_degree -> maxval (device int).
 *   geobi_mesh_normals   unit face normals + face centroids (dataset.py:222-224) and vertex normals =
 *                        normalised sum of incident face normals (openmesh update_vertex_normals,
 *                        dataset.py:198-199); fp64 inside, fp32 out.  vnormal may be NULL.
 *   geobi_ring_graph_*   kind 0: vertex graph = mesh edges in both directions (to_undirected(ev.T),
 *                        dataset.py:211-212); kind 1: facet graph = faces sharing >= 1 vertex
 *                        (data_util.build_facet_graph, code/data_util.py:436-456).  Output is the
 *                        (row, col)-sorted CSR WITHOUT self loops that the conv / pooling entry points
 *                        take (the reference appends / inlines one loop per node; FeaStConv and the
 *                        pooling layer drop them again).  _count writes rowptr_g [n+1] (read
 *                        rowptr_g[n] = E to size col), _fill writes col [E].
 *   geobi_calc_weight    data_util.calc_weight (code/data_util.py:383-398) over the loop-free edges:
 *                        clamp(n_i.n_j, 1e-3) * exp(|p_i-p_j|^2 / (-2 mean|p_i-p_j| + 1e-12)); the mean
 *                        runs over E + extra_zero_edges entries (the reference's edge list carries one
 *                        zero-length self loop per node).  w may be NULL; mean_len (device float,
 *                        optional) receives the mean -- center_and_scale's 1/scale (data_util.py:201-230). */
size_t geobi_vertex_faces_ws_bytes(int64_t F, int64_t V);
int geobi_vertex_faces(const int32_t* fv, int64_t F, int64_t V, int32_t* rowptr, int32_t* list, void* ws,
                       size_t ws_bytes, void* stream);
int geobi_vf_padded(const int32_t* rowptr, const int32_t* list, int64_t V, int maxval, int32_t* vf, void* stream);
int geobi_max_degree(const int32_t* rowptr, int64_t N, int32_t* out, void* stream);
int geobi_mesh_normals(const float* points, const int32_t* fv, int64_t F, int64_t V, const int32_t* rowptr,
                       const int32_t* list, float* fnormal, float* centroid, float* vnormal, void* stream);
size_t geobi_ring_graph_ws_bytes(int64_t n_nodes);
int geobi_ring_graph_count(int kind, const int32_t* fv, const int32_t* rowptr_vf, const int32_t* list,
                           int64_t n_nodes, int32_t* rowptr_g, void* ws, size_t ws_bytes, void* stream);
int geobi_ring_graph_fill(int kind, const int32_t* fv, const int32_t* rowptr_vf, const int32_t* list,
                          int64_t n_nodes, const int32_t* rowptr_g, int32_t* col, void* stream);
/* geobi_calc_weight_parts: geobi_calc_weight over a disjoint union of meshes (the patches of one network pass): node_ptr
 * [n_parts + 1] (device) cuts the nodes into parts, a part's edges are the CSR rows of its nodes, and every part gets its
 * OWN mean edge length (calc_weight normalises by the mean of the edges it is handed, code/data_util.py:383-398, and the
 * reference hands it one patch at a time) -- formed in the single-mesh kernel's order, so the weights of a part are
 * bit-identical to what the part alone gives.  The mean's denominator counts one zero-length self loop per node.        */
size_t geobi_calc_weight_parts_ws_bytes(int n_parts);
int geobi_calc_weight_parts(const float* pos, const float* normal, const int32_t* rowptr, const int32_t* row,
                            const int32_t* col, int64_t E, const int32_t* node_ptr, int n_parts, float* w, void* ws,
                            size_t ws_bytes, void* stream);
size_t geobi_calc_weight_ws_bytes(void);
int geobi_calc_weight(const float* pos, const float* normal, const int32_t* row, const int32_t* col, int64_t E,
                      int64_t extra_zero_edges, float* w, float* mean_len, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------- patch split / merge ----
 * Meshes with more faces than one pass takes are cut into overlapping patches, denoised patch by patch
 * and merged (code/dataset.py:156-193, code/test_dual.py:49-61).
 *   geobi_patch_grow       data_util.mesh_get_neighbor_np (code/data_util.py:55-84) on the device: face-ring growth from
 *                          a seed until neighbor_count faces (<= 0: unlimited) or ring_count rings (<= 0: unlimited) are
 *                          listed, in the reference's visiting order (ring faces in order, their vertices in order, each
 *                          vertex's incident faces in the order of its row of `vf`, the padded [V, maxval] incidence table
 *                          of geobi_vf_padded) -- one launch of one workgroup per patch, ring by ring; a
 *                          vertex is expanded where it is first met, a face met by several expanding (face, vertex,
 *                          incidence) slots of a ring is listed where the smallest slot meets it.
 *                          `state`: geobi_patch_grow_state_ints(F, V) int32 set up by geobi_patch_grow_init (which also picks
 *                          the first seed: the face with the largest d2, lowest id on ties -- code/dataset.py:159); it
 *                          carries the visited flags from patch to patch.  seed < 0: the seed the previous call picked
 *                          (pick_next != 0: the unvisited face with the largest d2, code/dataset.py:186-190), so a CHAIN
 *                          of launches splits a mesh with no host step in between; patch_id: 1, 2, ... (distinct per
 *                          patch of one split).  sel_out: room for min(neighbor_count, F) ids; n_out (device, optional)
 *                          and mailbox (mapped host int32, optional: 1 + 2 n, written last) receive the face count; 0
 *                          faces = every face had been visited.
 *   geobi_submesh          data_util.get_submesh (code/data_util.py:318-336) on the device: sel [n_sel]
 *                          face ids -> v_idx (original vertex ids in first-use order; capacity
 *                          min(V, 3 n_sel)), f_sub [n_sel,3] renumbered faces, count (device int) = number
 *                          of patch vertices.
 *   geobi_patch_accumulate Vp[v_idx] += vert_p, sum_v[v_idx] += 1, Np[f_idx] += norm_p
 *   geobi_patch_finalize   Vp = Vp / sum_v / scale + centroid; Np = normalize(Np) (eps 1e-12)            */
size_t geobi_patch_grow_state_ints(int64_t F, int64_t V);
int geobi_patch_grow_init(int32_t* state, int64_t F, int64_t V, const float* d2, void* stream);
int geobi_patch_grow(const int32_t* fv, const int32_t* vf, int maxval, int64_t F, int64_t V, const float* d2, int64_t seed,
                     int64_t neighbor_count, int64_t ring_count, int patch_id, int32_t* state, int32_t* sel_out,
                     int32_t* n_out, int32_t* mailbox, int pick_next, void* stream);
/* n zeroed int32 slots of mapped HOST memory that a kernel can write (the patch sizes above); per host thread, valid until
 * that thread asks for more slots */
int geobi_host_mailbox(int n, int32_t** host_ptr);
/* waits (spinning) until *word != 0 and returns it; 0 when `stream` -- the one the writing kernel runs on -- drained first */
int geobi_host_mailbox_wait(const int32_t* word, void* stream);
size_t geobi_submesh_ws_bytes(int64_t n_sel, int64_t V);
int geobi_submesh(const int32_t* fv, const int32_t* sel, int64_t n_sel, int64_t V, int32_t* v_idx, int32_t* f_sub,
                  int32_t* count, void* ws, size_t ws_bytes, void* stream);
int geobi_patch_accumulate(const float* vert_p, const float* norm_p, const int32_t* v_idx, const int32_t* f_idx,
                           int64_t nv, int64_t nf, float* Vp, float* Np, int32_t* sum_v, void* stream);
int geobi_patch_finalize(float* Vp, float* Np, const int32_t* sum_v, int64_t V, int64_t F, float scale, float cx,
                         float cy, float cz, void* stream);

/* ---------------------------------------------------------------- dense helpers ------------
 * Plain fp32 MFMA GEMMs used by the layers above, exported for tests and profiling.            */
int geobi_gemm_nn(const float* A, int lda, const float* B, int ldb, int transB, float* C, int ldc, int M, int N,
                  int K, const float* bias, float slope, void* stream);
size_t geobi_gemm_tn_ws_bytes(int I, int J, int64_t M);
int geobi_gemm_tn(const float* A, int lda, const float* B, int ldb, int64_t M, int I, int J, float* C, int ldc,
                  void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------- size read-back ------------
 * The ONE entry point that waits for the device: copies n (<= 64) int32 from device memory to `host`
 * after everything enqueued on `stream`, polling (not sleeping) until the copy has landed.  Used for the
 * data-dependent sizes of a pooling step (coarse node / edge counts; the reference's `unique` /
 * `numel() == 0` syncs, code/net_util.py:128,139).                                                  */
int geobi_read_i32(const int32_t* dev, int n, int32_t* host, void* stream);

/* ---------------------------------------------------------------- whole-network forward ----
 * The complete inference pass of code/network.py:318-343 (DualGNN.forward: vertex branch -> vertex head ->
 * geometry coupling -> facet branch -> normal head) as ONE call: the op sequence of GNNModule.forward
 * (code/network.py:270-300) and PoolingLayer.forward (code/net_util.py:76-158, edge_weight_type 10, two matching
 * steps, max or mean pooling) is driven by native host code instead of ~180 Python-level launches.  Same kernels,
 * same order, bit-identical results to the module-by-module path.
 *   arena : caller-provided device scratch; every intermediate (features, coarse graphs, cluster vectors) is
 *           bump-allocated in it and stays valid until the caller releases it.  geobi_net_forward_arena_bytes is a
 *           sufficient size for a mesh of the given level-0 sizes.
 *   graphs: loop-free, (row, col)-sorted, SYMMETRIC CSR of level 0 with the static edge weights in CSR order.
 *   out   : byte offsets into the arena of the results and of what the module surface exposes afterwards
 *           (PoolingLayer.unpooling_indices, the raw cluster vector of every matching step).
 * Host synchronisation: one read of the pooling sizes per pooling layer (4 per call).
 * Returns 0, or GEOBI_NET_FALLBACK (2) when a case outside the fast path turned up (edge-free level, a matching
 * that needs more rounds, a coarse row beyond the sort-free width) -- the caller then runs the module-by-module
 * path --, or GEOBI_NET_ARENA (3) when the arena is too small (out->used_bytes = bytes needed so far).        */
#define GEOBI_NET_FALLBACK 2
#define GEOBI_NET_ARENA 3
typedef struct { const float *lin_w, *u_w, *c, *bias; } geobi_conv_params_t;
typedef struct { geobi_conv_params_t conv[8]; } geobi_gnn_params_t;        /* l_conv1..4, r_conv1..4 */
typedef struct {
  geobi_gnn_params_t gnn_v, gnn_f;
  const float *fc_v1_w, *fc_v1_b, *fc_v2_w, *fc_v2_b, *fc_f1_w, *fc_f1_b, *fc_f2_w, *fc_f2_b;
  int32_t force_depth, pool_mean;
} geobi_net_params_t;
typedef struct {
  int64_t N, E;
  const int32_t *rowptr, *col, *row;
  const float* weight;
} geobi_level0_t;
typedef struct {
  int64_t nodes[3];               /* node count of levels 0, 1, 2 */
  int64_t unpool_off[2];          /* int32 [nodes[l]]: composed fine -> coarse index of pooling layer l + 1 */
  int64_t cluster_off[2][2];      /* int32 raw (graclus-style) cluster vector of [layer][step] */
  int64_t cluster_len[2][2];
} geobi_branch_out_t;
typedef struct {
  int64_t verts_off, normals_off, xf_off;     /* float [V,3], [F,3], [F,12] */
  int64_t used_bytes;
  geobi_branch_out_t v, f;
} geobi_net_out_t;
size_t geobi_net_forward_arena_bytes(int64_t V, int64_t Ev, int64_t F, int64_t Ef);
int geobi_net_forward(const geobi_net_params_t* prm, const geobi_level0_t* gv, const geobi_level0_t* gf,
                      const float* x_v, const float* x_f, const int32_t* fv, const float* depth_direction,
                      void* arena, size_t arena_bytes, geobi_net_out_t* out, void* stream);

/* Training through the same executor: geobi_net_forward_train additionally keeps what the backward needs in the arena
 * (size: geobi_net_train_arena_bytes) and returns a host-side record of it as *handle; geobi_net_backward replays it
 * in reverse -- the per-op backward passes of the module-by-module path, same kernels, same order -- and writes the
 * 72 parameter gradients through `grads` (same layout as the parameter struct; accumulate != 0 adds, the semantics of
 * autograd's accumulation into .grad, code/train_dual.py:211-218).  pos_rev_*: position of each level-0 edge's reverse
 * edge (geobi_csr_reverse_index).  corner_segptr / corner_members: vertex -> corner inverse lists of the face table
 * (geobi_segment_csr over fv viewed as [3F]).  The arena must stay untouched from the forward to the end of the
 * backward; geobi_net_release frees the host-side record (after the backward, or instead of it).                  */
size_t geobi_net_train_arena_bytes(int64_t V, int64_t Ev, int64_t F, int64_t Ef);
int geobi_net_forward_train(const geobi_net_params_t* prm, const geobi_level0_t* gv, const geobi_level0_t* gf,
                            const int32_t* pos_rev_v, const int32_t* pos_rev_f, const float* x_v, const float* x_f,
                            const int32_t* fv, const float* depth_direction, void* arena, size_t arena_bytes,
                            geobi_net_out_t* out, int64_t* handle, void* stream);
int geobi_net_backward(int64_t handle, const float* g_verts, const float* g_normals, const geobi_net_params_t* grads,
                       int accumulate, const int32_t* corner_segptr, const int32_t* corner_members, void* stream);
int geobi_net_release(int64_t handle);
/* Data-parallel overlap hook: the NEXT geobi_net_backward of the calling host thread records ev_main (on `stream`) and
 * ev_side (on the library's side stream, where that backward's weight-gradient products run; NULL: not wanted) at the
 * point where every gradient of gnn_f / fc_f1 / fc_f2 has been enqueued -- the facet branch's backward runs first.  A
 * caller that waits for both events on its communication stream can all-reduce that half of the gradient bucket under
 * the vertex branch's backward (hipEvent_t handles passed as void*; one-shot: cleared once recorded).              */
int geobi_net_backward_facet_events(void* ev_main, void* ev_side);
/* how often a pooling-size wait ran into its spin cap and fell back to the blocking read (diagnostic; 0 on a healthy run) */
int geobi_net_spin_cap_hits(void);

/* Several mesh groups of one optimiser step in flight together.  The reference walks the meshes of a batch one after
 * another (code/train_dual.py:199-218: forward, loss / batch_size, backward; optimiser step every batch_size meshes);
 * meshes are independent, so the iterations may overlap.  Each group (a disjoint-union graph of one or more meshes) is a
 * complete forward -> loss -> backward pipeline -- geobi_net_forward_train, the losses of code/network.py:364-396
 * (loss kinds 0 = L1, 1 = L2) and geobi_net_backward -- on its OWN stream with its OWN arena and gradient buffers, driven by
 * its own host thread inside the library (group 0 by the calling thread), so that one group's pooling chains and size
 * reads run under the other groups' FeaSt kernels.  Per-context state of the library (side streams, events, scan state)
 * is per host thread, so the groups share nothing but the read-only parameters.
 *   w_v / w_f       per-row loss weights (a union of meshes: 1 / (meshes in the group * rows of the row's mesh)); NULL: 1 / rows
 *   scale_v/scale_n factor on the group's two losses: dual_loss's v_scale / n_scale times (meshes in the group / meshes in
 *                   the step), so that the group losses ADD UP to the step's loss and the gradients to its gradient
 *   grads           where the group's 72 parameter gradients are ADDED; when they are views of one flat buffer, name it in
 *                   grad_flat / grad_count and the call zeroes it first (on the group's stream)
 *   losses          device float[2]: the group's share of loss_v and loss_n
 *   out, rc, error  filled per group (out as geobi_net_forward_train; out.used_bytes = the arena bytes this group needs)
 * Ordering: every group stream first waits for what `main_stream` holds at the call; on return `main_stream` waits for
 * every group, and, with sum_into != NULL, sum_into[i] = grad_flat_0[i] + grad_flat_1[i] + ... (fixed order: the step is
 * bit-reproducible) is enqueued on it.  The call returns when everything is ENQUEUED.  Return: 0, or the code of the first
 * failed group (GEOBI_NET_FALLBACK / GEOBI_NET_ARENA as for geobi_net_forward_train; the caller repeats the whole call). */
#define GEOBI_MAX_GROUPS 8
typedef struct {
  const geobi_level0_t *gv, *gf;
  const int32_t *pos_rev_v, *pos_rev_f;
  const float *x_v, *x_f;
  const int32_t* fv;
  const float* depth_direction;
  const float *y_v, *y_f;
  const float *w_v, *w_f;
  float scale_v, scale_n;
  const int32_t *corner_segptr, *corner_members;
  void* arena;
  size_t arena_bytes;
  geobi_net_params_t grads;
  float* grad_flat;
  int64_t grad_count;
  float* losses;
  void* stream;
  geobi_net_out_t out;
  int32_t rc;
  char error[252];
} geobi_train_group_t;
/* sizeof of a struct of this header as the library was compiled (which: 0 geobi_net_params_t, 1 geobi_level0_t,
 * 2 geobi_net_out_t, 3 geobi_train_group_t, 4 geobi_copy_seg_t; else 0): a binding checks its mirror against it */
size_t geobi_abi_sizeof(int which);
int geobi_net_train_groups(const geobi_net_params_t* prm, geobi_train_group_t* groups, int n_groups, int loss_kind_v,
                           int loss_kind_n, float* sum_into, int64_t sum_count, void* main_stream);

/* ---------------------------------------------------------------- concurrency --------------
 * Weight-gradient GEMMs are off the critical path of a backward call; by default they run on a
 * library-owned non-blocking HIP stream, forked from and joined back into `stream` INSIDE the call
 * (event wait on both ends), so the caller's stream semantics are unchanged.  0 disables it.     */
int geobi_set_overlap(int enable);
/* Deferred join (experimental, off by default): with geobi_side_defer(1) a backward call still forks the
 * side stream from `stream` but returns WITHOUT joining it.  The caller then owns the hazard: every buffer
 * handed to those calls must stay allocated and untouched until geobi_side_join(stream) has been enqueued
 * (it makes `stream` wait for everything on the side stream).                                          */
int geobi_side_defer(int on);
int geobi_side_join(void* stream);

/* ---------------------------------------------------------------- measurement --------------
 * When enabled, the selected kernel family is bracketed with HIP events on its launch stream
 * (kernel: 1 = FeaSt aggregation forward, 2 = transposed aggregation backward, 3 = backward row
 * pass, 4 = the dense GEMMs).  geobi_prof_collect synchronises the recorded events and returns the
 * launch count, the summed device time (ms) and the summed ALGORITHMIC bytes
 * of launches whose channel count equals `tag` (tag = 0: all).  For kernel 4 the third figure is the
 * flop count 2*M*N*K instead and the tag is 1 for gemm_nn (+ its split-K reduce), 2 for gemm_tn
 * (+ its slab reduce).                                                                            */
int geobi_prof_enable(int kernel);
/* diagnostic: y[i] = the kernels' own exp for the softmax (csrc/feast_dev.h: exp_le0, six instructions; arguments are
 * differences to the row maximum, i.e. FINITE and <= 0: <= 2 ulp against exp, results below 2^-126 flush to 0; a
 * non-finite argument is outside its contract).  Exposed so that a test can check it in isolation.              */
int geobi_debug_exp_le0(const float* x, float* y, int64_t n, void* stream);
int geobi_prof_collect(int tag, int64_t* launches, double* total_ms, double* total_bytes);

#ifdef __cplusplus
}
#endif
#endif /* GEOBI_HIP_H_ */
