// Graph pooling / unpooling kernels (PoolingLayer of the hot path,
// /root/reference/code/net_util.py:56-245, and pool_edge :289-295).
//
//   edge_weight_t10   w_e += exp(-|x_i - x_j|^2 / 2)                       (net_util.py:226-230)
//   match_*           heavy-edge matching, cluster id = min(u, v)          (graclus, net_util.py:127)
//   relabel           dense ids by rank of the representative              (consecutive_cluster, :128)
//   segment_csr       inverse lists  cluster -> members  (sorted, deterministic)
//   segment_max/mean  feature pooling with arg-max for the backward        (scatter, :131-134)
//   segment_sum       unpool backward (gather forward = `x[unpooling_indices]`, :242-245)
//   pool_edge         relabel endpoints, drop loops, sort, merge duplicates by mean  (:289-295)
//
// Matching specification (deterministic, unlike torch_cluster's randomised orders): an edge is
// taken iff it is the best remaining edge at BOTH endpoints under the total order
// (weight desc, min(u,v) asc, max(u,v) asc) -- i.e. greedy matching in globally descending edge
// order, computed by rounds of mutual proposals.  oracle/oracle_c.c:oracle_greedy_sorted is the
// sequential statement of the same rule.
#include "common.h"

#include <cstdlib>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace geobi {

namespace {

constexpr uint64_t kSentinel = ~0ull;

static inline int key_bits(int64_t N) {   // (1 << bits) > N: see graph.hip
  int b = 1;
  while ((1ll << b) <= N) ++b;
  return b;
}

// ------------------------------------------------------------------------ edge weights
// 8 lanes per edge, float4 per lane per pass.
__global__ __launch_bounds__(256) void edge_weight_t10_kernel(const float* __restrict__ x, int C,
                                                              const int* __restrict__ row,
                                                              const int* __restrict__ col,
                                                              const float* __restrict__ w_in, int64_t E,
                                                              float* __restrict__ w_out, int* __restrict__ zero8) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (zero8 != nullptr && t < 8) zero8[t] = 0;     // the pooling layer's counters, cleared on the way (one fill launch less)
  int64_t e = t >> 3;
  int l = (int)(t & 7);
  bool ok = e < E;
  int i = ok ? row[e] : 0, j = ok ? col[e] : 0;
  float d = 0.f;
  if (ok) {
    const float* xi = x + (size_t)i * C;
    const float* xj = x + (size_t)j * C;
    if ((C & 3) == 0) {
      for (int c = l * 4; c < C; c += 32) {
        float4 a = *reinterpret_cast<const float4*>(xi + c);
        float4 b = *reinterpret_cast<const float4*>(xj + c);
        float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z, dw = a.w - b.w;
        d = fmaf(dx, dx, d); d = fmaf(dy, dy, d); d = fmaf(dz, dz, d); d = fmaf(dw, dw, d);
      }
    } else {
      for (int c = l; c < C; c += 8) {
        float dd = xi[c] - xj[c];
        d = fmaf(dd, dd, d);
      }
    }
  }
  d += __shfl_xor(d, 1, 64);
  d += __shfl_xor(d, 2, 64);
  d += __shfl_xor(d, 4, 64);
  if (ok && l == 0) w_out[e] = (w_in ? w_in[e] : 0.f) + expf(d * -0.5f);
}

// Learned edge weights, PoolingLayer edge_weight_type 3 / 4 / 5 (/root/reference/code/net_util.py:182-206, GAT-style):
// per node al = x . att_l, ar = x . att_r (8 lanes per node, float4 per lane and pass), per edge
// sigmoid((al[r] + ar[c]) + (al[c] + ar[r])) in the reference's order of additions; type 5 averages with the given weight.
__global__ __launch_bounds__(256) void node_att_kernel(const float* __restrict__ x, int C, const float* __restrict__ att_l,
                                                       const float* __restrict__ att_r, int64_t N,
                                                       float2* __restrict__ alr) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i = t >> 3;
  const int l = (int)(t & 7);
  const bool ok = i < N;
  float a = 0.f, b = 0.f;
  if (ok) {
    const float* xi = x + (size_t)i * C;
    if ((C & 3) == 0) {
      for (int c = l * 4; c < C; c += 32) {
        const float4 v = *reinterpret_cast<const float4*>(xi + c);
        const float4 p = *reinterpret_cast<const float4*>(att_l + c);
        const float4 q = *reinterpret_cast<const float4*>(att_r + c);
        a = fmaf(v.x, p.x, a); a = fmaf(v.y, p.y, a); a = fmaf(v.z, p.z, a); a = fmaf(v.w, p.w, a);
        b = fmaf(v.x, q.x, b); b = fmaf(v.y, q.y, b); b = fmaf(v.z, q.z, b); b = fmaf(v.w, q.w, b);
      }
    } else {
      for (int c = l; c < C; c += 8) {
        a = fmaf(xi[c], att_l[c], a);
        b = fmaf(xi[c], att_r[c], b);
      }
    }
  }
#pragma unroll
  for (int m = 1; m < 8; m <<= 1) {
    a += __shfl_xor(a, m, 64);
    b += __shfl_xor(b, m, 64);
  }
  if (ok && l == 0) alr[i] = make_float2(a, b);
}

__global__ __launch_bounds__(256) void edge_weight_att_kernel(const float2* __restrict__ alr, const int* __restrict__ row,
                                                              const int* __restrict__ col, const float* __restrict__ w_in,
                                                              int64_t E, float* __restrict__ w_out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const float2 r = alr[row[e]], c = alr[col[e]];
  const float alpha = (r.x + c.y) + (c.x + r.y);
  const float sg = 1.0f / (1.0f + expf(-alpha));
  w_out[e] = w_in != nullptr ? (sg + w_in[e]) * 0.5f : sg;
}

// ---------------------------------------------------------------------------- matching
__device__ __forceinline__ bool edge_better(float w1, int a1, int b1, float w2, int a2, int b2) {
  // (a, b) = (min, max) endpoint ids
  if (w1 != w2) return w1 > w2;
  if (a1 != a2) return a1 < a2;
  return b1 < b2;
}

// One kernel per round: node u first commits the outcome of the PREVIOUS round from the proposal
// array alone (a pair is matched iff the proposals are mutual), then, if still free, proposes to its
// best free neighbour.  Whether a neighbour w is free is derived the same way (cluster[w] may or may
// not have been committed yet by w's own thread -- both views agree), so no second "resolve" launch
// and no grid-wide barrier is needed between the two halves of a round.
//   prop == -2 : no proposal information (before round 0)      prop == -1 : no free neighbour
// FIRST != 0: round 0 of a call -- every proposal is "no information" (-2) and, when the call starts from
// scratch (FIRST == 1), every node is undecided: nothing is read for it, and the round initialises the state
// itself (no separate init launch).
// A round is a chain of dependent loads per node, not bandwidth (9-11 us for 12 k nodes as for 82 k): the loads are
// therefore issued in four steps, each step's loads independent of one another --
//   A: own state, own proposal, row range      B: partner's proposal, up to MB neighbour ids + weights
//   C: the neighbours' states and proposals    D: the proposals of the neighbours' targets
// -- speculatively (a node that turns out to be decided has loaded a few words too many), instead of walking the
// neighbours one by one with four dependent loads each.
template <int FIRST>
__global__ __launch_bounds__(256) void match_round_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                                   const float* __restrict__ w, const int* __restrict__ prop_prev, int N,
                                   int* __restrict__ cluster, int* __restrict__ prop_next, int* __restrict__ status) {
  constexpr int MB = 4;                  // neighbours per batch (8 and 4 measure alike, 16 is slower)
  int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (FIRST != 0 && u == 0) *status = 0;
  if (u >= N) return;
  // ---- step A
  const int rs = rowptr[u], re = rowptr[u + 1];
  int cu = -1, pv = -2;
  if (FIRST != 1) cu = cluster[u];
  if (FIRST == 0) pv = prop_prev[u];
  // ---- step B (speculative: issued before the node's own outcome is known)
  int ppv = -2;
  if (FIRST == 0) ppv = prop_prev[pv >= 0 ? pv : u];
  int vb[MB];
  float wb[MB];
#pragma unroll
  for (int i = 0; i < MB; ++i) {
    const int e = rs + i < re ? rs + i : (re > rs ? re - 1 : 0);
    vb[i] = re > rs ? col[e] : u;
    wb[i] = (w && re > rs) ? w[e] : 1.0f;
  }
  if (FIRST == 1) {
    cluster[u] = -1;
  } else {
    if (cu >= 0) { prop_next[u] = -1; return; }
    if (FIRST == 0) {
      if (pv == -1) { cluster[u] = u; prop_next[u] = -1; return; }
      if (pv >= 0 && ppv == u) { cluster[u] = pv; prop_next[u] = -1; return; }    // state = partner
    }
  }
  int best = -1, ba = 0, bb = 0;
  float bw = 0.f;
  for (int e0 = rs; e0 < re; e0 += MB) {
    if (e0 > rs) {                        // rows longer than one batch: the next MB ids and weights
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        const int e = e0 + i < re ? e0 + i : re - 1;
        vb[i] = col[e];
        wb[i] = w ? w[e] : 1.0f;
      }
    }
    // ---- step C
    int cv[MB], pw[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      cv[i] = FIRST == 1 ? -1 : cluster[vb[i]];
      pw[i] = FIRST == 0 ? prop_prev[vb[i]] : -2;
    }
    // ---- step D
    int ppw[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i) ppw[i] = (FIRST == 0 && pw[i] >= 0) ? prop_prev[pw[i]] : -2;
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      if (e0 + i >= re) break;
      const int v = vb[i];
      if (v == u) continue;
      // free = undecided and neither closed as a singleton nor mutually matched by the previous round (derived the
      // same way whether or not v's own thread has committed it yet -- both views agree)
      bool free_v = true;
      if (FIRST != 1) {
        if (cv[i] >= 0) free_v = false;
        else if (FIRST == 0) {
          if (pw[i] == -1) free_v = false;
          else if (pw[i] >= 0 && ppw[i] == v) free_v = false;
        }
      }
      if (!free_v) continue;
      const float we = wb[i];
      const int a = u < v ? u : v, b = u < v ? v : u;
      if (best < 0 || edge_better(we, a, b, bw, ba, bb)) { best = v; bw = we; ba = a; bb = b; }
    }
  }
  prop_next[u] = best;
}

// State of a node between calls (`cluster` of the launchers): -1 undecided, u closed as a singleton,
// v != u matched with partner v.  The cluster id graclus reports is min(u, state).
//
// commit of the last round + count of nodes that are still undecided; `final` (optional) receives the
// clustering with the undecided nodes closed as singletons; flag / sz (optional, both or neither):
// flag[u] = u is the representative (smaller member) of its cluster, sz[u] = its cluster size (0 for non-reps)
__global__ void match_commit_kernel(const int* __restrict__ prop, int N, int* __restrict__ cluster,
                                    int* __restrict__ remaining, int* __restrict__ final, int* __restrict__ flag,
                                    int* __restrict__ sz) {
  int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u == 0 && flag) { flag[N] = 0; sz[N] = 0; }      // scan tails
  if (u >= N) return;
  int st = cluster[u];
  if (st < 0) {
    int v = prop[u];
    if (v == -1) st = u;
    else if (v >= 0 && prop[v] == u) st = v;
    else atomicAdd(remaining, 1);
    if (st >= 0) cluster[u] = st;
  }
  const int fin = st < 0 ? u : (u < st ? u : st);
  if (final) final[u] = fin;
  if (flag) {
    const bool rep = fin == u;
    flag[u] = rep ? 1 : 0;
    sz[u] = rep ? ((st >= 0 && st != u) ? 2 : 1) : 0;
  }
}

// dense ids + inverse lists of the matching from the two scans (rank of a representative, offset of its
// members): cnew[u] = rank[rep(u)]; segment c = [rep, partner]; segptr[nc] = N closes the list
__global__ void match_lists_kernel(const int* __restrict__ state, const int* __restrict__ final,
                                   const int* __restrict__ rank, const int* __restrict__ offs, int N,
                                   int* __restrict__ cnew, int* __restrict__ count, int* __restrict__ segptr,
                                   int* __restrict__ members) {
  int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= N) return;
  const int rep = final[u];
  cnew[u] = rank[rep];
  if (rep == u) {
    const int o = offs[u];
    segptr[rank[u]] = o;
    members[o] = u;
    const int st = state[u];
    if (st >= 0 && st != u) members[o + 1] = st;
  }
  if (u == N - 1) {
    const int nc = rank[N];                            // exclusive scan over N + 1 entries: total at [N]
    *count = nc;
    segptr[nc] = N;
  }
}

// Scan-free variant of (match_commit, exclusive scans, match_lists) for up to 256 * kMaxScanBlocks nodes: the commit
// kernel scans its own 256 (flag, size) pairs in LDS and writes block-local prefixes plus one total per block; the
// list kernel turns the <= kMaxScanBlocks totals into global prefixes in LDS at its start.  Two launches instead of
// three, and no single-workgroup walk over the whole array.
constexpr int kMaxScanBlocks = 4096;

__device__ __forceinline__ int block_exclusive_scan_256(int v, int* s_wave, int& total) {
  // 256 threads = 4 waves; returns the exclusive prefix of v inside the block and the block total
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int t = __shfl_up(inc, d, 64);
    if (lane >= d) inc += t;
  }
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) { const int t = s_wave[w]; off += w < wave ? t : 0; tot += t; }
  __syncthreads();
  total = tot;
  return off + inc - v;
}

__global__ __launch_bounds__(256) void match_commit_scan_kernel(const int* __restrict__ prop, int N,
                                                                int* __restrict__ cluster, int* __restrict__ remaining,
                                                                int* __restrict__ final, int* __restrict__ lrank,
                                                                int* __restrict__ loffs, int* __restrict__ bt_rank,
                                                                int* __restrict__ bt_offs) {
  __shared__ int s_wave[4];
  const int u = blockIdx.x * 256 + threadIdx.x;
  int rep = 0, size = 0;
  if (u < N) {
    int st = cluster[u];
    if (st < 0) {
      int v = prop[u];
      if (v == -1) st = u;
      else if (v >= 0 && prop[v] == u) st = v;
      else atomicAdd(remaining, 1);
      if (st >= 0) cluster[u] = st;
    }
    const int fin = st < 0 ? u : (u < st ? u : st);
    final[u] = fin;
    rep = fin == u ? 1 : 0;
    size = rep ? ((st >= 0 && st != u) ? 2 : 1) : 0;
  }
  int tr, to;
  const int pr = block_exclusive_scan_256(rep, s_wave, tr);
  const int po = block_exclusive_scan_256(size, s_wave, to);
  if (u < N) { lrank[u] = pr; loffs[u] = po; }
  if (threadIdx.x == 0) { bt_rank[blockIdx.x] = tr; bt_offs[blockIdx.x] = to; }
}

__global__ __launch_bounds__(256) void match_lists_scan_kernel(const int* __restrict__ state, const int* __restrict__ final,
                                                               const int* __restrict__ lrank, const int* __restrict__ loffs,
                                                               const int* __restrict__ bt_rank, const int* __restrict__ bt_offs,
                                                               int nblocks, int N, int* __restrict__ cnew,
                                                               int* __restrict__ count, int* __restrict__ segptr,
                                                               int* __restrict__ members, const int* __restrict__ rowptr,
                                                               int4* __restrict__ rowinfo) {
  // exclusive prefixes of the block totals (every block recomputes them: <= 4096 ints, L2-resident)
  __shared__ int s_rank[kMaxScanBlocks], s_offs[kMaxScanBlocks];
  __shared__ int s_wave[4];
  __shared__ int s_total[2];
  int carry_r = 0, carry_o = 0;
  for (int base = 0; base < nblocks; base += 256) {
    const int i = base + threadIdx.x;
    const int vr = i < nblocks ? bt_rank[i] : 0, vo = i < nblocks ? bt_offs[i] : 0;
    int tr, to;
    const int pr = block_exclusive_scan_256(vr, s_wave, tr);
    const int po = block_exclusive_scan_256(vo, s_wave, to);
    if (i < nblocks) { s_rank[i] = carry_r + pr; s_offs[i] = carry_o + po; }
    carry_r += tr; carry_o += to;
  }
  if (threadIdx.x == 0) { s_total[0] = carry_r; s_total[1] = carry_o; }
  __syncthreads();
  const int u = blockIdx.x * 256 + threadIdx.x;
  if (u >= N) return;
  const int rep = final[u];
  cnew[u] = s_rank[rep >> 8] + lrank[rep];
  if (rep == u) {
    const int o = s_offs[u >> 8] + loffs[u];
    const int cid = s_rank[u >> 8] + lrank[u];
    segptr[cid] = o;
    members[o] = u;
    const int st = state[u];
    const bool pair = st >= 0 && st != u;
    if (pair) members[o + 1] = st;
    if (rowinfo != nullptr) {
      // the members' rows of the matched graph, for the edge coarsening behind this call (it would otherwise walk
      // segptr -> members -> rowptr, three dependent loads per coarse node)
      const int r0 = rowptr[u], d0 = rowptr[u + 1] - r0;
      const int r1 = pair ? rowptr[st] : 0, d1 = pair ? rowptr[st + 1] - r1 : 0;
      rowinfo[cid] = make_int4(r0, d0, r1, d1);
    }
  }
  if (u == N - 1) {
    const int nc = s_total[0];
    *count = nc;
    segptr[nc] = N;
  }
}

__global__ void match_finish_kernel(int N, const int* __restrict__ state, int* __restrict__ cluster) {
  int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u < N) { int c = state[u]; cluster[u] = c < 0 ? u : (u < c ? u : c); }
}

// ----------------------------------------------------------------------------- relabel
// clusterings whose ids are member indices with cluster[id] == id (graclus: id = min member): no scatter
__global__ void rep_self_flag_kernel(const int* __restrict__ cluster, int N, int* __restrict__ flag) {
  int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u < N) flag[u] = cluster[u] == u ? 1 : 0;
}

__global__ void rep_flag_kernel(const int* __restrict__ cluster, int N, int* __restrict__ flag) {
  // flag every id that occurs (benign race: all writers store 1); ids must lie in [0, N)
  int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u < N) flag[cluster[u]] = 1;
}

__global__ void relabel_apply_kernel(const int* __restrict__ cluster, const int* __restrict__ flag,
                                     const int* __restrict__ rank, int N, int* __restrict__ cnew,
                                     int* __restrict__ count) {
  int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= N) return;
  cnew[u] = rank[cluster[u]];
  if (u == N - 1) *count = rank[u] + flag[u];
}

// ------------------------------------------------------------------------ inverse lists
__global__ void seg_keys_kernel(const int* __restrict__ seg, int64_t n, int bits, uint64_t* __restrict__ keys) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) keys[i] = ((uint64_t)(uint32_t)seg[i] << bits) | (uint64_t)(uint32_t)i;
}

__global__ void seg_unpack_kernel(const uint64_t* __restrict__ keys, int64_t n, int bits,
                                  int* __restrict__ members) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) members[i] = (int)(uint32_t)(keys[i] & ((1ull << bits) - 1));
}

__global__ void seg_ptr_kernel(const uint64_t* __restrict__ keys, int64_t n, int nseg, int bits,
                               int* __restrict__ segptr) {
  int sidx = blockIdx.x * blockDim.x + threadIdx.x;
  if (sidx > nseg) return;
  uint64_t target = (uint64_t)sidx << bits;
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if (keys[mid] < target) lo = mid + 1; else hi = mid;
  }
  segptr[sidx] = (int)lo;
}

// Inverse lists of a MATCHING (clusters of <= 2 nodes, raw id = smaller member, cnew = dense id):
// no sort needed -- the representative takes slot 0, its partner slot 1 (members stay ascending).
__global__ void pair_count_kernel(const int* __restrict__ cnew, const int* __restrict__ raw, int N, int nseg,
                                  int* __restrict__ cnt) {
  int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u < nseg) cnt[u] = 1;              // pass 1 of 2 (see launcher): every cluster has its representative
  (void)cnew; (void)raw; (void)N;
}

__global__ void pair_mark_kernel(const int* __restrict__ cnew, const int* __restrict__ raw, int N,
                                 int* __restrict__ cnt) {
  int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u < N && raw[u] != u) cnt[cnew[u]] = 2;       // exactly one writer per matched cluster
}

__global__ void pair_fill_kernel(const int* __restrict__ cnew, const int* __restrict__ raw, int N, int nseg,
                                 int* __restrict__ segptr, int* __restrict__ members) {
  int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u == 0) segptr[nseg] = N;
  if (u < N) members[segptr[cnew[u]] + (raw[u] == u ? 0 : 1)] = u;
}

// Inverse lists of a composition  fine --(idx1)--> mid --(idx2)--> coarse :
// members12(C) = concat over m in members2(C) of members1(m)   (fixed order -> deterministic sums)
__global__ void compose_count_kernel(const int* __restrict__ segptr1, const int* __restrict__ segptr2,
                                     const int* __restrict__ members2, int nseg2, int* __restrict__ cnt) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nseg2) return;
  int n = 0;
  for (int e = segptr2[c]; e < segptr2[c + 1]; ++e) { int m = members2[e]; n += segptr1[m + 1] - segptr1[m]; }
  cnt[c] = n;
}

__global__ void compose_fill_kernel(const int* __restrict__ segptr1, const int* __restrict__ members1,
                                    const int* __restrict__ segptr2, const int* __restrict__ members2, int nseg2,
                                    int n_fine, int* __restrict__ segptr12, int* __restrict__ members12) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0) segptr12[nseg2] = n_fine;
  if (c >= nseg2) return;
  int o = segptr12[c];
  for (int e = segptr2[c]; e < segptr2[c + 1]; ++e) {
    int m = members2[e];
    for (int f = segptr1[m]; f < segptr1[m + 1]; ++f) members12[o++] = members1[f];
  }
}

// ----------------------------------------------------------------------- segment reduce
__global__ void segment_max_fwd_kernel(const float* __restrict__ x, int C, const int* __restrict__ segptr,
                                       const int* __restrict__ members, int nseg, float* __restrict__ out,
                                       int* __restrict__ arg) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)nseg * C) return;
  int sidx = (int)(t / C), c = (int)(t % C);
  int rs = segptr[sidx], re = segptr[sidx + 1];
  float best = 0.f;          // empty segment -> 0 (torch_scatter fills untouched rows with 0)
  int bi = -1;
  for (int e = rs; e < re; ++e) {
    int m = members[e];      // members ascend, strict '>' keeps the first maximum
    float v = x[(size_t)m * C + c];
    if (bi < 0 || v > best) { best = v; bi = m; }
  }
  out[t] = best;
  arg[t] = bi;
}

// Max over the composition of two clusterings (a pooling layer's two matching steps) in ONE pass: segment c of the
// second step = concatenation, in order, of the first-step segments of its members.  The running maximum keeps the
// first maximum in that order -- the element the two-step form (max of the step-one maxima, first maximum at each
// step) routes to -- so out and the backward's routing are identical to two segment_max passes; the step-one
// maxima are never materialised.  arg12 holds the FINE row of the maximum.
__global__ void segment_max2_fwd_kernel(const float* __restrict__ x, int C, const int* __restrict__ segptr1,
                                        const int* __restrict__ members1, const int* __restrict__ segptr2,
                                        const int* __restrict__ members2, int nseg2, float* __restrict__ out,
                                        int* __restrict__ arg12) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)nseg2 * C) return;
  const int sidx = (int)(t / C), c = (int)(t % C);
  float best = 0.f;
  int bi = -1;
  for (int e2 = segptr2[sidx]; e2 < segptr2[sidx + 1]; ++e2) {
    const int m = members2[e2];
    float mb = 0.f;                               // the step-one maximum of segment m and its first position
    int mi = -1;
    for (int e1 = segptr1[m]; e1 < segptr1[m + 1]; ++e1) {
      const int f = members1[e1];
      const float v = x[(size_t)f * C + c];
      if (mi < 0 || v > mb) { mb = v; mi = f; }
    }
    if (mi >= 0 && (bi < 0 || mb > best)) { best = mb; bi = mi; }
  }
  out[t] = best;
  arg12[t] = bi;
}

// gather form of its backward: fine row n receives the gradient of its composed segment where it was the arg-max
__global__ void segment_max2_bwd_kernel(const float* __restrict__ gout, const int* __restrict__ arg12,
                                        const int* __restrict__ seg12, int C, int64_t total, int nseg2,
                                        float* __restrict__ gx, int add) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int n = (int)(t / C), c = (int)(t % C);
  const int sg = seg12[n];
  float v = 0.f;
  if (sg >= 0 && sg < nseg2) {
    const size_t o = (size_t)sg * C + c;
    if (arg12[o] == n) v = gout[o];
  }
  gx[t] = add ? gx[t] + v : v;                  // add: on top of what gx holds (a skip connection's gradient)
}

// gather form: fine row n receives the gradient of its segment where it was the arg-max (every
// element of gx is written exactly once -> no zero-fill pass)
__global__ void segment_max_bwd_kernel(const float* __restrict__ gout, const int* __restrict__ arg,
                                       const int* __restrict__ seg, int C, int64_t total, int nseg,
                                       float* __restrict__ gx) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  int n = (int)(t / C), c = (int)(t % C);
  int sg = seg[n];
  float v = 0.f;
  if (sg >= 0 && sg < nseg) {
    size_t o = (size_t)sg * C + c;
    if (arg[o] == n) v = gout[o];
  }
  gx[t] = v;
}

__global__ void segment_sum_kernel(const float* __restrict__ x, int C, const int* __restrict__ segptr,
                                   const int* __restrict__ members, int nseg, int mean, float* __restrict__ out) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)nseg * C) return;
  int sidx = (int)(t / C), c = (int)(t % C);
  int rs = segptr[sidx], re = segptr[sidx + 1];
  float s = 0.f;
  for (int e = rs; e < re; ++e) s += x[(size_t)members[e] * C + c];
  if (mean) s = s / (float)max(re - rs, 1);
  out[t] = s;
}

// Sum over the composition of two clusterings, the members walked in the order the composed list (compose_fill_kernel)
// would hold them -- same sums bit for bit, without building that list.
__global__ void segment_sum2_kernel(const float* __restrict__ x, int C, const int* __restrict__ segptr1,
                                    const int* __restrict__ members1, const int* __restrict__ segptr2,
                                    const int* __restrict__ members2, int nseg2, float* __restrict__ out) {
  // four channels per thread (16-B loads; C is 64 or 128 on the path), same summation order per channel as before
  const int Q = C >> 2;
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)nseg2 * Q) return;
  const int sidx = (int)(t / Q), c = (int)(t % Q) * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int e2 = segptr2[sidx]; e2 < segptr2[sidx + 1]; ++e2) {
    const int m = members2[e2];
    for (int e1 = segptr1[m]; e1 < segptr1[m + 1]; ++e1) {
      const float4 v = *reinterpret_cast<const float4*>(x + (size_t)members1[e1] * C + c);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  *reinterpret_cast<float4*>(out + (size_t)sidx * C + c) = s;
}

__global__ void segment_mean_bwd_kernel(const float* __restrict__ gout, const int* __restrict__ seg,
                                        const int* __restrict__ segptr, int C, int64_t total,
                                        float* __restrict__ gx) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  int n = (int)(t / C), c = (int)(t % C);
  int sidx = seg[n];
  gx[t] = gout[(size_t)sidx * C + c] / (float)max(segptr[sidx + 1] - segptr[sidx], 1);
}

__global__ void gather_rows_kernel(const float* __restrict__ x, const int* __restrict__ idx, int C, int64_t total,
                                   float* __restrict__ out) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  int n = (int)(t / C), c = (int)(t % C);
  out[t] = x[(size_t)idx[n] * C + c];
}

// ---------------------------------------------------------------------------- pool_edge
__global__ void pool_edge_keys_kernel(const int* __restrict__ cnew, const int* __restrict__ row,
                                      const int* __restrict__ col, int64_t E, int bits,
                                      uint64_t* __restrict__ keys, int* __restrict__ vals) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  int a = cnew[row[e]], b = cnew[col[e]];
  keys[e] = (a == b) ? kSentinel : (((uint64_t)(uint32_t)a << bits) | (uint64_t)(uint32_t)b);
  vals[e] = (int)e;
}

__global__ void pool_edge_heads_kernel(const uint64_t* __restrict__ keys, int64_t E, int* __restrict__ head) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  uint64_t k = keys[e];
  head[e] = (k != kSentinel && (e == 0 || keys[e - 1] != k)) ? 1 : 0;
}

__global__ void pool_edge_emit_kernel(const uint64_t* __restrict__ keys, const int* __restrict__ vals,
                                      const int* __restrict__ head, const int* __restrict__ rank,
                                      const float* __restrict__ w, int64_t E, int* __restrict__ row_c,
                                      int* __restrict__ col_c, float* __restrict__ w_c, uint64_t* __restrict__ ukeys,
                                      int* __restrict__ count, int bits) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  if (e == E - 1) *count = rank[e] + head[e];
  if (!head[e]) return;
  uint64_t k = keys[e];
  int o = rank[e];
  row_c[o] = (int)(uint32_t)(k >> bits);
  col_c[o] = (int)(uint32_t)(k & ((1ull << bits) - 1));
  ukeys[o] = k;
  if (w) {
    // mean of the merged duplicates; fp64 accumulation makes the result independent of the
    // order of the run, so w(a,b) == w(b,a) bit for bit and the matching stays symmetric
    double s = 0.0;
    int n = 0;
    for (int64_t f = e; f < E && keys[f] == k; ++f) { s += (double)w[vals[f]]; ++n; }
    w_c[o] = (float)(s / (double)n);
  }
}

// ---------------------------------------------------------- pool_edge, sort-free (matchings)
// For a matching every coarse node has 1-2 members, so its coarse row is the union of at most two
// fine rows: one wave per coarse node gathers those (<= 64) entries, relabels them, sorts them with a
// 64-lane bitonic network, drops the self entry and merges duplicates -- no global radix sort.
// PASS 0 counts the unique neighbours (-> exclusive scan -> rowptr), PASS 1 repeats and writes.
// Rows with more than 64 gathered entries set `overflow` (the caller falls back to the sorted path).
template <int PASS>
__global__ __launch_bounds__(256) void pool_edge_rows_kernel(
    const int* __restrict__ cnew, const int* __restrict__ segptr, const int* __restrict__ members,
    const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ w,
    const int* __restrict__ ncount, int nbound, int* __restrict__ cnt, const int* __restrict__ rowptr_c,
    int* __restrict__ row_c, int* __restrict__ col_c, float* __restrict__ w_c, int* __restrict__ overflow,
    int* __restrict__ total, int4* __restrict__ rowinfo, const int4* __restrict__ rowinfo_in) {
  const int lane = threadIdx.x & 63;
  const int A = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (PASS != 1 && blockIdx.x == 0 && threadIdx.x == 0) cnt[nbound] = 0;     // scan tail: rowptr_c[nbound] = total
  if (PASS == 1 && blockIdx.x == 0 && threadIdx.x == 0) *total = rowptr_c[nbound];
  if (A >= nbound) return;
  // requested before the coarse count is known (entries past it are never read back): one dependent load less
  int4 ri_in = make_int4(0, 0, 0, 0);
  if (PASS == 2 && rowinfo_in != nullptr) ri_in = rowinfo_in[A];
  const int nc = *ncount;
  if (A >= nc) { if (PASS != 1) cnt[A] = 0; return; }
  // the members' fine rows: PASS 0 walks segptr -> members -> rowptr (three dependent loads) and leaves the result
  // for PASS 1, whose chain then starts at the row entries
  int r0, d0, r1, d1;
  if (PASS == 2 && rowinfo_in != nullptr) {
    const int4 ri = ri_in;
    r0 = ri.x; d0 = ri.y; r1 = ri.z; d1 = ri.w;
    if (lane == 0) {
      rowinfo[A] = ri;
      if (d0 + d1 > 64) atomicOr(overflow, 1);
    }
  } else if (PASS != 1) {
    const int ms = segptr[A], me = segptr[A + 1];
    const int m0 = members[ms];
    const int m1 = (me - ms > 1) ? members[ms + 1] : -1;
    r0 = rowptr[m0]; d0 = rowptr[m0 + 1] - r0;
    r1 = m1 >= 0 ? rowptr[m1] : 0; d1 = m1 >= 0 ? rowptr[m1 + 1] - r1 : 0;
    if (lane == 0) {
      rowinfo[A] = make_int4(r0, d0, r1, d1);
      if (d0 + d1 > 64 || me - ms > 2) atomicOr(overflow, 1);
    }
  } else {
    const int4 ri = rowinfo[A];
    r0 = ri.x; d0 = ri.y; r1 = ri.z; d1 = ri.w;
  }
  int key = 0x7fffffff;
  float val = 0.f;
  if (lane < d0 + d1) {
    const int e = lane < d0 ? r0 + lane : r1 + (lane - d0);
    int k = cnew[col[e]];
    if (k != A) { key = k; val = w ? w[e] : 0.f; }
  }
  // bitonic sort, ascending by key; rows of up to 32 entries (the usual case: two mesh rows) skip the 64-wide merge.
  // Partner exchange at distance 1, 2, 4, 8 by DPP lane permutes (quad_perm, row_shl/shr:4, row_ror:8) -- 14 of the 15
  // stages of the 32-wide sort; a wave is a chain of ~20 dependent exchanges, and an LDS-crossbar permute costs ~10x
  // a DPP move in latency.  Every lane of the wave is active here (waves past the coarse count left as a whole).
  auto xchg = [&](int v, int j) -> int {
    switch (j) {
      case 1: return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);          // quad_perm [1,0,3,2]
      case 2: return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);          // quad_perm [2,3,0,1]
      case 4: {
        const int up = __builtin_amdgcn_update_dpp(0, v, 0x104, 0xf, 0xf, true);       // row_shl:4  lane i <- i + 4
        const int dn = __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);       // row_shr:4  lane i <- i - 4
        return (lane & 4) ? dn : up;
      }
      case 8: return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, true);          // row_ror:8 = xor 8 in a row
      default: return __shfl_xor(v, j, 64);
    }
  };
  auto stage = [&](int k, int j) {
    const int pk = xchg(key, j);
    const float pv = __int_as_float(xchg(__float_as_int(val), j));
    const bool keep_min = ((lane & j) == 0) == ((lane & k) == 0);
    const bool take = keep_min ? (pk < key) : (pk > key);
    if (take) { key = pk; val = pv; }
  };
#pragma unroll
  for (int k = 2; k <= 32; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) stage(k, j);
  }
  if (d0 + d1 > 32) {                            // wave-uniform
#pragma unroll
    for (int j = 32; j > 0; j >>= 1) stage(64, j);
  }
  const int prev = __shfl_up(key, 1, 64);
  const bool head = key != 0x7fffffff && (lane == 0 || prev != key);
  const unsigned long long hm = __ballot(head);
  if (PASS == 0) {
    if (lane == 0) cnt[A] = __popcll(hm);
    return;
  }
  if (PASS == 2 && lane == 0) cnt[A] = __popcll(hm);
  // run of duplicates starting at a head lane: up to the next head (or the first invalid lane)
  const unsigned long long vm = __ballot(key != 0x7fffffff);
  const int nvalid = __popcll(vm);
  const unsigned long long after = (lane >= 63) ? 0ull : (hm >> (lane + 1));
  const int next = after ? lane + 1 + (__ffsll((long long)after) - 1) : nvalid;
  const int len = head ? next - lane : 0;
  int maxlen = len;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, m, 64));
  double sum = (double)val;
  for (int t = 1; t < maxlen; ++t) {
    float v = __shfl_down(val, t, 64);
    if (t < len) sum += (double)v;
  }
  if (head) {
    const int t = __popcll(hm & ((1ull << lane) - 1ull));
    if (PASS == 2) {
      // one-pass form: the t-th unique entry parks in the t-th slot of the members' own fine rows (there are at most
      // d0 + d1 of them); pool_edge_compact_kernel moves the rows to their final places once the offsets are known
      const int pos = t < d0 ? r0 + t : r1 + (t - d0);
      col_c[pos] = key;                          // (col_c / w_c are the scratch pair here)
      if (w) w_c[pos] = (float)(sum / (double)len);
    } else {
      const int pos = rowptr_c[A] + t;
      row_c[pos] = A;
      col_c[pos] = key;
      if (w) w_c[pos] = (float)(sum / (double)len);
    }
  }
}

// second half of the one-pass form: 16 lanes per coarse node copy its parked entries to rowptr_c[A] ...
__global__ __launch_bounds__(256) void pool_edge_compact_kernel(const int* __restrict__ ncount, int nbound,
                                                                const int* __restrict__ rowptr_c,
                                                                const int4* __restrict__ rowinfo,
                                                                const int* __restrict__ tcol, const float* __restrict__ tw,
                                                                int* __restrict__ row_c, int* __restrict__ col_c,
                                                                float* __restrict__ w_c, int* __restrict__ total,
                                                                const int* publish_src, int* publish_host, int publish_seq) {
  const int A = (blockIdx.x * 256 + threadIdx.x) >> 4, k = threadIdx.x & 15;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const int tot = rowptr_c[nbound];
    *total = tot;
    if (publish_host != nullptr) {
      // The sizes the host is waiting for are all final once this kernel has started (its own total included):
      // thread 0 hands them over through mapped host memory right away -- the host learns them while the copy pass
      // below is still running, without a device-to-host copy launch and a stream drain in between.
      for (int i = 0; i < 8; ++i) {
        const int* src = publish_src + i;
        publish_host[i] = (src == total) ? tot : *src;
      }
      __threadfence_system();
      __hip_atomic_store(publish_host + 8, publish_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  if (A >= nbound || A >= *ncount) return;
  const int o = rowptr_c[A], c = rowptr_c[A + 1] - o;
  const int4 ri = rowinfo[A];
  for (int t = k; t < c; t += 16) {
    const int src = t < ri.y ? ri.x + t : ri.z + (t - ri.y);
    row_c[o + t] = A;
    col_c[o + t] = tcol[src];
    if (tw) w_c[o + t] = tw[src];
  }
}

__global__ void pool_edge_rowptr_kernel(const uint64_t* __restrict__ ukeys, const int* __restrict__ count, int nmax,
                                        int bits, int* __restrict__ rowptr) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n > nmax) return;
  uint64_t target = (uint64_t)n << bits;
  int lo = 0, hi = *count;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (ukeys[mid] < target) lo = mid + 1; else hi = mid;
  }
  rowptr[n] = lo;
}

__global__ void expand_rowptr_kernel(const int* __restrict__ rowptr, int N, int* __restrict__ row) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  for (int e = rowptr[n]; e < rowptr[n + 1]; ++e) row[e] = n;
}

__global__ void gather_f32_kernel(const float* __restrict__ src, const int* __restrict__ idx, int64_t n,
                                  float* __restrict__ dst) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { int k = idx[i]; dst[i] = k >= 0 ? src[k] : 0.f; }
}

// Exclusive scan of up to a few hundred thousand ints in ONE launch (one 1024-thread block walking the
// array with a running carry).  The graph levels of this path have <= ~10^5 nodes, where two rocPRIM
// launches (init + lookback) cost more in launch latency than the scan itself.  Each thread owns 16
// consecutive ints per step (16 K per step for the block); the loads of the next step are issued before
// the current one is scanned, the carry lives in a register and the wave sums are double-buffered, so
// a step costs one barrier.
__global__ __launch_bounds__(1024) void small_exclusive_scan_kernel(const int* __restrict__ in, int* __restrict__ out,
                                                                    int64_t n) {
  constexpr int EPT = 16, CHUNK = 1024 * EPT;
  __shared__ int wsum[2][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool vec = ((((uintptr_t)in) | ((uintptr_t)out)) & 15) == 0;
  int cur[EPT], nxt[EPT];
  auto load = [&](int64_t base, int* v) {
    const int64_t i0 = base + (int64_t)threadIdx.x * EPT;
    if (vec && i0 + EPT <= n) {
#pragma unroll
      for (int q = 0; q < EPT / 4; ++q) {
        int4 t = *reinterpret_cast<const int4*>(in + i0 + 4 * q);
        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
      }
    } else {
#pragma unroll
      for (int q = 0; q < EPT; ++q) v[q] = (i0 + q < n) ? in[i0 + q] : 0;
    }
  };
  int carry = 0, buf = 0;
  load(0, cur);
  for (int64_t base = 0; base < n; base += CHUNK, buf ^= 1) {
    if (base + CHUNK < n) load(base + CHUNK, nxt);
    int tsum = 0;
#pragma unroll
    for (int q = 0; q < EPT; ++q) tsum += cur[q];
    int inc = tsum;                                   // inclusive scan of thread sums inside the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      int t = __shfl_up(inc, d, 64);
      if (lane >= d) inc += t;
    }
    if (lane == 63) wsum[buf][wave] = inc;
    __syncthreads();
    int woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const int t = wsum[buf][w];
      woff += w < wave ? t : 0;
      total += t;
    }
    int ex = carry + woff + inc - tsum;               // exclusive prefix of this thread's first element
    carry += total;
    const int64_t i0 = base + (int64_t)threadIdx.x * EPT;
    if (vec && i0 + EPT <= n) {
#pragma unroll
      for (int q = 0; q < EPT / 4; ++q) {
        int4 t;
        t.x = ex; ex += cur[4 * q];
        t.y = ex; ex += cur[4 * q + 1];
        t.z = ex; ex += cur[4 * q + 2];
        t.w = ex; ex += cur[4 * q + 3];
        *reinterpret_cast<int4*>(out + i0 + 4 * q) = t;
      }
    } else {
#pragma unroll
      for (int q = 0; q < EPT; ++q) {
        if (i0 + q < n) out[i0 + q] = ex;
        ex += cur[q];
      }
    }
#pragma unroll
    for (int q = 0; q < EPT; ++q) cur[q] = nxt[q];
  }
}

// Two independent exclusive scans in one launch: the walk of small_exclusive_scan_kernel over two arrays
// (16 ints per thread, array and step; the next step's loads in flight while this one is scanned).
__global__ __launch_bounds__(1024) void small_exclusive_scan2_kernel(const int* __restrict__ in_a,
                                                                     const int* __restrict__ in_b,
                                                                     int* __restrict__ out_a, int* __restrict__ out_b,
                                                                     int64_t n) {
  constexpr int EPT = 16, CHUNK = 1024 * EPT;
  __shared__ int wsum[2][2][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool vec = ((((uintptr_t)in_a) | ((uintptr_t)in_b) | ((uintptr_t)out_a) | ((uintptr_t)out_b)) & 15) == 0;
  int ca[EPT], cb[EPT], na[EPT], nb[EPT];
  auto load = [&](const int* __restrict__ in, int64_t base, int* v) {
    const int64_t i0 = base + (int64_t)threadIdx.x * EPT;
    if (vec && i0 + EPT <= n) {
#pragma unroll
      for (int q = 0; q < EPT / 4; ++q) {
        int4 t = *reinterpret_cast<const int4*>(in + i0 + 4 * q);
        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
      }
    } else {
#pragma unroll
      for (int q = 0; q < EPT; ++q) v[q] = (i0 + q < n) ? in[i0 + q] : 0;
    }
  };
  auto store = [&](int* __restrict__ out, int64_t base, const int* v, int ex) {
    const int64_t i0 = base + (int64_t)threadIdx.x * EPT;
    if (vec && i0 + EPT <= n) {
#pragma unroll
      for (int q = 0; q < EPT / 4; ++q) {
        int4 t;
        t.x = ex; ex += v[4 * q];
        t.y = ex; ex += v[4 * q + 1];
        t.z = ex; ex += v[4 * q + 2];
        t.w = ex; ex += v[4 * q + 3];
        *reinterpret_cast<int4*>(out + i0 + 4 * q) = t;
      }
    } else {
#pragma unroll
      for (int q = 0; q < EPT; ++q) {
        if (i0 + q < n) out[i0 + q] = ex;
        ex += v[q];
      }
    }
  };
  int carry_a = 0, carry_b = 0, buf = 0;
  load(in_a, 0, ca);
  load(in_b, 0, cb);
  for (int64_t base = 0; base < n; base += CHUNK, buf ^= 1) {
    if (base + CHUNK < n) { load(in_a, base + CHUNK, na); load(in_b, base + CHUNK, nb); }
    int ta = 0, tb = 0;
#pragma unroll
    for (int q = 0; q < EPT; ++q) { ta += ca[q]; tb += cb[q]; }
    int ia = ta, ib = tb;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      int xa = __shfl_up(ia, d, 64), xb = __shfl_up(ib, d, 64);
      if (lane >= d) { ia += xa; ib += xb; }
    }
    if (lane == 63) { wsum[buf][0][wave] = ia; wsum[buf][1][wave] = ib; }
    __syncthreads();
    int woa = 0, wob = 0, tota = 0, totb = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const int xa = wsum[buf][0][w], xb = wsum[buf][1][w];
      woa += w < wave ? xa : 0; wob += w < wave ? xb : 0;
      tota += xa; totb += xb;
    }
    store(out_a, base, ca, carry_a + woa + ia - ta);
    store(out_b, base, cb, carry_b + wob + ib - tb);
    carry_a += tota; carry_b += totb;
#pragma unroll
    for (int q = 0; q < EPT; ++q) { ca[q] = na[q]; cb[q] = nb[q]; }
  }
}

constexpr int64_t kSmallScan = 1 << 18;
constexpr int64_t kOneBlockScan = 1 << 14;        // up to here the one-block walk is a single step

// Exclusive scan of 16 K .. 256 K ints in ONE launch of several blocks (decoupled look-back): the one-block walk above
// is bound by a single CU's memory throughput (3.3 us per 16 K ints: 20 us for the 82 K-entry row pointers of the
// facet level).  Blocks take their index from a ticket (so every predecessor of a block is already running: the
// look-back cannot wait on a block that has not started), publish (epoch | flag | value) as one 64-bit word -- flag 1:
// the block's own sum, flag 2: the inclusive prefix up to and including it -- and wave 0 of each block inspects 64
// predecessors at a time.  The state words carry the call's epoch, so the buffer is never cleared between calls.
// Library-global state: one scan at a time (every scan of this path runs on the caller's stream, in order).
__device__ __forceinline__ unsigned long long scan_pack(unsigned epoch, unsigned flag, int value) {
  return ((unsigned long long)((epoch << 2) | flag) << 32) | (unsigned)value;
}

__global__ __launch_bounds__(256) void lookback_exclusive_scan_kernel(const int* __restrict__ in, int* __restrict__ out,
                                                                      int64_t n, unsigned long long* state, int* ticket,
                                                                      unsigned epoch, int nblocks) {
  constexpr int EPT = 16, CHUNK = 256 * EPT;
  __shared__ int s_bid, s_prefix, wsum[4];
  if (threadIdx.x == 0) s_bid = atomicAdd(ticket, 1);
  __syncthreads();
  const int b = s_bid;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t i0 = (int64_t)b * CHUNK + (int64_t)threadIdx.x * EPT;
  const bool vec = ((((uintptr_t)in) | ((uintptr_t)out)) & 15) == 0 && i0 + EPT <= n;
  int v[EPT];
  if (vec) {
#pragma unroll
    for (int q = 0; q < EPT / 4; ++q) {
      const int4 t = *reinterpret_cast<const int4*>(in + i0 + 4 * q);
      v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int q = 0; q < EPT; ++q) v[q] = (i0 + q < n) ? in[i0 + q] : 0;
  }
  int tsum = 0;
#pragma unroll
  for (int q = 0; q < EPT; ++q) tsum += v[q];
  int inc = tsum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(inc, d, 64);
    if (lane >= d) inc += t;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int woff = 0, total = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const int t = wsum[w];
    woff += w < wave ? t : 0;
    total += t;
  }
  if (wave == 0) {
    int prefix = 0;
    if (b > 0) {
      if (lane == 0) __hip_atomic_store(state + b, scan_pack(epoch, 1, total), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      for (int hi = b - 1; hi >= 0; hi -= 64) {
        const int pidx = hi - lane;
        unsigned flag = 2;
        int val = 0;
        if (pidx >= 0) {
          unsigned long long st;
          int spins = 0;
          do {
            st = __hip_atomic_load(state + pidx, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
          } while (((unsigned)(st >> 34) != epoch || ((st >> 32) & 3) == 0) && ++spins < (1 << 26));
          flag = (unsigned)(st >> 32) & 3;
          val = (int)(unsigned)st;
        }
        const unsigned long long done = __ballot(flag == 2);
        const int first = done ? __ffsll((long long)done) - 1 : 64;       // nearest predecessor with a full prefix
        int c = lane <= first ? val : 0;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m, 64);
        prefix += c;
        if (done) break;
      }
    }
    if (lane == 0) {
      __hip_atomic_store(state + b, scan_pack(epoch, 2, prefix + total), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      s_prefix = prefix;
      if (b == nblocks - 1) *ticket = 0;                                   // every ticket of this call is taken
    }
  }
  __syncthreads();
  int ex = s_prefix + woff + inc - tsum;
  if (vec) {
#pragma unroll
    for (int q = 0; q < EPT / 4; ++q) {
      int4 t;
      t.x = ex; ex += v[4 * q];
      t.y = ex; ex += v[4 * q + 1];
      t.z = ex; ex += v[4 * q + 2];
      t.w = ex; ex += v[4 * q + 3];
      *reinterpret_cast<int4*>(out + i0 + 4 * q) = t;
    }
  } else {
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
      if (i0 + q < n) out[i0 + q] = ex;
      ex += v[q];
    }
  }
}

// persistent look-back state (zeroed once; see the kernel).  One per host thread: a thread drives one stream at a time
// (one mesh group of a concurrent step, executor.hip), so two scans in flight never share state words.
static thread_local unsigned long long* g_scan_state = nullptr;
static thread_local int* g_scan_ticket = nullptr;
static thread_local unsigned g_scan_epoch = 0;

static hipError_t lookback_scan(const int* in, int* out, int64_t n, hipStream_t s) {
  if (g_scan_state == nullptr) {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, 4096 * sizeof(unsigned long long) + 256);
    if (e != hipSuccess) return e;
    // zeroed ON THE LAUNCH STREAM: a null-stream memset is not ordered against a non-blocking stream, and a host thread
    // that starts later (one context per thread) initialises its state while other threads' kernels are in flight
    e = hipMemsetAsync(p, 0, 4096 * sizeof(unsigned long long) + 256, s);
    if (e != hipSuccess) return e;
    g_scan_state = (unsigned long long*)p;
    g_scan_ticket = (int*)((char*)p + 4096 * sizeof(unsigned long long));
  }
  g_scan_epoch = (g_scan_epoch + 1) & 0x3fffffffu;
  if (g_scan_epoch == 0) g_scan_epoch = 1;
  const int nblocks = cdiv(n, 4096);
  lookback_exclusive_scan_kernel<<<nblocks, 256, 0, s>>>(in, out, n, g_scan_state, g_scan_ticket, g_scan_epoch, nblocks);
  return hipGetLastError();
}


static hipError_t exclusive_scan_int(void* temp, size_t& tb, const int* in, int* out, int64_t n, hipStream_t s) {
  if (n <= kSmallScan) {
    if (temp == nullptr) { tb = 16; return hipSuccess; }
    static const bool lookback = [] { const char* f = getenv("GEOBI_SCAN_LOOKBACK"); return !f || atoi(f) != 0; }();   // A/B knob
    if (lookback && n > kOneBlockScan) return lookback_scan(in, out, n, s);
    small_exclusive_scan_kernel<<<1, 1024, 0, s>>>(in, out, n);
    return hipGetLastError();
  }
  return rocprim::exclusive_scan(temp, tb, in, out, 0, (size_t)n, rocprim::plus<int>(), s, false);
}

template <typename T>
size_t scan_temp_bytes(int64_t n) {
  size_t tb = 0;
  (void)rocprim::exclusive_scan(nullptr, tb, (T*)nullptr, (T*)nullptr, (T)0, (size_t)n, rocprim::plus<T>(), (hipStream_t)0,
                          false);
  return tb;
}

size_t sort_keys_temp_bytes(int64_t n) {
  size_t tb = 0;
  (void)rocprim::radix_sort_keys(nullptr, tb, (uint64_t*)nullptr, (uint64_t*)nullptr, (size_t)n, 0u, 64u, (hipStream_t)0,
                           false);
  return tb;
}

size_t sort_pairs_temp_bytes(int64_t n) {
  size_t tb = 0;
  (void)rocprim::radix_sort_pairs(nullptr, tb, (uint64_t*)nullptr, (uint64_t*)nullptr, (int*)nullptr, (int*)nullptr,
                            (size_t)n, 0u, 64u, (hipStream_t)0, false);
  return tb;
}

}  // namespace

size_t scan_ws_bytes(int64_t n) {
  size_t tb = n <= kSmallScan ? 16 : scan_temp_bytes<int>(n);
  return align_up(tb ? tb : 16);
}

int scan_exclusive_i32(void* temp, size_t temp_bytes, const int* in, int* out, int64_t n, hipStream_t s) {
  size_t tb = temp_bytes;
  GEOBI_HIP(exclusive_scan_int(temp, tb, in, out, n, s));
  return 0;
}

int edge_weight_t10(const float* x, int C, const int32_t* row, const int32_t* col, const float* w_in, int64_t E,
                    float* w_out, hipStream_t s, int32_t* zero8) {
  if (E <= 0) return 0;
  edge_weight_t10_kernel<<<cdiv(E * 8, 256), 256, 0, s>>>(x, C, row, col, w_in, E, w_out, zero8);
  GEOBI_LAUNCH_OK();
  return 0;
}

int edge_weight_att(const float* x, int C, const float* att_l, const float* att_r, const int32_t* row, const int32_t* col,
                    const float* w_in, int64_t N, int64_t E, float* node_ws, float* w_out, hipStream_t s) {
  if (E <= 0 || N <= 0) return 0;
  if (C <= 0) return set_error("edge_weight_att: C = %d", C);
  node_att_kernel<<<cdiv(N * 8, 256), 256, 0, s>>>(x, C, att_l, att_r, N, reinterpret_cast<float2*>(node_ws));
  GEOBI_LAUNCH_OK();
  edge_weight_att_kernel<<<cdiv(E, 256), 256, 0, s>>>(reinterpret_cast<const float2*>(node_ws), row, col, w_in, E, w_out);
  GEOBI_LAUNCH_OK();
  return 0;
}

int gather_f32(const float* src, const int32_t* idx, int64_t n, float* dst, hipStream_t s) {
  if (n <= 0) return 0;
  gather_f32_kernel<<<cdiv(n, 256), 256, 0, s>>>(src, idx, n, dst);
  GEOBI_LAUNCH_OK();
  return 0;
}

int expand_rowptr(const int32_t* rowptr, int64_t N, int32_t* row, hipStream_t s) {
  if (N <= 0) return 0;
  expand_rowptr_kernel<<<cdiv(N, 256), 256, 0, s>>>(rowptr, (int)N, row);
  GEOBI_LAUNCH_OK();
  return 0;
}

// `rounds` proposal rounds; the first one initialises the state (init != 0) and the undecided counter.
// On return pp holds the proposals of the last round.
// Test hook (geobi_set_match_round_cap): at most cap x (rounds / 8) rounds per call.  The callers ask for 8 rounds and
// double the number on every resume of their repair loops; a cap of 1 makes those 1, 2, 4, ... -- every pooling step then
// comes back with undecided nodes and is resumed, which an uncapped mesh graph almost never needs.
static int g_round_cap = 0;
void set_match_round_cap(int cap) { g_round_cap = cap > 0 ? cap : 0; }

static void launch_match_rounds(const int32_t* rowptr, const int32_t* col, const float* w, int N, int rounds, int init,
                                int32_t* cluster, int32_t* status, int*& pp, int*& pn, hipStream_t s) {
  const int blocks = cdiv(N, 256);
  if (g_round_cap > 0) {
    const int64_t capped = (int64_t)g_round_cap * (rounds > 8 ? rounds / 8 : 1);
    if (capped < rounds) rounds = (int)capped;
  }
  for (int r = 0; r < rounds; ++r) {
    if (r > 0)
      match_round_kernel<0><<<blocks, 256, 0, s>>>(rowptr, col, w, pp, N, cluster, pn, status);
    else if (init)
      match_round_kernel<1><<<blocks, 256, 0, s>>>(rowptr, col, w, pp, N, cluster, pn, status);
    else
      match_round_kernel<2><<<blocks, 256, 0, s>>>(rowptr, col, w, pp, N, cluster, pn, status);
    int* t = pp; pp = pn; pn = t;
  }
}

size_t match_ws_bytes(int64_t N) { return 2 * align_up((size_t)N * sizeof(int)) + 1024; }

// Runs `rounds` proposal rounds.  init != 0 starts from scratch, init == 0 continues from the state
// in `cluster` (entries < 0 = undecided).  `status[0]` receives the number of nodes still undecided
// after the last round (0 = converged).  `cluster` keeps the resumable state; `cluster_final`
// (optional) receives a copy with undecided nodes closed as singletons, i.e. always a valid clustering.
int match_heavy_edge(const int32_t* rowptr, const int32_t* col, const float* w, int64_t N, int rounds, int init,
                     int32_t* cluster, int32_t* cluster_final, int32_t* status, void* ws, size_t ws_bytes,
                     hipStream_t s) {
  GEOBI_REQUIRE(N > 0 && rounds > 0, "match: empty graph or no rounds");
  Arena a(ws, ws_bytes);
  int* prop0 = a.take<int>(N);
  int* prop1 = a.take<int>(N);
  GEOBI_REQUIRE(a.ok() && prop0, "match: workspace too small");
  int blocks = cdiv(N, 256);
  int* pp = prop0;
  int* pn = prop1;
  launch_match_rounds(rowptr, col, w, (int)N, rounds, init, cluster, status, pp, pn, s);
  match_commit_kernel<<<blocks, 256, 0, s>>>(pp, (int)N, cluster, status, cluster_final, nullptr, nullptr);
  GEOBI_LAUNCH_OK();
  return 0;
}

size_t match_coarsen_ws_bytes(int64_t N) {
  return match_ws_bytes(N) + 5 * align_up((size_t)(N + 1) * sizeof(int)) + 2 * scan_ws_bytes(N + 1) + 1024;
}

// One pooling step's integer front end in one call: matching rounds, commit, dense relabel and the inverse
// lists of the matching (what geobi_match_heavy_edge + geobi_relabel_compact + geobi_segment_csr_pairs
// produce, 6 launches fewer).  counters[0] = undecided nodes, counters[1] = coarse node count.
int match_coarsen(const int32_t* rowptr, const int32_t* col, const float* w, int64_t N, int rounds, int init,
                  int32_t* state, int32_t* cluster_final, int32_t* cnew, int32_t* segptr, int32_t* members,
                  int32_t* counters, void* ws, size_t ws_bytes, hipStream_t s, void* rowinfo_out, bool* rowinfo_made) {
  GEOBI_REQUIRE(N > 0 && rounds > 0, "match_coarsen: empty graph or no rounds");
  Arena a(ws, ws_bytes);
  int* prop0 = a.take<int>(N);
  int* prop1 = a.take<int>(N);
  int* flag = a.take<int>(N + 1);
  int* sz = a.take<int>(N + 1);
  int* rank = a.take<int>(N + 1);
  int* offs = a.take<int>(N + 1);
  size_t tb = scan_ws_bytes(N + 1);
  void* temp_a = a.take<char>(tb);
  void* temp_b = a.take<char>(tb);
  GEOBI_REQUIRE(a.ok() && prop0, "match_coarsen: workspace too small (%zu < %zu)", ws_bytes, a.off);
  const int blocks = cdiv(N, 256);
  int* pp = prop0;
  int* pn = prop1;
  launch_match_rounds(rowptr, col, w, (int)N, rounds, init, state, counters, pp, pn, s);
  static const bool scan_free = [] { const char* f = getenv("GEOBI_MATCH_SCANFREE"); return !f || atoi(f) != 0; }();   // A/B knob
  if (scan_free && blocks <= kMaxScanBlocks) {
    // scan-free pair: block-local prefixes in the commit kernel, block totals folded by the list kernel
    int* bt_rank = rank;                 // rank / offs are free in this variant: reuse them for the block totals
    int* bt_offs = offs;
    match_commit_scan_kernel<<<blocks, 256, 0, s>>>(pp, (int)N, state, counters, cluster_final, flag, sz, bt_rank, bt_offs);
    GEOBI_LAUNCH_OK();
    match_lists_scan_kernel<<<blocks, 256, 0, s>>>(state, cluster_final, flag, sz, bt_rank, bt_offs, blocks, (int)N, cnew,
                                                   counters + 1, segptr, members, rowptr, (int4*)rowinfo_out);
    GEOBI_LAUNCH_OK();
    if (rowinfo_made) *rowinfo_made = rowinfo_out != nullptr;
    return 0;
  }
  if (rowinfo_made) *rowinfo_made = false;
  match_commit_kernel<<<blocks, 256, 0, s>>>(pp, (int)N, state, counters, cluster_final, flag, sz);
  GEOBI_LAUNCH_OK();
  if (N + 1 <= kSmallScan) {
    small_exclusive_scan2_kernel<<<1, 1024, 0, s>>>(flag, sz, rank, offs, N + 1);
    GEOBI_LAUNCH_OK();
  } else {
    GEOBI_TRY(scan_exclusive_i32(temp_a, tb, flag, rank, N + 1, s));
    GEOBI_TRY(scan_exclusive_i32(temp_b, tb, sz, offs, N + 1, s));
  }
  match_lists_kernel<<<blocks, 256, 0, s>>>(state, cluster_final, rank, offs, (int)N, cnew, counters + 1, segptr,
                                            members);
  GEOBI_LAUNCH_OK();
  return 0;
}

size_t relabel_ws_bytes(int64_t N) {
  return align_up((size_t)N * sizeof(int)) * 2 + align_up(scan_temp_bytes<int>(N)) + 512;
}

int relabel_compact(const int32_t* cluster, int64_t N, int rep_is_self, int32_t* cnew, int32_t* count, void* ws,
                    size_t ws_bytes, hipStream_t s) {
  GEOBI_REQUIRE(N > 0, "relabel: empty");
  Arena a(ws, ws_bytes);
  int* flag = a.take<int>(N);
  int* rank = a.take<int>(N);
  size_t tb = scan_temp_bytes<int>(N);
  void* temp = a.take<char>(tb ? tb : 1);
  GEOBI_REQUIRE(a.ok() && flag, "relabel: workspace too small");
  int blocks = cdiv(N, 256);
  if (rep_is_self) {
    rep_self_flag_kernel<<<blocks, 256, 0, s>>>(cluster, (int)N, flag);
  } else {
    GEOBI_HIP(hipMemsetAsync(flag, 0, sizeof(int) * N, s));
    rep_flag_kernel<<<blocks, 256, 0, s>>>(cluster, (int)N, flag);
  }
  GEOBI_LAUNCH_OK();
  GEOBI_HIP(exclusive_scan_int(temp, tb, flag, rank, N, s));
  relabel_apply_kernel<<<blocks, 256, 0, s>>>(cluster, flag, rank, (int)N, cnew, count);
  GEOBI_LAUNCH_OK();
  return 0;
}

size_t segment_csr_ws_bytes(int64_t n) {
  return align_up((size_t)n * sizeof(uint64_t)) * 2 + align_up(sort_keys_temp_bytes(n)) + 512;
}

int segment_csr(const int32_t* seg, int64_t n, int64_t nseg, int32_t* segptr, int32_t* members, void* ws,
                size_t ws_bytes, hipStream_t s) {
  if (n <= 0) {
    GEOBI_HIP(hipMemsetAsync(segptr, 0, sizeof(int) * (nseg + 1), s));
    return 0;
  }
  Arena a(ws, ws_bytes);
  uint64_t* k_in = a.take<uint64_t>(n);
  uint64_t* k_out = a.take<uint64_t>(n);
  size_t tb = sort_keys_temp_bytes(n);
  void* temp = a.take<char>(tb ? tb : 1);
  GEOBI_REQUIRE(a.ok() && k_in, "segment_csr: workspace too small");
  const int bits = key_bits(n);                 // member index < n
  const int sbits = key_bits(nseg);
  seg_keys_kernel<<<cdiv(n, 256), 256, 0, s>>>(seg, n, bits, k_in);
  GEOBI_LAUNCH_OK();
  GEOBI_HIP(rocprim::radix_sort_keys(temp, tb, k_in, k_out, (size_t)n, 0u, (unsigned)(bits + sbits), s, false));
  seg_unpack_kernel<<<cdiv(n, 256), 256, 0, s>>>(k_out, n, bits, members);
  seg_ptr_kernel<<<cdiv(nseg + 1, 256), 256, 0, s>>>(k_out, n, (int)nseg, bits, segptr);
  GEOBI_LAUNCH_OK();
  return 0;
}

size_t segment_pairs_ws_bytes(int64_t nseg) {
  return align_up((size_t)nseg * sizeof(int)) + align_up(scan_temp_bytes<int>(nseg)) + 512;
}

int segment_csr_pairs(const int32_t* cnew, const int32_t* raw, int64_t N, int64_t nseg, int32_t* segptr,
                      int32_t* members, void* ws, size_t ws_bytes, hipStream_t s) {
  GEOBI_REQUIRE(N > 0 && nseg > 0, "segment_csr_pairs: empty");
  Arena a(ws, ws_bytes);
  int* cnt = a.take<int>(nseg);
  size_t tb = scan_temp_bytes<int>(nseg);
  void* temp = a.take<char>(tb ? tb : 1);
  GEOBI_REQUIRE(a.ok() && cnt, "segment_csr_pairs: workspace too small");
  pair_count_kernel<<<cdiv(nseg, 256), 256, 0, s>>>(cnew, raw, (int)N, (int)nseg, cnt);
  pair_mark_kernel<<<cdiv(N, 256), 256, 0, s>>>(cnew, raw, (int)N, cnt);
  GEOBI_LAUNCH_OK();
  GEOBI_HIP(exclusive_scan_int(temp, tb, cnt, segptr, nseg, s));
  pair_fill_kernel<<<cdiv(N, 256), 256, 0, s>>>(cnew, raw, (int)N, (int)nseg, segptr, members);
  GEOBI_LAUNCH_OK();
  return 0;
}

int segment_csr_compose(const int32_t* segptr1, const int32_t* members1, const int32_t* segptr2,
                        const int32_t* members2, int64_t nseg2, int64_t n_fine, int32_t* segptr12,
                        int32_t* members12, void* ws, size_t ws_bytes, hipStream_t s) {
  GEOBI_REQUIRE(nseg2 > 0, "segment_csr_compose: empty");
  Arena a(ws, ws_bytes);
  int* cnt = a.take<int>(nseg2);
  size_t tb = scan_temp_bytes<int>(nseg2);
  void* temp = a.take<char>(tb ? tb : 1);
  GEOBI_REQUIRE(a.ok() && cnt, "segment_csr_compose: workspace too small");
  compose_count_kernel<<<cdiv(nseg2, 256), 256, 0, s>>>(segptr1, segptr2, members2, (int)nseg2, cnt);
  GEOBI_LAUNCH_OK();
  GEOBI_HIP(exclusive_scan_int(temp, tb, cnt, segptr12, nseg2, s));
  compose_fill_kernel<<<cdiv(nseg2, 256), 256, 0, s>>>(segptr1, members1, segptr2, members2, (int)nseg2,
                                                       (int)n_fine, segptr12, members12);
  GEOBI_LAUNCH_OK();
  return 0;
}

int segment_max_fwd(const float* x, int C, const int32_t* segptr, const int32_t* members, int64_t nseg, float* out,
                    int32_t* arg, hipStream_t s) {
  if (nseg <= 0) return 0;
  segment_max_fwd_kernel<<<cdiv(nseg * C, 256), 256, 0, s>>>(x, C, segptr, members, (int)nseg, out, arg);
  GEOBI_LAUNCH_OK();
  return 0;
}

int segment_max_bwd(const float* gout, const int32_t* arg, const int32_t* seg, int C, int64_t nseg, int64_t n_fine,
                    float* gx, hipStream_t s) {
  if (n_fine <= 0) return 0;
  segment_max_bwd_kernel<<<cdiv(n_fine * C, 256), 256, 0, s>>>(gout, arg, seg, C, n_fine * C, (int)nseg, gx);
  GEOBI_LAUNCH_OK();
  return 0;
}

int segment_max2_fwd(const float* x, int C, const int32_t* segptr1, const int32_t* members1, const int32_t* segptr2,
                     const int32_t* members2, int64_t nseg2, float* out, int32_t* arg12, hipStream_t s) {
  if (nseg2 <= 0) return 0;
  segment_max2_fwd_kernel<<<cdiv(nseg2 * C, 256), 256, 0, s>>>(x, C, segptr1, members1, segptr2, members2, (int)nseg2, out,
                                                              arg12);
  GEOBI_LAUNCH_OK();
  return 0;
}

int segment_max2_bwd(const float* gout, const int32_t* arg12, const int32_t* seg12, int C, int64_t nseg2, int64_t n_fine,
                     float* gx, int add, hipStream_t s) {
  if (n_fine <= 0) return 0;
  segment_max2_bwd_kernel<<<cdiv(n_fine * C, 256), 256, 0, s>>>(gout, arg12, seg12, C, n_fine * C, (int)nseg2, gx, add);
  GEOBI_LAUNCH_OK();
  return 0;
}

int segment_sum(const float* x, int C, const int32_t* segptr, const int32_t* members, int64_t nseg, int mean,
                float* out, hipStream_t s) {
  if (nseg <= 0) return 0;
  segment_sum_kernel<<<cdiv(nseg * C, 256), 256, 0, s>>>(x, C, segptr, members, (int)nseg, mean, out);
  GEOBI_LAUNCH_OK();
  return 0;
}

int segment_sum2(const float* x, int C, const int32_t* segptr1, const int32_t* members1, const int32_t* segptr2,
                 const int32_t* members2, int64_t nseg2, float* out, hipStream_t s) {
  if (nseg2 <= 0) return 0;
  GEOBI_REQUIRE((C & 3) == 0, "segment_sum2: channel count %d is not a multiple of 4", C);
  segment_sum2_kernel<<<cdiv(nseg2 * (C >> 2), 256), 256, 0, s>>>(x, C, segptr1, members1, segptr2, members2, (int)nseg2, out);
  GEOBI_LAUNCH_OK();
  return 0;
}

int segment_mean_bwd(const float* gout, const int32_t* seg, const int32_t* segptr, int C, int64_t n_fine, float* gx,
                     hipStream_t s) {
  if (n_fine <= 0) return 0;
  segment_mean_bwd_kernel<<<cdiv(n_fine * C, 256), 256, 0, s>>>(gout, seg, segptr, C, n_fine * C, gx);
  GEOBI_LAUNCH_OK();
  return 0;
}

int gather_rows(const float* x, const int32_t* idx, int C, int64_t n_out, float* out, hipStream_t s) {
  if (n_out <= 0) return 0;
  gather_rows_kernel<<<cdiv(n_out * C, 256), 256, 0, s>>>(x, idx, C, n_out * C, out);
  GEOBI_LAUNCH_OK();
  return 0;
}

// with E (the fine edge count) the workspace also holds the scratch pair of the one-pass form
size_t pool_edge_rows_ws_bytes_onepass(int64_t nbound, int64_t E) {
  return pool_edge_rows_ws_bytes(nbound) + 2 * align_up((size_t)E * sizeof(int)) + 256;
}

size_t pool_edge_rows_ws_bytes(int64_t nbound) {
  return align_up((size_t)(nbound + 1) * sizeof(int)) + align_up(scan_temp_bytes<int>(nbound + 1)) +
         align_up((size_t)nbound * sizeof(int4)) + 512;
}

// cnew, (segptr, members) = pair lists built with the bound `nbound` (fine node count), ncount = device
// count of coarse nodes.  rowptr_c has nbound + 1 entries; count[0] = coarse edges; overflow[0] |= 1 if
// some row did not fit (outputs are then incomplete and the caller must use pool_edge).
int pool_edge_rows(const int32_t* cnew, const int32_t* segptr, const int32_t* members, const int32_t* rowptr,
                   const int32_t* col, const float* w, const int32_t* ncount, int64_t nbound, int32_t* rowptr_c,
                   int32_t* row_c, int32_t* col_c, float* w_c, int32_t* count, int32_t* overflow, void* ws,
                   size_t ws_bytes, hipStream_t s, int64_t E_fine, const void* rowinfo_in, const int32_t* publish_src,
                   int32_t* publish_host, int publish_seq) {
  GEOBI_REQUIRE(nbound > 0, "pool_edge_rows: empty");
  Arena a(ws, ws_bytes);
  int* cnt = a.take<int>(nbound + 1);
  size_t tb = scan_temp_bytes<int>(nbound + 1);
  void* temp = a.take<char>(tb ? tb : 1);
  int4* rowinfo = a.take<int4>(nbound);
  GEOBI_REQUIRE(a.ok() && cnt && rowinfo, "pool_edge_rows: workspace too small");
  int blocks = cdiv(nbound, 4);
  if (E_fine > 0) {
    // one-pass form (callers that sized the workspace with pool_edge_rows_ws_bytes_onepass): gather + relabel + sort +
    // merge ONCE, entries parked in scratch, then a short copy pass instead of a second merge
    int* tcol = a.take<int>(E_fine);
    float* tw = a.take<float>(E_fine);
    GEOBI_REQUIRE(a.ok() && tcol && tw, "pool_edge_rows: workspace too small for the one-pass form");
    pool_edge_rows_kernel<2><<<blocks, 256, 0, s>>>(cnew, segptr, members, rowptr, col, w, ncount, (int)nbound, cnt,
                                                     nullptr, nullptr, tcol, tw, overflow, nullptr, rowinfo,
                                                     (const int4*)rowinfo_in);
    GEOBI_LAUNCH_OK();
    GEOBI_HIP(exclusive_scan_int(temp, tb, cnt, rowptr_c, nbound + 1, s));
    pool_edge_compact_kernel<<<cdiv(nbound * 16, 256), 256, 0, s>>>(ncount, (int)nbound, rowptr_c, rowinfo, tcol,
                                                                    w ? tw : nullptr, row_c, col_c, w_c, count,
                                                                    publish_src, publish_host, publish_seq);
    GEOBI_LAUNCH_OK();
    return 0;
  }
  pool_edge_rows_kernel<0><<<blocks, 256, 0, s>>>(cnew, segptr, members, rowptr, col, w, ncount, (int)nbound, cnt,
                                                   nullptr, nullptr, nullptr, nullptr, overflow, nullptr, rowinfo, nullptr);
  GEOBI_LAUNCH_OK();
  GEOBI_HIP(exclusive_scan_int(temp, tb, cnt, rowptr_c, nbound + 1, s));
  pool_edge_rows_kernel<1><<<blocks, 256, 0, s>>>(cnew, segptr, members, rowptr, col, w, ncount, (int)nbound, cnt,
                                                   rowptr_c, row_c, col_c, w_c, overflow, count, rowinfo, nullptr);
  GEOBI_LAUNCH_OK();
  return 0;
}

size_t pool_edge_ws_bytes(int64_t E) {
  return align_up((size_t)E * sizeof(uint64_t)) * 3 + align_up((size_t)E * sizeof(int)) * 4 +
         align_up(sort_pairs_temp_bytes(E)) + align_up(scan_temp_bytes<int>(E)) + 1024;
}

// Outputs are sized for the worst case (E entries, nmax + 1 row pointers); the true edge count is
// written to count[0] on the device.
int pool_edge(const int32_t* cnew, const int32_t* row, const int32_t* col, const float* w, int64_t E, int64_t nmax,
              int32_t* rowptr_c, int32_t* row_c, int32_t* col_c, float* w_c, int32_t* count, void* ws,
              size_t ws_bytes, hipStream_t s) {
  if (E <= 0) {
    GEOBI_HIP(hipMemsetAsync(rowptr_c, 0, sizeof(int) * (nmax + 1), s));
    GEOBI_HIP(hipMemsetAsync(count, 0, sizeof(int), s));
    return 0;
  }
  Arena a(ws, ws_bytes);
  uint64_t* k_in = a.take<uint64_t>(E);
  uint64_t* k_out = a.take<uint64_t>(E);
  uint64_t* ukeys = a.take<uint64_t>(E);
  int* v_in = a.take<int>(E);
  int* v_out = a.take<int>(E);
  int* head = a.take<int>(E);
  int* rank = a.take<int>(E);
  size_t tb_sort = sort_pairs_temp_bytes(E), tb_scan = scan_temp_bytes<int>(E);
  void* t_sort = a.take<char>(tb_sort ? tb_sort : 1);
  void* t_scan = a.take<char>(tb_scan ? tb_scan : 1);
  GEOBI_REQUIRE(a.ok() && k_in, "pool_edge: workspace too small (%zu < %zu)", ws_bytes, a.off);
  int blocks = cdiv(E, 256);
  const int bits = key_bits(nmax);
  pool_edge_keys_kernel<<<blocks, 256, 0, s>>>(cnew, row, col, E, bits, k_in, v_in);
  GEOBI_LAUNCH_OK();
  GEOBI_HIP(rocprim::radix_sort_pairs(t_sort, tb_sort, k_in, k_out, v_in, v_out, (size_t)E, 0u, (unsigned)(2 * bits),
                                      s, false));
  pool_edge_heads_kernel<<<blocks, 256, 0, s>>>(k_out, E, head);
  GEOBI_LAUNCH_OK();
  GEOBI_HIP(exclusive_scan_int(t_scan, tb_scan, head, rank, E, s));
  pool_edge_emit_kernel<<<blocks, 256, 0, s>>>(k_out, v_out, head, rank, w, E, row_c, col_c, w_c, ukeys, count,
                                               bits);
  GEOBI_LAUNCH_OK();
  pool_edge_rowptr_kernel<<<cdiv(nmax + 1, 256), 256, 0, s>>>(ukeys, count, (int)nmax, bits, rowptr_c);
  GEOBI_LAUNCH_OK();
  return 0;
}

}  // namespace geobi
