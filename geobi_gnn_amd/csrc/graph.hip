// Edge-sorted adjacency (CSR) construction for the mesh graphs.
//
// The reference hands the path COO `edge_index [2,E] int64` (row = source j, col = target i;
// /root/reference/code/network.py:271 -> FeaStConv).  Every kernel of this library walks a
// CSR sorted by (segment node, neighbour) so gathers are row-coalesced and aggregation is a
// sorted-segment reduction (no atomics, fixed summation order -> bitwise reproducible).
// Self loops are dropped here: FeaStConv removes and re-adds exactly one per node, which
// the conv kernels apply implicitly, and the pooling layer drops them first too
// (/root/reference/code/net_util.py:163).
#include "common.h"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace geobi {

namespace {

constexpr uint64_t kSentinel = ~0ull;

// Keys are (segment << bits) | neighbour with (1 << bits) > N, so the radix sort only has to look at
// 2 * bits bits (34 for the 82 k-node level-0 facet graph) instead of 64.  The sentinel (all ones)
// still sorts last because every valid key is < 2^(2 bits) - 1.
static inline int key_bits(int64_t N) {
  int b = 1;
  while ((1ll << b) <= N) ++b;
  return b;
}

__global__ void make_keys_kernel(const int64_t* __restrict__ seg, const int64_t* __restrict__ nbr, int64_t E,
                                 int64_t N, int drop_self, int bits, uint64_t* __restrict__ keys,
                                 int32_t* __restrict__ vals, int32_t* __restrict__ bad) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  int64_t a = seg[e], b = nbr[e];
  const bool ok = a >= 0 && a < N && b >= 0 && b < N;       // ids outside [0, N) never reach the kernels
  if (!ok && bad != nullptr) atomicAdd(bad, 1);
  keys[e] = (!ok || (drop_self && a == b)) ? kSentinel : (((uint64_t)a << bits) | (uint64_t)(uint32_t)b);
  vals[e] = (int32_t)e;
}

__global__ void expand_keys_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, int N,
                                   int64_t Ecap, int bits, uint64_t* __restrict__ keys, int32_t* __restrict__ vals) {
  // one thread per node: emits (col << 32 | row) for its CSR segment
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n < N) {
    int rs = rowptr[n], re = rowptr[n + 1];
    for (int e = rs; e < re; ++e) {
      keys[e] = ((uint64_t)(uint32_t)col[e] << bits) | (uint64_t)(uint32_t)n;
      vals[e] = e;
    }
  }
  // tail [rowptr[N], Ecap) is padding
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t total = (int64_t)gridDim.x * blockDim.x;
  int64_t E = rowptr[N];
  for (int64_t e = E + t; e < Ecap; e += total) {
    keys[e] = kSentinel;
    vals[e] = -1;
  }
}

__global__ void unpack_sorted_kernel(const uint64_t* __restrict__ keys, int64_t E, int bits,
                                     int32_t* __restrict__ col) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  uint64_t k = keys[e];
  col[e] = (k == kSentinel) ? -1 : (int32_t)(uint32_t)(k & ((1ull << bits) - 1));
}

// rowptr[n] = first position whose key >= (n << 32)   (n = N gives the valid edge count)
__global__ void rowptr_search_kernel(const uint64_t* __restrict__ keys, int64_t E, int N, int bits,
                                     int32_t* __restrict__ rowptr) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n > N) return;
  uint64_t target = (uint64_t)n << bits;
  int64_t lo = 0, hi = E;
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if (keys[mid] < target) lo = mid + 1; else hi = mid;
  }
  rowptr[n] = (int32_t)lo;
}

__global__ void invert_perm_kernel(const int32_t* __restrict__ pos, int64_t Ecap, const int32_t* __restrict__ count,
                                   int32_t* __restrict__ inv) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= Ecap) return;
  if (e < *count) inv[pos[e]] = (int32_t)e;
}

// For every edge (i -> j) of a (row, col)-sorted CSR: the position of (j -> i), or -1 (and flag |= 1) if
// the reverse edge is missing.  For a symmetric graph the transposed CSR is the CSR itself and this
// index is the edge correspondence the backward pass needs -- no sort.
__global__ void reverse_index_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ row,
                                     const int32_t* __restrict__ col, int64_t E, int32_t* __restrict__ pos_rev,
                                     int32_t* __restrict__ flag) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  int i = row[e], j = col[e];
  int lo = rowptr[j], hi = rowptr[j + 1];
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (col[mid] < i) lo = mid + 1; else hi = mid;
  }
  if (lo < rowptr[j + 1] && col[lo] == i) {
    pos_rev[e] = lo;
  } else {
    pos_rev[e] = -1;
    if (flag) atomicOr(flag, 1);
  }
}

struct SortBuffers {
  uint64_t *k_in, *k_out;
  int32_t *v_in, *v_out;
  void* temp;
  size_t temp_bytes;
};

int carve_sort(Arena& a, int64_t E, SortBuffers& sb) {
  size_t tb = 0;
  if (E > 0) {
    hipError_t err = rocprim::radix_sort_pairs(nullptr, tb, (uint64_t*)nullptr, (uint64_t*)nullptr, (int32_t*)nullptr,
                                               (int32_t*)nullptr, (size_t)E, 0u, 64u, (hipStream_t)0, false);
    if (err != hipSuccess) return set_error("rocprim radix_sort size query failed: %s", hipGetErrorString(err));
  }
  sb.temp_bytes = tb;
  sb.k_in = a.take<uint64_t>(E);
  sb.k_out = a.take<uint64_t>(E);
  sb.v_in = a.take<int32_t>(E);
  sb.v_out = a.take<int32_t>(E);
  sb.temp = a.take<char>(tb ? tb : 1);
  return 0;
}

}  // namespace

size_t csr_ws_bytes(int64_t E, int64_t N) {
  (void)N;
  Arena a(nullptr, 0);
  SortBuffers sb;
  if (carve_sort(a, E, sb) != 0) return 0;
  return align_up(a.off) + 256;
}

int csr_from_coo(const int64_t* seg, const int64_t* nbr, int64_t E, int64_t N, int drop_self, int32_t* rowptr,
                 int32_t* col, int32_t* eid, int32_t* bad, void* ws, size_t ws_bytes, hipStream_t s) {
  GEOBI_REQUIRE(N >= 0 && E >= 0 && N < (1ll << 31) && E < (1ll << 31), "csr_from_coo: sizes out of int32 range");
  if (bad != nullptr) GEOBI_HIP(hipMemsetAsync(bad, 0, sizeof(int32_t), s));
  if (E == 0) {
    GEOBI_HIP(hipMemsetAsync(rowptr, 0, sizeof(int32_t) * (N + 1), s));
    return 0;
  }
  Arena a(ws, ws_bytes);
  SortBuffers sb;
  GEOBI_TRY(carve_sort(a, E, sb));
  GEOBI_REQUIRE(a.ok() && ws != nullptr, "csr_from_coo: workspace too small (%zu < %zu)", ws_bytes, a.off);
  const int T = 256;
  const int bits = key_bits(N);
  make_keys_kernel<<<cdiv(E, T), T, 0, s>>>(seg, nbr, E, N, drop_self, bits, sb.k_in, sb.v_in, bad);
  GEOBI_LAUNCH_OK();
  size_t tb = sb.temp_bytes;
  GEOBI_HIP(rocprim::radix_sort_pairs(sb.temp, tb, sb.k_in, sb.k_out, sb.v_in, eid, (size_t)E, 0u, (unsigned)(2 * bits), s,
                                      false));
  unpack_sorted_kernel<<<cdiv(E, T), T, 0, s>>>(sb.k_out, E, bits, col);
  GEOBI_LAUNCH_OK();
  rowptr_search_kernel<<<cdiv(N + 1, T), T, 0, s>>>(sb.k_out, E, (int)N, bits, rowptr);
  GEOBI_LAUNCH_OK();
  return 0;
}

int csr_reverse_index(const int32_t* rowptr, const int32_t* row, const int32_t* col, int64_t E, int32_t* pos_rev,
                      int32_t* flag, hipStream_t s) {
  if (flag != nullptr) GEOBI_HIP(hipMemsetAsync(flag, 0, sizeof(int32_t), s));
  if (E <= 0) return 0;
  reverse_index_kernel<<<cdiv(E, 256), 256, 0, s>>>(rowptr, row, col, E, pos_rev, flag);
  GEOBI_LAUNCH_OK();
  return 0;
}

// Transposed CSR of a CSR whose valid edge count lives on the device (rowptr[N]); `Ecap` is the
// allocated length of col.  pos_t[e_t] = position of that edge in the input CSR; inv_pos[e] =
// position of input edge e in the transposed CSR.
int csr_transpose(const int32_t* rowptr, const int32_t* col, int64_t N, int64_t Ecap, int32_t* rowptr_t,
                  int32_t* col_t, int32_t* pos_t, int32_t* inv_pos, void* ws, size_t ws_bytes, hipStream_t s) {
  if (Ecap == 0) {
    GEOBI_HIP(hipMemsetAsync(rowptr_t, 0, sizeof(int32_t) * (N + 1), s));
    return 0;
  }
  Arena a(ws, ws_bytes);
  SortBuffers sb;
  GEOBI_TRY(carve_sort(a, Ecap, sb));
  GEOBI_REQUIRE(a.ok() && ws != nullptr, "csr_transpose: workspace too small (%zu < %zu)", ws_bytes, a.off);
  const int T = 256;
  int blocks = cdiv(N > 0 ? N : 1, T);
  const int bits = key_bits(N);
  expand_keys_kernel<<<blocks, T, 0, s>>>(rowptr, col, (int)N, Ecap, bits, sb.k_in, sb.v_in);
  GEOBI_LAUNCH_OK();
  size_t tb = sb.temp_bytes;
  GEOBI_HIP(rocprim::radix_sort_pairs(sb.temp, tb, sb.k_in, sb.k_out, sb.v_in, pos_t, (size_t)Ecap, 0u,
                                      (unsigned)(2 * bits), s, false));
  // low 32 bits of the transposed key hold the original row = the neighbour in the transposed view
  unpack_sorted_kernel<<<cdiv(Ecap, T), T, 0, s>>>(sb.k_out, Ecap, bits, col_t);
  GEOBI_LAUNCH_OK();
  rowptr_search_kernel<<<cdiv(N + 1, T), T, 0, s>>>(sb.k_out, Ecap, (int)N, bits, rowptr_t);
  GEOBI_LAUNCH_OK();
  if (inv_pos) {
    invert_perm_kernel<<<cdiv(Ecap, T), T, 0, s>>>(pos_t, Ecap, rowptr + N, inv_pos);
    GEOBI_LAUNCH_OK();
  }
  return 0;
}


// ---------------------------------------------------------------- batched union copy (geobi_concat32)
namespace {
constexpr int kMaxSegs = 96;
struct ConcatJob {
  const uint32_t* src[kMaxSegs];
  uint32_t* dst[kMaxSegs];
  int64_t start[kMaxSegs + 1];     // element offsets of the jobs in the launch's index space
  uint32_t fill[kMaxSegs];         // int32 add, or the bits of the float fill value
  int n, is_float;
};

// One launch for every array of a union batch: 4 elements per thread and step; the job of an element is found by a
// binary search over <= 96 start offsets held in LDS.
__global__ __launch_bounds__(256) void concat32_kernel(ConcatJob job) {
  __shared__ int64_t s_start[kMaxSegs + 1];
  for (int i = threadIdx.x; i <= job.n; i += 256) s_start[i] = job.start[i];
  __syncthreads();
  const int64_t total = s_start[job.n];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int lo = 0, hi = job.n - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (s_start[mid] <= i) lo = mid; else hi = mid - 1;
    }
    const int64_t k = i - s_start[lo];
    const uint32_t* src = job.src[lo];
    uint32_t v = job.fill[lo];
    if (src != nullptr) v = job.is_float ? src[k] : src[k] + v;
    job.dst[lo][k] = v;
  }
}
}  // namespace

int concat32(const CopySeg* segs, int n_segs, int is_float, hipStream_t s) {
  for (int base = 0; base < n_segs; base += kMaxSegs) {
    ConcatJob job;
    job.n = n_segs - base < kMaxSegs ? n_segs - base : kMaxSegs;
    job.is_float = is_float;
    job.start[0] = 0;
    for (int i = 0; i < job.n; ++i) {
      const CopySeg& g = segs[base + i];
      GEOBI_REQUIRE(g.n >= 0 && (g.n == 0 || g.dst != nullptr), "concat32: bad segment %d", base + i);
      job.src[i] = (const uint32_t*)g.src;
      job.dst[i] = (uint32_t*)g.dst;
      uint32_t bits;
      if (is_float) memcpy(&bits, &g.value, 4); else bits = (uint32_t)g.add;
      job.fill[i] = bits;
      job.start[i + 1] = job.start[i] + g.n;
    }
    const int64_t total = job.start[job.n];
    if (total == 0) continue;
    int64_t blocks = (total + 1023) / 1024;
    if (blocks > 4096) blocks = 4096;
    concat32_kernel<<<(int)blocks, 256, 0, s>>>(job);
    GEOBI_LAUNCH_OK();
  }
  return 0;
}

}  // namespace geobi
