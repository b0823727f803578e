// Mesh -> hot-path inputs on the device: what the reference computes on the
// CPU with openmesh + PyG before the network runs (/root/reference/code/dataset.py:197-233
// process_one_submesh, code/data_util.py:383-398 calc_weight, :436-456 build_facet_graph).
//   vertex -> face incidence (openmesh vf_indices), face / vertex normals and face centroids,
//   the vertex graph (mesh edges, both directions) and the facet graph (faces sharing a vertex) as
//   (row, col)-sorted CSR without self loops -- the layout the conv / pooling kernels walk --
//   and the bilateral edge weights of both graphs.
// Integer results are exact; the graphs come out sorted and duplicate-free by construction (each node
// repeatedly selects the smallest not-yet-emitted neighbour of its 1-ring), no global sort.
#include "common.h"

namespace geobi {

namespace {

constexpr int kPartials = 256;       // blocks of the edge-length reduction (fixed: deterministic mean)

__global__ void vf_count_kernel(const int* __restrict__ fv, int64_t F3, int V, int* __restrict__ cnt,
                                int* __restrict__ bad) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) cnt[V] = 0;            // scan tail
  if (i >= F3) return;
  int v = fv[i];
  if (v < 0 || v >= V) { atomicAdd(bad, 1); return; }
  atomicAdd(&cnt[v], 1);
}

__global__ void vf_fill_kernel(const int* __restrict__ fv, int64_t F3, int V, const int* __restrict__ rowptr,
                               int* __restrict__ cursor, int* __restrict__ list) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= F3) return;
  int v = fv[i];
  if (v < 0 || v >= V) return;
  list[rowptr[v] + atomicAdd(&cursor[v], 1)] = (int)(i / 3);
}

// ascending face ids per vertex (the atomic cursor fills in arrival order); valences are small
__global__ void vf_sort_kernel(const int* __restrict__ rowptr, int V, int* __restrict__ list) {
  int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  const int b = rowptr[v], e = rowptr[v + 1];
  for (int i = b + 1; i < e; ++i) {
    int key = list[i], j = i - 1;
    while (j >= b && list[j] > key) { list[j + 1] = list[j]; --j; }
    list[j + 1] = key;
  }
}

__global__ void vf_pad_kernel(const int* __restrict__ rowptr, const int* __restrict__ list, int V, int maxval,
                              int* __restrict__ vf) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)V * maxval) return;
  int v = (int)(i / maxval), k = (int)(i % maxval);
  int b = rowptr[v], n = rowptr[v + 1] - b;
  vf[i] = k < n ? list[b + k] : -1;
}

__global__ void max_degree_kernel(const int* __restrict__ rowptr, int N, int* __restrict__ out) {
  int v = blockIdx.x * blockDim.x + threadIdx.x;
  int d = v < N ? rowptr[v + 1] - rowptr[v] : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) d = max(d, __shfl_xor(d, o, 64));
  if ((threadIdx.x & 63) == 0 && d > 0) atomicMax(out, d);
}

__device__ __forceinline__ void face_normal_d(const float* __restrict__ pts, const int* __restrict__ fv, int f,
                                              double (&n)[3]) {
  const int a = fv[3 * f], b = fv[3 * f + 1], c = fv[3 * f + 2];
  double p0[3], e1[3], e2[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    p0[k] = pts[3 * a + k];
    e1[k] = (double)pts[3 * b + k] - p0[k];
    e2[k] = (double)pts[3 * c + k] - p0[k];
  }
  n[0] = e1[1] * e2[2] - e1[2] * e2[1];
  n[1] = e1[2] * e2[0] - e1[0] * e2[2];
  n[2] = e1[0] * e2[1] - e1[1] * e2[0];
  double len = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
  len = len > 1e-12 ? len : 1e-12;
  n[0] /= len; n[1] /= len; n[2] /= len;
}

__global__ void face_geometry_kernel(const float* __restrict__ pts, const int* __restrict__ fv, int F,
                                     float* __restrict__ fnormal, float* __restrict__ centroid) {
  int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  double n[3];
  face_normal_d(pts, fv, f, n);
  const int a = fv[3 * f], b = fv[3 * f + 1], c = fv[3 * f + 2];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    fnormal[3 * f + k] = (float)n[k];
    centroid[3 * f + k] = ((pts[3 * a + k] + pts[3 * b + k]) + pts[3 * c + k]) / 3.0f;   // fp32 mean like torch
  }
}

__global__ void vertex_normal_kernel(const float* __restrict__ pts, const int* __restrict__ fv,
                                     const int* __restrict__ rowptr, const int* __restrict__ list, int V,
                                     float* __restrict__ vnormal) {
  int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  double s[3] = {0.0, 0.0, 0.0};
  for (int e = rowptr[v]; e < rowptr[v + 1]; ++e) {
    double n[3];
    face_normal_d(pts, fv, list[e], n);
    s[0] += n[0]; s[1] += n[1]; s[2] += n[2];
  }
  double len = sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
  len = len > 1e-12 ? len : 1e-12;
#pragma unroll
  for (int k = 0; k < 3; ++k) vnormal[3 * v + k] = (float)(s[k] / len);
}

// 1-ring of a node: KIND 0 = vertices sharing a face with vertex u; KIND 1 = faces sharing a vertex with
// face f.  `visit(e)` is called for every (possibly repeated) ring entry.
template <int KIND, typename Fn>
__device__ __forceinline__ void for_each_ring_entry(const int* __restrict__ fv, const int* __restrict__ rowptr,
                                                    const int* __restrict__ list, int node, Fn visit) {
  if (KIND == 0) {
    for (int e = rowptr[node]; e < rowptr[node + 1]; ++e) {
      const int f = list[e];
#pragma unroll
      for (int k = 0; k < 3; ++k) visit(fv[3 * f + k]);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int v = fv[3 * node + k];
      for (int e = rowptr[v]; e < rowptr[v + 1]; ++e) visit(list[e]);
    }
  }
}

// PASS 0: cnt[node] = number of distinct ring neighbours (the node itself excluded).
// PASS 1: col[rowptr_g[node] ..] = those neighbours, ascending.
template <int KIND, int PASS>
__global__ void ring_graph_kernel(const int* __restrict__ fv, const int* __restrict__ rowptr,
                                  const int* __restrict__ list, int n_nodes, int* __restrict__ cnt,
                                  const int* __restrict__ rowptr_g, int* __restrict__ col) {
  int node = blockIdx.x * blockDim.x + threadIdx.x;
  if (PASS == 0 && node == 0) cnt[n_nodes] = 0;      // scan tail
  if (node >= n_nodes) return;
  int last = -1, pos = 0;
  const int base = PASS == 1 ? rowptr_g[node] : 0;
  while (true) {
    int m = 0x7fffffff;
    for_each_ring_entry<KIND>(fv, rowptr, list, node, [&](int e) {
      if (e != node && e > last && e < m) m = e;
    });
    if (m == 0x7fffffff) break;
    if (PASS == 1) col[base + pos] = m;
    ++pos;
    last = m;
  }
  if (PASS == 0) cnt[node] = pos;
}

// sum of edge lengths, fixed blocking -> deterministic; double accumulation
__global__ __launch_bounds__(256) void edge_length_partial_kernel(const float* __restrict__ pos,
                                                                  const int* __restrict__ row,
                                                                  const int* __restrict__ col, int64_t E,
                                                                  double* __restrict__ partial) {
  __shared__ double red[256];
  double s = 0.0;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < E; e += (int64_t)gridDim.x * 256) {
    const int i = row[e], j = col[e];
    const float dx = pos[3 * i] - pos[3 * j], dy = pos[3 * i + 1] - pos[3 * j + 1], dz = pos[3 * i + 2] - pos[3 * j + 2];
    s += (double)sqrtf((dx * dx + dy * dy) + dz * dz);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__device__ __forceinline__ float mean_from_partials(const double* __restrict__ partial, int64_t denom) {
  double s = 0.0;
  for (int b = 0; b < kPartials; ++b) s += partial[b];
  return (float)(s / (double)(denom > 0 ? denom : 1));
}

__global__ void mean_edge_length_kernel(const double* __restrict__ partial, int64_t denom, float* __restrict__ out) {
  if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = mean_from_partials(partial, denom);
}

// calc_weight (data_util.py:383-398): clamp(n_i . n_j, 1e-3) * exp(|dp|^2 / (-2 mean|dp| + 1e-12))
__global__ void calc_weight_kernel(const float* __restrict__ pos, const float* __restrict__ nrm,
                                   const int* __restrict__ row, const int* __restrict__ col, int64_t E,
                                   const double* __restrict__ partial, int64_t denom, float* __restrict__ w) {
  const float mean = mean_from_partials(partial, denom);
  const float den = -2.0f * mean + 1e-12f;
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int i = row[e], j = col[e];
  const float dx = pos[3 * i] - pos[3 * j], dy = pos[3 * i + 1] - pos[3 * j + 1], dz = pos[3 * i + 2] - pos[3 * j + 2];
  const float len2 = (dx * dx + dy * dy) + dz * dz;
  float dn = (nrm[3 * i] * nrm[3 * j] + nrm[3 * i + 1] * nrm[3 * j + 1]) + nrm[3 * i + 2] * nrm[3 * j + 2];
  dn = dn > 0.001f ? dn : 0.001f;
  w[e] = dn * expf(len2 / den);
}

// The same two kernels over a disjoint union of meshes ("parts": node ranges node_ptr[p] .. node_ptr[p+1], whose edges are
// the CSR rows of those nodes): every part gets ITS OWN mean edge length, formed with the blocking the single-mesh kernels use
// (block b of the part sums the part's edges b * 256 + t, + 65 536, ...), so a part's weights are the bits the part alone gives.
__global__ __launch_bounds__(256) void edge_length_partial_parts_kernel(const float* __restrict__ pos,
                                                                        const int* __restrict__ rowptr,
                                                                        const int* __restrict__ row,
                                                                        const int* __restrict__ col,
                                                                        const int* __restrict__ node_ptr,
                                                                        double* __restrict__ partial) {
  __shared__ double red[256];
  const int part = blockIdx.y;
  const int64_t e0 = rowptr[node_ptr[part]], E = (int64_t)rowptr[node_ptr[part + 1]] - e0;
  double s = 0.0;
  for (int64_t el = (int64_t)blockIdx.x * 256 + threadIdx.x; el < E; el += (int64_t)gridDim.x * 256) {
    const int64_t e = e0 + el;
    const int i = row[e], j = col[e];
    const float dx = pos[3 * i] - pos[3 * j], dy = pos[3 * i + 1] - pos[3 * j + 1], dz = pos[3 * i + 2] - pos[3 * j + 2];
    s += (double)sqrtf((dx * dx + dy * dy) + dz * dz);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[(size_t)part * kPartials + blockIdx.x] = red[0];
}

__global__ void calc_weight_parts_kernel(const float* __restrict__ pos, const float* __restrict__ nrm,
                                         const int* __restrict__ rowptr, const int* __restrict__ row,
                                         const int* __restrict__ col, int64_t E, const int* __restrict__ node_ptr, int n_parts,
                                         const double* __restrict__ partial, float* __restrict__ w) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int i = row[e], j = col[e];
  int part = 0;
  while (part + 1 < n_parts && i >= node_ptr[part + 1]) ++part;
  const int a = node_ptr[part], b = node_ptr[part + 1];
  const int64_t denom = ((int64_t)rowptr[b] - rowptr[a]) + (b - a);        // the part's edges + one zero-length loop per node
  const float mean = mean_from_partials(partial + (size_t)part * kPartials, denom);
  const float den = -2.0f * mean + 1e-12f;
  const float dx = pos[3 * i] - pos[3 * j], dy = pos[3 * i + 1] - pos[3 * j + 1], dz = pos[3 * i + 2] - pos[3 * j + 2];
  const float len2 = (dx * dx + dy * dy) + dz * dz;
  float dn = (nrm[3 * i] * nrm[3 * j] + nrm[3 * i + 1] * nrm[3 * j + 1]) + nrm[3 * i + 2] * nrm[3 * j + 2];
  dn = dn > 0.001f ? dn : 0.001f;
  w[e] = dn * expf(len2 / den);
}

}  // namespace

size_t vertex_faces_ws_bytes(int64_t F, int64_t V) {
  return align_up((size_t)(V + 1) * sizeof(int)) * 2 + scan_ws_bytes(V + 1) + 1024;
}

int vertex_faces(const int32_t* fv, int64_t F, int64_t V, int32_t* rowptr, int32_t* list, void* ws, size_t ws_bytes,
                 hipStream_t s) {
  GEOBI_REQUIRE(F >= 0 && V > 0, "vertex_faces: empty mesh");
  Arena a(ws, ws_bytes);
  int* cnt = a.take<int>(V + 1);
  int* cursor = a.take<int>(V + 1);      // cursor[V] doubles as the bad-index counter
  size_t tb = scan_ws_bytes(V + 1);
  void* temp = a.take<char>(tb);
  GEOBI_REQUIRE(a.ok() && cnt, "vertex_faces: workspace too small (%zu < %zu)", ws_bytes, a.off);
  GEOBI_HIP(hipMemsetAsync(cnt, 0, sizeof(int) * (V + 1), s));
  GEOBI_HIP(hipMemsetAsync(cursor, 0, sizeof(int) * (V + 1), s));
  const int64_t F3 = 3 * F;
  const int blocks = cdiv(F3 > 0 ? F3 : 1, 256);
  vf_count_kernel<<<blocks, 256, 0, s>>>(fv, F3, (int)V, cnt, cursor + V);
  GEOBI_LAUNCH_OK();
  GEOBI_TRY(scan_exclusive_i32(temp, tb, cnt, rowptr, V + 1, s));
  vf_fill_kernel<<<blocks, 256, 0, s>>>(fv, F3, (int)V, rowptr, cursor, list);
  GEOBI_LAUNCH_OK();
  vf_sort_kernel<<<cdiv(V, 128), 128, 0, s>>>(rowptr, (int)V, list);
  GEOBI_LAUNCH_OK();
  return 0;
}

int vf_padded(const int32_t* rowptr, const int32_t* list, int64_t V, int maxval, int32_t* vf, hipStream_t s) {
  if (V <= 0 || maxval <= 0) return 0;
  vf_pad_kernel<<<cdiv(V * maxval, 256), 256, 0, s>>>(rowptr, list, (int)V, maxval, vf);
  GEOBI_LAUNCH_OK();
  return 0;
}

int max_degree(const int32_t* rowptr, int64_t N, int32_t* out, hipStream_t s) {
  GEOBI_HIP(hipMemsetAsync(out, 0, sizeof(int), s));
  if (N <= 0) return 0;
  max_degree_kernel<<<cdiv(N, 256), 256, 0, s>>>(rowptr, (int)N, out);
  GEOBI_LAUNCH_OK();
  return 0;
}

int mesh_normals(const float* points, const int32_t* fv, int64_t F, int64_t V, const int32_t* rowptr,
                 const int32_t* list, float* fnormal, float* centroid, float* vnormal, hipStream_t s) {
  if (F > 0) {
    face_geometry_kernel<<<cdiv(F, 256), 256, 0, s>>>(points, fv, (int)F, fnormal, centroid);
    GEOBI_LAUNCH_OK();
  }
  if (V > 0 && vnormal != nullptr) {
    vertex_normal_kernel<<<cdiv(V, 128), 128, 0, s>>>(points, fv, rowptr, list, (int)V, vnormal);
    GEOBI_LAUNCH_OK();
  }
  return 0;
}

size_t ring_graph_ws_bytes(int64_t n_nodes) { return align_up((size_t)(n_nodes + 1) * sizeof(int)) + scan_ws_bytes(n_nodes + 1) + 512; }

int ring_graph_count(int kind, const int32_t* fv, const int32_t* rowptr_vf, const int32_t* list, int64_t n_nodes,
                     int32_t* rowptr_g, void* ws, size_t ws_bytes, hipStream_t s) {
  GEOBI_REQUIRE(kind == 0 || kind == 1, "ring_graph: kind is 0 (vertex graph) or 1 (facet graph)");
  GEOBI_REQUIRE(n_nodes > 0, "ring_graph: empty");
  Arena a(ws, ws_bytes);
  int* cnt = a.take<int>(n_nodes + 1);
  size_t tb = scan_ws_bytes(n_nodes + 1);
  void* temp = a.take<char>(tb);
  GEOBI_REQUIRE(a.ok() && cnt, "ring_graph: workspace too small (%zu < %zu)", ws_bytes, a.off);
  const int blocks = cdiv(n_nodes, 128);
  if (kind == 0)
    ring_graph_kernel<0, 0><<<blocks, 128, 0, s>>>(fv, rowptr_vf, list, (int)n_nodes, cnt, nullptr, nullptr);
  else
    ring_graph_kernel<1, 0><<<blocks, 128, 0, s>>>(fv, rowptr_vf, list, (int)n_nodes, cnt, nullptr, nullptr);
  GEOBI_LAUNCH_OK();
  GEOBI_TRY(scan_exclusive_i32(temp, tb, cnt, rowptr_g, n_nodes + 1, s));
  return 0;
}

int ring_graph_fill(int kind, const int32_t* fv, const int32_t* rowptr_vf, const int32_t* list, int64_t n_nodes,
                    const int32_t* rowptr_g, int32_t* col, hipStream_t s) {
  GEOBI_REQUIRE(kind == 0 || kind == 1, "ring_graph: kind is 0 (vertex graph) or 1 (facet graph)");
  if (n_nodes <= 0) return 0;
  const int blocks = cdiv(n_nodes, 128);
  if (kind == 0)
    ring_graph_kernel<0, 1><<<blocks, 128, 0, s>>>(fv, rowptr_vf, list, (int)n_nodes, nullptr, rowptr_g, col);
  else
    ring_graph_kernel<1, 1><<<blocks, 128, 0, s>>>(fv, rowptr_vf, list, (int)n_nodes, nullptr, rowptr_g, col);
  GEOBI_LAUNCH_OK();
  return 0;
}

size_t calc_weight_ws_bytes() { return align_up(kPartials * sizeof(double)) + 256; }

int calc_weight(const float* pos, const float* normal, const int32_t* row, const int32_t* col, int64_t E,
                int64_t extra_zero_edges, float* w, float* mean_len, void* ws, size_t ws_bytes, hipStream_t s) {
  Arena a(ws, ws_bytes);
  double* partial = a.take<double>(kPartials);
  GEOBI_REQUIRE(a.ok() && partial, "calc_weight: workspace too small");
  edge_length_partial_kernel<<<kPartials, 256, 0, s>>>(pos, row, col, E, partial);
  GEOBI_LAUNCH_OK();
  const int64_t denom = E + extra_zero_edges;
  if (mean_len != nullptr) {
    mean_edge_length_kernel<<<1, 64, 0, s>>>(partial, denom, mean_len);
    GEOBI_LAUNCH_OK();
  }
  if (w != nullptr && E > 0) {
    calc_weight_kernel<<<cdiv(E, 256), 256, 0, s>>>(pos, normal, row, col, E, partial, denom, w);
    GEOBI_LAUNCH_OK();
  }
  return 0;
}

size_t calc_weight_parts_ws_bytes(int n_parts) { return align_up((size_t)(n_parts > 0 ? n_parts : 1) * kPartials * sizeof(double)) + 256; }

int calc_weight_parts(const float* pos, const float* normal, const int32_t* rowptr, const int32_t* row, const int32_t* col,
                      int64_t E, const int32_t* node_ptr, int n_parts, float* w, void* ws, size_t ws_bytes, hipStream_t s) {
  GEOBI_REQUIRE(n_parts > 0 && n_parts <= 4096, "calc_weight_parts: 1 .. 4096 parts (got %d)", n_parts);
  Arena a(ws, ws_bytes);
  double* partial = a.take<double>((size_t)n_parts * kPartials);
  GEOBI_REQUIRE(a.ok() && partial, "calc_weight_parts: workspace too small");
  if (E <= 0) return 0;
  edge_length_partial_parts_kernel<<<dim3(kPartials, n_parts), 256, 0, s>>>(pos, rowptr, row, col, node_ptr, partial);
  GEOBI_LAUNCH_OK();
  calc_weight_parts_kernel<<<cdiv(E, 256), 256, 0, s>>>(pos, normal, rowptr, row, col, E, node_ptr, n_parts, partial, w);
  GEOBI_LAUNCH_OK();
  return 0;
}

}  // namespace geobi
