// Patch split / merge for meshes larger than one network pass.
//   /root/reference/code/dataset.py:156-193          the split loop (seed = farthest unvisited face)
//   /root/reference/code/data_util.py:55-84          mesh_get_neighbor_np: face-ring growth from a seed
//   /root/reference/code/data_util.py:318-336        get_submesh: vertex renumbering in first-use order
//   /root/reference/code/test_dual.py:49-61          overlap merge: sum, count, divide / normalise
// The ring growth is a strictly ordered traversal (the patch is cut in the middle of a ring at
// `neighbor_count` faces, in visiting order), so it runs on the host over the CSR incidence; everything
// that touches per-vertex / per-face data stays on the device.
#include <vector>

#include "common.h"

namespace geobi {

namespace {

__global__ void submesh_first_use_kernel(const int* __restrict__ fv, const int* __restrict__ sel, int64_t n_sel,
                                         int* __restrict__ first) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= 3 * n_sel) return;
  const int v = fv[3 * (int64_t)sel[p / 3] + (p % 3)];
  atomicMin(&first[v], (int)p);
}

__global__ void submesh_flag_kernel(const int* __restrict__ fv, const int* __restrict__ sel, int64_t n_sel,
                                    const int* __restrict__ first, int* __restrict__ flag) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p == 0) flag[3 * n_sel] = 0;            // scan tail -> number of patch vertices
  if (p >= 3 * n_sel) return;
  const int v = fv[3 * (int64_t)sel[p / 3] + (p % 3)];
  flag[p] = first[v] == (int)p ? 1 : 0;
}

__global__ void submesh_assign_kernel(const int* __restrict__ fv, const int* __restrict__ sel, int64_t n_sel,
                                      const int* __restrict__ first, const int* __restrict__ rank,
                                      int* __restrict__ v_idx, int* __restrict__ f_sub, int* __restrict__ count) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p == 0) count[0] = rank[3 * n_sel];
  if (p >= 3 * n_sel) return;
  const int v = fv[3 * (int64_t)sel[p / 3] + (p % 3)];
  const int fp = first[v];
  const int id = rank[fp];                    // vertices numbered in order of first use
  f_sub[p] = id;
  if (fp == (int)p) v_idx[id] = v;
}

__global__ void patch_accumulate_kernel(const float* __restrict__ vert_p, const float* __restrict__ norm_p,
                                        const int* __restrict__ v_idx, const int* __restrict__ f_idx, int nv, int nf,
                                        float* __restrict__ Vp, float* __restrict__ Np, int* __restrict__ sum_v) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nv) {                               // a patch lists every vertex once: plain read-modify-write
    const int v = v_idx[i];
    Vp[3 * v] += vert_p[3 * i]; Vp[3 * v + 1] += vert_p[3 * i + 1]; Vp[3 * v + 2] += vert_p[3 * i + 2];
    sum_v[v] += 1;
  }
  if (i < nf) {
    const int f = f_idx[i];
    Np[3 * f] += norm_p[3 * i]; Np[3 * f + 1] += norm_p[3 * i + 1]; Np[3 * f + 2] += norm_p[3 * i + 2];
  }
}

__global__ void patch_finalize_kernel(float* __restrict__ Vp, float* __restrict__ Np, const int* __restrict__ sum_v,
                                      int V, int F, float scale, float cx, float cy, float cz) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < V) {
    const float c = (float)sum_v[i];          // 0 for a vertex no face uses: 0/0 like the reference
    Vp[3 * i] = Vp[3 * i] / c / scale + cx;
    Vp[3 * i + 1] = Vp[3 * i + 1] / c / scale + cy;
    Vp[3 * i + 2] = Vp[3 * i + 2] / c / scale + cz;
  }
  if (i < F) {
    const float x = Np[3 * i], y = Np[3 * i + 1], z = Np[3 * i + 2];
    float d = sqrtf((x * x + y * y) + z * z);
    d = d > 1e-12f ? d : 1e-12f;             // torch.nn.functional.normalize eps
    Np[3 * i] = x / d; Np[3 * i + 1] = y / d; Np[3 * i + 2] = z / d;
  }
}

}  // namespace

// HOST function over HOST arrays.  vf as CSR (rowptr [V+1], list), the order of `list` inside a vertex
// is the order the reference walks its padded vf_indices row.
int patch_grow_host(const int32_t* fv, const int32_t* vf_rowptr, const int32_t* vf_list, int64_t F, int64_t seed,
                    int64_t neighbor_count, int64_t ring_count, int32_t* out, int64_t* out_n) {
  GEOBI_REQUIRE(F > 0 && seed >= 0 && seed < F, "patch_grow: seed %lld outside [0, %lld)", (long long)seed, (long long)F);
  if (neighbor_count <= 0) neighbor_count = INT64_MAX;
  if (ring_count <= 0) ring_count = INT64_MAX;
  std::vector<uint8_t> sel((size_t)F, 0);
  int64_t n = 0;
  out[n++] = (int32_t)seed;
  sel[seed] = 1;
  int64_t ok_start = 0, ok_end = 1;
  for (int64_t ring = 0; ring < ring_count; ++ring) {
    for (int64_t q = ok_start; q < ok_end; ++q) {
      const int32_t face = out[q];
      for (int k = 0; k < 3; ++k) {
        const int32_t v = fv[3 * (int64_t)face + k];
        for (int32_t e = vf_rowptr[v]; e < vf_rowptr[v + 1]; ++e) {
          const int32_t g = vf_list[e];
          if (!sel[g]) {
            out[n++] = g;
            sel[g] = 1;
            if (n >= neighbor_count) { *out_n = n; return 0; }
          }
        }
      }
    }
    ok_start = ok_end;
    ok_end = n;
    if (ok_start == ok_end) break;
  }
  *out_n = n;
  return 0;
}

size_t submesh_ws_bytes(int64_t n_sel, int64_t V) {
  return align_up((size_t)V * sizeof(int)) + align_up((size_t)(3 * n_sel + 1) * sizeof(int)) * 2 +
         scan_ws_bytes(3 * n_sel + 1) + 1024;
}

int submesh(const int32_t* fv, const int32_t* sel, int64_t n_sel, int64_t V, int32_t* v_idx, int32_t* f_sub,
            int32_t* count, void* ws, size_t ws_bytes, hipStream_t s) {
  GEOBI_REQUIRE(n_sel > 0 && V > 0, "submesh: empty selection");
  Arena a(ws, ws_bytes);
  int* first = a.take<int>(V);
  int* flag = a.take<int>(3 * n_sel + 1);
  int* rank = a.take<int>(3 * n_sel + 1);
  size_t tb = scan_ws_bytes(3 * n_sel + 1);
  void* temp = a.take<char>(tb);
  GEOBI_REQUIRE(a.ok() && first, "submesh: workspace too small (%zu < %zu)", ws_bytes, a.off);
  GEOBI_HIP(hipMemsetAsync(first, 0x7f, sizeof(int) * V, s));       // 0x7f7f7f7f: larger than any position
  const int blocks = cdiv(3 * n_sel, 256);
  submesh_first_use_kernel<<<blocks, 256, 0, s>>>(fv, sel, n_sel, first);
  GEOBI_LAUNCH_OK();
  submesh_flag_kernel<<<blocks, 256, 0, s>>>(fv, sel, n_sel, first, flag);
  GEOBI_LAUNCH_OK();
  GEOBI_TRY(scan_exclusive_i32(temp, tb, flag, rank, 3 * n_sel + 1, s));
  submesh_assign_kernel<<<blocks, 256, 0, s>>>(fv, sel, n_sel, first, rank, v_idx, f_sub, count);
  GEOBI_LAUNCH_OK();
  return 0;
}

int patch_accumulate(const float* vert_p, const float* norm_p, const int32_t* v_idx, const int32_t* f_idx, int64_t nv,
                     int64_t nf, float* Vp, float* Np, int32_t* sum_v, hipStream_t s) {
  const int64_t n = nv > nf ? nv : nf;
  if (n <= 0) return 0;
  patch_accumulate_kernel<<<cdiv(n, 256), 256, 0, s>>>(vert_p, norm_p, v_idx, f_idx, (int)nv, (int)nf, Vp, Np, sum_v);
  GEOBI_LAUNCH_OK();
  return 0;
}

int patch_finalize(float* Vp, float* Np, const int32_t* sum_v, int64_t V, int64_t F, float scale, float cx, float cy,
                   float cz, hipStream_t s) {
  const int64_t n = V > F ? V : F;
  if (n <= 0) return 0;
  patch_finalize_kernel<<<cdiv(n, 256), 256, 0, s>>>(Vp, Np, sum_v, (int)V, (int)F, scale, cx, cy, cz);
  GEOBI_LAUNCH_OK();
  return 0;
}

}  // namespace geobi
