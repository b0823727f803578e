/* TEST INFRASTRUCTURE -- C restatement of the sequential greedy matching of
 * torch_cluster's graclus_cpu (third-party, un-vendored; called at
 * /root/reference/code/net_util.py:127).  Same algorithm as the Python loop in
 * oracle/pyg_ops.py:graclus; exists only so the CPU baseline is not dominated by
 * interpreter overhead.  PARITY UNPINNED (no reference fixture exists). */
#include <stdint.h>

void oracle_graclus_f32(int64_t n, const int64_t* rowptr, const int64_t* col, const float* w,
                        const int64_t* node_perm, int64_t* out) {
  for (int64_t k = 0; k < n; ++k) {
    int64_t u = node_perm[k];
    if (out[u] >= 0) continue;
    int64_t v_max = u;
    float w_max = 0.0f;
    for (int64_t e = rowptr[u]; e < rowptr[u + 1]; ++e) {
      int64_t v = col[e];
      if (out[v] >= 0) continue;
      if (w[e] >= w_max) { v_max = v; w_max = w[e]; }
    }
    int64_t m = u < v_max ? u : v_max;
    out[u] = m;
    out[v_max] = m;
  }
}

/* Greedy matching in globally descending edge order (the deterministic specification
 * the HIP handshake kernel implements: an edge is taken iff it is the heaviest remaining
 * edge at both endpoints, ties broken by smaller min(u,v) then smaller max(u,v)).
 * `order` lists directed CSR edge ids already sorted by that key; `row` is the source
 * node of each CSR edge. */
void oracle_greedy_sorted(int64_t n, int64_t m, const int64_t* order, const int64_t* row,
                          const int64_t* col, int64_t* out) {
  for (int64_t i = 0; i < n; ++i) out[i] = -1;
  for (int64_t k = 0; k < m; ++k) {
    int64_t e = order[k];
    int64_t u = row[e], v = col[e];
    if (u == v || out[u] >= 0 || out[v] >= 0) continue;
    int64_t mn = u < v ? u : v;
    out[u] = mn;
    out[v] = mn;
  }
  for (int64_t i = 0; i < n; ++i) if (out[i] < 0) out[i] = i;
}

/* Sequential statement of the reference's face-ring growth, /root/reference/code/data_util.py:55-84
 * (mesh_get_neighbor_np): the spec the device kernel (geobi_patch_grow) is checked against, itself PINNED by the face
 * lists the reference's own function produced (tests/golden/patches_n8.npz, oracle/gen_golden.py:gen_patches).
 * vf as CSR (rowptr [V+1], list), walked in list order = the order of the reference's padded vf_indices row. */
void oracle_patch_grow(const int32_t* fv, const int32_t* vf_rowptr, const int32_t* vf_list, int64_t F, int64_t seed,
                       int64_t neighbor_count, int64_t ring_count, uint8_t* sel, int32_t* out, int64_t* out_n) {
  if (neighbor_count <= 0) neighbor_count = INT64_MAX;
  if (ring_count <= 0) ring_count = INT64_MAX;
  for (int64_t i = 0; i < F; ++i) sel[i] = 0;
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
            if (n >= neighbor_count) { *out_n = n; return; }
          }
        }
      }
    }
    ok_start = ok_end;
    ok_end = n;
    if (ok_start == ok_end) break;
  }
  *out_n = n;
}
