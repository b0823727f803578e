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
