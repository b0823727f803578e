// Patch split / merge for meshes larger than one network pass.
//   /root/reference/code/dataset.py:156-193          the split loop (seed = farthest unvisited face)
//   /root/reference/code/data_util.py:55-84          mesh_get_neighbor_np: face-ring growth from a seed
//   /root/reference/code/data_util.py:318-336        get_submesh: vertex renumbering in first-use order
//   /root/reference/code/test_dual.py:49-61          overlap merge: sum, count, divide / normalise
// The ring growth is an ordered traversal (the patch is cut in the middle of a ring at `neighbor_count` faces, in
// visiting order); since round 4 it runs on the device as well, ring by ring, with the visiting order recovered from
// slot numbers (patch_grow_kernel).  Its sequential statement lives with the test infrastructure (oracle/oracle_c.c).

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


// ---------------------------------------------------------------------------------------- ring growth on the device
// data_util.mesh_get_neighbor_np (code/data_util.py:55-84) is an ORDERED traversal: ring r + 1 lists, in the order they are
// first met, the unselected faces found by walking ring r's faces in order, their three vertices in order, each vertex's
// incident faces in order; the patch is cut the moment `neighbor_count` faces are listed.
//
// Level-synchronous form, one workgroup per patch (a ring is a few hundred to a few thousand faces: a chip-wide launch per
// ring would cost more in launches than the ring in work).  Number the (ring face q, vertex k) pairs of a ring p = 3 q + k
// ("items") and an item's incidence entries (p, e).  Two facts carry the order:
//   * a vertex only matters where it is met FIRST in the patch: every face around it is selected right there, so later
//     meetings find nothing.  Items whose vertex was expanded in an earlier ring are dropped (a bitmap); among the items of
//     one ring the smallest p per vertex wins (atomicMin of p on the vertex's entry of an LDS hash table).  That leaves
//     ~1/6 of the (q, k, e) triples the sequential loop walks.
//   * a face met by several expanding items is listed where the smallest (p, e) meets it (atomicMin of p W + e on the
//     face's entry of a second table), and its place in the new ring is the number of winning (p, e) before its own: lanes
//     own items in ascending order, so that is a wave prefix sum plus the totals of the waves before.
// The cut is "the first `room` winners".  The current ring's vertex rows sit in LDS; the "selected" / "expanded" flags are
// LDS bitmaps when the mesh fits them (262 144 faces, 131 072 vertices), per-patch stamps in HBM otherwise.  The kernel
// ends by picking the next patch's seed -- the unvisited face farthest from the centroid, lowest id on ties like np.argmax
// (code/dataset.py:159,188-190) -- so a chain of these launches splits a whole mesh with no host step in between.
// Measured on the way (MI355X, 20 000-face patches of a 151 k-face mesh, ~59 rings): all (q, k, e) triples with atomicMin
// on a global per-face key 0.8-1.0 ms per patch; the same with the LDS table 0.8 ms (phase times by s_memtime: the 12.6 k
// incidence loads of a ring 11 k cycles, their hash inserts 11 k, append 5 k, barriers 3 k of 37 k per ring); the host
// loop it replaces 0.37 ms.
constexpr int kGrowThreads = 1024;
constexpr int kGrowRingLds = 2048;        // ring faces whose vertex rows (24 KB) and ids (8 KB) are kept in LDS
constexpr int kGrowIter = 4;              // items per lane and pass
constexpr int kGrowIter2 = 8;             // (vertex, entry) slots per lane and pass
constexpr int kGrowWinners = 2048;        // expanding vertices a pass lists
constexpr int kGrowFaceTab = 4096;        // (face id, first slot) entries: 32 KB
constexpr int kGrowVertTab = 2048;        // (vertex id, first item) entries: 16 KB
constexpr int kGrowFaceBits = 262144, kGrowVertBits = 131072; // 32 KB + 16 KB of LDS bitmaps

// dynamic LDS layout (ints)
constexpr int kOffFv = 0;
constexpr int kOffIds = kOffFv + 3 * kGrowRingLds;       // the ring being appended: face ids (their rows follow at the ring's end)
constexpr int kOffFaceTab = kOffIds + kGrowRingLds;
constexpr int kOffVertTab = kOffFaceTab + 2 * kGrowFaceTab;
constexpr int kOffWinners = kOffVertTab + 2 * kGrowVertTab;
constexpr int kOffFaceBits = kOffWinners + kGrowWinners;
constexpr int kOffVertBits = kOffFaceBits + kGrowFaceBits / 32;
constexpr int kGrowLdsInts = kOffVertBits + kGrowVertBits / 32;
static_assert(kGrowLdsInts * 4 <= 158 * 1024, "the growth kernel's LDS");

// next seed: the unvisited face with the largest d2 (lowest id on ties); state[0] = seed or -1, state[2] = unvisited count.
// scratch: 3 x kGrowThreads words of LDS.
__device__ void grow_pick_seed(const float* __restrict__ d2, const int* __restrict__ visited, int F, int* __restrict__ state,
                               int* scratch) {
  float* s_d = reinterpret_cast<float*>(scratch);
  int* s_f = scratch + kGrowThreads;
  int* s_n = scratch + 2 * kGrowThreads;
  float bd = 0.f;
  int bf = -1, left = 0;
  for (int f0 = threadIdx.x; f0 < F; f0 += 4 * kGrowThreads) {   // four independent (flag, distance) pairs in flight
    int vis[4];
    float dd[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int f = f0 + u * kGrowThreads;
      vis[u] = f < F ? visited[f] : 1;
      dd[u] = f < F ? d2[f] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (vis[u]) continue;
      ++left;
      if (bf < 0 || dd[u] > bd) { bd = dd[u]; bf = f0 + u * kGrowThreads; }     // ascending f per thread: the first maximum stays
    }
  }
  s_d[threadIdx.x] = bd; s_f[threadIdx.x] = bf; s_n[threadIdx.x] = left;
  __syncthreads();
  for (int o = kGrowThreads / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      const float od = s_d[threadIdx.x + o];
      const int of = s_f[threadIdx.x + o];
      const int mf = s_f[threadIdx.x];
      if (of >= 0 && (mf < 0 || od > s_d[threadIdx.x] || (od == s_d[threadIdx.x] && of < mf))) {
        s_d[threadIdx.x] = od; s_f[threadIdx.x] = of;
      }
      s_n[threadIdx.x] += s_n[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { state[0] = s_f[0]; state[2] = s_n[0]; }
}

__global__ __launch_bounds__(kGrowThreads) void patch_grow_init_kernel(const float* __restrict__ d2, int F,
                                                                       int* __restrict__ visited, int* __restrict__ state) {
  __shared__ int scratch[3 * kGrowThreads];
  if (threadIdx.x == 0) { state[1] = 0; state[3] = 0; }
  grow_pick_seed(d2, visited, F, state, scratch);
}

// Hash tables of a pass (LDS, open addressing, `size` a power of two): entry of `id` (claimed if new) with its value lowered
// to `rel`.  The FIRST probe of all of a lane's entries is issued back to back by the caller (grow_tab_first: the LDS
// atomics pipeline; one site after the other, each waiting for its compare-and-swap, was 9 k of a ring's 29 k cycles);
// grow_tab_rest finishes an entry whose first slot was taken by another id.  More than kGrowProbes probes = the table is
// too full: the flag makes the caller redo the pass with fewer items.
constexpr int kGrowProbes = 48;

__device__ __forceinline__ unsigned grow_hash(int id, int shift) { return ((unsigned)id * 2654435761u) >> shift; }

__device__ __forceinline__ int grow_tab_rest(int* t_id, int* t_val, int size, unsigned h, int first_old, int* full, int id,
                                             int rel) {
  int old = first_old;
  for (int probe = 0; probe < kGrowProbes; ++probe) {
    if (old == -1 || old == id) {
      atomicMin(&t_val[h], rel);
      return (int)h;
    }
    h = (h + 1) & (size - 1);
    old = atomicCAS(&t_id[h], -1, id);
  }
  full[1] = 1;
  return -1;
}

// exclusive prefix of one int per lane over the wave; total = the wave's sum
__device__ __forceinline__ int grow_wave_scan(int c, int lane, int& total) {
  int incl = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  total = __shfl(incl, 63, 64);
  return incl - c;
}

template <bool LDSBITS>
__global__ __launch_bounds__(kGrowThreads) void patch_grow_kernel(
    const int* __restrict__ fv, const int* __restrict__ vf, int W, int F, int V, const float* __restrict__ d2, int seed_arg,
    int neighbor_count, int ring_count, int patch_id, int* __restrict__ stamp, int* __restrict__ vstamp,
    int* __restrict__ visited, int* __restrict__ state, int* __restrict__ sel_out, int* __restrict__ n_out, int* mailbox,
    int pick_next) {
  extern __shared__ int s_dyn[];
  int* s_fv = s_dyn + kOffFv;                         // [3 * kGrowRingLds] vertex rows of the ring being walked
  int* s_ids = s_dyn + kOffIds;                       // [kGrowRingLds] ids of the ring being appended
  int* ft_id = s_dyn + kOffFaceTab;                   // face table
  int* ft_val = ft_id + kGrowFaceTab;
  int* vt_id = s_dyn + kOffVertTab;                   // vertex table
  int* vt_val = vt_id + kGrowVertTab;
  int* wl = s_dyn + kOffWinners;                      // this pass' expanding vertices, in item order
  unsigned* fbits = reinterpret_cast<unsigned*>(s_dyn + kOffFaceBits);
  unsigned* vbits = reinterpret_cast<unsigned*>(s_dyn + kOffVertBits);
  __shared__ int s_wave[32];
  __shared__ int s_full[2];                           // [1]: a table (or the winner list) overflowed: redo the pass smaller
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const int seed = seed_arg >= 0 ? seed_arg : state[0];
  if (seed < 0 || seed >= F) {                        // every face has been visited: no patch
    if (t == 0) {
      if (n_out) n_out[0] = 0;
      if (mailbox) { __threadfence_system(); *(volatile int*)mailbox = 1; }
    }
    return;
  }
  if (LDSBITS) {
    for (int i = t; i < kGrowFaceBits / 32 + kGrowVertBits / 32; i += kGrowThreads) fbits[i] = 0u;   // the two bitmaps are adjacent
    __syncthreads();
  }
  auto face_selected = [&](int g) -> bool { return LDSBITS ? ((fbits[g >> 5] >> (g & 31)) & 1u) != 0u : stamp[g] == patch_id; };
  auto vertex_expanded = [&](int v) -> bool { return LDSBITS ? ((vbits[v >> 5] >> (v & 31)) & 1u) != 0u : vstamp[v] == patch_id; };
  auto wave_offsets = [&](int mine, int& woff, int& total) {     // s_wave must hold every wave's count (barrier before)
    woff = 0; total = 0;
#pragma unroll
    for (int w = 0; w < kGrowThreads / 64; ++w) {
      const int c = s_wave[w];
      if (w < wave) woff += c;
      total += c;
    }
    (void)mine;
  };
  int n = 1, ring_start = 0, ring_end = 1;
  bool need_clear = true;                             // the tables are cleared by the lanes that used them; in full only here and after a redone pass
  if (t == 0) {
    sel_out[0] = seed; visited[seed] = 1;
    if (LDSBITS) atomicOr(&fbits[seed >> 5], 1u << (seed & 31)); else stamp[seed] = patch_id;
  }
  if (t < 3) s_fv[t] = fv[3 * (size_t)seed + t];
  __syncthreads();
  bool ring_lds = true;
  for (int ring = 0; ring < ring_count && n < neighbor_count; ++ring) {
    const int R = ring_end - ring_start;
    const long long items = 3ll * R;
    const int* ring_fv = s_fv;
    int added = 0;                                                // faces this ring has appended so far (uniform)
    // items of one pass: the whole ring when it fits (spread over the waves), halved whenever a pass overflows a table
    long long pass_items = items < (long long)kGrowThreads * kGrowIter ? items : (long long)kGrowThreads * kGrowIter;
    long long pass0 = 0;
    while (pass0 < items && n < neighbor_count) {
      const long long pass_end = pass0 + pass_items < items ? pass0 + pass_items : items;
      const int it1 = (int)((pass_end - pass0 + kGrowThreads - 1) / kGrowThreads);
      if (need_clear) {
        for (int i = t; i < kGrowFaceTab; i += kGrowThreads) { ft_id[i] = -1; ft_val[i] = 0x7fffffff; }
        for (int i = t; i < kGrowVertTab; i += kGrowThreads) { vt_id[i] = -1; vt_val[i] = 0x7fffffff; }
        if (t == 0) { s_full[0] = 0; s_full[1] = 0; }
        __syncthreads();
        need_clear = false;
      }
      // ---- 1: the item's vertex; unexpanded vertices enter the vertex table with their item number
      const long long base = pass0 + (long long)wave * (64 * it1) + lane;
      int vtx[kGrowIter], vh[kGrowIter];
#pragma unroll
      for (int i = 0; i < kGrowIter; ++i) {
        const long long p = base + 64 * i;
        vtx[i] = -1; vh[i] = -2;
        if (i < it1 && p < pass_end) {
          const int q = (int)(p / 3), k = (int)(p - 3ll * q);
          const int v = ring_lds ? ring_fv[3 * q + k] : fv[3 * (size_t)sel_out[ring_start + q] + k];
          if (!vertex_expanded(v)) { vtx[i] = v; vh[i] = atomicCAS(&vt_id[grow_hash(v, 21)], -1, v); }
        }
      }
#pragma unroll
      for (int i = 0; i < kGrowIter; ++i)
        vh[i] = vtx[i] >= 0 ? grow_tab_rest(vt_id, vt_val, kGrowVertTab, grow_hash(vtx[i], 21), vh[i], s_full, vtx[i],
                                             (int)(base + 64 * i - pass0))
                            : -1;
      __syncthreads();
      // ---- 1b: items that met their vertex first, compacted in item order -> wl
      unsigned long long fm[kGrowIter];
      int cnt = 0;
#pragma unroll
      for (int i = 0; i < kGrowIter; ++i) {
        const bool first = vh[i] >= 0 && vt_val[vh[i]] == (int)(base + 64 * i - pass0);
        fm[i] = __ballot(first);
        cnt += __popcll(fm[i]);
      }
      if (lane == 0) s_wave[wave] = cnt;
      __syncthreads();
      int woff, nw;
      wave_offsets(cnt, woff, nw);
      const long long slots = (long long)nw * W;
      if (s_full[1] || nw > kGrowWinners || slots > (long long)kGrowThreads * kGrowIter2) {   // too much for one pass
        __syncthreads();
        pass_items = pass_items > 1 ? pass_items / 2 : 1;        // one item expands at most W <= 8192 slots: always fits
        need_clear = true;
        continue;
      }
      {
        int running = woff;
#pragma unroll
        for (int i = 0; i < kGrowIter; ++i) {
          if (vh[i] >= 0) { vt_id[vh[i]] = -1; vt_val[vh[i]] = 0x7fffffff; }       // every reader is past the barrier above
          if ((fm[i] >> lane) & 1ull) {
            wl[running + __popcll(fm[i] & lt_mask)] = vtx[i];
            if (LDSBITS) atomicOr(&vbits[vtx[i] >> 5], 1u << (vtx[i] & 31)); else vstamp[vtx[i]] = patch_id;
          }
          running += __popcll(fm[i]);
        }
      }
      __syncthreads();
      // ---- 2: slots (expanding vertex j, incidence entry e): unselected faces enter the face table with their slot number
      int it2 = (int)((slots + kGrowThreads - 1) / kGrowThreads);
      it2 = it2 < 1 ? 1 : it2;                                    // <= kGrowIter2 (checked above)
      const int base2 = wave * (64 * it2) + lane;
      int gg[kGrowIter2], fh[kGrowIter2];
#pragma unroll
      for (int i = 0; i < kGrowIter2; ++i) {
        const int sl = base2 + 64 * i;
        gg[i] = -1;
        if (i < it2 && sl < slots) {
          const int j = sl / W, e = sl - j * W;
          gg[i] = vf[(size_t)wl[j] * W + e];
        }
      }
#pragma unroll
      for (int i = 0; i < kGrowIter2; ++i) {
        fh[i] = -2;
        if (gg[i] >= 0 && !face_selected(gg[i])) fh[i] = atomicCAS(&ft_id[grow_hash(gg[i], 20)], -1, gg[i]);
        else gg[i] = -1;
      }
#pragma unroll
      for (int i = 0; i < kGrowIter2; ++i)
        fh[i] = gg[i] >= 0 ? grow_tab_rest(ft_id, ft_val, kGrowFaceTab, grow_hash(gg[i], 20), fh[i], s_full, gg[i], base2 + 64 * i)
                           : -1;
      __syncthreads();
      if (s_full[1]) {                                            // more new faces than the table holds: smaller pass
        // the vertices this pass marked expanded have not listed their faces yet: take the marks back
        for (int j = t; j < nw; j += kGrowThreads) {
          const int v = wl[j];
          if (LDSBITS) atomicAnd(&vbits[v >> 5], ~(1u << (v & 31))); else vstamp[v] = 0;
        }
        __syncthreads();
        pass_items = pass_items > 1 ? pass_items / 2 : 1;
        need_clear = true;
        continue;
      }
      // ---- 3: winners in slot order -> positions
      unsigned long long wm[kGrowIter2];
      cnt = 0;
#pragma unroll
      for (int i = 0; i < kGrowIter2; ++i) {
        wm[i] = __ballot(fh[i] >= 0 && ft_val[fh[i]] == base2 + 64 * i);
        cnt += __popcll(wm[i]);
      }
      if (lane == 0) s_wave[wave] = cnt;
      __syncthreads();
      int total;
      wave_offsets(cnt, woff, total);
      // ---- 4: append (ids also into LDS: the ring's end fetches the rows from there, so nothing in this kernel waits for
      //         the global stores below)
      const int room = neighbor_count - n;
      const int take = total < room ? total : room;
      const bool ids_lds = added + take <= kGrowRingLds;
      {
        int running = woff;
#pragma unroll
        for (int i = 0; i < kGrowIter2; ++i) {
          if (fh[i] >= 0) { ft_id[fh[i]] = -1; ft_val[fh[i]] = 0x7fffffff; }        // every reader is past the barrier above
          if ((wm[i] >> lane) & 1ull) {
            const int idx = running + __popcll(wm[i] & lt_mask);
            if (idx < room) {
              const int g = gg[i];
              sel_out[n + idx] = g;
              visited[g] = 1;
              if (ids_lds) s_ids[added + idx] = g;
              if (LDSBITS) atomicOr(&fbits[g >> 5], 1u << (g & 31)); else stamp[g] = patch_id;
            }
          }
          running += __popcll(wm[i]);
        }
      }
      if (LDSBITS) {                                              // LDS traffic only: no need to drain the stores
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
      } else {
        __syncthreads();
      }
      n += take;
      added += take;
      pass0 = pass_end;
    }
    ring_lds = added <= kGrowRingLds;
    ring_start = ring_end;
    ring_end = n;
    if (ring_start == ring_end) break;
    if (ring_lds) {
      // the new ring's vertex rows into LDS, one face per thread and step: ONE round trip for the whole ring
      for (int j = t; j < added; j += kGrowThreads) {
        const int g = s_ids[j];
        int* row = s_fv + 3 * j;
        row[0] = fv[3 * (size_t)g]; row[1] = fv[3 * (size_t)g + 1]; row[2] = fv[3 * (size_t)g + 2];
      }
    }
    __syncthreads();                                              // rows parked; sel_out / stamps visible for the global forms
  }
  __syncthreads();
  if (pick_next) grow_pick_seed(d2, visited, F, state, s_dyn);    // the tables are dead: their LDS is the scratch
  __syncthreads();
  if (t == 0) {
    state[1] += 1;
    if (n_out) n_out[0] = n;
    if (mailbox) { __threadfence_system(); *(volatile int*)mailbox = 1 + 2 * n; }
  }
}

}  // namespace

size_t patch_grow_state_ints(int64_t F, int64_t V) { return (size_t)2 * (size_t)F + (size_t)V + 8; }

// state (int32): [stamp F | visited F | vertex stamp V | seed, patches, unvisited, error, ...]
int patch_grow_init(int32_t* state, int64_t F, int64_t V, const float* d2, hipStream_t s) {
  GEOBI_REQUIRE(F > 0 && V > 0, "patch_grow_init: empty mesh");
  GEOBI_HIP(hipMemsetAsync(state, 0, sizeof(int) * ((size_t)2 * F + V + 8), s));      // no patch has id 0
  patch_grow_init_kernel<<<1, kGrowThreads, 0, s>>>(d2, (int)F, state + F, state + 2 * F + V);
  GEOBI_LAUNCH_OK();
  return 0;
}

int patch_grow(const int32_t* fv, const int32_t* vf, int maxval, int64_t F, int64_t V, const float* d2, int64_t seed,
               int64_t neighbor_count, int64_t ring_count, int patch_id, int32_t* state, int32_t* sel_out, int32_t* n_out,
               int32_t* mailbox, int pick_next, hipStream_t s) {
  GEOBI_REQUIRE(F > 0 && V > 0 && maxval > 0, "patch_grow: empty mesh");
  GEOBI_REQUIRE(patch_id > 0, "patch_grow: patch ids start at 1 (0 marks a face no patch of this split holds)");
  GEOBI_REQUIRE(seed < F, "patch_grow: seed %lld outside [0, %lld)", (long long)seed, (long long)F);
  GEOBI_REQUIRE(maxval <= kGrowIter2 * kGrowThreads, "patch_grow: a vertex with %d incident faces", maxval);
  GEOBI_REQUIRE(pick_next == 0 || d2 != nullptr, "patch_grow: picking the next seed needs d2");
  const int nc = (neighbor_count <= 0 || neighbor_count > F) ? (int)F : (int)neighbor_count;
  const int rc = (ring_count <= 0 || ring_count > F) ? (int)F : (int)ring_count;
  constexpr size_t lds = (size_t)kGrowLdsInts * sizeof(int);
  int32_t* scalars = state + 2 * F + V;
  const bool bits = F <= kGrowFaceBits && V <= kGrowVertBits;
#define GEOBI_GROW(B)                                                                                                  \
  do {                                                                                                                 \
    static std::atomic<bool> attr_set{false};                                                                          \
    if (!attr_set) {                                                                                                   \
      GEOBI_HIP(hipFuncSetAttribute((const void*)patch_grow_kernel<B>, hipFuncAttributeMaxDynamicSharedMemorySize,     \
                                    (int)lds));                                                                        \
      attr_set = true;                                                                                                 \
    }                                                                                                                  \
    patch_grow_kernel<B><<<1, kGrowThreads, lds, s>>>(fv, vf, maxval, (int)F, (int)V, d2, (int)seed, nc, rc, patch_id,  \
                                                      state, state + 2 * F, state + F, scalars, sel_out, n_out,        \
                                                      mailbox, pick_next);                                             \
  } while (0)
  if (bits) GEOBI_GROW(true); else GEOBI_GROW(false);
#undef GEOBI_GROW
  GEOBI_LAUNCH_OK();
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
