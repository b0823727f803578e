// Geometry coupling between the two domains and the two regression heads.
//
//   face_geom   /root/reference/code/network.py:332-337 + data_util.py:182-198:
//               x_f = cat(x_f[:, :6], centroid(pred. verts), unit normal(pred. verts));
//               backward scatters into the vertices through vertex -> (face, corner) inverse
//               lists (segment_sum in pool.hip), not atomics.
//   head        network.py:324-343: Linear(32,1024) + leaky_relu(0.2) + Linear(1024, 1|3), then
//               vertex head: (* depth_direction) + xyz;   face head: F.normalize(dim=1).
#include "common.h"

namespace geobi {

namespace {

constexpr float kNormEps = 1e-12f;   // torch.nn.functional.normalize default eps

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 ld3(const float* p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 mul(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

__global__ void face_geom_fwd_kernel(const float* __restrict__ verts, const int* __restrict__ fv,
                                     const float* __restrict__ xf, int ldxf, int F, float* __restrict__ out) {
  int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  V3 v0 = ld3(verts + 3 * (size_t)fv[3 * f]), v1 = ld3(verts + 3 * (size_t)fv[3 * f + 1]),
     v2 = ld3(verts + 3 * (size_t)fv[3 * f + 2]);
  V3 cen = add(add(v0, v1), v2);
  cen = {cen.x / 3.0f, cen.y / 3.0f, cen.z / 3.0f};
  V3 n = cross(sub(v1, v0), sub(v2, v0));
  float len = fmaxf(sqrtf(dot(n, n)), kNormEps);
  float* o = out + (size_t)f * 12;
  const float* xi = xf + (size_t)f * ldxf;
#pragma unroll
  for (int c = 0; c < 6; ++c) o[c] = xi[c];
  o[6] = cen.x; o[7] = cen.y; o[8] = cen.z;
  o[9] = n.x / len; o[10] = n.y / len; o[11] = n.z / len;
}

// per-corner gradients cg[3f + corner][3] from g[f][6:12]
__global__ void face_geom_bwd_kernel(const float* __restrict__ verts, const int* __restrict__ fv,
                                     const float* __restrict__ g, int F, float* __restrict__ cg) {
  int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  V3 v0 = ld3(verts + 3 * (size_t)fv[3 * f]), v1 = ld3(verts + 3 * (size_t)fv[3 * f + 1]),
     v2 = ld3(verts + 3 * (size_t)fv[3 * f + 2]);
  const float* gr = g + (size_t)f * 12;
  V3 gc = {gr[6] / 3.0f, gr[7] / 3.0f, gr[8] / 3.0f};
  V3 gn = {gr[9], gr[10], gr[11]};
  V3 a = sub(v1, v0), b = sub(v2, v0);
  V3 n = cross(a, b);
  float len = sqrtf(dot(n, n));
  V3 dn;
  if (len > kNormEps) {
    V3 nh = mul(n, 1.0f / len);
    dn = mul(sub(gn, mul(nh, dot(nh, gn))), 1.0f / len);
  } else {
    dn = mul(gn, 1.0f / kNormEps);   // clamped branch of normalize: n / eps
  }
  V3 da = cross(b, dn), db = cross(dn, a);   // d(a x b): da = b x dn, db = dn x a
  V3 d1 = add(gc, da), d2 = add(gc, db), d0 = sub(sub(gc, da), db);
  float* o = cg + (size_t)f * 9;
  o[0] = d0.x; o[1] = d0.y; o[2] = d0.z;
  o[3] = d1.x; o[4] = d1.y; o[5] = d1.z;
  o[6] = d2.x; o[7] = d2.y; o[8] = d2.z;
}

// ------------------------------------------------------------------------------ heads
// one wave per node: raw[c] = h[n,:] . W2[c,:] + b2[c]; then the head-specific finish.
// mode 0 (vertex): out = raw (* dd) + resid      mode 1 (face): out = raw / max(|raw|, eps)
template <int NOUT>
__global__ __launch_bounds__(256) void head_out_kernel(const float* __restrict__ h, int K,
                                                       const float* __restrict__ w2, const float* __restrict__ b2,
                                                       int N, int mode, const float* __restrict__ dd,
                                                       const float* __restrict__ resid, int ld_resid,
                                                       float* __restrict__ raw, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 4 + wave;
  if (n >= N) return;
  float acc[NOUT];
#pragma unroll
  for (int c = 0; c < NOUT; ++c) acc[c] = 0.f;
  const float* hr = h + (size_t)n * K;
  for (int k = lane * 4; k < K; k += 256) {
    float4 hv = *reinterpret_cast<const float4*>(hr + k);
#pragma unroll
    for (int c = 0; c < NOUT; ++c) {
      float4 wv = *reinterpret_cast<const float4*>(w2 + (size_t)c * K + k);
      acc[c] = fmaf(hv.x, wv.x, acc[c]); acc[c] = fmaf(hv.y, wv.y, acc[c]);
      acc[c] = fmaf(hv.z, wv.z, acc[c]); acc[c] = fmaf(hv.w, wv.w, acc[c]);
    }
  }
#pragma unroll
  for (int c = 0; c < NOUT; ++c) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc[c] += __shfl_xor(acc[c], m, 64);
    acc[c] += b2[c];
  }
  if (lane != 0) return;
#pragma unroll
  for (int c = 0; c < NOUT; ++c) raw[(size_t)n * NOUT + c] = acc[c];
  float o[3];
  if (mode == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float v = (NOUT == 3) ? acc[c % NOUT] : acc[0] * dd[(size_t)n * 3 + c];
      o[c] = v + resid[(size_t)n * ld_resid + c];
    }
  } else {
    float len = fmaxf(sqrtf(acc[0] * acc[0] + acc[1 % NOUT] * acc[1 % NOUT] + acc[2 % NOUT] * acc[2 % NOUT]), kNormEps);
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = acc[c % NOUT] / len;
  }
  out[(size_t)n * 3] = o[0]; out[(size_t)n * 3 + 1] = o[1]; out[(size_t)n * 3 + 2] = o[2];
}

// gradient of the finish: graw from gout
__global__ void head_finish_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ raw, int nout,
                                       int mode, const float* __restrict__ dd, int N, float* __restrict__ graw) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float* g = gout + (size_t)n * 3;
  if (mode == 0) {
    if (nout == 3) {
      graw[(size_t)n * 3] = g[0]; graw[(size_t)n * 3 + 1] = g[1]; graw[(size_t)n * 3 + 2] = g[2];
    } else {
      const float* d = dd + (size_t)n * 3;
      graw[n] = g[0] * d[0] + g[1] * d[1] + g[2] * d[2];
    }
  } else {
    const float* v = raw + (size_t)n * 3;
    float len = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    float* o = graw + (size_t)n * 3;
    if (len > kNormEps) {
      float inv = 1.0f / len;
      float y0 = v[0] * inv, y1 = v[1] * inv, y2 = v[2] * inv;
      float yg = y0 * g[0] + y1 * g[1] + y2 * g[2];
      o[0] = (g[0] - y0 * yg) * inv; o[1] = (g[1] - y1 * yg) * inv; o[2] = (g[2] - y2 * yg) * inv;
    } else {
      o[0] = g[0] / kNormEps; o[1] = g[1] / kNormEps; o[2] = g[2] / kNormEps;
    }
  }
}

// dh[n,k] = (sum_c graw[n,c] W2[c,k]) * lrelu'(h[n,k])
__global__ void head_dh_kernel(const float* __restrict__ graw, int nout, const float* __restrict__ w2,
                               const float* __restrict__ h, int K, float slope, int64_t total,
                               float* __restrict__ dh) {
  int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (t >= total) return;
  int64_t n = t / K;
  int k = (int)(t % K);
  float4 hv = *reinterpret_cast<const float4*>(h + t);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int c = 0; c < nout; ++c) {
    float gr = graw[n * nout + c];
    float4 wv = *reinterpret_cast<const float4*>(w2 + (size_t)c * K + k);
    s.x = fmaf(gr, wv.x, s.x); s.y = fmaf(gr, wv.y, s.y); s.z = fmaf(gr, wv.z, s.z); s.w = fmaf(gr, wv.w, s.w);
  }
  s.x = hv.x > 0.f ? s.x : s.x * slope; s.y = hv.y > 0.f ? s.y : s.y * slope;
  s.z = hv.z > 0.f ? s.z : s.z * slope; s.w = hv.w > 0.f ? s.w : s.w * slope;
  *reinterpret_cast<float4*>(dh + t) = s;
}

// ------------------------------------------------------------------- losses and metrics
// network.py:364-413.  One pass per call: per-row term, optional per-row weight (per-mesh means of a
// disjoint-union batch), block partials, fixed-order final sum (deterministic, no atomics).
//   kind 0: L1  sum_c |a - b|      kind 1: L2  sum_c (a - b)^2         (loss_v / loss_n)
//   kind 2: Euclidean distance     kind 3: angle in degrees, acos(clamp(1 - |a - b|^2 / 2))   (error_v / error_n)
__device__ __forceinline__ float row_term(const float* __restrict__ a, const float* __restrict__ b, int kind) {
  float d0 = a[0] - b[0], d1 = a[1] - b[1], d2 = a[2] - b[2];
  if (kind == 0) return fabsf(d0) + fabsf(d1) + fabsf(d2);
  float sq = d0 * d0 + d1 * d1 + d2 * d2;
  if (kind == 1) return sq;
  if (kind == 2) return sqrtf(sq);
  float v = fminf(fmaxf(1.0f - sq * 0.5f, -1.0f), 1.0f);
  return acosf(v) * 57.29577951308232f;
}

__global__ __launch_bounds__(256) void row_loss_partial_kernel(const float* __restrict__ a,
                                                               const float* __restrict__ b,
                                                               const float* __restrict__ w, int64_t n, int kind,
                                                               float* __restrict__ partial) {
  __shared__ float red[256];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float t = row_term(a + 3 * i, b + 3 * i, kind);
    s += w ? t * w[i] : t;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int m = 128; m >= 1; m >>= 1) {
    if (threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ void row_loss_final_kernel(const float* __restrict__ partial, int blocks, float scale,
                                      float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float s = 0.f;
  for (int b = 0; b < blocks; ++b) s += partial[b];
  out[0] = s * scale;
}

// d loss / d a  for kinds 0 and 1:  g[i, c] = gout * scale * w_i * (sign(d) | 2 d)
__global__ void row_loss_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                    const float* __restrict__ w, const float* __restrict__ gout, float scale,
                                    int64_t n, int kind, float* __restrict__ ga) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 3 * n) return;
  float d = a[t] - b[t];
  float g = (gout ? gout[0] : 1.0f) * scale * (w ? w[t / 3] : 1.0f);       // gout == NULL: the loss is the root
  // torch: d|x|/dx = sign(x) with sign(0) = 0
  ga[t] = kind == 0 ? g * (d > 0.f ? 1.0f : (d < 0.f ? -1.0f : 0.f)) : g * 2.0f * d;
}

// ---------------------------------------------------------------- vertex update (f1)
__global__ void centroid_kernel(const float* __restrict__ pts, const int* __restrict__ fv, int F,
                                float* __restrict__ cent) {
  int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  V3 c = add(add(ld3(pts + 3 * (size_t)fv[3 * f]), ld3(pts + 3 * (size_t)fv[3 * f + 1])),
             ld3(pts + 3 * (size_t)fv[3 * f + 2]));
  cent[3 * (size_t)f] = c.x / 3.0f; cent[3 * (size_t)f + 1] = c.y / 3.0f; cent[3 * (size_t)f + 2] = c.z / 3.0f;
}

// p_v += 1/max(cnt,1) * sum_{f adj v} n_f (n_f . (c_f - p_v))      (optionally projected on dd_v)
// cent == nullptr: the centroid of every adjacent face is formed on the fly from the OLD positions (same
// expression as centroid_kernel, so the same bits) -- one launch per sweep instead of two, which is what the
// 60-sweep loop is bound by.
// A sweep is a chain of dependent loads per vertex (face list -> corner ids -> corner positions), 60 sweeps in a row:
// the loads of up to VB adjacent faces are issued step by step -- all face ids, then all corner ids and normals, then
// all corner positions -- instead of walking the faces one by one with three dependent loads each.  The sums keep the
// face order (same bits as the face-by-face form).
__global__ __launch_bounds__(256) void vertex_update_kernel(const float* __restrict__ pts, const float* __restrict__ cent,
                                     const int* __restrict__ fv, const float* __restrict__ nrm,
                                     const int* __restrict__ vf, int maxval, const float* __restrict__ dd, int V,
                                     float* __restrict__ out) {
  constexpr int VB = 8;
  int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  V3 p = ld3(pts + 3 * (size_t)v);
  V3 s = {0.f, 0.f, 0.f};
  int cnt = 0;
  for (int a0 = 0; a0 < maxval; a0 += VB) {
    int f[VB];
#pragma unroll
    for (int i = 0; i < VB; ++i) f[i] = a0 + i < maxval ? vf[(size_t)v * maxval + a0 + i] : -1;
    int c0[VB], c1[VB], c2[VB];
    V3 n[VB], c[VB];
#pragma unroll
    for (int i = 0; i < VB; ++i) {
      const int fi = f[i] >= 0 ? f[i] : 0;
      n[i] = ld3(nrm + 3 * (size_t)fi);
      if (cent != nullptr) {
        c[i] = ld3(cent + 3 * (size_t)fi);
      } else {
        c0[i] = fv[3 * fi]; c1[i] = fv[3 * fi + 1]; c2[i] = fv[3 * fi + 2];
      }
    }
    if (cent == nullptr) {
#pragma unroll
      for (int i = 0; i < VB; ++i) {
        V3 t = add(add(ld3(pts + 3 * (size_t)c0[i]), ld3(pts + 3 * (size_t)c1[i])), ld3(pts + 3 * (size_t)c2[i]));
        c[i].x = t.x / 3.0f; c[i].y = t.y / 3.0f; c[i].z = t.z / 3.0f;
      }
    }
#pragma unroll
    for (int i = 0; i < VB; ++i) {
      if (f[i] < 0) continue;
      ++cnt;
      float d = dot(n[i], sub(c[i], p));
      s = add(s, mul(n[i], d));
    }
  }
  float inv = 1.0f / (float)(cnt > 0 ? cnt : 1);
  s = mul(s, inv);
  if (dd) {
    V3 d = ld3(dd + 3 * (size_t)v);
    s = mul(d, dot(s, d));
  }
  out[3 * (size_t)v] = p.x + s.x; out[3 * (size_t)v + 1] = p.y + s.y; out[3 * (size_t)v + 2] = p.z + s.z;
}

}  // namespace

static int loss_blocks(int64_t n) {
  int64_t b = (n + 1023) / 1024;
  return (int)(b < 1 ? 1 : (b > 512 ? 512 : b));
}

size_t row_loss_ws_bytes(int64_t n) { return align_up((size_t)loss_blocks(n) * sizeof(float)) + 256; }

int row_loss_fwd(const float* a, const float* b, const float* w, int64_t n, int kind, float scale, float* out,
                 void* ws, size_t ws_bytes, hipStream_t s) {
  GEOBI_REQUIRE(n > 0 && kind >= 0 && kind <= 3, "row_loss: bad arguments");
  Arena ar(ws, ws_bytes);
  int blocks = loss_blocks(n);
  float* partial = ar.take<float>(blocks);
  GEOBI_REQUIRE(ar.ok() && partial, "row_loss: workspace too small");
  row_loss_partial_kernel<<<blocks, 256, 0, s>>>(a, b, w, n, kind, partial);
  row_loss_final_kernel<<<1, 64, 0, s>>>(partial, blocks, scale, out);
  GEOBI_LAUNCH_OK();
  return 0;
}

int row_loss_bwd(const float* a, const float* b, const float* w, const float* gout, int64_t n, int kind,
                 float scale, float* ga, hipStream_t s) {
  GEOBI_REQUIRE(n > 0 && (kind == 0 || kind == 1), "row_loss_bwd: only L1 / L2 are differentiable here");
  row_loss_bwd_kernel<<<cdiv(3 * n, 256), 256, 0, s>>>(a, b, w, gout, scale, n, kind, ga);
  GEOBI_LAUNCH_OK();
  return 0;
}

size_t update_position_ws_bytes(int64_t V, int64_t F) {
  return align_up((size_t)F * 3 * sizeof(float)) + align_up((size_t)V * 3 * sizeof(float)) + 512;
}

int update_position2(const float* points, const int32_t* fv, const int32_t* vf, int maxval, const float* normals,
                     const float* dd, int64_t V, int64_t F, int n_iter, float* out, void* ws, size_t ws_bytes,
                     hipStream_t s) {
  GEOBI_REQUIRE(V > 0 && F > 0 && n_iter >= 0, "update_position2: empty mesh");
  Arena a(ws, ws_bytes);
  float* tmp = a.take<float>((size_t)V * 3);
  GEOBI_REQUIRE(a.ok() && ws, "update_position2: workspace too small");
  // ping-pong so that the LAST iteration lands in `out`
  const float* src = points;
  if (n_iter == 0) {
    GEOBI_HIP(hipMemcpyAsync(out, points, sizeof(float) * V * 3, hipMemcpyDeviceToDevice, s));
    return 0;
  }
  for (int it = 0; it < n_iter; ++it) {
    float* dst = ((n_iter - 1 - it) % 2 == 0) ? out : tmp;
    vertex_update_kernel<<<cdiv(V, 256), 256, 0, s>>>(src, nullptr, fv, normals, vf, maxval, dd, (int)V, dst);
    src = dst;
  }
  GEOBI_LAUNCH_OK();
  return 0;
}

int face_geom_fwd(const float* verts, const int32_t* fv, const float* xf, int ldxf, int64_t F, float* out,
                  hipStream_t s) {
  if (F <= 0) return 0;
  face_geom_fwd_kernel<<<cdiv(F, 256), 256, 0, s>>>(verts, fv, xf, ldxf, (int)F, out);
  GEOBI_LAUNCH_OK();
  return 0;
}

int face_geom_bwd(const float* verts, const int32_t* fv, const float* g, int64_t F, float* corner_grad,
                  hipStream_t s) {
  if (F <= 0) return 0;
  face_geom_bwd_kernel<<<cdiv(F, 256), 256, 0, s>>>(verts, fv, g, (int)F, corner_grad);
  GEOBI_LAUNCH_OK();
  return 0;
}

int head_fwd(const float* x, int Cin, int64_t N, const float* w1, const float* b1, int K, const float* w2,
             const float* b2, int nout, float slope, int mode, const float* dd, const float* resid, int ld_resid,
             float* h, float* raw, float* out, hipStream_t s) {
  GEOBI_REQUIRE(N > 0 && (K % 256) == 0 && (nout == 1 || nout == 3), "head_fwd: unsupported shape");
  GEOBI_REQUIRE(!(mode == 0 && nout == 1 && dd == nullptr), "head_fwd: force_depth needs depth_direction");
  GEOBI_REQUIRE(!(mode == 1 && nout != 3), "head_fwd: the face head has 3 outputs");
  if (h == nullptr) {   // fused path: the hidden activation never leaves the matrix-core accumulators
    GEOBI_REQUIRE(head_fused_supported(Cin, K, nout), "head_fwd: the fused head needs Cin=32, K=1024 (got %d, %d)", Cin, K);
    return head_fwd_fused(x, N, w1, b1, w2, b2, nout, slope, mode, dd, resid, ld_resid, raw, out, s);
  }
  GemmEpilogue ep;
  ep.bias = b1;
  ep.slope = slope;
  GEOBI_TRY(gemm_nn(x, Cin, w1, Cin, 1, h, K, (int)N, K, Cin, ep, s));
  if (nout == 3)
    head_out_kernel<3><<<cdiv(N, 4), 256, 0, s>>>(h, K, w2, b2, (int)N, mode, dd, resid, ld_resid, raw, out);
  else
    head_out_kernel<1><<<cdiv(N, 4), 256, 0, s>>>(h, K, w2, b2, (int)N, mode, dd, resid, ld_resid, raw, out);
  GEOBI_LAUNCH_OK();
  return 0;
}

size_t head_bwd_ws_bytes(int64_t N, int Cin, int K) {
  size_t t1 = gemm_tn_ws_bytes(K, Cin + 1, N), t2 = gemm_tn_ws_bytes(3, K + 1, N);
  size_t c = colsum_ws_bytes(N, K);
  size_t unfused = align_up((size_t)N * 3 * sizeof(float)) + align_up((size_t)N * K * sizeof(float)) +
                   align_up(t1 > t2 ? t1 : t2) + align_up(c) + 1024;
  size_t fused = align_up((size_t)N * 3 * sizeof(float)) + head_bwd_fused_ws_bytes(N) + 1024;
  return unfused > fused ? unfused : fused;
}

int head_bwd(const float* x, int Cin, int64_t N, const float* w1, const float* b1, int K, const float* w2, int nout,
             float slope, int mode, const float* dd, const float* h, const float* raw, const float* gout, float* dx,
             float* dw1, float* db1, float* dw2, float* db2, int accumulate, void* ws, size_t ws_bytes, hipStream_t s) {
  Arena a(ws, ws_bytes);
  float* graw = a.take<float>((size_t)N * 3);
  if (h == nullptr) {   // fused path (forward did not save the hidden activation): recompute it in-kernel
    GEOBI_REQUIRE(head_fused_supported(Cin, K, nout) && dx != nullptr && b1 != nullptr,
                  "head_bwd: fused path needs Cin=32, K=1024, b1 and dx");
    size_t fb = head_bwd_fused_ws_bytes(N);
    void* fws = a.take<char>(fb);
    GEOBI_REQUIRE(a.ok() && ws, "head_bwd: workspace too small (%zu < %zu)", ws_bytes, a.off);
    head_finish_bwd_kernel<<<cdiv(N, 256), 256, 0, s>>>(gout, raw, nout, mode, dd, (int)N, graw);
    GEOBI_LAUNCH_OK();
    return head_bwd_fused(x, N, w1, b1, w2, nout, slope, graw, dx, dw1, db1, dw2, db2, accumulate, fws, fb, s);
  }
  float* dh = a.take<float>((size_t)N * K);
  size_t t1 = gemm_tn_ws_bytes(K, Cin + 1, N), t2 = gemm_tn_ws_bytes(3, K + 1, N);
  size_t tnb = t1 > t2 ? t1 : t2;
  void* tn_ws = a.take<char>(tnb);
  GEOBI_REQUIRE(a.ok() && ws, "head_bwd: workspace too small (%zu < %zu)", ws_bytes, a.off);
  head_finish_bwd_kernel<<<cdiv(N, 256), 256, 0, s>>>(gout, raw, nout, mode, dd, (int)N, graw);
  GEOBI_LAUNCH_OK();
  // dW2 = graw^T h and db2 = graw^T 1 in one pass (implicit ones column appended to h)
  TnOutput o2;
  o2.C = dw2; o2.ldc = K; o2.C2 = db2; o2.extra_col = 1; o2.accumulate = accumulate;
  GEOBI_TRY(gemm_tn(graw, nout, h, K, N, nout, K + 1, -1, K, o2, tn_ws, tnb, s));
  head_dh_kernel<<<cdiv(N * K / 4, 256), 256, 0, s>>>(graw, nout, w2, h, K, slope, N * K, dh);
  GEOBI_LAUNCH_OK();
  // dW1 = dh^T x and db1 = dh^T 1
  TnOutput o1;
  o1.C = dw1; o1.ldc = Cin; o1.C2 = db1; o1.extra_col = 1; o1.accumulate = accumulate;
  GEOBI_TRY(gemm_tn(dh, K, x, Cin, N, K, Cin + 1, -1, Cin, o1, tn_ws, tnb, s));
  if (dx) {
    GemmEpilogue ep;
    GEOBI_TRY(gemm_nn(dh, K, w1, Cin, 0, dx, Cin, (int)N, Cin, K, ep, s));
  }
  return 0;
}

// ---------------------------------------------------------------------------- optimiser step
// torch.optim.Adam's update (code/train_dual.py:162: the reference's optimiser) over ONE flat parameter vector --
//   g += wd p;  m = m + (1 - b1)(g - m);  v = b2 v + (1 - b2) g g;  p -= (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps)
// -- four floats per thread, the whole chip busy.  torch's own fused form hands one 65 536-element chunk to a workgroup:
// 15 workgroups for the network's 0.94 M parameters, 45 us per step where this launch takes ~6.
namespace {
__global__ __launch_bounds__(256) void adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, int64_t n, float lr_bc1,
                                                        float b1, float b2, float eps, float wd, float rsqrt_bc2) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  auto one = [&](float& pp, float gg, float& mm, float& vv) {
    gg = wd != 0.f ? fmaf(wd, pp, gg) : gg;
    mm = fmaf(1.0f - b1, gg - mm, mm);                 // lerp, as torch writes it
    vv = fmaf(b2, vv, (1.0f - b2) * gg * gg);
    const float denom = sqrtf(vv) * rsqrt_bc2 + eps;
    pp -= lr_bc1 * (mm / denom);
  };
  if (i + 4 <= n) {
    float4 P = *reinterpret_cast<float4*>(p + i), M = *reinterpret_cast<float4*>(m + i), V = *reinterpret_cast<float4*>(v + i);
    const float4 G = *reinterpret_cast<const float4*>(g + i);
    one(P.x, G.x, M.x, V.x); one(P.y, G.y, M.y, V.y); one(P.z, G.z, M.z, V.z); one(P.w, G.w, M.w, V.w);
    *reinterpret_cast<float4*>(p + i) = P; *reinterpret_cast<float4*>(m + i) = M; *reinterpret_cast<float4*>(v + i) = V;
  } else {
    for (int64_t k = i; k < n; ++k) one(p[k], g[k], m[k], v[k]);
  }
}
}  // namespace

int adam_flat(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps, float wd,
              float bias_corr1, float bias_corr2, hipStream_t s) {
  GEOBI_REQUIRE(n > 0 && bias_corr1 > 0.f && bias_corr2 > 0.f, "adam_flat: n = %lld, bias corrections %g, %g", (long long)n,
                bias_corr1, bias_corr2);
  GEOBI_REQUIRE(((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16 == 0, "adam_flat: 16-byte aligned vectors");
  adam_flat_kernel<<<cdiv(cdiv(n, 4), 256), 256, 0, s>>>(p, g, m, v, n, lr / bias_corr1, b1, b2, eps, wd,
                                                         1.0f / sqrtf(bias_corr2));
  GEOBI_LAUNCH_OK();
  return 0;
}

}  // namespace geobi
