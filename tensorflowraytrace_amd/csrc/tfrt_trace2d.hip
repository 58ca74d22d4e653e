// tfrt 2-D hot path for MI355X (gfx950): rays x (segments + arcs), nearest hit, stable
// classification/compaction, angle-form Snell, pass loop and reverse sweep.
//
// Reference: OpticalSystem2D.intersect / _segment_intersection / _arc_intersection /
// _seg_or_arc / _get_arc_norm (tfrt/engine.py:523-866), process_projection_2D
// (engine.py:1544-1986), geometry.snells_law_2D (geometry.py:565-653).
//
// 2-D scenes have few primitives (M ~ 1..1e3) and many rays, so one pass is per-ray work
// streaming over an LDS-resident primitive table: lane = ray, every primitive is decided
// exactly in float64 (trace_math2d.h), running minima in registers.  In a mixed system each
// pass emits, per class, segment-hit rays first and arc-hit rays second (engine.py:1955-1981),
// hence 7 compaction bins: {active, finished, stopped} x {segment, arc} + dead.
#include "tfrt_common.h"
#include "trace_math2d.h"

namespace tfrt {

constexpr int TILE2 = 256;
constexpr int NBIN = 8;
constexpr int BIN_DEAD = 6;

template <typename T>
__device__ __forceinline__ void load_ray2(const T* rays, int64_t stride, int64_t i, double s[2],
                                          double e[2]) {
  s[0] = ldd(rays, i);
  s[1] = ldd(rays, stride + i);
  e[0] = ldd(rays, 2 * stride + i);
  e[1] = ldd(rays, 3 * stride + i);
}

template <typename T>
__device__ __forceinline__ void store_ray2(T* rays, int64_t stride, int64_t i, const double s[2],
                                           const double e[2]) {
  rays[i] = static_cast<T>(s[0]);
  rays[stride + i] = static_cast<T>(s[1]);
  rays[2 * stride + i] = static_cast<T>(e[0]);
  rays[3 * stride + i] = static_cast<T>(e[1]);
}

__device__ __forceinline__ int cat_cls(int cat) {
  return cat == CAT_OPTICAL ? CLS_ACTIVE : (cat == CAT_TARGET ? CLS_FINISHED : CLS_STOPPED);
}

// Nearest segment and nearest arc for ray (s, e); then engine.py:652-657 seg-vs-arc.
// Returns merged primitive index (segments first) or -1.
// STATE_ULP: unit roundoff of the ray-state storage type (0 for float64 state)
template <int STATE_BITS>
__device__ __forceinline__ int nearest2d(const double s[2], const double e[2],
                                         const double* __restrict__ seg, int Ms,
                                         const double* __restrict__ arc, int Ma, double ei,
                                         double es, double er, int last_prim, double* lds,
                                         bool active, double* out_u, double* out_aux) {
  double su = INFINITY, au = INFINITY, aang = 0.0;
  int sj = -1, aj = -1;
  for (int t0 = 0; t0 < Ms; t0 += TILE2) {
    const int nt = min(TILE2, Ms - t0);
    __syncthreads();
    for (int k = threadIdx.x; k < nt * 4; k += BLOCK) lds[k] = seg[(int64_t)t0 * 4 + k];
    __syncthreads();
    if (active) {
      for (int j = 0; j < nt; ++j) {
        if (t0 + j == last_prim) continue;  // the segment the ray starts on
        const Hit2 h = exact_segment(s, e, lds + 4 * j, ei, es, er);
        if (h.valid && h.ray_u < su) {
          su = h.ray_u;
          sj = t0 + j;
        }
      }
    }
  }
  for (int t0 = 0; t0 < Ma; t0 += TILE2) {
    const int nt = min(TILE2, Ma - t0);
    __syncthreads();
    for (int k = threadIdx.x; k < nt * 5; k += BLOCK) lds[k] = arc[(int64_t)t0 * 5 + k];
    __syncthreads();
    if (active) {
      for (int j = 0; j < nt; ++j) {
        double er_j = er;
        if (STATE_BITS < 53 && Ms + t0 + j == last_prim) {
          // a rounded (float32 / float16) start sits up to ~1 ulp off the arc it left: do not
          // let the near root (u ~ 0) count as a new hit
          const double dl = sqrt((e[0] - s[0]) * (e[0] - s[0]) + (e[1] - s[1]) * (e[1] - s[1]));
          const double mag = fabs(s[0]) + fabs(s[1]) + fabs(lds[5 * j + 4]);
          er_j = fmax(er, 64.0 * ldexp(1.0, -STATE_BITS) * mag / fmax(dl, 1e-300));
        }
        const Hit2 h = exact_arc(s, e, lds + 5 * j, ei, er_j);
        if (h.valid && h.ray_u < au) {
          au = h.ray_u;
          aj = t0 + j;
          aang = h.prim_u;
        }
      }
    }
  }
  // engine.py:652-657
  if (sj >= 0 && aj >= 0) {
    if (su < au) aj = -1; else sj = -1;
  }
  if (sj >= 0) {
    *out_u = su;
    *out_aux = 0.0;
    return sj;
  }
  if (aj >= 0) {
    *out_u = au;
    *out_aux = aang;
    return Ms + aj;
  }
  *out_u = INFINITY;
  *out_aux = 0.0;
  return -1;
}

// ---------------------------------------------------------------------------------------
// Filtered nearest hit (the trace kernels; the seams keep the plain loop above).
//
// As in 3-D, throughput and decisions are separated: every primitive gets a bounding circle
// (segment: midpoint, half length; arc: the whole circle, or the circle on its chord when it
// spans less than pi), every ray the unit normal n of its line and the offset s.n; the line can
// touch the primitive only if (c.n - s.n)^2 <= r_eff^2 -- 6 float32 ops against ~60 float64 ops
// (two atan2 for an arc) of the exact test.  Survivors (a handful out of hundreds) go to a
// per-lane LDS queue and are decided exactly, in ascending primitive order, by a wave-uniform
// flush.  Conservative: r_eff = r (1 + 1e-5) + 64 * 2^-24 (|c| + r), segments are extended by
// size_eps at both ends (engine.py:722-724), and the reference's tangent snap `|rad| < eps -> 0`
// (geometry.py:480-483), which lets a line MISS a circle of radius R by up to
// d = eps R^3 / (8 |ray|^2), is covered by (r + d)^2 <= r^2 + 2 w R^4 + w^2 R^6, w = eps / (8 |ray|^2).
constexpr int KQ2 = 16;  // queue slots per lane

__device__ __forceinline__ float round_up2(double x) {
  float f = static_cast<float>(x);
  f = nextafterf(f, INFINITY);
  return nextafterf(f, INFINITY);
}

// filter entry (cx, cy, r_eff^2, R^2) of a segment
__device__ __forceinline__ float4 filter_segment(const double* g, double es) {
  const double cx = 0.5 * (g[0] + g[2]), cy = 0.5 * (g[1] + g[3]);
  const double len = sqrt((g[2] - g[0]) * (g[2] - g[0]) + (g[3] - g[1]) * (g[3] - g[1]));
  double r = 0.5 * len + (es > 0.0 ? es * len : 0.0);
  r = r * (1.0 + 1e-5) + 64.0 * 5.9604644775390625e-08 * (sqrt(cx * cx + cy * cy) + r);
  if (!(r == r) || r > 1e18) return make_float4(0.f, 0.f, INFINITY, 0.f);  // always a candidate
  return make_float4((float)cx, (float)cy, round_up2(r * r), 0.f);
}

__device__ __forceinline__ float4 filter_arc(const double* g) {
  const double R = fabs(g[4]);
  double span = g[3] - g[2];
  if (span < 0.0) span += 2 * PI_D;  // angle_in_interval: geometry.py:790-802
  double cx = g[0], cy = g[1], r = R;
  const double h = 0.5 * span;
  if (h < 0.5 * PI_D) {  // less than a half circle: the circle on the chord contains the arc
    const double tm = g[2] + h;
    cx += R * cos(h) * cos(tm);
    cy += R * cos(h) * sin(tm);
    r = R * sin(h) + 1e-9 * R;  // + slack for the rounding of cos / sin
  }
  r = r * (1.0 + 1e-5) + 64.0 * 5.9604644775390625e-08 * (sqrt(cx * cx + cy * cy) + r);
  if (!(r == r) || !(R == R) || r > 1e18 || R > 1e9)
    return make_float4(0.f, 0.f, INFINITY, 0.f);  // always a candidate
  return make_float4((float)cx, (float)cy, round_up2(r * r), round_up2(R * R));
}

// Two levels (round 4): the primitives of a tile, in the caller's order, form clusters of G2 = 8
// consecutive ones -- the segments of a polyline and the arcs of a lens array are neighbours in
// the tables the reference builds (boundaries.py: one row per segment along the curve) -- under a
// bounding circle of their bounding circles.  A lane first collects the clusters its line touches
// (Ms / 8 + Ma / 8 tests instead of Ms + Ma), then tests the members of ITS clusters, cluster by
// cluster in ascending order: primitives still reach the exact test in ascending index, so ties
// fall as before.  A line that passes a member's circle passes the cluster's:
// |t_C - t_m| <= |C - c_m|, hence |t_C| <= |C - c_m| + r_m + sqrt(s_m) <= R_C + q, with
// s_m = R_m^4 (2 w + w^2 R_m^2) the arc's tangent-snap term and
// q = Rq sqrt(w1) + Rq^1.5 sqrt(w2) >= sqrt(s_max)   (Rq = the largest R^2 of the cluster).
// Measured on cfg5b (4M rays x 320 primitives, random rays): 5,340 -> VALU instructions per
// wavefront, k_intersect2d 627 us per pass -> see DESIGN.md 5.
constexpr int G2 = 8;                     // primitives per cluster
constexpr int NC2 = TILE2 / G2;           // clusters per tile

// bounding circle (cx, cy, R, Rq) and Rq^1.5 of the filter entries f[0..G2) (z < 0: padding)
__device__ __forceinline__ float4 cluster_filter(const float4* f, float* rq15) {
  float sx = 0.f, sy = 0.f, rq = 0.f;
  int m = 0;
  bool always = false;
#pragma unroll
  for (int g = 0; g < G2; ++g) {
    const float4 v = f[g];
    if (v.z < 0.f) continue;
    always = always || !(v.z < INFINITY);
    sx += v.x;
    sy += v.y;
    rq = fmaxf(rq, v.w);
    ++m;
  }
  *rq15 = 0.f;
  if (m == 0) return make_float4(0.f, 0.f, -1.f, 0.f);          // never touched
  if (always || !(rq < 1e30f)) return make_float4(0.f, 0.f, INFINITY, 0.f);  // always a candidate
  const float cx = sx / (float)m, cy = sy / (float)m;
  float R = 0.f;
#pragma unroll
  for (int g = 0; g < G2; ++g) {
    const float4 v = f[g];
    if (v.z < 0.f) continue;
    const float dx = v.x - cx, dy = v.y - cy;
    const float d = sqrtf(dx * dx + dy * dy) * (1.f + 1e-6f);
    const float r = sqrtf(v.z) * (1.f + 1e-6f);
    R = fmaxf(R, d + r);
  }
  // (the same allowance for the float32 evaluation of the line's offset as the members have)
  R = R * (1.f + 1e-5f) + 64.f * 5.9604644775390625e-08f * (sqrtf(cx * cx + cy * cy) + R);
  if (!(R < INFINITY)) return make_float4(0.f, 0.f, INFINITY, 0.f);
  *rq15 = rq * sqrtf(rq) * (1.f + 1e-6f);
  return make_float4(cx, cy, R, rq);
}

template <int STATE_BITS>
__device__ __forceinline__ int nearest2d_filtered(const double s[2], const double e[2],
                                                  const double* __restrict__ seg, int Ms,
                                                  const double* __restrict__ arc, int Ma,
                                                  double ei, double es, double er, int last_prim,
                                                  double* lds, float4* filt, int32_t* queue,
                                                  float4* cfilt, float* crq15, uint8_t* cqueue,
                                                  bool active, double* out_u, double* out_aux) {
  // float32 state of this lane's ray
  // sn = NaN: a lane without a (valid) ray never passes the test, whatever the radius
  float nx = 0.f, ny = 0.f, sn = __builtin_nanf(""), w1 = 0.f, w2 = 0.f, k1 = 0.f, kw = 0.f;
  {
    const double dx = e[0] - s[0], dy = e[1] - s[1];
    const double l2 = dx * dx + dy * dy;
    if (active && l2 > 0.0 && l2 < INFINITY) {
      const double inv = 1.0 / sqrt(l2);
      const double ux = -dy * inv, uy = dx * inv;
      nx = (float)ux;
      ny = (float)uy;
      sn = (float)(s[0] * ux + s[1] * uy);
      const double w = (ei > 0.0 ? ei : 0.0) / (8.0 * l2);
      w1 = round_up2(2.0 * w * (1.0 + 1e-5));
      w2 = round_up2(w * w * (1.0 + 1e-5));
      k1 = round_up2(sqrt((double)w1) * (1.0 + 1e-6));
      kw = round_up2(sqrt((double)w2) * (1.0 + 1e-6));
    }
  }
  auto touches = [&](const float4 f) {
    const float t = fmaf(f.x, nx, fmaf(f.y, ny, -sn));
    const float lim = fmaf(f.w * f.w, fmaf(w2, f.w, w1), f.z);
    return t * t <= lim;
  };
  auto touches_cluster = [&](const float4 c, const float rq15) {
    const float t = fmaf(c.x, nx, fmaf(c.y, ny, -sn));
    return fabsf(t) <= c.z + fmaf(c.w, k1, rq15 * kw);
  };
  const int tid = threadIdx.x;
  int cnt = 0;
  double su = INFINITY, au = INFINITY, aang = 0.0;
  int sj = -1, aj = -1;

  // the clusters of the tile in `filt` (nt primitives, padded to whole clusters)
  auto build_clusters = [&](int nt) {
    const int nc = (nt + G2 - 1) / G2;
    for (int k = tid; k < nc; k += BLOCK) cfilt[k] = cluster_filter(filt + G2 * k, crq15 + k);
    return nc;
  };
  // this lane's clusters, ascending (always store, count only hits: no branch per test)
  auto collect = [&](int nc) {
    int cc = 0;
    for (int c = 0; c < nc; ++c) {
      cqueue[cc * BLOCK + tid] = (uint8_t)c;
      cc += touches_cluster(cfilt[c], crq15[c]) ? 1 : 0;
    }
    return cc;
  };

  for (int t0 = 0; t0 < Ms; t0 += TILE2) {
    const int nt = min(TILE2, Ms - t0);
    const int nt8 = (nt + G2 - 1) & ~(G2 - 1);
    __syncthreads();
    for (int k = tid; k < nt * 4; k += BLOCK) lds[k] = seg[(int64_t)t0 * 4 + k];
    __syncthreads();
    for (int k = tid; k < nt8; k += BLOCK)
      filt[k] = k < nt ? filter_segment(lds + 4 * k, es) : make_float4(0.f, 0.f, -1.f, 0.f);
    __syncthreads();
    const int nc = build_clusters(nt);
    __syncthreads();
    auto flush = [&]() {
      for (int k = 0; k < cnt; ++k) {
        const int j = queue[k * BLOCK + tid];
        if (t0 + j == last_prim) continue;  // the segment the ray starts on
        const Hit2 h = exact_segment(s, e, lds + 4 * j, ei, es, er);
        if (h.valid && h.ray_u < su) {
          su = h.ray_u;
          sj = t0 + j;
        }
      }
      cnt = 0;
    };
    const int cc = collect(nc);
    for (int k = 0; __any(k < cc); ++k) {
      if (k < cc) {
        const int j0 = G2 * (int)cqueue[k * BLOCK + tid];
#pragma unroll
        for (int g = 0; g < G2; ++g) {
          queue[cnt * BLOCK + tid] = j0 + g;
          cnt += touches(filt[j0 + g]) ? 1 : 0;
        }
      }
      if (__any(cnt > KQ2 - G2)) flush();
    }
    flush();  // the queue refers to this tile's LDS copy
  }
  for (int t0 = 0; t0 < Ma; t0 += TILE2) {
    const int nt = min(TILE2, Ma - t0);
    const int nt8 = (nt + G2 - 1) & ~(G2 - 1);
    __syncthreads();
    for (int k = tid; k < nt * 5; k += BLOCK) lds[k] = arc[(int64_t)t0 * 5 + k];
    __syncthreads();
    for (int k = tid; k < nt8; k += BLOCK)
      filt[k] = k < nt ? filter_arc(lds + 5 * k) : make_float4(0.f, 0.f, -1.f, 0.f);
    __syncthreads();
    const int nc = build_clusters(nt);
    __syncthreads();
    auto flush = [&]() {
      for (int k = 0; k < cnt; ++k) {
        const int j = queue[k * BLOCK + tid];
        double er_j = er;
        if (STATE_BITS < 53 && Ms + t0 + j == last_prim) {
          // a rounded (float32 / float16) start sits up to ~1 ulp off the arc it left: do not
          // let the near root (u ~ 0) count as a new hit
          const double dl = sqrt((e[0] - s[0]) * (e[0] - s[0]) + (e[1] - s[1]) * (e[1] - s[1]));
          const double mag = fabs(s[0]) + fabs(s[1]) + fabs(lds[5 * j + 4]);
          er_j = fmax(er, 64.0 * ldexp(1.0, -STATE_BITS) * mag / fmax(dl, 1e-300));
        }
        const Hit2 h = exact_arc_hit(s, e, lds + 5 * j, ei, er_j);
        if (h.valid && h.ray_u < au) {
          au = h.ray_u;
          aj = t0 + j;
          aang = h.prim_u;
        }
      }
      cnt = 0;
    };
    const int cc = collect(nc);
    for (int k = 0; __any(k < cc); ++k) {
      if (k < cc) {
        const int j0 = G2 * (int)cqueue[k * BLOCK + tid];
#pragma unroll
        for (int g = 0; g < G2; ++g) {
          queue[cnt * BLOCK + tid] = j0 + g;
          cnt += touches(filt[j0 + g]) ? 1 : 0;
        }
      }
      if (__any(cnt > KQ2 - G2)) flush();
    }
    flush();
  }
  // engine.py:652-657
  if (sj >= 0 && aj >= 0) {
    if (su < au) aj = -1; else sj = -1;
  }
  if (sj >= 0) {
    *out_u = su;
    *out_aux = 0.0;
    return sj;
  }
  if (aj >= 0) {
    *out_u = au;
    *out_aux = aang;
    return Ms + aj;
  }
  *out_u = INFINITY;
  *out_aux = 0.0;
  return -1;
}

// 39.7 KB of LDS per workgroup leave room for 4 wavefronts per SIMD; the compiler's own choice is
// 152 VGPRs (3 wavefronts).  Pinned to 4 (128 VGPRs, 18 spilled to scratch in the exact tests):
// cfg5b forward + backward 2.93 -> 2.82 ms, results unchanged.
#ifndef TFRT_I2D_WAVES
#define TFRT_I2D_WAVES 4
#endif
#define TFRT_I2D_ATTR __attribute__((amdgpu_waves_per_eu(TFRT_I2D_WAVES, TFRT_I2D_WAVES)))
template <typename T>
__global__ __launch_bounds__(BLOCK) TFRT_I2D_ATTR void k_intersect2d(
    const T* __restrict__ rays, int64_t stride, const int32_t* __restrict__ n_ptr,
    const int32_t* __restrict__ last_prim, tfrt_scene2d sc, int32_t* __restrict__ rec_prim,
    double* __restrict__ rec_u, double* __restrict__ rec_aux, uint8_t* __restrict__ rec_bin,
    int32_t* __restrict__ blockcnt) {
  const int n = *n_ptr;
  const int base = blockIdx.x * BLOCK;
  if (base >= n) return;
  __shared__ double lds[TILE2 * 5];
  __shared__ float4 filt[TILE2];
  __shared__ int32_t queue[KQ2 * BLOCK];
  __shared__ float4 cfilt[NC2];
  __shared__ float crq15[NC2];
  __shared__ uint8_t cqueue[NC2 * BLOCK];
  __shared__ int wc[WAVES][NBIN];
  const int i = base + threadIdx.x;
  const bool active = i < n;
  double s[2] = {0, 0}, e[2] = {0, 0};
  if (active) load_ray2(rays, stride, i, s, e);
  const int lp = (active && last_prim) ? last_prim[i] : -1;
  const int Ms = (int)sc.n_segments, Ma = (int)sc.n_arcs;
  double u, aux;
  const int prim = nearest2d_filtered<(sizeof(T) == 8 ? 53 : (sizeof(T) == 4 ? 24 : 11))>(
      s, e, sc.seg, Ms, sc.arc, Ma, sc.intersect_epsilion, sc.size_epsilion,
      sc.ray_start_epsilion, lp, lds, filt, queue, cfilt, crq15, cqueue, active, &u, &aux);
  int bin = -1;
  if (active) {
    if (prim < 0) {
      bin = BIN_DEAD;
    } else if (prim < Ms) {
      bin = cat_cls(sc.seg_cat[prim]) * 2;
    } else {
      bin = cat_cls(sc.arc_cat[prim - Ms]) * 2 + 1;
    }
    rec_prim[i] = prim;
    rec_u[i] = u;
    rec_aux[i] = aux;
    rec_bin[i] = (uint8_t)bin;
  }
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < NBIN; ++c) {
    const unsigned long long m = __ballot(bin == c);
    if (lane_id() == 0) wc[wave][c] = __popcll(m);
  }
  __syncthreads();
  if (threadIdx.x < NBIN) {
    int t = 0;
    for (int w = 0; w < WAVES; ++w) t += wc[w][threadIdx.x];
    blockcnt[blockIdx.x * NBIN + threadIdx.x] = t;
  }
}

// One block of 1024 threads scans every ray block's histogram (up to SCAN_GRID_MIN_ROWS rows;
// the offsets it writes are global, k_react2d gets no rowbase).
__global__ __launch_bounds__(1024) void k_scan2d_one(const int32_t* __restrict__ n_ptr,
                                                 const int32_t* __restrict__ blockcnt,
                                                 int32_t* __restrict__ blockoff,
                                                 int32_t* __restrict__ pass_counts,
                                                 int32_t* __restrict__ bin_counts,
                                                 int32_t* __restrict__ totals,
                                                 int32_t* __restrict__ n_next,
                                                 unsigned long long* __restrict__ n_tests,
                                                 int M) {
  const int n = *n_ptr;
  const int nblk = (n + BLOCK - 1) / BLOCK;
  const int per = (nblk + 1023) / 1024;
  const int b0 = min(nblk, (int)threadIdx.x * per), b1 = min(nblk, b0 + per);
  int loc[NBIN];
#pragma unroll
  for (int c = 0; c < NBIN; ++c) loc[c] = 0;
  for (int b = b0; b < b1; ++b)
    for (int c = 0; c < NBIN; ++c) loc[c] += blockcnt[b * NBIN + c];
  __shared__ int wsum[16][NBIN];
  __shared__ int wbase[16][NBIN];
  __shared__ int total[NBIN];
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  int pre[NBIN];
#pragma unroll
  for (int c = 0; c < NBIN; ++c) {
    int v = loc[c];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(v, d, 64);
      if (lane >= d) v += o;
    }
    pre[c] = v - loc[c];
    if (lane == 63) wsum[wave][c] = v;
  }
  __syncthreads();
  if (threadIdx.x < NBIN) {
    int run = 0;
    for (int w = 0; w < 16; ++w) {
      wbase[w][threadIdx.x] = run;
      run += wsum[w][threadIdx.x];
    }
    total[threadIdx.x] = run;
  }
  __syncthreads();
  int run[NBIN];
#pragma unroll
  for (int c = 0; c < NBIN; ++c) run[c] = wbase[wave][c] + pre[c];
  for (int b = b0; b < b1; ++b)
    for (int c = 0; c < NBIN; ++c) {
      blockoff[b * NBIN + c] = run[c];
      run[c] += blockcnt[b * NBIN + c];
    }
  if (threadIdx.x < NBIN) bin_counts[threadIdx.x] = total[threadIdx.x];
  if (threadIdx.x < 4) {
    const int c = threadIdx.x;
    const int t = (c == CLS_DEAD) ? total[BIN_DEAD] : total[2 * c] + total[2 * c + 1];
    pass_counts[c] = t;
    pass_counts[4 + c] = totals[c];
    totals[c] += t;
    if (c == CLS_ACTIVE) *n_next = t;
  }
  if (threadIdx.x == 0) *n_tests += (unsigned long long)n * (unsigned long long)M;
}

// From SCAN_GRID_MIN_ROWS ray blocks on (its ticket and fences cost ~13 us whatever the size; the
// single block takes 54 us for the 15.6k rows of a 4M-ray pass, this 20 us):
constexpr int SCAN_GRID_MIN_ROWS = 8192;
__global__ __launch_bounds__(1024) void k_scan2d(const int32_t* __restrict__ n_ptr,
                                                 const int32_t* __restrict__ blockcnt,
                                                 int32_t* __restrict__ blockoff,
                                                 int32_t* __restrict__ rowtot,
                                                 int32_t* __restrict__ rowbase,
                                                 unsigned int* __restrict__ ticket,
                                                 int32_t* __restrict__ pass_counts,
                                                 int32_t* __restrict__ bin_counts,
                                                 int32_t* __restrict__ totals,
                                                 int32_t* __restrict__ n_next,
                                                 unsigned long long* __restrict__ n_tests,
                                                 int M) {
  // grid of 1024-thread workgroups, one ray block per thread (scan_rows_grid); the workgroup
  // that finishes last closes the pass
  const int n = *n_ptr;
  const int nblk = (n + BLOCK - 1) / BLOCK;
  __shared__ int total[NBIN];
  if (!scan_rows_grid<NBIN>(blockcnt, blockoff, nblk, rowtot, rowbase, ticket, total)) return;
  if (threadIdx.x < NBIN) bin_counts[threadIdx.x] = total[threadIdx.x];
  if (threadIdx.x < 4) {
    const int c = threadIdx.x;
    const int t = (c == CLS_DEAD) ? total[BIN_DEAD] : total[2 * c] + total[2 * c + 1];
    pass_counts[c] = t;
    pass_counts[4 + c] = totals[c];
    totals[c] += t;
    if (c == CLS_ACTIVE) *n_next = t;
  }
  if (threadIdx.x == 0) *n_tests += (unsigned long long)n * (unsigned long long)M;
}

__device__ __forceinline__ void prim_indices(const tfrt_scene2d& sc, int prim, int rid,
                                             double* n_in, double* n_out) {
  const int Ms = (int)sc.n_segments;
  const bool is_arc = prim >= Ms;
  const int k = is_arc ? prim - Ms : prim;
  const int32_t* mi = is_arc ? sc.arc_mat_in : sc.seg_mat_in;
  const int32_t* mo = is_arc ? sc.arc_mat_out : sc.seg_mat_out;
  if (sc.n_table != nullptr && mi != nullptr && mo != nullptr) {
    *n_in = sc.n_table[(int64_t)mi[k] * sc.n_table_stride + rid];
    *n_out = sc.n_table[(int64_t)mo[k] * sc.n_table_stride + rid];
  } else {
    *n_in = (is_arc ? sc.arc_n_in : sc.seg_n_in)[k];
    *n_out = (is_arc ? sc.arc_n_out : sc.seg_n_out)[k];
  }
}

template <typename T>
__device__ __forceinline__ bool emit2(const tfrt_ray_out& o, int64_t slot, const double s[2],
                                      const double e[2], int rid, int face) {
  if (o.rays == nullptr) return true;
  if (slot >= o.capacity) return false;
  store_ray2(static_cast<T*>(o.rays), o.capacity, slot, s, e);
  if (o.ray_id) o.ray_id[slot] = rid;
  if (o.face) o.face[slot] = face;
  return true;
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void k_react2d(
    const T* __restrict__ rays_in, int64_t stride_in, const int32_t* __restrict__ n_ptr,
    const int32_t* __restrict__ ray_id_in, const int32_t* __restrict__ rec_prim,
    const double* __restrict__ rec_u, const double* __restrict__ rec_aux,
    const uint8_t* __restrict__ rec_bin, const int32_t* __restrict__ blockoff,
    const int32_t* __restrict__ rowbase,
    const int32_t* __restrict__ pass_counts, const int32_t* __restrict__ bin_counts,
    tfrt_scene2d sc, double L, double dead_len, uint32_t flags, T* __restrict__ rays_out,
    int64_t stride_out, int32_t* __restrict__ ray_id_out, int32_t* __restrict__ last_prim_out,
    int32_t* __restrict__ rec_slot, tfrt_ray_out fin, tfrt_ray_out act, tfrt_ray_out stp,
    tfrt_ray_out dead, int32_t* __restrict__ err) {
  const int n = *n_ptr;
  const int base = blockIdx.x * BLOCK;
  if (base >= n) return;
  const int i = base + threadIdx.x;
  const int bin = (i < n) ? (int)rec_bin[i] : -1;
  __shared__ int wc[WAVES][NBIN];
  const int wave = threadIdx.x >> 6;
  int rank = 0;
#pragma unroll
  for (int c = 0; c < NBIN; ++c) {
    const unsigned long long m = __ballot(bin == c);
    if (bin == c) rank = rank_below(m);
    if (lane_id() == 0) wc[wave][c] = __popcll(m);
  }
  __syncthreads();
  if (i >= n) return;
  for (int w = 0; w < wave; ++w) rank += wc[w][bin];
  const int cls = (bin == BIN_DEAD) ? CLS_DEAD : (bin >> 1);
  const int kind = bin & 1;
  int slot = blockoff[blockIdx.x * NBIN + bin] +
             (rowbase != nullptr ? rowbase[(blockIdx.x >> 10) * NBIN + bin] : 0) + rank;
  if (cls != CLS_DEAD && kind == 1) slot += bin_counts[2 * cls];  // arcs after segments
  const int64_t gslot = (int64_t)pass_counts[4 + cls] + slot;

  double s[2], e[2];
  load_ray2(rays_in, stride_in, i, s, e);
  const int rid = ray_id_in ? ray_id_in[i] : i;
  const int prim = rec_prim[i];
  bool ok = true;
  if (cls == CLS_DEAD) {
    if (flags & TFRT_COMPILE_DEAD) {
      double e2[2] = {e[0], e[1]};
      if (dead_len != 0.0)
        for (int k = 0; k < 2; ++k) e2[k] = advance_between(s[k], dead_len, e[k]);
      ok = emit2<T>(dead, gslot, s, e2, rid, -1);
    }
    rec_slot[i] = (int32_t)gslot;
  } else {
    const double u = rec_u[i];
    const double h[2] = {s[0] + u * (e[0] - s[0]), s[1] + u * (e[1] - s[1])};
    if (cls == CLS_FINISHED) {
      if (flags & TFRT_COMPILE_FINISHED) ok = emit2<T>(fin, gslot, s, h, rid, prim);
      rec_slot[i] = (int32_t)gslot;
    } else if (cls == CLS_STOPPED) {
      if (flags & TFRT_COMPILE_STOPPED) ok = emit2<T>(stp, gslot, s, h, rid, prim);
      rec_slot[i] = (int32_t)gslot;
    } else {
      if (flags & TFRT_COMPILE_ACTIVE) ok = emit2<T>(act, gslot, s, h, rid, prim);
      const int Ms = (int)sc.n_segments;
      double norm;
      if (prim >= Ms) {
        norm = arc_norm(sc.arc[(int64_t)(prim - Ms) * 5 + 4], rec_aux[i]);
      } else {
        norm = segment_norm(sc.seg + (int64_t)prim * 4);
      }
      double n_in, n_out;
      prim_indices(sc, prim, rid, &n_in, &n_out);
      const double a = snell2d_angle(s[0], s[1], h[0], h[1], norm, n_in, n_out);
      const double e2[2] = {advance(h[0], L, cos(a)), advance(h[1], L, sin(a))};
      store_ray2(rays_out, stride_out, slot, h, e2);
      ray_id_out[slot] = rid;
      last_prim_out[slot] = prim;
      rec_slot[i] = slot;
    }
  }
  if (!ok) atomicOr(err, 1);
}

__device__ __forceinline__ void add4(const double* g, int64_t cap, int64_t slot, double a[2],
                                     double b[2]) {
  if (g == nullptr) return;
  for (int k = 0; k < 2; ++k) {
    a[k] += g[k * cap + slot];
    b[k] += g[(2 + k) * cap + slot];
  }
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void k_backward2d(
    const T* __restrict__ rays_in, int64_t stride_in, const int32_t* __restrict__ n_ptr,
    const int32_t* __restrict__ ray_id_in, const int32_t* __restrict__ rec_prim,
    const double* __restrict__ rec_u, const uint8_t* __restrict__ rec_bin,
    const int32_t* __restrict__ rec_slot, const int32_t* __restrict__ pass_counts,
    tfrt_scene2d sc, double L, double dead_len, const double* __restrict__ g_child,
    int64_t child_stride, const double* __restrict__ g_fin, int64_t cap_fin,
    const double* __restrict__ g_act, int64_t cap_act, const double* __restrict__ g_stp,
    int64_t cap_stp, const double* __restrict__ g_dead, int64_t cap_dead,
    double* __restrict__ g_out, int64_t out_stride, double* __restrict__ g_seg,
    double* __restrict__ g_arc) {
  const int n = *n_ptr;
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const int bin = rec_bin[i];
  const int cls = (bin == BIN_DEAD) ? CLS_DEAD : (bin >> 1);
  const int slot = rec_slot[i];
  double s[2], e[2];
  load_ray2(rays_in, stride_in, i, s, e);
  double gs[2] = {0, 0}, ge[2] = {0, 0};
  if (cls == CLS_DEAD) {
    if (g_dead != nullptr) {
      double a[2] = {0, 0}, b[2] = {0, 0};
      add4(g_dead, cap_dead, slot, a, b);
      const double dl = (dead_len != 0.0) ? dead_len : 1.0;
      for (int k = 0; k < 2; ++k) {
        gs[k] = a[k] + (1.0 - dl) * b[k];
        ge[k] = dl * b[k];
      }
    }
  } else {
    double g_s[2] = {0, 0}, g_h[2] = {0, 0}, g_ce[2] = {0, 0};
    bool has_child = false;
    if (cls == CLS_FINISHED) {
      add4(g_fin, cap_fin, slot, g_s, g_h);
    } else if (cls == CLS_STOPPED) {
      add4(g_stp, cap_stp, slot, g_s, g_h);
    } else {
      // active-history slot: base + (arcs after segments) handled at forward time: the
      // child slot IS the within-pass active slot
      add4(g_act, cap_act, (int64_t)pass_counts[4 + CLS_ACTIVE] + slot, g_s, g_h);
      if (g_child != nullptr) {
        has_child = true;
        add4(g_child, child_stride, slot, g_h, g_ce);
      }
    }
    bool nz = has_child;
    for (int k = 0; k < 2; ++k) nz = nz || g_s[k] != 0.0 || g_h[k] != 0.0;
    if (nz) {
      const int prim = rec_prim[i];
      const int Ms = (int)sc.n_segments;
      const bool is_arc = prim >= Ms;
      const int rid = ray_id_in ? ray_id_in[i] : i;
      double n_in = 1.0, n_out = 1.0, gp[5];
      if (has_child) prim_indices(sc, prim, rid, &n_in, &n_out);
      const double* pp = is_arc ? sc.arc + (int64_t)(prim - Ms) * 5 : sc.seg + (int64_t)prim * 4;
      adjoint2d(s, e, pp, is_arc, rec_u[i], has_child, n_in, n_out, L, g_s, g_h, g_ce, gs, ge, gp,
                sc.finite_tir_gradient != 0);
      if (is_arc) {
        if (g_arc != nullptr)
          for (int q = 0; q < 5; ++q)
            if (gp[q] != 0.0) unsafeAtomicAdd(g_arc + (int64_t)(prim - Ms) * 5 + q, gp[q]);
      } else if (g_seg != nullptr) {
        for (int q = 0; q < 4; ++q)
          if (gp[q] != 0.0) unsafeAtomicAdd(g_seg + (int64_t)prim * 4 + q, gp[q]);
      }
    }
  }
  for (int k = 0; k < 2; ++k) {
    g_out[k * out_stride + i] = gs[k];
    g_out[(2 + k) * out_stride + i] = ge[k];
  }
}

__global__ void k_init2(int32_t* nrays0, int n, int32_t* tail8, unsigned int* scan_ticket) {
  if (threadIdx.x == 0) *nrays0 = n;
  if (threadIdx.x < 8) tail8[threadIdx.x] = 0;
  if (threadIdx.x == 0) *scan_ticket = 0u;
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void k_copy_rays2(const T* __restrict__ in, int64_t sin,
                                                      const int32_t* __restrict__ id_in,
                                                      const int32_t* __restrict__ n_ptr,
                                                      T* __restrict__ out, int64_t sout,
                                                      int32_t* __restrict__ id_out) {
  const int n = *n_ptr;
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  for (int k = 0; k < 4; ++k) out[k * sout + i] = in[k * sin + i];
  if (id_out) id_out[i] = id_in ? id_in[i] : i;
}

// seam kernels: one primitive kind, nearest per ray
template <typename T, bool ARC>
__global__ __launch_bounds__(BLOCK) void k_seam2d(const T* __restrict__ rays, int64_t stride, int n,
                                                  const double* __restrict__ prim, int M, double ei,
                                                  double es, double er, double* x, double* y,
                                                  uint8_t* valid, double* ray_u, double* prim_u,
                                                  int32_t* gather) {
  __shared__ double lds[TILE2 * 5];
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  const bool active = i < n;
  double s[2] = {0, 0}, e[2] = {0, 0};
  if (active) load_ray2(rays, stride, i, s, e);
  constexpr int W = ARC ? 5 : 4;
  Hit2 best;
  best.valid = false;
  best.ray_u = INFINITY;
  best.prim_u = best.x = best.y = 0.0;
  int bj = 0;
  Hit2 first = best;
  for (int t0 = 0; t0 < M; t0 += TILE2) {
    const int nt = min(TILE2, M - t0);
    __syncthreads();
    for (int k = threadIdx.x; k < nt * W; k += BLOCK) lds[k] = prim[(int64_t)t0 * W + k];
    __syncthreads();
    if (active) {
      for (int j = 0; j < nt; ++j) {
        Hit2 h;
        if (ARC) h = exact_arc(s, e, lds + 5 * j, ei, er);
        else h = exact_segment(s, e, lds + 4 * j, ei, es, er);
        if (t0 + j == 0) first = h;
        if (h.valid && h.ray_u < best.ray_u) {
          best = h;
          bj = t0 + j;
        }
      }
    }
  }
  if (!active) return;
  const Hit2& o = best.valid ? best : first;  // argmin of an all-sentinel column is 0
  x[i] = o.x;
  y[i] = o.y;
  valid[i] = best.valid;
  ray_u[i] = best.valid ? best.ray_u : INFINITY;
  prim_u[i] = o.prim_u;
  gather[i] = best.valid ? bj : 0;
}

struct Layout2 {
  size_t nrays, blockcnt, blockoff, rowtot, rowbase, ticket, bincnt, rays, rayid, lastprim, rec_prim, rec_slot, rec_u,
      rec_aux, rec_bin, gbuf, total;
  int nblk;
};

static Layout2 make_layout2(int64_t N, int P, int dtype) {
  Layout2 L;
  const size_t esz = dtype == TFRT_F64 ? 8 : (dtype == TFRT_F16 ? 2 : 4);
  const size_t n = N > 0 ? N : 1;
  L.nblk = cdiv(n, BLOCK);
  size_t o = 0;
  auto take = [&](size_t bytes) {
    size_t at = o;
    o = align_up(o + bytes);
    return at;
  };
  L.nrays = take((P + 2) * sizeof(int32_t));
  L.blockcnt = take((size_t)L.nblk * NBIN * sizeof(int32_t));
  L.blockoff = take((size_t)L.nblk * NBIN * sizeof(int32_t));
  L.rowtot = take((size_t)cdiv(L.nblk, 1024) * NBIN * sizeof(int32_t));
  L.rowbase = take((size_t)cdiv(L.nblk, 1024) * NBIN * sizeof(int32_t));
  L.ticket = take(sizeof(unsigned int));
  L.bincnt = take((size_t)(P + 1) * NBIN * sizeof(int32_t));
  L.rays = take((size_t)P * 4 * n * esz);
  L.rayid = take((size_t)P * n * sizeof(int32_t));
  L.lastprim = take((size_t)P * n * sizeof(int32_t));
  L.rec_prim = take((size_t)P * n * sizeof(int32_t));
  L.rec_slot = take((size_t)P * n * sizeof(int32_t));
  L.rec_u = take((size_t)P * n * sizeof(double));
  L.rec_aux = take((size_t)P * n * sizeof(double));
  L.rec_bin = take((size_t)P * n);
  L.gbuf = take((size_t)2 * 4 * n * sizeof(double));
  L.total = o;
  return L;
}

static bool scene2_ok(const tfrt_scene2d* sc) {
  if (!sc || sc->n_segments < 0 || sc->n_arcs < 0) return false;
  if (sc->n_segments > 0 && (!sc->seg || !sc->seg_cat)) return false;
  if (sc->n_arcs > 0 && (!sc->arc || !sc->arc_cat)) return false;
  if (sc->n_segments + sc->n_arcs >= (1ll << 30)) return false;
  const bool table = sc->n_table != nullptr;
  if (sc->n_segments > 0 && !((table && sc->seg_mat_in && sc->seg_mat_out) ||
                              (sc->seg_n_in && sc->seg_n_out)))
    return false;
  if (sc->n_arcs > 0 && !((table && sc->arc_mat_in && sc->arc_mat_out) ||
                          (sc->arc_n_in && sc->arc_n_out)))
    return false;
  return true;
}

template <typename T>
static int trace2d_forward_t(const void* src_rays, int64_t src_stride, int64_t N,
                             const tfrt_scene2d* sc, double L, double dead_len, int P, int dtype,
                             uint32_t flags, tfrt_ray_out* fin, tfrt_ray_out* act,
                             tfrt_ray_out* stp, tfrt_ray_out* dead, void* unfinished,
                             int32_t* unfinished_id, int32_t* counts, void* workspace,
                             size_t workspace_bytes, hipStream_t st) {
  const Layout2 lay = make_layout2(N, P, dtype);
  if (workspace_bytes < lay.total) return TFRT_E_WORKSPACE;
  char* ws = static_cast<char*>(workspace);
  int32_t* nrays = reinterpret_cast<int32_t*>(ws + lay.nrays);
  int32_t* blockcnt = reinterpret_cast<int32_t*>(ws + lay.blockcnt);
  int32_t* blockoff = reinterpret_cast<int32_t*>(ws + lay.blockoff);
  int32_t* rowtot = reinterpret_cast<int32_t*>(ws + lay.rowtot);
  int32_t* rowbase = reinterpret_cast<int32_t*>(ws + lay.rowbase);
  unsigned int* ticket = reinterpret_cast<unsigned int*>(ws + lay.ticket);
  int32_t* bincnt = reinterpret_cast<int32_t*>(ws + lay.bincnt);
  T* rays_ws = reinterpret_cast<T*>(ws + lay.rays);
  int32_t* rayid = reinterpret_cast<int32_t*>(ws + lay.rayid);
  int32_t* lastprim = reinterpret_cast<int32_t*>(ws + lay.lastprim);
  int32_t* rec_prim = reinterpret_cast<int32_t*>(ws + lay.rec_prim);
  int32_t* rec_slot = reinterpret_cast<int32_t*>(ws + lay.rec_slot);
  double* rec_u = reinterpret_cast<double*>(ws + lay.rec_u);
  double* rec_aux = reinterpret_cast<double*>(ws + lay.rec_aux);
  uint8_t* rec_bin = reinterpret_cast<uint8_t*>(ws + lay.rec_bin);
  int32_t* tail = counts + (size_t)P * TFRT_COUNTS_PER_PASS;
  const size_t n = N > 0 ? N : 1;
  const int M = (int)(sc->n_segments + sc->n_arcs);
  hipLaunchKernelGGL(k_init2, dim3(1), dim3(64), 0, st, nrays, (int)N, tail, ticket);
  const tfrt_ray_out none = {nullptr, nullptr, nullptr, 0};
  for (int p = 0; p < P; ++p) {
    const T* rin = p == 0 ? static_cast<const T*>(src_rays) : rays_ws + (size_t)(p - 1) * 4 * n;
    const int64_t sin = p == 0 ? src_stride : (int64_t)n;
    const int32_t* idin = p == 0 ? nullptr : rayid + (size_t)(p - 1) * n;
    const int32_t* lpin = p == 0 ? nullptr : lastprim + (size_t)(p - 1) * n;
    hipLaunchKernelGGL((k_intersect2d<T>), dim3(lay.nblk), dim3(BLOCK), 0, st, rin, sin, nrays + p,
                       lpin, *sc, rec_prim + (size_t)p * n, rec_u + (size_t)p * n,
                       rec_aux + (size_t)p * n, rec_bin + (size_t)p * n, blockcnt);
    const bool grid_scan = lay.nblk >= SCAN_GRID_MIN_ROWS;
    if (grid_scan)
      hipLaunchKernelGGL(k_scan2d, dim3(cdiv(lay.nblk, 1024)), dim3(1024), 0, st, nrays + p,
                         blockcnt, blockoff, rowtot, rowbase, ticket,
                         counts + (size_t)p * TFRT_COUNTS_PER_PASS, bincnt + (size_t)p * NBIN, tail,
                         nrays + p + 1, reinterpret_cast<unsigned long long*>(tail + 4), M);
    else
      hipLaunchKernelGGL(k_scan2d_one, dim3(1), dim3(1024), 0, st, nrays + p, blockcnt, blockoff,
                         counts + (size_t)p * TFRT_COUNTS_PER_PASS, bincnt + (size_t)p * NBIN, tail,
                         nrays + p + 1, reinterpret_cast<unsigned long long*>(tail + 4), M);
    hipLaunchKernelGGL((k_react2d<T>), dim3(lay.nblk), dim3(BLOCK), 0, st, rin, sin, nrays + p,
                       idin, rec_prim + (size_t)p * n, rec_u + (size_t)p * n,
                       rec_aux + (size_t)p * n, rec_bin + (size_t)p * n, blockoff,
                       grid_scan ? rowbase : static_cast<int32_t*>(nullptr),
                       counts + (size_t)p * TFRT_COUNTS_PER_PASS, bincnt + (size_t)p * NBIN, *sc, L,
                       dead_len, flags, rays_ws + (size_t)p * 4 * n, (int64_t)n,
                       rayid + (size_t)p * n, lastprim + (size_t)p * n, rec_slot + (size_t)p * n,
                       fin ? *fin : none, act ? *act : none, stp ? *stp : none,
                       dead ? *dead : none, tail + 6);
  }
  if (unfinished != nullptr && P > 0) {
    hipLaunchKernelGGL((k_copy_rays2<T>), dim3(lay.nblk), dim3(BLOCK), 0, st,
                       rays_ws + (size_t)(P - 1) * 4 * n, (int64_t)n, rayid + (size_t)(P - 1) * n,
                       nrays + P, static_cast<T*>(unfinished), (int64_t)N, unfinished_id);
  }
  return hipGetLastError() == hipSuccess ? 0 : TFRT_E_LAUNCH;
}

template <typename T>
static int trace2d_backward_t(const void* src_rays, int64_t src_stride, int64_t N,
                              const tfrt_scene2d* sc, double L, double dead_len, int P, int dtype,
                              const double* g_fin, int64_t cap_fin, const double* g_act,
                              int64_t cap_act, const double* g_stp, int64_t cap_stp,
                              const double* g_dead, int64_t cap_dead, double* g_seg, double* g_arc,
                              double* g_src, const int32_t* counts, void* workspace,
                              size_t workspace_bytes, hipStream_t st) {
  const Layout2 lay = make_layout2(N, P, dtype);
  if (workspace_bytes < lay.total) return TFRT_E_WORKSPACE;
  char* ws = static_cast<char*>(workspace);
  const int32_t* nrays = reinterpret_cast<int32_t*>(ws + lay.nrays);
  const T* rays_ws = reinterpret_cast<T*>(ws + lay.rays);
  const int32_t* rayid = reinterpret_cast<int32_t*>(ws + lay.rayid);
  const int32_t* rec_prim = reinterpret_cast<int32_t*>(ws + lay.rec_prim);
  const int32_t* rec_slot = reinterpret_cast<int32_t*>(ws + lay.rec_slot);
  const double* rec_u = reinterpret_cast<double*>(ws + lay.rec_u);
  const uint8_t* rec_bin = reinterpret_cast<uint8_t*>(ws + lay.rec_bin);
  double* gbuf = reinterpret_cast<double*>(ws + lay.gbuf);
  const size_t n = N > 0 ? N : 1;
  for (int p = P - 1; p >= 0; --p) {
    const T* rin = p == 0 ? static_cast<const T*>(src_rays) : rays_ws + (size_t)(p - 1) * 4 * n;
    const int64_t sin = p == 0 ? src_stride : (int64_t)n;
    const int32_t* idin = p == 0 ? nullptr : rayid + (size_t)(p - 1) * n;
    const double* g_child = (p == P - 1) ? nullptr : gbuf + (size_t)((p + 1) & 1) * 4 * n;
    double* g_out = (p == 0 && g_src != nullptr) ? g_src : gbuf + (size_t)(p & 1) * 4 * n;
    const int64_t out_stride = (p == 0 && g_src != nullptr) ? N : (int64_t)n;
    hipLaunchKernelGGL((k_backward2d<T>), dim3(lay.nblk), dim3(BLOCK), 0, st, rin, sin, nrays + p,
                       idin, rec_prim + (size_t)p * n, rec_u + (size_t)p * n,
                       rec_bin + (size_t)p * n, rec_slot + (size_t)p * n,
                       counts + (size_t)p * TFRT_COUNTS_PER_PASS, *sc, L, dead_len, g_child,
                       (int64_t)n, g_fin, cap_fin, g_act, cap_act, g_stp, cap_stp, g_dead, cap_dead,
                       g_out, out_stride, g_seg, g_arc);
  }
  return hipGetLastError() == hipSuccess ? 0 : TFRT_E_LAUNCH;
}

template <bool ARC>
static int seam2d(const void* rays, int64_t stride, int64_t n_rays, int32_t dtype,
                  const double* prim, int64_t M, double ei, double es, double er, double* x,
                  double* y, uint8_t* valid, double* ray_u, double* prim_u, int32_t* gather,
                  void* stream) {
  if (n_rays < 0 || M < 0 || stride < n_rays || (M > 0 && !prim)) return TFRT_E_BADARG;
  if (n_rays == 0) return 0;
  if (!rays || !x || !y || !valid || !ray_u || !prim_u || !gather) return TFRT_E_BADARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(cdiv(n_rays, BLOCK));
  if (dtype == TFRT_F32) {
    hipLaunchKernelGGL((k_seam2d<float, ARC>), grid, dim3(BLOCK), 0, st,
                       static_cast<const float*>(rays), stride, (int)n_rays, prim, (int)M, ei, es,
                       er, x, y, valid, ray_u, prim_u, gather);
  } else if (dtype == TFRT_F64) {
    hipLaunchKernelGGL((k_seam2d<double, ARC>), grid, dim3(BLOCK), 0, st,
                       static_cast<const double*>(rays), stride, (int)n_rays, prim, (int)M, ei, es,
                       er, x, y, valid, ray_u, prim_u, gather);
  } else if (dtype == TFRT_F16) {
    hipLaunchKernelGGL((k_seam2d<_Float16, ARC>), grid, dim3(BLOCK), 0, st,
                       static_cast<const _Float16*>(rays), stride, (int)n_rays, prim, (int)M, ei,
                       es, er, x, y, valid, ray_u, prim_u, gather);
  } else {
    return TFRT_E_UNSUPPORTED;
  }
  return hipGetLastError() == hipSuccess ? 0 : TFRT_E_LAUNCH;
}

}  // namespace tfrt

using namespace tfrt;

extern "C" {

int tfrt_segment_intersection(const void* rays, int64_t stride, int64_t n_rays,
                              int32_t state_dtype, const double* seg, int64_t n_segments,
                              double intersect_epsilion, double size_epsilion,
                              double ray_start_epsilion, double* x, double* y, uint8_t* valid,
                              double* ray_u, double* seg_u, int32_t* gather_segment,
                              void* stream) {
  return seam2d<false>(rays, stride, n_rays, state_dtype, seg, n_segments, intersect_epsilion,
                       size_epsilion, ray_start_epsilion, x, y, valid, ray_u, seg_u,
                       gather_segment, stream);
}

int tfrt_arc_intersection(const void* rays, int64_t stride, int64_t n_rays, int32_t state_dtype,
                          const double* arc, int64_t n_arcs, double intersect_epsilion,
                          double size_epsilion, double ray_start_epsilion, double* x, double* y,
                          uint8_t* valid, double* ray_u, double* arc_u, int32_t* gather_arc,
                          void* stream) {
  return seam2d<true>(rays, stride, n_rays, state_dtype, arc, n_arcs, intersect_epsilion,
                      size_epsilion, ray_start_epsilion, x, y, valid, ray_u, arc_u, gather_arc,
                      stream);
}

size_t tfrt_trace2d_workspace_bytes(int64_t n_rays, int64_t n_segments, int64_t n_arcs,
                                    int32_t max_passes, int32_t state_dtype) {
  if (n_rays < 0 || n_segments < 0 || n_arcs < 0 || max_passes < 0) return 0;
  return make_layout2(n_rays, max_passes, state_dtype).total;
}

int tfrt_trace2d_forward(const void* src_rays, int64_t src_stride, int64_t n_rays,
                         const tfrt_scene2d* scene, double new_ray_length,
                         double dead_ray_length, int32_t max_passes, int32_t state_dtype,
                         uint32_t flags, tfrt_ray_out* finished, tfrt_ray_out* active,
                         tfrt_ray_out* stopped, tfrt_ray_out* dead, void* unfinished,
                         int32_t* unfinished_id, int32_t* counts, void* workspace,
                         size_t workspace_bytes, void* stream) {
  if (!scene2_ok(scene) || n_rays < 0 || n_rays >= (1ll << 31) - 4096 || max_passes < 0 ||
      !counts || !workspace || (n_rays > 0 && !src_rays) || src_stride < n_rays)
    return TFRT_E_BADARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (state_dtype == TFRT_F32)
    return trace2d_forward_t<float>(src_rays, src_stride, n_rays, scene, new_ray_length,
                                    dead_ray_length, max_passes, state_dtype, flags, finished,
                                    active, stopped, dead, unfinished, unfinished_id, counts,
                                    workspace, workspace_bytes, st);
  if (state_dtype == TFRT_F64)
    return trace2d_forward_t<double>(src_rays, src_stride, n_rays, scene, new_ray_length,
                                     dead_ray_length, max_passes, state_dtype, flags, finished,
                                     active, stopped, dead, unfinished, unfinished_id, counts,
                                     workspace, workspace_bytes, st);
  if (state_dtype == TFRT_F16)
    return trace2d_forward_t<_Float16>(src_rays, src_stride, n_rays, scene, new_ray_length,
                                     dead_ray_length, max_passes, state_dtype, flags, finished,
                                     active, stopped, dead, unfinished, unfinished_id, counts,
                                     workspace, workspace_bytes, st);
  return TFRT_E_UNSUPPORTED;
}

int tfrt_trace2d_backward(const void* src_rays, int64_t src_stride, int64_t n_rays,
                          const tfrt_scene2d* scene, double new_ray_length,
                          double dead_ray_length, int32_t max_passes, int32_t state_dtype,
                          const double* grad_finished, int64_t cap_finished,
                          const double* grad_active, int64_t cap_active,
                          const double* grad_stopped, int64_t cap_stopped,
                          const double* grad_dead, int64_t cap_dead, double* grad_seg,
                          double* grad_arc, double* grad_src_rays, const int32_t* counts,
                          void* workspace, size_t workspace_bytes, void* stream) {
  if (!scene2_ok(scene) || n_rays < 0 || max_passes < 0 || !counts || !workspace)
    return TFRT_E_BADARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (state_dtype == TFRT_F32)
    return trace2d_backward_t<float>(src_rays, src_stride, n_rays, scene, new_ray_length,
                                     dead_ray_length, max_passes, state_dtype, grad_finished,
                                     cap_finished, grad_active, cap_active, grad_stopped,
                                     cap_stopped, grad_dead, cap_dead, grad_seg, grad_arc,
                                     grad_src_rays, counts, workspace, workspace_bytes, st);
  if (state_dtype == TFRT_F64)
    return trace2d_backward_t<double>(src_rays, src_stride, n_rays, scene, new_ray_length,
                                      dead_ray_length, max_passes, state_dtype, grad_finished,
                                      cap_finished, grad_active, cap_active, grad_stopped,
                                      cap_stopped, grad_dead, cap_dead, grad_seg, grad_arc,
                                      grad_src_rays, counts, workspace, workspace_bytes, st);
  if (state_dtype == TFRT_F16)
    return trace2d_backward_t<_Float16>(src_rays, src_stride, n_rays, scene, new_ray_length,
                                      dead_ray_length, max_passes, state_dtype, grad_finished,
                                      cap_finished, grad_active, cap_active, grad_stopped,
                                      cap_stopped, grad_dead, cap_dead, grad_seg, grad_arc,
                                      grad_src_rays, counts, workspace, workspace_bytes, st);
  return TFRT_E_UNSUPPORTED;
}

}  // extern "C"
