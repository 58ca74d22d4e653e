// tfrt 3-D hot path for MI355X (gfx950): ray x triangle intersection, nearest hit, stable
// classification/compaction, Snell update, pass loop and reverse sweep.
//
// Structure of one pass (reference: tfrt/engine.py:2193-2302 single_pass):
//
//   k_rayprep      per ray: float64 ray -> 8 float32 numbers (an orthonormal pair (a, b) spanning
//                  the plane perpendicular to the ray, and -s.a, -s.b in a mesh-centred frame).
//   intersect      decides every ray-face pair of the pass.  The unit of throughput is a
//                  conservative float32 line-vs-sphere test, (c.a - s.a)^2 + (c.b - s.b)^2 <= r^2
//                  (8 FMA-class ops); whatever it lets through is screened by a float32
//                  Moeller-Trumbore evaluation with error bounds (may_hit) and then decided
//                  EXACTLY in float64 with the reference's own Cramer sums and epsilons
//                  (trace_math.h exact_triangle), so hit/miss and nearest-hit decisions are the
//                  float64 reference's decisions; the float32 stages only remove pairs that
//                  cannot hit.  Two kernels share that scheme, a third (rays handed over in a
//                  coherent order) shares the walk of the hierarchy among a wavefront's rays:
//        k_intersect_beam   (tfrt_scene3d.coherent_rays) one bundle per wavefront against the
//                           sphere hierarchy with lane = node, the candidate faces as triangles
//                           seen along the bundle's axis, exact float64 decisions; wavefronts
//                           that are no narrow bundle are cut or left to k_intersect_group.
//        k_intersect_group  (default, scenes of >= 64 faces) sphere hierarchy over k-d face
//                           clusters: 8-cluster superclusters -> 16-face clusters -> faces;
//                           lanes queue the clusters their rays touch, the wave drains the
//                           queues together (member tests, screen, float64 decisions, each on
//                           compacted full wavefronts).
//        k_intersect3d      every pair through the sphere filter: grid (ray blocks, face
//                           chunks), lane = R rays, spheres stream through an LDS tile.
//   k_classify3d   min over chunks (lowest face index wins ties, like tf.argmin), boundary
//                  catagory -> ray class, per-block class histogram.
//   k_scan3d       one block: exclusive scan of the histograms -> stable output slots,
//                  per-pass counts, running totals, next pass's ray count (all on device:
//                  the host never synchronises inside a trace).
//   k_react3d      projects the ray end onto the hit, writes finished / stopped / dead /
//                  active-history rows at their stable slots, refracts/reflects active rays
//                  (float64 Snell) into the next pass's ray block, records the tape.
//
// Backward: k_backward3d walks the tape pass by pass in reverse, recomputes the per-ray
// forward in float64 and applies the hand-derived adjoint (trace_math.h adjoint3d); the face
// gradients of the rays are summed per wavefront in LDS (coherent rays) or left in a per-ray
// stash that k_face_accumulate sums in LDS face windows (natural order).
#include <vector>

#include <type_traits>
#include "tfrt_common.h"
#include "goal_finish.h"

namespace tfrt {

constexpr int TILE = 512;  // spheres per LDS tile (8 KiB; measured best on MI355X)
constexpr int KC = 24;     // candidate slots per lane (24 KiB per block)

// error bits written to counts[...error]
constexpr int ERR_CAPACITY = 1;

// ------------------------------------------------------------------------------ prep

// c0 = mean of the face corner P0 over (a sample of) the faces: origin of the filter's
// coordinate frame (keeps |c| small so the float32 rounding margin stays far below the sphere
// radii).  Any point near the mesh serves, so at most ~2048 evenly spaced faces are read.
// (Also does k_init's work -- the trace's counters -- so a trace with faces needs one set-up
// launch instead of two.)
__global__ __launch_bounds__(BLOCK) void k_center(const double* __restrict__ fverts, int M,
                                                  double* __restrict__ c0, int32_t* nrays0, int n,
                                                  int32_t* tail8, unsigned int* scan_ticket) {
  if (nrays0 != nullptr) {
    if (threadIdx.x == 0) *nrays0 = n;
    if (threadIdx.x < 8) tail8[threadIdx.x] = 0;
    if (threadIdx.x == 0 && scan_ticket != nullptr) *scan_ticket = 0u;
  }
  __shared__ double red[WAVES][3];
  __shared__ double ctr[3];
  const int step = M > 2048 ? M / 2048 : 1;
  const int ns = M > 0 ? (M + step - 1) / step : 0;  // samples j * step, j < ns
  const int wave = threadIdx.x >> 6;
  double a[3] = {0, 0, 0};
  for (int j = threadIdx.x; j < ns; j += BLOCK) {
    const double* P = fverts + 9 * (int64_t)j * step;
    a[0] += P[0];
    a[1] += P[1];
    a[2] += P[2];
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1)
    for (int k = 0; k < 3; ++k) a[k] += __shfl_xor(a[k], d, 64);
  if (lane_id() == 0)
    for (int k = 0; k < 3; ++k) red[wave][k] = a[k];
  __syncthreads();
  if (threadIdx.x < 3) {
    double sum = 0.0;
    for (int w = 0; w < WAVES; ++w) sum += red[w][threadIdx.x];
    // rounded to float32: (float)(s - c0) is then a plain float subtraction for float32 ray
    // state (k_intersect_group forms it for every candidate pair), same value, no conversions
    const float cf = (ns > 0) ? (float)(sum / ns) : 0.f;
    ctr[threadIdx.x] = (cf - cf == 0.f) ? (double)cf : 0.0;  // (inf / NaN: the origin serves)
    c0[threadIdx.x] = ctr[threadIdx.x];
  }
}

// Smallest enclosing sphere of a triangle, inflated so that the float32 filter is
// conservative: r_eff = r (1 + 1e-5) + 64 u32 (|c| + r) + max(size_eps, 0) (|E1| + |E2|).
// cc = centre relative to c0 (float64), reff = inflated radius.
__device__ __forceinline__ void face_sphere(const double* __restrict__ P, const double* c0,
                                            double size_eps, double cc[3], double* reff_out) {
  const double A[3] = {P[0], P[1], P[2]}, B[3] = {P[3], P[4], P[5]}, C[3] = {P[6], P[7], P[8]};
  double ab[3], ac[3], bc[3];
  for (int k = 0; k < 3; ++k) {
    ab[k] = B[k] - A[k];
    ac[k] = C[k] - A[k];
    bc[k] = C[k] - B[k];
  }
  double c[3], r2;
  const double dA = dot3(ab, ac);           // angle at A obtuse/right if <= 0
  const double dB = -dot3(ab, bc);          // (A-B).(C-B)
  const double dC = dot3(ac, bc);           // (A-C).(B-C)
  if (dA <= 0.0) {                          // longest edge BC
    for (int k = 0; k < 3; ++k) c[k] = 0.5 * (B[k] + C[k]);
    r2 = 0.25 * dot3(bc, bc);
  } else if (dB <= 0.0) {                   // longest edge AC
    for (int k = 0; k < 3; ++k) c[k] = 0.5 * (A[k] + C[k]);
    r2 = 0.25 * dot3(ac, ac);
  } else if (dC <= 0.0) {                   // longest edge AB
    for (int k = 0; k < 3; ++k) c[k] = 0.5 * (A[k] + B[k]);
    r2 = 0.25 * dot3(ab, ab);
  } else {                                  // acute: circumcentre
    double n[3], t1[3], t2[3];
    cross3(ab, ac, n);
    const double n2 = dot3(n, n);
    cross3(n, ab, t1);   // (ab x ac) x ab
    cross3(ac, n, t2);   // ac x (ab x ac)
    const double ab2 = dot3(ab, ab), ac2 = dot3(ac, ac);
    double off[3];
    for (int k = 0; k < 3; ++k) off[k] = (ac2 * t1[k] + ab2 * t2[k]) / (2.0 * n2);
    for (int k = 0; k < 3; ++k) c[k] = A[k] + off[k];
    r2 = dot3(off, off);
  }
  double r = sqrt(r2);
  // guard against a degenerate (zero-area) acute classification producing inf/nan
  if (!(r2 == r2) || r2 > 1e300) {
    for (int k = 0; k < 3; ++k) c[k] = (A[k] + B[k] + C[k]) / 3.0;
    r = 0.0;
    for (int v = 0; v < 3; ++v) {
      double d2 = 0;
      for (int k = 0; k < 3; ++k) d2 += (P[3 * v + k] - c[k]) * (P[3 * v + k] - c[k]);
      r = fmax(r, sqrt(d2));
    }
  }
  for (int k = 0; k < 3; ++k) cc[k] = c[k] - c0[k];
  const double cn = sqrt(dot3(cc, cc));
  const double u32 = 5.9604644775390625e-08;  // 2^-24
  double reff = r * (1.0 + 1e-5) + 64.0 * u32 * (cn + r);
  // size_epsilion: trig_u, trig_v >= -eps, trig_u + trig_v <= 1 + eps is the triangle with the
  // corners A - eps (ab + ac), B + eps (2 ab - ac), C + eps (2 ac - ab): every valid hit lies within
  // eps max(|ab + ac|, |2 ab - ac|, |2 ac - ab|) <= 2 eps (|ab| + |ac|) of the face.  (Until
  // round 3 the factor was 1: a hit in the far corner of that margin could be filtered out when
  // size_epsilion was not tiny -- found by the coherent-ray kernel, whose bound was right.)
  if (size_eps > 0.0) reff += 2.0 * size_eps * (sqrt(dot3(ab, ab)) + sqrt(dot3(ac, ac)));
  *reff_out = reff;
}

__device__ __forceinline__ float4 pack_sphere(const double cc[3], double reff) {
  float rf = static_cast<float>(reff * reff);
  rf = nextafterf(rf, INFINITY);
  rf = nextafterf(rf, INFINITY);
  return make_float4((float)cc[0], (float)cc[1], (float)cc[2], rf);
}

// What the reaction needs of a face besides the hit: the unit normal as snell3d uses it and,
// when the refractive indices do not depend on the ray (one wavelength for every ray, or "value"
// mode), the two index ratios and n_in -- formed ONCE per face by the trace's set-up launch
// (the same operations in the same order as snell3d's, hence the same bits) instead of once per
// ray and pass: a cross product, two square roots and six float64 divisions less per ray, 48 B
// gathered for 72 + two dependent index loads.
struct FaceTables {
  double* fnorm = nullptr;           // (M, 3)
  double* feta = nullptr;            // (M, 4): n1, n2, n_in, n_out -- or nullptr: indices per ray
  const int32_t* mat_in = nullptr;
  const int32_t* mat_out = nullptr;
  const double* n_table = nullptr;   // (column 0: n_table_uniform)
  int64_t n_table_stride = 0;
  const double* n_in = nullptr;
  const double* n_out = nullptr;
};

__device__ __forceinline__ void face_tables(const FaceTables& ft, const double* __restrict__ P9,
                                            int f) {
  if (ft.fnorm != nullptr) {
    double P[9], un[3];
    for (int q = 0; q < 9; ++q) P[q] = P9[q];
    snell_normal(P, un);
    for (int q = 0; q < 3; ++q) ft.fnorm[3 * (int64_t)f + q] = un[q];
  }
  if (ft.feta != nullptr) {
    double ni, no, n1, n2;
    if (ft.n_table != nullptr && ft.mat_in != nullptr) {
      ni = ft.n_table[(int64_t)ft.mat_in[f] * ft.n_table_stride];
      no = ft.n_table[(int64_t)ft.mat_out[f] * ft.n_table_stride];
    } else {
      ni = ft.n_in[f];
      no = ft.n_out[f];
    }
    snell_ratios(ni, no, &n1, &n2);
    ft.feta[4 * (int64_t)f] = n1;
    ft.feta[4 * (int64_t)f + 1] = n2;
    ft.feta[4 * (int64_t)f + 2] = ni;
    ft.feta[4 * (int64_t)f + 3] = no;
  }
}

__global__ __launch_bounds__(BLOCK) void k_spheres(const double* __restrict__ fverts, int M,
                                                   const double* __restrict__ c0,
                                                   double size_eps,
                                                   float4* __restrict__ sphere,
                                                   FaceTables ft,
                                                   double* __restrict__ clear_buf,
                                                   int64_t clear_n) {
  const int j = blockIdx.x * BLOCK + threadIdx.x;
  for (int64_t k = j; k < clear_n; k += (int64_t)gridDim.x * BLOCK) clear_buf[k] = 0.0;
  if (j >= M) return;
  double cc[3], reff;
  face_sphere(fverts + 9 * (int64_t)j, c0, size_eps, cc, &reff);
  sphere[j] = pack_sphere(cc, reff);
  face_tables(ft, fverts + 9 * (int64_t)j, j);
}

// Hierarchy modes: faces are visited in `order` (spatially coherent groups of CLUSTER faces).
// 16 lanes per cluster, one member face each: the lane writes its member's sphere (cluster
// order; padding entries never hit), the member -> face map and the float32 face record; the
// group then builds the cluster's bounding sphere together.
//
// The cluster sphere only has to contain the member TRIANGLES (plus the size_eps margin by
// which the exact test accepts points outside them), not the members' own bounding spheres:
// a line that reaches a valid hit point passes through it either way.  Its centre starts at the
// mean of the member centres and takes 12 Badoiu-Clarkson steps (move towards the farthest
// vertex by 1/(k+1)) towards the minimal enclosing ball of the 48 vertices: ~10 % less radius,
// ~20 % fewer rays per cluster.
constexpr int CLUSTER = 16;
// Badoiu-Clarkson steps of the cluster / supercluster balls (12 steps gave the same candidate
// counts per wavefront and ray, and a set-up launch 5 us longer)
constexpr int BC_STEPS = 6;

// Funnel counters of k_intersect_group: tuning builds only (-DTFRT_TUNING, csrc/tfrt_tuning.h;
// never in the shipped library).
#ifdef TFRT_TUNING
#include "tfrt_tuning.h"
#else
#define TFRT_STAT(k, v) do { } while (0)
#define TFRT_TICK_INIT do { } while (0)
#define TFRT_TICK(k) do { } while (0)
#define TFRT_WAVE_BEGIN do { } while (0)
#define TFRT_WAVE_END(qw) do { } while (0)
#define TFRT_WAVE_NOTE(k, v) do { } while (0)
#endif

__device__ __forceinline__ void cluster_spheres_block(
    const int block, const double* __restrict__ fverts, int M, const int32_t* __restrict__ order,
    const double* __restrict__ c0, double size_eps, int n_clusters,
    float4* __restrict__ csphere, int32_t* __restrict__ cface, float4* __restrict__ clsphere,
    float4* __restrict__ crec, const FaceTables& ft) {
  const int k = block * BLOCK + threadIdx.x;  // member slot = cluster * CLUSTER + member
  const int c = k / CLUSTER;
  const bool in_range = c < n_clusters;            // uniform over the 16 lanes of a cluster
  int f = (in_range && k < M) ? order[k] : -1;
  if (f < 0 || f >= M) f = -1;
  double V[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};  // vertices relative to c0
  double ctr[3] = {0, 0, 0}, rad = -1.0, edge = 0.0;
  if (f >= 0) {
    const double* P = fverts + 9 * (int64_t)f;
    face_sphere(P, c0, size_eps, ctr, &rad);
    for (int v = 0; v < 3; ++v)
      for (int q = 0; q < 3; ++q) V[v][q] = P[3 * v + q] - c0[q];
    for (int v = 0; v < 3; ++v) {
      double e2 = 0;
      for (int q = 0; q < 3; ++q) e2 += (V[(v + 1) % 3][q] - V[v][q]) * (V[(v + 1) % 3][q] - V[v][q]);
      edge = fmax(edge, sqrt(e2));
    }
    face_tables(ft, P, f);
  }
  if (in_range) {
    cface[k] = f;
    csphere[k] = f >= 0 ? pack_sphere(ctr, rad) : make_float4(0.f, 0.f, 0.f, -1.f);
    if (crec != nullptr) {  // float32 record for the screen: P0 - c0, P1 - P0, P2 - P0
      // (.w of the first entry carries the face index: the screen needs no second gather for it)
      crec[3 * (int64_t)k] =
          make_float4((float)V[0][0], (float)V[0][1], (float)V[0][2], __int_as_float(f));
      crec[3 * (int64_t)k + 1] = make_float4((float)(V[1][0] - V[0][0]), (float)(V[1][1] - V[0][1]),
                                             (float)(V[1][2] - V[0][2]), 0.f);
      crec[3 * (int64_t)k + 2] = make_float4((float)(V[2][0] - V[0][0]), (float)(V[2][1] - V[0][1]),
                                             (float)(V[2][2] - V[0][2]), 0.f);
    }
  }
  // reductions over the 16 lanes of the cluster (xor shuffles stay inside aligned groups of 16)
  auto sum16 = [&](double v) {
#pragma unroll
    for (int d = 8; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
  };
  auto max16 = [&](double v) {
#pragma unroll
    for (int d = 8; d > 0; d >>= 1) v = fmax(v, __shfl_xor(v, d, 64));
    return v;
  };
  const double cnt = sum16(f >= 0 ? 1.0 : 0.0);
  double mean[3];
  for (int q = 0; q < 3; ++q) mean[q] = sum16(f >= 0 ? ctr[q] : 0.0) / fmax(cnt, 1.0);
  const int lane = threadIdx.x & 63;
  double R = 0.0;
  for (int it = 1; it <= BC_STEPS + 1; ++it) {
    // this lane's farthest vertex from the current centre
    double best = -1.0;
    int bv = 0;
    if (f >= 0) {
      for (int v = 0; v < 3; ++v) {
        double d2 = 0;
        for (int q = 0; q < 3; ++q) d2 += (V[v][q] - mean[q]) * (V[v][q] - mean[q]);
        if (d2 > best) {
          best = d2;
          bv = v;
        }
      }
    }
    const double top = max16(best);
    if (it == BC_STEPS + 1) {  // final radius about the final centre
      R = sqrt(fmax(top, 0.0));
      break;
    }
    // the lowest lane of the group holding the maximum provides the vertex
    const unsigned long long holders = __ballot(best == top && f >= 0);
    const int group0 = lane & ~(CLUSTER - 1);
    const unsigned long long mine = (holders >> group0) & 0xFFFFull;
    const int src = group0 + (mine ? __ffsll((long long)mine) - 1 : 0);
    for (int q = 0; q < 3; ++q) {
      const double mine_q = bv == 0 ? V[0][q] : (bv == 1 ? V[1][q] : V[2][q]);  // no dynamic index
      const double fq = __shfl(mine_q, src, 64);
      mean[q] += (fq - mean[q]) / (it + 1);
    }
  }
  const double max_edge = max16(edge);
  if (in_range && (threadIdx.x & (CLUSTER - 1)) == 0) {
    if (cnt > 0.0) {
      if (size_eps > 0.0) R += 4.0 * size_eps * max_edge;  // as face_sphere(): 2 size_eps (|E1| + |E2|)
      const double cn = sqrt(dot3(mean, mean));
      R = R * (1.0 + 1e-5) + 64.0 * 5.9604644775390625e-08 * (cn + R);
      clsphere[c] = pack_sphere(mean, R);
    } else {
      clsphere[c] = make_float4(0.f, 0.f, 0.f, -1.f);
    }
  }
}

// Level 0 of the grouped filter: one bounding sphere per SUPER consecutive clusters (with a
// k-d face order every aligned run of SUPER * CLUSTER faces is one subtree, i.e. one patch).
// Like a cluster sphere it only has to contain the member triangles; one 128-thread block per
// supercluster (thread = face) walks the centre towards the minimal enclosing ball of the 384
// vertices (a sphere around the cluster spheres was ~30 % larger: 9.7 instead of ~6
// superclusters touched per ray, and the per-lane cluster loop runs once per touched one).
constexpr int SUPER = 8;

__device__ __forceinline__ void super_spheres_block(
    const int s, const double* __restrict__ fverts, int M, const int32_t* __restrict__ order,
    const double* __restrict__ c0, double size_eps, float4* __restrict__ susphere) {
  constexpr int NT = SUPER * CLUSTER;  // 128 threads = 2 waves (the caller retires the others)
  __shared__ double red[NT / 64][4];
  __shared__ double pick[3];
  const int t = threadIdx.x;
  const int k = s * NT + t;
  int f = k < M ? order[k] : -1;
  if (f < 0 || f >= M) f = -1;
  double V[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  double edge = 0.0;
  if (f >= 0) {
    const double* P = fverts + 9 * (int64_t)f;
    for (int v = 0; v < 3; ++v)
      for (int q = 0; q < 3; ++q) V[v][q] = P[3 * v + q] - c0[q];
    for (int v = 0; v < 3; ++v) {
      double e2 = 0;
      for (int q = 0; q < 3; ++q) e2 += (V[(v + 1) % 3][q] - V[v][q]) * (V[(v + 1) % 3][q] - V[v][q]);
      edge = fmax(edge, sqrt(e2));
    }
  }
  auto block_sum = [&](double v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    __syncthreads();
    if ((t & 63) == 0) red[t >> 6][0] = v;
    __syncthreads();
    return red[0][0] + red[1][0];
  };
  auto block_max = [&](double v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = fmax(v, __shfl_xor(v, d, 64));
    __syncthreads();
    if ((t & 63) == 0) red[t >> 6][0] = v;
    __syncthreads();
    return fmax(red[0][0], red[1][0]);
  };
  const double cnt = block_sum(f >= 0 ? 1.0 : 0.0);
  if (cnt == 0.0) {  // block-uniform
    if (t == 0) susphere[s] = make_float4(0.f, 0.f, 0.f, -1.f);
    return;
  }
  double mean[3];
  for (int q = 0; q < 3; ++q)
    mean[q] = block_sum(f >= 0 ? (V[0][q] + V[1][q] + V[2][q]) / 3.0 : 0.0) / cnt;
  double R = 0.0;
  for (int it = 1; it <= BC_STEPS + 1; ++it) {
    double best = -1.0;
    int bv = 0;
    if (f >= 0) {
      for (int v = 0; v < 3; ++v) {
        double d2 = 0;
        for (int q = 0; q < 3; ++q) d2 += (V[v][q] - mean[q]) * (V[v][q] - mean[q]);
        if (d2 > best) {
          best = d2;
          bv = v;
        }
      }
    }
    const double top = block_max(best);
    if (it == BC_STEPS + 1) {
      R = sqrt(fmax(top, 0.0));
      break;
    }
    // any thread holding the maximum publishes its vertex (equal distances: any of them serves)
    if (f >= 0 && best == top) {
      for (int q = 0; q < 3; ++q) pick[q] = bv == 0 ? V[0][q] : (bv == 1 ? V[1][q] : V[2][q]);
    }
    __syncthreads();
    for (int q = 0; q < 3; ++q) mean[q] += (pick[q] - mean[q]) / (it + 1);
    __syncthreads();
  }
  const double max_edge = block_max(edge);
  if (t == 0) {
    if (size_eps > 0.0) R += 4.0 * size_eps * max_edge;
    const double cn = sqrt(dot3(mean, mean));
    R = R * (1.0 + 1e-5) + 64.0 * 5.9604644775390625e-08 * (cn + R);
    susphere[s] = pack_sphere(mean, R);
  }
}

// Both sphere levels of the grouped filter in one launch (each is a latency-bound ~12 us
// kernel on its own and neither reads the other's output): blocks [0, n_super) walk one
// supercluster ball each with their first 128 threads, the rest do 16 clusters each.
//
// The launch also does k_center's work (one set-up launch per trace instead of two): every block
// derives the frame origin itself -- the mean of P0 over 64 evenly spaced faces, one wave, the
// same arithmetic in every block, rounded to float32 like k_center's -- and block 0 publishes it
// for the kernels that follow and clears the trace's counters.
// What k_trace_inplace needs once per pass, behind the beam walk (the reaction and the tape): kept
// in device memory -- the set-up launch writes it -- and re-read by scalar loads when a pass gets
// there.  As by-value kernel arguments these twenty pointers and sizes stayed in scalar registers
// across the whole walk; the compiler spilled them to lanes of a vector register and reloaded
// them sixteen at a time, ~80 v_readlane per pass.
struct InplaceTape {
  void* rays_ws;         // child of pass p (input of pass p + 1) at (p * 6 + k) * n + ray
  int32_t* rec_tri;      // [p * n + ray]
  double* rec_t;
  uint8_t* rec_cls;
  int64_t n;
  uint32_t* wcount;      // [p * wstride + wavefront]: four class counts, one byte each
  int64_t wstride;       // rows P and P + 1: the wavefront's executed work (beam_pass `work`)
  const int32_t* catagory;
  const double* fnorm;   // FaceTables
  const double* feta;    // ... or null: the indices depend on the ray (n_table, one column per ray)
  const double* n_table;
  const int32_t* mat_in;
  const int32_t* mat_out;
  int64_t n_table_stride;
  double L;
  // tfrt_scene3d.in_place == 2: every ray's finished row at the ray's own column (see there)
  void* fin_rows;        // 6 x fin_cap, or null
  int64_t fin_cap;
  int32_t* fin_face;     // face of the target hit, -1: the ray did not finish
  int32_t* fin_passes;   // passes the ray entered
};

__global__ __launch_bounds__(BLOCK) void k_hierarchy_spheres(
    const double* __restrict__ fverts, int M, const int32_t* __restrict__ order,
    double* __restrict__ c0_out, double size_eps, int n_clusters, int n_super,
    float4* __restrict__ csphere, int32_t* __restrict__ cface, float4* __restrict__ clsphere,
    float4* __restrict__ crec, float4* __restrict__ susphere, int32_t* nrays0, int n,
    int32_t* tail8, unsigned int* scan_ticket, int32_t* __restrict__ hist0, int hist_len,
    FaceTables ft, double* __restrict__ clear_buf, int64_t clear_n, InplaceTape tape,
    InplaceTape* __restrict__ tape_out) {
  if (tape_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *tape_out = tape;
  // (coherent-ray traces: the class histogram the first pass's intersect kernels add into)
  for (int k = blockIdx.x * BLOCK + threadIdx.x; k < hist_len; k += gridDim.x * BLOCK) hist0[k] = 0;
  // (tfrt_scene3d.clear_buffer: the block a reverse sweep will accumulate into)
  for (int64_t k = (int64_t)blockIdx.x * BLOCK + threadIdx.x; k < clear_n;
       k += (int64_t)gridDim.x * BLOCK)
    clear_buf[k] = 0.0;
  __shared__ double c0[3];
  if (threadIdx.x < 64) {
    const int step = M > 64 ? M / 64 : 1;
    const int ns = (M + step - 1) / step < 64 ? (M + step - 1) / step : 64;
    double a[3] = {0.0, 0.0, 0.0};
    if ((int)threadIdx.x < ns) {
      const double* P = fverts + 9 * (int64_t)threadIdx.x * step;
      a[0] = P[0];
      a[1] = P[1];
      a[2] = P[2];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1)
      for (int k = 0; k < 3; ++k) a[k] += __shfl_xor(a[k], d, 64);
    if (threadIdx.x < 3) {
      const float cf = (ns > 0) ? (float)(a[threadIdx.x] / ns) : 0.f;
      c0[threadIdx.x] = (cf - cf == 0.f) ? (double)cf : 0.0;  // (inf / NaN: the origin serves)
    }
  }
  __syncthreads();
  if (blockIdx.x == 0) {
    if (threadIdx.x < 3) c0_out[threadIdx.x] = c0[threadIdx.x];
    if (threadIdx.x == 0) *nrays0 = n;
    if (threadIdx.x < 8) tail8[threadIdx.x] = 0;
    if (threadIdx.x == 0 && scan_ticket != nullptr) *scan_ticket = 0u;
  }
  if ((int)blockIdx.x < n_super) {
    if (threadIdx.x >= SUPER * CLUSTER) return;  // (whole waves: 128 is a multiple of 64)
    super_spheres_block(blockIdx.x, fverts, M, order, c0, size_eps, susphere);
  } else {
    cluster_spheres_block(blockIdx.x - n_super, fverts, M, order, c0, size_eps, n_clusters,
                          csphere, cface, clsphere, crec, ft);
  }
}

// ---------------------------------------------------------------------- ray filter state

// Per ray and pass, once: the float32 filter state (an orthonormal pair (a, b) spanning the
// plane perpendicular to the ray and the offsets -s.a, -s.b, in the c0 frame), 8 floats per
// ray stored SoA.  Computing it here instead of in every (ray block, face chunk) workgroup of
// k_intersect3d keeps the float64 sqrt/divide work off the hot kernel when faces are chunked.
// o[0..5] = (a, b), o[6..7] = (-s.a, -s.b).  Returns false (NaN offsets: never a candidate)
// for a zero-length or non-finite ray, which can hit nothing (den = 0).
__device__ __forceinline__ bool ray_filter_state(const double s[3], const double e[3],
                                                 const double* __restrict__ c0, float o[8],
                                                 double u[3], double sc[3]) {
  for (int k = 0; k < 6; ++k) o[k] = 0.f;
  o[6] = o[7] = __builtin_nanf("");
  const double d[3] = {e[0] - s[0], e[1] - s[1], e[2] - s[2]};
  const double l2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
  if (!(l2 > 0.0 && l2 < INFINITY)) return false;
  const double inv = 1.0 / sqrt(l2);
  for (int k = 0; k < 3; ++k) {
    u[k] = d[k] * inv;
    sc[k] = s[k] - c0[k];
  }
  // a = normalize(u x e_k) with e_k the axis least aligned with u; b = u x a
  const double f0 = fabs(u[0]), f1 = fabs(u[1]), f2 = fabs(u[2]);
  const bool k0 = (f0 <= f1 && f0 <= f2), k1 = !k0 && (f1 <= f2);
  const double ek[3] = {k0 ? 1.0 : 0.0, k1 ? 1.0 : 0.0, (!k0 && !k1) ? 1.0 : 0.0};
  double a[3], b[3];
  cross3(u, ek, a);
  const double ia = 1.0 / sqrt(dot3(a, a));
  a[0] *= ia; a[1] *= ia; a[2] *= ia;
  cross3(u, a, b);
  o[0] = (float)a[0]; o[1] = (float)a[1]; o[2] = (float)a[2];
  o[3] = (float)b[0]; o[4] = (float)b[1]; o[5] = (float)b[2];
  o[6] = -(float)dot3(sc, a);
  o[7] = -(float)dot3(sc, b);
  return true;
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void k_rayprep(const T* __restrict__ rays, int64_t stride,
                                                   const int32_t* __restrict__ n_ptr,
                                                   const double* __restrict__ c0,
                                                   float* __restrict__ prep, int64_t pstride) {
  const int n = *n_ptr;
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  float o[8];
  double s[3], e[3], u[3], sc[3];
  load_ray3(rays, stride, i, s, e);
  ray_filter_state(s, e, c0, o, u, sc);
#pragma unroll
  for (int k = 0; k < 8; ++k) prep[k * pstride + i] = o[k];
}

// ------------------------------------------------------------------- float32 screen
//
// Second conservative screen between the bounding-sphere filter and the float64 decision: a
// float32 Moeller-Trumbore evaluation with running error bounds.  Returns false only when
// exact_triangle() is certain to report "no valid hit" -- or a hit that cannot displace the
// current nearest one (ray_u certainly above `best`) -- so skipping the float64 test cannot
// change any result.  A line that passes through a face's bounding sphere usually misses the
// face itself (or hits it behind the ray's start): most sphere survivors end here, at ~50
// float32 ops instead of the ~200 float64 ops (three divisions) of the exact sums.
//
// Error model: every quantity is a triple product of float32-rounded differences.  With
// u = 2^-24 and 1-norms nd, n1, n2 (edges, ray) and nt (error scale of t), e.g. for t.(d x e2):
// error of t: 2 u nt nd n2; rounding of d, e2 and of the cross product: <= 10 u |t| nd n2;
// rounding of the dot product: <= 6 u |t| nd n2 -- below 18 u nt nd n2 in total; K = 32 is used.
// d = ray end - start, (ux,uy,uz) = P1 - P0, (vx,vy,vz) = P2 - P0, t = start - P0, all float32;
// nt_err = 1-norm bounding the absolute error of t in units of 2^-24 (|t| itself when t is the
// rounding of an exact difference; larger when its operands were already rounded).
// (nd, n1, n2: 1-norms of d, e1, e2 -- callers that test one face against many rays, or one ray
// against many faces, form them once)
__device__ __forceinline__ bool may_hit_norms(float dx, float dy, float dz, float ux, float uy,
                                              float uz, float vx, float vy, float vz, float tx,
                                              float ty, float tz, float nt_err, float nd, float n1,
                                              float n2, float es, float er, double best) {
  // p = d x e2, q = t x e1
  const float px = dy * vz - dz * vy, py = dz * vx - dx * vz, pz = dx * vy - dy * vx;
  const float qx = ty * uz - tz * uy, qy = tz * ux - tx * uz, qz = tx * uy - ty * ux;
  const float det = ux * px + uy * py + uz * pz;
  const float bu = tx * px + ty * py + tz * pz;   // trig_u * det
  const float bv = dx * qx + dy * qy + dz * qz;   // trig_v * det
  const float bw = vx * qx + vy * qy + vz * qz;   // ray_u  * det
  const float nt = nt_err;
  const float k = 32.0f * 5.9604644775390625e-08f;
  const float e_det = k * nd * n1 * n2, e_u = k * nt * nd * n2, e_v = k * nd * nt * n1;
  const float e_w = k * n2 * nt * n1;
  const float ad = fabsf(det);
  if (!(ad > e_det)) return true;  // too close to parallel to decide here (also NaN/inf)
  const float sg = det > 0.f ? 1.f : -1.f;
  const float lo = ad - e_det, hi = ad + e_det;
  const float su = sg * bu, sv = sg * bv, sw = sg * bw;
  // trig_u < -eps_size, trig_v < -eps_size
  const float thr = fminf(-es * lo, -es * hi) - 1e-30f;
  if (su + e_u < thr) return false;
  if (sv + e_v < thr) return false;
  // trig_u + trig_v > 1 + eps_size
  if (su + sv - e_u - e_v > fmaxf((1.f + es) * lo, (1.f + es) * hi) + 1e-30f) return false;
  // ray_u < eps_start
  if (sw + e_w < fminf(er * lo, er * hi) - 1e-30f) return false;
  // ray_u > best: cannot displace the nearest hit found so far (ties still go to float64)
  if (best < 3.0e38) {
    float bf = (float)best;
    bf = nextafterf(bf, INFINITY);
    if (sw - e_w > fmaxf(bf * lo, bf * hi) + 1e-30f) return false;
  }
  return true;
}

__device__ __forceinline__ bool may_hit_core(float dx, float dy, float dz, float ux, float uy,
                                             float uz, float vx, float vy, float vz, float tx,
                                             float ty, float tz, float nt_err, float es, float er,
                                             double best) {
  return may_hit_norms(dx, dy, dz, ux, uy, uz, vx, vy, vz, tx, ty, tz, nt_err,
                       fabsf(dx) + fabsf(dy) + fabsf(dz), fabsf(ux) + fabsf(uy) + fabsf(uz),
                       fabsf(vx) + fabsf(vy) + fabsf(vz), es, er, best);
}

__device__ __forceinline__ bool may_hit(const double s[3], const double e[3], const double P[9],
                                        double eps_size, double eps_start, double best) {
  const float tx = (float)(s[0] - P[0]), ty = (float)(s[1] - P[1]), tz = (float)(s[2] - P[2]);
  return may_hit_core((float)(e[0] - s[0]), (float)(e[1] - s[1]), (float)(e[2] - s[2]),
                      (float)(P[3] - P[0]), (float)(P[4] - P[1]), (float)(P[5] - P[2]),
                      (float)(P[6] - P[0]), (float)(P[7] - P[1]), (float)(P[8] - P[2]), tx, ty, tz,
                      fabsf(tx) + fabsf(ty) + fabsf(tz), (float)eps_size, (float)eps_start, best);
}

// ------------------------------------------------------------------------- intersect

template <typename T, int R>
__global__ __launch_bounds__(BLOCK) void k_intersect3d(
    const T* __restrict__ rays, int64_t stride, const int32_t* __restrict__ n_ptr,
    const int32_t* __restrict__ last_tri, const float4* __restrict__ sphere,
    const double* __restrict__ fverts, const float* __restrict__ prep, int64_t pstride, int M,
    int chunk_faces, double eps_int, double eps_size, double eps_start,
    double* __restrict__ part_t, int32_t* __restrict__ part_i, int64_t part_stride) {
  const int n = *n_ptr;
  const int base = blockIdx.x * (BLOCK * R);
  if (base >= n) return;  // block-uniform
  const int tid = threadIdx.x;
  const int f0 = blockIdx.y * chunk_faces;
  const int f1 = min(M, f0 + chunk_faces);

  __shared__ float4 tile[TILE + 8];
  __shared__ int32_t cand[KC * BLOCK];

  // per-ray filter state (float32, c0 frame): an orthonormal pair (a, b) spanning the plane
  // perpendicular to the ray, and the ray's offsets -s.a, -s.b in it.  The squared distance
  // from a sphere centre c to the ray's line is (c.a - s.a)^2 + (c.b - s.b)^2: 8 VALU ops.
  float ax[R], ay[R], az[R], bx[R], by[R], bz[R], nsa[R], nsb[R];
  double bt[R];
  int32_t bi[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = base + r * BLOCK + tid;
    bt[r] = INFINITY;
    bi[r] = -1;
    ax[r] = ay[r] = az[r] = bx[r] = by[r] = bz[r] = 0.f;
    // NaN, not +inf: a lane without a ray must never pass `d2 <= r2`, even for a face whose
    // inflated radius overflowed to +inf (size_eps = 1e300 does that); an inf here let ray-less
    // lanes queue candidates and read ray slots >= n
    nsa[r] = nsb[r] = __builtin_nanf("");
    if (i < n) {
      ax[r] = prep[i];
      ay[r] = prep[pstride + i];
      az[r] = prep[2 * pstride + i];
      bx[r] = prep[3 * pstride + i];
      by[r] = prep[4 * pstride + i];
      bz[r] = prep[5 * pstride + i];
      nsa[r] = prep[6 * pstride + i];
      nsb[r] = prep[7 * pstride + i];
    }
  }

  static_assert(KC >= 4 * R + 4, "candidate queue too small");
  int cnt = 0;
  // exact float64 decision for every queued candidate of this lane
  auto flush = [&]() {
    for (int k = 0; k < cnt; ++k) {
      const int v = cand[k * BLOCK + tid];
      const int j = v >> 2;
      const int r = v & 3;
      const int i = base + r * BLOCK + tid;
      if (i >= n) continue;  // (cannot happen: ray-less lanes never queue)
      if (last_tri != nullptr && last_tri[i] == j) continue;  // face the ray starts on
      double s[3], e[3], P[9];
      load_ray3(rays, stride, i, s, e);
      const double* fp = fverts + 9 * (int64_t)j;
#pragma unroll
      for (int q = 0; q < 9; ++q) P[q] = fp[q];
      double best = bt[0];
#pragma unroll
      for (int rr = 1; rr < R; ++rr)
        if (rr == r) best = bt[rr];
      if (!may_hit(s, e, P, eps_size, eps_start, best)) continue;
      const TriHit h = exact_triangle(s, e, P, eps_int, eps_size, eps_start);
      if (h.valid) {
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
          if (rr == r && h.ray_u < bt[rr]) {  // strict <: lowest face index wins ties
            bt[rr] = h.ray_u;
            bi[rr] = j;
          }
        }
      }
    }
    cnt = 0;
  };

  // One sphere against this lane's R rays: 8 FMA-class ops per ray, a min tree and ONE
  // compare; the (rare) survivors are queued for the exact float64 stage.
  auto test = [&](const float4 sp, const int face) {
    float q[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float pa = fmaf(sp.x, ax[r], fmaf(sp.y, ay[r], fmaf(sp.z, az[r], nsa[r])));
      const float pb = fmaf(sp.x, bx[r], fmaf(sp.y, by[r], fmaf(sp.z, bz[r], nsb[r])));
      q[r] = fmaf(pa, pa, pb * pb);
    }
    float qmin = q[0];
#pragma unroll
    for (int r = 1; r < R; ++r) qmin = fminf(qmin, q[r]);
    if (qmin <= sp.w) {  // rare
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (q[r] <= sp.w) {
          cand[cnt * BLOCK + tid] = (face << 2) | r;
          ++cnt;
        }
      }
    }
  };

  const float4 never = make_float4(0.f, 0.f, 0.f, -1.f);  // |w|^2 <= -1 is never true
  for (int t0 = f0; t0 < f1; t0 += TILE) {
    const int nt = min(TILE, f1 - t0);
    const int nt4 = (nt + 3) & ~3;
    __syncthreads();  // previous tile fully consumed
    for (int k = tid; k < nt4; k += BLOCK) tile[k] = (k < nt) ? sphere[t0 + k] : never;
    __syncthreads();
    for (int j = 0; j < nt4; j += 4) {
      // same address in every lane: LDS broadcast reads, four in flight
      const float4 s0 = tile[j], s1 = tile[j + 1], s2 = tile[j + 2], s3 = tile[j + 3];
      test(s0, t0 + j);
      test(s1, t0 + j + 1);
      test(s2, t0 + j + 2);
      test(s3, t0 + j + 3);
      // Wave-uniform flush: when any lane could overflow within the next group of four
      // spheres, every lane decides its queued candidates now (one reconverged pass of
      // the float64 stage instead of one divergent pass per overflowing lane).
      if (__any(cnt > KC - 4 * R)) flush();
    }
  }
  flush();

#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = base + r * BLOCK + tid;
    if (i < n) {
      part_t[blockIdx.y * part_stride + i] = bt[r];
      part_i[blockIdx.y * part_stride + i] = bi[r];
    }
  }
}

// ---------------------------------------------------------------- grouped intersect
//
// Two-level filter in the rays' natural order (no sort).
//   level 1 (per lane)  : each ray against the bounding spheres of the face clusters (CLUSTER
//                         faces each: M/16 spheres, the loop that is M spheres long in
//                         k_intersect3d); clusters a ray's line touches go to the lane's queue.
//   level 2 (per wave)  : the wave drains its lanes' queues together -- 16 lanes take one
//                         queued (ray, cluster) and test the cluster's 16 member spheres, one
//                         each (the 256 B of member spheres are one coalesced read; per-lane
//                         gathers of them cost 16 uncoalesced loads per candidate and were the
//                         bottleneck); member hits become (ray, face) pairs.
//   decision (per wave) : pairs are dealt one per lane: float32 screen, then the exact float64
//                         test; the nearest hit per ray is kept in LDS with 64-bit atomic min
//                         on an order-preserving key of ray_u, ties to the lower face index.
// All filters are conservative, so the float64 stage sees every pair that can win and the
// results equal those of k_intersect3d.
__device__ __forceinline__ unsigned long long dkey(double x) {  // monotone double -> u64
  // (-0.0 and +0.0 get ONE key: they compare equal in the reference's argmin, engine.py:1148, so
  // the lower face index must win between them -- two coplanar faces through a ray's start give
  // ray_u = -0.0 and +0.0, valid hits once ray_start_epsilion <= 0)
  const unsigned long long b = x == 0.0 ? 0ull : (unsigned long long)__double_as_longlong(x);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double dkey_inv(unsigned long long k) {
  const unsigned long long b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
  return __longlong_as_double((long long)b);
}

// class byte of the tape: bits 0-1 the ray class, bits 2-3 the branches of the forward reaction
constexpr int TAPE_INTERNAL = 4, TAPE_REFLECT = 8;

__device__ __forceinline__ int cat_to_cls(int cat) {
  return cat == CAT_OPTICAL ? CLS_ACTIVE : (cat == CAT_TARGET ? CLS_FINISHED : CLS_STOPPED);
}

// acc = 2 acc + (q <= w), w wave-uniform (a scalar register): v_cmp + v_addc.  (NaN q: 0.)
__device__ __forceinline__ void shift_in_le(unsigned& acc, const float q, const float w) {
  __asm__ volatile("v_cmp_ge_f32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                   : "+v"(acc)
                   : "v"(q), "s"(w)
                   : "vcc");
}

// Five workgroups per CU (30.8 KB of LDS each: no copy of the rays in LDS, a 640-entry candidate
// list) and five waves per SIMD (94 VGPRs, no spills).  Measured at 1M rays, optimiser step: with the
// 7,100-instruction kernel of the round's first half 5 waves were no faster than 4 (236 vs 230 us per
// launch); after the instruction trimming, with the screen / decision stages' gather latency a
// larger share of a wave's life, 0.809 ms against 0.826 (the four-wave configuration kept a copy of
// the rays in LDS and a 1024-entry list).
// (only the shipped one-ray-per-lane instantiation: 2 or 4 rays per lane need more LDS than that)
#define TFRT_GROUP_ATTR __attribute__((amdgpu_waves_per_eu(5, 5)))
// ORD: the wave does one wavefront that k_intersect_beam has left (k_intersect_group_left); a
// compile-time switch, so that the natural-order kernel keeps its register budget
template <typename T, int R, bool ORD>
__device__ __forceinline__ void group_walk(
    const T* __restrict__ rays, int64_t stride, const int n,
    const int32_t* __restrict__ last_tri, const float4* __restrict__ susphere,
    const float4* __restrict__ clsphere, const float4* __restrict__ csphere,
    const float4* __restrict__ crec, const int32_t* __restrict__ cface,
    const double* __restrict__ fverts, const double* __restrict__ c0,
    const float* __restrict__ prep, int64_t pstride, int n_clusters, int chunk_clusters,
    double eps_int, double eps_size, double eps_start, double* __restrict__ part_t,
    int32_t* __restrict__ part_i, int64_t part_stride, const int32_t* __restrict__ catagory,
    int32_t* __restrict__ rec_tri, double* __restrict__ rec_t, uint8_t* __restrict__ rec_cls,
    int32_t* __restrict__ blockcnt, int32_t* __restrict__ hist, const int base,
    const int qwave) {
  constexpr int RW = 64 * R;      // rays per wave
  constexpr int GT = 256;   // cluster spheres per LDS tile (128: 0.868 ms per step against 0.823)
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  // Coherent-ray traces (R == 1 only): this wave does wavefront `qwave` -- one that
  // k_intersect_beam has left to this kernel -- i.e. rays 64 qwave ... 64 qwave + 63.  qwave < 0:
  // nothing; the wave still walks the tiles with its block (the staging barriers), without tests.
  constexpr bool ordered = ORD;
  const bool idle_wave = ordered && qwave < 0;
  if (!ordered && base >= n) return;  // block-uniform
  // ray of slot r * 64 + lane of this wave (-1: none)
  auto ray_index = [&](const int r) -> int {
    const int qq = ordered ? qwave * 64 + lane : base + r * BLOCK + tid;
    return (!idle_wave && qq < n) ? qq : -1;
  };
  const int c_lo = blockIdx.y * chunk_clusters;
  const int c_hi = min(n_clusters, c_lo + chunk_clusters);

  // cluster spheres of the current tile; one float4 of padding after every supercluster's 8 so
  // that lanes working on different superclusters read from different LDS banks
  __shared__ float4 tile[GT + GT / SUPER];
  // candidate list of the wave: (tile-local cluster << 8 | ray slot), 16 bits because it is
  // always drained before the tile changes (LDS footprint decides the waves in flight)
  constexpr int LIST_CAP = 640;   // > 64 * SUPER: one batch of 64 pairs must fit
  static_assert(LIST_CAP > 64 * SUPER, "candidate list too small for one batch");
  __shared__ uint16_t clist[WAVES][LIST_CAP];
  // (tile-local supercluster << 8 | ray slot) pairs waiting for their cluster tests
  __shared__ uint16_t rlist[WAVES][128];
  // (a, -s.a) and (b, -s.b) of the wave's rays, interleaved: one address serves both reads
  __shared__ float4 prep_ab[WAVES][2 * RW];
  __shared__ unsigned long long best_k[WAVES][RW];
  __shared__ int32_t best_i[WAVES][RW];
  constexpr int MU = 4;                     // 16-lane groups' candidates in flight per step
  constexpr int PAIRS = 64 + MU * 64;       // waiting pairs: < 64 left over + one step's hits
  __shared__ uint32_t pairs[WAVES][PAIRS];   // member slot << 8 | ray slot (member slot < 2^24)
  __shared__ uint8_t x_slot[WAVES][128];     // screen survivors waiting for the float64 test
  __shared__ int32_t x_face[WAVES][128];
  // The wave's rays for the screen / decision stages come from the lane that owns them (one ray
  // per lane) or are re-read from the ray block (R > 1); a copy in LDS, 6-12 KB, cost the fifth
  // workgroup per CU.  (6 waves per SIMD spill: 276 us against 230.)
  auto ray_at = [&](int slot) -> int64_t {  // (R > 1: natural order only)
    const int i = base + (slot >> 6) * BLOCK + wave * 64 + (slot & 63);
    return i < n ? i : 0;
  };
#define TFRT_RAYV(q, slot) ldd(rays, (int64_t)(q) * stride + ray_at(slot))
  __shared__ int32_t skip_l[WAVES][RW];      // face each ray starts on (-1: none)

  const double cx = c0[0], cy = c0[1], cz = c0[2];
  const float cxf = (float)cx, cyf = (float)cy, czf = (float)cz;  // exact: k_center rounds c0
  // (s - c0) in float32 of a start coordinate read from the ray state
  auto rel_c0 = [](const double v, const double c, const float cf) -> float {
    if constexpr (sizeof(T) <= 4) return (float)v - cf;  // v is a float32 / float16 value
    else return (float)(v - c);
  };
  float ax[R], ay[R], az[R], bx[R], by[R], bz[R], nsa[R], nsb[R];
  // float32 / float16 ray state, one ray per lane: every lane keeps its own ray in registers and
  // the screen / decision stages fetch a pair's ray from the lane that owns it (ds_bpermute)
  // instead of gathering six coordinates from the ray block
  constexpr bool RAY_SHUFFLE = (R == 1);
  using RT = std::conditional_t<sizeof(T) <= 4, float, double>;  // (exact for every ray state)
  RT own_ray[6] = {0, 0, 0, 0, 0, 0};
  if constexpr (RAY_SHUFFLE) {
    const int i0 = ray_index(0);
    const int64_t ii = i0 >= 0 ? i0 : 0;
#pragma unroll
    for (int q = 0; q < 6; ++q) own_ray[q] = static_cast<RT>(rays[q * stride + ii]);
  }
  // (all lanes active: a lane that is masked off would hand out zeros)
  auto ray_of = [&](const int slot, double s[3], double e[3]) {
    if constexpr (RAY_SHUFFLE) {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        s[q] = static_cast<double>(__shfl(own_ray[q], slot, 64));
        e[q] = static_cast<double>(__shfl(own_ray[3 + q], slot, 64));
      }
    } else {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        s[q] = TFRT_RAYV(q, slot);
        e[q] = TFRT_RAYV(3 + q, slot);
      }
    }
  };
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = ray_index(r);
    ax[r] = ay[r] = az[r] = bx[r] = by[r] = bz[r] = 0.f;
    // NaN, not +inf: a lane without a ray must never pass `d2 <= r2`, even for a face whose
    // inflated radius overflowed to +inf (size_eps = 1e300 does that); an inf here let ray-less
    // lanes queue candidates and read ray slots >= n
    nsa[r] = nsb[r] = __builtin_nanf("");
    {
      const int slot = r * 64 + lane;
      const int64_t ii = i >= 0 ? i : 0;
      skip_l[wave][slot] = (last_tri != nullptr && i >= 0) ? last_tri[ii] : -1;
    }
    if (i >= 0 && prep != nullptr && !ordered) {
      ax[r] = prep[i];
      ay[r] = prep[pstride + i];
      az[r] = prep[2 * pstride + i];
      bx[r] = prep[3 * pstride + i];
      by[r] = prep[4 * pstride + i];
      bz[r] = prep[5 * pstride + i];
      nsa[r] = prep[6 * pstride + i];
      nsb[r] = prep[7 * pstride + i];
    } else if (i >= 0) {
      // first pass of a trace: nobody has written the filter state of these rays yet.  Forming
      // it here (k_rayprep's arithmetic, ~60 float64 instructions) saves that kernel's launch,
      // its 32 B per ray written and the same 32 B read back
      // (coherent-ray traces: always -- k_react3d does not write a filter state that hardly any
      // wavefront would read)
      double s[3], e[3], u[3], scv[3];
      load_ray3(rays, stride, i, s, e);
      float o[8];
      ray_filter_state(s, e, c0, o, u, scv);
      ax[r] = o[0];
      ay[r] = o[1];
      az[r] = o[2];
      bx[r] = o[3];
      by[r] = o[4];
      bz[r] = o[5];
      nsa[r] = o[6];
      nsb[r] = o[7];
    }
    const int slot = r * 64 + lane;
    prep_ab[wave][2 * slot] = make_float4(ax[r], ay[r], az[r], nsa[r]);
    prep_ab[wave][2 * slot + 1] = make_float4(bx[r], by[r], bz[r], nsb[r]);
    best_k[wave][slot] = dkey(INFINITY);
    best_i[wave][slot] = -1;
  }

  int ln = 0;          // entries in clist (wave-uniform)
  int rn = 0;          // entries in rlist (wave-uniform)
  int t0 = c_lo - GT;  // first cluster of the current tile (c_lo is a multiple of SUPER)

  // Screen: one (ray, member) pair per lane against the face's float32 record (3 gathered
  // 16-byte loads; the ray comes from LDS).  Survivors -- about a third -- are appended to the
  // exact list so that the expensive float64 stage always runs on full wavefronts.
  const float es_f = (float)eps_size, er_f = (float)eps_start;
  int xn = 0;  // entries waiting in x_slot / x_face (wave-uniform)
  int pn = 0;  // entries waiting in pairs[] (wave-uniform)
  bool draining = false, drained = false;  // after the last tile: one final flush empties both
  auto screen = [&](const int nb) {
    bool keep = false;
    int j = -1;
    const uint32_t pr = lane < nb ? pairs[wave][lane] : 0u;
    const int slot = (int)(pr & 255u);
    double s[3], e[3];
    ray_of(slot, s, e);
    if (lane < nb) {
      const int memb = (int)(pr >> 8);
      float4 r0 = crec[3 * (int64_t)memb], r1 = crec[3 * (int64_t)memb + 1],
             r2 = crec[3 * (int64_t)memb + 2];
      // All three gathers are in flight before anything is decided: left to itself the compiler
      // sinks the record's loads behind the face-index and skip tests and the first early-out of
      // the screen -- three dependent round trips per batch instead of one.
      __asm__ volatile(""
                       : "+v"(r0.x), "+v"(r0.y), "+v"(r0.z), "+v"(r0.w), "+v"(r1.x), "+v"(r1.y),
                         "+v"(r1.z), "+v"(r2.x), "+v"(r2.y), "+v"(r2.z));
      j = __float_as_int(r0.w);
      const double best = dkey_inv(best_k[wave][slot]);
      // t = (s - c0) - (P0 - c0): both operands are float32 roundings, so the error of t scales
      // with their magnitudes, not with |t|
      const float sx = rel_c0(s[0], cx, cxf), sy = rel_c0(s[1], cy, cyf), sz = rel_c0(s[2], cz, czf);
      const float tx = sx - r0.x, ty = sy - r0.y, tz = sz - r0.z;
      const float nt_err = fabsf(tx) + fabsf(ty) + fabsf(tz) + fabsf(sx) + fabsf(sy) + fabsf(sz) +
                           fabsf(r0.x) + fabsf(r0.y) + fabsf(r0.z);
      keep = j >= 0 && j != skip_l[wave][slot] &&
             may_hit_core((float)(e[0] - s[0]), (float)(e[1] - s[1]), (float)(e[2] - s[2]), r1.x,
                          r1.y, r1.z, r2.x, r2.y, r2.z, tx, ty, tz, nt_err, es_f, er_f, best);
    }
    const unsigned long long km = __ballot(keep);
    if (keep) {
      const int pos = xn + rank_below(km);
      x_slot[wave][pos] = (uint8_t)slot;
      x_face[wave][pos] = j;
    }
    xn += __popcll(km);
    TFRT_STAT(4, __popcll(km));
    wave_fence();
  };

  // Decision: one (ray, face) per lane, exact float64 test; nearest hit per ray kept in LDS
  // (64-bit min on an order-preserving key of ray_u, ties to the lower face index).
  auto decide = [&](const int nb) {
    bool have = false;
    unsigned long long key = 0, old = 0;
    int j = -1;
    const int slot = lane < nb ? (int)x_slot[wave][lane] : 0;
    double s[3], e[3];
    ray_of(slot, s, e);
    if (lane < nb) {
      j = x_face[wave][lane];
      old = best_k[wave][slot];
      double P[9];
      const double* fp = fverts + 9 * (int64_t)j;
#pragma unroll
      for (int q = 0; q < 9; ++q) P[q] = fp[q];
      const TriHit h = exact_triangle(s, e, P, eps_int, eps_size, eps_start);
      if (h.valid) {
        have = true;
        key = dkey(h.ray_u);
      }
    }
    TFRT_STAT(5, __popcll(__ballot(have)));
    wave_fence();
    if (have) atomicMin(&best_k[wave][slot], key);
    wave_fence();
    bool win = false;
    if (have) {
      const unsigned long long now = best_k[wave][slot];
      win = key == now;
      if (win && now < old) best_i[wave][slot] = 0x7FFFFFFF;  // a nearer hit: restart the tie-break
    }
    wave_fence();
    if (win) atomicMin(&best_i[wave][slot], j);
    wave_fence();
  };

  // Drain the wave's candidate list.
  auto flush = [&]() {
    const uint16_t* list = &clist[wave][0];
    const int total = ln;
    // 2. member tests, 16 queued candidates per step: each 16-lane group takes four; their
    //    member spheres are fetched first (four independent coalesced 256-byte reads in
    //    flight).  Member hits become (ray, face) pairs; while 64 are waiting -- and once more
    //    at the end -- they are decided, one per lane.  (One call site each for decide() and
    //    flush(): the float64 test is big and copies of it only bloat the kernel.)
    // (pairs and screen survivors carry (member, ray) / (face, ray): they outlive the tile, so
    // partial batches wait for the next flush and only the last one -- `draining`, after the
    // last tile -- empties them.  Draining at every flush ran the screen and the ~350-instruction
    // float64 decision on a mostly empty wavefront three or four extra times per wave.)
    const int nsteps = (total + 4 * MU - 1) / (4 * MU);
    // Whole steps only: the list is padded with a harmless entry -- the tile's first cluster
    // against ray slot 0; one more candidate is always allowed -- so that the loads below need
    // no bounds test (compare, exec juggling and default values were a fifth of a step).
    static_assert(LIST_CAP % (4 * MU) == 0 && 4 * MU <= 64, "padding stays inside the list");
    if (total + lane < nsteps * 4 * MU) clist[wave][total + lane] = 0;
    wave_fence();
    for (int st = 0; st < nsteps + (draining ? 1 : 0); ++st) {
      const int q0 = st * 4 * MU;
      if (st < nsteps) {
        // 4 * MU candidates per step: 16 / ML lanes take one (ray, cluster) and ML members each
        // (the list entry, the ray's filter state and the addresses are shared by ML tests;
        // measured at 1M rays: ML = 1 -> 0.918 ms per optimiser step, 2 -> 0.897, 4 -> 0.913)
        constexpr int ML = 2;
        static_assert(MU % ML == 0 && CLUSTER % ML == 0, "whole candidates per step");
        constexpr int MH = MU / ML;            // rounds per step
        constexpr int LPC = CLUSTER / ML;      // lanes per candidate
        float4 sp[MH][ML];
        int slot[MH];
        unsigned memb[MH];
#pragma unroll
        for (int u = 0; u < MH; ++u) {
          const int q = q0 + (64 / LPC) * u + (lane / LPC);
          const unsigned v = list[q];
          slot[u] = (int)(v & 255u);
          memb[u] = ((unsigned)t0 + (v >> 8)) * CLUSTER + (unsigned)ML * (unsigned)(lane % LPC);
#pragma unroll
          for (int h = 0; h < ML; ++h) sp[u][h] = csphere[memb[u] + h];
        }
#pragma unroll
        for (int u = 0; u < MH; ++u) {
          const float4 fa = prep_ab[wave][2 * slot[u]], fb = prep_ab[wave][2 * slot[u] + 1];
#pragma unroll
          for (int h = 0; h < ML; ++h) {
            const float4 m = sp[u][h];
            const float pa = fmaf(m.x, fa.x, fmaf(m.y, fa.y, fmaf(m.z, fa.z, fa.w)));
            const float pb = fmaf(m.x, fb.x, fmaf(m.y, fb.y, fmaf(m.z, fb.z, fb.w)));
            const bool hit = fmaf(pa, pa, pb * pb) <= m.w;
            const unsigned long long hm = __ballot(hit);
            if (hit)
              pairs[wave][pn + rank_below(hm)] = ((memb[u] + (unsigned)h) << 8) | (uint32_t)slot[u];
            pn += __popcll(hm);
            TFRT_STAT(3, __popcll(hm));
          }
        }
      }
      const bool last = draining && st == nsteps;
      while (pn >= 64 || (last && (pn > 0 || xn > 0))) {
        const int nb = min(pn, 64);
        wave_fence();
        if (nb > 0) screen(nb);
        // keep the rest of the pairs: move them to the front (64 per round)
        for (int m0 = 0; m0 < pn - nb; m0 += 64) {
          uint32_t tp = 0;
          if (m0 + lane < pn - nb) tp = pairs[wave][nb + m0 + lane];
          wave_fence();
          if (m0 + lane < pn - nb) pairs[wave][m0 + lane] = tp;
          wave_fence();
        }
        pn -= nb;
        while (xn >= 64 || (last && pn == 0 && xn > 0)) {
          const int xb = min(xn, 64);
          decide(xb);
          int ts = 0, tf = 0;  // fewer than 64 can remain
          if (lane < xn - xb) {
            ts = x_slot[wave][xb + lane];
            tf = x_face[wave][xb + lane];
          }
          wave_fence();
          if (lane < xn - xb) {
            x_slot[wave][lane] = (uint8_t)ts;
            x_face[wave][lane] = tf;
          }
          xn -= xb;
          wave_fence();
        }
      }
    }
    ln = 0;
  };

  static_assert(GT / SUPER <= 32 && GT % SUPER == 0, "one 32-bit supercluster mask per ray and tile");
  static_assert(SUPER == 8, "8-lane groups test the 8 clusters of a supercluster");
  const float4 never = make_float4(0.f, 0.f, 0.f, -1.f);
  unsigned touched[R];  // per ray: superclusters of the current tile its line touches
#pragma unroll
  for (int r = 0; r < R; ++r) touched[r] = 0u;
  // One wave-uniform action per iteration (so flush() and the batch below have one call site):
  //   flush   when the candidate list is nearly full, or the tile is finished and it is not empty
  //   batch   8 lanes per waiting (ray, supercluster) pair test its 8 cluster spheres, one each
  //           (LDS tile); touched clusters are appended to the candidate list (ballot + rank)
  //   round   every lane with a touched supercluster left moves one to the pair list
  //   tile    stage the next tile and run level 0 on it (all four waves meet here), or finish
  for (;;) {
    bool pending = false;
#pragma unroll
    for (int r = 0; r < R; ++r) pending = pending || __any(touched[r] != 0u);
    if (ln > LIST_CAP - 64 * SUPER || (!pending && rn == 0 && ln > 0) ||
        (draining && !drained && !pending && rn == 0)) {
      flush();
      drained = draining;
    } else if (rn >= 64 || (!pending && rn > 0)) {
      const int nb = min(rn, 64);
      // One waiting (ray, supercluster) pair per lane: its 8 cluster spheres in turn, hits
      // collected in a per-lane mask and then moved to the candidate list one per lane and
      // round (ballot + rank).  (8 lanes per pair with one append per step cost ~345
      // instructions per 64 pairs, this ~160.)
      unsigned hits = 0u;
      const int v = lane < nb ? (int)rlist[wave][lane] : 0;
      const int cl0 = (v >> 8) * SUPER, sl = v & 255;
      double l1s[3] = {0.0, 0.0, 0.0};
      // ... and how far along its line this pair's ray has a hit already (inf: none).  A cluster
      // whose sphere lies wholly beyond that cannot hold the nearest hit, nor a tie with it: rays
      // that run along a long wall (a light guide: 48 clusters queued per ray and pass, against 5
      // on a lens) lose most of their queue here once the first batches have been decided.
      float reach = INFINITY;
      if (last_tri != nullptr && eps_start >= 0.0) {  // (wave-uniform: only the behind test needs it)
        double l1e[3];
        ray_of(sl, l1s, l1e);
        const double bu = dkey_inv(best_k[wave][sl]);
        if (bu < 1.0e300) {
          const double dx = l1e[0] - l1s[0], dy = l1e[1] - l1s[1], dz = l1e[2] - l1s[2];
          // (rounded up generously: 1e-4 relative dwarfs the float32 rounding of everything here)
          reach = (float)(bu * sqrt(dx * dx + dy * dy + dz * dz)) * 1.0001f;
        }
      }
      if (lane < nb) {
        const float4 fa = prep_ab[wave][2 * sl], fb = prep_ab[wave][2 * sl + 1];
        const float4* row = &tile[cl0 + (cl0 >> 3)];
        // The sphere tests are line tests; a cluster wholly behind the ray's start cannot hold
        // a valid hit (ray_u > 0) either.  Rays that leave a surface see half of the clusters
        // around them that way, and rays that graze a long wall (a light guide) many more.
        // u = a x b is the ray direction; t = (c - s).u + (error bound of the start's part)
        // must be below -r.  The centre's part of the rounding error (< 10 * 2^-24 |c|) is
        // inside the 64 * 2^-24 (|c| + r) the spheres are inflated by.
        const float ux = fa.y * fb.z - fa.z * fb.y, uy = fa.z * fb.x - fa.x * fb.z,
                    uz = fa.x * fb.y - fa.y * fb.x;
        const float sx = rel_c0(l1s[0], cx, cxf), sy = rel_c0(l1s[1], cy, cyf),
                    sz = rel_c0(l1s[2], cz, czf);
        const float t_err = (32.f * 5.9604644775390625e-08f) * (fabsf(sx) + fabsf(sy) + fabsf(sz));
        const float t_off = fmaf(-sx, ux, fmaf(-sy, uy, -sz * uz)) + t_err;
        // (a LOWER bound of a centre's coordinate along the ray is tq - 2 t_err; |u| = 1 to 1e-6:
        // inside reach's 1e-4)
        const float far0 = reach + 2.f * t_err;
        // (first pass of a trace -- no ray starts on a face yet: sources normally sit outside
        // the scene, nothing lies behind them and the test would only cost; wave-uniform)
        // (and only while hits must lie ahead of the start: ray_start_epsilion >= 0)
        if (last_tri != nullptr && eps_start >= 0.0) {
#pragma unroll
          for (int c = 0; c < SUPER; ++c) {
            const float4 sp = row[c];
            const float pa = fmaf(sp.x, fa.x, fmaf(sp.y, fa.y, fmaf(sp.z, fa.z, fa.w)));
            const float pb = fmaf(sp.x, fb.x, fmaf(sp.y, fb.y, fmaf(sp.z, fb.z, fb.w)));
            const float tq = fmaf(sp.x, ux, fmaf(sp.y, uy, fmaf(sp.z, uz, t_off)));
            const bool behind = tq < 0.f && tq * tq > sp.w;
            const float over = tq - far0;      // (reach = inf: -inf, never beyond; NaN: never)
            const bool beyond = over > 0.f && over * over > sp.w;
            hits |= ((fmaf(pa, pa, pb * pb) <= sp.w && !behind && !beyond) ? 1u : 0u) << c;
          }
        } else {
#pragma unroll
          for (int c = 0; c < SUPER; ++c) {
            const float4 sp = row[c];
            const float pa = fmaf(sp.x, fa.x, fmaf(sp.y, fa.y, fmaf(sp.z, fa.z, fa.w)));
            const float pb = fmaf(sp.x, fb.x, fmaf(sp.y, fb.y, fmaf(sp.z, fb.z, fb.w)));
            hits |= (fmaf(pa, pa, pb * pb) <= sp.w ? 1u : 0u) << c;
          }
        }
      }
      for (;;) {
        const bool has = hits != 0u;
        const unsigned long long hm = __ballot(has);
        if (hm == 0ull) break;
        if (has) {
          const int k = __ffs(hits) - 1;
          hits &= hits - 1u;
          clist[wave][ln + rank_below(hm)] = (uint16_t)(((cl0 + k) << 8) | sl);
        }
        ln += __popcll(hm);
        TFRT_STAT(2, __popcll(hm));
      }
      // keep the rest of the pairs (fewer than 64) at the front
      int keep = 0;
      if (lane < rn - nb) keep = rlist[wave][nb + lane];
      wave_fence();
      if (lane < rn - nb) rlist[wave][lane] = (uint16_t)keep;
      rn -= nb;
      wave_fence();
    } else if (pending) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const bool has = touched[r] != 0u;
        const unsigned long long m = __ballot(has);
        if (m != 0ull && rn < 64) {  // (rn < 64: room for one more round of up to 64 pairs)
          if (has) {
            const int k = __ffs(touched[r]) - 1;
            touched[r] &= touched[r] - 1u;
            rlist[wave][rn + rank_below(m)] = (uint16_t)((k << 8) | (r * 64 + lane));
          }
          rn += __popcll(m);
          TFRT_STAT(1, __popcll(m));
        }
      }
      wave_fence();
    } else {
      if (draining) break;  // (the final flush has run)
      t0 += GT;
      if (t0 >= c_hi) {
        draining = true;
        continue;
      }
      const int nt = min(GT, c_hi - t0);
      const int ns = (nt + SUPER - 1) / SUPER;
      __syncthreads();
      for (int k = tid; k < ns * SUPER; k += BLOCK)
        tile[k + (k >> 3)] = (k < nt) ? clsphere[t0 + k] : never;
      __syncthreads();
      // level 0: which superclusters of the tile does each ray's line touch.  The supercluster
      // spheres are wave-uniform data: read straight from global memory with a uniform index
      // (scalar loads into SGPRs, four spheres per round) -- no LDS staging, no LDS reads
      TFRT_STAT(0, (long long)ns * __popcll(__ballot(ray_index(0) >= 0)));
      const float4* __restrict__ su = susphere + t0 / SUPER;
      // (a "wholly behind the ray's start" test here, as at level 1, was measured and does not
      // pay: +4 instructions on each of the 83 supercluster tests for ~2 pairs saved per ray)
      // (descending, the result of each test shifted in at bit 0: compare + add-with-carry, two
      // instructions where `touched |= bit << k` took four; bit k still stands for supercluster k)
      auto level0 = [&](const float4 sp) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const float pa = fmaf(sp.x, ax[r], fmaf(sp.y, ay[r], fmaf(sp.z, az[r], nsa[r])));
          const float pb = fmaf(sp.x, bx[r], fmaf(sp.y, by[r], fmaf(sp.z, bz[r], nsb[r])));
          shift_in_le(touched[r], fmaf(pa, pa, pb * pb), sp.w);
        }
      };
      int k = idle_wave ? -1 : ns - 1;   // (wave-uniform)
      for (; k >= 3; k -= 4) {  // four scalar loads in flight
        const float4 s0 = su[k], s1 = su[k - 1], s2 = su[k - 2], s3 = su[k - 3];
        level0(s0);
        level0(s1);
        level0(s2);
        level0(s3);
      }
      for (; k >= 0; --k) level0(su[k]);
    }
  }

  if (rec_cls == nullptr) {  // several cluster chunks: k_classify3d merges the partial results
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = ray_index(r);
      if (i >= 0) {
        const int slot = r * 64 + lane;
        part_t[blockIdx.y * part_stride + i] = dkey_inv(best_k[wave][slot]);
        part_i[blockIdx.y * part_stride + i] = best_i[wave][slot];
      }
    }
    return;
  }
  // One cluster chunk: this workgroup has seen every face, so the nearest hit is final.  Do
  // k_classify3d's work here (hit record, class, per-256-ray class histogram for the scan).
  __shared__ int wc[WAVES][4];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = ray_index(r);
    int cls = -1;
    if (i >= 0) {
      const int slot = r * 64 + lane;
      const int bi = best_i[wave][slot];
      cls = (bi < 0) ? CLS_DEAD : cat_to_cls(catagory[bi]);
      rec_tri[i] = bi;
      rec_t[i] = dkey_inv(best_k[wave][slot]);
      rec_cls[i] = (uint8_t)cls;
    }
    if (ordered) {  // (R == 1) this wavefront's share of its 256-ray block's class histogram
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int cnt_c = __popcll(__ballot(cls == c));
        if (lane == c && cnt_c > 0 && qwave >= 0) atomicAdd(&hist[(qwave >> 2) * 4 + c], cnt_c);
      }
      break;  // (block-uniform)
    }
    if (base + r * BLOCK >= n) break;  // block-uniform: no rays in this 256-ray slice
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const unsigned long long m = __ballot(cls == c);
      if (lane == 0) wc[wave][c] = __popcll(m);
    }
    __syncthreads();
    if (tid < 4) {
      int sum = 0;
      for (int w = 0; w < WAVES; ++w) sum += wc[w][tid];
      blockcnt[(blockIdx.x * R + r) * 4 + tid] = sum;
    }
    __syncthreads();
  }
}

template <typename T>
__global__ __launch_bounds__(BLOCK) TFRT_GROUP_ATTR void k_intersect_group(
    const T* __restrict__ rays, int64_t stride, const int32_t* __restrict__ n_ptr,
    const int32_t* __restrict__ last_tri, const float4* __restrict__ susphere,
    const float4* __restrict__ clsphere, const float4* __restrict__ csphere,
    const float4* __restrict__ crec, const int32_t* __restrict__ cface,
    const double* __restrict__ fverts, const double* __restrict__ c0,
    const float* __restrict__ prep, int64_t pstride, int n_clusters, int chunk_clusters,
    double eps_int, double eps_size, double eps_start, double* __restrict__ part_t,
    int32_t* __restrict__ part_i, int64_t part_stride, const int32_t* __restrict__ catagory,
    int32_t* __restrict__ rec_tri, double* __restrict__ rec_t, uint8_t* __restrict__ rec_cls,
    int32_t* __restrict__ blockcnt) {
  // one workgroup per 256 rays
  group_walk<T, 1, false>(rays, stride, *n_ptr, last_tri, susphere, clsphere, csphere, crec, cface,
                          fverts, c0, prep, pstride, n_clusters, chunk_clusters, eps_int, eps_size,
                          eps_start, part_t, part_i, part_stride, catagory, rec_tri, rec_t, rec_cls,
                          blockcnt, nullptr, (int)blockIdx.x * BLOCK, -1);
}

// Coherent-ray traces: the wavefronts k_intersect_beam has left (usually none: the workgroups read
// one counter and retire), four per workgroup and round.  A kernel of its own: the loop and the
// list arguments cost the natural-order kernel 25 vector and 78 scalar register spills at its five
// waves per SIMD when both lived in one body.
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_intersect_group_left(
    const T* __restrict__ rays, int64_t stride, const int32_t* __restrict__ n_ptr,
    const int32_t* __restrict__ last_tri, const float4* __restrict__ susphere,
    const float4* __restrict__ clsphere, const float4* __restrict__ csphere,
    const float4* __restrict__ crec, const int32_t* __restrict__ cface,
    const double* __restrict__ fverts, const double* __restrict__ c0, int n_clusters,
    int chunk_clusters, double eps_int, double eps_size, double eps_start,
    const int32_t* __restrict__ catagory, int32_t* __restrict__ rec_tri,
    double* __restrict__ rec_t, uint8_t* __restrict__ rec_cls,
    const int32_t* __restrict__ left_list, const int32_t* __restrict__ left_count,
    int32_t* __restrict__ hist) {
  const int n = *n_ptr;
  const int total = *left_count;
  for (int item = blockIdx.x; item * WAVES < total; item += gridDim.x) {
    const int k = item * WAVES + (int)(threadIdx.x >> 6);
    const int qwave = k < total ? left_list[k] : -1;
    group_walk<T, 1, true>(rays, stride, n, last_tri, susphere, clsphere, csphere, crec, cface,
                           fverts, c0, nullptr, 0, n_clusters, chunk_clusters, eps_int, eps_size,
                           eps_start, nullptr, nullptr, 0, catagory, rec_tri, rec_t, rec_cls,
                           nullptr, hist, 0, qwave);
    __syncthreads();  // (the next round reuses the LDS tile and lists)
  }
}

// ------------------------------------------------------------- coherent-wave intersect
//
// k_intersect_beam: the same decisions as k_intersect_group for wavefronts whose 64 rays form a
// narrow bundle -- which they do when the caller hands the source rays over in a coherent order
// (tfrt_scene3d.coherent_rays: neighbouring rays have neighbouring lines, e.g. sorted along a
// Hilbert curve through their aperture points; the stable compaction keeps the children of
// neighbouring rays neighbours in every later pass).  Then a wave shares ONE walk of the hierarchy:
//
//   bundle     axis (o, w) = mean start / mean direction of the wave's rays; every point of every
//              ray's line at axial coordinate t lies within R0 + S |t| of the axis (R0, S: wave
//              maxima of the lines' offsets and slopes).  A sphere (c, r) can contain a point of
//              some line only if dist(c, axis) <= r + R0 + S (|t_c| + r)   [triangle inequality]
//              -- and, from the second pass on, only if t_c + r >= min start coordinate (valid
//              hits lie ahead of the starts).
//   levels     LANE = NODE: 64 supercluster / cluster / member spheres per instruction against the
//              bundle (no per-ray work at all), survivors compacted into the next level's list.
//   faces      LANE = FACE: the candidate faces (a dozen for a coherent wave) as TRIANGLES, seen
//              along the axis, against the bundle (face_frame below); those left are ordered by
//              depth and taken two at a time, LANE = RAY: the faces' records travel in scalar
//              registers (v_readlane), a ray keeps a face if its image lies within the face's
//              (a dozen instructions); the pairs kept queue for the exact float64 decision, one
//              (ray, face) per lane, nearest hit per ray by 64-bit ds_min -- the same function
//              as in the grouped kernel.  The walk ends when every ray has a decided hit nearer
//              than every face left.
//
// Every stage only removes pairs that cannot win (the bundle bounds are inflated well beyond
// their float32 rounding), so results are bit-identical to k_intersect_group / k_intersect3d.
// A wave whose rays do NOT form a narrow bundle (directions spread, or more than BEAM_CLIST
// clusters / BEAM_FLIST faces touched) gives up early and puts itself on the list of wavefronts
// that the grouped kernel -- launched behind this one -- does.
constexpr int BEAM_SLIST = 64;    // touched superclusters a narrow bundle may have
constexpr int BEAM_CLIST = 32;    // ... clusters
constexpr int BEAM_FLIST = 192;   // ... candidate faces
constexpr int BEAM_WIDE = 48;     // candidate faces beyond which a bundle of more than 8 rays is cut

struct Beam {  // wave-uniform
  float ox, oy, oz, wx, wy, wz, R0, S, tmin;      // axis, bounds (see k_intersect_beam)
  float e1x, e1y, e1z, e2x, e2y, e2z;             // image plane (see face_frame)
  float R0b, tb;                                  // the bundle's radius in a second plane, t = tb
  // radius bound of the bundle at axial coordinate t (+ pad on |t|): the smaller of the two
  __device__ __forceinline__ float radius(float t, float pad) const {
    return fminf(R0 + S * (fabsf(t) + pad), R0b + S * (fabsf(t - tb) + pad));
  }
};

__device__ __forceinline__ bool beam_touch(const Beam& b, const float4 sp) {
  // radius from the stored (inflated) r^2, rounded up; padding entries have w < 0: NaN, never true
  // (v_sqrt_f32 / v_rcp_f32 / v_rsq_f32 as they are, 1 ulp: every bound of this kernel is inflated
  // far beyond that, and the IEEE forms cost ten instructions each)
  const float r = __builtin_amdgcn_sqrtf(sp.w) * 1.000002f;
  const float vx = sp.x - b.ox, vy = sp.y - b.oy, vz = sp.z - b.oz;
  const float t = vx * b.wx + vy * b.wy + vz * b.wz;
  const float v2 = vx * vx + vy * vy + vz * vz;
  const float B = r + b.radius(t, r);
  // (2e-6 v2: ten times the rounding error of v2 - t^2 and of a not exactly unit w)
  return (v2 - t * t <= B * B + 2e-6f * v2) && !(t + r < b.tmin);
}

// The faces themselves against the bundle and its rays, seen ALONG THE AXIS: points are mapped
// to (x . e1, x . e2) with e1, e2 (about) orthonormal and perpendicular to w -- a linear map, so
// a hit X of a ray on a face lies in the face's image, and on the ray's image
//     p(t) = A + M t,   t = (X - o) . w,   M = (d . e1, d . e2) / (d . w),   A = p(start) - M t_start
// (exact identities for ANY e1, e2, w).  A valid hit lies within `slack` of the triangle (trig_u,
// trig_v >= -eps_size, trig_u + trig_v <= 1 + eps_size put X = P0 + u E1 + v E2 at most
// 5 eps_size max(|E1|, |E2|) outside it), so t_X lies in the face's axial range [tlo, thi]
// (corner values widened by the slack) and p(t_X) within |M| (thi - tlo) / 2 of p(tc), tc the
// middle of the range.  With n_k, c_k the outward unit normals and offsets of the image
// triangle's edges (n_k . p + c_k = signed distance outside edge k):
//   LANE = FACE  face_frame(): the bundle can touch the face only if the axis' image (the origin)
//                is within rho = R0 + S max|t| of every edge's inner half-plane, c_k <= rho, and
//                the face does not lie behind every start;  it writes the face's record
//                (n_k, c_k, tlo, thi, tc, h, slack, face index) for
//   LANE = RAY   on_face(): n_k . p(tc) + c_k <= |M| h + slack for the three edges.
// Both are necessary conditions of a valid hit with every bound inflated far beyond its float32
// rounding (NaN: true); a triangle seen (almost) edge-on has no reliable orientation and passes
// every ray.  r0 = (P0 - c0 | face index), r1 = E1, r2 = E2: the float32 face record.
__device__ __forceinline__ bool face_frame(const Beam& b, const float4 r0, const float4 r1,
                                           const float4 r2, const float es, float4 rec[4]) {
  const float n1 = fabsf(r1.x) + fabsf(r1.y) + fabsf(r1.z);
  const float n2 = fabsf(r2.x) + fabsf(r2.y) + fabsf(r2.z);
  const float scale = fabsf(r0.x) + fabsf(r0.y) + fabsf(r0.z) + fabsf(b.ox) + fabsf(b.oy) +
                      fabsf(b.oz) + n1 + n2;
  const float slack = 5.05f * es * (n1 + n2) + 8e-6f * scale;
  const float ax = r0.x - b.ox, ay = r0.y - b.oy, az = r0.z - b.oz;
  const float t0 = ax * b.wx + ay * b.wy + az * b.wz;
  const float t1 = t0 + (r1.x * b.wx + r1.y * b.wy + r1.z * b.wz);
  const float t2 = t0 + (r2.x * b.wx + r2.y * b.wy + r2.z * b.wz);
  const float tlo = fminf(t0, fminf(t1, t2)) - slack, thi = fmaxf(t0, fmaxf(t1, t2)) + slack;
  // images of the corners: q0, q0 + g1, q0 + g2
  const float q0x = ax * b.e1x + ay * b.e1y + az * b.e1z, q0y = ax * b.e2x + ay * b.e2y + az * b.e2z;
  const float g1x = r1.x * b.e1x + r1.y * b.e1y + r1.z * b.e1z,
              g1y = r1.x * b.e2x + r1.y * b.e2y + r1.z * b.e2z;
  const float g2x = r2.x * b.e1x + r2.y * b.e1y + r2.z * b.e1z,
              g2y = r2.x * b.e2x + r2.y * b.e2y + r2.z * b.e2z;
  const float area = g1x * g2y - g1y * g2x;
  const float l1 = g1x * g1x + g1y * g1y, l2 = g2x * g2x + g2y * g2y;
  const bool flat = !(area * area > 1e-8f * l1 * l2);  // (edge-on, degenerate or NaN)
  const float sg = area > 0.f ? 1.f : -1.f;
  auto edge = [&](float px, float py, float dx, float dy, float* nx, float* ny, float* c) {
    const float k = sg * __builtin_amdgcn_rsqf(dx * dx + dy * dy);
    *nx = dy * k;
    *ny = -dx * k;
    *c = -(*nx * px + *ny * py);
  };
  float nx0, ny0, c0, nx1, ny1, c1, nx2, ny2, c2;
  edge(q0x, q0y, g1x, g1y, &nx0, &ny0, &c0);
  edge(q0x + g1x, q0y + g1y, g2x - g1x, g2y - g1y, &nx1, &ny1, &c1);
  edge(q0x + g2x, q0y + g2y, -g2x, -g2y, &nx2, &ny2, &c2);
  if (flat) {
    nx0 = ny0 = nx1 = ny1 = nx2 = ny2 = 0.f;
    c0 = c1 = c2 = -INFINITY;
  }
  const float slack2 = 1.5f * slack;  // (in the image: |e1|, |e2| = 1 to rounding, two components)
  rec[0] = make_float4(nx0, ny0, c0, tlo);
  rec[1] = make_float4(nx1, ny1, c1, thi);
  rec[2] = make_float4(nx2, ny2, c2, 0.5f * (tlo + thi));
  rec[3] = make_float4(0.5001f * (thi - tlo), slack2, r0.w, 0.f);
  const float rho = (fminf(b.R0 + b.S * fmaxf(fabsf(tlo), fabsf(thi)),
                           b.R0b + b.S * fmaxf(fabsf(tlo - b.tb), fabsf(thi - b.tb))) +
                     slack2) * 1.0001f;
  return !(c0 > rho) && !(c1 > rho) && !(c2 > rho) && !(thi < b.tmin);
}

// Wave-wide reductions with DPP (data-parallel primitives: the operand of a VALU instruction
// comes from another lane of the row / a neighbouring row): six vector instructions, no trip
// through the LDS crossbar (ds_bpermute costs an LDS round trip per step).  Every lane must hold
// a value (the caller substitutes the neutral element).  The result is wave-uniform.
template <typename Op>
__device__ __forceinline__ float wave_reduce_f(float v, Op op) {
  auto dpp = [](float x, auto ctrl, auto row_mask) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x),
                                                      decltype(ctrl)::value,
                                                      decltype(row_mask)::value, 0xF, false));
  };
  using std::integral_constant;
  v = op(v, dpp(v, integral_constant<int, 0xB1>{}, integral_constant<int, 0xF>{}));   // quad_perm [1,0,3,2]
  v = op(v, dpp(v, integral_constant<int, 0x4E>{}, integral_constant<int, 0xF>{}));   // quad_perm [2,3,0,1]
  v = op(v, dpp(v, integral_constant<int, 0x141>{}, integral_constant<int, 0xF>{}));  // row_half_mirror
  v = op(v, dpp(v, integral_constant<int, 0x140>{}, integral_constant<int, 0xF>{}));  // row_mirror
  // now every lane holds its row's result; fold the four rows into lane 63
  v = op(v, dpp(v, integral_constant<int, 0x142>{}, integral_constant<int, 0xA>{}));  // row_bcast:15 -> rows 1, 3
  v = op(v, dpp(v, integral_constant<int, 0x143>{}, integral_constant<int, 0xC>{}));  // row_bcast:31 -> rows 2, 3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_sum_f(float v) {
  return wave_reduce_f(v, [](float a, float b) { return a + b; });
}
__device__ __forceinline__ float wave_max_f(float v) {
  return wave_reduce_f(v, [](float a, float b) { return fmaxf(a, b); });
}
__device__ __forceinline__ float wave_min_f(float v) {
  return wave_reduce_f(v, [](float a, float b) { return fminf(a, b); });
}
// Several wave-wide reductions at once: one DPP instruction per value and step (the operand comes
// from another lane, the operation is part of the same instruction), the values taken in turn so
// that no instruction reads a register written by the one before it (a DPP read needs two idle
// slots after the write -- inside inline assembly nobody else keeps count of them).  Results are
// valid in lane 63; rows a step does not address keep their value.
#define TFRT_DPP_STEP4(OPA, OPB, OPC, OPD, CTRL)      \
  OPA " %0, %0, %0 " CTRL "\n\t" OPB " %1, %1, %1 " CTRL "\n\t" \
  OPC " %2, %2, %2 " CTRL "\n\t" OPD " %3, %3, %3 " CTRL "\n\t"
#define TFRT_DPP_ALL4(OPA, OPB, OPC, OPD)                                        \
  "s_nop 1\n\t"                                                                  \
  TFRT_DPP_STEP4(OPA, OPB, OPC, OPD, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") \
  TFRT_DPP_STEP4(OPA, OPB, OPC, OPD, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf") \
  TFRT_DPP_STEP4(OPA, OPB, OPC, OPD, "row_half_mirror row_mask:0xf bank_mask:0xf")     \
  TFRT_DPP_STEP4(OPA, OPB, OPC, OPD, "row_mirror row_mask:0xf bank_mask:0xf")          \
  TFRT_DPP_STEP4(OPA, OPB, OPC, OPD, "row_bcast:15 row_mask:0xa bank_mask:0xf")        \
  TFRT_DPP_STEP4(OPA, OPB, OPC, OPD, "row_bcast:31 row_mask:0xc bank_mask:0xf")
// four sums
__device__ __forceinline__ void wave_sum4(float& a, float& b, float& c, float& d) {
  __asm__ volatile(TFRT_DPP_ALL4("v_add_f32_dpp", "v_add_f32_dpp", "v_add_f32_dpp", "v_add_f32_dpp")
                   : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), 63));
  b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b), 63));
  c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c), 63));
  d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), 63));
}
// three maxima and a minimum
__device__ __forceinline__ void wave_max3_min(float& a, float& b, float& c, float& d) {
  __asm__ volatile(TFRT_DPP_ALL4("v_max_f32_dpp", "v_max_f32_dpp", "v_max_f32_dpp", "v_min_f32_dpp")
                   : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), 63));
  b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b), 63));
  c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c), 63));
  d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), 63));
}

__device__ __forceinline__ float uniform_f(float v) {  // v wave-uniform: into a scalar register
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}
__device__ __forceinline__ float bcast_f(float v, int src_lane) {  // src_lane wave-uniform
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}

// The exact float64 test of k_intersect_beam, one (ray, face) pair per lane; nearest hit per ray
// by 64-bit min on an order-preserving key of ray_u, ties to the lower face index (tf.argmin's
// first index).  x_pair[k] = face << 6 | ray slot, rt[6][64] = the wavefront's rays as stored.
// (A function of its own, not inlined: the four places that call it share one copy, and its
// registers -- nine float64 vertices, the six-term sums -- are not added to the caller's.)
template <typename U>
using LdsPtr = __attribute__((address_space(3))) U*;

template <typename RT>
__device__ __forceinline__ void beam_decide(const int nb, const int lane, LdsPtr<const uint32_t> x_pair,
                                         LdsPtr<const RT> rt, LdsPtr<unsigned long long> best_k,
                                         LdsPtr<int32_t> best_i, const double* __restrict__ fverts,
                                         const double eps_int, const double eps_size,
                                         const double eps_start) {
  bool have = false;
  unsigned long long key = 0, old = 0;
  int j = -1;
  const uint32_t xp = lane < nb ? x_pair[lane] : 0u;
  const int slot = (int)(xp & 63u);
  if (lane < nb) {
    double s[3], e[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      s[k] = static_cast<double>(rt[k * 64 + slot]);
      e[k] = static_cast<double>(rt[(3 + k) * 64 + slot]);
    }
    j = (int)(xp >> 6);
    old = best_k[slot];
    double P[9];
    const double* fp = fverts + 9 * (int64_t)j;
#pragma unroll
    for (int k = 0; k < 9; ++k) P[k] = fp[k];
    const TriHit h = exact_triangle(s, e, P, eps_int, eps_size, eps_start);
    if (h.valid) {
      have = true;
      key = dkey(h.ray_u);
    }
  }
  wave_fence();
  if (have) atomicMin((unsigned long long*)&best_k[slot], key);
  wave_fence();
  bool win = false;
  if (have) {
    const unsigned long long now = best_k[slot];
    win = key == now;
    if (win && now < old) best_i[slot] = 0x7FFFFFFF;  // a nearer hit: restart the tie-break
  }
  wave_fence();
  if (win) atomicMin((int32_t*)&best_i[slot], j);
  wave_fence();
}

// LDS of one wavefront of the beam walk (k_intersect_beam, k_trace_inplace)
template <typename RT>
struct BeamLds {
  uint16_t slist[BEAM_SLIST];
  uint16_t clist[BEAM_CLIST];
  uint32_t flist[BEAM_FLIST];
  uint32_t x_pair[192];  // face << 6 | lane of the ray (faces < 2^24)
  float4 ftab[64][4];    // records of a chunk's faces, nearest first (face_frame)
  RT rtab[6][64];        // the wave's rays as stored (the exact test reads them by slot)
  unsigned long long best_k[64];
  int32_t best_i[64];
};

// the scene as the beam walk reads it (device pointers of one trace)
struct BeamScene {
  const float4* susphere;
  const float4* clsphere;
  const float4* csphere;
  const float4* crec;
  const double* fverts;
  const double* c0;
  int n_clusters, n_super;
  double eps_int, eps_size, eps_start;
};

#ifdef TFRT_TICKS
#define TFRT_WI_ARG _wi
#else
#define TFRT_WI_ARG nullptr
#endif

// ONE pass of one wavefront: the rays wait in W.rtab (as stored), `live` lanes carry a ray of this
// pass, `skip` is the face a ray starts on (-1: none).  Leaves every live ray's nearest valid hit in
// W.best_k (order-preserving key of ray_u) / W.best_i (face, -1: none).  Returns false when the
// wavefront is no (few) narrow bundle(s) and the caller should leave it to the grouped kernel
// (never with coherent_only).
template <typename T, typename RT>
__device__ __forceinline__ bool beam_pass(BeamLds<RT>& W, const BeamScene& g, const int lane,
                                          const int bundle, const bool live, const int skip,
                                          const bool first_pass, const int coherent_only,
                                          unsigned* _wi, unsigned* work = nullptr) {
  // work (wave-uniform, optional): [0] += (ray, face) pairs that reach the exact float64 test,
  // [1] += candidate faces tested as triangles against a bundle (face_frame)
  TFRT_TICK_INIT;
  const float4* __restrict__ susphere = g.susphere;
  const float4* __restrict__ clsphere = g.clsphere;
  const float4* __restrict__ csphere = g.csphere;
  const float4* __restrict__ crec = g.crec;
  const double* __restrict__ fverts = g.fverts;
  const double* __restrict__ c0 = g.c0;
  const int n_clusters = g.n_clusters, n_super = g.n_super;
  const double eps_int = g.eps_int, eps_size = g.eps_size, eps_start = g.eps_start;
  const float4 never = make_float4(0.f, 0.f, 0.f, -1.f);
  W.best_k[lane] = dkey(INFINITY);
  W.best_i[lane] = -1;
  const double cx = c0[0], cy = c0[1], cz = c0[2];
  // (exact: c0 is rounded to float32)
  const float cxf = uniform_f((float)cx), cyf = uniform_f((float)cy), czf = uniform_f((float)cz);
  auto rel = [](const RT v, const double c, const float cf) -> float {
    if constexpr (sizeof(T) <= 4) return (float)v - cf;
    else return (float)((double)v - c);
  };
  const float es_f = (float)eps_size;
  int xn = 0;  // (ray, face) pairs waiting for the exact test (wave-uniform)

  // exact float64 test, one (ray, face) per lane (beam_decide)
  auto decide = [&](const int nb) {
    TFRT_STAT(28, 1);
    TFRT_WAVE_NOTE(1, 1);
    wave_fence();
    beam_decide<RT>(nb, lane, (LdsPtr<const uint32_t>)&W.x_pair[0],
                    (LdsPtr<const RT>)&W.rtab[0][0], (LdsPtr<unsigned long long>)&W.best_k[0],
                    (LdsPtr<int32_t>)&W.best_i[0], fverts, eps_int, eps_size, eps_start);
  };

  // The wavefront's lanes are taken as ONE bundle; if that bundle is not narrow (the ray order
  // jumps inside the wavefront: a Hilbert curve leaves and re-enters a round aperture) it is
  // cut where neighbouring lanes lie farthest apart, and
  // the parts are taken one after the other, down to single rays if need be (one ray is always a
  // narrow bundle).  `cuts` bit k: a bundle ends before lane k.  A wavefront whose rays are
  // simply not coherent -- directions spread, a quarter of the scene's clusters touched, or more
  // than BEAM_ATTEMPTS bundles tried -- is left to the grouped kernel instead: sixty-four
  // single-ray walks cost ten times its work.
  TFRT_STAT(8, 1);
  constexpr int BEAM_ATTEMPTS = 20;
  TFRT_TICK(0);
  float gap = -1.f;  // (formed when the first cut is needed)
  unsigned long long cuts = 0ull;
  int lo = 0, attempts = 0;
  bool spread = false;
  while (lo < bundle) {
    ++attempts;
    TFRT_STAT(30, 1);
    TFRT_WAVE_NOTE(0, 1);
    const unsigned long long above = lo < 63 ? (cuts >> (lo + 1)) << (lo + 1) : 0ull;
    const int hi = above != 0ull ? __ffsll((long long)above) - 1 : bundle;
    const int len = hi - lo;
    // this lane's ray in float32, relative to c0 (formed anew for every bundle, from the LDS copy:
    // nothing of it has to stay in registers while the faces are walked)
    wave_fence();
    RT own[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) own[k] = W.rtab[k][lane];
    const float sx = rel(own[0], cx, cxf), sy = rel(own[1], cy, cyf), sz = rel(own[2], cz, czf);
    const float dx = (float)(own[3] - own[0]), dy = (float)(own[4] - own[1]),
                dz = (float)(own[5] - own[2]);
    const float len2 = dx * dx + dy * dy + dz * dz;
    // (a zero-length or non-finite ray can hit nothing: it takes no part and ends up dead)
    const bool ok = live && len2 > 0.f && len2 < 3.0e38f;
    const float inv = ok ? __builtin_amdgcn_rsqf(len2) : 0.f;
    const float ux = dx * inv, uy = dy * inv, uz = dz * inv;
    const float rabs = fabsf(sx) + fabsf(sy) + fabsf(sz);
    const bool sel = ok && lane >= lo && lane < hi;
    // (the first supercluster spheres are fetched while the bundle is formed: one round trip
    // less in the chain ray -> bundle -> level 0 -> level 1 -> level 2 -> faces)
    const float4 su_first = lane < n_super ? susphere[lane] : never;
    const float cnt = (float)__popcll(__ballot(sel));
    bool narrow = true, brute = false;
    int ns = 0, nc = 0, nf = 0;
    Beam bm = {0.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, -INFINITY, 1.f, 0.f, 0.f, 0.f, 1.f, 0.f,
               INFINITY, 0.f};
    float ray_t = 0.f, ray_dt = 0.f, ray_terr = 0.f;  // this lane's ray along the axis: start, d . w
    float Ax = 0.f, Ay = 0.f, Mx = 0.f, My = 0.f, Mlen = 0.f, Perr = 0.f, Pm = 0.f;  // ... its image
    if (cnt > 0.f) {
      // ---- the bundle
      float swx = sel ? ux : 0.f, swy = sel ? uy : 0.f, swz = sel ? uz : 0.f;
      float sox = sel ? sx : 0.f, soy = sel ? sy : 0.f, soz = sel ? sz : 0.f;
      float pad0 = 0.f, pad1 = 0.f;
      wave_sum4(swx, swy, swz, sox);
      wave_sum4(soy, soz, pad0, pad1);
      const float wl2 = swx * swx + swy * swy + swz * swz;
      narrow = wl2 > 0.49f * cnt * cnt;  // (NaN: false)
      spread = !narrow;
      const float wi = narrow ? __builtin_amdgcn_rsqf(wl2) : 0.f;
      bm.wx = swx * wi; bm.wy = swy * wi; bm.wz = swz * wi;
      const float icnt = __builtin_amdgcn_rcpf(cnt);
      bm.ox = sox * icnt; bm.oy = soy * icnt; bm.oz = soz * icnt;
      const float cosk = ux * bm.wx + uy * bm.wy + uz * bm.wz;
      if (__any(sel && !(cosk > 0.7f))) narrow = false, spread = true;
      const float px = sx - bm.ox, py = sy - bm.oy, pz = sz - bm.oz;
      const float ts = px * bm.wx + py * bm.wy + pz * bm.wz;
      ray_t = ts;
      ray_dt = dx * bm.wx + dy * bm.wy + dz * bm.wz;
      ray_terr = 1e-5f * (rabs + fabsf(bm.ox) + fabsf(bm.oy) + fabsf(bm.oz));
      // the image plane: e1 = normalize(w x axis least aligned with w), e2 = w x e1
      {
        const float f0 = fabsf(bm.wx), f1 = fabsf(bm.wy), f2 = fabsf(bm.wz);
        const bool k0 = f0 <= f1 && f0 <= f2, k1 = !k0 && f1 <= f2;
        const float kx = k0 ? 1.f : 0.f, ky = k1 ? 1.f : 0.f, kz = (!k0 && !k1) ? 1.f : 0.f;
        const float cxx = bm.wy * kz - bm.wz * ky, cyy = bm.wz * kx - bm.wx * kz,
                    czz = bm.wx * ky - bm.wy * kx;
        const float il = __builtin_amdgcn_rsqf(cxx * cxx + cyy * cyy + czz * czz);
        bm.e1x = cxx * il; bm.e1y = cyy * il; bm.e1z = czz * il;
        bm.e2x = bm.wy * bm.e1z - bm.wz * bm.e1y;
        bm.e2y = bm.wz * bm.e1x - bm.wx * bm.e1z;
        bm.e2z = bm.wx * bm.e1y - bm.wy * bm.e1x;
      }
      // this ray's image p(t) = A + M t
      {
        const float idt = __builtin_amdgcn_rcpf(sel ? ray_dt : 1.f);
        Mx = (dx * bm.e1x + dy * bm.e1y + dz * bm.e1z) * idt;
        My = (dx * bm.e2x + dy * bm.e2y + dz * bm.e2z) * idt;
        const float p1 = px * bm.e1x + py * bm.e1y + pz * bm.e1z,
                    p2 = px * bm.e2x + py * bm.e2y + pz * bm.e2z;
        Ax = p1 - ts * Mx;
        Ay = p2 - ts * My;
        Mlen = __builtin_amdgcn_sqrtf(Mx * Mx + My * My) * 1.0001f;
        Pm = 2e-5f * (fabsf(Mx) + fabsf(My));
        Perr = 2e-5f * (fabsf(p1) + fabsf(p2) + fabsf(ts) * (fabsf(Mx) + fabsf(My)) + rabs +
                        fabsf(bm.ox) + fabsf(bm.oy) + fabsf(bm.oz));
      }
      // (offset and slope of the ray's line from the axis are those of its image: e1, e2, w are
      // orthonormal to rounding, which the inflation below covers many times over)
      float R0 = sel ? __builtin_amdgcn_sqrtf(Ax * Ax + Ay * Ay) : 0.f, S = sel ? Mlen : 0.f;
      float Lw = sel ? rabs : 0.f, tmin = sel ? ts : INFINITY;
      wave_max3_min(R0, S, Lw, tmin);
      Lw += fabsf(bm.ox) + fabsf(bm.oy) + fabsf(bm.oz);
      // A second anchor of the same bound: R0 + S |t| is tight near the plane where R0 was taken
      // and loosens with the distance from it, because S is the largest slope, not the spread.
      // The starts' plane is the right one once the rays are inside the scene -- but rays sorted
      // by where they cross the scene and coming from all over an extended object ten units away
      // (a source re-drawn at random every step, dev/hexalens.py:36-48) are a double cone whose
      // narrow part is at the lens: measured from their starts the bundle looked as wide there as
      // the object.  So the bundle's radius is also taken in the plane through the frame origin
      // c0 (which lies in the mesh), at axial coordinate tb, and a node is tested against the
      // smaller of the two bounds.
      const float tb = -(bm.ox * bm.wx + bm.oy * bm.wy + bm.oz * bm.wz);
      float R0b = 0.f;
      if (sel) {
        const float bx = Ax + Mx * tb, by = Ay + My * tb;
        R0b = __builtin_amdgcn_sqrtf(bx * bx + by * by);
      }
      R0b = wave_max_f(R0b);
      // bounds inflated far beyond their float32 rounding (offsets ~ 2^-23 Lw, slopes ~ 2^-23)
      bm.R0 = R0 * 1.001f + 4e-6f * Lw;
      bm.S = S * 1.001f + 2e-6f;
      bm.tb = tb;
      bm.R0b = R0b * 1.001f + 4e-6f * (Lw + fabsf(tb)) + 2e-6f * fabsf(tb) * S;
      if (!(bm.R0b < 3.0e38f)) bm.R0b = INFINITY;   // (NaN or overflow: the first bound alone)
      // (first pass of a trace: sources normally sit outside the scene, nothing lies behind
      // them; and only while hits must lie ahead of the start: ray_start_epsilion >= 0)
      bm.tmin = (!first_pass && eps_start >= 0.0) ? tmin - 1e-5f * Lw - 1e-5f * fabsf(tmin)
                                                  : -INFINITY;
      if (!(bm.R0 < 3.0e38f && bm.S < 3.0e38f)) narrow = false;  // (also NaN)
      // (wave-uniform, but computed by vector instructions: into scalar registers)
      bm.ox = uniform_f(bm.ox); bm.oy = uniform_f(bm.oy); bm.oz = uniform_f(bm.oz);
      bm.wx = uniform_f(bm.wx); bm.wy = uniform_f(bm.wy); bm.wz = uniform_f(bm.wz);
      bm.R0 = uniform_f(bm.R0); bm.S = uniform_f(bm.S); bm.tmin = uniform_f(bm.tmin);
      bm.R0b = uniform_f(bm.R0b); bm.tb = uniform_f(bm.tb);
      bm.e1x = uniform_f(bm.e1x); bm.e1y = uniform_f(bm.e1y); bm.e1z = uniform_f(bm.e1z);
      bm.e2x = uniform_f(bm.e2x); bm.e2y = uniform_f(bm.e2y); bm.e2z = uniform_f(bm.e2z);
      TFRT_TICK(1);

      // ---- levels: lane = node
      if (narrow) {
        float4 nxt = su_first;
        for (int b = 0; b < n_super; b += 64) {
          const int node = b + lane;
          const float4 cur = nxt;  // (the next 64 spheres are on their way while these are tested)
          nxt = node + 64 < n_super ? susphere[node + 64] : never;
          const bool hit = beam_touch(bm, cur);
          const unsigned long long m = __ballot(hit);
          if (hit) {
            const int pos = ns + rank_below(m);
            if (pos < BEAM_SLIST) W.slist[pos] = (uint16_t)node;
          }
          ns += __popcll(m);
        }
        narrow = ns <= BEAM_SLIST && n_super <= 65536;
      }
      wave_fence();
      TFRT_TICK(2);
      if (narrow && ns > 0) {
        for (int b = 0; b < ns * SUPER; b += 64) {
          const int k = b + lane;
          bool hit = false;
          int cl = 0;
          if (k < ns * SUPER) {
            cl = (int)W.slist[k >> 3] * SUPER + (k & 7);
            if (cl < n_clusters) hit = beam_touch(bm, clsphere[cl]);
          }
          const unsigned long long m = __ballot(hit);
          if (hit) {
            const int pos = nc + rank_below(m);
            if (pos < BEAM_CLIST) W.clist[pos] = (uint16_t)cl;
          }
          nc += __popcll(m);
        }
        narrow = nc <= BEAM_CLIST && n_clusters <= 65536;
      }
      wave_fence();
      TFRT_TICK(3);
      if (narrow && nc > 0) {
        for (int b = 0; b < nc * CLUSTER; b += 64) {
          const int k = b + lane;
          bool hit = false;
          unsigned slot = 0;
          if (k < nc * CLUSTER) {
            slot = (unsigned)W.clist[k >> 4] * CLUSTER + (unsigned)(k & 15);
            hit = beam_touch(bm, csphere[slot]);
          }
          const unsigned long long m = __ballot(hit);
          if (hit) {
            const int pos = nf + rank_below(m);
            if (pos < BEAM_FLIST) W.flist[pos] = slot;
          }
          nf += __popcll(m);
        }
        // (a bundle that touches several times the faces a coherent one does -- the ray order
        // jumps inside it -- is cheaper as two: every ray is tested against every face left)
        narrow = nf <= BEAM_FLIST && !(nf > BEAM_WIDE && len > 8 && attempts <= 6);
      }
      wave_fence();
      TFRT_TICK(4);
    }
    if (!narrow) {
      TFRT_TICK(10);
      // (coherent_only: the caller launches no grouped kernel behind this one -- it has seen this
      // source leave no wavefront over -- so every wavefront is finished here, however wide)
      const bool hopeless = !coherent_only &&
                            ((len == bundle && (spread || (nc > 64 && nc > n_clusters / 4))) ||
                             attempts >= BEAM_ATTEMPTS);
      if (len > 1 && !hopeless) {  // cut at the widest gap inside [lo, hi), try the first part
        if (cuts == 0ull) {
          // distance of this lane's line from the previous lane's, near the scene: feet of the
          // perpendiculars from the frame origin, plus the directions' difference over that distance
          const float su = sx * ux + sy * uy + sz * uz;
          const float fx = sx - su * ux, fy = sy - su * uy, fz = sz - su * uz;
          const float gx = fx - __shfl_up(fx, 1, 64), gy = fy - __shfl_up(fy, 1, 64),
                      gz = fz - __shfl_up(fz, 1, 64);
          const float hx = ux - __shfl_up(ux, 1, 64), hy = uy - __shfl_up(uy, 1, 64),
                      hz = uz - __shfl_up(uz, 1, 64);
          const bool prev_ok = __shfl_up(ok ? 1 : 0, 1, 64) != 0;
          gap = 0.f;
          if (ok && prev_ok && lane > 0)
            gap = gx * gx + gy * gy + gz * gz +
                  (sx * sx + sy * sy + sz * sz) * (hx * hx + hy * hy + hz * hz);
          if (!(gap >= 0.f)) gap = 0.f;  // (NaN)
        }
        const bool inside = lane > lo && lane < hi;
        const float widest = wave_max_f(inside ? gap : -1.f);
        const unsigned long long at = __ballot(inside && gap == widest);
        // (all gaps equal -- e.g. zero: lanes without a ray -- : halve)
        const int g = (widest > 0.f && at != 0ull) ? __ffsll((long long)at) - 1 : lo + (len >> 1);
        cuts |= 1ull << g;
        continue;
      }
      if (len == 1 && !hopeless) {
        // ONE ray whose line touches more nodes than the lists hold (it runs along a surface, or
        // the scene is huge): that ray against EVERY face, sixty-four member slots at a time
        // through the same stage as the lists' faces (its bundle is the ray itself: face_frame
        // tests the triangles against its line).  Rare and slow (M / 64 rounds), but never wrong.
        brute = true;
      } else {
        // not a wavefront of (a few) narrow bundles: the grouped kernel does it
        TFRT_STAT(ns > BEAM_SLIST ? 10 : (nc > BEAM_CLIST ? 11 : (nf > BEAM_FLIST ? 12 : 9)), 1);
        TFRT_STAT(15, lo);
        return false;
      }
    }

    // ---- faces (see face_frame)
    // (1) LANE = FACE: the triangle itself against the bundle -- a bounding sphere has 2.4 times
    //     the area of its face -- and the records of the faces left, nearest (along the axis)
    //     first, into the LDS;
    // (2) LANE = RAY, one face at a time: is the ray's image within the face's?  (a dozen
    //     instructions on a record every lane reads from the same LDS address) -- the (ray, face)
    //     pairs that pass queue up for
    // (3) LANE = PAIR: the exact float64 decision.
    // The walk ends as soon as every ray has a hit nearer than the nearest point of every face
    // left (a lens: the second surface and the target are never tested in the first pass).
    TFRT_STAT(13, nf);
    TFRT_WAVE_NOTE(3, nf);
    TFRT_TICK(5);
    // upper bound of the axial coordinate of this ray's nearest hit so far (inf: none)
    auto reach_now = [&]() {
      const double best = dkey_inv(W.best_k[lane]);
      const float bu = nextafterf((float)best, INFINITY) * ray_dt;  // (d . w > 0: cos > 0.7)
      return ray_t + bu + 1e-5f * fabsf(bu) + ray_terr;
    };
    float myreach = reach_now();
    bool queued = false;  // a pair of this ray waits for its decision
    // Chunks of up to 64 candidate faces: the entries of flist -- or, for a single ray whose lists
    // overflowed, every member slot of the scene.  The LAST chunk of the wavefront's last bundle
    // also decides what is still queued (one place for all decisions: the float64 test's
    // registers are not multiplied by the number of places that run it).
    const int n_cand = brute ? n_clusters * CLUSTER : nf;
    int f0 = 0;
    do {
      const int nb = min(64, n_cand - f0);
      if (work != nullptr) work[1] += (unsigned)nb;
      float4 rec[4] = {never, never, never, never};  // (set: nothing is carried around the loop)
      bool touch = false;
      float tnear = INFINITY;
      bool cand = lane < nb;
      if (brute && cand) {
        // (every slot of the scene: first the ray's line -- the axis of its bundle -- against the
        // slot's sphere, or this would be two hundred instructions per 64 slots)
        const float4 sp = csphere[f0 + lane];
        const float vx = sp.x - bm.ox, vy = sp.y - bm.oy, vz = sp.z - bm.oz;
        const float vu = vx * bm.wx + vy * bm.wy + vz * bm.wz;
        const float v2 = vx * vx + vy * vy + vz * vz;
        const float rr = __builtin_amdgcn_sqrtf(sp.w) * 1.000002f + bm.radius(vu, 0.f);
        cand = v2 - vu * vu <= rr * rr + 4e-6f * v2;  // (padding: w < 0, NaN, never)
      }
      if (cand) {
        const int64_t memb = brute ? (int64_t)(f0 + lane) : (int64_t)W.flist[f0 + lane];
        const float4 r0 = crec[3 * memb], r1 = crec[3 * memb + 1], r2 = crec[3 * memb + 2];
        touch = face_frame(bm, r0, r1, r2, es_f, rec) && __float_as_int(r0.w) >= 0;
        tnear = rec[0].w;
        if (tnear != tnear) tnear = -INFINITY;  // (NaN: never skipped)
        rec[0].w = tnear;
      }
      f0 += 64;
      const bool closing = f0 >= n_cand && hi == bundle;  // nothing comes after this chunk
      // nearest first: a face's place = the number of faces that begin nearer (ties: lower lane)
      const unsigned long long tm = __ballot(touch);
      const int nt = __popcll(tm);
      int place = 0;
      for (unsigned long long rest = tm; rest != 0ull; rest &= rest - 1ull) {
        const int o = __ffsll((long long)rest) - 1;
        const float to = bcast_f(tnear, o);
        place += (to < tnear || (to == tnear && o < lane)) ? 1 : 0;
      }
      wave_fence();  // (the previous chunk's records have been read)
      if (touch) {
        W.ftab[place][0] = rec[0];
        W.ftab[place][1] = rec[1];
        W.ftab[place][2] = rec[2];
        W.ftab[place][3] = rec[3];
      }
      wave_fence();
      TFRT_STAT(27, nt);
      TFRT_TICK(6);
      float span = -INFINITY;  // farthest axial coordinate of the faces taken so far
      // one face of the record table against this lane's ray: the record's sixteen words have been
      // read by sixteen lanes (from lane `b` on) and are handed round with v_readlane -- scalar
      // operands; sixty-four lanes reading ONE LDS address would be served one after the other
      auto on_face = [&](const float word, const int b, int* j, float* thi) {
        const float tn = bcast_f(word, b + 3), tc = bcast_f(word, b + 11);
        *thi = bcast_f(word, b + 7);
        *j = __float_as_int(bcast_f(word, b + 14));
        const float ix = Ax + Mx * tc, iy = Ay + My * tc;
        const float room = Mlen * bcast_f(word, b + 12) + bcast_f(word, b + 13) + Perr + Pm * fabsf(tc);
        // (valid hits lie ahead of the start -- where bm.tmin says so --, and nearer than the
        // ray's nearest hit so far)
        return sel && *j != skip && !(myreach < tn) &&
               !(bm.tmin > -INFINITY && *thi < ray_t - ray_terr) &&
               !(bcast_f(word, b) * ix + bcast_f(word, b + 1) * iy + bcast_f(word, b + 2) > room) &&
               !(bcast_f(word, b + 4) * ix + bcast_f(word, b + 5) * iy + bcast_f(word, b + 6) > room) &&
               !(bcast_f(word, b + 8) * ix + bcast_f(word, b + 9) * iy + bcast_f(word, b + 10) > room);
      };
      // two faces per round (the rounds' scalar bookkeeping and branches cost as much as a face)
      for (int c = 0;; c += 2) {
        const bool end = c >= nt;  // (behind the chunk's last face: only decisions, if any)
        const bool two = c + 1 < nt;
        const float word =
            reinterpret_cast<const float*>(&W.ftab[end ? 0 : c][0])[lane & (two ? 31 : 15)];
        const float tn = end ? INFINITY : bcast_f(word, 3);
        // decide what is queued: full batches; or at a gap in depth when every ray has a hit or
        // a candidate (the walk may end here); or at the very end
        const bool flush =
            end ? closing
                : (tn > span && __ballot(sel && !(queued || myreach < INFINITY)) == 0ull);
        while (xn >= 64 || (flush && xn > 0)) {
          const int nd = min(xn, 64);
          wave_fence();
          decide(nd);
          // (what is left, fewer than 128 pairs, moves to the front)
          uint32_t tp = 0u, tq = 0u;
          if (lane < xn - nd) tp = W.x_pair[nd + lane];
          if (lane + 64 < xn - nd) tq = W.x_pair[nd + 64 + lane];
          wave_fence();
          if (lane < xn - nd) W.x_pair[lane] = tp;
          if (lane + 64 < xn - nd) W.x_pair[64 + lane] = tq;
          xn -= nd;
          wave_fence();
          // (rays with a pair among those moved stay "queued")
          if (xn == 0) queued = false;
          myreach = reach_now();
          TFRT_TICK(8);
        }
        if (end) break;
        if (__ballot(sel && !(myreach < tn)) == 0ull) {  // all the faces left lie farther
          c = nt - 2;  // (the next round is the closing one)
          continue;
        }
        TFRT_STAT(29, two ? 2 : 1);
        TFRT_WAVE_NOTE(2, two ? 2 : 1);
        int ja, jb = -1;
        float thia, thib = -INFINITY;
        const bool keepa = on_face(word, 0, &ja, &thia);
        bool keepb = false;
        if (two) keepb = on_face(word, 16, &jb, &thib);
        span = fmaxf(span, fmaxf(thia, thib));
        const unsigned long long kma = __ballot(keepa), kmb = __ballot(keepb);
        const int na = __popcll(kma);
        if (keepa) W.x_pair[xn + rank_below(kma)] = ((uint32_t)ja << 6) | (uint32_t)lane;
        if (keepb) W.x_pair[xn + na + rank_below(kmb)] = ((uint32_t)jb << 6) | (uint32_t)lane;
        queued = queued || keepa || keepb;
        xn += na + __popcll(kmb);
        if (work != nullptr) work[0] += (unsigned)(na + __popcll(kmb));
        TFRT_STAT(14, na + __popcll(kmb));
        TFRT_TICK(7);
      }
    } while (f0 < n_cand);
    lo = hi;  // this bundle is done: the next one starts behind it
  }
  wave_fence();
  return true;
}

// BW wavefronts per workgroup.  The wavefronts of a workgroup share nothing and never meet at a
// barrier, so one per workgroup (BW = 1) lets the dispatcher hand out work wavefront by wavefront:
// a four-wavefront workgroup holds its LDS and its wave slots until the slowest of the four is done.
template <typename T, int BW>
__global__ __launch_bounds__(64 * BW) void k_intersect_beam(
    const T* __restrict__ rays, int64_t stride, const int32_t* __restrict__ n_ptr,
    const int32_t* __restrict__ last_tri, BeamScene g, const int32_t* __restrict__ catagory,
    int32_t* __restrict__ rec_tri, double* __restrict__ rec_t, uint8_t* __restrict__ rec_cls,
    int32_t* __restrict__ hist, int32_t* __restrict__ left_list, int32_t* __restrict__ left_count,
    int32_t* __restrict__ left_total, int coherent_only, int bundle) {
  // bundle = rays per wavefront: 64, or 32 (lanes 32..63 carry no ray; they still test nodes) for
  // launches that leave the chip half empty -- twice the wavefronts, each with fewer candidate
  // faces and half the decisions: the launch is as long as ONE wavefront's chain of dependent
  // steps then, not as the work.  (32 only with coherent_only: the grouped kernel takes whole
  // 64-ray wavefronts.)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qwave = blockIdx.x * BW + wave;
  const int q = qwave * bundle + lane;
  const int n = *n_ptr;
  if (qwave * bundle >= n) return;  // (whole wave; no block-level synchronisation in this kernel)
  TFRT_WAVE_BEGIN;

  // this lane's ray: coalesced reads of the ray block
  using RT = std::conditional_t<sizeof(T) <= 4, float, double>;
  const int i = (lane < bundle && q < n) ? q : -1;
  const bool live = i >= 0;
  const int64_t ii = live ? i : 0;
  __shared__ BeamLds<RT> lds[BW];
  BeamLds<RT>& W = lds[wave];
#pragma unroll
  for (int k = 0; k < 6; ++k) W.rtab[k][lane] = static_cast<RT>(rays[k * stride + ii]);
  const int skip = (live && last_tri != nullptr) ? last_tri[ii] : -1;
  if (!beam_pass<T, RT>(W, g, lane, bundle, live, skip, last_tri == nullptr, coherent_only,
                        TFRT_WI_ARG)) {
    // not a wavefront of (a few) narrow bundles: the grouped kernel does it
    if (lane == 0) {
      left_list[atomicAdd(left_count, 1)] = qwave;
      atomicAdd(left_total, 1);
    }
    return;
  }
  wave_fence();

  // ---- hit record, class and this wavefront's share of its 256-ray block's class histogram
  int cls = -1;
  if (live) {
    const int bi = W.best_i[lane];
    cls = (bi < 0) ? CLS_DEAD : cat_to_cls(catagory[bi]);
    rec_tri[i] = bi;
    rec_t[i] = dkey_inv(W.best_k[lane]);
    rec_cls[i] = (uint8_t)cls;
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int cnt_c = __popcll(__ballot(cls == c));
    if (lane == c && cnt_c > 0) atomicAdd(&hist[((qwave * bundle) >> 8) * 4 + c], cnt_c);
  }
  TFRT_WAVE_END(qwave);
}

// -------------------------------------------------------------------------- classify

__global__ __launch_bounds__(BLOCK) void k_classify3d(
    const int32_t* __restrict__ n_ptr, int chunks, const double* __restrict__ part_t,
    const int32_t* __restrict__ part_i, int64_t part_stride,
    const int32_t* __restrict__ catagory, int32_t* __restrict__ rec_tri,
    double* __restrict__ rec_t, uint8_t* __restrict__ rec_cls, int32_t* __restrict__ blockcnt) {
  const int n = *n_ptr;
  const int base = blockIdx.x * BLOCK;
  if (base >= n) return;
  const int i = base + threadIdx.x;
  int cls = -1;
  if (i < n) {
    double bt = INFINITY;
    int bi = -1;
    for (int c = 0; c < chunks; ++c) {  // ties: the lowest face index wins (tf.argmin)
      const double t = part_t[c * part_stride + i];
      const int pi = part_i[c * part_stride + i];
      if (t < bt || (t == bt && pi >= 0 && pi < bi)) {
        bt = t;
        bi = pi;
      }
    }
    cls = (bi < 0) ? CLS_DEAD : cat_to_cls(catagory[bi]);
    rec_tri[i] = bi;
    rec_t[i] = bt;
    rec_cls[i] = (uint8_t)cls;
  }
  __shared__ int wc[WAVES][4];
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const unsigned long long m = __ballot(cls == c);
    if (lane_id() == 0) wc[wave][c] = __popcll(m);
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    int s = 0;
    for (int w = 0; w < WAVES; ++w) s += wc[w][threadIdx.x];
    blockcnt[blockIdx.x * 4 + threadIdx.x] = s;
  }
}

// ------------------------------------------------------------------------------ scan

// One block of 1024 threads: exclusive scan of the per-block class histograms (up to
// SCAN_GRID_MIN_ROWS ray blocks; the offsets it writes are global, k_react3d gets no rowbase).
__global__ __launch_bounds__(1024) void k_scan3d_one(const int32_t* __restrict__ n_ptr,
                                                 const int32_t* __restrict__ blockcnt,
                                                 int32_t* __restrict__ blockoff,
                                                 int32_t* __restrict__ pass_counts,
                                                 int32_t* __restrict__ totals,
                                                 int32_t* __restrict__ n_next,
                                                 unsigned long long* __restrict__ n_tests,
                                                 int M) {
  const int n = *n_ptr;
  const int nblk = (n + BLOCK - 1) / BLOCK;
  const int per = (nblk + 1023) / 1024;
  const int b0 = min(nblk, (int)threadIdx.x * per), b1 = min(nblk, b0 + per);
  int loc[4] = {0, 0, 0, 0};
  for (int b = b0; b < b1; ++b)
    for (int c = 0; c < 4; ++c) loc[c] += blockcnt[b * 4 + c];

  __shared__ int wsum[16][4];
  __shared__ int wbase[16][4];
  __shared__ int total[4];
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  int pre[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    int v = loc[c];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(v, d, 64);
      if (lane >= d) v += o;
    }
    pre[c] = v - loc[c];  // exclusive within the wave
    if (lane == 63) wsum[wave][c] = v;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    int run = 0;
    for (int w = 0; w < 16; ++w) {
      wbase[w][threadIdx.x] = run;
      run += wsum[w][threadIdx.x];
    }
    total[threadIdx.x] = run;
  }
  __syncthreads();
  int run[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) run[c] = wbase[wave][c] + pre[c];
  for (int b = b0; b < b1; ++b)
    for (int c = 0; c < 4; ++c) {
      blockoff[b * 4 + c] = run[c];
      run[c] += blockcnt[b * 4 + c];
    }
  if (threadIdx.x < 4) {
    const int c = threadIdx.x;
    pass_counts[c] = total[c];
    pass_counts[4 + c] = totals[c];
    totals[c] += total[c];
  }
  if (threadIdx.x == 0) {
    *n_next = total[CLS_ACTIVE];
    *n_tests += (unsigned long long)n * (unsigned long long)M;
  }
}

// Exclusive scan of the per-block class histograms over a grid of 1024-thread workgroups
// (scan_rows_grid); the workgroup that finishes last also closes the pass: class totals, the
// next pass's ray count and the test counter.  Its ticket and fences cost ~13 us whatever the
// size (the single-block scan takes 7 us for the 3,907 rows of 1M rays, 54 us for 15.6k rows):
// used from SCAN_GRID_MIN_ROWS ray blocks on.
constexpr int SCAN_GRID_MIN_ROWS = 8192;
__global__ __launch_bounds__(1024) void k_scan3d(const int32_t* __restrict__ n_ptr,
                                                 const int32_t* __restrict__ blockcnt,
                                                 int32_t* __restrict__ blockoff,
                                                 int32_t* __restrict__ rowtot,
                                                 int32_t* __restrict__ rowbase,
                                                 unsigned int* __restrict__ ticket,
                                                 int32_t* __restrict__ pass_counts,
                                                 int32_t* __restrict__ totals,
                                                 int32_t* __restrict__ n_next,
                                                 unsigned long long* __restrict__ n_tests,
                                                 int M) {
  const int n = *n_ptr;
  const int nblk = (n + BLOCK - 1) / BLOCK;
  __shared__ int total[4];
  if (!scan_rows_grid<4>(blockcnt, blockoff, nblk, rowtot, rowbase, ticket, total)) return;
  if (threadIdx.x < 4) {
    const int c = threadIdx.x;
    pass_counts[c] = total[c];
    pass_counts[4 + c] = totals[c];
    totals[c] += total[c];
  }
  if (threadIdx.x == 0) {
    *n_next = total[CLS_ACTIVE];
    *n_tests += (unsigned long long)n * (unsigned long long)M;
  }
}

// ----------------------------------------------------------------------------- react

__device__ __forceinline__ void face_indices(const tfrt_scene3d& sc, int tri, int rid,
                                             double* n_in, double* n_out) {
  if (sc.n_table != nullptr && sc.mat_in != nullptr) {
    // (one wavelength for every ray: one column, the same few values for every lane -- not
    // sixteen bytes per ray and pass from a table as long as the source)
    const int64_t col = sc.n_table_uniform ? 0 : rid;
    *n_in = sc.n_table[(int64_t)sc.mat_in[tri] * sc.n_table_stride + col];
    *n_out = sc.n_table[(int64_t)sc.mat_out[tri] * sc.n_table_stride + col];
  } else {
    *n_in = sc.n_in[tri];
    *n_out = sc.n_out[tri];
  }
}

template <typename T>
__device__ __forceinline__ bool emit(const tfrt_ray_out& o, int64_t slot, const double s[3],
                                     const double e[3], int rid, int face) {
  if (o.rays == nullptr) return true;
  if (slot >= o.capacity) return false;
  store_ray3(static_cast<T*>(o.rays), o.capacity, slot, s, e);
  if (o.ray_id) o.ray_id[slot] = rid;
  if (o.face) o.face[slot] = face;
  return true;
}

// k_react3d's own scan (launches of <= SELF_SCAN_MAX_BLOCKS ray blocks: one launch less per pass)
constexpr int SELF_SCAN_MAX_BLOCKS = 4096;
struct SelfScan {
  const int32_t* blockcnt = nullptr;     // per-block class histograms; nullptr: a scan kernel ran
  const int32_t* prev_counts = nullptr;  // counts row of the previous pass (nullptr: first pass)
  int32_t* counts_row = nullptr;         // counts row of this pass (written by the last block)
  int32_t* totals = nullptr;             // running class totals of the trace
  int32_t* n_next = nullptr;             // ray count of the next pass
  unsigned long long* n_tests = nullptr;
  int M = 0;
  // coherent-ray traces: the class histogram of the NEXT pass (the intersect kernels add into
  // it with atomics); every block clears its own row here, one pass ahead
  int32_t* hist_next = nullptr;
  int hist_rows = 0;
};

// (80 scalar registers: the argument list alone -- scene, four output classes, the scan's pointers --
// asks for 106, which admits six workgroups per CU; with eight the launch's 3,907 workgroups need
// two rounds instead of three: 24 -> 21.6 us at a million rays)
template <typename T>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_num_sgpr(80))) void k_react3d(
    const T* __restrict__ rays_in, int64_t stride_in, const int32_t* __restrict__ n_ptr,
    const int32_t* __restrict__ ray_id_in, const int32_t* __restrict__ rec_tri,
    const double* __restrict__ rec_t, uint8_t* __restrict__ rec_cls,
    const int32_t* __restrict__ blockoff, const int32_t* __restrict__ rowbase,
    const int32_t* __restrict__ pass_counts,
    tfrt_scene3d sc, double L, double dead_len, uint32_t flags, T* __restrict__ rays_out,
    int64_t stride_out, int32_t* __restrict__ ray_id_out, int32_t* __restrict__ last_tri_out,
    int32_t* __restrict__ rec_slot, tfrt_ray_out fin, tfrt_ray_out act, tfrt_ray_out stp,
    tfrt_ray_out dead, int32_t* __restrict__ err, float* __restrict__ prep_next, int64_t pstride,
    const double* __restrict__ c0, SelfScan ss, const double* __restrict__ fnorm,
    const double* __restrict__ feta) {
  const int n = *n_ptr;
  const int base = blockIdx.x * BLOCK;
  // Self-scan mode (few ray blocks): no scan launch ran.  Every block sums the class histograms
  // of the blocks before it itself, and the last one closes the pass like k_scan3d_one does.
  const bool self = ss.blockcnt != nullptr;
  int before[4] = {0, 0, 0, 0};  // rays of each class emitted by earlier passes
  if (self && ss.prev_counts != nullptr) {
#pragma unroll
    for (int c = 0; c < 4; ++c) before[c] = ss.prev_counts[4 + c] + ss.prev_counts[c];
  }
  if (self && n == 0 && blockIdx.x == 0) {  // nothing left to trace: the pass is empty
    if (threadIdx.x < 4) {
      ss.counts_row[threadIdx.x] = 0;
      ss.counts_row[4 + threadIdx.x] = before[threadIdx.x];
    }
    if (threadIdx.x == 0) *ss.n_next = 0;
  }
  if (base >= n) return;
  if (ss.hist_next != nullptr && threadIdx.x < 4) ss.hist_next[blockIdx.x * 4 + threadIdx.x] = 0;
  if (ss.hist_next != nullptr && blockIdx.x == 0 && threadIdx.x == 4)
    ss.hist_next[ss.hist_rows * 4] = 0;  // (the count of wavefronts left to the grouped kernel)
  const int i = base + threadIdx.x;
  const int cls = (i < n) ? (int)rec_cls[i] : -1;

  // stable rank of this ray inside its class within the block
  __shared__ int wc[WAVES][4];
  __shared__ int wpre[WAVES][4];
  const int wave = threadIdx.x >> 6;
  int rank = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const unsigned long long m = __ballot(cls == c);
    if (cls == c) rank = rank_below(m);
    if (lane_id() == 0) wc[wave][c] = __popcll(m);
  }
  if (self) {
    int loc[4] = {0, 0, 0, 0};
    const int4* __restrict__ rows = reinterpret_cast<const int4*>(ss.blockcnt);
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += BLOCK) {
      const int4 v = rows[b];
      loc[0] += v.x;
      loc[1] += v.y;
      loc[2] += v.z;
      loc[3] += v.w;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) loc[c] += __shfl_xor(loc[c], d, 64);
      if (lane_id() == 0) wpre[wave][c] = loc[c];
    }
  }
  __syncthreads();
  if (self && blockIdx.x == (unsigned)((n - 1) / BLOCK) && threadIdx.x < 4) {
    const int c = threadIdx.x;
    int total = 0;
    for (int w = 0; w < WAVES; ++w) total += wpre[w][c] + wc[w][c];
    ss.counts_row[c] = total;
    ss.counts_row[4 + c] = before[c];
    ss.totals[c] = before[c] + total;
    if (c == CLS_ACTIVE) *ss.n_next = total;
    if (c == 1) *ss.n_tests += (unsigned long long)n * (unsigned long long)ss.M;
  }
  if (i >= n) return;
  for (int w = 0; w < wave; ++w) rank += wc[w][cls];
  int slot;
  int64_t gslot;
  if (self) {
    int off = 0;
    for (int w = 0; w < WAVES; ++w) off += wpre[w][cls];
    slot = off + rank;
    gslot = (int64_t)before[cls] + slot;
  } else {
    // within this pass: offset inside the scan workgroup's 1024 blocks + that workgroup's base
    slot = blockoff[blockIdx.x * 4 + cls] +
           (rowbase != nullptr ? rowbase[(blockIdx.x >> 10) * 4 + cls] : 0) + rank;
    gslot = (int64_t)pass_counts[4 + cls] + slot;  // within the whole trace
  }

  double s[3], e[3];
  load_ray3(rays_in, stride_in, i, s, e);
  const int rid = ray_id_in ? ray_id_in[i] : i;
  const int tri = rec_tri[i];
  bool ok = true;

  if (cls == CLS_DEAD) {
    if (flags & TFRT_COMPILE_DEAD) {
      double e2[3] = {e[0], e[1], e[2]};
      if (dead_len != 0.0)
        for (int k = 0; k < 3; ++k) e2[k] = advance_between(s[k], dead_len, e[k]);
      ok = emit<T>(dead, gslot, s, e2, rid, -1);
    }
    rec_slot[i] = (int32_t)gslot;
  } else {
    double h[3];
    hit_point(s, e, rec_t[i], h);
    if (cls == CLS_FINISHED) {
      if (flags & TFRT_COMPILE_FINISHED) ok = emit<T>(fin, gslot, s, h, rid, tri);
      rec_slot[i] = (int32_t)gslot;
    } else if (cls == CLS_STOPPED) {
      if (flags & TFRT_COMPILE_STOPPED) ok = emit<T>(stp, gslot, s, h, rid, tri);
      rec_slot[i] = (int32_t)gslot;
    } else {  // ACTIVE
      if (flags & TFRT_COMPILE_ACTIVE) ok = emit<T>(act, gslot, s, h, rid, tri);
      // (the face's unit normal was formed once per face by the trace's set-up launch: the cross
      // product, a square root and four divisions of float64 less per ray, 24 B gathered for 72)
      double un[3], n1, n2, n_in;
      const double* fp = fnorm + 3 * (int64_t)tri;
#pragma unroll
      for (int q = 0; q < 3; ++q) un[q] = fp[q];
      if (feta != nullptr) {
        const double* fe = feta + 4 * (int64_t)tri;
        n1 = fe[0];
        n2 = fe[1];
        n_in = fe[2];
      } else {
        double n_out;
        face_indices(sc, tri, rid, &n_in, &n_out);
        snell_ratios(n_in, n_out, &n1, &n2);
      }
      const Snell3 f = snell3d_core(s, h, un, n1, n2, n_in == 0.0);
      // the branches this reaction took go on the tape (bits 2, 3 of the class byte): the reverse
      // sweep re-derives everything else, but with other roundings (reciprocals instead of
      // quotients), and at grazing incidence or at the critical angle must not take another side
      rec_cls[i] = (uint8_t)(cls | (f.nu > 0.0 ? TAPE_INTERNAL : 0) | (f.reflect ? TAPE_REFLECT : 0));
      double e2[3];
      for (int k = 0; k < 3; ++k) e2[k] = advance(h[k], L, f.w[k]);
      store_ray3(rays_out, stride_out, slot, h, e2);
      ray_id_out[slot] = rid;
      last_tri_out[slot] = tri;
      rec_slot[i] = slot;
      if (prep_next != nullptr) {
        // the child's filter state for the next pass (saves a k_rayprep launch and a re-read of
        // the ray block), from the child AS STORED: the exact tests see the rounded state
        double sr[3], er[3], u[3], scv[3];
        for (int k = 0; k < 3; ++k) {
          sr[k] = static_cast<double>(static_cast<T>(h[k]));
          er[k] = static_cast<double>(static_cast<T>(e2[k]));
        }
        float o[8];
        ray_filter_state(sr, er, c0, o, u, scv);
#pragma unroll
        for (int k = 0; k < 8; ++k) prep_next[k * pstride + slot] = o[k];
      }
    }
  }
  if (!ok) atomicOr(err, ERR_CAPACITY);
}

// ------------------------------------------------------------------ in-place trace
//
// tfrt_scene3d.in_place: ALL passes of a trace over coherent rays in ONE launch.  The per-pass
// sequence above (intersect -> [scan] -> react) exists because the reference compacts the children
// between passes (engine.py:2069-2111) -- but StandardReaction emits exactly one child per active
// ray (operation.py:255-307), so a lane can keep its ray: intersect (beam_pass), classify, Snell,
// next pass, with the ray in LDS / registers and one tape record per pass at slot = ray index.
// Nothing is compacted while tracing: no scan, no block of children written and read back per
// pass, one launch ramp instead of 2 P.  A wavefront whose rays have all ended leaves the loop;
// wavefronts never meet (no grid barrier).  The reference's output order -- every class lists its
// rays pass after pass in source order, engine.py:2095, 1379-1403 -- is made afterwards from the
// per-wavefront class counts the trace leaves: one scan (k_inplace_scan: the per-pass counts the
// host reads, and every wavefront's bases) and, only when a ray set is asked for, one gather
// (k_inplace_gather: rows recomputed from the tape, which holds everything they are made of).
// Results are those of the per-pass path bit for bit: a ray's nearest hit does not depend on which
// rays share its wavefront (every stage of the walk only removes pairs that cannot win), and the
// child is rounded to the state type exactly where the per-pass path stores it.
inline int inplace_bundle(int64_t N) { return N <= 160 * 1024 ? 32 : 64; }
// words per pass of the per-wavefront count rows (wavefronts of 32 rays at least; rows start on
// 256-byte boundaries)
inline size_t inplace_wstride(int64_t N) { return ((size_t)((N > 0 ? N : 1) + 31) / 32 + 63) / 64 * 64; }

template <typename T>
struct InplaceArgs {
  const T* src;          // source rays (inputs of pass 1)
  int64_t src_stride;
  int32_t N, P, bundle, nwaves;
  const InplaceTape* tape;   // (device memory: see InplaceTape)
};

// Wavefronts per SIMD: five for the plain kernel (96 registers, no scratch; with the 8 KB of LDS a
// wavefront takes, five is also what a CU's LDS holds), 0.244 against 0.250 ms per step with four.
// The kernel that also writes the finished rows needs 119 registers: four (pinned -- left to itself
// the compiler took 162, three wavefronts, and the step 0.270 ms).  float64 state: the LDS (9.7 KB
// per wavefront) admits four either way.
#ifndef TFRT_INPLACE_WAVES   // (tuning builds set it: scratch/build_variants.py)
#define TFRT_INPLACE_WAVES 5
#endif
#define TFRT_INPLACE_ATTR __attribute__((amdgpu_waves_per_eu(TFRT_INPLACE_WAVES, TFRT_INPLACE_WAVES)))
// ROWS: tfrt_scene3d.in_place == 2 (the finished rows at the rays' own columns)
template <typename T, bool ROWS>
__device__ __forceinline__ void trace_inplace(const InplaceArgs<T>& a, const BeamScene& g) {
  using RT = std::conditional_t<sizeof(T) <= 4, float, double>;
  const int lane = threadIdx.x, qwave = blockIdx.x;
  const int q = qwave * a.bundle + lane;
  const bool has = lane < a.bundle && q < a.N;
  const int64_t i = has ? q : 0;
  __shared__ BeamLds<RT> lds;
  BeamLds<RT>& W = lds;
#pragma unroll
  for (int k = 0; k < 6; ++k) W.rtab[k][lane] = static_cast<RT>(a.src[k * a.src_stride + i]);
  bool active = has;
  int skip = -1;
  int p = 0;
  static_assert(sizeof(InplaceTape) % 4 == 0 && sizeof(InplaceTape) / 4 <= 64, "one word per lane");
  const uint32_t tape_words =
      reinterpret_cast<const uint32_t*>(a.tape)[lane < (int)(sizeof(InplaceTape) / 4) ? lane : 0];
  TFRT_WAVE_BEGIN;
  unsigned work[2] = {0u, 0u};
  for (; p < a.P; ++p) {
    if (__ballot(active) == 0ull) break;  // (wave-uniform: every ray of the wavefront has ended)
    beam_pass<T, RT>(W, g, lane, a.bundle, active, skip, p == 0, /*coherent_only=*/1, TFRT_WI_ARG,
                     work);
    wave_fence();
#ifdef TFRT_BURN   // (tuning experiment: extra independent float32 FMAs per pass -- does the launch notice?)
    {
      float x0 = (float)lane, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
#pragma unroll
      for (int k = 0; k < TFRT_BURN; ++k) {
        x0 = fmaf(x0, 1.0001f, 0.5f);
        x1 = fmaf(x1, 1.0001f, 0.5f);
        x2 = fmaf(x2, 1.0001f, 0.5f);
        x3 = fmaf(x3, 1.0001f, 0.5f);
        __asm__ volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
      }
      if (x0 + x1 + x2 + x3 == 12345.678f) W.best_i[lane] = -2;
    }
#endif
    TFRT_TICK_INIT;
    // (the struct waits in ONE vector register, a word per lane, and is unpacked here, pass by
    // pass: the register is made opaque first, so that the unpacking stays below the walk instead
    // of being hoisted in front of the loop -- forty scalar registers kept across it.  Scalar
    // loads at this point were the other way to keep them out of the walk, and stalled every
    // wavefront for a memory round trip per pass)
    uint32_t tw = tape_words;
    __asm__ volatile("" : "+v"(tw));
    InplaceTape tq;
    {
      uint32_t w[sizeof(InplaceTape) / 4];
#pragma unroll
      for (int k = 0; k < (int)(sizeof(InplaceTape) / 4); ++k)
        w[k] = (uint32_t)__builtin_amdgcn_readlane((int)tw, k);
      __builtin_memcpy(&tq, w, sizeof(InplaceTape));
    }
    int cls = -1;
    if (active) {
      const int bi = W.best_i[lane];
      const double t = dkey_inv(W.best_k[lane]);
      cls = (bi < 0) ? CLS_DEAD : cat_to_cls(tq.catagory[bi]);
      const size_t at = (size_t)p * tq.n + i;
      tq.rec_tri[at] = bi;
      tq.rec_t[at] = t;
      int tape = cls;
      if (cls == CLS_ACTIVE) {
        // k_react3d's reaction: project the end onto the hit, refract / reflect (float64 Snell)
        double s[3], e[3], h[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          s[k] = static_cast<double>(W.rtab[k][lane]);
          e[k] = static_cast<double>(W.rtab[3 + k][lane]);
        }
        hit_point(s, e, t, h);
        double un[3], n1, n2, n_in;
        const double* fp = tq.fnorm + 3 * (int64_t)bi;
#pragma unroll
        for (int k = 0; k < 3; ++k) un[k] = fp[k];
        if (tq.feta != nullptr) {
          const double* fe = tq.feta + 4 * (int64_t)bi;
          n1 = fe[0];
          n2 = fe[1];
          n_in = fe[2];
        } else {
          // (feta is only absent in "index" mode with one table column per source ray)
          n_in = tq.n_table[(int64_t)tq.mat_in[bi] * tq.n_table_stride + i];
          const double n_out = tq.n_table[(int64_t)tq.mat_out[bi] * tq.n_table_stride + i];
          snell_ratios(n_in, n_out, &n1, &n2);
        }
        const Snell3 f = snell3d_core(s, h, un, n1, n2, n_in == 0.0);
        tape |= (f.nu > 0.0 ? TAPE_INTERNAL : 0) | (f.reflect ? TAPE_REFLECT : 0);
        // the child, rounded to the state type where the per-pass path stores it; it stays here
        T* out = static_cast<T*>(tq.rays_ws) + (size_t)p * 6 * tq.n;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const T hs = static_cast<T>(h[k]);
          const T es = static_cast<T>(advance(h[k], tq.L, f.w[k]));
          out[k * tq.n + i] = hs;
          out[(3 + k) * tq.n + i] = es;
          W.rtab[k][lane] = static_cast<RT>(hs);
          W.rtab[3 + k][lane] = static_cast<RT>(es);
        }
        skip = bi;
      } else {
        active = false;
        if (ROWS) {
          // this ray's row of the in-place finished block: the row a compaction would store
          // (start, hit point, rounded to the state type), or -- the ray stopped / died -- the
          // source ray itself, a finite stand-in that no gradient is ever read for
          const bool fin = cls == CLS_FINISHED;
          T* fin_rows = static_cast<T*>(tq.fin_rows);
          double h[3] = {0.0, 0.0, 0.0};
          if (fin) {
            double s[3], e[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              s[k] = static_cast<double>(W.rtab[k][lane]);
              e[k] = static_cast<double>(W.rtab[3 + k][lane]);
            }
            hit_point(s, e, t, h);
          }
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            fin_rows[k * tq.fin_cap + i] = fin ? static_cast<T>(W.rtab[k][lane]) : a.src[k * a.src_stride + i];
            fin_rows[(3 + k) * tq.fin_cap + i] = fin ? static_cast<T>(h[k]) : a.src[(3 + k) * a.src_stride + i];
          }
          tq.fin_face[i] = fin ? bi : -1;
          tq.fin_passes[i] = p + 1;
        }
      }
      tq.rec_cls[at] = (uint8_t)tape;
    }
    uint32_t word = 0u;
#pragma unroll
    for (int c = 0; c < 4; ++c) word |= (uint32_t)__popcll(__ballot(cls == c)) << (8 * c);
    if (lane == 0) tq.wcount[(size_t)p * tq.wstride + qwave] = word;
    TFRT_TICK(11);
  }
  InplaceTape te;
  {
    uint32_t w[sizeof(InplaceTape) / 4];
#pragma unroll
    for (int k = 0; k < (int)(sizeof(InplaceTape) / 4); ++k)
      w[k] = (uint32_t)__builtin_amdgcn_readlane((int)tape_words, k);
    __builtin_memcpy(&te, w, sizeof(InplaceTape));
  }
  if (ROWS && active) {   // still active after the last pass: not finished either
    T* fin_rows = static_cast<T*>(te.fin_rows);
#pragma unroll
    for (int k = 0; k < 6; ++k) fin_rows[k * te.fin_cap + i] = a.src[k * a.src_stride + i];
    te.fin_face[i] = -1;
    te.fin_passes[i] = a.P;
  }
  // (passes this wavefront never reached: no rays)
  for (int pp = p + lane; pp < a.P; pp += 64) te.wcount[(size_t)pp * te.wstride + qwave] = 0u;
  if (lane < 2) te.wcount[(size_t)(a.P + lane) * te.wstride + qwave] = work[lane];
  TFRT_WAVE_END(qwave);
}

template <typename T>
__global__ __launch_bounds__(64) TFRT_INPLACE_ATTR void k_trace_inplace(InplaceArgs<T> a, BeamScene g) {
  trace_inplace<T, false>(a, g);
}
template <typename T>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_trace_inplace_rows(
    InplaceArgs<T> a, BeamScene g) {
  trace_inplace<T, true>(a, g);
}

// The counts of an in-place trace: workgroup p sums the class counts of the passes before its own
// (the rays every class has listed so far: base_*), scans its own pass's per-wavefront counts
// (wbase: where a wavefront's rays of each class begin within the pass) and writes its row of
// `counts`; the last one also writes the trailing totals and the test count.  No workgroup waits
// for another (each reads (p + 1) x nwaves words: a few hundred KB from the L2).  Every loop keeps
// several independent loads in flight per thread: a first version that read word after word took
// 35 us for 3 x 15,625 words -- sixteen dependent round trips per thread.
__global__ __launch_bounds__(1024) void k_inplace_scan(const uint32_t* __restrict__ wcount,
                                                       int nwaves, int wstride, int P, int N, int M,
                                                       int4* __restrict__ wbase,
                                                       int32_t* __restrict__ counts) {
  const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ int wsum[16][4];
  __shared__ int red[4];
  auto unpack = [](uint32_t w, int v[4]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] += (int)((w >> (8 * c)) & 0xFFu);
  };
  // rays of each class listed by earlier passes (rows are padded to wstride words; the padding
  // is never written: masked)
  int before[4] = {0, 0, 0, 0};
  if (p > 0) {
    int acc[4] = {0, 0, 0, 0};
    for (int q = 0; q < p; ++q) {
      const uint32_t* rq = wcount + (size_t)q * wstride;
      for (int k0 = tid; k0 < nwaves; k0 += 8 * 1024) {
        uint32_t w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = (k0 + j * 1024 < nwaves) ? rq[k0 + j * 1024] : 0u;
#pragma unroll
        for (int j = 0; j < 8; ++j) unpack(w[j], acc);
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) acc[c] += __shfl_xor(acc[c], d, 64);
      if (lane == 0) wsum[wave][c] = acc[c];
    }
    __syncthreads();
    if (tid < 4) {
      int r = 0;
      for (int w = 0; w < 16; ++w) r += wsum[w][tid];
      red[tid] = r;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c) before[c] = red[c];
    __syncthreads();
  }
  // this pass: every thread takes sixteen consecutive wavefronts (four 16-byte loads), the
  // workgroup 16,384 at a time
  const uint4* row4 = reinterpret_cast<const uint4*>(wcount + (size_t)p * wstride);
  int4* out = wbase + (size_t)p * wstride;
  int tot[4] = {0, 0, 0, 0};   // of the wavefronts before this round's
  for (int base = 0; base < nwaves; base += 16 * 1024) {
    const int w0 = base + tid * 16;
    uint32_t w[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint4 v = (w0 + 4 * j < nwaves) ? row4[(w0 >> 2) + j] : make_uint4(0u, 0u, 0u, 0u);
      w[4 * j] = v.x;
      w[4 * j + 1] = v.y;
      w[4 * j + 2] = v.z;
      w[4 * j + 3] = v.w;
    }
    int mine[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (w0 + j >= nwaves) w[j] = 0u;
      unpack(w[j], mine);
    }
    int pre[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      int x = mine[c];
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(x, d, 64);
        if (lane >= d) x += o;
      }
      pre[c] = x - mine[c] + tot[c];
      if (lane == 63) wsum[wave][c] = x;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      int r = 0;
      for (int ww = 0; ww < 16; ++ww) {
        if (ww == wave) pre[c] += r;
        r += wsum[ww][c];
      }
      tot[c] += r;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (w0 + j < nwaves) out[w0 + j] = make_int4(pre[0], pre[1], pre[2], pre[3]);
      unpack(w[j], pre);
    }
    __syncthreads();   // (wsum is written again by the next round)
  }
  if (tid < 4) {
    counts[(size_t)p * TFRT_COUNTS_PER_PASS + tid] = tot[tid];
    counts[(size_t)p * TFRT_COUNTS_PER_PASS + 4 + tid] = before[tid];
  }
  if (p == P - 1) {
    int32_t* tail = counts + (size_t)P * TFRT_COUNTS_PER_PASS;
    if (tid < 4) tail[tid] = before[tid] + tot[tid];
    if (tid == 4) {
      // rays entering pass 1..P: the source, then the rays still active after each pass
      const unsigned long long tests =
          ((unsigned long long)N + (unsigned long long)before[CLS_ACTIVE]) * (unsigned long long)M;
      tail[4] = (int32_t)(uint32_t)(tests & 0xFFFFFFFFull);
      tail[5] = (int32_t)(uint32_t)(tests >> 32);
    }
  }
}

// sums of the two work rows (tfrt_trace3d_executed)
__global__ __launch_bounds__(BLOCK) void k_inplace_work(const uint32_t* __restrict__ rows, int nwaves,
                                                        int wstride, unsigned long long* out) {
  __shared__ unsigned long long part[WAVES];
  const uint32_t* r = rows + (size_t)blockIdx.x * wstride;
  unsigned long long acc = 0ull;
  for (int k = threadIdx.x; k < nwaves; k += BLOCK) acc += r[k];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
  if (lane_id() == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0ull;
    for (int w = 0; w < WAVES; ++w) t += part[w];
    out[blockIdx.x] = t;
  }
}

// Per-wavefront class counts of an in-place trace in the CALLER's ray numbering
// (tfrt_scene3d.ray_slot: the rays were handed over in another order, e.g. a coherent one): 64
// consecutive rays of the caller per wavefront, every lane walks its ray's class bytes through
// slot_of.  The scan and the gather then work in that numbering, and the ray sets come out as a
// trace of the caller's own order would list them -- the per-pass boolean_mask order of
// engine.py:2069-2111 -- without ever being listed in the order of the trace.
__global__ __launch_bounds__(64) void k_inplace_count(const uint8_t* __restrict__ rec_cls, int64_t n,
                                                      int N, int P, const int32_t* __restrict__ slot_of,
                                                      uint32_t* __restrict__ wcount, int wstride) {
  const int lane = threadIdx.x, qwave = blockIdx.x;
  const int r = qwave * 64 + lane;
  bool alive = r < N;
  const int64_t i = alive ? slot_of[r] : 0;
  int p = 0;
  for (; p < P; ++p) {
    if (__ballot(alive) == 0ull) break;
    const int cls = alive ? ((int)rec_cls[(size_t)p * n + i] & 3) : -1;
    uint32_t word = 0u;
#pragma unroll
    for (int c = 0; c < 4; ++c) word |= (uint32_t)__popcll(__ballot(cls == c)) << (8 * c);
    if (lane == 0) wcount[(size_t)p * wstride + qwave] = word;
    if (cls != CLS_ACTIVE) alive = false;
  }
  for (int pp = p + lane; pp < P; pp += 64) wcount[(size_t)pp * wstride + qwave] = 0u;
}

// The ray sets of an in-place trace in the reference's order: a wavefront's lanes walk their
// rays' records, rank themselves inside their class (ballots) behind their wavefront's base, and
// write the rows k_react3d would have written at the same slots -- recomputed from the tape, which
// holds everything a row is made of (the pass's input ray and the hit parameter).  rec_slot
// receives every record's output slot within its class (the reverse sweep reads class gradients
// through it).
template <typename T>
struct GatherArgs {
  const T* src;
  int64_t src_stride;
  int32_t N, P, bundle, nwaves;
  const T* rays_ws;
  const int32_t* rec_tri;
  const double* rec_t;
  const uint8_t* rec_cls;
  int32_t* rec_slot;
  int64_t n;
  const int4* wbase;     // [p * wstride + wavefront]
  int32_t wstride;
  // the caller's ray r sits at slot_of[r] of the trace (tfrt_scene3d.ray_slot); null: r itself.
  // With it the lanes walk the rays in the CALLER's numbering: the sets come out in the caller's
  // order with the caller's ids, and wbase / counts are those of that numbering (k_inplace_count)
  const int32_t* slot_of;
  const int32_t* counts;
  uint32_t flags;
  double dead_len;
  tfrt_ray_out fin, act, stp, dead;
  T* unfinished;
  int32_t* unfinished_id;
  int32_t* err;
};

template <typename T>
__global__ __launch_bounds__(64) void k_inplace_gather(GatherArgs<T> a) {
  const int lane = threadIdx.x, qwave = blockIdx.x;
  const int q = qwave * a.bundle + lane;
  const bool has = lane < a.bundle && q < a.N;
  const int rid = has ? q : 0;                                       // the ray as the caller numbers it
  const int64_t i = a.slot_of != nullptr ? a.slot_of[rid] : rid;     // ... and where the trace kept it
  bool alive = has, ok = true;
  int p = 0;
  for (; p < a.P; ++p) {
    if (__ballot(alive) == 0ull) break;
    const size_t at = (size_t)p * a.n + i;
    const int cls = alive ? ((int)a.rec_cls[at] & 3) : -1;
    int rank = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const unsigned long long m = __ballot(cls == c);
      if (cls == c) rank = rank_below(m);
    }
    if (alive) {
      const int4 wb = a.wbase[(size_t)p * a.wstride + qwave];
      const int in_pass = (cls == 0 ? wb.x : (cls == 1 ? wb.y : (cls == 2 ? wb.z : wb.w))) + rank;
      const int64_t gslot = (int64_t)a.counts[(size_t)p * TFRT_COUNTS_PER_PASS + 4 + cls] + in_pass;
      // Only a record whose class is compiled is read at all -- and only such a record can carry a
      // gradient, so only it needs its row in rec_slot.  With the caller's numbering every access to
      // the tape here is a scattered one (the lanes' rays lie anywhere in the trace's order): the
      // records of rays that merely went on to the next pass, two thirds of them, cost nine of those
      // each while active rays were not asked for.
      const uint32_t want = cls == CLS_DEAD       ? TFRT_COMPILE_DEAD
                            : cls == CLS_FINISHED ? TFRT_COMPILE_FINISHED
                            : cls == CLS_STOPPED  ? TFRT_COMPILE_STOPPED
                                                  : TFRT_COMPILE_ACTIVE;
      if (a.flags & want) {
        a.rec_slot[at] = (int32_t)gslot;
        const T* rin = p == 0 ? a.src : a.rays_ws + (size_t)(p - 1) * 6 * a.n;
        const int64_t sin = p == 0 ? a.src_stride : a.n;
        double s[3], e[3];
        load_ray3(rin, sin, i, s, e);
        if (cls == CLS_DEAD) {
          double e2[3] = {e[0], e[1], e[2]};
          if (a.dead_len != 0.0)
            for (int k = 0; k < 3; ++k) e2[k] = advance_between(s[k], a.dead_len, e[k]);
          ok = emit<T>(a.dead, gslot, s, e2, rid, -1) && ok;
        } else {
          const int tri = a.rec_tri[at];
          double h[3];
          hit_point(s, e, a.rec_t[at], h);
          const tfrt_ray_out& o = cls == CLS_FINISHED ? a.fin : (cls == CLS_STOPPED ? a.stp : a.act);
          ok = emit<T>(o, gslot, s, h, rid, tri) && ok;
        }
      }
      if (cls != CLS_ACTIVE) alive = false;
    }
  }
  if (!ok) atomicOr(a.err, ERR_CAPACITY);
  // the rays still active after the last pass (what single_pass returns, engine.py:2302)
  if (p == a.P && a.P > 0 && a.unfinished != nullptr) {
    const unsigned long long m = __ballot(alive);
    if (alive) {
      const int slot = a.wbase[(size_t)(a.P - 1) * a.wstride + qwave].x + rank_below(m);
      const T* rin = a.rays_ws + (size_t)(a.P - 1) * 6 * a.n;
#pragma unroll
      for (int k = 0; k < 6; ++k) a.unfinished[(int64_t)k * a.N + slot] = rin[k * a.n + i];
      if (a.unfinished_id != nullptr) a.unfinished_id[slot] = rid;
    }
  }
}

// -------------------------------------------------------------------------- backward

template <typename G>
__device__ __forceinline__ void add6(const G* g, int64_t cap, int64_t slot, double a[3],
                                     double b[3]) {
  if (g == nullptr) return;
  for (int k = 0; k < 3; ++k) {
    a[k] += static_cast<double>(g[k * cap + slot]);
    b[k] += static_cast<double>(g[(3 + k) * cap + slot]);
  }
}

// Storage type of the reverse sweep's own intermediates (the ray gradients handed from pass to
// pass and the per-ray face-gradient terms): float32 next to float32 / float16 ray state --
// the forward's hit points carry 2^-24 relative rounding there already -- float64 next to float64
// state.  All arithmetic and every sum stay float64.  (96 + 72 B per ray and pass halved: the
// sweep is bound by its memory traffic.)
template <typename T>
struct SweepStore {
  using type = float;
};
template <>
struct SweepStore<double> {
  using type = double;
};

template <typename T>
__device__ __forceinline__ int backward_ray(
    int i, const T* __restrict__ rays_in, int64_t stride_in, const int32_t* __restrict__ ray_id_in,
    const int32_t* __restrict__ rec_tri, const double* __restrict__ rec_t,
    const uint8_t* __restrict__ rec_cls, const int32_t* __restrict__ rec_slot,
    const int32_t* __restrict__ pass_counts, const tfrt_scene3d& sc, double L, double dead_len,
    const typename SweepStore<T>::type* __restrict__ g_child, int64_t child_stride,
    const double* __restrict__ g_fin, int64_t cap_fin, const double* __restrict__ g_act,
    int64_t cap_act, const double* __restrict__ g_stp, int64_t cap_stp,
    const double* __restrict__ g_dead, int64_t cap_dead,
    typename SweepStore<T>::type* __restrict__ g_out, double* __restrict__ g_src_out,
    int64_t out_stride, double gP[9]);

// (106 VGPRs = 4 waves per SIMD; the kernel is bound by float64 VALU issue -- ~2,000 executed
// instructions per ray, 28 of them divisions -- and forcing 5 or 6 waves with
// amdgpu_waves_per_eu spills: 55/66 us for the lens passes became 62/74 and 89/103)
// a float64 from another lane of the row (DPP control word CTRL), both halves
template <int CTRL>
__device__ __forceinline__ double dpp_f64(const double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xFFFFFFFFll), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Coherent rays (tfrt_scene3d.coherent_rays): the 64 rays of a wavefront hit a handful of
// faces.  Their terms are summed per face in LDS first (one slot per distinct face of the
// wavefront, LDS float64 adds), then every (face, term) sum goes to memory with ONE atomic:
// no per-ray stash written and read back, no accumulate launch.
// (several copies of every sum, taken by the lane's low bits: lanes that add to ONE LDS address
// are served one after the other -- with a single copy a third of this kernel's time -- and
// neighbouring lanes are the ones that share a face.  Eight copies for up to 8 faces, four for
// up to 16, two for up to 32: sparse rays share a face with few lanes anyway)
constexpr int WSUM_SLOTS = 32, WSUM_CELLS = 576;
__device__ __forceinline__ void wave_face_sums(int tri, const double gP[9], double* wacc,
                                               int32_t* wface, double* __restrict__ g_fverts) {
  const int lane = threadIdx.x & 63;
  int slot = -1, ns = 0;
  unsigned long long todo = __ballot(tri >= 0);
  if (todo == 0ull) return;  // (wave-uniform)
  while (todo != 0ull && ns < WSUM_SLOTS) {
    const int leader = __ffsll((long long)todo) - 1;
    const int k = __builtin_amdgcn_readlane(tri, leader);
    const bool mine = tri == k;
    if (mine) slot = ns;
    if (lane == leader) wface[ns] = k;
    ++ns;
    todo &= ~__ballot(mine);
  }
  const int copies = ns <= 8 ? 8 : (ns <= 16 ? 4 : 2);
  const int cells = ns * 9 * copies;
  for (int k = lane; k < cells; k += 64) wacc[k] = 0.0;
  wave_fence();
  if (tri >= 0) {
    if (slot >= 0) {
      double* w = &wacc[slot * 9 * copies + (lane & (copies - 1))];
#pragma unroll
      for (int c = 0; c < 9; ++c)
        if (gP[c] != 0.0) unsafeAtomicAdd(w + c * copies, gP[c]);
    } else {  // (more distinct faces than slots: rays that are not coherent after all)
      double* gp = g_fverts + 9 * (int64_t)tri;
#pragma unroll
      for (int c = 0; c < 9; ++c)
        if (gP[c] != 0.0) unsafeAtomicAdd(gp + c, gP[c]);
    }
  }
  wave_fence();
  // the copies of a sum lie in neighbouring lanes now: folded with DPP, then ONE atomic per
  // (face, term) of the wavefront goes to memory
  for (int k0 = 0; k0 < cells; k0 += 64) {
    const int k = k0 + lane;
    double v = k < cells ? wacc[k] : 0.0;
    v += dpp_f64<0xB1>(v);                    // quad_perm [1,0,3,2]
    if (copies >= 4) v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
    if (copies >= 8) v += dpp_f64<0x141>(v);  // row_half_mirror: the other quad of the eight
    if ((lane & (copies - 1)) == 0 && k < cells && v != 0.0) {
      const int term = k / copies;
      unsafeAtomicAdd(g_fverts + 9 * (int64_t)wface[term / 9] + (term % 9), v);
    }
  }
  wave_fence();  // (the next use of wacc / wface must not overtake these reads)
}

// BW wavefronts per workgroup (the wavefronts share nothing: see k_intersect_beam)
template <typename T, int BW>
__global__ __launch_bounds__(64 * BW) void k_backward3d(
    const T* __restrict__ rays_in, int64_t stride_in, const int32_t* __restrict__ n_ptr,
    const int32_t* __restrict__ ray_id_in, const int32_t* __restrict__ rec_tri,
    const double* __restrict__ rec_t, const uint8_t* __restrict__ rec_cls,
    const int32_t* __restrict__ rec_slot, const int32_t* __restrict__ pass_counts,
    tfrt_scene3d sc, double L, double dead_len,
    const typename SweepStore<T>::type* __restrict__ g_child, int64_t child_stride,
    const double* __restrict__ g_fin, int64_t cap_fin, const double* __restrict__ g_act,
    int64_t cap_act, const double* __restrict__ g_stp, int64_t cap_stp,
    const double* __restrict__ g_dead, int64_t cap_dead,
    typename SweepStore<T>::type* __restrict__ g_out, double* __restrict__ g_src_out,
    int64_t out_stride, double* __restrict__ g_fverts,
    typename SweepStore<T>::type* __restrict__ stash_g, int32_t* __restrict__ stash_face,
    int wave_sums) {
  const int n = *n_ptr;
  const int q0 = blockIdx.x * (64 * BW) + threadIdx.x;
  const int i = q0;
  if (wave_sums) {
    if ((q0 & ~63) >= n) return;  // (whole wave; the sums: wave_face_sums)
    __shared__ double wacc[BW][WSUM_CELLS];  // [slot][term][copy]
    __shared__ int32_t wface[BW][WSUM_SLOTS];
    const int wave = threadIdx.x >> 6;
    double gP[9];
    int tri = -1;
    if (i < n)
      tri = backward_ray<T>(i, rays_in, stride_in, ray_id_in, rec_tri, rec_t, rec_cls, rec_slot,
                            pass_counts, sc, L, dead_len, g_child, child_stride, g_fin, cap_fin,
                            g_act, cap_act, g_stp, cap_stp, g_dead, cap_dead, g_out, g_src_out,
                            out_stride, gP);
    wave_face_sums(tri, gP, wacc[wave], wface[wave], g_fverts);
    return;
  }
  if (i >= n) return;
  double gP[9];
  const int tri = backward_ray<T>(i, rays_in, stride_in, ray_id_in, rec_tri, rec_t, rec_cls,
                                  rec_slot, pass_counts, sc, L, dead_len, g_child, child_stride,
                                  g_fin, cap_fin, g_act, cap_act, g_stp, cap_stp, g_dead, cap_dead,
                                  g_out, g_src_out, out_stride, gP);
  if (stash_face != nullptr) {
    // face gradients are summed by k_face_accumulate: leave this ray's 9 terms and its face
    stash_face[i] = tri;
    if (tri >= 0) {
      using S = typename SweepStore<T>::type;
      S* o = stash_g + 9 * (int64_t)i;
#pragma unroll
      for (int c = 0; c < 9; ++c) o[c] = static_cast<S>(gP[c]);
    }
    return;
  }
  if (tri >= 0) {
    double* gp = g_fverts + 9 * (int64_t)tri;
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      if (gP[c] != 0.0) unsafeAtomicAdd(gp + c, gP[c]);
    }
  }
}

// (A ray whose terms are NaN / Inf poisons the sums it enters, exactly as in the reference's
// tape; SGD_Optimizer zeroes non-finite entries of the summed parameter gradient afterwards,
// optimizer.py:226-229.  `x != 0.0` is true for NaN, so the adds below let it through.)
//
// Sums the per-ray face-gradient terms left by k_backward3d into g_fverts without hammering
// memory with contended float64 atomics (9 per ray, up to thousands of rays per face: 93 % of
// the reverse sweep before).  blockIdx.y owns a window of FACE_WINDOW faces whose 9 sums live
// in LDS, blockIdx.x a chunk of ray slots; rays that hit a face of the window add their terms
// with LDS atomics, and the window is flushed once per block.  2048 faces = 144 KB of LDS, one
// block per CU: every window block re-reads its chunk's face indices and flushes 9 sums per face,
// so fewer, larger windows win (1M rays x 10,574 faces, optimiser step: 512 faces 0.815 ms,
// 1024 0.790, 2048 0.774).
constexpr int FACE_WINDOW = 2048;

template <typename S>
__global__ __launch_bounds__(1024) void k_face_accumulate(
    const int32_t* __restrict__ nrays, int passes, int64_t pass_stride,
    const int32_t* __restrict__ stash_face, const S* __restrict__ stash_g, int chunk, int M,
    double* __restrict__ g_fverts) {
  __shared__ double acc[FACE_WINDOW * 9];
  const int lo = blockIdx.x * chunk;
  if (lo >= nrays[0]) return;  // block-uniform (the ray count never grows from pass to pass)
  const int w0 = blockIdx.y * FACE_WINDOW;
  const int w1 = min(M, w0 + FACE_WINDOW);
  for (int k = threadIdx.x; k < FACE_WINDOW * 9; k += 1024) acc[k] = 0.0;
  __syncthreads();
  // All passes of the sweep in one launch: the window is cleared and flushed once per block
  // instead of once per block and pass (the flush is 9,216 global float64 atomics).
  for (int p = 0; p < passes; ++p) {
    const int hi = min(nrays[p], lo + chunk);
    const int32_t* __restrict__ face = stash_face + (int64_t)p * pass_stride;
    const S* __restrict__ terms = stash_g + 9 * (int64_t)p * pass_stride;
    // The block is a chain of dependent round trips (face -> 9 terms -> LDS add): fetch the
    // faces of four slots first, then the terms of those that fall into the window, then add.
    constexpr int U = 4;
    for (int base = lo + threadIdx.x; base < hi; base += 1024 * U) {
      int t[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = base + u * 1024;
        t[u] = i < hi ? face[i] : -1;
      }
      double g[U][9];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (t[u] >= w0 && t[u] < w1) {
          const S* src = terms + 9 * (int64_t)(base + u * 1024);
#pragma unroll
          for (int c = 0; c < 9; ++c) g[u][c] = static_cast<double>(src[c]);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (t[u] >= w0 && t[u] < w1) {
          double* a = acc + 9 * (t[u] - w0);
#pragma unroll
          for (int c = 0; c < 9; ++c)
            if (g[u][c] != 0.0) unsafeAtomicAdd(a + c, g[u][c]);
        }
      }
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < (w1 - w0) * 9; k += 1024) {
    const double v = acc[k];
    if (v != 0.0) unsafeAtomicAdd(g_fverts + 9 * (int64_t)w0 + k, v);
  }
}

// ---- ordered (deterministic) accumulation: tfrt_scene3d.deterministic
// float64 sums depend on the order of the adds; integer sums do not.  Per pass: the largest
// finite |term| of the stash fixes a power-of-two scale, every term becomes round(term * scale)
// in 64 bits (|.| <= 2^40, so 2^22 rays per face cannot overflow), integer atomics sum them, and
// k_fixed_finish adds the converted sums to g_fverts in face order.  Non-finite terms poison
// their slot (NaN), as a float sum would.
constexpr int FIXED_BITS = 40;

template <typename S>
__global__ __launch_bounds__(BLOCK) void k_stash_absmax(const int32_t* __restrict__ n_ptr,
                                                        const int32_t* __restrict__ stash_face,
                                                        const S* __restrict__ stash_g,
                                                        unsigned long long* __restrict__ maxbits) {
  const int n = *n_ptr;
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  double m = 0.0;
  if (i < n && stash_face[i] >= 0) {
    const S* g = stash_g + 9 * (int64_t)i;
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      const double a = fabs(static_cast<double>(g[c]));
      if (a > m && a < INFINITY) m = a;  // (NaN compares false)
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) m = fmax(m, __shfl_xor(m, d, 64));
  // non-negative doubles order like their bit patterns
  if (lane_id() == 0 && m > 0.0) atomicMax(maxbits, (unsigned long long)__double_as_longlong(m));
}

__device__ __forceinline__ double fixed_scale(unsigned long long maxbits) {
  if (maxbits == 0ull) return 0.0;
  int e;
  (void)frexp(__longlong_as_double((long long)maxbits), &e);  // max = f * 2^e, f in [0.5, 1)
  // |term| * scale < 2^40.  (A pass whose largest term lies below 2^-980 would need a scale
  // beyond the float64 range: the exponent is clamped, such terms round to zero -- they are
  // ~1e-295 of anything that matters.)
  int shift = FIXED_BITS - e;
  if (shift > 1000) shift = 1000;
  return ldexp(1.0, shift);
}

template <typename S>
__global__ __launch_bounds__(BLOCK) void k_face_accumulate_fixed(
    const int32_t* __restrict__ n_ptr, const int32_t* __restrict__ stash_face,
    const S* __restrict__ stash_g, const unsigned long long* __restrict__ maxbits,
    unsigned long long* __restrict__ acc, uint8_t* __restrict__ flag) {
  const int n = *n_ptr;
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const int f = stash_face[i];
  if (f < 0) return;
  const double scale = fixed_scale(*maxbits);
  const S* g = stash_g + 9 * (int64_t)i;
#pragma unroll
  for (int c = 0; c < 9; ++c) {
    const double x = static_cast<double>(g[c]);
    if (!(fabs(x) < INFINITY)) {
      flag[9 * (int64_t)f + c] = 1;  // NaN / Inf: the slot's sum is not a number
    } else {
      const long long q = llrint(x * scale);
      if (q != 0) atomicAdd(acc + 9 * (int64_t)f + c, (unsigned long long)q);  // two's complement
    }
  }
}

__global__ __launch_bounds__(BLOCK) void k_fixed_finish(int64_t m9,
                                                        const unsigned long long* __restrict__ max_this,
                                                        unsigned long long* __restrict__ max_next,
                                                        unsigned long long* __restrict__ acc,
                                                        uint8_t* __restrict__ flag,
                                                        double* __restrict__ g_fverts) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  const double scale = fixed_scale(*max_this);
  if (i < m9) {
    const long long q = (long long)acc[i];
    if (flag[i] != 0) {
      g_fverts[i] = __builtin_nan("");
    } else if (q != 0) {
      g_fverts[i] += (double)q / scale;
    }
    acc[i] = 0ull;
    flag[i] = 0;
  }
  // the other slot of the two-entry scale buffer is the next pass's: clear it here (this pass's
  // is still being read by the other workgroups of this launch)
  if (i == 0) *max_next = 0ull;
}

// Reverse of one ray slot of one pass, in registers: `child` holds the gradient w.r.t. the slot's
// child ray (start, end) when it has one; gs / ge receive the gradient w.r.t. the slot's input
// ray.  Returns the face whose gradient gP must be accumulated (-1: none).
template <typename T, bool GN = true>   // GN: the caller may want d error / d (per-face indices)
__device__ __forceinline__ int backward_core(
    int i, int tape, int slot, const T* __restrict__ rays_in, int64_t stride_in,
    const int32_t* __restrict__ ray_id_in, const int32_t* __restrict__ rec_tri,
    const double* __restrict__ rec_t, const int32_t* __restrict__ pass_counts,
    const tfrt_scene3d& sc, double L, double dead_len, bool child_pass, const double child[6],
    const double* seed_fin,   // the finished row's gradient in registers (else read from g_fin)
    const double* __restrict__ g_fin, int64_t cap_fin, const double* __restrict__ g_act,
    int64_t cap_act, const double* __restrict__ g_stp, int64_t cap_stp,
    const double* __restrict__ g_dead, int64_t cap_dead, double gs[3], double ge[3],
    double gP[9], int tri_known = -2, const double* __restrict__ feta = nullptr,
    bool slot_global = false) {
  // slot_global: `slot` is the row of the slot's output class in the whole trace (in-place traces:
  // rec_slot as k_inplace_gather left it), not the row within its pass
  // tri_known >= 0: the caller already holds the slot's face (k_backward_chain reads it on its
  // way forward); the face, the hit parameter and the indices are then asked for together with
  // the ray, before anything decides whether the slot carries a gradient at all -- one round
  // trip to memory per pass instead of three dependent ones.
  int face_out = -1;
  const int cls = tape & 3;
  double s[3], e[3];
  load_ray3(rays_in, stride_in, i, s, e);
  for (int k = 0; k < 3; ++k) gs[k] = ge[k] = 0.0;
  double P[9], t_rec = 0.0, n_in = 1.0, n_out = 1.0;
  int tri = tri_known;
  auto load_face = [&](bool with_indices) {
    const double* fp = sc.face_verts + 9 * (int64_t)tri;
#pragma unroll
    for (int q = 0; q < 9; ++q) P[q] = fp[q];
    t_rec = rec_t[i];
    if (with_indices) {
      if (feta != nullptr) {
        n_in = feta[4 * (int64_t)tri + 2];
        n_out = feta[4 * (int64_t)tri + 3];
      } else {
        face_indices(sc, tri, ray_id_in ? ray_id_in[i] : i, &n_in, &n_out);
      }
    }
  };
  const bool early = tri_known >= 0 && cls != CLS_DEAD;
  if (early) load_face(child_pass && cls == CLS_ACTIVE);

  if (cls == CLS_DEAD) {
    if (g_dead != nullptr) {
      double a[3] = {0, 0, 0}, b[3] = {0, 0, 0};
      add6(g_dead, cap_dead, slot, a, b);
      const double dl = (dead_len != 0.0) ? dead_len : 1.0;
      for (int k = 0; k < 3; ++k) {
        gs[k] = a[k] + (1.0 - dl) * b[k];
        ge[k] = dl * b[k];
      }
    }
  } else {
    double g_s[3] = {0, 0, 0}, g_h[3] = {0, 0, 0}, g_ce[3] = {0, 0, 0};
    bool has_child = false;
    if (cls == CLS_FINISHED) {
      if (seed_fin != nullptr) {
        for (int k = 0; k < 3; ++k) {
          g_s[k] += seed_fin[k];
          g_h[k] += seed_fin[3 + k];
        }
      } else {
        add6(g_fin, cap_fin, slot, g_s, g_h);
      }
    } else if (cls == CLS_STOPPED) {
      add6(g_stp, cap_stp, slot, g_s, g_h);
    } else {
      if (g_act != nullptr)
        add6(g_act, cap_act, (slot_global ? 0 : (int64_t)pass_counts[4 + CLS_ACTIVE]) + slot, g_s, g_h);
      if (child_pass) {
        has_child = true;
        for (int k = 0; k < 3; ++k) {
          g_h[k] += child[k];
          g_ce[k] += child[3 + k];
        }
      }
    }
    bool nz = has_child;
    for (int k = 0; k < 3; ++k) nz = nz || g_s[k] != 0.0 || g_h[k] != 0.0;
    if (nz) {
      if (!early) {
        tri = rec_tri[i];
        load_face(has_child);
      }
      double gn[2];
      const bool want_n = GN && has_child && sc.grad_n_in != nullptr && sc.n_table == nullptr;
      const int branch = ((tape & TAPE_INTERNAL) ? 1 : 0) | ((tape & TAPE_REFLECT) ? 2 : 0);
      adjoint3d(s, e, P, t_rec, has_child, n_in, n_out, L, g_s, g_h, g_ce, gs, ge, gP, gn, branch,
                want_n);
      if (want_n) {  // "value" mode: d error / d (per-face refractive indices)
        if (gn[0] != 0.0) unsafeAtomicAdd(sc.grad_n_in + tri, gn[0]);
        if (gn[1] != 0.0) unsafeAtomicAdd(sc.grad_n_out + tri, gn[1]);
      }
      if (sc.face_grad_mask == nullptr || sc.face_grad_mask[tri] != 0) face_out = tri;
    }
  }
  return face_out;
}

// The same with the child's gradient read from, and the slot's own written to, the sweep's
// per-pass buffers (k_backward3d).  The first pass writes to the caller's g_src, or nowhere.
template <typename T>
__device__ __forceinline__ int backward_ray(
    int i, const T* __restrict__ rays_in, int64_t stride_in, const int32_t* __restrict__ ray_id_in,
    const int32_t* __restrict__ rec_tri, const double* __restrict__ rec_t,
    const uint8_t* __restrict__ rec_cls, const int32_t* __restrict__ rec_slot,
    const int32_t* __restrict__ pass_counts, const tfrt_scene3d& sc, double L, double dead_len,
    const typename SweepStore<T>::type* __restrict__ g_child, int64_t child_stride,
    const double* __restrict__ g_fin, int64_t cap_fin, const double* __restrict__ g_act,
    int64_t cap_act, const double* __restrict__ g_stp, int64_t cap_stp,
    const double* __restrict__ g_dead, int64_t cap_dead,
    typename SweepStore<T>::type* __restrict__ g_out, double* __restrict__ g_src_out,
    int64_t out_stride, double gP[9]) {
  const int tape = rec_cls[i];
  const int slot = rec_slot[i];
  double child[6] = {0, 0, 0, 0, 0, 0};
  const bool child_pass = g_child != nullptr;
  if (child_pass && (tape & 3) == CLS_ACTIVE) {
    for (int k = 0; k < 6; ++k) child[k] = static_cast<double>(g_child[k * child_stride + slot]);
  }
  double gs[3], ge[3];
  const int face_out = backward_core<T>(i, tape, slot, rays_in, stride_in, ray_id_in, rec_tri,
                                        rec_t, pass_counts, sc, L, dead_len, child_pass, child,
                                        nullptr, g_fin, cap_fin, g_act, cap_act, g_stp, cap_stp, g_dead,
                                        cap_dead, gs, ge, gP);
  if (g_out != nullptr) {
    using G = typename SweepStore<T>::type;
    for (int k = 0; k < 3; ++k) {
      g_out[k * out_stride + i] = static_cast<G>(gs[k]);
      g_out[(3 + k) * out_stride + i] = static_cast<G>(ge[k]);
    }
  } else if (g_src_out != nullptr) {
    for (int k = 0; k < 3; ++k) {
      g_src_out[k * out_stride + i] = gs[k];
      g_src_out[(3 + k) * out_stride + i] = ge[k];
    }
  }
  return face_out;
}

// The whole reverse sweep of a coherent trace in ONE launch.  A ray's gradient only ever flows
// along the ray's own chain of slots (pass p slot -> rec_slot -> pass p + 1 slot), so a lane takes a
// SOURCE ray, follows its chain forward to the pass where the ray ended (finished / stopped / dead /
// still active after the last pass), and walks back pass by pass with the gradient w.r.t. the
// child ray in registers: the per-pass gradient blocks (2 x 24..48 B per ray and pass written
// and read back) and P - 1 dependent launches are gone, and the intermediates stay float64.
// Stable compaction keeps the slots of neighbouring lanes neighbours in every pass, so the tape
// reads stay coalesced and a wavefront's rays still share their faces (wave_face_sums, once per
// pass).  The slots of a chain wait in LDS (one column per lane) because P is not a constant.
constexpr int CHAIN_MAXP = 8;

template <typename T>
struct ChainArgs {
  const T* src;            // source rays (inputs of pass 1)
  int64_t src_stride;
  const T* rays_ws;        // inputs of pass 2..P, (P - 1) blocks of 6 x n
  const int32_t* nrays;    // rays entering pass 1..P
  const int32_t* rayid;    // per pass (from pass 2 on): source-ray index of a slot
  const int32_t* rec_tri;
  const int32_t* rec_slot;
  const double* rec_t;
  const uint8_t* rec_cls;
  const int32_t* counts;
  int64_t n;               // slot stride of the per-pass arrays
  int32_t P;
  double L, dead_len;
  const double *g_fin, *g_act, *g_stp, *g_dead;
  int64_t cap_fin, cap_act, cap_stp, cap_dead;
  double* g_src;           // (6 x N) or null
  int64_t N;
  double* g_fverts;
  // GOAL: the finished rows' gradient is that of the built-in goal error, formed here
  const T* fin_rays;       // the finished-ray block of the forward (6 x fin_cap)
  int64_t fin_cap;
  GoalFields gf;
  const double* goal;
  int64_t goal_stride, goal_ray_stride;
  double* partial;         // one partial error sum per wavefront of the launch
  int32_t* partial_cnt;    // ... and {finished rays, passes entered} per wavefront, or null
  const double* feta;      // per-face indices (FaceTables) or null
  // tape of an in-place trace (tfrt_scene3d.in_place): a ray keeps its slot through all passes.
  // 1: rec_slot holds every record's row in its output class (k_inplace_gather ran: the class
  // gradients are read through it); 2: no class gradient but the built-in goal's -- rec_slot is
  // never read, the finished row is recomputed from the tape
  int32_t inplace;
  int32_t chain_in_lds;    // the chain's records wait in LDS (P <= CHAIN_MAXP), else re-read (in place only)
};

#ifndef TFRT_CHAIN_WAVES   // (tuning builds set it: scratch/build_variants.py)
#define TFRT_CHAIN_WAVES 4
#endif
// GN = false: no gradient with respect to the per-face indices is asked for (tfrt_scene3d.grad_n_in
// null -- the usual case; the adjoint's index terms and the two atomics are not compiled in)
template <typename T, int BW, bool GOAL, bool GN>
__global__ __launch_bounds__(64 * BW)
__attribute__((amdgpu_waves_per_eu(TFRT_CHAIN_WAVES, TFRT_CHAIN_WAVES))) void k_backward_chain(
    ChainArgs<T> a, tfrt_scene3d sc) {
  // the reference's squared_difference and reduce_sum are separate ops: no contraction
#pragma clang fp contract(off)
  const int n0 = a.nrays[0];
  const int i0 = blockIdx.x * (64 * BW) + threadIdx.x;
  if ((i0 & ~63) >= n0) {  // (whole wave)
    if (GOAL && (threadIdx.x & 63) == 0) {
      a.partial[i0 >> 6] = 0.0;
      if (a.partial_cnt != nullptr) a.partial_cnt[2 * (i0 >> 6)] = a.partial_cnt[2 * (i0 >> 6) + 1] = 0;
    }
    return;
  }
  __shared__ double wacc[BW][WSUM_CELLS];
  __shared__ int32_t wface[BW][WSUM_SLOTS];
  extern __shared__ int4 chain_lds[];   // [wave][pass][lane]: slot, tape byte, output slot, face
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int P = a.P;
  int4* chain = chain_lds + (size_t)wave * P * 64;
  // forward: the slots of this ray's chain and the pass it ends in; what the walk back needs of
  // every slot's record is read here, three independent loads per pass
  int last = -1;
  if (i0 < n0) {
    int j = i0;
    for (int p = 0; p < P; ++p) {
      const size_t at = (size_t)p * a.n + j;
      const int tape = a.rec_cls[at], tri = a.rec_tri[at];
      const int slot = a.inplace >= 2 ? j : a.rec_slot[at];
      if (a.chain_in_lds) chain[p * 64 + lane] = make_int4(j, tape, slot, tri);
      last = p;
      if ((tape & 3) != CLS_ACTIVE || p == P - 1) break;
      if (!a.inplace) j = slot;
    }
  }
  // (wave-uniform bound of the walk back)
  int top = last;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) top = max(top, __shfl_xor(top, d, 64));
  double child[6] = {0, 0, 0, 0, 0, 0};
  double err = 0.0;
  int n_fin = 0, n_entered = 0;   // (wave-uniform) finished rays, passes the wavefront's rays entered
  for (int p = top; p >= 0; --p) {
    double gP[9];
    int tri = -1;
    bool fin_here = false;
    if (p <= last) {
      int4 rec;
      if (a.chain_in_lds) {
        rec = chain[p * 64 + lane];
      } else {  // (an in-place tape of more passes than the LDS columns hold: the ray stays at i0)
        const size_t at = (size_t)p * a.n + i0;
        rec = make_int4(i0, a.rec_cls[at], a.inplace >= 2 ? i0 : a.rec_slot[at], a.rec_tri[at]);
      }
      const int j = rec.x, tape = rec.y, slot = rec.z;
      const size_t off = (size_t)p * a.n;
      const T* rin = p == 0 ? a.src : a.rays_ws + (size_t)(p - 1) * 6 * a.n;
      const int64_t sin = p == 0 ? a.src_stride : a.n;
      // (in place a slot's source ray is the slot itself)
      const int32_t* idin = (p == 0 || a.inplace) ? nullptr : a.rayid + (size_t)(p - 1) * a.n;
      double seed[6] = {0, 0, 0, 0, 0, 0};
      if (GOAL && p == last && (tape & 3) == CLS_FINISHED) {
        fin_here = true;
        // tfrt_goal_error3d's terms for this ray: the output AS STORED in the finished block
        // minus the goal row of the source ray; d (sum of squares) = 2 (output - goal)
        // (in-place traces write no finished block: the row k_inplace_gather would store is
        // recomputed from the tape -- the pass's input ray and the hit parameter -- and rounded
        // to the state type like the stored one)
        double fin_row[6] = {0, 0, 0, 0, 0, 0};
        if (a.inplace) {
          double s0[3], e0[3], h0[3];
          load_ray3(rin, sin, j, s0, e0);
          hit_point(s0, e0, a.rec_t[off + j], h0);
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            fin_row[k] = s0[k];
            fin_row[3 + k] = static_cast<double>(static_cast<T>(h0[k]));
          }
        }
        for (int c = 0; c < a.gf.n; ++c) {
          const int row = a.gf.row[c];
          double out_c = 0.0;
          if (a.inplace) {
#pragma unroll
            for (int k = 0; k < 6; ++k)   // (no dynamic register index)
              if (k == row) out_c = fin_row[k];
          } else {
            out_c = ldd(a.fin_rays, (int64_t)row * a.fin_cap + slot);
          }
          const double r = out_c -
                           a.goal[(int64_t)c * a.goal_stride + (int64_t)i0 * a.goal_ray_stride];
          const double g = 2.0 * r;
#pragma unroll
          for (int q = 0; q < 6; ++q)   // (no dynamic register index)
            if (q == row) seed[q] = g;
          err += r * r;
        }
      }
      double gs[3], ge[3];
      tri = backward_core<T, GN>(j, tape, slot, rin, sin, idin, a.rec_tri + off, a.rec_t + off,
                                 a.counts + (size_t)p * TFRT_COUNTS_PER_PASS, sc, a.L, a.dead_len,
                                 p < P - 1, child, GOAL ? seed : nullptr, a.g_fin, a.cap_fin, a.g_act,
                             a.cap_act, a.g_stp, a.cap_stp, a.g_dead, a.cap_dead, gs, ge, gP,
                             rec.w, a.feta, a.inplace != 0);
      for (int k = 0; k < 3; ++k) {
        child[k] = gs[k];
        child[3 + k] = ge[k];
      }
    }
    if (GOAL) {
      n_fin += __popcll(__ballot(fin_here));
      n_entered += __popcll(__ballot(p <= last));
    }
    wave_face_sums(tri, gP, wacc[wave], wface[wave], a.g_fverts);
  }
  if (a.g_src != nullptr && i0 < n0) {
    // (a ray without a chain cannot be: every source ray enters pass 1 when P > 0)
    for (int k = 0; k < 6; ++k) a.g_src[k * a.N + i0] = child[k];
  }
  if (GOAL) {
    // fixed-shape reduction: xor butterflies inside the wave (k_goal_finish sums the partials)
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) err += __shfl_xor(err, d, 64);
    if (lane == 0) {
      a.partial[i0 >> 6] = err;
      if (a.partial_cnt != nullptr) {
        a.partial_cnt[2 * (i0 >> 6)] = n_fin;
        a.partial_cnt[2 * (i0 >> 6) + 1] = n_entered;
      }
    }
  }
}

// ------------------------------------------------------------------------------ misc

__global__ void k_init(int32_t* nrays0, int n, int32_t* tail8, unsigned int* scan_ticket) {
  if (threadIdx.x == 0) *nrays0 = n;
  if (threadIdx.x < 8) tail8[threadIdx.x] = 0;
  if (threadIdx.x == 0 && scan_ticket != nullptr) *scan_ticket = 0u;
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void k_copy_rays(const T* __restrict__ in, int64_t sin,
                                                     const int32_t* __restrict__ id_in,
                                                     const int32_t* __restrict__ n_ptr,
                                                     T* __restrict__ out, int64_t sout,
                                                     int32_t* __restrict__ id_out) {
  const int n = *n_ptr;
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  for (int k = 0; k < 6; ++k) out[k * sout + i] = in[k * sin + i];
  if (id_out) id_out[i] = id_in ? id_in[i] : i;
}

// finalize for the seam-level tfrt_intersect3d
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_finalize_seam(
    const T* __restrict__ rays, int64_t stride, int n, int chunks,
    const double* __restrict__ part_t, const int32_t* __restrict__ part_i, int64_t part_stride,
    const double* __restrict__ fverts, int M, double eps_int, double eps_size, double eps_start,
    double* x, double* y, double* z, uint8_t* valid, double* ray_u, double* trig_u,
    double* trig_v, int32_t* gather_trig) {
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  double bt = INFINITY;
  int bi = -1;
  for (int c = 0; c < chunks; ++c) {
    const double t = part_t[c * part_stride + i];
    if (t < bt) {
      bt = t;
      bi = part_i[c * part_stride + i];
    }
  }
  const int tri = bi < 0 ? 0 : bi;  // tf.argmin over an all-sentinel column returns 0
  double s[3], e[3], P[9], h[3] = {0, 0, 0};
  TriHit th;
  th.ray_u = th.trig_u = th.trig_v = 0.0;
  th.valid = false;
  if (M > 0) {
    load_ray3(rays, stride, i, s, e);
    for (int q = 0; q < 9; ++q) P[q] = fverts[9 * (int64_t)tri + q];
    th = exact_triangle(s, e, P, eps_int, eps_size, eps_start);
    hit_point(s, e, th.ray_u, h);
  }
  x[i] = h[0];
  y[i] = h[1];
  z[i] = h[2];
  valid[i] = bi >= 0;
  ray_u[i] = bi >= 0 ? th.ray_u : INFINITY;
  trig_u[i] = th.trig_u;
  trig_v[i] = th.trig_v;
  gather_trig[i] = tri;
}

// ------------------------------------------------------------------- host-side plan

struct Plan3 {
  int R, ray_blocks, chunks, chunk_faces, nblk;
  // grouped (two-level) kernel: grid and clusters per chunk
  int g_blocks, g_chunks, g_chunk_clusters;
};

// device buffers of the hierarchy for one trace (order == nullptr: all-pairs filter)
struct Accel3 {
  const int32_t* order;
  int n_clusters;
  float4* csphere;
  int32_t* cface;
  float4* clsphere;
  float4* susphere;  // one per SUPER clusters
  float4* crec;      // 3 per member: float32 face record for the screen
};

static Plan3 make_plan(int64_t N, int64_t M) {
  Plan3 p;
  p.R = (N >= 32768) ? 4 : (N >= 8192 ? 2 : 1);
  p.ray_blocks = cdiv(N > 0 ? N : 1, (int64_t)BLOCK * p.R);
  int target = 4096;  // ~16 workgroups per CU on 256 CUs (measured best: finer tail)
  int chunks = cdiv(target, p.ray_blocks);
  const int max_chunks = cdiv(M > 0 ? M : 1, 256);
  if (chunks > max_chunks) chunks = max_chunks;
  if (chunks < 1) chunks = 1;
  p.chunk_faces = cdiv(M > 0 ? M : 1, chunks);
  p.chunks = cdiv(M > 0 ? M : 1, p.chunk_faces);
  p.nblk = cdiv(N > 0 ? N : 1, BLOCK);
  // grouped kernel: level 1 is 16x shorter, so favour more workgroups over rays per lane
  const int n_clusters = cdiv(M > 0 ? M : 1, 16);
  // (one ray per lane: measured to win at every size -- 16M rays: 11.9 vs 17.7 ms for 2; more rays
  // per lane cost LDS, i.e. workgroups per CU)
  p.g_blocks = cdiv(N > 0 ? N : 1, (int64_t)BLOCK);
  // One cluster chunk (classification then happens in the kernel's epilogue: one launch less per
  // pass) from a few dozen ray blocks on; measured on the 10,574-face scene, optimiser step with
  // 1 chunk / 2048-workgroup target: 15k rays 0.255 / 0.261 ms, 60k 0.255 / 0.263, 125k 0.263 /
  // 0.263, 250k 0.307 / 0.333.  Only tiny launches spread the scene over more workgroups.
  int gtarget = 64;
  int gch = cdiv(gtarget, p.g_blocks);
  const int gmax = cdiv(n_clusters, 64);       // at least 64 clusters (1024 faces) per chunk
  if (gch > gmax) gch = gmax;
  if (gch > p.chunks) gch = p.chunks;           // part_t / part_i are sized for p.chunks
  if (gch < 1) gch = 1;
  p.g_chunk_clusters = (cdiv(n_clusters, gch) + 7) / 8 * 8;  // whole superclusters
  p.g_chunks = cdiv(n_clusters, p.g_chunk_clusters);
  return p;
}

struct Layout3 {
  size_t c0, sphere, nrays, blockcnt, blockoff, rowtot, rowbase, ticket, part_t, part_i, prep;
  size_t csphere, cface, clsphere, susphere, crec, fnorm, feta, hist_a, hist_b, left_list, wcount,
      wbase, tape_args;
  size_t rays, rayid, lasttri, rec_tri, rec_slot, rec_t, rec_cls, gbuf, stash_g, stash_face, fix_acc,
      fix_flag, fix_max, total;
};

static Layout3 make_layout(int64_t N, int64_t M, int P, int dtype, const Plan3& pl) {
  Layout3 L;
  const size_t esz = dtype == TFRT_F64 ? 8 : (dtype == TFRT_F16 ? 2 : 4);
  const size_t n = N > 0 ? N : 1, m = M > 0 ? M : 1;
  size_t o = 0;
  auto take = [&](size_t bytes) {
    size_t at = o;
    o = align_up(o + bytes);
    return at;
  };
  L.c0 = take(4 * sizeof(double));
  L.sphere = take(m * sizeof(float4));
  L.nrays = take((P + 2) * sizeof(int32_t));
  L.blockcnt = take((size_t)pl.nblk * 4 * sizeof(int32_t));
  L.blockoff = take((size_t)pl.nblk * 4 * sizeof(int32_t));
  L.rowtot = take((size_t)cdiv(pl.nblk, 1024) * 4 * sizeof(int32_t));
  L.rowbase = take((size_t)cdiv(pl.nblk, 1024) * 4 * sizeof(int32_t));
  L.ticket = take(sizeof(unsigned int));
  L.part_t = take((size_t)pl.chunks * n * sizeof(double));
  L.part_i = take((size_t)pl.chunks * n * sizeof(int32_t));
  L.prep = take((size_t)8 * n * sizeof(float));
  const size_t ncl = (m + CLUSTER - 1) / CLUSTER;
  L.csphere = take(ncl * CLUSTER * sizeof(float4));
  L.cface = take(ncl * CLUSTER * sizeof(int32_t));
  L.clsphere = take(ncl * sizeof(float4));
  L.susphere = take((ncl + SUPER - 1) / SUPER * sizeof(float4));
  L.crec = take(ncl * CLUSTER * 3 * sizeof(float4));
  L.fnorm = take(m * 3 * sizeof(double));   // the reaction's unit normal per face (snell_normal)
  L.feta = take(m * 4 * sizeof(double));    // ... and its indices and their ratios (FaceTables)
  // coherent-ray traces (tfrt_scene3d.coherent_rays): two class histograms (one being read, one
  // being built with atomics by both intersect kernels), the wavefronts left to the grouped kernel
  L.hist_a = take(((size_t)pl.nblk * 4 + 1) * sizeof(int32_t));   // (+ the count of wavefronts
  L.hist_b = take(((size_t)pl.nblk * 4 + 1) * sizeof(int32_t));   //  left to the grouped kernel)
  L.left_list = take((n + 63) / 64 * sizeof(int32_t));
  // in-place traces (tfrt_scene3d.in_place): per pass and wavefront (of 32 rays at least) the packed
  // class counts and the bases k_inplace_scan makes of them
  // (P rows of the trace's own counts, two work rows, P rows of counts in the caller's numbering)
  L.wcount = take((size_t)(2 * (P > 0 ? P : 1) + 2) * inplace_wstride(n) * sizeof(uint32_t));
  L.wbase = take((size_t)(P > 0 ? P : 1) * inplace_wstride(n) * sizeof(int4));
  L.tape_args = take(sizeof(InplaceTape));
  L.rays = take((size_t)P * 6 * n * esz);        // inputs of pass 1..P
  L.rayid = take((size_t)P * n * sizeof(int32_t));
  L.lasttri = take((size_t)P * n * sizeof(int32_t));
  L.rec_tri = take((size_t)P * n * sizeof(int32_t));
  L.rec_slot = take((size_t)P * n * sizeof(int32_t));
  L.rec_t = take((size_t)P * n * sizeof(double));
  L.rec_cls = take((size_t)P * n);
  L.gbuf = take((size_t)2 * 6 * n * sizeof(double));
  L.stash_g = take((size_t)(P > 0 ? P : 1) * 9 * n * sizeof(double));   // per pass: summed in one launch
  L.stash_face = take((size_t)(P > 0 ? P : 1) * n * sizeof(int32_t));
  L.fix_acc = take((size_t)9 * m * sizeof(unsigned long long));  // ordered accumulation
  L.fix_flag = take((size_t)9 * m);
  L.fix_max = take(2 * sizeof(unsigned long long));
  L.total = o;
  return L;
}

// ---- optional per-launch timing of the hot kernels (benchmark use only; see tfrt_profile_*)
struct ProfRec {
  hipEvent_t a, b;
  int32_t kind;  // TFRT_PROF_*
};
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
// brackets the launches made while it is alive with an event pair on the launch stream
struct ProfScope {
  ProfRec rec;
  hipStream_t st;
  bool on;
  ProfScope(int kind, hipStream_t stream) : st(stream), on(g_prof_on) {
    if (!on) return;
    rec.kind = kind;
    (void)hipEventCreate(&rec.a);
    (void)hipEventCreate(&rec.b);
    (void)hipEventRecord(rec.a, st);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(rec.b, st);
    g_prof.push_back(rec);
  }
};

// Where the grouped kernel leaves the classified hit records when it runs as one cluster chunk
// (it then does k_classify3d's work in its epilogue).
struct Classify3 {
  const int32_t* catagory = nullptr;
  int32_t* rec_tri = nullptr;
  double* rec_t = nullptr;
  uint8_t* rec_cls = nullptr;
  int32_t* blockcnt = nullptr;
};

// Coherent-ray trace (tfrt_scene3d.coherent_rays): what the two intersect kernels of a pass need.
struct Ordered3 {
  int nq = 0;                        // source rays
  int32_t* hist = nullptr;           // class histogram of this pass (atomics)
  int32_t* left_list = nullptr;      // wavefronts k_intersect_beam leaves to the grouped kernel
  int32_t* left_count = nullptr;
  int32_t* left_total = nullptr;     // ... summed over the passes of the trace (counts tail [7])
  int coherent_only = 0;             // no grouped-kernel launch: k_intersect_beam finishes all
  int n_super = 0;
};

template <typename T>
static int launch_intersect(const Plan3& pl, hipStream_t st, const T* rays, int64_t stride,
                            const int32_t* n_ptr, const int32_t* last_tri, const float4* sphere,
                            const double* fverts, const double* c0, float* prep, int64_t pstride,
                            int M, double ei, double es, double er, double* part_t,
                            int32_t* part_i, int64_t part_stride, const Accel3* ac,
                            bool prep_ready = false, const Classify3* classify = nullptr,
                            bool* classified = nullptr, const Ordered3* od = nullptr) {
  const bool grouped = ac != nullptr && ac->order != nullptr;
  Classify3 fz;
  if (grouped && pl.g_chunks == 1 && classify != nullptr && classify->rec_cls != nullptr)
    fz = *classify;
  if (classified != nullptr) *classified = fz.rec_cls != nullptr;
  // (the grouped kernel forms the filter state of a trace's first pass itself: prep = nullptr)
  const bool prep_inline = grouped && !prep_ready;
  if (!prep_ready && !prep_inline)
    hipLaunchKernelGGL((k_rayprep<T>), dim3(pl.nblk), dim3(BLOCK), 0, st, rays, stride, n_ptr, c0,
                       prep, pstride);
  dim3 grid(pl.ray_blocks, pl.chunks);
  if (grouped) grid = dim3(pl.g_blocks, pl.g_chunks);
  ProfScope prof(TFRT_PROF_INTERSECT, st);
  if (grouped && od != nullptr) {
    // coherent wavefronts first; the grouped kernel then does the wavefronts that were not
    // (half-size wavefronts while the launch would leave the chip half empty: §3.6)
    int bundle = 64;
    if (od->coherent_only && od->nq <= 160 * 1024) bundle = 32;  // (125k rays: 0.191 against 0.200 ms per step; 250k: 0.223 against 0.211)
    constexpr int BEAM_BW = 1;
    const BeamScene bs = {ac->susphere, ac->clsphere, ac->csphere, ac->crec, fverts, c0,
                          ac->n_clusters, od->n_super, ei, es, er};
    hipLaunchKernelGGL((k_intersect_beam<T, BEAM_BW>), dim3(cdiv(od->nq, BEAM_BW * bundle)),
                       dim3(64 * BEAM_BW), 0, st, rays, stride, n_ptr, last_tri, bs, fz.catagory,
                       fz.rec_tri, fz.rec_t, fz.rec_cls, od->hist, od->left_list,
                       od->left_count, od->left_total, od->coherent_only, bundle);
    // (enough workgroups to fill the chip when every wavefront is left over; they loop)
    grid = dim3(min(cdiv(od->nq, BLOCK), 1280), 1);
  }
  if (grouped && od != nullptr) {
    // (every wavefront was finished by k_intersect_beam: no launch)
    if (!od->coherent_only)
      hipLaunchKernelGGL((k_intersect_group_left<T>), grid, dim3(BLOCK), 0, st, rays, stride, n_ptr,
                         last_tri, ac->susphere, ac->clsphere, ac->csphere, ac->crec, ac->cface,
                         fverts, c0, ac->n_clusters, pl.g_chunk_clusters, ei, es, er, fz.catagory,
                         fz.rec_tri, fz.rec_t, fz.rec_cls, od->left_list, od->left_count, od->hist);
  } else if (grouped) {
    hipLaunchKernelGGL((k_intersect_group<T>), grid, dim3(BLOCK), 0, st, rays, stride, n_ptr,
                       last_tri, ac->susphere, ac->clsphere, ac->csphere, ac->crec, ac->cface,
                       fverts, c0, prep_inline ? nullptr : prep, pstride, ac->n_clusters,
                       pl.g_chunk_clusters, ei, es, er, part_t, part_i, part_stride, fz.catagory,
                       fz.rec_tri, fz.rec_t, fz.rec_cls, fz.blockcnt);
  } else {
#define TFRT_LAUNCH_R(RR)                                                                      \
    hipLaunchKernelGGL((k_intersect3d<T, RR>), grid, dim3(BLOCK), 0, st, rays, stride, n_ptr,  \
                       last_tri, sphere, fverts, prep, pstride, M, pl.chunk_faces, ei, es, er, \
                       part_t, part_i, part_stride)
    if (pl.R == 1) { TFRT_LAUNCH_R(1); }
    else if (pl.R == 4) { TFRT_LAUNCH_R(4); }
    else { TFRT_LAUNCH_R(2); }
#undef TFRT_LAUNCH_R
  }
  return 0;
}

static bool scene_ok(const tfrt_scene3d* sc) {
  if (!sc || sc->n_faces < 0) return false;
  if (sc->n_faces > 0 && (!sc->face_verts || !sc->catagory)) return false;
  if (sc->n_faces >= (1ll << 29)) return false;
  const bool index_mode = sc->mat_in && sc->mat_out && sc->n_table;
  const bool value_mode = sc->n_in && sc->n_out;
  return sc->n_faces == 0 || index_mode || value_mode;
}

// does this trace take the in-place route (tfrt_scene3d.in_place)?  The same test in the forward,
// the reverse sweep and tfrt_trace3d_compact.
static bool inplace_trace(const tfrt_scene3d* sc, int64_t N, int64_t M, int P) {
  const bool hierarchy = M >= 4 * CLUSTER && M < (1 << 24) - CLUSTER && sc->cluster_order != nullptr;
  return sc->in_place != 0 && sc->coherent_rays != 0 && sc->deterministic == 0 && hierarchy &&
         N >= 64 && P >= 1;
}

template <typename T>
static int inplace_gather_t(const void* src_rays, int64_t src_stride, int64_t N, int64_t M,
                            const int32_t* ray_slot, double dead_len, int P, uint32_t flags, const tfrt_ray_out* fin, const tfrt_ray_out* act,
                            const tfrt_ray_out* stp, const tfrt_ray_out* dead, void* unfinished,
                            int32_t* unfinished_id, int32_t* counts, char* ws, const Layout3& lay,
                            hipStream_t st) {
  const tfrt_ray_out none = {nullptr, nullptr, nullptr, 0};
  const size_t n = N > 0 ? N : 1;
  GatherArgs<T> a;
  a.src = static_cast<const T*>(src_rays);
  a.src_stride = src_stride;
  a.N = (int32_t)N;
  a.P = P;
  a.bundle = inplace_bundle(N);
  a.nwaves = cdiv(N, a.bundle);
  a.rays_ws = reinterpret_cast<const T*>(ws + lay.rays);
  a.rec_tri = reinterpret_cast<const int32_t*>(ws + lay.rec_tri);
  a.rec_t = reinterpret_cast<const double*>(ws + lay.rec_t);
  a.rec_cls = reinterpret_cast<const uint8_t*>(ws + lay.rec_cls);
  a.rec_slot = reinterpret_cast<int32_t*>(ws + lay.rec_slot);
  a.n = (int64_t)n;
  a.wbase = reinterpret_cast<const int4*>(ws + lay.wbase);
  a.wstride = (int32_t)inplace_wstride(N);
  a.counts = counts;
  a.flags = flags;
  a.dead_len = dead_len;
  a.fin = fin ? *fin : none;
  a.act = act ? *act : none;
  a.stp = stp ? *stp : none;
  a.dead = dead ? *dead : none;
  a.unfinished = static_cast<T*>(unfinished);
  a.unfinished_id = unfinished_id;
  a.err = counts + (size_t)P * TFRT_COUNTS_PER_PASS + 6;
  a.slot_of = ray_slot;
  const uint32_t* wcount = reinterpret_cast<const uint32_t*>(ws + lay.wcount);
  if (ray_slot != nullptr) {
    // the caller's numbering: wavefronts of 64 of ITS consecutive rays, counted from the tape
    // (their rows: behind the trace's own count rows and the two work rows)
    a.bundle = 64;
    a.nwaves = cdiv(N, 64);
    uint32_t* wnat = reinterpret_cast<uint32_t*>(ws + lay.wcount) + (size_t)(P + 2) * a.wstride;
    hipLaunchKernelGGL(k_inplace_count, dim3(a.nwaves), dim3(64), 0, st, a.rec_cls, a.n, (int)N, P,
                       ray_slot, wnat, a.wstride);
    wcount = wnat;
  }
  // the counts (per pass and class, bases, totals, tests) and every wavefront's bases, then the rows
  hipLaunchKernelGGL(k_inplace_scan, dim3(P), dim3(1024), 0, st, wcount, a.nwaves, a.wstride, P,
                     (int)N, (int)M, reinterpret_cast<int4*>(ws + lay.wbase), counts);
  hipLaunchKernelGGL((k_inplace_gather<T>), dim3(a.nwaves), dim3(64), 0, st, a);
  return hipGetLastError() == hipSuccess ? 0 : TFRT_E_LAUNCH;
}

template <typename T>
static int trace3d_forward_t(const void* src_rays, int64_t src_stride, int64_t N,
                             const tfrt_scene3d* sc, double L, double dead_len, int P,
                             int dtype, uint32_t flags, tfrt_ray_out* fin, tfrt_ray_out* act,
                             tfrt_ray_out* stp, tfrt_ray_out* dead, void* unfinished,
                             int32_t* unfinished_id, int32_t* counts, void* workspace,
                             size_t workspace_bytes, hipStream_t st) {
  const int M = (int)sc->n_faces;
  Plan3 pl = make_plan(N, M);
  const Layout3 lay = make_layout(N, M, P, dtype, pl);
  if (workspace_bytes < lay.total) return TFRT_E_WORKSPACE;
  char* ws = static_cast<char*>(workspace);
  double* c0 = reinterpret_cast<double*>(ws + lay.c0);
  float4* sphere = reinterpret_cast<float4*>(ws + lay.sphere);
  int32_t* nrays = reinterpret_cast<int32_t*>(ws + lay.nrays);
  int32_t* blockcnt = reinterpret_cast<int32_t*>(ws + lay.blockcnt);
  int32_t* blockoff = reinterpret_cast<int32_t*>(ws + lay.blockoff);
  int32_t* rowtot = reinterpret_cast<int32_t*>(ws + lay.rowtot);
  int32_t* rowbase = reinterpret_cast<int32_t*>(ws + lay.rowbase);
  unsigned int* ticket = reinterpret_cast<unsigned int*>(ws + lay.ticket);
  double* part_t = reinterpret_cast<double*>(ws + lay.part_t);
  int32_t* part_i = reinterpret_cast<int32_t*>(ws + lay.part_i);
  float* prep = reinterpret_cast<float*>(ws + lay.prep);
  T* rays_ws = reinterpret_cast<T*>(ws + lay.rays);
  int32_t* rayid = reinterpret_cast<int32_t*>(ws + lay.rayid);
  int32_t* lasttri = reinterpret_cast<int32_t*>(ws + lay.lasttri);
  int32_t* rec_tri = reinterpret_cast<int32_t*>(ws + lay.rec_tri);
  int32_t* rec_slot = reinterpret_cast<int32_t*>(ws + lay.rec_slot);
  double* rec_t = reinterpret_cast<double*>(ws + lay.rec_t);
  uint8_t* rec_cls = reinterpret_cast<uint8_t*>(ws + lay.rec_cls);
  FaceTables ft;
  ft.fnorm = reinterpret_cast<double*>(ws + lay.fnorm);
  // (the indices of a face do not depend on the ray: one table column, or "value" mode)
  const bool index_mode = sc->n_table != nullptr && sc->mat_in != nullptr;
  if (index_mode ? sc->n_table_uniform != 0 : (sc->n_in != nullptr && sc->n_out != nullptr))
    ft.feta = reinterpret_cast<double*>(ws + lay.feta);
  ft.mat_in = sc->mat_in;
  ft.mat_out = sc->mat_out;
  ft.n_table = sc->n_table;
  ft.n_table_stride = sc->n_table_stride;
  ft.n_in = sc->n_in;
  ft.n_out = sc->n_out;
  int32_t* tail = counts + (size_t)P * TFRT_COUNTS_PER_PASS;
  const size_t n = N > 0 ? N : 1;

  if (M <= 0) hipLaunchKernelGGL(k_init, dim3(1), dim3(64), 0, st, nrays, (int)N, tail, ticket);
  if (M <= 0 && sc->clear_buffer != nullptr && sc->clear_count > 0)   // (no set-up launch to do it)
    (void)hipMemsetAsync(sc->clear_buffer, 0, (size_t)sc->clear_count * sizeof(double), st);
  Accel3 ac;
  // (the grouped kernel packs member slot and ray slot into 32 bits: member slots < 2^24)
  ac.order = (M >= 4 * CLUSTER && M < (1 << 24) - CLUSTER) ? sc->cluster_order : nullptr;
  ac.n_clusters = cdiv(M > 0 ? M : 1, CLUSTER);
  ac.csphere = reinterpret_cast<float4*>(ws + lay.csphere);
  ac.cface = reinterpret_cast<int32_t*>(ws + lay.cface);
  ac.clsphere = reinterpret_cast<float4*>(ws + lay.clsphere);
  ac.susphere = reinterpret_cast<float4*>(ws + lay.susphere);
  ac.crec = reinterpret_cast<float4*>(ws + lay.crec);
  // Coherent rays: wavefronts take k_intersect_beam's shared walk first; one cluster chunk,
  // classification in the kernels' epilogues
  const bool coherent = sc->coherent_rays != 0 && ac.order != nullptr && N >= 64;
  int32_t* hist_ab[2] = {reinterpret_cast<int32_t*>(ws + lay.hist_a),
                         reinterpret_cast<int32_t*>(ws + lay.hist_b)};
  if (coherent) {
    pl.g_blocks = cdiv(N, (int64_t)BLOCK);
    pl.g_chunks = 1;
    pl.g_chunk_clusters = (ac.n_clusters + 7) / 8 * 8;
  }
  const bool inplace = inplace_trace(sc, N, M, P);
  const bool rows_in_place = inplace && sc->in_place == 2;
  InplaceTape tape = {};
  if (inplace) {
    tape.rays_ws = rays_ws;
    tape.rec_tri = rec_tri;
    tape.rec_t = rec_t;
    tape.rec_cls = rec_cls;
    tape.n = (int64_t)n;
    tape.wcount = reinterpret_cast<uint32_t*>(ws + lay.wcount);
    tape.wstride = (int64_t)inplace_wstride(N);
    tape.catagory = sc->catagory;
    tape.fnorm = ft.fnorm;
    tape.feta = ft.feta;
    tape.n_table = sc->n_table;
    tape.mat_in = sc->mat_in;
    tape.mat_out = sc->mat_out;
    tape.n_table_stride = sc->n_table_stride;
    tape.L = L;
    if (rows_in_place) {
      // the finished rows at the rays' own columns: nothing is compacted, nothing else is written
      if (!fin || !fin->rays || !fin->face || !fin->ray_id || fin->capacity < N ||
          (act && act->rays) || (stp && stp->rays) || (dead && dead->rays) || unfinished != nullptr)
        return TFRT_E_BADARG;
      tape.fin_rows = fin->rays;
      tape.fin_cap = fin->capacity;
      tape.fin_face = fin->face;
      tape.fin_passes = fin->ray_id;
    }
  }
  if (M > 0) {
    if (ac.order != nullptr) {  // (the hierarchy kernel also does k_center's work)
      const int cl_blocks = cdiv((int64_t)ac.n_clusters * CLUSTER, BLOCK);
      const int n_super = cdiv(ac.n_clusters, SUPER);
      hipLaunchKernelGGL(k_hierarchy_spheres, dim3(n_super + cl_blocks), dim3(BLOCK), 0, st,
                         sc->face_verts, M, ac.order, c0, sc->size_epsilion, ac.n_clusters,
                         n_super, ac.csphere, ac.cface, ac.clsphere, ac.crec, ac.susphere, nrays,
                         (int)N, tail, ticket, hist_ab[0],
                         (coherent && !inplace) ? pl.nblk * 4 + 1 : 0, ft,
                         sc->clear_buffer, sc->clear_buffer ? sc->clear_count : 0, tape,
                         inplace ? reinterpret_cast<InplaceTape*>(ws + lay.tape_args) : nullptr);
    } else {
      hipLaunchKernelGGL(k_center, dim3(1), dim3(BLOCK), 0, st, sc->face_verts, M, c0, nrays,
                         (int)N, tail, ticket);
      hipLaunchKernelGGL(k_spheres, dim3(cdiv(M, BLOCK)), dim3(BLOCK), 0, st, sc->face_verts, M,
                         c0, sc->size_epsilion, sphere, ft, sc->clear_buffer,
                         sc->clear_buffer ? sc->clear_count : 0);
    }
  }
  const tfrt_ray_out none = {nullptr, nullptr, nullptr, 0};
  if (inplace) {
    // every pass in one launch, rays in place; then the counts, then (only if asked) the ray sets
    InplaceArgs<T> a;
    a.src = static_cast<const T*>(src_rays);
    a.src_stride = src_stride;
    a.N = (int32_t)N;
    a.P = P;
    a.bundle = inplace_bundle(N);
    a.nwaves = cdiv(N, a.bundle);
    a.tape = reinterpret_cast<const InplaceTape*>(ws + lay.tape_args);
    const BeamScene bs = {ac.susphere, ac.clsphere, ac.csphere, ac.crec, sc->face_verts, c0,
                          ac.n_clusters, cdiv(ac.n_clusters, SUPER), sc->intersect_epsilion,
                          sc->size_epsilion, sc->ray_start_epsilion};
    {
      ProfScope prof(TFRT_PROF_INTERSECT, st);
      if (rows_in_place)
        hipLaunchKernelGGL((k_trace_inplace_rows<T>), dim3(a.nwaves), dim3(64), 0, st, a, bs);
      else
        hipLaunchKernelGGL((k_trace_inplace<T>), dim3(a.nwaves), dim3(64), 0, st, a, bs);
    }
    // (no room for ray sets: no scan either -- tfrt_trace3d_compact makes counts and sets later)
    const bool want_rows = !rows_in_place &&
                           ((fin && fin->rays) || (act && act->rays) || (stp && stp->rays) ||
                            (dead && dead->rays) || unfinished != nullptr);
    if (want_rows)
      return inplace_gather_t<T>(src_rays, src_stride, N, M, sc->ray_slot, dead_len, P, flags, fin,
                                 act, stp, dead, unfinished, unfinished_id, counts, ws, lay, st);
    return hipGetLastError() == hipSuccess ? 0 : TFRT_E_LAUNCH;
  }
  const int chunks_used = ac.order == nullptr ? pl.chunks : pl.g_chunks;
  for (int p = 0; p < P; ++p) {
    const T* rin = p == 0 ? static_cast<const T*>(src_rays) : rays_ws + (size_t)(p - 1) * 6 * n;
    const int64_t sin = p == 0 ? src_stride : (int64_t)n;
    const int32_t* idin = p == 0 ? nullptr : rayid + (size_t)(p - 1) * n;
    const int32_t* ltin = p == 0 ? nullptr : lasttri + (size_t)(p - 1) * n;
    T* rout = rays_ws + (size_t)p * 6 * n;
    Classify3 fz;
    fz.catagory = sc->catagory;
    fz.rec_tri = rec_tri + (size_t)p * n;
    fz.rec_t = rec_t + (size_t)p * n;
    fz.rec_cls = rec_cls + (size_t)p * n;
    fz.blockcnt = blockcnt;
    bool classified = false;
    Ordered3 od;
    if (coherent) {
      od.nq = (int)N;
      od.hist = hist_ab[p & 1];
      od.left_list = reinterpret_cast<int32_t*>(ws + lay.left_list);
      od.left_count = od.hist + (size_t)pl.nblk * 4;
      od.left_total = tail + 7;
      od.coherent_only = sc->coherent_only != 0;
      od.n_super = cdiv(ac.n_clusters, SUPER);
      fz.blockcnt = od.hist;
    }
    if (launch_intersect<T>(pl, st, rin, sin, nrays + p, ltin, sphere, sc->face_verts, c0, prep,
                            (int64_t)n, M, sc->intersect_epsilion, sc->size_epsilion,
                            sc->ray_start_epsilion, part_t, part_i, (int64_t)n, &ac,
                            /*prep_ready=*/p > 0, &fz, &classified,
                            coherent ? &od : nullptr) != 0)
      return TFRT_E_LAUNCH;
    int32_t* blockcnt_p = coherent ? od.hist : blockcnt;
    if (!classified)
      hipLaunchKernelGGL(k_classify3d, dim3(pl.nblk), dim3(BLOCK), 0, st, nrays + p, chunks_used,
                         part_t, part_i, (int64_t)n, sc->catagory, rec_tri + (size_t)p * n,
                         rec_t + (size_t)p * n, rec_cls + (size_t)p * n, blockcnt_p);
    const bool grid_scan = pl.nblk >= SCAN_GRID_MIN_ROWS;
    SelfScan ss;
    if (coherent) {
      ss.hist_next = hist_ab[(p + 1) & 1];
      ss.hist_rows = pl.nblk;
    }
    if (pl.nblk <= SELF_SCAN_MAX_BLOCKS) {
      ss.blockcnt = blockcnt_p;
      ss.prev_counts = p > 0 ? counts + (size_t)(p - 1) * TFRT_COUNTS_PER_PASS : nullptr;
      ss.counts_row = counts + (size_t)p * TFRT_COUNTS_PER_PASS;
      ss.totals = tail;
      ss.n_next = nrays + p + 1;
      ss.n_tests = reinterpret_cast<unsigned long long*>(tail + 4);
      ss.M = M;
    } else if (grid_scan)
      hipLaunchKernelGGL(k_scan3d, dim3(cdiv(pl.nblk, 1024)), dim3(1024), 0, st, nrays + p,
                         blockcnt_p, blockoff, rowtot, rowbase, ticket,
                         counts + (size_t)p * TFRT_COUNTS_PER_PASS, tail, nrays + p + 1,
                         reinterpret_cast<unsigned long long*>(tail + 4), M);
    else
      hipLaunchKernelGGL(k_scan3d_one, dim3(1), dim3(1024), 0, st, nrays + p, blockcnt_p, blockoff,
                         counts + (size_t)p * TFRT_COUNTS_PER_PASS, tail, nrays + p + 1,
                         reinterpret_cast<unsigned long long*>(tail + 4), M);
    ProfScope prof_react(TFRT_PROF_REACT, st);
    hipLaunchKernelGGL((k_react3d<T>), dim3(pl.nblk), dim3(BLOCK), 0, st, rin, sin, nrays + p,
                       idin, rec_tri + (size_t)p * n, rec_t + (size_t)p * n,
                       rec_cls + (size_t)p * n, blockoff,
                       grid_scan ? rowbase : static_cast<int32_t*>(nullptr),
                       counts + (size_t)p * TFRT_COUNTS_PER_PASS, *sc, L, dead_len, flags, rout,
                       (int64_t)n, rayid + (size_t)p * n, lasttri + (size_t)p * n,
                       rec_slot + (size_t)p * n, fin ? *fin : none, act ? *act : none,
                       stp ? *stp : none, dead ? *dead : none, tail + 6,
                       (p + 1 < P && !coherent) ? prep : nullptr,
                       (int64_t)n, c0, ss, ft.fnorm, ft.feta);
  }
  if (unfinished != nullptr && P > 0) {
    hipLaunchKernelGGL((k_copy_rays<T>), dim3(pl.nblk), dim3(BLOCK), 0, st,
                       rays_ws + (size_t)(P - 1) * 6 * n, (int64_t)n, rayid + (size_t)(P - 1) * n,
                       nrays + P, static_cast<T*>(unfinished), (int64_t)N, unfinished_id);
  }
  return hipGetLastError() == hipSuccess ? 0 : TFRT_E_LAUNCH;
}

// the built-in goal error folded into the sweep (tfrt_trace3d_backward_goal)
struct ChainGoal {
  const void* fin_rays;
  int64_t fin_cap;
  GoalFields gf;
  const double* goal;
  int64_t goal_stride, goal_ray_stride;
  double* partial;
  int32_t* partial_cnt;
};

template <typename T>
static int trace3d_backward_t(const void* src_rays, int64_t src_stride, int64_t N,
                              const tfrt_scene3d* sc, double L, double dead_len, int P, int dtype,
                              const double* g_fin, int64_t cap_fin, const double* g_act,
                              int64_t cap_act, const double* g_stp, int64_t cap_stp,
                              const double* g_dead, int64_t cap_dead, double* g_fverts,
                              double* g_src, const int32_t* counts, void* workspace,
                              size_t workspace_bytes, hipStream_t st,
                              const ChainGoal* goal = nullptr) {
  const int M = (int)sc->n_faces;
  const Plan3 pl = make_plan(N, M);
  const Layout3 lay = make_layout(N, M, P, dtype, pl);
  if (workspace_bytes < lay.total) return TFRT_E_WORKSPACE;
  char* ws = static_cast<char*>(workspace);
  const int32_t* nrays = reinterpret_cast<int32_t*>(ws + lay.nrays);
  const T* rays_ws = reinterpret_cast<T*>(ws + lay.rays);
  const int32_t* rayid = reinterpret_cast<int32_t*>(ws + lay.rayid);
  const int32_t* rec_tri = reinterpret_cast<int32_t*>(ws + lay.rec_tri);
  const int32_t* rec_slot = reinterpret_cast<int32_t*>(ws + lay.rec_slot);
  const double* rec_t = reinterpret_cast<double*>(ws + lay.rec_t);
  const uint8_t* rec_cls = reinterpret_cast<uint8_t*>(ws + lay.rec_cls);
  using G = typename SweepStore<T>::type;  // (the regions are sized for float64)
  G* gbuf = reinterpret_cast<G*>(ws + lay.gbuf);
  const size_t n = N > 0 ? N : 1;
  // Windowed LDS accumulation of the face gradients (k_face_accumulate): every window block
  // scans its chunk's face ids, so it is used while the windows are few; beyond that (and for
  // small ray counts) k_backward3d adds straight into g_fverts with float64 atomics -- faces are
  // then so many that they see little contention.
  const int windows = cdiv(M > 0 ? M : 1, FACE_WINDOW);
  const bool ordered = sc->deterministic != 0 && M > 0 && g_fverts != nullptr;
  // (coherent rays: the wavefront's rays hit few faces; k_backward3d sums them itself)
  const bool wave_sums = !ordered && sc->coherent_rays != 0 && M > 0 && g_fverts != nullptr;
  const bool stash = ordered || (!wave_sums && M > 0 && N >= 16384 && windows <= 32);
  unsigned long long* fix_acc = reinterpret_cast<unsigned long long*>(ws + lay.fix_acc);
  uint8_t* fix_flag = reinterpret_cast<uint8_t*>(ws + lay.fix_flag);
  unsigned long long* fix_max = reinterpret_cast<unsigned long long*>(ws + lay.fix_max);
  if (ordered)  // (fix_acc, fix_flag and fix_max are adjacent: one clear)
    (void)hipMemsetAsync(fix_acc, 0, lay.total - lay.fix_acc, st);
  G* stash_g_all = reinterpret_cast<G*>(ws + lay.stash_g);
  int32_t* stash_face_all = reinterpret_cast<int32_t*>(ws + lay.stash_face);
  // ray slots per accumulate block, measured at 1M rays x 11 windows (us for the three passes,
  // target pass first): 2048 -> 37/50/54, 4096 -> 21/33/38, 8192 -> 15/37/34, 16384 -> 12/49/36.
  // Every block zeroes and flushes its window, so big chunks win while the blocks still fill
  // the chip (two 1024-thread blocks per CU).
  int acc_chunk = 8192;
  // (one launch sums all passes now: a block per CU is enough -- 125k rays x 11 windows, step time
  // with 1024 / 2048 / 4096 / 8192 slots per block: 0.302 / 0.278 / 0.265 / 0.268 ms)
  while (acc_chunk > 1024 && (int64_t)cdiv(N, acc_chunk) * windows < 256) acc_chunk /= 2;
  const bool inplace = inplace_trace(sc, N, M, P);
  if (inplace && !wave_sums) return TFRT_E_UNSUPPORTED;   // (deterministic: not with in_place)
  if ((wave_sums && P >= 1 && (P <= CHAIN_MAXP || inplace)) || goal != nullptr) {
    // coherent rays: the whole sweep in one launch (k_backward_chain)
    if (!(wave_sums && P >= 1 && (P <= CHAIN_MAXP || inplace))) return TFRT_E_UNSUPPORTED;
    ChainArgs<T> a;
    a.src = static_cast<const T*>(src_rays);
    a.src_stride = src_stride;
    a.rays_ws = rays_ws;
    a.nrays = nrays;
    a.rayid = rayid;
    a.rec_tri = rec_tri;
    a.rec_slot = rec_slot;
    a.rec_t = rec_t;
    a.rec_cls = rec_cls;
    a.counts = counts;
    a.n = (int64_t)n;
    a.P = P;
    a.L = L;
    a.dead_len = dead_len;
    a.g_fin = g_fin;
    a.g_act = g_act;
    a.g_stp = g_stp;
    a.g_dead = g_dead;
    a.cap_fin = cap_fin;
    a.cap_act = cap_act;
    a.cap_stp = cap_stp;
    a.cap_dead = cap_dead;
    a.g_src = g_src;
    a.N = N;
    a.g_fverts = g_fverts;
    // (in-place tape: class gradients are read through rec_slot, which k_inplace_gather fills;
    // a sweep that is handed none but the built-in goal's never reads it)
    // 3 (tfrt_scene3d.in_place == 2): the finished rows' gradient sits at the rays' own columns,
    // no other class carries one -- like 2, rec_slot is never read
    if (inplace && sc->in_place == 2 && (g_act || g_stp || g_dead || goal != nullptr || cap_fin < N))
      return TFRT_E_BADARG;
    a.inplace = !inplace ? 0
                : (sc->in_place == 2 ? 3
                   : ((g_fin || g_act || g_stp || g_dead || goal == nullptr) ? 1 : 2));
    a.chain_in_lds = P <= CHAIN_MAXP ? 1 : 0;
    a.fin_rays = nullptr;
    a.fin_cap = 0;
    a.gf.n = 0;
    a.goal = nullptr;
    a.goal_stride = a.goal_ray_stride = 0;
    a.partial = nullptr;
    a.partial_cnt = nullptr;
    {  // (the per-face indices the forward's set-up launch left, under the same condition)
      const bool index_mode = sc->n_table != nullptr && sc->mat_in != nullptr;
      const bool per_face = index_mode ? sc->n_table_uniform != 0
                                       : (sc->n_in != nullptr && sc->n_out != nullptr);
      a.feta = per_face ? reinterpret_cast<const double*>(ws + lay.feta) : nullptr;
    }
    const size_t chain_lds = a.chain_in_lds ? (size_t)P * 64 * sizeof(int4) : 0;
    ProfScope prof_bwd(TFRT_PROF_BACKWARD, st);
    if (goal != nullptr) {
      a.fin_rays = static_cast<const T*>(goal->fin_rays);
      a.fin_cap = goal->fin_cap;
      a.gf = goal->gf;
      a.goal = goal->goal;
      a.goal_stride = goal->goal_stride;
      a.goal_ray_stride = goal->goal_ray_stride;
      a.partial = goal->partial;
      a.partial_cnt = goal->partial_cnt;
      if (N > 0 && sc->grad_n_in != nullptr)
        hipLaunchKernelGGL((k_backward_chain<T, 1, true, true>), dim3(cdiv(N, 64)), dim3(64),
                           chain_lds, st, a, *sc);
      else if (N > 0)
        hipLaunchKernelGGL((k_backward_chain<T, 1, true, false>), dim3(cdiv(N, 64)), dim3(64),
                           chain_lds, st, a, *sc);
    } else if (N > 0 && sc->grad_n_in != nullptr) {
      hipLaunchKernelGGL((k_backward_chain<T, 1, false, true>), dim3(cdiv(N, 64)), dim3(64),
                         chain_lds, st, a, *sc);
    } else if (N > 0) {
      hipLaunchKernelGGL((k_backward_chain<T, 1, false, false>), dim3(cdiv(N, 64)), dim3(64),
                         chain_lds, st, a, *sc);
    }
    return hipGetLastError() == hipSuccess ? 0 : TFRT_E_LAUNCH;
  }
  for (int p = P - 1; p >= 0; --p) {
    const T* rin = p == 0 ? static_cast<const T*>(src_rays) : rays_ws + (size_t)(p - 1) * 6 * n;
    const int64_t sin = p == 0 ? src_stride : (int64_t)n;
    const int32_t* idin = p == 0 ? nullptr : rayid + (size_t)(p - 1) * n;
    const G* g_child = (p == P - 1) ? nullptr : gbuf + (size_t)((p + 1) & 1) * 6 * n;
    // (the first pass's ray gradient goes to the caller's g_src, or nowhere)
    G* g_out = p == 0 ? nullptr : gbuf + (size_t)(p & 1) * 6 * n;
    double* g_src_out = p == 0 ? g_src : nullptr;
    const int64_t out_stride = p == 0 ? N : (int64_t)n;
    G* stash_g = stash_g_all + (size_t)p * 9 * n;
    int32_t* stash_face = stash_face_all + (size_t)p * n;
    ProfScope prof_bwd(TFRT_PROF_BACKWARD, st);
    constexpr int BWD_BW = 1;
    hipLaunchKernelGGL((k_backward3d<T, BWD_BW>), dim3(cdiv(N, 64 * BWD_BW)), dim3(64 * BWD_BW), 0,
                       st, rin, sin, nrays + p,
                       idin, rec_tri + (size_t)p * n, rec_t + (size_t)p * n,
                       rec_cls + (size_t)p * n, rec_slot + (size_t)p * n,
                       counts + (size_t)p * TFRT_COUNTS_PER_PASS, *sc, L, dead_len, g_child,
                       (int64_t)n, g_fin, cap_fin, g_act, cap_act, g_stp, cap_stp, g_dead,
                       cap_dead, g_out, g_src_out, out_stride, g_fverts,
                       stash ? stash_g : nullptr, stash ? stash_face : nullptr, wave_sums ? 1 : 0);
    if (ordered) {
      // two-entry scale buffer, alternating per pass (each pass's conversion clears the other)
      unsigned long long* mx = fix_max + ((P - 1 - p) & 1);
      hipLaunchKernelGGL((k_stash_absmax<G>), dim3(pl.nblk), dim3(BLOCK), 0, st, nrays + p, stash_face,
                         stash_g, mx);
      hipLaunchKernelGGL((k_face_accumulate_fixed<G>), dim3(pl.nblk), dim3(BLOCK), 0, st, nrays + p,
                         stash_face, stash_g, mx, fix_acc, fix_flag);
      hipLaunchKernelGGL(k_fixed_finish, dim3(cdiv((int64_t)M * 9, BLOCK)), dim3(BLOCK), 0, st,
                         (int64_t)M * 9, mx, fix_max + (((P - 1 - p) & 1) ^ 1), fix_acc, fix_flag,
                         g_fverts);
    }
  }
  if (stash && !ordered && P > 0) {
    ProfScope prof_acc(TFRT_PROF_ACCUMULATE, st);
    hipLaunchKernelGGL((k_face_accumulate<G>), dim3(cdiv(N, acc_chunk), windows), dim3(1024), 0, st,
                       nrays, P, (int64_t)n, stash_face_all, stash_g_all, acc_chunk, M, g_fverts);
  }
  return hipGetLastError() == hipSuccess ? 0 : TFRT_E_LAUNCH;
}

}  // namespace tfrt

// ================================================================================ C ABI
using namespace tfrt;

extern "C" {

size_t tfrt_trace3d_workspace_bytes(int64_t n_rays, int64_t n_faces, int32_t max_passes,
                                    int32_t state_dtype) {
  if (n_rays < 0 || n_faces < 0 || max_passes < 0) return 0;
  const Plan3 pl = make_plan(n_rays, n_faces);
  return make_layout(n_rays, n_faces, max_passes, state_dtype, pl).total;
}

int tfrt_trace3d_forward(const void* src_rays, int64_t src_stride, int64_t n_rays,
                         const tfrt_scene3d* scene, double new_ray_length,
                         double dead_ray_length, int32_t max_passes, int32_t state_dtype,
                         uint32_t flags, tfrt_ray_out* finished, tfrt_ray_out* active,
                         tfrt_ray_out* stopped, tfrt_ray_out* dead, void* unfinished,
                         int32_t* unfinished_id, int32_t* counts, void* workspace,
                         size_t workspace_bytes, void* stream) {
  if (!scene_ok(scene) || n_rays < 0 || n_rays >= (1ll << 31) - 4096 || max_passes < 0 ||
      !counts || !workspace || (n_rays > 0 && !src_rays) || src_stride < n_rays)
    return TFRT_E_BADARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (state_dtype == TFRT_F32)
    return trace3d_forward_t<float>(src_rays, src_stride, n_rays, scene, new_ray_length,
                                    dead_ray_length, max_passes, state_dtype, flags, finished,
                                    active, stopped, dead, unfinished, unfinished_id, counts,
                                    workspace, workspace_bytes, st);
  if (state_dtype == TFRT_F64)
    return trace3d_forward_t<double>(src_rays, src_stride, n_rays, scene, new_ray_length,
                                     dead_ray_length, max_passes, state_dtype, flags, finished,
                                     active, stopped, dead, unfinished, unfinished_id, counts,
                                     workspace, workspace_bytes, st);
  if (state_dtype == TFRT_F16)
    return trace3d_forward_t<_Float16>(src_rays, src_stride, n_rays, scene, new_ray_length,
                                     dead_ray_length, max_passes, state_dtype, flags, finished,
                                     active, stopped, dead, unfinished, unfinished_id, counts,
                                     workspace, workspace_bytes, st);
  return TFRT_E_UNSUPPORTED;
}

int tfrt_trace3d_in_place(const tfrt_scene3d* scene, int64_t n_rays, int32_t max_passes) {
  if (!scene_ok(scene) || n_rays < 0 || max_passes < 0) return TFRT_E_BADARG;
  return inplace_trace(scene, n_rays, scene->n_faces, max_passes) ? 1 : 0;
}

int tfrt_trace3d_executed(int64_t n_rays, int64_t n_faces, int32_t max_passes, int32_t state_dtype,
                          const void* workspace, size_t workspace_bytes, int64_t* executed,
                          void* stream) {
  if (n_rays < 64 || n_faces < 0 || max_passes < 1 || !workspace || !executed) return TFRT_E_BADARG;
  const Plan3 pl = make_plan(n_rays, n_faces);
  const Layout3 lay = make_layout(n_rays, n_faces, max_passes, state_dtype, pl);
  if (workspace_bytes < lay.total) return TFRT_E_WORKSPACE;
  const int wstride = (int)inplace_wstride(n_rays);
  const uint32_t* rows = reinterpret_cast<const uint32_t*>(static_cast<const char*>(workspace) +
                                                           lay.wcount) + (size_t)max_passes * wstride;
  hipLaunchKernelGGL(k_inplace_work, dim3(2), dim3(BLOCK), 0, static_cast<hipStream_t>(stream), rows,
                     cdiv(n_rays, inplace_bundle(n_rays)), wstride,
                     reinterpret_cast<unsigned long long*>(executed));
  return hipGetLastError() == hipSuccess ? 0 : TFRT_E_LAUNCH;
}

int tfrt_trace3d_compact(const void* src_rays, int64_t src_stride, int64_t n_rays,
                         double dead_ray_length, int32_t max_passes, int32_t state_dtype,
                         uint32_t flags, tfrt_ray_out* finished, tfrt_ray_out* active,
                         tfrt_ray_out* stopped, tfrt_ray_out* dead, void* unfinished,
                         int32_t* unfinished_id, int32_t* counts, int64_t n_faces,
                         const int32_t* ray_slot, void* workspace, size_t workspace_bytes,
                         void* stream) {
  if (n_rays < 64 || n_rays >= (1ll << 31) - 4096 || max_passes < 1 || !counts || !workspace ||
      !src_rays || src_stride < n_rays || n_faces < 0)
    return TFRT_E_BADARG;
  const Plan3 pl = make_plan(n_rays, n_faces);
  const Layout3 lay = make_layout(n_rays, n_faces, max_passes, state_dtype, pl);
  if (workspace_bytes < lay.total) return TFRT_E_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* ws = static_cast<char*>(workspace);
  if (state_dtype == TFRT_F32)
    return inplace_gather_t<float>(src_rays, src_stride, n_rays, n_faces, ray_slot, dead_ray_length, max_passes, flags,
                                   finished, active, stopped, dead, unfinished, unfinished_id,
                                   counts, ws, lay, st);
  if (state_dtype == TFRT_F64)
    return inplace_gather_t<double>(src_rays, src_stride, n_rays, n_faces, ray_slot, dead_ray_length, max_passes,
                                    flags, finished, active, stopped, dead, unfinished,
                                    unfinished_id, counts, ws, lay, st);
  if (state_dtype == TFRT_F16)
    return inplace_gather_t<_Float16>(src_rays, src_stride, n_rays, n_faces, ray_slot, dead_ray_length, max_passes,
                                      flags, finished, active, stopped, dead, unfinished,
                                      unfinished_id, counts, ws, lay, st);
  return TFRT_E_UNSUPPORTED;
}

int tfrt_trace3d_backward(const void* src_rays, int64_t src_stride, int64_t n_rays,
                          const tfrt_scene3d* scene, double new_ray_length,
                          double dead_ray_length, int32_t max_passes, int32_t state_dtype,
                          const double* grad_finished, int64_t cap_finished,
                          const double* grad_active, int64_t cap_active,
                          const double* grad_stopped, int64_t cap_stopped,
                          const double* grad_dead, int64_t cap_dead, double* grad_face_verts,
                          double* grad_src_rays, const int32_t* counts, void* workspace,
                          size_t workspace_bytes, void* stream) {
  if (!scene_ok(scene) || n_rays < 0 || max_passes < 0 || !counts || !workspace ||
      !grad_face_verts)
    return TFRT_E_BADARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (state_dtype == TFRT_F32)
    return trace3d_backward_t<float>(src_rays, src_stride, n_rays, scene, new_ray_length,
                                     dead_ray_length, max_passes, state_dtype, grad_finished,
                                     cap_finished, grad_active, cap_active, grad_stopped,
                                     cap_stopped, grad_dead, cap_dead, grad_face_verts,
                                     grad_src_rays, counts, workspace, workspace_bytes, st);
  if (state_dtype == TFRT_F64)
    return trace3d_backward_t<double>(src_rays, src_stride, n_rays, scene, new_ray_length,
                                      dead_ray_length, max_passes, state_dtype, grad_finished,
                                      cap_finished, grad_active, cap_active, grad_stopped,
                                      cap_stopped, grad_dead, cap_dead, grad_face_verts,
                                      grad_src_rays, counts, workspace, workspace_bytes, st);
  if (state_dtype == TFRT_F16)
    return trace3d_backward_t<_Float16>(src_rays, src_stride, n_rays, scene, new_ray_length,
                                      dead_ray_length, max_passes, state_dtype, grad_finished,
                                      cap_finished, grad_active, cap_active, grad_stopped,
                                      cap_stopped, grad_dead, cap_dead, grad_face_verts,
                                      grad_src_rays, counts, workspace, workspace_bytes, st);
  return TFRT_E_UNSUPPORTED;
}

size_t tfrt_trace3d_backward_goal_workspace_bytes(int64_t n_rays) {
  if (n_rays < 0) return 0;
  // per 64 rays: one partial error sum + two int32 counts (in-place tapes)
  return 2 * align_up((size_t)cdiv(n_rays > 0 ? n_rays : 1, 64) * sizeof(double));
}

int tfrt_trace3d_backward_goal(const void* src_rays, int64_t src_stride, int64_t n_rays,
                               const tfrt_scene3d* scene, double new_ray_length,
                               double dead_ray_length, int32_t max_passes, int32_t state_dtype,
                               const tfrt_ray_out* finished, const int32_t* fields,
                               int32_t n_fields, const double* goal, int64_t goal_stride,
                               int64_t goal_ray_stride, double* error_out, int64_t* tests_total,
                               void* goal_workspace, size_t goal_workspace_bytes,
                               tfrt_goal_pending* pending, const double* grad_active,
                               int64_t cap_active, const double* grad_stopped,
                               int64_t cap_stopped, const double* grad_dead, int64_t cap_dead,
                               double* grad_face_verts, double* grad_src_rays,
                               const int32_t* counts, void* workspace, size_t workspace_bytes,
                               void* stream) {
  if (!scene_ok(scene) || n_rays < 0 || max_passes < 0 || !counts || !workspace ||
      !grad_face_verts || !finished || !fields || n_fields < 1 || n_fields > 6 || !error_out ||
      !pending || !goal_workspace || goal_stride < 0 || goal_ray_stride < 0 ||
      goal_workspace_bytes < tfrt_trace3d_backward_goal_workspace_bytes(n_rays))
    return TFRT_E_BADARG;
  // (an in-place trace leaves no finished block: the rows are recomputed from its tape)
  const bool tape_rows = inplace_trace(scene, n_rays, scene->n_faces, max_passes);
  if (n_rays > 0 && (!goal || (!tape_rows && (!finished->rays || finished->capacity <= 0))))
    return TFRT_E_BADARG;
  ChainGoal g;
  g.fin_rays = finished->rays;
  g.fin_cap = finished->capacity;
  g.gf.n = n_fields;
  for (int c = 0; c < 6; ++c) {
    g.gf.row[c] = c < n_fields ? fields[c] : 0;
    if (g.gf.row[c] < 0 || g.gf.row[c] > 5) return TFRT_E_BADARG;
  }
  g.goal = goal;
  g.goal_stride = goal_stride;
  g.goal_ray_stride = goal_ray_stride;
  g.partial = static_cast<double*>(goal_workspace);
  // (an in-place trace that was given no room for ray sets ran no scan: `counts` holds nothing yet,
  // the sweep counts the finished rays and the tests itself)
  const bool own_counts = tape_rows && !grad_active && !grad_stopped && !grad_dead;
  g.partial_cnt = own_counts
                      ? reinterpret_cast<int32_t*>(static_cast<char*>(goal_workspace) +
                                                   align_up((size_t)cdiv(n_rays > 0 ? n_rays : 1, 64) *
                                                            sizeof(double)))
                      : nullptr;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int rc = TFRT_E_UNSUPPORTED;
  if (state_dtype == TFRT_F32)
    rc = trace3d_backward_t<float>(src_rays, src_stride, n_rays, scene, new_ray_length,
                                   dead_ray_length, max_passes, state_dtype, nullptr, 0,
                                   grad_active, cap_active, grad_stopped, cap_stopped, grad_dead,
                                   cap_dead, grad_face_verts, grad_src_rays, counts, workspace,
                                   workspace_bytes, st, &g);
  else if (state_dtype == TFRT_F64)
    rc = trace3d_backward_t<double>(src_rays, src_stride, n_rays, scene, new_ray_length,
                                    dead_ray_length, max_passes, state_dtype, nullptr, 0,
                                    grad_active, cap_active, grad_stopped, cap_stopped, grad_dead,
                                    cap_dead, grad_face_verts, grad_src_rays, counts, workspace,
                                    workspace_bytes, st, &g);
  else if (state_dtype == TFRT_F16)
    rc = trace3d_backward_t<_Float16>(src_rays, src_stride, n_rays, scene, new_ray_length,
                                      dead_ray_length, max_passes, state_dtype, nullptr, 0,
                                      grad_active, cap_active, grad_stopped, cap_stopped,
                                      grad_dead, cap_dead, grad_face_verts, grad_src_rays, counts,
                                      workspace, workspace_bytes, st, &g);
  if (rc != 0) return rc;
  // trailing counters of the trace: {total_active, total_finished, ..., n_tests_lo, n_tests_hi}
  const int32_t* tail = counts + (size_t)max_passes * TFRT_COUNTS_PER_PASS;
  pending->partial = g.partial;
  pending->n_partial = n_rays > 0 ? cdiv(n_rays, 64) : 0;
  pending->n_finished = tail + 1;
  pending->n_fields = n_fields;
  pending->error_out = error_out;
  pending->tests_lo_hi = tail + 4;
  pending->tests_total = tests_total;
  pending->partial_counts = g.partial_cnt;
  pending->n_faces = scene->n_faces;
  pending->counts_tail = g.partial_cnt != nullptr ? const_cast<int32_t*>(tail) : nullptr;
  return 0;
}

size_t tfrt_intersect3d_workspace_bytes(int64_t n_rays, int64_t n_faces) {
  if (n_rays < 0 || n_faces < 0) return 0;
  const Plan3 pl = make_plan(n_rays, n_faces);
  const size_t n = n_rays > 0 ? n_rays : 1, m = n_faces > 0 ? n_faces : 1;
  return align_up(4 * sizeof(double)) + align_up(m * sizeof(float4)) + align_up(64) +
         align_up((size_t)pl.chunks * n * sizeof(double)) +
         align_up((size_t)pl.chunks * n * sizeof(int32_t)) + align_up((size_t)8 * n * sizeof(float));
}

int tfrt_intersect3d(const void* rays, int64_t stride, int64_t n_rays, int32_t state_dtype,
                     const double* face_verts, int64_t n_faces, double intersect_epsilion,
                     double size_epsilion, double ray_start_epsilion, double* x, double* y,
                     double* z, uint8_t* valid, double* ray_u, double* trig_u, double* trig_v,
                     int32_t* gather_trig, void* workspace, size_t workspace_bytes,
                     void* stream) {
  if (n_rays < 0 || n_faces < 0 || n_faces >= (1ll << 29) || stride < n_rays || !workspace ||
      (n_faces > 0 && !face_verts))
    return TFRT_E_BADARG;
  if (workspace_bytes < tfrt_intersect3d_workspace_bytes(n_rays, n_faces)) return TFRT_E_WORKSPACE;
  if (n_rays == 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int M = (int)n_faces;
  const Plan3 pl = make_plan(n_rays, n_faces);
  const size_t n = n_rays, m = n_faces > 0 ? n_faces : 1;
  char* ws = static_cast<char*>(workspace);
  size_t o = 0;
  double* c0 = reinterpret_cast<double*>(ws + o);
  o += align_up(4 * sizeof(double));
  float4* sphere = reinterpret_cast<float4*>(ws + o);
  o += align_up(m * sizeof(float4));
  int32_t* nptr = reinterpret_cast<int32_t*>(ws + o);
  o += align_up(64);
  double* part_t = reinterpret_cast<double*>(ws + o);
  o += align_up((size_t)pl.chunks * n * sizeof(double));
  int32_t* part_i = reinterpret_cast<int32_t*>(ws + o);
  o += align_up((size_t)pl.chunks * n * sizeof(int32_t));
  float* prep = reinterpret_cast<float*>(ws + o);
  if (M <= 0)
    hipLaunchKernelGGL(k_init, dim3(1), dim3(64), 0, st, nptr, (int)n_rays, nptr + 8,
                       (unsigned int*)nullptr);
  if (M > 0) {
    hipLaunchKernelGGL(k_center, dim3(1), dim3(BLOCK), 0, st, face_verts, M, c0, nptr, (int)n_rays,
                       nptr + 8, (unsigned int*)nullptr);
    hipLaunchKernelGGL(k_spheres, dim3(cdiv(M, BLOCK)), dim3(BLOCK), 0, st, face_verts, M, c0,
                       size_epsilion, sphere, FaceTables(), static_cast<double*>(nullptr),
                       (int64_t)0);
  }
#define TFRT_SEAM(TT)                                                                          \
  launch_intersect<TT>(pl, st, static_cast<const TT*>(rays), stride, nptr, nullptr, sphere,    \
                       face_verts, c0, prep, (int64_t)n, M, intersect_epsilion, size_epsilion, \
                       ray_start_epsilion, part_t, part_i, (int64_t)n, nullptr);               \
  hipLaunchKernelGGL((k_finalize_seam<TT>), dim3(pl.nblk), dim3(BLOCK), 0, st,                 \
                     static_cast<const TT*>(rays), stride, (int)n_rays, pl.chunks, part_t,     \
                     part_i, (int64_t)n, face_verts, M, intersect_epsilion, size_epsilion,     \
                     ray_start_epsilion, x, y, z, valid, ray_u, trig_u, trig_v, gather_trig)
  if (state_dtype == TFRT_F32) {
    TFRT_SEAM(float);
  } else if (state_dtype == TFRT_F64) {
    TFRT_SEAM(double);
  } else if (state_dtype == TFRT_F16) {
    TFRT_SEAM(_Float16);
  } else {
    return TFRT_E_UNSUPPORTED;
  }
#undef TFRT_SEAM
  return hipGetLastError() == hipSuccess ? 0 : TFRT_E_LAUNCH;
}

int tfrt_profile_enable(int enable) {
  for (auto& r : g_prof) {
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  g_prof.clear();
  g_prof_on = enable != 0;
  return 0;
}

int tfrt_profile_read(float* ms, int32_t max_records) {
  return tfrt_profile_read_kind(TFRT_PROF_INTERSECT, ms, max_records);
}

int tfrt_profile_read_kind(int32_t kind, float* ms, int32_t max_records) {
  int n = 0;
  for (auto& r : g_prof) {
    if (r.kind != kind) continue;
    if (n >= max_records) break;
    if (hipEventSynchronize(r.b) != hipSuccess) return TFRT_E_LAUNCH;
    float t = 0.f;
    (void)hipEventElapsedTime(&t, r.a, r.b);
    ms[n++] = t;
  }
  return n;
}

#ifdef TFRT_TUNING
TFRT_TUNING_EXPORTS
#endif

}  // extern "C"
