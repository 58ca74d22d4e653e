// Coherent ray order on the device, permutations of ray sets, and the way back.
//
// The reference's ray sets are ordered: every class lists, pass after pass, its rays in the order
// of the source set (tfrt/engine.py:2069-2111 boolean_mask per pass, :1379-1403 the ray-set
// properties, :2311-2330 ray_trace owns both).  The trace kernels are fastest when 64 consecutive
// rays are neighbours in space (k_intersect_beam, tfrt_scene3d.coherent_rays).  This file is what
// a caller needs to have both:
//
//   tfrt_ray_order      rays -> int32 permutation along a Hilbert curve through the points where
//                       their lines pass the scene (k_order_xy, k_order_key, then a stable
//                       two-pass LSD radix sort of (key, index) with digits of up to 13 bits)
//   tfrt_permute_rays   ray block -> ray block in that order (through 8-element records, so the
//                       random access is one read per ray instead of six)
//   tfrt_gather_rows    any per-ray rows (n(lambda) table, goal rows, fields) through an index
//   tfrt_restore_order  ray ids of one output class of a trace over permuted rays -> the row
//                       permutation that puts the class back into the reference's order
//                       (a bitmap over (pass, original id), a popcount scan, a rank per row)
//
// Nothing here allocates or synchronises; every launch has data-independent arguments, so the
// calls can be captured into the graph of an optimiser step (a source re-drawn every step is
// ordered every step).
#include "tfrt_common.h"
#include "source_programs.h"

namespace tfrt {

// where the rays to be ordered come from: a ray block, or a source program (then they are never
// written in source order at all)
template <typename T>
struct BlockRays {
  static constexpr bool HAS_F32 = false;
  const T* rays;
  int64_t stride;
  __device__ __forceinline__ void load(int64_t i, double s[3], double e[3]) const {
    load_ray3(rays, stride, i, s, e);
  }
};
struct ProgramRays {
  static constexpr bool HAS_F32 = true;
  tfrt_source3d_program sp;
  int64_t first;
  __device__ __forceinline__ void load(int64_t i, double s[3], double e[3]) const {
    eval_ray(sp, first + i, s, e);
  }
  // (float32 evaluation: enough for the order's keys, a fraction of the float64 one's time)
  __device__ __forceinline__ void load_f(int64_t i, float s[3], float e[3]) const {
    eval_ray<float>(sp, first + i, s, e);
  }
};

// ------------------------------------------------------------------------------ order keys

constexpr int ORD_FACE_SAMPLES = 64;

__device__ __forceinline__ unsigned enc_f(float f) {  // monotone float -> u32
  const unsigned u = __float_as_uint(f);
  return (u >> 31) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dec_f(unsigned k) {
  return __uint_as_float((k >> 31) ? (k & 0x7FFFFFFFu) : ~k);
}

// sum of v[0..3] over the 256 threads of the block, the same on every run (a fixed tree)
__device__ __forceinline__ void block_sum4(double v[4], double (*red)[4]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int q = 0; q < 4; ++q) red[tid][q] = v[q];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
#pragma unroll
      for (int q = 0; q < 4; ++q) red[tid][q] += red[tid + s][q];
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = red[0][q];
  __syncthreads();
}

struct OrderFrame {
  double c[3], w[3], a[3], b[3];
  int planar;
};

__device__ __forceinline__ void order_coords_f(const OrderFrame& fr, const float s[3],
                                               const float e[3], float* xo, float* yo);

// The frame in which the rays' lines are compared: the middle of the scene (mean centroid of 64
// sampled faces, or the mean end point of 256 sampled rays) and the plane perpendicular to the
// bundle's mean direction (that of 256 sampled rays unless the caller gives an axis); rays without
// a common direction, |mean| <= 1/2 (an isotropic point source): no plane -- octahedral map of the
// directions.  One workgroup, a fixed reduction tree: the same frame on every run.  Also arms the
// extents mm[0..3] (running minima of enc(x), ~enc(x), enc(y), ~enc(y)).
// (the work of one workgroup; `red` is BLOCK x 4 doubles of LDS; thread 0 returns the frame in *out)
template <typename R, bool F32>
__device__ __forceinline__ void block_frame(const R& src, int n, const double* __restrict__ fverts,
                                            int M, double ax0, double ax1, double ax2, int has_axis,
                                            double (*red)[4], OrderFrame* out, float sf[3],
                                            float ef[3]) {
  const int tid = threadIdx.x;
  const int64_t is = n > 0 ? (int64_t)tid * n / BLOCK : 0;  // this thread's sample ray
  double ss[3] = {0, 0, 0}, se[3] = {0, 0, 0};
  if (n > 0) {
    if constexpr (F32) {
      src.load_f(is, sf, ef);
      for (int q = 0; q < 3; ++q) {
        ss[q] = sf[q];
        se[q] = ef[q];
      }
    } else {
      src.load(is, ss, se);
    }
  }
  double acc[4] = {0, 0, 0, 0};
  if (fverts != nullptr && M > 0) {
    if (tid < ORD_FACE_SAMPLES) {
      const double* f = fverts + 9 * ((int64_t)tid * M / ORD_FACE_SAMPLES);
      acc[0] = (f[0] + f[3] + f[6]) / 3.0;
      acc[1] = (f[1] + f[4] + f[7]) / 3.0;
      acc[2] = (f[2] + f[5] + f[8]) / 3.0;
      acc[3] = 1.0;
      if (!(isfinite(acc[0]) && isfinite(acc[1]) && isfinite(acc[2]))) acc[0] = acc[1] = acc[2] = acc[3] = 0.0;
    }
  } else if (n > 0 && isfinite(se[0]) && isfinite(se[1]) && isfinite(se[2])) {
    acc[0] = se[0];
    acc[1] = se[1];
    acc[2] = se[2];
    acc[3] = 1.0;
  }
  block_sum4(acc, red);
  const double cn = acc[3] > 0.0 ? acc[3] : 1.0;
  const double cx = acc[0] / cn, cy = acc[1] / cn, cz = acc[2] / cn;
  double dir[4] = {0, 0, 0, 0};
  if (!has_axis && n > 0) {
    const double dx = se[0] - ss[0], dy = se[1] - ss[1], dz = se[2] - ss[2];
    const double len = sqrt(dx * dx + dy * dy + dz * dz);
    if (isfinite(len) && len > 0.0) {
      dir[0] = dx / len;
      dir[1] = dy / len;
      dir[2] = dz / len;
      dir[3] = 1.0;
    }
  }
  if (!has_axis) block_sum4(dir, red);
  if (tid == 0) {
    OrderFrame fr;
    double w[3] = {dir[0], dir[1], dir[2]};
    double ng = dir[3] > 0.0 ? dir[3] : 1.0;
    if (has_axis) {
      w[0] = ax0;
      w[1] = ax1;
      w[2] = ax2;
    }
    const double wl = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    if (has_axis) ng = wl;  // (any length of a given axis counts as "one direction")
    fr.c[0] = cx;
    fr.c[1] = cy;
    fr.c[2] = cz;
    fr.planar = (wl > 0.5 * ng && wl > 0.0) ? 1 : 0;
#pragma unroll
    for (int q = 0; q < 3; ++q) fr.w[q] = fr.a[q] = fr.b[q] = 0.0;
    if (fr.planar) {
      w[0] /= wl;
      w[1] /= wl;
      w[2] /= wl;
      int k = 0;
      if (fabs(w[1]) < fabs(w[k])) k = 1;
      if (fabs(w[2]) < fabs(w[k])) k = 2;
      const double e[3] = {k == 0 ? 1.0 : 0.0, k == 1 ? 1.0 : 0.0, k == 2 ? 1.0 : 0.0};
      double a[3] = {w[1] * e[2] - w[2] * e[1], w[2] * e[0] - w[0] * e[2], w[0] * e[1] - w[1] * e[0]};
      const double al = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
      a[0] /= al;
      a[1] /= al;
      a[2] /= al;
      fr.b[0] = w[1] * a[2] - w[2] * a[1];
      fr.b[1] = w[2] * a[0] - w[0] * a[2];
      fr.b[2] = w[0] * a[1] - w[1] * a[0];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        fr.w[q] = w[q];
        fr.a[q] = a[q];
      }
    }
    *out = fr;
  }
}

template <typename R>
__global__ __launch_bounds__(BLOCK) void k_order_frame(const R src,
                                                       int n, const double* __restrict__ fverts,
                                                       int M, double ax0, double ax1, double ax2,
                                                       int has_axis, OrderFrame* __restrict__ frame,
                                                       unsigned* __restrict__ mm) {
  __shared__ double red[BLOCK][4];
  mm[threadIdx.x] = 0xFFFFFFFFu;      // (MM_SLOTS * 4 == BLOCK values)
  float sf[3], ef[3];
  block_frame<R, false>(src, n, fverts, M, ax0, ax1, ax2, has_axis, red, frame, sf, ef);
}

// where a ray's line passes the middle of the scene, as two coordinates of the frame (NaN: no line)
__device__ __forceinline__ void order_coords(const OrderFrame& fr, const double s[3],
                                             const double e[3], float* xo, float* yo) {
  float x = __builtin_nanf(""), y = __builtin_nanf("");
  const double dx = e[0] - s[0], dy = e[1] - s[1], dz = e[2] - s[2];
  const double len = sqrt(dx * dx + dy * dy + dz * dz);
  if (isfinite(len) && len > 0.0) {
    const double u[3] = {dx / len, dy / len, dz / len};
    if (fr.planar) {
      // foot of the perpendicular from the centre to the line, relative to the centre
      const double t = (fr.c[0] - s[0]) * u[0] + (fr.c[1] - s[1]) * u[1] + (fr.c[2] - s[2]) * u[2];
      const double p[3] = {s[0] + t * u[0] - fr.c[0], s[1] + t * u[1] - fr.c[1],
                           s[2] + t * u[2] - fr.c[2]};
      x = (float)(p[0] * fr.a[0] + p[1] * fr.a[1] + p[2] * fr.a[2]);
      y = (float)(p[0] * fr.b[0] + p[1] * fr.b[1] + p[2] * fr.b[2]);
    } else {
      const double l1 = fabs(u[0]) + fabs(u[1]) + fabs(u[2]);
      const double ox = u[0] / l1, oy = u[1] / l1, oz = u[2] / l1;
      x = (float)(oz < 0.0 ? (1.0 - fabs(oy)) * (ox >= 0.0 ? 1.0 : -1.0) : ox);
      y = (float)(oz < 0.0 ? (1.0 - fabs(ox)) * (oy >= 0.0 ? 1.0 : -1.0) : oy);
    }
    if (!(isfinite(x) && isfinite(y))) x = y = __builtin_nanf("");
  }
  *xo = x;
  *yo = y;
}

// the same in float32, for rays evaluated in float32 (ProgramRays::load_f)
__device__ __forceinline__ void order_coords_f(const OrderFrame& fr, const float s[3],
                                               const float e[3], float* xo, float* yo) {
  float x = __builtin_nanf(""), y = __builtin_nanf("");
  const float dx = e[0] - s[0], dy = e[1] - s[1], dz = e[2] - s[2];
  const float len = sqrtf(dx * dx + dy * dy + dz * dz);
  if (isfinite(len) && len > 0.f) {
    const float il = 1.f / len;
    const float u[3] = {dx * il, dy * il, dz * il};
    if (fr.planar) {
      const float c[3] = {(float)fr.c[0], (float)fr.c[1], (float)fr.c[2]};
      const float q[3] = {s[0] - c[0], s[1] - c[1], s[2] - c[2]};   // start relative to the centre
      const float t = -(q[0] * u[0] + q[1] * u[1] + q[2] * u[2]);
      const float p[3] = {q[0] + t * u[0], q[1] + t * u[1], q[2] + t * u[2]};
      x = p[0] * (float)fr.a[0] + p[1] * (float)fr.a[1] + p[2] * (float)fr.a[2];
      y = p[0] * (float)fr.b[0] + p[1] * (float)fr.b[1] + p[2] * (float)fr.b[2];
    } else {
      const float l1 = fabsf(u[0]) + fabsf(u[1]) + fabsf(u[2]);
      const float ox = u[0] / l1, oy = u[1] / l1, oz = u[2] / l1;
      x = oz < 0.f ? (1.f - fabsf(oy)) * (ox >= 0.f ? 1.f : -1.f) : ox;
      y = oz < 0.f ? (1.f - fabsf(ox)) * (oy >= 0.f ? 1.f : -1.f) : oy;
    }
    if (!(isfinite(x) && isfinite(y))) x = y = __builtin_nanf("");
  }
  *xo = x;
  *yo = y;
}

// extents of the block's finite coordinates into mm: wave minima by shuffles, one atomic per
// block and value -- spread over MM_SLOTS copies of the four values (thousands of atomics on ONE
// address are served one after the other: 47 us of a 50 us kernel at a million rays)
constexpr int MM_SLOTS = 64;
__device__ __forceinline__ void order_extents(float x, float y, unsigned (*wmm)[4],
                                              unsigned* __restrict__ mm) {
  const int tid = threadIdx.x;
  const bool ok = x == x;
  unsigned v[4] = {ok ? enc_f(x) : 0xFFFFFFFFu, ok ? ~enc_f(x) : 0xFFFFFFFFu,
                   ok ? enc_f(y) : 0xFFFFFFFFu, ok ? ~enc_f(y) : 0xFFFFFFFFu};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v[q] = min(v[q], (unsigned)__shfl_xor((int)v[q], d, 64));
  }
  if ((tid & 63) == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) wmm[tid >> 6][q] = v[q];
  }
  __syncthreads();
  if (tid < 4) {
    unsigned m = wmm[0][tid];
    for (int w = 1; w < WAVES; ++w) m = min(m, wmm[w][tid]);
    if (m != 0xFFFFFFFFu) atomicMin(&mm[(blockIdx.x % MM_SLOTS) * 4 + tid], m);
  }
}

// the four extents from their MM_SLOTS copies (every lane of a wave gets them)
__device__ __forceinline__ void order_extents_read(const unsigned* __restrict__ mm, float* xlo,
                                                   float* xhi, float* ylo, float* yhi) {
  const int lane = threadIdx.x & 63;
  uint4 v = reinterpret_cast<const uint4*>(mm)[lane];     // (MM_SLOTS == 64: one slot per lane)
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    v.x = min(v.x, (unsigned)__shfl_xor((int)v.x, d, 64));
    v.y = min(v.y, (unsigned)__shfl_xor((int)v.y, d, 64));
    v.z = min(v.z, (unsigned)__shfl_xor((int)v.z, d, 64));
    v.w = min(v.w, (unsigned)__shfl_xor((int)v.w, d, 64));
  }
  *xlo = dec_f(v.x);
  *xhi = dec_f(~v.y);
  *ylo = dec_f(v.z);
  *yhi = dec_f(~v.w);
}

template <typename R>
__global__ __launch_bounds__(BLOCK) void k_order_xy(const R src,
                                                    int n, const OrderFrame* __restrict__ frame,
                                                    float2* __restrict__ xy,
                                                    unsigned* __restrict__ mm) {
  __shared__ unsigned wmm[WAVES][4];
  const OrderFrame fr = *frame;
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  float x = __builtin_nanf(""), y = __builtin_nanf("");
  if (i < n) {
    double s[3], e[3];
    src.load(i, s, e);
    order_coords(fr, s, e, &x, &y);
    xy[i] = make_float2(x, y);
  }
  order_extents(x, y, wmm, mm);
}

// Index of grid point (x, y), 0 <= x, y < 2^bits, along the Hilbert curve: unlike a Morton code
// the curve has no jumps, so 64 consecutive rays cover a compact patch (a Morton order made the
// 99th-percentile wavefront touch 136 faces instead of 18).
__device__ __forceinline__ unsigned hilbert_index(unsigned x, unsigned y, int bits) {
  unsigned d = 0;
  const unsigned n1 = (1u << bits) - 1u;
  for (unsigned s = 1u << (bits - 1); s > 0; s >>= 1) {
    const unsigned rx = (x & s) ? 1u : 0u, ry = (y & s) ? 1u : 0u;
    d += s * s * ((3u * rx) ^ ry);
    if (ry == 0u) {
      if (rx == 1u) {
        x = n1 - x;
        y = n1 - y;
      }
      const unsigned t = x;
      x = y;
      y = t;
    }
  }
  return d;
}

// keys + the histogram of their low digits per sort tile (tile = BLOCK * ITEMS consecutive rays)
template <int ITEMS>
__global__ __launch_bounds__(BLOCK) void k_order_key(const float2* __restrict__ xy, int n,
                                                     const unsigned* __restrict__ mm, int bits,
                                                     unsigned* __restrict__ keys,
                                                     unsigned* __restrict__ hist, int nblk) {
  extern __shared__ unsigned h_lds[];
  const int tid = threadIdx.x;
  const int bins = 1 << bits;
  for (int d = tid; d < bins; d += BLOCK) h_lds[d] = 0u;
  __syncthreads();
  float xlo, xhi, ylo, yhi;
  order_extents_read(mm, &xlo, &xhi, &ylo, &yhi);
  const float g1 = (float)(bins - 1);
  const float sx = xhi > xlo ? g1 / (xhi - xlo) : 0.f, sy = yhi > ylo ? g1 / (yhi - ylo) : 0.f;
  const unsigned kmax = (bits >= 16) ? 0xFFFFFFFFu : ((1u << (2 * bits)) - 1u);
  const int base = blockIdx.x * (BLOCK * ITEMS);
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const int i = base + r * BLOCK + tid;
    if (i < n) {
      const float2 p = xy[i];
      unsigned key = kmax;  // rays that are no line (zero length, non-finite): last
      if (p.x == p.x) {
        const float fx = fminf(fmaxf((p.x - xlo) * sx, 0.f), g1);
        const float fy = fminf(fmaxf((p.y - ylo) * sy, 0.f), g1);
        key = hilbert_index((unsigned)fx, (unsigned)fy, bits);
      }
      keys[i] = key;
      atomicAdd(&h_lds[key & (unsigned)(bins - 1)], 1u);
    }
  }
  __syncthreads();
  for (int d = tid; d < bins; d += BLOCK) hist[(int64_t)blockIdx.x * bins + d] = h_lds[d];
}


// Program sources: rays -> coordinates -> keys in ONE launch, float32 evaluation.  Every workgroup
// makes the frame itself from the same 256 sample rays (the same arithmetic in every workgroup:
// block_frame) and takes the extents of the key grid from those samples, widened by 1/16 -- a ray
// outside them lands in an edge cell --, so nothing has to pass over all rays before the keys are
// made.  Also leaves the histogram of the low digits per sort tile -- of the HIGH digits when HIGH
// (tfrt_source3d_order_cells: the scatter by the high digit comes first, see k_bucket_sort).
template <int ITEMS, bool HIGH>
__global__ __launch_bounds__(BLOCK) void k_order_pkey(const ProgramRays src, int n,
                                                      const double* __restrict__ fverts, int M,
                                                      double ax0, double ax1, double ax2,
                                                      int has_axis, int bits,
                                                      unsigned* __restrict__ keys,
                                                      unsigned* __restrict__ hist, int nblk) {
  extern __shared__ unsigned h_lds[];
  __shared__ double red[BLOCK][4];
  __shared__ OrderFrame fr;
  __shared__ unsigned wmm[WAVES][4];
  const int tid = threadIdx.x;
  const int bins = 1 << bits;
  for (int d = tid; d < bins; d += BLOCK) h_lds[d] = 0u;
  float sf[3] = {0.f, 0.f, 0.f}, ef[3] = {0.f, 0.f, 0.f};
  block_frame<ProgramRays, true>(src, n, fverts, M, ax0, ax1, ax2, has_axis, red, &fr, sf, ef);
  __syncthreads();
  float xlo, xhi, ylo, yhi;
  {
    float x, y;
    order_coords_f(fr, sf, ef, &x, &y);
    const bool ok = n > 0 && x == x;
    unsigned v[4] = {ok ? enc_f(x) : 0xFFFFFFFFu, ok ? ~enc_f(x) : 0xFFFFFFFFu,
                     ok ? enc_f(y) : 0xFFFFFFFFu, ok ? ~enc_f(y) : 0xFFFFFFFFu};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) v[q] = min(v[q], (unsigned)__shfl_xor((int)v[q], d, 64));
    }
    if ((tid & 63) == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) wmm[tid >> 6][q] = v[q];
    }
    __syncthreads();
    unsigned m[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      m[q] = wmm[0][q];
      for (int w = 1; w < WAVES; ++w) m[q] = min(m[q], wmm[w][q]);
    }
    xlo = dec_f(m[0]);
    xhi = dec_f(~m[1]);
    ylo = dec_f(m[2]);
    yhi = dec_f(~m[3]);
    const float mx = 0.0625f * (xhi - xlo), my = 0.0625f * (yhi - ylo);
    xlo -= mx;
    xhi += mx;
    ylo -= my;
    yhi += my;
  }
  const float g1 = (float)(bins - 1);
  const float sx = xhi > xlo ? g1 / (xhi - xlo) : 0.f, sy = yhi > ylo ? g1 / (yhi - ylo) : 0.f;
  const unsigned kmax = (bits >= 16) ? 0xFFFFFFFFu : ((1u << (2 * bits)) - 1u);
  const int base = blockIdx.x * (BLOCK * ITEMS);
  for (int r = 0; r < ITEMS; ++r) {
    const int i = base + r * BLOCK + tid;
    if (i < n) {
      float s3[3], e3[3], x, y;
      src.load_f(i, s3, e3);
      order_coords_f(fr, s3, e3, &x, &y);
      unsigned key = kmax;  // rays that are no line (zero length, non-finite): last
      if (x == x) {
        const float fx = fminf(fmaxf((x - xlo) * sx, 0.f), g1);
        const float fy = fminf(fmaxf((y - ylo) * sy, 0.f), g1);
        key = hilbert_index((unsigned)fx, (unsigned)fy, bits);
      }
      keys[i] = key;
      atomicAdd(&h_lds[(HIGH ? key >> bits : key) & (unsigned)(bins - 1)], 1u);
    }
  }
  __syncthreads();
  for (int d = tid; d < bins; d += BLOCK) hist[(int64_t)blockIdx.x * bins + d] = h_lds[d];
}


// ------------------------------------------------------------------------------ radix sort
//
// Stable LSD radix sort of (key, value) pairs, two passes with digits of `bits` bits each
// (bits <= 13: 8192 bins).  Per pass: tile histograms hist[tile][digit] (rows written and read
// whole: a digit-major matrix meant a million scattered 4-byte accesses per pass), their prefix
// down the columns (k_colscan_rows: within segments of 32 tiles; k_colscan_segs: the segments'
// bases and the digits' totals, whose scan every tile does for itself), and the scatter: every tile ranks its
// items (wave-wide match of the digit by ballots, a counter per wave and digit in LDS), sorts them
// by digit in LDS and writes runs of equal digits to consecutive addresses.
constexpr int COL_SEG = 32;   // tiles per column segment
constexpr int SCAN_CHUNK = 4096;  // words per block of the restore scan

// hist[r][d] <- sum of hist[segment start .. r)[d];  seg[s][d] <- the segment's sum
__global__ __launch_bounds__(BLOCK) void k_colscan_rows(unsigned* __restrict__ hist, int nblk,
                                                        int bins, unsigned* __restrict__ seg) {
  const int d = blockIdx.x * BLOCK + threadIdx.x;
  if (d >= bins) return;
  const int r0 = blockIdx.y * COL_SEG, r1 = min(nblk, r0 + COL_SEG);
  unsigned run = 0;
  int r = r0;
  for (; r + 8 <= r1; r += 8) {
    unsigned v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = hist[(int64_t)(r + q) * bins + d];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      hist[(int64_t)(r + q) * bins + d] = run;
      run += v[q];
    }
  }
  for (; r < r1; ++r) {
    const unsigned v = hist[(int64_t)r * bins + d];
    hist[(int64_t)r * bins + d] = run;
    run += v;
  }
  seg[(int64_t)blockIdx.y * bins + d] = run;
}

// seg[s][d] <- sum of seg[0 .. s)[d];  total[d] <- the digit's items in all tiles
__global__ __launch_bounds__(BLOCK) void k_colscan_segs(unsigned* __restrict__ seg, int nseg,
                                                        int bins, unsigned* __restrict__ total) {
  const int d = blockIdx.x * BLOCK + threadIdx.x;
  if (d >= bins) return;
  unsigned run = 0;
  int s = 0;
  for (; s + 8 <= nseg; s += 8) {
    unsigned v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = seg[(int64_t)(s + q) * bins + d];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      seg[(int64_t)(s + q) * bins + d] = run;
      run += v[q];
    }
  }
  for (; s < nseg; ++s) {
    const unsigned v = seg[(int64_t)s * bins + d];
    seg[(int64_t)s * bins + d] = run;
    run += v;
  }
  total[d] = run;
}

template <int ITEMS>
__global__ __launch_bounds__(BLOCK) void k_sort_hist(const uint2* __restrict__ pairs, int n,
                                                     int shift, int bits,
                                                     unsigned* __restrict__ hist, int nblk) {
  extern __shared__ unsigned h_lds[];
  const int tid = threadIdx.x;
  const int bins = 1 << bits;
  for (int d = tid; d < bins; d += BLOCK) h_lds[d] = 0u;
  __syncthreads();
  const int base = blockIdx.x * (BLOCK * ITEMS);
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const int i = base + r * BLOCK + tid;
    if (i < n) atomicAdd(&h_lds[(pairs[i].x >> shift) & (unsigned)(bins - 1)], 1u);
  }
  __syncthreads();
  for (int d = tid; d < bins; d += BLOCK) hist[(int64_t)blockIdx.x * bins + d] = h_lds[d];
}

template <int ITEMS>
__global__ __launch_bounds__(BLOCK) void k_sort_hist_keys(const unsigned* __restrict__ keys, int n,
                                                          int bits, unsigned* __restrict__ hist) {
  extern __shared__ unsigned h_lds[];
  const int tid = threadIdx.x;
  const int bins = 1 << bits;
  for (int d = tid; d < bins; d += BLOCK) h_lds[d] = 0u;
  __syncthreads();
  const int base = blockIdx.x * (BLOCK * ITEMS);
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const int i = base + r * BLOCK + tid;
    if (i < n) atomicAdd(&h_lds[keys[i] & (unsigned)(bins - 1)], 1u);
  }
  __syncthreads();
  for (int d = tid; d < bins; d += BLOCK) hist[(int64_t)blockIdx.x * bins + d] = h_lds[d];
}

// exclusive prefix of `v` over the 256 threads of the block (wave scan + one LDS exchange);
// returns the block total in *total
__device__ __forceinline__ unsigned block_exclusive(unsigned v, unsigned* wsum, unsigned* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned o = (unsigned)__shfl_up((int)x, d, 64);
    if (lane >= d) x += o;
  }
  if (lane == 63) wsum[wave] = x;
  __syncthreads();
  unsigned base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < WAVES; ++w) {
    const unsigned s = wsum[w];
    if (w < wave) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + x - v;
}

// exclusive scan, in LDS, of `cnt` values loaded from `src` (every block does this for the chunk
// totals: a few hundred values; saves a launch).  All 256 threads take part.
__device__ __forceinline__ void lds_exclusive_from(const unsigned* __restrict__ src, int cnt,
                                                   unsigned* dst, unsigned* wsum) {
  const int per = (cnt + BLOCK - 1) / BLOCK;
  const int lo = threadIdx.x * per;
  unsigned sum = 0;
  for (int q = 0; q < per; ++q) {
    const int i = lo + q;
    const unsigned t = i < cnt ? src[i] : 0u;
    if (i < cnt) dst[i] = sum;
    sum += t;
  }
  unsigned total;
  const unsigned base = block_exclusive(sum, wsum, &total);
  for (int q = 0; q < per; ++q) {
    const int i = lo + q;
    if (i < cnt) dst[i] += base;
  }
  __syncthreads();
}

// One pass of the sort for one tile.  FIRST: keys from `keys_in`, the values are the items' own
// indices; otherwise (key, value) pairs from `pairs_in`.  LAST: only the values are written.  Dynamic LDS: cnt u16 [WAVES][bins] | delta i32 [bins] |
// stage_k u32 [TILE] | stage_v i32 [TILE].
template <int ITEMS, bool FIRST, bool LAST>
__global__ __launch_bounds__(BLOCK) void k_sort_scatter(
    const unsigned* __restrict__ keys_in, const uint2* __restrict__ pairs_in, int n, int shift,
    int bits, const unsigned* __restrict__ hist, const unsigned* __restrict__ seg,
    const unsigned* __restrict__ dtotal, uint2* __restrict__ pairs_out,
    int32_t* __restrict__ vals_out) {
  constexpr int TILE = BLOCK * ITEMS;
  extern __shared__ unsigned lds[];
  __shared__ unsigned wsum[WAVES];
  const int bins = 1 << bits;
  uint16_t* cnt = reinterpret_cast<uint16_t*>(lds);                    // [WAVES][bins]
  int32_t* delta = reinterpret_cast<int32_t*>(lds + (WAVES * bins) / 2);  // [bins]
  unsigned* stage_k = lds + (WAVES * bins) / 2 + bins;
  int32_t* stage_v = reinterpret_cast<int32_t*>(stage_k + TILE);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int base = blockIdx.x * TILE;
  const int wbase = base + wave * (64 * ITEMS);
  const unsigned dmask = (unsigned)(bins - 1);

  unsigned key[ITEMS];
  int32_t val[ITEMS];
  unsigned dr[ITEMS];  // digit | rank << 16
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const int i = wbase + r * 64 + lane;
    const bool ok = i < n;
    if constexpr (FIRST) {
      key[r] = ok ? keys_in[i] : 0xFFFFFFFFu;
      val[r] = i;
    } else {
      const uint2 kv = ok ? pairs_in[i] : make_uint2(0xFFFFFFFFu, 0u);
      key[r] = kv.x;
      val[r] = (int32_t)kv.y;
    }
    // (a slot past the end ranks as the largest digit: it is also last by position, so it ends
    // up behind every item of the tile and is simply not written)
    dr[r] = ok ? ((key[r] >> shift) & dmask) : dmask;
  }
  for (int w = tid; w < (WAVES * bins) / 2; w += BLOCK) lds[w] = 0u;
  __syncthreads();

  // rank of every item among the items of its wave with the same digit (stable: rounds in
  // order, lanes in order)
  uint16_t* my = cnt + wave * bins;
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const unsigned d = dr[r];
    unsigned long long m = ~0ull;
    for (int b = 0; b < bits; ++b) {
      const bool bit = (d >> b) & 1u;
      const unsigned long long bal = __ballot(bit);
      m &= bit ? bal : ~bal;
    }
    const int below = rank_below(m);
    const unsigned c = my[d];
    wave_fence();
    if (below == 0) my[d] = (uint16_t)(c + (unsigned)__popcll(m));
    wave_fence();
    dr[r] = d | ((c + (unsigned)below) << 16);
  }
  __syncthreads();

  // start of every (digit, wave) run within the tile, and the digit's distance to its place in
  // the output: (items of smaller digits in all tiles: a scan of the digits' totals, which every
  // tile does for itself) + (items of this digit in the tiles before: the column segments before
  // this tile's + its row of hist)
  {
    const int per = bins >= BLOCK ? bins / BLOCK : 1;
    const int d0 = tid * per;
    const int myseg = blockIdx.x / COL_SEG;
    unsigned tsum = 0, gsum = 0;
    if (d0 < bins) {
      for (int q = 0; q < per; ++q) {
        const int d = d0 + q;
        // wave w's start within the digit's run: exclusive prefixes in cnt[1..3]; cnt[0] (always
        // 0) keeps the digit's count in this tile until the second loop
        unsigned run = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
          const unsigned c = cnt[w * bins + d];
          cnt[w * bins + d] = (uint16_t)run;
          run += c;
        }
        cnt[d] = (uint16_t)run;
        tsum += run;
        // (for now: items of this thread's smaller digits + of this digit in the tiles before)
        delta[d] = (int32_t)(gsum + seg[(int64_t)myseg * bins + d] +
                             hist[(int64_t)blockIdx.x * bins + d]);
        gsum += dtotal[d];
      }
    }
    unsigned total;
    unsigned dbase = block_exclusive(tsum, wsum, &total);
    const unsigned gbase = block_exclusive(gsum, wsum, &total);
    if (d0 < bins) {
      for (int q = 0; q < per; ++q) {
        const int d = d0 + q;
        const unsigned c = cnt[d];
        cnt[d] = (uint16_t)dbase;
#pragma unroll
        for (int w = 1; w < WAVES; ++w) cnt[w * bins + d] = (uint16_t)(cnt[w * bins + d] + dbase);
        delta[d] = (int32_t)(gbase + (unsigned)delta[d]) - (int32_t)dbase;
        dbase += c;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const unsigned d = dr[r] & 0xFFFFu;
    const int pos = (int)cnt[wave * bins + d] + (int)(dr[r] >> 16);
    stage_k[pos] = key[r];
    stage_v[pos] = val[r];
  }
  __syncthreads();
  const int nvalid = min(TILE, n - base);
  for (int p = tid; p < nvalid; p += BLOCK) {
    const unsigned k = stage_k[p];
    const int dest = p + delta[(k >> shift) & dmask];
    // (key and value travel as one 8-byte store: the writes are scattered, and their number is
    // what this kernel costs)
    if constexpr (!LAST) pairs_out[dest] = make_uint2(k, (unsigned)stage_v[p]);
    else vals_out[dest] = stage_v[p];
  }
}

// ------------------------------------------------------------------------------ permutations

// Second half of the most-significant-digit-first sort (tfrt_source3d_order_cells): workgroup d
// takes the bucket of high digit d -- consecutive pairs after the stable scatter by that digit --
// and places its items by their low digit: a histogram in LDS, its scan, one atomic counter per
// digit.  Any bucket size (a source whose rays all fall in one bucket is sorted by one workgroup:
// slow, still right); items with the same key land in an order that may vary from run to run.
__global__ __launch_bounds__(BLOCK) void k_bucket_sort(const uint2* __restrict__ pairs, int bits,
                                                       const unsigned* __restrict__ dtotal,
                                                       int32_t* __restrict__ perm) {
  extern __shared__ unsigned lds[];   // low-digit counters [bins] | first places of the buckets [bins]
  __shared__ unsigned wsum[WAVES];
  const int bins = 1 << bits, tid = threadIdx.x;
  unsigned* cnt = lds;
  unsigned* pre = lds + bins;
  lds_exclusive_from(dtotal, bins, pre, wsum);
  const unsigned start = pre[blockIdx.x], len = dtotal[blockIdx.x];
  if (len == 0u) return;   // (block-uniform)
  for (int k = tid; k < bins; k += BLOCK) cnt[k] = 0u;
  __syncthreads();
  const unsigned mask = (unsigned)(bins - 1);
  // (a bucket normally holds n / bins items, a thousand at a million rays: a thread's first KEEP
  // items stay in registers between the two loops, their loads all in flight together)
  constexpr int KEEP = 8;
  uint2 mine[KEEP];
#pragma unroll
  for (int q = 0; q < KEEP; ++q) {
    const unsigned i = tid + q * BLOCK;
    mine[q] = i < len ? pairs[start + i] : make_uint2(0u, 0u);
  }
#pragma unroll
  for (int q = 0; q < KEEP; ++q)
    if (tid + q * BLOCK < len) atomicAdd(&cnt[mine[q].x & mask], 1u);
  for (unsigned i = tid + KEEP * BLOCK; i < len; i += BLOCK)
    atomicAdd(&cnt[pairs[start + i].x & mask], 1u);
  __syncthreads();
  {  // exclusive scan of the counters, in place (a thread's own run of consecutive digits)
    const int per = (bins + BLOCK - 1) / BLOCK;
    const int lo = tid * per;
    unsigned sum = 0;
    for (int q = 0; q < per; ++q) {
      const int i = lo + q;
      if (i < bins) {
        const unsigned t = cnt[i];
        cnt[i] = sum;
        sum += t;
      }
    }
    unsigned total;
    const unsigned base = block_exclusive(sum, wsum, &total);
    for (int q = 0; q < per; ++q) {
      const int i = lo + q;
      if (i < bins) cnt[i] += base;
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < KEEP; ++q) {
    if (tid + q * BLOCK < len) {
      const unsigned pos = atomicAdd(&cnt[mine[q].x & mask], 1u);
      perm[start + pos] = (int32_t)mine[q].y;
    }
  }
  for (unsigned i = tid + KEEP * BLOCK; i < len; i += BLOCK) {
    const uint2 p = pairs[start + i];
    const unsigned pos = atomicAdd(&cnt[p.x & mask], 1u);
    perm[start + pos] = (int32_t)p.y;
  }
}

template <int BYTES>
struct Vec;
template <>
struct Vec<16> { using type = uint4; };
template <>
struct Vec<32> { struct alignas(16) type { uint4 a, b; }; };
template <>
struct Vec<64> { struct alignas(16) type { uint4 a, b, c, d; }; };

// ray block (SoA) -> records of 8 elements (six used), one ray each
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_rays_to_records(const T* __restrict__ rays,
                                                           int64_t stride, int n,
                                                           T* __restrict__ rec) {
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  using V = typename Vec<8 * sizeof(T)>::type;
  T e[8];
  V v;
#pragma unroll
  for (int q = 0; q < 6; ++q) e[q] = rays[q * stride + i];
  e[6] = e[7] = T(0);
  __builtin_memcpy(&v, e, sizeof(V));
  reinterpret_cast<V*>(rec)[i] = v;
}

// records, through an index -> ray block: dst[:, j] = record[index[j]]
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_records_to_rays(const T* __restrict__ rec,
                                                           const int32_t* __restrict__ index,
                                                           int n, T* __restrict__ dst,
                                                           int64_t dstride) {
  const int j = blockIdx.x * BLOCK + threadIdx.x;
  if (j >= n) return;
  using V = typename Vec<8 * sizeof(T)>::type;
  T e[8];
  const V v = reinterpret_cast<const V*>(rec)[index[j]];
  __builtin_memcpy(e, &v, sizeof(V));
#pragma unroll
  for (int q = 0; q < 6; ++q) dst[q * dstride + j] = e[q];
}

// dst[k][j] = src[k][index[j]], k < rows (elements of E bytes)
template <typename E>
__global__ __launch_bounds__(BLOCK) void k_gather_rows(const E* __restrict__ src, int64_t sstride,
                                                       int rows, const int32_t* __restrict__ index,
                                                       int64_t n,
                                                       const int32_t* __restrict__ n_valid,
                                                       E* __restrict__ dst, int64_t dstride) {
  const int64_t j = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (j >= n || (n_valid != nullptr && j >= *n_valid)) return;
  const int64_t i = index[j];
  for (int k = 0; k < rows; ++k) dst[k * dstride + j] = src[k * sstride + i];
}

// ------------------------------------------------------------------------------ the way back
//
// One output class of a trace over permuted rays lists, pass after pass, the rays of that pass in
// the PERMUTED order; the reference lists them by original index.  Rows carry unique keys
// (pass, original id): set the key's bit in a bitmap, scan the words' popcounts, and a row's place
// is the number of bits before its own -- a counting sort, three small launches.

struct Segments {  // the passes' rows of one class: n_seg segments [base, base + n)
  const int32_t* n;
  const int32_t* base;
  int32_t stride, n_seg;
  const int32_t* total;  // rows of the class (device), or NULL: `rows`
  int32_t rows;
};

constexpr int SEG_LDS = 1024;

// segment of row r (binary search over the bases); -1: r is past the last row
__device__ __forceinline__ int segment_of(const Segments& sg, const int32_t* sbase,
                                          const int32_t* sn, int r) {
  if (sg.n == nullptr) return r < sg.rows ? 0 : -1;
  int lo = 0, hi = sg.n_seg - 1;
  while (lo < hi) {  // last segment with base <= r
    const int mid = (lo + hi + 1) >> 1;
    if (sbase[mid] <= r) lo = mid;
    else hi = mid - 1;
  }
  // (empty segments share their base with the next one: step to the one that holds r)
  while (lo < sg.n_seg && r >= sbase[lo] + sn[lo]) ++lo;
  return lo < sg.n_seg ? lo : -1;
}

template <int STAGE>  // 0: mark bits, 1: place rows
__global__ __launch_bounds__(BLOCK) void k_restore_rows(
    Segments sg, const int32_t* __restrict__ ray_id, const int32_t* __restrict__ perm, int n_src,
    int wn, uint2* __restrict__ words, const unsigned* __restrict__ totals, int nchunks,
    int32_t* __restrict__ inv, int32_t* __restrict__ dest_of, int32_t* __restrict__ id_out) {
  extern __shared__ unsigned lds[];
  __shared__ unsigned wsum[WAVES];
  __shared__ int32_t sbase[SEG_LDS], sn[SEG_LDS];
  const int nseg = sg.n == nullptr ? 0 : min(sg.n_seg, SEG_LDS);
  for (int k = threadIdx.x; k < nseg; k += BLOCK) {
    sbase[k] = sg.base[(int64_t)k * sg.stride];
    sn[k] = sg.n[(int64_t)k * sg.stride];
  }
  if constexpr (STAGE == 1) lds_exclusive_from(totals, nchunks, lds, wsum);
  else __syncthreads();
  const int rows = sg.total != nullptr ? *sg.total : sg.rows;
  const int r = blockIdx.x * BLOCK + threadIdx.x;
  if (r >= rows) return;
  Segments s2 = sg;
  s2.n_seg = nseg;
  const int p = segment_of(s2, sbase, sn, r);
  if (p < 0) return;
  const int id = ray_id[r];
  const int orig = perm != nullptr ? perm[id] : id;
  const int64_t w = (int64_t)p * wn + (orig >> 5);
  const unsigned bit = 1u << (orig & 31);
  if constexpr (STAGE == 0) {
    atomicOr(&words[w].x, bit);
  } else {
    const uint2 wd = words[w];
    const int dest = (int)(lds[w / SCAN_CHUNK] + wd.y + (unsigned)__popc(wd.x & (bit - 1u)));
    if (inv != nullptr) inv[dest] = r;
    if (dest_of != nullptr) dest_of[r] = dest;
    if (id_out != nullptr) id_out[dest] = orig;
  }
}

// words[i].y <- popcounts of words[chunk start .. i).x, totals[chunk] <- the chunk's popcount
__global__ __launch_bounds__(BLOCK) void k_restore_scan(uint2* __restrict__ words, int64_t len,
                                                        unsigned* __restrict__ totals) {
  __shared__ unsigned wsum[WAVES];
  constexpr int PER = SCAN_CHUNK / BLOCK;
  const int64_t i0 = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * PER;
  unsigned v[PER];
  unsigned sum = 0;
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const unsigned c = (i0 + q < len) ? (unsigned)__popc(words[i0 + q].x) : 0u;
    v[q] = sum;
    sum += c;
  }
  unsigned total;
  const unsigned base = block_exclusive(sum, wsum, &total);
#pragma unroll
  for (int q = 0; q < PER; ++q)
    if (i0 + q < len) words[i0 + q].y = v[q] + base;
  if (threadIdx.x == 0) totals[blockIdx.x] = total;
}

// ------------------------------------------------------------------------------ face clusters
//
// tfrt_cluster_order: the permutation of the faces that makes every aligned run of `leaf`
// consecutive entries (and of `group` entries: the trace kernels' clusters and superclusters) a
// compact patch -- a k-d ordering: recursive median split of the face centroids along the longest
// axis of their bounding box, the left part a multiple of `group` (of `leaf` below that), down to
// parts of <= `leaf` faces; outsized faces (a target plane behind a fine lens mesh) go last, or
// every ray would test the fifteen small faces that happen to share a huge one's cluster.
// Level by level on the device: every position carries its segment (start, length, path in the
// tree); per level the segments' bounding boxes (atomics), a key (path, quantised coordinate along
// the segment's longest axis), the stable radix sort of those keys -- which sorts inside every
// segment at once, because the path is monotone in the position --, and the split of every
// segment at its aligned median.  The split sizes depend on the lengths alone, so nothing is read
// back: no host sync.

constexpr int CO_SIZE_BINS = 4096;   // positive float32 >> 19: exponent + 4 mantissa bits

__global__ __launch_bounds__(BLOCK) void k_co_prepare(const double* __restrict__ fverts, int M,
                                                      float4* __restrict__ cent,
                                                      unsigned* __restrict__ size_hist) {
  const int f = blockIdx.x * BLOCK + threadIdx.x;
  if (f >= M) return;
  const double* v = fverts + 9 * (int64_t)f;
  const double cx = (v[0] + v[3] + v[6]) / 3.0, cy = (v[1] + v[4] + v[7]) / 3.0,
               cz = (v[2] + v[5] + v[8]) / 3.0;
  double r2 = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double dx = v[3 * k] - cx, dy = v[3 * k + 1] - cy, dz = v[3 * k + 2] - cz;
    r2 = fmax(r2, dx * dx + dy * dy + dz * dz);
  }
  const float size = (float)sqrt(r2);
  cent[f] = make_float4((float)cx, (float)cy, (float)cz, size);
  unsigned bin = __float_as_uint(size) >> 19;
  if (!(size >= 0.f) || bin >= CO_SIZE_BINS) bin = CO_SIZE_BINS - 1;   // (NaN / inf: the last bin)
  atomicAdd(&size_hist[bin], 1u);
}

// thr[0] <- 8 x (upper edge of the bin that holds the median size); counters cleared
__global__ __launch_bounds__(BLOCK) void k_co_threshold(const unsigned* __restrict__ size_hist,
                                                        int M, float* __restrict__ thr,
                                                        int32_t* __restrict__ n_big) {
  __shared__ unsigned wsum[WAVES];
  constexpr int PER = CO_SIZE_BINS / BLOCK;
  unsigned v[PER], sum = 0;
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    v[q] = size_hist[threadIdx.x * PER + q];
    sum += v[q];
  }
  unsigned total;
  unsigned run = block_exclusive(sum, wsum, &total);
  const unsigned half = ((unsigned)M + 1u) / 2u;
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    if (run < half && run + v[q] >= half) {
      const unsigned bin = threadIdx.x * PER + q;
      thr[0] = 8.0f * __uint_as_float((bin + 1u) << 19);
    }
    run += v[q];
  }
  if (threadIdx.x == 0) n_big[0] = 0;
}

__global__ __launch_bounds__(BLOCK) void k_co_flag(const float4* __restrict__ cent, int M,
                                                   const float* __restrict__ thr,
                                                   unsigned* __restrict__ keys,
                                                   int32_t* __restrict__ n_big) {
  const int f = blockIdx.x * BLOCK + threadIdx.x;
  bool big = false;
  if (f < M) {
    big = !(cent[f].w <= thr[0]);
    keys[f] = big ? 1u : 0u;
  }
  const unsigned long long m = __ballot(big);
  if ((threadIdx.x & 63) == 0 && m != 0ull) atomicAdd(n_big, __popcll(m));
}

__global__ __launch_bounds__(BLOCK) void k_co_init(int M, const int32_t* __restrict__ n_big,
                                                   int32_t* __restrict__ seg_start,
                                                   int32_t* __restrict__ seg_len,
                                                   unsigned* __restrict__ path) {
  const int pos = blockIdx.x * BLOCK + threadIdx.x;
  if (pos >= M) return;
  const int n_small = M - *n_big;
  const bool small = pos < n_small;
  seg_start[pos] = small ? 0 : n_small;
  seg_len[pos] = small ? n_small : 0;      // (0: never split)
  path[pos] = small ? 0u : 1u;
}

// bounding boxes of the segments still to be split: bbox[6 * start + q], running minima of
// enc(x), ~enc(x), enc(y), ...
__global__ __launch_bounds__(BLOCK) void k_co_bbox(const float4* __restrict__ cent,
                                                   const int32_t* __restrict__ idx,
                                                   const int32_t* __restrict__ seg_start,
                                                   const int32_t* __restrict__ seg_len, int leaf,
                                                   int M, unsigned* __restrict__ bbox) {
  const int pos = blockIdx.x * BLOCK + threadIdx.x;
  const bool on = pos < M && seg_len[pos] > leaf;
  const int start = on ? seg_start[pos] : -1;
  unsigned v[6];
  if (on) {
    const float4 c = cent[idx[pos]];
    v[0] = enc_f(c.x); v[1] = ~enc_f(c.x);
    v[2] = enc_f(c.y); v[3] = ~enc_f(c.y);
    v[4] = enc_f(c.z); v[5] = ~enc_f(c.z);
  } else {
#pragma unroll
    for (int q = 0; q < 6; ++q) v[q] = 0xFFFFFFFFu;
  }
  // a wave whose lanes all belong to one segment (the rule once segments are long) folds its
  // minima first: six atomics per wave instead of six per lane on the same six addresses
  const int first = __shfl(start, 0, 64);
  const bool same = __all(start == first || !on) && first >= 0;
  if (same) {
#pragma unroll
    for (int q = 0; q < 6; ++q) {
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) v[q] = min(v[q], (unsigned)__shfl_xor((int)v[q], d, 64));
    }
    const int lane = threadIdx.x & 63;
    unsigned mine = v[0];
#pragma unroll
    for (int q = 1; q < 6; ++q) mine = lane == q ? v[q] : mine;
    if (lane < 6 && mine != 0xFFFFFFFFu) atomicMin(&bbox[6 * (int64_t)first + lane], mine);
  } else if (on) {
#pragma unroll
    for (int q = 0; q < 6; ++q) atomicMin(&bbox[6 * (int64_t)start + q], v[q]);
  }
}

__global__ __launch_bounds__(BLOCK) void k_co_key(const float4* __restrict__ cent,
                                                  const int32_t* __restrict__ idx,
                                                  const int32_t* __restrict__ seg_start,
                                                  const int32_t* __restrict__ seg_len,
                                                  const unsigned* __restrict__ path,
                                                  const unsigned* __restrict__ bbox, int leaf,
                                                  int cbits, int M, unsigned* __restrict__ keys) {
  const int pos = blockIdx.x * BLOCK + threadIdx.x;
  if (pos >= M) return;
  unsigned q = 0;
  if (seg_len[pos] > leaf) {
    const unsigned* b = bbox + 6 * (int64_t)seg_start[pos];
    const float lo[3] = {dec_f(b[0]), dec_f(b[2]), dec_f(b[4])};
    const float hi[3] = {dec_f(~b[1]), dec_f(~b[3]), dec_f(~b[5])};
    int axis = 0;
    float ext = hi[0] - lo[0];
    if (hi[1] - lo[1] > ext) { axis = 1; ext = hi[1] - lo[1]; }
    if (hi[2] - lo[2] > ext) { axis = 2; ext = hi[2] - lo[2]; }
    const float4 c = cent[idx[pos]];
    const float x = axis == 0 ? c.x : (axis == 1 ? c.y : c.z);
    const float top = (float)((1u << cbits) - 1u);
    const float t = ext > 0.f ? (x - lo[axis]) / ext * top : 0.f;
    q = (unsigned)fminf(fmaxf(t, 0.f), top);      // (NaN: 0)
  }
  keys[pos] = (path[pos] << cbits) | q;
}

__global__ __launch_bounds__(BLOCK) void k_co_apply(const int32_t* __restrict__ idx_in,
                                                    const int32_t* __restrict__ perm, int M,
                                                    int32_t* __restrict__ idx_out) {
  const int pos = blockIdx.x * BLOCK + threadIdx.x;
  if (pos < M) idx_out[pos] = idx_in != nullptr ? idx_in[perm[pos]] : perm[pos];
}

__global__ __launch_bounds__(BLOCK) void k_co_split(int32_t* __restrict__ seg_start,
                                                    int32_t* __restrict__ seg_len,
                                                    unsigned* __restrict__ path, int leaf,
                                                    int group, int M) {
  const int pos = blockIdx.x * BLOCK + threadIdx.x;
  if (pos >= M) return;
  const int len = seg_len[pos];
  unsigned p = path[pos] << 1;
  if (len > leaf) {
    const int start = seg_start[pos];
    const int unit = len > group ? group : leaf;
    const int n_left = unit * (((len + unit - 1) / unit) / 2);
    if (pos - start < n_left) {
      seg_len[pos] = n_left;
    } else {
      seg_start[pos] = start + n_left;
      seg_len[pos] = len - n_left;
      p |= 1u;
    }
  }
  path[pos] = p;
}

// ------------------------------------------------------------------------------ host side

static int order_bits(int64_t n) {
  int lg = 0;
  while ((1ll << lg) < n) ++lg;       // ceil(log2 n)
  int b = (lg + 1) / 2;               // about one cell per ray
  if (b < 4) b = 4;
  if (b > 13) b = 13;
  return b;
}
// items per thread of a sort tile: big tiles make long runs of equal digits (the scatter's writes
// are what it costs), small ones fill the chip when the rays are few
static int order_items(int64_t n) { return n < (128 << 10) ? 4 : 8; }
// (16 items per thread -- 244 tiles at a million rays, one wavefront per SIMD -- measured slower
// than 8 with every digit width: the fused step over a re-drawn source 0.521 -> 0.513 ms)

struct OrderLayout {
  size_t head, xy, keys_a, pairs, hist, seg, dtotal, total;
  int bits, items, nblk, nseg;
};

static OrderLayout order_layout(int64_t n, int bits = -1) {
  OrderLayout L;
  const size_t m = n > 0 ? (size_t)n : 1;
  L.bits = bits > 0 ? bits : order_bits(n);
  L.items = order_items(n);
  L.nblk = cdiv((int64_t)m, (int64_t)BLOCK * L.items);
  const int64_t hlen = (int64_t)(1 << L.bits) * L.nblk;
  L.nseg = cdiv(L.nblk, COL_SEG);
  size_t o = 0;
  auto take = [&](size_t bytes) {
    size_t at = o;
    o = align_up(o + bytes);
    return at;
  };
  L.head = take(MM_SLOTS * 16 + sizeof(OrderFrame));   // extents (MM_SLOTS x 4 u32) | frame
  L.xy = take(m * sizeof(float2));
  L.keys_a = take(m * sizeof(unsigned));
  L.pairs = take(m * sizeof(uint2));
  L.hist = take((size_t)hlen * sizeof(unsigned));
  L.seg = take((size_t)L.nseg * ((size_t)1 << L.bits) * sizeof(unsigned));
  L.dtotal = take(((size_t)1 << L.bits) * sizeof(unsigned));
  L.total = o;
  return L;
}

static size_t scatter_lds_bytes(int bits, int items) {
  const size_t bins = (size_t)1 << bits;
  return (WAVES * bins) * 2 + bins * 4 + (size_t)BLOCK * items * 8;
}

template <int ITEMS>
static int sort_passes(const OrderLayout& L, char* ws, int n, int32_t* perm, unsigned* keys_nat,
                       hipStream_t st) {
  uint2* pairs = reinterpret_cast<uint2*>(ws + L.pairs);
  unsigned* hist = reinterpret_cast<unsigned*>(ws + L.hist);
  unsigned* seg = reinterpret_cast<unsigned*>(ws + L.seg);
  unsigned* dtotal = reinterpret_cast<unsigned*>(ws + L.dtotal);
  const int bins = 1 << L.bits;
  const size_t lds = scatter_lds_bytes(L.bits, ITEMS);
  if (lds > 160 * 1024) return TFRT_E_UNSUPPORTED;
  const dim3 cgrid(cdiv(bins, BLOCK), L.nseg);
  // pass 0 (its histogram came with the keys)
  hipLaunchKernelGGL(k_colscan_rows, cgrid, dim3(BLOCK), 0, st, hist, L.nblk, bins, seg);
  hipLaunchKernelGGL(k_colscan_segs, dim3(cgrid.x), dim3(BLOCK), 0, st, seg, L.nseg, bins, dtotal);
  hipLaunchKernelGGL((k_sort_scatter<ITEMS, true, false>), dim3(L.nblk), dim3(BLOCK), lds, st,
                     keys_nat, static_cast<const uint2*>(nullptr), n, 0, L.bits, hist, seg,
                     dtotal, pairs, static_cast<int32_t*>(nullptr));
  // pass 1
  hipLaunchKernelGGL((k_sort_hist<ITEMS>), dim3(L.nblk), dim3(BLOCK), bins * sizeof(unsigned), st,
                     pairs, n, L.bits, L.bits, hist, L.nblk);
  hipLaunchKernelGGL(k_colscan_rows, cgrid, dim3(BLOCK), 0, st, hist, L.nblk, bins, seg);
  hipLaunchKernelGGL(k_colscan_segs, dim3(cgrid.x), dim3(BLOCK), 0, st, seg, L.nseg, bins, dtotal);
  hipLaunchKernelGGL((k_sort_scatter<ITEMS, false, true>), dim3(L.nblk), dim3(BLOCK), lds, st,
                     static_cast<const unsigned*>(nullptr), pairs, n, L.bits, L.bits, hist, seg,
                     dtotal, static_cast<uint2*>(nullptr), perm);
  return 0;
}

// perm = argsort(keys, stable) for keys of 2 * L.bits bits
static int sort_keys(const OrderLayout& L, char* ws, int n, unsigned* keys, int32_t* perm,
                     hipStream_t st) {
  unsigned* hist = reinterpret_cast<unsigned*>(ws + L.hist);
  const size_t hl = ((size_t)1 << L.bits) * sizeof(unsigned);
  if (L.items == 4) {
    hipLaunchKernelGGL((k_sort_hist_keys<4>), dim3(L.nblk), dim3(BLOCK), hl, st, keys, n, L.bits, hist);
    return sort_passes<4>(L, ws, n, perm, keys, st);
  }
  hipLaunchKernelGGL((k_sort_hist_keys<8>), dim3(L.nblk), dim3(BLOCK), hl, st, keys, n, L.bits, hist);
  return sort_passes<8>(L, ws, n, perm, keys, st);
}

template <typename R>
static int ray_order_t(const R& src, int64_t N, const double* fverts, int64_t M,
                       const double* axis, int32_t* perm, uint32_t* keys_out, char* ws,
                       const OrderLayout& L, hipStream_t st, bool cells = false) {
  const int n = (int)N;
  unsigned* mm = reinterpret_cast<unsigned*>(ws + L.head);
  OrderFrame* frame = reinterpret_cast<OrderFrame*>(ws + L.head + MM_SLOTS * 16);
  float2* xy = reinterpret_cast<float2*>(ws + L.xy);
  unsigned* keys = keys_out != nullptr ? keys_out : reinterpret_cast<unsigned*>(ws + L.keys_a);
  unsigned* hist = reinterpret_cast<unsigned*>(ws + L.hist);
  const double a0 = axis ? axis[0] : 0.0, a1 = axis ? axis[1] : 0.0, a2 = axis ? axis[2] : 0.0;
  if constexpr (!R::HAS_F32) {
    hipLaunchKernelGGL((k_order_frame<R>), dim3(1), dim3(BLOCK), 0, st, src, n, fverts, (int)M, a0,
                       a1, a2, axis ? 1 : 0, frame, mm);
    hipLaunchKernelGGL((k_order_xy<R>), dim3(cdiv(N, BLOCK)), dim3(BLOCK), 0, st, src, n, frame,
                       xy, mm);
  }
  const size_t hl = ((size_t)1 << L.bits) * sizeof(unsigned);
  int rc = 0;
  if constexpr (R::HAS_F32) {
    if (cells) {
      // most significant digit first: one stable scatter by the high digit (its histogram comes
      // with the keys), then every bucket of it is sorted by the low digit in LDS
      uint2* pairs = reinterpret_cast<uint2*>(ws + L.pairs);
      unsigned* seg = reinterpret_cast<unsigned*>(ws + L.seg);
      unsigned* dtotal = reinterpret_cast<unsigned*>(ws + L.dtotal);
      const int bins = 1 << L.bits;
      const size_t lds = scatter_lds_bytes(L.bits, L.items);
      if (lds > 160 * 1024) return TFRT_E_UNSUPPORTED;
      const dim3 cgrid(cdiv(bins, BLOCK), L.nseg);
#define TFRT_ORDER_MSD(I)                                                                          \
  {                                                                                                \
    hipLaunchKernelGGL((k_order_pkey<I, true>), dim3(L.nblk), dim3(BLOCK), hl, st, src, n, fverts, \
                       (int)M, a0, a1, a2, axis ? 1 : 0, L.bits, keys, hist, L.nblk);              \
    hipLaunchKernelGGL(k_colscan_rows, cgrid, dim3(BLOCK), 0, st, hist, L.nblk, bins, seg);        \
    hipLaunchKernelGGL(k_colscan_segs, dim3(cgrid.x), dim3(BLOCK), 0, st, seg, L.nseg, bins,       \
                       dtotal);                                                                    \
    hipLaunchKernelGGL((k_sort_scatter<I, true, false>), dim3(L.nblk), dim3(BLOCK), lds, st, keys, \
                       static_cast<const uint2*>(nullptr), n, L.bits, L.bits, hist, seg, dtotal,   \
                       pairs, static_cast<int32_t*>(nullptr));                                     \
  }
      if (L.items == 4) TFRT_ORDER_MSD(4)
      else TFRT_ORDER_MSD(8)
#undef TFRT_ORDER_MSD
      hipLaunchKernelGGL(k_bucket_sort, dim3(bins), dim3(BLOCK), 2 * bins * sizeof(unsigned), st,
                         pairs, L.bits, dtotal, perm);
      return 0;
    }
  }
#define TFRT_ORDER_ITEMS(I)                                                                       \
  {                                                                                               \
    if constexpr (R::HAS_F32)   /* a program's rays: frame, extents and keys in one launch */     \
      hipLaunchKernelGGL((k_order_pkey<I, false>), dim3(L.nblk), dim3(BLOCK), hl, st, src, n,     \
                         fverts, (int)M, a0, a1, a2, axis ? 1 : 0, L.bits, keys, hist, L.nblk);    \
    else                                                                                          \
      hipLaunchKernelGGL((k_order_key<I>), dim3(L.nblk), dim3(BLOCK), hl, st, xy, n, mm, L.bits,  \
                         keys, hist, L.nblk);                                                     \
    rc = sort_passes<I>(L, ws, n, perm, keys, st);                                                \
  }
  if (L.items == 4) TFRT_ORDER_ITEMS(4)
  else TFRT_ORDER_ITEMS(8)
#undef TFRT_ORDER_ITEMS
  return rc;
}

template <typename T>
static void permute_rays_t(const void* src, int64_t sstride, int64_t n, const int32_t* perm,
                           void* dst, int64_t dstride, void* ws, hipStream_t st) {
  T* rec = static_cast<T*>(ws);
  hipLaunchKernelGGL((k_rays_to_records<T>), dim3(cdiv(n, BLOCK)), dim3(BLOCK), 0, st,
                     static_cast<const T*>(src), sstride, (int)n, rec);
  hipLaunchKernelGGL((k_records_to_rays<T>), dim3(cdiv(n, BLOCK)), dim3(BLOCK), 0, st, rec, perm,
                     (int)n, static_cast<T*>(dst), dstride);
}

struct RestoreLayout {
  size_t words, totals, total;
  int wn, nchunks;
  int64_t len;
};
static RestoreLayout restore_layout(int64_t n_src, int64_t n_seg) {
  RestoreLayout L;
  L.wn = (int)((n_src + 31) / 32);
  if (L.wn < 1) L.wn = 1;
  L.len = (int64_t)L.wn * (n_seg > 0 ? n_seg : 1);
  L.nchunks = cdiv(L.len, SCAN_CHUNK);
  L.words = 0;
  L.totals = align_up((size_t)L.len * sizeof(uint2));
  L.total = L.totals + align_up((size_t)L.nchunks * sizeof(unsigned));
  return L;
}

static int cluster_levels(int64_t M, int leaf, int group) {
  int L = 0;
  int64_t m = M;
  while (m > leaf) {
    const int64_t unit = m > group ? group : leaf;
    const int64_t n_left = unit * (((m + unit - 1) / unit) / 2);
    m = n_left > m - n_left ? n_left : m - n_left;
    ++L;
  }
  return L;
}

struct ClusterLayout {
  size_t cent, size_hist, scalars, idx_a, idx_b, seg_start, seg_len, path, bbox, keys, perm, sort, total;
  OrderLayout S;
  int levels;
};

static ClusterLayout cluster_layout(int64_t M, int leaf, int group) {
  ClusterLayout L;
  const size_t m = M > 0 ? (size_t)M : 1;
  L.levels = cluster_levels(M, leaf, group);
  L.S = order_layout(M, 13);     // keys of 26 bits
  size_t o = 0;
  auto take = [&](size_t bytes) {
    size_t at = o;
    o = align_up(o + bytes);
    return at;
  };
  L.cent = take(m * sizeof(float4));
  L.size_hist = take(CO_SIZE_BINS * sizeof(unsigned));
  L.scalars = take(64);            // thr (float) | n_big (int32)
  L.idx_a = take(m * 4);
  L.idx_b = take(m * 4);
  L.seg_start = take(m * 4);
  L.seg_len = take(m * 4);
  L.path = take(m * 4);
  L.bbox = take(m * 24);
  L.keys = take(m * 4);
  L.perm = take(m * 4);
  L.sort = take(L.S.total);
  L.total = o;
  return L;
}

}  // namespace tfrt

// ================================================================================ C ABI
using namespace tfrt;

extern "C" {

size_t tfrt_ray_order_workspace_bytes(int64_t n_rays) { return order_layout(n_rays).total; }

int tfrt_ray_order(const void* rays, int64_t stride, int64_t n_rays, int32_t state_dtype,
                   const double* face_verts, int64_t n_faces, const double* axis, int32_t* perm,
                   uint32_t* keys_out, void* workspace, size_t workspace_bytes, void* stream) {
  if (n_rays < 0 || n_rays >= (1ll << 31) || n_faces < 0 || n_faces >= (1ll << 31))
    return TFRT_E_BADARG;
  if (n_rays == 0) return 0;
  if (!rays || !perm || !workspace || stride < n_rays) return TFRT_E_BADARG;
  const OrderLayout L = order_layout(n_rays);
  if (workspace_bytes < L.total) return TFRT_E_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* ws = static_cast<char*>(workspace);
  int rc;
  switch (state_dtype) {
    case TFRT_F32:
      rc = ray_order_t(BlockRays<float>{static_cast<const float*>(rays), stride}, n_rays,
                       face_verts, n_faces, axis, perm, keys_out, ws, L, st);
      break;
    case TFRT_F64:
      rc = ray_order_t(BlockRays<double>{static_cast<const double*>(rays), stride}, n_rays,
                       face_verts, n_faces, axis, perm, keys_out, ws, L, st);
      break;
    case TFRT_F16:
      rc = ray_order_t(BlockRays<_Float16>{static_cast<const _Float16*>(rays), stride}, n_rays,
                       face_verts, n_faces, axis, perm, keys_out, ws, L, st);
      break;
    default:
      return TFRT_E_BADARG;
  }
  if (rc != 0) return rc;
  return hipGetLastError() == hipSuccess ? 0 : TFRT_E_LAUNCH;
}

static int source3d_order(const tfrt_source3d_program* program, int64_t first, int64_t n_rays,
                          const double* face_verts, int64_t n_faces, const double* axis,
                          int32_t* perm, uint32_t* keys_out, void* workspace,
                          size_t workspace_bytes, void* stream, bool cells) {
  // (the same validation as tfrt_source3d_generate: k_order_pkey evaluates the program)
  if (!source_program_ok(program) || n_rays < 0 || n_rays >= (1ll << 31) || n_faces < 0 ||
      n_faces >= (1ll << 31) || first < 0 || first + n_rays > program->n_rays)
    return TFRT_E_BADARG;
  if (n_rays == 0) return 0;
  if (!perm || !workspace) return TFRT_E_BADARG;
  const OrderLayout L = order_layout(n_rays);
  if (workspace_bytes < L.total) return TFRT_E_WORKSPACE;
  ProgramRays src;
  src.sp = *program;
  src.first = first;
  const int rc = ray_order_t(src, n_rays, face_verts, n_faces, axis, perm, keys_out,
                             static_cast<char*>(workspace), L, static_cast<hipStream_t>(stream),
                             cells);
  if (rc != 0) return rc;
  return hipGetLastError() == hipSuccess ? 0 : TFRT_E_LAUNCH;
}

int tfrt_source3d_order(const tfrt_source3d_program* program, int64_t first, int64_t n_rays,
                        const double* face_verts, int64_t n_faces, const double* axis,
                        int32_t* perm, uint32_t* keys_out, void* workspace,
                        size_t workspace_bytes, void* stream) {
  return source3d_order(program, first, n_rays, face_verts, n_faces, axis, perm, keys_out,
                        workspace, workspace_bytes, stream, false);
}

int tfrt_source3d_order_cells(const tfrt_source3d_program* program, int64_t first, int64_t n_rays,
                              const double* face_verts, int64_t n_faces, const double* axis,
                              int32_t* perm, uint32_t* keys_out, void* workspace,
                              size_t workspace_bytes, void* stream) {
  return source3d_order(program, first, n_rays, face_verts, n_faces, axis, perm, keys_out,
                        workspace, workspace_bytes, stream, true);
}

size_t tfrt_permute_rays_workspace_bytes(int64_t n_rays, int32_t state_dtype) {
  const size_t esz = state_dtype == TFRT_F64 ? 8 : (state_dtype == TFRT_F16 ? 2 : 4);
  return align_up((size_t)(n_rays > 0 ? n_rays : 1) * 8 * esz);
}

int tfrt_permute_rays(const void* src_rays, int64_t src_stride, int64_t n_rays,
                      int32_t state_dtype, const int32_t* index, void* dst_rays,
                      int64_t dst_stride, void* workspace, size_t workspace_bytes, void* stream) {
  if (n_rays < 0 || n_rays >= (1ll << 31)) return TFRT_E_BADARG;
  if (n_rays == 0) return 0;
  if (!src_rays || !dst_rays || !index || !workspace || src_stride < n_rays || dst_stride < n_rays)
    return TFRT_E_BADARG;
  if (workspace_bytes < tfrt_permute_rays_workspace_bytes(n_rays, state_dtype))
    return TFRT_E_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (state_dtype) {
    case TFRT_F32:
      permute_rays_t<float>(src_rays, src_stride, n_rays, index, dst_rays, dst_stride, workspace, st);
      break;
    case TFRT_F64:
      permute_rays_t<double>(src_rays, src_stride, n_rays, index, dst_rays, dst_stride, workspace, st);
      break;
    case TFRT_F16:
      permute_rays_t<_Float16>(src_rays, src_stride, n_rays, index, dst_rays, dst_stride, workspace, st);
      break;
    default:
      return TFRT_E_BADARG;
  }
  return hipGetLastError() == hipSuccess ? 0 : TFRT_E_LAUNCH;
}

int tfrt_gather_rows(const void* src, int64_t src_stride, int32_t n_rows, int32_t elem_bytes,
                     const int32_t* index, int64_t n, const int32_t* n_valid, void* dst,
                     int64_t dst_stride, void* stream) {
  if (n < 0 || n_rows < 0) return TFRT_E_BADARG;
  if (n == 0 || n_rows == 0) return 0;
  if (!src || !dst || !index) return TFRT_E_BADARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(cdiv(n, BLOCK));
  switch (elem_bytes) {
    case 1:
      hipLaunchKernelGGL((k_gather_rows<uint8_t>), grid, dim3(BLOCK), 0, st,
                         static_cast<const uint8_t*>(src), src_stride, n_rows, index, n, n_valid,
                         static_cast<uint8_t*>(dst), dst_stride);
      break;
    case 2:
      hipLaunchKernelGGL((k_gather_rows<uint16_t>), grid, dim3(BLOCK), 0, st,
                         static_cast<const uint16_t*>(src), src_stride, n_rows, index, n, n_valid,
                         static_cast<uint16_t*>(dst), dst_stride);
      break;
    case 4:
      hipLaunchKernelGGL((k_gather_rows<uint32_t>), grid, dim3(BLOCK), 0, st,
                         static_cast<const uint32_t*>(src), src_stride, n_rows, index, n, n_valid,
                         static_cast<uint32_t*>(dst), dst_stride);
      break;
    case 8:
      hipLaunchKernelGGL((k_gather_rows<uint64_t>), grid, dim3(BLOCK), 0, st,
                         static_cast<const uint64_t*>(src), src_stride, n_rows, index, n, n_valid,
                         static_cast<uint64_t*>(dst), dst_stride);
      break;
    default:
      return TFRT_E_BADARG;
  }
  return hipGetLastError() == hipSuccess ? 0 : TFRT_E_LAUNCH;
}

size_t tfrt_cluster_order_workspace_bytes(int64_t n_faces, int32_t leaf, int32_t group) {
  if (leaf < 1 || group < leaf) return 0;
  return cluster_layout(n_faces, leaf, group).total;
}

int tfrt_cluster_order(const double* face_verts, int64_t n_faces, int32_t leaf, int32_t group,
                       int32_t* order, void* workspace, size_t workspace_bytes, void* stream) {
  if (n_faces < 0 || n_faces >= (1ll << 31) || leaf < 1 || group < leaf || group % leaf != 0)
    return TFRT_E_BADARG;
  if (n_faces == 0) return 0;
  if (!face_verts || !order || !workspace) return TFRT_E_BADARG;
  const ClusterLayout L = cluster_layout(n_faces, leaf, group);
  if (L.levels + 1 > 22) return TFRT_E_UNSUPPORTED;     // (the key keeps >= 4 coordinate bits)
  if (workspace_bytes < L.total) return TFRT_E_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* ws = static_cast<char*>(workspace);
  const int M = (int)n_faces;
  float4* cent = reinterpret_cast<float4*>(ws + L.cent);
  unsigned* size_hist = reinterpret_cast<unsigned*>(ws + L.size_hist);
  float* thr = reinterpret_cast<float*>(ws + L.scalars);
  int32_t* n_big = reinterpret_cast<int32_t*>(ws + L.scalars + 16);
  int32_t* idx[2] = {reinterpret_cast<int32_t*>(ws + L.idx_a), reinterpret_cast<int32_t*>(ws + L.idx_b)};
  int32_t* seg_start = reinterpret_cast<int32_t*>(ws + L.seg_start);
  int32_t* seg_len = reinterpret_cast<int32_t*>(ws + L.seg_len);
  unsigned* path = reinterpret_cast<unsigned*>(ws + L.path);
  unsigned* bbox = reinterpret_cast<unsigned*>(ws + L.bbox);
  unsigned* keys = reinterpret_cast<unsigned*>(ws + L.keys);
  int32_t* perm = reinterpret_cast<int32_t*>(ws + L.perm);
  char* sort_ws = ws + L.sort;
  const dim3 grid(cdiv(M, BLOCK)), block(BLOCK);
  (void)hipMemsetAsync(size_hist, 0, CO_SIZE_BINS * sizeof(unsigned), st);
  hipLaunchKernelGGL(k_co_prepare, grid, block, 0, st, face_verts, M, cent, size_hist);
  hipLaunchKernelGGL(k_co_threshold, dim3(1), block, 0, st, size_hist, M, thr, n_big);
  hipLaunchKernelGGL(k_co_flag, grid, block, 0, st, cent, M, thr, keys, n_big);
  int rc = sort_keys(L.S, sort_ws, M, keys, perm, st);     // small faces first, outsized last
  if (rc != 0) return rc;
  int cur = 0;
  hipLaunchKernelGGL(k_co_apply, grid, block, 0, st, static_cast<const int32_t*>(nullptr), perm, M,
                     idx[cur]);
  hipLaunchKernelGGL(k_co_init, grid, block, 0, st, M, n_big, seg_start, seg_len, path);
  for (int lev = 0; lev < L.levels; ++lev) {
    int cbits = 26 - (lev + 1);
    if (cbits > 10) cbits = 10;
    (void)hipMemsetAsync(bbox, 0xFF, (size_t)M * 24, st);
    hipLaunchKernelGGL(k_co_bbox, grid, block, 0, st, cent, idx[cur], seg_start, seg_len, leaf, M,
                       bbox);
    hipLaunchKernelGGL(k_co_key, grid, block, 0, st, cent, idx[cur], seg_start, seg_len, path, bbox,
                       leaf, cbits, M, keys);
    rc = sort_keys(L.S, sort_ws, M, keys, perm, st);
    if (rc != 0) return rc;
    hipLaunchKernelGGL(k_co_apply, grid, block, 0, st, idx[cur], perm, M, idx[cur ^ 1]);
    cur ^= 1;
    hipLaunchKernelGGL(k_co_split, grid, block, 0, st, seg_start, seg_len, path, leaf, group, M);
  }
  (void)hipMemcpyAsync(order, idx[cur], (size_t)M * 4, hipMemcpyDeviceToDevice, st);
  return hipGetLastError() == hipSuccess ? 0 : TFRT_E_LAUNCH;
}

size_t tfrt_restore_order_workspace_bytes(int64_t n_src, int32_t n_segments) {
  return restore_layout(n_src, n_segments).total;
}

int tfrt_restore_order(const int32_t* ray_id, int64_t n_rows, const int32_t* seg_n,
                       const int32_t* seg_base, int32_t seg_stride, int32_t n_segments,
                       const int32_t* total_rows, const int32_t* perm, int64_t n_src,
                       int32_t* inv, int32_t* dest_of, int32_t* ray_id_out, void* workspace,
                       size_t workspace_bytes, void* stream) {
  if (n_rows < 0 || n_rows >= (1ll << 31) || n_src < 0 || n_src >= (1ll << 31))
    return TFRT_E_BADARG;
  if (n_rows == 0) return 0;
  if (!ray_id || !workspace) return TFRT_E_BADARG;
  const bool segs = seg_n != nullptr && seg_base != nullptr;
  if (segs && (n_segments < 1 || n_segments > SEG_LDS || seg_stride < 1)) return TFRT_E_UNSUPPORTED;
  const RestoreLayout L = restore_layout(n_src, segs ? n_segments : 1);
  if (workspace_bytes < L.total) return TFRT_E_WORKSPACE;
  if ((size_t)L.nchunks * 4 > 96 * 1024) return TFRT_E_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* ws = static_cast<char*>(workspace);
  uint2* words = reinterpret_cast<uint2*>(ws + L.words);
  unsigned* totals = reinterpret_cast<unsigned*>(ws + L.totals);
  Segments sg;
  sg.n = segs ? seg_n : nullptr;
  sg.base = segs ? seg_base : nullptr;
  sg.stride = seg_stride;
  sg.n_seg = segs ? n_segments : 1;
  sg.total = total_rows;
  sg.rows = (int32_t)n_rows;
  (void)hipMemsetAsync(words, 0, (size_t)L.len * sizeof(uint2), st);
  const dim3 grid(cdiv(n_rows, BLOCK));
  hipLaunchKernelGGL((k_restore_rows<0>), grid, dim3(BLOCK), 0, st, sg, ray_id, perm, (int)n_src,
                     L.wn, words, totals, L.nchunks, inv, dest_of, ray_id_out);
  hipLaunchKernelGGL(k_restore_scan, dim3(L.nchunks), dim3(BLOCK), 0, st, words, L.len, totals);
  hipLaunchKernelGGL((k_restore_rows<1>), grid, dim3(BLOCK), (size_t)L.nchunks * 4, st, sg, ray_id,
                     perm, (int)n_src, L.wn, words, totals, L.nchunks, inv, dest_of, ray_id_out);
  return hipGetLastError() == hipSuccess ? 0 : TFRT_E_LAUNCH;
}

}  // extern "C"
