// Source rays made on the device, in place.
//
// The reference's optimisation scripts re-draw their rays every step: dev/hexalens.py:36-48 builds
// its AperatureSource from two RandomUniformCircle distributions (tfrt/distributions.py:1586-1598:
// tf.random.uniform in _update), lifted and moved by BasePointTransformation (:2014-2120), and
// SGD_Optimizer.single_step calls optical_system.update() first (tfrt/optimizer.py:217).  With
// stock tensor ops that is ~20 small kernels and a new set of tensors per step -- which also ends
// every per-source cache and the launch graph of the step.  Here a source is a small PROGRAM:
//
//   points program   how sample i of a distribution is made: a table row (static distributions),
//                    or two uniform numbers of a counter-based generator (Philox4x32-10, counter =
//                    (sample, epoch), key = (seed, stream)) pushed through the distribution's
//                    formula (circle / square / spherical cap, uniform or Lambertian:
//                    distributions.py:1375-1393, 1586-1598, 1751-1775, 1814-1850) and the
//                    transformation (lift to 3-D, scale, quaternion, translation)
//   source program   AperatureSource / PointSource / AngularSource assembly of two of those
//                    (tfrt/sources.py:464-1095, undense: sample i of each input makes ray i)
//
// Rays and points are functions of (program, epoch, i): they are written into the caller's
// persistent buffers by one launch, any subset of them can be made again later (the sorted copy of
// an ordered source, a field somebody asks for after the step), and nothing but the device-side
// epoch counters changes from step to step, so the step stays one launch graph.
#include "tfrt_common.h"

namespace tfrt {

__device__ __forceinline__ void philox_round(uint32_t c[4], const uint32_t k[2]) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
  c[0] = n0;
  c[1] = (uint32_t)p1;
  c[2] = n2;
  c[3] = (uint32_t)p0;
}

// two uniform float64 in [0, 1) (53 bits each) for (seed, stream, epoch, sample)
__device__ __forceinline__ void uniform2(uint64_t seed, uint32_t stream, uint64_t epoch,
                                         uint64_t sample, double* u0, double* u1) {
  uint32_t c[4] = {(uint32_t)sample, (uint32_t)(sample >> 32), (uint32_t)epoch,
                   (uint32_t)(epoch >> 32)};
  uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32) ^ stream};
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k);
    k[0] += 0x9E3779B9u;
    k[1] += 0xBB67AE85u;
  }
  const uint64_t a = ((uint64_t)c[0] << 32) | c[1], b = ((uint64_t)c[2] << 32) | c[3];
  *u0 = (double)(a >> 11) * 0x1.0p-53;
  *u1 = (double)(b >> 11) * 0x1.0p-53;
}

__device__ __forceinline__ void quat_rotate(const double q[4], double v[3]) {
  // v' = v + w t + u x t, t = 2 u x v   (q = (w, u) a unit quaternion)
  const double t0 = 2.0 * (q[2] * v[2] - q[3] * v[1]);
  const double t1 = 2.0 * (q[3] * v[0] - q[1] * v[2]);
  const double t2 = 2.0 * (q[1] * v[1] - q[2] * v[0]);
  const double r0 = v[0] + q[0] * t0 + (q[2] * t2 - q[3] * t1);
  const double r1 = v[1] + q[0] * t1 + (q[3] * t0 - q[1] * t2);
  const double r2 = v[2] + q[0] * t2 + (q[1] * t1 - q[2] * t0);
  v[0] = r0;
  v[1] = r1;
  v[2] = r2;
}

constexpr double TWO_PI = 6.283185307179586476925286766559;
constexpr double GOLDEN_TURN = 3.14159265358979323846 * (1.0 + 2.2360679774997896964);  // pi (1 + sqrt 5)

// Sample `i` of a points program: the 3-D point (after the transformation) and the two numbers the
// distribution's rank properties are made of (circle: r in [0, 1], theta; sphere: phi, theta).
__device__ __forceinline__ void eval_points(const tfrt_points_program& pg, int64_t i,
                                            double out[3], double aux[2]) {
  double p[3] = {0.0, 0.0, 0.0};
  aux[0] = aux[1] = 0.0;
  if (pg.kind == TFRT_PTS_TABLE) {
    const double* row = pg.table + 3 * i;
    p[0] = row[0];
    p[1] = row[1];
    p[2] = row[2];
  } else {
    double u0, u1;
    uniform2(pg.seed, pg.stream, (uint64_t)*pg.epoch, (uint64_t)i, &u0, &u1);
    auto theta_mod = [&](double th) {
      if (pg.p[1] == 0.0 && pg.p[2] == TWO_PI) return th;
      const double span = pg.p[2] - pg.p[1];
      double m = fmod(th, span);
      if (m != 0.0 && ((m < 0.0) != (span < 0.0))) m += span;   // (sign of the divisor)
      return m + pg.p[1];
    };
    if (pg.kind == TFRT_PTS_CIRCLE) {            // p = {radius, theta_start, theta_end}
      const double r = sqrt(u0);
      const double th = theta_mod(TWO_PI * u1);
      double sn, cs;
      sincos(th, &sn, &cs);
      p[1] = pg.p[0] * (r * cs);
      p[2] = pg.p[0] * (r * sn);
      aux[0] = r;
      aux[1] = th;
    } else if (pg.kind == TFRT_PTS_SQUARE) {     // p = {x_size, -, -, y_size}
      p[1] = -pg.p[0] + (2.0 * pg.p[0]) * u0;
      p[2] = -pg.p[3] + (2.0 * pg.p[3]) * u1;
      aux[0] = p[1];
      aux[1] = p[2];
    } else {                                     // p = {radius, theta_start, theta_end, lower bound}
      const double c = pg.p[3] + (1.0 - pg.p[3]) * u0;
      const double phi = acos(pg.kind == TFRT_PTS_SPHERE_LAMBERT ? sqrt(c) : c);
      const double th = theta_mod(GOLDEN_TURN * u1);
      double sp, cp, sn, cs;
      sincos(phi, &sp, &cp);
      sincos(th, &sn, &cs);
      p[0] = pg.p[0] * cp;
      p[1] = pg.p[0] * (sp * cs);
      p[2] = pg.p[0] * (sp * sn);
      aux[0] = phi;
      aux[1] = th;
    }
    // BasePointTransformation (distributions.py:2014-2120): scale, rotate, translate
    if (pg.has_scale) {
      p[0] *= pg.scale[0];
      p[1] *= pg.scale[1];
      p[2] *= pg.scale[2];
    }
    if (pg.has_quat) quat_rotate(pg.quat, p);
    if (pg.has_shift) {
      p[0] += pg.shift[0];
      p[1] += pg.shift[1];
      p[2] += pg.shift[2];
    }
  }
  out[0] = p[0];
  out[1] = p[1];
  out[2] = p[2];
}

__global__ __launch_bounds__(BLOCK) void k_points(tfrt_points_program pg, const int32_t* index,
                                                  int64_t first, int64_t n,
                                                  double* __restrict__ points,
                                                  int32_t cols, double* __restrict__ aux0,
                                                  double* __restrict__ aux1) {
  const int64_t j = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (j >= n) return;
  const int64_t i = first + (index != nullptr ? index[j] : j);
  double p[3], aux[2];
  eval_points(pg, i, p, aux);
  if (points != nullptr) {
    if (cols == 3) {
      points[3 * j] = p[0];
      points[3 * j + 1] = p[1];
      points[3 * j + 2] = p[2];
    } else {                 // the untransformed distribution's own plane
      points[2 * j] = p[1];
      points[2 * j + 1] = p[2];
    }
  }
  if (aux0 != nullptr) aux0[j] = aux[0];
  if (aux1 != nullptr) aux1[j] = aux[1];
}

// ray i of the source (natural numbering)
__device__ __forceinline__ void eval_ray(const tfrt_source3d_program& sp, int64_t i, double s[3],
                                         double e[3]) {
  double a[3] = {0, 0, 0}, b[3] = {0, 0, 0}, aux[2];
  const int64_t ia = sp.a.count == 1 ? 0 : i, ib = sp.b.count == 1 ? 0 : i;
  if (sp.kind == TFRT_SRC_APERTURE) {
    eval_points(sp.a, ia, s, aux);
    eval_points(sp.b, ib, e, aux);
    return;
  }
  eval_points(sp.b, ib, b, aux);   // the direction vectors
  if (sp.has_quat) quat_rotate(sp.quat, b);
  if (sp.kind == TFRT_SRC_ANGULAR) {
    eval_points(sp.a, ia, a, aux);
    if (sp.has_quat) quat_rotate(sp.quat, a);
  }
  double st[3], en[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    st[q] = sp.center[q] + a[q];
    en[q] = st[q] + sp.ray_length * b[q];
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    s[q] = sp.swap ? en[q] : st[q];
    e[q] = sp.swap ? st[q] : en[q];
  }
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void k_source3d(tfrt_source3d_program sp,
                                                    const int32_t* __restrict__ index,
                                                    int64_t first, int64_t n,
                                                    T* __restrict__ rays, int64_t stride,
                                                    double* __restrict__ fields,
                                                    int64_t fstride) {
  const int64_t j = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (j >= n) return;
  const int64_t i = first + (index != nullptr ? index[j] : j);
  double s[3], e[3];
  eval_ray(sp, i, s, e);
  if (rays != nullptr) store_ray3(rays, stride, j, s, e);
  if (fields != nullptr) store_ray3(fields, fstride, j, s, e);
}

__global__ void k_epoch_advance(int64_t* p0, int64_t* p1, int64_t* p2, int64_t* p3, int64_t* p4,
                                int64_t* p5, int64_t* p6, int64_t* p7, int n) {
  int64_t* p[8] = {p0, p1, p2, p3, p4, p5, p6, p7};
  const int t = threadIdx.x;
  if (t < n && p[t] != nullptr) p[t][0] += 1;
}

static bool points_ok(const tfrt_points_program* pg) {
  if (!pg || pg->count < 0) return false;
  if (pg->kind == TFRT_PTS_TABLE) return pg->count == 0 || pg->table != nullptr;
  if (pg->kind < TFRT_PTS_TABLE || pg->kind > TFRT_PTS_SPHERE_LAMBERT) return false;
  return pg->epoch != nullptr;
}

}  // namespace tfrt

using namespace tfrt;

extern "C" {

int tfrt_epoch_advance(int64_t* const* epochs, int32_t n, void* stream) {
  if (n < 0 || n > 8 || (n > 0 && !epochs)) return TFRT_E_BADARG;
  if (n == 0) return 0;
  int64_t* p[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  for (int i = 0; i < n; ++i) p[i] = epochs[i];
  for (int i = 0; i < n; ++i)           // (the same counter twice would race)
    for (int j = 0; j < i; ++j)
      if (p[i] != nullptr && p[i] == p[j]) return TFRT_E_BADARG;
  hipLaunchKernelGGL(k_epoch_advance, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), p[0],
                     p[1], p[2], p[3], p[4], p[5], p[6], p[7], n);
  return hipGetLastError() == hipSuccess ? 0 : TFRT_E_LAUNCH;
}

int tfrt_points_generate(const tfrt_points_program* program, const int32_t* index,
                         int64_t first, int64_t n, double* points, int32_t point_columns, double* aux0, double* aux1,
                         void* stream) {
  if (!points_ok(program) || n < 0 || (point_columns != 2 && point_columns != 3))
    return TFRT_E_BADARG;
  if (first < 0 || (index == nullptr && first + n > program->count)) return TFRT_E_BADARG;
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_points, dim3(cdiv(n, BLOCK)), dim3(BLOCK), 0,
                     static_cast<hipStream_t>(stream), *program, index, first, n, points, point_columns,
                     aux0, aux1);
  return hipGetLastError() == hipSuccess ? 0 : TFRT_E_LAUNCH;
}

int tfrt_source3d_generate(const tfrt_source3d_program* program, const int32_t* index,
                           int64_t first, int64_t n, int32_t state_dtype, void* rays, int64_t stride, double* fields,
                           int64_t field_stride, void* stream) {
  if (!program || n < 0 || program->kind < TFRT_SRC_APERTURE || program->kind > TFRT_SRC_ANGULAR)
    return TFRT_E_BADARG;
  if (!points_ok(&program->b)) return TFRT_E_BADARG;
  if (program->kind != TFRT_SRC_POINT && !points_ok(&program->a)) return TFRT_E_BADARG;
  if (first < 0 || (index == nullptr && first + n > program->n_rays)) return TFRT_E_BADARG;
  const int64_t ca = program->kind == TFRT_SRC_POINT ? 1 : program->a.count, cb = program->b.count;
  if ((ca != 1 && ca != program->n_rays) || (cb != 1 && cb != program->n_rays))
    return TFRT_E_BADARG;   // (undense: every input has one sample or one per ray)
  if ((rays != nullptr && stride < n) || (fields != nullptr && field_stride < n))
    return TFRT_E_BADARG;
  if (n == 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(cdiv(n, BLOCK));
  switch (state_dtype) {
    case TFRT_F32:
      hipLaunchKernelGGL((k_source3d<float>), grid, dim3(BLOCK), 0, st, *program, index, first, n,
                         static_cast<float*>(rays), stride, fields, field_stride);
      break;
    case TFRT_F64:
      hipLaunchKernelGGL((k_source3d<double>), grid, dim3(BLOCK), 0, st, *program, index, first, n,
                         static_cast<double*>(rays), stride, fields, field_stride);
      break;
    case TFRT_F16:
      hipLaunchKernelGGL((k_source3d<_Float16>), grid, dim3(BLOCK), 0, st, *program, index, first, n,
                         static_cast<_Float16*>(rays), stride, fields, field_stride);
      break;
    default:
      return TFRT_E_BADARG;
  }
  return hipGetLastError() == hipSuccess ? 0 : TFRT_E_LAUNCH;
}

}  // extern "C"
