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
#include "source_programs.h"

namespace tfrt {

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
  if (!points_program_ok(program) || n < 0 || (point_columns != 2 && point_columns != 3))
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
  if (n < 0 || !source_program_ok(program)) return TFRT_E_BADARG;
  if (first < 0 || (index == nullptr && first + n > program->n_rays)) return TFRT_E_BADARG;
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
