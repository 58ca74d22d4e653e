// Built-in image-forming error over the finished rays and its gradient seed, on the device.
//
// The error functions of the reference's optimisation scripts have one shape
// (dev/hexalens.py:144-168, dev/light_guide.py, the hexalens "inner / outer goal"):
//
//     output = stack(finished[field_0], finished[field_1], ...)           one row per finished ray
//     goal   = f(inherited source fields of that ray)
//     error  = squared_difference(output, goal)                           tf.math.squared_difference
//
// `goal` only depends on fields a finished ray inherits unchanged from its source ray
// (engine.py:2242-2281), so it is a table with one row per SOURCE ray, looked up through the
// source-ray index the trace carries along.  With that table on the device the whole optimiser
// step needs no data-dependent host code: no ray counts are read back, nothing is sliced, and the
// launch sequence is identical from step to step (hipGraph-capturable).
//
// k_goal_error: one lane per finished-ray slot (the number of finished rays is read from the
// device-side counters of the trace).  Writes d(sum error)/d(field) = 2 (output - goal) into the
// matching rows of the (6 x capacity) float64 seed block tfrt_trace3d_backward consumes, and the
// error sum in a FIXED summation order (per-workgroup partial sums, combined by k_goal_finish), so
// two runs give bit-identical errors.  HBM-bound: 4-8 B x n_fields read + 8 B x
// n_fields written per finished ray.
#include "tfrt_common.h"
#include "goal_finish.h"

namespace tfrt {

template <typename T>
__global__ __launch_bounds__(BLOCK) void k_goal_error(
    const T* __restrict__ fin, int64_t cap, const int32_t* __restrict__ fin_id,
    const int32_t* __restrict__ n_ptr, GoalFields gf, const double* __restrict__ goal,
    int64_t goal_stride, int64_t goal_ray_stride, double* __restrict__ g_fin, double* __restrict__ partial,
    double* __restrict__ zero_buf, int64_t zero_n) {
  // the reference's squared_difference and reduce_sum are separate ops: no contraction
#pragma clang fp contract(off)
  __shared__ double wsum[WAVES];
  const int n = *n_ptr;
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  // (optional) clear the buffer the reverse sweep accumulates into: saves the caller a fill launch
  for (int64_t k = i; k < zero_n; k += (int64_t)gridDim.x * BLOCK) zero_buf[k] = 0.0;
  double acc = 0.0;
  if (i < n) {
    const int64_t id = fin_id[i];
    for (int c = 0; c < gf.n; ++c) {
      const int64_t at = (int64_t)gf.row[c] * cap + i;
      const double r = ldd(fin, at) - goal[(int64_t)c * goal_stride + id * goal_ray_stride];
      g_fin[at] = 2.0 * r;
      acc += r * r;
    }
  }
  // fixed-shape reduction: xor butterflies inside the wave, waves in index order
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
  if (lane_id() == 0) wsum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int w = 0; w < WAVES; ++w) s += wsum[w];
    partial[blockIdx.x] = s;
  }
}

// Second stage (one workgroup): the partial sums in a fixed order.  A separate launch rather than
// "the workgroup that finishes last": that needs a device-scope release fence per workgroup, and
// with per-XCD L2s every such fence writes the XCD's dirty lines (the seed rows just stored) back
// -- 117 us at 1M rays against ~10 us for the two launches.
__global__ __launch_bounds__(BLOCK) void k_goal_finish(tfrt_goal_pending g) {
  goal_finish_block(g);
}

}  // namespace tfrt

using namespace tfrt;

extern "C" {

size_t tfrt_goal_error3d_workspace_bytes(int64_t capacity) {
  if (capacity < 0) return 0;
  return align_up((size_t)cdiv(capacity > 0 ? capacity : 1, BLOCK) * sizeof(double));
}

static int goal_error_launch(const void* finished_rays, int64_t capacity,
                             const int32_t* finished_id, int32_t state_dtype,
                             const int32_t* counts, int32_t max_passes, const int32_t* fields,
                             int32_t n_fields, const double* goal, int64_t goal_stride,
                             int64_t goal_ray_stride, double* grad_finished, double* error_out,
                             double* zero_buffer,
                             int64_t zero_count, int64_t* tests_total, void* workspace,
                             size_t workspace_bytes, tfrt_goal_pending* pending, void* stream) {
  if (capacity < 0 || n_fields < 1 || n_fields > 6 || !fields || !counts || max_passes < 0 ||
      !error_out || !workspace || workspace_bytes < tfrt_goal_error3d_workspace_bytes(capacity) ||
      zero_count < 0 || (zero_count > 0 && !zero_buffer) || goal_stride < 0 || goal_ray_stride < 0)
    return TFRT_E_BADARG;
  if (capacity > 0 && (!finished_rays || !finished_id || !goal || !grad_finished))
    return TFRT_E_BADARG;
  GoalFields gf;
  gf.n = n_fields;
  for (int c = 0; c < 6; ++c) {
    gf.row[c] = c < n_fields ? fields[c] : 0;
    if (gf.row[c] < 0 || gf.row[c] > 5) return TFRT_E_BADARG;
  }
  // trailing counters of the trace: {total_active, total_finished, ..., n_tests_lo, n_tests_hi}
  const int32_t* tail = counts + (size_t)max_passes * TFRT_COUNTS_PER_PASS;
  const int32_t* n_finished = tail + 1;
  double* partial = static_cast<double*>(workspace);
  const int nblk = cdiv(capacity > 0 ? capacity : 1, BLOCK);
  hipStream_t st = static_cast<hipStream_t>(stream);
#define TFRT_GOAL(T)                                                                           \
  hipLaunchKernelGGL((k_goal_error<T>), dim3(nblk), dim3(BLOCK), 0, st,                        \
                     static_cast<const T*>(finished_rays), capacity, finished_id, n_finished,  \
                     gf, goal, goal_stride, goal_ray_stride, grad_finished, partial, zero_buffer,   \
                     zero_count)
  if (state_dtype == TFRT_F32) {
    TFRT_GOAL(float);
  } else if (state_dtype == TFRT_F64) {
    TFRT_GOAL(double);
  } else if (state_dtype == TFRT_F16) {
    TFRT_GOAL(_Float16);
  } else {
    return TFRT_E_BADARG;
  }
#undef TFRT_GOAL
  tfrt_goal_pending g = {};
  g.partial = partial;
  g.n_partial = nblk;
  g.n_finished = n_finished;
  g.n_fields = n_fields;
  g.error_out = error_out;
  g.tests_lo_hi = tail + 4;
  g.tests_total = tests_total;
  if (pending != nullptr) {
    *pending = g;  // (the caller has the sum finished: tfrt_sgd_process_multi_finish / tfrt_goal_finish)
  } else {
    hipLaunchKernelGGL(k_goal_finish, dim3(1), dim3(BLOCK), 0, st, g);
  }
  return hipGetLastError() == hipSuccess ? 0 : TFRT_E_LAUNCH;
}

int tfrt_goal_error3d(const void* finished_rays, int64_t capacity, const int32_t* finished_id,
                      int32_t state_dtype, const int32_t* counts, int32_t max_passes,
                      const int32_t* fields, int32_t n_fields, const double* goal,
                      int64_t goal_stride, int64_t goal_ray_stride, double* grad_finished,
                      double* error_out, double* zero_buffer, int64_t zero_count,
                      int64_t* tests_total, void* workspace, size_t workspace_bytes,
                      void* stream) {
  return goal_error_launch(finished_rays, capacity, finished_id, state_dtype, counts, max_passes,
                           fields, n_fields, goal, goal_stride, goal_ray_stride, grad_finished,
                           error_out,
                           zero_buffer, zero_count, tests_total, workspace, workspace_bytes,
                           nullptr, stream);
}

int tfrt_goal_error3d_deferred(const void* finished_rays, int64_t capacity,
                               const int32_t* finished_id, int32_t state_dtype,
                               const int32_t* counts, int32_t max_passes, const int32_t* fields,
                               int32_t n_fields, const double* goal, int64_t goal_stride,
                               int64_t goal_ray_stride, double* grad_finished, double* error_out,
                               double* zero_buffer, int64_t zero_count, int64_t* tests_total,
                               void* workspace, size_t workspace_bytes,
                               tfrt_goal_pending* pending, void* stream) {
  if (!pending) return TFRT_E_BADARG;
  return goal_error_launch(finished_rays, capacity, finished_id, state_dtype, counts, max_passes,
                           fields, n_fields, goal, goal_stride, goal_ray_stride, grad_finished,
                           error_out,
                           zero_buffer, zero_count, tests_total, workspace, workspace_bytes,
                           pending, stream);
}

int tfrt_goal_finish(const tfrt_goal_pending* pending, void* stream) {
  if (!pending || !pending->partial || (!pending->n_finished && !pending->partial_counts) ||
      !pending->error_out ||
      pending->n_partial < 0)
    return TFRT_E_BADARG;
  hipLaunchKernelGGL(k_goal_finish, dim3(1), dim3(BLOCK), 0, static_cast<hipStream_t>(stream),
                     *pending);
  return hipGetLastError() == hipSuccess ? 0 : TFRT_E_LAUNCH;
}

}  // extern "C"
