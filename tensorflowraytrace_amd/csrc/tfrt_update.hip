// Parameter-update kernels of the optimiser step (tfrt/optimizer.py:223-282, 316).
//
// k_sgd_process : non-finite -> 0, scale, clip of one gradient tensor and, optionally, the
//                 Keras-SGD apply `param -= sgd_lr * processed` in the same launch: the host
//                 path of the reference is six eager ops per parameter per step.
// k_csr_matvec  : y = A x for a CSR matrix: the accumulator / smoother products
//                 (optimizer.py:250-255, 277-282).  The matrices the mesh tools build
//                 (mesh_tools.py:221-421) have a handful of non-zeros per row, a dense (P,P)
//                 product reads P^2 doubles (213 MB for the 5167-vertex lens) to use ~P*k of them.
//
// All of it is HBM/latency bound (tens of KB per launch); one thread per element / one wave per
// row, coalesced loads, nothing else to tune.
#include "tfrt_common.h"
#include "goal_finish.h"

namespace tfrt {

// the reference runs these as separate multiply / subtract ops: keep them unfused
#pragma clang fp contract(off)

template <typename T>
__global__ __launch_bounds__(BLOCK) void k_sgd_process(const T* __restrict__ grad,
                                                       T* __restrict__ processed,
                                                       T* __restrict__ param, int64_t n, T scale,
                                                       T clip, T sgd_lr,
                                                       const double* __restrict__ hyper) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  if (hyper != nullptr) {  // step-dependent values live on the device (replayed launch graphs)
    scale = static_cast<T>(hyper[0]);
    clip = static_cast<T>(hyper[1]);
    sgd_lr = static_cast<T>(hyper[2]);
  }
  T g = grad[i];
  // optimizer.py:226-229 (tf.where(is_finite(g), g, 0)), :233 scale, :236-247 clip_by_value
  g = isfinite(g) ? g : T(0);
  g = g * scale;
  g = g < -clip ? -clip : (g > clip ? clip : g);
  if (processed != nullptr) processed[i] = g;
  if (param != nullptr) param[i] = param[i] - sgd_lr * g;
}

// The same for up to SGD_BATCH tensors in one launch (an optimiser with several parameter
// tensors -- the two surfaces of a lens -- otherwise pays one ~4 us launch each per step).
constexpr int SGD_BATCH = 8;
struct SgdBatch {
  const double* grad[SGD_BATCH];
  double* processed[SGD_BATCH];
  double* param[SGD_BATCH];
  int64_t n[SGD_BATCH];
  int32_t first_block[SGD_BATCH + 1];
  int32_t count;
};

__global__ __launch_bounds__(BLOCK) void k_sgd_process_multi(SgdBatch b,
                                                             const double* __restrict__ hyper,
                                                             tfrt_goal_pending goal) {
  // (one workgroup more than the tensors need, when the step's error sum is still to be finished)
  if ((int)blockIdx.x == b.first_block[SGD_BATCH]) {
    goal_finish_block(goal);
    return;
  }
  int k = 0;
  while (k + 1 < b.count && (int)blockIdx.x >= b.first_block[k + 1]) ++k;  // block-uniform
  const int64_t i = (int64_t)((int)blockIdx.x - b.first_block[k]) * BLOCK + threadIdx.x;
  if (i >= b.n[k]) return;
  const double scale = hyper[3 * k], clip = hyper[3 * k + 1], sgd_lr = hyper[3 * k + 2];
  double g = b.grad[k][i];
  g = isfinite(g) ? g : 0.0;
  g = g * scale;
  g = g < -clip ? -clip : (g > clip ? clip : g);
  if (b.processed[k] != nullptr) b.processed[k][i] = g;
  if (b.param[k] != nullptr) b.param[k][i] = b.param[k][i] - sgd_lr * g;
}

// one wave per row; lanes stride over the row's non-zeros, butterfly-sum at the end
__global__ __launch_bounds__(BLOCK) void k_csr_matvec(const int64_t* __restrict__ crow,
                                                      const int64_t* __restrict__ col,
                                                      const double* __restrict__ val,
                                                      const double* __restrict__ x,
                                                      double* __restrict__ y, int64_t n_rows) {
  const int64_t row = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
  if (row >= n_rows) return;  // whole wave leaves together: row is wave-uniform
  const int64_t lo = crow[row], hi = crow[row + 1];
  double acc = 0.0;
  for (int64_t k = lo + lane_id(); k < hi; k += 64) acc += val[k] * x[col[k]];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (lane_id() == 0) y[row] = acc;
}

}  // namespace tfrt

using namespace tfrt;

extern "C" {

static int sgd_process_launch(const void* grad, void* processed, void* param, int64_t n,
                              int32_t dtype, double scale, double clip, double sgd_learning_rate,
                              const double* hyper, void* stream) {
  if (n == 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(cdiv(n, BLOCK));
  if (dtype == TFRT_F64)
    hipLaunchKernelGGL((k_sgd_process<double>), grid, dim3(BLOCK), 0, st,
                       static_cast<const double*>(grad), static_cast<double*>(processed),
                       static_cast<double*>(param), n, scale, clip, sgd_learning_rate, hyper);
  else
    hipLaunchKernelGGL((k_sgd_process<float>), grid, dim3(BLOCK), 0, st,
                       static_cast<const float*>(grad), static_cast<float*>(processed),
                       static_cast<float*>(param), n, (float)scale, (float)clip,
                       (float)sgd_learning_rate, hyper);
  return hipGetLastError() == hipSuccess ? 0 : TFRT_E_LAUNCH;
}

int tfrt_sgd_process(const void* grad, void* processed, void* param, int64_t n, int32_t dtype,
                     double scale, double clip, double sgd_learning_rate, void* stream) {
  if (n < 0 || (n > 0 && !grad) || clip < 0.0 || (dtype != TFRT_F32 && dtype != TFRT_F64))
    return TFRT_E_BADARG;
  return sgd_process_launch(grad, processed, param, n, dtype, scale, clip, sgd_learning_rate,
                            nullptr, stream);
}

int tfrt_sgd_process_dev(const void* grad, void* processed, void* param, int64_t n, int32_t dtype,
                         const double* hyper, void* stream) {
  if (n < 0 || (n > 0 && !grad) || !hyper || (dtype != TFRT_F32 && dtype != TFRT_F64))
    return TFRT_E_BADARG;
  return sgd_process_launch(grad, processed, param, n, dtype, 0.0, 0.0, 0.0, hyper, stream);
}

static int sgd_multi_launch(int32_t n_tensors, const void* const* grad, void* const* processed,
                            void* const* param, const int64_t* n, const double* hyper,
                            const tfrt_goal_pending* pending, void* stream) {
  if (n_tensors < 0 || n_tensors > SGD_BATCH || (n_tensors > 0 && (!grad || !n || !hyper)))
    return TFRT_E_BADARG;
  if (pending != nullptr && (!pending->partial || (!pending->n_finished && !pending->partial_counts) ||
      !pending->error_out ||
                             pending->n_partial < 0))
    return TFRT_E_BADARG;
  SgdBatch b;
  int blocks = 0;
  for (int k = 0; k < SGD_BATCH; ++k) {
    const bool on = k < n_tensors;
    if (on && (n[k] < 0 || (n[k] > 0 && !grad[k]))) return TFRT_E_BADARG;
    b.grad[k] = on ? static_cast<const double*>(grad[k]) : nullptr;
    b.processed[k] = (on && processed) ? static_cast<double*>(processed[k]) : nullptr;
    b.param[k] = (on && param) ? static_cast<double*>(param[k]) : nullptr;
    b.n[k] = on ? n[k] : 0;
    b.first_block[k] = blocks;
    if (on) blocks += cdiv(n[k], BLOCK);
  }
  b.first_block[SGD_BATCH] = blocks;
  b.count = n_tensors;
  const int grid = blocks + (pending != nullptr ? 1 : 0);
  if (grid == 0) return 0;
  hipLaunchKernelGGL(k_sgd_process_multi, dim3(grid), dim3(BLOCK), 0,
                     static_cast<hipStream_t>(stream), b, hyper,
                     pending != nullptr ? *pending : tfrt_goal_pending{});
  return hipGetLastError() == hipSuccess ? 0 : TFRT_E_LAUNCH;
}

int tfrt_sgd_process_multi(int32_t n_tensors, const void* const* grad, void* const* processed,
                           void* const* param, const int64_t* n, const double* hyper,
                           void* stream) {
  return sgd_multi_launch(n_tensors, grad, processed, param, n, hyper, nullptr, stream);
}

int tfrt_sgd_process_multi_finish(int32_t n_tensors, const void* const* grad,
                                  void* const* processed, void* const* param, const int64_t* n,
                                  const double* hyper, const tfrt_goal_pending* pending,
                                  void* stream) {
  if (!pending) return TFRT_E_BADARG;
  return sgd_multi_launch(n_tensors, grad, processed, param, n, hyper, pending, stream);
}

int tfrt_csr_matvec(const int64_t* crow_indices, const int64_t* col_indices, const double* values,
                    const double* x, double* y, int64_t n_rows, void* stream) {
  if (n_rows < 0 || (n_rows > 0 && (!crow_indices || !x || !y))) return TFRT_E_BADARG;
  if (n_rows == 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(k_csr_matvec, dim3(cdiv(n_rows, WAVES)), dim3(BLOCK), 0, st, crow_indices,
                     col_indices, values, x, y, n_rows);
  return hipGetLastError() == hipSuccess ? 0 : TFRT_E_LAUNCH;
}

}  // extern "C"
