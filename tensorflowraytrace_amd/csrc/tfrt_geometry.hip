// The public pairwise functions of tfrt/geometry.py on the device:
//
//   line_intersect / raw_line_intersect                    geometry.py:27-167
//   line_triangle_intersect / raw_line_triangle_intersect  geometry.py:191-320
//   line_circle_intersect / raw_line_circle_intersect      geometry.py:338-547
//
// The reference evaluates them on tf.meshgrid copies of its inputs and returns dense
// (n_rows, n_cols) grids.  Here one lane computes one output element and reads its operands
// through (column stride, row stride) pairs, so the meshgrid copies never exist: the
// "meshgrid" forms pass strides (1, 0) for the first operand set and (0, 1) for the second, the
// element-wise "raw" forms pass (1, 0) for both with one row.  float64, operation order of the
// reference (FP contraction off), results bit-identical to an unfused CPU evaluation.
//
// The trace kernels do NOT go through these dense grids (that is the point of the fused path);
// these entry points exist so that code written against the reference's geometry API keeps
// working.  All of it is HBM-bound: 40..56 B written per element.
#include "tfrt_common.h"
#include "trace_math2d.h"

namespace tfrt {

struct Strided {
  const double* p;
  int64_t sc, sr;  // element (row, col) = p[col * sc + row * sr]
};

__device__ __forceinline__ double at(const Strided& a, int64_t row, int64_t col) {
  return a.p[col * a.sc + row * a.sr];
}

struct LineArgs {
  Strided x1s, y1s, x1e, y1e, x2s, y2s, x2e, y2e;
};

__global__ __launch_bounds__(BLOCK) void k_line_intersect(LineArgs a, int64_t n_cols,
                                                          int64_t n_rows, double eps,
                                                          double* __restrict__ x,
                                                          double* __restrict__ y,
                                                          uint8_t* __restrict__ valid,
                                                          double* __restrict__ u,
                                                          double* __restrict__ v) {
#pragma clang fp contract(off)
  const int64_t k = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (k >= n_cols * n_rows) return;
  const int64_t row = k / n_cols, col = k - row * n_cols;
  const double x1s = at(a.x1s, row, col), y1s = at(a.y1s, row, col);
  const double x2s = at(a.x2s, row, col), y2s = at(a.y2s, row, col);
  // geometry.py:136-160
  const double x1 = at(a.x1e, row, col) - x1s, y1 = at(a.y1e, row, col) - y1s;
  const double x2 = at(a.x2e, row, col) - x2s, y2 = at(a.y2e, row, col) - y2s;
  const double den = x1 * y2 - y1 * x2;
  const bool ok = fabs(den) >= eps;
  const double inv = 1.0 / (ok ? den : 1.0);
  const double uu = ok ? (x2 * (y1s - y2s) - y2 * (x1s - x2s)) * inv : 1.0;
  const double vv = ok ? (y1 * (x2s - x1s) - x1 * (y2s - y1s)) * inv : 1.0;
  x[k] = x1s + uu * x1;
  y[k] = y1s + uu * y1;
  valid[k] = ok;
  u[k] = uu;
  v[k] = vv;
}

struct TriArgs {
  Strided rx1, ry1, rz1, rx2, ry2, rz2, xp, yp, zp, x1, y1, z1, x2, y2, z2;
};

__global__ __launch_bounds__(BLOCK) void k_line_triangle_intersect(
    TriArgs t, int64_t n_cols, int64_t n_rows, double eps, double* __restrict__ x,
    double* __restrict__ y, double* __restrict__ z, uint8_t* __restrict__ valid,
    double* __restrict__ ray_u, double* __restrict__ trig_u, double* __restrict__ trig_v) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n_cols * n_rows) return;
  const int64_t row = i / n_cols, col = i - row * n_cols;
  const double rx1 = at(t.rx1, row, col), ry1 = at(t.ry1, row, col), rz1 = at(t.rz1, row, col);
  const double xp = at(t.xp, row, col), yp = at(t.yp, row, col), zp = at(t.zp, row, col);
  // geometry.py:286-318
  const double a = rx1 - at(t.rx2, row, col), d = ry1 - at(t.ry2, row, col),
               h = rz1 - at(t.rz2, row, col);
  const double b = at(t.x1, row, col) - xp, f = at(t.y1, row, col) - yp,
               k = at(t.z1, row, col) - zp;
  const double c = at(t.x2, row, col) - xp, g = at(t.y2, row, col) - yp,
               l = at(t.z2, row, col) - zp;
  const double q = rx1 - xp, r = ry1 - yp, s = rz1 - zp;
  const double den = a * g * k + b * d * l + c * f * h - a * f * l - b * g * h - c * d * k;
  const double nr = b * l * r + c * f * s + g * k * q - b * g * s - c * k * r - f * l * q;
  const double nu = a * g * s + c * h * r + d * l * q - a * l * r - c * d * s - g * h * q;
  const double nv = a * k * r + b * d * s + f * h * q - a * f * s - b * h * r - d * k * q;
  const bool ok = fabs(den) >= eps;
  const double sd = ok ? den : 1.0;
  const double ru = nr / sd;
  x[i] = rx1 - ru * a;
  y[i] = ry1 - ru * d;
  z[i] = rz1 - ru * h;
  valid[i] = ok;
  ray_u[i] = ru;
  trig_u[i] = nu / sd;
  trig_v[i] = nv / sd;
}

struct CircleArgs {
  Strided xs, ys, xe, ye, xc, yc, r;
};

// out: 5 arrays per root (x, y, u, v f64 and valid u8), plus root first then minus root
struct RootOut {
  double* x;
  double* y;
  uint8_t* valid;
  double* u;
  double* v;
};

__global__ __launch_bounds__(BLOCK) void k_line_circle_intersect(CircleArgs c, int64_t n_cols,
                                                                 int64_t n_rows, double eps,
                                                                 RootOut plus, RootOut minus) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n_cols * n_rows) return;
  const int64_t row = i / n_cols, col = i - row * n_cols;
  const double xs = at(c.xs, row, col), ys = at(c.ys, row, col);
  const double xe = at(c.xe, row, col), ye = at(c.ye, row, col);
  const double xc = at(c.xc, row, col), yc = at(c.yc, row, col);
  // geometry.py:464-530
  const double inv_r = 1.0 / at(c.r, row, col);
  const double xr = (xs - xc) * inv_r, yr = (ys - yc) * inv_r;
  const double xd = (xe - xs) * inv_r, yd = (ye - ys) * inv_r;
  const double a = xd * xd + yd * yd;
  const double b = 2.0 * xr * xd + 2.0 * yr * yd;
  const double cc = xr * xr + yr * yr - 1.0;
  double rad = b * b - 4.0 * a * cc;
  if (fabs(rad) < eps) rad = 0.0;  // tangent snap
  const bool rad_less = rad < 0.0;
  const double sr = sqrt(rad_less ? 1.0 : rad);
  double um = rad_less ? 1.0 : (-b - sr);
  double up = rad_less ? 1.0 : (-b + sr);
  const bool azero = fabs(a) < eps;
  const double inv = 1.0 / (azero ? 1.0 : 2 * a);
  um = azero ? 1.0 : um * inv;
  up = azero ? 1.0 : up * inv;
  const bool ok = !rad_less && !azero;
  const double xm = xs + (xe - xs) * um, ym = ys + (ye - ys) * um;
  const double xq = xs + (xe - xs) * up, yq = ys + (ye - ys) * up;
  plus.x[i] = xq;
  plus.y[i] = yq;
  plus.valid[i] = ok;
  plus.u[i] = up;
  plus.v[i] = atan2(yq - yc, xq - xc);
  minus.x[i] = xm;
  minus.y[i] = ym;
  minus.valid[i] = ok;
  minus.u[i] = um;
  minus.v[i] = atan2(ym - yc, xm - xc);
}

static bool grid_ok(int64_t n_cols, int64_t n_rows) {
  return n_cols >= 0 && n_rows >= 0 && (n_rows == 0 || n_cols < (1ll << 40) / (n_rows > 0 ? n_rows : 1));
}

}  // namespace tfrt

using namespace tfrt;

extern "C" {

int tfrt_line_intersect(int64_t n_cols, int64_t n_rows, const double* const first[4],
                        int64_t first_col_stride, int64_t first_row_stride,
                        const double* const second[4], int64_t second_col_stride,
                        int64_t second_row_stride, double epsilion, double* x, double* y,
                        uint8_t* valid, double* u, double* v, void* stream) {
  if (!grid_ok(n_cols, n_rows) || !first || !second) return TFRT_E_BADARG;
  const int64_t n = n_cols * n_rows;
  if (n == 0) return 0;
  if (!x || !y || !valid || !u || !v) return TFRT_E_BADARG;
  for (int k = 0; k < 4; ++k)
    if (!first[k] || !second[k]) return TFRT_E_BADARG;
  LineArgs a;
  Strided* f = &a.x1s;
  for (int k = 0; k < 4; ++k) {
    f[k] = {first[k], first_col_stride, first_row_stride};
    f[4 + k] = {second[k], second_col_stride, second_row_stride};
  }
  hipLaunchKernelGGL(k_line_intersect, dim3(cdiv(n, BLOCK)), dim3(BLOCK), 0,
                     static_cast<hipStream_t>(stream), a, n_cols, n_rows, epsilion, x, y, valid, u,
                     v);
  return hipGetLastError() == hipSuccess ? 0 : TFRT_E_LAUNCH;
}

int tfrt_line_triangle_intersect(int64_t n_cols, int64_t n_rows, const double* const rays[6],
                                 int64_t ray_col_stride, int64_t ray_row_stride,
                                 const double* const triangles[9], int64_t tri_col_stride,
                                 int64_t tri_row_stride, double epsilion, double* x, double* y,
                                 double* z, uint8_t* valid, double* ray_u, double* trig_u,
                                 double* trig_v, void* stream) {
  if (!grid_ok(n_cols, n_rows) || !rays || !triangles) return TFRT_E_BADARG;
  const int64_t n = n_cols * n_rows;
  if (n == 0) return 0;
  if (!x || !y || !z || !valid || !ray_u || !trig_u || !trig_v) return TFRT_E_BADARG;
  TriArgs t;
  Strided* f = &t.rx1;
  for (int k = 0; k < 6; ++k) {
    if (!rays[k]) return TFRT_E_BADARG;
    f[k] = {rays[k], ray_col_stride, ray_row_stride};
  }
  for (int k = 0; k < 9; ++k) {
    if (!triangles[k]) return TFRT_E_BADARG;
    f[6 + k] = {triangles[k], tri_col_stride, tri_row_stride};
  }
  hipLaunchKernelGGL(k_line_triangle_intersect, dim3(cdiv(n, BLOCK)), dim3(BLOCK), 0,
                     static_cast<hipStream_t>(stream), t, n_cols, n_rows, epsilion, x, y, z, valid,
                     ray_u, trig_u, trig_v);
  return hipGetLastError() == hipSuccess ? 0 : TFRT_E_LAUNCH;
}

int tfrt_line_circle_intersect(int64_t n_cols, int64_t n_rows, const double* const lines[4],
                               int64_t line_col_stride, int64_t line_row_stride,
                               const double* const circles[3], int64_t circle_col_stride,
                               int64_t circle_row_stride, double epsilion, double* const plus[4],
                               uint8_t* plus_valid, double* const minus[4], uint8_t* minus_valid,
                               void* stream) {
  if (!grid_ok(n_cols, n_rows) || !lines || !circles) return TFRT_E_BADARG;
  const int64_t n = n_cols * n_rows;
  if (n == 0) return 0;
  if (!plus || !minus || !plus_valid || !minus_valid) return TFRT_E_BADARG;
  CircleArgs c;
  Strided* f = &c.xs;
  for (int k = 0; k < 4; ++k) {
    if (!lines[k] || !plus[k] || !minus[k]) return TFRT_E_BADARG;
    f[k] = {lines[k], line_col_stride, line_row_stride};
  }
  for (int k = 0; k < 3; ++k) {
    if (!circles[k]) return TFRT_E_BADARG;
    f[4 + k] = {circles[k], circle_col_stride, circle_row_stride};
  }
  const RootOut p = {plus[0], plus[1], plus_valid, plus[2], plus[3]};
  const RootOut m = {minus[0], minus[1], minus_valid, minus[2], minus[3]};
  hipLaunchKernelGGL(k_line_circle_intersect, dim3(cdiv(n, BLOCK)), dim3(BLOCK), 0,
                     static_cast<hipStream_t>(stream), c, n_cols, n_rows, epsilion, p, m);
  return hipGetLastError() == hipSuccess ? 0 : TFRT_E_LAUNCH;
}

}  // extern "C"
