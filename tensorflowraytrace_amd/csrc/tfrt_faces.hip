// Triangle-boundary face construction (forward + reverse), the element-wise Snell seams and
// the library bookkeeping entry points.
//
//   k_build_faces      tfrt/boundaries.py:890-923 update_fields_from_vertices
//   k_build_faces_bwd  reverse of the same gather/cross/normalize incl. the per-corner
//                      stop_gradient mask (vertex_update_map, boundaries.py:900-913)
//   k_snell3d/2d       tfrt/geometry.py:671-753 / 565-653, one ray per lane
#include "tfrt_common.h"

namespace tfrt {

__global__ __launch_bounds__(BLOCK) void k_build_faces(const double* __restrict__ vertices,
                                                       int64_t V, const int32_t* __restrict__ faces,
                                                       int64_t F, double* __restrict__ fverts,
                                                       double* __restrict__ norm) {
  const int64_t f = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (f >= F) return;
  double P[9];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    int64_t v = faces[3 * f + c];
    if (v < 0 || v >= V) v = 0;  // host validates; keep the access in range regardless
#pragma unroll
    for (int k = 0; k < 3; ++k) P[3 * c + k] = vertices[3 * v + k];
  }
#pragma unroll
  for (int q = 0; q < 9; ++q) fverts[9 * f + q] = P[q];
  if (norm != nullptr) {
    double N[3], C[3], clen;
    face_normal(P, N, C, &clen);
    norm[3 * f] = N[0];
    norm[3 * f + 1] = N[1];
    norm[3 * f + 2] = N[2];
  }
}

__global__ __launch_bounds__(BLOCK) void k_build_faces_bwd(
    const double* __restrict__ g_fverts, const double* __restrict__ g_norm,
    const double* __restrict__ fverts, const int32_t* __restrict__ faces,
    const uint8_t* __restrict__ mask, int64_t F, int64_t V, double* __restrict__ g_vertices) {
  const int64_t f = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (f >= F) return;
  double g[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) g[q] = g_fverts ? g_fverts[9 * f + q] : 0.0;
  if (g_norm != nullptr) {
    // N = C/|C|, C = A x B, A = P1-P0, B = P2-P1
    double P[9], N[3], C[3], clen;
#pragma unroll
    for (int q = 0; q < 9; ++q) P[q] = fverts[9 * f + q];
    face_normal(P, N, C, &clen);
    const double gn[3] = {g_norm[3 * f], g_norm[3 * f + 1], g_norm[3 * f + 2]};
    const double nn = dot3(N, gn);
    double Cb[3];
    for (int k = 0; k < 3; ++k) Cb[k] = (gn[k] - N[k] * nn) / clen;
    const double A[3] = {P[3] - P[0], P[4] - P[1], P[5] - P[2]};
    const double B[3] = {P[6] - P[3], P[7] - P[4], P[8] - P[5]};
    double Ab[3], Bb[3];
    cross3(B, Cb, Ab);
    cross3(Cb, A, Bb);
    for (int k = 0; k < 3; ++k) {
      g[k] -= Ab[k];
      g[3 + k] += Ab[k] - Bb[k];
      g[6 + k] += Bb[k];
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if (mask != nullptr && mask[3 * f + c] == 0) continue;
    const int64_t v = faces[3 * f + c];
    if (v < 0 || v >= V) continue;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double x = g[3 * c + k];
      if (x != 0.0) unsafeAtomicAdd(g_vertices + 3 * v + k, x);
    }
  }
}

// Parametric surfaces (boundaries.py:1065-1078): vertices = zero_points + parameters * vectors
// followed by the face gather above, in one launch; the reverse folds the scatter through the
// gather, the per-corner stop_gradient mask and the product with `vectors` into one atomic per
// corner.  The reference runs the product and the sum as separate ops: keep them unfused so the
// faces are bit-identical with the two-step path.
#pragma clang fp contract(off)
__global__ __launch_bounds__(BLOCK) void k_param_faces(
    const double* __restrict__ zero, const double* __restrict__ vectors,
    const double* __restrict__ params, int64_t V, const int32_t* __restrict__ faces, int64_t F,
    double* __restrict__ fverts, double* __restrict__ norm) {
  const int64_t f = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (f >= F) return;
  double P[9];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    int64_t v = faces[3 * f + c];
    if (v < 0 || v >= V) v = 0;  // host validates; keep the access in range regardless
    const double p = params[v];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double step = p * vectors[3 * v + k];
      P[3 * c + k] = zero[3 * v + k] + step;
    }
  }
#pragma unroll
  for (int q = 0; q < 9; ++q) fverts[9 * f + q] = P[q];
  if (norm != nullptr) {
    double N[3], C[3], clen;
    face_normal(P, N, C, &clen);
    norm[3 * f] = N[0];
    norm[3 * f + 1] = N[1];
    norm[3 * f + 2] = N[2];
  }
}

// shared by the two reverse kernels: gradient of the 9 face coordinates including the part
// that arrives through the unit normal
__device__ inline void face_grad(const double* __restrict__ g_fverts,
                                 const double* __restrict__ g_norm,
                                 const double* __restrict__ fverts, int64_t f, double g[9]) {
#pragma unroll
  for (int q = 0; q < 9; ++q) g[q] = g_fverts ? g_fverts[9 * f + q] : 0.0;
  if (g_norm != nullptr) {
    // N = C/|C|, C = A x B, A = P1-P0, B = P2-P1
    double P[9], N[3], C[3], clen;
#pragma unroll
    for (int q = 0; q < 9; ++q) P[q] = fverts[9 * f + q];
    face_normal(P, N, C, &clen);
    const double gn[3] = {g_norm[3 * f], g_norm[3 * f + 1], g_norm[3 * f + 2]};
    const double nn = dot3(N, gn);
    double Cb[3];
    for (int k = 0; k < 3; ++k) Cb[k] = (gn[k] - N[k] * nn) / clen;
    const double A[3] = {P[3] - P[0], P[4] - P[1], P[5] - P[2]};
    const double B[3] = {P[6] - P[3], P[7] - P[4], P[8] - P[5]};
    double Ab[3], Bb[3];
    cross3(B, Cb, Ab);
    cross3(Cb, A, Bb);
    for (int k = 0; k < 3; ++k) {
      g[k] -= Ab[k];
      g[3 + k] += Ab[k] - Bb[k];
      g[6 + k] += Bb[k];
    }
  }
}

__global__ __launch_bounds__(BLOCK) void k_param_faces_bwd(
    const double* __restrict__ g_fverts, const double* __restrict__ g_norm,
    const double* __restrict__ fverts, const int32_t* __restrict__ faces,
    const uint8_t* __restrict__ mask, const double* __restrict__ vectors, int64_t F, int64_t V,
    double* __restrict__ g_params) {
  const int64_t f = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (f >= F) return;
  double g[9];
  face_grad(g_fverts, g_norm, fverts, f, g);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if (mask != nullptr && mask[3 * f + c] == 0) continue;
    const int64_t v = faces[3 * f + c];
    if (v < 0 || v >= V) continue;
    const double x = g[3 * c] * vectors[3 * v] + g[3 * c + 1] * vectors[3 * v + 1] +
                     g[3 * c + 2] * vectors[3 * v + 2];
    if (x != 0.0) unsafeAtomicAdd(g_params + v, x);
  }
}

// Gather forms of the two reverse kernels: a lane group per VERTEX walks the face corners that
// reference it (corner_start / corner_list: the corners f*3+c sorted by vertex, built once per
// mesh topology) and sums their gradients in that fixed order.  No atomics, no zero-filled
// output, and the same bits on every run (a float64 atomic sum depends on arrival order).
// (8 lanes per vertex: lane j takes the vertex's corners j, j + 8, ... and the eight partial sums
// are combined by a fixed xor butterfly -- a vertex has ~6 corners, one lane per vertex left the
// chip at 21 workgroups for a 5,167-vertex surface.)
constexpr int GATHER_LANES = 8;

__global__ __launch_bounds__(BLOCK) void k_build_faces_bwd_gather(
    const double* __restrict__ g_fverts, const double* __restrict__ g_norm,
    const double* __restrict__ fverts, const uint8_t* __restrict__ mask,
    const int32_t* __restrict__ corner_start, const int32_t* __restrict__ corner_list, int64_t V,
    double* __restrict__ g_vertices) {
  const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  const int64_t v = t / GATHER_LANES;
  const int sub = (int)(t % GATHER_LANES);
  double acc[3] = {0.0, 0.0, 0.0};
  if (v < V) {
    for (int q = corner_start[v] + sub; q < corner_start[v + 1]; q += GATHER_LANES) {
      const int fc = corner_list[q], f = fc / 3, c = fc - 3 * f;
      if (mask != nullptr && mask[fc] == 0) continue;
      double g[9];
      face_grad(g_fverts, g_norm, fverts, f, g);
      acc[0] += g[3 * c];
      acc[1] += g[3 * c + 1];
      acc[2] += g[3 * c + 2];
    }
  }
#pragma unroll
  for (int d = GATHER_LANES / 2; d > 0; d >>= 1)
    for (int k = 0; k < 3; ++k) acc[k] += __shfl_xor(acc[k], d, 64);
  if (v < V && sub == 0) {
    g_vertices[3 * v] = acc[0];
    g_vertices[3 * v + 1] = acc[1];
    g_vertices[3 * v + 2] = acc[2];
  }
}

__global__ __launch_bounds__(BLOCK) void k_param_faces_bwd_gather(
    const double* __restrict__ g_fverts, const double* __restrict__ g_norm,
    const double* __restrict__ fverts, const uint8_t* __restrict__ mask,
    const double* __restrict__ vectors, const int32_t* __restrict__ corner_start,
    const int32_t* __restrict__ corner_list, int64_t V, double* __restrict__ g_params) {
  const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  const int64_t v = t / GATHER_LANES;
  const int sub = (int)(t % GATHER_LANES);
  double acc = 0.0;
  if (v < V) {
    for (int q = corner_start[v] + sub; q < corner_start[v + 1]; q += GATHER_LANES) {
      const int fc = corner_list[q], f = fc / 3, c = fc - 3 * f;
      if (mask != nullptr && mask[fc] == 0) continue;
      double g[9];
      face_grad(g_fverts, g_norm, fverts, f, g);
      acc += g[3 * c] * vectors[3 * v] + g[3 * c + 1] * vectors[3 * v + 1] +
             g[3 * c + 2] * vectors[3 * v + 2];
    }
  }
#pragma unroll
  for (int d = GATHER_LANES / 2; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
  if (v < V && sub == 0) g_params[v] = acc;
}

// Several surfaces of one optical system in ONE launch each way (a system's update is a chain
// of launches of a few thousand faces each: a dependent launch costs ~4.5 us whatever it does).
// The descriptors travel by value in the kernel arguments; workgroups are dealt to the surfaces
// in order (first_block).  A surface with `copy_from` is a fixed one: its rows are copied into
// its place of the merged (M, 9) block, so the system's faces need no concatenation either.
struct FaceSurfaces {
  int32_t count;
  int32_t first_block[TFRT_MAX_SURFACES + 1];
  tfrt_face_surface s[TFRT_MAX_SURFACES];
};
struct FaceSurfaceGrads {
  int32_t count;
  int32_t first_block[TFRT_MAX_SURFACES + 1];
  tfrt_face_surface_grad s[TFRT_MAX_SURFACES];
};

#pragma clang fp contract(off)
__global__ __launch_bounds__(BLOCK) void k_param_faces_multi(FaceSurfaces b) {
  int k = 0;
  while (k + 1 < b.count && (int)blockIdx.x >= b.first_block[k + 1]) ++k;  // block-uniform
  const tfrt_face_surface& u = b.s[k];
  const int64_t f = (int64_t)((int)blockIdx.x - b.first_block[k]) * BLOCK + threadIdx.x;
  if (f >= u.n_faces) return;
  if (u.copy_from != nullptr) {
#pragma unroll
    for (int q = 0; q < 9; ++q) u.face_verts[9 * f + q] = u.copy_from[9 * f + q];
    return;
  }
  // (the same arithmetic as k_param_faces: product and sum stay separate roundings)
  double P[9];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    int64_t v = u.faces[3 * f + c];
    if (v < 0 || v >= u.n_vertices) v = 0;  // host validates; keep the access in range regardless
    const double p = u.parameters[v];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const double step = p * u.vectors[3 * v + q];
      P[3 * c + q] = u.zero_points[3 * v + q] + step;
    }
  }
#pragma unroll
  for (int q = 0; q < 9; ++q) u.face_verts[9 * f + q] = P[q];
  if (u.norm != nullptr) {
    double N[3], C[3], clen;
    face_normal(P, N, C, &clen);
    u.norm[3 * f] = N[0];
    u.norm[3 * f + 1] = N[1];
    u.norm[3 * f + 2] = N[2];
  }
}

__global__ __launch_bounds__(BLOCK) void k_param_faces_bwd_gather_multi(FaceSurfaceGrads b) {
  int k = 0;
  while (k + 1 < b.count && (int)blockIdx.x >= b.first_block[k + 1]) ++k;  // block-uniform
  const tfrt_face_surface_grad& u = b.s[k];
  const int64_t t = (int64_t)((int)blockIdx.x - b.first_block[k]) * BLOCK + threadIdx.x;
  const int64_t v = t / GATHER_LANES;
  const int sub = (int)(t % GATHER_LANES);
  double acc = 0.0;
  if (v < u.n_vertices) {
    for (int q = u.corner_start[v] + sub; q < u.corner_start[v + 1]; q += GATHER_LANES) {
      const int fc = u.corner_list[q], f = fc / 3, c = fc - 3 * f;
      if (u.update_mask != nullptr && u.update_mask[fc] == 0) continue;
      double g[9];
      face_grad(u.grad_face_verts, u.grad_norm, u.face_verts, f, g);
      acc += g[3 * c] * u.vectors[3 * v] + g[3 * c + 1] * u.vectors[3 * v + 1] +
             g[3 * c + 2] * u.vectors[3 * v + 2];
    }
  }
#pragma unroll
  for (int d = GATHER_LANES / 2; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
  if (v < u.n_vertices && sub == 0) u.grad_parameters[v] = acc;
}

__global__ __launch_bounds__(BLOCK) void k_snell3d(int64_t n, const double* xs, const double* ys,
                                                   const double* zs, const double* xe,
                                                   const double* ye, const double* ze,
                                                   const double* norm, const double* n_in,
                                                   const double* n_out, double L, double* out) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const double s[3] = {xs[i], ys[i], zs[i]}, h[3] = {xe[i], ye[i], ze[i]};
  const double N[3] = {norm[3 * i], norm[3 * i + 1], norm[3 * i + 2]};
  const Snell3 f = snell3d(s, h, N, n_in[i], n_out[i]);
  for (int k = 0; k < 3; ++k) {
    out[k * n + i] = h[k];
    out[(3 + k) * n + i] = advance(h[k], L, f.w[k]);
  }
}

__global__ __launch_bounds__(BLOCK) void k_snell2d(int64_t n, const double* xs, const double* ys,
                                                   const double* xe, const double* ye,
                                                   const double* norm, const double* n_in,
                                                   const double* n_out, double L, double* out) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const double a = snell2d_angle(xs[i], ys[i], xe[i], ye[i], norm[i], n_in[i], n_out[i]);
  out[i] = xe[i];
  out[n + i] = ye[i];
  out[2 * n + i] = advance(xe[i], L, cos(a));
  out[3 * n + i] = advance(ye[i], L, sin(a));
}

// Arithmetic self-test: the float64 primitives the decisions rest on, one result per element,
// compiled with this library's flags.  Parity with the reference's eager float64 ops needs each
// of them correctly rounded (IEEE 754), like the host's.
__global__ __launch_bounds__(BLOCK) void k_selftest_f64(int op, int64_t n, const double* a,
                                                        const double* b, double* out) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const double x = a[i], y = b ? b[i] : 0.0;
  double r;
  switch (op) {
    case TFRT_SELFTEST_DIV: r = x / y; break;
    case TFRT_SELFTEST_SQRT: r = sqrt(x); break;
    case TFRT_SELFTEST_RSQRT: r = 1.0 / sqrt(x); break;          // l2_normalize3's scale
    case TFRT_SELFTEST_MULADD: r = advance(x, y, y); break;       // x + y*y, two roundings
    case TFRT_SELFTEST_ADJ_RCP: r = adj_rcp(x); break;            // the reverse sweep's own
    case TFRT_SELFTEST_ADJ_RSQRT: r = adj_rsqrt(x); break;
    default: r = 0.0;
  }
  out[i] = r;
}

}  // namespace tfrt

using namespace tfrt;

extern "C" {

int tfrt_version(void) { return TFRT_VERSION; }

const char* tfrt_strerror(int code) {
  switch (code) {
    case 0: return "ok";
    case TFRT_E_BADARG: return "bad argument (null pointer, negative size, stride < n, or too many faces)";
    case TFRT_E_WORKSPACE: return "workspace too small (query tfrt_*_workspace_bytes)";
    case TFRT_E_LAUNCH: return "HIP kernel launch failed";
    case TFRT_E_UNSUPPORTED: return "unsupported state dtype or option";
    default: return "unknown tfrt error";
  }
}

int tfrt_build_faces_forward(const double* vertices, int64_t n_vertices, const int32_t* faces,
                             int64_t n_faces, double* face_verts, double* norm, void* stream) {
  if (n_faces < 0 || n_vertices < 0) return TFRT_E_BADARG;
  if (n_faces == 0) return 0;
  if (!vertices || !faces || !face_verts || n_vertices == 0) return TFRT_E_BADARG;
  hipLaunchKernelGGL(k_build_faces, dim3(cdiv(n_faces, BLOCK)), dim3(BLOCK), 0,
                     static_cast<hipStream_t>(stream), vertices, n_vertices, faces, n_faces,
                     face_verts, norm);
  return hipGetLastError() == hipSuccess ? 0 : TFRT_E_LAUNCH;
}

int tfrt_build_faces_backward(const double* grad_face_verts, const double* grad_norm,
                              const double* face_verts, const int32_t* faces,
                              const uint8_t* update_mask, int64_t n_faces, int64_t n_vertices,
                              const int32_t* corner_start, const int32_t* corner_list,
                              double* grad_vertices, void* stream) {
  if (n_faces < 0 || n_vertices < 0) return TFRT_E_BADARG;
  if (n_faces == 0) return 0;
  if (!faces || !grad_vertices || (!grad_face_verts && !grad_norm) ||
      (grad_norm && !face_verts) || ((corner_start == nullptr) != (corner_list == nullptr)))
    return TFRT_E_BADARG;
  if (corner_start != nullptr) {
    hipLaunchKernelGGL(k_build_faces_bwd_gather, dim3(cdiv(n_vertices * GATHER_LANES, BLOCK)),
                       dim3(BLOCK), 0,
                       static_cast<hipStream_t>(stream), grad_face_verts, grad_norm, face_verts,
                       update_mask, corner_start, corner_list, n_vertices, grad_vertices);
    return hipGetLastError() == hipSuccess ? 0 : TFRT_E_LAUNCH;
  }
  hipLaunchKernelGGL(k_build_faces_bwd, dim3(cdiv(n_faces, BLOCK)), dim3(BLOCK), 0,
                     static_cast<hipStream_t>(stream), grad_face_verts, grad_norm, face_verts,
                     faces, update_mask, n_faces, n_vertices, grad_vertices);
  return hipGetLastError() == hipSuccess ? 0 : TFRT_E_LAUNCH;
}

int tfrt_param_faces_forward(const double* zero_points, const double* vectors,
                             const double* parameters, int64_t n_vertices, const int32_t* faces,
                             int64_t n_faces, double* face_verts, double* norm, void* stream) {
  if (n_faces < 0 || n_vertices < 0) return TFRT_E_BADARG;
  if (n_faces == 0) return 0;
  if (!zero_points || !vectors || !parameters || !faces || !face_verts || n_vertices == 0)
    return TFRT_E_BADARG;
  hipLaunchKernelGGL(k_param_faces, dim3(cdiv(n_faces, BLOCK)), dim3(BLOCK), 0,
                     static_cast<hipStream_t>(stream), zero_points, vectors, parameters,
                     n_vertices, faces, n_faces, face_verts, norm);
  return hipGetLastError() == hipSuccess ? 0 : TFRT_E_LAUNCH;
}

int tfrt_param_faces_backward(const double* grad_face_verts, const double* grad_norm,
                              const double* face_verts, const int32_t* faces,
                              const uint8_t* update_mask, const double* vectors, int64_t n_faces,
                              int64_t n_vertices, const int32_t* corner_start,
                              const int32_t* corner_list, double* grad_parameters, void* stream) {
  if (n_faces < 0 || n_vertices < 0) return TFRT_E_BADARG;
  if (n_faces == 0) return 0;
  if (!faces || !vectors || !grad_parameters || (!grad_face_verts && !grad_norm) ||
      (grad_norm && !face_verts) || ((corner_start == nullptr) != (corner_list == nullptr)))
    return TFRT_E_BADARG;
  if (corner_start != nullptr) {
    hipLaunchKernelGGL(k_param_faces_bwd_gather, dim3(cdiv(n_vertices * GATHER_LANES, BLOCK)),
                       dim3(BLOCK), 0,
                       static_cast<hipStream_t>(stream), grad_face_verts, grad_norm, face_verts,
                       update_mask, vectors, corner_start, corner_list, n_vertices,
                       grad_parameters);
    return hipGetLastError() == hipSuccess ? 0 : TFRT_E_LAUNCH;
  }
  hipLaunchKernelGGL(k_param_faces_bwd, dim3(cdiv(n_faces, BLOCK)), dim3(BLOCK), 0,
                     static_cast<hipStream_t>(stream), grad_face_verts, grad_norm, face_verts,
                     faces, update_mask, vectors, n_faces, n_vertices, grad_parameters);
  return hipGetLastError() == hipSuccess ? 0 : TFRT_E_LAUNCH;
}

int tfrt_param_faces_forward_multi(const tfrt_face_surface* surfaces, int32_t n_surfaces,
                                   void* stream) {
  if (n_surfaces < 0 || n_surfaces > TFRT_MAX_SURFACES || (n_surfaces > 0 && !surfaces))
    return TFRT_E_BADARG;
  FaceSurfaces b;
  int blocks = 0;
  for (int k = 0; k < TFRT_MAX_SURFACES; ++k) {
    b.first_block[k] = blocks;
    if (k >= n_surfaces) {
      b.s[k] = tfrt_face_surface{};
      continue;
    }
    const tfrt_face_surface& u = surfaces[k];
    if (u.n_faces < 0 || u.n_vertices < 0) return TFRT_E_BADARG;
    if (u.n_faces > 0) {
      if (!u.face_verts) return TFRT_E_BADARG;
      if (!u.copy_from &&
          (!u.zero_points || !u.vectors || !u.parameters || !u.faces || u.n_vertices == 0))
        return TFRT_E_BADARG;
    }
    b.s[k] = u;
    blocks += cdiv(u.n_faces, BLOCK);
  }
  b.first_block[TFRT_MAX_SURFACES] = blocks;
  b.count = n_surfaces;
  if (blocks == 0) return 0;
  hipLaunchKernelGGL(k_param_faces_multi, dim3(blocks), dim3(BLOCK), 0,
                     static_cast<hipStream_t>(stream), b);
  return hipGetLastError() == hipSuccess ? 0 : TFRT_E_LAUNCH;
}

int tfrt_param_faces_backward_multi(const tfrt_face_surface_grad* surfaces, int32_t n_surfaces,
                                    void* stream) {
  if (n_surfaces < 0 || n_surfaces > TFRT_MAX_SURFACES || (n_surfaces > 0 && !surfaces))
    return TFRT_E_BADARG;
  FaceSurfaceGrads b;
  int blocks = 0;
  for (int k = 0; k < TFRT_MAX_SURFACES; ++k) {
    b.first_block[k] = blocks;
    if (k >= n_surfaces) {
      b.s[k] = tfrt_face_surface_grad{};
      continue;
    }
    const tfrt_face_surface_grad& u = surfaces[k];
    if (u.n_vertices < 0) return TFRT_E_BADARG;
    if (u.n_vertices > 0 &&
        (!u.vectors || !u.grad_parameters || !u.corner_start || !u.corner_list ||
         (!u.grad_face_verts && !u.grad_norm) || (u.grad_norm && !u.face_verts)))
      return TFRT_E_BADARG;
    b.s[k] = u;
    blocks += cdiv(u.n_vertices * GATHER_LANES, BLOCK);
  }
  b.first_block[TFRT_MAX_SURFACES] = blocks;
  b.count = n_surfaces;
  if (blocks == 0) return 0;
  hipLaunchKernelGGL(k_param_faces_bwd_gather_multi, dim3(blocks), dim3(BLOCK), 0,
                     static_cast<hipStream_t>(stream), b);
  return hipGetLastError() == hipSuccess ? 0 : TFRT_E_LAUNCH;
}

int tfrt_snell3d(int64_t n, const double* x_start, const double* y_start, const double* z_start,
                 const double* x_end, const double* y_end, const double* z_end,
                 const double* norm, const double* n_in, const double* n_out,
                 double new_ray_length, double* out6, void* stream) {
  if (n < 0) return TFRT_E_BADARG;
  if (n == 0) return 0;
  if (!x_start || !y_start || !z_start || !x_end || !y_end || !z_end || !norm || !n_in ||
      !n_out || !out6)
    return TFRT_E_BADARG;
  hipLaunchKernelGGL(k_snell3d, dim3(cdiv(n, BLOCK)), dim3(BLOCK), 0,
                     static_cast<hipStream_t>(stream), n, x_start, y_start, z_start, x_end, y_end,
                     z_end, norm, n_in, n_out, new_ray_length, out6);
  return hipGetLastError() == hipSuccess ? 0 : TFRT_E_LAUNCH;
}

int tfrt_snell2d(int64_t n, const double* x_start, const double* y_start, const double* x_end,
                 const double* y_end, const double* norm, const double* n_in,
                 const double* n_out, double new_ray_length, double* out4, void* stream) {
  if (n < 0) return TFRT_E_BADARG;
  if (n == 0) return 0;
  if (!x_start || !y_start || !x_end || !y_end || !norm || !n_in || !n_out || !out4)
    return TFRT_E_BADARG;
  hipLaunchKernelGGL(k_snell2d, dim3(cdiv(n, BLOCK)), dim3(BLOCK), 0,
                     static_cast<hipStream_t>(stream), n, x_start, y_start, x_end, y_end, norm,
                     n_in, n_out, new_ray_length, out4);
  return hipGetLastError() == hipSuccess ? 0 : TFRT_E_LAUNCH;
}

int tfrt_selftest_f64(int op, int64_t n, const double* a, const double* b, double* out,
                      void* stream) {
  if (n < 0 || op < TFRT_SELFTEST_DIV || op > TFRT_SELFTEST_ADJ_RSQRT) return TFRT_E_BADARG;
  if (n == 0) return 0;
  if (!a || !out || (!b && (op == TFRT_SELFTEST_DIV || op == TFRT_SELFTEST_MULADD)))
    return TFRT_E_BADARG;
  hipLaunchKernelGGL(k_selftest_f64, dim3(cdiv(n, BLOCK)), dim3(BLOCK), 0,
                     static_cast<hipStream_t>(stream), op, n, a, b, out);
  return hipGetLastError() == hipSuccess ? 0 : TFRT_E_LAUNCH;
}

}  // extern "C"
