// CPU harness around tensorflowraytrace_amd/csrc/trace_math.h (TEST ONLY).
// Built by tests/conftest.py with g++ into tests/host_math/_build/libhost_math.so so the
// per-ray float64 math used by the HIP kernels can be checked against the oracle on a
// machine with no GPU.  It is never loaded by the product package.
#include <stdint.h>
#include "trace_math.h"
#include "trace_math2d.h"

extern "C" {

void hm_exact_triangle(int64_t n, const double* s, const double* e, const double* P,
                       double eps_int, double eps_size, double eps_start,
                       double* ray_u, double* trig_u, double* trig_v, uint8_t* valid,
                       double* hit) {
  for (int64_t i = 0; i < n; ++i) {
    tfrt::TriHit h = tfrt::exact_triangle(s + 3 * i, e + 3 * i, P + 9 * i, eps_int, eps_size, eps_start);
    ray_u[i] = h.ray_u; trig_u[i] = h.trig_u; trig_v[i] = h.trig_v; valid[i] = h.valid;
    tfrt::hit_point(s + 3 * i, e + 3 * i, h.ray_u, hit + 3 * i);
  }
}

void hm_snell3d(int64_t n, const double* s, const double* h, const double* P,
                const double* n_in, const double* n_out, double L, double* e_new) {
  for (int64_t i = 0; i < n; ++i) {
    double N[3], C[3], clen;
    tfrt::face_normal(P + 9 * i, N, C, &clen);
    tfrt::Snell3 f = tfrt::snell3d(s + 3 * i, h + 3 * i, N, n_in[i], n_out[i]);
    for (int k = 0; k < 3; ++k) e_new[3 * i + k] = tfrt::advance(h[3 * i + k], L, f.w[k]);
  }
}

void hm_face_normal(int64_t n, const double* P, double* norm) {
  for (int64_t i = 0; i < n; ++i) {
    double C[3], clen;
    tfrt::face_normal(P + 9 * i, norm + 3 * i, C, &clen);
  }
}

// the seam form (geometry.py:671-673): the caller supplies the normal; writes the 6 x n block
void hm_snell3d_norm(int64_t n, const double* s, const double* h, const double* norm,
                     const double* n_in, const double* n_out, double L, double* out6) {
  for (int64_t i = 0; i < n; ++i) {
    tfrt::Snell3 f = tfrt::snell3d(s + 3 * i, h + 3 * i, norm + 3 * i, n_in[i], n_out[i]);
    for (int k = 0; k < 3; ++k) {
      out6[k * n + i] = h[3 * i + k];
      out6[(3 + k) * n + i] = tfrt::advance(h[3 * i + k], L, f.w[k]);
    }
  }
}

void hm_snell2d(int64_t n, const double* xs, const double* ys, const double* xe, const double* ye,
                const double* norm, const double* n_in, const double* n_out, double L,
                double* oxe, double* oye) {
  for (int64_t i = 0; i < n; ++i) {
    double a = tfrt::snell2d_angle(xs[i], ys[i], xe[i], ye[i], norm[i], n_in[i], n_out[i]);
    oxe[i] = xe[i] + L * cos(a);
    oye[i] = ye[i] + L * sin(a);
  }
}

void hm_adjoint3d(int64_t n, const double* s, const double* e, const double* P,
                  const double* ray_u, const uint8_t* has_child, const double* n_in,
                  const double* n_out, double L, const double* g_s, const double* g_h,
                  const double* g_ce, double* gs, double* ge, double* gP) {
  for (int64_t i = 0; i < n; ++i) {
    tfrt::adjoint3d(s + 3 * i, e + 3 * i, P + 9 * i, ray_u[i], has_child[i] != 0, n_in[i], n_out[i], L,
                    g_s + 3 * i, g_h + 3 * i, g_ce + 3 * i, gs + 3 * i, ge + 3 * i, gP + 9 * i);
  }
}

// ... with the gradient w.r.t. the two refractive indices ("value" mode, operation.py:268-272)
void hm_adjoint3d_n(int64_t n, const double* s, const double* e, const double* P,
                    const double* ray_u, const uint8_t* has_child, const double* n_in,
                    const double* n_out, double L, const double* g_s, const double* g_h,
                    const double* g_ce, double* gs, double* ge, double* gP, double* gn) {
  for (int64_t i = 0; i < n; ++i) {
    tfrt::adjoint3d(s + 3 * i, e + 3 * i, P + 9 * i, ray_u[i], has_child[i] != 0, n_in[i], n_out[i], L,
                    g_s + 3 * i, g_h + 3 * i, g_ce + 3 * i, gs + 3 * i, ge + 3 * i, gP + 9 * i,
                    gn + 2 * i);
  }
}

void hm_exact_segment(int64_t n, const double* s, const double* e, const double* seg,
                      double eps_int, double eps_size, double eps_start, double* ray_u,
                      double* seg_u, double* xy, uint8_t* valid) {
  for (int64_t i = 0; i < n; ++i) {
    tfrt::Hit2 h = tfrt::exact_segment(s + 2 * i, e + 2 * i, seg + 4 * i, eps_int, eps_size, eps_start);
    ray_u[i] = h.ray_u; seg_u[i] = h.prim_u; xy[2 * i] = h.x; xy[2 * i + 1] = h.y; valid[i] = h.valid;
  }
}

void hm_exact_arc(int64_t n, const double* s, const double* e, const double* arc,
                  double eps_int, double eps_start, double* ray_u, double* arc_u, double* xy,
                  uint8_t* valid, double* norm) {
  for (int64_t i = 0; i < n; ++i) {
    tfrt::Hit2 h = tfrt::exact_arc(s + 2 * i, e + 2 * i, arc + 5 * i, eps_int, eps_start);
    ray_u[i] = h.ray_u; arc_u[i] = h.prim_u; xy[2 * i] = h.x; xy[2 * i + 1] = h.y; valid[i] = h.valid;
    norm[i] = tfrt::arc_norm(arc[5 * i + 4], h.prim_u);
  }
}

// the trace kernels' variant (valid hits only)
void hm_exact_arc_hit(int64_t n, const double* s, const double* e, const double* arc,
                      double eps_int, double eps_start, double* ray_u, double* arc_u, double* xy,
                      uint8_t* valid) {
  for (int64_t i = 0; i < n; ++i) {
    tfrt::Hit2 h = tfrt::exact_arc_hit(s + 2 * i, e + 2 * i, arc + 5 * i, eps_int, eps_start);
    ray_u[i] = h.ray_u; arc_u[i] = h.prim_u; xy[2 * i] = h.x; xy[2 * i + 1] = h.y; valid[i] = h.valid;
  }
}

void hm_adjoint2d(int64_t n, const double* s, const double* e, const double* prim, int prim_stride,
                  int is_arc, const double* u, const uint8_t* has_child, const double* n_in,
                  const double* n_out, double L, const double* g_s, const double* g_h,
                  const double* g_ce, double* gs, double* ge, double* gprim) {
  for (int64_t i = 0; i < n; ++i) {
    tfrt::adjoint2d(s + 2 * i, e + 2 * i, prim + prim_stride * i, is_arc != 0, u[i], has_child[i] != 0,
                    n_in[i], n_out[i], L, g_s + 2 * i, g_h + 2 * i, g_ce + 2 * i, gs + 2 * i,
                    ge + 2 * i, gprim + 5 * i);
  }
}

// the opt-in finite form of the total-internal-reflection gradient (tfrt_scene2d.finite_tir_gradient)
void hm_adjoint2d_finite(int64_t n, const double* s, const double* e, const double* prim,
                         int prim_stride, int is_arc, const double* u, const uint8_t* has_child,
                         const double* n_in, const double* n_out, double L, const double* g_s,
                         const double* g_h, const double* g_ce, double* gs, double* ge,
                         double* gprim) {
  for (int64_t i = 0; i < n; ++i) {
    tfrt::adjoint2d(s + 2 * i, e + 2 * i, prim + prim_stride * i, is_arc != 0, u[i], has_child[i] != 0,
                    n_in[i], n_out[i], L, g_s + 2 * i, g_h + 2 * i, g_ce + 2 * i, gs + 2 * i,
                    ge + 2 * i, gprim + 5 * i, true);
  }
}
}
