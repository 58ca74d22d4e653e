// Per-ray float64 math of the tfrt hot path, shared by the HIP kernels (device) and by the
// CPU test harness in tests/host_math (host).  No memory traffic, no torch, no HIP types.
//
// Reference formulas restated (ecpoppenheimer/TensorFlowRayTrace):
//   exact_triangle   tfrt/geometry.py:286-311 (Cramer six-term sums) + the range tests of
//                    tfrt/engine.py:1138-1141
//   face_normal      tfrt/boundaries.py:918-923  normalize((P1-P0) x (P2-P1))
//   snell3d          tfrt/geometry.py:715-753
//   snell2d          tfrt/geometry.py:601-651
//   adjoint3d        hand-derived reverse of  hit -> projected end -> snell3d  (the part of
//                    tf.GradientTape's work in tfrt/optimizer.py:216-220 that touches rays)
//
// Everything here evaluates in double with FP contraction off so that discrete decisions
// (valid masks, nearest hit) agree with an unfused float64 CPU evaluation of the same sums.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define TFRT_HD __host__ __device__ __forceinline__
#else
#define TFRT_HD inline
#endif

namespace tfrt {

constexpr int CLS_ACTIVE = 0;    // hit an OPTICAL boundary  (engine.py:14 OPTICAL = 0)
constexpr int CLS_FINISHED = 1;  // hit a TARGET boundary    (engine.py:16 TARGET  = 2)
constexpr int CLS_STOPPED = 2;   // hit a STOP boundary      (engine.py:15 STOP    = 1)
constexpr int CLS_DEAD = 3;      // hit nothing

constexpr int CAT_OPTICAL = 0, CAT_STOP = 1, CAT_TARGET = 2;

struct TriHit {
  double ray_u, trig_u, trig_v;
  bool valid;
};

// geometry.py:286-311 + engine.py:1138-1141.  s = ray start (r1), e = ray end (r2),
// P = {xp,yp,zp, x1,y1,z1, x2,y2,z2}.
TFRT_HD TriHit exact_triangle(const double s[3], const double e[3], const double P[9],
                              double eps_int, double eps_size, double eps_start) {
#pragma clang fp contract(off)
  const double a = s[0] - e[0], d = s[1] - e[1], h = s[2] - e[2];
  const double b = P[3] - P[0], f = P[4] - P[1], k = P[5] - P[2];
  const double c = P[6] - P[0], g = P[7] - P[1], l = P[8] - P[2];
  const double q = s[0] - P[0], r = s[1] - P[1], t = s[2] - P[2];

  const double den = a * g * k + b * d * l + c * f * h - a * f * l - b * g * h - c * d * k;
  const double nr = b * l * r + c * f * t + g * k * q - b * g * t - c * k * r - f * l * q;
  const double nu = a * g * t + c * h * r + d * l * q - a * l * r - c * d * t - g * h * q;
  const double nv = a * k * r + b * d * t + f * h * q - a * f * t - b * h * r - d * k * q;

  TriHit o;
  bool valid = fabs(den) >= eps_int;
  const double sd = valid ? den : 1.0;
  o.ray_u = nr / sd;
  o.trig_u = nu / sd;
  o.trig_v = nv / sd;
  valid = valid && (o.trig_u >= -eps_size);
  valid = valid && (o.trig_v >= -eps_size);
  valid = valid && (o.trig_u + o.trig_v <= 1.0 + eps_size);
  valid = valid && (o.ray_u >= eps_start);
  o.valid = valid;
  return o;
}

// hit point = r1 - ray_u * (r1 - r2)      geometry.py:316-318
TFRT_HD void hit_point(const double s[3], const double e[3], double ray_u, double h[3]) {
#pragma clang fp contract(off)
  h[0] = s[0] - ray_u * (s[0] - e[0]);
  h[1] = s[1] - ray_u * (s[1] - e[1]);
  h[2] = s[2] - ray_u * (s[2] - e[2]);
}

// new ray end = hit + L * w (geometry.py:751-752) and the shortened dead ray
// start + L * (end - start) (engine.py:2043-2049): a product and a sum, rounded separately like
// the reference's eager ops.  (Written inline in a kernel these contract to one fma under hipcc's
// default -ffp-contract=fast, and the child ray's end then differs from the reference's in the
// last bit -- enough to flip a nearest-hit tie between coplanar overlapping faces a pass later.)
TFRT_HD double advance(double origin, double L, double dir) {
#pragma clang fp contract(off)
  return origin + L * dir;
}

TFRT_HD double advance_between(double s, double L, double e) {
#pragma clang fp contract(off)
  return s + L * (e - s);
}

TFRT_HD void cross3(const double a[3], const double b[3], double o[3]) {
#pragma clang fp contract(off)
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

TFRT_HD double dot3(const double a[3], const double b[3]) {
#pragma clang fp contract(off)
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

// boundaries.py:918-923: norm = normalize((P1-P0) x (P2-P1)).  Also returns the raw cross
// product C and its length (needed by the adjoint).
TFRT_HD void face_normal(const double P[9], double N[3], double C[3], double* clen) {
#pragma clang fp contract(off)
  const double A[3] = {P[3] - P[0], P[4] - P[1], P[5] - P[2]};
  const double B[3] = {P[6] - P[3], P[7] - P[4], P[8] - P[5]};
  cross3(A, B, C);
  const double len = sqrt(dot3(C, C));
  *clen = len;
  N[0] = C[0] / len;
  N[1] = C[1] / len;
  N[2] = C[2] / len;
}

// tf.math.l2_normalize: x * rsqrt(max(sum(x^2), 1e-12)).  Returns the scale that was applied.
TFRT_HD double l2_normalize3(const double v[3], double o[3]) {
#pragma clang fp contract(off)
  double sq = dot3(v, v);
  if (sq < 1e-12) sq = 1e-12;
  const double inv = 1.0 / sqrt(sq);
  o[0] = v[0] * inv;
  o[1] = v[1] * inv;
  o[2] = v[2] * inv;
  return inv;
}

struct Snell3 {
  double u[3], n[3];  // normalised ray direction and surface normal
  double nu, eta, k;  // n.u, index ratio, radicand
  bool reflect;       // TIR or mirror
  double w[3];        // new direction
};

// The normal of a face as snell3d() uses it: boundaries.py:918-923 followed by the l2_normalize of
// geometry.py:721 -- a function of the face alone, so a trace may form it once per face
// (k_hierarchy_spheres / k_spheres) instead of once per ray and pass: the same operations in the
// same order, hence the same bits.
TFRT_HD void snell_normal(const double P[9], double n[3]) {
  double N[3], C[3], clen;
  face_normal(P, N, C, &clen);
  l2_normalize3(N, n);
}

// The two index ratios of geometry.py:727-733 (tf.math.divide_no_nan both ways): functions of
// the face and the wavelength alone.
TFRT_HD void snell_ratios(double n_in, double n_out, double* n1, double* n2) {
#pragma clang fp contract(off)
  const bool in_safe = n_in != 0.0, out_safe = n_out != 0.0;
  const double nis = in_safe ? n_in : 1.0, nos = out_safe ? n_out : 1.0;
  *n1 = out_safe ? nis / nos : 0.0;
  *n2 = in_safe ? nos / nis : 0.0;
}

// geometry.py:715-753 from the normal and the ratios on (un = snell_normal() of the face,
// (n1, n2) = snell_ratios(), mirror = (n_in == 0)).
TFRT_HD Snell3 snell3d_core(const double s[3], const double h[3], const double un[3], double n1,
                            double n2, bool mirror) {
#pragma clang fp contract(off)
  Snell3 o;
  const double r[3] = {h[0] - s[0], h[1] - s[1], h[2] - s[2]};
  l2_normalize3(r, o.u);
  o.n[0] = un[0];
  o.n[1] = un[1];
  o.n[2] = un[2];
  o.nu = dot3(o.n, o.u);
  const bool internal = o.nu > 0.0;
  o.eta = internal ? n1 : n2;
  const double nu_eta = o.eta * o.nu;
  o.k = 1.0 - o.eta * o.eta + nu_eta * nu_eta;
  const bool tir = o.k < 0.0;
  o.reflect = tir || mirror;
  if (o.reflect) {
    for (int i = 0; i < 3; ++i) o.w[i] = -2.0 * o.nu * o.n[i] + o.u[i];
  } else {
    const double sg = (o.nu > 0.0) ? 1.0 : ((o.nu < 0.0) ? -1.0 : 0.0);
    const double alpha = sg * sqrt(o.k) - nu_eta;
    for (int i = 0; i < 3; ++i) o.w[i] = alpha * o.n[i] + o.eta * o.u[i];
  }
  return o;
}

TFRT_HD Snell3 snell3d_unit(const double s[3], const double h[3], const double un[3],
                            double n_in, double n_out) {
  double n1, n2;
  snell_ratios(n_in, n_out, &n1, &n2);
  return snell3d_core(s, h, un, n1, n2, n_in == 0.0);
}

// geometry.py:715-753.  s = ray start, h = projected ray end (the hit), norm = face normal.
TFRT_HD Snell3 snell3d(const double s[3], const double h[3], const double norm[3],
                       double n_in, double n_out) {
  double un[3];
  l2_normalize3(norm, un);
  return snell3d_unit(s, h, un, n_in, n_out);
}

// floor-mod like tf.math.mod / python %
TFRT_HD double fmod_floor(double x, double m) {
  double r = fmod(x, m);
  if (r != 0.0 && ((r < 0.0) != (m < 0.0))) r += m;
  return r;
}

// geometry.py:601-651.  Returns the new ray angle; new ray = (h, h + L*(cos,sin)).
TFRT_HD double snell2d_angle(double xs, double ys, double xe, double ye, double norm,
                             double n_in, double n_out) {
#pragma clang fp contract(off)
  const double PI = 3.141592653589793;
  norm = fmod_floor(norm, 2 * PI);
  double ray_angle = atan2(ys - ye, xs - xe);
  ray_angle = fmod_floor(ray_angle, 2 * PI);
  double theta1 = norm - ray_angle;
  if (theta1 > PI) theta1 = theta1 - 2 * PI;
  if (theta1 < -PI) theta1 = theta1 + 2 * PI;
  const bool internal = fabs(theta1) >= PI / 2;
  const bool in_safe = n_in != 0.0, out_safe = n_out != 0.0;
  const double nis = in_safe ? n_in : 1.0, nos = out_safe ? n_out : 1.0;
  const double n1 = out_safe ? nis / nos : 0.0;
  const double n2 = in_safe ? nos / nis : 0.0;
  const double n = internal ? n1 : n2;
  if (!internal) norm = norm + PI;
  if (internal) theta1 = theta1 + PI;
  const double theta2 = n * sin(theta1);
  if (fabs(theta2) <= 1.0 && n != 0.0) return norm - asin(theta2);
  return norm + theta1 + PI;
}

// Reciprocal and reciprocal square root for the REVERSE sweep only (the forward's quotients decide
// hits and must round like the reference's): on the device the hardware estimate refined by
// Newton steps to the last bit or two -- 6-8 instructions instead of the ~15 of an IEEE float64
// division or square root; the reverse sweep is bound by float64 issue and its tolerance is
// 1e-8, not the last bit.  On the host (tests/host_math) the plain operations.
TFRT_HD double adj_rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  return r;
#else
  return 1.0 / x;
#endif
}
TFRT_HD double adj_rsqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rsq(x);
  // y <- y + y (1 - x y^2) / 2, twice (the estimate carries ~26 bits)
  double e = __builtin_fma(-x * y, y, 1.0);
  y = __builtin_fma(0.5 * y, e, y);
  e = __builtin_fma(-x * y, y, 1.0);
  y = __builtin_fma(0.5 * y, e, y);
  return y;
#else
  return 1.0 / sqrt(x);
#endif
}

// ------------------------------------------------------------------------------------------
// Reverse-mode of one ray through one pass.
//
// Forward (per ray):   d = e - s ; C = (P1-P0) x (P2-P0) ; t = ((P0-s).C)/(d.C) ; h = s + t d
//   finished/stopped/active-history output : (s, h)
//   active child                            : (h, h + L w),  w = snell3d(s, h, normalize(C))
// The reference's ray_u (six-term Cramer sums) is the same rational function as t above, so
// the derivative is taken through the compact form.
//
// Inputs:  g_s   upstream gradient on s used as an output start (history/finished/stopped)
//          g_h   upstream gradient on h from outputs whose *end* is h, plus the child's start
//          g_ce  upstream gradient on the child's end (h + L w); ignored unless has_child
// Outputs: gs, ge (gradient wrt this pass's input ray), gP[9] (wrt P0,P1,P2 of the hit face),
//          gn[2] (optional: wrt n_in, n_out of the reaction -- "value" mode reads them as ordinary
//          tensors, operation.py:268-272; zero for a mirror / total internal reflection, whose
//          direction does not depend on the ratio, geometry.py:735-747).
TFRT_HD void adjoint3d(const double s[3], const double e[3], const double P[9], double ray_u,
                       bool has_child, double n_in, double n_out, double L,
                       const double g_s[3], const double g_h[3], const double g_ce[3],
                       double gs[3], double ge[3], double gP[9], double* gn = nullptr,
                       int branch = -1, bool gn_wanted = true) {
  // branch: -1 = re-derive the forward's branches here; else bit 0 = the ray met the face from
  // the inside (n.u > 0), bit 1 = it was reflected (mirror or total internal reflection), as the
  // forward pass decided them
  // (gn_wanted: a caller that decides at run time passes its two-element array either way -- a
  // pointer that is sometimes null keeps the array in scratch memory on the GPU)
  if (gn != nullptr) gn[0] = gn[1] = 0.0;
  const double d[3] = {e[0] - s[0], e[1] - s[1], e[2] - s[2]};
  const double E1[3] = {P[3] - P[0], P[4] - P[1], P[5] - P[2]};
  const double E2[3] = {P[6] - P[0], P[7] - P[1], P[8] - P[2]};
  double C[3];
  cross3(E1, E2, C);
  const double t = ray_u;
  double h[3] = {s[0] + t * d[0], s[1] + t * d[1], s[2] + t * d[2]};

  double hb[3] = {g_h[0], g_h[1], g_h[2]};
  double sb[3] = {g_s[0], g_s[1], g_s[2]};
  for (int i = 0; i < 9; ++i) gP[i] = 0.0;

  if (has_child) {
    // Forward quantities recomputed for the reverse step.  They need not reproduce the forward
    // pass bit for bit (the gradient tolerance is 1e-8, not the last bit), so the unit vectors
    // come from one reciprocal each instead of the forward's nine quotients, n = C / |C| skips
    // the forward's second normalisation of an already-unit vector, and only the index ratio
    // that is used is formed: 5 float64 divisions and 3 square roots instead of 14 and 5
    // (a float64 division is ~30 instructions; the kernel is bound by float64 VALU issue).
    const double inv_c = adj_rsqrt(dot3(C, C));
    const double n[3] = {C[0] * inv_c, C[1] * inv_c, C[2] * inv_c};
    const double r[3] = {h[0] - s[0], h[1] - s[1], h[2] - s[2]};
    const double rsq = dot3(r, r);
    const bool clamped = rsq < 1e-12;               // l2_normalize's max(sum x^2, 1e-12)
    const double inv_r = clamped ? 1e6 : adj_rsqrt(rsq);
    const double u[3] = {r[0] * inv_r, r[1] * inv_r, r[2] * inv_r};
    const double nu = dot3(n, u);
    const bool in_safe = n_in != 0.0, out_safe = n_out != 0.0;
    const double nis = in_safe ? n_in : 1.0, nos = out_safe ? n_out : 1.0;
    const bool internal = branch >= 0 ? (branch & 1) != 0 : nu > 0.0;
    const double eta = internal ? (out_safe ? nis / nos : 0.0) : (in_safe ? nos / nis : 0.0);
    const double nu_eta = eta * nu;
    double k = 1.0 - eta * eta + nu_eta * nu_eta;
    const bool reflect = branch >= 0 ? (branch & 2) != 0 : ((k < 0.0) || (n_in == 0.0));
    // (the forward refracted, so ITS radicand was >= 0; the one re-derived here may come out a few
    // ulp below zero at the critical angle)
    if (!reflect && k < 0.0) k = 0.0;
    double wb[3], ub[3], nb[3];
    for (int i = 0; i < 3; ++i) {
      hb[i] += g_ce[i];
      wb[i] = L * g_ce[i];
    }
    double nub;
    if (reflect) {
      nub = -2.0 * dot3(wb, n);
      for (int i = 0; i < 3; ++i) {
        ub[i] = wb[i];
        nb[i] = -2.0 * nu * wb[i];
      }
    } else {
      const double sg = branch >= 0 ? (internal ? 1.0 : (nu < 0.0 ? -1.0 : 0.0))
                                    : ((nu > 0.0) ? 1.0 : ((nu < 0.0) ? -1.0 : 0.0));
      // (k = 0 at the critical angle: 1 / sqrt(k) is infinite there, like the reference's gradient)
      const double irk = k > 0.0 ? adj_rsqrt(k) : INFINITY;
      const double rk = k > 0.0 ? k * irk : 0.0;
      const double alpha = sg * rk - nu_eta;
      const double ab = dot3(wb, n);
      nub = ab * (sg * eta * nu_eta * irk - eta);
      if (gn != nullptr && gn_wanted) {
        // w = alpha n + eta u, alpha = sg sqrt(1 - eta^2 + eta^2 nu^2) - eta nu
        const double etab = dot3(wb, u) + ab * (sg * eta * (nu * nu - 1.0) * irk - nu);
        if (internal) {            // eta = n_in / n_out
          if (out_safe) {
            gn[0] = in_safe ? etab / nos : 0.0;
            gn[1] = -etab * eta / nos;
          }
        } else if (in_safe) {      // eta = n_out / n_in
          gn[1] = out_safe ? etab / nis : 0.0;
          gn[0] = -etab * eta / nis;
        }
      }
      for (int i = 0; i < 3; ++i) {
        nb[i] = alpha * wb[i];
        ub[i] = eta * wb[i];
      }
    }
    for (int i = 0; i < 3; ++i) {
      nb[i] += nub * u[i];
      ub[i] += nub * n[i];
    }
    // u = l2_normalize(h - s)
    if (!clamped) {
      const double uu = dot3(u, ub);
      for (int i = 0; i < 3; ++i) {
        const double rb = (ub[i] - u[i] * uu) * inv_r;
        hb[i] += rb;
        sb[i] -= rb;
      }
    } else {  // clamped branch of l2_normalize: u = r * 1e6
      for (int i = 0; i < 3; ++i) {
        hb[i] += ub[i] * 1e6;
        sb[i] -= ub[i] * 1e6;
      }
    }
    // n = l2_normalize(N), N = C/|C|  ->  dC = (nb - n (n.nb)) / |C|   (|N| = 1)
    const double nn = dot3(n, nb);
    double Cb[3];
    for (int i = 0; i < 3; ++i) Cb[i] = (nb[i] - n[i] * nn) * inv_c;
    // C = E1 x E2
    double E1b[3], E2b[3];
    cross3(E2, Cb, E1b);
    cross3(Cb, E1, E2b);
    for (int i = 0; i < 3; ++i) {
      gP[3 + i] += E1b[i];
      gP[6 + i] += E2b[i];
      gP[i] -= E1b[i] + E2b[i];
    }
  }

  // h = s + t d
  double db[3];
  double tb = 0.0;
  for (int i = 0; i < 3; ++i) {
    sb[i] += hb[i];
    tb += hb[i] * d[i];
    db[i] = t * hb[i];
  }
  // t = num / den, num = (P0 - s).C, den = d.C
  const double den = dot3(d, C);
  const double numb = tb * adj_rcp(den);
  const double denb = -numb * t;
  double Cb[3];
  for (int i = 0; i < 3; ++i) {
    gP[i] += numb * C[i];
    sb[i] -= numb * C[i];
    Cb[i] = numb * (P[i] - s[i]) + denb * d[i];
    db[i] += denb * C[i];
  }
  double E1b[3], E2b[3];
  cross3(E2, Cb, E1b);
  cross3(Cb, E1, E2b);
  for (int i = 0; i < 3; ++i) {
    gP[3 + i] += E1b[i];
    gP[6 + i] += E2b[i];
    gP[i] -= E1b[i] + E2b[i];
    ge[i] = db[i];
    gs[i] = sb[i] - db[i];
  }
}

}  // namespace tfrt
