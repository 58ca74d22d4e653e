// Per-ray float64 math of the 2-D tfrt hot path (segments + arcs), shared by the HIP kernels
// and the CPU test harness.  Restates:
//   exact_segment   tfrt/geometry.py:136-160 + range tests of tfrt/engine.py:722-724
//   exact_arc       tfrt/geometry.py:464-530 + tfrt/engine.py:803-845 (root choice) +
//                   geometry.angle_in_interval (geometry.py:790-802)
//   arc_norm        tfrt/engine.py:667-670
//   adjoint2d       hand-derived reverse of  hit -> projected end -> snells_law_2D
#pragma once
#include "trace_math.h"

namespace tfrt {

constexpr double PI_D = 3.141592653589793;

struct Hit2 {
  double ray_u;  // parameter along the ray
  double prim_u; // segment parameter, or arc angle of the hit
  double x, y;
  bool valid;
};

// seg = {x_start, y_start, x_end, y_end}
TFRT_HD Hit2 exact_segment(const double s[2], const double e[2], const double seg[4],
                           double eps_int, double eps_size, double eps_start) {
#pragma clang fp contract(off)
  const double x1 = e[0] - s[0], y1 = e[1] - s[1];
  const double x2 = seg[2] - seg[0], y2 = seg[3] - seg[1];
  const double den = x1 * y2 - y1 * x2;
  Hit2 o;
  bool valid = fabs(den) >= eps_int;
  const double inv = 1.0 / (valid ? den : 1.0);
  const double u = valid ? (x2 * (s[1] - seg[1]) - y2 * (s[0] - seg[0])) * inv : 1.0;
  const double v = valid ? (y1 * (seg[0] - s[0]) - x1 * (seg[1] - s[1])) * inv : 1.0;
  o.ray_u = u;
  o.prim_u = v;
  o.x = s[0] + u * x1;
  o.y = s[1] + u * y1;
  valid = valid && (v >= -eps_size) && (v <= 1.0 + eps_size) && (u >= eps_start);
  o.valid = valid;
  return o;
}

TFRT_HD bool angle_in_interval(double angle, double start, double end) {
#pragma clang fp contract(off)
  double ra = angle - start;
  if (ra < 0.0) ra = ra + 2 * PI_D;
  double re = end - start;
  if (re < 0.0) re = re + 2 * PI_D;
  return ra <= re;
}

// arc = {x_center, y_center, angle_start, angle_end, radius}.  `eps_start_self` replaces
// eps_start for the root test (used to keep a float32-rounded ray from re-hitting, at u ~ 0,
// the arc it starts on).
TFRT_HD Hit2 exact_arc(const double s[2], const double e[2], const double arc[5],
                       double eps_int, double eps_start) {
#pragma clang fp contract(off)
  const double xc = arc[0], yc = arc[1], a1 = arc[2], a2 = arc[3], r = arc[4];
  const double inv_r = 1.0 / r;
  const double xr = (s[0] - xc) * inv_r, yr = (s[1] - yc) * inv_r;
  const double xd = (e[0] - s[0]) * inv_r, yd = (e[1] - s[1]) * inv_r;
  const double a = xd * xd + yd * yd;
  const double b = 2.0 * xr * xd + 2.0 * yr * yd;
  const double c = xr * xr + yr * yr - 1.0;
  double rad = b * b - 4.0 * a * c;
  if (fabs(rad) < eps_int) rad = 0.0;
  const bool rad_less = rad < 0.0;
  const double sr = sqrt(rad_less ? 1.0 : rad);
  double um = rad_less ? 1.0 : (-b - sr);
  double up = rad_less ? 1.0 : (-b + sr);
  const bool azero = fabs(a) < eps_int;
  const double inv = 1.0 / (azero ? 1.0 : 2 * a);
  um = azero ? 1.0 : um * inv;
  up = azero ? 1.0 : up * inv;
  const bool base = !rad_less && !azero;
  const double xm = s[0] + (e[0] - s[0]) * um, ym = s[1] + (e[1] - s[1]) * um;
  const double xp = s[0] + (e[0] - s[0]) * up, yp = s[1] + (e[1] - s[1]) * up;
  bool vm = base && (um >= eps_start);
  bool vp = base && (up >= eps_start);
  double angm = 0.0, angp = 0.0;
  if (vm) {
    angm = atan2(ym - yc, xm - xc);
    vm = angle_in_interval(angm, a1, a2);
  }
  if (vp) {
    angp = atan2(yp - yc, xp - xc);
    vp = angle_in_interval(angp, a1, a2);
  }
  // engine.py:836-845: minus wins only if strictly closer (after sentinel fill)
  const bool choose_minus = vm && (!vp || um < up);
  Hit2 o;
  o.valid = vm || vp;
  if (choose_minus) {
    o.ray_u = um; o.prim_u = angm; o.x = xm; o.y = ym;
  } else {
    o.ray_u = up; o.prim_u = vp ? angp : atan2(yp - yc, xp - xc); o.x = xp; o.y = yp;
  }
  return o;
}

// exact_arc() for callers that only use VALID hits (the trace kernels): the same roots, the same
// atan2 of the same points and the same choice, but an angle is only computed for a root that can
// be the answer -- the minus root when it lies in range, else the plus root; the plus root a
// second time only when the minus root lay in range and missed the arc's interval.  (exact_arc
// evaluates three atan2 -- ~100 float64 instructions each -- on nearly every call of a wavefront;
// this one one, sometimes two.)  When !valid the other fields are unspecified.
TFRT_HD Hit2 exact_arc_hit(const double s[2], const double e[2], const double arc[5],
                           double eps_int, double eps_start) {
#pragma clang fp contract(off)
  const double xc = arc[0], yc = arc[1], a1 = arc[2], a2 = arc[3], r = arc[4];
  const double inv_r = 1.0 / r;
  const double xr = (s[0] - xc) * inv_r, yr = (s[1] - yc) * inv_r;
  const double xd = (e[0] - s[0]) * inv_r, yd = (e[1] - s[1]) * inv_r;
  const double a = xd * xd + yd * yd;
  const double b = 2.0 * xr * xd + 2.0 * yr * yd;
  const double c = xr * xr + yr * yr - 1.0;
  double rad = b * b - 4.0 * a * c;
  if (fabs(rad) < eps_int) rad = 0.0;
  const bool rad_less = rad < 0.0;
  const double sr = sqrt(rad_less ? 1.0 : rad);
  double um = rad_less ? 1.0 : (-b - sr);
  double up = rad_less ? 1.0 : (-b + sr);
  const bool azero = fabs(a) < eps_int;
  const double inv = 1.0 / (azero ? 1.0 : 2 * a);
  um = azero ? 1.0 : um * inv;
  up = azero ? 1.0 : up * inv;
  const bool base = !rad_less && !azero;
  const bool em = base && (um >= eps_start), ep = base && (up >= eps_start);
  Hit2 o;
  o.valid = false;
  o.ray_u = up;
  o.prim_u = o.x = o.y = 0.0;
  if (em || ep) {
    // (um <= up: when both are valid the minus root wins -- and when they coincide the two
    // candidates are the same point with the same angle)
    const double u1 = em ? um : up;
    const double x1 = s[0] + (e[0] - s[0]) * u1, y1 = s[1] + (e[1] - s[1]) * u1;
    const double ang1 = atan2(y1 - yc, x1 - xc);
    if (angle_in_interval(ang1, a1, a2)) {
      o.valid = true;
      o.ray_u = u1;
      o.prim_u = ang1;
      o.x = x1;
      o.y = y1;
    } else if (em && ep) {
      const double xp = s[0] + (e[0] - s[0]) * up, yp = s[1] + (e[1] - s[1]) * up;
      const double angp = atan2(yp - yc, xp - xc);
      if (angle_in_interval(angp, a1, a2)) {
        o.valid = true;
        o.ray_u = up;
        o.prim_u = angp;
        o.x = xp;
        o.y = yp;
      }
    }
  }
  return o;
}

// engine.py:667-670
TFRT_HD double arc_norm(double radius, double arc_u) {
#pragma clang fp contract(off)
  const double n = (radius < 0.0) ? arc_u + PI_D : arc_u;
  return fmod_floor(n + PI_D, 2 * PI_D) - PI_D;
}

// engine.py:580-586
TFRT_HD double segment_norm(const double seg[4]) {
#pragma clang fp contract(off)
  return atan2(seg[3] - seg[1], seg[2] - seg[0]) + PI_D / 2.0;
}

// ------------------------------------------------------------------------------------------
// Reverse of one 2-D ray through one pass.  Forward: d = e - s; hit h = s + u d on a segment
// (prim = {A, B}) or an arc (prim = {c, a1, a2, r}); outputs (s, h) for finished / stopped /
// history, child (h, h + L (cos t, sin t)) with t = snells_law_2D(s, h, norm, n_in, n_out).
//   g_s, g_h, g_ce as in adjoint3d.  gprim: 4 (segment) or 5 (arc) gradients.
// Total internal reflection: the reference evaluates asin(theta2), |theta2| > 1, in the branch
// tf.where does not select (geometry.py:640-646); that branch's gradient is 0 * d asin = NaN,
// whatever the upstream gradient (zero included), so theta1 -- i.e. the norm angle and the ray
// angle -- receive NaN and with them the boundary entries and the parent ray.  Reproduced unless
// finite_tir (tfrt_scene2d.finite_tir_gradient): then the reflect branch's own gradient.
TFRT_HD void adjoint2d(const double s[2], const double e[2], const double* prim, bool is_arc,
                       double u, bool has_child, double n_in, double n_out, double L,
                       const double g_s[2], const double g_h[2], const double g_ce[2],
                       double gs[2], double ge[2], double gprim[5], bool finite_tir = false) {
  const double d[2] = {e[0] - s[0], e[1] - s[1]};
  const double h[2] = {s[0] + u * d[0], s[1] + u * d[1]};
  double hb[2] = {g_h[0], g_h[1]};
  double sb[2] = {g_s[0], g_s[1]};
  for (int i = 0; i < 5; ++i) gprim[i] = 0.0;
  double normb = 0.0;  // gradient on the surface-normal angle

  if (has_child) {
    hb[0] += g_ce[0];
    hb[1] += g_ce[1];
    // replay snells_law_2D to get the branch and the angles
    double norm;
    if (is_arc) {
      norm = arc_norm(prim[4], atan2(h[1] - prim[1], h[0] - prim[0]));
    } else {
      norm = segment_norm(prim);
    }
    const double PI = PI_D;
    double nm = fmod_floor(norm, 2 * PI);
    double ra = fmod_floor(atan2(s[1] - h[1], s[0] - h[0]), 2 * PI);
    double th1 = nm - ra;
    if (th1 > PI) th1 -= 2 * PI;
    if (th1 < -PI) th1 += 2 * PI;
    const bool internal = fabs(th1) >= PI / 2;
    const bool in_safe = n_in != 0.0, out_safe = n_out != 0.0;
    const double nis = in_safe ? n_in : 1.0, nos = out_safe ? n_out : 1.0;
    const double n1 = out_safe ? nis / nos : 0.0, n2 = in_safe ? nos / nis : 0.0;
    const double n = internal ? n1 : n2;
    const double norm_eff = internal ? nm : nm + PI;
    const double th1e = internal ? th1 + PI : th1;
    const double th2 = n * sin(th1e);
    const bool refr = (fabs(th2) <= 1.0) && (n != 0.0);
    const double newa = refr ? norm_eff - asin(th2) : norm_eff + th1e + PI;
    const double newb = L * (-sin(newa) * g_ce[0] + cos(newa) * g_ce[1]);
    double rab;  // gradient on ray_angle
    if (refr) {
      const double k = n * cos(th1e) / sqrt(1.0 - th2 * th2);
      normb = newb * (1.0 - k);
      rab = newb * k;
    } else if (fabs(th2) > 1.0 && !finite_tir) {
      normb = rab = __builtin_nan("");
    } else {  // (mirror, n == 0: theta2 = 0 and asin's unselected gradient is a plain 0)
      normb = newb * 2.0;
      rab = -newb;
    }
    // ray_angle = atan2(s.y - h.y, s.x - h.x)
    const double qx = s[0] - h[0], qy = s[1] - h[1];
    const double q2 = qx * qx + qy * qy;
    if (q2 > 0.0) {
      const double gx = -qy / q2 * rab, gy = qx / q2 * rab;  // d atan2(qy,qx) = (-qy dqx + qx dqy)/q2
      sb[0] += gx; sb[1] += gy;
      hb[0] -= gx; hb[1] -= gy;
    }
    if (is_arc) {
      // norm = atan2(h - c) (+ const)
      const double px = h[0] - prim[0], py = h[1] - prim[1];
      const double p2 = px * px + py * py;
      const double gx = -py / p2 * normb, gy = px / p2 * normb;
      hb[0] += gx; hb[1] += gy;
      gprim[0] -= gx; gprim[1] -= gy;
    } else {
      // norm = atan2(Sy, Sx) + pi/2, S = B - A
      const double Sx = prim[2] - prim[0], Sy = prim[3] - prim[1];
      const double S2 = Sx * Sx + Sy * Sy;
      const double gSx = -Sy / S2 * normb, gSy = Sx / S2 * normb;
      gprim[0] -= gSx; gprim[1] -= gSy; gprim[2] += gSx; gprim[3] += gSy;
    }
  }

  // h = s + u d
  double ub = hb[0] * d[0] + hb[1] * d[1];
  double db[2] = {u * hb[0], u * hb[1]};
  sb[0] += hb[0];
  sb[1] += hb[1];
  if (is_arc) {
    // F(u) = |s + u d - c|^2 - r^2 = 0
    const double px = h[0] - prim[0], py = h[1] - prim[1];
    const double fu = px * d[0] + py * d[1];  // (1/2) dF/du
    const double k = -ub / fu;
    // du = -[p.(ds + u dd - dc) - r dr] / fu
    sb[0] += k * px; sb[1] += k * py;
    db[0] += k * u * px; db[1] += k * u * py;
    gprim[0] -= k * px; gprim[1] -= k * py;
    gprim[4] -= k * prim[4];
  } else {
    // u = num/den, num = Sx (sy - Ay) - Sy (sx - Ax), den = dx Sy - dy Sx
    const double Ax = prim[0], Ay = prim[1];
    const double Sx = prim[2] - Ax, Sy = prim[3] - Ay;
    const double den = d[0] * Sy - d[1] * Sx;
    const double numb = ub / den, denb = -ub * u / den;
    double gSx = numb * (s[1] - Ay) - denb * d[1];
    double gSy = -numb * (s[0] - Ax) + denb * d[0];
    sb[0] += -numb * Sy;
    sb[1] += numb * Sx;
    double gAx = numb * Sy, gAy = -numb * Sx;
    db[0] += denb * Sy;
    db[1] += -denb * Sx;
    gprim[0] += gAx - gSx; gprim[1] += gAy - gSy; gprim[2] += gSx; gprim[3] += gSy;
  }
  ge[0] = db[0]; ge[1] = db[1];
  gs[0] = sb[0] - db[0]; gs[1] = sb[1] - db[1];
}

}  // namespace tfrt
