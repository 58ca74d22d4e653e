// Device functions of the source programs (tfrt_points_program / tfrt_source3d_program): shared by
// csrc/tfrt_source.hip (rays and points into the caller's buffers) and csrc/tfrt_order.hip (the
// coherent order of a program's rays without ever writing them in source order).
#pragma once
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

// (the programs evaluate in float64 for the rays and points they hand out, and in float32 for the
// coherent order's keys, where a cell of the key grid is 2^-10 of the extent)
__device__ __forceinline__ double m_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float m_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double m_fmod(double x, double y) { return fmod(x, y); }
__device__ __forceinline__ float m_fmod(float x, float y) { return fmodf(x, y); }
__device__ __forceinline__ double m_acos(double x) { return acos(x); }
__device__ __forceinline__ float m_acos(float x) { return acosf(x); }
__device__ __forceinline__ double m_fmax(double x, double y) { return fmax(x, y); }
__device__ __forceinline__ float m_fmax(float x, float y) { return fmaxf(x, y); }
__device__ __forceinline__ void m_sincos(double x, double* s, double* c) { sincos(x, s, c); }
__device__ __forceinline__ void m_sincos(float x, float* s, float* c) { sincosf(x, s, c); }
__device__ __forceinline__ void m_sincospi(double x, double* s, double* c) { sincospi(x, s, c); }
__device__ __forceinline__ void m_sincospi(float x, float* s, float* c) { sincospif(x, s, c); }

template <typename F>
__device__ __forceinline__ void quat_rotate(const double qd[4], F v[3]) {
  // v' = v + w t + u x t, t = 2 u x v   (q = (w, u) a unit quaternion)
  const F q[4] = {(F)qd[0], (F)qd[1], (F)qd[2], (F)qd[3]};
  const F t0 = (F)2 * (q[2] * v[2] - q[3] * v[1]);
  const F t1 = (F)2 * (q[3] * v[0] - q[1] * v[2]);
  const F t2 = (F)2 * (q[1] * v[1] - q[2] * v[0]);
  const F r0 = v[0] + q[0] * t0 + (q[2] * t2 - q[3] * t1);
  const F r1 = v[1] + q[0] * t1 + (q[3] * t0 - q[1] * t2);
  const F r2 = v[2] + q[0] * t2 + (q[1] * t1 - q[2] * t0);
  v[0] = r0;
  v[1] = r1;
  v[2] = r2;
}

constexpr double TWO_PI = 6.283185307179586476925286766559;
constexpr double GOLDEN_TURN = 3.14159265358979323846 * (1.0 + 2.2360679774997896964);  // pi (1 + sqrt 5)

// Sample `i` of a points program: the 3-D point (after the transformation) and the two numbers the
// distribution's rank properties are made of (circle: r in [0, 1], theta; sphere: phi, theta).
template <typename F>
__device__ __forceinline__ void eval_points(const tfrt_points_program& pg, int64_t i, F out[3],
                                            F aux[2]) {
  F p[3] = {(F)0, (F)0, (F)0};
  aux[0] = aux[1] = (F)0;
  if (pg.kind == TFRT_PTS_TABLE) {
    const double* row = pg.table + 3 * i;
    p[0] = (F)row[0];
    p[1] = (F)row[1];
    p[2] = (F)row[2];
  } else {
    double ud0, ud1;
    uniform2(pg.seed, pg.stream, (uint64_t)*pg.epoch, (uint64_t)i, &ud0, &ud1);
    const F u0 = (F)ud0, u1 = (F)ud1;
    const F P0 = (F)pg.p[0], P1 = (F)pg.p[1], P2 = (F)pg.p[2], P3 = (F)pg.p[3];
    // theta = turns * pi, folded into the wedge [theta_start, theta_end) when there is one
    // (distributions.py:1396-1447); sine and cosine straight from the half-turns when there is
    // none (sincospi: no range reduction against a rounded pi)
    const bool wedge = !(pg.p[1] == 0.0 && pg.p[2] == TWO_PI);
    auto angle = [&](F half_turns, F* th, F* sn, F* cs) {
      F t = half_turns * (F)3.14159265358979323846;
      if (wedge) {
        const F span = P2 - P1;
        F m = m_fmod(t, span);
        if (m != (F)0 && ((m < (F)0) != (span < (F)0))) m += span;   // (sign of the divisor)
        t = m + P1;
        m_sincos(t, sn, cs);
      } else {
        m_sincospi(half_turns, sn, cs);
      }
      *th = t;
    };
    if (pg.kind == TFRT_PTS_CIRCLE) {            // p = {radius, theta_start, theta_end}
      const F r = m_sqrt(u0);
      F th, sn, cs;
      angle((F)2 * u1, &th, &sn, &cs);
      p[1] = P0 * (r * cs);
      p[2] = P0 * (r * sn);
      aux[0] = r;
      aux[1] = th;
    } else if (pg.kind == TFRT_PTS_SQUARE) {     // p = {x_size, -, -, y_size}
      p[1] = -P0 + ((F)2 * P0) * u0;
      p[2] = -P3 + ((F)2 * P3) * u1;
      aux[0] = p[1];
      aux[1] = p[2];
    } else {                                     // p = {radius, theta_start, theta_end, lower bound}
      const F c = P3 + ((F)1 - P3) * u0;
      // cos(phi) = c (uniform cap) or sqrt(c) (Lambertian: cos^2 is uniform); sin from it
      const F cp = pg.kind == TFRT_PTS_SPHERE_LAMBERT ? m_sqrt(c) : c;
      const F sp = m_sqrt(m_fmax((F)0, ((F)1 - cp) * ((F)1 + cp)));
      F th, sn, cs;
      angle((F)(1.0 + 2.2360679774997896964) * u1, &th, &sn, &cs);   // theta = pi (1 + sqrt 5) u
      p[0] = P0 * cp;
      p[1] = P0 * (sp * cs);
      p[2] = P0 * (sp * sn);
      aux[0] = m_acos(cp);
      aux[1] = th;
    }
    // BasePointTransformation (distributions.py:2014-2120): scale, rotate, translate
    if (pg.has_scale) {
      p[0] *= (F)pg.scale[0];
      p[1] *= (F)pg.scale[1];
      p[2] *= (F)pg.scale[2];
    }
    if (pg.has_quat) quat_rotate<F>(pg.quat, p);
    if (pg.has_shift) {
      p[0] += (F)pg.shift[0];
      p[1] += (F)pg.shift[1];
      p[2] += (F)pg.shift[2];
    }
  }
  out[0] = p[0];
  out[1] = p[1];
  out[2] = p[2];
}

// ray i of the source (natural numbering)
template <typename F>
__device__ __forceinline__ void eval_ray(const tfrt_source3d_program& sp, int64_t i, F s[3],
                                         F e[3]) {
  F a[3] = {(F)0, (F)0, (F)0}, b[3] = {(F)0, (F)0, (F)0}, aux[2];
  const int64_t ia = sp.a.count == 1 ? 0 : i, ib = sp.b.count == 1 ? 0 : i;
  if (sp.kind == TFRT_SRC_APERTURE) {
    eval_points<F>(sp.a, ia, s, aux);
    eval_points<F>(sp.b, ib, e, aux);
    return;
  }
  eval_points<F>(sp.b, ib, b, aux);   // the direction vectors
  if (sp.has_quat) quat_rotate<F>(sp.quat, b);
  if (sp.kind == TFRT_SRC_ANGULAR) {
    eval_points<F>(sp.a, ia, a, aux);
    if (sp.has_quat) quat_rotate<F>(sp.quat, a);
  }
  F st[3], en[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    st[q] = (F)sp.center[q] + a[q];
    en[q] = st[q] + (F)sp.ray_length * b[q];
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    s[q] = sp.swap ? en[q] : st[q];
    e[q] = sp.swap ? st[q] : en[q];
  }
}

// Host-side validity of the programs: everything a kernel dereferences or indexes by
// (tfrt_points_generate, tfrt_source3d_generate and tfrt_source3d_order refuse what fails here with
// TFRT_E_BADARG instead of launching on it).
inline bool points_program_ok(const tfrt_points_program* pg) {
  if (!pg || pg->count < 0) return false;
  if (pg->kind == TFRT_PTS_TABLE) return pg->count == 0 || pg->table != nullptr;
  if (pg->kind < TFRT_PTS_TABLE || pg->kind > TFRT_PTS_SPHERE_LAMBERT) return false;
  return pg->epoch != nullptr;
}

inline bool source_program_ok(const tfrt_source3d_program* sp) {
  if (!sp || sp->n_rays < 0) return false;
  if (sp->kind < TFRT_SRC_APERTURE || sp->kind > TFRT_SRC_ANGULAR) return false;
  if (!points_program_ok(&sp->b)) return false;
  if (sp->kind != TFRT_SRC_POINT && !points_program_ok(&sp->a)) return false;
  // (undense: every input has one sample or one per ray)
  const int64_t ca = sp->kind == TFRT_SRC_POINT ? 1 : sp->a.count, cb = sp->b.count;
  return (ca == 1 || ca == sp->n_rays) && (cb == 1 || cb == sp->n_rays);
}

}  // namespace tfrt
