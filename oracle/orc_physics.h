/*
 * orc_physics.h -- TEST INFRASTRUCTURE (oracle), not product code.
 *
 * Plain-C (real) restatement of the formula library the reference's lgrngn path uses.
 * Every expression keeps the reference's operator order so that results agree with the
 * reference serial backend to the last bit wherever only +,-,*,/,sqrt are involved
 * (libm calls are glibc's, as in the reference's CPU build).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this.
 */
#ifndef ORC_PHYSICS_H
#define ORC_PHYSICS_H
#include <tgmath.h>          /* type-generic libm: sqrt(x) is sqrtf for a float x -- the calls a real_t = float build of the reference makes */
#include <float.h>
#include <stdint.h>

/* real = the reference's real_t.  The oracle proper is the real build (liblcx_oracle.so, pinned on the reference's data); the SAME
 * source with -DORC_REAL=float -fsingle-precision-constant is the float flavour (liblcx_oracle_f32.so): float storage, float libm,
 * float literals -- what `particles_t<float, ...>` computes with, e.g. in the kinematic_2D set-up that the reference runs in float
 * (models/kinematic_2D/tests/paper_GMD_2015/fig_a/calc.cpp:36-39).  dbl = what stays double in EVERY flavour: the C ABI's test hooks
 * and by-value arguments, the random engine's draws, and the two places where the reference itself forces double
 * (hskpng_ijk.ipp:171, kappa_koehler.hpp:154-166). */
#ifndef ORC_REAL
#define ORC_REAL double
#endif
typedef ORC_REAL real;
typedef double dbl;
#define ORC_EPS ((real)(sizeof(real) == 4 ? FLT_EPSILON : DBL_EPSILON))       /* std::numeric_limits<real_t> */
#define ORC_MAX ((real)(sizeof(real) == 4 ? FLT_MAX : DBL_MAX))
#define ORC_MIN ((real)(sizeof(real) == 4 ? FLT_MIN : DBL_MIN))

/* ---- constants: include/libcloudph++/common/moist_air.hpp:26-112, const_cp.hpp:22-26,
 *      earth.hpp:16-22, theta_std.hpp:20, molar_mass.hpp:23-24 ---- */
static const real c_pd = 1005, c_pv = 1850, c_pw = 4218;
static const real M_d = 0.02897;
#define M_v ((1 * 1e-3) + (17 * 1e-3))
static const real kaBoNA = 8.3144621;
#define R_d (kaBoNA / M_d)
#define R_v (kaBoNA / M_v)
#define eps_v (M_v / M_d)
static const real rho_w = 1e3, D_0 = 2.26e-5, K_0 = 2.4e-2;
static const real p_tri = 611.73, T_tri = 273.16, l_tri = 2.5e6;
static const real p_1000 = 100000, g_earth = 9.81, p_stp = 101325;
#define T_stp (273.15 + 15)
#define rho_stp (p_stp / T_stp / R_d)
#define ORC_PI 3.141592653589793238462643383279502884

static inline real dmin(real a, real b) { return b < a ? b : a; }  /* std::min */
static inline real dmax(real a, real b) { return a < b ? b : a; }  /* std::max */

/* theta_dry.hpp:24-55 */
static inline real theta_dry_T(real th, real rhod)
{
  return pow(th * pow(rhod * R_d / p_1000, R_d / c_pd), c_pd / (c_pd - R_d));
}
static inline real theta_dry_p(real rhod, real r, real T) { return rhod * (R_d + r * R_v) * T; }
/* theta_std.hpp:35-41 */
static inline real theta_std_exner(real p) { return pow(p / p_1000, R_d / c_pd); }
/* const_cp.hpp:82-86 */
static inline real l_v(real T) { return l_tri + (c_pv - c_pw) * (T - T_tri); }
/* theta_dry.hpp:60-65 */
static inline real d_th_d_rv(real T, real th) { return -th / T * l_v(T) / c_pd; }
/* moist_air.hpp:77-83 */
static inline real p_v(real p, real r) { return p * r / (r + eps_v); }
/* const_cp.hpp:34-43 */
static inline real p_vs(real T)
{
  return p_tri * exp(
    (l_tri + (c_pw - c_pv) * T_tri) / R_v * (1. / T_tri - 1. / T)
    - (c_pw - c_pv) / R_v * log(T / T_tri));
}
/* const_cp.hpp:58-64 */
static inline real r_vs(real T, real p) { return eps_v / (p / p_vs(T) - 1); }
/* tetens.hpp:12-37 */
static inline real tet_p_vs(real T)
{
  const real Tc = T - 273.15;
  return 6.1078e2 * exp((17.27 * Tc) / (Tc + 237.3));
}
static inline real tet_r_vs(real T, real p)
{
  const real Tc = T - 273.15;
  return 380. / (p * exp(-17.2693882 * Tc / (T - 35.86)) - 610.9);
}
/* hskpng_Tpr.ipp:64-97 */
static inline real RH_of(int formula, real p, real rv, real T)
{
  switch (formula) {
    case 0: return p_v(p, rv) / p_vs(T);
    case 1: return rv / r_vs(T, p);
    case 2: return p_v(p, rv) / tet_p_vs(T);
    default: return rv / tet_r_vs(T, p);
  }
}
/* vterm.hpp:22-31 */
static inline real visc(real T)
{
  const real tt = T / T_tri;
  return (1.72 * 1e-5) * (393. / (T + 120.)) * (tt * sqrt(tt));
}
/* mean_free_path.hpp:16-51 */
static inline real lambda_D_of(real T) { return 2. * D_0 / sqrt(2. * (R_v * T)); }
static inline real lambda_K_of(real T, real p) { return .8 * (K_0 * T / p) / sqrt(2. * (R_d * T)); }
/* kelvin_term.hpp:25-50 */
static inline real sg_surf(real T) { return 0.07275 * (1. - 0.002 * (T - 291.)); }
static inline real kelvin_A(real T) { return 2. * sg_surf(T) / R_v / T / rho_w; }
static inline real klvntrm(real r, real T) { return exp(kelvin_A(T) / r); }
/* kappa_koehler.hpp:31-54 */
static inline real rw3_eq_nokelvin(real rd3, real kappa, real RH)
{
  return rd3 * (1 - RH * (1 - kappa)) / (1 - RH);
}
static inline real a_w(real rw3, real rd3, real kappa)
{
  return (rw3 - rd3) / (rw3 - rd3 * (1. - kappa));
}
/* ventil.hpp:16-80, transition_regime.hpp:15-20 */
static inline real vent_Re(real vt, real rw, real rho, real eta) { return vt * (2. * rw) * rho / eta; }
static inline real vent_Nu(real Pr, real Re)
{
  return 1. + cbrt(1. + Re * Pr) * dmax(1., pow(Re, .077));
}
static inline real trans_beta(real Kn) { return (1 + Kn) / (1 + 1.71 * Kn + 1.33 * Kn * Kn); }
/* maxwell-mason.hpp:15-47 */
static inline real mm_rdrdt(real D, real K, real rho_v, real T, real RH, real aw, real klv)
{
  const real lv = l_v(T);
  return (1. - aw * klv / RH) / rho_w /
         (1. / D / rho_v + lv / K / RH / T * (lv / R_v / T - 1.));
}

/* ---- TOMS 748 (Alefeld, Potra, Shi 1995), the variant vendored by the reference:
 *      include/libcloudph++/common/detail/toms748.hpp:60-454 ---- */
typedef real (*orc_fn)(real x, void *ctx);

static inline int orc_tol_reached(real eps, real a, real b)
{                                                   /* toms748.hpp:267-282 */
  return fabs(a - b) <= eps * dmin(fabs(a), fabs(b));
}
static inline real orc_eps_tolerance(unsigned bits)
{
  return dmax((real)ldexpf(1.0f, 1 - (int)bits), 4 * ORC_EPS);
}
/* The root finders' tolerance follows sizeof(real_t) in the reference (eps_tolerance<real_t>(sizeof(real_t) * 8 / 4), config.hpp:39,
 * toms748.hpp:445-471): 2^-15 for real, 2^-7 for float.  This oracle computes in real; orc_set_real_bytes(4) makes it iterate to
 * FLOAT's tolerance, so that a float build of the product is compared with the iterates its own arithmetic is meant to take
 * (tests/test_hip_configs.py, the icicle set-up, which the reference runs in float).  Process-wide: an object takes its condensation
 * tolerance from it when it is created, rw3_eq reads it at every call -- a test sets it for the length of its run and resets it to 8. */
static unsigned orc_real_bytes_v = sizeof(real);
static inline real orc_real_eps(void) { return orc_eps_tolerance(orc_real_bytes_v * 8 / 4); }
static inline real t748_safe_div(real num, real denom, real r)
{                                                   /* toms748.hpp:124-138 */
  if (fabs(denom) < 1 && fabs(denom * ORC_MAX) <= fabs(num)) return r;
  return num / denom;
}
static inline real t748_secant(real a, real b, real fa, real fb)
{                                                   /* toms748.hpp:140-160 */
  const real tol = ORC_EPS * 5;
  const real c = a - (fa / (fb - fa)) * (b - a);
  if (c <= a + fabs(a) * tol || c >= b - fabs(b) * tol) return (a + b) / 2;
  return c;
}
static inline real t748_quadratic(real a, real b, real d, real fa, real fb, real fd, unsigned count)
{                                                   /* toms748.hpp:162-222 */
  real B = t748_safe_div(fb - fa, b - a, ORC_MAX);
  real A = t748_safe_div(fd - fb, d - b, ORC_MAX);
  A = t748_safe_div(A - B, d - a, 0.);
  if (A == 0) return t748_secant(a, b, fa, fb);
  real c = copysign(1., A * fa) > 0 ? a : b;
  for (unsigned i = 1; i <= count; ++i)
    c -= t748_safe_div(fa + (B + A * (c - b)) * (c - a), B + A * (2 * c - a - b), 1 + c - a);
  if (c <= a || c >= b) c = t748_secant(a, b, fa, fb);
  return c;
}
static inline real t748_cubic(real a, real b, real d, real e, real fa, real fb, real fd, real fe)
{                                                   /* toms748.hpp:224-262 */
  const real q11 = (d - e) * fd / (fe - fd);
  const real q21 = (b - d) * fb / (fd - fb);
  const real q31 = (a - b) * fa / (fb - fa);
  const real d21 = (b - d) * fd / (fd - fb);
  const real d31 = (a - b) * fb / (fb - fa);
  const real q22 = (d21 - q11) * fb / (fe - fb);
  const real q32 = (d31 - q21) * fa / (fd - fa);
  const real d32 = (d31 - q21) * fd / (fd - fa);
  const real q33 = (d32 - q22) * fa / (fe - fa);
  real c = q31 + q32 + q33 + a;
  if (c <= a || c >= b) c = t748_quadratic(a, b, d, fa, fb, fd, 3);
  return c;
}
typedef struct { real a, b, fa, fb, d, fd; } t748_state;
static inline void t748_bracket(orc_fn f, void *ctx, t748_state *s, real c)
{                                                   /* toms748.hpp:60-122 */
  const real tol = ORC_EPS * 2;
  if ((s->b - s->a) < 2 * tol * s->a) c = s->a + (s->b - s->a) / 2;
  else if (c <= s->a + fabs(s->a) * tol) c = s->a + fabs(s->a) * tol;
  else if (c >= s->b - fabs(s->b) * tol) c = s->b - fabs(s->a) * tol;
  const real fc = f(c, ctx);
  if (fc == 0) { s->a = c; s->fa = 0; s->d = 0; s->fd = 0; return; }
  if (copysign(1., s->fa * fc) < 0) { s->d = s->b; s->fd = s->fb; s->b = c; s->fb = fc; }
  else                              { s->d = s->a; s->fd = s->fa; s->a = c; s->fa = fc; }
}
static inline int t748_prof(const t748_state *s, real fe)
{
  const real md = ORC_MIN * 32;
  return fabs(s->fa - s->fb) < md || fabs(s->fa - s->fd) < md || fabs(s->fa - fe) < md ||
         fabs(s->fb - s->fd) < md || fabs(s->fb - fe) < md || fabs(s->fd - fe) < md;
}
static inline real orc_toms748(orc_fn f, void *ctx, real ax, real bx, real fax, real fbx,
                                 real eps, uintmax_t *max_iter)
{                                                   /* toms748.hpp:289-431 */
  uintmax_t count = *max_iter;
  t748_state s = {ax, bx, fax, fbx, 0, 0};
  real c, u, fu, a0, b0, e, fe;
  const real mu = 0.5;
  if (orc_tol_reached(eps, s.a, s.b) || s.fa == 0 || s.fb == 0) {
    *max_iter = 0;
    if (s.fa == 0) s.b = s.a; else if (s.fb == 0) s.a = s.b;
    return (s.a + s.b) / 2;
  }
  fe = e = s.fd = 1e5f;
  if (s.fa != 0) {
    c = t748_secant(s.a, s.b, s.fa, s.fb);
    t748_bracket(f, ctx, &s, c);
    --count;
    if (count && s.fa != 0 && !orc_tol_reached(eps, s.a, s.b)) {
      c = t748_quadratic(s.a, s.b, s.d, s.fa, s.fb, s.fd, 2);
      e = s.d; fe = s.fd;
      t748_bracket(f, ctx, &s, c);
      --count;
    }
  }
  while (count && s.fa != 0 && !orc_tol_reached(eps, s.a, s.b)) {
    a0 = s.a; b0 = s.b;
    c = t748_prof(&s, fe) ? t748_quadratic(s.a, s.b, s.d, s.fa, s.fb, s.fd, 2)
                          : t748_cubic(s.a, s.b, s.d, e, s.fa, s.fb, s.fd, fe);
    e = s.d; fe = s.fd;
    t748_bracket(f, ctx, &s, c);
    if (0 == --count || s.fa == 0 || orc_tol_reached(eps, s.a, s.b)) break;
    c = t748_prof(&s, fe) ? t748_quadratic(s.a, s.b, s.d, s.fa, s.fb, s.fd, 3)
                          : t748_cubic(s.a, s.b, s.d, e, s.fa, s.fb, s.fd, fe);
    t748_bracket(f, ctx, &s, c);
    if (0 == --count || s.fa == 0 || orc_tol_reached(eps, s.a, s.b)) break;
    if (fabs(s.fa) < fabs(s.fb)) { u = s.a; fu = s.fa; } else { u = s.b; fu = s.fb; }
    c = u - 2 * (fu / (s.fb - s.fa)) * (s.b - s.a);
    if (fabs(c - u) > (s.b - s.a) / 2) c = s.a + (s.b - s.a) / 2;
    e = s.d; fe = s.fd;
    t748_bracket(f, ctx, &s, c);
    if (0 == --count || s.fa == 0 || orc_tol_reached(eps, s.a, s.b)) break;
    if ((s.b - s.a) < mu * (b0 - a0)) continue;
    e = s.d; fe = s.fd;
    t748_bracket(f, ctx, &s, s.a + (s.b - s.a) / 2);
    --count;
  }
  *max_iter -= count;
  if (s.fa == 0) s.b = s.a; else if (s.fb == 0) s.a = s.b;
  return (s.a + s.b) / 2;
}

/* ---- equilibrium wet radius, kappa_koehler.hpp:58-146 ---- */
typedef struct { real RH, rd3, kappa, T; } rw3eq_ctx;
static inline real rw3_eq_minfun(real rw3, void *vc)
{
  const rw3eq_ctx *c = (const rw3eq_ctx *)vc;
  return c->RH - a_w(rw3, c->rd3, c->kappa) * klvntrm(cbrt(rw3), c->T);
}
static inline real rw3_eq(real rd3, real kappa, real RH, real T)
{
  if (kappa == 0) return rd3;
  rw3eq_ctx c = {RH, rd3, kappa, T};
  const real a = rd3, b = rw3_eq_nokelvin(rd3, kappa, RH);
  uintmax_t it = 100;      /* toms748.hpp:445-471: default n_iter 100, eps_tolerance(sizeof(T)*8/4) */
  return orc_toms748(rw3_eq_minfun, &c, a, b, rw3_eq_minfun(a, &c), rw3_eq_minfun(b, &c), orc_real_eps(), &it);
}

/* ---- critical radius, kappa_koehler.hpp:88-166.  The reference evaluates it in double whatever real_t is (its products underflow a
 *      float); the float flavour of this file has no double instance of the root finder, so it does not serve the two options that
 *      need rw3_cr (sstp_cond_act > 1, diag_RH_ge_Sc / diag_rw_ge_rc: orc_create refuses them there) ---- */
typedef struct { real rd3, kappa, T; } rw3cr_ctx;
static inline real rw3_cr_minfun(real rw3, void *vc)
{
  const rw3cr_ctx *c = (const rw3cr_ctx *)vc;
  return kelvin_A(c->T) * (c->rd3 - rw3) * ((c->kappa - 1) * c->rd3 + rw3) + 3 * c->kappa * c->rd3 * rw3 * cbrt(rw3);
}
static inline real rw3_cr(real rd3, real kappa, real T)
{
  rw3cr_ctx c = {rd3, kappa, T};
  const real a = 1e0 * rd3, b = 1e8 * rd3;
  uintmax_t it = 100;
  return orc_toms748(rw3_cr_minfun, &c, a, b, rw3_cr_minfun(a, &c), rw3_cr_minfun(b, &c),
                     orc_eps_tolerance(sizeof(dbl) * 8 / 4), &it);
}

/* critical supersaturation, kappa_koehler.hpp:168-189 */
static inline real S_cr(real rd3, real kappa, real T)
{
  const real rw3 = rw3_cr(rd3, kappa, T);
  return a_w(rw3, rd3, kappa) * klvntrm(cbrt(rw3), T);
}

/* ---- condensation: src/impl/condensation/common/particles_impl_cond_common.ipp:80-338 ---- */
typedef struct {
  real rw2_old, dt, rhod, rv, T, p, RH, eta, rd3, kpa, vt, RH_max, lambda_D, lambda_K;
} cond_ctx;
static inline real drw2_dt(const cond_ctx *c, real rw2)
{
  const real rw = sqrt(rw2);
  const real rw3 = rw * rw * rw;
  const real Re = vent_Re(c->vt, rw, c->rhod, c->eta);
  const real Sc = c->eta / c->rhod / D_0;
  const real Pr = c_pd * c->eta / K_0;
  const real D = D_0 * trans_beta(c->lambda_D / rw) * (vent_Nu(Sc, Re) / 2);
  const real K = K_0 * trans_beta(c->lambda_K / rw) * (vent_Nu(Pr, Re) / 2);
  return 2. * mm_rdrdt(D, K, c->rhod * c->rv, c->T, c->RH > c->RH_max ? c->RH_max : c->RH,
                       a_w(rw3, c->rd3, c->kpa), klvntrm(rw, c->T));
}
static inline real cond_minfun(real rw2, void *vc)
{
  const cond_ctx *c = (const cond_ctx *)vc;
  return c->rw2_old + c->dt * drw2_dt(c, rw2) - rw2;
}
static inline real advance_rw2_apply(cond_ctx *c, real eps, real cond_mlt, uintmax_t n_iter, int apply);
static inline real advance_rw2(cond_ctx *c, real eps, real cond_mlt, uintmax_t n_iter)
{ return advance_rw2_apply(c, eps, cond_mlt, n_iter, 1); }
static inline real advance_rw2_apply(cond_ctx *c, real eps, real cond_mlt, uintmax_t n_iter, int apply)
{
  const real rw2_old = c->rw2_old;
  if (rw2_old <= 0) return rw2_old;
  const real drw2 = c->dt * drw2_dt(c, rw2_old);
  if (drw2 == 0) return apply ? rw2_old : 0.;
  const real rd = cbrt(c->rd3);
  const real rd2 = rd * rd;
  const real a = dmax(rd2, rw2_old + dmin(0., cond_mlt * drw2)),
               b = rw2_old + dmax(0., cond_mlt * drw2);
  if (a == b) return apply ? rw2_old : 0.;
  real fa, fb;
  if (drw2 > 0) { fa = drw2; fb = cond_minfun(b, c); }
  else          { fa = cond_minfun(a, c); fb = drw2; }
  real rw2_new;
  if (fa * fb > 0) rw2_new = rw2_old + drw2;
  else { uintmax_t it = n_iter; rw2_new = orc_toms748(cond_minfun, c, a, b, fa, fb, eps, &it); }
  if (rw2_new < rd2) rw2_new = rd2;
  return apply ? rw2_new : rw2_new - rw2_old;
}

/* ---- terminal velocities: include/libcloudph++/common/vterm.hpp:33-220 ---- */
static inline real vt_khvorostyanov(real r, real T, real rhoa, real eta, int spherical)
{
  (void)T;
  const real X = (32. / 3) * (rho_w - rhoa) / rhoa * g_earth * r * r * r / eta / eta * rhoa * rhoa;
  const real b = (.0902 / 2) * sqrt(X) /
                   ((sqrt(1. + .0902 * sqrt(X)) - 1.) * (sqrt(1. + .0902 * sqrt(X))));
  const real pow_hlpr = sqrt(1. + .0902 * sqrt(X)) - 1.;
  const real a = (9.06 * 9.06 / 4) * pow_hlpr * pow_hlpr / pow(X, b);
  real Av;
  if (spherical)
    Av = a * pow(eta / rhoa * 1e4, 1. - 2. * b) * pow((4. / 3) * rho_w / rhoa * g_earth * 1e2, b);
  else {
    const real lambda_half = 2.35e-3;
    const real ksi = exp(-r / lambda_half) + (1. - exp(-r / lambda_half)) / (1. + r / lambda_half);
    const real alfa = ORC_PI / 6. * rho_w * ksi;
    Av = a * pow(eta / rhoa * 1e4, 1. - 2. * b) * pow(2.546479 * alfa / rhoa * g_earth * 1e2, b);
  }
  const real Bv = 3. * b - 1.;
  return (Av * pow((2 * 1e2) * r, Bv)) / 1e2;
}
static inline real vt_beard77_v0(real r)
{
  const real m_s[4] = {0.105035e2, 0.108750e1, -0.133245, -0.659969e-2};
  const real m_l[8] = {0.65639e1, -0.10391e1, -0.14001e1, -0.82736e0, -0.34277e0, -0.83072e-1, -0.10583e-1, -0.54208e-3};
  const real x = log(2 * 100 * r);
  real y = 0;
  if (r <= 20e-6) for (int i = 0; i < 4; ++i) y += m_s[i] * pow(x, (real)i);
  else            for (int i = 0; i < 8; ++i) y += m_l[i] * pow(x, (real)i);
  return exp(y) / 100.;
}
static inline real vt_beard77_fact(real r, real p, real rhoa, real eta)
{
  const real eta_0 = 1.818e-5;
  if (r <= 20e-6) {
    const real l_0 = 6.62e-8;
    const real l = l_0 * (eta / eta_0) * sqrt(p_stp / p * rho_stp / rhoa);
    return (eta_0 / eta) * (1 + 1.255 * (l / r)) / (1 + 1.255 * (l_0 / r));
  } else {
    const real eps_s = (eta_0 / eta) - 1;
    const real eps_c = sqrt(rho_stp / rhoa) - 1;
    return 1.104 * eps_s + ((1.058 * eps_c - 1.104 * eps_s) * (5.52 + log(2 * 100 * r)) / 5.01) + 1;
  }
}
static inline real vt_beard76(real r, real T, real p, real rhoa, real eta)
{
  if (r <= 9.5e-6) {
    const real l = 6.62e-8 * (eta / 1.818e-5) * (p_stp / p) * sqrt(T / 293.15);
    const real C_ac = 1. + 1.255 * l / r;
    return (rho_w - rhoa) * g_earth / (4.5 * eta) * C_ac * r * r;
  } else if (r <= 5.035e-4) {
    const real b[7] = {-0.318657e1, 0.992696, -0.153193e-2, -0.987059e-3, -0.578878e-3, 0.855176e-4, -0.327815e-5};
    const real l = 6.62e-8 * (eta / 1.818e-5) * (p_stp / p) * sqrt(T / 293.15);
    const real C_ac = 1. + 1.255 * l / r;
    const real log_N_Da = log((32. / 3.) * r * r * r * rhoa * (rho_w - rhoa) * g_earth / eta / eta);
    real Y = 0.;
    for (int i = 0; i < 7; ++i) Y = Y + b[i] * pow(log_N_Da, (real)i);
    const real N_Re = C_ac * exp(Y);
    return eta * N_Re / rhoa / 2. / r;
  } else {
    const real b[6] = {-0.500015e1, 0.523778e1, -0.204914e1, 0.475294, -0.542819e-1, 0.238449e-2};
    const real sg = sg_surf(T);
    const real Bo = (16. / 3.) * r * r * (rho_w - rhoa) * g_earth / sg;
    const real N_p = sg * sg * sg * rhoa * rhoa / eta / eta / eta / eta / g_earth / (rho_w - rhoa);
    const real X = log(Bo * pow(N_p, 1. / 6.));
    real Y = 0.;
    for (int i = 0; i < 6; ++i) Y = Y + b[i] * pow(X, (real)i);
    const real N_Re = pow(N_p, 1. / 6.) * exp(Y);
    return eta * N_Re / rhoa / 2. / r;
  }
}
#endif
