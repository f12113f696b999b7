/*
 * lcx_oracle.c -- TEST INFRASTRUCTURE.  CPU oracle for the lgrngn hot path.
 *
 * A serial, plain-C restatement of what the reference's serial backend
 * (thrust::cpp) does on particles_t::init / step_sync / step_async / diag_*, following the
 * reference's control flow and evaluation order file by file (citations at each function,
 * paths relative to the reference checkout).  It exports the same entry points as the product
 * C ABI (include/lcx.h) with the prefix orc_ so that one test harness can drive both.
 *
 * PINNING.  The reference itself cannot be built in this image (its headers need Boost.units /
 * Boost.math, which are absent, and stand-ins are not allowed), so this restatement is pinned
 * against the reference's own known answers and golden data instead (tests/test_oracle_pins.py):
 *   - tests/python/physics/refdata/lgrngn_cond_substepping_refdata.csv (56 per-cell-substepping rows)
 *   - tests/python/physics/lgrngn_cond.py:52-56,131-132,152-187 (th / rv known answers, sstp scaling)
 *   - tests/python/physics/puddle.py:74-81 (precipitation totals), test_coal.py:95-101 (conservation)
 *   - tests/python/physics/coalescence_golovin.py:147-153 (analytic Golovin RMSD)
 *   - tests/python/unit/lgrngn_adve.py:97-105, tests/common/test_common_pvs.cpp:6-8, tests/toms748
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * It is never on the product path.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <stdint.h>
#include "../include/lcx.h"
#include "orc_physics.h"

typedef unsigned long long n_t;       /* impl::n_t, src/impl/particles_impl.ipp:29 */
typedef size_t sz;                    /* thrust_size_t, src/detail/thrust.hpp:11 */

static __thread char orc_err[512];
#define FAIL(...) do { snprintf(orc_err, sizeof orc_err, __VA_ARGS__); return 1; } while (0)

/* ---------------- RNG: std::mt19937 + libstdc++ distributions (src/detail/urand.hpp:24-86) ------------- */
typedef struct { uint32_t mt[624]; int idx; } mt19937_t;
static void mt_seed(mt19937_t *g, uint32_t s)
{
  g->mt[0] = s;
  for (int i = 1; i < 624; ++i) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
  g->idx = 624;
}
static uint32_t mt_next(mt19937_t *g)
{
  if (g->idx >= 624) {
    for (int i = 0; i < 624; ++i) {
      uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
      g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    g->idx = 0;
  }
  uint32_t y = g->mt[g->idx++];
  y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
  return y;
}
/* uniform_real_distribution<real>(0,1) == generate_canonical<real,53>: two 32-bit draws */
static dbl rng_u01_dbl(mt19937_t *g)
{
  const dbl x0 = (dbl)mt_next(g);
  const dbl x1 = (dbl)mt_next(g);
  dbl r = (x0 + x1 * (dbl)4294967296.0) / (dbl)18446744073709551616.0;
  if (r >= (dbl)1.0) r = nextafter((dbl)1.0, (dbl)0.0);
  return r;
}
/* (the float flavour takes the double draw -- the SAME stream as the double flavour, so that one replay serves both -- and keeps the
 * distribution's promise u < 1 in its own type; the reference's float run draws generate_canonical<float, 24> from one 32-bit word
 * instead: no reference-held data pins either) */
static real rng_u01(mt19937_t *g)
{
  real r = (real)rng_u01_dbl(g);
  if (r >= 1) r = nextafter((real)1, (real)0);
  return r;
}
/* uniform_int_distribution<unsigned>(0,UINT_MAX): one draw; fnctr_un returns it through real_t */
static dbl rng_un_dbl(mt19937_t *g) { return (dbl)mt_next(g); }      /* (through double in every flavour: a float would lose eight of the 32 bits) */

/* std::normal_distribution<real>(0,1) of libstdc++ (bits/random.tcc): Marsaglia polar method; every second call returns
 * the value saved by the previous one.  The distribution object is a member of the reference's rng (urand.hpp:30,43), so
 * the saved value survives from one generate_normal_n call to the next. */
typedef struct { int saved_available; dbl saved; } normal_state;
static real rng_normal(mt19937_t *g, normal_state *ns)
{
  if (ns->saved_available) { ns->saved_available = 0; return (real)ns->saved; }
  dbl x, y, r2;
  do {
    x = (dbl)2.0 * rng_u01_dbl(g) - (dbl)1.0;
    y = (dbl)2.0 * rng_u01_dbl(g) - (dbl)1.0;
    r2 = x * x + y * y;
  } while (r2 > (dbl)1.0 || r2 == (dbl)0.0);
  const dbl mult = sqrt(-2 * log(r2) / r2);
  ns->saved = x * mult;
  ns->saved_available = 1;
  return (real)(y * mult);
}

/* ---------------- state (src/impl/particles_impl.ipp:26-325) ---------------- */
typedef struct { real *q; sz len, pos; } fifo_arr;

struct orc_particles {
  lcx_opts_init_t o;
  lcx_distro_t *distros; lcx_dry_size_t *sizes;
  real *kernel_parameters; sz n_kernel_parameters; real kernel_r_max; int n_user_params; int n_size_keys;
  real *w_LS, *aerosol_conc_factor;
  int n_dims; sz n_cell, n_part, n_part_old, n_part_to_init, cap;
  int init_called, should_now_run_async, should_now_run_cond, selected_before_counting, var_rho, sorted;
  int sstp_cond, sstp_coal, allow_sstp_cond, pure_const_multi, increase_sstp_coal;
  real dt; int adve_scheme; int halo;      /* halo: x-planes of Courant halo on each side (2 with pred_corr, particles_impl.ipp:361) */
  mt19937_t rng; normal_state rng_ns;
  /* SGS turbulence (turb_adve / turb_cond): cell field diss_rate (holds TKE after hskpng_tke), SGS mixing length profile,
   * per-particle velocity perturbations and supersaturation perturbation (particles_impl.ipp:141-144,461-473) */
  real *diss_rate, *SGS_mix_len, *tau_cell;
  /* test hooks of the "reverse replay" (tests/test_hip_reverse_replay.py; no reference counterpart): LCX_DBG_TAG gives every super-droplet
   * a persistent tag that is compacted and migrates with it like any attribute, and orc_rng_replay_push queues random arrays that the
   * next hskpng_shuffle_and_sort / coal consume INSTEAD of drawing from the engine -- so that this restatement can be run on the
   * device generator's stream while the device stays on its production path */
  real *tag;
  struct { int kind; dbl *v; sz n; } rq[64]; int rq_head, rq_tail;      /* (dbl: the un of a shuffle are 32-bit integers passed as doubles) */
  real *ict;     /* opts_init.diag_incloud_time: time each SD has been activated (particles_impl.ipp:93,475-476) */
  real *up, *vp, *wp, *ssp, *dot_ssp;
  /* particle attributes */
  n_t *n; real *rd3, *rw2, *kpa, *x, *y, *z, *vt;
  sz *ijk, *sorted_id, *sorted_ijk;
  real *n_filtered, *tmp_part, *col, *mom_vals;
  /* cell fields */
  real *rhod, *th, *rv, *p, *T, *RH, *eta, *dv, *lambda_D, *lambda_K;
  real *sstp_tmp_rv, *sstp_tmp_th, *sstp_tmp_rh, *drw_mom3, *rw_mom3, *scl;
  /* per-particle substepping (exact_sstp_cond): the reference re-uses sstp_tmp_{rv,th,rh,p} as n_part-long attributes
   * (particles_impl.ipp:452-459); rc2 only with sstp_cond_act > 1 (:488-491) */
  int exact, use_rc2, sstp_cond_act;
  real *pp_rv, *pp_th, *pp_rh, *pp_p, *rc2;
  real *dlt_rv, *dlt_th, *dlt_rh, *dlt_p, *rwX, *drwX, *Tp; unsigned *pp_sstp;
  real *courant_x, *courant_y, *courant_z; sz n_cx, n_cy, n_cz;
  sz *count_ijk, *off; n_t *count_num; real *count_mom; sz count_n;
  real vt_0[10000]; real vt0_ln_r_min, vt0_ln_r_max;
  real log_rd_min, log_rd_max, multiplier;
  real puddle[LCX_OUT_COUNT];
  real *outbuf;
  real eps_tol;
  /* distmem */
  sz *lft_id, *rgt_id; sz lft_count, rgt_count;
  /* message buffers of the device-driven exchange protocol (orc_exch_*, the CPU twin of include/lcx.h lcx_exch_*) */
  unsigned char *xbox[4]; sz xcap; unsigned xflags;
};
typedef struct orc_particles orc_particles;

static int m1(int n) { return n == 0 ? 1 : n; }
#define NEW(T, n) ((T *)calloc((n) ? (n) : 1, sizeof(T)))
/* Elementwise loops carry this marker; it only has an effect in the -fopenmp build (liblcx_oracle_omp.so), which
 * exists for bench.py's cpu_baseline leg (the counterpart of the reference's OpenMP backend, thrust::omp transforms).
 * Reductions and scans stay serial so that every build sums in the same order. */
#define OMP_FOR _Pragma("omp parallel for schedule(static)")

const char *orc_last_error(void) { return orc_err; }
/* see orc_physics.h: the root finders iterate to the tolerance of a real_t of that many bytes from now on */
void orc_set_real_bytes(int bytes) { orc_real_bytes_v = bytes == 4 ? 4u : 8u; }
#ifdef _OPENMP
#include <omp.h>
const char *orc_version(void) { return "lcx-oracle 1 (OpenMP elementwise loops)"; }
int orc_num_threads(void) { return omp_get_max_threads(); }
void orc_set_num_threads(int n) { if (n > 0) omp_set_num_threads(n); }
#else
void orc_set_num_threads(int n) { (void)n; }
const char *orc_version(void) { return "lcx-oracle 1 (serial)"; }
int orc_num_threads(void) { return 1; }
#endif
/* Stage timers of the oracle itself (ORC_TIMERS=1 in the environment; orc_timers_dump prints and clears them): used once to see
 * which stages of the OpenMP build were still serial (bench.py's cpu_baseline leg). */
#include <time.h>
enum { TM_SORT, TM_COND, TM_MOMS, TM_TPR, TM_VTERM, TM_COAL, TM_MOVE, TM_POST, TM_OTHER, TM_N };
static real tm_acc[TM_N];
static int tm_on = -1;
static real tm_now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
#define TMR(k, stmt) do { if (tm_on < 0) tm_on = getenv("ORC_TIMERS") != NULL; if (tm_on) { const real t0_ = tm_now(); stmt; tm_acc[k] += tm_now() - t0_; } else { stmt; } } while (0)
void orc_timers_dump(void)
{
  static const char *nm[TM_N] = {"sort", "cond", "th_rv", "Tpr", "vterm", "coal", "adve+sedi+bcnd", "post_copy", "other"};
  for (int k = 0; k < TM_N; ++k) { fprintf(stderr, "orc timer %-16s %9.3f s\n", nm[k], tm_acc[k]); tm_acc[k] = 0; }
}

void orc_opts_init_default(lcx_opts_init_t *o)
{                                           /* opts_init.hpp:186-247 */
  memset(o, 0, sizeof *o);
  o->dx = o->dy = o->dz = 1; o->x1 = o->y1 = o->z1 = 1;
  o->sstp_cond = o->sstp_coal = o->sstp_chem = o->sstp_cond_act = 1;
  o->sedi_switch = 1; o->coal_switch = 1; o->sstp_cond_mix = 1;
  o->RH_max = .95; o->rng_seed = 44; o->rng_seed_init = 44;
  o->sstp_cond_adapt_drw2_eps = 1e-4; o->sstp_cond_adapt_drw2_max = 4; o->rc2_T = 10;
  o->adve_scheme = LCX_ADVE_IMPLICIT; o->RH_formula = LCX_RH_PV_CC;
  o->dev_id = -1; o->rd_min = -1; o->rd_max = -1; o->th_dry = 1; o->strict_fp = 0; o->cond_solver = 1;     /* (the header's default since round 5; the oracle computes in ONE arithmetic whatever these say) */
}
void orc_opts_default(lcx_opts_t *o)
{                                           /* opts.hpp:41-48 */
  memset(o, 0, sizeof *o);
  o->adve = o->sedi = o->cond = o->coal = 1; o->RH_max = 44; o->dt = -1;
}

static int distmem(const orc_particles *s) { return s->o.bcond_lft == 1 || s->o.bcond_rgt == 1; }

int orc_create(const lcx_opts_init_t *oi, int real_kind, orc_particles **out)
{
  if (real_kind != (int)sizeof(real)) FAIL("oracle: this flavour computes in a real_t of %d bytes (liblcx_oracle.so: double, liblcx_oracle_f32.so: float)", (int)sizeof(real));
  if (oi->chem_switch || oi->ice_switch || oi->rlx_switch || oi->src_type)
    FAIL("libcloudph++: option outside the accelerated hot path (chem/ice/src/rlx)");
  if (sizeof(real) == 4 && oi->sstp_cond_act > 1) FAIL("oracle (float flavour): sstp_cond_act > 1 needs the critical radius in double (orc_physics.h)");
  orc_particles *s = NEW(orc_particles, 1);
  s->o = *oi;
  s->distros = NEW(lcx_distro_t, oi->n_dry_distros);
  memcpy(s->distros, oi->dry_distros, sizeof(lcx_distro_t) * oi->n_dry_distros);
  s->sizes = NEW(lcx_dry_size_t, oi->n_dry_sizes);
  if (oi->n_dry_sizes) memcpy(s->sizes, oi->dry_sizes, sizeof(lcx_dry_size_t) * oi->n_dry_sizes);
  s->n_user_params = oi->n_kernel_parameters;
  for (int d = 0; d < oi->n_dry_sizes; ++d)        /* dry_sizes.size() of the reference = number of (kappa, rd_insol) keys */
    if (d == 0 || oi->dry_sizes[d].kappa != oi->dry_sizes[d - 1].kappa || oi->dry_sizes[d].rd_insol != oi->dry_sizes[d - 1].rd_insol) s->n_size_keys++;
  s->n_kernel_parameters = oi->n_kernel_parameters;
  s->kernel_parameters = NEW(real, oi->n_kernel_parameters);
  for (int i = 0; i < oi->n_kernel_parameters; ++i) s->kernel_parameters[i] = (real)oi->kernel_parameters[i];
  s->w_LS = NEW(real, oi->n_w_LS);
  for (int i = 0; i < oi->n_w_LS; ++i) s->w_LS[i] = (real)oi->w_LS[i];
  s->aerosol_conc_factor = NEW(real, oi->n_aerosol_conc_factor);
  for (int i = 0; i < oi->n_aerosol_conc_factor; ++i) s->aerosol_conc_factor[i] = (real)oi->aerosol_conc_factor[i];
  /* particles_impl.ipp:327-345 */
  s->n_dims = oi->nx / m1(oi->nx) + oi->ny / m1(oi->ny) + oi->nz / m1(oi->nz);
  s->n_cell = (sz)m1(oi->nx) * m1(oi->ny) * m1(oi->nz);
  s->sstp_cond = oi->sstp_cond; s->sstp_coal = oi->sstp_coal;
  s->allow_sstp_cond = oi->sstp_cond > 1 || oi->sstp_cond_act > 1;
  s->sstp_cond_act = oi->sstp_cond_act;
  s->exact = s->allow_sstp_cond && oi->exact_sstp_cond;
  s->use_rc2 = oi->sstp_cond_act > 1 && s->allow_sstp_cond;
  s->pure_const_multi = (oi->sd_conc == 0) && (oi->sd_const_multi > 0 || oi->n_dry_sizes > 0);
  s->adve_scheme = oi->adve_scheme;
  s->halo = oi->adve_scheme == LCX_ADVE_PRED_CORR ? 2 : 0;
  if (s->o.n_x_tot == 0) s->o.n_x_tot = oi->nx;
  mt_seed(&s->rng, (uint32_t)oi->rng_seed);
  s->eps_tol = orc_real_eps();                               /* src/detail/config.hpp:39: eps_tolerance(sizeof(real_t) * 8 / 4) */
  s->vt0_ln_r_min = log(5e-7); s->vt0_ln_r_max = log(3e-3); /* config.hpp:36-38 */
  s->cap = (sz)oi->n_sd_max;
  sz c = s->cap, nc = s->n_cell;
  s->n = NEW(n_t, c); s->rd3 = NEW(real, c); s->rw2 = NEW(real, c); s->kpa = NEW(real, c);
  s->x = NEW(real, c); s->y = NEW(real, c); s->z = NEW(real, c); s->vt = NEW(real, c);
  s->ijk = NEW(sz, c); s->sorted_id = NEW(sz, c); s->sorted_ijk = NEW(sz, c);
  s->n_filtered = NEW(real, c); s->tmp_part = NEW(real, c); s->col = NEW(real, c); s->mom_vals = NEW(real, c);
  s->lft_id = NEW(sz, c); s->rgt_id = NEW(sz, c);
  s->rhod = NEW(real, nc); s->th = NEW(real, nc); s->rv = NEW(real, nc); s->p = NEW(real, nc);
  s->T = NEW(real, nc); s->RH = NEW(real, nc); s->eta = NEW(real, nc); s->dv = NEW(real, nc);
  s->lambda_D = NEW(real, nc); s->lambda_K = NEW(real, nc);
  s->sstp_tmp_rv = NEW(real, nc); s->sstp_tmp_th = NEW(real, nc); s->sstp_tmp_rh = NEW(real, nc);
  s->drw_mom3 = NEW(real, nc); s->rw_mom3 = NEW(real, nc); s->scl = NEW(real, nc);
  if (s->exact) {
    s->pp_rv = NEW(real, c); s->pp_th = NEW(real, c); s->pp_rh = NEW(real, c); s->pp_p = NEW(real, c);
    s->dlt_rv = NEW(real, c); s->dlt_th = NEW(real, c); s->dlt_rh = NEW(real, c); s->dlt_p = NEW(real, c);
    s->rwX = NEW(real, c); s->drwX = NEW(real, c); s->Tp = NEW(real, c); s->pp_sstp = NEW(unsigned, c);
  }
  if (s->use_rc2) s->rc2 = NEW(real, c);
  if (oi->diag_incloud_time) s->ict = NEW(real, c);                    /* init_incloud_time.ipp:14-17: zero */
  if (oi->dbg_flags & LCX_DBG_TAG) s->tag = NEW(real, c);
  if (oi->turb_coal_switch && !(oi->turb_adve_switch || oi->turb_cond_switch)) s->diss_rate = NEW(real, nc);
  if (oi->turb_adve_switch || oi->turb_cond_switch) {
    s->diss_rate = NEW(real, nc); s->tau_cell = NEW(real, nc);
    s->SGS_mix_len = NEW(real, oi->n_SGS_mix_len);
    for (int i = 0; i < oi->n_SGS_mix_len; ++i) s->SGS_mix_len[i] = (real)oi->SGS_mix_len[i];
    s->up = NEW(real, c); s->vp = NEW(real, c); s->wp = NEW(real, c);       /* resized with the initial value 0 */
    if (oi->turb_cond_switch) { s->ssp = NEW(real, c); s->dot_ssp = NEW(real, c); }
  }
  s->count_ijk = NEW(sz, nc); s->off = NEW(sz, nc + 1); s->count_num = NEW(n_t, nc); s->count_mom = NEW(real, nc);
  s->outbuf = NEW(real, nc);
  *out = s;
  return 0;
}
void orc_destroy(orc_particles *s)
{
  if (!s) return;
  void *ptrs[] = {s->distros, s->sizes, s->kernel_parameters, s->w_LS, s->aerosol_conc_factor, s->n, s->rd3, s->rw2,
    s->kpa, s->x, s->y, s->z, s->vt, s->ijk, s->sorted_id, s->sorted_ijk, s->n_filtered, s->tmp_part, s->col, s->mom_vals,
    s->diss_rate, s->SGS_mix_len, s->tau_cell, s->up, s->vp, s->wp, s->ssp, s->dot_ssp, s->ict, s->tag,
    s->pp_rv, s->pp_th, s->pp_rh, s->pp_p, s->rc2, s->dlt_rv, s->dlt_th, s->dlt_rh, s->dlt_p, s->rwX, s->drwX, s->Tp, s->pp_sstp,
    s->lft_id, s->rgt_id, s->rhod, s->th, s->rv, s->p, s->T, s->RH, s->eta, s->dv, s->lambda_D, s->lambda_K,
    s->sstp_tmp_rv, s->sstp_tmp_th, s->sstp_tmp_rh, s->drw_mom3, s->rw_mom3, s->scl, s->count_ijk, s->off,
    s->count_num, s->count_mom, s->outbuf, s->courant_x, s->courant_y, s->courant_z, s->xbox[0], s->xbox[1], s->xbox[2], s->xbox[3]};
  for (sz i = 0; i < sizeof ptrs / sizeof *ptrs; ++i) free(ptrs[i]);
  while (s->rq_head != s->rq_tail) { free(s->rq[s->rq_head].v); s->rq_head = (s->rq_head + 1) % 64; }
  free(s);
}

/* ---------------- Eulerian <-> Lagrangian sync (particles_impl_sync.ipp:15-68, init_e2l.ipp:34-114) ----- */
/* device index c (z fastest) of a field with extents (nx+ex, ny+ey, nz+ez) -> element offset in the user's array */
/* halo > 0: the device array starts `halo` x-planes left of the user's (Courant numbers with pred_corr); planes outside the
 * user's array wrap around it cyclically (init_e2l.ipp:44-46,109-113: periodic_cellno over the whole array of
 * n_x_tot + ext_x planes), across MPI / device boundaries they are overwritten by the halo exchange afterwards */
static ptrdiff_t l2e_halo(const orc_particles *s, const lcx_arrinfo_t *a, sz c, int ex, int ey, int ez, int halo)
{
  const int ny = s->o.ny + ey, nz = s->o.nz + ez;
  const ptrdiff_t n_planes = (ptrdiff_t)s->o.n_x_tot + ex;
  ptrdiff_t i;
  switch (s->n_dims) {
    case 0: return 0;
    case 1: i = (ptrdiff_t)c; break;
    case 2: i = (ptrdiff_t)(c / nz); break;
    default: i = (ptrdiff_t)(c / ((sz)nz * ny));
  }
  i += s->o.n_x_bfr - halo;
  if (halo) { if (i >= n_planes) i -= n_planes; else if (i < 0) i += n_planes; }
  switch (s->n_dims) {
    case 1: return i;
    case 2: return a->strides[0] * i + a->strides[1] * (ptrdiff_t)(c % nz);
    default: return a->strides[0] * i + a->strides[1] * (ptrdiff_t)((c / nz) % ny) + a->strides[2] * (ptrdiff_t)(c % nz);
  }
}
static ptrdiff_t l2e(const orc_particles *s, const lcx_arrinfo_t *a, sz c, int ex, int ey, int ez) { return l2e_halo(s, a, c, ex, ey, ez, 0); }
static int arr_null(const lcx_arrinfo_t *a) { return !a || !a->data || !a->strides; }
static void sync_in_arr(const orc_particles *s, const lcx_arrinfo_t *a, real *to, sz n, int ex, int ey, int ez)
{
  if (arr_null(a)) return;
  const real *d = (const real *)a->data;
  for (sz c = 0; c < n; ++c) to[c] = d[l2e(s, a, c, ex, ey, ez)];
}
static void sync_in_courant(const orc_particles *s, const lcx_arrinfo_t *a, real *to, sz n, int ex, int ey, int ez)
{
  if (arr_null(a)) return;
  const real *d = (const real *)a->data;
  for (sz c = 0; c < n; ++c) to[c] = d[l2e_halo(s, a, c, ex, ey, ez, s->halo)];
}
static void sync_out_arr(const orc_particles *s, const real *from, const lcx_arrinfo_t *a, sz n)
{
  if (arr_null(a)) return;
  real *d = (real *)a->data;
  for (sz c = 0; c < n; ++c) d[l2e(s, a, c, 0, 0, 0)] = from[c];
}

/* ---------------- housekeeping ---------------- */
/* hskpng_Tpr.ipp:219-305 */
static void hskpng_Tpr(orc_particles *s)
{
  OMP_FOR
  for (sz c = 0; c < s->n_cell; ++c) {
    if (s->o.th_dry) s->T[c] = theta_dry_T(s->th[c], s->rhod[c]);
    else             s->T[c] = s->th[c] * theta_std_exner(s->p[c]);
    if (!s->o.const_p) s->p[c] = theta_dry_p(s->rhod[c], s->rv[c], s->T[c]);
    s->RH[c] = RH_of(s->o.RH_formula, s->p[c], s->rv[c], s->T[c]);
    s->eta[c] = visc(s->T[c]);
    if (s->n_dims == 0) s->dv[c] = 1. / s->rhod[c];
  }
}
/* hskpng_mfp.ipp:42-51 */
static void hskpng_mfp(orc_particles *s)
{
  OMP_FOR
  for (sz c = 0; c < s->n_cell; ++c) {
    s->lambda_D[c] = lambda_D_of(s->T[c]);
    s->lambda_K[c] = lambda_K_of(s->T[c], s->p[c]);
  }
}
/* hskpng_ijk.ipp:159-200, :33-82 : size_t(real(x)/real(dx)), z fastest */
static void hskpng_ijk(orc_particles *s)
{
  const lcx_opts_init_t *o = &s->o;
  OMP_FOR
  for (sz p = 0; p < s->n_part; ++p) {
    sz i = o->nx ? (sz)(s->x[p] / o->dx) : 0, j = o->ny ? (sz)(s->y[p] / o->dy) : 0, k = o->nz ? (sz)(s->z[p] / o->dz) : 0;
    switch (s->n_dims) {
      case 0: break;
      case 1: s->ijk[p] = i; break;
      case 2: s->ijk[p] = i * o->nz + k; break;
      default: s->ijk[p] = i * ((sz)o->nz * o->ny) + j * o->nz + k;
    }
  }
  s->sorted = 0;
}
#ifdef _OPENMP
/* ---- The OpenMP build's counterparts of the reference backend's parallel primitives (thrust::omp sort_by_key, reduce_by_key,
 * copy_if / remove_if): every one of them returns what the serial loop beside it returns, bit for bit (a stable sort has one answer;
 * a run of equal keys is summed by ONE thread in the serial order; selections keep the index order), which
 * tests/test_oracle_pins.py checks against the serial build.  Only bench.py's cpu_baseline leg and the production-size parity tests
 * load this build. */
/* Work arrays of these primitives: kept between calls (fresh gigabyte allocations are page-faulted in on every call, and 128 threads
 * faulting at once serialise in the kernel).  One set per process: the OpenMP build is driven by one host thread at a time. */
enum { SCR_K, SCR_V, SCR_START, SCR_FLAG, SCR_IDX, SCR_TMP, SCR_UN, SCR_N };
static void *scr_buf[SCR_N];
static size_t scr_cap[SCR_N];
static void *scr_get(int slot, size_t bytes)
{
  if (scr_cap[slot] < bytes) { free(scr_buf[slot]); scr_cap[slot] = bytes + bytes / 8 + 64; scr_buf[slot] = malloc(scr_cap[slot]); }
  return scr_buf[slot];
}
static void omp_chunk(sz n, int t, int T, sz *lo, sz *hi)
{
  const sz per = (n + (sz)T - 1) / (sz)T;
  *lo = per * (sz)t > n ? n : per * (sz)t;
  *hi = *lo + per > n ? n : *lo + per;
}
/* one stable pass of an LSD radix sort on the digit (key >> sh) & mask: per-thread histograms over contiguous chunks */
static void par_sort_pass(const sz *key, const sz *val, sz *k2, sz *v2, sz n, int sh, sz mask)
{
  const sz nb = mask + 1;
  const int Tmax = omp_get_max_threads();
  const sz n_hist = (sz)Tmax * nb;
  sz *hist = NEW(sz, n_hist);
#pragma omp parallel
  {
    const int t = omp_get_thread_num(), T = omp_get_num_threads();
    sz lo, hi; omp_chunk(n, t, T, &lo, &hi);
    sz *h = hist + (sz)t * nb;
    for (sz i = lo; i < hi; ++i) h[(key[i] >> sh) & mask]++;
#pragma omp barrier
#pragma omp single
    { sz acc = 0; for (sz b = 0; b < nb; ++b) for (int u = 0; u < T; ++u) { const sz c = hist[(sz)u * nb + b]; hist[(sz)u * nb + b] = acc; acc += c; } }
    for (sz i = lo; i < hi; ++i) { const sz d = h[(key[i] >> sh) & mask]++; k2[d] = key[i]; v2[d] = val[i]; }
  }
  free(hist);
}
/* indices p (ascending) with flag[p] != 0 into idx; returns their number */
static sz par_select(const unsigned char *flag, sz n, sz *idx)
{
  const int Tmax = omp_get_max_threads();
  sz *cnt = NEW(sz, (sz)Tmax + 1);
#pragma omp parallel
  {
    const int t = omp_get_thread_num(), T = omp_get_num_threads();
    sz lo, hi; omp_chunk(n, t, T, &lo, &hi);
    sz c = 0;
    for (sz i = lo; i < hi; ++i) c += flag[i] != 0;
    cnt[t + 1] = c;
#pragma omp barrier
#pragma omp single
    for (int u = 0; u < T; ++u) cnt[u + 1] += cnt[u];
    sz w = cnt[t];
    for (sz i = lo; i < hi; ++i) if (flag[i]) idx[w++] = i;
  }
  sz total = 0;                                      /* cnt[T] of the team that ran (entries behind it stay 0) */
  for (int u = 0; u <= Tmax; ++u) if (cnt[u] > total) total = cnt[u];
  free(cnt);
  return total;
}
/* first index of every run of equal keys (key sorted) into start[0 .. runs], start[runs] = n; returns the number of runs */
static sz par_runs(const sz *key, sz n, sz *start)
{
  unsigned char *head = (unsigned char *)scr_get(SCR_FLAG, n);
  OMP_FOR
  for (sz p = 0; p < n; ++p) head[p] = p == 0 || key[p] != key[p - 1];
  const sz runs = par_select(head, n, start);
  start[runs] = n;
  return runs;
}
#endif
/* stable sort of (key,val) pairs by key; keys < nkeys when nkeys>0 (counting sort) else 32-bit LSD radix */
static void stable_sort_by_key(sz *key, sz *val, sz n, sz nkeys)
{
#ifdef _OPENMP
  {                                                  /* LSD radix, 11 bits per pass, over the bits the keys can have */
    sz *ka = key, *va = val, *kb = (sz *)scr_get(SCR_K, n * sizeof(sz)), *vb = (sz *)scr_get(SCR_V, n * sizeof(sz));
    int bits = 32;
    if (nkeys) { bits = 1; while (((sz)1 << bits) < nkeys) ++bits; }
    for (int sh = 0; sh < bits; sh += 11) {
      const int w = bits - sh < 11 ? bits - sh : 11;
      par_sort_pass(ka, va, kb, vb, n, sh, ((sz)1 << w) - 1);
      sz *t = ka; ka = kb; kb = t; t = va; va = vb; vb = t;
    }
    if (ka != key) {
      OMP_FOR
      for (sz i = 0; i < n; ++i) { key[i] = ka[i]; val[i] = va[i]; }
    }
    return;
  }
#endif
  sz *k2 = NEW(sz, n), *v2 = NEW(sz, n);
  if (nkeys) {
    sz *cnt = NEW(sz, nkeys + 1);
    for (sz i = 0; i < n; ++i) cnt[key[i] + 1]++;
    for (sz c = 0; c < nkeys; ++c) cnt[c + 1] += cnt[c];
    for (sz i = 0; i < n; ++i) { sz d = cnt[key[i]]++; k2[d] = key[i]; v2[d] = val[i]; }
    memcpy(key, k2, n * sizeof(sz)); memcpy(val, v2, n * sizeof(sz));
    free(cnt);
  } else {
    for (int pass = 0; pass < 2; ++pass) {
      sz *cnt = NEW(sz, 65537);
      const int sh = 16 * pass;
      for (sz i = 0; i < n; ++i) cnt[((key[i] >> sh) & 0xffff) + 1]++;
      for (sz c = 0; c < 65536; ++c) cnt[c + 1] += cnt[c];
      for (sz i = 0; i < n; ++i) { sz d = cnt[(key[i] >> sh) & 0xffff]++; k2[d] = key[i]; v2[d] = val[i]; }
      memcpy(key, k2, n * sizeof(sz)); memcpy(val, v2, n * sizeof(sz));
      free(cnt);
    }
  }
  free(k2); free(v2);
}
/* next queued random array of the given kind (orc_rng_replay_push), or NULL when the queue is empty: the engine draws then */
static dbl *replay_pop(orc_particles *s, int kind, sz n)
{
  if (s->rq_head == s->rq_tail) return NULL;
  if (s->rq[s->rq_head].kind != kind || s->rq[s->rq_head].n < n) { fprintf(stderr, "oracle: rng replay queue does not match the request (kind %d, %zu values)\n", kind, (size_t)n); abort(); }
  dbl *v = s->rq[s->rq_head].v;
  s->rq_head = (s->rq_head + 1) % 64;
  return v;
}
/* hskpng_sort.ipp:15-57 */
static void hskpng_sort_helper(orc_particles *s, int shuffle)
{
  const sz n = s->n_part;
  OMP_FOR
  for (sz p = 0; p < n; ++p) s->sorted_id[p] = p;
  if (!shuffle) memcpy(s->sorted_ijk, s->ijk, n * sizeof(sz));
  else {
#ifdef _OPENMP
    sz *un = (sz *)scr_get(SCR_UN, n * sizeof(sz));
#else
    sz *un = NEW(sz, n);
#endif
    dbl *rq = replay_pop(s, 1, n);
    if (rq) { for (sz p = 0; p < n; ++p) un[p] = (sz)(unsigned int)rq[p]; free(rq); }
    else for (sz p = 0; p < n; ++p) un[p] = (sz)(unsigned int)rng_un_dbl(&s->rng);
    stable_sort_by_key(un, s->sorted_id, n, 0);
    OMP_FOR
    for (sz p = 0; p < n; ++p) s->sorted_ijk[p] = s->ijk[s->sorted_id[p]];
#ifndef _OPENMP
    free(un);
#endif
  }
  stable_sort_by_key(s->sorted_ijk, s->sorted_id, n, s->n_cell);
  s->sorted = 1;
}
static void hskpng_sort(orc_particles *s) { if (!s->sorted) hskpng_sort_helper(s, 0); }
/* hskpng_count.ipp:16-48: cells in use and their populations from the sorted cell numbers */
static void count_runs(orc_particles *s)
{
#ifdef _OPENMP
  {
    sz *start = (sz *)scr_get(SCR_START, (s->n_part + 1) * sizeof(sz));
    const sz runs = par_runs(s->sorted_ijk, s->n_part, start);
    OMP_FOR
    for (sz i = 0; i < runs; ++i) { s->count_ijk[i] = s->sorted_ijk[start[i]]; s->count_num[i] = (n_t)(start[i + 1] - start[i]); }
    s->count_n = runs;
    return;
  }
#endif
  sz cn = 0;
  for (sz p = 0; p < s->n_part; ++p) {
    if (p == 0 || s->sorted_ijk[p] != s->sorted_ijk[p - 1]) { s->count_ijk[cn] = s->sorted_ijk[p]; s->count_num[cn] = 0; ++cn; }
    s->count_num[cn - 1] += 1;
  }
  s->count_n = cn;
}
static void hskpng_count(orc_particles *s)
{
  hskpng_sort(s);
  count_runs(s);
}
/* hskpng_vterm.ipp:15-33,185-342 */
static int vt0_bin(const orc_particles *s, real rw2)
{
  const int n_bin = 10000;
  const real dlnr = (s->vt0_ln_r_max - s->vt0_ln_r_min) / n_bin;
  const real lnr = .5 * log(rw2);
  return lnr <= s->vt0_ln_r_min ? 0 : lnr >= s->vt0_ln_r_max ? n_bin - 1 : (int)((lnr - s->vt0_ln_r_min) / dlnr);
}
static real vt_of(const orc_particles *s, real rw2, sz c)
{
  const real r = sqrt(rw2);
  switch (s->o.terminal_velocity) {
    case LCX_VT_BEARD76: return vt_beard76(r, s->T[c], s->p[c], s->rhod[c], s->eta[c]);
    case LCX_VT_BEARD77: return vt_beard77_fact(r, s->p[c], s->rhod[c], s->eta[c]) * vt_beard77_v0(r);
    case LCX_VT_BEARD77FAST: return vt_beard77_fact(r, s->p[c], s->rhod[c], s->eta[c]) * s->vt_0[vt0_bin(s, rw2)];
    case LCX_VT_KHVOROSTYANOV_SPHERICAL: return vt_khvorostyanov(r, s->T[c], s->rhod[c], s->eta[c], 1);
    case LCX_VT_KHVOROSTYANOV_NONSPHERICAL: return vt_khvorostyanov(r, s->T[c], s->rhod[c], s->eta[c], 0);
    default: return 0.;
  }
}
static void hskpng_vterm(orc_particles *s, int only_invalid)
{
  OMP_FOR
  for (sz p = 0; p < s->n_part; ++p) {
    if (only_invalid ? (s->vt[p] == -1. && s->rw2[p] > 0) : (s->rw2[p] > 0))
      s->vt[p] = vt_of(s, s->rw2[p], s->ijk[p]);
  }
}
/* init_vterm.ipp:36-59 */
static void init_vterm(orc_particles *s)
{
  if (s->o.terminal_velocity != LCX_VT_BEARD77FAST) return;
  const int n_bin = 10000;
  const real dlnr = (s->vt0_ln_r_max - s->vt0_ln_r_min) / n_bin;
  for (int it = 0; it < n_bin; ++it) s->vt_0[it] = vt_beard77_v0(exp(s->vt0_ln_r_min + (it + 0.5) * dlnr));
}
/* hskpng_remove.ipp:20-76 (stable), hskpng_resize.ipp:7-32 */
static int mig_attrs(orc_particles *s, real **a);
static int hskpng_remove_n0(orc_particles *s)
{
  real *attrs[24]; const int na = mig_attrs(s, attrs);       /* every registered attribute (distmem_real_vctrs) + n */
#ifdef _OPENMP
  {
    const sz n = s->n_part;
    unsigned char *keep = (unsigned char *)scr_get(SCR_FLAG, n);
    sz *idx = (sz *)scr_get(SCR_IDX, n * sizeof(sz));
    OMP_FOR
    for (sz p = 0; p < n; ++p) keep[p] = s->n[p] != 0;
    const sz m = par_select(keep, n, idx);
    if (m != n) {
      real *tmp = (real *)scr_get(SCR_TMP, m * sizeof(real));
      for (int a = 0; a < na; ++a) {
        OMP_FOR
        for (sz i = 0; i < m; ++i) tmp[i] = attrs[a][idx[i]];
        OMP_FOR
        for (sz i = 0; i < m; ++i) attrs[a][i] = tmp[i];
      }
      n_t *tn = (n_t *)tmp;                              /* (n_t and real are both 8 bytes) */
      OMP_FOR
      for (sz i = 0; i < m; ++i) tn[i] = s->n[idx[i]];
      OMP_FOR
      for (sz i = 0; i < m; ++i) s->n[i] = tn[i];
    }
    s->n_part = m;
    return 0;
  }
#endif
  sz w = 0;
  for (sz p = 0; p < s->n_part; ++p) {
    if (s->n[p] == 0) continue;
    if (w != p) {
      s->n[w] = s->n[p];
      for (int a = 0; a < na; ++a) attrs[a][w] = attrs[a][p];
    }
    ++w;
  }
  s->n_part = w;
  return 0;
}

/* ---------------- moments (particles_impl_moms.ipp:50-387) ---------------- */
static void moms_all(orc_particles *s)
{
  hskpng_sort(s);
  OMP_FOR
  for (sz p = 0; p < s->n_part; ++p) s->n_filtered[p] = (real)s->n[p];
  s->selected_before_counting = 1;
}
static void moms_rng(orc_particles *s, real mn, real mx, const real *vec, int cons)
{
  hskpng_sort(s);
  for (sz p = 0; p < s->n_part; ++p) {
    const real y = cons ? s->n_filtered[p] : (real)s->n[p];
    s->n_filtered[p] = (vec[p] >= mn && vec[p] < mx) ? y : 0;
  }
  s->selected_before_counting = 1;
}
static void moms_gt0(orc_particles *s, const real *vec, int cons)
{
  hskpng_sort(s);
  for (sz p = 0; p < s->n_part; ++p) {
    const real y = cons ? s->n_filtered[p] : (real)s->n[p];
    s->n_filtered[p] = y * (vec[p] > 0);
  }
  s->selected_before_counting = 1;
}
static real moment_counter(real n, real x, real xp)
{
  return x >= 0 ? n * pow(x, xp) : n * pow(x, (real)(int)xp);
}
static void moms_calc(orc_particles *s, const real *vec, real power, int specific)
{
  sz cn = 0;
  real *vals = s->mom_vals;            /* not tmp_part: diag_precip_rate passes that one as vec */
  OMP_FOR
  for (sz p = 0; p < s->n_part; ++p) { const sz id = s->sorted_id[p]; vals[p] = moment_counter(s->n_filtered[id], vec[id], power); }
#ifdef _OPENMP
  {                                      /* reduce_by_key: one thread per run of a cell, summed in the serial order */
    sz *start = (sz *)scr_get(SCR_START, (s->n_part + 1) * sizeof(sz));
    cn = par_runs(s->sorted_ijk, s->n_part, start);
    OMP_FOR
    for (sz i = 0; i < cn; ++i) {
      real acc = vals[start[i]];
      for (sz p = start[i] + 1; p < start[i + 1]; ++p) acc = acc + vals[p];
      s->count_ijk[i] = s->sorted_ijk[start[i]]; s->count_mom[i] = acc;
    }
  }
#else
  for (sz p = 0; p < s->n_part; ++p) {
    const real v = vals[p];
    if (p == 0 || s->sorted_ijk[p] != s->sorted_ijk[p - 1]) { s->count_ijk[cn] = s->sorted_ijk[p]; s->count_mom[cn] = v; ++cn; }
    else s->count_mom[cn - 1] = s->count_mom[cn - 1] + v;
  }
#endif
  s->count_n = cn;
  if (specific && s->n_dims > 0)
    OMP_FOR
    for (sz i = 0; i < cn; ++i) {
      s->count_mom[i] = s->count_mom[i] / s->dv[s->count_ijk[i]];
      s->count_mom[i] = s->count_mom[i] / s->rhod[s->count_ijk[i]];
    }
}

/* ---------------- condensation driver ---------------- */
/* sstp_save.ipp:7-30 (per-cell version) */
static void sstp_save(orc_particles *s)
{
  if (!s->allow_sstp_cond) return;
  if (s->o.exact_sstp_cond) {                  /* per-particle version: copy through ijk (sstp_save.ipp:17-22) */
    for (sz p = 0; p < s->n_part; ++p) {
      const sz c = s->ijk[p];
      s->pp_rv[p] = s->rv[c]; s->pp_th[p] = s->th[c]; s->pp_rh[p] = s->rhod[c];
      if (s->o.const_p) s->pp_p[p] = s->p[c];
    }
    return;
  }
  memcpy(s->sstp_tmp_rv, s->rv, s->n_cell * sizeof(real));
  memcpy(s->sstp_tmp_th, s->th, s->n_cell * sizeof(real));
  memcpy(s->sstp_tmp_rh, s->rhod, s->n_cell * sizeof(real));
}
/* sstp_percell_step.ipp:7-48 */
static void sstp_percell_step(orc_particles *s, int step)
{
  if (s->sstp_cond == 1) return;
  real *scl[3] = {s->rv, s->th, s->rhod}, *tmp[3] = {s->sstp_tmp_rv, s->sstp_tmp_th, s->sstp_tmp_rh};
  const real sstp = s->sstp_cond;
  for (int ix = 0; ix < (s->var_rho ? 3 : 2); ++ix)
    for (sz c = 0; c < s->n_cell; ++c) {
      if (step == 0) {
        tmp[ix][c] = scl[ix][c] - tmp[ix][c];
        scl[ix][c] = scl[ix][c] - (sstp - 1) * tmp[ix][c] / sstp;
      } else scl[ix][c] = scl[ix][c] + tmp[ix][c] / sstp;
    }
}
/* save_liq_ice_content_before_change.ipp:13-31 */
static void save_liq_before(orc_particles *s)
{
  hskpng_sort(s);
  moms_all(s);
  moms_calc(s, s->rw2, 3. / 2., 1);
  if (s->count_n != s->n_cell) for (sz c = 0; c < s->n_cell; ++c) s->drw_mom3[c] = 0.;
  OMP_FOR
  for (sz i = 0; i < s->count_n; ++i) s->drw_mom3[s->count_ijk[i]] = -s->count_mom[i];
}
/* percell/particles_impl_cond.ipp:13-139 */
static void cond(orc_particles *s, real dt, real RH_max, int step, int turb_cond)
{
  hskpng_sort(s);
  if (step == 0) { if (s->count_n != s->n_cell) for (sz c = 0; c < s->n_cell; ++c) s->rw_mom3[c] = 0.; }
  else for (sz c = 0; c < s->n_cell; ++c) s->drw_mom3[c] = -s->rw_mom3[c];
  /* (dynamic chunks: the root finder's iteration count differs between haze and cloud droplets, i.e. between regions of the box) */
  _Pragma("omp parallel for schedule(dynamic, 2048)")
  for (sz p = 0; p < s->n_part; ++p) {
    const sz c = s->ijk[p];
    cond_ctx cc = {s->rw2[p], dt / s->sstp_cond, s->rhod[c], s->rv[c], s->T[c], s->p[c], s->RH[c] + (turb_cond ? s->ssp[p] : 0.), s->eta[c],
                   s->rd3[p], s->kpa[p], s->vt[p], RH_max, s->lambda_D[c], s->lambda_K[c]};   /* RH_sgs, percell/particles_impl_cond.ipp:50-72 */
    s->rw2[p] = advance_rw2(&cc, s->eps_tol, 2., 100);     /* config.hpp:13,26: n_iter 100, cond_mlt 2 */
  }
  moms_all(s);
  moms_calc(s, s->rw2, 3. / 2., 1);
  if (step < s->sstp_cond - 1) {
    for (sz i = 0; i < s->count_n; ++i) s->rw_mom3[s->count_ijk[i]] = s->count_mom[i];
    for (sz c = 0; c < s->n_cell; ++c) s->drw_mom3[c] = s->rw_mom3[c] + s->drw_mom3[c];
  } else
  {
    OMP_FOR
    for (sz i = 0; i < s->count_n; ++i) s->drw_mom3[s->count_ijk[i]] = s->count_mom[i] + s->drw_mom3[s->count_ijk[i]];
  }
}
/* particles_impl_update_th_rv.ipp:74-191 */
static void update_th_rv(orc_particles *s)
{
  const real mlt = rho_w * (4. / 3) * ORC_PI;
  OMP_FOR
  for (sz c = 0; c < s->n_cell; ++c) s->drw_mom3[c] = s->drw_mom3[c] * mlt;
  OMP_FOR
  for (sz c = 0; c < s->n_cell; ++c) s->rv[c] = s->rv[c] - s->drw_mom3[c];
  OMP_FOR
  for (sz c = 0; c < s->n_cell; ++c) s->th[c] = s->th[c] - s->drw_mom3[c] * d_th_d_rv(s->T[c], s->th[c]);
}
/* ---------------- per-particle condensation substepping (src/impl/condensation/perparticle/, particles_step.ipp:199-236) ---------------- */
/* hskpng_rc2.ipp:14-32, particles_diag.ipp:41-62: rc2 = rw3_cr(rd3, kappa, rc2_T + 273.15)^(2/3) where rc2 == invalid */
static void hskpng_approximate_rc2_invalid(orc_particles *s)
{
  if (s->sstp_cond_act == 1 || !s->allow_sstp_cond) return;
  for (sz p = 0; p < s->n_part; ++p)
    if (s->rc2[p] == -1.) s->rc2[p] = pow(rw3_cr(s->rd3[p], s->kpa[p], s->o.rc2_T + 273.15), 2. / 3);
}
/* calculate_noncond_perparticle_sstp_delta.ipp:13-43 */
static void calculate_noncond_perparticle_sstp_delta(orc_particles *s)
{
  for (sz p = 0; p < s->n_part; ++p) {
    const sz c = s->ijk[p];
    s->dlt_rv[p] = s->rv[c] - s->pp_rv[p];
    s->dlt_th[p] = s->th[c] - s->pp_th[p];
    s->dlt_rh[p] = s->rhod[c] - s->pp_rh[p];
    if (s->o.const_p) s->dlt_p[p] = s->p[c] - s->pp_p[p];
  }
}
/* apply_noncond_perparticle_sstp_delta.ipp:11-31 */
static void apply_noncond_perparticle_sstp_delta(orc_particles *s)
{
  for (sz p = 0; p < s->n_part; ++p) {
    s->pp_rv[p] = s->pp_rv[p] + s->dlt_rv[p] / s->sstp_cond;
    s->pp_th[p] = s->pp_th[p] + s->dlt_th[p] / s->sstp_cond;
    s->pp_rh[p] = s->pp_rh[p] + s->dlt_rh[p] / s->sstp_cond;
    if (s->o.const_p) s->pp_p[p] = s->pp_p[p] + s->dlt_p[p] / s->sstp_cond;
  }
}
static real rw2torw3(real rw2) { return rw2 * sqrt(rw2); }       /* cond_common.ipp:57-67 */
/* cond_common.ipp:24-41 */
static real rw3diff2drv(const orc_particles *s, real rw3diff, real rhod, n_t n, real dv)
{
  const real mlt = -rho_w * (4. / 3) * ORC_PI;
  if (s->n_dims > 0) return mlt * rw3diff * (real)n / rhod / dv;
  return mlt * rw3diff * (real)n;
}
static real pp_T(const orc_particles *s, real th, real rhod, real p)
{ return s->o.th_dry ? theta_dry_T(th, rhod) : th * theta_std_exner(p); }     /* hskpng_Tpr.ipp:26-46 */
/* cond_perparticle_advance_rw2.ipp:30-125 + perparticle_advance_rw2.ipp:8-43 */
static void cond_perparticle_advance_rw2(orc_particles *s, real RH_max, int turb_cond)
{
  for (sz p = 0; p < s->n_part; ++p) s->Tp[p] = pp_T(s, s->pp_th[p], s->pp_rh[p], s->pp_p[p]);
  for (sz p = 0; p < s->n_part; ++p) {
    const sz c = s->ijk[p];
    const real pr = s->o.const_p ? s->pp_p[p] : theta_dry_p(s->pp_rh[p], s->pp_rv[p], s->Tp[p]);
    const real RH = RH_of(s->o.RH_formula, pr, s->pp_rv[p], s->Tp[p]) + (turb_cond ? s->ssp[p] : 0.);   /* RH_sgs, :8-22 */
    cond_ctx cc = {s->rw2[p], s->dt / s->sstp_cond, s->pp_rh[p], s->pp_rv[p], s->Tp[p], pr, RH, visc(s->Tp[p]),
                   s->rd3[p], s->kpa[p], s->vt[p], RH_max, s->lambda_D[c], s->lambda_K[c]};
    s->rw2[p] = advance_rw2(&cc, s->eps_tol, 2., 100);
  }
}
/* update_th_rv.ipp:243-283: per-cell sum (sorted order) of a per-particle change, added to every particle of the cell */
static void update_pstate(orc_particles *s, real *pstate, const real *pdstate)
{
  real *dstate = s->scl;
  for (sz c = 0; c < s->n_cell; ++c) dstate[c] = 0.;
  sz nseg = 0;
  for (sz q = 0; q < s->n_part; ++q) {                     /* thrust::reduce_by_key over the sorted order */
    const real v = pdstate[s->sorted_id[q]];
    if (q == 0 || s->sorted_ijk[q] != s->sorted_ijk[q - 1]) { s->count_ijk[nseg] = s->sorted_ijk[q]; s->count_mom[nseg] = v; ++nseg; }
    else s->count_mom[nseg - 1] = s->count_mom[nseg - 1] + v;
  }
  s->count_n = nseg;
  for (sz i = 0; i < nseg; ++i) dstate[s->count_ijk[i]] = s->count_mom[i] + dstate[s->count_ijk[i]];
  for (sz p = 0; p < s->n_part; ++p) pstate[p] = pstate[p] + dstate[s->ijk[p]];
}
/* apply_perparticle_drw3_to_perparticle_rv_and_th.ipp:13-60 */
static void apply_perparticle_drw3_to_perparticle_rv_and_th(orc_particles *s)
{
  real *drw3 = s->drwX;
  for (sz p = 0; p < s->n_part; ++p) drw3[p] = rw3diff2drv(s, drw3[p], s->pp_rh[p], s->n[p], s->dv[s->ijk[p]]);
  if (s->o.sstp_cond_mix) update_pstate(s, s->pp_rv, drw3);
  else for (sz p = 0; p < s->n_part; ++p) s->pp_rv[p] = drw3[p] + s->pp_rv[p];
  for (sz p = 0; p < s->n_part; ++p) drw3[p] = drw3[p] * d_th_d_rv(s->Tp[p], s->pp_th[p]);
  if (s->o.sstp_cond_mix) update_pstate(s, s->pp_th, drw3);
  else for (sz p = 0; p < s->n_part; ++p) s->pp_th[p] = drw3[p] + s->pp_th[p];
}
/* calc_liq_ice_content_change.ipp:12-26 */
static void calc_liq_content_change(orc_particles *s)
{
  moms_all(s);
  moms_calc(s, s->rw2, 3. / 2., 1);
  for (sz i = 0; i < s->count_n; ++i) s->drw_mom3[s->count_ijk[i]] = s->count_mom[i] + s->drw_mom3[s->count_ijk[i]];
}
/* perparticle_nomixing_adaptive_sstp_cond.ipp:56-265, one super-droplet */
static void adaptive_sstp_cond_one(orc_particles *s, sz p, real RH_max, int turb_cond)
{
  const lcx_opts_init_t *o = &s->o;
  const sz c = s->ijk[p];
  const real dlt_rv = s->dlt_rv[p], dlt_th = s->dlt_th[p], dlt_rhod = s->dlt_rh[p], dlt_p = s->dlt_p[p];
  const n_t n = s->n[p];
  const real dv = s->dv[c], lambda_D = s->lambda_D[c], lambda_K = s->lambda_K[c], rd3 = s->rd3[p], kpa = s->kpa[p], vt = s->vt[p];
  const int sstp_cond_max = s->sstp_cond, sstp_cond_act = s->sstp_cond_act;
  unsigned sstp_cond;
  real t_rv = s->pp_rv[p], t_th = s->pp_th[p], t_rh = s->pp_rh[p], t_p = o->const_p ? s->pp_p[p] : 0., rw2 = s->rw2[p];
  real drw2 = 0, Tp = 0, RH = 0, delta_fraction_applied = 0;
  const real dot_ssp = turb_cond ? s->dot_ssp[p] : 0.;                       /* :66,74 */
  real ssp = turb_cond ? s->ssp[p] : 0.;
#define APPLY_DELTA(mult) do { const real m_ = (mult); t_rv += dlt_rv * m_; t_th += dlt_th * m_; t_rh += dlt_rhod * m_; \
                               if (o->const_p) { t_p += dlt_p * m_; } if (turb_cond) { ssp += dot_ssp * s->dt * m_; } } while (0)
#define CALC_STATE() do { Tp = pp_T(s, t_th, t_rh, t_p); if (!o->const_p) t_p = theta_dry_p(t_rh, t_rv, Tp); \
                          RH = RH_of(o->RH_formula, t_p, t_rv, Tp) + (turb_cond ? ssp : 0.); } while (0)
  int first_cond_step_done_in_adaptation = sstp_cond_max == 1 ? 1 : 0;
  {
    real drw2_new = 0;
    sstp_cond = (unsigned)sstp_cond_max;
    for (int sstp_cond_try = 1; sstp_cond_try <= sstp_cond_max; sstp_cond_try *= 2) {
      delta_fraction_applied = sstp_cond_try == 1 ? 1 : -1. / sstp_cond_try;
      APPLY_DELTA(delta_fraction_applied);
      CALC_STATE();
      cond_ctx cc = {rw2, s->dt / sstp_cond_try, t_rh, t_rv, Tp, t_p, RH, visc(Tp), rd3, kpa, vt, RH_max, lambda_D, lambda_K};
      const real d = advance_rw2_apply(&cc, s->eps_tol, 2., 100, 0);
      if (sstp_cond_try == 1) drw2 = d; else drw2_new = d;
      if (sstp_cond_try > 1) {
        if ((fabs(drw2_new * 2 - drw2) <= o->sstp_cond_adapt_drw2_eps * rw2) && (fabs(drw2) < o->sstp_cond_adapt_drw2_max * rw2)) {
          sstp_cond = (unsigned)(sstp_cond_try / 2);
          APPLY_DELTA(-delta_fraction_applied);
          first_cond_step_done_in_adaptation = 1;
          break;
        }
        drw2 = drw2_new;
      }
    }
    if (sstp_cond_act > 1) {
      const real rc2 = s->rc2[p];
      if ((rw2 < rc2 && (rw2 + sstp_cond * drw2) > rc2) || (rw2 > rc2 && (rw2 + sstp_cond * drw2) < rc2)) {
        sstp_cond = (unsigned)sstp_cond_act;
        first_cond_step_done_in_adaptation = 0;
      }
    }
    if (!first_cond_step_done_in_adaptation)
      APPLY_DELTA(sstp_cond_max == 1 ? -delta_fraction_applied : delta_fraction_applied);
  }
  delta_fraction_applied = 1. / sstp_cond;
  real rw3 = drw2, drw3;          /* `real_t &rw3 = drw2`: drw2 is only needed at the start of the first step */
  for (unsigned step = 0; step < sstp_cond; ++step) {
    drw3 = step > 0 ? -rw3 : -rw2torw3(rw2);
    if (first_cond_step_done_in_adaptation && step == 0) rw2 += rw3;      /* rw3 aliases drw2 here */
    else {
      APPLY_DELTA(delta_fraction_applied);
      CALC_STATE();
      cond_ctx cc = {rw2, s->dt / sstp_cond, t_rh, t_rv, Tp, t_p, RH, visc(Tp), rd3, kpa, vt, RH_max, lambda_D, lambda_K};
      rw2 = advance_rw2_apply(&cc, s->eps_tol, 2., 100, 1);
    }
    if (step < sstp_cond - 1) { rw3 = rw2torw3(rw2); drw3 += rw3; }
    else drw3 += rw2torw3(rw2);
    drw3 = rw3diff2drv(s, drw3, t_rh, n, dv);
    t_rv += drw3;
    drw3 = drw3 * d_th_d_rv(Tp, t_th);
    t_th += drw3;
  }
#undef APPLY_DELTA
#undef CALC_STATE
  s->pp_sstp[p] = sstp_cond;
  s->pp_rv[p] = t_rv; s->pp_th[p] = t_th; s->pp_rh[p] = t_rh;
  if (o->const_p) s->pp_p[p] = t_p;
  if (turb_cond) s->ssp[p] = ssp;
  s->rw2[p] = rw2;
}
/* particles_step.ipp:199-236 */
static void cond_perparticle(orc_particles *s, real RH_max, int turb_cond)
{
  if (!s->o.sstp_cond_mix) save_liq_before(s);
  calculate_noncond_perparticle_sstp_delta(s);
  if (s->o.adaptive_sstp_cond) {
    for (sz p = 0; p < s->n_part; ++p) adaptive_sstp_cond_one(s, p, RH_max, turb_cond);
  } else {
    for (int step = 0; step < s->sstp_cond; ++step) {
      apply_noncond_perparticle_sstp_delta(s);
      if (turb_cond) for (sz p = 0; p < s->n_part; ++p) s->ssp[p] = s->ssp[p] + s->dt / s->sstp_cond * s->dot_ssp[p];   /* apply_perparticle_sgs_supersat.ipp */
      if (step == 0) for (sz p = 0; p < s->n_part; ++p) s->drwX[p] = -rw2torw3(s->rw2[p]);   /* set_perparticle_drwX_to_minus_rwX */
      else           for (sz p = 0; p < s->n_part; ++p) s->drwX[p] = -s->rwX[p];
      cond_perparticle_advance_rw2(s, RH_max, turb_cond);
      if (step < s->sstp_cond - 1) for (sz p = 0; p < s->n_part; ++p) { s->rwX[p] = rw2torw3(s->rw2[p]); s->drwX[p] = s->rwX[p] + s->drwX[p]; }
      else                         for (sz p = 0; p < s->n_part; ++p) s->drwX[p] = rw2torw3(s->rw2[p]) + s->drwX[p];
      apply_perparticle_drw3_to_perparticle_rv_and_th(s);
    }
  }
  /* apply_perparticle_cond_change_to_percell_rv_and_th.ipp:11-24 */
  if (s->o.sstp_cond_mix) {
    for (sz p = 0; p < s->n_part; ++p) s->rv[s->ijk[p]] = s->pp_rv[p];      /* update_state: last particle of a cell wins */
    for (sz p = 0; p < s->n_part; ++p) s->th[s->ijk[p]] = s->pp_th[p];
  } else {
    calc_liq_content_change(s);
    update_th_rv(s);
  }
}

/* ---------------- SGS turbulence (hskpng_tke.ipp, hskpng_turb_vel.ipp, hskpng_turb_ss.ipp; common/GA17_turbulence.hpp) ---------------- */
static void hskpng_tke(orc_particles *s)
{                                                   /* diss_rate := TKE = ((L eps) / C_E)^(2/3), L = SGS_mix_len[k] */
  const sz nz = m1(s->o.nz);
  for (sz c = 0; c < s->n_cell; ++c) {
    const real ret = cbrt((s->SGS_mix_len[c % nz] * s->diss_rate[c]) / 0.845);
    s->diss_rate[c] = ret * ret;
  }
}
static void hskpng_turb_vel(orc_particles *s, real dt, int only_vertical)
{
  const sz nz = m1(s->o.nz);
  const real cube_root_of_two_pi = pow(2. * ORC_PI, 1. / 3.);
  for (sz c = 0; c < s->n_cell; ++c) s->tau_cell[c] = s->SGS_mix_len[c % nz] / cube_root_of_two_pi * sqrt(1.5 / s->diss_rate[c]);
  real *vel[3] = {s->up, s->wp, s->vp};
  for (int i = only_vertical ? 1 : 0; i < (only_vertical ? 2 : s->n_dims); ++i) {
    for (sz p = 0; p < s->n_part; ++p) s->tmp_part[p] = rng_normal(&s->rng, &s->rng_ns);
    for (sz p = 0; p < s->n_part; ++p) {
      const sz c = s->ijk[p];
      const real e = exp(-dt / s->tau_cell[c]);                              /* update_turb_vel */
      vel[i][p] = vel[i][p] * e + sqrt((1. - e * e) * (2. / 3.) * s->diss_rate[c]) * s->tmp_part[p];
    }
  }
}
static void hskpng_turb_dot_ss(orc_particles *s)
{
  real *tau_rlx = s->scl;
  moms_all(s);
  moms_calc(s, s->rw2, 1. / 2, 0);
  for (sz i = 0; i < s->count_n; ++i) tau_rlx[s->count_ijk[i]] = 1. / (2.8e-4 * (s->count_mom[i] / s->dv[s->count_ijk[i]]));   /* tau_relax */
  for (sz p = 0; p < s->n_part; ++p) s->dot_ssp[p] = 3e-4 * s->wp[p] - s->ssp[p] / tau_rlx[s->ijk[p]];                          /* dot_turb_ss */
}

/* particles_impl_adjust_timesteps.ipp:13-24 */
static int adjust_timesteps(orc_particles *s, real dt)
{
  if (dt > 0 && !s->o.variable_dt_switch) FAIL("libcloudph++: opts.dt specified, but opts_init.variable_dt_switch is false.");
  s->sstp_cond = dt > 0 && s->o.sstp_cond > 1 ? (int)ceil(s->o.sstp_cond * dt / s->o.dt) : s->o.sstp_cond;
  s->sstp_cond_act = dt > 0 && s->o.sstp_cond_act > 1 ? (int)ceil(s->o.sstp_cond_act * dt / s->o.dt) : s->o.sstp_cond_act;
  s->sstp_coal = dt > 0 && s->o.sstp_coal > 1 ? (int)ceil(s->o.sstp_coal * dt / s->o.dt) : s->o.sstp_coal;
  s->dt = dt > 0 ? dt : s->o.dt;
  return 0;
}

/* ---------------- coalescence (particles_impl_coal.ipp:99-546, src/detail/kernels.hpp:38-202) ---------------- */
static int kernel_index(n_t R) { return R <= 100. ? (int)R : (int)(100 + (R - 100.) / 10.); }       /* kernel_utils.hpp:10-18 */
static sz kernel_vector_index(int i, int j, n_t nup)
{                                                                                             /* kernel_utils.hpp:21-29 */
  return i >= j ? (sz)(0.5 * i * (i + 1) + j + nup) : (sz)(0.5 * j * (j + 1) + i + nup);
}
static real interpolated_efficiency(const orc_particles *s, real r1, real r2)
{                                                                                             /* kernel_interpolation.hpp:9-65 */
  const real *kp = s->kernel_parameters; const n_t nup = s->n_user_params;
  r1 *= 1e6; r2 *= 1e6;
  if (r1 >= s->kernel_r_max) r1 = s->kernel_r_max - 1e-6;
  if (r2 >= s->kernel_r_max) r2 = s->kernel_r_max - 1e-6;
  n_t dx, dy, x[4];
  if (r1 >= 100.) { x[0] = (n_t)(floor(r1 / 10.) * 10); dx = 10; } else { x[0] = (n_t)floor(r1); dx = 1; }
  if (r2 >= 100.) { x[2] = (n_t)(floor(r2 / 10.) * 10); dy = 10; } else { x[2] = (n_t)floor(r2); dy = 1; }
  x[1] = x[0] + dx; x[3] = x[2] + dy;
  sz iv[4];
  iv[0] = kernel_vector_index(kernel_index(x[0]), kernel_index(x[2]), nup);
  iv[1] = kernel_vector_index(kernel_index(x[1]), kernel_index(x[2]), nup);
  iv[2] = kernel_vector_index(kernel_index(x[0]), kernel_index(x[3]), nup);
  iv[3] = kernel_vector_index(kernel_index(x[1]), kernel_index(x[3]), nup);
  real w[4];
  w[0] = r1 - x[0]; w[1] = x[1] - r1; w[2] = r2 - x[2]; w[3] = x[3] - r2;
  return (kp[iv[0]] * w[1] * w[3] + kp[iv[1]] * w[0] * w[3] + kp[iv[2]] * w[1] * w[2] + kp[iv[3]] * w[0] * w[2]) / dx / dy;
}
static real k_geometric(n_t na, n_t nb, real rw2a, real rw2b, real vta, real vtb)
{
  const n_t nmax = na < nb ? nb : na;
  return ORC_PI * nmax * fabs(vta - vtb) * (rw2a + rw2b + 2. * sqrt(rw2a * rw2b));
}
/* Onishi turbulent kernel without gravitational settling, 2 pi R^2 <|Wr|> g(R) (src/detail/kernel_onishi_nograv.hpp:29-153).
   The reference computes the Kolmogorov length as pow(nu^3/eps, real_t(1/4)) with an INTEGER 1/4 == 0, i.e. leta == 1:
   reproduced, the results are what the reference produces. */
static real kernel_onishi_nograv(real r1, real r2, real Re_l, real eps, real dnu, real ratio_den)
{
  if (eps < 1e-10) return 0.;
  const real urms = sqrt(Re_l / sqrt(15. / dnu / eps));
  const real CR = r1 + r2;
  const real taup1 = ratio_den * 4. * r1 * r1 / 18. / dnu, taup2 = ratio_den * 4. * r2 * r2 / 18. / dnu;
  const real leta = pow(dnu * dnu * dnu / eps, 0.);
  const real tauk = leta * leta / dnu;
  const real Te = Re_l * tauk / sqrt(15.);
  const real theta1 = 2.5 * taup1 / Te, theta2 = 2.5 * taup2 / Te;
  const real phi = dmax(theta2 / theta1, theta1 / theta2);
  const real cw = 1. + 0.6 * exp(-pow(phi - 1., 1.5));
  real gamma = 0.183 * urms * urms / (dnu * dnu / leta / leta);
  gamma = phi * gamma;
  const real WrS2 = (dnu * dnu * CR * CR) / (leta * leta * leta * leta) / 15.;
  real WrA2 = urms * urms * gamma / (gamma - 1.)
    * ((theta1 + theta2) - 4. * theta1 * theta2 / (theta1 + theta2) * sqrt((1. + theta1 + theta2) / (1. + theta1) / (1. + theta2)))
    * (1. / (1. + theta1) / (1. + theta2) - 1. / (1. + gamma * theta1) / (1. + gamma * theta2));
  WrA2 = cw * WrA2;
  WrA2 = WrA2 / 3.;
  const real Wr = sqrt(2. / ORC_PI * (WrA2 + WrS2));
  const real A1 = 110.0, A2 = 0.38, A3 = 0.16;
  real alpha = log10(0.26 * sqrt(Re_l)) / log10(2.0);
  alpha = dmax(alpha, 1.e-20);
  const real CA = 0.06 * pow(Re_l, 0.30), CB = 0.4;
  const real StA = pow(A2 / A1 * Re_l, 0.25);
  const real hlpr = cbrt(A2 / A3);
  const real StB = hlpr * hlpr * cbrt(Re_l);
  const real St1 = taup1 / tauk, St2 = taup2 / tauk;
  real y11, y21, y12, y22;
  if (St2 <= StA) { y11 = A1 * St1 * St1; y21 = 0.; } else { y11 = 0.; y21 = A2 * Re_l / (St1 * St1); }
  const real y31 = A3 * sqrt(Re_l / St1);
  if (St1 <= StA) { y12 = A1 * St2 * St2; y22 = 0.; } else { y12 = 0.; y22 = A2 * Re_l / (St2 * St2); }
  const real y32 = A3 * sqrt(Re_l / St2);
  const real za1 = 0.5 * (1. - tanh((log10(St1) - log10(StA)) / CA));
  const real zb1 = 0.5 * (1. + tanh((log10(St1) - log10(StB)) / CB));
  const real za2 = 0.5 * (1. - tanh((log10(St2) - log10(StA)) / CA));
  const real zb2 = 0.5 * (1. + tanh((log10(St2) - log10(StB)) / CB));
  const real gR1 = y11 * pow(za1, alpha) + y21 * pow(1. - za1, alpha) + y31 * zb1 + 1.;
  const real gR2 = y12 * pow(za2, alpha) + y22 * pow(1. - za2, alpha) + y32 * zb2 + 1.;
  const real xai = dmax(taup2 / taup1, taup1 / taup2);
  const real RG12 = 2.6 * exp(-xai) + 0.205 * exp(-0.0206 * xai) * 0.5 * (1.0 + tanh(xai - 3.0));
  const real gR = 1. + RG12 * sqrt(gR1 - 1.) * sqrt(gR2 - 1.);
  return 2. * ORC_PI * CR * CR * Wr * gR;
}
/* Wang et al. (2009) turbulent enhancement of the collision efficiency (src/detail/wang_collision_enhancement.hpp:13-92);
   table values: the published Table 1 as the reference tabulates it, [ratio][eps class][collector radius] */
static real wang_collision_enhancement(real r1, real r2, real eps)
{
  static const real R0[7] = {10e-6, 20e-6, 30e-6, 40e-6, 50e-6, 60e-6, 100e-6};
  static const real rat[11] = {0., .1, .2, .3, .4, .5, .6, .7, .8, .9, 1.};
  static const real eta_e[11][2][7] = {
    {{1.74, 1.74, 1.773, 1.49, 1.207, 1.207, 1.0}, {4.976, 4.976, 3.593, 2.519, 1.445, 1.445, 1.0}},
    {{1.46, 1.46, 1.421, 1.245, 1.069, 1.069, 1.0}, {2.984, 2.984, 2.181, 1.691, 1.201, 1.201, 1.0}},
    {{1.32, 1.32, 1.245, 1.123, 1.000, 1.000, 1.0}, {1.988, 1.988, 1.475, 1.313, 1.150, 1.150, 1.0}},
    {{1.250, 1.250, 1.148, 1.087, 1.025, 1.025, 1.0}, {1.490, 1.490, 1.187, 1.156, 1.126, 1.126, 1.0}},
    {{1.186, 1.186, 1.066, 1.060, 1.056, 1.056, 1.0}, {1.249, 1.249, 1.088, 1.090, 1.092, 1.092, 1.0}},
    {{1.045, 1.045, 1.000, 1.014, 1.028, 1.028, 1.0}, {1.139, 1.139, 1.130, 1.091, 1.051, 1.051, 1.0}},
    {{1.070, 1.070, 1.030, 1.038, 1.046, 1.046, 1.0}, {1.220, 1.220, 1.190, 1.138, 1.086, 1.086, 1.0}},
    {{1.000, 1.000, 1.054, 1.042, 1.029, 1.029, 1.0}, {1.325, 1.325, 1.267, 1.165, 1.063, 1.063, 1.0}},
    {{1.223, 1.223, 1.117, 1.069, 1.021, 1.021, 1.0}, {1.716, 1.716, 1.345, 1.223, 1.100, 1.100, 1.0}},
    {{1.570, 1.570, 1.244, 1.166, 1.088, 1.088, 1.0}, {3.788, 3.788, 1.501, 1.311, 1.120, 1.120, 1.0}},
    {{20.3, 20.3, 14.6, 8.61, 2.60, 2.60, 1.0}, {36.52, 36.52, 19.16, 22.80, 26.0, 26.0, 1.0}}};
  const real R = r1 > r2 ? r1 : r2, r = r1 > r2 ? r2 : r1;
  if (R > 100e-6) return 1.;
  const int n_eps = eps <= 2.5e-2 ? 0 : 1;
  int n_R0, n_rat;
  for (n_R0 = 0; n_R0 < 7; ++n_R0) if (R0[n_R0] > R) break;
  const real ratio = r / R;
  for (n_rat = 1; n_rat < 11; ++n_rat) if (rat[n_rat] > ratio) break;
  if (n_R0 == 0) return eta_e[n_rat][n_eps][n_R0];
  const real w0 = R - R0[n_R0 - 1], w1 = R0[n_R0] - R, w2 = ratio - rat[n_rat - 1], w3 = rat[n_rat] - ratio;
  return (eta_e[n_rat - 1][n_eps][n_R0 - 1] * w1 * w3 + eta_e[n_rat - 1][n_eps][n_R0] * w0 * w3 +
          eta_e[n_rat][n_eps][n_R0 - 1] * w1 * w2 + eta_e[n_rat][n_eps][n_R0] * w0 * w2)
         / (R0[n_R0] - R0[n_R0 - 1]) / (rat[n_rat] - rat[n_rat - 1]);
}
/* c = cell of the pair, diss = its TKE dissipation rate when opts.turb_coal, else 0 (coal.ipp:392-416,439-451) */
static real kernel_calc(const orc_particles *s, n_t na, n_t nb, real rw2a, real rw2b, real vta, real vtb, sz c, real diss)
{
  switch (s->o.kernel) {
    case LCX_KERNEL_ONISHI_HALL:
    case LCX_KERNEL_ONISHI_HALL_DAVIS_NO_WAALS: {                                   /* kernel_onishi::calc, kernels.hpp:209-250 */
      const real rwa = sqrt(rw2a), rwb = sqrt(rw2b);
      const real Re_l = s->kernel_parameters[0];
      const real nograv = kernel_onishi_nograv(rwa, rwb, Re_l, diss, s->eta[c] / s->rhod[c], 1e3 / s->rhod[c]);
      const real geometric = k_geometric(na, nb, rw2a, rw2b, vta, vtb);
      /* the reference hands k_params[0] (Re_lambda) to the enhancement's dissipation-rate argument: reproduced */
      return interpolated_efficiency(s, rwa, rwb) * wang_collision_enhancement(rwa, rwb, Re_l) *
             sqrt(geometric * geometric + nograv * nograv);
    }
    case LCX_KERNEL_GOLOVIN: {
      const n_t nmax = na < nb ? nb : na;
      return ORC_PI * 4. / 3. * s->kernel_parameters[0] * nmax * (rw2a * sqrt(rw2a) + rw2b * sqrt(rw2b));
    }
    case LCX_KERNEL_GEOMETRIC:
      if (s->n_user_params == 1) return k_geometric(na, nb, rw2a, rw2b, vta, vtb) * s->kernel_parameters[0];
      return k_geometric(na, nb, rw2a, rw2b, vta, vtb);
    case LCX_KERNEL_LONG: {
      real res = k_geometric(na, nb, rw2a, rw2b, vta, vtb);
      const real r_L = dmax(sqrt(rw2a), sqrt(rw2b));
      if (r_L < 50.e-6) {
        const real r_s = dmin(sqrt(rw2a), sqrt(rw2b));
        if (r_s <= 3e-6) res = 0.; else res *= 4.5e8 * r_L * r_L * (1. - 3e-6 / r_s);
      }
      return res;
    }
    default: /* geometric with tabulated efficiencies */
      return interpolated_efficiency(s, sqrt(rw2a), sqrt(rw2b)) * k_geometric(na, nb, rw2a, rw2b, vta, vtb);
  }
}
/* collide<>, coal.ipp:110-143: _a has the higher multiplicity */
static void collide(orc_particles *s, sz a, sz b, n_t col_no)
{
  s->n[a] -= col_no * s->n[b];
  const real rw_b = cbrt(col_no * s->rw2[a] * sqrt(s->rw2[a]) + s->rw2[b] * sqrt(s->rw2[b]));
  s->rw2[b] = rw_b * rw_b;
  s->rd3[b] = col_no * s->rd3[a] + s->rd3[b];
  s->vt[b] = -1.;
  if (s->use_rc2) s->rc2[b] = -1.;        /* invalidator, coal.ipp:33-44,527-545 */
}
static void coal(orc_particles *s, real dt, int turb_coal)
{
  hskpng_sort_helper(s, 1);
  s->sorted = 1;
  count_runs(s);                                 /* hskpng_count on the shuffled order */
  OMP_FOR
  for (sz c = 0; c < s->n_cell; ++c) { s->scl[c] = 0.; s->off[c] = 0; }
  OMP_FOR
  for (sz i = 0; i < s->count_n; ++i) {
    const n_t n = s->count_num[i];
    s->scl[s->count_ijk[i]] = n > 1 ? ((real)(n * (n - 1)) / 2) / (n / 2) : 0;      /* scale_factor, coal.ipp:99-107 */
    s->off[s->count_ijk[i]] = (sz)n;
  }
  { sz acc = 0; for (sz c = 0; c < s->n_cell; ++c) { sz t = s->off[c]; s->off[c] = acc; acc += t; } }
  real *u01 = s->col;
  { dbl *rq = replay_pop(s, 0, s->n_part);
    if (rq) { for (sz p = 0; p < s->n_part; ++p) u01[p] = (real)rq[p]; free(rq); }
    else for (sz p = 0; p < s->n_part; ++p) u01[p] = rng_u01(&s->rng); }
  const sz n_pairs_end = s->n_part ? s->n_part - 1 : 0;
  OMP_FOR
  for (sz p = 0; p < n_pairs_end; ++p) {                   /* collider::operator(), coal.ipp:180-267 */
    const sz ca = s->sorted_ijk[p], cb = s->sorted_ijk[p + 1];
    const sz cix_a = p - s->off[ca];
    if (cix_a % 2 != 0) continue;
    const sz cix_b = (p + 1) - s->off[cb];
    if (cix_a != cix_b - 1) { s->col[p] = 0.; continue; }
    const sz a = s->sorted_id[p], b = s->sorted_id[p + 1];
    const real prob = dt / s->dv[ca] * s->scl[ca] *
      kernel_calc(s, s->n[a], s->n[b], s->rw2[a], s->rw2[b], s->vt[a], s->vt[b], ca, turb_coal ? s->diss_rate[ca] : 0.);
    n_t col_no = (n_t)prob;
    if (s->pure_const_multi && col_no >= 1) s->increase_sstp_coal = 1;
    if (u01[p] < prob - col_no) ++col_no;
    if (col_no == 0) { s->col[p] = 0.; s->col[p + 1] = 0.; continue; }
    if (s->n[a] >= s->n[b]) {
      if (s->n[b] > 0) { n_t q = s->n[a] / s->n[b]; if (q < col_no) col_no = q; }
      collide(s, a, b, col_no);
      s->col[p + 1] = -2.;
    } else {
      if (s->n[a] > 0) { n_t q = s->n[b] / s->n[a]; if (q < col_no) col_no = q; }
      collide(s, b, a, col_no);
      s->col[p + 1] = -1.;
    }
    s->col[p] = (real)col_no;
  }
  if (s->o.n_dry_distros + s->n_size_keys > 1)             /* weighted_summator, coal.ipp:57-97,458-480 */
    for (sz p = 0; p + 1 < s->n_part; ++p) {
      if (s->col[p] <= 0) continue;
      const sz a = s->sorted_id[p], b = s->sorted_id[p + 1];
      const int na_ge_nb = s->col[p + 1] == -2.;
      real rd3_old = na_ge_nb ? s->rd3[b] - s->col[p] * s->rd3[a] : s->rd3[a] - s->col[p] * s->rd3[b];
      for (int ci = 0; ci < s->col[p]; ++ci) {
        if (na_ge_nb) { s->kpa[b] = (s->kpa[a] * s->rd3[a] + s->kpa[b] * rd3_old) / (s->rd3[a] + rd3_old); rd3_old += s->rd3[a]; }
        else          { s->kpa[a] = (s->kpa[b] * s->rd3[b] + s->kpa[a] * rd3_old) / (s->rd3[b] + rd3_old); rd3_old += s->rd3[b]; }
      }
    }
  if (s->o.diag_incloud_time)                              /* selector, coal.ipp:17-31,505-525: the changed SD keeps the larger value */
    for (sz p = 0; p + 1 < s->n_part; ++p) {
      if (s->col[p] <= 0) continue;
      const sz a = s->sorted_id[p], b = s->sorted_id[p + 1];
      const real m = dmax(s->ict[a], s->ict[b]);
      if (s->col[p + 1] == -2.) s->ict[b] = m; else s->ict[a] = m;
    }
}

static int mig_attrs(orc_particles *s, real **a);
/* ---------------- advection, sedimentation, boundary (adve.ipp:28-304, sedi.ipp:13-25, subs.ipp:13-25, bcnd.ipp:99-368) -------- */
static real adve_1d(int scheme, real x, sz fl, real C_l, real C_r, real dx)
{
  if (scheme == LCX_ADVE_IMPLICIT) return (x + dx * (C_l - fl * (C_r - C_l))) / (1 - (C_r - C_l));
  return 1 * x + (C_r - C_l) * (x - dx * fl) + dx * C_l;
}
/* Courant numbers at the faces of extended-grid cell ce (grid with nx + 2 halo planes), init_grid.ipp:57-158 */
static void faces(const orc_particles *s, sz ce, real *Cxl, real *Cxr, real *Cyl, real *Cyr, real *Czl, real *Czr)
{
  const lcx_opts_init_t *o = &s->o;
  const sz nz = m1(o->nz), ny = m1(o->ny);
  const sz rgt = ce + (s->n_dims == 3 ? nz * ny : (sz)o->nz);
  *Cxl = s->courant_x[ce]; *Cxr = s->courant_x[rgt];
  if (s->n_dims > 2) { const sz fre = ce + (ce / (nz * ny)) * nz; *Cyl = s->courant_y[fre]; *Cyr = s->courant_y[fre + nz]; }
  if (s->n_dims > 1) {
    const sz blw = s->n_dims == 2 ? ce + ce / nz : ce + ny * (ce / (nz * ny)) + (ce - (ce / (nz * ny)) * (nz * ny)) / nz;
    *Czl = s->courant_z[blw]; *Czr = s->courant_z[blw + 1];
  }
}
static sz cell_ext(const orc_particles *s, real x, real y, real z)
{                                                   /* hskpng_ijk in the coordinates that start at the halo's left edge */
  const lcx_opts_init_t *o = &s->o;
  const sz i = o->nx ? (sz)(x / o->dx) : 0, j = o->ny ? (sz)(y / o->dy) : 0, k = o->nz ? (sz)(z / o->dz) : 0;
  switch (s->n_dims) { case 1: return i; case 2: return i * o->nz + k; default: return i * ((sz)o->nz * o->ny) + j * o->nz + k; }
}
static real periodic(real x, real a, real b);
/* adve.ipp:184-304: predictor-corrector with nearest-neighbour interpolation */
static void adve_pred_corr(orc_particles *s)
{
  const lcx_opts_init_t *o = &s->o;
  const sz nz = m1(o->nz), ny = m1(o->ny);
  const real shift = (real)s->halo * o->dx;
  for (sz p = 0; p < s->n_part; ++p) {
    real x = s->x[p] + shift, y = s->y[p], z = s->z[p];
    real Cxl, Cxr, Cyl = 0, Cyr = 0, Czl = 0, Czr = 0;
    sz ce = cell_ext(s, x, y, z);
    real x_old = x, y_old = y, z_old = z;
    faces(s, ce, &Cxl, &Cxr, &Cyl, &Cyr, &Czl, &Czr);             /* predictor: explicit Euler */
    {
      sz i, j = 0, k = 0;
      if (s->n_dims == 1) i = ce; else if (s->n_dims == 2) { i = ce / nz; k = ce % nz; } else { i = ce / (nz * ny); j = (ce / nz) % ny; k = ce % nz; }
      x = 1 * x + (Cxr - Cxl) * (x - o->dx * i) + o->dx * Cxl;
      if (s->n_dims > 2) y = 1 * y + (Cyr - Cyl) * (y - o->dy * j) + o->dy * Cyl;
      if (s->n_dims > 1) z = 1 * z + (Czr - Czl) * (z - o->dz * k) + o->dz * Czl;
    }
    if (s->n_dims > 1) {
      if (z >= o->z1) z = o->z1 - 1e-8 * o->dz;
      if (z <= o->z0) z = o->z0 + 1e-8 * o->dz;
    }
    if (s->n_dims == 3) {
      if (y >= o->y1) y_old = y_old + (o->y1 - o->y0);
      if (y < o->y0) y_old = y_old - (o->y1 - o->y0);
      y = periodic(y, o->y0, o->y1);
    }
    ce = cell_ext(s, x, y, z);                                      /* cell after the predictor step */
    s->ijk[p] = ce;
    x_old = x + x_old;
    if (s->n_dims > 2) y_old = y + y_old;
    if (s->n_dims > 1) z_old = z + z_old;
    faces(s, ce, &Cxl, &Cxr, &Cyl, &Cyr, &Czl, &Czr);             /* corrector: rhs at the midpoint position */
    {
      sz i, j = 0, k = 0;
      if (s->n_dims == 1) i = ce; else if (s->n_dims == 2) { i = ce / nz; k = ce % nz; } else { i = ce / (nz * ny); j = (ce / nz) % ny; k = ce % nz; }
      x = 0 * x + (Cxr - Cxl) * (x - o->dx * i) + o->dx * Cxl;
      if (s->n_dims > 2) y = 0 * y + (Cyr - Cyl) * (y - o->dy * j) + o->dy * Cyl;
      if (s->n_dims > 1) z = 0 * z + (Czr - Czl) * (z - o->dz * k) + o->dz * Czl;
    }
    x = (x + x_old) / 2.;
    if (s->n_dims > 2) y = (y + y_old) / 2.;
    if (s->n_dims > 1) z = (z + z_old) / 2.;
    s->x[p] = x - shift;
    if (s->n_dims > 2) s->y[p] = y;
    if (s->n_dims > 1) s->z[p] = z;
  }
}
static void adve(orc_particles *s)
{
  if (s->n_dims == 0) return;
  if (s->adve_scheme == LCX_ADVE_PRED_CORR) { adve_pred_corr(s); return; }
  const lcx_opts_init_t *o = &s->o;
  const sz nz = m1(o->nz), ny = m1(o->ny);
  const sz halo_cells = (sz)s->halo * (s->n_dims == 1 ? 1 : s->n_dims == 2 ? nz : nz * ny);    /* adve_calc(true, halo_x) */
  OMP_FOR
  for (sz p = 0; p < s->n_part; ++p) {
    const sz c0 = s->ijk[p];
    sz i, j = 0, k = 0;
    { const sz c = c0;
    if (s->n_dims == 1) i = c; else if (s->n_dims == 2) { i = c / nz; k = c % nz; } else { i = c / (nz * ny); j = (c / nz) % ny; k = c % nz; } }
    const sz c = c0 + halo_cells;                                   /* index into the halo-extended Courant arrays */
    /* init_grid.ipp:57-158 face maps */
    const sz lft = c, rgt = c + (s->n_dims == 3 ? nz * ny : (sz)o->nz); /* 1-D: rgt == lft as in init_grid.ipp:96-107 */
    s->x[p] = adve_1d(s->adve_scheme, s->x[p], i, s->courant_x[lft], s->courant_x[rgt], o->dx);
    if (s->n_dims > 2) {
      const sz fre = c + (c / (nz * ny)) * nz, hnd = fre + nz;
      s->y[p] = adve_1d(s->adve_scheme, s->y[p], j, s->courant_y[fre], s->courant_y[hnd], o->dy);
    }
    if (s->n_dims > 1) {
      const sz blw = s->n_dims == 2 ? c + c / nz : c + ny * (c / (nz * ny)) + (c - (c / (nz * ny)) * (nz * ny)) / nz;
      s->z[p] = adve_1d(s->adve_scheme, s->z[p], k, s->courant_z[blw], s->courant_z[blw + 1], o->dz);
    }
  }
}
static void sedi(orc_particles *s, real dt)
{
  OMP_FOR
  for (sz p = 0; p < s->n_part; ++p) s->z[p] = s->z[p] - dt * s->vt[p];
}
static void subs(orc_particles *s, real dt)
{
  const sz nz = m1(s->o.nz);
  for (sz p = 0; p < s->n_part; ++p) s->z[p] = s->z[p] - dt * s->w_LS[s->ijk[p] % nz];
}
static real periodic(real x, real a, real b) { return a + fmod((x - a) + 10 * (b - a), b - a); }
static void bcnd(orc_particles *s)
{
  if (s->n_dims == 0) return;
  const lcx_opts_init_t *o = &s->o;
  if (!distmem(s)) {
    if (!o->open_side_walls) { OMP_FOR for (sz p = 0; p < s->n_part; ++p) s->x[p] = periodic(s->x[p], o->x0, o->x1); }
    else { OMP_FOR for (sz p = 0; p < s->n_part; ++p) if (s->x[p] >= o->x1 || s->x[p] < o->x0) s->n[p] = 0; }
  } else {
    s->lft_count = s->rgt_count = 0;
    for (sz p = 0; p < s->n_part; ++p) if (s->x[p] < o->x0) s->lft_id[s->lft_count++] = p;
    for (sz p = 0; p < s->n_part; ++p) if (s->x[p] >= o->x1) s->rgt_id[s->rgt_count++] = p;
    if (o->bcond_lft == 3) for (sz i = 0; i < s->lft_count; ++i) s->n[s->lft_id[i]] = 0;
    if (o->bcond_rgt == 3) for (sz i = 0; i < s->rgt_count; ++i) s->n[s->rgt_id[i]] = 0;
  }
  if (s->n_dims == 3) {
    if (!o->open_side_walls) { OMP_FOR for (sz p = 0; p < s->n_part; ++p) s->y[p] = periodic(s->y[p], o->y0, o->y1); }
    else { OMP_FOR for (sz p = 0; p < s->n_part; ++p) if (s->y[p] >= o->y1 || s->y[p] < o->y0) s->n[p] = 0; }
  }
  if (s->n_dims > 1) {
    if (!o->periodic_topbot_walls) {
      OMP_FOR
      for (sz p = 0; p < s->n_part; ++p) if (s->z[p] >= o->z1) s->n[p] = 0;
      real liq_vol = 0, dry_vol = 0, liq_num = 0, prtcl_num = 0;
      OMP_FOR
      for (sz p = 0; p < s->n_part; ++p) s->n_filtered[p] = s->z[p] < o->z0 ? (real)s->n[p] : 0.;
#ifdef _OPENMP
      {                        /* the four sums over the SDs below the floor only, in index order: every other term is +0.0, which
                                  leaves a non-negative running sum as it is -- the same bits as the full serial loops below */
        const sz n = s->n_part;
        unsigned char *below = (unsigned char *)scr_get(SCR_FLAG, n);
        sz *idx = (sz *)scr_get(SCR_IDX, n * sizeof(sz));
        OMP_FOR
        for (sz p = 0; p < n; ++p) below[p] = s->n_filtered[p] != 0.;
        const sz m = par_select(below, n, idx);
        for (sz i = 0; i < m; ++i) { const sz p = idx[i]; liq_vol = liq_vol + 4. / 3. * ORC_PI * s->n_filtered[p] * pow(s->rw2[p], 3. / 2.); }
        for (sz i = 0; i < m; ++i) { const sz p = idx[i]; dry_vol = dry_vol + 4. / 3. * ORC_PI * s->n_filtered[p] * pow(s->rd3[p], 1.); }
        for (sz i = 0; i < m; ++i) { const sz p = idx[i]; liq_num = liq_num + (s->rw2[p] == 0. ? 0. : s->n_filtered[p]); }
        for (sz i = 0; i < m; ++i) { const sz p = idx[i]; prtcl_num = prtcl_num + s->n_filtered[p]; }
      }
#else
      for (sz p = 0; p < s->n_part; ++p) liq_vol = liq_vol + 4. / 3. * ORC_PI * s->n_filtered[p] * pow(s->rw2[p], 3. / 2.);
      for (sz p = 0; p < s->n_part; ++p) dry_vol = dry_vol + 4. / 3. * ORC_PI * s->n_filtered[p] * pow(s->rd3[p], 1.);
      for (sz p = 0; p < s->n_part; ++p) liq_num = liq_num + (s->rw2[p] == 0. ? 0. : s->n_filtered[p]);
      for (sz p = 0; p < s->n_part; ++p) prtcl_num = prtcl_num + s->n_filtered[p];
#endif
      s->puddle[LCX_OUT_LIQ_VOL] += liq_vol; s->puddle[LCX_OUT_DRY_VOL] += dry_vol;
      s->puddle[LCX_OUT_LIQ_NUM] += liq_num; s->puddle[LCX_OUT_PRTCL_NUM] += prtcl_num;
      OMP_FOR
      for (sz p = 0; p < s->n_part; ++p) if (s->z[p] < o->z0) s->n[p] = 0;
    } else { OMP_FOR for (sz p = 0; p < s->n_part; ++p) s->z[p] = periodic(s->z[p], o->z0, o->z1); }
  }
}
/* post_copy.ipp:18-35 */
/* housekeeping/particles_impl_rcyc.ipp:44-140: SDs with n == 0 are re-used as halves of the SDs with the highest
 * multiplicities.  thrust::sort_by_key of the multiplicities is taken as the stable (radix) sort the CPU backends use. */
static void rcyc(orc_particles *s)
{
  const sz N = s->n_part;
  sz n_flagged = 0;
  for (sz p = 0; p < N; ++p) n_flagged += s->n[p] == 0;
  if (n_flagged == 0) return;
  const sz n_to_rcyc = n_flagged;
  if (s->pure_const_multi) { hskpng_remove_n0(s); return; }
  s->sorted = 0;
  sz *key = s->sorted_ijk, *sid = s->sorted_id;
  for (sz p = 0; p < N; ++p) { sid[p] = p; key[p] = (sz)s->n[p]; }
  {                                                   /* stable LSD radix sort of the 64-bit keys, 16 bits per pass */
    sz *k2 = NEW(sz, N), *v2 = NEW(sz, N);
    for (int pass = 0; pass < 4; ++pass) {
      sz *cnt = NEW(sz, 65537);
      const int sh = 16 * pass;
      for (sz i = 0; i < N; ++i) cnt[((key[i] >> sh) & 0xffff) + 1]++;
      for (sz c = 0; c < 65536; ++c) cnt[c + 1] += cnt[c];
      for (sz i = 0; i < N; ++i) { const sz d = cnt[(key[i] >> sh) & 0xffff]++; k2[d] = key[i]; v2[d] = sid[i]; }
      memcpy(key, k2, N * sizeof(sz)); memcpy(sid, v2, N * sizeof(sz));
      free(cnt);
    }
    free(k2); free(v2);
  }
  sz n_splittable = 0;                               /* entries behind the last multiplicity-1 SD of the sorted sequence */
  while (n_splittable < N && key[N - 1 - n_splittable] != 1) ++n_splittable;
  if (n_splittable == 0) { hskpng_remove_n0(s); return; }
  if (n_splittable < n_flagged) n_flagged = n_splittable;
  real *attrs[24]; const int na = mig_attrs(s, attrs);             /* distmem_real_vctrs: everything but n */
  for (int a = 0; a < na; ++a)
    for (sz t = 0; t < n_flagged; ++t) attrs[a][sid[t]] = attrs[a][sid[N - 1 - t]];
  for (sz t = 0; t < n_flagged; ++t) { const n_t big = s->n[sid[N - 1 - t]]; s->n[sid[t]] = big - big / 2; }
  for (sz t = 0; t < n_flagged; ++t) { const n_t big = s->n[sid[N - 1 - t]]; s->n[sid[N - 1 - t]] = big / 2; }
  if (n_flagged < n_to_rcyc) hskpng_remove_n0(s);
}
static int post_copy(orc_particles *s, const lcx_opts_t *opts)
{
  if (opts->rcyc) rcyc(s);
  else hskpng_remove_n0(s);
  hskpng_ijk(s);
  hskpng_count(s);
  return 0;
}

/* ---------------- initialisation (particles_init.ipp:16-131 and src/impl/initialization/) ---------------- */
static real eval_distro(const lcx_distro_t *d, real lnrd)
{
  if (d->fn) return d->fn(lnrd, d->user);
  real res = 0;                                   /* common/lognormal.hpp:25-37, sum of modes */
  if (d->n_modes < 0) { const real q = pow(exp(lnrd), 3) / pow(d->mean_rd[0], 3); return d->n_stp[0] * 3. * q * exp(-q); }   /* lcx.h: exponential in volume */
  for (int m = 0; m < d->n_modes; ++m)
    res += d->n_stp[m] / sqrt(2 * ORC_PI) / log(d->sdev[m]) *
           exp(-pow((lnrd - log(d->mean_rd[m])), 2) / 2. / pow(log(d->sdev[m]), 2));
  return res;
}
/* init_dist_analysis.ipp:17-77 */
static int init_dist_analysis_sd_conc(orc_particles *s, const lcx_distro_t *d, n_t sd_conc, real dt)
{
  const lcx_opts_init_t *o = &s->o;
  const real vol = s->n_dims == 0 ? s->dv[0] : (o->dx * o->dy * o->dz);
  if (o->rd_min >= 0 && o->rd_max >= 0) {
    s->multiplier = log(o->rd_max / o->rd_min) / sd_conc * dt * vol;
    s->log_rd_min = log(o->rd_min); s->log_rd_max = log(o->rd_max);
  } else if (o->rd_min < 0 && o->rd_max < 0) {
    real rd_min = 1e-14, rd_max = 1e-3;            /* config.hpp:23-24 */
    int found = 0;
    while (!found) {
      s->multiplier = log(rd_max / rd_min) / sd_conc * dt * vol;
      s->log_rd_min = log(rd_min); s->log_rd_max = log(rd_max);
      const n_t n_min = (n_t)(eval_distro(d, s->log_rd_min) * s->multiplier),
                n_max = (n_t)(eval_distro(d, s->log_rd_max) * s->multiplier);
      if (rd_min == 1e-14 && n_min != 0) FAIL("Initial dry radii distribution is non-zero (%llu) for rd_min_init (1e-14)", n_min);
      if (rd_max == 1e-3 && n_max != 0) FAIL("Initial dry radii distribution is non-zero (%llu) for rd_max_init (0.001)", n_max);
      if (n_min == 0) rd_min *= 1.01; else if (n_max == 0) rd_max /= 1.01; else found = 1;
    }
  } else FAIL("opts_init.rd_min * opts_init.rd_max < 0");
  return 0;
}
static int resize_npart(orc_particles *s)
{                                                   /* hskpng_resize.ipp:7-32; new vt = invalid */
  if (s->n_part > s->o.n_sd_max) FAIL("n_sd_max (%llu) < n_part (%zu)", s->o.n_sd_max, s->n_part);
  return 0;
}
/* init_SD_with_distros_finalize: init_kappa, init_wet (init_wet.ipp:17-78), init_xyz (init_xyz.ipp:40-74) for the SDs
 * [n_part_old, n_part) */
static void init_finalize(orc_particles *s, real kappa)
{
  const lcx_opts_init_t *o = &s->o;
  OMP_FOR
  for (sz p = s->n_part_old; p < s->n_part; ++p) s->kpa[p] = kappa;
  OMP_FOR
  for (sz p = s->n_part_old; p < s->n_part; ++p) {
    const sz c = s->ijk[p];
    s->rw2[p] = pow(rw3_eq(s->rd3[p], s->kpa[p], dmin(s->RH[c], o->RH_max), s->T[c]), 2. / 3);
  }
  const int nn[3] = {o->nx, o->ny, o->nz};
  const real a[3] = {o->x0, o->y0, o->z0}, b[3] = {o->x1, o->y1, o->z1}, dd3[3] = {o->dx, o->dy, o->dz};
  real *v[3] = {s->x, s->y, s->z};
  const sz nz = m1(o->nz), ny = m1(o->ny);
  for (int ix = 0; ix < 3; ++ix) {
    if (nn[ix] == 0) continue;
    for (sz g = 0; g < s->n_part_to_init; ++g) s->tmp_part[g] = rng_u01(&s->rng);
    OMP_FOR
    for (sz g = 0; g < s->n_part_to_init; ++g) {
      const sz p = s->n_part_old + g, c = s->ijk[p];
      sz ii;
      if (s->n_dims == 1) ii = c;
      else if (s->n_dims == 2) ii = ix == 0 ? c / nz : c % nz;
      else ii = ix == 0 ? c / (nz * ny) : ix == 1 ? (c / nz) % ny : c % nz;
      const real u = s->tmp_part[g];
      v[ix][p] = u * dmin(b[ix], (ii + 1) * dd3[ix]) + (1. - u) * dmax(a[ix], ii * dd3[ix]);
    }
  }
}

/* ---- constant-multiplicity and large-tail initialisation (init_SD_with_distros_const_multi.ipp, ..._tail.ipp) ---- */
/* Brent's minimiser: the reference calls boost::math::tools::brent_find_minima (init_dist_analysis.ipp:95; Boost is NOT
 * vendored and its version is not pinned by the reference: parity UNPINNED for this routine).  Restated from the
 * published algorithm (R. P. Brent, Algorithms for Minimization without Derivatives, 1973, ch. 5) in the form Boost
 * documents: golden ratio 0.3819660, tolerance 2^(1-bits) with bits = min(digits/2, requested). */
typedef real (*orc_fn1)(real, void *);
static real brent_find_minimum(orc_fn1 f, void *ctx, real min, real max, int bits, uintmax_t *max_iter, real *fmin)
{
  if (bits > 53 / 2) bits = 53 / 2;
  const real tolerance = ldexp(1.0, 1 - bits), golden = 0.3819660f;
  real x, w, v, u, delta, delta2, fu, fv, fw, fx, mid, fract1, fract2;
  x = w = v = max;
  fw = fv = fx = f(x, ctx);
  delta2 = delta = 0;
  uintmax_t count = *max_iter;
  do {
    mid = (min + max) / 2;
    fract1 = tolerance * fabs(x) + tolerance / 4;
    fract2 = 2 * fract1;
    if (fabs(x - mid) <= (fract2 - (max - min) / 2)) break;
    if (fabs(delta2) > fract1) {
      real r = (x - w) * (fx - fv), q = (x - v) * (fx - fw), p = (x - v) * q - (x - w) * r;
      q = 2 * (q - r);
      if (q > 0) p = -p;
      q = fabs(q);
      const real td = delta2;
      delta2 = delta;
      if ((fabs(p) >= fabs(q * td / 2)) || (p <= q * (min - x)) || (p >= q * (max - x))) {
        delta2 = (x >= mid) ? min - x : max - x;
        delta = golden * delta2;
      } else {
        delta = p / q;
        u = x + delta;
        if (((u - min) < fract2) || ((max - u) < fract2)) delta = (mid - x) < 0 ? -fabs(fract1) : fabs(fract1);
      }
    } else {
      delta2 = (x >= mid) ? min - x : max - x;
      delta = golden * delta2;
    }
    u = (fabs(delta) >= fract1) ? x + delta : (delta > 0 ? x + fabs(fract1) : x - fabs(fract1));
    fu = f(u, ctx);
    if (fu <= fx) {
      if (u >= x) min = x; else max = x;
      v = w; w = x; x = u; fv = fw; fw = fx; fx = fu;
    } else {
      if (u < x) min = u; else max = u;
      if ((fu <= fw) || (w == x)) { v = w; w = u; fv = fw; fw = fu; }
      else if ((fu <= fv) || (v == x) || (v == w)) { v = u; fv = fu; }
    }
  } while (--count);
  *max_iter -= count;
  *fmin = fx;
  return x;
}
typedef struct { const lcx_distro_t *d; real mul, add; } distro_ctx;
static real distro_mul_add(real lnrd, void *vc) { const distro_ctx *c = (const distro_ctx *)vc; return eval_distro(c->d, lnrd) * c->mul + c->add; }
/* init_dist_analysis.ipp:80-120 */
static int init_dist_analysis_const_multi(orc_particles *s, const lcx_distro_t *d)
{
  const lcx_opts_init_t *o = &s->o;
  if (o->rd_min >= 0 && o->rd_max >= 0) { s->log_rd_min = log(o->rd_min); s->log_rd_max = log(o->rd_max); }
  else if (o->rd_min < 0 && o->rd_max < 0) {
    distro_ctx neg = {d, -1., 0.};
    uintmax_t n_iter = 100;
    real fmin;
    const real lnrd_max = brent_find_minimum(distro_mul_add, &neg, log(1e-14), log(1e-3), 200, &n_iter, &fmin);
    const real bound = -fmin / 1e20;                         /* config.hpp:21 threshold */
    distro_ctx lvl = {d, 1., -bound};
    n_iter = 100;
    s->log_rd_min = orc_toms748(distro_mul_add, &lvl, log(1e-14), lnrd_max, distro_mul_add(log(1e-14), &lvl), distro_mul_add(lnrd_max, &lvl),
                                s->eps_tol, &n_iter);
    n_iter = 100;
    s->log_rd_max = orc_toms748(distro_mul_add, &lvl, lnrd_max, log(1e-3), distro_mul_add(lnrd_max, &lvl), distro_mul_add(log(1e-3), &lvl),
                                s->eps_tol, &n_iter);
  } else FAIL("opts_init.rd_min * opts_init.rd_max < 0");
  return 0;
}
/* init_count_num.ipp:14-24,41-101 + init_ijk + init_dry_const_multi.ipp:20-80 + init_n_const_multi: SDs of multiplicity
 * const_multi whose dry radii are drawn from the CDF of the spectrum on [log_rd_min, log_rd_max] */
static int init_const_multi_like(orc_particles *s, const lcx_distro_t *d, n_t const_multi)
{
  const lcx_opts_init_t *o = &s->o;
  const real bin = 1e-4, lo = s->log_rd_min, hi = s->log_rd_max;         /* config.hpp:20 bin_precision */
  const int nb = (int)((hi - lo) / bin);
  real integral = (eval_distro(d, lo) + eval_distro(d, hi)) / 2.;
  for (int i = 1; i < nb; ++i) integral += eval_distro(d, lo + i * bin);
  integral = integral * bin;
  sz total = 0;
  for (sz c = 0; c < s->n_cell; ++c) {                                     /* init_count_num_hlpr + conc_to_number */
    real conc = integral;
    conc = conc * s->dv[c];
    if (!o->aerosol_independent_of_rhod) conc = s->rhod[c] / rho_stp * conc;
    if (o->n_aerosol_conc_factor > 0) conc = conc * s->aerosol_conc_factor[c % o->nz];
    s->count_num[c] = (n_t)(conc / const_multi + 0.5);
    total += (sz)s->count_num[c];
  }
  s->n_part_old = s->n_part;
  s->n_part_to_init = total;
  s->n_part += total;
  if (resize_npart(s)) return 1;
  for (sz p = s->n_part_old; p < s->n_part; ++p) { s->vt[p] = -1.; if (s->use_rc2) s->rc2[p] = -1.; }
  { sz w = s->n_part_old; for (sz c = 0; c < s->n_cell; ++c) for (n_t q = 0; q < s->count_num[c]; ++q) s->ijk[w++] = c; }
  const sz ncdf = (sz)((hi - lo) / bin + 1);
  real *cdf = NEW(real, ncdf);
  for (sz i = 0; i < ncdf; ++i) cdf[i] = eval_distro(d, lo + bin * i) * 1;
  for (sz i = 1; i < ncdf; ++i) cdf[i] = cdf[i - 1] + cdf[i];
  { const real back = cdf[ncdf - 1]; for (sz i = 0; i < ncdf; ++i) cdf[i] = cdf[i] / back; }
  for (sz g = 0; g < total; ++g) s->tmp_part[g] = rng_u01(&s->rng);
  for (sz g = 0; g < total; ++g) {
    const real u = s->tmp_part[g];
    sz a = 0, b = ncdf;                                                    /* thrust::upper_bound: first index with cdf > u */
    while (a < b) { const sz m = a + (b - a) / 2; if (!(u < cdf[m])) a = m + 1; else b = m; }
    const real lnrd = lo + (real)a * bin;
    s->rd3[s->n_part_old + g] = exp(3 * lnrd);
  }
  free(cdf);
  for (sz p = s->n_part_old; p < s->n_part; ++p) s->n[p] = const_multi;
  return 0;
}
static int init_SD_with_distros(orc_particles *s)
{
  const lcx_opts_init_t *o = &s->o;
  real tot_lnrd_rng = 0.;
  if (o->sd_conc > 0)
    for (int d = 0; d < o->n_dry_distros; ++d) {
      if (init_dist_analysis_sd_conc(s, &s->distros[d], o->sd_conc, 1.)) return 1;
      tot_lnrd_rng += s->log_rd_max - s->log_rd_min;
    }
  for (int d = 0; d < o->n_dry_distros; ++d) {
    const lcx_distro_t *dd = &s->distros[d];
    if (o->sd_conc > 0) {
    /* init_SD_with_distros_sd_conc.ipp:14-46 */
    if (init_dist_analysis_sd_conc(s, dd, o->sd_conc, 1.)) return 1;
    if (s->log_rd_min >= s->log_rd_max) FAIL("Distribution analysis error: rd_min(%g) >= rd_max(%g)", exp(s->log_rd_min), exp(s->log_rd_max));
    const real fraction = (s->log_rd_max - s->log_rd_min) / tot_lnrd_rng;
    s->multiplier *= o->sd_conc / (n_t)(int)(fraction * o->sd_conc + 0.5);
    const n_t per_cell = (n_t)(fraction * o->sd_conc);               /* init_count_num.ipp:32-35 */
    for (sz c = 0; c < s->n_cell; ++c) s->count_num[c] = per_cell;
    s->n_part_old = s->n_part;
    s->n_part_to_init = (sz)per_cell * s->n_cell;
    s->n_part += s->n_part_to_init;
    if (resize_npart(s)) return 1;
    OMP_FOR
    for (sz p = s->n_part_old; p < s->n_part; ++p) { s->vt[p] = -1.; if (s->use_rc2) s->rc2[p] = -1.; }
    /* init_ijk.ipp:36-52 */
    OMP_FOR
    for (sz g = 0; g < s->n_part_to_init; ++g) s->ijk[s->n_part_old + g] = g / (sz)per_cell;
    /* init_dry_sd_conc.ipp:43-86 */
    for (sz g = 0; g < s->n_part_to_init; ++g) s->tmp_part[g] = rng_u01(&s->rng);
    OMP_FOR
    for (sz g = 0; g < s->n_part_to_init; ++g) {
      const sz c = s->ijk[s->n_part_old + g];
      const sz ptr = (sz)per_cell * c;
      const real lnrd = s->log_rd_min + ((real)(g - ptr) + s->tmp_part[g]) * (s->log_rd_max - s->log_rd_min) / (real)s->count_num[c];
      s->rd3[s->n_part_old + g] = exp(3 * lnrd);
    }
    /* init_n.ipp:48-143 (a distribution given as a function pointer may be a Python callback: those stay on one thread) */
#pragma omp parallel for schedule(static) if (dd->fn == NULL)
    for (sz g = 0; g < s->n_part_to_init; ++g) {
      const sz p = s->n_part_old + g, c = s->ijk[p];
      const real lnrd = log(s->rd3[p]) / 3.;
      real v = s->multiplier * eval_distro(dd, lnrd);
      if (!o->aerosol_independent_of_rhod) v = v * s->rhod[c] / rho_stp;
      if (o->n_aerosol_conc_factor > 0) v = v * s->aerosol_conc_factor[c % o->nz];
      if (s->n_dims > 0) v = v * s->dv[c] / (o->dx * o->dy * o->dz);
      s->n[p] = (n_t)(v + 0.5);
    }
    init_finalize(s, dd->kappa);
    if (o->sd_conc_large_tail) {                                            /* init_SD_with_distros_tail.ipp:14-40 */
      const real log_rd_min_init = s->log_rd_max;
      if (init_dist_analysis_const_multi(s, dd)) return 1;
      s->log_rd_min = log_rd_min_init;
      if (s->log_rd_min >= s->log_rd_max) FAIL("Distribution analysis error: rd_min(%g) >= rd_max(%g)", exp(s->log_rd_min), exp(s->log_rd_max));
      if (init_const_multi_like(s, dd, 1)) return 1;
      init_finalize(s, dd->kappa);
    }
    }
    if (o->sd_const_multi > 0) {                                            /* init_SD_with_distros_const_multi.ipp:14-38 */
      if (init_dist_analysis_const_multi(s, dd)) return 1;
      if (s->log_rd_min >= s->log_rd_max) FAIL("Distribution analysis error: rd_min(%g) >= rd_max(%g)", exp(s->log_rd_min), exp(s->log_rd_max));
      if (init_const_multi_like(s, dd, o->sd_const_multi)) return 1;
      init_finalize(s, dd->kappa);
    }
  }
  return 0;
}
/* init_SD_with_sizes.ipp:14-77, init_count_num.ipp:41-70,88-92 (conc_to_number), init_n.ipp:130-143, init_dry_dry_sizes.ipp:14-20 */
static int init_SD_with_sizes(orc_particles *s)
{
  const lcx_opts_init_t *o = &s->o;
  for (int d = 0; d < o->n_dry_sizes; ++d) {
    const lcx_dry_size_t *ds = &s->sizes[d];
    const n_t per_cell = (n_t)ds->sd_count;
    for (sz c = 0; c < s->n_cell; ++c) s->count_num[c] = per_cell;
    s->n_part_old = s->n_part;
    s->n_part_to_init = (sz)per_cell * s->n_cell;
    s->n_part += s->n_part_to_init;
    if (resize_npart(s)) return 1;
    { sz w = s->n_part_old; for (sz c = 0; c < s->n_cell; ++c) for (n_t q = 0; q < per_cell; ++q) s->ijk[w++] = c; }
    const real rad3 = ds->radius * ds->radius * ds->radius;
    for (sz p = s->n_part_old; p < s->n_part; ++p) { s->rd3[p] = rad3; s->kpa[p] = ds->kappa; s->vt[p] = -1.; if (s->use_rc2) s->rc2[p] = -1.; }
    for (sz p = s->n_part_old; p < s->n_part; ++p) {
      const sz c = s->ijk[p];
      real conc = ds->conc;
      conc = conc * s->dv[c];
      if (!o->aerosol_independent_of_rhod) conc = s->rhod[c] / rho_stp * conc;
      if (o->n_aerosol_conc_factor > 0) conc = conc * s->aerosol_conc_factor[c % o->nz];
      s->n[p] = (n_t)(conc / (sz)ds->sd_count + .5);
    }
    for (sz p = s->n_part_old; p < s->n_part; ++p) {
      const sz c = s->ijk[p];
      s->rw2[p] = pow(rw3_eq(s->rd3[p], s->kpa[p], dmin(s->RH[c], o->RH_max), s->T[c]), 2. / 3);
    }
    const int nn[3] = {o->nx, o->ny, o->nz};
    const real a[3] = {o->x0, o->y0, o->z0}, b[3] = {o->x1, o->y1, o->z1}, dd3[3] = {o->dx, o->dy, o->dz};
    real *v[3] = {s->x, s->y, s->z};
    const sz nz = m1(o->nz), ny = m1(o->ny);
    for (int ix = 0; ix < 3; ++ix) {
      if (nn[ix] == 0) continue;
      for (sz g = 0; g < s->n_part_to_init; ++g) s->tmp_part[g] = rng_u01(&s->rng);
      for (sz g = 0; g < s->n_part_to_init; ++g) {
        const sz p = s->n_part_old + g, c = s->ijk[p];
        sz ii;
        if (s->n_dims == 1) ii = c;
        else if (s->n_dims == 2) ii = ix == 0 ? c / nz : c % nz;
        else ii = ix == 0 ? c / (nz * ny) : ix == 1 ? (c / nz) % ny : c % nz;
        const real u = s->tmp_part[g];
        v[ix][p] = u * dmin(b[ix], (ii + 1) * dd3[ix]) + (1. - u) * dmax(a[ix], ii * dd3[ix]);
      }
    }
  }
  return 0;
}
/* init_grid.ipp:14-54 */
static void init_grid(orc_particles *s)
{
  const lcx_opts_init_t *o = &s->o;
  if (s->n_dims == 0) return;
  const int nz = m1(o->nz), ny = m1(o->ny);
  for (sz c = 0; c < s->n_cell; ++c) {
    const int ijk = (int)c;
    const int i = (ijk / nz) / ny, j = (ijk / nz) % ny, k = ijk % nz;
    s->dv[c] = dmax(0., (dmin((i + 1) * o->dx, o->x1) - dmax(i * o->dx, o->x0)) *
                        (dmin((j + 1) * o->dy, o->y1) - dmax(j * o->dy, o->y0)) *
                        (dmin((k + 1) * o->dz, o->z1) - dmax(k * o->dz, o->z0)));
  }
}
#include "orc_tables.h"
/* init_kernel.ipp:6-233 */
static int init_kernel(orc_particles *s)
{
  const lcx_opts_init_t *o = &s->o;
  switch (o->kernel) {
    case LCX_KERNEL_GEOMETRIC:
      if (s->n_user_params > 1) FAIL("Not more than 1 parameter is required by the geometric kernel, %d given", s->n_user_params);
      return 0;
    case LCX_KERNEL_GOLOVIN:
      if (s->n_user_params != 1) FAIL("Golovin kernel accepts exactly one parameter, %d given", s->n_user_params);
      return 0;
    case LCX_KERNEL_LONG:
      if (s->n_user_params != 0) FAIL("Long kernel doesn't accept parameters, %d given", s->n_user_params);
      return 0;
    default: {
      const int onishi = o->kernel == LCX_KERNEL_ONISHI_HALL || o->kernel == LCX_KERNEL_ONISHI_HALL_DAVIS_NO_WAALS;
      if (onishi) {                                                                          /* init_kernel.ipp:183-231 */
        if (s->n_user_params != 1) FAIL("libcloudph++: Please supply one kernel parameter: Taylor microscale Reynolds number.");
        if (!o->turb_coal_switch) FAIL("libcloudph++: To use the turbulent Onishis kernel, set turb_coal_switch=True");
      } else if (s->n_user_params != 0) FAIL("this kernel doesn't accept parameters");
      sz n = 0; real r_max = 0;
      const int eff = o->kernel == LCX_KERNEL_ONISHI_HALL ? LCX_KERNEL_HALL :
                      o->kernel == LCX_KERNEL_ONISHI_HALL_DAVIS_NO_WAALS ? LCX_KERNEL_HALL_DAVIS_NO_WAALS : o->kernel;
      dbl r_max_d = 0;
      const dbl *tab = orc_efficiency_table(eff, &n, &r_max_d);      /* (the tables are data: doubles in every flavour) */
      r_max = (real)r_max_d;
      if (!tab) FAIL("libcloudph++: kernel %d not available in this backend", o->kernel);
      real *kp = NEW(real, n + s->n_user_params);                    /* user parameters first, then the efficiencies */
      for (int i = 0; i < s->n_user_params; ++i) kp[i] = s->kernel_parameters[i];
      for (sz i = 0; i < n; ++i) kp[s->n_user_params + i] = (real)tab[i];
      free(s->kernel_parameters);
      s->kernel_parameters = kp;
      s->n_kernel_parameters = n + s->n_user_params; s->kernel_r_max = r_max;
      return 0;
    }
  }
}
/* init_sanity_check.ipp (subset relevant to the supported options) */
static int init_sanity_check(orc_particles *s, const lcx_arrinfo_t *th, const lcx_arrinfo_t *rv, const lcx_arrinfo_t *rhod,
                             const lcx_arrinfo_t *p, const lcx_arrinfo_t *cx, const lcx_arrinfo_t *cy, const lcx_arrinfo_t *cz)
{
  const lcx_opts_init_t *o = &s->o;
  if (s->init_called) FAIL("libcloudph++: init() may be called just once");
  s->init_called = 1;
  if (arr_null(th) || arr_null(rv) || arr_null(rhod)) FAIL("libcloudph++: passing th, rv and rhod is mandatory");
  if (!arr_null(cx) || !arr_null(cy) || !arr_null(cz)) {
    if (s->n_dims == 0) FAIL("libcloudph++: Courant numbers passed in 0D setup");
    if (s->n_dims == 1 && (arr_null(cx) || !arr_null(cy) || !arr_null(cz))) FAIL("libcloudph++: Only X Courant number allowed in 1D setup");
    if (s->n_dims == 2 && (arr_null(cx) || !arr_null(cy) || arr_null(cz))) FAIL("libcloudph++: Only X and Z Courant numbers allowed in 2D setup");
    if (s->n_dims == 3 && (arr_null(cx) || arr_null(cy) || arr_null(cz))) FAIL("libcloudph++: All XYZ Courant number components required in 3D setup");
  }
  if (o->n_dry_distros == 0 && o->n_dry_sizes == 0) FAIL("libcloudph++: Both dry_distros and dry_sizes are undefined");
  if (s->n_dims > 0) {
    if (!(o->x0 >= 0 && o->x0 < m1(o->nx) * o->dx)) FAIL("libcloudph++: !(x0 >= 0 & x0 < min(1,nx)*dz)");
    if (!(o->y0 >= 0 && o->y0 < m1(o->ny) * o->dy)) FAIL("libcloudph++: !(y0 >= 0 & y0 < min(1,ny)*dy)");
    if (!(o->z0 >= 0 && o->z0 < m1(o->nz) * o->dz)) FAIL("libcloudph++: !(z0 >= 0 & z0 < min(1,nz)*dz)");
    if (!(o->y1 > o->y0 && o->y1 <= m1(o->ny) * o->dy)) FAIL("libcloudph++: !(y1 > y0 & y1 <= min(1,ny)*dy)");
    if (!(o->z1 > o->z0 && o->z1 <= m1(o->nz) * o->dz)) FAIL("libcloudph++: !(z1 > z0 & z1 <= min(1,nz)*dz)");
  }
  if (o->dt == 0) FAIL("libcloudph++: please specify opts_init.dt");
  if (o->sd_conc * o->sd_const_multi != 0) FAIL("libcloudph++: specify either opts_init.sd_conc or opts_init.sd_const_multi, not both");
  if (o->sd_conc == 0 && o->sd_const_multi == 0 && o->n_dry_sizes == 0) FAIL("libcloudph++: please specify opts_init.sd_conc, opts_init.sd_const_multi or opts_init.dry_sizes");
  if (o->coal_switch) {
    if (o->terminal_velocity == LCX_VT_UNDEFINED) FAIL("libcloudph++: please specify opts_init.terminal_velocity or turn off opts_init.coal_switch");
    if (o->kernel == LCX_KERNEL_UNDEFINED) FAIL("libcloudph++: please specify opts_init.kernel");
  }
  if (o->sedi_switch && o->terminal_velocity == LCX_VT_UNDEFINED) FAIL("libcloudph++: please specify opts_init.terminal_velocity or turn off opts_init.sedi_switch");
  if (o->sedi_switch && o->nz == 0) FAIL("libcloudph++: opts_init.sedi_switch can be True only if n_dims > 1");
  if (o->subs_switch && o->nz == 0) FAIL("libcloudph++: opts_init.subs_switch can be True only if n_dims > 1");
  if (o->subs_switch && o->nz != o->n_w_LS) FAIL("libcloudph++: opts_init.subs_switch == True, but subsidence velocity profile size != nz");
  if (o->n_aerosol_conc_factor && s->n_dims < 2) FAIL("libcloudph++: aerosol_conc_factor can only be used in 2D and 3D");
  if (o->n_aerosol_conc_factor && o->nz != o->n_aerosol_conc_factor) FAIL("libcloudph++: aerosol_conc_factor size needs to be either 0 or nz");
  if (o->n_aerosol_conc_factor && !o->aerosol_independent_of_rhod) FAIL("libcloudph++: aerosol_conc_factor can only be used if aerosol_independent_of_rhod==true");
  if (o->const_p && arr_null(p)) FAIL("libcloudph++: In const_p option, pressure profile must be passed (p in init())");
  if (!o->const_p && !arr_null(p)) FAIL("libcloudph++: pressure profile was passed in init(), but the constant pressure option was not used");
  if (o->sstp_cond < 1) FAIL("libcloudph++: opts_init.sstp_cond needs to be greater than 0");
  if (o->turb_adve_switch && o->nz == 0) FAIL("libcloudph++: opts_init.turb_adve_switch can be True only if n_dims > 1");
  if (o->turb_cond_switch && o->nz == 0) FAIL("libcloudph++: opts_init.turb_cond_switch can be True only if n_dims > 1");
  if ((o->turb_adve_switch || o->turb_cond_switch) && o->nz != o->n_SGS_mix_len) FAIL("libcloudph++: at least one of opts_init.turb_adve_switch, opts_init.turb_cond_switch is true, but SGS mixing length profile size != nz");
  for (int k = 0; k < o->n_SGS_mix_len; ++k) if (s->SGS_mix_len && s->SGS_mix_len[k] <= 0) FAIL("libcloudph++: SGS_mix_len <= 0");
  if (o->adaptive_sstp_cond && !o->exact_sstp_cond) FAIL("libcloudph++: Adaptive condensation substepping (opts_init.adaptive_sstp_cond) works oly for per-particle substepping (opts_init.exact_sstp_cond)");
  if (!o->sstp_cond_mix && !o->exact_sstp_cond) FAIL("libcloudph++: Mixing of rv and th (opts_init.sstp_cond_mix) can only be disable for per-particle substepping (opts_init.exact_sstp_cond)");
  if (o->sstp_cond_mix && o->adaptive_sstp_cond && o->exact_sstp_cond) FAIL("libcloudph++: Adaptive cond substepping (opts_init.adaptive_sstp_cond) with per-particle substepping (opts_init.exact_sstp_cond) requires mixing of th and rv between subteps (opts_init.sstp_cond_mix) to be disabled");
  if (o->sstp_cond_act > 1 && (o->sstp_cond_mix || !o->exact_sstp_cond || !o->adaptive_sstp_cond)) FAIL("libcloudph++: number of substeps for activation (opts_init.sstp_cond_act) can be greater than 1 only if mixing of rv and th (opts_init.sstp_cond_mix) is disabled and if per-particle condensation substepping is used (opts_init.exact_sstp_cond) and if adaptive substepping is used (opts_init.adaptive_sstp_cond)");
  return 0;
}
static void alloc_courants(orc_particles *s)
{                                                   /* init_sync.ipp:28-44, particles_impl.ipp:413-431 */
  const lcx_opts_init_t *o = &s->o;
  const int nxh = o->nx + 2 * s->halo;
  switch (s->n_dims) {
    case 3: s->n_cx = (sz)(nxh + 1) * o->ny * o->nz; s->n_cy = (sz)nxh * (o->ny + 1) * o->nz; s->n_cz = (sz)nxh * o->ny * (o->nz + 1); break;
    case 2: s->n_cx = (sz)(nxh + 1) * o->nz; s->n_cz = (sz)nxh * (o->nz + 1); break;
    case 1: s->n_cx = (sz)nxh + 1; break;
    default: break;
  }
  s->courant_x = NEW(real, s->n_cx); s->courant_y = NEW(real, s->n_cy); s->courant_z = NEW(real, s->n_cz);
}
int orc_init(orc_particles *s, const lcx_arrinfo_t *th, const lcx_arrinfo_t *rv, const lcx_arrinfo_t *rhod,
             const lcx_arrinfo_t *p, const lcx_arrinfo_t *cx, const lcx_arrinfo_t *cy, const lcx_arrinfo_t *cz)
{
  if (init_sanity_check(s, th, rv, rhod, p, cx, cy, cz)) return 1;
  if (s->o.rng_seed_init_switch) mt_seed(&s->rng, (uint32_t)s->o.rng_seed_init);
  alloc_courants(s);
  sync_in_arr(s, th, s->th, s->n_cell, 0, 0, 0);
  sync_in_arr(s, rv, s->rv, s->n_cell, 0, 0, 0);
  sync_in_arr(s, rhod, s->rhod, s->n_cell, 0, 0, 0);
  sync_in_arr(s, p, s->p, s->n_cell, 0, 0, 0);
  sync_in_courant(s, cx, s->courant_x, s->n_cx, 1, 0, 0);
  sync_in_courant(s, cy, s->courant_y, s->n_cy, 0, 1, 0);
  sync_in_courant(s, cz, s->courant_z, s->n_cz, 0, 0, 1);
  init_grid(s);
  hskpng_Tpr(s);
  if (!s->o.no_ccn_at_init) {
    if (s->o.n_dry_distros > 0 && init_SD_with_distros(s)) return 1;
    if (s->o.n_dry_sizes > 0 && init_SD_with_sizes(s)) return 1;
  }
  if (s->o.coal_switch && init_kernel(s)) return 1;
  init_vterm(s);
  hskpng_vterm(s, 1);
  hskpng_approximate_rc2_invalid(s);         /* particles_init.ipp:116-117 */
  sstp_save(s);
  hskpng_count(s);
  mt_seed(&s->rng, (uint32_t)s->o.rng_seed);
  if (s->tag) for (sz p = 0; p < s->n_part; ++p) s->tag[p] = (real)p;
  return 0;
}

/* ---------------- time stepping (particles_step.ipp) ---------------- */
int orc_sync_in(orc_particles *s, const lcx_arrinfo_t *th, const lcx_arrinfo_t *rv, const lcx_arrinfo_t *rhod,
                const lcx_arrinfo_t *cx, const lcx_arrinfo_t *cy, const lcx_arrinfo_t *cz, const lcx_arrinfo_t *diss)
{
  if (!s->init_called) FAIL("libcloudph++: please call init() before calling step_sync()");
  if (s->should_now_run_async) FAIL("libcloudph++: please call step_async() before calling step_sync() again");
  if (arr_null(th) || arr_null(rv)) FAIL("libcloudph++: passing th and rv is mandatory");
  if (!arr_null(cx) || !arr_null(cy) || !arr_null(cz)) {
    if (s->n_dims == 0) FAIL("libcloudph++: Courant numbers passed in 0D setup");
    if (s->n_dims == 1 && (arr_null(cx) || !arr_null(cy) || !arr_null(cz))) FAIL("libcloudph++: Only X Courant number allowed in 1D setup");
    if (s->n_dims == 2 && (arr_null(cx) || !arr_null(cy) || arr_null(cz))) FAIL("libcloudph++: Only X and Z Courant numbers allowed in 2D setup");
    if (s->n_dims == 3 && (arr_null(cx) || arr_null(cy) || arr_null(cz))) FAIL("libcloudph++: All XYZ Courant number components required in 3D setup");
  }
  { const int turb = s->o.turb_adve_switch || s->o.turb_cond_switch || s->o.turb_coal_switch;              /* particles_step.ipp:74-78 */
    if (turb && arr_null(diss)) FAIL("libcloudph++: turbulent advection, coalescence and condesation are not switched off and diss_rate is empty");
    if (!turb && !arr_null(diss)) FAIL("libcloudph++: turbulent advection, coalescence and condesation are switched off and diss_rate is not empty"); }
  s->var_rho = !arr_null(rhod);
  sync_in_arr(s, th, s->th, s->n_cell, 0, 0, 0);
  sync_in_arr(s, rv, s->rv, s->n_cell, 0, 0, 0);
  sync_in_arr(s, rhod, s->rhod, s->n_cell, 0, 0, 0);
  if (s->diss_rate) sync_in_arr(s, diss, s->diss_rate, s->n_cell, 0, 0, 0);
  sync_in_courant(s, cx, s->courant_x, s->n_cx, 1, 0, 0);
  sync_in_courant(s, cy, s->courant_y, s->n_cy, 0, 1, 0);
  sync_in_courant(s, cz, s->courant_z, s->n_cz, 0, 0, 1);
  /* particles_step.ipp:127-142: Courant numbers beyond the 2-cell halo break the predictor-corrector: first order this step */
  if (s->o.adve_scheme == LCX_ADVE_PRED_CORR && !arr_null(cx))
    for (sz c = 0; c < s->n_cx; ++c) if (s->courant_x[c] < -2. || s->courant_x[c] > 2.) { s->adve_scheme = LCX_ADVE_EULER; break; }
  s->should_now_run_cond = 1;
  return 0;
}
int orc_step_cond(orc_particles *s, const lcx_opts_t *opts, const lcx_arrinfo_t *th, const lcx_arrinfo_t *rv)
{
  if (!s->should_now_run_cond) FAIL("libcloudph++: please call sync_in() before calling step_cond()");
  if (opts->turb_cond && !s->o.turb_cond_switch) FAIL("libcloudph++: turb_cond_swtich=False, but turb_cond==True");
  s->should_now_run_cond = 0;
  if (adjust_timesteps(s, opts->dt)) return 1;
  if (s->o.diag_incloud_time)                                  /* update_incloud_time.ipp:36-66: += dt while rw2 > rc2, else 0 */
    for (sz p = 0; p < s->n_part; ++p) {
      const real rc2 = pow(rw3_cr(s->rd3[p], s->kpa[p], s->T[s->ijk[p]]), 2. / 3);
      if (s->rw2[p] > rc2) s->ict[p] += s->dt; else s->ict[p] = 0;
    }
  if (opts->cond) {
    TMR(TM_SORT, hskpng_sort(s));
    TMR(TM_TPR, hskpng_mfp(s));
    if (s->o.exact_sstp_cond && (s->sstp_cond > 1 || s->sstp_cond_act > 1)) cond_perparticle(s, opts->RH_max, opts->turb_cond);
    else for (int step = 0; step < s->sstp_cond; ++step) {
      TMR(TM_OTHER, sstp_percell_step(s, step));
      if (opts->turb_cond) for (sz p = 0; p < s->n_part; ++p) s->ssp[p] = s->ssp[p] + s->dt / s->sstp_cond * s->dot_ssp[p];   /* apply_perparticle_sgs_supersat.ipp */
      TMR(TM_TPR, hskpng_Tpr(s));
      if (step == 0) TMR(TM_MOMS, save_liq_before(s));
      TMR(TM_COND, cond(s, s->dt, opts->RH_max, step, opts->turb_cond));
      TMR(TM_MOMS, update_th_rv(s));
    }
    TMR(TM_OTHER, sstp_save(s));
    sync_out_arr(s, s->th, th, s->n_cell);
    sync_out_arr(s, s->rv, rv, s->n_cell);
  }
  s->should_now_run_async = 1;
  s->selected_before_counting = 0;
  return 0;
}
int orc_step_sync(orc_particles *s, const lcx_opts_t *opts, const lcx_arrinfo_t *th, const lcx_arrinfo_t *rv,
                  const lcx_arrinfo_t *rhod, const lcx_arrinfo_t *cx, const lcx_arrinfo_t *cy,
                  const lcx_arrinfo_t *cz, const lcx_arrinfo_t *diss)
{
  if (orc_sync_in(s, th, rv, rhod, cx, cy, cz, diss)) return 1;
  return orc_step_cond(s, opts, th, rv);
}
int orc_step_async(orc_particles *s, const lcx_opts_t *opts)
{
  if (!s->should_now_run_async) FAIL("libcloudph++: please call step_sync() before calling step_async() again");
  s->should_now_run_async = 0;
  if (opts->chem_dsl || opts->chem_dsc || opts->chem_rct) FAIL("libcloudph++: all chemistry was switched off in opts_init");
  if (opts->coal && !s->o.coal_switch) FAIL("libcloudph++: coalescence was switched off in opts_init");
  if (opts->sedi && !s->o.sedi_switch) FAIL("libcloudph++: sedimentation was switched off in opts_init");
  if (opts->subs && !s->o.subs_switch) FAIL("libcloudph++: subsidence was switched off in opts_init");
  if (opts->turb_adve && !s->o.turb_adve_switch) FAIL("libcloudph++: turb_adve_switch=False, but turb_adve==True");
  if (opts->turb_adve && s->n_dims == 0) FAIL("libcloudph++: turbulent advection does not work in 0D");
  if (opts->turb_coal && !s->o.turb_coal_switch) FAIL("libcloudph++: turb_coal_switch=False, but turb_coal==True");  /* the reference reads an empty diss_rate here */
  if (opts->src) FAIL("libcloudph++: aerosol source was switched off in opts_init");
  if (opts->rlx) FAIL("libcloudph++: aerosol relaxation was switched off in opts_init");
  if (adjust_timesteps(s, opts->dt)) return 1;
  TMR(TM_TPR, hskpng_Tpr(s));
  if (opts->sedi || opts->coal || opts->cond) TMR(TM_VTERM, hskpng_vterm(s, 0));
  if (opts->coal) {
    for (int step = 0; step < s->sstp_coal; ++step) {
      TMR(TM_COAL, coal(s, s->dt / s->sstp_coal, opts->turb_coal));
      if (step + 1 != s->sstp_coal) hskpng_vterm(s, 1);
    }
    if (s->increase_sstp_coal) { ++s->sstp_coal; s->increase_sstp_coal = 0; }
    TMR(TM_OTHER, hskpng_approximate_rc2_invalid(s));       /* particles_step.ipp:402-403 */
  }
  if (opts->turb_adve || opts->turb_cond) hskpng_tke(s);                     /* particles_step.ipp:406-427 */
  if (opts->turb_adve) hskpng_turb_vel(s, s->dt, 0);
  else if (opts->turb_cond) hskpng_turb_vel(s, s->dt, 1);
  if (opts->turb_cond) hskpng_turb_dot_ss(s);
  if (opts->adve) TMR(TM_MOVE, adve(s));
  s->adve_scheme = s->o.adve_scheme;
  if (opts->turb_adve) {                                                      /* turb_adve.ipp:13-33: x, z, y get up, wp, vp */
    real *pos[3] = {s->x, s->z, s->y}, *vel[3] = {s->up, s->wp, s->vp};
    for (int i = 0; i < s->n_dims; ++i) for (sz p = 0; p < s->n_part; ++p) pos[i][p] = pos[i][p] + vel[i][p] * s->dt;
  }
  if (opts->sedi) TMR(TM_MOVE, sedi(s, s->dt));
  if (opts->subs) subs(s, s->dt);
  TMR(TM_MOVE, bcnd(s));
  if (!distmem(s)) { int e_ = 0; TMR(TM_POST, e_ = post_copy(s, opts)); if (e_) return 1; }
  s->selected_before_counting = 0;
  return 0;
}

/* ---------------- diagnostics (particles_diag.ipp, fill_outbuf.ipp) ---------------- */
static void diag_cellfield(orc_particles *s, const real *f)
{
  hskpng_Tpr(s);
  memcpy(s->count_mom, f, s->n_cell * sizeof(real));
  s->count_n = s->n_cell;
  for (sz c = 0; c < s->n_cell; ++c) s->count_ijk[c] = c;
}
/* particles_diag.ipp:499-555: divergence of the Courant field per cell, y then z then x differences, each / opts_init.dt */
int orc_diag_vel_div(orc_particles *s)
{
  if (s->n_dims == 0) return 0;
  const sz nz = m1(s->o.nz), ny = m1(s->o.ny);
  const sz plane = s->n_dims == 1 ? 1 : s->n_dims == 2 ? nz : nz * ny;
  for (sz c = 0; c < s->n_cell; ++c) {
    real Cxl, Cxr, Cyl = 0, Cyr = 0, Czl = 0, Czr = 0, d = 0.;
    faces(s, c + (sz)s->halo * plane, &Cxl, &Cxr, &Cyl, &Cyr, &Czl, &Czr);
    if (s->n_dims == 3) d = d + (Cyr - Cyl) / s->o.dt;
    if (s->n_dims >= 2) d = d + (Czr - Czl) / s->o.dt;
    d = d + (Cxr - Cxl) / s->o.dt;
    s->count_mom[c] = d; s->count_ijk[c] = c;
  }
  s->count_n = s->n_cell;
  return 0;
}
int orc_diag_pressure(orc_particles *s) { diag_cellfield(s, s->p); return 0; }
int orc_diag_temperature(orc_particles *s) { diag_cellfield(s, s->T); return 0; }
int orc_diag_RH(orc_particles *s) { diag_cellfield(s, s->RH); return 0; }
int orc_diag_sd_conc(orc_particles *s)
{
  if (!s->selected_before_counting) FAIL("libcloudph++: please select super-droplets (diag_all / diag_*_rng) before counting moments");
  hskpng_sort(s);
  sz cn = 0;
  for (sz p = 0; p < s->n_part; ++p) {
    const real v = s->n_filtered[s->sorted_id[p]] > 0. ? 1 : 0;
    if (p == 0 || s->sorted_ijk[p] != s->sorted_ijk[p - 1]) { s->count_ijk[cn] = s->sorted_ijk[p]; s->count_mom[cn] = v; ++cn; }
    else s->count_mom[cn - 1] += v;
  }
  s->count_n = cn;
  return 0;
}
int orc_diag_all(orc_particles *s) { moms_all(s); return 0; }
int orc_diag_water(orc_particles *s) { moms_gt0(s, s->rw2, 0); return 0; }
int orc_diag_water_cons(orc_particles *s) { moms_gt0(s, s->rw2, 1); return 0; }                  /* particles_diag.ipp:346-349 */
static int diag_sgs_mom(orc_particles *s, const real *v, int k)                                /* particles_diag.ipp:463-480 */
{
  if (!v) FAIL("libcloudph++: moment of an SGS velocity perturbation that this set-up does not carry (turb_adve_switch / turb_cond_switch, dimensions)");
  if (!s->selected_before_counting) FAIL("libcloudph++: please select super-droplets (diag_all / diag_*_rng) before counting moments");
  moms_calc(s, v, k, 1); return 0;
}
int orc_diag_up_mom(orc_particles *s, int k) { return diag_sgs_mom(s, s->o.turb_adve_switch && s->o.nx ? s->up : NULL, k); }
int orc_diag_vp_mom(orc_particles *s, int k) { return diag_sgs_mom(s, s->o.turb_adve_switch && s->o.ny ? s->vp : NULL, k); }
int orc_diag_wp_mom(orc_particles *s, int k) { return diag_sgs_mom(s, (s->o.turb_adve_switch && s->o.nz) || s->o.turb_cond_switch ? s->wp : NULL, k); }
int orc_diag_dry_rng(orc_particles *s, dbl a, dbl b) { moms_rng(s, pow(a, 3), pow(b, 3), s->rd3, 0); return 0; }
int orc_diag_wet_rng(orc_particles *s, dbl a, dbl b) { moms_rng(s, pow(a, 2), pow(b, 2), s->rw2, 0); return 0; }
int orc_diag_kappa_rng(orc_particles *s, dbl a, dbl b) { moms_rng(s, a, b, s->kpa, 0); return 0; }
int orc_diag_dry_rng_cons(orc_particles *s, dbl a, dbl b) { moms_rng(s, pow(a, 3), pow(b, 3), s->rd3, 1); return 0; }
int orc_diag_wet_rng_cons(orc_particles *s, dbl a, dbl b) { moms_rng(s, pow(a, 2), pow(b, 2), s->rw2, 1); return 0; }
int orc_diag_kappa_rng_cons(orc_particles *s, dbl a, dbl b) { moms_rng(s, a, b, s->kpa, 1); return 0; }
#define NEED_SELECTION if (!s->selected_before_counting) FAIL("libcloudph++: please select super-droplets (diag_all / diag_*_rng) before counting moments")
int orc_diag_dry_mom(orc_particles *s, int k) { NEED_SELECTION; moms_calc(s, s->rd3, k / 3., 1); return 0; }
int orc_diag_wet_mom(orc_particles *s, int k) { NEED_SELECTION; moms_calc(s, s->rw2, k / 2., 1); return 0; }
int orc_diag_kappa_mom(orc_particles *s, int k) { NEED_SELECTION; moms_calc(s, s->kpa, k, 1); return 0; }
int orc_diag_incloud_time_mom(orc_particles *s, int k)                        /* particles_diag.ipp:482-490 */
{
  if (!s->o.diag_incloud_time) FAIL("libcloudph++: diag_incloud_time_mom called, but opts_init.diag_incloud_time==false");
  NEED_SELECTION; moms_calc(s, s->ict, k, 1); return 0;
}
/* particles_diag.ipp:555-584: sum of n_filtered * rw^3 * vt per cell (not specific); refreshes vt as a side effect */
int orc_diag_precip_rate(orc_particles *s)
{
  if (!s->selected_before_counting) FAIL("libcloudph++: please select super-droplets (diag_all / diag_*_rng) before counting moments");
  hskpng_vterm(s, 0);
  for (sz p = 0; p < s->n_part; ++p) s->tmp_part[p] = pow(s->rw2[p], 3. / 2) * s->vt[p];
  moms_calc(s, s->tmp_part, 1., 0);
  return 0;
}
/* particles_diag.ipp:350-407 + moms.ipp:112-155 */
int orc_diag_RH_ge_Sc(orc_particles *s)
{
  hskpng_sort(s);
  for (sz p = 0; p < s->n_part; ++p) {
    const sz c = s->ijk[p];
    const real v = s->RH[c] - S_cr(s->rd3[p], s->kpa[p], s->T[c]);
    s->n_filtered[p] = (real)s->n[p] * (v >= 0);
  }
  s->selected_before_counting = 1;
  return 0;
}
int orc_diag_rw_ge_rc(orc_particles *s)
{
  hskpng_sort(s);
  for (sz p = 0; p < s->n_part; ++p) {
    const real rc2 = pow(rw3_cr(s->rd3[p], s->kpa[p], s->T[s->ijk[p]]), 2. / 3);
    s->n_filtered[p] = s->rw2[p] >= rc2 ? (real)s->n[p] : 0;
  }
  s->selected_before_counting = 1;
  return 0;
}
/* particles_diag.ipp:494-497, mass_dens.ipp:7-120 (the kernel width uses the number of SDs of the SD's cell) */
int orc_diag_wet_mass_dens(orc_particles *s, dbl rad, dbl sig0)
{
  NEED_SELECTION;
  hskpng_sort(s);
  hskpng_count(s);
  for (sz c = 0; c < s->n_cell; ++c) s->scl[c] = 0.;
  for (sz i = 0; i < s->count_n; ++i) s->scl[s->count_ijk[i]] = (real)s->count_num[i];
  sz cn = 0;
  for (sz q = 0; q < s->n_part; ++q) {
    const sz id = s->sorted_id[q], c = s->sorted_ijk[q];
    const real x = s->rw2[id], sig = sig0 / pow(s->scl[c], 0.2);
    const real v = s->n_filtered[id] / sig * pow(x, 3 * .5) * exp(-pow((log(pow(x, .5)) - log(rad)) / sig, 2) / 2.);
    if (q == 0 || c != s->sorted_ijk[q - 1]) { s->count_ijk[cn] = c; s->count_mom[cn] = v; ++cn; }
    else s->count_mom[cn - 1] = s->count_mom[cn - 1] + v;
  }
  s->count_n = cn;
  const real prefactor = 4. / 3. * rho_w * sqrt(ORC_PI / 2.);
  for (sz i = 0; i < cn; ++i) s->count_mom[i] = prefactor * s->count_mom[i] / s->dv[s->count_ijk[i]];
  return 0;
}
/* particles_diag.ipp:607-640: max wet radius per cell */
int orc_diag_max_rw(orc_particles *s)
{
  hskpng_sort(s);
  sz cn = 0;
  for (sz p = 0; p < s->n_part; ++p) {
    const real v = sqrt(s->rw2[s->sorted_id[p]]);
    if (p == 0 || s->sorted_ijk[p] != s->sorted_ijk[p - 1]) { s->count_ijk[cn] = s->sorted_ijk[p]; s->count_mom[cn] = v; ++cn; }
    else if (s->count_mom[cn - 1] < v) s->count_mom[cn - 1] = v;
  }
  s->count_n = cn;
  return 0;
}
int orc_outbuf(orc_particles *s, const void **data, size_t *n)
{
  for (sz c = 0; c < s->n_cell; ++c) s->outbuf[c] = 0;
  for (sz i = 0; i < s->count_n; ++i) s->outbuf[s->count_ijk[i]] = s->count_mom[i];
  hskpng_count(s);                                  /* particles_ctor.ipp:83-92 */
  *data = s->outbuf; *n = s->n_cell;
  return 0;
}
int orc_get_attr(orc_particles *s, const char *name, void *out, size_t cap, size_t *n)
{
  const real *v = !strcmp(name, "rw2") ? s->rw2 : !strcmp(name, "rd3") ? s->rd3 : !strcmp(name, "kappa") ? s->kpa :
                    !strcmp(name, "x") ? s->x : !strcmp(name, "y") ? s->y : !strcmp(name, "z") ? s->z : NULL;
  if (!v) FAIL("Unknown attribute name passed to get_attr.");
  *n = s->n_part;
  if (out) { if (cap < s->n_part) FAIL("get_attr: buffer too small"); memcpy(out, v, s->n_part * sizeof(real)); }
  return 0;
}
int orc_diag_puddle(orc_particles *s, dbl out[LCX_OUT_COUNT]) { for (int i = 0; i < LCX_OUT_COUNT; ++i) out[i] = (dbl)s->puddle[i]; return 0; }

/* ---------------- introspection hooks ---------------- */
int orc_n_part(orc_particles *s, size_t *n) { *n = s->n_part; return 0; }
int orc_n_cell(orc_particles *s, size_t *n) { *n = s->n_cell; return 0; }
int orc_real_kind(orc_particles *s, int *k) { (void)s; *k = 8; return 0; }
int orc_get_state_u64(orc_particles *s, const char *name, unsigned long long *out, size_t cap, size_t *n)
{
  sz len = s->n_part; const sz *v = NULL;
  if (!strcmp(name, "n")) { *n = len; if (out) { if (cap < len) FAIL("buffer too small"); memcpy(out, s->n, len * sizeof(n_t)); } return 0; }
  if (!strcmp(name, "ijk")) v = s->ijk; else if (!strcmp(name, "sorted_id")) v = s->sorted_id;
  else if (!strcmp(name, "sorted_ijk")) v = s->sorted_ijk;
  else if (!strcmp(name, "count_ijk")) { v = s->count_ijk; len = s->count_n; }
  else if (!strcmp(name, "count_num")) { *n = s->count_n; if (out) { if (cap < s->count_n) FAIL("buffer too small"); memcpy(out, s->count_num, s->count_n * sizeof(n_t)); } return 0; }
  else FAIL("unknown u64 state '%s'", name);
  *n = len;
  if (out) { if (cap < len) FAIL("buffer too small"); for (sz i = 0; i < len; ++i) out[i] = v[i]; }
  return 0;
}
int orc_get_state_real(orc_particles *s, const char *name, dbl *out, size_t cap, size_t *n)
{
  struct { const char *nm; const real *v; sz len; } tab[] = {
    {"vt", s->vt, s->n_part}, {"T", s->T, s->n_cell}, {"p", s->p, s->n_cell}, {"RH", s->RH, s->n_cell},
    {"eta", s->eta, s->n_cell}, {"th", s->th, s->n_cell}, {"rv", s->rv, s->n_cell}, {"rhod", s->rhod, s->n_cell},
    {"dv", s->dv, s->n_cell}, {"lambda_D", s->lambda_D, s->n_cell}, {"lambda_K", s->lambda_K, s->n_cell},
    {"courant_x", s->courant_x, s->n_cx}, {"courant_y", s->courant_y, s->n_cy}, {"courant_z", s->courant_z, s->n_cz},
    {"vt_0", s->vt_0, 10000}, {"count_mom", s->count_mom, s->count_n}, {"col", s->col, s->n_part},
    {"rw2", s->rw2, s->n_part}, {"rd3", s->rd3, s->n_part}, {"kappa", s->kpa, s->n_part},
    {"x", s->x, s->n_part}, {"y", s->y, s->n_part}, {"z", s->z, s->n_part},
    {"sstp_tmp_rv", s->exact ? s->pp_rv : s->sstp_tmp_rv, s->exact ? s->n_part : s->n_cell},
    {"sstp_tmp_th", s->exact ? s->pp_th : s->sstp_tmp_th, s->exact ? s->n_part : s->n_cell},
    {"sstp_tmp_rh", s->exact ? s->pp_rh : s->sstp_tmp_rh, s->exact ? s->n_part : s->n_cell},
    {"sstp_tmp_p", s->pp_p, s->exact && s->o.const_p ? s->n_part : 0}, {"rc2", s->rc2, s->use_rc2 ? s->n_part : 0},
    {"up", s->up, s->up ? s->n_part : 0}, {"vp", s->vp, s->vp ? s->n_part : 0}, {"wp", s->wp, s->wp ? s->n_part : 0},
    {"incloud_time", s->ict, s->ict ? s->n_part : 0}, {"ssp", s->ssp, s->ssp ? s->n_part : 0}, {"dot_ssp", s->dot_ssp, s->dot_ssp ? s->n_part : 0},
    {"tag", s->tag, s->tag ? s->n_part : 0}, {"diss_rate", s->diss_rate, s->diss_rate ? s->n_cell : 0}};
  for (sz i = 0; i < sizeof tab / sizeof *tab; ++i)
    if (!strcmp(name, tab[i].nm)) {
      *n = tab[i].len;
      if (out) { if (cap < tab[i].len) FAIL("buffer too small"); for (sz k = 0; k < tab[i].len; ++k) out[k] = (dbl)tab[i].v[k]; }
      return 0;
    }
  FAIL("unknown real state '%s'", name);
}
int orc_set_particles(orc_particles *s, size_t n, const unsigned long long *mult, const dbl *rd3, const dbl *rw2,
                      const dbl *kpa, const dbl *vt, const dbl *x, const dbl *y, const dbl *z)
{
  if (n > s->cap) FAIL("n_sd_max (%llu) < n_part (%zu)", s->o.n_sd_max, n);
  s->n_part = n;
  memcpy(s->n, mult, n * sizeof(n_t));
  for (sz p = 0; p < n; ++p) { s->rd3[p] = (real)rd3[p]; s->rw2[p] = (real)rw2[p]; s->kpa[p] = (real)kpa[p]; s->vt[p] = (real)vt[p]; }
  if (x) for (sz p = 0; p < n; ++p) s->x[p] = (real)x[p];
  if (y) for (sz p = 0; p < n; ++p) s->y[p] = (real)y[p];
  if (z) for (sz p = 0; p < n; ++p) s->z[p] = (real)z[p];
  hskpng_ijk(s);
  if (s->use_rc2) { for (sz p = 0; p < n; ++p) s->rc2[p] = -1.; hskpng_approximate_rc2_invalid(s); }
  if (s->up) { memset(s->up, 0, n * sizeof(real)); memset(s->vp, 0, n * sizeof(real)); memset(s->wp, 0, n * sizeof(real)); }
  if (s->ssp) { memset(s->ssp, 0, n * sizeof(real)); memset(s->dot_ssp, 0, n * sizeof(real)); }
  if (s->ict) memset(s->ict, 0, n * sizeof(real));
  if (s->tag) for (sz p = 0; p < n; ++p) s->tag[p] = (real)p;
  sstp_save(s);
  hskpng_count(s);
  return 0;
}
/* preview of the next random arrays WITHOUT advancing the engine: kinds[i] 0 = u01, 1 = un; lens[i] values each;
 * out receives the concatenation.  Used to replay the CPU stream on the device. */
int orc_rng_preview(orc_particles *s, const int *kinds, const size_t *lens, int ncalls, dbl *out)
{
  mt19937_t g = s->rng;
  normal_state ns = s->rng_ns;
  for (int c = 0; c < ncalls; ++c)
    for (sz i = 0; i < lens[c]; ++i) *out++ = kinds[c] == 0 ? (dbl)rng_u01(&g) : kinds[c] == 1 ? rng_un_dbl(&g) : (dbl)rng_normal(&g, &ns);
  return 0;
}
/* test hook of the reverse replay: overwrite ONE particle attribute in place ("rw2": the wet radii after a condensation step, so that
 * the stages behind it are compared from identical inputs); nothing else is touched */
int orc_set_state_real(orc_particles *s, const char *name, const dbl *data, size_t n)
{
  const int cell = !strcmp(name, "th") || !strcmp(name, "rv");      /* (the cell fields that condensation has just updated) */
  if (n != (cell ? s->n_cell : s->n_part)) FAIL("oracle: set_state_real: %zu values for '%s'", n, name);
  real *dst = !strcmp(name, "rw2") ? s->rw2 : !strcmp(name, "rd3") ? s->rd3 : !strcmp(name, "vt") ? s->vt :
                !strcmp(name, "th") ? s->th : !strcmp(name, "rv") ? s->rv : !strcmp(name, "tag") ? s->tag : NULL;
  if (!dst) FAIL("oracle: set_state_real: unknown attribute '%s'", name);
  for (sz i = 0; i < n; ++i) dst[i] = (real)data[i];
  return 0;
}
/* queue a random array for the next consumer of its kind (0: the u01 of a coalescence call, 1: the un of a shuffle), see struct */
int orc_rng_replay_push(orc_particles *s, int kind, const dbl *data, size_t n)
{
  if ((s->rq_tail + 1) % 64 == s->rq_head) FAIL("oracle: rng replay queue is full");
  if (kind != 0 && kind != 1) FAIL("oracle: rng replay kind must be 0 (u01) or 1 (un)");
  dbl *v = NEW(dbl, n);
  memcpy(v, data, n * sizeof(dbl));
  s->rq[s->rq_tail].kind = kind; s->rq[s->rq_tail].v = v; s->rq[s->rq_tail].n = n;
  s->rq_tail = (s->rq_tail + 1) % 64;
  return 0;
}
int orc_rng_replay_pending(orc_particles *s, size_t *n) { *n = (size_t)((s->rq_tail - s->rq_head + 64) % 64); return 0; }
int orc_stage(orc_particles *s, const char *st, const lcx_opts_t *opts)
{
  if (!strcmp(st, "hskpng_Tpr")) hskpng_Tpr(s);
  else if (!strcmp(st, "hskpng_mfp")) hskpng_mfp(s);
  else if (!strcmp(st, "hskpng_ijk")) hskpng_ijk(s);
  else if (!strcmp(st, "hskpng_sort")) hskpng_sort(s);
  else if (!strcmp(st, "hskpng_shuffle_and_sort")) hskpng_sort_helper(s, 1);
  else if (!strcmp(st, "hskpng_count")) hskpng_count(s);
  else if (!strcmp(st, "hskpng_vterm_all")) hskpng_vterm(s, 0);
  else if (!strcmp(st, "hskpng_vterm_invalid")) hskpng_vterm(s, 1);
  else if (!strcmp(st, "coal")) { if (adjust_timesteps(s, opts ? opts->dt : -1)) return 1; coal(s, s->dt / s->sstp_coal, opts ? opts->turb_coal : 0); }
  else if (!strcmp(st, "adve")) adve(s);
  else if (!strcmp(st, "sedi")) { if (adjust_timesteps(s, opts ? opts->dt : -1)) return 1; sedi(s, s->dt); }
  else if (!strcmp(st, "bcnd")) bcnd(s);
  else if (!strcmp(st, "post_copy")) { lcx_opts_t o; orc_opts_default(&o); return post_copy(s, opts ? opts : &o); }
  else FAIL("unknown stage '%s'", st);
  return 0;
}

/* ---------------- 1-D decomposition helpers (pack.ipp:14-133, unpack.ipp:14-143) ---------------- */
int orc_migrate_counts(orc_particles *s, size_t *l, size_t *r) { *l = s->lft_count; *r = s->rgt_count; return 0; }
/* attributes that travel with a super-droplet: distmem_real_vctrs, particles_impl.ipp:440-491 */
static int mig_attrs(orc_particles *s, real **a)
{
  int k = 0;
  a[k++] = s->rd3; a[k++] = s->rw2; a[k++] = s->kpa; a[k++] = s->vt;
  if (s->o.nx != 0) a[k++] = s->x;
  if (s->o.ny != 0) a[k++] = s->y;
  if (s->o.nz != 0) a[k++] = s->z;
  if (s->exact) { a[k++] = s->pp_rv; a[k++] = s->pp_th; a[k++] = s->pp_rh; if (s->o.const_p) a[k++] = s->pp_p; }
  if (s->o.turb_adve_switch) { if (s->o.nx != 0) a[k++] = s->up; if (s->o.ny != 0) a[k++] = s->vp; if (s->o.nz != 0) a[k++] = s->wp; }
  if (s->o.turb_cond_switch) { if (!(s->o.turb_adve_switch && s->o.nz != 0)) a[k++] = s->wp; a[k++] = s->ssp; a[k++] = s->dot_ssp; }
  if (s->o.diag_incloud_time) a[k++] = s->ict;
  if (s->use_rc2) a[k++] = s->rc2;
  if (s->tag) a[k++] = s->tag;
  return k;
}
size_t orc_migrate_record_bytes(orc_particles *s) { real *a[24]; return sizeof(n_t) + sizeof(real) * (size_t)mig_attrs(s, a); }
int orc_migrate_pack(orc_particles *s, int side, dbl x_rmt, void *buf, size_t cap_bytes)
{
  const sz cnt = side == 0 ? s->lft_count : s->rgt_count;
  const sz *id = side == 0 ? s->lft_id : s->rgt_id;
  if (cap_bytes < cnt * orc_migrate_record_bytes(s)) FAIL("migrate_pack: buffer too small");
  const real x_lcl = side == 0 ? s->o.x0 : s->o.x1;
  for (sz i = 0; i < cnt; ++i) s->x[id[i]] = x_rmt + s->x[id[i]] - x_lcl;   /* detail::remote, pack.ipp:14-26 */
  n_t *nb = (n_t *)buf; real *rb = (real *)(nb + cnt);
  real *attrs[24]; const int na = mig_attrs(s, attrs);
  for (sz i = 0; i < cnt; ++i) nb[i] = s->n[id[i]];
  for (int a = 0; a < na; ++a) for (sz i = 0; i < cnt; ++i) rb[(sz)a * cnt + i] = attrs[a][id[i]];
  return 0;
}
int orc_migrate_unpack(orc_particles *s, const void *buf, size_t cnt)
{
  if (cnt == 0) return 0;
  const sz old = s->n_part;
  if (old + cnt > s->cap) FAIL("n_sd_max (%llu) < n_part (%zu)", s->o.n_sd_max, old + cnt);
  const n_t *nb = (const n_t *)buf; const real *rb = (const real *)(nb + cnt);
  real *attrs[24]; const int na = mig_attrs(s, attrs);
  for (sz i = 0; i < cnt; ++i) s->n[old + i] = nb[i];
  for (int a = 0; a < na; ++a) for (sz i = 0; i < cnt; ++i) attrs[a][old + i] = rb[(sz)a * cnt + i];
  const real tol = 5e-4;                          /* config.hpp:31, tolerance_away_from_bcond */
  for (sz i = old; i < old + cnt; ++i) { const real x = s->x[i]; s->x[i] = x >= s->o.x1 ? x - tol : x < s->o.x0 ? x + tol : x; }
  s->n_part = old + cnt;
  return 0;
}
int orc_migrate_finish(orc_particles *s, const lcx_opts_t *opts)
{
  /* emigrants were the first lft_count/rgt_count ids recorded by bcnd: flag them (flag_lft/rgt, unpack.ipp:118-141) */
  for (sz i = 0; i < s->lft_count; ++i) s->n[s->lft_id[i]] = 0;
  for (sz i = 0; i < s->rgt_count; ++i) s->n[s->rgt_id[i]] = 0;
  s->lft_count = s->rgt_count = 0;
  return post_copy(s, opts);
}

/* ---------------- the same three steps behind the message interface of include/lcx.h lcx_exch_* ----------------
 * CPU twin of the product's device-driven exchange so that libcloudphxx_amd/multi.py runs ONE protocol with either engine (the gloo
 * tests drive it with this oracle).  Message = 256-byte header {count, overflow, next capacity} + tiles of 256 records, each tile
 * n[256] | attr_0[256] | attr_1[256] ... (csrc/lcx_kernels.hpp k_pack_dev states the layout; pack.ipp:30-133 / unpack.ipp:50-143 the
 * reference's attribute-major buffers it replaces). */
#define XHDR 256
#define XTILE 256
static size_t x_rec_bytes(orc_particles *s) { real *a[24]; return sizeof(n_t) + sizeof(real) * (size_t)mig_attrs(s, a); }
static size_t x_msg_bytes(orc_particles *s, size_t n_rec) { return XHDR + ((n_rec + XTILE - 1) / XTILE) * XTILE * x_rec_bytes(s); }
int orc_exch_enable(orc_particles *s, int nx_min, size_t *cap_rec)
{
  sz c = 2 * s->cap / (sz)(nx_min > 0 ? nx_min : 1) + 1024;
  if (c > s->cap) c = s->cap;
  s->xcap = (c + XTILE - 1) / XTILE * XTILE;
  for (int k = 0; k < 4; ++k) s->xbox[k] = (unsigned char *)calloc(x_msg_bytes(s, s->xcap), 1);
  *cap_rec = s->xcap;
  return 0;
}
int orc_exch_buffers(orc_particles *s, void *ptrs[4]) { for (int k = 0; k < 4; ++k) ptrs[k] = s->xbox[k]; return 0; }
size_t orc_exch_message_bytes(orc_particles *s, size_t n_rec) { return x_msg_bytes(s, n_rec); }
static void x_pack_side(orc_particles *s, unsigned char *msg, const sz *id, sz cnt, real x_rmt, real x_lcl, unsigned next_cap)
{
  unsigned *h = (unsigned *)msg;
  h[0] = (unsigned)cnt; h[1] = cnt > s->xcap; h[2] = next_cap;
  if (cnt > s->xcap) return;
  real *attrs[24]; const int na = mig_attrs(s, attrs);
  const size_t tile_bytes = XTILE * x_rec_bytes(s);
  for (sz i = 0; i < cnt; ++i) {
    s->x[id[i]] = x_rmt + s->x[id[i]] - x_lcl;                       /* detail::remote, pack.ipp:14-26 */
    unsigned char *t = msg + XHDR + (i / XTILE) * tile_bytes;
    const sz j = i % XTILE;
    ((n_t *)t)[j] = s->n[id[i]];
    for (int a = 0; a < na; ++a) ((real *)(t + XTILE * sizeof(n_t)))[(sz)a * XTILE + j] = attrs[a][id[i]];
    s->n[id[i]] = 0;                                                   /* flag_lft / flag_rgt, unpack.ipp:118-141 */
  }
}
int orc_exch_pack(orc_particles *s, int has_lft, dbl lft_x1, int has_rgt, dbl rgt_x0, unsigned next_lft, unsigned next_rgt)
{
  if (has_lft) x_pack_side(s, s->xbox[0], s->lft_id, s->lft_count, lft_x1, s->o.x0, next_lft);
  if (has_rgt) x_pack_side(s, s->xbox[1], s->rgt_id, s->rgt_count, rgt_x0, s->o.x1, next_rgt);
  return 0;
}
static int x_unpack_side(orc_particles *s, const unsigned char *msg)
{
  const unsigned *h = (const unsigned *)msg;
  const sz cnt = h[1] ? 0 : h[0], old = s->n_part;
  if (old + cnt > s->cap) FAIL("n_sd_max (%llu) < n_part (%zu)", s->o.n_sd_max, old + cnt);
  real *attrs[24]; const int na = mig_attrs(s, attrs);
  const size_t tile_bytes = XTILE * x_rec_bytes(s);
  const real tol = 5e-4;                                             /* config.hpp:31, tolerance_away_from_bcond */
  for (sz i = 0; i < cnt; ++i) {
    const unsigned char *t = msg + XHDR + (i / XTILE) * tile_bytes;
    const sz j = i % XTILE;
    s->n[old + i] = ((const n_t *)t)[j];
    for (int a = 0; a < na; ++a) attrs[a][old + i] = ((const real *)(t + XTILE * sizeof(n_t)))[(sz)a * XTILE + j];
    const real x = s->x[old + i];
    s->x[old + i] = x >= s->o.x1 ? x - tol : x < s->o.x0 ? x + tol : x;
  }
  s->n_part = old + cnt;
  return 0;
}
int orc_exch_unpack(orc_particles *s, int from_lft, int from_rgt, unsigned have_lft, unsigned have_rgt)
{
  const unsigned *hl = (const unsigned *)s->xbox[2], *hr = (const unsigned *)s->xbox[3];
  s->xflags = 0;
  if ((from_lft && !hl[1] && hl[0] > have_lft) || (from_rgt && !hr[1] && hr[0] > have_rgt)) { s->xflags = 4; return 0; }
  if ((from_lft && hl[1]) || (from_rgt && hr[1])) s->xflags |= 1;
  if (from_lft && x_unpack_side(s, s->xbox[2])) return 1;             /* the left neighbour's first (unpack.ipp: lft, then rgt) */
  if (from_rgt && x_unpack_side(s, s->xbox[3])) return 1;
  return 0;
}
int orc_exch_finish(orc_particles *s, const lcx_opts_t *opts, unsigned rec[12], int *complete)
{
  const unsigned *hl = (const unsigned *)s->xbox[2], *hr = (const unsigned *)s->xbox[3];
  memset(rec, 0, 12 * sizeof *rec);
  rec[1] = (unsigned)s->lft_count; rec[2] = (unsigned)s->rgt_count; rec[3] = hl[0]; rec[4] = hr[0]; rec[5] = s->xflags; rec[9] = hl[2]; rec[10] = hr[2];
  *complete = !(s->xflags & 4);
  if (!*complete) return 0;
  if (s->xflags & 1) FAIL("libcloudph++: more super-droplets crossed a slab face in one step than the exchange buffer holds (%zu records); raise opts_init.n_sd_max", s->xcap);
  s->lft_count = s->rgt_count = 0;
  return post_copy(s, opts);
}
int orc_exch_sort_interior(orc_particles *s) { (void)s; return 0; }      /* (the product's overlap of the re-sort with the transport: nothing to do here) */
int orc_stream(orc_particles *s, void **stream) { (void)s; *stream = NULL; return 0; }

/* ---------------- Courant halo exchange of pred_corr (xchng_courants.ipp:15-160) ---------------- */
/* element ranges inside the halo-extended arrays: [send to left, send to right, recv from left, recv from right] */
static sz courant_halo_geom(orc_particles *s, int which, real **arr, sz off[4])
{
  const lcx_opts_init_t *o = &s->o;
  if (!s->halo || s->n_dims == 0) return 0;
  const sz ny = m1(o->ny), nz = m1(o->nz), h = (sz)s->halo;
  sz plane, n;                                       /* reals per x-plane of this array, total */
  if (which == 0) { plane = s->n_dims == 1 ? 1 : s->n_dims == 2 ? nz : nz * ny; *arr = s->courant_x; n = s->n_cx; }
  else if (which == 1) { if (s->n_dims < 3) return 0; plane = (ny + 1) * nz; *arr = s->courant_y; n = s->n_cy; }
  else { if (s->n_dims < 2) return 0; plane = s->n_dims == 2 ? nz + 1 : (nz + 1) * ny; *arr = s->courant_z; n = s->n_cz; }
  const sz cnt = h * plane;
  if (which == 0) { off[0] = (h + 1) * plane; off[1] = (sz)o->nx * plane; }      /* cx_lft_internal_idx, cx_rgt_internal_idx = n_cell */
  else            { off[0] = cnt;             off[1] = (sz)o->nx * plane; }      /* c[yz]_lft_internal_idx = halo, _rgt = nx planes in */
  off[2] = 0; off[3] = n - cnt;
  return cnt;
}
size_t orc_courant_halo_count(orc_particles *s, int which) { real *a; sz off[4]; return courant_halo_geom(s, which, &a, off); }
int orc_courant_halo_pack(orc_particles *s, int which, int side, void *buf)
{
  real *a; sz off[4]; const sz cnt = courant_halo_geom(s, which, &a, off);
  if (cnt) memcpy(buf, a + off[side], cnt * sizeof(real));
  return 0;
}
int orc_courant_halo_unpack(orc_particles *s, int which, int side, const void *buf)
{
  real *a; sz off[4]; const sz cnt = courant_halo_geom(s, which, &a, off);
  if (cnt) memcpy(a + off[2 + side], buf, cnt * sizeof(real));
  return 0;
}

/* libcloudphxx.common of the reference's Python module (bindings/python/common.hpp:19-172, lib.cpp:55-66,129-144), scalar */
static int orc_common_eval_real(const char *name, const real *a, int n, real *out);
int orc_common_eval(const char *name, const dbl *a_, int n, dbl *out_)
{
  real a[16], out_v = 0, *out = &out_v;
  for (int i = 0; i < n && i < 16; ++i) a[i] = (real)a_[i];
  const int rc_ = orc_common_eval_real(name, a, n, out);
  *out_ = (dbl)out_v;
  return rc_;
}
static int orc_common_eval_real(const char *name, const real *a, int n, real *out)
{
#define IS(nm) (!strcmp(name, nm))
  const real kap = R_d / c_pd;
  if (IS("th_dry2std") && n == 2) *out = a[0] / pow(1 + a[1] * R_v / R_d, kap);                   /* theta_dry.hpp:101-113 */
  else if (IS("th_std2dry") && n == 2) *out = a[0] * pow(1 + a[1] * R_v / R_d, kap);              /* theta_dry.hpp:86-99 */
  else if (IS("exner") && n == 1) *out = theta_std_exner(a[0]);
  else if (IS("p_v") && n == 2) *out = p_v(a[0], a[1]);
  else if (IS("p_vs") && n == 1) *out = p_vs(a[0]);
  else if (IS("r_vs") && n == 2) *out = r_vs(a[0], a[1]);
  else if (IS("p_vs_tet") && n == 1) *out = tet_p_vs(a[0]);
  else if (IS("l_v") && n == 1) *out = l_v(a[0]);
  else if (IS("T") && n == 2) *out = theta_dry_T(a[0], a[1]);
  else if (IS("p") && n == 3) *out = theta_dry_p(a[0], a[1], a[2]);
  else if (IS("visc") && n == 1) *out = visc(a[0]);
  else if (IS("rw3_cr") && n == 3) *out = rw3_cr(a[0], a[1], a[2]);
  else if (IS("S_cr") && n == 3) *out = S_cr(a[0], a[1], a[2]);
  else if (IS("p_hydro") && n == 5) {                                                             /* hydrostatic.hpp:24-38 */
    const real R_moist = (R_d + a[2] * R_v) / (1 + a[2]);                                       /* moist_air.hpp:54-70 */
    *out = p_1000 * pow(pow(a[4] / p_1000, kap) - kap * g_earth / a[1] / R_moist * (a[0] - a[3]), c_pd / R_d);
  }
  else if (IS("rhod") && n == 3) *out = (a[0] - p_v(a[0], a[2])) / (pow(a[0] / p_1000, kap) * R_d * a[1]);   /* theta_std.hpp:23-32 */
  else if (IS("R_d") && n == 0) *out = R_d; else if (IS("R_v") && n == 0) *out = R_v;
  else if (IS("c_pd") && n == 0) *out = c_pd; else if (IS("c_pv") && n == 0) *out = c_pv;
  else if (IS("c_pw") && n == 0) *out = c_pw; else if (IS("g") && n == 0) *out = g_earth;
  else if (IS("p_1000") && n == 0) *out = p_1000; else if (IS("eps") && n == 0) *out = eps_v;
  else if (IS("rho_stp") && n == 0) *out = rho_stp; else if (IS("rho_w") && n == 0) *out = rho_w;
  else FAIL("libcloudph++: common.%s with %d argument(s) is not provided by this backend", name, n);
  return 0;
#undef IS
}
