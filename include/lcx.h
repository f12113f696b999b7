/*
 * lcx.h -- C ABI of the MI355X-native super-droplet ("lgrngn") backend.
 *
 * This is the drop-in boundary: plain C, opaque handle, raw pointers + sizes,
 * no C++ / torch types.  Every entry point replaces one virtual of the
 * reference's particles_proto_t<real_t>
 * (reference: include/libcloudph++/lgrngn/particles.hpp:17-134) or one of the
 * structs passed to it (opts_init.hpp:29-253, opts.hpp:20-50, arrinfo.hpp:11-49).
 * The C++ mirror in include/libcloudphxx_amd/lgrngn/ and the ctypes mirror in
 * libcloudphxx_amd/lgrngn.py are thin shims over these functions.
 *
 * Error convention: every function returns 0 on success, non-zero on failure;
 * lcx_last_error() then returns the message ("libcloudph++: ..." texts follow
 * the reference's std::runtime_error strings, e.g. src/particles_step.ipp:44-78)
 * and the C++ shim re-throws it as std::runtime_error.
 *
 * real_t: a handle is created for float (real_kind=4) or double (real_kind=8);
 * all `void *` array arguments are arrays of that type.
 */
#ifndef LCX_H
#define LCX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* enum values follow the declaration order of the reference enums
 * (kernel.hpp:8, terminal_velocity.hpp:8, advection_scheme.hpp:8, RH_formula.hpp:8) */
enum lcx_kernel {
  LCX_KERNEL_UNDEFINED = 0, LCX_KERNEL_GEOMETRIC, LCX_KERNEL_GOLOVIN, LCX_KERNEL_HALL,
  LCX_KERNEL_HALL_DAVIS_NO_WAALS, LCX_KERNEL_LONG, LCX_KERNEL_ONISHI_HALL,
  LCX_KERNEL_ONISHI_HALL_DAVIS_NO_WAALS, LCX_KERNEL_HALL_PINSKY_1000MB_GRAV,
  LCX_KERNEL_HALL_PINSKY_CUMULONIMBUS, LCX_KERNEL_HALL_PINSKY_STRATOCUMULUS,
  LCX_KERNEL_VOHL_DAVIS_NO_WAALS
};
enum lcx_vt {
  LCX_VT_UNDEFINED = 0, LCX_VT_BEARD76, LCX_VT_BEARD77, LCX_VT_BEARD77FAST,
  LCX_VT_KHVOROSTYANOV_SPHERICAL, LCX_VT_KHVOROSTYANOV_NONSPHERICAL
};
enum lcx_adve { LCX_ADVE_UNDEFINED = 0, LCX_ADVE_IMPLICIT, LCX_ADVE_EULER, LCX_ADVE_PRED_CORR };
enum lcx_rh   { LCX_RH_PV_CC = 0, LCX_RH_RV_CC, LCX_RH_PV_TET, LCX_RH_RV_TET };

/* indices of lcx_diag_puddle() output, = common::output_t (common/output.hpp:8-24) */
enum lcx_puddle {
  LCX_OUT_HNO3 = 0, LCX_OUT_NH3, LCX_OUT_CO2, LCX_OUT_SO2, LCX_OUT_H2O2, LCX_OUT_O3,
  LCX_OUT_S_VI, LCX_OUT_H, LCX_OUT_LIQ_VOL, LCX_OUT_DRY_VOL, LCX_OUT_PRTCL_NUM,
  LCX_OUT_ICE_MASS, LCX_OUT_LIQ_NUM, LCX_OUT_ICE_NUM, LCX_OUT_COUNT
};

/* n(ln rd) at STP of one dry-aerosol mode; evaluated ON THE HOST exactly like the
 * reference evaluates unary_function::funval (init_n.ipp:56-84, init_dist_analysis.ipp:62-64) */
typedef double (*lcx_distro_fn)(double lnrd, void *user);

typedef struct {
  double kappa, rd_insol;        /* key of dry_distros_t (distro_t.hpp:10-36) */
  lcx_distro_fn fn;              /* used when fn != NULL */
  void *user;
  /* built-in alternative (fn == NULL): sum of n_modes lognormal modes
   * n_stp/ (sqrt(2pi) ln sdev) exp(-(lnrd-ln mean_rd)^2 / (2 ln^2 sdev)), common/lognormal.hpp:25-37;
   * n_modes = -1: the exponential-in-volume spectrum of Shima et al. 2009 that the reference's Golovin test passes as a Python function
   * (tests/python/physics/coalescence_golovin.py:41-44), n(ln r) = 3 n_stp[0] (r / mean_rd[0])^3 exp(-(r / mean_rd[0])^3), so that
   * a 1e8-droplet box of it can be initialised without a host callback per droplet */
  int n_modes;
  double mean_rd[4], sdev[4], n_stp[4];
} lcx_distro_t;

/* one (kappa, radius) -> (STP concentration, SD count) entry of dry_sizes_t (distro_t.hpp:39-46) */
typedef struct { double kappa, rd_insol, radius, conc; int sd_count; } lcx_dry_size_t;

/* POD mirror of opts_init_t<real_t> (opts_init.hpp:29-253); same field names and defaults.
 * Fields of sub-systems that are out of scope (chemistry, ice, sources, relaxation, SGS
 * turbulence) are kept so that a caller's settings are checked, not silently dropped:
 * lcx_create() fails if one of them is switched on. */
typedef struct {
  int nx, ny, nz;
  double dx, dy, dz, dt;
  int sstp_cond, sstp_coal, sstp_cond_act, sstp_chem;
  double x0, y0, z0, x1, y1, z1;
  unsigned long long sd_conc;
  int sd_conc_large_tail, aerosol_independent_of_rhod, variable_dt_switch;
  unsigned long long sd_const_multi, n_sd_max;
  int kernel, terminal_velocity, adve_scheme, RH_formula;
  const double *kernel_parameters; int n_kernel_parameters;
  int chem_switch, coal_switch, sedi_switch, subs_switch, rlx_switch,
      turb_adve_switch, turb_cond_switch, turb_coal_switch, ice_switch,
      exact_sstp_cond, sstp_cond_mix, adaptive_sstp_cond, time_dep_ice_nucl;
  double RH_max;
  double sstp_cond_adapt_drw2_eps, sstp_cond_adapt_drw2_max, rc2_T;   /* opts_init.hpp:105-106,149 */
  int rng_seed, rng_seed_init, rng_seed_init_switch;
  int dev_count, dev_id;
  const double *w_LS; int n_w_LS;
  const double *SGS_mix_len; int n_SGS_mix_len;             /* opts_init.hpp: SGS mixing length profile [m], size nz */
  const double *aerosol_conc_factor; int n_aerosol_conc_factor;
  double rd_min, rd_max;
  int no_ccn_at_init, open_side_walls, periodic_topbot_walls;
  int src_type;
  int th_dry, const_p;
  int diag_incloud_time;
  const lcx_distro_t *dry_distros; int n_dry_distros;      /* sorted by (kappa, rd_insol) like std::map */
  const lcx_dry_size_t *dry_sizes; int n_dry_sizes;        /* sorted by (kappa, rd_insol, radius) */
  /* --- extensions (no reference counterpart) --- */
  int n_x_tot;          /* total nx of the decomposed domain (ctor arg n_x_tot, particles.hpp:233) */
  int n_x_bfr;          /* x-planes owned by ranks to the left (distmem_opts.hpp:27) */
  int bcond_lft, bcond_rgt; /* 0 sharedmem, 1 distmem, 3 open  (src/detail/bcond.hpp) */
  int strict_fp;        /* 0 (default since round 5): fast arithmetic -- growth rate collected into one rational expression + FMA, refined
                         *    reciprocals, the way the reference's own Release build is compiled (-Ofast), and the condensation equation
                         *    solved by cond_solver below;
                         * 1: IEEE order-preserving arithmetic, the reference's TOMS748 iterates and its ordered per-cell sums (the parity
                         *    mode that the tests pin; 2x the step time of the default) */
  int cond_solver;      /* fast arithmetic only.
                         * 1 (default since round 5): the reference's TOMS748 iterates in the fast arithmetic (the storage-order kernel with
                         *    TOMS748 in it): the reference's answers as closely as its own builds follow each other -- held to SURVEY 8a's
                         *    bars and to the reference's refdata tolerances, like strict_fp = 1, in every test;
                         * 0: a lean bracketed secant on the reference's bracket, to the reference's tolerance 2^-15 (csrc/lcx_math.hpp
                         *    advance_rw2_lean2_with): the ROOT of rw2' = rw2 + dt f(rw2') itself, within that tolerance of the reference's
                         *    answer -- which is the midpoint of TOMS748's last bracket, up to 1.5e-5 from the root it brackets.  A droplet
                         *    whose bracket can hold SEVERAL roots (one that can evaporate down to its dry core within the step, a bracket
                         *    across more than a factor of four in radius in supersaturated air: about 0.1 % of bench.py's boxes) is solved with
                         *    TOMS748 as under 1, so that WHICH root it ends on is the reference's choice (k_cond_lean_listed).  Half the
                         *    kernel time of 1, what bench.py's headline runs */
  int reorder_every;    /* physical re-ordering of the super-droplet storage into the cell-sorted order (keeps the per-cell gathers
                         * line-coalesced in long runs: 18.1 instead of 21.5 ms per step after 400 steps of the 128^3 box).
                         * N > 0: every N steps, and whenever dead super-droplets are compacted away anyway (the same one pass over
                         * the attributes, gathered in sorted order instead of storage order); 0 (default): N = 64, or 16 for a slab that has
                         * neighbours (one extra pass of ~3.4 ms per 64 steps of ~15 ms);
                         * -1: never -- stable compaction, storage order == the reference's id order at all times.
                         * A re-ordering renumbers the ids: SDs keep their relative order inside a cell, the relative order of SDs of
                         * different cells and the id -> random-number association change (statistically equivalent).  An object
                         * that was ever fed a replayed random stream (lcx_rng_replay_push, i.e. a parity run) behaves like -1. */
  int stream_ordered;   /* 0 (default): step_sync / step_cond return when th and rv have been written, as the reference's do.
                         * 1: with DEVICE arrays (lcx_arrinfo_t.on_device) they return once the work is queued on the object's stream, like any
                         *    GPU library call: th and rv are valid for work ordered behind lcx_stream() (an event wait on the caller's stream),
                         *    the host does not wait -- it is already queueing step_async while condensation runs.  Calls that involve host
                         *    arrays, and everything that hands data to the host (diag, outbuf, get_attr ...), synchronise as before.
                         *    THE OTHER DIRECTION IS THE CALLER'S TOO: the next step_sync reads the caller's device arrays (th, rv, rhod, the
                         *    Courant numbers) on lcx_stream() with no host wait in between, so a caller that rewrites them on a stream of its
                         *    own makes lcx_stream() wait for that work first (hipStreamWaitEvent(lcx_stream(h), its_event), or a device
                         *    synchronisation) -- with stream_ordered = 0 the host waits at the end of step_cond and only the usual rule is
                         *    left: what the caller queued on its streams before calling in must have been ordered by the caller. */
  /* --- test / measurement switches (no reference counterpart; all 0 in production).  They used to be LCX_* environment variables read
   * here and there in the library; a stray variable in a user's environment silently changed the kernel path.  Now they are part of the
   * options an object is created with, read once, and the library reads no environment variable but LCX_DATA_DIR (where the collision
   * efficiency tables live) and LCX_MULTI_DEVICE_MAP (slab -> device map of a multi-device object). */
  unsigned dbg_flags;   /* LCX_DBG_* bits below */
  int dbg_cond_budget;  /* cond_solver = 1 with LCX_DBG_COND_TOMS_TWO_PASS: iteration budget of the first condensation pass (stragglers go to a dense second launch);
                         * 0: the library's choice (6 from 2^25 super-droplets, else one pass), > 0: that budget, < 0: one pass.
                         * The lean solver's folded kernel (k_cond_lean_fold): > 0 = slots of its LDS stage in use (at most 128; a test makes it
                         * small so that unconverged droplets stay in their own lanes) */
  int dbg_pack_delay_us;/* multi-device tests: slabs with an odd first plane send their messages so many microseconds late */
} lcx_opts_init_t;

enum lcx_dbg {
  LCX_DBG_NO_COND_PRE = 1 << 0,        /* fast arithmetic: evaluate the per-cell set-up of the growth rate per droplet (k_cond<T, true>) */
  LCX_DBG_NO_WAVE_FLAGS = 1 << 1,      /* exchange: scatter the boundary super-droplets without the per-wave flags */
  LCX_DBG_EAGER_COMPACT = 1 << 2,      /* compact dead super-droplets away in every step (the reference's remove_n0) */
  LCX_DBG_SHUFFLE_PHILOX = 1 << 3,     /* shuffle keys drawn from Philox, ranked on 64 bits (round 2's form) */
  LCX_DBG_NO_DEFERRED_SORT = 1 << 4,   /* the end-of-step re-sort finished at once instead of riding on the next condensation kernel */
  LCX_DBG_COND_NO_FOLD = 1 << 5,       /* cond_solver = 1: the plain storage-order kernel instead of the one folded behind TOMS748's head (k_cond_lean_fold<.., 2>,
                                          the same bits); with LCX_DBG_COND_TOMS_TWO_PASS: k_cond_fast instead of k_cond_fast_fold */
  LCX_DBG_COND_SORTED_ORDER = 1 << 6,  /* k_cond_lean over the sorted order (gathers) instead of the storage order */
  LCX_DBG_NO_OVERLAP = 1 << 7,         /* exchange: no re-sort of the interior while the messages travel */
  LCX_DBG_MULTI_NO_PEER = 1 << 8,      /* multi-device object: treat the devices as unable to map each other's memory (staged copies) */
  LCX_DBG_MULTI_SERIALIZE = 1 << 9,    /* multi-device object: the slabs take turns (per-slab timing on one GPU) */
  LCX_DBG_TAG = 1 << 10,               /* every super-droplet carries a persistent tag (its index at init / set_particles) as one more
                                        * attribute that is compacted, re-ordered and migrates with it (lcx_get_state_real "raw_tag"), and
                                        * every coalescence call records what it consumed of the random generator (lcx_rng_dump) */
  LCX_DBG_KPA_ARRAY = 1 << 11,         /* k_cond_lean reads the hygroscopicity array even when the run has a single value (then passed as a scalar) */
  LCX_DBG_HOST_SYNC_LOOP = 1 << 12,    /* host arrays in sync_in / sync_out through the plain host loop (the form rounds 1-3 had) */
  LCX_DBG_EXCH_SORT_NOW = 1 << 14,     /* a slab with neighbours re-sorts inside its exchange (interior while the messages travel, boundary behind
                                        * them) even when the next condensation kernel could carry the scatter */
  LCX_DBG_COND_LEAN_R3 = 1 << 13,      /* k_cond_lean with round 3's form of the solver's bookkeeping and helper functions (the same rw2 bit for bit) */
  LCX_DBG_FINISH_STAGED = 1 << 16,     /* fast arithmetic: the per-cell finish through its LDS stage (k_cond_cellfinish) also where the changes lie in
                                        * the sorted order and k_cond_cellfinish_direct would read them straight from memory */
  LCX_DBG_NO_RANK_OVERLAP = 1 << 17,   /* the in-cell ranking of a carried re-sort on the object's one stream, not next to the per-cell finish and
                                        * the terminal velocities on a stream of its own */
  LCX_DBG_RANK_BY_COUNTING = 1 << 18,  /* the in-cell shuffled order ranked by counting smaller keys (k_cellrank<uint32_t, true>) instead of by buckets */
  LCX_DBG_COND_FOLD = 1 << 19,         /* the lean solver's kernel with its workgroup folded behind the solver's first loop trip (k_cond_lean_fold: the
                                          unconverged droplets handed to the workgroup's lowest lanes through LDS; the same rw2 bit for bit, not faster) */
  LCX_DBG_COND_NO_LIST = 1 << 20,      /* cond_solver = 0: no list of droplets for the reference's iterates (brackets that may hold several roots, k_cond_lean /
                                        * k_cond_lean_listed): the lean solver takes every droplet, as in rounds 3-4 */
  LCX_DBG_COND_WQ = 1 << 21,           /* cond_solver = 0, measured and not adopted (round 6): k_cond_lean_wq -- a wave walks several batches of 64 storage slots and keeps
                                        * the droplets whose first loop trip has not converged on a queue of its own in LDS, taking them up again 64 at a time
                                        * (the same rw2 bit for bit; dbg_cond_budget & 255 = batches per wave) -- instead of k_cond_lean with its budget */
  LCX_DBG_COND_WQ_CAP128 = 1 << 22,    /* k_cond_lean_wq with a queue of 128 entries per wave (its trips always on full waves; four workgroups per CU) instead of 96 */
  LCX_DBG_COND_WQ_PF = 1 << 23,        /* k_cond_lean_wq issues the next batch's slot-indexed loads before it computes the current batch */
  LCX_DBG_COND_WQ_PF2 = 1 << 24,       /* ... and its cell-indexed loads as well (queue of 128 entries, three waves per SIMD) */
  LCX_DBG_COND_BUDGET = 1 << 25,       /* cond_solver = 0, measured and not adopted (round 6): the first pass gives every droplet's loop a budget of two trips
                                        * (dbg_cond_budget & 255: that many); a droplet that has not converged by then leaves the loop's state in a record and
                                        * k_cond_lean_resume goes on with it (the same bits; dbg_cond_budget >> 8: records per part, tests) */
  LCX_DBG_COND_PROBE = 1 << 26,        /* measurement only: k_cond_lean cut short at seven stages, launched ahead of the real kernel (k_cond_probe: instruction counts per part) */
  LCX_DBG_COND_NO_FUSED_SUBSTEPS = 1 << 27, /* fast arithmetic, sstp_cond > 1: a cell pass, a condensation kernel and a per-cell finish per substep (rounds 1-5) instead of
                                        * every substep of the step in one launch (k_cond_substeps: the same bits) */
  LCX_DBG_VTERM_INVALID_OWN_PASS = 1 << 28, /* sstp_coal > 1: hskpng_vterm_invalid as a launch of its own between the substeps (rounds 1-5) instead of on the
                                        * next substep's in-cell ranking (the same bits) */
  LCX_DBG_COND_TOMS_TWO_PASS = 1 << 15 /* cond_solver = 1 through round 2's kernels (k_cond_fast_fold + k_cond_fast over the sorted order, iteration budget and
                                        * straggler launch) instead of the storage-order kernel with TOMS748 in it */
};

/* lcx_get_state_u64("raw_mode") = { strict_fp, cond_solver, the kernel of the last condensation launch (below; 0: none yet), dbg_flags } as
 * the OBJECT holds them: a test that claims the benchmarked or the default path reads them back instead of trusting what it meant to set */
enum lcx_cond_kernel {
  LCX_CK_STRICT = 1,                   /* k_cond<T, false>: strict_fp = 1 */
  LCX_CK_FAST_PER_DROPLET_SETUP = 2,   /* k_cond<T, true>: fast arithmetic with turb_cond, or LCX_DBG_NO_COND_PRE */
  LCX_CK_LEAN = 3,                     /* k_cond_lean over the storage order: cond_solver = 0, what bench.py's headline runs */
  LCX_CK_LEAN_SORTED = 4,              /* ... over the sorted order (LCX_DBG_COND_SORTED_ORDER, or a droplet order that the storage does not have) */
  LCX_CK_FOLD_TOMS748 = 5,             /* k_cond_lean_fold<.., 2>: cond_solver = 1, the API default */
  LCX_CK_LEAN_TOMS748 = 6,             /* k_cond_lean<.., 2> (LCX_DBG_COND_NO_FOLD) */
  LCX_CK_LEAN_TOMS748_SORTED = 7,
  LCX_CK_TOMS748_TWO_PASS = 8,         /* k_cond_fast(_fold) (LCX_DBG_COND_TOMS_TWO_PASS) */
  LCX_CK_LEAN_R3 = 9, LCX_CK_FOLD_LEAN = 10, LCX_CK_LEAN_WQ = 11,      /* LCX_DBG_COND_LEAN_R3, _COND_FOLD, _COND_WQ */
  LCX_CK_PER_PARTICLE = 12,            /* exact_sstp_cond: k_pp_cond_* */
  LCX_CK_SUBSTEPS = 13                 /* k_cond_substeps: fast arithmetic, sstp_cond > 1 -- every substep of the step in one launch */
};

/* POD mirror of opts_t<real_t> (opts.hpp:20-50) */
typedef struct {
  int adve, sedi, subs, cond, coal, src, rlx, rcyc, turb_adve, turb_cond, turb_coal, ice_nucl;
  int chem_dsl, chem_dsc, chem_rct;
  double RH_max;
  double dt;
} lcx_opts_t;

/* arrinfo_t (arrinfo.hpp:11-49): data == NULL <=> "not provided".  strides in elements.
 * on_device (extension: lets a GPU-resident host model skip the PCIe round trip of particles_impl_sync.ipp:15-68):
 *   0 host array; 1 data is a device pointer (for a multi-device object: the global array on a device every slab's device can read);
 *   2 multi-device objects only: data is a host table of dev_count device pointers (void *[]), entry i = the planes of slab i as an
 *     array of its own on slab i's device, same strides for all (a host model decomposed the same way keeps its fields like that). */
typedef struct {
  void *data;
  const ptrdiff_t *strides;
  int on_device;
} lcx_arrinfo_t;

typedef struct lcx_particles lcx_particles;

void lcx_opts_init_default(lcx_opts_init_t *);   /* opts_init.hpp:186-247 defaults */
void lcx_opts_default(lcx_opts_t *);             /* opts.hpp:41-48 defaults */
const char *lcx_last_error(void);
const char *lcx_version(void);

/* factory<real_t>(backend, opts_init)  (factory.hpp:12-15, src/lib.cpp:13-40) + ctor (particles_ctor.ipp:22-75) */
int lcx_create(const lcx_opts_init_t *, int real_kind, lcx_particles **out);
void lcx_destroy(lcx_particles *);
/* factory<real_t>(multi_CUDA | multi_HIP, opts_init): ONE object that drives opts_init.dev_count devices of this process (0: all
 * visible ones) -- replaces particles_t<real_t, multi_CUDA> (particles.hpp:246-340, src/particles_multi_gpu_*.ipp,
 * src/impl_multi_gpu/particles_multi_gpu_impl.ipp:17-227).  The domain is cut into x-slabs (src/detail/distmem_opts.hpp:10-52), one
 * per device; the arrays passed to init / sync_in / step_cond are the GLOBAL ones (each device reads and writes its planes), outbuf()
 * is gathered into one host array, diag_puddle() is summed; get_attr throws as in the reference; opts.rcyc is refused as there.
 * step_async exchanges the super-droplets that crossed a slab face between neighbouring devices: packed on the device straight into
 * the neighbour's buffer over the peer mapping (xGMI), count in the message header, one host synchronisation per step
 * (replaces impl_multi_gpu/particles_multi_gpu_impl_step_async_and_copy.ipp:28-206).  Every other entry point of this header takes
 * the handle unchanged.  LCX_MULTI_DEVICE_MAP="0,0,1,1" (environment) maps slabs to devices explicitly; several slabs may share one. */
int lcx_create_multi(const lcx_opts_init_t *, int real_kind, lcx_particles **out);
/* number of slabs of a handle (1 for a single-device object) */
int lcx_multi_dev_count(lcx_particles *, int *n);
/* handle of slab i of a multi-device object, for the calls that address one device's storage (state getters, set_particles, random
 * replay: what the reference exposes as the public pimpl->particles[i], particles_multi_gpu_impl.ipp:21); owned by the parent, valid
 * until the parent is destroyed, not to be destroyed itself */
int lcx_multi_slab(lcx_particles *, int i, lcx_particles **slab);

/* particles_t::init (particles_init.ipp:16-131) */
int lcx_init(lcx_particles *, const lcx_arrinfo_t *th, const lcx_arrinfo_t *rv, const lcx_arrinfo_t *rhod,
             const lcx_arrinfo_t *p, const lcx_arrinfo_t *courant_x, const lcx_arrinfo_t *courant_y,
             const lcx_arrinfo_t *courant_z);
/* particles_t::sync_in / step_cond / step_sync / step_async (particles_step.ipp:15-29,32-158,161-336,339-494) */
int lcx_sync_in(lcx_particles *, const lcx_arrinfo_t *th, const lcx_arrinfo_t *rv, const lcx_arrinfo_t *rhod,
                const lcx_arrinfo_t *courant_x, const lcx_arrinfo_t *courant_y, const lcx_arrinfo_t *courant_z,
                const lcx_arrinfo_t *diss_rate);
int lcx_step_cond(lcx_particles *, const lcx_opts_t *, const lcx_arrinfo_t *th, const lcx_arrinfo_t *rv);
int lcx_step_sync(lcx_particles *, const lcx_opts_t *, const lcx_arrinfo_t *th, const lcx_arrinfo_t *rv,
                  const lcx_arrinfo_t *rhod, const lcx_arrinfo_t *courant_x, const lcx_arrinfo_t *courant_y,
                  const lcx_arrinfo_t *courant_z, const lcx_arrinfo_t *diss_rate);
int lcx_step_async(lcx_particles *, const lcx_opts_t *);

/* diagnostics (particles_diag.ipp:148-222,224-248,411-421) */
int lcx_diag_sd_conc(lcx_particles *);
int lcx_diag_pressure(lcx_particles *);
int lcx_diag_temperature(lcx_particles *);
int lcx_diag_RH(lcx_particles *);
int lcx_diag_all(lcx_particles *);
int lcx_diag_vel_div(lcx_particles *);     /* particles_diag.ipp:499-555: divergence of the Courant field, per cell */
int lcx_diag_water(lcx_particles *);
int lcx_diag_dry_rng(lcx_particles *, double r_mi, double r_mx);
int lcx_diag_wet_rng(lcx_particles *, double r_mi, double r_mx);
int lcx_diag_kappa_rng(lcx_particles *, double k_mi, double k_mx);
int lcx_diag_dry_rng_cons(lcx_particles *, double r_mi, double r_mx);
int lcx_diag_wet_rng_cons(lcx_particles *, double r_mi, double r_mx);
int lcx_diag_kappa_rng_cons(lcx_particles *, double k_mi, double k_mx);
int lcx_diag_dry_mom(lcx_particles *, int k);
int lcx_diag_wet_mom(lcx_particles *, int k);
int lcx_diag_kappa_mom(lcx_particles *, int k);
int lcx_diag_water_cons(lcx_particles *);                 /* particles_diag.ipp:346-349: rw2 > 0 among the selected */
int lcx_diag_up_mom(lcx_particles *, int k);              /* :463-480: moments of the SGS velocity perturbations (turb_adve / turb_cond) */
int lcx_diag_vp_mom(lcx_particles *, int k);
int lcx_diag_wp_mom(lcx_particles *, int k);
int lcx_diag_incloud_time_mom(lcx_particles *, int k);   /* particles_diag.ipp:482-490; needs opts_init.diag_incloud_time */
/* selections by activation state (particles_diag.ipp:350-407): RH >= critical supersaturation / r_w >= critical radius */
int lcx_diag_RH_ge_Sc(lcx_particles *);
int lcx_diag_rw_ge_rc(lcx_particles *);
/* kernel-density estimate of the wet mass density function at radius rad (particles_diag.ipp:494-497, mass_dens.ipp:7-120) */
int lcx_diag_wet_mass_dens(lcx_particles *, double rad, double sig0);
int lcx_diag_precip_rate(lcx_particles *);
int lcx_diag_max_rw(lcx_particles *);
/* outbuf(): pointer into library-owned HOST memory, n_cell reals, valid until the next
 * outbuf call (particles_ctor.ipp:83-92, fill_outbuf.ipp:13-37) */
int lcx_outbuf(lcx_particles *, const void **data, size_t *n);
/* get_attr(name) for "rw2","rd3","kappa","x","y","z" (fill_outbuf.ipp:40-77); call with
 * out == NULL to query the length */
int lcx_get_attr(lcx_particles *, const char *name, void *out, size_t cap, size_t *n);
int lcx_diag_puddle(lcx_particles *, double out[LCX_OUT_COUNT]);   /* particles_diag.ipp:411-421 */

/* ---- introspection / parity hooks (no reference counterpart; used by tests and bench) ---- */
int lcx_n_part(lcx_particles *, size_t *n);
int lcx_n_cell(lcx_particles *, size_t *n);
int lcx_real_kind(lcx_particles *, int *kind);
/* integer state: "n","ijk","sorted_id","sorted_ijk","count_ijk","count_num","cell_start" (as uint64) */
int lcx_get_state_u64(lcx_particles *, const char *name, unsigned long long *out, size_t cap, size_t *n);
/* real state not covered by get_attr: "vt","T","p","RH","eta","th","rv","rhod","dv","lambda_D","lambda_K",
 * "courant_x","courant_y","courant_z","vt_0" (as double whatever the real kind).
 * Names that begin with "raw_" ("raw_n","raw_ijk" as u64; "raw_rw2","raw_rd3","raw_kappa","raw_vt","raw_x","raw_y","raw_z","raw_tag" as
 * reals) return the STORAGE as it is -- its whole extent, dead slots (n == 0) included, nothing compacted or sorted on the way: reading
 * them does not disturb a production run, whereas every other particle-state getter first puts the storage into the reference's order.
 * "raw_collided" (u64, one value): living super-droplets that carry coalescence's invalid terminal velocity, i.e. the number of pairs that
 * collided in the last lcx_step_async (bench.py's coal-stress workload reports it).  "raw_sorted_id" (u64): the cell-sorted order as the
 * last sort left it -- after a step_sync with condensation, the shuffled order that the step's coalescence will pair up (an error while
 * the re-sort is still deferred). */
int lcx_get_state_real(lcx_particles *, const char *name, double *out, size_t cap, size_t *n);
/* overwrite particle state (all arrays of length n; x/y/z may be NULL for absent dimensions);
 * lets a test start the device from an oracle state */
int lcx_set_particles(lcx_particles *, size_t n, const unsigned long long *mult, const double *rd3,
                      const double *rw2, const double *kpa, const double *vt,
                      const double *x, const double *y, const double *z);
/* random-number replay: queue host-generated arrays that the next rand_un / rand_u01 calls consume
 * instead of the device generator (kind 0 = u01 reals, 1 = un 32-bit integers passed as doubles).
 * With replayed streams the integer results (sort permutation, multiplicities) are bit-comparable
 * with the reference CPU backends (src/detail/urand.hpp:24-86). */
int lcx_rng_replay_push(lcx_particles *, int kind, const double *data, size_t n);
int lcx_rng_replay_pending(lcx_particles *, size_t *n_arrays);
/* the reverse of the replay (needs LCX_DBG_TAG): what coalescence call `call` of the LAST lcx_step_async (0 .. sstp_coal-1) consumed of the
 * device's generator, so that the oracle can be run on the DEVICE's stream while the device stays on its production path (no replay,
 * production storage order, pre-shuffled deferred sort): which = 0: the uniforms u01[p] of the candidate pair that starts at position p
 * of the sorted order (n_part values, coal.ipp:369-450), 1: the shuffle key un[id] that ordered super-droplet `id` inside its cell
 * (hskpng_sort.ipp:28-47), by STORAGE index at coalescence time (storage extent values, dead slots included),
 * 2: the tags by storage index at coalescence time, 3: the cell index by storage index at coalescence time (0xFFFFFFFF: a dead slot).
 * out == NULL queries the length. */
int lcx_rng_dump(lcx_particles *, int call, int which, double *out, size_t cap, size_t *n);
/* overwrite ONE per-super-droplet attribute in storage order, the whole storage extent (test hook; "tag" only, needs LCX_DBG_TAG: a test
 * that follows droplets across the slabs of a decomposed domain gives them tags that are unique over all slabs) */
int lcx_set_state_real(lcx_particles *, const char *name, const double *data, size_t n);
/* run single housekeeping stages (for stage-level parity tests) */
int lcx_stage(lcx_particles *, const char *stage, const lcx_opts_t *opts);
/* per-stage device time of the last step in ms: fills names/values up to cap, returns count in *n */
int lcx_timings(lcx_particles *, const char **names, double *ms, size_t cap, size_t *n);
/* on: 0 off; 1 every stage (the stages then run one after the other: the in-cell ranking, which otherwise runs on a stream of its own
 * next to the per-cell finish and the terminal velocities, stays on the object's stream); 2 the condensation kernel's stage only */
int lcx_set_profiling(lcx_particles *, int on);

/* ---- 1-D domain decomposition (replaces impl_multi_gpu/..._step_async_and_copy.ipp:28-206 and
 *      distributed_memory/particles_impl_{pack,unpack}.ipp): after lcx_step_async() on a handle
 *      created with bcond_lft/rgt = distmem, the host exchanges packed migrants with its neighbours
 *      (RCCL send/recv) and finishes the step with lcx_migrate_finish(). ---- */
/* counts of SDs that left through the left / right face */
int lcx_migrate_counts(lcx_particles *, size_t *n_lft, size_t *n_rgt);
/* pack migrants of one side (0 = left, 1 = right) into a device buffer of
 * lcx_migrate_record_bytes() * count bytes, attribute-major: n[count] then rd3,rw2,kpa,vt,x,(y),(z)[count];
 * x is re-based to the receiver's frame (pack.ipp:14-26,110-133) given the receiver's edge x_rmt */
size_t lcx_migrate_record_bytes(lcx_particles *);
int lcx_migrate_pack(lcx_particles *, int side, double x_rmt, void *dev_buf, size_t cap_bytes);
/* append `count` immigrants from a device buffer (unpack.ipp:50-143) */
int lcx_migrate_unpack(lcx_particles *, const void *dev_buf, size_t count);
/* flag emigrants n=0 and run post_copy (post_copy.ipp:18-35) */
int lcx_migrate_finish(lcx_particles *, const lcx_opts_t *);
/* ---- the same exchange driven from the DEVICE, for one process per GPU (what particles_t<real_t, multi_CUDA>::step_async does between
 *      the devices of one process, impl_multi_gpu/..._step_async_and_copy.ipp:28-206; replaces the MPI flavour
 *      distributed_memory/particles_impl_mpi_exchange.ipp:20-330): the migrant counts never visit the host, one host synchronisation
 *      per step.  A message = 256-byte header {count, overflow flag, records the sender ships in the first part of its NEXT message}
 *      + the records in tiles of 256 super-droplets, each tile attribute-major (csrc/lcx_kernels.hpp, k_pack_dev), so that any whole
 *      number of tiles behind the header is a self-contained prefix: a transport that must fix the size of a message before its count
 *      is known (RCCL send / recv) ships a first part of agreed size and the rest only in the rare step whose count exceeds it.
 *   lcx_exch_enable   once, before init: two outboxes + two inboxes of lcx_exch_message_bytes(*cap_rec) bytes each on the object's device
 *                     (nx_min = x-planes of the thinnest slab of the decomposition: one capacity for all ranks); from then on
 *                     lcx_step_async() leaves the emigrant lists and their counts on the device
 *   lcx_exch_buffers  ptrs[4] = {outbox to the left, outbox to the right, inbox from the left, inbox from the right}
 *   lcx_exch_pack     after lcx_step_async(): emigrants -> outboxes (x re-based to the receiver's frame, pack.ipp:14-26), their
 *                     multiplicities cleared; queued on the object's stream (lcx_stream), no host synchronisation
 *   lcx_exch_sort_interior  (optional, right after the transport has been STARTED) queues what does not depend on the neighbours --
 *                     scan, scatter and in-cell ranking of the slab's interior cells -- so that it runs while the messages travel
 *   [transport: outbox to the left -> the left neighbour's "inbox from the right", and vice versa -- header + have_* records at least]
 *   lcx_exch_unpack   immigrants of both inboxes -> storage (left neighbour's first, unpack.ipp:50-143), have_lft / have_rgt = records
 *                     of each message that have arrived; a message whose count exceeds that is not touched; queued, no host sync
 *   lcx_exch_finish   the step's ONE host synchronisation: rec[12] = {dead, out_lft, out_rgt, in_lft, in_rgt, flags, crowded cells,
 *                     largest cell, shift, next_cap_lft, next_cap_rgt, 0}; *complete = 0: a message had not arrived in full, nothing
 *                     was unpacked -- ship the remaining tiles, call lcx_exch_unpack and lcx_exch_finish again; else post_copy ran
 *   lcx_stream        the hipStream_t all of the object's work is queued on (for ordering a transport against it without host syncs) */
int lcx_exch_enable(lcx_particles *, int nx_min, size_t *cap_rec);
int lcx_exch_buffers(lcx_particles *, void *ptrs[4]);
size_t lcx_exch_message_bytes(lcx_particles *, size_t n_rec);
int lcx_exch_pack(lcx_particles *, int has_lft, double lft_x1, int has_rgt, double rgt_x0, unsigned next_cap_lft, unsigned next_cap_rgt);
int lcx_exch_sort_interior(lcx_particles *);
int lcx_exch_unpack(lcx_particles *, int from_lft, int from_rgt, unsigned have_lft, unsigned have_rgt);
int lcx_exch_finish(lcx_particles *, const lcx_opts_t *, unsigned rec[12], int *complete);
int lcx_stream(lcx_particles *, void **hip_stream);
/* Courant halo of pred_corr advection on a decomposed domain (xchng_courants.ipp:15-160).  which: 0 Cx, 1 Cy, 2 Cz;
 * side: 0 left, 1 right.  _count: reals per side (0: nothing to exchange); _pack: the interior planes next to `side`
 * (what that neighbour needs) -> device buffer; _unpack: device buffer received from the neighbour at `side` -> the halo
 * planes on that side.  Call after sync_in, before step_async. */
size_t lcx_courant_halo_count(lcx_particles *, int which);
int lcx_courant_halo_pack(lcx_particles *, int which, int side, void *dev_buf);
int lcx_courant_halo_unpack(lcx_particles *, int which, int side, const void *dev_buf);
/* device pointer helpers so that a host language without device allocation can stage buffers */
int lcx_dev_alloc(void **ptr, size_t bytes);
int lcx_dev_free(void *ptr);
int lcx_dev_copy(void *dst, const void *src, size_t bytes, int kind /*1 H2D, 2 D2H, 3 D2D*/);
int lcx_dev_sync(void);
/* parity hook: evaluate one of the device elementary functions on a host array of doubles
 * (which: 0 seeded cbrt, 1 reduced exp [the fast-mode growth rate uses these], 2 library cbrt, 3 library exp,
 *  4 refined reciprocal [the fast-mode root finder divides with it], 5 lean logarithm [fast-mode terminal velocities]) */
int lcx_math_probe(int which, const double *x_host, double *y_host, size_t n);
/* parity hook: raw output blocks of the library's counter-based generator, Philox4x32-10 (Salmon et al. 2011; what the reference
 * draws from cuRAND / mt19937 in src/detail/urand.hpp:24-86 is drawn here inside the consuming kernel).  ics = n triples
 * (index, call, seed): counter = {index lo, index hi, call lo, call hi}, key = {seed lo, seed hi}; out = 4 n words.
 * on_device = 0 evaluates the same routine on the host (needs no GPU). */
int lcx_philox_probe(const unsigned long long *ics, size_t n, unsigned int *out, int on_device);
/* host-side scalar evaluation of the formula library, = the functions the reference's Python module exposes as
 * libcloudphxx.common (ref: bindings/python/common.hpp:19-172, lib.cpp:129-144): name is one of
 *   th_dry2std(th_dry,r) th_std2dry(th_std,r) exner(p) p_v(p,r) p_vs(T) r_vs(T,p) p_vs_tet(T) l_v(T) T(th,rhod)
 *   p(rhod,r,T) visc(T) rw3_cr(rd3,kappa,T) S_cr(rd3,kappa,T) p_hydro(z,th_0,r_0,z_0,p_0) rhod(p,th_std,rv)
 * or a constant (no arguments): R_d R_v c_pd c_pv c_pw g p_1000 eps rho_stp rho_w.  Needs no GPU. */
int lcx_common_eval(const char *name, const double *args, int n_args, double *out);

#ifdef __cplusplus
}
#endif
#endif
