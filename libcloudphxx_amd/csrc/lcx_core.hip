// lcx_core.hip -- host orchestration of the HIP kernels + the C ABI (include/lcx.h).
//
// Particles<real_t> owns the device state of one particles_t<real_t, HIP> object and implements the
// reference's time-step orchestration (src/particles_step.ipp, src/particles_init.ipp, src/particles_diag.ipp)
// as a sequence of kernel launches on one HIP stream.  The C ABI at the bottom is what the host-language
// shims bind (C++: include/libcloudphxx_amd/lgrngn/particles.hpp, Python: libcloudphxx_amd/lgrngn.py).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>
#include <functional>

#include <atomic>
#include "../../include/lcx.h"
#include "lcx_kernels.hpp"
#include "lcx_pool.hpp"

// every kernel launch of this translation unit and every host wait for a stream are COUNTED (lcx_get_state_u64 "raw_launches"): what a
// step costs a thin slab or a 2-D set-up is its launches and its waits, not its bytes (bench.py's c2 leg reports both per step)
namespace lcx { static std::atomic<unsigned long long> g_launches{0}, g_host_waits{0}; }
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...) \
  do { lcx::g_launches.fetch_add(1, std::memory_order_relaxed); kernelName<<<(numBlocks), (numThreads), (memPerBlock), (streamId)>>>(__VA_ARGS__); } while (0)

namespace lcx {

struct lcx_error : std::runtime_error { using std::runtime_error::runtime_error; };

#define HIPCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
  throw lcx_error(std::string("libcloudph++ (HIP): ") + hipGetErrorString(e_) + " at " #expr); } while (0)

static inline unsigned nblk(size_t n, unsigned bs = BS) { return unsigned((n + bs - 1) / bs); }

template <class T> struct DevBuf {
  T *p = nullptr; size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete; DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { release(); }
  void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
  void alloc(size_t m) { if (m <= n && p) return; release(); HIPCHK(hipMalloc((void **)&p, std::max<size_t>(m, 1) * sizeof(T))); n = std::max<size_t>(m, 1); }
  void alloc_zero(size_t m, hipStream_t s) { alloc(m); HIPCHK(hipMemsetAsync(p, 0, n * sizeof(T), s)); }
  // fine-grained device memory: coherent with writes that arrive from ANOTHER device while kernels of this one run before and
  // after them (the exchange inboxes that a neighbour's pack kernel fills over xGMI).  Returns false when the runtime refuses and
  // plain (coarse-grained) memory was taken instead: such an inbox must only be filled by copies (hipMemcpyPeerAsync), never by a
  // neighbour's kernel -- the caller routes its senders through the staged path then (lcx_multi.hpp)
  bool alloc_finegrained(size_t m)
  {
    release();
    bool fine = true;
    if (hipExtMallocWithFlags((void **)&p, std::max<size_t>(m, 1) * sizeof(T), hipDeviceMallocFinegrained) != hipSuccess) {
      (void)hipGetLastError(); p = nullptr; fine = false;
      HIPCHK(hipMalloc((void **)&p, std::max<size_t>(m, 1) * sizeof(T)));
    }
    n = std::max<size_t>(m, 1);
    return fine;
  }
  void swap(DevBuf &o) { std::swap(p, o.p); std::swap(n, o.n); }
};

// ---- efficiency tables (numeric data files, see tools/extract_efficiency_tables.py) ----
// looked up in $LCX_DATA_DIR if set, else in ../data next to this shared object (csrc/liblcx_hip.so -> data/), so that a
// C / C++ consumer finds them from any working directory
static std::string data_dir()
{
  if (const char *dir = getenv("LCX_DATA_DIR")) return dir;
  Dl_info info;
  if (dladdr(reinterpret_cast<const void *>(&data_dir), &info) && info.dli_fname) {
    std::string so(info.dli_fname);
    const size_t slash = so.rfind('/');
    return (slash == std::string::npos ? std::string(".") : so.substr(0, slash)) + "/../data";
  }
  return "libcloudphxx_amd/data";
}
static bool load_efficiency_table(int kernel, std::vector<double> &tab, double &r_max, std::string &path)
{
  path = data_dir() + "/kernel_eff_" + std::to_string(kernel) + ".f64";
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) return false;
  double hdr[2];
  bool ok = fread(hdr, sizeof(double), 2, f) == 2;
  if (ok) { tab.resize(size_t(hdr[1])); ok = fread(tab.data(), sizeof(double), tab.size(), f) == tab.size(); r_max = hdr[0]; }
  fclose(f);
  return ok;
}

struct IParticles {
  virtual ~IParticles() {}
  virtual void bind() {}      // make the object's device current on the calling thread
  virtual int real_kind() const = 0;
  virtual void init(const lcx_arrinfo_t *th, const lcx_arrinfo_t *rv, const lcx_arrinfo_t *rhod, const lcx_arrinfo_t *p,
                    const lcx_arrinfo_t *cx, const lcx_arrinfo_t *cy, const lcx_arrinfo_t *cz) = 0;
  virtual void sync_in(const lcx_arrinfo_t *th, const lcx_arrinfo_t *rv, const lcx_arrinfo_t *rhod, const lcx_arrinfo_t *cx,
                       const lcx_arrinfo_t *cy, const lcx_arrinfo_t *cz, const lcx_arrinfo_t *diss) = 0;
  virtual void step_cond(const lcx_opts_t &, const lcx_arrinfo_t *th, const lcx_arrinfo_t *rv) = 0;
  // particles_t::step_sync = sync_in + step_cond (particles_step.ipp:15-29); an implementation may interleave the two where that
  // cannot be observed
  virtual void step_sync(const lcx_opts_t &o, const lcx_arrinfo_t *th, const lcx_arrinfo_t *rv, const lcx_arrinfo_t *rhod, const lcx_arrinfo_t *cx,
                         const lcx_arrinfo_t *cy, const lcx_arrinfo_t *cz, const lcx_arrinfo_t *diss) { sync_in(th, rv, rhod, cx, cy, cz, diss); step_cond(o, th, rv); }
  virtual void step_async(const lcx_opts_t &) = 0;
  virtual void diag_cell(int which) = 0;               // 0 p, 1 T, 2 RH
  virtual void diag_vel_div() = 0;
  virtual void diag_sd_conc() = 0;
  virtual void diag_select(int mode, int cons, int attr, double a, double b) = 0;   // attr: 0 rd3, 1 rw2, 2 kpa
  virtual void diag_mom(int attr, double power) = 0;
  virtual void diag_precip_rate() = 0;
  virtual void diag_act(int which) = 0;             // 0 RH >= Sc, 1 rw >= rc
  virtual void diag_wet_mass_dens(double rad, double sig0) = 0;
  virtual void diag_max_rw() = 0;
  virtual void outbuf(const void **data, size_t *n) = 0;
  virtual void get_attr(const char *name, void *out, size_t cap, size_t *n) = 0;
  virtual void diag_puddle(double *out) = 0;
  virtual size_t n_part() = 0;
  virtual size_t n_cell() = 0;
  virtual void get_state_u64(const char *name, unsigned long long *out, size_t cap, size_t *n) = 0;
  virtual void get_state_real(const char *name, double *out, size_t cap, size_t *n) = 0;
  virtual void set_particles(size_t n, const unsigned long long *mult, const double *rd3, const double *rw2, const double *kpa,
                             const double *vt, const double *x, const double *y, const double *z) = 0;
  virtual void rng_replay_push(int kind, const double *data, size_t n) = 0;
  virtual size_t rng_replay_pending() = 0;
  virtual void rng_dump(int call, int which, double *out, size_t cap, size_t *n) = 0;
  virtual void set_state_real(const char *name, const double *data, size_t n) = 0;
  virtual void stage(const char *name, const lcx_opts_t *opts) = 0;
  virtual void timings(const char **names, double *ms, size_t cap, size_t *n) = 0;
  virtual void set_profiling(int on) = 0;
  virtual void migrate_counts(size_t *l, size_t *r) = 0;
  virtual size_t migrate_record_bytes() = 0;
  virtual void migrate_pack(int side, double x_rmt, void *buf, size_t cap) = 0;
  virtual void migrate_unpack(const void *buf, size_t count) = 0;
  virtual void migrate_finish(const lcx_opts_t &) = 0;
  virtual size_t courant_halo_count(int which) = 0;
  virtual void courant_halo_copy(int which, int side, void *buf, bool pack) = 0;
  // device-driven exchange for one process per GPU (include/lcx.h, lcx_exch_*); a multi-device object does it inside step_async
  [[noreturn]] static void no_exch() { throw std::runtime_error("libcloudph++: lcx_exch_* drives a single-device object (one process per GPU)"); }
  virtual size_t x_enable(int) { no_exch(); }
  virtual void x_buffers(void **) { no_exch(); }
  virtual size_t x_message_bytes(size_t) { no_exch(); }
  virtual void x_pack(bool, double, bool, double, unsigned, unsigned) { no_exch(); }
  virtual void x_unpack(bool, bool, unsigned, unsigned) { no_exch(); }
  virtual void x_sort_interior() { no_exch(); }
  virtual bool x_finish(const lcx_opts_t &, unsigned *) { no_exch(); }
  virtual void *stream() { return nullptr; }
};

template <class real_t>
struct Particles : IParticles {
  using T = real_t;
  // ---- options (deep copies) ----
  lcx_opts_init_t o;
  std::vector<lcx_distro_t> distros;
  std::vector<lcx_dry_size_t> sizes; int n_size_keys = 0;   // dry_sizes.size() of the reference = number of (kappa, rd_insol) keys
  std::vector<double> kernel_parameters_h, w_LS_h, conc_factor_h;
  int n_dims; size_t ncell, npart = 0, nphys = 0, cap;   // npart: living SDs (API); nphys: storage extent incl. not yet compacted dead SDs
  bool eager_compact = false, fused_pending = false; size_t n_before_unpack = 0; int steps_since_reorder = 0;
  grid_t g;
  // ---- order-of-operation flags (particles_impl.ipp:32) ----
  bool init_called = false, should_now_run_async = false, should_now_run_cond = false, selected_before_counting = false;
  bool var_rho = false, sorted = false, sorted_shuffled = false;
  // production order (no replayed stream): the re-sort at the end of step_async ranks the cells straight by the NEXT coalescence's
  // random keys (un[id], id) -- nothing between the two needs ids ascending inside a cell (condensation treats a droplet on its
  // own, the fast per-cell sums take any fixed order, and this one is as deterministic as the other), so the ascending-id ranking
  // of every step (1.0 of 17.3 ms on C3) is not run at all and coalescence finds its shuffled order ready.  The reference's order
  // (plain sort, then shuffle at coalescence) stays in every parity run and with opts_init.reorder_every < 0.
  bool shuffle_fresh = false, last_async_coal = false;
  // an SD that coalescence uses up is marked in ijk by k_coal when the step ends in the fused move + re-index (which then reads no
  // multiplicities: 8 B per SD less); zero multiplicities of other origin (initialisation, set_particles) make the next move look
  bool coal_marks_dead = false, zero_n_unmarked = true;
  int sstp_cond, sstp_coal; bool allow_sstp_cond, pure_const_multi; double dt;
  int adve_scheme, halo = 0;      // halo: x-planes of Courant halo on each side (pred_corr)
  hipStream_t st = nullptr;
  // ---- particle attributes (two buffer sets: stable compaction writes from one into the other) ----
  struct Attrs { DevBuf<n_t> n; DevBuf<T> rd3, rw2, kpa, vt, x, y, z, ext[MAX_EXT]; } A, B;
  // per-particle condensation substepping (exact_sstp_cond): the private rv, th, rhod(, p) of a super-droplet and rc2 are
  // further attributes (ext[]) that are compacted and migrate with it
  // the initial sampling may have its own seed (opts_init.rng_seed_init_switch; particles_ctor.ipp / particles_init.ipp: the
  // generator is re-seeded with rng_seed at the end of init)
  bool in_init = false;
  long long seed_now() const { return in_init && o.rng_seed_init_switch ? o.rng_seed_init : o.rng_seed; }
  bool replay_used = false;   // a parity run: storage stays in the reference's id order (see opts_init.reorder_every)
  bool dbg(unsigned f) const { return (o.dbg_flags & f) != 0; }      // test / measurement switches (include/lcx.h, lcx_dbg): read from the options, never from the environment
  bool no_cond_pre = dbg(LCX_DBG_NO_COND_PRE);   // test switch: evaluate the per-cell set-up per droplet instead
  uint64_t cells_version = 0;   // order_cells: the list of cells above CELLRANK_MAX is remembered per cell_start
  bool exact = false, use_rc2 = false; int sstp_cond_act = 1, n_ext = 0, ix_rv = -1, ix_th = -1, ix_rh = -1, ix_p = -1, ix_rc2 = -1;
  DevBuf<T> pp_dlt[4], pp_rw3s, pp_dst_rv, pp_dst_th;
  int ix_up = -1, ix_vp = -1, ix_wp = -1, ix_ssp = -1, ix_dot_ssp = -1;
  int ix_tag = -1;                    // LCX_DBG_TAG: a persistent tag per super-droplet (parity tests match droplets across re-orderings by it)
  int ix_ict = -1;                    // opts_init.diag_incloud_time: time spent activated, travels with the SD (particles_impl.ipp:475-476)
  std::vector<double> SGS_mix_len_h; DevBuf<T> SGS_mix_len, diss_rate, tau_cell, tau_rlx;
  bool turb() const { return o.turb_adve_switch || o.turb_cond_switch; }
  bool turb_any() const { return turb() || o.turb_coal_switch; }     // diss_rate is synced in for any of the three (particles_step.ipp:74-78,121)
  DevBuf<uint32_t> ijk, sorted_id, sorted_ijk, rank, cell_cnt, cell_start, tile_sums, scan_total, big_list, step_cnt, mig_ids[2];
  // The cell-sorted order begins at sorted_id.p + sort_base.  0 for an object without neighbours.  A slab with neighbours sorts its
  // interior while their messages travel (exch_sort_interior) and lets the order begin `shift` = (immigrants of the left boundary
  // planes) below a fixed headroom, instead of moving the interior's entries once that number is known: sort_base = headroom - shift.
  size_t sort_base = 0, sort_headroom = 0;
  DevBuf<uint32_t> sorted_alt;          // (exchange only) the in-cell ranking's output while `rank` still holds the boundary SDs' arrival ranks
  // The in-cell ranking of a carried re-sort runs on a stream of its own (st_rank) next to what follows the condensation kernel on `st`:
  // the per-cell finish, sync_out and, across the step boundary, the terminal velocities.  Those are bound by memory, the ranking by
  // integer instructions and LDS, and none of them reads the sorted order.  Whoever does -- every user of sid() / sijk() / rnk(), the
  // scan that rewrites cell_start, a host synchronisation -- makes `st` wait for the ranking first (join_rank): ordered by construction.
  hipStream_t st_rank = nullptr; hipEvent_t ev_fork = nullptr, ev_rank = nullptr; mutable bool rank_pending = false;
  void join_rank() const { if (rank_pending) { rank_pending = false; HIPCHK(hipStreamWaitEvent(st, ev_rank, 0)); } }
  uint32_t *sid() const { join_rank(); return sorted_id.p + sort_base; }
  uint32_t *sijk() const { join_rank(); return sorted_ijk.p + sort_base; }
  uint32_t *rnk() const { join_rank(); return rank.p; }
  DevBuf<uint8_t> mig, cond_pre, wave_flag, cond_records; DevBuf<uint32_t> defer_cnt, wg_mig, cond_listed;
  const bool use_wave_flags = !dbg(LCX_DBG_NO_WAVE_FLAGS);
  void alloc_mig() { mig.alloc((cap + BS - 1) / BS * BS + 16); mig_ids[0].alloc(cap); mig_ids[1].alloc(cap); wg_mig.alloc(3 * (size_t(nblk(cap)) + 1)); }      // (+ the two offset arrays)
  DevBuf<uint64_t> sort_scratch;
  DevBuf<T> col, m3_before, m3_after, n_filtered, fvals;
  // ---- cell fields ----
  DevBuf<T> rhod, th, rv, p, Tk, RH, eta, dv, lambda_D, lambda_K, sstp_tmp_rv, sstp_tmp_th, sstp_tmp_rh, rw_mom3, count_mom;
  DevBuf<T> courant_x, courant_y, courant_z, w_LS, conc_factor, vt_0, kparams;
  size_t n_cx = 0, n_cy = 0, n_cz = 0;
  DevBuf<T> stage_dev; std::vector<T> stage_host, outbuf_h;
  DevBuf<double> puddle_partial, puddle_sum, puddle_acc;
  DevBuf<int> d_flag;
  // step_cnt: [0] dead SDs counted by the fused move, [1] number of crowded cells, [2] largest cell occupancy -- one read-back
  unsigned int *d_dead_p() { return step_cnt.p; }
  uint32_t *big_meta_p() { return step_cnt.p + 1; }      // filled by list_big_from_hist only (cleared behind every scan)
  uint32_t *big_meta_own_p() { return step_cnt.p + 4; }  // order_cells' own listing (cleared by itself)
  void *pinned = nullptr;      // 256 B of page-locked host memory for the small per-step read-backs (counts, sums)
  double puddle[LCX_OUT_COUNT];
  bool count_mom_valid_all = true;
  double kernel_r_max = 0; int n_user_params = 0;
  double log_rd_min = 0, log_rd_max = 0, multiplier = 0;
  T eps_tol;
  vt_cfg vtc;
  uint64_t rng_call = 0;
  size_t lft_count = 0, rgt_count = 0;
  // replay queues: device arrays consumed by the next rand_u01 / rand_un
  struct Replay { int kind; std::unique_ptr<DevBuf<T>> u01; std::unique_ptr<DevBuf<uint32_t>> un; size_t n; };
  std::deque<Replay> replay;
  std::vector<std::unique_ptr<DevBuf<T>>> replay_keep_T; std::vector<std::unique_ptr<DevBuf<uint32_t>>> replay_keep_u;
  // profiling
  int profiling = 0;
  std::vector<std::pair<std::string, std::pair<hipEvent_t, hipEvent_t>>> prof_events;
  std::map<std::string, double> prof_ms; std::vector<std::string> prof_order;

  int real_kind() const override { return int(sizeof(T)); }
  void bind() override { if (o.dev_id >= 0) (void)hipSetDevice(o.dev_id); }
  bool distmem() const { return o.bcond_lft == 1 || o.bcond_rgt == 1; }
  static int m1(int n) { return n == 0 ? 1 : n; }

  // ------------------------------------------------------------------------------------------
  explicit Particles(const lcx_opts_init_t &oi) : o(oi)
  {
    if (oi.chem_switch || oi.ice_switch || oi.rlx_switch || oi.src_type)
      throw lcx_error("libcloudph++: option outside the accelerated hot path (chem/ice/src/rlx)");
    if (oi.n_sd_max >= (1ull << 32)) throw lcx_error("libcloudph++: n_sd_max must be < 2^32 per device (32-bit super-droplet ids)");
    distros.assign(oi.dry_distros, oi.dry_distros + oi.n_dry_distros);
    sizes.assign(oi.dry_sizes, oi.dry_sizes + oi.n_dry_sizes);
    for (size_t d = 0; d < sizes.size(); ++d)
      if (d == 0 || sizes[d].kappa != sizes[d - 1].kappa || sizes[d].rd_insol != sizes[d - 1].rd_insol) ++n_size_keys;
    kernel_parameters_h.assign(oi.kernel_parameters, oi.kernel_parameters + oi.n_kernel_parameters);
    w_LS_h.assign(oi.w_LS, oi.w_LS + oi.n_w_LS);
    conc_factor_h.assign(oi.aerosol_conc_factor, oi.aerosol_conc_factor + oi.n_aerosol_conc_factor);
    o.dry_distros = nullptr; o.kernel_parameters = nullptr; o.w_LS = nullptr; o.aerosol_conc_factor = nullptr; o.dry_sizes = nullptr;
    n_user_params = oi.n_kernel_parameters;
    n_dims = oi.nx / m1(oi.nx) + oi.ny / m1(oi.ny) + oi.nz / m1(oi.nz);             // particles_impl.ipp:335-345
    ncell = size_t(m1(oi.nx)) * m1(oi.ny) * m1(oi.nz);
    if (ncell >= (1ull << 32)) throw lcx_error("libcloudph++: n_cell must be < 2^32");
    g = grid_t{oi.nx, oi.ny, oi.nz, n_dims, double(T(oi.dx)), double(T(oi.dy)), double(T(oi.dz))};
    sstp_cond = oi.sstp_cond; sstp_coal = oi.sstp_coal;
    allow_sstp_cond = oi.sstp_cond > 1 || oi.sstp_cond_act > 1;
    sstp_cond_act = oi.sstp_cond_act;
    exact = allow_sstp_cond && oi.exact_sstp_cond;                                   // particles_impl.ipp:452-459
    use_rc2 = oi.sstp_cond_act > 1 && allow_sstp_cond;                               // :488-491
    if (exact) { ix_rv = n_ext++; ix_th = n_ext++; ix_rh = n_ext++; if (oi.const_p) ix_p = n_ext++; }
    if (use_rc2) ix_rc2 = n_ext++;
    // SGS turbulence: velocity perturbations (turb_adve: one per dimension; turb_cond: the vertical one) and the supersaturation
    // perturbation with its tendency travel with the SD as well (particles_impl.ipp:461-473)
    if (oi.turb_adve_switch) { if (oi.nx) ix_up = n_ext++; if (oi.ny) ix_vp = n_ext++; if (oi.nz) ix_wp = n_ext++; }
    if (oi.turb_cond_switch) { if (ix_wp < 0) ix_wp = n_ext++; ix_ssp = n_ext++; ix_dot_ssp = n_ext++; }
    if (oi.diag_incloud_time) ix_ict = n_ext++;
    if (dbg(LCX_DBG_TAG)) ix_tag = n_ext++;
    SGS_mix_len_h.assign(oi.SGS_mix_len, oi.SGS_mix_len + oi.n_SGS_mix_len);
    o.SGS_mix_len = nullptr;
    pure_const_multi = (oi.sd_conc == 0) && (oi.sd_const_multi > 0 || oi.n_dry_sizes > 0);
    adve_scheme = oi.adve_scheme;
    halo = oi.adve_scheme == LCX_ADVE_PRED_CORR ? 2 : 0;                               // particles_impl.ipp:361
    if (o.n_x_tot == 0) o.n_x_tot = oi.nx;
    dt = oi.dt;
    eps_tol = eps_tolerance<T>(sizeof(T) * 8 / 4);                                   // src/detail/config.hpp:39
    vtc = vt_cfg{oi.terminal_velocity, double(T(std::log(5e-7))), double(T(std::log(3e-3))), 10000};   // config.hpp:27-38
    cap = size_t(oi.n_sd_max);
    eager_compact = dbg(LCX_DBG_EAGER_COMPACT);
    for (double &v : puddle) v = 0;
    if (oi.dev_id >= 0) HIPCHK(hipSetDevice(oi.dev_id));                             // particles_ctor.ipp:60-63
    HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    HIPCHK(hipHostMalloc(&pinned, 256, hipHostMallocDefault));
    alloc_attrs(A);
    for (int ix : {ix_up, ix_vp, ix_wp, ix_ssp, ix_dot_ssp, ix_ict}) if (ix >= 0) HIPCHK(hipMemsetAsync(A.ext[ix].p, 0, cap * sizeof(T), st));
    if (turb_any()) diss_rate.alloc_zero(ncell, st);
    if (turb()) {
      tau_cell.alloc_zero(ncell, st); tau_rlx.alloc_zero(ncell, st);
      std::vector<T> hm(SGS_mix_len_h.begin(), SGS_mix_len_h.end());
      SGS_mix_len.alloc(hm.size());
      h2d(SGS_mix_len.p, hm.data(), hm.size() * sizeof(T));
    }
    if (use_rc2) hipLaunchKernelGGL(k_fill<T>, dim3(nblk(cap)), dim3(BS), 0, st, A.ext[ix_rc2].p, cap, T(-1));   // detail::invalid, particles_impl.ipp:490
    ijk.alloc(cap); sorted_id.alloc(cap); sorted_ijk.alloc(cap); rank.alloc(cap);
    cell_cnt.alloc_zero(ncell, st); cell_start.alloc_zero(ncell + 1, st);
    tile_sums.alloc(2 * (std::max(cap, ncell) / SCAN_TILE + 2)); scan_total.alloc(4);   // (two halves for the two migrant lists)
    big_list.alloc(ncell + 1); step_cnt.alloc_zero(8, st);        // (every cell can be listed: list_thr)
    m3_before.alloc(cap); m3_after.alloc(cap);
    if (oi.coal_switch) col.alloc(cap);
    for (DevBuf<T> *b : {&rhod, &th, &rv, &p, &Tk, &RH, &eta, &dv, &lambda_D, &lambda_K, &sstp_tmp_rv, &sstp_tmp_th, &sstp_tmp_rh, &rw_mom3, &count_mom})
      b->alloc_zero(ncell, st);
    d_flag.alloc_zero(1, st);
    puddle_partial.alloc(size_t(nblk(cap)) * 4); puddle_sum.alloc(4 + 256 * 4); puddle_acc.alloc_zero(4, st);
    outbuf_h.assign(ncell, T(0));
    if (distmem()) alloc_mig();
  }
  ~Particles() override
  {
    if (st_rank) { (void)hipStreamSynchronize(st_rank); (void)hipStreamDestroy(st_rank); (void)hipEventDestroy(ev_fork); (void)hipEventDestroy(ev_rank); }
    if (st) (void)hipStreamSynchronize(st);
    for (auto &e : prof_events) { (void)hipEventDestroy(e.second.first); (void)hipEventDestroy(e.second.second); }
    for (hipEvent_t e : prof_pool) (void)hipEventDestroy(e);
    if (st) (void)hipStreamDestroy(st);
    if (pinned) (void)hipHostFree(pinned);
    // (the copy stream reads the staging areas: it is drained before they are freed)
    if (st_copy) { (void)hipStreamSynchronize(st_copy); (void)hipStreamDestroy(st_copy); }
    for (HostStage *h : {&hstage_in, &hstage_out}) if (h->p) (void)hipHostFree(h->p);
    for (hipEvent_t e : out_events) (void)hipEventDestroy(e);
    if (ev_copy) (void)hipEventDestroy(ev_copy);
  }
  // small device -> host read-back through page-locked memory (a pageable destination makes the copy synchronous and slow)
  template <class S> void read_back(S *dst, const S *src, size_t n)
  {
    HIPCHK(hipMemcpyAsync(pinned, src, n * sizeof(S), hipMemcpyDeviceToHost, st));
    sync();
    memcpy(dst, pinned, n * sizeof(S));
  }

  // host -> device on OUR stream.  (A plain hipMemcpy runs on the null stream, which a non-blocking stream does not wait
  // for: it may return with the DMA from its staging buffer still in flight and the next kernel on `st` could read stale
  // device memory.)
  void h2d(void *dst, const void *src, size_t bytes)
  {
    if (!bytes) return;
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
    sync();
  }
  void alloc_attrs(Attrs &a)
  {
    a.n.alloc(cap); a.rd3.alloc(cap); a.rw2.alloc(cap); a.kpa.alloc(cap); a.vt.alloc(cap);
    if (o.nx) a.x.alloc(cap); if (o.ny) a.y.alloc(cap); if (o.nz) a.z.alloc(cap);
    for (int e = 0; e < n_ext; ++e) a.ext[e].alloc(cap);
  }
  attr_set<T> aset(Attrs &a)
  {
    attr_set<T> s{a.n.p, a.rd3.p, a.rw2.p, a.kpa.p, a.vt.p, a.x.p, a.y.p, a.z.p, {}, n_ext};
    for (int e = 0; e < n_ext; ++e) s.ext[e] = a.ext[e].p;
    return s;
  }
  void sync() { join_rank(); g_host_waits.fetch_add(1, std::memory_order_relaxed); HIPCHK(hipStreamSynchronize(st)); hstage_busy = false; }
  // the end of step_cond: th and rv are written on `st`; a ranking on st_rank goes on while the host queues step_async
  void sync_results_only() { g_host_waits.fetch_add(1, std::memory_order_relaxed); HIPCHK(hipStreamSynchronize(st)); hstage_busy = false; }

  // ---- profiling ranges (hipEvents on OUR stream) ----
  // profiling: 0 off; 1 every stage; 2 the condensation kernel's stage only -- two event records per step instead of fifty: every record
  // is a packet between two kernels, and on a 16-plane slab of C3 the full set costs 0.11 ms of a 1.33 ms step (bench.py times with
  // level 2 and takes the stage table from a few extra steps at level 1)
  struct Range {
    Particles *self; const char *name; hipEvent_t a = nullptr, b = nullptr;
    Range(Particles *s, const char *nm) : self(s), name(nm)
    {
      if (self->profiling == 1 || (self->profiling == 2 && (!strcmp(nm, "cond") || !strcmp(nm, "cond_listed")))) { a = self->prof_event(); b = self->prof_event(); (void)hipEventRecord(a, self->st); }
    }
    ~Range() { if (a) { (void)hipEventRecord(b, self->st); self->prof_events.push_back({std::string(name), {a, b}}); } }
  };
  std::vector<hipEvent_t> prof_pool;
  hipEvent_t prof_event()
  {
    hipEvent_t e = nullptr;
    if (!prof_pool.empty()) { e = prof_pool.back(); prof_pool.pop_back(); } else (void)hipEventCreate(&e);
    return e;
  }
  void collect_profile()
  {
    if (!profiling) return;
    sync();
    for (auto &e : prof_events) {
      float ms = 0; (void)hipEventElapsedTime(&ms, e.second.first, e.second.second);
      if (!prof_ms.count(e.first)) prof_order.push_back(e.first);
      prof_ms[e.first] += ms;
      prof_pool.push_back(e.second.first); prof_pool.push_back(e.second.second);
    }
    prof_events.clear();
  }
  void set_profiling(int on) override { collect_profile(); profiling = on < 0 ? 0 : on > 2 ? 1 : on; prof_ms.clear(); prof_order.clear(); }
  void timings(const char **names, double *ms, size_t capn, size_t *n) override
  {
    collect_profile();
    size_t k = 0;
    for (auto &nm : prof_order) { if (k >= capn) break; names[k] = nm.c_str(); ms[k] = prof_ms[nm]; ++k; }
    *n = k;
  }

  // ---- device exclusive scan: out[0..m) = exclusive scan of in, out[m] (if out_last) = total; returns nothing (async) ----
  // total_slot: where the grand total is kept (default scan_total[0]; the emigrant counts of a slab with neighbours live in
  // scan_total[0..1] from the move to the end of the exchange, so a scan in between names another slot)
  void exclusive_scan(const uint32_t *in, uint32_t *out, size_t m, uint32_t *out_last, uint32_t *zero_in = nullptr, uint32_t *zero_words = nullptr,
                      int n_zero_words = 0, uint32_t *total_slot = nullptr)
  {
    join_rank();                       // (cell_start, which a ranking on st_rank may still be reading)
    uint32_t *total = total_slot ? total_slot : scan_total.p;
    const size_t tiles = (m + SCAN_TILE - 1) / SCAN_TILE;
    if (tiles == 0) { if (out_last) HIPCHK(hipMemsetAsync(out_last, 0, sizeof(uint32_t), st)); HIPCHK(hipMemsetAsync(total, 0, sizeof(uint32_t), st)); return; }
    hipLaunchKernelGGL(k_scan_tiles, dim3(unsigned(tiles)), dim3(BS), 0, st, in, out, tile_sums.p, m);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, st, tile_sums.p, tiles, total);
    hipLaunchKernelGGL(k_scan_add, dim3(nblk(m)), dim3(BS), 0, st, out, tile_sums.p, m, total, out_last, zero_in, zero_words, n_zero_words);
  }

  // ------------------------------------------------------------------------------------------
  // Eulerian <-> Lagrangian sync (particles_impl_sync.ipp:15-68, init_e2l.ipp:34-114)
  // ------------------------------------------------------------------------------------------
  static bool is_null(const lcx_arrinfo_t *a) { return !a || !a->data || !a->strides; }
  void arr_geom(const lcx_arrinfo_t *a, int ex, int ey, int ez, int &n1, int &n2, long &s0, long &s1, long &s2) const
  {
    (void)ex;
    n1 = o.ny + ey; n2 = o.nz + ez; s0 = s1 = s2 = 0;
    switch (n_dims) {
      case 0: break;
      case 1: s0 = 1; break;
      case 2: s0 = a->strides[0]; s1 = a->strides[1]; break;
      default: s0 = a->strides[0]; s1 = a->strides[1]; s2 = a->strides[2];
    }
  }
  // halo_planes > 0 (Courant numbers with pred_corr): the device array starts that many x-planes left of the user's array;
  // planes outside it wrap around the n_x_tot + ex planes cyclically (init_e2l.ipp:44-46,109-113)
  // device-resident caller arrays are gathered / scattered by ONE launch per call (flush_sync_jobs)
  sync_jobs<T> jobs_in{}, jobs_out{};
  void add_job(sync_jobs<T> &J, T *lib, const T *user, size_t n, int n1, int n2, long s0, long s1, long s2, long i_off, long wrap)
  {
    const int j = J.n_jobs++;
    if (j == 0) J.first_block[0] = 0;
    J.ndims = n_dims;
    J.dst[j] = lib; J.src[j] = user; J.n[j] = n; J.n1[j] = n1; J.n2[j] = n2; J.s0[j] = s0; J.s1[j] = s1; J.s2[j] = s2; J.i_off[j] = i_off; J.wrap[j] = wrap;
    J.first_block[j + 1] = J.first_block[j] + nblk(n);
  }
  void flush_sync_jobs()
  {
    if (jobs_in.n_jobs) hipLaunchKernelGGL((k_sync_multi<T, true>), dim3(jobs_in.first_block[jobs_in.n_jobs]), dim3(BS), 0, st, jobs_in);
    if (jobs_out.n_jobs) hipLaunchKernelGGL((k_sync_multi<T, false>), dim3(jobs_out.first_block[jobs_out.n_jobs]), dim3(BS), 0, st, jobs_out);
    jobs_in.n_jobs = jobs_out.n_jobs = 0;
  }
  void sync_in_arr(const lcx_arrinfo_t *a, DevBuf<T> &to, size_t n, int ex, int ey, int ez, int halo_planes = 0)
  {
    if (is_null(a)) return;
    int n1, n2; long s0, s1, s2;
    arr_geom(a, ex, ey, ez, n1, n2, s0, s1, s2);
    // on_device == 3 (set by the multi-device front end for per-slab arrays): a device array of THIS slab, indexed from 0; its
    // Courant halo wraps inside the slab and is then overwritten by the halo exchange
    const bool local = a->on_device == 3;
    const long wrap = halo_planes ? (local ? long(o.nx) : long(o.n_x_tot)) + ex : 0;
    if (a->on_device) {
      if (jobs_in.n_jobs == MAX_SYNC_JOBS) flush_sync_jobs();
      add_job(jobs_in, to.p, (const T *)a->data, n, n1, n2, s0, s1, s2, (local ? 0l : long(o.n_x_bfr)) - halo_planes, wrap);
      return;
    }
    const long ioff = long(o.n_x_bfr) - halo_planes;
    if (!dbg(LCX_DBG_HOST_SYNC_LOOP)) {
      // the caller's HOST array (what an unchanged icicle / UWLCM passes, particles_impl_sync.ipp:15-68): its rows are gathered into
      // page-locked staging by a few host threads, ONE asynchronous copy on the COPY stream takes the field into a device-side staging
      // area (it starts at once, whatever the object's own stream is still busy with -- the kernels of the previous step_async, or
      // this step's condensation for the Courant numbers), and a device-to-device copy on the object's stream, behind an event, puts
      // it into the library's array (flush_host_in).  Nothing waits here: the staging area is not the caller's memory, and the
      // caller's array has been read completely when this returns
      T *stg = stage_reserve(hstage_in, n);
      const size_t at = stg - (T *)hstage_in.p;      // (stage_reserve may have started over)
      host_copy_rows(true, stg, (T *)a->data, n, n1, n2, s0, s1, s2, ioff, wrap);
      need_copy_stream();
      HIPCHK(hipMemcpyAsync(dstage.p + at, stg, n * sizeof(T), hipMemcpyHostToDevice, st_copy));
      host_in_jobs.push_back(InJob{to.p, dstage.p + at, n});
      hstage_busy = true;
      return;
    }
    stage_host.resize(n);
    const T *d = (const T *)a->data;
    auto pl = [&](long i) { i += ioff; if (wrap) { if (i >= wrap) i -= wrap; else if (i < 0) i += wrap; } return i; };
    switch (n_dims) {
      case 0: stage_host[0] = d[0]; break;
      case 1: for (size_t c = 0; c < n; ++c) stage_host[c] = d[pl(long(c)) * s0]; break;
      case 2: for (size_t c = 0; c < n; ++c) stage_host[c] = d[pl(long(c / n2)) * s0 + long(c % n2) * s1]; break;
      default: for (size_t c = 0; c < n; ++c) stage_host[c] = d[pl(long(c / (size_t(n2) * n1))) * s0 + long((c / n2) % n1) * s1 + long(c % n2) * s2];
    }
    HIPCHK(hipMemcpyAsync(to.p, stage_host.data(), n * sizeof(T), hipMemcpyHostToDevice, st));
    sync();     // stage_host is reused by the next field
  }
  // ---- page-locked staging for host arrays + the host threads that fill / drain it.  One area per direction, sized once for all the
  // fields of a call (4 cell fields + the three Courant arrays in; th and rv out); `used` starts from zero in every call
  struct HostStage { void *p = nullptr; size_t cap = 0, used = 0; } hstage_in, hstage_out;
  bool hstage_busy = false;        // copies out of / into the staging areas are queued and not yet waited for
  std::unique_ptr<WorkerPool> hpool;
  void stage_begin() { if (hstage_busy) { flush_host_in(); sync(); } hstage_in.used = hstage_out.used = 0; }
  DevBuf<T> dstage;                // device-side twin of hstage_in
  struct InJob { T *lib; const T *stg; size_t n; };
  std::vector<InJob> host_in_jobs;
  hipStream_t st_copy = nullptr; hipEvent_t ev_copy = nullptr;
  void need_copy_stream()
  { if (!st_copy) { HIPCHK(hipStreamCreateWithFlags(&st_copy, hipStreamNonBlocking)); HIPCHK(hipEventCreateWithFlags(&ev_copy, hipEventDisableTiming)); } }
  // the fields that the copy stream has been given so far -> the library's arrays, on the object's stream behind the copies
  void flush_host_in()
  {
    if (host_in_jobs.empty()) return;
    HIPCHK(hipEventRecord(ev_copy, st_copy));
    HIPCHK(hipStreamWaitEvent(st, ev_copy, 0));
    for (const InJob &j : host_in_jobs) HIPCHK(hipMemcpyAsync(j.lib, j.stg, j.n * sizeof(T), hipMemcpyDeviceToDevice, st));
    host_in_jobs.clear();
  }
  T *stage_reserve(HostStage &h, size_t n)
  {
    if (h.used + n > h.cap / sizeof(T)) {
      // (more fields than planned for: wait, drain what the area still holds for the caller, start over)
      if (h.used) { flush_host_in(); sync(); if (&h == &hstage_out) finish_sync_out(); h.used = 0; }
      if (n > h.cap / sizeof(T)) {
        const size_t want = std::max(n, &h == &hstage_in ? 4 * ncell + n_cx + n_cy + n_cz : 2 * ncell) * sizeof(T);
        if (h.p) HIPCHK(hipHostFree(h.p));
        h.p = nullptr; h.cap = 0;
        HIPCHK(hipHostMalloc(&h.p, want, hipHostMallocDefault));
        h.cap = want;
        if (&h == &hstage_in) dstage.alloc(want / sizeof(T));
      }
    }
    T *r = (T *)h.p + h.used;
    h.used += n;
    return r;
  }
  int host_threads() const
  {
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    return int(std::max(1u, std::min({16u / unsigned(std::max(o.dev_count, 1)), hw, 16u})));
  }
  // rows of the library's array (its innermost extent, contiguous there) <-> the caller's strided array; in: caller -> dense
  void host_copy_rows(bool in, T *dense, T *user, size_t n, int n1, int n2, long s0, long s1, long s2, long ioff, long wrap)
  {
    const int nd = n_dims;
    // (a 1-D domain without wrap is ONE row of n elements at stride s0, not n rows of one element)
    const bool one_row = nd == 1 && !wrap;
    const size_t inner = nd >= 2 ? size_t(n2) : one_row ? n : 1, rows = nd == 0 ? 1 : n / inner;
    const long s_in = nd == 3 ? s2 : nd == 2 ? s1 : one_row ? s0 : 1;
    auto work = [=](size_t r0, size_t r1) {
      for (size_t r = r0; r < r1; ++r) {
        long i = nd == 3 ? long(r / size_t(n1)) : long(r);
        i += ioff; if (wrap) { if (i >= wrap) i -= wrap; else if (i < 0) i += wrap; }
        const long base = nd == 0 ? 0 : nd == 3 ? i * s0 + long(r % size_t(n1)) * s1 : i * s0;
        T *d = dense + r * inner, *u = user + base;
        if (s_in == 1) { if (in) memcpy(d, u, inner * sizeof(T)); else memcpy(u, d, inner * sizeof(T)); }
        else if (in) for (size_t k = 0; k < inner; ++k) d[k] = u[long(k) * s_in];
        else for (size_t k = 0; k < inner; ++k) u[long(k) * s_in] = d[k];
      }
    };
    const int nt = host_threads();
    if (nt == 1 || n * sizeof(T) < (size_t(1) << 18)) { work(0, rows); return; }
    if (!hpool) hpool.reset(new WorkerPool(nt));
    hpool->run([&](int t) { work(rows * size_t(t) / size_t(nt), rows * size_t(t + 1) / size_t(nt)); });
  }
  // host arrays of sync_out: the copies into the staging area are queued by sync_out_arr, the rows go to the caller's arrays once the
  // stream has been waited for (finish_sync_out)
  struct OutJob { T *stg, *user; size_t n; int n1, n2; long s0, s1, s2, ioff; hipEvent_t done; };
  std::vector<OutJob> out_jobs;
  std::vector<hipEvent_t> out_events;
  // (a field's rows are written while the next field's copy is still in flight: an event per field)
  void finish_sync_out()
  {
    for (const OutJob &j : out_jobs) {
      HIPCHK(hipEventSynchronize(j.done));
      host_copy_rows(false, j.stg, j.user, j.n, j.n1, j.n2, j.s0, j.s1, j.s2, j.ioff, 0);
    }
    out_jobs.clear();
  }
  // ---- the Courant numbers of a combined step_sync, when they are host arrays: nothing reads them before step_async, so their rows
  // are gathered and copied while the condensation kernels run, see step_sync
  const lcx_arrinfo_t *late_c[3] = {nullptr, nullptr, nullptr}; bool courants_late = false;
  void late_courants()
  {
    if (!courants_late) return;
    courants_late = false;
    sync_in_arr(late_c[0], courant_x, n_cx, 1, 0, 0, halo); sync_in_arr(late_c[1], courant_y, n_cy, 0, 1, 0, halo);
    sync_in_arr(late_c[2], courant_z, n_cz, 0, 0, 1, halo);
    // a Courant array of the same call that lives on the DEVICE has only been listed (add_job): step_cond's flush_sync_jobs ran before
    // this -- launch it here, or step_async advects with the previous step's numbers for that direction
    flush_sync_jobs();
    flush_host_in();                                     // (whatever is queued on the object's stream from here on sees them)
  }
  static bool on_host(const lcx_arrinfo_t *a) { return !is_null(a) && !a->on_device; }
  void step_sync(const lcx_opts_t &opts, const lcx_arrinfo_t *th_, const lcx_arrinfo_t *rv_, const lcx_arrinfo_t *rhod_, const lcx_arrinfo_t *cx,
                 const lcx_arrinfo_t *cy, const lcx_arrinfo_t *cz, const lcx_arrinfo_t *diss) override
  {
    // (pred_corr looks at the Courant numbers inside sync_in: no deferral then)
    courants_late = (on_host(cx) || on_host(cy) || on_host(cz)) && !dbg(LCX_DBG_HOST_SYNC_LOOP) && o.adve_scheme != LCX_ADVE_PRED_CORR && opts.cond;
    late_c[0] = cx; late_c[1] = cy; late_c[2] = cz;
    // (a failed call leaves no job behind that holds a caller pointer: the next call would read or write through it)
    try { sync_in(th_, rv_, rhod_, cx, cy, cz, diss); step_cond(opts, th_, rv_); }
    catch (...) { courants_late = false; out_jobs.clear(); host_in_jobs.clear(); jobs_in.n_jobs = jobs_out.n_jobs = 0; throw; }
  }
  void sync_out_arr(DevBuf<T> &from, const lcx_arrinfo_t *a, size_t n)
  {
    if (is_null(a)) return;
    int n1, n2; long s0, s1, s2;
    arr_geom(a, 0, 0, 0, n1, n2, s0, s1, s2);
    if (a->on_device) {
      if (jobs_out.n_jobs == MAX_SYNC_JOBS) flush_sync_jobs();
      add_job(jobs_out, from.p, (const T *)a->data, n, n1, n2, s0, s1, s2, a->on_device == 3 ? 0l : long(o.n_x_bfr), 0);
      return;
    }
    if (!dbg(LCX_DBG_HOST_SYNC_LOOP)) {
      T *stg = stage_reserve(hstage_out, n);
      HIPCHK(hipMemcpyAsync(stg, from.p, n * sizeof(T), hipMemcpyDeviceToHost, st));
      hstage_busy = true;
      const size_t k = out_jobs.size();
      while (out_events.size() <= k) { hipEvent_t e; HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); out_events.push_back(e); }
      HIPCHK(hipEventRecord(out_events[k], st));
      out_jobs.push_back(OutJob{stg, (T *)a->data, n, n1, n2, s0, s1, s2, long(o.n_x_bfr), out_events[k]});
      return;
    }
    stage_host.resize(n);
    HIPCHK(hipMemcpyAsync(stage_host.data(), from.p, n * sizeof(T), hipMemcpyDeviceToHost, st));
    sync();
    T *d = (T *)a->data;
    const long ioff = o.n_x_bfr;
    switch (n_dims) {
      case 0: d[0] = stage_host[0]; break;
      case 1: for (size_t c = 0; c < n; ++c) d[(long(c) + ioff) * s0] = stage_host[c]; break;
      case 2: for (size_t c = 0; c < n; ++c) d[(long(c / n2) + ioff) * s0 + long(c % n2) * s1] = stage_host[c]; break;
      default: for (size_t c = 0; c < n; ++c) d[(long(c / (size_t(n2) * n1)) + ioff) * s0 + long((c / n2) % n1) * s1 + long(c % n2) * s2] = stage_host[c];
    }
  }

  // ------------------------------------------------------------------------------------------
  // random numbers (src/detail/urand.hpp): Philox on the fly, or a replayed host stream
  // ------------------------------------------------------------------------------------------
  u01_src<T> rand_u01(size_t n)
  {
    if (!replay.empty()) {
      Replay r = std::move(replay.front()); replay.pop_front();
      if (r.kind != 0 || r.n < n) throw lcx_error("libcloudph++: rng replay queue does not match the requested rand_u01 call");
      const T *ptr = r.u01->p;
      replay_keep_T.push_back(std::move(r.u01));
      return u01_src<T>{ptr, 0, 0};
    }
    return u01_src<T>{nullptr, ++rng_call, uint64_t(uint32_t(seed_now()))};
  }
  const bool shuffle_philox = dbg(LCX_DBG_SHUFFLE_PHILOX);      // measurement switch: round 2's shuffle keys (Philox, 64-bit ranking)
  rng_src rand_un(size_t n)
  {
    if (!replay.empty()) {
      Replay r = std::move(replay.front()); replay.pop_front();
      if (r.kind != 1 || r.n < n) throw lcx_error("libcloudph++: rng replay queue does not match the requested rand_un call");
      const uint32_t *ptr = r.un->p;
      replay_keep_u.push_back(std::move(r.un));
      return rng_src{ptr, 0, 0, 0u, 0u};
    }
    // the device generator's shuffle keys: a salted bijection of the ids (lcx_kernels.hpp, rng_src); the two salt words of this call
    rng_src rs{nullptr, ++rng_call, uint64_t(uint32_t(seed_now())), 0u, 0u};
    uint32_t w[4];
    philox::gen(0x756e73616c74ull /* "unsalt" */, rs.call, rs.seed, w);
    rs.s1 = w[0]; rs.s2 = w[1] | 1u;                 // (never both zero: that selects the switch below)
    if (shuffle_philox) rs.s1 = rs.s2 = 0u;
    return rs;
  }
  void release_replay_keep() { if (replay_keep_T.empty() && replay_keep_u.empty()) return; sync(); replay_keep_T.clear(); replay_keep_u.clear(); }
  void rng_replay_push(int kind, const double *data, size_t n) override
  {
    replay_used = true;
    Replay r; r.kind = kind; r.n = n;
    if (kind == 0 || kind == 2) {                        // 0: uniform [0,1), 2: standard normal
      std::vector<T> h(n); for (size_t i = 0; i < n; ++i) h[i] = T(data[i]);
      r.u01.reset(new DevBuf<T>()); r.u01->alloc(n);
      h2d(r.u01->p, h.data(), n * sizeof(T));
    } else if (kind == 1) {
      std::vector<uint32_t> h(n); for (size_t i = 0; i < n; ++i) h[i] = uint32_t(T(data[i]));   // fnctr_un goes through real_t
      r.un.reset(new DevBuf<uint32_t>()); r.un->alloc(n);
      h2d(r.un->p, h.data(), n * sizeof(uint32_t));
    } else throw lcx_error("libcloudph++: unknown rng replay kind");
    replay.push_back(std::move(r));
  }
  size_t rng_replay_pending() override { return replay.size(); }
  // LCX_DBG_TAG: what each coalescence call of the last step_async consumed (lcx_rng_dump, include/lcx.h)
  struct RngRec { DevBuf<T> u01, tag; DevBuf<uint32_t> un, ijk; size_t n_pos = 0, n_store = 0; };
  std::vector<std::unique_ptr<RngRec>> rng_recs;
  rng_src last_shuffle_rs{nullptr, 0, 0, 0u, 0u};      // the keys that put the cells into their present shuffled order
  void record_rng(const u01_src<T> &ru)
  {
    std::unique_ptr<RngRec> r(new RngRec);
    r->n_pos = npart; r->n_store = nphys;
    r->u01.alloc(npart); r->tag.alloc(nphys); r->un.alloc(nphys); r->ijk.alloc(nphys);
    hipLaunchKernelGGL(k_rng_record<T>, dim3(nblk(std::max(npart, nphys))), dim3(BS), 0, st, npart, nphys, last_shuffle_rs, ru, A.ext[ix_tag].p, ijk.p,
                       r->u01.p, r->un.p, r->tag.p, r->ijk.p);
    rng_recs.push_back(std::move(r));
  }
  void set_state_real(const char *name, const double *data, size_t n) override
  {
    if (std::string(name) != "tag" || ix_tag < 0) throw lcx_error("libcloudph++: lcx_set_state_real sets \"tag\" only (opts_init.dbg_flags & LCX_DBG_TAG)");
    if (n != nphys) throw lcx_error("libcloudph++: lcx_set_state_real: one value per storage slot (" + std::to_string(nphys) + ")");
    std::vector<T> h(data, data + n);
    h2d(A.ext[ix_tag].p, h.data(), n * sizeof(T));
  }
  void rng_dump(int call, int which, double *out, size_t capn, size_t *n) override
  {
    if (ix_tag < 0) throw lcx_error("libcloudph++: lcx_rng_dump needs opts_init.dbg_flags & LCX_DBG_TAG");
    if (call < 0 || size_t(call) >= rng_recs.size()) throw lcx_error("libcloudph++: lcx_rng_dump: the last step_async made no such coalescence call");
    if (which < 0 || which > 3) throw lcx_error("libcloudph++: lcx_rng_dump: which must be 0 ... 3");
    RngRec &r = *rng_recs[size_t(call)];
    const size_t len = which == 0 ? r.n_pos : r.n_store;
    *n = len;
    if (!out) return;
    if (capn < len) throw lcx_error("buffer too small");
    if (which == 0 || which == 2) { auto h = d2h(which == 0 ? r.u01.p : r.tag.p, len); for (size_t i = 0; i < len; ++i) out[i] = double(h[i]); }
    else { auto h = d2h(which == 1 ? r.un.p : r.ijk.p, len); for (size_t i = 0; i < len; ++i) out[i] = double(h[i]); }
  }

  // ------------------------------------------------------------------------------------------
  // housekeeping
  // ------------------------------------------------------------------------------------------
  bool vtpre_valid = false;       // vt_pre (the cell part of beard77) matches the current T, p, eta
  // with_vtpre: the cell pass of step_async -- hskpng_Tpr and the cell part of the beard77 terminal velocity in one launch
  void hskpng_Tpr(bool with_vtpre = false)
  {
    Range r(this, "hskpng_Tpr");
    if (with_vtpre && (vtc.formula == LCX_VT_BEARD77 || vtc.formula == LCX_VT_BEARD77FAST)) {
      vt_pre.alloc(ncell);
      hipLaunchKernelGGL(k_cell_Tpr_vtpre<T>, dim3(nblk(ncell)), dim3(BS), 0, st, ncell, th.p, rhod.p, rv.p, p.p, Tk.p, RH.p, eta.p, dv.p,
                         o.th_dry, o.const_p, o.RH_formula, n_dims, vt_pre.p);
      vtpre_valid = true;
      return;
    }
    hipLaunchKernelGGL(k_cell_Tpr<T>, dim3(nblk(ncell)), dim3(BS), 0, st, ncell, th.p, rhod.p, rv.p, p.p, Tk.p, RH.p, eta.p, dv.p,
                       o.th_dry, o.const_p, o.RH_formula, n_dims);
    vtpre_valid = false;
  }
  void hskpng_mfp()
  {
    hipLaunchKernelGGL(k_cell_mfp<T>, dim3(nblk(ncell)), dim3(BS), 0, st, ncell, Tk.p, p.p, lambda_D.p, lambda_K.p);
  }
  // cell index of every SD (+ optionally the cell histogram with per-SD ranks in the same pass)
  void ijk_and_hist(int do_ijk, bool do_hist)      // do_ijk: see k_ijk_hist
  {
    if (do_hist) HIPCHK(hipMemsetAsync(cell_cnt.p, 0, ncell * sizeof(uint32_t), st));
    if (nphys)
      hipLaunchKernelGGL(k_ijk_hist<T>, dim3(nblk(nphys)), dim3(BS), 0, st, size_t(0), nphys, g, A.n.p, A.x.p, A.y.p, A.z.p, ijk.p,
                         do_hist ? cell_cnt.p : nullptr, rnk(), do_ijk);
  }
  void hskpng_ijk() { Range r(this, "hskpng_ijk"); ijk_and_hist(2, false); sorted = false; sort_deferred = false; }   // (a sort left undone is void)
  // finish a sort given cell_cnt/rank: scan -> scatter -> per-cell order
  // meta_known: {number of cells above CELLRANK_MAX, largest occupancy} already on the host (listed from the histogram ahead of the
  // step's read-back), else order_cells lists them from the CSR offsets and pays a host round trip of its own
  uint32_t big_n = 0, big_mx = 0; uint64_t meta_version = ~0ull;
  // Which cells are listed for the one-wave-per-cell sorts.  Where cells are crowded on average (above CELLRANK_MAX / 2 per cell: C5's 512)
  // k_cellrank has nothing to rank -- round 4 ran it all the same, every workgroup voting "all crowded" and copying its ids to the
  // other buffer: 20 B per super-droplet moved for nothing, a third of C5's re-sort.  Round 5: in such a box k_cellrank is not launched,
  // one wave per cell orders EVERY cell's scattered ids in place without a list (sort_listed_cells), and only the cells beyond a wave's
  // capacity are listed (threshold CELLSORT_WAVE_MAX) for the workgroup-wide sorts.  big_thr: the threshold the list in big_list was
  // made with (a slab with neighbours keeps CELLRANK_MAX: its boundary ranges are ranked by k_cellrank)
  uint32_t big_thr = uint32_t(CELLRANK_MAX);
  uint32_t list_thr() const
  { return (!distmem() && ncell && npart / ncell > size_t(CELLRANK_MAX) / 2) ? uint32_t(CELLSORT_WAVE_MAX) : uint32_t(CELLRANK_MAX); }
  bool every_cell_by_a_wave() const { return big_thr == uint32_t(CELLSORT_WAVE_MAX); }
  // defer: the scan only -- the scatter and the in-cell ranking are left to the next step's condensation (finish_deferred_sort, or the
  // storage-order kernel that carries the scatter, cond_substep); the random keys of a shuffle are drawn NOW, at their place in the
  // generator's sequence
  bool sort_deferred = false, deferred_shuffle = false; rng_src deferred_rs{nullptr, 0, 0, 0u, 0u};
  const bool defer_sort_ok = !dbg(LCX_DBG_NO_DEFERRED_SORT);
  void sort_from_hist(bool shuffle, const uint32_t *meta_known = nullptr, bool defer = false)
  {
    sort_deferred = false;
    // (the scan leaves the histogram and the step's counters cleared for the next fused move)
    exclusive_scan(cell_cnt.p, cell_start.p, ncell, cell_start.p + ncell, cell_cnt.p, step_cnt.p, 3);
    ++cells_version;
    if (meta_known) { big_n = meta_known[0]; big_mx = meta_known[1]; meta_version = cells_version; }
    if (defer && nphys && npart) {
      deferred_shuffle = shuffle;
      deferred_rs = shuffle ? rand_un(npart) : rng_src{nullptr, 0, 0, 0u, 0u};
      sort_deferred = true; sorted = false;
      return;
    }
    if (nphys)
      hipLaunchKernelGGL(k_scatter_sorted, dim3(nblk(nphys)), dim3(BS), 0, st, nphys, ijk.p, rnk(), cell_start.p, sid(), sijk());
    order_cells(shuffle);
  }
  // scattered == false: nobody has carried the scatter
  void finish_deferred_sort(bool scattered = false)
  {
    if (!sort_deferred) return;
    sort_deferred = false;
    Range r(this, "post_copy");
    if (!scattered && nphys)
      hipLaunchKernelGGL(k_scatter_sorted, dim3(nblk(nphys)), dim3(BS), 0, st, nphys, ijk.p, rnk(), cell_start.p, sid(), sijk());
    order_cells(deferred_shuffle, &deferred_rs);
    shuffle_fresh = deferred_shuffle;
  }
  // puts every cell segment of sorted_id into the reference's order: ascending id, or ascending (un[id], id)
  void order_cells(bool shuffle, const rng_src *drawn = nullptr)
  {
    sort_deferred = false;
    if (npart) {
      rng_src rs{nullptr, 0, 0, 0u, 0u};
      if (drawn) rs = *drawn;
      else if (shuffle) rs = rand_un(npart);     // (a replayed stream is indexed by compact ids: coal() compacts first)
      if (shuffle) last_shuffle_rs = rs;
      if (ncell == 1 && !shuffle && nphys == npart) hipLaunchKernelGGL(k_iota, dim3(nblk(npart)), dim3(BS), 0, st, sid(), npart);
      else {
        // the list of cells too big for k_cellrank costs a host round trip unless it came with the step's read-back (sort_from_hist);
        // the in-cell shuffle of coalescence re-orders the SAME segments as the sort before it, so the list is kept until cell_start changes
        if (meta_version != cells_version) {
          big_thr = list_thr();
          HIPCHK(hipMemsetAsync(big_meta_own_p(), 0, 2 * sizeof(uint32_t), st));
          hipLaunchKernelGGL(k_list_big_cells, dim3(nblk(ncell)), dim3(BS), 0, st, ncell, cell_start.p, big_thr, big_list.p, big_meta_own_p(),
                             big_meta_own_p() + 1, (const uint32_t *)nullptr);
        }
        const int crowded = npart / (ncell ? ncell : 1) > size_t(CELLRANK_MAX) / 2;
        if (every_cell_by_a_wave()) ;           // every cell is sorted in place below: nothing to rank and no buffer swap
        else if (shuffle && !rs.un && !crowded && !shuffle_philox && (rs.s1 | rs.s2) && !dbg(LCX_DBG_RANK_BY_COUNTING) && vt_fix_pending && nphys) {
          // (between two coalescence substeps: hskpng_vterm_invalid rides on the ranking, see step_async)
          const bool b77 = vtc.formula == LCX_VT_BEARD77 || vtc.formula == LCX_VT_BEARD77FAST;
          const rank_vt_fix<T> fx{vtc, b77 ? (vtc.formula == LCX_VT_BEARD77FAST ? 2 : 1) : 0, o.strict_fp ? 0 : 1, Tk.p, p.p, rhod.p, eta.p, vt_0.p,
                                  b77 ? vt_pre.p : nullptr, A.rw2.p, A.vt.p};
          hipLaunchKernelGGL((k_cellrank_bkt<true, rank_vt_fix<T>>), dim3(nblk(npart)), dim3(BS), 0, st, npart, sijk(), cell_start.p, sid(), rnk() + sort_base, rs,
                             rank_range{nullptr, nullptr, nullptr}, fx);
          vt_fix_pending = false;
        }
        else if (shuffle && !rs.un && !crowded && !shuffle_philox && (rs.s1 | rs.s2) && !dbg(LCX_DBG_RANK_BY_COUNTING)) hipLaunchKernelGGL(k_cellrank_bkt<>, dim3(nblk(npart)), dim3(BS), 0, st, npart, sijk(), cell_start.p, sid(), rnk() + sort_base, rs, rank_range{nullptr, nullptr, nullptr});
        else if (shuffle && !rs.un && !crowded && !shuffle_philox) hipLaunchKernelGGL((k_cellrank<uint32_t, true>), dim3(nblk(npart)), dim3(BS), 0, st, npart, sijk(), cell_start.p, sid(), rnk() + sort_base, rs, crowded, rank_range{nullptr, nullptr, nullptr});
        else if (shuffle) hipLaunchKernelGGL(k_cellrank<uint64_t>, dim3(nblk(npart)), dim3(BS), 0, st, npart, sijk(), cell_start.p, sid(), rnk() + sort_base, rs, crowded, rank_range{nullptr, nullptr, nullptr});
        else hipLaunchKernelGGL(k_cellrank<uint32_t>, dim3(nblk(npart)), dim3(BS), 0, st, npart, sijk(), cell_start.p, sid(), rnk() + sort_base, rs, crowded, rank_range{nullptr, nullptr, nullptr});
        if (!every_cell_by_a_wave()) sorted_id.swap(rank);        // `rank` is free after the scatter: it serves as the output buffer
        if (meta_version != cells_version) {
          uint32_t m2[2];
          read_back(m2, big_meta_own_p(), 2);
          big_n = m2[0]; big_mx = m2[1]; meta_version = cells_version;
        }
        if (big_n || every_cell_by_a_wave()) sort_listed_cells(shuffle, rs);
      }
    }
    sorted = true; sorted_shuffled = shuffle; shuffle_fresh = false;
  }
  // the cells too crowded for k_cellrank (big_list, big_n of them, the largest holds big_mx): one wave, one workgroup or global scratch each
  void sort_listed_cells(bool shuffle, const rng_src &rs)
  {
    const uint32_t meta[2] = {big_n, big_mx};
    // every_cell_by_a_wave(): no list for the one-wave sorts -- every cell of the box in turn (the list then holds what is beyond a wave)
    const uint32_t *lst = every_cell_by_a_wave() ? (const uint32_t *)nullptr : big_list.p;
    const uint32_t n_w = every_cell_by_a_wave() ? uint32_t(ncell) : meta[0];
    const unsigned nbw = std::min<unsigned>((n_w + BS / WAVE - 1) / (BS / WAVE), 256u * 32u);
    // (the device generator's salted bijection keys: 32 bits order a cell)
    if (shuffle && !rs.un && (rs.s1 | rs.s2) && !dbg(LCX_DBG_RANK_BY_COUNTING))
    {
      hipLaunchKernelGGL(k_cellsort_wave_bkt<10>, dim3(nbw), dim3(BS), 0, st, lst, n_w, cell_start.p, sid(), rs, 1u);
      hipLaunchKernelGGL((k_cellsort_wave_bkt<CELLSORT_WAVE_MAX / WAVE>), dim3(nbw), dim3(BS), 0, st, lst, n_w, cell_start.p, sid(), rs, 10u * WAVE);
    }
    else if (shuffle && !rs.un && (rs.s1 | rs.s2)) hipLaunchKernelGGL((k_cellsort_wave<uint32_t, true>), dim3(nbw), dim3(BS), 0, st, lst, n_w, cell_start.p, sid(), rs);
    else if (shuffle) hipLaunchKernelGGL(k_cellsort_wave<uint64_t>, dim3(nbw), dim3(BS), 0, st, lst, n_w, cell_start.p, sid(), rs);
    else         hipLaunchKernelGGL(k_cellsort_wave<uint32_t>, dim3(nbw), dim3(BS), 0, st, lst, n_w, cell_start.p, sid(), rs);
    if (!meta[0]) return;
    if (meta[1] > uint32_t(CELLSORT_WAVE_MAX)) {
      const unsigned nbl = std::min<unsigned>(meta[0], 256u * 16u);
      if (shuffle) hipLaunchKernelGGL(k_cellsort_lds<uint64_t>, dim3(nbl), dim3(BS), 0, st, big_list.p, meta[0], cell_start.p, sid(), rs);
      else         hipLaunchKernelGGL(k_cellsort_lds<uint32_t>, dim3(nbl), dim3(BS), 0, st, big_list.p, meta[0], cell_start.p, sid(), rs);
    }
    if (meta[1] > uint32_t(CELLSORT_LDS_MAX)) {
      size_t P = 1; while (P < meta[1]) P <<= 1;
      const unsigned nb = std::min<unsigned>(meta[0], 64u);
      sort_scratch.alloc(P * nb);
      hipLaunchKernelGGL(k_cellsort_big, dim3(nb), dim3(1024), 0, st, big_list.p, meta[0], cell_start.p, sid(), int(shuffle), rs,
                         sort_scratch.p, P);
    }
  }
  void hskpng_sort_helper(bool shuffle)
  {
    Range r(this, shuffle ? "hskpng_shuffle_and_sort" : "hskpng_sort");
    finish_deferred_sort();
    if (sorted && shuffle && sorted_shuffled && shuffle_fresh && replay.empty()) { shuffle_fresh = false; return; }   // post_copy has shuffled for us already
    if (sorted && shuffle) { order_cells(true); return; }   // cells unchanged since the last sort: re-order the segments only
    ijk_and_hist(0, true);
    sort_from_hist(shuffle);
  }
  void hskpng_sort() { finish_deferred_sort(); if (!sorted) hskpng_sort_helper(false); }
  void hskpng_count() { hskpng_sort(); }
  void hskpng_vterm(bool only_invalid)
  {
    Range r(this, only_invalid ? "hskpng_vterm_invalid" : "hskpng_vterm_all");
    if (!nphys) return;
    if (vtc.formula == LCX_VT_BEARD77 || vtc.formula == LCX_VT_BEARD77FAST) {
      if (!vtpre_valid) {
        vt_pre.alloc(ncell);
        hipLaunchKernelGGL(k_vterm_cellpre<T>, dim3(nblk(ncell)), dim3(BS), 0, st, ncell, p.p, rhod.p, eta.p, vt_pre.p);
        vtpre_valid = true;
      }
      const dim3 gv(nblk(nphys, 2 * BS * VT_CHUNKS));                   // VT_CHUNKS pairs of super-droplets per lane
      auto launch = [&](auto kern) { hipLaunchKernelGGL(kern, gv, dim3(BS), 0, st, nphys, int(only_invalid), vtc, A.rw2.p, ijk.p, vt_pre.p, vt_0.p, A.vt.p); };
      const bool table = vtc.formula == LCX_VT_BEARD77FAST;
      if (o.strict_fp) { if (table) launch(k_vterm_b77<T, false, true>); else launch(k_vterm_b77<T, false, false>); }
      else             { if (table) launch(k_vterm_b77<T, true, true>); else launch(k_vterm_b77<T, true, false>); }
      return;
    }
    hipLaunchKernelGGL(k_vterm<T>, dim3(nblk(nphys)), dim3(BS), 0, st, nphys, int(only_invalid), vtc, A.rw2.p, ijk.p, Tk.p, p.p, rhod.p, eta.p, vt_0.p, A.vt.p);
  }
  // cells per workgroup of the LDS-staged per-cell walks: as many as fit the staging buffer at the mean occupancy (+25 %)
  int cf_cells() const
  {
    const size_t mean = npart / std::max<size_t>(ncell, 1) + 1;
    return int(std::max<size_t>(1, std::min<size_t>(CF_CELLS, size_t(CF_CAP) * 4 / 5 / mean)));
  }
  void check_npart(size_t n) const
  {                                                                                      // hskpng_resize.ipp:9
    if (n > o.n_sd_max) throw lcx_error("n_sd_max (" + std::to_string(o.n_sd_max) + ") < n_part (" + std::to_string(n) + ")");
  }

  // post_copy.ipp:18-35: remove n==0 (stable) -> ijk -> count(sort).
  // The removal is LAZY: a dead SD (n == 0) is excluded from the cell histogram at once (so no kernel that walks the
  // sorted order ever sees it and n_part() reports living SDs only), but the stable compaction of the storage --
  // 2 x 64 B per SD of traffic for typically a handful of deaths per step -- is deferred until dead SDs exceed 1/32 of
  // the storage, or until something observes storage order (get_attr / state getters, a replayed random stream,
  // set_particles, capacity pressure).  Relative order of the living SDs, hence every tie-break of the stable sort, is
  // the same as after the reference's eager remove_if.  LCX_EAGER_COMPACT=1 forces a compaction every step.
  // ordered list of the ids whose flag byte is 1 (the scan-compaction of the migrant lists), returns the count
  size_t ordered_ids(const uint8_t *flag, DevBuf<uint32_t> &out)
  {
    const size_t tiles = (nphys + SCAN_TILE - 1) / SCAN_TILE;
    out.alloc(cap);
    hipLaunchKernelGGL(k_mig_tiles, dim3(unsigned(tiles)), dim3(BS), 0, st, flag, nphys, uint8_t(1), tile_sums.p);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, st, tile_sums.p, tiles, scan_total.p);
    hipLaunchKernelGGL(k_mig_ids, dim3(unsigned(tiles)), dim3(BS), 0, st, flag, nphys, uint8_t(1), tile_sums.p, out.p);
    uint32_t tot = 0;
    read_back(&tot, scan_total.p, 1);
    return tot;
  }
  // housekeeping/particles_impl_rcyc.ipp:44-140.  The reference sorts all multiplicities (stable) and pairs the t-th SD of
  // the sorted sequence (the zeros, ids ascending) with the t-th from its end (largest n first, higher id first among equal
  // n).  Here: counts -> radix select of the k-th largest multiplicity -> ordered id lists of the zeros, of n > n* and of
  // n == n* -> the same pairs.  Storage must hold no stale dead SDs (post_copy compacts eagerly while rcyc is on).
  void rcyc()
  {
    if (!nphys) return;
    const size_t N = nphys;
    rc_stat.alloc_zero(4, st); rc_max.alloc_zero(1, st);
    hipLaunchKernelGGL(k_rcyc_stat, dim3(nblk(N)), dim3(BS), 0, st, A.n.p, N, rc_stat.p, rc_max.p);
    unsigned int stat[4]; unsigned long long nmax = 0;
    read_back(stat, rc_stat.p, 4); read_back(&nmax, rc_max.p, 1);
    size_t n_flagged = stat[0];
    if (n_flagged == 0 || pure_const_multi) return;                      // (pure_const_multi: removal only)
    const size_t n_splittable = stat[1] > 0 ? size_t(stat[2]) : N;      // entries behind the last n == 1 of the sorted sequence
    if (n_splittable == 0) return;
    const size_t k = std::min(n_flagged, n_splittable);
    // radix select: n* = k-th largest multiplicity, rem = how many donors carry exactly n*
    int bytes = 1; while (bytes < 8 && (nmax >> (8 * bytes))) ++bytes;
    n_t prefix = 0; size_t rem = k;
    rc_hist.alloc(256);
    for (int b = bytes - 1; b >= 0; --b) {
      HIPCHK(hipMemsetAsync(rc_hist.p, 0, 256 * sizeof(unsigned int), st));
      hipLaunchKernelGGL(k_rcyc_hist, dim3(nblk(N)), dim3(BS), 0, st, A.n.p, N, prefix, 8 * b, rc_hist.p);
      std::vector<unsigned int> hst = d2h(rc_hist.p, 256);
      size_t above = 0; int bin = 255;
      for (; bin > 0; --bin) { if (above + hst[bin] >= rem) break; above += hst[bin]; }
      rem -= above;
      prefix |= n_t(bin) << (8 * b);
    }
    const n_t thr = prefix;
    const size_t m = k - rem;                                            // donors with n > n*
    if (!mig.p) mig.alloc(cap);
    std::vector<uint32_t> recv(k), donor(k);
    hipLaunchKernelGGL(k_rcyc_flag, dim3(nblk(N)), dim3(BS), 0, st, A.n.p, N, 0, n_t(0), mig.p);
    ordered_ids(mig.p, mig_ids[0]);
    HIPCHK(hipMemcpyAsync(recv.data(), mig_ids[0].p, k * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    sync();
    if (m) {
      hipLaunchKernelGGL(k_rcyc_flag, dim3(nblk(N)), dim3(BS), 0, st, A.n.p, N, 2, thr, mig.p);
      const size_t got = ordered_ids(mig.p, mig_ids[0]);
      if (got != m) throw lcx_error("libcloudph++ (HIP): rcyc selection inconsistent");
      rc_vals.alloc(m);
      hipLaunchKernelGGL(k_gather_n, dim3(nblk(m)), dim3(BS), 0, st, mig_ids[0].p, m, A.n.p, rc_vals.p);
      std::vector<uint32_t> ids = d2h(mig_ids[0].p, m);
      std::vector<n_t> vals = d2h(rc_vals.p, m);
      std::vector<size_t> ord(m);
      for (size_t i = 0; i < m; ++i) ord[i] = i;
      std::sort(ord.begin(), ord.end(), [&](size_t a_, size_t b_) { return vals[a_] != vals[b_] ? vals[a_] > vals[b_] : ids[a_] > ids[b_]; });
      for (size_t i = 0; i < m; ++i) donor[i] = ids[ord[i]];
    }
    if (rem) {
      hipLaunchKernelGGL(k_rcyc_flag, dim3(nblk(N)), dim3(BS), 0, st, A.n.p, N, 1, thr, mig.p);
      const size_t e = ordered_ids(mig.p, mig_ids[1]);
      if (e < rem) throw lcx_error("libcloudph++ (HIP): rcyc selection inconsistent");
      std::vector<uint32_t> tail(rem);
      HIPCHK(hipMemcpyAsync(tail.data(), mig_ids[1].p + (e - rem), rem * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
      sync();
      for (size_t i = 0; i < rem; ++i) donor[m + i] = tail[rem - 1 - i];       // highest id first
    }
    rc_pairs.alloc(2 * k);
    HIPCHK(hipMemcpyAsync(rc_pairs.p, recv.data(), k * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(rc_pairs.p + k, donor.data(), k * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_rcyc_apply<T>, dim3(nblk(k)), dim3(BS), 0, st, k, rc_pairs.p, rc_pairs.p + k, aset(A), g);
    sync();
  }
  DevBuf<unsigned int> rc_stat, rc_hist; DevBuf<unsigned long long> rc_max; DevBuf<n_t> rc_vals; DevBuf<uint32_t> rc_pairs;
  void post_copy(const lcx_opts_t &opts, bool force_compact = false)
  {
    if (opts.rcyc) { Range r(this, "rcyc"); rcyc(); force_compact = true; }      // what could not be recycled is removed at once
    Range r(this, "post_copy");
    const size_t tiles = (nphys + SCAN_TILE - 1) / SCAN_TILE;
    uint32_t alive = 0;
    if (tiles) {
      hipLaunchKernelGGL(k_alive_tiles, dim3(unsigned(tiles)), dim3(BS), 0, st, A.n.p, nphys, tile_sums.p);
      hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, st, tile_sums.p, tiles, scan_total.p);
      read_back(&alive, scan_total.p, 1);
    }
    const size_t dead = nphys - alive;
    if (dead && (force_compact || eager_compact || dead * 32 > nphys)) {
      if (!B.n.p) alloc_attrs(B);
      HIPCHK(hipMemsetAsync(cell_cnt.p, 0, ncell * sizeof(uint32_t), st));
      hipLaunchKernelGGL(k_compact<T>, dim3(unsigned(tiles)), dim3(BS), 0, st, nphys, aset(A), aset(B), tile_sums.p, g, ijk.p, cell_cnt.p, rnk());
      swap_attr_sets();
      nphys = alive;
    } else ijk_and_hist(1, true);                    // re-index in place (dead SDs get DEAD_CELL)
    npart = alive;
    sort_from_hist(false);
  }
  // post_copy when k_move has already produced ijk / histogram / ranks / the dead count
  static constexpr int SLAB_REORDER_EVERY = 16;      // (see post_copy_after_fused_move)
  bool listed_from_hist = false, meta_known_valid = false; uint32_t meta_known_v[2] = {0, 0};
  void list_big_from_hist()
  {   // (big_meta was cleared behind the previous sort's scan)
    big_thr = list_thr();
    hipLaunchKernelGGL(k_list_big_cells, dim3(nblk(ncell)), dim3(BS), 0, st, ncell, (const uint32_t *)nullptr, big_thr, big_list.p, big_meta_p(), big_meta_p() + 1,
                       (const uint32_t *)cell_cnt.p);
    listed_from_hist = true;
  }
  void post_copy_after_fused_move(const lcx_opts_t &opts, long dead_known = -1)
  {
    unsigned int dead = 0;
    uint32_t meta[2] = {0, 0}; const uint32_t *meta_p = nullptr;
    if (dead_known >= 0) { dead = unsigned(dead_known); if (meta_known_valid) { meta[0] = meta_known_v[0]; meta[1] = meta_known_v[1]; meta_p = meta; } }
    else {
      uint32_t h[3];
      read_back(h, step_cnt.p, 3);                                // dead count (+ the crowded cells, if the move listed them)
      dead = h[0];
      if (listed_from_hist) { meta[0] = h[1]; meta[1] = h[2]; meta_p = meta; }
      dead -= unsigned(std::min<size_t>(reused_total, dead));     // emigrants' slots that immigrants have taken over are alive again
    }
    listed_from_hist = meta_known_valid = false;
    reused_total = 0;
    // reference storage order (stable compaction only) when asked for, and in every parity run (a replayed stream is indexed by id)
    const bool strict_order = this->strict_order();
    const bool compact_now = dead && (eager_compact || size_t(dead) * 32 > nphys);
    if (compact_now && strict_order) { post_copy(opts, true); return; }
    Range r(this, "post_copy");
    npart = nphys - dead;
    // default period: 64 steps; 16 for a slab with neighbours, whose storage the immigrants and emigrants disorder faster -- a third of a
    // 16-plane slab's droplets are replaced within 32 steps at |C| = 0.15, and the newcomers sit wherever a slot was free: coalescence
    // 130 -> 233 us, the move 213 -> 317, condensation 339 -> 413 between two re-orderings of 0.85 ms (profiles/r04x_*; ms per step of the
    // slab with its exchange, period 8: 1.400, 12: 1.388, 16: 1.367, 24: 1.385, 32: 1.408; a single device: 64 and 32 alike)
    // (crowded cells, 512 per cell on C5: the pairs of coalescence are gathered from anywhere in a cell's 4 KB per attribute, and from
    // further away the longer the storage has drifted from the cell order -- 66 steps of C5: period 64 79.8 ms per step, 32: 78.0, 16: 77.7)
    const int every_ = o.reorder_every > 0 ? o.reorder_every : (distmem() ? SLAB_REORDER_EVERY : (ncell && npart / ncell >= 256 ? 16 : 64));
    const bool reorder_due = compact_now || (!strict_order && steps_since_reorder + 1 >= every_);     // (the re-ordering wants the plain order)
    // (strict arithmetic sums a cell's droplets in the reference's order, ascending id: k_cond_cellfinish<T, 1> walks the sorted order
    // as it finds it, so the cells must not be left in the shuffled order there -- the in-cell ranking by id stays, coalescence
    // shuffles for itself)
    const bool preshuffle = !strict_order && !o.strict_fp && last_async_coal && o.coal_switch && !reorder_due && npart >= 2;
    // (the scatter rides on the next condensation kernel when that will be the storage-order one and no re-ordering is due -- round 4: a
    // slab with neighbours as well, once its immigrants are in the histogram; its exchange then skips the overlapped interior re-sort)
    const bool defer = defer_sort_ok && lean_storage_cond() && !reorder_due && meta_p != nullptr && replay.empty();
    sort_from_hist(preshuffle, meta_p, defer);
    shuffle_fresh = preshuffle && !sort_deferred;
    // dropping the dead SDs costs one pass over all attributes either way: gather it in sorted order (opts_init.reorder_every)
    if (compact_now || (!strict_order && ++steps_since_reorder >= every_)) reorder_storage();
  }
  // opts_init.reorder_every: storage := cell-sorted order (ids renumbered, dead SDs dropped); needs the plain sorted order
  void reorder_storage()
  {
    steps_since_reorder = 0;
    if (!npart) return;
    Range r(this, "reorder_storage");
    if (!B.n.p) alloc_attrs(B);
    hipLaunchKernelGGL(k_reorder<T>, dim3(nblk(npart)), dim3(BS), 0, st, npart, sid(), sijk(), aset(A), aset(B), g, rnk());
    swap_attr_sets();
    ijk.swap(rank);
    hipLaunchKernelGGL(k_iota, dim3(nblk(npart)), dim3(BS), 0, st, sid(), npart);
    nphys = npart;
  }
  void swap_attr_sets()
  {
    A.n.swap(B.n); A.rd3.swap(B.rd3); A.rw2.swap(B.rw2); A.kpa.swap(B.kpa); A.vt.swap(B.vt); A.x.swap(B.x); A.y.swap(B.y); A.z.swap(B.z);
    for (int e = 0; e < n_ext; ++e) A.ext[e].swap(B.ext[e]);
  }
  // make storage order == the reference's (no dead SDs in it) before anything that exposes storage order
  void ensure_compact()
  {
    if (lft_count || rgt_count || fused_pending)
      throw lcx_error("libcloudph++: the neighbour exchange of this step is pending (migrate_pack / _unpack / _finish) -- particle state cannot be read or compacted now");
    if (nphys == npart) return;
    ensure_compact_forced();
  }
  void ensure_compact_forced() { lcx_opts_t od; lcx_opts_default(&od); post_copy(od, true); }

  // ------------------------------------------------------------------------------------------
  // condensation (particles_step.ipp:188-267)
  // ------------------------------------------------------------------------------------------
  void sstp_save()
  {
    if (!allow_sstp_cond) return;
    if (o.exact_sstp_cond) {                                  // per-particle version (sstp_save.ipp:17-22); ijk is valid here
      if (nphys)
        hipLaunchKernelGGL(k_pp_save<T>, dim3(nblk(nphys)), dim3(BS), 0, st, nphys, ijk.p, rv.p, th.p, rhod.p, p.p,
                           A.ext[ix_rv].p, A.ext[ix_th].p, A.ext[ix_rh].p, o.const_p ? A.ext[ix_p].p : nullptr);
      return;
    }
    HIPCHK(hipMemcpyAsync(sstp_tmp_rv.p, rv.p, ncell * sizeof(T), hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(sstp_tmp_th.p, th.p, ncell * sizeof(T), hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(sstp_tmp_rh.p, rhod.p, ncell * sizeof(T), hipMemcpyDeviceToDevice, st));
  }
  // round 6: every condensation substep of the step in ONE launch (k_cond_substeps: a workgroup owns a run of cells and their droplets);
  // fast arithmetic, sstp_cond > 1, the sorted order in place
  void cond_substeps_fused(double RH_max)
  {
    Range r(this, "cond");
    cond_pre.alloc(ncell * sizeof(cond_cell_fast<T>));
    const cell_pre_args<T> P{ncell, th.p, rhod.p, rv.p, p.p, Tk.p, RH.p, eta.p, dv.p, lambda_D.p, lambda_K.p, o.th_dry, o.const_p, o.RH_formula, n_dims,
                             T(RH_max), reinterpret_cast<cond_cell_fast<T> *>(cond_pre.p)};
    cond_args<T> a{sid(), sijk(), A.n.p, A.rd3.p, A.kpa.p, A.vt.p, A.rw2.p, rhod.p, rv.p, Tk.p, eta.p, RH.p,
                   lambda_D.p, lambda_K.p, m3_before.p, m3_after.p, T(T(dt) / sstp_cond), T(RH_max), eps_tol, T(2.), 100u, 1, ncell,
                   0u, nullptr, P.pre, nullptr, nullptr, nullptr, nullptr, nullptr, 0u};
    const unsigned per_cell = unsigned(std::max<size_t>(1, npart / std::max<size_t>(1, ncell)));
    const unsigned cells_per_wg = std::max(1u, std::min(unsigned(SUBSTEP_CELLS), unsigned(BS) / per_cell));
    const dim3 grid(nblk(ncell, cells_per_wg)), bl(BS);
    const bool toms = o.cond_solver == 1;
    const int ask = dbg(LCX_DBG_COND_NO_LIST) ? 0 : 1;
    const T kv = kpa_uniform ? kpa_value : T(0);
    last_cond_kernel = LCX_CK_SUBSTEPS;
    if (toms && kpa_uniform) hipLaunchKernelGGL((k_cond_substeps<T, true, 2>), grid, bl, 0, st, P, a, kv, cell_start.p, cells_per_wg, sstp_cond, sstp_fused(0), rw_mom3.p, ask);
    else if (toms) hipLaunchKernelGGL((k_cond_substeps<T, false, 2>), grid, bl, 0, st, P, a, kv, cell_start.p, cells_per_wg, sstp_cond, sstp_fused(0), rw_mom3.p, ask);
    else if (kpa_uniform) hipLaunchKernelGGL((k_cond_substeps<T, true, 0>), grid, bl, 0, st, P, a, kv, cell_start.p, cells_per_wg, sstp_cond, sstp_fused(0), rw_mom3.p, ask);
    else hipLaunchKernelGGL((k_cond_substeps<T, false, 0>), grid, bl, 0, st, P, a, kv, cell_start.p, cells_per_wg, sstp_cond, sstp_fused(0), rw_mom3.p, ask);
    vtpre_valid = false;
  }
  // sstp_percell_step.ipp:7-48, as an argument of the substep's cell pass (k_cell_cond_pre)
  sstp_fields<T> sstp_fused(int step)
  {
    sstp_fields<T> ss{0, step, T(sstp_cond), {rv.p, th.p, rhod.p}, {sstp_tmp_rv.p, sstp_tmp_th.p, sstp_tmp_rh.p}};
    if (sstp_cond > 1) ss.n = var_rho ? 3 : 2;
    return ss;
  }
  bool lean_storage_cond() const
  { return !o.strict_fp && !no_cond_pre && cond_storage_order && !cond_toms_two_pass() && !o.exact_sstp_cond; }
  bool cond_toms_two_pass() const { return o.cond_solver == 1 && dbg(LCX_DBG_COND_TOMS_TWO_PASS); }
  void cond_substep(double RH_max, int step, bool turb_cond = false)
  {
    const bool carry_scatter = sort_deferred && lean_storage_cond() && !turb_cond && npart;
    if (!carry_scatter) hskpng_sort();
    // fast arithmetic (no SGS supersaturation): per-cell set-up hoisted (k_cond_cellpre) and one scratch value per droplet (the
    // change of n rw^3); strict arithmetic: n rw^3 before and after in position order + the ordered per-cell walk
    const bool fast = !o.strict_fp && !turb_cond && !no_cond_pre;
    {
      // the substep's cell pass in one launch: mean free paths (substep 0, from the previous T and p as the reference's hskpng_mfp
      // ahead of the loop), hskpng_Tpr, and in fast arithmetic the droplet-independent set-up of the growth rate
      Range r(this, "hskpng_Tpr");
      // (the last 16 words: the counter of k_cond_lean's list of droplets for the reference's iterates, see cond_list -- cleared with the rest)
      // (two sets of DEFER_SHARDS counters, 64 B apart: the list of droplets for the reference's iterates and the records of the first pass's budget, see cond_list -- cleared here)
      if (fast) { cond_pre.alloc(ncell * sizeof(cond_cell_fast<T>)); defer_cnt.alloc(2 * DEFER_SHARDS * DEFER_CNT_STRIDE); }
      const int n_defer_words = 2 * DEFER_SHARDS * DEFER_CNT_STRIDE;
      hipLaunchKernelGGL(k_cell_cond_pre<T>, dim3(std::max(nblk(ncell), nblk(size_t(n_defer_words)))), dim3(BS), 0, st, ncell, th.p, rhod.p, rv.p, p.p,
                         Tk.p, RH.p, eta.p, dv.p, lambda_D.p, lambda_K.p, o.th_dry, o.const_p, o.RH_formula, n_dims, int(step == 0), T(RH_max),
                         fast ? reinterpret_cast<cond_cell_fast<T> *>(cond_pre.p) : (cond_cell_fast<T> *)nullptr,
                         fast ? defer_cnt.p : (uint32_t *)nullptr, fast ? n_defer_words : 0, sstp_fused(step));
      vtpre_valid = false;
    }
    if (npart) {
      Range r(this, "cond");
      cond_args<T> a{sid(), sijk(), A.n.p, A.rd3.p, A.kpa.p, A.vt.p, A.rw2.p, rhod.p, rv.p, Tk.p, eta.p, RH.p,
                     lambda_D.p, lambda_K.p, m3_before.p, m3_after.p, T(T(dt) / sstp_cond), T(RH_max), eps_tol, T(2.), 100u, step == 0, ncell,
                     xcd_group(npart, ncell),
                     turb_cond ? A.ext[ix_ssp].p : nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                     unsigned(o.dbg_cond_budget > 0 ? std::min(o.dbg_cond_budget, FOLD_CAP) : FOLD_CAP)};
      const dim3 gr(nblk(npart)), bl(BS);
      // fast arithmetic: k_cond_lean with the lean bracketed secant, or (opts_init.cond_solver = 1) with TOMS748 -- the reference's
      // iterates in fast arithmetic; round 2's kernels for that (iteration budget + straggler launch, fold) behind a switch
      const bool cond_toms = o.cond_solver == 1;
      if (fast && !cond_toms_two_pass()) {
        a.pre = reinterpret_cast<const cond_cell_fast<T> *>(cond_pre.p);
        // the lean solver's list (cond_list): room for every droplet, 8 B each
        cond_list lst{nullptr, nullptr};
        bool listed = false;
        launch_listed = nullptr;               // (a call that threw between the two launches must not leave its second half behind)
        // round 6: in DEFER_SHARDS parts (one counter each), and the first pass's loop with a budget of trips -- a droplet that needs more
        // is listed as well (k_cond_lean).  list_parts(): the part's capacity for a launch of `blocks` workgroups of `per_block` droplets
        auto list_parts = [&](size_t blocks, size_t per_block) {
          const size_t shard_cap = nblk(blocks, DEFER_SHARDS) * per_block;
          cond_listed.alloc(2 * size_t(DEFER_SHARDS) * shard_cap);
          // (the budget: measured and not adopted -- the first pass 3.0 -> 2.75 ms, its second pass 0.26 ms beside the ranking: the step as before)
          const unsigned budget = !dbg(LCX_DBG_COND_BUDGET) || dbg(LCX_DBG_COND_WQ) ? 100u : o.dbg_cond_budget > 0 && (o.dbg_cond_budget & 255) ? unsigned(o.dbg_cond_budget & 255) : 2u;
          cond_list l{cond_listed.p, defer_cnt.p, shard_cap, budget, nullptr, nullptr, 0u};
          if (budget < 100u) {
            // records for 3 % of the droplets (bench.py's settled boxes leave 1 %; dbg_cond_budget >> 8: a test's own capacity per part)
            l.rec_cap = o.dbg_cond_budget > 255 ? uint32_t(o.dbg_cond_budget >> 8) : uint32_t(std::max<size_t>(256, shard_cap * 3 / 100));
            cond_records.alloc(size_t(DEFER_SHARDS) * l.rec_cap * sizeof(lean_record<T>));
            l.rec = cond_records.p; l.rcount = defer_cnt.p + DEFER_SHARDS * DEFER_CNT_STRIDE;
          }
          return l;
        };
        const bool want_list = !cond_toms && !dbg(LCX_DBG_COND_NO_LIST);
        cond_in_storage_order = cond_storage_order;
        if (cond_in_storage_order) {
          a.storage_ijk = ijk.p; a.xcd_group = xcd_group(nphys, ncell);
          if (carry_scatter) { a.sc_rank = rnk(); a.sc_cell_start = cell_start.p; a.sc_sorted_id = sid(); a.sc_sorted_ijk = sijk(); }
          // (one hygroscopicity in the whole run: a scalar instead of 8 B per droplet, see kpa_uniform)
          const dim3 gs(nblk(nphys));
          // (cond_solver = 1: the workgroup folded behind TOMS748's head -- this kernel is bound by instruction issue at half-empty
          // waves, unlike the lean solver's; LCX_DBG_COND_NO_FOLD: the plain kernel, the same bits)
          last_cond_kernel = cond_toms ? (dbg(LCX_DBG_COND_NO_FOLD) ? LCX_CK_LEAN_TOMS748 : LCX_CK_FOLD_TOMS748) : dbg(LCX_DBG_COND_LEAN_R3) ? LCX_CK_LEAN_R3 : dbg(LCX_DBG_COND_FOLD) ? LCX_CK_FOLD_LEAN
                           : dbg(LCX_DBG_COND_WQ) ? LCX_CK_LEAN_WQ : LCX_CK_LEAN;
          if (cond_toms && !dbg(LCX_DBG_COND_NO_FOLD) && kpa_uniform) hipLaunchKernelGGL((k_cond_lean_fold<T, true, 2>), gs, bl, 0, st, nphys, a, kpa_value);
          else if (cond_toms && !dbg(LCX_DBG_COND_NO_FOLD)) hipLaunchKernelGGL((k_cond_lean_fold<T, false, 2>), gs, bl, 0, st, nphys, a, T(0));
          else if (cond_toms && kpa_uniform) hipLaunchKernelGGL((k_cond_lean<T, 15, true, 2>), gs, bl, 0, st, nphys, a, kpa_value);
          else if (cond_toms) hipLaunchKernelGGL((k_cond_lean<T, 15, false, 2>), gs, bl, 0, st, nphys, a, T(0));
          else if (dbg(LCX_DBG_COND_LEAN_R3)) hipLaunchKernelGGL((k_cond_lean<T, 11, false, 1>), gs, bl, 0, st, nphys, a, T(0));
          // (round 5, measured and kept behind a switch: the workgroup folded behind the solver's first loop trip -- the same bits, 6 % fewer
          // vector instructions per wave at a higher lane use, and not faster: the chip runs this kernel at its package power cap, where
          // what a launch costs is the lanes that compute, not the instructions that issue; see k_cond_lean_fold)
          else if (dbg(LCX_DBG_COND_FOLD) && kpa_uniform) hipLaunchKernelGGL((k_cond_lean_fold<T, true>), gs, bl, 0, st, nphys, a, kpa_value);
          else if (dbg(LCX_DBG_COND_FOLD)) hipLaunchKernelGGL((k_cond_lean_fold<T, false>), gs, bl, 0, st, nphys, a, T(0));
          else if (!dbg(LCX_DBG_COND_WQ)) {
            if (dbg(LCX_DBG_COND_PROBE) && kpa_uniform) {      // (measurement only, see k_cond_probe)
              hipLaunchKernelGGL((k_cond_probe<T, 0>), gs, bl, 0, st, nphys, a, kpa_value); hipLaunchKernelGGL((k_cond_probe<T, 1>), gs, bl, 0, st, nphys, a, kpa_value);
              hipLaunchKernelGGL((k_cond_probe<T, 2>), gs, bl, 0, st, nphys, a, kpa_value); hipLaunchKernelGGL((k_cond_probe<T, 3>), gs, bl, 0, st, nphys, a, kpa_value);
              hipLaunchKernelGGL((k_cond_probe<T, 4>), gs, bl, 0, st, nphys, a, kpa_value); hipLaunchKernelGGL((k_cond_probe<T, 5>), gs, bl, 0, st, nphys, a, kpa_value);
              hipLaunchKernelGGL((k_cond_probe<T, 6>), gs, bl, 0, st, nphys, a, kpa_value);
            }
            if (want_list) lst = list_parts(gs.x, BS);
            if (lst.rec && lst.budget == 2u) {                  // (dbg COND_BUDGET: two trips as straight-line code)
              if (kpa_uniform) hipLaunchKernelGGL((k_cond_lean<T, 15, true, 0, 2>), gs, bl, 0, st, nphys, a, kpa_value, lst);
              else hipLaunchKernelGGL((k_cond_lean<T, 15, false, 0, 2>), gs, bl, 0, st, nphys, a, T(0), lst);
            }
            else if (lst.rec) {                                 // (any other budget: the run-time form of the loop)
              if (kpa_uniform) hipLaunchKernelGGL((k_cond_lean<T, 15, true, 0, 0>), gs, bl, 0, st, nphys, a, kpa_value, lst);
              else hipLaunchKernelGGL((k_cond_lean<T, 15, false, 0, 0>), gs, bl, 0, st, nphys, a, T(0), lst);
            }
            else if (kpa_uniform) hipLaunchKernelGGL((k_cond_lean<T, 15, true>), gs, bl, 0, st, nphys, a, kpa_value, lst);
            else hipLaunchKernelGGL((k_cond_lean<T, 15, false>), gs, bl, 0, st, nphys, a, T(0), lst);
            listed = lst.ent != nullptr;
          }
          else {
            // round 6: a wave walks n_batch batches of 64 slots and keeps the droplets whose first loop trip has not converged on a queue
            // of its own in LDS (k_cond_lean_wq: the same bits).  dbg_cond_budget > 0: that many batches (tests, measurements)
            const unsigned chunks = nblk(nphys);
            unsigned nb = 8u;
            while (nb > 1 && chunks / nb < 2048u) nb /= 2;           // (a small box: enough workgroups to fill the device first)
            if (o.dbg_cond_budget > 0 && (o.dbg_cond_budget & 255)) nb = unsigned(o.dbg_cond_budget & 255);
            a.xcd_group = std::max(1u, a.xcd_group / nb);
            const dim3 gw((chunks + nb - 1) / nb);
            if (want_list) { lst = list_parts(gw.x, size_t(BS) * nb); lst.budget = 100u; }      // (its queue takes the droplets that need more trips)
            const T kv = kpa_uniform ? kpa_value : T(0);
            launch_cond_lean_wq<T>(gw, st, wq_params<T>{nphys, a, kv, lst, nb}, kpa_uniform, dbg(LCX_DBG_COND_WQ_CAP128) ? 128 : 96, dbg(LCX_DBG_COND_WQ_PF2) ? 2 : dbg(LCX_DBG_COND_WQ_PF) ? 1 : 0);
            listed = lst.ent != nullptr;
          }
        }
        else if (cond_toms) { last_cond_kernel = LCX_CK_LEAN_TOMS748_SORTED; hipLaunchKernelGGL((k_cond_lean<T, 15, false, 2>), gr, bl, 0, st, npart, a, T(0)); }
        else {
          last_cond_kernel = LCX_CK_LEAN_SORTED;
          if (want_list) lst = list_parts(gr.x, BS);
          if (lst.rec) hipLaunchKernelGGL((k_cond_lean<T, 15, false, 0, 0>), gr, bl, 0, st, npart, a, T(0), lst);
          else hipLaunchKernelGGL((k_cond_lean<T, 15, false>), gr, bl, 0, st, npart, a, T(0), lst);
          listed = lst.ent != nullptr;
        }
        // the listed droplets (brackets that may hold several roots): the reference's iterates, on the same stream ahead of the per-cell
        // finish -- launched BEHIND the fork of the in-cell ranking below, so that its few thousand waves run next to the ranking's
        if (listed) {
          const bool uni = kpa_uniform && cond_in_storage_order;
          const T kv = uni ? kpa_value : T(0);
          launch_listed = [this, a, lst, uni, kv, bl]() {
            // (the host does not know the counts: workgroups for 1 % of the droplets -- the settled boxes list 0.1 %, a swinging one 1 % -- at a droplet per lane,
            // more droplets by the stride; a workgroup that finds nothing leaves at once)
            const unsigned per_part = unsigned(std::min<size_t>(2048, std::max<size_t>(8, lst.shard_cap / 100 / BS + 1)));
            if (lst.rec) {
              const unsigned n_resume = DEFER_SHARDS * ((lst.rec_cap + BS - 1) / BS);
              if (uni) hipLaunchKernelGGL((k_cond_lean_resume<T, true>), dim3(n_resume), bl, 0, st, a, lst, kv);
              else hipLaunchKernelGGL((k_cond_lean_resume<T, false>), dim3(n_resume), bl, 0, st, a, lst, kv);
            }
            if (uni) hipLaunchKernelGGL((k_cond_lean_listed<T, true>), dim3(DEFER_SHARDS * per_part), bl, 0, st, a, lst, kv);
            else hipLaunchKernelGGL((k_cond_lean_listed<T, false>), dim3(DEFER_SHARDS * per_part), bl, 0, st, a, lst, kv);
          };
        }
      }
      else if (fast) {
        a.pre = reinterpret_cast<const cond_cell_fast<T> *>(cond_pre.p);
        // two passes: a short iteration budget first, the droplets that need more in a dense second launch (k_cond_fast)
        // The second launch costs what its slowest wave costs (~60 us) however few droplets it holds, and the first pass saves ~2.3 us
        // per million droplets: one pass below 2^25 droplets (a 16-plane slab of C3, 16.7e6 SDs: 0.96 ms in two passes, 0.91 in one).
        // opts_init.dbg_cond_budget: test / measurement switch (the parity tests force the two-pass form at their small sizes with it)
        const int budget = o.dbg_cond_budget > 0 ? o.dbg_cond_budget : o.dbg_cond_budget < 0 ? 0 : (npart >= (size_t(1) << 25) ? 6 : 0);
        // (`rank` is free between the sorts; part s holds at most the positions of the workgroups b with b % DEFER_SHARDS == s)
        cond_defer df{rnk(), defer_cnt.p, size_t(nblk(nblk(npart), DEFER_SHARDS)) * BS, unsigned(budget)};
        if (size_t(DEFER_SHARDS) * df.shard_cap > cap) df.budget = 0;           // (tiny set-ups: the parts do not fit the scratch)
        const bool fold = !dbg(LCX_DBG_COND_NO_FOLD);                             // (test / measurement switch)
        last_cond_kernel = LCX_CK_TOMS748_TWO_PASS;
        if (fold) hipLaunchKernelGGL((k_cond_fast_fold<T, 11>), gr, bl, 0, st, npart, a, df);
        else hipLaunchKernelGGL((k_cond_fast<T, 11, false>), gr, bl, 0, st, npart, a, df);
        if (df.budget) {
          const unsigned per_shard = std::max(1u, std::min(nblk(npart / 8 + 1), 256u * 64u) / DEFER_SHARDS);
          hipLaunchKernelGGL((k_cond_fast<T, 11, true>), dim3(per_shard * DEFER_SHARDS), bl, 0, st, npart, a, df);
        }
      }
      else if (o.strict_fp) { last_cond_kernel = LCX_CK_STRICT; hipLaunchKernelGGL((k_cond<T, false>), gr, bl, 0, st, npart, a); }
      else { last_cond_kernel = LCX_CK_FAST_PER_DROPLET_SETUP; hipLaunchKernelGGL((k_cond<T, true>), gr, bl, 0, st, npart, a); }
    }
    if (carry_scatter) {                                      // the in-cell ranking, behind the kernel that scattered
      // (on its own stream when the list of crowded cells is on the host already, i.e. nothing in it waits for the device)
      // (not while every stage is being timed, lcx_set_profiling(1): the stage table is of stages that run one after the other)
      if (!dbg(LCX_DBG_NO_RANK_OVERLAP) && profiling != 1 && meta_version == cells_version) {
        if (!st_rank) {
          HIPCHK(hipStreamCreateWithFlags(&st_rank, hipStreamNonBlocking));
          HIPCHK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&ev_rank, hipEventDisableTiming));
        }
        HIPCHK(hipEventRecord(ev_fork, st));
        HIPCHK(hipStreamWaitEvent(st_rank, ev_fork, 0));
        {
          // (swapped back whatever happens inside: an exception between the two swaps would leave every later launch, and the stream
          // that lcx_stream() hands out, on the side stream for the object's lifetime)
          struct SwapBack { hipStream_t &a, &b; SwapBack(hipStream_t &a_, hipStream_t &b_) : a(a_), b(b_) { std::swap(a, b); } ~SwapBack() { std::swap(a, b); } } guard(st, st_rank);
          finish_deferred_sort(true);
          HIPCHK(hipEventRecord(ev_rank, st));
        }
        rank_pending = true;
      } else finish_deferred_sort(true);
    }
    // ("cond_listed": a stage of its own since round 6 -- "cond" is the first pass's kernel alone, what the roofline prices)
    if (launch_listed) { Range r(this, "cond_listed"); launch_listed(); launch_listed = nullptr; }
    {
      Range r(this, "cond_cellfinish");
      // (a kernel that carried the scatter has left each droplet's change at the droplet's place in the sorted order: no gather)
      launch_cellfinish(step, sstp_cond, fast, cond_in_storage_order && !carry_scatter ? sid() : (const uint32_t *)nullptr);
      cond_in_storage_order = false;
    }
  }
  // every super-droplet of the run carries the same hygroscopicity: one dry distribution (or one (kappa, rd_insol) key of dry_sizes),
  // nothing set by the caller (set_particles), so that coalescence never mixes two values -- the condensation kernel then takes it as
  // a scalar.  (LCX_DBG_KPA_ARRAY: the array is read all the same.)
  bool kpa_uniform = false; T kpa_value = T(0);
  int last_cond_kernel = 0;                  // (enum lcx_cond_kernel of the last condensation launch: "raw_mode")
  std::function<void()> launch_listed;       // (cond_substep: k_cond_lean_listed, queued behind the fork of the in-cell ranking)
  const bool cond_storage_order = !dbg(LCX_DBG_COND_SORTED_ORDER);      // (measurement switch: the positional form)
  bool cond_in_storage_order = false;
  // per-cell sums of n rw^3 before / after the substep + update_th_rv.  Strict arithmetic: the ordered single-lane walk (the
  // reference's summation order); fast: eight lanes per cell, or a whole wave per cell where cells are crowded
  void launch_cellfinish(int step, int sstp, bool delta = false, const uint32_t *gather = nullptr)
  {
    const int dl = delta ? 1 : 0;
    if (!o.strict_fp && ncell >= 4096 && npart / ncell >= 192)
      hipLaunchKernelGGL(k_cond_cellfinish_wave<T>, dim3(nblk(ncell, BS / WAVE)), dim3(BS), 0, st, ncell, cell_start.p, m3_before.p, m3_after.p, dv.p, rhod.p,
                         rv.p, th.p, Tk.p, rw_mom3.p, step, sstp, n_dims, dl, gather);
    else if (!o.strict_fp && delta && !gather && !dbg(LCX_DBG_FINISH_STAGED))
      hipLaunchKernelGGL(k_cond_cellfinish_direct<T>, dim3(nblk(ncell * 8)), dim3(BS), 0, st, ncell, cell_start.p, m3_after.p, dv.p, rhod.p, rv.p, th.p, Tk.p,
                         rw_mom3.p, n_dims);
    else if (!o.strict_fp) {
      const int cfc = std::min(cf_cells(), BS / 8);
      hipLaunchKernelGGL((k_cond_cellfinish<T, 8>), dim3(nblk(ncell, cfc)), dim3(BS), 0, st, ncell, cfc, cell_start.p, m3_before.p, m3_after.p, dv.p, rhod.p,
                         rv.p, th.p, Tk.p, rw_mom3.p, step, sstp, n_dims, dl, gather);
    } else
      hipLaunchKernelGGL((k_cond_cellfinish<T, 1>), dim3(nblk(ncell, cf_cells())), dim3(BS), 0, st, ncell, cf_cells(), cell_start.p, m3_before.p, m3_after.p, dv.p, rhod.p,
                         rv.p, th.p, Tk.p, rw_mom3.p, step, sstp, n_dims, dl, gather);
  }
  // hskpng_rc2.ipp:14-32
  void hskpng_approximate_rc2_invalid()
  {
    if (sstp_cond_act == 1 || !allow_sstp_cond || !nphys) return;
    hipLaunchKernelGGL(k_rc2<T>, dim3(nblk(nphys)), dim3(BS), 0, st, nphys, A.rd3.p, A.kpa.p, T(T(o.rc2_T) + T(273.15)), A.ext[ix_rc2].p);
  }
  // particles_step.ipp:199-236: per-particle substepping; sorted (plain order) on entry
  void cond_perparticle(double RH_max, bool turb_cond)
  {
    if (!npart) return;
    last_cond_kernel = LCX_CK_PER_PARTICLE;
    pp_args<T> a{};
    a.sorted_id = sid(); a.sorted_ijk = sijk();
    a.n = A.n.p; a.rd3 = A.rd3.p; a.kpa = A.kpa.p; a.vt = A.vt.p; a.rw2 = A.rw2.p;
    a.pp_rv = A.ext[ix_rv].p; a.pp_th = A.ext[ix_th].p; a.pp_rh = A.ext[ix_rh].p; a.pp_p = o.const_p ? A.ext[ix_p].p : nullptr;
    a.rv = rv.p; a.th = th.p; a.rhod = rhod.p; a.p = p.p;
    a.dv = dv.p; a.lambda_D = lambda_D.p; a.lambda_K = lambda_K.p; a.rc2 = use_rc2 ? A.ext[ix_rc2].p : nullptr;
    a.ssp = turb_cond ? A.ext[ix_ssp].p : nullptr; a.dot_ssp = turb_cond ? A.ext[ix_dot_ssp].p : nullptr;
    a.m3_before = m3_before.p; a.m3_after = m3_after.p;
    a.dt = T(dt); a.RH_max = T(RH_max); a.eps = eps_tol; a.cond_mlt = T(2.); a.n_iter = 100u;
    a.adapt_eps = T(o.sstp_cond_adapt_drw2_eps); a.adapt_max = T(o.sstp_cond_adapt_drw2_max);
    a.sstp_cond = sstp_cond; a.sstp_cond_act = sstp_cond_act; a.th_dry = o.th_dry; a.const_p = o.const_p; a.RH_formula = o.RH_formula;
    a.n_dims = n_dims;
    const dim3 grid(nblk(npart)), blk(BS);
    if (!o.sstp_cond_mix) {
      {
        Range r(this, o.adaptive_sstp_cond ? "cond_perparticle_adaptive" : "cond_perparticle");
        if (o.adaptive_sstp_cond) {
          if (o.strict_fp) hipLaunchKernelGGL((k_pp_cond_adaptive<T, false>), grid, blk, 0, st, npart, a);
          else             hipLaunchKernelGGL((k_pp_cond_adaptive<T, true>), grid, blk, 0, st, npart, a);
        } else {
          if (o.strict_fp) hipLaunchKernelGGL((k_pp_cond_nomix<T, false>), grid, blk, 0, st, npart, a);
          else             hipLaunchKernelGGL((k_pp_cond_nomix<T, true>), grid, blk, 0, st, npart, a);
        }
      }
      // save_liq_ice_content_before_change + calc_liq_ice_content_change + update_th_rv: ordered per-cell sums of n rw^3
      Range r(this, "cond_cellfinish");
      launch_cellfinish(0, 1);
      return;
    }
    Range r(this, "cond_perparticle_mix");
    for (int k = 0; k < (o.const_p ? 4 : 3); ++k) pp_dlt[k].alloc(cap);
    pp_rw3s.alloc(cap); pp_dst_rv.alloc(ncell); pp_dst_th.alloc(ncell);
    a.dlt_rv = pp_dlt[0].p; a.dlt_th = pp_dlt[1].p; a.dlt_rh = pp_dlt[2].p; a.dlt_p = pp_dlt[3].p; a.rw3s = pp_rw3s.p;
    a.drv = m3_before.p; a.dth = m3_after.p; a.dst_rv = pp_dst_rv.p; a.dst_th = pp_dst_th.p;
    for (int step = 0; step < sstp_cond; ++step) {
      a.step = step;
      if (o.strict_fp) hipLaunchKernelGGL((k_pp_cond_mix<T, false>), grid, blk, 0, st, npart, a);
      else             hipLaunchKernelGGL((k_pp_cond_mix<T, true>), grid, blk, 0, st, npart, a);
      hipLaunchKernelGGL(k_cell_seqsum<T>, dim3(nblk(ncell, cf_cells())), dim3(BS), 0, st, ncell, cf_cells(), cell_start.p, m3_before.p, dv.p, rhod.p, 0, pp_dst_rv.p);
      hipLaunchKernelGGL(k_cell_seqsum<T>, dim3(nblk(ncell, cf_cells())), dim3(BS), 0, st, ncell, cf_cells(), cell_start.p, m3_after.p, dv.p, rhod.p, 0, pp_dst_th.p);
    }
    hipLaunchKernelGGL(k_pp_mix_finish<T>, dim3(nblk(ncell)), dim3(BS), 0, st, ncell, cell_start.p, sid(), A.ext[ix_rv].p, A.ext[ix_th].p,
                       pp_dst_rv.p, pp_dst_th.p, rv.p, th.p);
  }
  bool turb_adve_now = false;
  normal_src<T> rand_normal(size_t n)
  {
    if (!replay.empty()) {
      Replay r = std::move(replay.front()); replay.pop_front();
      if (r.kind != 2 || r.n < n) throw lcx_error("libcloudph++: rng replay queue does not match the requested rand_normal call");
      const T *ptr = r.u01->p;
      replay_keep_T.push_back(std::move(r.u01));
      return normal_src<T>{ptr, 0, 0};
    }
    return normal_src<T>{nullptr, ++rng_call, uint64_t(uint32_t(seed_now()))};
  }
  // hskpng_tke + hskpng_turb_vel (+ hskpng_turb_dot_ss): particles_step.ipp:406-427
  void sgs_turbulence(const lcx_opts_t &opts)
  {
    Range r(this, "sgs_turbulence");
    if (!replay.empty()) ensure_compact();
    hipLaunchKernelGGL(k_tke_tau<T>, dim3(nblk(ncell)), dim3(BS), 0, st, ncell, m1(o.nz), SGS_mix_len.p, diss_rate.p, tau_cell.p);
    const int comp[3] = {ix_up, ix_wp, ix_vp};                                           // the reference's order: up, wp, vp
    const int lo = opts.turb_adve ? 0 : 1, hi = opts.turb_adve ? n_dims : 2;
    for (int i = lo; i < hi; ++i) {
      const normal_src<T> rs = rand_normal(npart);
      hipLaunchKernelGGL(k_turb_vel<T>, dim3(nblk(nphys)), dim3(BS), 0, st, nphys, ijk.p, tau_cell.p, diss_rate.p, T(dt), rs, A.ext[comp[i]].p);
    }
    if (opts.turb_cond) {
      need_nfiltered();
      hskpng_sort();
      hipLaunchKernelGGL(k_nfilt<T>, dim3(nblk(nphys)), dim3(BS), 0, st, nphys, 0, 0, A.n.p, A.rw2.p, T(0), T(0), n_filtered.p);     // moms_all
      selected_before_counting = true;
      moms_sum(A.rw2.p, T(1. / 2), 0, false);                                            // sum n r_w per cell, not specific
      hipLaunchKernelGGL(k_tau_rlx<T>, dim3(nblk(ncell)), dim3(BS), 0, st, ncell, cell_start.p, count_mom.p, dv.p, tau_rlx.p);
      hipLaunchKernelGGL(k_turb_dot_ss<T>, dim3(nblk(nphys)), dim3(BS), 0, st, nphys, ijk.p, tau_rlx.p, A.ext[ix_ssp].p, A.ext[ix_wp].p, A.ext[ix_dot_ssp].p);
      selected_before_counting = false;
    }
  }
  void adjust_timesteps(double dt_)
  {                                                                                      // particles_impl_adjust_timesteps.ipp:13-24
    if (dt_ > 0 && !o.variable_dt_switch) throw lcx_error("libcloudph++: opts.dt specified, but opts_init.variable_dt_switch is false.");
    sstp_cond = dt_ > 0 && o.sstp_cond > 1 ? int(std::ceil(o.sstp_cond * dt_ / o.dt)) : o.sstp_cond;
    sstp_cond_act = dt_ > 0 && o.sstp_cond_act > 1 ? int(std::ceil(o.sstp_cond_act * dt_ / o.dt)) : o.sstp_cond_act;
    sstp_coal = dt_ > 0 && o.sstp_coal > 1 ? int(std::ceil(o.sstp_coal * dt_ / o.dt)) : o.sstp_coal;
    dt = dt_ > 0 ? dt_ : o.dt;
  }

  // ------------------------------------------------------------------------------------------
  // coalescence (coal.ipp:273-546)
  // ------------------------------------------------------------------------------------------
  // the previous coalescence substep has left velocities invalid: the in-cell ranking of the next one refreshes them on its way
  // (order_cells), or the pass of its own does where another ranking kernel runs
  bool vt_fix_pending = false;
  void coal(double dt_sub, bool turb_coal = false)
  {
    if (!replay.empty()) ensure_compact();     // un[id] of a replayed CPU stream is indexed by the reference's (compact) ids
    hskpng_sort_helper(true);
    if (vt_fix_pending) { vt_fix_pending = false; hskpng_vterm(true); }
    if (npart < 2) { if (npart) (void)rand_u01(npart); return; }
    Range r(this, "coal");
    const u01_src<T> rs = rand_u01(npart);
    if (ix_tag >= 0) record_rng(rs);
    const bool onishi = o.kernel == LCX_KERNEL_ONISHI_HALL || o.kernel == LCX_KERNEL_ONISHI_HALL_DAVIS_NO_WAALS;
    // diss == nullptr stands for the reference's constant-zero dissipation rate when opts.turb_coal is off (coal.ipp:392-403,439-451)
    coal_kernel_cfg<T> kc{o.kernel, n_user_params, T(kernel_r_max), kparams.p, eta.p, rhod.p, turb_coal ? diss_rate.p : nullptr};
    const bool kappa_pass = o.n_dry_distros + n_size_keys > 1;
    auto launch = [&](auto kern, bool need_col = true) {
      hipLaunchKernelGGL(kern, dim3(nblk((npart + 1) / 2)), dim3(BS), 0, st, npart, sid(), sijk(), cell_start.p, A.n.p, A.rw2.p, A.vt.p,
                         A.rd3.p, need_col ? col.p : (T *)nullptr, dv.p, T(dt_sub), kc, rs, int(pure_const_multi), d_flag.p, use_rc2 ? A.ext[ix_rc2].p : nullptr,
                         ix_ict >= 0 ? A.ext[ix_ict].p : nullptr, coal_marks_dead ? ijk.p : nullptr);
    };
    const bool tabulated = o.kernel != LCX_KERNEL_GOLOVIN && o.kernel != LCX_KERNEL_GEOMETRIC && o.kernel != LCX_KERNEL_LONG;
    if (onishi) launch(k_coal<T, true>);
    // (the production kernel writes the collision record only for the kappa pass: nothing else reads it outside a replayed run)
    else if (tabulated && !pure_const_multi && !rs.arr && !use_rc2 && ix_ict < 0 && coal_marks_dead) launch(k_coal<T, false, true>, kappa_pass);
    else launch(k_coal<T, false>);
    if (kappa_pass)
      hipLaunchKernelGGL(k_coal_kappa<T>, dim3(nblk(npart)), dim3(BS), 0, st, npart, sid(), col.p, A.kpa.p, A.rd3.p);
  }

  // ------------------------------------------------------------------------------------------
  // adve + sedi + subs + bcnd in one pass
  // ------------------------------------------------------------------------------------------
  // reindex: also produce the new cell index / histogram / rank / dead count (single-device post_copy fused in)
  void move(bool do_adve, bool do_sedi, bool do_subs, bool do_bcnd, bool reindex = false)
  {
    if (n_dims == 0 || nphys == 0) {
      if (do_bcnd) { lft_count = rgt_count = 0; if (dev_exchange) HIPCHK(hipMemsetAsync(scan_total.p, 0, 2 * sizeof(uint32_t), st)); }
      return;
    }
    Range r(this, "move(adve+sedi+bcnd)");
    move_args<T> a;
    a.n_part = nphys; a.g = g;
    a.dx = T(o.dx); a.dy = T(o.dy); a.dz = T(o.dz); a.x0 = T(o.x0); a.y0 = T(o.y0); a.z0 = T(o.z0); a.x1 = T(o.x1); a.y1 = T(o.y1); a.z1 = T(o.z1);
    a.dt = T(dt);
    a.x = A.x.p; a.y = A.y.p; a.z = A.z.p; a.vt = A.vt.p; a.rw2 = A.rw2.p; a.rd3 = A.rd3.p; a.n = A.n.p; a.ijk = ijk.p;
    a.courant_x = courant_x.p; a.courant_y = courant_y.p; a.courant_z = courant_z.p; a.w_LS = w_LS.p;
    a.up = a.vp = a.wp = nullptr;
    if (turb_adve_now && do_bcnd) {                 // (the full step: turb_adve follows adve, turb_adve.ipp)
      a.up = ix_up >= 0 ? A.ext[ix_up].p : A.ext[ix_wp].p;      // a.up != nullptr switches the block on; unused components are not read
      a.vp = ix_vp >= 0 ? A.ext[ix_vp].p : nullptr; a.wp = ix_wp >= 0 ? A.ext[ix_wp].p : nullptr;
    }
    a.do_adve = do_adve; a.scheme = adve_scheme; a.halo = halo; a.do_sedi = do_sedi; a.do_subs = do_subs; a.do_bcnd = do_bcnd;
    a.distmem = distmem(); a.bcond_lft = o.bcond_lft; a.bcond_rgt = o.bcond_rgt;
    a.open_side_walls = o.open_side_walls; a.periodic_topbot = o.periodic_topbot_walls;
    const bool want_puddle = do_bcnd && n_dims > 1 && !o.periodic_topbot_walls;
    const unsigned blocks = nblk(nphys);
    a.puddle_partial = want_puddle ? puddle_partial.p : nullptr;
    a.mig = mig.p; a.wg_mig = (do_bcnd && distmem()) ? wg_mig.p : nullptr;
    // (no memsets: every lane stores its migrant flag, and the histogram and the dead count are cleared behind each sort's scan)
    a.reindex = reindex; a.ijk_out = ijk.p; a.cnt = cell_cnt.p; a.rank = rnk(); a.dead_count = d_dead_p();
    a.check_n = !coal_marks_dead || zero_n_unmarked;
    if (reindex) zero_n_unmarked = false;
    const bool pc = adve_scheme == LCX_ADVE_PRED_CORR, tb = a.up != nullptr;
    if (pc && tb) hipLaunchKernelGGL((k_move<T, true, true>), dim3(blocks), dim3(BS), 0, st, a);
    else if (pc) hipLaunchKernelGGL((k_move<T, true, false>), dim3(blocks), dim3(BS), 0, st, a);
    else if (tb) hipLaunchKernelGGL((k_move<T, false, true>), dim3(blocks), dim3(BS), 0, st, a);
    else if (n_dims == 3) {
      const bool full = do_adve && do_sedi && !do_subs && do_bcnd && reindex && halo == 0 && !o.open_side_walls && !o.periodic_topbot_walls &&
                        (adve_scheme == LCX_ADVE_EULER || adve_scheme == LCX_ADVE_IMPLICIT);
      const int spec = !full ? MOVE_3D : MOVE_3D | MOVE_FULL | (a.distmem ? MOVE_DISTMEM : 0) | (adve_scheme == LCX_ADVE_IMPLICIT ? MOVE_IMPLICIT : 0);
      switch (spec) {
        case MOVE_3D | MOVE_FULL: hipLaunchKernelGGL((k_move<T, false, false, MOVE_3D | MOVE_FULL>), dim3(blocks), dim3(BS), 0, st, a); break;
        case MOVE_3D | MOVE_FULL | MOVE_DISTMEM: hipLaunchKernelGGL((k_move<T, false, false, MOVE_3D | MOVE_FULL | MOVE_DISTMEM>), dim3(blocks), dim3(BS), 0, st, a); break;
        case MOVE_3D | MOVE_FULL | MOVE_IMPLICIT: hipLaunchKernelGGL((k_move<T, false, false, MOVE_3D | MOVE_FULL | MOVE_IMPLICIT>), dim3(blocks), dim3(BS), 0, st, a); break;
        case MOVE_3D | MOVE_FULL | MOVE_DISTMEM | MOVE_IMPLICIT: hipLaunchKernelGGL((k_move<T, false, false, MOVE_3D | MOVE_FULL | MOVE_DISTMEM | MOVE_IMPLICIT>), dim3(blocks), dim3(BS), 0, st, a); break;
        default: hipLaunchKernelGGL((k_move<T, false, false, MOVE_3D>), dim3(blocks), dim3(BS), 0, st, a);
      }
    }
    else hipLaunchKernelGGL((k_move<T, false, false>), dim3(blocks), dim3(BS), 0, st, a);
    if (reindex && !distmem()) list_big_from_hist();                    // (with neighbours: after their immigrants are in, exch_unpack)
    if (want_puddle && dev_exchange) puddle_pending_blocks = blocks;      // (reduced after the emigrants are on their way, see lcx_multi.hpp)
    else if (want_puddle) puddle_reduce(blocks);
    if (do_bcnd && distmem()) build_migrant_lists();
  }
  unsigned puddle_pending_blocks = 0;
  void puddle_reduce_deferred() { if (puddle_pending_blocks) puddle_reduce(puddle_pending_blocks); puddle_pending_blocks = 0; }
  void puddle_reduce(unsigned blocks)
  {
    {
      const size_t slices = 256, per = (size_t(blocks) + slices - 1) / slices;
      hipLaunchKernelGGL(k_sum_partials, dim3(unsigned(slices)), dim3(BS), 0, st, puddle_partial.p, size_t(blocks), per, puddle_sum.p + 4, (double *)nullptr);
      // running totals stay on the device (same additions in the same order as on the host); diag_puddle reads them
      hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(BS), 0, st, puddle_sum.p + 4, slices, slices, puddle_sum.p, puddle_acc.p);
    }
  }
  void build_migrant_lists()
  {
    // from the per-workgroup counts of k_move (one word per 256 SDs): tiles of BS such words
    const size_t n_wg = nblk(nphys), tiles = (n_wg + BS - 1) / BS;
    size_t *cnt[2] = {&lft_count, &rgt_count};
    if (!tiles) { lft_count = rgt_count = 0; if (dev_exchange) HIPCHK(hipMemsetAsync(scan_total.p, 0, 2 * sizeof(uint32_t), st)); return; }
    uint32_t *off_l = wg_mig.p + (size_t(nblk(cap)) + 1), *off_r = off_l + (size_t(nblk(cap)) + 1);
    hipLaunchKernelGGL(k_mig_tiles3, dim3(unsigned(tiles)), dim3(BS), 0, st, wg_mig.p, n_wg, uint32_t(tiles), tile_sums.p, d_dead_p());
    hipLaunchKernelGGL(k_scan_sums2, dim3(2), dim3(1024), 0, st, tile_sums.p, tiles, scan_total.p);
    hipLaunchKernelGGL(k_mig_off, dim3(unsigned(tiles)), dim3(BS), 0, st, wg_mig.p, n_wg, uint32_t(tiles), tile_sums.p, off_l, off_r);
    hipLaunchKernelGGL(k_mig_ids4, dim3(nblk(n_wg, BS / WAVE)), dim3(BS), 0, st, mig.p, nphys, wg_mig.p, n_wg, off_l, off_r, mig_ids[0].p, mig_ids[1].p);
    if (dev_exchange) return;                         // multi_HIP: the counts stay on the device (scan_total[0..1]), see exch_*
    uint32_t tot[2];
    read_back(tot, scan_total.p, 2);                  // one host sync for both directions
    *cnt[0] = tot[0]; *cnt[1] = tot[1];
  }

  // ------------------------------------------------------------------------------------------
  // initialisation (particles_init.ipp:16-131)
  // ------------------------------------------------------------------------------------------
  double eval_distro(const lcx_distro_t &d, double lnrd) const
  {
    if (d.fn) return d.fn(lnrd, d.user);
    double res = 0;
    if (d.n_modes < 0) { const double q = std::pow(std::exp(lnrd), 3) / std::pow(d.mean_rd[0], 3); return d.n_stp[0] * 3. * q * std::exp(-q); }
    for (int m = 0; m < d.n_modes; ++m)
      res += d.n_stp[m] / std::sqrt(2 * M_PI) / std::log(d.sdev[m]) * std::exp(-std::pow((lnrd - std::log(d.mean_rd[m])), 2) / 2. / std::pow(std::log(d.sdev[m]), 2));
    return res;
  }
  void init_dist_analysis_sd_conc(const lcx_distro_t &d, n_t sd_conc, T dv0)
  {                                                                                      // init_dist_analysis.ipp:17-77
    const T vol = n_dims == 0 ? dv0 : T(T(o.dx) * T(o.dy) * T(o.dz));
    if (o.rd_min >= 0 && o.rd_max >= 0) {
      const T rd_min = T(o.rd_min), rd_max = T(o.rd_max);
      multiplier = T(std::log(rd_max / rd_min) / sd_conc * T(1) * vol);
      log_rd_min = T(std::log(rd_min)); log_rd_max = T(std::log(rd_max));
    } else if (o.rd_min < 0 && o.rd_max < 0) {
      T rd_min = T(1e-14), rd_max = T(1e-3);                                             // config.hpp:23-24
      bool found = false;
      while (!found) {
        multiplier = T(std::log(rd_max / rd_min) / sd_conc * T(1) * vol);
        log_rd_min = T(std::log(rd_min)); log_rd_max = T(std::log(rd_max));
        const n_t n_min = n_t(T(eval_distro(d, log_rd_min)) * T(multiplier)), n_max = n_t(T(eval_distro(d, log_rd_max)) * T(multiplier));
        if (rd_min == T(1e-14) && n_min != 0) throw lcx_error("Initial dry radii distribution is non-zero (" + std::to_string(n_min) + ") for rd_min_init (1e-14)");
        if (rd_max == T(1e-3) && n_max != 0) throw lcx_error("Initial dry radii distribution is non-zero (" + std::to_string(n_max) + ") for rd_max_init (0.001)");
        if (n_min == 0) rd_min *= T(1.01); else if (n_max == 0) rd_max /= T(1.01); else found = true;
      }
    } else throw lcx_error("opts_init.rd_min * opts_init.rd_max < 0");
  }
  // ---- constant-multiplicity and large-tail initialisation (host analysis as in the reference, sampling on the device) ----
  // Brent's minimiser: the reference calls boost::math::tools::brent_find_minima (init_dist_analysis.ipp:95); Boost is not
  // vendored by the reference and its version is not pinned, so this is the published algorithm (Brent 1973, ch. 5)
  // restated, identical to the oracle's (parity with the reference UNPINNED for this routine).
  template <class F> static double brent_find_minimum(F f, double min, double max, int bits, unsigned &max_iter, double &fmin)
  {
    if (bits > 53 / 2) bits = 53 / 2;
    const double tolerance = std::ldexp(1.0, 1 - bits), golden = 0.3819660f;
    double x, w, v, u, delta, delta2, fu, fv, fw, fx, mid, fract1, fract2;
    x = w = v = max;
    fw = fv = fx = f(x);
    delta2 = delta = 0;
    unsigned count = max_iter;
    do {
      mid = (min + max) / 2;
      fract1 = tolerance * std::fabs(x) + tolerance / 4;
      fract2 = 2 * fract1;
      if (std::fabs(x - mid) <= (fract2 - (max - min) / 2)) break;
      if (std::fabs(delta2) > fract1) {
        double r = (x - w) * (fx - fv), q = (x - v) * (fx - fw), p = (x - v) * q - (x - w) * r;
        q = 2 * (q - r);
        if (q > 0) p = -p;
        q = std::fabs(q);
        const double td = delta2;
        delta2 = delta;
        if ((std::fabs(p) >= std::fabs(q * td / 2)) || (p <= q * (min - x)) || (p >= q * (max - x))) {
          delta2 = (x >= mid) ? min - x : max - x;
          delta = golden * delta2;
        } else {
          delta = p / q;
          u = x + delta;
          if (((u - min) < fract2) || ((max - u) < fract2)) delta = (mid - x) < 0 ? -std::fabs(fract1) : std::fabs(fract1);
        }
      } else {
        delta2 = (x >= mid) ? min - x : max - x;
        delta = golden * delta2;
      }
      u = (std::fabs(delta) >= fract1) ? x + delta : (delta > 0 ? x + std::fabs(fract1) : x - std::fabs(fract1));
      fu = f(u);
      if (fu <= fx) {
        if (u >= x) min = x; else max = x;
        v = w; w = x; x = u; fv = fw; fw = fx; fx = fu;
      } else {
        if (u < x) min = u; else max = u;
        if ((fu <= fw) || (w == x)) { v = w; w = u; fv = fw; fw = fu; }
        else if ((fu <= fv) || (v == x) || (v == w)) { v = u; fv = fu; }
      }
    } while (--count);
    max_iter -= count;
    fmin = fx;
    return x;
  }
  void init_dist_analysis_const_multi(const lcx_distro_t &d)
  {                                                                                      // init_dist_analysis.ipp:80-120
    if (o.rd_min >= 0 && o.rd_max >= 0) { log_rd_min = T(std::log(T(o.rd_min))); log_rd_max = T(std::log(T(o.rd_max))); }
    else if (o.rd_min < 0 && o.rd_max < 0) {
      unsigned n_iter = 100;
      double fmin;
      const T lnrd_max = T(brent_find_minimum([&](double x) { return double(T(eval_distro(d, T(x))) * T(-1)); }, double(T(std::log(T(1e-14)))),
                                              double(T(std::log(T(1e-3)))), 200, n_iter, fmin));
      const T bound = T(-T(fmin) / T(1e20));                                             // config.hpp:21 threshold
      struct lvl_t { const Particles *self; const lcx_distro_t *d; T bound; T operator()(T x) const { return T(self->eval_distro(*d, x)) + (-bound); } };
      const lvl_t lvl{this, &d, bound};
      const T lo = T(std::log(T(1e-14))), hi = T(std::log(T(1e-3)));
      log_rd_min = toms748_solve(lvl, lo, lnrd_max, lvl(lo), lvl(lnrd_max), eps_tol, 100u);
      log_rd_max = toms748_solve(lvl, lnrd_max, hi, lvl(lnrd_max), lvl(hi), eps_tol, 100u);
    } else throw lcx_error("opts_init.rd_min * opts_init.rd_max < 0");
  }
  DevBuf<uint32_t> init_off; DevBuf<T> init_cdf; DevBuf<beard77_cell<T>> vt_pre;
  // init_count_num.ipp:14-24,41-101 + init_ijk + init_dry_const_multi.ipp:20-80 + init_n_const_multi, then finalize
  void init_const_multi_like(const lcx_distro_t &d, n_t const_multi)
  {
    const T bin = T(1e-4), lo = T(log_rd_min), hi = T(log_rd_max);                       // config.hpp:20 bin_precision
    const int nb = int((hi - lo) / bin);
    T integral = (T(eval_distro(d, lo)) + T(eval_distro(d, hi))) / T(2.);
    for (int i = 1; i < nb; ++i) integral += T(eval_distro(d, lo + i * bin));
    integral = integral * bin;
    std::vector<T> dv_h = d2h(dv.p, ncell), rhod_h = d2h(rhod.p, ncell);
    std::vector<uint32_t> off(ncell + 1, 0);
    size_t total = 0;
    for (size_t c = 0; c < ncell; ++c) {                                                 // init_count_num_hlpr + conc_to_number
      T conc = integral;
      conc = conc * dv_h[c];
      if (!o.aerosol_independent_of_rhod) conc = rhod_h[c] / cst<T>::rho_stp * conc;
      if (!conc_factor_h.empty()) conc = conc * T(conc_factor_h[c % size_t(m1(o.nz))]);
      off[c] = uint32_t(total);
      total += size_t(n_t(conc / const_multi + T(0.5)));
      if (total >= (1ull << 32)) throw lcx_error("libcloudph++: n_sd_max must be < 2^32 per device (32-bit super-droplet ids)");
    }
    off[ncell] = uint32_t(total);
    const size_t n_old = npart, n_new = total;
    check_npart(n_old + n_new);
    npart = nphys = n_old + n_new;
    if (n_new == 0) return;
    const size_t ncdf = size_t((hi - lo) / bin + 1);
    std::vector<T> cdf(ncdf);
    for (size_t i = 0; i < ncdf; ++i) cdf[i] = T(eval_distro(d, lo + bin * i)) * 1;
    for (size_t i = 1; i < ncdf; ++i) cdf[i] = cdf[i - 1] + cdf[i];
    { const T back = cdf[ncdf - 1]; for (size_t i = 0; i < ncdf; ++i) cdf[i] = cdf[i] / back; }
    init_off.alloc(ncell + 1); init_cdf.alloc(ncdf);
    HIPCHK(hipMemcpyAsync(init_off.p, off.data(), (ncell + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(init_cdf.p, cdf.data(), ncdf * sizeof(T), hipMemcpyHostToDevice, st));
    const u01_src<T> rs = rand_u01(n_new);
    hipLaunchKernelGGL(k_init_const_multi<T>, dim3(nblk(n_new)), dim3(BS), 0, st, n_new, n_old, init_off.p, uint32_t(ncell), init_cdf.p, uint32_t(ncdf),
                       lo, bin, rs, const_multi, ijk.p, A.rd3.p, A.kpa.p, T(d.kappa), A.vt.p, A.n.p);
    hipLaunchKernelGGL(k_init_wet<T>, dim3(nblk(n_new)), dim3(BS), 0, st, n_new, n_old, A.rd3.p, A.kpa.p, ijk.p, RH.p, Tk.p, T(o.RH_max), A.rw2.p);
    init_positions(n_new, n_old);
    sync();                                                                              // the host tables go out of scope
  }
  void init_SD_with_distros()
  {
    T dv0 = 0;
    if (n_dims == 0) { HIPCHK(hipMemcpyAsync(&dv0, dv.p, sizeof(T), hipMemcpyDeviceToHost, st)); sync(); }
    T tot_lnrd_rng = 0;
    if (o.sd_conc > 0)
      for (auto &d : distros) { init_dist_analysis_sd_conc(d, o.sd_conc, dv0); tot_lnrd_rng += T(log_rd_max - log_rd_min); }
    for (auto &d : distros) {
      if (o.sd_const_multi > 0) {                                                        // init_SD_with_distros_const_multi.ipp:14-38
        init_dist_analysis_const_multi(d);
        if (log_rd_min >= log_rd_max) throw lcx_error("Distribution analysis error: rd_min >= rd_max");
        init_const_multi_like(d, o.sd_const_multi);
        continue;
      }
      init_dist_analysis_sd_conc(d, o.sd_conc, dv0);                                     // init_SD_with_distros_sd_conc.ipp:14-46
      if (log_rd_min >= log_rd_max) throw lcx_error("Distribution analysis error: rd_min >= rd_max");
      const T fraction = T(log_rd_max - log_rd_min) / tot_lnrd_rng;
      multiplier = T(T(multiplier) * T(o.sd_conc / n_t(int(fraction * o.sd_conc + 0.5))));
      const n_t per_cell = n_t(fraction * o.sd_conc);                                    // init_count_num.ipp:32-35
      const size_t n_old = npart, n_new = size_t(per_cell) * ncell;
      check_npart(n_old + n_new);
      npart = nphys = n_old + n_new;
      if (n_new == 0) continue;
      const unsigned nb = nblk(n_new);
      {
        const u01_src<T> rs = rand_u01(n_new);
        if (d.fn) fvals.alloc(n_new);
        hipLaunchKernelGGL(k_init_dry<T>, dim3(nb), dim3(BS), 0, st, n_new, n_old, per_cell, T(log_rd_min), T(log_rd_max), rs, ijk.p, A.rd3.p, A.kpa.p, T(d.kappa), A.vt.p,
                           d.fn ? fvals.p : (T *)nullptr);
      }
      lognormal_modes lm{d.n_modes, {0}, {0}, {0}};
      for (int m = 0; m < 4; ++m) { lm.mean_rd[m] = d.mean_rd[m]; lm.sdev[m] = d.sdev[m]; lm.n_stp[m] = d.n_stp[m]; }
      const T *fv = nullptr;
      if (d.fn) {                                                                        // host evaluation of the user functor (init_n.ipp:56-84)
        // The host evaluates n(ln rd) -- and, since round 4, the dry volume rd3 = exp(3 ln rd) that the reference takes ln rd back from
        // (init_dry_sd_conc.ipp:26-34, init_n.ipp:62-66): with the host's exp and log on both sides of that round trip the argument of
        // the user's function, hence the integer multiplicity, is the reference's bit for bit (the device's exp differs from the
        // host's in the last place now and then, which used to move n by one for < 0.1 % of the super-droplets)
        std::vector<T> h(n_new), r3(n_new);
        HIPCHK(hipMemcpyAsync(h.data(), fvals.p, n_new * sizeof(T), hipMemcpyDeviceToHost, st));      // (the drawn ln rd)
        sync();
        for (size_t i = 0; i < n_new; ++i) {
          r3[i] = T(std::exp(3 * h[i]));
          const T lnrd = T(std::log(r3[i]) / 3.);
          h[i] = T(d.fn(lnrd, d.user));
        }
        HIPCHK(hipMemcpyAsync(A.rd3.p + n_old, r3.data(), n_new * sizeof(T), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(fvals.p, h.data(), n_new * sizeof(T), hipMemcpyHostToDevice, st));
        sync();
        fv = fvals.p;
      }
      hipLaunchKernelGGL(k_init_n<T>, dim3(nb), dim3(BS), 0, st, n_new, n_old, A.rd3.p, ijk.p, fv, lm, T(multiplier), rhod.p, dv.p,
                         conc_factor_h.empty() ? (const T *)nullptr : conc_factor.p, m1(o.nz), o.aerosol_independent_of_rhod, n_dims,
                         T(T(o.dx) * T(o.dy) * T(o.dz)), A.n.p);
      hipLaunchKernelGGL(k_init_wet<T>, dim3(nb), dim3(BS), 0, st, n_new, n_old, A.rd3.p, A.kpa.p, ijk.p, RH.p, Tk.p, T(o.RH_max), A.rw2.p);
      init_positions(n_new, n_old);
      if (o.sd_conc_large_tail) {                                                        // init_SD_with_distros_tail.ipp:14-40
        const double log_rd_min_init = log_rd_max;
        init_dist_analysis_const_multi(d);
        log_rd_min = log_rd_min_init;
        if (log_rd_min >= log_rd_max) throw lcx_error("Distribution analysis error: rd_min >= rd_max");
        init_const_multi_like(d, 1);
      }
    }
    release_replay_keep();
  }
  void init_positions(size_t n_new, size_t n_old)
  {                                                                                      // init_xyz.ipp:40-74
    const int nn[3] = {o.nx, o.ny, o.nz};
    const T a0[3] = {T(o.x0), T(o.y0), T(o.z0)}, b1[3] = {T(o.x1), T(o.y1), T(o.z1)}, dd[3] = {T(o.dx), T(o.dy), T(o.dz)};
    T *pos[3] = {A.x.p, A.y.p, A.z.p};
    for (int ix = 0; ix < 3; ++ix) {
      if (!nn[ix]) continue;
      const u01_src<T> rs = rand_u01(n_new);
      hipLaunchKernelGGL(k_init_pos<T>, dim3(nblk(n_new)), dim3(BS), 0, st, n_new, n_old, ix, g, ijk.p, rs, a0[ix], b1[ix], dd[ix], pos[ix]);
    }
  }
  void init_SD_with_sizes()
  {                                                                                      // init_SD_with_sizes.ipp:14-77
    for (const auto &ds : sizes) {
      const n_t per_cell = n_t(ds.sd_count);
      const size_t n_old = npart, n_new = size_t(per_cell) * ncell;
      check_npart(n_old + n_new);
      npart = nphys = n_old + n_new;
      if (n_new == 0) continue;
      const T r = T(ds.radius);
      hipLaunchKernelGGL(k_init_sizes<T>, dim3(nblk(n_new)), dim3(BS), 0, st, n_new, n_old, per_cell, T(r * r * r), T(ds.kappa), T(ds.conc), dv.p, rhod.p,
                         conc_factor_h.empty() ? (const T *)nullptr : conc_factor.p, m1(o.nz), o.aerosol_independent_of_rhod, ijk.p, A.rd3.p, A.kpa.p, A.vt.p, A.n.p);
      hipLaunchKernelGGL(k_init_wet<T>, dim3(nblk(n_new)), dim3(BS), 0, st, n_new, n_old, A.rd3.p, A.kpa.p, ijk.p, RH.p, Tk.p, T(o.RH_max), A.rw2.p);
      init_positions(n_new, n_old);
    }
    release_replay_keep();
  }
  void init_kernel()
  {                                                                                      // init_kernel.ipp:6-233
    std::vector<double> params = kernel_parameters_h;
    switch (o.kernel) {
      case LCX_KERNEL_GEOMETRIC: if (n_user_params > 1) throw lcx_error("Not more than 1 parameter is required by the geometric kernel"); break;
      case LCX_KERNEL_GOLOVIN: if (n_user_params != 1) throw lcx_error("Golovin kernel accepts exactly one parameter"); break;
      case LCX_KERNEL_LONG: if (n_user_params != 0) throw lcx_error("Long kernel doesn't accept parameters"); break;
      default: {
        const bool onishi = o.kernel == LCX_KERNEL_ONISHI_HALL || o.kernel == LCX_KERNEL_ONISHI_HALL_DAVIS_NO_WAALS;
        if (onishi) {                                                                    // init_kernel.ipp:183-231
          if (n_user_params != 1) throw lcx_error("libcloudph++: Please supply one kernel parameter: Taylor microscale Reynolds number.");
          if (!o.turb_coal_switch) throw lcx_error("libcloudph++: To use the turbulent Onishis kernel, set turb_coal_switch=True");
        } else if (n_user_params != 0) throw lcx_error("this kernel doesn't accept parameters");
        const int eff = o.kernel == LCX_KERNEL_ONISHI_HALL ? LCX_KERNEL_HALL :
                        o.kernel == LCX_KERNEL_ONISHI_HALL_DAVIS_NO_WAALS ? LCX_KERNEL_HALL_DAVIS_NO_WAALS : o.kernel;
        std::vector<double> tab;
        std::string tried;
        if (!load_efficiency_table(eff, tab, kernel_r_max, tried))
          throw lcx_error("libcloudph++: collision efficiency table of kernel " + std::to_string(o.kernel) + " not found or unreadable (" + tried +
                          "); set LCX_DATA_DIR to the directory that holds kernel_eff_*.f64");
        params.insert(params.end(), tab.begin(), tab.end());                             // user parameters first, then the efficiencies
      }
    }
    std::vector<T> h(params.begin(), params.end());
    kparams.alloc(h.size());
    h2d(kparams.p, h.data(), h.size() * sizeof(T));
  }
  void sanity_init(const lcx_arrinfo_t *th_, const lcx_arrinfo_t *rv_, const lcx_arrinfo_t *rhod_, const lcx_arrinfo_t *p_,
                   const lcx_arrinfo_t *cx, const lcx_arrinfo_t *cy, const lcx_arrinfo_t *cz)
  {                                                                                      // init_sanity_check.ipp
    if (init_called) throw lcx_error("libcloudph++: init() may be called just once");
    init_called = true;
    if (is_null(th_) || is_null(rv_) || is_null(rhod_)) throw lcx_error("libcloudph++: passing th, rv and rhod is mandatory");
    courant_checks(cx, cy, cz);
    if (distros.empty() && sizes.empty()) throw lcx_error("libcloudph++: Both dry_distros and dry_sizes are undefined");
    if (n_dims > 0) {
      if (!(o.x0 >= 0 && o.x0 < m1(o.nx) * o.dx)) throw lcx_error("libcloudph++: !(x0 >= 0 & x0 < min(1,nx)*dz)");
      if (!(o.y0 >= 0 && o.y0 < m1(o.ny) * o.dy)) throw lcx_error("libcloudph++: !(y0 >= 0 & y0 < min(1,ny)*dy)");
      if (!(o.z0 >= 0 && o.z0 < m1(o.nz) * o.dz)) throw lcx_error("libcloudph++: !(z0 >= 0 & z0 < min(1,nz)*dz)");
      if (!(o.y1 > o.y0 && o.y1 <= m1(o.ny) * o.dy)) throw lcx_error("libcloudph++: !(y1 > y0 & y1 <= min(1,ny)*dy)");
      if (!(o.z1 > o.z0 && o.z1 <= m1(o.nz) * o.dz)) throw lcx_error("libcloudph++: !(z1 > z0 & z1 <= min(1,nz)*dz)");
    }
    if (o.dt == 0) throw lcx_error("libcloudph++: please specify opts_init.dt");
    if (o.sd_conc * o.sd_const_multi != 0) throw lcx_error("libcloudph++: specify either opts_init.sd_conc or opts_init.sd_const_multi, not both");
    if (o.sd_conc == 0 && o.sd_const_multi == 0 && o.n_dry_sizes == 0) throw lcx_error("libcloudph++: please specify opts_init.sd_conc, opts_init.sd_const_multi or opts_init.dry_sizes");
    if (o.coal_switch) {
      if (o.terminal_velocity == LCX_VT_UNDEFINED) throw lcx_error("libcloudph++: please specify opts_init.terminal_velocity or turn off opts_init.coal_switch");
      if (o.kernel == LCX_KERNEL_UNDEFINED) throw lcx_error("libcloudph++: please specify opts_init.kernel");
    }
    if (o.sedi_switch && o.terminal_velocity == LCX_VT_UNDEFINED) throw lcx_error("libcloudph++: please specify opts_init.terminal_velocity or turn off opts_init.sedi_switch");
    if (o.sedi_switch && o.nz == 0) throw lcx_error("libcloudph++: opts_init.sedi_switch can be True only if n_dims > 1");
    if (o.subs_switch && o.nz == 0) throw lcx_error("libcloudph++: opts_init.subs_switch can be True only if n_dims > 1");
    if (o.subs_switch && o.nz != int(w_LS_h.size())) throw lcx_error("libcloudph++: opts_init.subs_switch == True, but subsidence velocity profile size != nz");
    if (!conc_factor_h.empty() && n_dims < 2) throw lcx_error("libcloudph++: aerosol_conc_factor can only be used in 2D and 3D");
    if (!conc_factor_h.empty() && o.nz != int(conc_factor_h.size())) throw lcx_error("libcloudph++: aerosol_conc_factor size needs to be either 0 or nz");
    if (!conc_factor_h.empty() && !o.aerosol_independent_of_rhod) throw lcx_error("libcloudph++: aerosol_conc_factor can only be used if aerosol_independent_of_rhod==true");
    if (o.const_p && is_null(p_)) throw lcx_error("libcloudph++: In const_p option, pressure profile must be passed (p in init())");
    if (!o.const_p && !is_null(p_)) throw lcx_error("libcloudph++: pressure profile was passed in init(), but the constant pressure option was not used");
    if (o.sstp_cond < 1) throw lcx_error("libcloudph++: opts_init.sstp_cond needs to be greater than 0");
    if (o.turb_adve_switch && o.nz == 0) throw lcx_error("libcloudph++: opts_init.turb_adve_switch can be True only if n_dims > 1");
    if (o.turb_cond_switch && o.nz == 0) throw lcx_error("libcloudph++: opts_init.turb_cond_switch can be True only if n_dims > 1");
    if (turb() && size_t(o.nz) != SGS_mix_len_h.size()) throw lcx_error("libcloudph++: at least one of opts_init.turb_adve_switch, opts_init.turb_cond_switch is true, but SGS mixing length profile size != nz");
    for (double v : SGS_mix_len_h) if (v <= 0) throw lcx_error("libcloudph++: SGS_mix_len <= 0");
    if (o.adaptive_sstp_cond && !o.exact_sstp_cond) throw lcx_error("libcloudph++: Adaptive condensation substepping (opts_init.adaptive_sstp_cond) works oly for per-particle substepping (opts_init.exact_sstp_cond)");
    if (!o.sstp_cond_mix && !o.exact_sstp_cond) throw lcx_error("libcloudph++: Mixing of rv and th (opts_init.sstp_cond_mix) can only be disable for per-particle substepping (opts_init.exact_sstp_cond)");
    if (o.sstp_cond_mix && o.adaptive_sstp_cond && o.exact_sstp_cond) throw lcx_error("libcloudph++: Adaptive cond substepping (opts_init.adaptive_sstp_cond) with per-particle substepping (opts_init.exact_sstp_cond) requires mixing of th and rv between subteps (opts_init.sstp_cond_mix) to be disabled");
    if (o.sstp_cond_act > 1 && (o.sstp_cond_mix || !o.exact_sstp_cond || !o.adaptive_sstp_cond)) throw lcx_error("libcloudph++: number of substeps for activation (opts_init.sstp_cond_act) can be greater than 1 only if mixing of rv and th (opts_init.sstp_cond_mix) is disabled and if per-particle condensation substepping is used (opts_init.exact_sstp_cond) and if adaptive substepping is used (opts_init.adaptive_sstp_cond)");
  }
  void courant_checks(const lcx_arrinfo_t *cx, const lcx_arrinfo_t *cy, const lcx_arrinfo_t *cz) const
  {
    if (!is_null(cx) || !is_null(cy) || !is_null(cz)) {
      if (n_dims == 0) throw lcx_error("libcloudph++: Courant numbers passed in 0D setup");
      if (n_dims == 1 && (is_null(cx) || !is_null(cy) || !is_null(cz))) throw lcx_error("libcloudph++: Only X Courant number allowed in 1D setup");
      if (n_dims == 2 && (is_null(cx) || !is_null(cy) || is_null(cz))) throw lcx_error("libcloudph++: Only X and Z Courant numbers allowed in 2D setup");
      if (n_dims == 3 && (is_null(cx) || is_null(cy) || is_null(cz))) throw lcx_error("libcloudph++: All XYZ Courant number components required in 3D setup");
    }
  }
  void init(const lcx_arrinfo_t *th_, const lcx_arrinfo_t *rv_, const lcx_arrinfo_t *rhod_, const lcx_arrinfo_t *p_,
            const lcx_arrinfo_t *cx, const lcx_arrinfo_t *cy, const lcx_arrinfo_t *cz) override
  {
    sanity_init(th_, rv_, rhod_, p_, cx, cy, cz);
    in_init = true;
    struct Leave { bool &f; ~Leave() { f = false; } } leave{in_init};
    const int nxh = o.nx + 2 * halo;
    switch (n_dims) {                                                                    // init_sync.ipp:28-44, particles_impl.ipp:413-431
      case 3: n_cx = size_t(nxh + 1) * o.ny * o.nz; n_cy = size_t(nxh) * (o.ny + 1) * o.nz; n_cz = size_t(nxh) * o.ny * (o.nz + 1); break;
      case 2: n_cx = size_t(nxh + 1) * o.nz; n_cz = size_t(nxh) * (o.nz + 1); break;
      case 1: n_cx = size_t(nxh) + 1; break;
      default: break;
    }
    courant_x.alloc_zero(n_cx, st); courant_y.alloc_zero(n_cy, st); courant_z.alloc_zero(n_cz, st);
    stage_begin();
    if (!w_LS_h.empty()) { std::vector<T> h(w_LS_h.begin(), w_LS_h.end()); w_LS.alloc(h.size()); h2d(w_LS.p, h.data(), h.size() * sizeof(T)); }
    if (!conc_factor_h.empty()) { std::vector<T> h(conc_factor_h.begin(), conc_factor_h.end()); conc_factor.alloc(h.size()); h2d(conc_factor.p, h.data(), h.size() * sizeof(T)); }
    sync_in_arr(th_, th, ncell, 0, 0, 0); sync_in_arr(rv_, rv, ncell, 0, 0, 0); sync_in_arr(rhod_, rhod, ncell, 0, 0, 0);
    sync_in_arr(p_, p, ncell, 0, 0, 0);
    sync_in_arr(cx, courant_x, n_cx, 1, 0, 0, halo); sync_in_arr(cy, courant_y, n_cy, 0, 1, 0, halo); sync_in_arr(cz, courant_z, n_cz, 0, 0, 1, halo);
    flush_sync_jobs(); flush_host_in();
    if (n_dims > 0)
      hipLaunchKernelGGL(k_init_dv<T>, dim3(nblk(ncell)), dim3(BS), 0, st, ncell, dv.p, m1(o.ny), m1(o.nz), T(o.dx), T(o.dy), T(o.dz),
                         T(o.x0), T(o.y0), T(o.z0), T(o.x1), T(o.y1), T(o.z1));
    hskpng_Tpr();
    if (!o.no_ccn_at_init && !distros.empty()) init_SD_with_distros();
    if (!o.no_ccn_at_init && !sizes.empty()) init_SD_with_sizes();
    // (a slab with neighbours takes in what they send: their caller may have set other values there)
    kpa_uniform = !o.no_ccn_at_init && int(distros.size()) + n_size_keys == 1 && !distmem() && !dbg(LCX_DBG_KPA_ARRAY);
    kpa_value = distros.size() == 1 ? T(distros[0].kappa) : !sizes.empty() ? T(sizes[0].kappa) : T(0);
    if (o.coal_switch) init_kernel();
    if (o.terminal_velocity == LCX_VT_BEARD77FAST) {
      vt_0.alloc(size_t(vtc.n_bin));
      hipLaunchKernelGGL(k_init_vt0<T>, dim3(nblk(size_t(vtc.n_bin))), dim3(BS), 0, st, vt_0.p, vtc);
    }
    if (ix_tag >= 0 && nphys) hipLaunchKernelGGL(k_fill_index<T>, dim3(nblk(nphys)), dim3(BS), 0, st, A.ext[ix_tag].p, nphys);
    hskpng_vterm(true);
    hskpng_approximate_rc2_invalid();                                                    // particles_init.ipp:116-117
    sstp_save();
    sorted = false; sort_deferred = false;
    hskpng_count();
    if (!B.n.p) alloc_attrs(B);      // the compaction target: allocated here, not inside the first step that compacts (GBs of hipMalloc)
    sync();
  }

  // ------------------------------------------------------------------------------------------
  // time stepping (particles_step.ipp)
  // ------------------------------------------------------------------------------------------
  void sync_in(const lcx_arrinfo_t *th_, const lcx_arrinfo_t *rv_, const lcx_arrinfo_t *rhod_, const lcx_arrinfo_t *cx,
               const lcx_arrinfo_t *cy, const lcx_arrinfo_t *cz, const lcx_arrinfo_t *diss) override
  {
    if (!init_called) throw lcx_error("libcloudph++: please call init() before calling step_sync()");
    if (should_now_run_async) throw lcx_error("libcloudph++: please call step_async() before calling step_sync() again");
    if (is_null(th_) || is_null(rv_)) throw lcx_error("libcloudph++: passing th and rv is mandatory");
    courant_checks(cx, cy, cz);
    if (turb_any() && is_null(diss)) throw lcx_error("libcloudph++: turbulent advection, coalescence and condesation are not switched off and diss_rate is empty");
    if (!turb_any() && !is_null(diss)) throw lcx_error("libcloudph++: turbulent advection, coalescence and condesation are switched off and diss_rate is not empty");
    Range r(this, "sync_in");
    var_rho = !is_null(rhod_);
    stage_begin();
    sync_in_arr(th_, th, ncell, 0, 0, 0); sync_in_arr(rv_, rv, ncell, 0, 0, 0); sync_in_arr(rhod_, rhod, ncell, 0, 0, 0);
    if (turb_any()) sync_in_arr(diss, diss_rate, ncell, 0, 0, 0);
    if (!courants_late) { sync_in_arr(cx, courant_x, n_cx, 1, 0, 0, halo); sync_in_arr(cy, courant_y, n_cy, 0, 1, 0, halo); sync_in_arr(cz, courant_z, n_cz, 0, 0, 1, halo); }
    flush_sync_jobs(); flush_host_in();
    if (o.adve_scheme == LCX_ADVE_PRED_CORR && !is_null(cx) && n_cx) {                  // particles_step.ipp:127-142
      HIPCHK(hipMemsetAsync(d_flag.p, 0, sizeof(int), st));
      hipLaunchKernelGGL(k_flag_outside<T>, dim3(nblk(n_cx)), dim3(BS), 0, st, courant_x.p, n_cx, T(-2.), T(2.), d_flag.p);
      int flag = 0;
      read_back(&flag, d_flag.p, 1);
      if (flag) { adve_scheme = LCX_ADVE_EULER; HIPCHK(hipMemsetAsync(d_flag.p, 0, sizeof(int), st)); }
    }
    should_now_run_cond = true;
  }
  void step_cond(const lcx_opts_t &opts, const lcx_arrinfo_t *th_, const lcx_arrinfo_t *rv_) override
  {
    if (!should_now_run_cond) throw lcx_error("libcloudph++: please call sync_in() before calling step_cond()");
    if (opts.turb_cond && !o.turb_cond_switch) throw lcx_error("libcloudph++: turb_cond_swtich=False, but turb_cond==True");
    should_now_run_cond = false;
    adjust_timesteps(opts.dt);
    if (ix_ict >= 0 && nphys)                                                            // update_incloud_time, particles_step.ipp:180-181
      hipLaunchKernelGGL(k_incloud_time<T>, dim3(nblk(nphys)), dim3(BS), 0, st, nphys, ijk.p, A.rd3.p, A.kpa.p, A.rw2.p, Tk.p, T(dt), A.ext[ix_ict].p);
    if (opts.cond) {
      // (a measurement switch that names one of the per-substep kernels gets that kernel)
      constexpr unsigned per_substep_variants = LCX_DBG_COND_NO_FUSED_SUBSTEPS | LCX_DBG_COND_FOLD | LCX_DBG_COND_WQ | LCX_DBG_COND_BUDGET | LCX_DBG_COND_PROBE |
                                                LCX_DBG_COND_LEAN_R3 | LCX_DBG_FINISH_STAGED | LCX_DBG_COND_NO_FOLD;
      const bool fused_substeps = sstp_cond > 1 && lean_storage_cond() && !opts.turb_cond && npart && !(o.dbg_flags & per_substep_variants);
      if (fused_substeps || !(sort_deferred && lean_storage_cond() && !opts.turb_cond && !(o.exact_sstp_cond && (sstp_cond > 1 || sstp_cond_act > 1)))) hskpng_sort();
      if (o.exact_sstp_cond && (sstp_cond > 1 || sstp_cond_act > 1)) { hskpng_mfp(); cond_perparticle(opts.RH_max, opts.turb_cond); }
      else if (fused_substeps) cond_substeps_fused(opts.RH_max);
      else for (int step = 0; step < sstp_cond; ++step) {
        // (the Eulerian fields' substep rides on the cell pass of cond_substep: sstp_fused)
        if (opts.turb_cond && nphys)                                                     // apply_perparticle_sgs_supersat.ipp
          hipLaunchKernelGGL(k_sgs_supersat<T>, dim3(nblk(nphys)), dim3(BS), 0, st, nphys, T(T(dt) / sstp_cond), A.ext[ix_dot_ssp].p, A.ext[ix_ssp].p);
        cond_substep(opts.RH_max, step, opts.turb_cond);                                 // (hskpng_mfp at substep 0 and hskpng_Tpr inside)
      }
      sstp_save();
      { Range r(this, "sync_out"); sync_out_arr(th, th_, ncell); sync_out_arr(rv, rv_, ncell); flush_sync_jobs(); }
    }
    late_courants();               // (host rows and copies while the kernels queued above run)
    // opts_init.stream_ordered: nothing of this call touches host memory -- the results are ordered on the stream, the host goes on
    const bool no_wait = o.stream_ordered && out_jobs.empty() && !hstage_busy && (is_null(th_) || th_->on_device) && (is_null(rv_) || rv_->on_device);
    finish_sync_out();
    if (!no_wait) sync_results_only();
    should_now_run_async = true;
    selected_before_counting = false;
  }
  void step_async(const lcx_opts_t &opts) override
  {
    if (!should_now_run_async) throw lcx_error("libcloudph++: please call step_sync() before calling step_async() again");
    should_now_run_async = false;
    if (opts.chem_dsl || opts.chem_dsc || opts.chem_rct) throw lcx_error("libcloudph++: all chemistry was switched off in opts_init");
    if (opts.coal && !o.coal_switch) throw lcx_error("libcloudph++: coalescence was switched off in opts_init");
    if (opts.sedi && !o.sedi_switch) throw lcx_error("libcloudph++: sedimentation was switched off in opts_init");
    if (opts.subs && !o.subs_switch) throw lcx_error("libcloudph++: subsidence was switched off in opts_init");
    if (opts.turb_adve && !o.turb_adve_switch) throw lcx_error("libcloudph++: turb_adve_switch=False, but turb_adve==True");
    if (opts.turb_coal && !o.turb_coal_switch) throw lcx_error("libcloudph++: turb_coal_switch=False, but turb_coal==True");   // the reference reads an empty diss_rate here
    if (opts.turb_adve && n_dims == 0) throw lcx_error("libcloudph++: turbulent advection does not work in 0D");
    if (opts.src) throw lcx_error("libcloudph++: aerosol source was switched off in opts_init");
    if (opts.rlx) throw lcx_error("libcloudph++: aerosol relaxation was switched off in opts_init");
    adjust_timesteps(opts.dt);
    rng_recs.clear();
    last_async_coal = opts.coal != 0;
    coal_marks_dead = n_dims > 0 && nphys > 0 && !opts.rcyc && sstp_coal == 1;   // (= the fused move below; with coalescence
                                                       // substeps a used-up SD still takes part in the later ones and keeps its cell)
    hskpng_Tpr(opts.sedi || opts.coal || opts.cond);
    if (opts.sedi || opts.coal || opts.cond) hskpng_vterm(false);
    if (opts.coal) {
      for (int step = 0; step < sstp_coal; ++step) {
        coal(dt / sstp_coal, opts.turb_coal);
        // (hskpng_vterm_invalid between the substeps: left to the next substep's ranking, which every droplet passes anyway)
        if (step + 1 != sstp_coal) { if (dbg(LCX_DBG_VTERM_INVALID_OWN_PASS)) hskpng_vterm(true); else vt_fix_pending = true; }
      }
      if (pure_const_multi) {
        int flag = 0;
        read_back(&flag, d_flag.p, 1);
        if (flag) { ++sstp_coal; HIPCHK(hipMemsetAsync(d_flag.p, 0, sizeof(int), st)); }
      }
      release_replay_keep();
      hskpng_approximate_rc2_invalid();                                                  // particles_step.ipp:402-403
    }
    // single device, > 0 dimensions: advection + sedimentation + boundary + re-indexing in ONE pass over the positions
    if ((opts.turb_adve || opts.turb_cond) && nphys) sgs_turbulence(opts);              // particles_step.ipp:406-427
    turb_adve_now = opts.turb_adve;
    // > 0 dimensions: advection + sedimentation + boundary + re-indexing in ONE pass over the positions; with a decomposed
    // domain the histogram is completed by the immigrants in migrate_finish
    const bool fused = n_dims > 0 && nphys > 0 && !opts.rcyc;
    move(opts.adve, opts.sedi, opts.subs, true, fused);
    adve_scheme = o.adve_scheme;
    fused_pending = fused && distmem();
    n_before_unpack = nphys;
    if (fused && !distmem()) post_copy_after_fused_move(opts);
    else if (!distmem()) post_copy(opts);
    // no host synchronisation here: step_async hands no array back, everything later is ordered on the object's stream and
    // every call that returns data to the host synchronises itself -- the caller's next step_sync is queued while this one runs
    selected_before_counting = false;
  }

  // ------------------------------------------------------------------------------------------
  // diagnostics (particles_diag.ipp, moms.ipp, fill_outbuf.ipp)
  // ------------------------------------------------------------------------------------------
  void diag_cell(int which) override
  {
    hskpng_Tpr();
    const T *src = which == 0 ? p.p : which == 1 ? Tk.p : RH.p;
    HIPCHK(hipMemcpyAsync(count_mom.p, src, ncell * sizeof(T), hipMemcpyDeviceToDevice, st));
    sync();
  }
  void diag_vel_div() override
  {
    if (n_dims == 0) return;
    hipLaunchKernelGGL(k_vel_div<T>, dim3(nblk(ncell)), dim3(BS), 0, st, ncell, g, halo, T(o.dt), courant_x.p, courant_y.p, courant_z.p, count_mom.p);
    sync();
  }
  void need_nfiltered() { if (!n_filtered.p) n_filtered.alloc(cap); }
  T *attr_ptr(int attr)
  {
    if (attr == 4) {
      if (ix_ict < 0) throw lcx_error("libcloudph++: diag_incloud_time_mom called, but opts_init.diag_incloud_time==false");
      return A.ext[ix_ict].p;
    }
    if (attr >= 5 && attr <= 7) {                                   // up, vp, wp
      const int ix = attr == 5 ? ix_up : attr == 6 ? ix_vp : ix_wp;
      if (ix < 0) throw lcx_error("libcloudph++: moment of an SGS velocity perturbation that this set-up does not carry (turb_adve_switch / turb_cond_switch, dimensions)");
      return A.ext[ix].p;
    }
    return attr == 0 ? A.rd3.p : attr == 1 ? A.rw2.p : attr == 2 ? A.kpa.p : A.vt.p;
  }
  void diag_select(int mode, int cons, int attr, double a, double b) override
  {
    hskpng_sort();
    need_nfiltered();
    if (cons && !selected_before_counting) throw lcx_error("libcloudph++: consecutive selection without a previous selection");
    if (nphys)
      hipLaunchKernelGGL(k_nfilt<T>, dim3(nblk(nphys)), dim3(BS), 0, st, nphys, mode, cons, A.n.p, attr_ptr(attr), T(a), T(b), n_filtered.p);
    selected_before_counting = true;
    sync();
  }
  void moms_sum(const T *vec, T power, int kind, bool specific, const T *vec2 = nullptr)
  {
    if (!selected_before_counting) throw lcx_error("libcloudph++: please select super-droplets (diag_all / diag_*_rng) before counting moments");
    hskpng_sort();
    if (npart)
      hipLaunchKernelGGL(k_mom_vals<T>, dim3(nblk(npart)), dim3(BS), 0, st, npart, sid(), n_filtered.p, vec, vec2, power, kind, m3_after.p);
    hipLaunchKernelGGL(k_cell_seqsum<T>, dim3(nblk(ncell, cf_cells())), dim3(BS), 0, st, ncell, cf_cells(), cell_start.p, m3_after.p, dv.p, rhod.p,
                       int(specific && n_dims > 0), count_mom.p);
    sync();
  }
  void diag_act(int which) override
  {                                                                                      // particles_diag.ipp:350-407
    hskpng_sort();
    need_nfiltered();
    if (nphys)
      hipLaunchKernelGGL(k_nfilt_act<T>, dim3(nblk(nphys)), dim3(BS), 0, st, nphys, which, A.n.p, A.rd3.p, A.kpa.p, A.rw2.p, ijk.p, Tk.p, RH.p, n_filtered.p);
    selected_before_counting = true;
    sync();
  }
  void diag_wet_mass_dens(double rad, double sig0) override
  {                                                                                      // mass_dens.ipp:36-118
    if (!selected_before_counting) throw lcx_error("libcloudph++: please select super-droplets (diag_all / diag_*_rng) before counting moments");
    hskpng_sort();
    if (npart)
      hipLaunchKernelGGL(k_massdens_vals<T>, dim3(nblk(npart)), dim3(BS), 0, st, npart, sid(), sijk(), cell_start.p, n_filtered.p, A.rw2.p,
                         T(rad), T(sig0), m3_after.p);
    hipLaunchKernelGGL(k_cell_seqsum<T>, dim3(nblk(ncell, cf_cells())), dim3(BS), 0, st, ncell, cf_cells(), cell_start.p, m3_after.p, dv.p, rhod.p, 0, count_mom.p);
    hipLaunchKernelGGL(k_massdens_scale<T>, dim3(nblk(ncell)), dim3(BS), 0, st, ncell, cell_start.p, dv.p,
                       T(T(4. / 3.) * cst<T>::rho_w * T(std::sqrt(M_PI / 2.))), count_mom.p);
    sync();
  }
  void diag_sd_conc() override { moms_sum(A.rw2.p, T(0), 1, false); }
  void diag_mom(int attr, double power) override { moms_sum(attr_ptr(attr), T(power), 0, true); }
  void diag_precip_rate() override
  {                                                                                      // particles_diag.ipp:529-547
    hskpng_vterm(false);
    moms_sum(A.rw2.p, T(1), 2, false, A.vt.p);
  }
  void diag_max_rw() override
  {
    hskpng_sort();
    hipLaunchKernelGGL(k_cell_max<T>, dim3(nblk(ncell)), dim3(BS), 0, st, ncell, cell_start.p, sid(), A.rw2.p, count_mom.p);
    sync();
  }
  void outbuf(const void **data, size_t *n) override
  {
    HIPCHK(hipMemcpyAsync(outbuf_h.data(), count_mom.p, ncell * sizeof(T), hipMemcpyDeviceToHost, st));
    hskpng_count();                                                                      // particles_ctor.ipp:83-92
    sync();
    *data = outbuf_h.data(); *n = ncell;
  }
  void get_attr(const char *name, void *out, size_t capn, size_t *n) override
  {
    ensure_compact();
    const std::string s(name);
    const T *v = s == "rw2" ? A.rw2.p : s == "rd3" ? A.rd3.p : s == "kappa" ? A.kpa.p : s == "x" ? A.x.p : s == "y" ? A.y.p : s == "z" ? A.z.p : nullptr;
    if (s != "rw2" && s != "rd3" && s != "kappa" && s != "x" && s != "y" && s != "z") throw lcx_error("Unknown attribute name passed to get_attr.");
    *n = npart;
    if (!out) return;
    if (capn < npart) throw lcx_error("get_attr: buffer too small");
    if (v && npart) { HIPCHK(hipMemcpyAsync(out, v, npart * sizeof(T), hipMemcpyDeviceToHost, st)); sync(); }
    else if (npart) memset(out, 0, npart * sizeof(T));
  }
  void diag_puddle(double *out) override
  {
    double s4[4];
    read_back(s4, puddle_acc.p, 4);
    puddle[LCX_OUT_LIQ_VOL] = s4[0]; puddle[LCX_OUT_DRY_VOL] = s4[1]; puddle[LCX_OUT_LIQ_NUM] = s4[2]; puddle[LCX_OUT_PRTCL_NUM] = s4[3];
    for (int i = 0; i < LCX_OUT_COUNT; ++i) out[i] = puddle[i];
  }
  size_t n_part() override { return npart; }
  size_t n_cell() override { return ncell; }

  // ------------------------------------------------------------------------------------------
  // introspection hooks
  // ------------------------------------------------------------------------------------------
  template <class S> std::vector<S> d2h(const S *p_, size_t n) { std::vector<S> h(n); if (n) { HIPCHK(hipMemcpyAsync(h.data(), p_, n * sizeof(S), hipMemcpyDeviceToHost, st)); sync(); } return h; }
  void get_state_u64(const char *name, unsigned long long *out, size_t capn, size_t *n) override
  {
    const std::string s(name);
    std::vector<unsigned long long> v;
    // "raw_*": the storage as it is (whole extent, dead slots included), nothing compacted or sorted on the way
    const bool raw = s.rfind("raw_", 0) == 0;
    if (!raw) ensure_compact();
    if (s == "raw_collided") {
      DevBuf<unsigned long long> cnt; cnt.alloc_zero(64 * 8, st);
      if (nphys) hipLaunchKernelGGL(k_count_collided<T>, dim3(nblk(nphys)), dim3(BS), 0, st, nphys, A.vt.p, ijk.p, cnt.p);
      auto h = d2h(cnt.p, 64 * 8);
      unsigned long long tot = 0; for (auto x : h) tot += x;
      v.assign(1, tot);
    }
    else if (s == "raw_launches") {                // kernel launches and host waits of the library in this process so far (the wq kernel's own launches not counted)
      v = {g_launches.load(), g_host_waits.load()};
    }
    else if (s == "raw_mode") {                    // what this object runs: strict_fp, cond_solver, the kernel of its last condensation launch (enum lcx_cond_kernel), dbg_flags
      v = {(unsigned long long)(o.strict_fp ? 1 : 0), (unsigned long long)o.cond_solver, (unsigned long long)last_cond_kernel, (unsigned long long)o.dbg_flags};
    }
    else if (s == "raw_cond_listed") {             // droplets that the last condensation substep handed to the reference's iterates (cond_list)
      unsigned long long c = 0;
      if (defer_cnt.p && defer_cnt.n >= size_t(DEFER_SHARDS * DEFER_CNT_STRIDE) && o.cond_solver == 0) {
        auto h = d2h(defer_cnt.p, size_t(DEFER_SHARDS * DEFER_CNT_STRIDE));
        for (int sh = 0; sh < DEFER_SHARDS; ++sh) c += h[size_t(sh) * DEFER_CNT_STRIDE];
      }
      v.assign(1, c);
    }
    else if (s == "raw_cond_resumed") {            // droplets whose loop the last condensation substep's first pass left to k_cond_lean_resume (its budget)
      unsigned long long c = 0;
      if (defer_cnt.p && defer_cnt.n >= size_t(2 * DEFER_SHARDS * DEFER_CNT_STRIDE) && o.cond_solver == 0) {
        auto h = d2h(defer_cnt.p + DEFER_SHARDS * DEFER_CNT_STRIDE, size_t(DEFER_SHARDS * DEFER_CNT_STRIDE));
        for (int sh = 0; sh < DEFER_SHARDS; ++sh) c += h[size_t(sh) * DEFER_CNT_STRIDE];
      }
      v.assign(1, c);
    }
    else if (s == "raw_n") { auto h = d2h(A.n.p, nphys); v.assign(h.begin(), h.end()); }
    else if (s == "raw_ijk") { auto h = d2h(ijk.p, nphys); v.assign(h.begin(), h.end()); }
    else if (s == "raw_sorted_id") {               // the cell-sorted order as the last sort left it (the shuffled order of the next coalescence, mostly)
      if (sort_deferred || !sorted) throw lcx_error("raw_sorted_id: the re-sort of the last step has not been finished yet");
      auto h = d2h(sid(), npart); v.assign(h.begin(), h.end());
    }
    else if (s == "n") { auto h = d2h(A.n.p, npart); v.assign(h.begin(), h.end()); }
    else if (s == "ijk") { auto h = d2h(ijk.p, npart); v.assign(h.begin(), h.end()); }
    else if (s == "sorted_id") {
      hskpng_sort();
      if (sorted_shuffled && shuffle_fresh) order_cells(false);   // (production order: post_copy has pre-shuffled for the next coalescence;
                                                                  //  the getter shows the reference's state at this point, ids ascending inside a cell)
      auto h = d2h(sid(), npart); v.assign(h.begin(), h.end());
    }
    else if (s == "sorted_ijk") { hskpng_sort(); auto h = d2h(sijk(), npart); v.assign(h.begin(), h.end()); }
    else if (s == "cell_start") { hskpng_sort(); auto h = d2h(cell_start.p, ncell + 1); v.assign(h.begin(), h.end()); }
    else if (s == "count_ijk" || s == "count_num") {
      hskpng_sort();
      auto h = d2h(cell_start.p, ncell + 1);
      for (size_t c = 0; c < ncell; ++c) if (h[c + 1] > h[c]) v.push_back(s == "count_ijk" ? c : (unsigned long long)(h[c + 1] - h[c]));
    } else throw lcx_error("unknown u64 state '" + s + "'");
    *n = v.size();
    if (!out) return;
    if (capn < v.size()) throw lcx_error("buffer too small");
    std::copy(v.begin(), v.end(), out);
  }
  void get_state_real(const char *name, double *out, size_t capn, size_t *n) override
  {
    const std::string s(name);
    if (s.rfind("raw_", 0) != 0) ensure_compact();
    struct E { const char *nm; const T *p; size_t len; };
    const E tab[] = {{"raw_rw2", A.rw2.p, nphys}, {"raw_rd3", A.rd3.p, nphys}, {"raw_kappa", A.kpa.p, nphys}, {"raw_vt", A.vt.p, nphys},
      {"raw_x", A.x.p, A.x.p ? nphys : 0}, {"raw_y", A.y.p, A.y.p ? nphys : 0}, {"raw_z", A.z.p, A.z.p ? nphys : 0},
      {"raw_tag", ix_tag >= 0 ? A.ext[ix_tag].p : nullptr, ix_tag >= 0 ? nphys : 0}, {"tag", ix_tag >= 0 ? A.ext[ix_tag].p : nullptr, ix_tag >= 0 ? npart : 0},
      {"vt", A.vt.p, npart}, {"T", Tk.p, ncell}, {"p", p.p, ncell}, {"RH", RH.p, ncell}, {"eta", eta.p, ncell}, {"th", th.p, ncell},
      {"rv", rv.p, ncell}, {"rhod", rhod.p, ncell}, {"dv", dv.p, ncell}, {"lambda_D", lambda_D.p, ncell}, {"lambda_K", lambda_K.p, ncell},
      {"courant_x", courant_x.p, n_cx}, {"courant_y", courant_y.p, n_cy}, {"courant_z", courant_z.p, n_cz},
      {"vt_0", vt_0.p, vt_0.p ? size_t(vtc.n_bin) : 0}, {"count_mom", count_mom.p, ncell}, {"col", col.p, col.p ? npart : 0},
      {"rw2", A.rw2.p, npart}, {"rd3", A.rd3.p, npart}, {"kappa", A.kpa.p, npart},
      {"x", A.x.p, A.x.p ? npart : 0}, {"y", A.y.p, A.y.p ? npart : 0}, {"z", A.z.p, A.z.p ? npart : 0},
      {"sstp_tmp_rv", exact ? A.ext[ix_rv].p : sstp_tmp_rv.p, exact ? npart : ncell},
      {"sstp_tmp_th", exact ? A.ext[ix_th].p : sstp_tmp_th.p, exact ? npart : ncell},
      {"sstp_tmp_rh", exact ? A.ext[ix_rh].p : sstp_tmp_rh.p, exact ? npart : ncell},
      {"sstp_tmp_p", exact && o.const_p ? A.ext[ix_p].p : nullptr, exact && o.const_p ? npart : 0},
      {"rc2", use_rc2 ? A.ext[ix_rc2].p : nullptr, use_rc2 ? npart : 0},
      {"up", ix_up >= 0 ? A.ext[ix_up].p : nullptr, ix_up >= 0 ? npart : 0}, {"vp", ix_vp >= 0 ? A.ext[ix_vp].p : nullptr, ix_vp >= 0 ? npart : 0},
      {"wp", ix_wp >= 0 ? A.ext[ix_wp].p : nullptr, ix_wp >= 0 ? npart : 0}, {"ssp", ix_ssp >= 0 ? A.ext[ix_ssp].p : nullptr, ix_ssp >= 0 ? npart : 0},
      {"incloud_time", ix_ict >= 0 ? A.ext[ix_ict].p : nullptr, ix_ict >= 0 ? npart : 0},
      {"dot_ssp", ix_dot_ssp >= 0 ? A.ext[ix_dot_ssp].p : nullptr, ix_dot_ssp >= 0 ? npart : 0},
      {"diss_rate", diss_rate.p, turb_any() ? ncell : 0}};
    for (const E &e : tab)
      if (s == e.nm) {
        *n = e.len;
        if (!out) return;
        if (capn < e.len) throw lcx_error("buffer too small");
        auto h = d2h(e.p, e.len);
        for (size_t i = 0; i < e.len; ++i) out[i] = double(h[i]);
        return;
      }
    throw lcx_error("unknown real state '" + s + "'");
  }
  void set_particles(size_t n, const unsigned long long *mult, const double *rd3_, const double *rw2_, const double *kpa_, const double *vt_,
                     const double *x_, const double *y_, const double *z_) override
  {
    check_npart(n);
    npart = nphys = n;
    zero_n_unmarked = true;
    kpa_uniform = false;                 // (the caller's own values)
    auto up = [&](DevBuf<T> &b, const double *src) {
      if (!src || !b.p || !n) return;
      std::vector<T> h(n); for (size_t i = 0; i < n; ++i) h[i] = T(src[i]);
      h2d(b.p, h.data(), n * sizeof(T));
    };
    h2d(A.n.p, mult, n * sizeof(n_t));
    up(A.rd3, rd3_); up(A.rw2, rw2_); up(A.kpa, kpa_); up(A.vt, vt_); up(A.x, x_); up(A.y, y_); up(A.z, z_);
    if (n) HIPCHK(hipMemsetAsync(ijk.p, 0, n * sizeof(uint32_t), st));     // every SD is in the order again, n == 0 included
    hskpng_ijk();
    for (int ix : {ix_up, ix_vp, ix_wp, ix_ssp, ix_dot_ssp, ix_ict}) if (ix >= 0 && n) HIPCHK(hipMemsetAsync(A.ext[ix].p, 0, n * sizeof(T), st));
    if (ix_tag >= 0 && n) hipLaunchKernelGGL(k_fill_index<T>, dim3(nblk(n)), dim3(BS), 0, st, A.ext[ix_tag].p, n);
    if (use_rc2 && n) { hipLaunchKernelGGL(k_fill<T>, dim3(nblk(n)), dim3(BS), 0, st, A.ext[ix_rc2].p, n, T(-1)); hskpng_approximate_rc2_invalid(); }
    sstp_save();
    hskpng_count();
    sync();
  }
  void stage(const char *name, const lcx_opts_t *opts) override
  {
    const std::string s(name);
    coal_marks_dead = false; zero_n_unmarked = true;                   // (single stages: nothing is fused)
    if (s == "hskpng_Tpr") hskpng_Tpr();
    else if (s == "hskpng_mfp") hskpng_mfp();
    else if (s == "hskpng_ijk") hskpng_ijk();
    else if (s == "hskpng_sort") hskpng_sort();
    else if (s == "hskpng_shuffle_and_sort") hskpng_sort_helper(true);
    else if (s == "hskpng_count") hskpng_count();
    else if (s == "hskpng_vterm_all") hskpng_vterm(false);
    else if (s == "hskpng_vterm_invalid") hskpng_vterm(true);
    else if (s == "coal") { adjust_timesteps(opts ? opts->dt : -1); coal(dt / sstp_coal, opts && opts->turb_coal); }
    else if (s == "adve") move(true, false, false, false);
    else if (s == "sedi") { adjust_timesteps(opts ? opts->dt : -1); move(false, true, false, false); }
    else if (s == "bcnd") move(false, false, false, true);
    else if (s == "post_copy") { lcx_opts_t od; lcx_opts_default(&od); post_copy(opts ? *opts : od); }
    else throw lcx_error("unknown stage '" + s + "'");
    sync();
    release_replay_keep();
  }

  // ------------------------------------------------------------------------------------------
  // 1-D decomposition primitives
  // ------------------------------------------------------------------------------------------
  void migrate_counts(size_t *l, size_t *r) override { *l = lft_count; *r = rgt_count; }
  size_t migrate_record_bytes() override { return sizeof(n_t) + sizeof(T) * (4 + size_t(n_dims) + size_t(n_ext)); }
  void migrate_pack(int side, double x_rmt, void *buf, size_t capb) override
  {
    const size_t cnt = side == 0 ? lft_count : rgt_count;
    if (capb < cnt * migrate_record_bytes()) throw lcx_error("migrate_pack: buffer too small");
    if (!cnt) return;
    Range r(this, "migrate_pack");
    n_t *nb = (n_t *)buf; T *rb = (T *)((n_t *)buf + cnt);
    hipLaunchKernelGGL(k_pack<T>, dim3(nblk(cnt)), dim3(BS), 0, st, cnt, mig_ids[side].p, aset(A), g, T(x_rmt), T(side == 0 ? o.x0 : o.x1), nb, rb);
    sync();
  }
  // emigrants leave (n = 0) once they are packed: the protocol is pack (both sides) -> unpack -> finish
  // emigrants leave (n = 0) once they are packed; their slots are offered to the immigrants of the same step (free_n / free_used)
  size_t free_n[2] = {0, 0}, free_used = 0, reused_total = 0;
  bool strict_order() const { return o.reorder_every < 0 || eager_compact || replay_used; }
  void flag_emigrants()
  {
    if (lft_count || rgt_count) {
      free_n[0] = strict_order() ? 0 : lft_count; free_n[1] = strict_order() ? 0 : rgt_count; free_used = 0;
    }
    if (lft_count) hipLaunchKernelGGL(k_flag_ids, dim3(nblk(lft_count)), dim3(BS), 0, st, lft_count, mig_ids[0].p, A.n.p);
    if (rgt_count) hipLaunchKernelGGL(k_flag_ids, dim3(nblk(rgt_count)), dim3(BS), 0, st, rgt_count, mig_ids[1].p, A.n.p);
    lft_count = rgt_count = 0;
  }
  void migrate_unpack(const void *buf, size_t cnt) override
  {
    if (!cnt) return;
    flag_emigrants();
    const size_t n_free = free_n[0] + free_n[1];
    size_t reuse = n_free > free_used ? std::min(cnt, n_free - free_used) : 0;
    if (nphys + (cnt - reuse) > cap) {                  // make room: compaction re-indexes everything, the fused histogram is void
      ensure_compact_forced();
      fused_pending = false;
      free_n[0] = free_n[1] = 0; free_used = 0; reuse = 0;
    }
    check_npart(nphys + (cnt - reuse));
    Range r(this, "migrate_unpack");
    const n_t *nb = (const n_t *)buf; const T *rb = (const T *)((const n_t *)buf + cnt);
    hipLaunchKernelGGL(k_unpack<T>, dim3(nblk(cnt)), dim3(BS), 0, st, cnt, nphys, aset(A), g, nb, rb, T(o.x0), T(o.x1), T(5e-4),
                       mig_ids[0].p, uint32_t(free_n[0]), mig_ids[1].p, uint32_t(free_n[1]), uint32_t(free_used),
                       ijk.p, fused_pending ? cell_cnt.p : nullptr, rnk());
    free_used += reuse; reused_total += reuse;
    nphys += cnt - reuse;
    sync();
  }
  // Courant halo of pred_corr on a decomposed domain (xchng_courants.ipp:15-160): element ranges inside the halo-extended
  // arrays [send to left, send to right, recv from left, recv from right]
  size_t courant_halo_geom(int which, T **arr, size_t off[4])
  {
    if (!halo || n_dims == 0) return 0;
    const size_t ny = m1(o.ny), nz = m1(o.nz), h = size_t(halo);
    size_t plane, n;
    if (which == 0) { plane = n_dims == 1 ? 1 : n_dims == 2 ? nz : nz * ny; *arr = courant_x.p; n = n_cx; }
    else if (which == 1) { if (n_dims < 3) return 0; plane = (ny + 1) * nz; *arr = courant_y.p; n = n_cy; }
    else { if (n_dims < 2) return 0; plane = n_dims == 2 ? nz + 1 : (nz + 1) * ny; *arr = courant_z.p; n = n_cz; }
    const size_t cnt = h * plane;
    if (which == 0) { off[0] = (h + 1) * plane; off[1] = size_t(o.nx) * plane; }
    else            { off[0] = cnt;             off[1] = size_t(o.nx) * plane; }
    off[2] = 0; off[3] = n - cnt;
    return cnt;
  }
  size_t courant_halo_count(int which) override { T *a; size_t off[4]; return courant_halo_geom(which, &a, off); }
  void courant_halo_copy(int which, int side, void *buf, bool pack) override
  {
    T *a; size_t off[4];
    const size_t cnt = courant_halo_geom(which, &a, off);
    if (!cnt) return;
    if (pack) HIPCHK(hipMemcpyAsync(buf, a + off[side], cnt * sizeof(T), hipMemcpyDeviceToDevice, st));
    else      HIPCHK(hipMemcpyAsync(a + off[2 + side], buf, cnt * sizeof(T), hipMemcpyDeviceToDevice, st));
    sync();
  }
  // ---- device-driven neighbour exchange (multi_HIP: lcx_multi.hpp; one process per GPU: lcx_exch_* of the C ABI): the three steps of
  // migrate_pack / _unpack / _finish without a host round trip for the counts.  inbox[0] receives from the left neighbour, inbox[1]
  // from the right one; outbox[0] / [1] hold what goes to the left / right when the transport is a copy (no peer mapping, RCCL).
  bool dev_exchange = false, inbox_finegrained = true;
  DevBuf<uint8_t> inbox[2], outbox[2]; size_t inbox_cap_rec = 0;
  DevBuf<uint32_t> xcnt;                 // [0..11] the step's record (k_collect_counts), [16] flags, [17] shift of the sorted order
  uint32_t exch_rec_h[12] = {0};         // the last step's record on the host
  int n_attr() const { return 4 + n_dims + n_ext; }
  size_t exch_bytes(size_t n_rec) const { return exch_msg_bytes<T>(n_rec, n_attr()); }
  // ONE capacity for every slab of a decomposition (a sender checks its count against the RECEIVER's inbox): two x-planes of the
  // thinnest slab's share of n_sd_max -- the reference sizes its buffers to half a plane (reserve_hskpng_npart.ipp:84-94,
  // config.hpp:25); a Courant number of 1 (the ring test) moves a whole plane, pred_corr allows 2 -- in whole tiles
  static size_t exch_capacity(size_t cap_sd, int nx_min)
  {
    const size_t c = std::min<size_t>(cap_sd, 2 * cap_sd / size_t(std::max(nx_min, 1)) + 1024);
    return (c + EXCH_TILE - 1) / EXCH_TILE * EXCH_TILE;
  }
  void exch_alloc(size_t cap_rec, bool with_outbox = false)
  {
    dev_exchange = true;
    inbox_cap_rec = cap_rec;
    for (auto &b : inbox) { inbox_finegrained = b.alloc_finegrained(exch_bytes(inbox_cap_rec)) && inbox_finegrained; HIPCHK(hipMemsetAsync(b.p, 0, EXCH_HDR, st)); }
    if (with_outbox) for (auto &b : outbox) { b.alloc(exch_bytes(inbox_cap_rec)); HIPCHK(hipMemsetAsync(b.p, 0, EXCH_HDR, st)); }
    xcnt.alloc_zero(24, st);
    if (!wg_mig.p) alloc_mig();
    // headroom in front of the sorted order (see sort_base): as many entries as a message can bring.  `rank` and `ijk` trade places
    // with the sorted arrays now and then (order_cells, reorder_storage), so all of them get it; nothing is stored in them yet
    sort_headroom = inbox_cap_rec;
    sync();
    for (DevBuf<uint32_t> *b : {&ijk, &sorted_id, &sorted_ijk, &rank, &sorted_alt}) { b->release(); b->alloc(cap + sort_headroom); }
    sync();
  }
  // x-planes at either end of the slab that an immigrant can reach (a Courant number of 1; pred_corr: 2)
  int bnd_planes() const { return halo ? halo : 1; }
  size_t plane_cells() const { return ncell / size_t(std::max(o.nx, 1)); }
  const bool no_overlap = dbg(LCX_DBG_NO_OVERLAP);      // measurement / test switch: the exchange without the overlapped re-sort
  // the overlapped re-sort needs the fused move's histogram, an interior, and the production rules for the storage order are not in
  // its way: a slab so thin that every plane is a boundary plane, rcyc and the unfused paths take the plain sequence
  // Round 4: when the step's re-sort can be left to the next condensation kernel (post_copy_after_fused_move: the storage-order kernel
  // carries the scatter, the in-cell ranking follows it) there is nothing to overlap -- the slab only scans its completed histogram behind
  // the unpack.  The overlapped form stays for the steps that re-order the storage, for strict arithmetic and for the other solvers.
  bool sort_will_be_deferred() const
  {
    const int every_ = o.reorder_every > 0 ? o.reorder_every : SLAB_REORDER_EVERY;
    const bool reorder_sched = !strict_order() && steps_since_reorder + 1 >= every_;
    return defer_sort_ok && lean_storage_cond() && !reorder_sched && replay.empty() && !dbg(LCX_DBG_EXCH_SORT_NOW);
  }
  bool overlap_possible() const
  { return dev_exchange && !no_overlap && fused_pending && n_dims > 0 && o.nx > 2 * bnd_planes() && sort_headroom > 0 && nphys > 0 && !sort_will_be_deferred(); }
  // ---- phase A of the overlapped re-sort: everything that does not depend on the neighbours, queued behind the pack kernel while
  // their messages travel: the stayers' scan, scatter and in-cell ranking of the interior cells [c_lo, c_hi).
  bool overlap_active = false, overlap_preshuffle = false; rng_src overlap_rs{nullptr, 0, 0, 0u, 0u}; uint32_t ov_c_lo = 0, ov_c_hi = 0;
  void exch_sort_interior()
  {
    overlap_active = false;
    if (!overlap_possible()) return;
    Range r(this, "exchange_sort_interior");
    overlap_active = true;
    ov_c_lo = uint32_t(size_t(bnd_planes()) * plane_cells()); ov_c_hi = uint32_t(ncell - size_t(bnd_planes()) * plane_cells());
    // (the storage re-ordering wants the plain order: its period is known ahead; a compaction that turns out to be due is not -- rare,
    // exch_finish re-ranks then)
    const int every_ = o.reorder_every > 0 ? o.reorder_every : SLAB_REORDER_EVERY;
    const bool reorder_sched = !strict_order() && steps_since_reorder + 1 >= every_;
    overlap_preshuffle = !strict_order() && !o.strict_fp && last_async_coal && o.coal_switch && !reorder_sched;
    // crowded interior cells from the stayers' histogram, which is final there (big_meta was cleared behind the previous sort)
    hipLaunchKernelGGL(k_list_big_cells, dim3(nblk(ov_c_hi - ov_c_lo)), dim3(BS), 0, st, size_t(ov_c_hi), (const uint32_t *)nullptr, uint32_t(CELLRANK_MAX), big_list.p,
                       big_meta_p(), big_meta_p() + 1, (const uint32_t *)cell_cnt.p, ov_c_lo);
    listed_from_hist = true;
    // stayers' CSR offsets; the histogram stays (the immigrants are added to it, their ranks continue behind the stayers), and so do the
    // step's counters
    exclusive_scan(cell_cnt.p, cell_start.p, ncell, cell_start.p + ncell, nullptr, nullptr, 0, scan_total.p + 3);      // (scan_total[0..1]: the emigrant counts)
    ++cells_version;
    join_rank();
    uint32_t *sid_h = sorted_id.p + sort_headroom, *sijk_h = sorted_ijk.p + sort_headroom, *alt_h = sorted_alt.p + sort_headroom;
    if (use_wave_flags && !wave_flag.p) wave_flag.alloc_zero(cap / WAVE + 64, st);
    hipLaunchKernelGGL(k_scatter_sorted, dim3(nblk(nphys)), dim3(BS), 0, st, nphys, ijk.p, rnk(), cell_start.p, sid_h, sijk_h,
                       sort_part{ov_c_lo, ov_c_hi, 1, nullptr, nullptr}, use_wave_flags ? wave_flag.p : (uint8_t *)nullptr);
    overlap_rs = rng_src{nullptr, 0, 0, 0u, 0u};
    if (overlap_preshuffle) overlap_rs = rand_un(nphys);
    launch_cellrank_range(overlap_preshuffle, overlap_rs, sijk_h, sid_h, alt_h, rank_range{cell_start.p + ov_c_lo, cell_start.p + ov_c_hi, nullptr}, nblk(nphys));
  }
  size_t bnd_pop_hint = 0;
  void rank_boundary(unsigned blocks)
  {
    join_rank();
    uint32_t *sid_h = sorted_id.p + sort_headroom, *sijk_h = sorted_ijk.p + sort_headroom, *alt_h = sorted_alt.p + sort_headroom;
    const uint32_t *shift = xcnt.p + 17;
    launch_cellrank_range(overlap_preshuffle, overlap_rs, sijk_h, sid_h, alt_h, rank_range{xcnt.p + 19 /* = 0 */, cell_start.p + ov_c_lo, shift}, blocks);
    launch_cellrank_range(overlap_preshuffle, overlap_rs, sijk_h, sid_h, alt_h, rank_range{cell_start.p + ov_c_hi, cell_start.p + ncell, shift}, blocks);
  }
  void launch_cellrank_range(bool shuffle, const rng_src &rs, const uint32_t *sijk_p, const uint32_t *in, uint32_t *out, const rank_range &rg, unsigned blocks)
  {
    const int crowded = 0;       // (cells above CELLRANK_MAX keep their arrival order here and are sorted from the list, as everywhere)
    if (shuffle && !rs.un && !shuffle_philox && (rs.s1 | rs.s2) && !dbg(LCX_DBG_RANK_BY_COUNTING)) hipLaunchKernelGGL(k_cellrank_bkt<>, dim3(blocks), dim3(BS), 0, st, size_t(0), sijk_p, cell_start.p, in, out, rs, rg);
    else if (shuffle && !rs.un && !shuffle_philox) hipLaunchKernelGGL((k_cellrank<uint32_t, true>), dim3(blocks), dim3(BS), 0, st, size_t(0), sijk_p, cell_start.p, in, out, rs, crowded, rg);
    else if (shuffle) hipLaunchKernelGGL(k_cellrank<uint64_t>, dim3(blocks), dim3(BS), 0, st, size_t(0), sijk_p, cell_start.p, in, out, rs, crowded, rg);
    else hipLaunchKernelGGL(k_cellrank<uint32_t>, dim3(blocks), dim3(BS), 0, st, size_t(0), sijk_p, cell_start.p, in, out, rs, crowded, rg);
  }
  // emigrants of both faces -> the neighbours' inboxes (pointers this device can write: peer-mapped, inboxes on this very device, or
  // this slab's own outboxes; nullptr: no neighbour behind that face), their multiplicities cleared in the same launch.
  // cap_l / cap_r: the capacity of the inbox behind each pointer; next_l / next_r: header word 2 (see k_pack_dev)
  const unsigned test_pack_delay_us = o.dbg_pack_delay_us > 0 ? unsigned(o.dbg_pack_delay_us) : 0u;
  void exch_pack(uint8_t *dst_l, double lft_x1, size_t cap_l, uint8_t *dst_r, double rgt_x0, size_t cap_r, uint32_t next_l = 0, uint32_t next_r = 0)
  {
    // test / measurement: the slabs with an odd first plane are late with their messages by so many microseconds
    if (test_pack_delay_us && ((o.n_x_bfr / std::max(o.nx, 1)) & 1)) hipLaunchKernelGGL(k_spin_us, dim3(1), dim3(1), 0, st, test_pack_delay_us);
    const unsigned half = nblk(std::max(cap_l, cap_r));
    hipLaunchKernelGGL(k_pack_dev<T>, dim3(2 * half), dim3(BS), 0, st, scan_total.p, half,
                       pack_side<T>{mig_ids[0].p, dst_l, T(lft_x1), T(o.x0), uint32_t(cap_l), next_l},
                       pack_side<T>{mig_ids[1].p, dst_r, T(rgt_x0), T(o.x1), uint32_t(cap_r), next_r}, aset(A), g);
  }
  // have_l / have_r: records of each message that have arrived (all of it unless the transport ships in two parts)
  // With the overlapped re-sort (phase B): the immigrants' ranks continue behind the stayers of their cells, the boundary planes'
  // offsets are redone, the interior's shifted, boundary SDs scattered and ranked -- all queued, nothing known to the host yet.
  void exch_unpack(bool from_l, bool from_r, uint32_t have_l = ~0u, uint32_t have_r = ~0u, bool retry = false)
  {
    Range r(this, "exchange_unpack");
    HIPCHK(hipMemsetAsync(xcnt.p + 16, 0, 6 * sizeof(uint32_t), st));      // flags, shift, extent, the constant 0, -, boundary population
    const bool ov = overlap_active;
    if (retry && fused_pending && !ov) HIPCHK(hipMemsetAsync(big_meta_p(), 0, 2 * sizeof(uint32_t), st));   // (the first attempt listed the crowded cells already)
    const uint8_t *in_l = from_l ? inbox[0].p : nullptr, *in_r = from_r ? inbox[1].p : nullptr;
    const uint32_t *n_free = strict_order() ? (const uint32_t *)nullptr : scan_total.p;
    // the boundary ranking's grid: sized from the last step's boundary populations (twice that and half a message), the whole extent
    // when there is no history; a step that outgrows it says so (flag 16) and the host ranks the boundary again
    const size_t ext_max = std::min(cap, nphys + 2 * inbox_cap_rec);
    const unsigned bnd_blocks = ov ? nblk(bnd_pop_hint ? std::min(ext_max, 2 * bnd_pop_hint + inbox_cap_rec / 2) : ext_max) : 0u;
    // (second call of the step: the first one's scan has cleared the histogram -- the stayers' counts come back from its offsets)
    if (ov && retry) hipLaunchKernelGGL(k_csr_to_counts, dim3(nblk(ncell)), dim3(BS), 0, st, cell_start.p, cell_cnt.p, ncell);
    hipLaunchKernelGGL(k_unpack_dev<T>, dim3(nblk(2 * inbox_cap_rec)), dim3(BS), 0, st, in_l, in_r, have_l, have_r,
                       nphys, cap, aset(A), g, T(o.x0), T(o.x1), T(5e-4), mig_ids[0].p, mig_ids[1].p, n_free,
                       ijk.p, fused_pending ? cell_cnt.p : nullptr, rnk(), xcnt.p + 16, int(ov), ov_c_lo, ov_c_hi, big_meta_p(), xcnt.p + 22, int(retry),
                       (ov && use_wave_flags) ? wave_flag.p : (uint8_t *)nullptr);
    if (ov) {
      const uint32_t *shift = xcnt.p + 17, *extent = xcnt.p + 18;
      // crowded boundary cells from the completed histogram, then the final offsets of every cell (the histogram is cleared behind them)
      hipLaunchKernelGGL(k_list_big_cells, dim3(nblk(ov_c_lo)), dim3(BS), 0, st, size_t(ov_c_lo), (const uint32_t *)nullptr, uint32_t(CELLRANK_MAX), big_list.p,
                         big_meta_p(), big_meta_p() + 1, (const uint32_t *)cell_cnt.p, 0u);
      hipLaunchKernelGGL(k_list_big_cells, dim3(nblk(ncell - ov_c_hi)), dim3(BS), 0, st, ncell, (const uint32_t *)nullptr, uint32_t(CELLRANK_MAX), big_list.p,
                         big_meta_p(), big_meta_p() + 1, (const uint32_t *)cell_cnt.p, ov_c_hi);
      exclusive_scan(cell_cnt.p, cell_start.p, ncell, cell_start.p + ncell, cell_cnt.p, nullptr, 0, scan_total.p + 3);
      join_rank();
      uint32_t *sid_h = sorted_id.p + sort_headroom, *sijk_h = sorted_ijk.p + sort_headroom;
      if (use_wave_flags)
        hipLaunchKernelGGL(k_scatter_flagged, dim3(nblk((ext_max + 16 * WAVE - 1) / (16 * WAVE), BS / WAVE)), dim3(BS), 0, st, ext_max, wave_flag.p, ijk.p, rnk(),
                           cell_start.p, sid_h, sijk_h, sort_part{ov_c_lo, ov_c_hi, 2, extent, shift});
      else
        hipLaunchKernelGGL(k_scatter_outside4, dim3(nblk((ext_max + 3) / 4)), dim3(BS), 0, st, ext_max, ijk.p, rnk(), cell_start.p, sid_h, sijk_h,
                           sort_part{ov_c_lo, ov_c_hi, 2, extent, shift});
      rank_boundary(bnd_blocks);
    }
    else if (fused_pending) list_big_from_hist();       // the histogram is complete now: crowded cells for order_cells, same read-back
    hipLaunchKernelGGL(k_collect_counts, dim3(1), dim3(64), 0, st, step_cnt.p, scan_total.p, in_l, in_r,
                       xcnt.p + 16, ov ? (const uint32_t *)(xcnt.p + 17) : (const uint32_t *)nullptr, xcnt.p, ov ? step_cnt.p : (uint32_t *)nullptr,
                       ov ? (const uint32_t *)(cell_start.p + ov_c_lo) : (const uint32_t *)nullptr, cell_start.p + ov_c_hi, cell_start.p + ncell, bnd_blocks * unsigned(BS));
  }
  size_t exch_moved = 0;       // super-droplets this slab has sent so far (bench / diagnostics)
  // returns false when a message had not arrived in full (nothing was unpacked: ship the rest, call exch_unpack and this again)
  bool exch_finish(const lcx_opts_t &opts)
  {
    uint32_t *h = exch_rec_h;
    read_back(h, xcnt.p, 12);                           // the step's one host synchronisation
    // (overlapped re-sort: the boundary pass has run without the immigrants of the incomplete message -- offsets with a shift of zero,
    // boundary SDs scattered and ranked among themselves; the second call redoes exactly that pass with them, nothing else is lost)
    if (h[5] & 4u) return false;
    if (h[5] & 1u) throw lcx_error("libcloudph++: more super-droplets crossed a slab face in one step than the exchange buffer holds (" +
                                    std::to_string(inbox_cap_rec) + " records); raise opts_init.n_sd_max");
    if (h[5] & 2u) throw lcx_error("n_sd_max (" + std::to_string(o.n_sd_max) + ") < n_part after the neighbour exchange");
    if (h[5] & 8u) throw lcx_error("libcloudph++: an immigrant landed beyond the boundary planes of its slab (Courant number above the scheme's limit?)");
    const size_t n_in = size_t(h[3]) + h[4], n_free = strict_order() ? 0 : size_t(h[1]) + h[2], reuse = std::min(n_in, n_free);
    exch_moved += size_t(h[1]) + h[2];
    nphys += n_in - reuse;
    lft_count = rgt_count = 0; free_n[0] = free_n[1] = 0; free_used = 0; reused_total = 0;
    if (fused_pending && overlap_active) {
      if (h[5] & 16u) rank_boundary(nblk(std::min(cap, nphys + 2 * inbox_cap_rec)));      // (this step's boundary planes outgrew the planned grid)
      bnd_pop_hint = h[11];
      fused_pending = false; overlap_active = false;
      finish_overlapped(opts, long(h[0]) - long(reuse), h[8], h[6], h[7]);
    }
    else if (fused_pending) {
      fused_pending = false;
      meta_known_valid = listed_from_hist; meta_known_v[0] = h[6]; meta_known_v[1] = h[7];
      post_copy_after_fused_move(opts, long(h[0]) - long(reuse));
    }
    else post_copy(opts);
    return true;
  }
  // phase C of the overlapped re-sort: the order is in place (interior ranked before the messages arrived, boundary planes behind
  // them); the host learns the counts, sorts the crowded cells from their list and applies the storage rules of post_copy_after_fused_move
  void finish_overlapped(const lcx_opts_t &opts, long dead_l, uint32_t shift, uint32_t n_big, uint32_t max_big)
  {
    const size_t dead = size_t(std::max(0l, dead_l));
    listed_from_hist = meta_known_valid = false;
    sorted_id.swap(sorted_alt);
    sort_base = sort_headroom - shift;
    npart = nphys - dead;
    big_n = n_big; big_mx = max_big; meta_version = cells_version;
    sorted = true; sorted_shuffled = overlap_preshuffle; shuffle_fresh = overlap_preshuffle;
    if (overlap_preshuffle) last_shuffle_rs = overlap_rs;
    const bool strict = strict_order();
    const bool compact_now = dead && (eager_compact || dead * 32 > nphys);
    if (compact_now && strict) { post_copy(opts, true); return; }
    Range r(this, "post_copy");
    if (big_n) sort_listed_cells(overlap_preshuffle, overlap_rs);
    const int every_ = o.reorder_every > 0 ? o.reorder_every : SLAB_REORDER_EVERY;
    if (compact_now || (!strict && ++steps_since_reorder >= every_)) {
      if (sorted_shuffled) { order_cells(false); shuffle_fresh = false; }      // (a compaction that was not foreseen: the re-ordering wants ascending ids)
      reorder_storage();
    }
  }

  // ---- lcx_exch_*: the device-driven exchange with a copying transport between processes (RCCL / host-staged, libcloudphxx_amd/multi.py)
  size_t x_enable(int nx_min) override
  {
    if (init_called) throw lcx_error("libcloudph++: lcx_exch_enable must be called before init()");
    if (!distmem()) throw lcx_error("libcloudph++: lcx_exch_enable needs a slab with a neighbour (bcond_lft / bcond_rgt = distmem)");
    exch_alloc(exch_capacity(cap, nx_min), true);
    return inbox_cap_rec;
  }
  void need_exch() const { if (!dev_exchange) throw lcx_error("libcloudph++: call lcx_exch_enable first"); }
  void x_buffers(void **p4) override { need_exch(); p4[0] = outbox[0].p; p4[1] = outbox[1].p; p4[2] = inbox[0].p; p4[3] = inbox[1].p; }
  size_t x_message_bytes(size_t n_rec) override { return exch_bytes(n_rec); }
  void x_pack(bool has_l, double lft_x1, bool has_r, double rgt_x0, unsigned next_l, unsigned next_r) override
  {
    need_exch();
    { Range r(this, "exchange_pack");
      exch_pack(has_l ? outbox[0].p : nullptr, lft_x1, inbox_cap_rec, has_r ? outbox[1].p : nullptr, rgt_x0, inbox_cap_rec, next_l, next_r); }
    puddle_reduce_deferred();
  }
  int x_unpack_calls = 0;             // within the current step (the second call is the retry with the whole messages)
  void x_unpack(bool from_l, bool from_r, unsigned have_l, unsigned have_r) override { need_exch(); exch_unpack(from_l, from_r, have_l, have_r, x_unpack_calls++ > 0); }
  void x_sort_interior() override { need_exch(); exch_sort_interior(); }
  bool x_finish(const lcx_opts_t &opts, unsigned *rec) override
  {
    need_exch();
    const bool done = exch_finish(opts);
    if (done) x_unpack_calls = 0;
    for (int k = 0; k < 12; ++k) rec[k] = exch_rec_h[k];
    return done;
  }
  void *stream() override { return (void *)st; }

  void migrate_finish(const lcx_opts_t &opts) override
  {
    flag_emigrants();
    free_n[0] = free_n[1] = 0; free_used = 0;
    if (fused_pending) {
      // k_move left the histogram of the SDs that stayed and k_unpack added the immigrants': scan / scatter / rank remain
      fused_pending = false;
      post_copy_after_fused_move(opts);
    } else { reused_total = 0; post_copy(opts); }
  }
};

} // namespace lcx

#include "lcx_multi.hpp"

// =================================================================================================
// C ABI
// =================================================================================================
using lcx::IParticles;
static thread_local std::string g_err;
// a handle owns its object, except the per-slab handles of a multi-device object (lcx_multi_slab), which the parent keeps
struct lcx_particles {
  IParticles *p = nullptr; bool own = true;
  std::vector<std::unique_ptr<lcx_particles>> slabs;
  ~lcx_particles() { slabs.clear(); if (own) delete p; }
};
static inline IParticles *bound(lcx_particles *h) { h->p->bind(); return h->p; }
// An entry point makes the object's device current on the CALLER's thread (bind) and a multi-device object walks over its devices:
// the caller's own current device is put back when the call returns, so that a host model that allocates or launches on "its" GPU
// next to this library keeps doing so
struct DevGuard {
  int prev = -1;
  DevGuard() { if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipGetLastError(); } }
  ~DevGuard() { int now = -1; if (prev >= 0 && hipGetDevice(&now) == hipSuccess && now != prev) (void)hipSetDevice(prev); }
};

#define LCX_TRY(body) try { DevGuard dev_guard_; body; return 0; } catch (const std::exception &e) { g_err = e.what(); return 1; } catch (...) { g_err = "libcloudph++: unknown error"; return 1; }
#define H (bound(h))

template <class T> static bool multi_slabs(lcx_particles *h, int *n, int i, lcx_particles **out)
{
  auto *m = dynamic_cast<lcx::MultiParticles<T> *>(h->p);
  if (!m) return false;
  if (n) *n = m->D;
  if (out) {
    if (i < 0 || i >= m->D) throw std::runtime_error("libcloudph++: no such slab");
    if (h->slabs.empty()) for (int k = 0; k < m->D; ++k) { h->slabs.emplace_back(new lcx_particles); h->slabs[k]->p = m->slab[k].get(); h->slabs[k]->own = false; }
    *out = h->slabs[i].get();
  }
  return true;
}

extern "C" {

const char *lcx_last_error(void) { return g_err.c_str(); }
const char *lcx_version(void) { return "libcloudphxx_amd 0.1 (HIP gfx950)"; }

void lcx_opts_init_default(lcx_opts_init_t *o)
{
  memset(o, 0, sizeof *o);
  o->dx = o->dy = o->dz = 1; o->x1 = o->y1 = o->z1 = 1;
  o->sstp_cond = o->sstp_coal = o->sstp_chem = o->sstp_cond_act = 1;
  o->sedi_switch = 1; o->coal_switch = 1; o->sstp_cond_mix = 1;
  o->RH_max = .95; o->rng_seed = 44; o->rng_seed_init = 44;
  o->sstp_cond_adapt_drw2_eps = 1e-4; o->sstp_cond_adapt_drw2_max = 4; o->rc2_T = 10;
  o->adve_scheme = LCX_ADVE_IMPLICIT; o->RH_formula = LCX_RH_PV_CC;
  o->dev_id = -1; o->rd_min = -1; o->rd_max = -1; o->th_dry = 1; o->strict_fp = 0; o->cond_solver = 1;   /* (round 5: the API default, see lcx.h) */
}
void lcx_opts_default(lcx_opts_t *o)
{
  memset(o, 0, sizeof *o);
  o->adve = o->sedi = o->cond = o->coal = 1; o->RH_max = 44; o->dt = -1;
}
int lcx_create(const lcx_opts_init_t *oi, int real_kind, lcx_particles **out)
{
  LCX_TRY({
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
      throw std::runtime_error("libcloudph++: no HIP device available (this backend has no CPU fallback)");
    std::unique_ptr<lcx_particles> h(new lcx_particles);
    if (real_kind == 8) h->p = new lcx::Particles<double>(*oi);
    else if (real_kind == 4) h->p = new lcx::Particles<float>(*oi);
    else throw std::runtime_error("libcloudph++: real_kind must be 4 (float) or 8 (double)");
    *out = h.release();
  })
}
int lcx_create_multi(const lcx_opts_init_t *oi, int real_kind, lcx_particles **out)
{
  LCX_TRY({
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
      throw std::runtime_error("libcloudph++: no HIP device available (this backend has no CPU fallback)");
    std::unique_ptr<lcx_particles> h(new lcx_particles);
    if (real_kind == 8) h->p = new lcx::MultiParticles<double>(*oi);
    else if (real_kind == 4) h->p = new lcx::MultiParticles<float>(*oi);
    else throw std::runtime_error("libcloudph++: real_kind must be 4 (float) or 8 (double)");
    *out = h.release();
  })
}
int lcx_multi_dev_count(lcx_particles *h, int *n)
{ LCX_TRY({ if (!multi_slabs<double>(h, n, 0, nullptr) && !multi_slabs<float>(h, n, 0, nullptr)) *n = 1; }) }
int lcx_multi_slab(lcx_particles *h, int i, lcx_particles **slab)
{
  LCX_TRY({
    if (!multi_slabs<double>(h, nullptr, i, slab) && !multi_slabs<float>(h, nullptr, i, slab))
      throw std::runtime_error("libcloudph++: not a multi-device object");
  })
}
void lcx_destroy(lcx_particles *h) { if (h && h->own) { DevGuard dev_guard_; delete h; } }
int lcx_init(lcx_particles *h, const lcx_arrinfo_t *th, const lcx_arrinfo_t *rv, const lcx_arrinfo_t *rhod, const lcx_arrinfo_t *p,
             const lcx_arrinfo_t *cx, const lcx_arrinfo_t *cy, const lcx_arrinfo_t *cz) { LCX_TRY(H->init(th, rv, rhod, p, cx, cy, cz)) }
int lcx_sync_in(lcx_particles *h, const lcx_arrinfo_t *th, const lcx_arrinfo_t *rv, const lcx_arrinfo_t *rhod, const lcx_arrinfo_t *cx,
                const lcx_arrinfo_t *cy, const lcx_arrinfo_t *cz, const lcx_arrinfo_t *diss) { LCX_TRY(H->sync_in(th, rv, rhod, cx, cy, cz, diss)) }
int lcx_step_cond(lcx_particles *h, const lcx_opts_t *o, const lcx_arrinfo_t *th, const lcx_arrinfo_t *rv) { LCX_TRY(H->step_cond(*o, th, rv)) }
int lcx_step_sync(lcx_particles *h, const lcx_opts_t *o, const lcx_arrinfo_t *th, const lcx_arrinfo_t *rv, const lcx_arrinfo_t *rhod,
                  const lcx_arrinfo_t *cx, const lcx_arrinfo_t *cy, const lcx_arrinfo_t *cz, const lcx_arrinfo_t *diss)
{ LCX_TRY(H->step_sync(*o, th, rv, rhod, cx, cy, cz, diss)) }
int lcx_step_async(lcx_particles *h, const lcx_opts_t *o) { LCX_TRY(H->step_async(*o)) }
int lcx_diag_sd_conc(lcx_particles *h) { LCX_TRY(H->diag_sd_conc()) }
int lcx_diag_pressure(lcx_particles *h) { LCX_TRY(H->diag_cell(0)) }
int lcx_diag_temperature(lcx_particles *h) { LCX_TRY(H->diag_cell(1)) }
int lcx_diag_RH(lcx_particles *h) { LCX_TRY(H->diag_cell(2)) }
int lcx_diag_vel_div(lcx_particles *h) { LCX_TRY(H->diag_vel_div()) }
int lcx_diag_all(lcx_particles *h) { LCX_TRY(H->diag_select(0, 0, 1, 0, 0)) }
int lcx_diag_water(lcx_particles *h) { LCX_TRY(H->diag_select(2, 0, 1, 0, 0)) }
int lcx_diag_dry_rng(lcx_particles *h, double a, double b) { LCX_TRY(H->diag_select(1, 0, 0, std::pow(a, 3), std::pow(b, 3))) }
int lcx_diag_wet_rng(lcx_particles *h, double a, double b) { LCX_TRY(H->diag_select(1, 0, 1, std::pow(a, 2), std::pow(b, 2))) }
int lcx_diag_kappa_rng(lcx_particles *h, double a, double b) { LCX_TRY(H->diag_select(1, 0, 2, a, b)) }
int lcx_diag_dry_rng_cons(lcx_particles *h, double a, double b) { LCX_TRY(H->diag_select(1, 1, 0, std::pow(a, 3), std::pow(b, 3))) }
int lcx_diag_wet_rng_cons(lcx_particles *h, double a, double b) { LCX_TRY(H->diag_select(1, 1, 1, std::pow(a, 2), std::pow(b, 2))) }
int lcx_diag_kappa_rng_cons(lcx_particles *h, double a, double b) { LCX_TRY(H->diag_select(1, 1, 2, a, b)) }
int lcx_diag_dry_mom(lcx_particles *h, int k) { LCX_TRY(H->diag_mom(0, k / 3.)) }
int lcx_diag_wet_mom(lcx_particles *h, int k) { LCX_TRY(H->diag_mom(1, k / 2.)) }
int lcx_diag_kappa_mom(lcx_particles *h, int k) { LCX_TRY(H->diag_mom(2, k)) }
int lcx_diag_incloud_time_mom(lcx_particles *h, int k) { LCX_TRY(H->diag_mom(4, k)) }
int lcx_diag_up_mom(lcx_particles *h, int k) { LCX_TRY(H->diag_mom(5, k)) }
int lcx_diag_vp_mom(lcx_particles *h, int k) { LCX_TRY(H->diag_mom(6, k)) }
int lcx_diag_wp_mom(lcx_particles *h, int k) { LCX_TRY(H->diag_mom(7, k)) }
int lcx_diag_water_cons(lcx_particles *h) { LCX_TRY(H->diag_select(2, 1, 1, 0, 0)) }
int lcx_diag_RH_ge_Sc(lcx_particles *h) { LCX_TRY(H->diag_act(0)) }
int lcx_diag_rw_ge_rc(lcx_particles *h) { LCX_TRY(H->diag_act(1)) }
int lcx_diag_wet_mass_dens(lcx_particles *h, double rad, double sig0) { LCX_TRY(H->diag_wet_mass_dens(rad, sig0)) }
int lcx_diag_precip_rate(lcx_particles *h) { LCX_TRY(H->diag_precip_rate()) }
int lcx_diag_max_rw(lcx_particles *h) { LCX_TRY(H->diag_max_rw()) }
int lcx_outbuf(lcx_particles *h, const void **data, size_t *n) { LCX_TRY(H->outbuf(data, n)) }
int lcx_get_attr(lcx_particles *h, const char *name, void *out, size_t cap, size_t *n) { LCX_TRY(H->get_attr(name, out, cap, n)) }
int lcx_diag_puddle(lcx_particles *h, double *out) { LCX_TRY(H->diag_puddle(out)) }
int lcx_n_part(lcx_particles *h, size_t *n) { LCX_TRY(*n = H->n_part()) }
int lcx_n_cell(lcx_particles *h, size_t *n) { LCX_TRY(*n = H->n_cell()) }
int lcx_real_kind(lcx_particles *h, int *k) { LCX_TRY(*k = H->real_kind()) }
int lcx_get_state_u64(lcx_particles *h, const char *name, unsigned long long *out, size_t cap, size_t *n) { LCX_TRY(H->get_state_u64(name, out, cap, n)) }
int lcx_get_state_real(lcx_particles *h, const char *name, double *out, size_t cap, size_t *n) { LCX_TRY(H->get_state_real(name, out, cap, n)) }
int lcx_set_particles(lcx_particles *h, size_t n, const unsigned long long *mult, const double *rd3, const double *rw2, const double *kpa,
                      const double *vt, const double *x, const double *y, const double *z) { LCX_TRY(H->set_particles(n, mult, rd3, rw2, kpa, vt, x, y, z)) }
int lcx_rng_replay_push(lcx_particles *h, int kind, const double *data, size_t n) { LCX_TRY(H->rng_replay_push(kind, data, n)) }
int lcx_rng_replay_pending(lcx_particles *h, size_t *n) { LCX_TRY(*n = H->rng_replay_pending()) }
int lcx_rng_dump(lcx_particles *h, int call, int which, double *out, size_t cap, size_t *n) { LCX_TRY(H->rng_dump(call, which, out, cap, n)) }
int lcx_set_state_real(lcx_particles *h, const char *name, const double *data, size_t n) { LCX_TRY(H->set_state_real(name, data, n)) }
int lcx_stage(lcx_particles *h, const char *stage, const lcx_opts_t *o) { LCX_TRY(H->stage(stage, o)) }
int lcx_timings(lcx_particles *h, const char **names, double *ms, size_t cap, size_t *n) { LCX_TRY(H->timings(names, ms, cap, n)) }
int lcx_set_profiling(lcx_particles *h, int on) { LCX_TRY(H->set_profiling(on)) }
int lcx_migrate_counts(lcx_particles *h, size_t *l, size_t *r) { LCX_TRY(H->migrate_counts(l, r)) }
size_t lcx_migrate_record_bytes(lcx_particles *h) { DevGuard dev_guard_; return H->migrate_record_bytes(); }
int lcx_migrate_pack(lcx_particles *h, int side, double x_rmt, void *buf, size_t cap) { LCX_TRY(H->migrate_pack(side, x_rmt, buf, cap)) }
int lcx_migrate_unpack(lcx_particles *h, const void *buf, size_t count) { LCX_TRY(H->migrate_unpack(buf, count)) }
int lcx_migrate_finish(lcx_particles *h, const lcx_opts_t *o) { LCX_TRY(H->migrate_finish(*o)) }
int lcx_exch_enable(lcx_particles *h, int nx_min, size_t *cap_rec) { LCX_TRY(*cap_rec = H->x_enable(nx_min)) }
int lcx_exch_buffers(lcx_particles *h, void *ptrs[4]) { LCX_TRY(H->x_buffers(ptrs)) }
size_t lcx_exch_message_bytes(lcx_particles *h, size_t n_rec) { try { DevGuard dev_guard_; return H->x_message_bytes(n_rec); } catch (const std::exception &e) { g_err = e.what(); return 0; } }
int lcx_exch_pack(lcx_particles *h, int has_lft, double lft_x1, int has_rgt, double rgt_x0, unsigned next_cap_lft, unsigned next_cap_rgt)
{ LCX_TRY(H->x_pack(has_lft != 0, lft_x1, has_rgt != 0, rgt_x0, next_cap_lft, next_cap_rgt)) }
int lcx_exch_sort_interior(lcx_particles *h) { LCX_TRY(H->x_sort_interior()) }
int lcx_exch_unpack(lcx_particles *h, int from_lft, int from_rgt, unsigned have_lft, unsigned have_rgt) { LCX_TRY(H->x_unpack(from_lft != 0, from_rgt != 0, have_lft, have_rgt)) }
int lcx_exch_finish(lcx_particles *h, const lcx_opts_t *o, unsigned rec[12], int *complete) { LCX_TRY(*complete = H->x_finish(*o, rec) ? 1 : 0) }
int lcx_stream(lcx_particles *h, void **hip_stream) { LCX_TRY(*hip_stream = H->stream()) }
size_t lcx_courant_halo_count(lcx_particles *h, int which) { DevGuard dev_guard_; return H->courant_halo_count(which); }
int lcx_courant_halo_pack(lcx_particles *h, int which, int side, void *buf) { LCX_TRY(H->courant_halo_copy(which, side, buf, true)) }
int lcx_courant_halo_unpack(lcx_particles *h, int which, int side, const void *buf) { LCX_TRY(H->courant_halo_copy(which, side, const_cast<void *>(buf), false)) }
int lcx_dev_alloc(void **ptr, size_t bytes) { LCX_TRY({ if (hipMalloc(ptr, bytes ? bytes : 1) != hipSuccess) throw std::runtime_error("libcloudph++ (HIP): hipMalloc failed"); }) }
int lcx_dev_free(void *ptr) { LCX_TRY({ if (hipFree(ptr) != hipSuccess) throw std::runtime_error("libcloudph++ (HIP): hipFree failed"); }) }
int lcx_dev_copy(void *dst, const void *src, size_t bytes, int kind)
{
  LCX_TRY({
    const hipMemcpyKind k = kind == 1 ? hipMemcpyHostToDevice : kind == 2 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    if (hipMemcpy(dst, src, bytes, k) != hipSuccess) throw std::runtime_error("libcloudph++ (HIP): hipMemcpy failed");
  })
}
int lcx_dev_sync(void) { LCX_TRY({ if (hipDeviceSynchronize() != hipSuccess) throw std::runtime_error("libcloudph++ (HIP): device synchronize failed"); }) }

int lcx_math_probe(int which, const double *x, double *y, size_t n)
{
  LCX_TRY({
    double *d = nullptr;
    if (hipMalloc(&d, (n ? n : 1) * sizeof(double)) != hipSuccess) throw std::runtime_error("libcloudph++ (HIP): hipMalloc failed");
    (void)hipMemcpy(d, x, n * sizeof(double), hipMemcpyHostToDevice);
    if (n) hipLaunchKernelGGL(lcx::k_math_probe, dim3((n + 255) / 256), dim3(256), 0, 0, which, d, n);
    const hipError_t e = hipMemcpy(y, d, n * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) throw std::runtime_error(std::string("libcloudph++ (HIP): math probe failed: ") + hipGetErrorString(e));
  })
}

int lcx_philox_probe(const unsigned long long *ics, size_t n, unsigned int *out, int on_device)
{
  LCX_TRY({
    if (!on_device) {
      for (size_t i = 0; i < n; ++i) { uint32_t r[4]; lcx::philox::gen(ics[3 * i], ics[3 * i + 1], ics[3 * i + 2], r); for (int k = 0; k < 4; ++k) out[4 * i + k] = r[k]; }
    } else {
      uint64_t *d_in = nullptr; uint32_t *d_out = nullptr;
      if (hipMalloc(&d_in, (n ? n : 1) * 3 * sizeof(uint64_t)) != hipSuccess || hipMalloc(&d_out, (n ? n : 1) * 4 * sizeof(uint32_t)) != hipSuccess)
        throw std::runtime_error("libcloudph++ (HIP): hipMalloc failed");
      (void)hipMemcpy(d_in, ics, n * 3 * sizeof(uint64_t), hipMemcpyHostToDevice);
      if (n) hipLaunchKernelGGL(lcx::k_philox_probe, dim3((n + 255) / 256), dim3(256), 0, 0, d_in, n, d_out);
      const hipError_t e = hipMemcpy(out, d_out, n * 4 * sizeof(uint32_t), hipMemcpyDeviceToHost);
      (void)hipFree(d_in); (void)hipFree(d_out);
      if (e != hipSuccess) throw std::runtime_error(std::string("libcloudph++ (HIP): philox probe failed: ") + hipGetErrorString(e));
    }
  })
}

int lcx_common_eval(const char *name, const double *a, int n, double *out)
{
  LCX_TRY({
    const std::string s(name);
    auto need = [&](int k) { if (n != k) throw std::runtime_error("libcloudph++: common." + s + " takes " + std::to_string(k) + " argument(s)"); };
    using c = lcx::cst<double>;
    const double kap = c::R_d / c::c_pd;
    if (s == "th_dry2std") { need(2); *out = a[0] / pow(1 + a[1] * c::R_v / c::R_d, kap); }                    // theta_dry.hpp:101-113
    else if (s == "th_std2dry") { need(2); *out = a[0] * pow(1 + a[1] * c::R_v / c::R_d, kap); }               // theta_dry.hpp:86-99
    else if (s == "exner") { need(1); *out = lcx::exner(a[0]); }
    else if (s == "p_v") { need(2); *out = lcx::p_v(a[0], a[1]); }
    else if (s == "p_vs") { need(1); *out = lcx::p_vs(a[0]); }
    else if (s == "r_vs") { need(2); *out = c::eps / (a[1] / lcx::p_vs(a[0]) - 1); }                            // const_cp.hpp r_vs
    else if (s == "p_vs_tet") { need(1); *out = lcx::tet_p_vs(a[0]); }
    else if (s == "l_v") { need(1); *out = lcx::l_v(a[0]); }
    else if (s == "T") { need(2); *out = lcx::theta_dry_T(a[0], a[1]); }
    else if (s == "p") { need(3); *out = lcx::theta_dry_p(a[0], a[1], a[2]); }
    else if (s == "visc") { need(1); *out = lcx::visc(a[0]); }
    else if (s == "rw3_cr") { need(3); *out = lcx::rw3_cr_of(a[0], a[1], a[2]); }
    else if (s == "S_cr") { need(3); *out = lcx::S_cr(a[0], a[1], a[2]); }
    else if (s == "p_hydro") {                                                                                  // hydrostatic.hpp:24-38
      need(5);
      const double R_moist = (c::R_d + a[2] * c::R_v) / (1 + a[2]);                                             // moist_air.hpp:54-70
      *out = c::p_1000 * pow(pow(a[4] / c::p_1000, kap) - kap * c::g / a[1] / R_moist * (a[0] - a[3]), c::c_pd / c::R_d);
    }
    else if (s == "rhod") { need(3); *out = (a[0] - lcx::p_v(a[0], a[2])) / (pow(a[0] / c::p_1000, kap) * c::R_d * a[1]); }   // theta_std.hpp:23-32
    else if (s == "R_d") { need(0); *out = c::R_d; } else if (s == "R_v") { need(0); *out = c::R_v; }
    else if (s == "c_pd") { need(0); *out = c::c_pd; } else if (s == "c_pv") { need(0); *out = c::c_pv; }
    else if (s == "c_pw") { need(0); *out = c::c_pw; } else if (s == "g") { need(0); *out = c::g; }
    else if (s == "p_1000") { need(0); *out = c::p_1000; } else if (s == "eps") { need(0); *out = c::eps; }
    else if (s == "rho_stp") { need(0); *out = c::rho_stp; } else if (s == "rho_w") { need(0); *out = c::rho_w; }
    else throw std::runtime_error("libcloudph++: common." + s + " is not provided by this backend");
  })
}

} // extern "C"
