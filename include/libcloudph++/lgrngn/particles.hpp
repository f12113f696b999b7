// particles_proto_t<real_t> / particles_t<real_t, backend>: the reference's user-facing classes
// (reference: include/libcloudph++/lgrngn/particles.hpp:17-134, :136-246) re-implemented as a thin,
// header-only shim over the C ABI of the HIP library (include/lcx.h, liblcx_hip.so).
//
// Same virtuals, same default arguments, same exception type (std::runtime_error with the library's
// "libcloudph++: ..." message), same call-order rules (init once; then sync_in -> step_cond [= step_sync]
// -> step_async, diag_* in between), same ownership (caller owns every array; th / rv are written back at
// the end of step_cond; outbuf() points into library memory valid until the next outbuf()).
#pragma once
#include "extincl.hpp"
#include "opts.hpp"
#include "opts_init.hpp"
#include "arrinfo.hpp"
#include "backend.hpp"
#include "../../lcx.h"

namespace libcloudphxx { namespace lgrngn {
  namespace chem = common::chem;

  template <typename real_t>
  struct particles_proto_t
  {
    typedef std::map<enum chem::chem_species_t, const arrinfo_t<real_t>> cchem_t;
    typedef std::map<enum chem::chem_species_t, arrinfo_t<real_t>> chem_t;
    typedef arrinfo_t<real_t> arr;

    virtual void init(const arr th, const arr rv, const arr rhod, const arr p = arr(), const arr courant_x = arr(),
                      const arr courant_y = arr(), const arr courant_z = arr(), const cchem_t ambient_chem = cchem_t()) { assert(false); }
    virtual void step_sync(const opts_t<real_t> &, arr th, arr rv, const arr rhod = arr(), const arr courant_x = arr(),
                           const arr courant_y = arr(), const arr courant_z = arr(), const arr diss_rate = arr(), chem_t ambient_chem = chem_t()) { assert(false); }
    virtual void sync_in(arr th, arr rv, const arr rhod = arr(), const arr courant_x = arr(), const arr courant_y = arr(),
                         const arr courant_z = arr(), const arr diss_rate = arr(), chem_t ambient_chem = chem_t()) { assert(false); }
    virtual void step_cond(const opts_t<real_t> &, arr th, arr rv, chem_t ambient_chem = chem_t()) { assert(false); }
    virtual void step_async(const opts_t<real_t> &) { assert(false); }

    virtual void diag_sd_conc() { assert(false); }
    virtual void diag_pressure() { assert(false); }
    virtual void diag_temperature() { assert(false); }
    virtual void diag_RH() { assert(false); }
    virtual void diag_all() { assert(false); }
    virtual void diag_water() { assert(false); }
    virtual void diag_dry_rng(const real_t &, const real_t &) { assert(false); }
    virtual void diag_wet_rng(const real_t &, const real_t &) { assert(false); }
    virtual void diag_kappa_rng(const real_t &, const real_t &) { assert(false); }
    virtual void diag_dry_rng_cons(const real_t &, const real_t &) { assert(false); }
    virtual void diag_wet_rng_cons(const real_t &, const real_t &) { assert(false); }
    virtual void diag_kappa_rng_cons(const real_t &, const real_t &) { assert(false); }
    virtual void diag_dry_mom(const int &) { assert(false); }
    virtual void diag_wet_mom(const int &) { assert(false); }
    virtual void diag_vel_div() { assert(false); }
    virtual void diag_kappa_mom(const int &) { assert(false); }
    virtual void diag_incloud_time_mom(const int &) { assert(false); }
    virtual void diag_up_mom(const int &) { assert(false); }
    virtual void diag_vp_mom(const int &) { assert(false); }
    virtual void diag_wp_mom(const int &) { assert(false); }
    virtual void diag_water_cons() { assert(false); }
    // diagnostics of the parts outside this library (chemistry, ice): declared for source compatibility, never served
    virtual void diag_chem(const enum common::chem::chem_species_t &) { assert(false); }
    virtual void diag_ice() { assert(false); }
    virtual void diag_ice_cons() { assert(false); }
    virtual void diag_ice_a_rng(const real_t &, const real_t &) { assert(false); }
    virtual void diag_ice_c_rng(const real_t &, const real_t &) { assert(false); }
    virtual void diag_ice_a_rng_cons(const real_t &, const real_t &) { assert(false); }
    virtual void diag_ice_c_rng_cons(const real_t &, const real_t &) { assert(false); }
    virtual void diag_ice_a_mom(const int &) { assert(false); }
    virtual void diag_ice_c_mom(const int &) { assert(false); }
    virtual void diag_ice_mix_ratio() { assert(false); }
    virtual void diag_precip_rate_ice_mass() { assert(false); }
    virtual void diag_max_rw() { assert(false); }
    virtual void diag_precip_rate() { assert(false); }
    virtual void diag_RH_ge_Sc() { assert(false); }
    virtual void diag_rw_ge_rc() { assert(false); }
    virtual void diag_wet_mass_dens(const real_t &, const real_t &) { assert(false); }
    virtual std::map<common::output_t, real_t> diag_puddle() { assert(false); return std::map<common::output_t, real_t>(); }
    virtual std::vector<real_t> get_attr(const std::string &) { assert(false); return std::vector<real_t>(); }
    virtual real_t *outbuf() { assert(false); return nullptr; }

    opts_init_t<real_t> *opts_init = nullptr;        // points at the live internal copy (particles.hpp:129)
    virtual ~particles_proto_t() {}
  };

  namespace detail
  {
    inline void lcx_check(int rc) { if (rc) throw std::runtime_error(lcx_last_error()); }

    template <typename real_t> inline double distro_trampoline(double lnrd, void *user)
    { return double(static_cast<common::unary_function<real_t> *>(user)->funval(real_t(lnrd))); }

    template <typename real_t> struct carr
    {
      lcx_arrinfo_t a; bool null;
      explicit carr(const arrinfo_t<real_t> &x) : null(x.is_null())
      { a.data = const_cast<void *>(static_cast<const void *>(x.data)); a.strides = x.strides; a.on_device = x.on_device ? 1 : 0; }
      const lcx_arrinfo_t *ptr() const { return null ? nullptr : &a; }
    };
  }

  // generic declaration: only the HIP slots are implemented by this library
  template <typename real_t, backend_t backend> struct particles_t;

  template <typename real_t>
  struct particles_t<real_t, HIP> : particles_proto_t<real_t>
  {
    typedef particles_proto_t<real_t> parent_t;
    typedef arrinfo_t<real_t> arr;
    typedef typename parent_t::cchem_t cchem_t;
    typedef typename parent_t::chem_t chem_t;

    // pimpl kept public like in the reference (particles.hpp:229-230); it only holds the C handle and the options copy
    struct impl { lcx_particles *h = nullptr; opts_init_t<real_t> opts_init; std::vector<real_t> outbuf_copy; ~impl() { if (h) lcx_destroy(h); } };
    std::unique_ptr<impl> pimpl;

    // multi: the object spans opts_init.dev_count devices (lcx_create_multi) -- used by the multi_HIP specialisation below
    explicit particles_t(opts_init_t<real_t> oi, int n_x_tot = 0, bool multi = false) : pimpl(new impl)
    {
      if (!oi.rlx_dry_distros.empty()) throw std::runtime_error("libcloudph++: option outside the accelerated hot path (rlx)");
      pimpl->opts_init = oi;
      this->opts_init = &pimpl->opts_init;
      lcx_opts_init_t c;
      lcx_opts_init_default(&c);
      const opts_init_t<real_t> &o = pimpl->opts_init;
      c.nx = o.nx; c.ny = o.ny; c.nz = o.nz; c.dx = o.dx; c.dy = o.dy; c.dz = o.dz; c.dt = o.dt;
      c.sstp_cond = o.sstp_cond; c.sstp_coal = o.sstp_coal; c.sstp_cond_act = o.sstp_cond_act; c.sstp_chem = o.sstp_chem;
      c.x0 = o.x0; c.y0 = o.y0; c.z0 = o.z0; c.x1 = o.x1; c.y1 = o.y1; c.z1 = o.z1;
      c.sd_conc = o.sd_conc; c.sd_conc_large_tail = o.sd_conc_large_tail; c.aerosol_independent_of_rhod = o.aerosol_independent_of_rhod;
      c.variable_dt_switch = o.variable_dt_switch; c.sd_const_multi = o.sd_const_multi; c.n_sd_max = o.n_sd_max;
      c.kernel = int(o.kernel); c.terminal_velocity = int(o.terminal_velocity); c.adve_scheme = int(o.adve_scheme); c.RH_formula = int(o.RH_formula);
      std::vector<double> kp(o.kernel_parameters.begin(), o.kernel_parameters.end()), wls(o.w_LS.begin(), o.w_LS.end()),
                          acf(o.aerosol_conc_factor.begin(), o.aerosol_conc_factor.end()), sgs(o.SGS_mix_len.begin(), o.SGS_mix_len.end());
      c.SGS_mix_len = sgs.data(); c.n_SGS_mix_len = int(sgs.size());
      c.kernel_parameters = kp.data(); c.n_kernel_parameters = int(kp.size());
      c.w_LS = wls.data(); c.n_w_LS = int(wls.size());
      c.aerosol_conc_factor = acf.data(); c.n_aerosol_conc_factor = int(acf.size());
      c.chem_switch = o.chem_switch; c.coal_switch = o.coal_switch; c.sedi_switch = o.sedi_switch; c.subs_switch = o.subs_switch;
      c.rlx_switch = o.rlx_switch; c.turb_adve_switch = o.turb_adve_switch; c.turb_cond_switch = o.turb_cond_switch;
      c.turb_coal_switch = o.turb_coal_switch; c.ice_switch = o.ice_switch; c.exact_sstp_cond = o.exact_sstp_cond;
      c.sstp_cond_mix = o.sstp_cond_mix; c.adaptive_sstp_cond = o.adaptive_sstp_cond; c.time_dep_ice_nucl = o.time_dep_ice_nucl;
      c.RH_max = o.RH_max; c.sstp_cond_adapt_drw2_eps = o.sstp_cond_adapt_drw2_eps; c.sstp_cond_adapt_drw2_max = o.sstp_cond_adapt_drw2_max; c.rc2_T = o.rc2_T;
      c.rng_seed = o.rng_seed; c.rng_seed_init = o.rng_seed_init; c.rng_seed_init_switch = o.rng_seed_init_switch;
      c.dev_count = o.dev_count; c.dev_id = o.dev_id; c.rd_min = o.rd_min; c.rd_max = o.rd_max;
      c.no_ccn_at_init = o.no_ccn_at_init; c.open_side_walls = o.open_side_walls; c.periodic_topbot_walls = o.periodic_topbot_walls;
      c.src_type = int(o.src_type); c.th_dry = o.th_dry; c.const_p = o.const_p; c.diag_incloud_time = o.diag_incloud_time;
      c.n_x_tot = n_x_tot; c.strict_fp = o.strict_fp; c.cond_solver = o.cond_solver; c.reorder_every = o.reorder_every; c.stream_ordered = o.stream_ordered;
      c.dbg_flags = o.dbg_flags;
      c.n_x_bfr = o.n_x_bfr; c.bcond_lft = o.bcond_lft; c.bcond_rgt = o.bcond_rgt;
      // std::map iterates in (kappa, rd_insol) order, which is the order the library expects
      std::vector<lcx_distro_t> dd;
      for (const auto &kv : pimpl->opts_init.dry_distros) {
        lcx_distro_t d{};
        d.kappa = kv.first.kappa; d.rd_insol = kv.first.rd_insol;
        d.fn = &detail::distro_trampoline<real_t>; d.user = kv.second.get();     // shared_ptr kept alive by pimpl->opts_init
        dd.push_back(d);
      }
      c.dry_distros = dd.data(); c.n_dry_distros = int(dd.size());
      std::vector<lcx_dry_size_t> ds;
      for (const auto &kv : o.dry_sizes) for (const auto &rc : kv.second)
        ds.push_back(lcx_dry_size_t{double(kv.first.kappa), double(kv.first.rd_insol), double(rc.first), double(rc.second.first), rc.second.second});
      c.dry_sizes = ds.data(); c.n_dry_sizes = int(ds.size());
      detail::lcx_check(multi ? lcx_create_multi(&c, int(sizeof(real_t)), &pimpl->h) : lcx_create(&c, int(sizeof(real_t)), &pimpl->h));
    }
    ~particles_t() override {}

    void init(const arr th, const arr rv, const arr rhod, const arr p = arr(), const arr courant_x = arr(), const arr courant_y = arr(),
              const arr courant_z = arr(), const cchem_t ambient_chem = cchem_t()) override
    {
      if (!ambient_chem.empty()) throw std::runtime_error("libcloudph++: chemistry was switched off and ambient_chem is not empty");
      detail::carr<real_t> a(th), b(rv), c(rhod), d(p), e(courant_x), f(courant_y), g(courant_z);
      detail::lcx_check(lcx_init(pimpl->h, a.ptr(), b.ptr(), c.ptr(), d.ptr(), e.ptr(), f.ptr(), g.ptr()));
    }
    void sync_in(arr th, arr rv, const arr rhod = arr(), const arr courant_x = arr(), const arr courant_y = arr(), const arr courant_z = arr(),
                 const arr diss_rate = arr(), chem_t ambient_chem = chem_t()) override
    {
      if (!ambient_chem.empty()) throw std::runtime_error("libcloudph++: chemistry was switched off and ambient_chem is not empty");
      detail::carr<real_t> a(th), b(rv), c(rhod), e(courant_x), f(courant_y), g(courant_z), d(diss_rate);
      detail::lcx_check(lcx_sync_in(pimpl->h, a.ptr(), b.ptr(), c.ptr(), e.ptr(), f.ptr(), g.ptr(), d.ptr()));
    }
    void step_cond(const opts_t<real_t> &opts, arr th, arr rv, chem_t ambient_chem = chem_t()) override
    {
      (void)ambient_chem;
      const lcx_opts_t oc = conv(opts);
      detail::carr<real_t> a(th), b(rv);
      detail::lcx_check(lcx_step_cond(pimpl->h, &oc, a.ptr(), b.ptr()));
    }
    void step_sync(const opts_t<real_t> &opts, arr th, arr rv, const arr rhod = arr(), const arr courant_x = arr(), const arr courant_y = arr(),
                   const arr courant_z = arr(), const arr diss_rate = arr(), chem_t ambient_chem = chem_t()) override
    {
      sync_in(th, rv, rhod, courant_x, courant_y, courant_z, diss_rate, ambient_chem);
      step_cond(opts, th, rv, ambient_chem);
    }
    void step_async(const opts_t<real_t> &opts) override
    {
      if (!opts.src_dry_distros.empty() || !opts.src_dry_sizes.empty()) throw std::runtime_error("libcloudph++: aerosol source was switched off in opts_init");
      const lcx_opts_t oc = conv(opts);
      detail::lcx_check(lcx_step_async(pimpl->h, &oc));
    }

    void diag_sd_conc() override { detail::lcx_check(lcx_diag_sd_conc(pimpl->h)); }
    void diag_pressure() override { detail::lcx_check(lcx_diag_pressure(pimpl->h)); }
    void diag_temperature() override { detail::lcx_check(lcx_diag_temperature(pimpl->h)); }
    void diag_RH() override { detail::lcx_check(lcx_diag_RH(pimpl->h)); }
    void diag_all() override { detail::lcx_check(lcx_diag_all(pimpl->h)); }
    void diag_water() override { detail::lcx_check(lcx_diag_water(pimpl->h)); }
    void diag_dry_rng(const real_t &a, const real_t &b) override { detail::lcx_check(lcx_diag_dry_rng(pimpl->h, a, b)); }
    void diag_wet_rng(const real_t &a, const real_t &b) override { detail::lcx_check(lcx_diag_wet_rng(pimpl->h, a, b)); }
    void diag_kappa_rng(const real_t &a, const real_t &b) override { detail::lcx_check(lcx_diag_kappa_rng(pimpl->h, a, b)); }
    void diag_dry_rng_cons(const real_t &a, const real_t &b) override { detail::lcx_check(lcx_diag_dry_rng_cons(pimpl->h, a, b)); }
    void diag_wet_rng_cons(const real_t &a, const real_t &b) override { detail::lcx_check(lcx_diag_wet_rng_cons(pimpl->h, a, b)); }
    void diag_kappa_rng_cons(const real_t &a, const real_t &b) override { detail::lcx_check(lcx_diag_kappa_rng_cons(pimpl->h, a, b)); }
    void diag_dry_mom(const int &k) override { detail::lcx_check(lcx_diag_dry_mom(pimpl->h, k)); }
    void diag_wet_mom(const int &k) override { detail::lcx_check(lcx_diag_wet_mom(pimpl->h, k)); }
    void diag_vel_div() override { detail::lcx_check(lcx_diag_vel_div(pimpl->h)); }
    void diag_kappa_mom(const int &k) override { detail::lcx_check(lcx_diag_kappa_mom(pimpl->h, k)); }
    void diag_incloud_time_mom(const int &k) override { detail::lcx_check(lcx_diag_incloud_time_mom(pimpl->h, k)); }
    void diag_up_mom(const int &k) override { detail::lcx_check(lcx_diag_up_mom(pimpl->h, k)); }
    void diag_vp_mom(const int &k) override { detail::lcx_check(lcx_diag_vp_mom(pimpl->h, k)); }
    void diag_wp_mom(const int &k) override { detail::lcx_check(lcx_diag_wp_mom(pimpl->h, k)); }
    void diag_water_cons() override { detail::lcx_check(lcx_diag_water_cons(pimpl->h)); }
    void diag_max_rw() override { detail::lcx_check(lcx_diag_max_rw(pimpl->h)); }
    void diag_precip_rate() override { detail::lcx_check(lcx_diag_precip_rate(pimpl->h)); }
    void diag_RH_ge_Sc() override { detail::lcx_check(lcx_diag_RH_ge_Sc(pimpl->h)); }
    void diag_rw_ge_rc() override { detail::lcx_check(lcx_diag_rw_ge_rc(pimpl->h)); }
    void diag_wet_mass_dens(const real_t &rad, const real_t &sig0) override { detail::lcx_check(lcx_diag_wet_mass_dens(pimpl->h, rad, sig0)); }
    std::map<common::output_t, real_t> diag_puddle() override
    {
      double v[LCX_OUT_COUNT];
      detail::lcx_check(lcx_diag_puddle(pimpl->h, v));
      std::map<common::output_t, real_t> m;
      for (int i = 0; i < LCX_OUT_COUNT; ++i) m[static_cast<common::output_t>(i)] = real_t(v[i]);
      return m;
    }
    std::vector<real_t> get_attr(const std::string &name) override
    {
      size_t n = 0;
      detail::lcx_check(lcx_get_attr(pimpl->h, name.c_str(), nullptr, 0, &n));
      std::vector<real_t> out(n);
      detail::lcx_check(lcx_get_attr(pimpl->h, name.c_str(), out.data(), n, &n));
      return out;
    }
    real_t *outbuf() override
    {
      const void *p = nullptr; size_t n = 0;
      detail::lcx_check(lcx_outbuf(pimpl->h, &p, &n));
      return const_cast<real_t *>(static_cast<const real_t *>(p));
    }

  private:
    static lcx_opts_t conv(const opts_t<real_t> &o)
    {
      lcx_opts_t c;
      lcx_opts_default(&c);
      c.adve = o.adve; c.sedi = o.sedi; c.subs = o.subs; c.cond = o.cond; c.coal = o.coal; c.src = o.src; c.rlx = o.rlx; c.rcyc = o.rcyc;
      c.turb_adve = o.turb_adve; c.turb_cond = o.turb_cond; c.turb_coal = o.turb_coal; c.ice_nucl = o.ice_nucl;
      c.chem_dsl = o.chem_dsl; c.chem_dsc = o.chem_dsc; c.chem_rct = o.chem_rct; c.RH_max = o.RH_max; c.dt = o.dt;
      return c;
    }
  };

  // particles_t<real_t, multi_HIP>: one object over all (or opts_init.dev_count) devices of the process, the reference's
  // particles_t<real_t, multi_CUDA> (reference: lgrngn/particles.hpp:246-340, src/particles_multi_gpu_*.ipp).  Same virtuals;
  // arrays are the GLOBAL ones, outbuf() the global field, get_attr() throws (particles_multi_gpu_ctor.ipp:53-57), opts.rcyc is
  // refused (particles_multi_gpu_step.ipp:63-65).  The slabs, worker threads and the device-driven exchange live inside the
  // library (libcloudphxx_amd/csrc/lcx_multi.hpp).
  template <typename real_t>
  struct particles_t<real_t, multi_HIP> : particles_t<real_t, HIP>
  {
    explicit particles_t(opts_init_t<real_t> oi) : particles_t<real_t, HIP>(oi, 0, true)
    {
      int n = 0;
      detail::lcx_check(lcx_multi_dev_count(this->pimpl->h, &n));
      this->pimpl->opts_init.dev_count = n;              // the reference stores the actual count in the live copy
    }
  };
  // the reference's own backend names: a driver that instantiates or dynamic_casts to them gets the HIP objects
  template <typename real_t>
  struct particles_t<real_t, CUDA> : particles_t<real_t, HIP>
  { explicit particles_t(opts_init_t<real_t> oi, int n_x_tot = 0) : particles_t<real_t, HIP>(oi, n_x_tot) {} };
  template <typename real_t>
  struct particles_t<real_t, multi_CUDA> : particles_t<real_t, multi_HIP>
  { explicit particles_t(opts_init_t<real_t> oi) : particles_t<real_t, multi_HIP>(oi) {} };
} }
