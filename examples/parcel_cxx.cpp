// A driver written against the reference's C++ interface (factory / opts_init_t / arrinfo_t / step_sync /
// step_async / diag_* / outbuf), compiled against include/libcloudph++ of THIS repository and linked with
// liblcx_hip.so.  It mirrors the reference's 0-D parcel set-up (tests/python/physics/lgrngn_cond.py) and a small
// 2-D box with advection and sedimentation, and prints a few invariants that tests/test_cxx_api.py checks.
#include <cmath>
#include <cstdio>
#include <memory>
#include <libcloudph++/lgrngn/factory.hpp>

using namespace libcloudphxx::lgrngn;
typedef double real_t;

struct lognormal : libcloudphxx::common::unary_function<real_t>
{
  real_t mean_r, stdev, n_tot;
  lognormal(real_t m, real_t s, real_t n) : mean_r(m), stdev(s), n_tot(n) {}
  real_t funval(const real_t lnr) const override
  { return n_tot * std::exp(-std::pow((lnr - std::log(mean_r)), 2) / 2 / std::pow(std::log(stdev), 2)) / std::log(stdev) / std::sqrt(2 * M_PI); }
};

int main()
{
  // ---- a driver that prints its options (as UWLCM / icicle do at start-up) needs the `<enum>_name` tables of the option headers
  {
    opts_init_t<real_t> oi;
    std::printf("options backend=%s kernel=%s vt=%s adve=%s RH=%s src=%s n_kernels=%zu\n", backend_name.at(HIP).c_str(),
                kernel_name.at(kernel_t::hall_pinsky_stratocumulus).c_str(), vt_name.at(vt_t::beard77fast).c_str(),
                as_name.at(oi.adve_scheme).c_str(), RH_formula_name.at(oi.RH_formula).c_str(), src_name.at(src_t::off).c_str(),
                kernel_name.size());
  }
  // ---- 0-D parcel: condensation only
  {
    opts_init_t<real_t> oi;
    oi.dry_distros.emplace(kappa_rd_insol_t<real_t>(.61, 0.), std::make_shared<lognormal>(.04e-6 / 2, 1.4, 60e6));
    oi.coal_switch = oi.sedi_switch = false;
    oi.RH_max = 0.999; oi.dt = 1; oi.sd_conc = 100; oi.n_sd_max = 100;
    std::unique_ptr<particles_proto_t<real_t>> prtcls(factory<real_t>(HIP, oi));
    real_t th = 300, rv = 0.02, rhod = 1;
    const ptrdiff_t strides[] = {1};
    prtcls->init(arrinfo_t<real_t>(&th, strides), arrinfo_t<real_t>(&rv, strides), arrinfo_t<real_t>(&rhod, strides));
    bool threw = false;
    try { opts_t<real_t> o; prtcls->step_async(o); } catch (const std::runtime_error &) { threw = true; }
    std::printf("call_order_exception %d\n", int(threw));
    opts_t<real_t> opts; opts.adve = opts.sedi = opts.coal = false; opts.cond = false;
    for (int step = 0; step < 40; ++step) {
      prtcls->step_sync(opts, arrinfo_t<real_t>(&th, strides), arrinfo_t<real_t>(&rv, strides), arrinfo_t<real_t>(&rhod, strides));
      prtcls->step_async(opts);
      opts.cond = true;
    }
    prtcls->diag_all(); prtcls->diag_sd_conc();
    std::printf("parcel th %.10g rv %.10g sd_conc %g\n", th, rv, prtcls->outbuf()[0]);
  }
  // ---- 2-D box: advection by one cell per step + everything else off
  {
    opts_init_t<real_t> oi;
    oi.dry_distros.emplace(kappa_rd_insol_t<real_t>(.61, 0.), std::make_shared<lognormal>(.04e-6 / 2, 1.4, 60e6));
    oi.coal_switch = oi.sedi_switch = false;
    oi.dt = 1; oi.nx = 6; oi.nz = 5; oi.dx = oi.dz = 1; oi.x1 = 6; oi.z1 = 5; oi.sd_conc = 10; oi.n_sd_max = 300;
    std::unique_ptr<particles_proto_t<real_t>> prtcls(factory<real_t>(HIP, oi));
    std::vector<real_t> th(30, 300.), rv(30, .01), rhod(30, 1.), Cx(35, 1.), Cz(36, 0.);
    const std::vector<ptrdiff_t> s{5, 1}, sz{6, 1};
    prtcls->init(arrinfo_t<real_t>(th.data(), s), arrinfo_t<real_t>(rv.data(), s), arrinfo_t<real_t>(rhod.data(), s), arrinfo_t<real_t>(),
                 arrinfo_t<real_t>(Cx.data(), s), arrinfo_t<real_t>(), arrinfo_t<real_t>(Cz.data(), sz));
    opts_t<real_t> opts; opts.cond = opts.coal = opts.sedi = false;
    double tot = 0;
    for (int step = 0; step < 6; ++step) {
      prtcls->step_sync(opts, arrinfo_t<real_t>(th.data(), s), arrinfo_t<real_t>(rv.data(), s), arrinfo_t<real_t>(rhod.data(), s));
      prtcls->step_async(opts);
    }
    prtcls->diag_all(); prtcls->diag_sd_conc();
    const real_t *out = prtcls->outbuf();
    for (int c = 0; c < 30; ++c) tot += out[c];
    std::printf("box total_sd %g n_attr %zu\n", tot, prtcls->get_attr("rw2").size());
  }
  return 0;
}
