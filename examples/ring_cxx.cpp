// Multi-GPU from C++, exactly as a driver written for the reference does it: factory<real_t>(multi_CUDA, opts_init) returns ONE
// object that spans the devices of the process (here opts_init.dev_count = 2 slabs; on a one-GPU box run it with
// LCX_MULTI_DEVICE_MAP=0,0 so that both slabs share the device).  The set-up of the reference's tests/mpi/mpi_adve_test.cpp:130-255:
// Courant number 1 in x, so that after nx steps every super-droplet has travelled once around the periodic domain -- through both
// slab faces -- and the per-cell diagnostics are bit for bit what they were.
#include <cstdio>
#include <cmath>
#include <memory>
#include <vector>
#include <libcloudph++/lgrngn/factory.hpp>

using namespace libcloudphxx::lgrngn;
typedef double real_t;

struct lognormal : libcloudphxx::common::unary_function<real_t>
{
  real_t funval(const real_t lnr) const override
  { return 60e6 * std::exp(-std::pow((lnr - std::log(.02e-6)), 2) / 2 / std::pow(std::log(1.4), 2)) / std::log(1.4) / std::sqrt(2 * M_PI); }
};

int main()
{
  const int nx = 6, nz = 4;
  opts_init_t<real_t> oi;
  oi.dry_distros.emplace(kappa_rd_insol_t<real_t>(.61, 0.), std::make_shared<lognormal>());
  oi.coal_switch = oi.sedi_switch = false;
  oi.dt = 1; oi.nx = nx; oi.nz = nz; oi.dx = oi.dz = 1; oi.x1 = nx; oi.z1 = nz; oi.sd_conc = 8; oi.n_sd_max = 8 * nx * nz * 4;
  oi.dev_count = 2;
  std::unique_ptr<particles_proto_t<real_t>> prtcls(factory<real_t>(multi_CUDA, oi));
  // GLOBAL arrays: every device reads and writes its own x-planes of them
  std::vector<real_t> th(nx * nz, 300.), rv(nx * nz, .01), rhod(nx * nz, 1.), Cx((nx + 1) * nz, 1.), Cz(nx * (nz + 1), 0.);
  const std::vector<ptrdiff_t> s{nz, 1}, sz{nz + 1, 1};
  auto ai = [](std::vector<real_t> &v, const std::vector<ptrdiff_t> &st) { return arrinfo_t<real_t>(v.data(), st); };
  prtcls->init(ai(th, s), ai(rv, s), ai(rhod, s), arrinfo_t<real_t>(), ai(Cx, s), arrinfo_t<real_t>(), ai(Cz, sz));
  auto conc = [&]() {
    prtcls->diag_all(); prtcls->diag_sd_conc();
    const real_t *o = prtcls->outbuf();
    return std::vector<real_t>(o, o + nx * nz);
  };
  const std::vector<real_t> before = conc();
  opts_t<real_t> opts; opts.cond = opts.coal = opts.sedi = false;
  for (int step = 0; step < nx; ++step) {
    prtcls->step_sync(opts, ai(th, s), ai(rv, s), ai(rhod, s));
    prtcls->step_async(opts);
  }
  const std::vector<real_t> after = conc();
  const int same = before == after;
  double tot = 0; for (real_t v : after) tot += v;
  bool threw = false;
  try { prtcls->get_attr("rw2"); } catch (const std::runtime_error &) { threw = true; }      // as the reference's multi_CUDA
  std::printf("ring round_trip_identical %d sd_total %g dev_count %d get_attr_throws %d\n", same, tot, prtcls->opts_init->dev_count, int(threw));
  return same && threw ? 0 : 1;
}
