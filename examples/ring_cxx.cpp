// The 1-D decomposition from C++: two slabs of one periodic domain held by two particle objects (one per GPU in a real run,
// both on one device here), super-droplets that cross a slab face handed over with the C ABI's migration calls
// (lcx_migrate_counts / _pack / _unpack / _finish, include/lcx.h) through device buffers -- what an MPI or RCCL client does
// with its own transport in between.  The set-up of the reference's tests/mpi/mpi_adve_test.cpp:130-255: Courant number 1,
// so after nx steps every super-droplet is back in its cell and the per-cell diagnostics are what they were.
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
static void check(int rc) { if (rc) throw std::runtime_error(lcx_last_error()); }

int main()
{
  const int nx_tot = 6, nz = 4, size = 2, nx = nx_tot / size;
  std::vector<std::unique_ptr<particles_t<real_t, HIP>>> slab;
  for (int r = 0; r < size; ++r) {
    opts_init_t<real_t> oi;
    oi.dry_distros.emplace(kappa_rd_insol_t<real_t>(.61, 0.), std::make_shared<lognormal>());
    oi.coal_switch = oi.sedi_switch = false;
    oi.dt = 1; oi.nx = nx; oi.nz = nz; oi.dx = oi.dz = 1; oi.x1 = nx; oi.z1 = nz; oi.sd_conc = 8; oi.n_sd_max = 8 * nx * nz * 3;
    oi.rng_seed = 44 + r;
    oi.bcond_lft = oi.bcond_rgt = 1;                       // both x-faces lead to a neighbour slab
    slab.emplace_back(new particles_t<real_t, HIP>(oi, nx));
  }
  std::vector<real_t> th(nx * nz, 300.), rv(nx * nz, .01), rhod(nx * nz, 1.), Cx((nx + 1) * nz, 1.), Cz(nx * (nz + 1), 0.);
  const std::vector<ptrdiff_t> s{nz, 1}, sz{nz + 1, 1};
  auto ai = [](std::vector<real_t> &v, const std::vector<ptrdiff_t> &st) { return arrinfo_t<real_t>(v.data(), st); };
  for (auto &p : slab) p->init(ai(th, s), ai(rv, s), ai(rhod, s), arrinfo_t<real_t>(), ai(Cx, s), arrinfo_t<real_t>(), ai(Cz, sz));
  auto conc = [&]() {
    std::vector<real_t> all;
    for (auto &p : slab) { p->diag_all(); p->diag_sd_conc(); real_t *o = p->outbuf(); all.insert(all.end(), o, o + nx * nz); }
    return all;
  };
  const std::vector<real_t> before = conc();
  opts_t<real_t> opts; opts.cond = opts.coal = opts.sedi = false;
  size_t moved = 0;
  for (int step = 0; step < nx_tot; ++step) {
    for (auto &p : slab) { p->step_sync(opts, ai(th, s), ai(rv, s), ai(rhod, s)); p->step_async(opts); }
    // pack on every slab, then unpack on the neighbours, then finish (re-index + re-sort)
    struct msg { size_t n[2]; void *buf[2]; };
    std::vector<msg> out(size);
    lcx_opts_t copts; lcx_opts_default(&copts); copts.cond = copts.coal = copts.sedi = 0;
    for (int r = 0; r < size; ++r) {
      lcx_particles *h = slab[r]->pimpl->h;
      check(lcx_migrate_counts(h, &out[r].n[0], &out[r].n[1]));
      const size_t rec = lcx_migrate_record_bytes(h);
      for (int side = 0; side < 2; ++side) {
        check(lcx_dev_alloc(&out[r].buf[side], out[r].n[side] * rec + 8));
        // the receiver's edge in ITS frame: left neighbour's x1, right neighbour's x0
        if (out[r].n[side]) check(lcx_migrate_pack(h, side, side == 0 ? real_t(nx) : real_t(0), out[r].buf[side], out[r].n[side] * rec));
        moved += out[r].n[side];
      }
    }
    for (int r = 0; r < size; ++r) {
      lcx_particles *h = slab[r]->pimpl->h;
      const int lft = (r + size - 1) % size, rgt = (r + 1) % size;
      if (out[lft].n[1]) check(lcx_migrate_unpack(h, out[lft].buf[1], out[lft].n[1]));      // left neighbour's right-going
      if (out[rgt].n[0]) check(lcx_migrate_unpack(h, out[rgt].buf[0], out[rgt].n[0]));      // right neighbour's left-going
    }
    for (int r = 0; r < size; ++r) check(lcx_migrate_finish(slab[r]->pimpl->h, &copts));
    check(lcx_dev_sync());
    for (auto &m : out) for (void *b : m.buf) check(lcx_dev_free(b));
  }
  const std::vector<real_t> after = conc();
  int same = before == after;
  double tot = 0; for (real_t v : after) tot += v;
  std::printf("ring round_trip_identical %d sd_total %g migrants %zu\n", same, tot, moved);
  return same ? 0 : 1;
}
