// Aerosol spectrum containers with the reference's types (reference: lgrngn/distro_t.hpp:10-58).
#pragma once
#include "extincl.hpp"
namespace libcloudphxx { namespace lgrngn {
  using common::unary_function;
  template <typename real_t> struct kappa_rd_insol_t
  {
    real_t kappa, rd_insol;
    kappa_rd_insol_t(real_t k, real_t r) : kappa(k), rd_insol(r) {}
    bool operator<(const kappa_rd_insol_t &o) const { return kappa != o.kappa ? kappa < o.kappa : rd_insol < o.rd_insol; }
  };
  template <typename real_t> using dry_distros_t = std::map<kappa_rd_insol_t<real_t>, std::shared_ptr<unary_function<real_t>>>;
  template <typename real_t> using dry_sizes_t = std::map<kappa_rd_insol_t<real_t>, std::map<real_t, std::pair<real_t, int>>>;
  template <typename real_t> using src_dry_distros_t = std::map<kappa_rd_insol_t<real_t>, std::tuple<std::shared_ptr<unary_function<real_t>>, int, int>>;
  template <typename real_t> using src_dry_sizes_t = std::map<kappa_rd_insol_t<real_t>, std::map<real_t, std::tuple<real_t, int, int>>>;
} }
