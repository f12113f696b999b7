// Per-call switches -- field names and defaults of the reference's opts_t (reference: lgrngn/opts.hpp:20-50).
#pragma once
#include "distro_t.hpp"
namespace libcloudphxx { namespace lgrngn {
  template <typename real_t>
  struct opts_t
  {
    bool adve = true, sedi = true, subs = false, cond = true, coal = true, src = false, rlx = false, rcyc = false,
         turb_adve = false, turb_cond = false, turb_coal = false, ice_nucl = false;
    real_t RH_max = 44;                         // anything above 1.1 means "no limit"
    bool chem_dsl = false, chem_dsc = false, chem_rct = false;
    real_t dt = -1;                             // < 0: use opts_init.dt
    src_dry_distros_t<real_t> src_dry_distros;  // aerosol sources: not part of the accelerated path
    src_dry_sizes_t<real_t> src_dry_sizes;
  };
} }
