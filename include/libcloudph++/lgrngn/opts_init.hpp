// Construction-time options -- field names, types and defaults of the reference's opts_init_t
// (reference: lgrngn/opts_init.hpp:29-253), written with default member initialisers.
// Everything is forwarded to lcx_opts_init_t (include/lcx.h).  Served: per-cell and per-particle condensation
// substepping (exact_sstp_cond, sstp_cond_mix, adaptive_sstp_cond, sstp_cond_act), the SGS turbulence switches
// (turb_adve / turb_cond / turb_coal), all initialisation modes, pred_corr advection, open side walls.  Switches of
// sub-systems outside the accelerated path (chemistry, ice, aerosol sources, relaxation) make the constructor throw
// instead of being ignored.
#pragma once
#include "kernel.hpp"
#include "terminal_velocity.hpp"
#include "advection_scheme.hpp"
#include "RH_formula.hpp"
#include "ccn_source.hpp"
#include "distro_t.hpp"
namespace libcloudphxx { namespace lgrngn {
  enum class INP_t { mineral };
  template <typename real_t>
  struct opts_init_t
  {
    // aerosol
    dry_distros_t<real_t> dry_distros;
    dry_sizes_t<real_t> dry_sizes;
    unsigned long long sd_conc = 0, sd_const_multi = 0, n_sd_max = 0;
    bool sd_conc_large_tail = false, aerosol_independent_of_rhod = false, no_ccn_at_init = false;
    std::vector<real_t> aerosol_conc_factor;
    real_t rd_min = -1, rd_max = -1;
    // grid, domain, time
    int nx = 0, ny = 0, nz = 0;
    real_t dx = 1, dy = 1, dz = 1, dt = 0;
    real_t x0 = 0, y0 = 0, z0 = 0, x1 = 1, y1 = 1, z1 = 1;
    int sstp_cond = 1, sstp_coal = 1, sstp_cond_act = 1, sstp_chem = 1;
    bool variable_dt_switch = false;
    bool open_side_walls = false, periodic_topbot_walls = false;
    // physics choices
    kernel_t kernel = kernel_t::undefined;
    std::vector<real_t> kernel_parameters;
    vt_t terminal_velocity = vt_t::undefined;
    as_t adve_scheme = as_t::implicit;
    RH_formula_t RH_formula = RH_formula_t::pv_cc;
    real_t RH_max = real_t(.95);
    bool th_dry = true, const_p = false;
    std::vector<real_t> w_LS, SGS_mix_len;
    // process switches
    bool chem_switch = false, coal_switch = true, sedi_switch = true, subs_switch = false, rlx_switch = false,
         turb_adve_switch = false, turb_cond_switch = false, turb_coal_switch = false, ice_switch = false,
         exact_sstp_cond = false, sstp_cond_mix = true, adaptive_sstp_cond = false, time_dep_ice_nucl = false,
         diag_incloud_time = false;
    real_t sstp_cond_adapt_drw2_eps = real_t(1e-4), sstp_cond_adapt_drw2_max = 4, rc2_T = 10, chem_rho = 0;
    INP_t inp_type = INP_t::mineral;
    // random numbers, devices
    int rng_seed = 44, rng_seed_init = 44;
    bool rng_seed_init_switch = false;
    int dev_count = 0, dev_id = -1;
    // sources / relaxation (accepted for source compatibility; must stay off)
    src_t src_type = src_t::off;
    real_t src_x0 = 0, src_y0 = 0, src_z0 = 0, src_x1 = 0, src_y1 = 0, src_z1 = 0;
    typedef std::unordered_map<real_t, std::tuple<std::shared_ptr<unary_function<real_t>>, std::pair<real_t, real_t>, std::pair<real_t, real_t>>> rlx_dry_distros_t;
    rlx_dry_distros_t rlx_dry_distros;
    unsigned long long rlx_bins = 0;
    real_t rlx_sd_per_bin = 0, rlx_timescale = 1;
    int supstp_rlx = 1;
    // --- extensions of this backend (no reference counterpart, see include/lcx.h) ---
    bool strict_fp = false;    // false (default since round 5): contracted one-division form of the condensational growth rate; true: IEEE operator order
    int cond_solver = 1;       // fast arithmetic only: 1 the reference's TOMS748 iterates (default since round 5), 0 lean bracketed secant (lcx.h)
    // a slab of a 1-D decomposed domain (what detail::distmem_opts derives from the MPI rank inside the reference, distmem_opts.hpp:20-52):
    // x-planes owned by the ranks to the left, and the kind of the two x-faces (0 this process owns the whole domain, 1 neighbour
    // slab: leaving SDs are listed for lcx_migrate_pack, 3 open wall)
    int n_x_bfr = 0, bcond_lft = 0, bcond_rgt = 0;
    unsigned dbg_flags = 0;    // test / measurement switches (lcx.h, enum lcx_dbg): all off in production
    int reorder_every = 0;     // storage re-ordering into the cell-sorted order: every N steps and with every compaction (0: N = 64), -1 never (lcx.h)
    bool stream_ordered = false; // device arrays: step_sync returns once its work is queued on the object's stream (lcx.h)
  };
} }
