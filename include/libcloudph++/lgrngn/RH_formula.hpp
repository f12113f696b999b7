#pragma once   // reference lgrngn/RH_formula.hpp:8-12 == enum lcx_rh; RH_formula_name as reference RH_formula.hpp:14-19
#include "enum_names.hpp"
namespace libcloudphxx { namespace lgrngn {
  // pv_cc / rv_cc: RH = pv / pvs resp. rv / rvs, saturation from the Clausius-Clapeyron equation; *_tet: from the Tetens formula
  enum class RH_formula_t { pv_cc, rv_cc, pv_tet, rv_tet };
  const std::unordered_map<RH_formula_t, std::string> RH_formula_name = detail::enum_names<RH_formula_t>({"pv_cc", "rv_cc", "pv_tet", "rv_tet"});
} }
