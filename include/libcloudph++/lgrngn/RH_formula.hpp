#pragma once   // reference lgrngn/RH_formula.hpp:8-12 == enum lcx_rh
namespace libcloudphxx { namespace lgrngn { enum class RH_formula_t { pv_cc, rv_cc, pv_tet, rv_tet }; } }
