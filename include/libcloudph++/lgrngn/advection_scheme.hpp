#pragma once   // reference lgrngn/advection_scheme.hpp:8 == enum lcx_adve; as_name as reference advection_scheme.hpp:10-15
#include "enum_names.hpp"
namespace libcloudphxx { namespace lgrngn {
  enum class as_t { undefined, implicit, euler, pred_corr };
  const std::unordered_map<as_t, std::string> as_name = detail::enum_names<as_t>({"undefined", "implicit", "euler", "pred_corr"});
} }
