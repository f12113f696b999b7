#pragma once   // reference lgrngn/terminal_velocity.hpp:8 == enum lcx_vt; vt_name as reference terminal_velocity.hpp:10-17
#include "enum_names.hpp"
namespace libcloudphxx { namespace lgrngn {
  enum class vt_t { undefined, beard76, beard77, beard77fast, khvorostyanov_spherical, khvorostyanov_nonspherical };
  const std::unordered_map<vt_t, std::string> vt_name = detail::enum_names<vt_t>(
    {"undefined", "beard76", "beard77", "beard77fast", "khvorostyanov_spherical", "khvorostyanov_nonspherical"});
} }
