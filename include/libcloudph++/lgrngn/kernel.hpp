#pragma once   // enumerators and order as reference lgrngn/kernel.hpp:8 == enum lcx_kernel; kernel_name as reference kernel.hpp:11-24
#include "enum_names.hpp"
namespace libcloudphxx { namespace lgrngn {
  enum class kernel_t { undefined, geometric, golovin, hall, hall_davis_no_waals, Long, onishi_hall, onishi_hall_davis_no_waals,
                        hall_pinsky_1000mb_grav, hall_pinsky_cumulonimbus, hall_pinsky_stratocumulus, vohl_davis_no_waals };
  const std::unordered_map<kernel_t, std::string> kernel_name = detail::enum_names<kernel_t>(
    {"undefined", "geometric", "golovin", "hall", "hall_davis_no_waals", "Long", "onishi_hall", "onishi_hall_davis_no_waals",
     "hall_pinsky_1000mb_grav", "hall_pinsky_cumulonimbus", "hall_pinsky_stratocumulus", "vohl_davis_no_waals"});
} }
