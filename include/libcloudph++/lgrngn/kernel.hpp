#pragma once   // enumerators and order as reference lgrngn/kernel.hpp:8 == enum lcx_kernel
namespace libcloudphxx { namespace lgrngn {
  enum class kernel_t { undefined, geometric, golovin, hall, hall_davis_no_waals, Long, onishi_hall, onishi_hall_davis_no_waals,
                        hall_pinsky_1000mb_grav, hall_pinsky_cumulonimbus, hall_pinsky_stratocumulus, vohl_davis_no_waals };
} }
