#pragma once   // reference lgrngn/terminal_velocity.hpp:8 == enum lcx_vt
namespace libcloudphxx { namespace lgrngn {
  enum class vt_t { undefined, beard76, beard77, beard77fast, khvorostyanov_spherical, khvorostyanov_nonspherical };
} }
