#pragma once   // reference lgrngn/advection_scheme.hpp:8 == enum lcx_adve
namespace libcloudphxx { namespace lgrngn { enum class as_t { undefined, implicit, euler, pred_corr }; } }
