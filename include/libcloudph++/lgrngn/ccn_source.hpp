#pragma once   // reference lgrngn/ccn_source.hpp:8 (sources are outside the accelerated path: only `off` is accepted)
namespace libcloudphxx { namespace lgrngn { enum class src_t { off, simple, matching }; } }
