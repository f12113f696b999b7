#pragma once   // reference lgrngn/ccn_source.hpp:8 (sources are outside the accelerated path: only `off` is accepted); src_name: ccn_source.hpp:14-18
#include "enum_names.hpp"
namespace libcloudphxx { namespace lgrngn {
  enum class src_t { off, simple, matching };
  const std::unordered_map<src_t, std::string> src_name = detail::enum_names<src_t>({"off", "simple", "matching"});
} }
