// Host-side functor base used for n(ln rd) spectra -- same contract as the reference's
// common::unary_function<real_t> (reference: include/libcloudph++/common/unary_function.hpp:11-18):
// derive, override funval(); the library evaluates it on the HOST during init().
#pragma once
namespace libcloudphxx { namespace common {
  template <typename real_t>
  struct unary_function
  {
    virtual ~unary_function() = default;
    virtual real_t funval(const real_t) const = 0;
    real_t operator()(const real_t x) const { return funval(x); }
  };
} }
