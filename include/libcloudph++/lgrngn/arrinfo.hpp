// (pointer, strides-in-elements) view of a caller-owned n-d array -- same members and constructors as the
// reference's arrinfo_t (reference: lgrngn/arrinfo.hpp:11-49).  `on_device` is an extension: data may be a
// device pointer (lcx_arrinfo_t.on_device).
#pragma once
#include "extincl.hpp"
namespace libcloudphxx { namespace lgrngn {
  template <typename real_t>
  struct arrinfo_t
  {
    const std::vector<ptrdiff_t> strvec;
    real_t *const data;
    const ptrdiff_t *strides;
    bool on_device = false;

    arrinfo_t() : data(nullptr), strides(nullptr) {}
    arrinfo_t(real_t *const d, const ptrdiff_t *s) : data(d), strides(s) {}
    arrinfo_t(real_t *const d, const std::vector<ptrdiff_t> &sv) : strvec(sv), data(d), strides(strvec.data()) {}
    arrinfo_t(const arrinfo_t &o) : strvec(o.strvec), data(o.data), strides(strvec.empty() ? o.strides : strvec.data()), on_device(o.on_device) {}
    arrinfo_t(arrinfo_t &&o) : strvec(std::move(o.strvec)), data(o.data), strides(strvec.empty() ? o.strides : strvec.data()), on_device(o.on_device) {}
    bool is_null() const { return data == nullptr || strides == nullptr; }
  };
} }
