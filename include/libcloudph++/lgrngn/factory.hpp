// factory<real_t>(backend, opts_init): same signature and failure mode as the reference
// (reference: lgrngn/factory.hpp:12-15, src/lib.cpp:13-40 -- std::runtime_error for a backend that is not built in).
#pragma once
#include "particles.hpp"
namespace libcloudphxx { namespace lgrngn {
  template <typename real_t>
  inline particles_proto_t<real_t> *factory(const backend_t backend, opts_init_t<real_t> opts_init)
  {
    switch (backend) {
      case HIP: return new particles_t<real_t, HIP>(opts_init);
      case CUDA: return new particles_t<real_t, CUDA>(opts_init);               // "the GPU backend" of a driver written for the reference
      case multi_HIP: return new particles_t<real_t, multi_HIP>(opts_init);     // one object, opts_init.dev_count devices (0: all)
      case multi_CUDA: return new particles_t<real_t, multi_CUDA>(opts_init);
      default:
        throw std::runtime_error(std::string("libcloudph++: backend ") + backend_str(backend) + " is not part of the HIP library (available: HIP, multi_HIP)");
    }
  }
} }
