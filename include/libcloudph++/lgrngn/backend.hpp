// backend_t with the reference's enumerators (reference: lgrngn/backend.hpp:8) plus the two HIP slots.
// CUDA / multi_CUDA are accepted as aliases of HIP / multi_HIP by factory() so that a driver that asks for
// "the GPU backend" keeps working; serial / OpenMP are not part of this library.
#pragma once
#include "enum_names.hpp"
namespace libcloudphxx { namespace lgrngn {
  enum backend_t { undefined, serial, OpenMP, CUDA, multi_CUDA, HIP, multi_HIP };
  inline const char *backend_str(backend_t b)
  {
    switch (b) { case serial: return "serial"; case OpenMP: return "OpenMP"; case CUDA: return "CUDA"; case multi_CUDA: return "multi_CUDA";
                 case HIP: return "HIP"; case multi_HIP: return "multi_HIP"; default: return "undefined"; }
  }
  // keyed by the enum like the reference's table (backend.hpp:10-16); an unscoped enum hashes through std::hash<backend_t> (C++14)
  const std::unordered_map<backend_t, std::string> backend_name = detail::enum_names<backend_t>(
    {"undefined", "serial", "OpenMP", "CUDA", "multi_CUDA", "HIP", "multi_HIP"});
} }
