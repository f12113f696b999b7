// lcx_cond_wq.hip -- k_cond_lean_wq in a translation unit of its own (see lcx_cond_wq.hpp): built with -mllvm -disable-machine-licm
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstddef>
#include "lcx_cond_wq.hpp"

namespace lcx {

template <class T, bool UNI, int CAP, int PF> static void launch1(dim3 grid, hipStream_t st, const wq_params<T> &p)
{
  hipLaunchKernelGGL((k_cond_lean_wq<T, UNI, CAP, PF>), grid, dim3(BS), 0, st, p);
}
template <class T> void launch_cond_lean_wq(dim3 grid, hipStream_t st, const wq_params<T> &p, bool kpa_uniform, int cap, int prefetch)
{
  const int sel = (cap == 128 ? 4 : 0) | (prefetch == 1 ? 2 : 0) | (kpa_uniform ? 1 : 0);
  if (prefetch == 2) {
    if (kpa_uniform) launch1<T, true, 128, 2>(grid, st, p); else launch1<T, false, 128, 2>(grid, st, p);
    return;
  }
  switch (sel) {
    case 0: launch1<T, false, 96, 0>(grid, st, p); break;
    case 1: launch1<T, true, 96, 0>(grid, st, p); break;
    case 2: launch1<T, false, 96, 1>(grid, st, p); break;
    case 3: launch1<T, true, 96, 1>(grid, st, p); break;
    case 4: launch1<T, false, 128, 0>(grid, st, p); break;
    case 5: launch1<T, true, 128, 0>(grid, st, p); break;
    case 6: launch1<T, false, 128, 1>(grid, st, p); break;
    default: launch1<T, true, 128, 1>(grid, st, p); break;
  }
}
template void launch_cond_lean_wq<double>(dim3, hipStream_t, const wq_params<double> &, bool, int, int);
template void launch_cond_lean_wq<float>(dim3, hipStream_t, const wq_params<float> &, bool, int, int);

}  // namespace lcx
