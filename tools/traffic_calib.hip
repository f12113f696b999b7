// traffic_calib.hip -- known-byte access patterns for calibrating rocprofv3's FETCH_SIZE on gfx950 (MI355X_MICROARCH.md: the counter is
// calibrated for wide coalesced reads only).  Each pattern is its own kernel (its own row in the counter csv) over a table far larger
// than the 256 MiB Infinity Cache; tools/calibrate_traffic.sh runs this under `rocprofv3 --pmc` and tools/calibrate_traffic.py divides.
//   stream16 / stream8 / stream4: lane i reads element i (16, 8, 4 bytes): expected = the table's bytes
//   gather8_far:   lane i reads the 8-byte element (i * ODD) mod N -- neighbours 2^7+ lines apart, every 64-B sector touched by 8 lanes far
//                  apart in time: expected >= N * 8 (each sector fetched at least once), <= N * 64 (once per touch)
//   gather8_near:  the condensation kernel's shape -- lane i reads element i + d(i), |d| <= 512 elements pseudo-random: within a wave the
//                  64 loads fall into a 8 KiB window; expected ~ N * 8 plus what is fetched twice
//   gather8_mix:   half of the lanes (pseudo-randomly chosen) read their own element, the other half one up to 64 Ki elements away: what
//                  storage looks like a few dozen steps after its last re-ordering
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ __forceinline__ uint32_t mix32(uint32_t h) { h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h; }
__global__ void stream16(const float4 *p, size_t n, float *out) { size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; if (i < n) { float4 v = p[i]; if (v.x == 123.f) out[0] = v.y; } }
__global__ void stream8(const double *p, size_t n, double *out) { size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; if (i < n) { double v = p[i]; if (v == 123.) out[0] = v; } }
__global__ void stream4(const uint32_t *p, size_t n, uint32_t *out) { size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; if (i < n) { uint32_t v = p[i]; if (v == 123u) out[0] = v; } }
__global__ void gather8_far(const double *p, size_t n, double *out) { size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; if (i < n) { double v = p[(i * 2654435761ull) & (n - 1)]; if (v == 123.) out[0] = v; } }
__global__ void gather8_near(const double *p, size_t n, double *out)
{ size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; if (i < n) { size_t j = (i + (mix32(uint32_t(i)) & 1023u)) & (n - 1); double v = p[j]; if (v == 123.) out[0] = v; } }
__global__ void gather8_mix(const double *p, size_t n, double *out)
{ size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; if (i < n) { const uint32_t h = mix32(uint32_t(i)); size_t j = (h & 1u) ? i : (i + ((h >> 1) & 0xFFFFu)) & (n - 1); double v = p[j]; if (v == 123.) out[0] = v; } }
int main()
{
  const size_t bytes = size_t(2) << 30;            // 2 GiB table
  void *p, *out;
  CHK(hipMalloc(&p, bytes)); CHK(hipMalloc(&out, 64));
  CHK(hipMemset(p, 0, bytes)); CHK(hipDeviceSynchronize());
  const unsigned bs = 256;
  for (int rep = 0; rep < 2; ++rep) {
    size_t n = bytes / 16; hipLaunchKernelGGL(stream16, dim3(unsigned((n + bs - 1) / bs)), dim3(bs), 0, 0, (const float4 *)p, n, (float *)out);
    n = bytes / 8; hipLaunchKernelGGL(stream8, dim3(unsigned((n + bs - 1) / bs)), dim3(bs), 0, 0, (const double *)p, n, (double *)out);
    n = bytes / 4; hipLaunchKernelGGL(stream4, dim3(unsigned((n + bs - 1) / bs)), dim3(bs), 0, 0, (const uint32_t *)p, n, (uint32_t *)out);
    n = bytes / 8; hipLaunchKernelGGL(gather8_far, dim3(unsigned((n + bs - 1) / bs)), dim3(bs), 0, 0, (const double *)p, n, (double *)out);
    hipLaunchKernelGGL(gather8_near, dim3(unsigned((n + bs - 1) / bs)), dim3(bs), 0, 0, (const double *)p, n, (double *)out);
    hipLaunchKernelGGL(gather8_mix, dim3(unsigned((n + bs - 1) / bs)), dim3(bs), 0, 0, (const double *)p, n, (double *)out);
    CHK(hipDeviceSynchronize());
  }
  printf("table_bytes %zu\n", bytes);
  return 0;
}
