#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>
#include "../include/lcx.h"
#include "../libcloudphxx_amd/csrc/lcx_math.hpp"
using namespace lcx;
struct In { double Tk, rhod, RH, rd3, rw2, vt; };
__global__ void k(const In *in, size_t n, double *o3, double *o7, unsigned *its)
{
  size_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
  const In q = in[i];
  const double p = q.rhod * 287 * q.Tk * 1.01;
  const double rv = q.RH * 0.622 * p_vs(q.Tk) / (p - p_vs(q.Tk));
  const double eta = visc(q.Tk);
  cond_cell_fast<double> cc = make_cond_cell_fast(q.rhod, rv, q.Tk, eta, lambda_D_of(q.Tk), lambda_K_of(q.Tk, p), q.RH, 1.05);
  cond_fun_fast<double, 3> f3; f3.setup_cell(cc, q.rw2, 1.0, q.rd3, 0.61, q.vt);
  cond_fun_fast<double, 7> f7; f7.setup_cell(cc, q.rw2, 1.0, q.rd3, 0.61, q.vt);
  unsigned it = 0;
  const double eps = eps_tolerance<double>(16);
  o3[i] = advance_rw2_lean_with(f3, q.rw2, q.rd3, 1.0, eps, 2.0, 100u);
  o7[i] = advance_rw2_lean2_with(f7, q.rw2, q.rd3, 1.0, eps, 2.0, 100u);
  its[i] = it;
}
int main() {
  const size_t n = 4000000;
  std::mt19937_64 g(1);
  std::uniform_real_distribution<double> U(0, 1);
  std::vector<In> h(n);
  for (auto &q : h) {
    q.Tk = 280 + 10 * U(g); q.rhod = 1.0 + 0.2 * U(g); q.RH = 0.85 + 0.17 * U(g);
    const double rd = 1e-9 * std::pow(10., 2.5 * U(g));
    q.rd3 = rd * rd * rd;
    const double rw = rd * (1.05 + std::pow(10., 2.5 * U(g) - 1.5));
    q.rw2 = rw * rw;
    q.vt = U(g) < 0.1 ? -1. : 1e-4 * U(g);
    if (U(g) < 0.1) { q.rw2 = 3.6e-9; }
  }
  In *d; double *o3, *o7; unsigned *its;
  hipMalloc(&d, n * sizeof(In)); hipMalloc(&o3, n * 8); hipMalloc(&o7, n * 8); hipMalloc(&its, n * 4);
  hipMemcpy(d, h.data(), n * sizeof(In), hipMemcpyHostToDevice);
  k<<<(n + 255) / 256, 256>>>(d, n, o3, o7, its);
  std::vector<double> a(n), b(n); std::vector<unsigned> it(n);
  hipMemcpy(a.data(), o3, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), o7, n * 8, hipMemcpyDeviceToHost); hipMemcpy(it.data(), its, n * 4, hipMemcpyDeviceToHost);
  long diff = 0; double maxrel = 0; long hist[8] = {0};
  for (size_t i = 0; i < n; ++i) {
    hist[it[i] < 7 ? it[i] : 7]++;
    if (std::memcmp(&a[i], &b[i], 8)) { ++diff; double rel = std::fabs(a[i] / b[i] - 1); if (rel > maxrel) maxrel = rel;
      if (diff <= 12) printf("diff i=%zu rw2=%.17g rd3=%.17g RH=%g Tk=%g vt=%g r3=%.17g r7=%.17g rel=%g it=%u\n", i, h[i].rw2, h[i].rd3, h[i].RH, h[i].Tk, h[i].vt, a[i], b[i], rel, it[i]); }
  }
  printf("n %zu diff %ld maxrel %g\nits hist:", n, diff, maxrel);
  for (int k2 = 0; k2 < 8; ++k2) printf(" %ld", hist[k2]);
  printf("\n");
}
