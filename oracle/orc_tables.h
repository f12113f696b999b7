/* orc_tables.h -- TEST INFRASTRUCTURE (oracle).  Collision-efficiency tables for the
 * "geometric x tabulated efficiency" kernels (reference: src/detail/kernel_definitions/, six *_efficiencies.hpp files).
 * The tables are numeric data loaded at run time from libcloudphxx_amd/data/kernel_eff_<id>.f64
 * (see tools/extract_efficiency_tables.py); LCX_DATA_DIR overrides the directory. */
#ifndef ORC_TABLES_H
#define ORC_TABLES_H
#include <stdio.h>
#include <stdlib.h>
static const double *orc_efficiency_table(int kernel, size_t *n, double *r_max)
{
  static double *cache[16]; static size_t cache_n[16]; static double cache_rmax[16];
  if (kernel < 0 || kernel >= 16) return NULL;
  if (!cache[kernel]) {
    const char *dir = getenv("LCX_DATA_DIR");
    char path[1024];
    snprintf(path, sizeof path, "%s/kernel_eff_%d.f64", dir ? dir : "libcloudphxx_amd/data", kernel);
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    double hdr[2];                       /* [r_max, count] then count doubles */
    if (fread(hdr, sizeof(double), 2, f) != 2) { fclose(f); return NULL; }
    size_t cnt = (size_t)hdr[1];
    double *t = (double *)malloc(cnt * sizeof(double));
    if (fread(t, sizeof(double), cnt, f) != cnt) { fclose(f); free(t); return NULL; }
    fclose(f);
    cache[kernel] = t; cache_n[kernel] = cnt; cache_rmax[kernel] = hdr[0];
  }
  *n = cache_n[kernel]; *r_max = cache_rmax[kernel];
  return cache[kernel];
}
#endif
