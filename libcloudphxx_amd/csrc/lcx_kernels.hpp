// lcx_kernels.hpp -- hand-written HIP kernels of the lgrngn hot path for gfx950 (CDNA4, wave64).
//
// Data layout in HBM: structure of arrays, one contiguous array per super-droplet attribute
//   n (u64), rd3, rw2, kpa, vt, x, y, z (real_t)   -- storage order == the reference's storage order
//   ijk, sorted_id, sorted_ijk (u32)               -- housekeeping (u32: n_cell, n_part < 2^32)
//   cell_start (u32, n_cell+1)                     -- CSR offsets of the cell-sorted order
// Every streaming kernel is launched dense over its index space (lane i <-> element i) so that the
// 64 lanes of a wave touch 64 consecutive elements (512 B of fp64) per load; kernels that walk the
// cell-sorted order read the gather index coalesced and the cell fields as wave-wide broadcasts.
// No MFMA anywhere: there is no dense contraction on this path; the roofline is HBM (and fp64 VALU
// for the condensation root finder).
#pragma once
#include "lcx_math.hpp"
#include "lcx_cond_wq.hpp"

namespace lcx {

// ============================================================================================
// generic helpers
// ============================================================================================
template <class T> __global__ void k_fill(T *a, size_t n, T v) { size_t i = gid(); if (i < n) a[i] = v; }
__global__ void k_iota(uint32_t *a, size_t n) { size_t i = gid(); if (i < n) a[i] = uint32_t(i); }
template <class D, class S> __global__ void k_convert(D *d, const S *s, size_t n) { size_t i = gid(); if (i < n) d[i] = D(s[i]); }

// particles_step.ipp:127-142: is any Courant number outside [-2, 2] (beyond the pred_corr halo)?
template <class T> __global__ void k_flag_outside(const T *v, size_t n, T lo, T hi, int *flag)
{ const size_t i = gid(); if (i < n && (v[i] < lo || v[i] > hi)) *flag = 1; }
// strided host-layout array -> contiguous library layout (particles_impl_sync.ipp:15-68, init_e2l.ipp:34-114)
// ext: extents (n0,n1,n2) of the device-layout field (z fastest), st: element strides of the source,
// i_off: index offset in the first dimension (x-planes owned by ranks to the left)
template <class T>
__global__ void k_gather_strided(T *dst, const T *src, size_t n, int ndims, int n1, int n2, long s0, long s1, long s2, long i_off,
                                 long wrap_planes = 0)
{
  // wrap_planes > 0 (Courant numbers with an x-halo): plane indices outside [0, wrap_planes) wrap around the user's array
  // cyclically (init_e2l.ipp:109-113)
  size_t c = gid(); if (c >= n) return;
  long i = ndims == 0 ? 0 : ndims == 1 ? long(c) : ndims == 2 ? long(c / n2) : long(c / (size_t(n2) * n1));
  i += i_off;
  if (wrap_planes > 0) { if (i >= wrap_planes) i -= wrap_planes; else if (i < 0) i += wrap_planes; }
  long off;
  if (ndims == 0) off = 0;
  else if (ndims == 1) off = i * s0;
  else if (ndims == 2) off = i * s0 + long(c % n2) * s1;
  else off = i * s0 + long((c / n2) % n1) * s1 + long(c % n2) * s2;
  dst[c] = src[off];
}
template <class T>
__global__ void k_scatter_strided(T *dst, const T *src, size_t n, int ndims, int n1, int n2, long s0, long s1, long s2, long i_off)
{
  size_t c = gid(); if (c >= n) return;
  long off;
  if (ndims == 0) off = 0;
  else if (ndims == 1) off = (long(c) + i_off) * s0;
  else if (ndims == 2) off = (long(c / n2) + i_off) * s0 + long(c % n2) * s1;
  else off = (long(c / (size_t(n2) * n1)) + i_off) * s0 + long((c / n2) % n1) * s1 + long(c % n2) * s2;
  dst[off] = src[c];
}

// several fields in ONE launch (sync_in moves up to seven arrays per step; on a slab of a decomposed domain seven launches of a few
// microseconds each were a measurable share of a 2 ms step): job j owns the blocks [first_block[j], first_block[j+1])
constexpr int MAX_SYNC_JOBS = 8;
template <class T> struct sync_jobs {
  int n_jobs, ndims;
  T *dst[MAX_SYNC_JOBS]; const T *src[MAX_SYNC_JOBS]; size_t n[MAX_SYNC_JOBS];
  int n1[MAX_SYNC_JOBS], n2[MAX_SYNC_JOBS]; long s0[MAX_SYNC_JOBS], s1[MAX_SYNC_JOBS], s2[MAX_SYNC_JOBS], i_off[MAX_SYNC_JOBS], wrap[MAX_SYNC_JOBS];
  unsigned first_block[MAX_SYNC_JOBS + 1];
};
template <class T, bool GATHER>          // GATHER: strided caller array -> contiguous library array; else the reverse (sync out)
__global__ void k_sync_multi(sync_jobs<T> J)
{
  int j = 0;
  while (j + 1 < J.n_jobs && blockIdx.x >= J.first_block[j + 1]) ++j;
  const size_t c = size_t(blockIdx.x - J.first_block[j]) * BS + threadIdx.x;
  if (c >= J.n[j]) return;
  const int n1 = J.n1[j], n2 = J.n2[j];
  long i = J.ndims == 0 ? 0 : J.ndims == 1 ? long(c) : J.ndims == 2 ? long(c / n2) : long(c / (size_t(n2) * n1));
  i += J.i_off[j];
  if (J.wrap[j] > 0) { if (i >= J.wrap[j]) i -= J.wrap[j]; else if (i < 0) i += J.wrap[j]; }
  long off;
  if (J.ndims == 0) off = 0;
  else if (J.ndims == 1) off = i * J.s0[j];
  else if (J.ndims == 2) off = i * J.s0[j] + long(c % n2) * J.s1[j];
  else off = i * J.s0[j] + long((c / n2) % n1) * J.s1[j] + long(c % n2) * J.s2[j];
  if (GATHER) J.dst[j][c] = J.src[j][off];
  else const_cast<T *>(J.src[j])[off] = J.dst[j][c];     // (sync out: `src` is the caller's array, `dst` the library's)
}

// ---- exclusive scan of u32 (three launches: per-tile scan, scan of tile sums, add) ----
constexpr int SCAN_TILE = 2048;         // 256 threads x 8 items, items interleaved for coalescing

// Inclusive prefix sum over the 64 lanes of a wave in SIX vector instructions: Hillis-Steele inside each row of 16 lanes with the data
// parallel primitives' row shifts (lanes without a source keep their value: `old` = 0 is added), then the last lane of rows 0 and 2
// broadcast into rows 1 and 3, then lane 31 into the upper half (row_bcast:15 / row_bcast:31, GFX9).  The same sums as the shuffle
// form below (integer addition), which costs a cross-lane LDS instruction, the lane test and the add per step: ~30 vector + 6 LDS
// instructions -- an eighth of k_cellrank_bkt's.
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v)
{
  v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x111, 0xf, 0xf, false));     // row_shr:1
  v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x112, 0xf, 0xf, false));     // row_shr:2
  v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x114, 0xf, 0xf, false));     // row_shr:4
  v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x118, 0xf, 0xf, false));     // row_shr:8
  v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x142, 0xa, 0xf, false));     // row_bcast:15 into rows 1 and 3
  v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x143, 0xc, 0xf, false));     // row_bcast:31 into rows 2 and 3
  return v;
}
// block-wide exclusive prefix of one value per thread (wave shuffles + LDS across the 4 waves)
constexpr int SCAN_MAX_WAVES = 16;      // block_exclusive_scan serves workgroups of up to 1024 threads (k_scan_sums)
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t &total, uint32_t *lds /* SCAN_MAX_WAVES entries */)
{
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) { uint32_t t = __shfl_up(inc, d); if (lane_id() >= unsigned(d)) inc += t; }
  if (lane_id() == WAVE - 1) lds[wave_id()] = inc;
  __syncthreads();
  uint32_t woff = 0, tot = 0;
  for (unsigned w = 0; w < blockDim.x / WAVE; ++w) { uint32_t s = lds[w]; if (w < wave_id()) woff += s; tot += s; }
  __syncthreads();
  total = tot;
  return woff + inc - v;
}
__global__ void k_scan_tiles(const uint32_t *in, uint32_t *out, uint32_t *tile_sums, size_t n)
{
  __shared__ uint32_t lds[SCAN_MAX_WAVES];
  const size_t base = size_t(blockIdx.x) * SCAN_TILE;
  uint32_t run = 0;
  for (int it = 0; it < SCAN_TILE / BS; ++it) {
    const size_t i = base + size_t(it) * BS + threadIdx.x;
    const uint32_t v = i < n ? in[i] : 0u;
    uint32_t tot;
    const uint32_t ex = block_exclusive_scan(v, tot, lds);
    if (i < n) out[i] = run + ex;
    run += tot;
  }
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = run;
}
// single workgroup: in-place exclusive scan of the tile sums; writes the grand total to *total
__global__ void k_scan_sums(uint32_t *sums, size_t m, uint32_t *total)
{
  __shared__ uint32_t lds[SCAN_MAX_WAVES];
  uint32_t run = 0;
  for (size_t base = 0; base < m; base += blockDim.x) {
    const size_t i = base + threadIdx.x;
    const uint32_t v = i < m ? sums[i] : 0u;
    uint32_t tot;
    const uint32_t ex = block_exclusive_scan(v, tot, lds);
    if (i < m) sums[i] = run + ex;
    run += tot;
  }
  if (threadIdx.x == 0) *total = run;
}
// zero_in: the scanned input is cleared behind the scan (the cell histogram is all zeros whenever no sort is in flight, so the
// passes that fill it need no memset of their own); zero_words: a few single counters cleared on the same occasion
__global__ void k_scan_add(uint32_t *out, const uint32_t *tile_offs, size_t n, const uint32_t *total, uint32_t *out_last,
                           uint32_t *zero_in = nullptr, uint32_t *zero_words = nullptr, int n_zero_words = 0)
{
  const size_t i = gid();
  if (i < n) { out[i] += tile_offs[i / SCAN_TILE]; if (zero_in) zero_in[i] = 0u; }
  if (i == 0 && out_last) *out_last = *total;        // out[n] = total (CSR end)
  if (i < size_t(n_zero_words)) zero_words[i] = 0u;
}

// ============================================================================================
// cell kernels (n_cell threads)
// ============================================================================================
// hskpng_Tpr.ipp:219-305
template <class T>
__global__ void k_cell_Tpr(size_t n_cell, const T *th, const T *rhod, const T *rv, T *p, T *Tk, T *RH, T *eta, T *dv,
                           int th_dry, int const_p, int RH_formula, int ndims)
{
  const size_t c = gid(); if (c >= n_cell) return;
  const T t = th_dry ? theta_dry_T(th[c], rhod[c]) : T(th[c] * exner(p[c]));
  Tk[c] = t;
  T pp = p[c];
  if (!const_p) { pp = theta_dry_p(rhod[c], rv[c], t); p[c] = pp; }
  RH[c] = RH_of(RH_formula, pp, rv[c], t);
  eta[c] = visc(t);
  if (ndims == 0) dv[c] = T(1) / rhod[c];
}
// hskpng_mfp.ipp:42-51
template <class T>
__global__ void k_cell_mfp(size_t n_cell, const T *Tk, const T *p, T *lambda_D, T *lambda_K)
{
  const size_t c = gid(); if (c >= n_cell) return;
  lambda_D[c] = lambda_D_of(Tk[c]);
  lambda_K[c] = lambda_K_of(Tk[c], p[c]);
}
// condensation substep's cell pass in one launch: (substep 0) the mean free paths from the temperature and pressure of the PREVIOUS
// housekeeping, as the reference's hskpng_mfp placed ahead of the substep loop (particles_step.ipp:193-196) -- then hskpng_Tpr --
// then (fast arithmetic) the droplet-independent set-up of the growth rate.  Same expressions as the three kernels it replaces.
template <class T> struct sstp_fields { int n, step; T sstp; T *scl[3], *tmp[3]; };
// one cell of that pass (shared with k_cond_substeps): returns the growth rate's set-up of the cell
template <class T> struct cell_pre_args {
  size_t n_cell; const T *th, *rhod, *rv; T *p, *Tk, *RH, *eta, *dv, *lambda_D, *lambda_K;
  int th_dry, const_p, RH_formula, ndims; T RH_max; cond_cell_fast<T> *pre;
};
template <class T>
__device__ __forceinline__ cond_cell_fast<T> cell_cond_pre_one(const cell_pre_args<T> &A, size_t c, int do_mfp, const sstp_fields<T> &ss)
{
  // (round 6) the substep's share of the Eulerian fields' change (sstp_percell_step.ipp:7-48: k_sstp_step's operations, per field) ahead of
  // the cell's own pass over the same fields: two or three launches fewer per substep -- a 2-D set-up of ten substeps is a queue of
  // five-microsecond kernels, and what it costs is their number (bench.py's c2 leg)
#pragma unroll
  for (int f = 0; f < 3; ++f) {            // (unrolled: the fields' pointers stay in scalar registers, a run-time index would put the struct on the stack)
    if (f >= ss.n) break;
    T *scl = ss.scl[f], *tmp = ss.tmp[f];
    if (ss.step == 0) {
      const T d = scl[c] - tmp[c];
      tmp[c] = d;
      scl[c] = scl[c] - (ss.sstp - 1) * d / ss.sstp;
    } else scl[c] = scl[c] + tmp[c] / ss.sstp;
  }
  T lD, lK;
  if (do_mfp) { lD = lambda_D_of(A.Tk[c]); lK = lambda_K_of(A.Tk[c], A.p[c]); A.lambda_D[c] = lD; A.lambda_K[c] = lK; }
  else { lD = A.lambda_D[c]; lK = A.lambda_K[c]; }
  const T t = A.th_dry ? theta_dry_T(A.th[c], A.rhod[c]) : T(A.th[c] * exner(A.p[c]));
  A.Tk[c] = t;
  T pp = A.p[c];
  if (!A.const_p) { pp = theta_dry_p(A.rhod[c], A.rv[c], t); A.p[c] = pp; }
  const T rh = RH_of(A.RH_formula, pp, A.rv[c], t), et = visc(t);
  A.RH[c] = rh;
  A.eta[c] = et;
  if (A.ndims == 0) A.dv[c] = T(1) / A.rhod[c];
  cond_cell_fast<T> cc{};
  if (A.pre) { cc = make_cond_cell_fast(A.rhod[c], A.rv[c], t, et, lD, lK, rh, A.RH_max); A.pre[c] = cc; }
  return cc;
}
template <class T>
__global__ void k_cell_cond_pre(size_t n_cell, const T *th, const T *rhod, const T *rv, T *p, T *Tk, T *RH, T *eta, T *dv, T *lambda_D, T *lambda_K,
                                int th_dry, int const_p, int RH_formula, int ndims, int do_mfp, T RH_max, cond_cell_fast<T> *pre,
                                uint32_t *zero_words = nullptr, int n_zero_words = 0, sstp_fields<T> ss = sstp_fields<T>{0, 0, T(1), {nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}})
{
  if (gid() < size_t(n_zero_words)) zero_words[gid()] = 0u;         // (the straggler counters of the condensation kernel that follows)
  const size_t c = gid(); if (c >= n_cell) return;
  const cell_pre_args<T> A{n_cell, th, rhod, rv, p, Tk, RH, eta, dv, lambda_D, lambda_K, th_dry, const_p, RH_formula, ndims, RH_max, pre};
  (void)cell_cond_pre_one(A, c, do_mfp, ss);
}
// sstp_percell_step.ipp:7-48 for one field (the fields of a substep as ONE argument of the cell pass: sstp_fields)
template <class T>
__global__ void k_sstp_step(size_t n_cell, int step, T sstp, T *scl, T *tmp)
{
  const size_t c = gid(); if (c >= n_cell) return;
  if (step == 0) {
    const T d = scl[c] - tmp[c];
    tmp[c] = d;
    scl[c] = scl[c] - (sstp - 1) * d / sstp;
  } else scl[c] = scl[c] + tmp[c] / sstp;
}
// init_grid.ipp:14-54
template <class T>
__global__ void k_init_dv(size_t n_cell, T *dv, int ny, int nz, T dx, T dy, T dz, T x0, T y0, T z0, T x1, T y1, T z1)
{
  const size_t c = gid(); if (c >= n_cell) return;
  const int ijk = int(c);
  const int i = (ijk / nz) / ny, j = (ijk / nz) % ny, k = ijk % nz;
  dv[c] = mx(T(0), (mn(T((i + 1) * dx), x1) - mx(T(i * dx), x0)) *
                   (mn(T((j + 1) * dy), y1) - mx(T(j * dy), y0)) *
                   (mn(T((k + 1) * dz), z1) - mx(T(k * dz), z0)));
}

// ============================================================================================
// housekeeping: cell index, sort
// ============================================================================================
struct grid_t { int nx, ny, nz, ndims; double dx, dy, dz; };
// (DEAD_CELL, the ijk of a super-droplet with n == 0 that has not been compacted away yet: lcx_cond_wq.hpp)

// hskpng_ijk.ipp:159-200,33-82: size_t(double(x)/double(dx)), z fastest
template <class T>
__device__ __forceinline__ uint32_t cell_of(const grid_t &g, T x, T y, T z)
{
  // size_t(double(x) / dx) of the reference; indices are < 2^32 here, so the native f64 -> u32 conversion gives the same value
  // (measured: the three IEEE divisions replaced by a reciprocal + exact-remainder check, bit-identical -- k_move stays at 2.53 ms)
  const uint32_t i = g.nx ? uint32_t(double(x) / g.dx) : 0u, j = g.ny ? uint32_t(double(y) / g.dy) : 0u, k = g.nz ? uint32_t(double(z) / g.dz) : 0u;
  switch (g.ndims) {
    case 0: return 0u;
    case 1: return i;
    case 2: return i * uint32_t(g.nz) + k;
    default: return i * (uint32_t(g.nz) * uint32_t(g.ny)) + j * uint32_t(g.nz) + k;
  }
}
// Histogram with per-SD arrival rank, wave-aggregated: lanes of a wave that fall into the same cell are
// counted with ONE atomic (leader adds the group size, members take consecutive ranks).  SDs that are close in
// storage are close in space, so a wave touches a handful of cells and the number of global atomics drops by
// ~10x compared with one atomic per SD.  Rank order is arbitrary; the per-cell sort below makes the final order
// deterministic and equal to the stable sort of the reference.  Must be called by ALL lanes of the wave.
// (Measured: capping the grouping loop at 6 rounds and letting the remaining lanes count themselves, one atomic each -- k_move 1.79 ->
// 2.36 ms on C3: between two storage re-orderings a wave's 64 neighbours in storage spread over a dozen cells, and it is the returning
// atomics that cost, not the rounds of ballots.  This is also why k_move is a quarter slower on a slab with neighbours than on the same
// slab alone: immigrants take the emigrants' slots in the boundary planes, whose waves then meet a cell per lane.)
__device__ __forceinline__ uint32_t wave_hist_rank(uint32_t *cnt, uint32_t c, bool active)
{
  // 1) group the lanes by cell with ballots/shuffles only (no memory traffic inside the loop)
  const unsigned l = lane_id();
  unsigned my_leader = l;
  uint32_t my_rank = 0, my_size = 0;
  unsigned long long todo = __ballot(active);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const uint32_t lc = __shfl(c, leader);
    const unsigned long long same = __ballot(active && c == lc) & todo;
    if ((same >> l) & 1ull) {
      my_leader = unsigned(leader);
      my_rank = uint32_t(__popcll(same & ((1ull << l) - 1ull)));
      my_size = uint32_t(__popcll(same));
    }
    todo &= ~same;
  }
  // 2) ONE atomic instruction per wave: every group leader reserves its group's slots at once
  uint32_t base = 0;
  if (active && l == my_leader) base = atomicAdd(&cnt[c], my_size);
  base = __shfl(base, int(my_leader));
  return base + my_rank;
}
// ijk (optional) + histogram/rank in one pass over the storage.  SDs with n == 0 ("dead": removed by the
// reference's hskpng_remove_n0, here possibly still in storage until the next compaction) get ijk = DEAD_CELL
// and take no part in the histogram, so they never enter the sorted order.
template <class T>
__global__ void __launch_bounds__(BS)
k_ijk_hist(size_t first, size_t n, grid_t g, const n_t *mult, const T *x, const T *y, const T *z, uint32_t *ijk, uint32_t *cnt, uint32_t *rank, int do_ijk)
{
  const size_t i = first + gid();                    // [first, n): the whole storage, or only the immigrants appended by unpack
  bool active = i < n;
  uint32_t c = 0;
  if (active) {
    // do_ijk: 0 keep ijk; 1 removal point (post_copy): n == 0 leaves the order; 2 plain hskpng_ijk: an SD with n == 0 stays in
    // the order until the next removal point, exactly as in the reference (e.g. zero-multiplicity SDs right after init)
    if (do_ijk) {
      if (do_ijk == 1 ? mult[i] == 0 : ijk[i] == DEAD_CELL) c = DEAD_CELL;
      else c = cell_of(g, g.nx ? x[i] : T(0), g.ny ? y[i] : T(0), g.nz ? z[i] : T(0));
      ijk[i] = c;
    } else c = ijk[i];
    active = c != DEAD_CELL;
  }
  if (cnt) { const uint32_t r = wave_hist_rank(cnt, c, active); if (active) rank[i] = r; }
}
// (measured: several super-droplets per lane -- four consecutive ids, or one from each of four chunks -- do not help here, 2.13-2.43
// against 2.07 ms for the whole re-sort: the pass is bound by its scattered 4-byte stores, not by the latency of its loads)
// Which part of the order a pass serves.  The overlapped re-sort of a slab with neighbours (lcx_core.hip, exch_*) puts its INTERIOR
// cells [c_lo, c_hi) in order while the neighbours' messages are still travelling and the boundary planes afterwards; the sorted
// arrays of such a slab begin `shift` entries BELOW the address the host knows (the left boundary planes' immigrants go in front).
//   mode 0: every cell;  1: cells in [c_lo, c_hi) only;  2: the others only.
//   n_dev / shift != nullptr: the extent of the storage / the shift are read from device memory (not known to the host yet)
struct sort_part { uint32_t c_lo, c_hi; int mode; const uint32_t *n_dev, *shift; };
__device__ __forceinline__ bool part_has(const sort_part &sp, uint32_t c)
{ return sp.mode == 0 || ((c >= sp.c_lo && c < sp.c_hi) == (sp.mode == 1)); }
// (Measured and dropped: four consecutive super-droplets per lane with 16-byte loads of ijk and rank -- the re-sort 1.83 -> 2.09 ms on C3:
// a lane's four targets are neighbours, the wave's stores no longer are.)
// skipped != nullptr (the interior pass of the overlapped re-sort): one byte per wave of 64 super-droplets, set when the wave holds a
// living super-droplet that this pass leaves out -- the boundary pass (k_scatter_flagged) then reads the cell indices of those waves only
__global__ void k_scatter_sorted(size_t n, const uint32_t *ijk, const uint32_t *rank, const uint32_t *cell_start,
                                 uint32_t *sorted_id, uint32_t *sorted_ijk, sort_part sp = sort_part{0u, 0u, 0, nullptr, nullptr}, uint8_t *skipped = nullptr)
{
  if (sp.n_dev) n = *sp.n_dev;
  const size_t i = gid();
  const uint32_t c = i < n ? ijk[i] : DEAD_CELL;
  if (skipped) {
    const bool left_out = c != DEAD_CELL && !part_has(sp, c);
    const unsigned long long b = __ballot(left_out);
    if (lane_id() == 0 && (i & ~size_t(63)) < n) skipped[i >> 6] = b ? uint8_t(1) : uint8_t(0);
  }
  if (c == DEAD_CELL || !part_has(sp, c)) return;
  const size_t pos = size_t(cell_start[c]) + rank[i] - (sp.shift ? *sp.shift : 0u);      // (shift <= the headroom in front of the arrays)
  (sorted_id + pos)[0] = uint32_t(i);
  (sorted_ijk + pos)[0] = c;
}

// The boundary pass of the overlapped re-sort over the waves that are flagged: those in which the interior pass left a super-droplet
// out (k_scatter_sorted's `skipped`) and those that have just taken an immigrant (k_unpack_dev).  One wave looks at sixteen flags and
// visits the flagged sixty-fours; on a 16-plane slab one wave in eight is flagged, on C3 one in sixty-four.
__global__ void __launch_bounds__(BS)
k_scatter_flagged(size_t n_max, const uint8_t *wave_flag, const uint32_t *ijk, const uint32_t *rank, const uint32_t *cell_start,
                  uint32_t *sorted_id, uint32_t *sorted_ijk, sort_part sp)
{
  const size_t n = sp.n_dev ? size_t(*sp.n_dev) : n_max;
  const size_t cb = (size_t(blockIdx.x) * (BS / WAVE) + wave_id()) * 16;             // first of this wave's sixteen chunks
  if (cb * WAVE >= n) return;
  const unsigned l = lane_id();
  const uint8_t f = (l < 16 && (cb + l) * WAVE < n) ? wave_flag[cb + l] : uint8_t(0);
  unsigned long long todo = __ballot(f != 0);
  if (!todo) return;
  const uint32_t shift = sp.shift ? *sp.shift : 0u;
  while (todo) {
    const int k = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const size_t i = (cb + size_t(k)) * WAVE + l;
    if (i >= n) continue;
    const uint32_t c = ijk[i];
    if (c == DEAD_CELL || (c >= sp.c_lo && c < sp.c_hi)) continue;
    const size_t pos = size_t(cell_start[c]) + rank[i] - shift;
    sorted_id[pos] = uint32_t(i);
    sorted_ijk[pos] = c;
  }
}
// the same pass over every cell index (kept for LCX_NO_WAVE_FLAGS=1 and as the flagged form's reference): only one SD in eight or
// sixteen has anything to store, the pass is the read of ijk -- four ids per lane with one 16-byte load (41 us on a 16.7e6-SD slab,
// 117 on C3)
__global__ void __launch_bounds__(BS)
k_scatter_outside4(size_t n_max, const uint32_t *ijk, const uint32_t *rank, const uint32_t *cell_start, uint32_t *sorted_id, uint32_t *sorted_ijk, sort_part sp)
{
  const size_t n = sp.n_dev ? size_t(*sp.n_dev) : n_max;
  const size_t i0 = gid() * 4;
  if (i0 >= n) return;
  const uint32_t shift = sp.shift ? *sp.shift : 0u;
  uint32_t c[4];
  if (i0 + 4 <= n) { const uint4 v = *reinterpret_cast<const uint4 *>(ijk + i0); c[0] = v.x; c[1] = v.y; c[2] = v.z; c[3] = v.w; }
  else for (int k = 0; k < 4; ++k) c[k] = i0 + k < n ? ijk[i0 + k] : DEAD_CELL;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (c[k] == DEAD_CELL || (c[k] >= sp.c_lo && c[k] < sp.c_hi)) continue;
    const size_t pos = size_t(cell_start[c[k]]) + rank[i0 + k] - shift;
    sorted_id[pos] = uint32_t(i0 + k);
    sorted_ijk[pos] = c[k];
  }
}

// Per-cell ordering.  Reference semantics (hskpng_sort.ipp:15-57): sorted_id = stable_sort_by_key(ijk) of the
// sequence 0..n-1  => inside a cell ids ascend; shuffled variant: first stable sort by the random key un[id],
// then stable sort by cell => inside a cell (un[id], id) ascends.
// One lane per position of the cell-grouped order (dense, coalesced).  A workgroup stages the keys of all cells
// its 256 positions touch (its own range plus the overhang of the first and last cell) in LDS; every lane then
// ranks its key inside its cell by counting smaller keys (keys are unique) with LDS broadcast reads and writes
// its id to `out` at cell_start + rank.  O(count) reads per lane, no data-dependent branching, no atomics.
// cells with more SDs than cellrank_max are listed and sorted by a bitonic network instead (O(n log^2 n))
constexpr int CELLRANK_MAX = 256;
template <class KEY> constexpr int cellrank_max = CELLRANK_MAX;
constexpr int CELLSORT_LDS_MAX = 2048;  // ... in LDS up to this size (k_cellsort_lds), in global scratch beyond (k_cellsort_big)
template <class KEY> constexpr int cr_cap = 1024;                               // keys staged per workgroup (4 / 8 KiB)
// un == nullptr: the device generator.  The random key of the in-cell shuffle is then un(id) = mix(mix(id ^ s1) + s2) with the 32-bit
// finaliser of MurmurHash3 for mix -- a BIJECTION of the 32-bit ids, salted per call with two words of Philox(call, seed) that the host
// draws (rand_un): the keys of a call are all different, so 32 bits order a cell without a tie-break (half the LDS traffic of the
// ranking, and a dozen integer operations per key instead of Philox's ten rounds: re-sort on C3 2.0 -> 1.7 ms).  Any function
// of (id, call, seed) alone is as deterministic as any other; the reference draws un from its generator's stream the same way
// (hskpng_sort.ipp:38-46, urand.hpp:57-86).  (Round 5, measured and dropped: ONE round of the finaliser instead of two -- two of the four
// quarter-rate multiplications per key gone: the ranking 0.780 -> 0.769 ms, not worth the weaker mixing.)
struct rng_src { const uint32_t *un; uint64_t call, seed; uint32_t s1, s2; };
LCX_HD uint32_t mix32(uint32_t h) { h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h; }
LCX_HD uint32_t shuffle_un(uint32_t id, uint32_t s1, uint32_t s2) { return mix32(mix32(id ^ s1) + s2); }

__device__ __forceinline__ uint64_t sort_key(uint32_t id, int shuffle, const rng_src &r)
{
  if (!shuffle) return id;
  // (s1 == s2 == 0: the measurement switch LCX_SHUFFLE_PHILOX -- keys drawn from Philox as in round 2, 64-bit ranking)
  const uint32_t u = r.un ? r.un[id] : (r.s1 | r.s2) ? shuffle_un(id, r.s1, r.s2) : philox::un(id, r.call, r.seed);
  return (uint64_t(u) << 32) | id;
}
// KEY = uint32_t with shuffle: the device generator's keys alone (unique, see above)
template <class KEY, bool SHUFFLE> __device__ __forceinline__ KEY rank_key(uint32_t id, const rng_src &r)
{
  if constexpr (!SHUFFLE) return KEY(id);
  else if constexpr (sizeof(KEY) == 4) return KEY(shuffle_un(id, r.s1, r.s2));
  else return KEY(sort_key(id, 1, r));
}
// KEY = uint32_t for the plain order (key == id), uint64_t for the shuffled order ((un << 32) | id).
// The kernel is bound by the latency of its dependent loads (three quarters of its wave-cycles are waits), so every lane
// fetches its own cell index, id and segment bounds up front (two dependent levels: sorted_ijk / in, then cell_start) and the
// workgroup only adds the overhang of its first and last cell before the one barrier ahead of the ranking loop (C3: 1.17 -> 1.04 ms
// for the ids, 1.41 -> 1.32 ms for the shuffle keys).  Tried and dropped: 32-bit shuffle keys (un alone) with the position in the
// plain order as tie-break -- lane-dependent loop bounds, 1.9 ms; a cell-major form (a workgroup owns a few whole cells, no
// overhang, no sorted_ijk) -- ragged ranges and the cell look-up cost more than the saved load level, 1.49 ms.
// Also tried: ranking by buckets as k_cellsort_wave does for crowded cells (bucket = the key's expected rank in its cell, LDS histogram +
// workgroup scan + in-bucket compares instead of 64 compares per key) -- a third of the instructions but five barriers instead of
// two: post_copy 2.03 against 2.09 ms on C3, not worth a second kernel.
// rank_range: the positions [*lo, *hi) of the order this launch serves (both CSR offsets in device memory, i.e. cell boundaries; nullptr:
// [0, n)), workgroup b taking the 256 positions from *lo + 256 b; shift: see sort_part
struct rank_range { const uint32_t *lo, *hi, *shift; };
// PROD: keys from the device generator (no replayed array), no crowded cells expected -- the uniform tests folded away; with KEY =
// uint32_t it is the SHUFFLED order on 32-bit keys (rank_key)
template <class KEY, bool PROD = false>
__global__ void __launch_bounds__(BS)
k_cellrank(size_t n, const uint32_t *sorted_ijk, const uint32_t *cell_start, const uint32_t *in, uint32_t *out, rng_src r, int crowded,
           rank_range rg = rank_range{nullptr, nullptr, nullptr})
{
  constexpr bool shuffle = sizeof(KEY) == 8 || PROD;
  if (PROD) { r.un = nullptr; crowded = 0; }
  __shared__ KEY lds[cr_cap<KEY>];
  __shared__ uint32_t bounds[2];
  size_t r_lo = 0;
  if (rg.lo) { r_lo = *rg.lo; n = *rg.hi; }
  if (rg.shift) { const uint32_t sh = *rg.shift; in -= sh; out -= sh; sorted_ijk -= sh; }
  const size_t p0 = r_lo + size_t(blockIdx.x) * BS;
  if (p0 >= n) return;                                     // (uniform: the grid covers an upper bound of the range)
  const size_t plast = (p0 + BS < n ? p0 + BS : n) - 1;
  const size_t p = p0 + threadIdx.x;
  const bool active = p < n;
  uint32_t c = 0, s = 0, e = 0, id = 0;
  // SPEC positions on either side of the workgroup's range are fetched together with its own ids, before the cell bounds are known: the
  // overhang of the first and last cell (half a cell on average) is then staged without a third dependent load level
  constexpr int SPEC = BS / 2;
  const size_t pe = threadIdx.x < SPEC ? p0 - SPEC + threadIdx.x : p0 + BS + (threadIdx.x - SPEC);      // (wraps below zero: caught by pe < n)
  const bool have_e = (threadIdx.x < SPEC ? p0 >= r_lo + (size_t(SPEC) - threadIdx.x) : true) && pe < n;
  uint32_t id_e = 0;
  if (have_e) id_e = in[pe];
  if (active) { c = sorted_ijk[p]; id = in[p]; s = cell_start[c]; e = cell_start[c + 1]; }
  // a workgroup that sees only crowded cells (every cell of C5) has nothing to rank: the listed-cell sorts take the ids as they are.
  // `crowded`: the host's guess from the mean SDs per cell -- the vote is a barrier behind the loads, 0.15 ms on C3 where it never hits
  if (crowded && __syncthreads_and(!active || (e - s) > uint32_t(cellrank_max<KEY>))) { if (active) out[p] = id; return; }
  const KEY mine = active ? rank_key<KEY, shuffle>(id, r) : KEY(0);
  const KEY key_e = have_e ? rank_key<KEY, shuffle>(id_e, r) : KEY(0);
  if (threadIdx.x == 0) bounds[0] = s;
  if (p == plast) bounds[1] = e;
  __syncthreads();
  const uint32_t lo = bounds[0], hi = bounds[1];
  const bool staged = (hi - lo) <= uint32_t(cr_cap<KEY>);
  if (staged) {
    if (active) lds[p - lo] = mine;
    if (have_e && pe >= lo && pe < hi) lds[pe - lo] = key_e;
    // what the speculative window does not cover (cells above SPEC super-droplets)
    if (p0 >= r_lo + size_t(SPEC))
      for (size_t q = size_t(lo) + threadIdx.x; q < p0 - SPEC; q += BS) lds[q - lo] = rank_key<KEY, shuffle>(in[q], r);  // the first cell's part before the block
    for (size_t q = p0 + BS + SPEC + threadIdx.x; q < hi; q += BS) lds[q - lo] = rank_key<KEY, shuffle>(in[q], r);         // the last cell's part behind it
  }
  __syncthreads();
  if (!active) return;
  const uint32_t cnt = e - s;
  if (cnt > uint32_t(cellrank_max<KEY>)) { out[p] = id; return; }      // listed by k_list_big_cells, sorted by a bitonic network
  uint32_t rank = 0;
  if (staged) {
    const KEY *seg = lds + (s - lo);
    for (uint32_t q = 0; q < cnt; ++q) rank += seg[q] < mine;
  } else {
    for (uint32_t q = s; q < e; ++q) rank += rank_key<KEY, shuffle>(in[q], r) < mine;
  }
  out[s + rank] = id;
}
// Round 4: the shuffled order on the device generator's 32-bit keys, ranked by BUCKETS instead of by counting.  A key's bucket is its
// expected rank in its cell, (key x count) >> 32 -- the keys are uniform, so a bucket holds one key on average.  The buckets of a cell are
// as many as its droplets and lie where the cell's positions lie, so ONE prefix sum over the workgroup's staged range turns the bucket
// counts into final positions: a key goes to (first position of its bucket) + (keys of its bucket that are smaller), the latter found by
// comparing with the one to four keys that share the bucket instead of with all 64 of the cell.  210 vector instructions per wave
// where the counting form (k_cellrank) has 285 (83 -> 50 LDS instructions), five barriers where it has two -- which is why round 3 dropped it when the kernel ran
// alone and was bound by its waits; now it runs next to the per-cell finish and the terminal velocities (Particles::st_rank), whose waves
// fill the barriers, and the vector ALU is what the three share.  Same order as k_cellrank<uint32_t, true> (the keys are unique).
// A workgroup whose first or last cell reaches beyond the speculative window, or whose staged range exceeds the stage, ranks by counting.
// EXTRA (round 6): something else that every droplet of the order needs before the kernel that consumes the ranking -- a coalescence
// substep's pass over the invalid terminal velocities (rank_vt_fix below: one launch less per substep); nothing by default
struct rank_no_extra { __device__ __forceinline__ void operator()(uint32_t, uint32_t) const {} };
template <bool DUMMY = true, class EXTRA = rank_no_extra>
__global__ void __launch_bounds__(BS)
k_cellrank_bkt(size_t n, const uint32_t *sorted_ijk, const uint32_t *cell_start, const uint32_t *in, uint32_t *out, rng_src r,
               rank_range rg = rank_range{nullptr, nullptr, nullptr}, EXTRA extra = EXTRA())
{
  using KEY = uint32_t;
  constexpr int CAP = cr_cap<KEY>;
  static_assert(CAP == 4 * BS, "four counters per lane in the scan");
  __shared__ KEY lds[CAP];                 // the keys grouped by bucket (counting fallback: by position)
  __shared__ uint32_t bkt[CAP];            // bucket counts, then their exclusive prefix sums
  __shared__ uint32_t bounds[4];
  __shared__ uint32_t wsum[BS / WAVE];
  size_t r_lo = 0;
  if (rg.lo) { r_lo = *rg.lo; n = *rg.hi; }
  if (rg.shift) { const uint32_t sh = *rg.shift; in -= sh; out -= sh; sorted_ijk -= sh; }
  const size_t p0 = r_lo + size_t(blockIdx.x) * BS;
  if (p0 >= n) return;
  const size_t plast = (p0 + BS < n ? p0 + BS : n) - 1;
  const size_t p = p0 + threadIdx.x;
  const bool active = p < n;
  uint32_t c = 0, s = 0, e = 0, id = 0;
  constexpr int SPEC = BS / 2;             // (see k_cellrank)
  const size_t pe = threadIdx.x < SPEC ? p0 - SPEC + threadIdx.x : p0 + BS + (threadIdx.x - SPEC);
  const bool have_e = (threadIdx.x < SPEC ? p0 >= r_lo + (size_t(SPEC) - threadIdx.x) : true) && pe < n;
  uint32_t id_e = 0;
  if (have_e) id_e = in[pe];
  if (active) { c = sorted_ijk[p]; id = in[p]; s = cell_start[c]; e = cell_start[c + 1]; }
  if (active) extra(id, c);
#pragma unroll
  for (int k = 0; k < CAP / BS; ++k) bkt[threadIdx.x + k * BS] = 0u;
  const KEY mine = active ? shuffle_un(id, r.s1, r.s2) : KEY(0);
  const KEY key_e = have_e ? shuffle_un(id_e, r.s1, r.s2) : KEY(0);
  if (threadIdx.x == 0) { bounds[0] = s; bounds[2] = e; }
  if (p == plast) { bounds[1] = e; bounds[3] = s; }
  __syncthreads();
  const uint32_t lo = bounds[0], hi = bounds[1], e_first = bounds[2], s_last = bounds[3];
  const uint32_t m = hi - lo;
  const uint32_t cnt = e - s;
  const bool by_buckets = m <= uint32_t(CAP) && size_t(lo) + SPEC >= p0 && size_t(hi) <= p0 + BS + SPEC;
  if (by_buckets) {
    uint32_t bi = 0, slot = 0, bi_e = 0, slot_e = 0;
    const bool ins_e = have_e && pe >= lo && pe < hi;
    if (active) { bi = (s - lo) + __umulhi(mine, cnt); slot = atomicAdd(&bkt[bi], 1u); }
    if (ins_e) {
      const uint32_t se = pe < p0 ? lo : s_last, ee = pe < p0 ? e_first : hi;
      bi_e = (se - lo) + __umulhi(key_e, ee - se); slot_e = atomicAdd(&bkt[bi_e], 1u);
    }
    __syncthreads();
    // exclusive prefix sum of the m counters, four per lane
    uint32_t v[4], tsum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = bkt[4 * threadIdx.x + k]; tsum += v[k]; }     // (counters behind m are zero)
    const uint32_t incl = wave_inclusive_scan(tsum);
    if (lane_id() == WAVE - 1) wsum[wave_id()] = incl;
    __syncthreads();
    uint32_t run = incl - tsum;
#pragma unroll
    for (int w = 0; w < BS / WAVE; ++w) if (w < int(wave_id())) run += wsum[w];
#pragma unroll
    for (int k = 0; k < 4; ++k) { bkt[4 * threadIdx.x + k] = run; run += v[k]; }
    __syncthreads();
    if (active) lds[bkt[bi] + slot] = mine;
    if (ins_e) lds[bkt[bi_e] + slot_e] = key_e;
    __syncthreads();
    if (!active) return;
    if (cnt > uint32_t(cellrank_max<KEY>)) { out[p] = id; return; }      // listed by k_list_big_cells, sorted by a bitonic network
    const uint32_t b0 = bkt[bi], b1 = bi + 1 < m ? bkt[bi + 1] : m;
    uint32_t rank = b0;
    for (uint32_t q = b0; q < b1; ++q) rank += lds[q] < mine;
    out[lo + rank] = id;
    return;
  }
  // counting, as k_cellrank
  const bool staged = m <= uint32_t(CAP);
  if (staged) {
    if (active) lds[p - lo] = mine;
    if (have_e && pe >= lo && pe < hi) lds[pe - lo] = key_e;
    if (p0 >= r_lo + size_t(SPEC))
      for (size_t q = size_t(lo) + threadIdx.x; q < p0 - SPEC; q += BS) lds[q - lo] = shuffle_un(in[q], r.s1, r.s2);
    for (size_t q = p0 + BS + SPEC + threadIdx.x; q < hi; q += BS) lds[q - lo] = shuffle_un(in[q], r.s1, r.s2);
  }
  __syncthreads();
  if (!active) return;
  if (cnt > uint32_t(cellrank_max<KEY>)) { out[p] = id; return; }
  uint32_t rank = 0;
  if (staged) {
    const KEY *seg = lds + (s - lo);
    for (uint32_t q = 0; q < cnt; ++q) rank += seg[q] < mine;
  } else {
    for (uint32_t q = s; q < e; ++q) rank += shuffle_un(in[q], r.s1, r.s2) < mine;
  }
  out[s + rank] = id;
}
// the cells with more than `thr` SDs: wave-aggregated append (one atomic per wave of 64 cells, not one per cell -- with
// 512 SDs in every cell a per-cell atomicAdd on one counter cost more than the sort itself)
// counts != nullptr: straight from the cell histogram (before the scan: the fused move has just produced it), else from the CSR offsets
// c_first: the launch covers the cells [c_first, n_cell)
__global__ void k_list_big_cells(size_t n_cell, const uint32_t *cell_start, uint32_t thr, uint32_t *big_list, uint32_t *big_count, uint32_t *big_max,
                                 const uint32_t *counts = nullptr, uint32_t c_first = 0u)
{
  const size_t c = size_t(c_first) + gid();
  uint32_t cnt = 0;
  if (c < n_cell) cnt = counts ? counts[c] : cell_start[c + 1] - cell_start[c];
  const bool big = cnt > thr;
  const unsigned long long bal = __ballot(big);
  if (!bal) return;
  uint32_t mx = cnt;
#pragma unroll
  for (int d = WAVE / 2; d > 0; d >>= 1) { const uint32_t o = __shfl_down(mx, d); mx = o > mx ? o : mx; }
  uint32_t base = 0;
  if (lane_id() == 0) { base = atomicAdd(big_count, uint32_t(__popcll(bal))); atomicMax(big_max, mx); }
  base = __shfl(base, 0);
  if (big) big_list[base + __popcll(bal & ((1ull << lane_id()) - 1ull))] = uint32_t(c);
}
// listed segments of up to CELLSORT_WAVE_MAX keys (e.g. 512 SDs per cell): ONE WAVE per segment, bitonic network on a
// power-of-two padded copy in a wave-private LDS slice.  No workgroup barriers: a wave executes its LDS instructions in
// order, so only the compiler has to be kept from moving them across a pass (wave-scope fence); the four waves of a
// workgroup sort four cells independently.
constexpr int CELLSORT_WAVE_MAX = 1024;
// one wave sorts a[0, cnt) in LDS ascending, cnt <= P = a power of two.  Bitonic network in its normalised form (every
// comparator ascending; the first pass of a merge pairs i with its mirror in the block): comparators that reach beyond cnt
// would only meet virtual +inf padding and are skipped.  No workgroup barrier: a wave executes its LDS instructions in order.
template <class KEY>
__device__ __forceinline__ void wave_bitonic(KEY *a, uint32_t cnt, uint32_t P, uint32_t lane)
{
  for (uint32_t k = 2; k <= P; k <<= 1) {
    const uint32_t hk = k >> 1;
    for (uint32_t t = lane; t < (P >> 1) && t < cnt; t += WAVE) {
      const uint32_t base = (t & ~(hk - 1)) << 1, off = t & (hk - 1);
      const uint32_t lo = base | off, hi = base + (k - 1 - off);
      if (hi < cnt) { const KEY u = a[lo], v = a[hi]; if (u > v) { a[lo] = v; a[hi] = u; } }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    for (uint32_t j = k >> 2; j > 0; j >>= 1) {
      for (uint32_t t = lane; t < (P >> 1) && t < cnt; t += WAVE) {
        const uint32_t lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
        if (hi < cnt) { const KEY u = a[lo], v = a[hi]; if (u > v) { a[lo] = v; a[hi] = u; } }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    }
  }
}
// Shuffle keys ((un << 32) | id, un uniform in 32 bits) are ranked by BUCKETS instead of a network: the top 9 bits of un spread a
// cell's keys over 512 buckets (one or two keys each at 512 SDs per cell); an LDS histogram + one wave scan give every bucket its
// first rank, and a key's rank is that plus the number of smaller keys in its own bucket.  ~12 LDS operations per key against ~90
// for the bitonic network of 512..1024 keys (C5: 18.5 -> see DESIGN.md); keys are unique, so the order does not depend on the
// arrival order of the atomics.  A degenerate un (a replayed constant array) only makes the in-bucket loop long, never wrong.
constexpr int CSW_BUCKETS = 512;
// U32 (round 4): the device generator's shuffle keys are a bijection of the ids (rng_src), so 32 bits order a cell by themselves -- half the
// LDS of the 64-bit (un << 32) | id form and one-word compares, the same order; the ids wait in registers (C5: 7.4 -> see DESIGN.md)
template <class KEY, bool U32 = false>
__global__ void __launch_bounds__(BS)
k_cellsort_wave(const uint32_t *big_list, uint32_t n_big, const uint32_t *cell_start, uint32_t *sorted_id, rng_src r)
{
  constexpr int shuffle = sizeof(KEY) == 8 || U32;
  static_assert(!U32 || sizeof(KEY) == 4, "the 32-bit shuffle keys");
  constexpr int PER = CELLSORT_WAVE_MAX / WAVE;
  __shared__ KEY lds[BS / WAVE][CELLSORT_WAVE_MAX];
  __shared__ uint32_t bkt[shuffle ? BS / WAVE : 1][shuffle ? CSW_BUCKETS : 1];
  __shared__ uint16_t idx[shuffle ? BS / WAVE : 1][shuffle ? CELLSORT_WAVE_MAX : 1];
  KEY *a = lds[wave_id()];
  const uint32_t lane = lane_id();
  auto wave_sync = [] { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); };
  for (uint32_t b = blockIdx.x * (BS / WAVE) + wave_id(); b < n_big; b += gridDim.x * (BS / WAVE)) {
    const uint32_t c = big_list ? big_list[b] : b, start = cell_start[c], cnt = cell_start[c + 1] - start;      // (no list: every cell in turn)
    if (cnt > uint32_t(CELLSORT_WAVE_MAX) || cnt < 2u) continue;         // k_cellsort_lds / k_cellsort_big take it
    if constexpr (shuffle) {
      uint32_t *cb = bkt[wave_id()];
      uint16_t *ix = idx[wave_id()];
      constexpr int CPL = CSW_BUCKETS / WAVE;                // counters per lane in the scan
      constexpr int SH = int(sizeof(KEY)) * 8 - 9;
      static_assert(CSW_BUCKETS == 512, "the bucket is the key's top 9 bits");
#pragma unroll
      for (int j = 0; j < CPL; ++j) cb[lane + j * WAVE] = 0u;
      wave_sync();
      uint32_t slot[PER];
      uint32_t idr[U32 ? PER : 1];
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const uint32_t i = lane + k * WAVE;
        if (i < cnt) {
          const uint32_t id = sorted_id[start + i];
          KEY key;
          if constexpr (U32) { key = shuffle_un(id, r.s1, r.s2); idr[k] = id; } else key = KEY(sort_key(id, shuffle, r));
          a[i] = key; slot[k] = atomicAdd(&cb[uint32_t(key >> SH)], 1u);
        }
      }
      wave_sync();
      uint32_t loc[CPL], sum = 0;
#pragma unroll
      for (int j = 0; j < CPL; ++j) { loc[j] = cb[lane * CPL + j]; sum += loc[j]; }
      const uint32_t incl = wave_inclusive_scan(sum);
      uint32_t run = incl - sum;
      wave_sync();
#pragma unroll
      for (int j = 0; j < CPL; ++j) { cb[lane * CPL + j] = run; run += loc[j]; }       // first rank of every bucket
      wave_sync();
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const uint32_t i = lane + k * WAVE;
        if (i < cnt) ix[cb[uint32_t(a[i] >> SH)] + slot[k]] = uint16_t(i);
      }
      wave_sync();
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const uint32_t i = lane + k * WAVE;
        if (i < cnt) {
          const KEY key = a[i];
          const uint32_t bk = uint32_t(key >> SH), s = cb[bk], e = bk + 1 < uint32_t(CSW_BUCKETS) ? cb[bk + 1] : cnt;
          uint32_t rank = s;
          for (uint32_t q = s; q < e; ++q) rank += a[ix[q]] < key;
          if constexpr (U32) sorted_id[start + rank] = idr[k]; else sorted_id[start + rank] = uint32_t(key);
        }
      }
      wave_sync();
    } else {
      uint32_t P = 2 * WAVE; while (P < cnt) P <<= 1;
      for (uint32_t i = lane; i < cnt; i += WAVE) a[i] = KEY(sort_key(sorted_id[start + i], shuffle, r));
      wave_sync();
      wave_bitonic(a, cnt, P, lane);
      for (uint32_t i = lane; i < cnt; i += WAVE) sorted_id[start + i] = uint32_t(a[i]);
      wave_sync();
    }
  }
}
// Round 5: the crowded cells' shuffled order on the device generator's 32-bit keys (C5: every cell, 512 keys each), one wave per cell,
// the scheme of k_cellrank_bkt inside a wave.  A key's bucket is HALF its expected rank in its cell, (key x count) >> 33: two keys per bucket
// on average whatever the cell holds (k_cellsort_wave's 512 fixed buckets are one or two keys at 512 per cell and ten at 1024); the keys
// are written GROUPED BY BUCKET behind one wave scan of the bucket counts, and a key's rank is the first position of its bucket + the
// smaller keys inside it.  Against k_cellsort_wave<uint32_t, true>: no position-ordered copy of the keys and no index array in LDS (the
// keys wait in registers, the bucket's members are read where they lie) -- eight LDS operations per key instead of fourteen, 6 KB of
// LDS per wave instead of 8.  big_list == nullptr: every cell of [0, n_big) in turn (round 5: a box whose cells are crowded on average
// is not listed at all -- the list cost C5 0.75 ms of single-address atomics, and round 4 copied every id through k_cellrank first).
// Cells above CELLSORT_WAVE_MAX are left to k_cellsort_lds / k_cellsort_big as before.  The same order: the keys are unique.
// PER: keys per lane at most -- the launch takes the cells of more than cnt_above and at most PER x 64 keys.  Ten serve C5's cells (512
// +- 60) from 68 vector registers, six waves per SIMD; sixteen (a wave's capacity) need 116: the host launches both, the second finds
// next to nothing to do.
template <int PER>
__global__ void __launch_bounds__(BS)
k_cellsort_wave_bkt(const uint32_t *big_list, uint32_t n_big, const uint32_t *cell_start, uint32_t *sorted_id, rng_src r, uint32_t cnt_above)
{
  static_assert(PER * WAVE <= CELLSORT_WAVE_MAX, "a wave's stage");
  constexpr int NB = CELLSORT_WAVE_MAX / 2;                // buckets at most
  constexpr int CPL = NB / WAVE;                           // counters per lane in the scan
  __shared__ uint32_t grp[BS / WAVE][CELLSORT_WAVE_MAX];   // the keys grouped by bucket
  __shared__ uint32_t bkt[BS / WAVE][NB];                  // bucket counts, then their exclusive prefix sums
  uint32_t *g = grp[wave_id()], *cb = bkt[wave_id()];
  const uint32_t lane = lane_id();
  auto wave_sync = [] { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); };
  for (uint32_t b = blockIdx.x * (BS / WAVE) + wave_id(); b < n_big; b += gridDim.x * (BS / WAVE)) {
    const uint32_t c = big_list ? big_list[b] : b, start = cell_start[c], cnt = cell_start[c + 1] - start;
    if (cnt <= cnt_above || cnt > uint32_t(PER * WAVE)) continue;
    const uint32_t nb = (cnt + 1u) >> 1;
#pragma unroll
    for (int j = 0; j < CPL; ++j) cb[lane + j * WAVE] = 0u;
    uint32_t key[PER], idr[PER], slot[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) { const uint32_t i = lane + k * WAVE; idr[k] = i < cnt ? sorted_id[start + i] : 0u; }
    wave_sync();
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const uint32_t i = lane + k * WAVE;
      if (i < cnt) { key[k] = shuffle_un(idr[k], r.s1, r.s2); slot[k] = atomicAdd(&cb[__umulhi(key[k], cnt) >> 1], 1u); }
    }
    wave_sync();
    uint32_t loc[CPL], sum = 0;
#pragma unroll
    for (int j = 0; j < CPL; ++j) { loc[j] = cb[lane * CPL + j]; sum += loc[j]; }      // (counters behind nb are zero)
    const uint32_t incl = wave_inclusive_scan(sum);
    uint32_t run = incl - sum;
    wave_sync();
#pragma unroll
    for (int j = 0; j < CPL; ++j) { cb[lane * CPL + j] = run; run += loc[j]; }         // first position of every bucket
    wave_sync();
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const uint32_t i = lane + k * WAVE;
      if (i < cnt) { slot[k] += cb[__umulhi(key[k], cnt) >> 1]; g[slot[k]] = key[k]; }
    }
    wave_sync();
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const uint32_t i = lane + k * WAVE;
      if (i < cnt) {
        const uint32_t bk = __umulhi(key[k], cnt) >> 1, s = cb[bk], e = bk + 1u < nb ? cb[bk + 1u] : cnt;
        uint32_t rank = s;
        for (uint32_t q = s; q < e; ++q) rank += g[q] < key[k];
        sorted_id[start + rank] = idr[k];
      }
    }
    wave_sync();
  }
}
// listed segments of up to CELLSORT_LDS_MAX keys: one workgroup per segment, bitonic network
// on a power-of-two padded copy in LDS; each thread does one compare-exchange per pass per 2*BS keys
template <class KEY>
__global__ void __launch_bounds__(BS)
k_cellsort_lds(const uint32_t *big_list, uint32_t n_big, const uint32_t *cell_start, uint32_t *sorted_id, rng_src r)
{
  constexpr int shuffle = sizeof(KEY) == 8;
  __shared__ KEY a[CELLSORT_LDS_MAX];
  for (uint32_t b = blockIdx.x; b < n_big; b += gridDim.x) {
    const uint32_t c = big_list[b], start = cell_start[c], cnt = cell_start[c + 1] - start;
    if (cnt > uint32_t(CELLSORT_LDS_MAX) || cnt <= uint32_t(CELLSORT_WAVE_MAX)) continue;   // k_cellsort_big / k_cellsort_wave take it
    uint32_t P = 2; while (P < cnt) P <<= 1;
    for (uint32_t i = threadIdx.x; i < cnt; i += BS) a[i] = KEY(sort_key(sorted_id[start + i], shuffle, r));
    __syncthreads();
    for (uint32_t k = 2; k <= P; k <<= 1) {                   // normalised network, comparators beyond cnt skipped (see k_cellsort_wave)
      const uint32_t hk = k >> 1;
      for (uint32_t t = threadIdx.x; t < (P >> 1) && t < cnt; t += BS) {
        const uint32_t base = (t & ~(hk - 1)) << 1, off = t & (hk - 1);
        const uint32_t lo = base | off, hi = base + (k - 1 - off);
        if (hi < cnt) { const KEY u = a[lo], v = a[hi]; if (u > v) { a[lo] = v; a[hi] = u; } }
      }
      __syncthreads();
      for (uint32_t j = k >> 2; j > 0; j >>= 1) {
        for (uint32_t t = threadIdx.x; t < (P >> 1) && t < cnt; t += BS) {
          const uint32_t lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
          if (hi < cnt) { const KEY u = a[lo], v = a[hi]; if (u > v) { a[lo] = v; a[hi] = u; } }
        }
        __syncthreads();
      }
    }
    for (uint32_t i = threadIdx.x; i < cnt; i += BS) sorted_id[start + i] = uint32_t(a[i]);
    __syncthreads();
  }
}
// still larger segments (e.g. a 0-D parcel): one workgroup per segment, bitonic network on a
// power-of-two padded scratch copy in global memory
__global__ void __launch_bounds__(1024)
k_cellsort_big(const uint32_t *big_list, uint32_t n_big, const uint32_t *cell_start, uint32_t *sorted_id, int shuffle, rng_src r,
               uint64_t *scratch, size_t scratch_stride)
{
  uint64_t *a = scratch + size_t(blockIdx.x) * scratch_stride;
  for (uint32_t b = blockIdx.x; b < n_big; b += gridDim.x) {
    const uint32_t c = big_list[b], start = cell_start[c], cnt = cell_start[c + 1] - start;
    if (cnt <= uint32_t(CELLSORT_LDS_MAX)) continue;         // done by k_cellsort_lds
    size_t P = 1; while (P < cnt) P <<= 1;
    for (size_t i = threadIdx.x; i < P; i += blockDim.x) a[i] = i < cnt ? sort_key(sorted_id[start + i], shuffle, r) : ~0ull;
    __syncthreads();
    for (size_t k = 2; k <= P; k <<= 1)
      for (size_t j = k >> 1; j > 0; j >>= 1) {
        for (size_t i = threadIdx.x; i < P; i += blockDim.x) {
          const size_t p = i ^ j;
          if (p > i) {
            const uint64_t u = a[i], v = a[p];
            const bool up = (i & k) == 0;
            if ((u > v) == up) { a[i] = v; a[p] = u; }
          }
        }
        __syncthreads();
      }
    for (size_t i = threadIdx.x; i < cnt; i += blockDim.x) sorted_id[start + i] = uint32_t(a[i]);
    __syncthreads();
  }
}

// ============================================================================================
// terminal velocity (hskpng_vterm.ipp:15-342, init_vterm.ipp:36-59)
// ============================================================================================
struct vt_cfg { int formula; double ln_r_min, ln_r_max; int n_bin; };

template <class T>
__global__ void k_init_vt0(T *vt_0, vt_cfg v)
{
  const size_t it = gid(); if (it >= size_t(v.n_bin)) return;
  const T lnmin = T(v.ln_r_min), dlnr = (T(v.ln_r_max) - T(v.ln_r_min)) / v.n_bin;
  const T r = exp(lnmin + (int(it) + 0.5) * dlnr);         // bin_mid: exp(real + double*real) as in the reference
  vt_0[it] = T(vt_beard77_v0(double(r)));
}
template <class T>
__device__ __forceinline__ T vt_eval(const vt_cfg &v, T rw2, T Tk, T p, T rhod, T eta, const T *vt_0)
{
  const T r = sqrt(rw2);
  switch (v.formula) {
    case LCX_VT_BEARD76: return vt_beard76(r, Tk, p, rhod, eta);
    case LCX_VT_BEARD77: return vt_beard77_fact(r, p, rhod, eta) * T(vt_beard77_v0(double(r)));
    case LCX_VT_BEARD77FAST: {
      const T lnmin = T(v.ln_r_min), lnmax = T(v.ln_r_max), dlnr = (lnmax - lnmin) / v.n_bin;
      const T lnr = .5 * log(rw2);
      const int bin = lnr <= lnmin ? 0 : lnr >= lnmax ? v.n_bin - 1 : int((lnr - lnmin) / dlnr);
      return vt_beard77_fact(r, p, rhod, eta) * vt_0[bin];
    }
    case LCX_VT_KHVOROSTYANOV_SPHERICAL: return T(vt_khvorostyanov(double(r), double(rhod), double(eta), true));
    case LCX_VT_KHVOROSTYANOV_NONSPHERICAL: return T(vt_khvorostyanov(double(r), double(rhod), double(eta), false));
    default: return T(0);
  }
}
// beard77 / beard77fast: the cell-only part of the correction factor (two square roots and four divisions of every
// evaluation) is computed once per cell; k_vterm_b77 then needs one load per SD of it instead of p, rhod, eta
template <class T>
__global__ void k_vterm_cellpre(size_t n_cell, const T *p, const T *rhod, const T *eta, beard77_cell<T> *out)
{ const size_t c = gid(); if (c < n_cell) out[c] = vt_beard77_cellpart(p[c], rhod[c], eta[c]); }
// hskpng_Tpr of step_async + the cell part of beard77 in one launch
template <class T>
__global__ void k_cell_Tpr_vtpre(size_t n_cell, const T *th, const T *rhod, const T *rv, T *p, T *Tk, T *RH, T *eta, T *dv,
                                 int th_dry, int const_p, int RH_formula, int ndims, beard77_cell<T> *out)
{
  const size_t c = gid(); if (c >= n_cell) return;
  const T t = th_dry ? theta_dry_T(th[c], rhod[c]) : T(th[c] * exner(p[c]));
  Tk[c] = t;
  T pp = p[c];
  if (!const_p) { pp = theta_dry_p(rhod[c], rv[c], t); p[c] = pp; }
  RH[c] = RH_of(RH_formula, pp, rv[c], t);
  const T et = visc(t);
  eta[c] = et;
  if (ndims == 0) dv[c] = T(1) / rhod[c];
  out[c] = vt_beard77_cellpart(pp, rhod[c], et);
}
// FAST (opts_init.strict_fp == 0): reciprocal square root, refined-reciprocal division and the lean logarithm (<= 2 ulp each)
// one super-droplet of the pass below
template <class T, bool FAST, bool TABLE>
__device__ __forceinline__ bool vterm_b77_one(const vt_cfg &v, T r2, uint32_t c, const beard77_cell<T> *pre, const T *vt_0, T &out)
{
  if (!(r2 > T(0)) || c == DEAD_CELL) return false;
  T r, f;
  if (FAST) {
    const T irw = rsqrt(r2);
    r = r2 * irw;
    f = r <= T(20e-6) ? vt_beard77_fact_pre_small_fast(irw, pre[c]) : vt_beard77_fact_pre(r, pre[c]);
  } else {
    r = sqrt(r2);
    f = vt_beard77_fact_pre(r, pre[c]);
  }
  if constexpr (!TABLE) { out = f * T(vt_beard77_v0(double(r))); return true; }      // beard77: the polynomial itself
  else {                                                                            // beard77fast: its table over ln r
    const T lnmin = T(v.ln_r_min), lnmax = T(v.ln_r_max), dlnr = (lnmax - lnmin) / v.n_bin;
    const T lnr = .5 * (FAST ? log_lean(r2) : T(log(r2)));
    const int bin = lnr <= lnmin ? 0 : lnr >= lnmax ? v.n_bin - 1 : int((lnr - lnmin) / dlnr);
    out = f * vt_0[bin];
    return true;
  }
}
// Several super-droplets per lane, all of their loads issued before the arithmetic: the pass is bound by memory latency (81 % of its
// wave-cycles were waits with one 8-byte element per lane).  A lane takes VT_CHUNKS pairs of neighbours, one pair from each
// 2*BS-element chunk of its workgroup's range, so that every load instruction of a wave is one contiguous 1 KiB (rw2, vt) or 512 B (ijk).
// C3: 1.05 ms with one element per lane, 0.82 ms with one pair, 1.31 ms with four CONSECUTIVE elements (strided lanes).
constexpr int VT_CHUNKS = 2;
template <class T, bool FAST, bool TABLE>
__global__ void __launch_bounds__(BS)
k_vterm_b77(size_t n, int only_invalid, vt_cfg v, const T *rw2, const uint32_t *ijk, const beard77_cell<T> *pre, const T *vt_0, T *vt)
{
  struct alignas(2 * sizeof(T)) pairT { T a, b; };
  struct alignas(8) pairU { uint32_t a, b; };
  const size_t base = size_t(blockIdx.x) * (2 * BS * VT_CHUNKS) + 2 * threadIdx.x;
  pairT r2[VT_CHUNKS], o[VT_CHUNKS]; pairU c[VT_CHUNKS];
#pragma unroll
  for (int k = 0; k < VT_CHUNKS; ++k) {
    const size_t i = base + size_t(k) * 2 * BS;
    // (the old velocities are read only by the pass that refreshes the invalid ones: a third of the all-droplets pass's reads otherwise)
    if (i + 1 < n) {
      r2[k] = *reinterpret_cast<const pairT *>(rw2 + i); c[k] = *reinterpret_cast<const pairU *>(ijk + i);
      o[k] = only_invalid ? *reinterpret_cast<const pairT *>(vt + i) : pairT{T(0), T(0)};
    } else if (i < n) { r2[k] = pairT{rw2[i], T(0)}; c[k] = pairU{ijk[i], DEAD_CELL}; o[k] = pairT{only_invalid ? vt[i] : T(0), T(0)}; }
    else { r2[k] = pairT{T(0), T(0)}; c[k] = pairU{DEAD_CELL, DEAD_CELL}; o[k] = r2[k]; }
  }
#pragma unroll
  for (int k = 0; k < VT_CHUNKS; ++k) {
    const size_t i = base + size_t(k) * 2 * BS;
    if (i >= n) break;
    T va = o[k].a, vb = o[k].b;
    const bool wa = !(only_invalid && !(va == T(-1))) && vterm_b77_one<T, FAST, TABLE>(v, r2[k].a, c[k].a, pre, vt_0, va);
    const bool wb = i + 1 < n && !(only_invalid && !(vb == T(-1))) && vterm_b77_one<T, FAST, TABLE>(v, r2[k].b, c[k].b, pre, vt_0, vb);
    if (wa && wb) *reinterpret_cast<pairT *>(vt + i) = pairT{va, vb};
    else if (wa) vt[i] = va;
    else if (wb) vt[i + 1] = vb;
  }
}
template <class T>
__global__ void k_vterm(size_t n, int only_invalid, vt_cfg v, const T *rw2, const uint32_t *ijk, const T *Tk, const T *p,
                        const T *rhod, const T *eta, const T *vt_0, T *vt)
{
  const size_t i = gid(); if (i >= n) return;
  const T r2 = rw2[i];
  if (!(r2 > T(0))) return;
  if (only_invalid && !(vt[i] == T(-1))) return;
  const uint32_t c = ijk[i];
  if (c == DEAD_CELL) return;
  vt[i] = vt_eval(v, r2, Tk[c], p[c], rhod[c], eta[c], vt_0);
}

// hskpng_vterm_invalid for ONE droplet of the sorted order (k_vterm's / k_vterm_b77's expressions, whichever the run's formula takes), as
// the EXTRA of the in-cell ranking that the next coalescence substep waits for anyway (sstp_coal > 1: a launch less per substep)
// b77: 0 k_vterm's formulas (vt_eval), 1 beard77 / 2 beard77fast through the cells' prepared part; fast: opts_init.strict_fp == 0
template <class T> struct rank_vt_fix {
  vt_cfg v; int b77, fast; const T *Tk, *p, *rhod, *eta, *vt_0; const beard77_cell<T> *pre; const T *rw2; T *vt;
  __device__ __forceinline__ void operator()(uint32_t id, uint32_t c) const
  {
    const T r2 = rw2[id];
    if (!(r2 > T(0)) || !(vt[id] == T(-1)) || c == DEAD_CELL) return;
    T out = T(0);
    if (b77 == 0) out = vt_eval(v, r2, Tk[c], p[c], rhod[c], eta[c], vt_0);
    else if (b77 == 1) { if (fast) vterm_b77_one<T, true, false>(v, r2, c, pre, vt_0, out); else vterm_b77_one<T, false, false>(v, r2, c, pre, vt_0, out); }
    else { if (fast) vterm_b77_one<T, true, true>(v, r2, c, pre, vt_0, out); else vterm_b77_one<T, false, true>(v, r2, c, pre, vt_0, out); }
    vt[id] = out;
  }
};

// ============================================================================================
// condensation (percell/particles_impl_cond.ipp:13-139 + moms.ipp:277-350 + update_th_rv.ipp:74-191)
// ============================================================================================
// One lane per position of the cell-sorted order.  Reads: sorted_id, sorted_ijk (coalesced), the SD's
// rw2, rd3, kpa, vt, n (gathers, near-coalesced while storage order ~ cell order), 8 cell fields
// (wave broadcast).  Writes rw2 and the SD's contribution(s) to the 3rd wet moment into position-ordered
// scratch (coalesced) for the order-preserving per-cell sum of k_cond_cellfinish.
// (cond_args: lcx_cond_wq.hpp)
template <class T>
__global__ void k_cond_cellpre(size_t n_cell, const T *rhod, const T *rv, const T *Tk, const T *eta, const T *RH, const T *lambda_D,
                               const T *lambda_K, T RH_max, cond_cell_fast<T> *pre)
{
  const size_t c = gid(); if (c >= n_cell) return;
  pre[c] = make_cond_cell_fast(rhod[c], rv[c], Tk[c], eta[c], lambda_D[c], lambda_K[c], RH[c], RH_max);
}
// Strict arithmetic, and the fast arithmetic with an SGS supersaturation perturbation per droplet (turb_cond); the production
// fast path is k_cond_fast below.  Register budget: 128 VGPRs = 4 waves per SIMD (the kernel wants 136; 3 waves: 10.7 ms,
// 4 waves with 24 B of scratch per lane: 10.1 ms, 5 waves: 13.0 ms).
// (Measured and dropped: this kernel in storage order with the re-sort's scatter on the side, as k_cond_lean -- the same bits, and the
// strict step 19.06 ms against 19.01: at four waves per SIMD and 32 B of scratch the extra loads and stores cost the kernel what the
// re-sort saves, and the per-cell finish pays for two gathers.)
template <class T, bool FAST>
__global__ void __launch_bounds__(BS) __attribute__((amdgpu_waves_per_eu(4, 4))) k_cond(size_t n_part, cond_args<T> a)
{
  const size_t pos = gid_xcd(a.xcd_group); if (pos >= n_part) return;
  const uint32_t id = a.sorted_id[pos], c = a.sorted_ijk[pos];
  const T rw2_old = a.rw2[id];
  const T nn = T(a.n[id]);                                            // n_filtered is a real_t copy of n (moms.ipp:55-61)
  if (a.first) a.m3_before[pos] = rw2_old >= 0 ? nn * (rw2_old * T(sqrt(rw2_old))) : nn * rw2_old;
  const T r = advance_rw2<T, FAST>(rw2_old, a.dt_sub, a.rhod[c], a.rv[c], a.Tk[c], a.eta[c], a.rd3[id], a.kpa[id], a.vt[id],
                             a.lambda_D[c], a.lambda_K[c], a.ssp ? T(a.RH[c] + a.ssp[id]) : a.RH[c], a.RH_max, a.eps, a.cond_mlt, a.n_iter);
  a.rw2[id] = r;
  a.m3_after[pos] = r >= 0 ? nn * (r * T(sqrt(r))) : nn * r;
}
// Per-cell finish: sums the cell's contributions IN SORTED ORDER (the same addition order as the reference's
// serial reduce_by_key), converts to the specific moment and applies the condensational feedback.
// A workgroup owns CF_CELLS consecutive cells = one contiguous range of the position-ordered scratch, which it
// stages in LDS with coalesced loads; then one lane per cell walks its segment in LDS.  (A lane-per-cell walk
// straight from global memory touches 64 different cache lines per load instruction and ran 10x slower.)
constexpr int CF_CELLS = 64;            // cells per workgroup at most; the host lowers it (cf_cells) when cells hold many SDs
constexpr int CF_CAP = 2048;            // reals staged per workgroup (16 KiB of fp64 = 10 workgroups per CU; measured 6144: 1.05 ms, 2048: 0.64, 1024: 0.80)
template <class T, int G = 1>
__device__ __forceinline__ T seg_sum(const T *lds, const T *glob, bool staged, uint32_t base, uint32_t s, uint32_t e)
{
  T acc = 0;
  if (s >= e) return acc;                  // (G > 1: a lane beyond the end of a short segment)
  if (staged) { acc = lds[s - base]; for (uint32_t q = s + G; q < e; q += G) acc = acc + lds[q - base]; }
  else        { acc = glob[s];       for (uint32_t q = s + G; q < e; q += G) acc = acc + glob[q]; }
  return acc;
}
template <class T, int G = 1>
__device__ __forceinline__ T seg_sum_gather(const T *glob, const uint32_t *gather, uint32_t s, uint32_t e)
{
  T acc = 0;
  if (s >= e) return acc;
  acc = glob[gather[s]];
  for (uint32_t q = s + G; q < e; q += G) acc = acc + glob[gather[q]];
  return acc;
}
// moment bookkeeping across substeps + update_th_rv for one cell, given its sums of n rw^3 before / after the substep
template <class T>
__device__ __forceinline__ void cellfinish_apply(size_t c, bool has, T after, T before, const T *dv, const T *rhod, T *rv, T *th,
                                                 const T *Tk, T *rw_mom3, int step, int sstp, int ndims)
{
  T drw;
  if (has && ndims > 0) { after = after / dv[c]; after = after / rhod[c]; }
  if (step == 0) {
    drw = 0;
    if (has) {
      if (ndims > 0) { before = before / dv[c]; before = before / rhod[c]; }
      drw = -before;
    }
    if (!has) rw_mom3[c] = 0;
  } else drw = -rw_mom3[c];
  if (step < sstp - 1) {
    if (has) rw_mom3[c] = after;
    drw = rw_mom3[c] + drw;
  } else if (has) drw = after + drw;
  // update_th_rv
  drw = drw * (cst<T>::rho_w * T(4. / 3) * cst<T>::pi);
  rv[c] = rv[c] - drw;
  th[c] = th[c] - drw * d_th_d_rv(Tk[c], th[c]);
}
// ---- fast arithmetic (opts_init.strict_fp == 0, no SGS supersaturation): the production form of the condensation kernel.
// Per-cell set-up hoisted (k_cond_cellpre); ONE scratch value per droplet, delta[pos] = n (rw_new^3 - rw_old^3), instead of the
// strict path's two (n rw^3 before and after): the cell's change of the third moment is the sum of the deltas, so the finishing
// pass reads half as much, nothing is carried between substeps (rw_mom3) and the difference of two large sums is never formed.
// Measured on C3 (MI355X, fp64, cond + cond_cellfinish per step): round-1 pair 8.45 + 0.62 ms; this form 8.42 + 0.42; with the
// root finder's reciprocals at one Newton step (OPT bit 0) 8.20; with the cube-root series (bit 1) 8.35; both 8.15 + 0.42.
// HBM traffic of this kernel (PMC): 13.0 GB read + 5.4 GB written per launch against 6.4 + 2.1 algorithmic at the start of round 2,
// 11.2 + 3.3 at its end.  2.7 GB of the writes were the 28-36 B of scratch per lane: they went when the library pow left the growth
// rate's rare branch (lcx_math.hpp pow_core) and the kernel fitted 114 VGPRs.  (A variant with 12 B of scratch had moved the same bytes
// as the 28-B one and taken 8.06 instead of 7.92 ms, which is why the scratch was acquitted for a while.)  The rest of the excess is
// the 40 B gathered and 8 B scattered per droplet through sorted_id, which touch whole 64-B lines once droplets have drifted from their
// storage neighbours (half of them after two steps at |C| = 0.3).
// Walking the STORAGE order instead (unit-stride attributes, only the 8-B delta gathered by the finishing pass) removes that
// traffic and was measured as well: 9.28 + 0.63 ms -- the lanes of a wave then sit in different cells, their root finders need
// different numbers of iterations, and the kernel is bound by exactly that (fp64 issue and divergence), not by HBM.
// Also measured and dropped: the per-cell sums fused in as a wave-shuffle segmented scan with one store per (wave, cell)
// fragment (each scan of 6 shuffle steps costs 0.35-0.4 ms, more than the 0.2 ms of stores + 0.4 ms finishing pass it
// replaces: 9.2-9.4 + 0.05 ms); the cell's nine constants re-read from LDS at every evaluation instead of living in 18 VGPRs
// (no scratch, but 9.4 ms); 3 waves per SIMD without scratch 10.2 ms, 5 waves with 116 B of scratch 10.8 ms.
template <class T> __device__ __forceinline__ T rw2_to_rw3_signed(T x) { return x >= 0 ? x * T(sqrt(x)) : x; }
// Stragglers.  A wave is as slow as its slowest lane: the root finder needs 4.0 growth-rate evaluations per droplet on average,
// but 7.3 per wave (C3, step 12: 3 evaluations 28 %, 4: 34 %, 5: 16 %, 6: 10 %, 7: 3.5 %, more: 0.7 %).  The first pass therefore
// runs the root finder with a short iteration budget; a droplet that has not converged inside it is left untouched and its position
// appended to a list (one atomic per wave), and a second, dense pass solves the listed droplets from scratch with the reference's
// budget of 100.  Same arithmetic per droplet either way, hence the same results; the list lives in the `rank` scratch.
// Root-finder iterations per droplet on C3 (bracket updates behind the two end evaluations): 0: 8 %, 1: 30 %, 2: 25 %, 3: 26 %, 4: 7 %,
// 5: 3.7 %, 6+: 0.2 % -- mean 2.1, but 92 % of the waves hold a droplet that needs 5.  Dealing a workgroup's droplets to its waves by the
// count each needed in the PREVIOUS step (a byte per droplet, counting sort in LDS) was built and measured: the count repeats for only
// 51 % of the droplets (83 % within +-1), so the waves stay as uneven as before, and the three barriers + the hint's gather and
// byte-wide scatter cost 0.66 ms (cond 7.30 -> 8.02 ms).
// Measured on C3 (ms per launch pair): no deferral 7.69; budget 6 (= 8 evaluations, 0.7 % deferred) 7.39; 7 and 8: 7.56; 5 (4 % deferred)
// 7.99; 4 (14 %) 8.69 -- dense waves of stragglers pay the maximum over 64 hard droplets, so only the far tail is worth deferring.
// The list is kept in DEFER_SHARDS parts, workgroup b appending to part b % DEFER_SHARDS with the part's own counter (one counter
// for everybody saturates at ~90 appends per microsecond -- with a few per cent of stragglers nearly every wave appends, and the
// first pass went from 7.8 to 25 ms); part s can hold every position of the workgroups that feed it, so it cannot overflow.
// (DEFER_SHARDS, DEFER_CNT_STRIDE: lcx_cond_wq.hpp)
struct cond_defer { uint32_t *list, *count; size_t shard_cap; unsigned budget; };      // budget 0: no deferral (a single pass with the full budget)
// one droplet; returns true when it was set aside for the second pass
template <class T, int OPT, bool SECOND>
__device__ __forceinline__ bool cond_fast_one(size_t pos, const cond_args<T> &a, unsigned budget, volatile T *my)
{
  const uint32_t id = a.sorted_id[pos], c = a.sorted_ijk[pos];
  // Every gather of the droplet is issued HERE, in one batch behind the two index loads: left to itself the compiler sinks them
  // into the branch below and the wave walks five dependent memory levels (sorted_id -> rw2 -> sorted_ijk, rd3, kpa, vt -> the
  // cell's constants ... -> n at the very end), a third of its lifetime at four waves per SIMD.  The empty asm pins the values.
  T rw2_old = a.rw2[id], rd3 = a.rd3[id], kpa = a.kpa[id], vt = a.vt[id];
  T nn = T(a.n[id]);
  cond_cell_fast<T> cc = a.pre[c];
  asm volatile("" : "+v"(rw2_old), "+v"(rd3), "+v"(kpa), "+v"(vt), "+v"(nn), "+v"(cc.Sc), "+v"(cc.Pr), "+v"(cc.lambda_D), "+v"(cc.lambda_K),
               "+v"(cc.A), "+v"(cc.RH_eff), "+v"(cc.c1), "+v"(cc.c2_rho), "+v"(cc.RH_rho_w), "+v"(cc.rhod), "+v"(cc.eta));
  *my = nn;                                                // the multiplicity waits in LDS while the root finder has the registers
  T r = rw2_old;
  bool deferred = false;
  if (rw2_old > 0) {
    cond_fun_fast<T, OPT> ff;
    ff.setup_cell(cc, rw2_old, a.dt_sub, rd3, kpa, vt);
    unsigned left = 1;
    const bool budgeted = !SECOND && budget != 0;
    r = advance_rw2_with(ff, rw2_old, rd3, a.dt_sub, a.eps, a.cond_mlt, budgeted ? budget : a.n_iter, &left);
    deferred = budgeted && left == 0;
    if (!deferred) a.rw2[id] = r;
  }
  if (!deferred) a.m3_after[pos] = *my * (rw2_to_rw3_signed(r) - rw2_to_rw3_signed(rw2_old));
  return deferred;
}
template <class T, int OPT = 3, bool SECOND = false>
__global__ void __launch_bounds__(BS) __attribute__((amdgpu_waves_per_eu(4, 4))) k_cond_fast(size_t n_part, cond_args<T> a, cond_defer df)
{
  __shared__ T stash[BS];
  volatile T *my = &stash[threadIdx.x];
  if (SECOND) {                                            // the listed droplets: workgroup b walks part b % DEFER_SHARDS with the stride
    const unsigned shard = blockIdx.x % DEFER_SHARDS;      // of the workgroups that share it (the host does not know the counts)
    const size_t count = df.count[shard * DEFER_CNT_STRIDE];
    const uint32_t *part = df.list + size_t(shard) * df.shard_cap;
    for (size_t q = size_t(blockIdx.x / DEFER_SHARDS) * BS + threadIdx.x; q < count; q += size_t(gridDim.x / DEFER_SHARDS) * BS)
      cond_fast_one<T, OPT, true>(part[q], a, 0u, my);
    return;
  }
  const size_t pos = gid_xcd(a.xcd_group); if (pos >= n_part) return;
  const bool deferred = cond_fast_one<T, OPT, false>(pos, a, df.budget, my);
  const unsigned long long bal = __ballot(deferred);       // (lanes behind n_part have left: the ballot covers the rest)
  if (bal) {
    const int leader = __ffsll((long long)bal) - 1;
    uint32_t base = 0;
    const unsigned shard = blockIdx.x % DEFER_SHARDS;
    if (int(lane_id()) == leader) base = atomicAdd(df.count + shard * DEFER_CNT_STRIDE, uint32_t(__popcll(bal)));
    base = __shfl(base, leader);
    if (deferred) df.list[size_t(shard) * df.shard_cap + base + __popcll(bal & ((1ull << lane_id()) - 1ull))] = uint32_t(pos);
  }
}

// Fast arithmetic's production kernel since round 3: the growth rate of k_cond_fast under the lean bracketed secant of lcx_math.hpp
// (advance_rw2_lean_with) instead of TOMS748 -- no iteration budget, no second launch, no fold: the iteration counts are short and even.
// Round 4.  The solver's bookkeeping pared down (lcx_math.hpp, lean2_*: the same operations on the droplet's numbers, straight-line
// loop body) and the growth rate's helper functions without the instructions that are identities here (OPT bit 2): 776 -> 671 vector
// instructions per wave in the spin-up steps that rounds 1-3 profiled (637 in the settled box), the same rw2 bit for bit (SOLVER = 1 keeps
// round 3's form for the test that shows it); the launch 3.43 -> 3.0 ms.  At that point the kernel is no longer bound by its vector
// ALU alone: 0.83 busy, and 10.4 GB of traffic (its own 56 B per droplet + the carried scatter's 20) in 2.9 ms is 3.6 of the 4.85 TB/s
// that a copy reaches on the same box.  UNI: a run with ONE hygroscopicity (one dry distribution, no user-set particles) passes it as
// a scalar -- 8 B per droplet that the kernel does not read.
// Measured and dropped on the way (profiles/r04_*):
//  * Dealing a workgroup's droplets to its waves by the iteration count each needed in the previous step (a byte per storage slot, one
//    ballot, two barriers), so that the slow ones share a wave: the count does not repeat well enough -- 96 % of the droplets that needed at
//    most one loop evaluation need at most one again, but a wave of 64 then still holds two or three that do not; the slowest of a wave
//    came down from 2.50 to 2.46 evaluations where a perfect sort of the workgroup gives 1.42.  +34 instructions per wave, 3.00 -> 3.34 ms.
//  * Two passes: every droplet's loop stopped after ONE evaluation, the 9 % that have not converged listed with the loop's state
//    (seven reals, one atomic per wave) and taken up where they stand by a dense second launch -- the same bits; the first pass 637 -> 590
//    instructions per wave and not a microsecond faster (2.98 against 3.00 ms: the memory side), the second pass 0.75 ms of gathers.
// SOLVER: 0 the lean solver in round 4's form, 1 in round 3's (the same bits, kept for the test that shows it), 2 TOMS748 -- the
// reference's iterates on this kernel's growth-rate arithmetic (opts_init.cond_solver = 1), in the same storage-order walk
// Round 5 (late): droplets whose bracket may hold SEVERAL roots (lcx_math.hpp lean2_head, `suspicious`) are not solved here: they
// are listed (one atomic per wave that has any: about 0.1 % of the droplets of bench.py's settled boxes, a tenth of a box whose aerosol is just
// activating) and k_cond_lean_listed takes them through TOMS748 -- the reference's iterates decide which root such a droplet ends on.
// ent: two words per entry -- the kernel's own index of the droplet (storage slot, or position in the sorted order) and the position of
// its change (what the carried scatter made of its rank: the rank buffer itself is the in-cell ranking's OUTPUT, and that ranking runs
// on its side stream next to k_cond_lean_listed); room for every droplet -- a list that could overflow would make WHO is listed depend
// on the order in which the atomics are served; count: the entries
// (struct cond_list: lcx_cond_wq.hpp)
// BUDGET (SOLVER 0 with records, dbg COND_BUDGET): loop trips of this pass as a compile-time number -- the trips are then straight-line
// code (a loop's back edge copies the five reals of its state: 34 vector instructions per wave in the run-time form); 0: lst.budget at
// run time; -1 (production): no budget, the solver as one call -- the kernel as round 5 had it (the budget's code costs the kernel 21
// vector instructions per wave even where it never runs)
template <class T, int OPT = 7, bool UNI = false, int SOLVER = 0, int BUDGET = -1>
__global__ void __launch_bounds__(BS) k_cond_lean(size_t n_part, cond_args<T> a, T kpa_uniform = T(0), cond_list lst = cond_list{nullptr, nullptr, 0, 0u, nullptr, nullptr, 0u})
{
  // a.storage_ijk != nullptr: the droplets are taken in STORAGE order -- n_part is the storage extent, the cell comes from ijk, the
  // attributes and the change (m3_after, storage-indexed; the per-cell finish gathers it through sorted_id) are read and written
  // coalesced.  A droplet's answer does not depend on who computes it, and the sums per cell keep their order.
  const size_t pos = gid_xcd(a.xcd_group); if (pos >= n_part) return;
  uint32_t id, c;
  size_t m3_pos = pos;
  T rw2_old, rd3, vt, kpa = kpa_uniform;
  n_t n_raw;
  cond_cell_fast<T> cc;
  if (a.storage_ijk) {
    // Round 5: TWO dependent memory levels instead of four.  Left to itself the compiler walked  blockDim (a vector load from the
    // dispatch packet) -> ijk -> rank, cell_start -> the scatter's stores -> the droplet's attributes and the cell's constants, each
    // behind a full wait: four round trips of a wave's life before its first fp64 instruction.  Everything the storage slot indexes
    // is issued in ONE batch (a dead slot's attributes are read for nothing: fewer than 1 in 32), everything the cell indexes in a
    // second; the empty asm statements are compiler barriers for memory operations -- loads are neither sunk nor hoisted across them.
    id = uint32_t(pos);
    c = a.storage_ijk[pos];
    uint32_t rk = 0;
    if (a.sc_rank) rk = a.sc_rank[pos];
    rw2_old = a.rw2[pos]; rd3 = a.rd3[pos]; vt = a.vt[pos];
    if (!UNI) kpa = a.kpa[pos];
    n_raw = a.n[pos];
    asm volatile("" ::: "memory");
    if (c == DEAD_CELL) return;
    uint32_t cs = 0;
    if (a.sc_rank) cs = a.sc_cell_start[c];
    cc = a.pre[c];
    asm volatile("" ::: "memory");
    // (the droplet's change goes where the droplet goes in the sorted order: the per-cell finish then reads one stretch, see there)
    if (a.sc_rank) { const size_t q = size_t(cs) + rk; a.sc_sorted_id[q] = id; a.sc_sorted_ijk[q] = c; m3_pos = q; }
  }
  else {
    id = a.sorted_id[pos]; c = a.sorted_ijk[pos];
    rw2_old = a.rw2[id]; rd3 = a.rd3[id]; vt = a.vt[id];
    if (!UNI) kpa = a.kpa[id];
    n_raw = a.n[id];
    cc = a.pre[c];
  }
  T nn = T(n_raw);
  asm volatile("" : "+v"(rw2_old), "+v"(rd3), "+v"(vt), "+v"(nn), "+v"(cc.Sc), "+v"(cc.Pr), "+v"(cc.lambda_D), "+v"(cc.lambda_K),
               "+v"(cc.A), "+v"(cc.RH_eff), "+v"(cc.c1), "+v"(cc.c2_rho), "+v"(cc.RH_rho_w));
  if (!UNI) asm volatile("" : "+v"(kpa));
  if constexpr (cond_fun_fast<T, OPT>::trim) asm volatile("" : "+v"(cc.two_rho_eta)); else asm volatile("" : "+v"(cc.rhod), "+v"(cc.eta));
  T delta = 0;
  bool several = false;                 // (SOLVER 0 with a list: the bracket may hold several roots, see cond_list)
  if (!(rw2_old <= 0)) {                // (cond_common.ipp:197-199; a NaN goes through the solver and poisons its cell as in the reference)
    cond_fun_fast<T, OPT> ff;
    ff.setup_cell(cc, rw2_old, a.dt_sub, rd3, kpa, vt);
    T r;
    if constexpr (SOLVER == 2) r = advance_rw2_with(ff, rw2_old, rd3, a.dt_sub, a.eps, a.cond_mlt, a.n_iter);
    else if constexpr (SOLVER == 1) r = advance_rw2_lean_with(ff, rw2_old, rd3, a.dt_sub, a.eps, a.cond_mlt, a.n_iter);
    else if constexpr (BUDGET < 0) r = advance_rw2_lean2_with(ff, rw2_old, rd3, a.dt_sub, a.eps, a.cond_mlt, a.n_iter, &several, lst.ent != nullptr);
    else {
      // (round 6) with records: the loop has a BUDGET of trips; a droplet that has not converged by then leaves the loop's state where it
      // stands in a record and k_cond_lean_resume goes on with it (lean2_loop is resumable: the same bits) -- the wave no longer waits
      // for its slowest droplet (cond_list)
      lean_state<T> s;
      T rd2 = 0;
      const bool ask = lst.ent != nullptr, budgeted = lst.rec != nullptr;
      if (!lean2_head(ff, rw2_old, rd3, a.dt_sub, a.eps, a.cond_mlt, s, r, rd2, &several, ask)) {
        const unsigned budget = BUDGET > 0 ? unsigned(BUDGET) : budgeted ? lst.budget : a.n_iter;
        bool done = lean2_loop(ff, a.eps, budget, s, r);
        if (budgeted && !done) {
          // (the lanes of this branch that go on: one atomic per wave that has any; `resumed` says that the record found room)
          const unsigned long long bs = __ballot(true);
          const int leader = __ffsll((long long)bs) - 1;
          const unsigned shard = blockIdx.x % DEFER_SHARDS;
          uint32_t base = 0;
          if (int(lane_id()) == leader) base = atomicAdd(lst.rcount + shard * DEFER_CNT_STRIDE, uint32_t(__popcll(bs)));
          base = __shfl(base, leader);
          const uint32_t e = base + uint32_t(__popcll(bs & ((1ull << lane_id()) - 1ull)));
          if (e < lst.rec_cap) {
            lean_record<T> *rc = static_cast<lean_record<T> *>(lst.rec) + (size_t(shard) * lst.rec_cap + e);
            const bool grows = s.b != rw2_old;            // (shrinking: b = rw2_old + 0, growing: a = max(rd2, rw2_old + 0), see lean2_head)
            rc->v[0] = s.x0; rc->v[1] = s.f0; rc->v[2] = s.x1; rc->v[3] = s.f1; rc->v[4] = s.c;
            rc->v[5] = grows ? s.b : s.a; rc->v[6] = grows ? -rd2 : rd2;
            rc->idx = a.storage_ijk ? id : uint32_t(m3_pos); rc->m3_pos = uint32_t(m3_pos);
            return;                                       // (its rw2 and its change: k_cond_lean_resume)
          }
          lean2_loop(ff, a.eps, a.n_iter - budget, s, r);      // (no room: on in its own lane)
        }
        r = lean2_tail(s, r, rd2);
      }
    }
    if (!several) {
      a.rw2[id] = r;
      // n (rw_new^3 - rw_old^3), the radii in the growth rate's own form rw2 * rsqrt(rw2) (its first evaluation has the old one already)
      delta = nn * (r * (r * rsqrt_pos(r)) - rw2_old * (rw2_old * rsqrt_pos(rw2_old)));
    }
  }
  if constexpr (SOLVER == 0) {
    const unsigned long long bal = __ballot(several);
    if (bal) {
      const int leader = __ffsll((long long)bal) - 1;
      uint32_t base = 0;
      const unsigned shard = blockIdx.x % DEFER_SHARDS;
      if (int(lane_id()) == leader) base = atomicAdd(lst.count + shard * DEFER_CNT_STRIDE, uint32_t(__popcll(bal)));
      base = __shfl(base, leader);
      // (the kernel's own index of the droplet, from what is live anyway: the storage slot is `id`, the position in the sorted order `m3_pos`)
      if (several) {
        const size_t q = 2 * (size_t(shard) * lst.shard_cap + base + uint32_t(__popcll(bal & ((1ull << lane_id()) - 1ull))));
        lst.ent[q] = a.storage_ijk ? id : uint32_t(m3_pos); lst.ent[q + 1] = uint32_t(m3_pos);
        return;                                                  // (its change: k_cond_lean_listed)
      }
    }
  }
  a.m3_after[m3_pos] = delta;
}
// MEASUREMENT ONLY (opts_init.dbg_flags & COND_PROBE): k_cond_lean<T, 15, true, 0> cut short at STAGE, launched ahead of the real kernel on
// the same inputs and writing only the scratch m3_before -- the vector instructions that each part of the kernel issues, read off the
// counters of consecutive stages (tools/pmc_steady.sh; profiles/r06_cond_stages.txt).  0: the two load levels and the stores; 1: + the set-up
// and the near end's evaluation; 2: + the rest of the head (bracket, far end, first secant point); 3 / 4 / 5: + one / two / all loop trips; 6: +
// the change of n rw^3 (the whole kernel but its list)
template <class T, int STAGE>
__global__ void __launch_bounds__(BS) k_cond_probe(size_t n_part, cond_args<T> a, T kpa_uniform)
{
  const size_t pos = gid_xcd(a.xcd_group); if (pos >= n_part) return;
  const uint32_t id = uint32_t(pos), c = a.storage_ijk[pos];
  uint32_t rk = 0;
  if (a.sc_rank) rk = a.sc_rank[pos];
  T rw2_old = a.rw2[pos], rd3 = a.rd3[pos], vt = a.vt[pos], kpa = kpa_uniform;
  const n_t n_raw = a.n[pos];
  asm volatile("" ::: "memory");
  if (c == DEAD_CELL) return;
  uint32_t cs = 0;
  if (a.sc_rank) cs = a.sc_cell_start[c];
  cond_cell_fast<T> cc = a.pre[c];
  asm volatile("" ::: "memory");
  const size_t m3_pos = a.sc_rank ? size_t(cs) + rk : pos;
  T nn = T(n_raw);
  asm volatile("" : "+v"(rw2_old), "+v"(rd3), "+v"(vt), "+v"(nn), "+v"(cc.Sc), "+v"(cc.Pr), "+v"(cc.lambda_D), "+v"(cc.lambda_K),
               "+v"(cc.A), "+v"(cc.RH_eff), "+v"(cc.c1), "+v"(cc.c2_rho), "+v"(cc.RH_rho_w));
  asm volatile("" : "+v"(cc.two_rho_eta));
  T out = nn + rw2_old + rd3 + vt + cc.Sc + cc.Pr + cc.lambda_D + cc.lambda_K + cc.A + cc.RH_eff + cc.c1 + cc.c2_rho + cc.RH_rho_w + cc.two_rho_eta + T(id);
  if constexpr (STAGE >= 1) {
    out = nn;
    if (!(rw2_old <= 0)) {
      cond_fun_fast<T, 15> ff;
      ff.setup_cell(cc, rw2_old, a.dt_sub, rd3, kpa, vt);
      if constexpr (STAGE == 1) out = a.dt_sub * ff.drw2_dt(rw2_old);
      else {
        lean_state<T> s;
        T r, rd2 = 0;
        bool several = false;
        if (!lean2_head(ff, rw2_old, rd3, a.dt_sub, a.eps, a.cond_mlt, s, r, rd2, &several, true)) {
          if constexpr (STAGE >= 3) {
            lean2_loop(ff, a.eps, STAGE == 3 ? 1u : STAGE == 4 ? 2u : a.n_iter, s, r);
            r = lean2_tail(s, r, rd2);
          }
        }
        out = r;
        if constexpr (STAGE >= 6) out = nn * (r * (r * rsqrt_pos(r)) - rw2_old * (rw2_old * rsqrt_pos(rw2_old)));
      }
    }
  }
  a.m3_before[m3_pos] = out;
}

// the listed droplets of k_cond_lean through TOMS748 on the same growth-rate arithmetic (what k_cond_lean<T, 15, UNI, 2> computes for them,
// bit for bit): a grid-stride walk over the list, a droplet per lane
template <class T, bool UNI>
__device__ __forceinline__ void cond_lean_listed(const cond_args<T> &a, const cond_list &lst, T kpa_uniform, unsigned block, unsigned n_blocks)
{
  // workgroup b walks part b % DEFER_SHARDS with the stride of the workgroups that share it (the host does not know the counts):
  // consecutive lanes take consecutive entries, and a wave's entries are the droplets of a few neighbouring waves of the first pass
  const unsigned shard = block % DEFER_SHARDS;
  const uint32_t n = lst.count[shard * DEFER_CNT_STRIDE];
  const uint32_t *part = lst.ent + 2 * size_t(shard) * lst.shard_cap;
  for (size_t q = size_t(block / DEFER_SHARDS) * BS + threadIdx.x; q < n; q += size_t(n_blocks / DEFER_SHARDS) * BS) {
    const uint32_t pos = part[2 * q], m3_pos = part[2 * q + 1];
    uint32_t id, c;
    if (a.storage_ijk) { id = pos; c = a.storage_ijk[pos]; }
    else { id = a.sorted_id[pos]; c = a.sorted_ijk[pos]; }
    const T rw2_old = a.rw2[id], rd3 = a.rd3[id], vt = a.vt[id], kpa = UNI ? kpa_uniform : a.kpa[id], nn = T(a.n[id]);
    const cond_cell_fast<T> cc = a.pre[c];
    cond_fun_fast<T, 15> ff;
    ff.setup_cell(cc, rw2_old, a.dt_sub, rd3, kpa, vt);
    const T r = advance_rw2_with(ff, rw2_old, rd3, a.dt_sub, a.eps, a.cond_mlt, a.n_iter);
    a.rw2[id] = r;
    a.m3_after[m3_pos] = nn * (r * (r * rsqrt_pos(r)) - rw2_old * (rw2_old * rsqrt_pos(rw2_old)));
  }
}

// the records of k_cond_lean's first pass (droplets whose loop had not converged within its budget): the loop goes on where it stands,
// a droplet per lane, consecutive lanes on consecutive records of a part -- the droplets of a few neighbouring waves of the first pass.
// The droplet's attributes and its cell's constants are read again and set up with the same expressions: the same bits.
template <class T, bool UNI>
__device__ __forceinline__ void cond_lean_resume(const cond_args<T> &a, const cond_list &lst, T kpa_uniform, unsigned block, unsigned n_blocks)
{
  const unsigned shard = block % DEFER_SHARDS;
  const uint32_t cnt = lst.rcount[shard * DEFER_CNT_STRIDE];
  const uint32_t n = cnt < lst.rec_cap ? cnt : lst.rec_cap;
  const lean_record<T> *part = static_cast<const lean_record<T> *>(lst.rec) + size_t(shard) * lst.rec_cap;
  for (size_t q = size_t(block / DEFER_SHARDS) * BS + threadIdx.x; q < n; q += size_t(n_blocks / DEFER_SHARDS) * BS) {
    const lean_record<T> rc = part[q];
    const uint32_t pos = rc.idx, m3_pos = rc.m3_pos;
    uint32_t id, c;
    if (a.storage_ijk) { id = pos; c = a.storage_ijk[pos]; }
    else { id = a.sorted_id[pos]; c = a.sorted_ijk[pos]; }
    const T rw2_old = a.rw2[id], rd3 = a.rd3[id], vt = a.vt[id], kpa = UNI ? kpa_uniform : a.kpa[id], nn = T(a.n[id]);
    const cond_cell_fast<T> cc = a.pre[c];
    cond_fun_fast<T, 15> ff;
    ff.setup_cell(cc, rw2_old, a.dt_sub, rd3, kpa, vt);
    lean_state<T> s;
    s.x0 = rc.v[0]; s.f0 = rc.v[1]; s.x1 = rc.v[2]; s.f1 = rc.v[3]; s.c = rc.v[4];
    const T far = rc.v[5], rd2s = rc.v[6];
    const bool grows = __builtin_signbit(rd2s);
    const T rd2 = fabs(rd2s);
    s.a = grows ? mx(rd2, rw2_old) : far; s.b = grows ? far : rw2_old;
    T r = s.c;
    lean2_loop(ff, a.eps, a.n_iter - lst.budget, s, r);
    r = lean2_tail(s, r, rd2);
    a.rw2[id] = r;
    a.m3_after[m3_pos] = nn * (r * (r * rsqrt_pos(r)) - rw2_old * (rw2_old * rsqrt_pos(rw2_old)));
  }
}

// (Measured and dropped: both in ONE launch, the first workgroups on the records and the others on the list -- 0.49 ms beside the in-cell
// ranking against 0.26 + 0.12 for the two launches.)
template <class T, bool UNI>
__global__ void __launch_bounds__(BS) k_cond_lean_listed(cond_args<T> a, cond_list lst, T kpa_uniform) { cond_lean_listed<T, UNI>(a, lst, kpa_uniform, blockIdx.x, gridDim.x); }
template <class T, bool UNI>
__global__ void __launch_bounds__(BS) k_cond_lean_resume(cond_args<T> a, cond_list lst, T kpa_uniform) { cond_lean_resume<T, UNI>(a, lst, kpa_uniform, blockIdx.x, gridDim.x); }

// Round 5, measured and kept behind opts_init.dbg_flags & COND_FOLD: the lean kernel with its workgroup FOLDED behind the solver's first
// loop trip.  Evaluations per droplet (below): 1 for the 7 % that take an early out, 3 for 72 % (near end, far end, the first secant point,
// which already confirms itself), 4...8 for the 21 % whose first secant point does not -- but a wave runs until its slowest droplet is
// done: 5.2.  Every lane runs the uniform part (head + ONE loop trip); the droplets that have not converged hand the loop's state
// where it stands (lean_state, the clamps, the droplet's constants: 13 reals + 3 words) through LDS to the lowest lanes of the
// workgroup, three of the four waves leave, and one (now and then two) runs the remaining trips densely.  The same operations on every
// droplet's numbers in the same order as k_cond_lean<T, 7, UNI, 0>: the same rw2 and the same change of n rw^3 bit for bit (lean2_loop
// is resumable, see lcx_math.hpp; tests/test_hip_parity.py).  Droplets beyond the stage's capacity carry on in their own lanes.
// Counters of the settled box (profiles/r05*_pmc_fold.txt): 638 -> 601 vector instructions per wave, fp64 FMAs 205 -> 172, lane use
// 0.66 -> 0.76, vector ALU 0.90 -> 0.85 busy -- and 2.86-2.92 ms against 2.81-2.88 for the plain kernel on the same box.  Like
// round 4's second-pass experiment it removes issued instructions whose lanes were mostly masked off, and those were not what the
// launch costs: the package sits at 1.31 kW of its 1.40 kW cap through the whole step and the shader clock at 2.19 GHz instead of 2.40
// (profiles/r05*_power.txt) -- the launch is priced in lanes that compute (energy), not in instructions that issue.  Storage order only.
constexpr int FOLD_CAP = 128;
// SOLVER = 2 (round 5): the same fold for TOMS748 (opts_init.cond_solver = 1, the API default) -- behind the root finder's HEAD (the far
// end, the secant and the first quadratic step: four evaluations that every droplet makes), where the 37 % of the droplets whose
// bracket is not yet inside the tolerance enter its main loop: toms_carry + the clamp + the droplet's constants, 14 reals + 4 words.
// Unlike the lean solver's, THIS kernel is bound by instruction issue, not by power (vector ALU 0.94 busy at a lane use of 0.50,
// profiles/r05b_pmc_toms.txt: half of the lanes it issues for are idle, and idle lanes draw no power), so emptying three of four waves
// ahead of the loop is time: see DESIGN.md section 4.  The same operations per droplet, the same bits as k_cond_lean<T, 15, UNI, 2>.
// (Measured and dropped: workgroups of 512 threads, so that the 190 droplets that enter the loop fill three dense waves of eight instead
// of 2 x (64 + 31): 6.5 ms against 5.45 -- eight waves meeting at the fold's barriers cost more than the emptier waves did.)
template <class T, bool UNI, int SOLVER = 0>
__global__ void __launch_bounds__(BS) k_cond_lean_fold(size_t n_part, cond_args<T> a, T kpa_uniform = T(0))
{
  constexpr bool TOMS = SOLVER == 2;
  __shared__ T xs[TOMS ? 14 : 13][FOLD_CAP];
  __shared__ uint32_t xw[TOMS ? 4 : 3][FOLD_CAP];
  __shared__ uint32_t wcnt[BS / WAVE];
  const size_t pos = gid_xcd(a.xcd_group);
  bool live = pos < n_part;
  uint32_t id = uint32_t(pos), c = DEAD_CELL, m3_pos = uint32_t(pos);
  T rw2_old = 0, nn = 0, r = 0, rd2 = 0;
  cond_fun_fast<T, 15> ff;
  lean_state<T> s;
  toms_carry<T> k;
  k.count = a.n_iter;
  bool need = false;
  if (live) {
    // (the two load levels of k_cond_lean, see there)
    c = a.storage_ijk[pos];
    uint32_t rk = 0;
    if (a.sc_rank) rk = a.sc_rank[pos];
    rw2_old = a.rw2[pos];
    T rd3 = a.rd3[pos], vt = a.vt[pos], kpa = kpa_uniform;
    if (!UNI) kpa = a.kpa[pos];
    const n_t n_raw = a.n[pos];
    asm volatile("" ::: "memory");
    live = c != DEAD_CELL;
    if (live) {
      uint32_t cs = 0;
      if (a.sc_rank) cs = a.sc_cell_start[c];
      cond_cell_fast<T> cc = a.pre[c];
      asm volatile("" ::: "memory");
      if (a.sc_rank) { m3_pos = cs + rk; a.sc_sorted_id[m3_pos] = id; a.sc_sorted_ijk[m3_pos] = c; }
      nn = T(n_raw);
      asm volatile("" : "+v"(rw2_old), "+v"(rd3), "+v"(vt), "+v"(nn), "+v"(cc.Sc), "+v"(cc.Pr), "+v"(cc.lambda_D), "+v"(cc.lambda_K),
                   "+v"(cc.A), "+v"(cc.RH_eff), "+v"(cc.c1), "+v"(cc.c2_rho), "+v"(cc.RH_rho_w));
      if (!UNI) asm volatile("" : "+v"(kpa));
      asm volatile("" : "+v"(cc.two_rho_eta));
      T delta = 0;
      if (!(rw2_old <= 0)) {
        ff.setup_cell(cc, rw2_old, a.dt_sub, rd3, kpa, vt);
        if constexpr (TOMS) need = !advance_rw2_head_with(ff, rw2_old, rd3, a.dt_sub, a.eps, a.cond_mlt, a.n_iter, k, r, &rd2);
        else if (!lean2_head(ff, rw2_old, rd3, a.dt_sub, a.eps, a.cond_mlt, s, r, rd2)) {
          need = !lean2_loop(ff, a.eps, 1u, s, r);
          if (!need) r = lean2_tail(s, r, rd2);
        }
        if (!need) {
          a.rw2[id] = r;
          delta = nn * (r * (r * rsqrt_pos(r)) - rw2_old * (rw2_old * rsqrt_pos(rw2_old)));
        }
      }
      if (!need) a.m3_after[m3_pos] = delta;
    }
  }
  // ---- the fold: unconverged droplets -> lanes 0 .. total-1 of the workgroup
  const unsigned long long bal = __ballot(need);
  if (lane_id() == 0) wcnt[wave_id()] = uint32_t(__popcll(bal));
  __syncthreads();
  uint32_t first = 0, total = 0;
#pragma unroll
  for (unsigned w = 0; w < BS / WAVE; ++w) { const uint32_t n_w = wcnt[w]; if (w < wave_id()) first += n_w; total += n_w; }
  if (total == 0) return;
  const uint32_t slot = first + uint32_t(__popcll(bal & ((1ull << lane_id()) - 1ull)));
  const uint32_t cap = a.fold_cap;                            // (FOLD_CAP; a test makes it small so that droplets stay behind)
  const bool moved = need && slot < cap;
  if (moved) {
    xw[0][slot] = id; xw[1][slot] = c; xw[2][slot] = m3_pos;
    if constexpr (TOMS) {
      xw[3][slot] = k.count;
      xs[0][slot] = k.s.a; xs[1][slot] = k.s.b; xs[2][slot] = k.s.fa; xs[3][slot] = k.s.fb; xs[4][slot] = k.s.d; xs[5][slot] = k.s.fd;
      xs[6][slot] = k.e; xs[13][slot] = k.fe;
    } else {
      xs[0][slot] = s.x0; xs[1][slot] = s.f0; xs[2][slot] = s.x1; xs[3][slot] = s.f1; xs[4][slot] = s.c; xs[5][slot] = s.a; xs[6][slot] = s.b;
    }
    xs[7][slot] = rd2; xs[8][slot] = rw2_old; xs[9][slot] = ff.rd3; xs[10][slot] = ff.rd3_1mk; xs[11][slot] = ff.c_Re; xs[12][slot] = nn;
  }
  __syncthreads();
  bool run = need && !moved;                                  // (beyond the stage's capacity: carries on where it is)
  const uint32_t t = threadIdx.x;
  if (t < (total < cap ? total : cap)) {
    id = xw[0][t]; c = xw[1][t]; m3_pos = xw[2][t];
    if constexpr (TOMS) {
      k.count = xw[3][t];
      k.s.a = xs[0][t]; k.s.b = xs[1][t]; k.s.fa = xs[2][t]; k.s.fb = xs[3][t]; k.s.d = xs[4][t]; k.s.fd = xs[5][t];
      k.e = xs[6][t]; k.fe = xs[13][t];
    } else {
      s.x0 = xs[0][t]; s.f0 = xs[1][t]; s.x1 = xs[2][t]; s.f1 = xs[3][t]; s.c = xs[4][t]; s.a = xs[5][t]; s.b = xs[6][t];
    }
    rd2 = xs[7][t]; rw2_old = xs[8][t]; nn = xs[12][t];
    // (the droplet's own products arrive as they were formed: the same bits as setup_cell's)
    ff.rd3 = xs[9][t]; ff.rd3_1mk = xs[10][t]; ff.c_Re = xs[11][t];
    if constexpr (!TOMS) r = s.c;
    run = true;
  }
  if (!run) return;
  {
    // the cell's constants are read again by whoever goes on (also by a droplet that stays in its lane): nothing of them lives across
    // the barriers, so that the kernel keeps the 80 vector registers of six waves per SIMD
    const cond_cell_fast<T> cc = a.pre[c];
    ff.rw2_old = rw2_old; ff.dt = a.dt_sub;
    ff.Sc = cc.Sc; ff.Pr = cc.Pr; ff.lambda_D = cc.lambda_D; ff.lambda_K = cc.lambda_K; ff.A = cc.A; ff.RH_eff = cc.RH_eff;
    ff.c1 = cc.c1; ff.c2_rho = cc.c2_rho; ff.RH_rho_w = cc.RH_rho_w;
    if constexpr (decltype(ff)::trim) { ff.Sc = ff.c_Re * cc.Sc; ff.Pr = ff.c_Re * cc.Pr; }      // (setup_cell's products, see cond_fun_fast)
  }
  if (!run) return;
  if constexpr (TOMS) r = advance_rw2_tail_with(ff, ff.rd3, a.eps, k, (unsigned *)nullptr, &rd2);
  else {
    lean2_loop(ff, a.eps, a.n_iter - 1u, s, r);
    r = lean2_tail(s, r, rd2);
  }
  a.rw2[id] = r;
  a.m3_after[m3_pos] = nn * (r * (r * rsqrt_pos(r)) - rw2_old * (rw2_old * rsqrt_pos(rw2_old)));
}

// Growth-rate evaluations per droplet (an offline count over 2.1e6 droplets of the oracle's state on the bench's fields): 3 for 72 % of the
// droplets, 4 for 19 %, 5...8 for 2.4 % (haze far from its equilibrium, droplets activating), 1 for the 7 % that take an early out -- mean
// 3.1, and 5.2 for the slowest droplet of a wave of 64 (TOMS748: 4.1 and 7.7).
// (Measured and dropped: an iteration budget of two secant updates with the 2.4 % of the droplets that exceed it solved by a dense second
// launch, as k_cond_fast does -- bit-identical, and no faster: 4.16 ms against 4.09 for the same kernel without budget and 3.9 for this
// plain form; the tail of the iteration counts is not what the kernel's time is made of.)
// (Measured and dropped: two droplets per lane with all index loads and attribute gathers of both issued up front, so that the second
// droplet's memory latency passes behind the first one's root search -- 126 VGPRs, no scratch, 3.96 against 3.89 ms: the 42 % of
// wave-cycles that the counters show as waiting are not the gathers at the head of the kernel.)

// The first pass with the workgroup FOLDED at the root finder's loop entry: every lane runs the head (the two end evaluations, the secant
// and the first quadratic step: uniform work); the ~37 % of the droplets that enter the loop hand their state (bracket, function values,
// the droplet's own constants: 13 reals + 4 words) through LDS to the lowest lanes of the workgroup, the emptied waves leave, and the
// remaining one or two run the loop densely.  Same arithmetic per droplet, same results as k_cond_fast.
// Measured on C3: 6.96-7.06 ms against 7.22-7.26 for the pair of launches.  The wave-evaluations drop by a fifth (4 waves x 4 + 2 x 3.3
// against 4 x 7), the time by 3-4 %: at four waves per SIMD the kernel runs at the latency of its dependent fp64 chains, and the folded
// workgroups leave their SIMDs with fewer waves to interleave.  Variants: the droplet's own values gathered again instead of moved
// (22 KB of LDS instead of 31): no gain at all; a 160-slot stage with an unfolded fallback (values live across the barrier): 7.34.
template <class T, int OPT = 3>
__global__ void __launch_bounds__(BS) __attribute__((amdgpu_waves_per_eu(4, 4))) k_cond_fast_fold(size_t n_part, cond_args<T> a, cond_defer df)
{
  constexpr int NR = 13;
  __shared__ T xr[NR][BS];                                     // [12] doubles as the lanes' own stash of the multiplicity
  __shared__ uint32_t xw[4][BS];
  __shared__ uint32_t wcnt[BS / WAVE];
  const size_t pos0 = gid_xcd(a.xcd_group);
  const unsigned budget = df.budget ? df.budget : a.n_iter;
  bool need = false;
  toms_carry<T> k;
  k.count = budget;
  T rw2_old = 0, rd3 = 0, kpa = 0, vt = 0;
  uint32_t id = 0, c = 0;
  if (pos0 < n_part) {
    id = a.sorted_id[pos0]; c = a.sorted_ijk[pos0];
    rw2_old = a.rw2[id]; rd3 = a.rd3[id]; kpa = a.kpa[id]; vt = a.vt[id];
    T nn = T(a.n[id]);
    cond_cell_fast<T> cc = a.pre[c];
    asm volatile("" : "+v"(rw2_old), "+v"(rd3), "+v"(kpa), "+v"(vt), "+v"(nn), "+v"(cc.Sc), "+v"(cc.Pr), "+v"(cc.lambda_D), "+v"(cc.lambda_K),
                 "+v"(cc.A), "+v"(cc.RH_eff), "+v"(cc.c1), "+v"(cc.c2_rho), "+v"(cc.RH_rho_w), "+v"(cc.rhod), "+v"(cc.eta));
    volatile T *my = &xr[12][threadIdx.x];
    *my = nn;                                                  // (the multiplicity waits in LDS while the root finder has the registers)
    T r = rw2_old;
    if (rw2_old > 0) {
      cond_fun_fast<T, OPT> ff;
      ff.setup_cell(cc, rw2_old, a.dt_sub, rd3, kpa, vt);
      need = !advance_rw2_head_with(ff, rw2_old, rd3, a.dt_sub, a.eps, a.cond_mlt, budget, k, r);
      if (df.budget && k.count == 0) need = true;              // (a budget the head itself exhausts: the loop below defers it)
      if (!need) a.rw2[id] = r;
    }
    if (!need) a.m3_after[pos0] = *my * (rw2_to_rw3_signed(r) - rw2_to_rw3_signed(rw2_old));
  }
  // fold: the droplets still in the root finder move to lanes 0 .. total-1 of the workgroup
  const unsigned long long bal = __ballot(need);
  if (lane_id() == 0) wcnt[wave_id()] = uint32_t(__popcll(bal));
  const T nn_move = need ? xr[12][threadIdx.x] : T(0);         // (slot and threadIdx.x index the same array: read before anybody writes)
  __syncthreads();
  uint32_t first = 0, total = 0;
#pragma unroll
  for (unsigned w = 0; w < BS / WAVE; ++w) { const uint32_t n_w = wcnt[w]; if (w < wave_id()) first += n_w; total += n_w; }
  if (need) {
    const uint32_t slot = first + uint32_t(__popcll(bal & ((1ull << lane_id()) - 1ull)));
    xw[0][slot] = id; xw[1][slot] = c; xw[2][slot] = uint32_t(pos0); xw[3][slot] = k.count;
    xr[0][slot] = k.s.a; xr[1][slot] = k.s.b; xr[2][slot] = k.s.fa; xr[3][slot] = k.s.fb; xr[4][slot] = k.s.d; xr[5][slot] = k.s.fd;
    xr[6][slot] = k.e; xr[7][slot] = k.fe; xr[8][slot] = rw2_old; xr[9][slot] = rd3; xr[10][slot] = kpa; xr[11][slot] = vt;
    xr[12][slot] = nn_move;
  }
  __syncthreads();
  if (threadIdx.x >= total) return;
  const uint32_t t = threadIdx.x;
  id = xw[0][t]; c = xw[1][t];
  const size_t pos = xw[2][t];
  k.count = xw[3][t];
  k.s.a = xr[0][t]; k.s.b = xr[1][t]; k.s.fa = xr[2][t]; k.s.fb = xr[3][t]; k.s.d = xr[4][t]; k.s.fd = xr[5][t];
  k.e = xr[6][t]; k.fe = xr[7][t]; rw2_old = xr[8][t]; rd3 = xr[9][t]; kpa = xr[10][t]; vt = xr[11][t];
  const cond_cell_fast<T> cc = a.pre[c];
  cond_fun_fast<T, OPT> ff;
  ff.setup_cell(cc, rw2_old, a.dt_sub, rd3, kpa, vt);
  unsigned left = 1;
  const T r = advance_rw2_tail_with(ff, rd3, a.eps, k, &left);
  const bool deferred = df.budget != 0 && left == 0;
  if (!deferred) { a.rw2[id] = r; a.m3_after[pos] = xr[12][t] * (rw2_to_rw3_signed(r) - rw2_to_rw3_signed(rw2_old)); }
  const unsigned long long dbal = __ballot(deferred);
  if (dbal) {
    const int leader = __ffsll((long long)dbal) - 1;
    uint32_t base = 0;
    const unsigned shard = blockIdx.x % DEFER_SHARDS;
    if (int(lane_id()) == leader) base = atomicAdd(df.count + shard * DEFER_CNT_STRIDE, uint32_t(__popcll(dbal)));
    base = __shfl(base, leader);
    if (deferred) df.list[size_t(shard) * df.shard_cap + base + __popcll(dbal & ((1ull << lane_id()) - 1ull))] = uint32_t(pos);
  }
}

// Measured and dropped (round 3): the fold REPEATED between the bracket updates.  Of the droplets that enter the loop 70 % end after one
// more update, 19 % after two, 10 % after three, so the folded waves run their second and third update at a lane use of 30 and 10 %.  With
// the loop advanced one update at a time (a state machine with ONE evaluation site, the same operations as toms748_tail: bit-identical
// results) and the workgroup dealt again whenever that frees a wave (95 droplets -> 2 waves, 28 -> 1): 128 VGPRs + 28 B of scratch, cond on
// C3 7.6 ms against 6.3 -- the selects that pick the step's abscissa, the run-time trip count of the quadratic step and the spills cost
// more than the emptier waves did.
// G lanes per cell: 1 = the ordered walk (strict arithmetic: the reference's summation order); 8 = fast arithmetic, every lane
// sums each 8th value of the staged segment and a fixed 3-step shuffle tree joins them (deterministic, different rounding)
//
// Round 4: the fast arithmetic's sums of the droplets' CHANGES (delta mode) are taken in FIXED POINT.  The cell's addends are scaled by
// an exact power of two chosen from the largest of them and their number (2^k with count x max|x| < 2^61 / 2^k: both independent of the
// addends' order), rounded to integers once -- to 2^-50 of the largest addend or better -- and summed in 64-bit integer arithmetic:
// integer addition is associative, so the sum does not depend on the ORDER of the addends.  That is what lets the storage-order condensation kernel leave a droplet's
// change at the position where the re-sort's scatter puts the droplet (its ARRIVAL rank inside the cell -- the order in which the
// move's histogram atomics happened to be served, different from run to run) and the finish read one contiguous stretch instead of
// gathering 8 bytes per droplet through sorted_id: the same bits whichever way the addends come (the gathered form, the positional
// form, the crowded cells' wave form), run after run.
__device__ __forceinline__ int fx_shift(double amax, uint32_t cnt)
{ int e; (void)frexp(amax, &e); return 61 - e - (32 - __clz(int(cnt | 1u))); }      // (amax < 2^e, cnt < 2^(32 - clz): cnt x amax x 2^k < 2^61)
__device__ __forceinline__ long long to_fx(double x, int k) { return __double2ll_rn(ldexp(x, k)); }
// the larger magnitude; a NaN (or an infinity) among the addends stays and makes the cell's sum NaN as the floating-point sum would be
__device__ __forceinline__ double nanmax(double a, double b) { return (b > a || b != b) ? b : a; }
// one lane's share (every G-th value of the cell's segment): the largest magnitude, then the fixed-point sum
template <class T, int G>
__device__ __forceinline__ double seg_amax(const T *lds, const T *glob, const uint32_t *gather, bool staged, uint32_t base, uint32_t s, uint32_t e)
{
  double m = 0;
  if (staged) for (uint32_t q = s; q < e; q += G) m = nanmax(m, fabs(double(lds[q - base])));
  else if (gather) for (uint32_t q = s; q < e; q += G) m = nanmax(m, fabs(double(glob[gather[q]])));
  else for (uint32_t q = s; q < e; q += G) m = nanmax(m, fabs(double(glob[q])));
  return m;
}
template <class T, int G>
__device__ __forceinline__ long long seg_sum_fx(const T *lds, const T *glob, const uint32_t *gather, bool staged, uint32_t base, uint32_t s, uint32_t e, int k)
{
  long long acc = 0;
  if (staged) for (uint32_t q = s; q < e; q += G) acc += to_fx(double(lds[q - base]), k);
  else if (gather) for (uint32_t q = s; q < e; q += G) acc += to_fx(double(glob[gather[q]]), k);
  else for (uint32_t q = s; q < e; q += G) acc += to_fx(double(glob[q]), k);
  return acc;
}
template <class T, int G>
__global__ void __launch_bounds__(BS)
k_cond_cellfinish(size_t n_cell, int cfc, const uint32_t *cell_start, const T *m3_before, const T *m3_after,
                  const T *dv, const T *rhod, T *rv, T *th, const T *Tk, T *rw_mom3, int step, int sstp, int ndims, int delta = 0,
                  const uint32_t *gather = nullptr /* m3_after is indexed by gather[position] (the storage-order condensation kernel) */)
{
  // delta: m3_after holds the droplets' CHANGES n (rw_new^3 - rw_old^3) (k_cond_fast): their sum is the whole answer of every substep
  if (delta) { step = 0; sstp = 1; }
  __shared__ T lds[CF_CAP];
  __shared__ uint32_t cs[CF_CELLS + 1];
  const size_t c0 = size_t(blockIdx.x) * cfc;
  const size_t c1 = c0 + cfc < n_cell ? c0 + cfc : n_cell;
  const int nc = int(c1 - c0);
  if (int(threadIdx.x) <= nc) cs[threadIdx.x] = cell_start[c0 + threadIdx.x];
  __syncthreads();
  const int cl = int(threadIdx.x) / G, sub = int(threadIdx.x) % G;
  const size_t c = c0 + cl;
  const bool mine = cl < nc;
  const uint32_t s = mine ? cs[cl] : 0u, e = mine ? cs[cl + 1] : 0u;
  const bool has = e > s;
  T after = 0, before = 0;
  long long after_fx = 0;
  int fxk = 0;
  bool bad_fx = false;
  // the workgroup's cells are taken in runs that fit the LDS stage (normally one run; crowded neighbourhoods split instead of
  // falling back to uncoalesced global reads); a single cell above CF_CAP is summed from global memory
  for (int cb = 0; cb < nc;) {
    int ce = cb + 1;
    while (ce < nc && cs[ce + 1] - cs[cb] <= uint32_t(CF_CAP)) ++ce;
    const uint32_t base = cs[cb], end = cs[ce];
    const bool staged = (end - base) <= uint32_t(CF_CAP);
    const bool in_run = mine && has && cl >= cb && cl < ce;
    if (staged && gather) {
      // two dependent loads per value: all of a thread's ids first, then all of its gathers (CF_CAP / BS of each in flight)
      constexpr int PER = CF_CAP / BS;
      uint32_t gi[PER]; T gv[PER];
#pragma unroll
      for (int k = 0; k < PER; ++k) { const uint32_t q = base + threadIdx.x + uint32_t(k) * BS; gi[k] = q < end ? gather[q] : 0u; }
#pragma unroll
      for (int k = 0; k < PER; ++k) { const uint32_t q = base + threadIdx.x + uint32_t(k) * BS; gv[k] = q < end ? m3_after[gi[k]] : T(0); }
#pragma unroll
      for (int k = 0; k < PER; ++k) { const uint32_t q = base + threadIdx.x + uint32_t(k) * BS; if (q < end) lds[q - base] = gv[k]; }
    }
    else if (staged) for (uint32_t q = base + threadIdx.x; q < end; q += BS) lds[q - base] = m3_after[q];
    __syncthreads();
    if (delta) {
      // (every lane of the workgroup takes part in the shuffles; lanes outside the run carry zeros)
      double amax = in_run ? seg_amax<T, G>(lds, m3_after, gather, staged, base, s + sub, e) : 0.;
#pragma unroll
      for (int d = G / 2; d > 0; d >>= 1) amax = nanmax(amax, __shfl_xor(amax, d));
      if (in_run) {
        bad_fx = !(amax < 1e300);                    // (NaN or infinite addends: the sum is NaN)
        fxk = fx_shift(amax, e - s);
        after_fx = (amax > 0 && !bad_fx) ? seg_sum_fx<T, G>(lds, m3_after, gather, staged, base, s + sub, e, fxk) : 0ll;
      }
    }
    else if (in_run) after = (gather && !staged) ? seg_sum_gather<T, G>(m3_after, gather, s + sub, e) : seg_sum<T, G>(lds, m3_after, staged, base, s + sub, e);
    if (step == 0 && !delta) {
      __syncthreads();
      if (staged) for (uint32_t q = base + threadIdx.x; q < end; q += BS) lds[q - base] = m3_before[q];
      __syncthreads();
      if (in_run) before = seg_sum<T, G>(lds, m3_before, staged, base, s + sub, e);
    }
    __syncthreads();
    cb = ce;
  }
  if (G > 1) {
#pragma unroll
    for (int d = G / 2; d > 0; d >>= 1) { after = after + __shfl_xor(after, d); before = before + __shfl_xor(before, d); after_fx += __shfl_xor(after_fx, d); }
  }
  if (!mine || sub != 0) return;
  if (delta) after = bad_fx ? T(NAN) : T(ldexp(double(after_fx), -fxk));
  cellfinish_apply(c, has, after, before, dv, rhod, rv, th, Tk, rw_mom3, step, sstp, ndims);
}
// The fast arithmetic's finish when the changes lie in the sorted order already (the condensation kernel that carried the re-sort's
// scatter put them there): no LDS stage, no barrier.  Eight lanes per cell as above; a lane reads every 8th value of the cell's stretch
// straight from global memory (the eight lanes of a cell cover 64 consecutive bytes per load), keeps the first 64 values of the cell in
// registers between the pass that finds the largest magnitude and the fixed-point sum, and re-reads only what a cell holds beyond that.
// The sums are integer sums: the same bits as k_cond_cellfinish / _wave give.
template <class T>
__global__ void __launch_bounds__(BS)
k_cond_cellfinish_direct(size_t n_cell, const uint32_t *cell_start, const T *change, const T *dv, const T *rhod, T *rv, T *th, const T *Tk,
                         T *rw_mom3, int ndims)
{
  constexpr int G = 8, PER = 8;
  const size_t c = (size_t(blockIdx.x) * BS + threadIdx.x) / G;
  const uint32_t sub = threadIdx.x % G;
  const bool mine = c < n_cell;
  const uint32_t s = mine ? cell_start[c] : 0u, e = mine ? cell_start[c + 1] : 0u;
  double v[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) { const uint32_t q = s + sub + uint32_t(k) * G; v[k] = q < e ? double(change[q]) : 0.; }
  double amax = 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) amax = nanmax(amax, fabs(v[k]));
  for (uint32_t q = s + sub + uint32_t(PER) * G; q < e; q += G) amax = nanmax(amax, fabs(double(change[q])));
#pragma unroll
  for (int d = G / 2; d > 0; d >>= 1) amax = nanmax(amax, __shfl_xor(amax, d));
  const bool bad = !(amax < 1e300);                  // (NaN or infinite addends: the sum is NaN)
  const int fxk = fx_shift(amax, e - s);
  long long acc = 0;
  if (amax > 0 && !bad) {
#pragma unroll
    for (int k = 0; k < PER; ++k) acc += to_fx(v[k], fxk);
    for (uint32_t q = s + sub + uint32_t(PER) * G; q < e; q += G) acc += to_fx(double(change[q]), fxk);
  }
#pragma unroll
  for (int d = G / 2; d > 0; d >>= 1) acc += __shfl_xor(acc, d);
  if (!mine || sub != 0) return;
  const T after = bad ? T(NAN) : T(ldexp(double(acc), -fxk));
  cellfinish_apply(c, e > s, after, T(0), dv, rhod, rv, th, Tk, rw_mom3, 0, 1, ndims);
}
// Round 6: ALL the condensation substeps of a step in ONE launch (fast arithmetic, sstp_cond > 1).  Condensation couples droplets
// through their cell alone -- the cell's th and rv take the droplets' change after every substep -- so a workgroup that owns a run of
// cells and every droplet in them can make the substeps one after the other with workgroup barriers only: per substep the cell pass
// (one lane per cell: the Eulerian fields' substep, hskpng_Tpr, the growth rate's set-up -- cell_cond_pre_one, kept in LDS), the droplets
// (a lane per droplet of the cells' stretch of the sorted order: the same solver calls as k_cond_lean / k_cond_lean_fold), the cells'
// fixed-point sums of the changes (LDS atomics: maximum, then integers -- k_cond_cellfinish_direct's numbers, whose sums do not depend
// on the addends' order) and cellfinish_apply.  The same operations on the same numbers as the launches it replaces: bit-identical
// (tests/test_hip_parity.py).  BASELINE configs[1] (76 x 76 cells x 64, ten substeps) made thirty launches of 5-16 us each for this:
// 0.27 of its 0.62 ms step.  A lane walks its droplets (position S + lane, + 256 ...) in every substep and a cell lane its cell, so
// everything a substep reads of what the previous one wrote was written by the same lane.
// SOLVER 0: the lean solver, droplets whose bracket may hold several roots through TOMS748 in place (what k_cond_lean_listed computes
// for them); 2: TOMS748 for every droplet.
constexpr int SUBSTEP_CELLS = 64;              // cells per workgroup at most (the host picks 256 / (droplets per cell), at least one)
template <class T, bool UNI, int SOLVER>
__global__ void __launch_bounds__(BS) k_cond_substeps(cell_pre_args<T> P, cond_args<T> a, T kpa_uniform, const uint32_t *cell_start, unsigned cells_per_wg,
                                                      int sstp, sstp_fields<T> ss, T *rw_mom3, int ask_several)
{
  __shared__ cond_cell_fast<T> cc_s[SUBSTEP_CELLS];
  __shared__ unsigned long long amax_s[SUBSTEP_CELLS];
  __shared__ long long fx_s[SUBSTEP_CELLS];
  __shared__ int fxk_s[SUBSTEP_CELLS];
  const size_t c0 = size_t(blockIdx.x) * cells_per_wg;
  if (c0 >= P.n_cell) return;
  const size_t c1 = c0 + cells_per_wg < P.n_cell ? c0 + cells_per_wg : P.n_cell;
  const unsigned ncl = unsigned(c1 - c0), t = threadIdx.x;
  const uint32_t S = cell_start[c0], E = cell_start[c1];
  // a lane's FIRST droplet (its only one unless the cells hold more than 256 between them) stays in registers through the substeps: its
  // attributes are read once, its wet radius is written once, its change never leaves the lane
  const bool has0 = S + t < E;
  uint32_t id0 = 0, lc0 = 0;
  T rw2_0 = 0, rd3_0 = 0, vt_0 = 0, kpa_0 = kpa_uniform, nn_0 = 0, delta0 = 0;
  if (has0) {
    id0 = a.sorted_id[S + t]; lc0 = a.sorted_ijk[S + t] - uint32_t(c0);
    rw2_0 = a.rw2[id0]; rd3_0 = a.rd3[id0]; vt_0 = a.vt[id0]; nn_0 = T(a.n[id0]);
    if (!UNI) kpa_0 = a.kpa[id0];
  }
  for (int step = 0; step < sstp; ++step) {
    if (t < ncl) {
      ss.step = step;
      cc_s[t] = cell_cond_pre_one(P, c0 + t, int(step == 0), ss);
      amax_s[t] = 0ull; fx_s[t] = 0ll;
    }
    __syncthreads();
    for (uint32_t j = 0, q = S + t; q < E; ++j, q += BS) {
      uint32_t id = id0, lc = lc0;
      T rw2_old = rw2_0, rd3 = rd3_0, vt = vt_0, kpa = kpa_0, nn = nn_0;
      if (j) {
        id = a.sorted_id[q]; lc = a.sorted_ijk[q] - uint32_t(c0);
        rw2_old = a.rw2[id]; rd3 = a.rd3[id]; vt = a.vt[id]; nn = T(a.n[id]);
        if (!UNI) kpa = a.kpa[id];
      }
      T delta = 0;
      if (!(rw2_old <= 0)) {
        cond_fun_fast<T, 15> ff;
        ff.setup_cell(cc_s[lc], rw2_old, a.dt_sub, rd3, kpa, vt);
        T r;
        if constexpr (SOLVER == 2) r = advance_rw2_with(ff, rw2_old, rd3, a.dt_sub, a.eps, a.cond_mlt, a.n_iter);
        else {
          bool several = false;
          r = advance_rw2_lean2_with(ff, rw2_old, rd3, a.dt_sub, a.eps, a.cond_mlt, a.n_iter, &several, ask_several != 0);
          if (several) r = advance_rw2_with(ff, rw2_old, rd3, a.dt_sub, a.eps, a.cond_mlt, a.n_iter);
        }
        if (j) a.rw2[id] = r; else rw2_0 = r;
        delta = nn * (r * (r * rsqrt_pos(r)) - rw2_old * (rw2_old * rsqrt_pos(rw2_old)));
      }
      if (j) a.m3_after[q] = delta; else delta0 = delta;
      // (the bits of a non-negative double order like the double; a NaN's lie above every number's: it stays, as nanmax keeps it)
      atomicMax(&amax_s[lc], (unsigned long long)__double_as_longlong(fabs(double(delta))));
    }
    __syncthreads();
    if (t < ncl) {
      const double amax = __longlong_as_double((long long)amax_s[t]);
      const uint32_t cnt = cell_start[c0 + t + 1] - cell_start[c0 + t];
      fxk_s[t] = (amax > 0 && amax < 1e300) ? fx_shift(amax, cnt) : 0;
    }
    __syncthreads();
    for (uint32_t j = 0, q = S + t; q < E; ++j, q += BS) {
      const uint32_t lc = j ? a.sorted_ijk[q] - uint32_t(c0) : lc0;
      const T delta = j ? a.m3_after[q] : delta0;
      const double amax = __longlong_as_double((long long)amax_s[lc]);
      if (amax > 0 && amax < 1e300) atomicAdd((unsigned long long *)&fx_s[lc], (unsigned long long)to_fx(double(delta), fxk_s[lc]));
    }
    __syncthreads();
    if (t < ncl) {
      const size_t c = c0 + t;
      const double amax = __longlong_as_double((long long)amax_s[t]);
      const bool bad = !(amax < 1e300);
      const T after = bad ? T(NAN) : T(ldexp(double(fx_s[t]), -fxk_s[t]));
      cellfinish_apply(c, cell_start[c + 1] > cell_start[c], after, T(0), P.dv, P.rhod, const_cast<T *>(P.rv), const_cast<T *>(P.th), P.Tk, rw_mom3, 0, 1, P.ndims);
    }
    __syncthreads();
  }
  if (has0 && !(rw2_0 <= 0)) a.rw2[id0] = rw2_0;
}
// Fast arithmetic, crowded cells (hundreds of SDs per cell): ONE WAVE per cell sums the segment with coalesced loads and a
// fixed shuffle tree -- deterministic, but not the reference's serial order, so the sums differ from the ordered ones in the last
// bits like everything else in this mode.  (With 64 SDs per cell the staged kernel above is faster: 0.63 against 1.16 ms;
// with 512 per cell a single lane walking 512 staged values is not: 15.7 ms against 4.)
template <class T>
__global__ void __launch_bounds__(BS)
k_cond_cellfinish_wave(size_t n_cell, const uint32_t *cell_start, const T *m3_before, const T *m3_after,
                       const T *dv, const T *rhod, T *rv, T *th, const T *Tk, T *rw_mom3, int step, int sstp, int ndims, int delta = 0,
                       const uint32_t *gather = nullptr)
{
  const size_t c = size_t(blockIdx.x) * (BS / WAVE) + wave_id();
  if (c >= n_cell) return;
  if (delta) { step = 0; sstp = 1; }
  const uint32_t s = cell_start[c], e = cell_start[c + 1];
  T after = 0, before = 0;
  if (delta) {                          // (fixed point: see k_cond_cellfinish)
    // (the first 512 values of the cell wait in registers between the two passes, as in k_cond_cellfinish_direct)
    constexpr int PER = 8;
    double v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) { const uint32_t q = s + lane_id() + uint32_t(k) * WAVE; v[k] = q < e ? double(m3_after[gather ? gather[q] : q]) : 0.; }
    double amax = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) amax = nanmax(amax, fabs(v[k]));
    for (uint32_t q = s + lane_id() + uint32_t(PER) * WAVE; q < e; q += WAVE) amax = nanmax(amax, fabs(double(m3_after[gather ? gather[q] : q])));
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) amax = nanmax(amax, __shfl_xor(amax, d));
    const bool bad = !(amax < 1e300);
    const int fxk = fx_shift(amax, e - s);
    long long acc = 0;
    if (amax > 0 && !bad) {
#pragma unroll
      for (int k = 0; k < PER; ++k) acc += to_fx(v[k], fxk);
      for (uint32_t q = s + lane_id() + uint32_t(PER) * WAVE; q < e; q += WAVE) acc += to_fx(double(m3_after[gather ? gather[q] : q]), fxk);
    }
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) acc += __shfl_down(acc, d);
    after = bad ? T(NAN) : T(ldexp(double(acc), -fxk));
  } else {
    for (uint32_t q = s + lane_id(); q < e; q += WAVE) { after = after + m3_after[gather ? gather[q] : q]; if (step == 0) before = before + m3_before[q]; }
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) { after = after + __shfl_down(after, d); before = before + __shfl_down(before, d); }
  }
  if (lane_id() == 0) cellfinish_apply(c, e > s, after, before, dv, rhod, rv, th, Tk, rw_mom3, step, sstp, ndims);
}

// ============================================================================================
// per-particle condensation substepping (exact_sstp_cond): src/impl/condensation/perparticle/*.ipp,
// particles_step.ipp:199-236.  The reference runs ~8 thrust passes per substep over n_part-long temporaries
// (sstp_dlt_*, rwX, drwX, Tp); here a super-droplet keeps its private (rv, th, rhod, p) in registers:
//   no mixing            : ONE launch does all substeps of a super-droplet (k_pp_cond_nomix)
//   adaptive, no mixing  : ONE launch, the substep count is chosen per super-droplet (k_pp_cond_adaptive)
//   mixing               : one launch per substep + the ordered per-cell sums that couple the droplets of a cell
// ============================================================================================
template <class T>
struct pp_args {
  const uint32_t *sorted_id, *sorted_ijk;
  const n_t *n; const T *rd3, *kpa, *vt; T *rw2;
  T *pp_rv, *pp_th, *pp_rh, *pp_p;            // per-particle "old" state (sstp_tmp_* of the reference)
  const T *rv, *th, *rhod, *p;                // cell state after sync_in
  const T *dv, *lambda_D, *lambda_K, *rc2;
  T *ssp; const T *dot_ssp;                   // opts.turb_cond: SGS supersaturation perturbation and its tendency, else nullptr
  T *m3_before, *m3_after;                    // position-ordered: n rw^3 before / after (no mixing)
  T *dlt_rv, *dlt_th, *dlt_rh, *dlt_p, *rw3s; // mixing only: per-particle deltas and stored rw^3
  T *drv, *dth;                               // mixing only, position-ordered: this substep's change of rv, th
  const T *dst_rv, *dst_th;                   // mixing only: per-cell sums of the previous substep
  T dt, RH_max, eps, cond_mlt, adapt_eps, adapt_max; unsigned n_iter;
  int sstp_cond, sstp_cond_act, th_dry, const_p, RH_formula, n_dims, step;
};
template <class T> __device__ __forceinline__ T rw2torw3(T rw2) { return rw2 * T(sqrt(rw2)); }       // cond_common.ipp:57-67
template <class T> __device__ __forceinline__ T rw3diff2drv(T d, T rhod, n_t n, T dv, int n_dims)
{                                                                                                  // cond_common.ipp:24-41
  const T mlt = -cst<T>::rho_w * T(4. / 3) * cst<T>::pi;
  return n_dims > 0 ? mlt * d * T(n) / rhod / dv : mlt * d * T(n);
}
// temperature, pressure and RH of one super-droplet's private air (cond_perparticle_advance_rw2.ipp:30-125)
// ssp: SGS supersaturation perturbation, RH_sgs = RH + ssp (cond_perparticle_advance_rw2.ipp:8-22); 0 without turb_cond
template <class T> __device__ __forceinline__ void pp_state(const pp_args<T> &a, T t_th, T t_rv, T t_rh, T &t_p, T &Tp, T &RH, T ssp)
{
  Tp = a.th_dry ? theta_dry_T(t_th, t_rh) : T(t_th * exner(t_p));
  if (!a.const_p) t_p = theta_dry_p(t_rh, t_rv, Tp);
  RH = RH_of(a.RH_formula, t_p, t_rv, Tp);
  if (a.ssp) RH = RH + ssp;
}
template <class T, bool FAST> __device__ __forceinline__ T pp_advance(const pp_args<T> &a, T rw2, T dt, T t_rh, T t_rv, T Tp, T RH,
                                                                      T rd3, T kpa, T vt, T lD, T lK)
{ return advance_rw2<T, FAST>(rw2, dt, t_rh, t_rv, Tp, visc(Tp), rd3, kpa, vt, lD, lK, RH, a.RH_max, a.eps, a.cond_mlt, a.n_iter); }

// sstp_save.ipp:17-22 / init_perparticle_sstp.ipp: per-particle copy of the cell state
template <class T>
__global__ void k_pp_save(size_t n, const uint32_t *ijk, const T *rv, const T *th, const T *rhod, const T *p,
                          T *pp_rv, T *pp_th, T *pp_rh, T *pp_p)
{
  const size_t i = gid(); if (i >= n) return;
  const uint32_t c = ijk[i];
  if (c == DEAD_CELL) return;
  pp_rv[i] = rv[c]; pp_th[i] = th[c]; pp_rh[i] = rhod[c];
  if (pp_p) pp_p[i] = p[c];
}
// update_incloud_time.ipp:36-66: the time a super-droplet has been activated grows by dt while rw2 > rc2(T of its cell), else 0
template <class T>
__global__ void k_incloud_time(size_t n, const uint32_t *ijk, const T *rd3, const T *kpa, const T *rw2, const T *Tk, T dt, T *ict)
{
  const size_t i = gid(); if (i >= n) return;
  const uint32_t c = ijk[i];
  if (c == DEAD_CELL) return;
  const T rc2 = rc2_of(rd3[i], kpa[i], Tk[c]);
  ict[i] = rw2[i] > rc2 ? ict[i] + dt : T(0);
}
// hskpng_rc2.ipp:14-32
template <class T>
__global__ void k_rc2(size_t n, const T *rd3, const T *kpa, T Tk, T *rc2)
{
  const size_t i = gid(); if (i >= n) return;
  if (rc2[i] == T(-1)) rc2[i] = rc2_of(rd3[i], kpa[i], Tk);
}

template <class T, bool FAST>
__global__ void __launch_bounds__(BS) k_pp_cond_nomix(size_t n_part, pp_args<T> a)
{
  const size_t pos = gid(); if (pos >= n_part) return;
  const uint32_t id = a.sorted_id[pos], c = a.sorted_ijk[pos];
  const n_t n = a.n[id];
  const T rd3 = a.rd3[id], kpa = a.kpa[id], vt = a.vt[id], dv = a.dv[c], lD = a.lambda_D[c], lK = a.lambda_K[c];
  T t_rv = a.pp_rv[id], t_th = a.pp_th[id], t_rh = a.pp_rh[id], t_p = a.const_p ? a.pp_p[id] : T(0);
  const T d_rv = a.rv[c] - t_rv, d_th = a.th[c] - t_th, d_rh = a.rhod[c] - t_rh, d_p = a.const_p ? a.p[c] - t_p : T(0);
  T rw2 = a.rw2[id], rw3 = 0, Tp, RH;
  T ssp = a.ssp ? a.ssp[id] : T(0);
  const T dot_ssp = a.ssp ? a.dot_ssp[id] : T(0);
  a.m3_before[pos] = rw2 >= 0 ? T(n) * rw2torw3(rw2) : T(n) * rw2;
  for (int step = 0; step < a.sstp_cond; ++step) {
    t_rv = t_rv + d_rv / a.sstp_cond; t_th = t_th + d_th / a.sstp_cond; t_rh = t_rh + d_rh / a.sstp_cond;      // apply_noncond_...ipp
    if (a.const_p) t_p = t_p + d_p / a.sstp_cond;
    if (a.ssp) ssp = ssp + a.dt / a.sstp_cond * dot_ssp;                                                      // apply_perparticle_sgs_supersat.ipp
    T drw3 = step > 0 ? -rw3 : -rw2torw3(rw2);
    pp_state(a, t_th, t_rv, t_rh, t_p, Tp, RH, ssp);
    rw2 = pp_advance<T, FAST>(a, rw2, a.dt / a.sstp_cond, t_rh, t_rv, Tp, RH, rd3, kpa, vt, lD, lK);
    rw3 = rw2torw3(rw2);
    drw3 = rw3 + drw3;
    drw3 = rw3diff2drv(drw3, t_rh, n, dv, a.n_dims);
    t_rv = drw3 + t_rv;
    drw3 = drw3 * d_th_d_rv(Tp, t_th);
    t_th = drw3 + t_th;
  }
  a.rw2[id] = rw2;
  if (a.ssp) a.ssp[id] = ssp;
  a.m3_after[pos] = rw2 >= 0 ? T(n) * rw2torw3(rw2) : T(n) * rw2;
}

// perparticle_nomixing_adaptive_sstp_cond.ipp:56-265
template <class T, bool FAST>
__global__ void __launch_bounds__(BS) k_pp_cond_adaptive(size_t n_part, pp_args<T> a)
{
  const size_t pos = gid(); if (pos >= n_part) return;
  const uint32_t id = a.sorted_id[pos], c = a.sorted_ijk[pos];
  const n_t n = a.n[id];
  const T rd3 = a.rd3[id], kpa = a.kpa[id], vt = a.vt[id], dv = a.dv[c], lD = a.lambda_D[c], lK = a.lambda_K[c];
  T t_rv = a.pp_rv[id], t_th = a.pp_th[id], t_rh = a.pp_rh[id], t_p = a.const_p ? a.pp_p[id] : T(0);
  const T d_rv = a.rv[c] - t_rv, d_th = a.th[c] - t_th, d_rh = a.rhod[c] - t_rh, d_p = a.const_p ? a.p[c] - t_p : T(0);
  T rw2 = a.rw2[id], drw2 = 0, Tp = 0, RH = 0, frac = 0;
  T ssp = a.ssp ? a.ssp[id] : T(0);
  const T dot_ssp = a.ssp ? a.dot_ssp[id] : T(0);
  a.m3_before[pos] = rw2 >= 0 ? T(n) * rw2torw3(rw2) : T(n) * rw2;
  auto apply_delta = [&](T m) { t_rv += d_rv * m; t_th += d_th * m; t_rh += d_rh * m; if (a.const_p) t_p += d_p * m;
                                if (a.ssp) ssp += dot_ssp * a.dt * m; };
  const int sstp_max = a.sstp_cond;
  unsigned sstp = unsigned(sstp_max);
  bool first_done = sstp_max == 1;
  {
    T drw2_new = 0;
    for (int tr = 1; tr <= sstp_max; tr *= 2) {
      frac = tr == 1 ? T(1) : -T(1) / tr;
      apply_delta(frac);
      pp_state(a, t_th, t_rv, t_rh, t_p, Tp, RH, ssp);
      T d = rw2;                                                    // advance_rw2<real_t, false>: the increment (rw2 <= 0: rw2 itself)
      if (rw2 > 0) d = pp_advance<T, FAST>(a, rw2, a.dt / tr, t_rh, t_rv, Tp, RH, rd3, kpa, vt, lD, lK) - rw2;
      if (tr == 1) drw2 = d; else drw2_new = d;
      if (tr > 1) {
        if (fabs(drw2_new * 2 - drw2) <= a.adapt_eps * rw2 && fabs(drw2) < a.adapt_max * rw2) {
          sstp = unsigned(tr / 2);
          apply_delta(-frac);
          first_done = true;
          break;
        }
        drw2 = drw2_new;
      }
    }
    if (a.sstp_cond_act > 1) {
      const T rc2 = a.rc2[id];
      if ((rw2 < rc2 && (rw2 + sstp * drw2) > rc2) || (rw2 > rc2 && (rw2 + sstp * drw2) < rc2)) {
        sstp = unsigned(a.sstp_cond_act);
        first_done = false;
      }
    }
    if (!first_done) apply_delta(sstp_max == 1 ? -frac : frac);
  }
  frac = T(1) / sstp;
  T rw3 = drw2;                                                     // `real_t &rw3 = drw2` in the reference
  for (unsigned step = 0; step < sstp; ++step) {
    T drw3 = step > 0 ? -rw3 : -rw2torw3(rw2);
    if (first_done && step == 0) rw2 += rw3;
    else {
      apply_delta(frac);
      pp_state(a, t_th, t_rv, t_rh, t_p, Tp, RH, ssp);
      rw2 = pp_advance<T, FAST>(a, rw2, a.dt / sstp, t_rh, t_rv, Tp, RH, rd3, kpa, vt, lD, lK);
    }
    if (step < sstp - 1) { rw3 = rw2torw3(rw2); drw3 += rw3; }
    else drw3 += rw2torw3(rw2);
    drw3 = rw3diff2drv(drw3, t_rh, n, dv, a.n_dims);
    t_rv += drw3;
    drw3 = drw3 * d_th_d_rv(Tp, t_th);
    t_th += drw3;
  }
  a.rw2[id] = rw2;
  if (a.ssp) a.ssp[id] = ssp;
  a.m3_after[pos] = rw2 >= 0 ? T(n) * rw2torw3(rw2) : T(n) * rw2;
}

// one substep with mixing (particles_step.ipp:221-232); the per-cell sums of drv / dth are taken by k_cell_seqsum and
// added to every droplet of the cell at the start of the next substep (update_pstate, update_th_rv.ipp:243-283)
template <class T, bool FAST>
__global__ void __launch_bounds__(BS) k_pp_cond_mix(size_t n_part, pp_args<T> a)
{
  const size_t pos = gid(); if (pos >= n_part) return;
  const uint32_t id = a.sorted_id[pos], c = a.sorted_ijk[pos];
  const n_t n = a.n[id];
  T t_rv = a.pp_rv[id], t_th = a.pp_th[id], t_rh = a.pp_rh[id], t_p = a.const_p ? a.pp_p[id] : T(0);
  T d_rv, d_th, d_rh, d_p = 0;
  if (a.step == 0) {
    d_rv = a.rv[c] - t_rv; d_th = a.th[c] - t_th; d_rh = a.rhod[c] - t_rh;
    a.dlt_rv[id] = d_rv; a.dlt_th[id] = d_th; a.dlt_rh[id] = d_rh;
    if (a.const_p) { d_p = a.p[c] - t_p; a.dlt_p[id] = d_p; }
  } else {
    t_rv = t_rv + a.dst_rv[c]; t_th = t_th + a.dst_th[c];
    d_rv = a.dlt_rv[id]; d_th = a.dlt_th[id]; d_rh = a.dlt_rh[id];
    if (a.const_p) d_p = a.dlt_p[id];
  }
  t_rv = t_rv + d_rv / a.sstp_cond; t_th = t_th + d_th / a.sstp_cond; t_rh = t_rh + d_rh / a.sstp_cond;
  if (a.const_p) t_p = t_p + d_p / a.sstp_cond;
  T rw2 = a.rw2[id], Tp, RH;
  T ssp = T(0);
  if (a.ssp) { ssp = a.ssp[id] + a.dt / a.sstp_cond * a.dot_ssp[id]; a.ssp[id] = ssp; }                     // apply_perparticle_sgs_supersat.ipp
  T drw3 = a.step > 0 ? -a.rw3s[id] : -rw2torw3(rw2);
  pp_state(a, t_th, t_rv, t_rh, t_p, Tp, RH, ssp);
  rw2 = pp_advance<T, FAST>(a, rw2, a.dt / a.sstp_cond, t_rh, t_rv, Tp, RH, a.rd3[id], a.kpa[id], a.vt[id], a.lambda_D[c], a.lambda_K[c]);
  const T rw3 = rw2torw3(rw2);
  if (a.step < a.sstp_cond - 1) a.rw3s[id] = rw3;
  drw3 = rw3 + drw3;
  drw3 = rw3diff2drv(drw3, t_rh, n, a.dv[c], a.n_dims);
  a.rw2[id] = rw2;
  a.pp_rv[id] = t_rv; a.pp_th[id] = t_th; a.pp_rh[id] = t_rh;
  if (a.const_p) a.pp_p[id] = t_p;
  a.drv[pos] = drw3;
  a.dth[pos] = drw3 * d_th_d_rv(Tp, t_th);
}
// update_state (update_th_rv.ipp:287-299) after the last substep: the cell takes the value of its last droplet
template <class T>
__global__ void k_pp_mix_finish(size_t n_cell, const uint32_t *cell_start, const uint32_t *sorted_id, const T *pp_rv, const T *pp_th,
                                const T *dst_rv, const T *dst_th, T *rv, T *th)
{
  const size_t c = gid(); if (c >= n_cell) return;
  const uint32_t s = cell_start[c], e = cell_start[c + 1];
  if (e == s) return;
  const uint32_t id = sorted_id[e - 1];
  rv[c] = pp_rv[id] + dst_rv[c];
  th[c] = pp_th[id] + dst_th[c];
}

// ============================================================================================
// SGS turbulence (hskpng_tke.ipp, hskpng_turb_vel.ipp, hskpng_turb_ss.ipp, apply_perparticle_sgs_supersat.ipp;
// formulas common/GA17_turbulence.hpp:52-113)
// ============================================================================================
template <class T> struct normal_src { const T *arr; uint64_t call, seed; };
// diss_rate := TKE = ((L eps) / C_E)^(2/3);  tau = L / (2 pi)^(1/3) sqrt(C_tau / TKE);  L = SGS_mix_len[k]
template <class T>
__global__ void k_tke_tau(size_t n_cell, int nz, const T *mix_len, T *diss_rate, T *tau)
{
  const size_t c = gid(); if (c >= n_cell) return;
  const T L = mix_len[c % size_t(nz)];
  const T ret = cbrt((L * diss_rate[c]) / T(0.845));
  const T tke = ret * ret;
  diss_rate[c] = tke;
  tau[c] = L / T(pow(T(2) * cst<T>::pi, T(1. / 3.))) * sqrt(T(1.5) / tke);
}
// update_turb_vel: Ornstein-Uhlenbeck step of one velocity component
template <class T>
__global__ void k_turb_vel(size_t n, const uint32_t *ijk, const T *tau, const T *tke, T dt, normal_src<T> rs, T *vel)
{
  const size_t i = gid(); if (i >= n) return;
  const uint32_t c = ijk[i];
  if (c == DEAD_CELL) return;
  const T r = rs.arr ? rs.arr[i] : philox::normal<T>(i, rs.call, rs.seed);
  const T e = exp(-dt / tau[c]);
  vel[i] = vel[i] * e + sqrt((T(1) - e * e) * T(2. / 3.) * tke[c]) * r;
}
// tau_relax from the first wet moment per volume (count_mom holds sum n r_w of the cell), then dot_turb_ss per SD
template <class T>
__global__ void k_tau_rlx(size_t n_cell, const uint32_t *cell_start, const T *mom1, const T *dv, T *tau_rlx)
{
  const size_t c = gid(); if (c >= n_cell) return;
  if (cell_start[c + 1] > cell_start[c]) tau_rlx[c] = T(1) / (T(2.8e-4) * (mom1[c] / dv[c]));
}
template <class T>
__global__ void k_turb_dot_ss(size_t n, const uint32_t *ijk, const T *tau_rlx, const T *ssp, const T *wp, T *dot_ssp)
{
  const size_t i = gid(); if (i >= n) return;
  const uint32_t c = ijk[i];
  if (c == DEAD_CELL) return;
  dot_ssp[i] = T(3e-4) * wp[i] - ssp[i] / tau_rlx[c];
}
template <class T>
__global__ void k_sgs_supersat(size_t n, T dt_sub, const T *dot_ssp, T *ssp)
{ const size_t i = gid(); if (i < n) ssp[i] = ssp[i] + dt_sub * dot_ssp[i]; }

// ============================================================================================
// coalescence (particles_impl_coal.ipp:99-546, src/detail/kernels.hpp:38-202, kernel_interpolation.hpp:9-65)
// ============================================================================================
template <class T> struct coal_kernel_cfg {
  int kernel; int n_user_params; T r_max; const T *params;
  const T *eta, *rhod, *diss;     // per-cell fields of the turbulent (Onishi) kernel; diss == nullptr: dissipation rate 0 (opts.turb_coal off)
};

__device__ __forceinline__ int kernel_index(n_t R) { return R <= 100. ? int(R) : int(100 + (R - 100.) / 10.); }
__device__ __forceinline__ size_t kernel_vector_index(int i, int j, n_t nup)
{
  return i >= j ? size_t(0.5 * i * (i + 1) + j + nup) : size_t(0.5 * j * (j + 1) + i + nup);
}
template <class T>
__device__ __forceinline__ T interpolated_efficiency(const coal_kernel_cfg<T> &k, T r1, T r2)
{
  r1 *= 1e6; r2 *= 1e6;
  if (r1 >= k.r_max) r1 = k.r_max - 1e-6;
  if (r2 >= k.r_max) r2 = k.r_max - 1e-6;
  n_t dx, dy, x[4];
  if (r1 >= 100.) { x[0] = n_t(floor(r1 / 10.) * 10); dx = 10; } else { x[0] = n_t(floor(r1)); dx = 1; }
  if (r2 >= 100.) { x[2] = n_t(floor(r2 / 10.) * 10); dy = 10; } else { x[2] = n_t(floor(r2)); dy = 1; }
  x[1] = x[0] + dx; x[3] = x[2] + dy;
  const n_t nup = k.n_user_params;
  const size_t iv0 = kernel_vector_index(kernel_index(x[0]), kernel_index(x[2]), nup),
               iv1 = kernel_vector_index(kernel_index(x[1]), kernel_index(x[2]), nup),
               iv2 = kernel_vector_index(kernel_index(x[0]), kernel_index(x[3]), nup),
               iv3 = kernel_vector_index(kernel_index(x[1]), kernel_index(x[3]), nup);
  const T w0 = r1 - x[0], w1 = x[1] - r1, w2 = r2 - x[2], w3 = x[3] - r2;
  return (k.params[iv0] * w1 * w3 + k.params[iv1] * w0 * w3 + k.params[iv2] * w1 * w2 + k.params[iv3] * w0 * w2) / dx / dy;
}
template <class T>
__device__ __forceinline__ T k_geometric(n_t na, n_t nb, T rw2a, T rw2b, T vta, T vtb)
{
  const n_t nmax = na < nb ? nb : na;
  return cst<T>::pi * nmax * fabs(vta - vtb) * (rw2a + rw2b + 2. * sqrt(rw2a * rw2b));
}
// Onishi turbulent kernel without gravitational settling, 2 pi R^2 <|Wr|> g(R) (src/detail/kernel_onishi_nograv.hpp:29-153).
// The reference evaluates the Kolmogorov length as pow(nu^3/eps, real_t(1/4)) with an integer 1/4 == 0, i.e. leta == 1:
// kept, so that the kernel values are the reference's.
template <class T>
__device__ T kernel_onishi_nograv(T r1, T r2, T Re_l, T eps, T dnu, T ratio_den)
{
  if (eps < 1e-10) return T(0);
  const T urms = sqrt(Re_l / sqrt(15. / dnu / eps));
  const T CR = r1 + r2;
  const T taup1 = ratio_den * 4. * r1 * r1 / 18. / dnu, taup2 = ratio_den * 4. * r2 * r2 / 18. / dnu;
  const T leta = T(1);
  const T tauk = leta * leta / dnu;
  const T Te = Re_l * tauk / sqrt(15.);
  const T theta1 = 2.5 * taup1 / Te, theta2 = 2.5 * taup2 / Te;
  const T phi = mx(T(theta2 / theta1), T(theta1 / theta2));
  const T cw = 1. + 0.6 * exp(-pow(phi - 1., 1.5));
  T gamma = 0.183 * urms * urms / (dnu * dnu / leta / leta);
  gamma = phi * gamma;
  const T WrS2 = (dnu * dnu * CR * CR) / (leta * leta * leta * leta) / 15.;
  T WrA2 = urms * urms * gamma / (gamma - 1.)
    * ((theta1 + theta2) - 4. * theta1 * theta2 / (theta1 + theta2) * sqrt((1. + theta1 + theta2) / (1. + theta1) / (1. + theta2)))
    * (1. / (1. + theta1) / (1. + theta2) - 1. / (1. + gamma * theta1) / (1. + gamma * theta2));
  WrA2 = cw * WrA2;
  WrA2 = WrA2 / 3.;
  const T Wr = sqrt(2. / cst<T>::pi * (WrA2 + WrS2));
  const T A1 = 110.0, A2 = 0.38, A3 = 0.16;
  T alpha = log10(0.26 * sqrt(Re_l)) / log10(T(2.0));
  alpha = mx(alpha, T(1.e-20));
  const T CA = 0.06 * pow(Re_l, T(0.30)), CB = 0.4;
  const T StA = pow(A2 / A1 * Re_l, T(0.25));
  const T hlpr = cbrt(A2 / A3);
  const T StB = hlpr * hlpr * cbrt(Re_l);
  const T St1 = taup1 / tauk, St2 = taup2 / tauk;
  T y11, y21, y12, y22;
  if (St2 <= StA) { y11 = A1 * St1 * St1; y21 = 0.; } else { y11 = 0.; y21 = A2 * Re_l / (St1 * St1); }
  const T y31 = A3 * sqrt(Re_l / St1);
  if (St1 <= StA) { y12 = A1 * St2 * St2; y22 = 0.; } else { y12 = 0.; y22 = A2 * Re_l / (St2 * St2); }
  const T y32 = A3 * sqrt(Re_l / St2);
  const T za1 = 0.5 * (1. - tanh((log10(St1) - log10(StA)) / CA));
  const T zb1 = 0.5 * (1. + tanh((log10(St1) - log10(StB)) / CB));
  const T za2 = 0.5 * (1. - tanh((log10(St2) - log10(StA)) / CA));
  const T zb2 = 0.5 * (1. + tanh((log10(St2) - log10(StB)) / CB));
  const T gR1 = y11 * pow(za1, alpha) + y21 * pow(T(1.) - za1, alpha) + y31 * zb1 + 1.;
  const T gR2 = y12 * pow(za2, alpha) + y22 * pow(T(1.) - za2, alpha) + y32 * zb2 + 1.;
  const T xai = mx(T(taup2 / taup1), T(taup1 / taup2));
  const T RG12 = 2.6 * exp(-xai) + 0.205 * exp(-0.0206 * xai) * 0.5 * (1.0 + tanh(xai - 3.0));
  const T gR = 1. + RG12 * sqrt(gR1 - 1.) * sqrt(gR2 - 1.);
  return 2. * cst<T>::pi * CR * CR * Wr * gR;
}
// Wang et al. (2009) turbulent enhancement of the collision efficiency, [ratio][eps class][collector radius]
// (src/detail/wang_collision_enhancement.hpp:13-92)
__device__ const double wang_eta_e[11][2][7] = {
  {{1.74, 1.74, 1.773, 1.49, 1.207, 1.207, 1.0}, {4.976, 4.976, 3.593, 2.519, 1.445, 1.445, 1.0}},
  {{1.46, 1.46, 1.421, 1.245, 1.069, 1.069, 1.0}, {2.984, 2.984, 2.181, 1.691, 1.201, 1.201, 1.0}},
  {{1.32, 1.32, 1.245, 1.123, 1.000, 1.000, 1.0}, {1.988, 1.988, 1.475, 1.313, 1.150, 1.150, 1.0}},
  {{1.250, 1.250, 1.148, 1.087, 1.025, 1.025, 1.0}, {1.490, 1.490, 1.187, 1.156, 1.126, 1.126, 1.0}},
  {{1.186, 1.186, 1.066, 1.060, 1.056, 1.056, 1.0}, {1.249, 1.249, 1.088, 1.090, 1.092, 1.092, 1.0}},
  {{1.045, 1.045, 1.000, 1.014, 1.028, 1.028, 1.0}, {1.139, 1.139, 1.130, 1.091, 1.051, 1.051, 1.0}},
  {{1.070, 1.070, 1.030, 1.038, 1.046, 1.046, 1.0}, {1.220, 1.220, 1.190, 1.138, 1.086, 1.086, 1.0}},
  {{1.000, 1.000, 1.054, 1.042, 1.029, 1.029, 1.0}, {1.325, 1.325, 1.267, 1.165, 1.063, 1.063, 1.0}},
  {{1.223, 1.223, 1.117, 1.069, 1.021, 1.021, 1.0}, {1.716, 1.716, 1.345, 1.223, 1.100, 1.100, 1.0}},
  {{1.570, 1.570, 1.244, 1.166, 1.088, 1.088, 1.0}, {3.788, 3.788, 1.501, 1.311, 1.120, 1.120, 1.0}},
  {{20.3, 20.3, 14.6, 8.61, 2.60, 2.60, 1.0}, {36.52, 36.52, 19.16, 22.80, 26.0, 26.0, 1.0}}};
template <class T>
__device__ T wang_collision_enhancement(T r1, T r2, T eps)
{
  const T R0[7] = {10e-6, 20e-6, 30e-6, 40e-6, 50e-6, 60e-6, 100e-6};
  const T rat[11] = {0., .1, .2, .3, .4, .5, .6, .7, .8, .9, 1.};
  const T R = r1 > r2 ? r1 : r2, r = r1 > r2 ? r2 : r1;
  if (R > 100e-6) return T(1);
  const int n_eps = eps <= 2.5e-2 ? 0 : 1;
  int n_R0, n_rat;
  for (n_R0 = 0; n_R0 < 7; ++n_R0) if (R0[n_R0] > R) break;
  const T ratio = r / R;
  for (n_rat = 1; n_rat < 11; ++n_rat) if (rat[n_rat] > ratio) break;
  if (n_R0 == 0) return T(wang_eta_e[n_rat][n_eps][n_R0]);
  const T w0 = R - R0[n_R0 - 1], w1 = R0[n_R0] - R, w2 = ratio - rat[n_rat - 1], w3 = rat[n_rat] - ratio;
  return (T(wang_eta_e[n_rat - 1][n_eps][n_R0 - 1]) * w1 * w3 + T(wang_eta_e[n_rat - 1][n_eps][n_R0]) * w0 * w3 +
          T(wang_eta_e[n_rat][n_eps][n_R0 - 1]) * w1 * w2 + T(wang_eta_e[n_rat][n_eps][n_R0]) * w0 * w2)
         / (R0[n_R0] - R0[n_R0 - 1]) / (rat[n_rat] - rat[n_rat - 1]);
}
// ONISHI is a template parameter so that the other kernels' instantiation carries none of the code above
template <class T, bool ONISHI>
__device__ __forceinline__ T kernel_calc(const coal_kernel_cfg<T> &k, n_t na, n_t nb, T rw2a, T rw2b, T vta, T vtb, uint32_t cell)
{
  if constexpr (ONISHI) {                                                         // kernel_onishi::calc, kernels.hpp:209-250
    const T rwa = sqrt(rw2a), rwb = sqrt(rw2b);
    const T Re_l = k.params[0];
    const T rhod = k.rhod[cell];
    const T nograv = kernel_onishi_nograv<T>(rwa, rwb, Re_l, k.diss ? k.diss[cell] : T(0), k.eta[cell] / rhod, T(1e3) / rhod);
    const T geometric = k_geometric(na, nb, rw2a, rw2b, vta, vtb);
    // the reference passes k_params[0] (Re_lambda) as the enhancement's dissipation-rate argument: kept
    return interpolated_efficiency(k, rwa, rwb) * wang_collision_enhancement<T>(rwa, rwb, Re_l) * sqrt(geometric * geometric + nograv * nograv);
  }
  switch (k.kernel) {
    case LCX_KERNEL_GOLOVIN: {
      const n_t nmax = na < nb ? nb : na;
      return T(cst<T>::pi * 4. / 3. * k.params[0] * nmax * (rw2a * sqrt(rw2a) + rw2b * sqrt(rw2b)));
    }
    case LCX_KERNEL_GEOMETRIC:
      if (k.n_user_params == 1) return k_geometric(na, nb, rw2a, rw2b, vta, vtb) * k.params[0];
      return k_geometric(na, nb, rw2a, rw2b, vta, vtb);
    case LCX_KERNEL_LONG: {
      T res = k_geometric(na, nb, rw2a, rw2b, vta, vtb);
      const T r_L = mx(T(sqrt(rw2a)), T(sqrt(rw2b)));
      if (r_L < 50.e-6) {
        const T r_s = mn(T(sqrt(rw2a)), T(sqrt(rw2b)));
        if (r_s <= 3e-6) res = 0.; else res *= 4.5e8 * r_L * r_L * (1. - 3e-6 / r_s);
      }
      return res;
    }
    default:
      return interpolated_efficiency(k, T(sqrt(rw2a)), T(sqrt(rw2b))) * k_geometric(na, nb, rw2a, rw2b, vta, vtb);
  }
}
template <class T> struct u01_src { const T *arr; uint64_t call, seed; };

// One lane per CANDIDATE PAIR: lane t looks at position 2t of the (shuffled) cell-sorted order and owns the pair
// (p, p+1) that starts at 2t if 2t sits at an even in-cell offset with a cell-mate behind it, else the one that may
// start at 2t+1.  Every pair of the reference (even in-cell offset a, b = a+1 in the same cell, coal.ipp:198-212) is
// owned by exactly one lane and all 64 lanes of a wave carry work.  Pairs are disjoint, so the read-modify-write
// of the two SDs needs no atomics.  The collision count / who-was-bigger flags go to col[] exactly as in the
// reference (coal.ipp:209,233-267) because the kappa update (and tests) read them -- col == nullptr: nobody will (one kappa, the
// production kernel): 8 bytes per super-droplet and step that were written for nothing, 1.55 -> 1.40 ms on C3.
// TAB (as k_move's SPEC: uniform branches are dear): the production configuration compiled for itself -- a tabulated-efficiency
// kernel (hall*, vohl*), random numbers from Philox, no per-particle rc2 / in-cloud time, used-up super-droplets marked in ijk
// (Measured and dropped, round 3: the terminal velocities of hskpng_vterm_all computed HERE by the lane that owns the pair, from the wet
// radii it has gathered anyway, and stored -- instead of the separate streaming pass (28 B per SD, 0.81 ms on C3).  Same values, but
// k_coal 1.55 -> 2.56 ms and k_move +0.1: the scattered 8-byte stores of vt and the logarithm per droplet cost more than the pass.
// Round 5, once more without the scattered stores: k_coal evaluating the pair's velocities WITHOUT storing them (the droplet that grew
// marked by a bit of ijk instead of the invalid flag) and k_move evaluating, storing and using every droplet's -- the same bits in every
// attribute, no separate pass in the stretch where the in-cell ranking waits for the memory system; k_coal 1.44 -> 1.67 ms, k_move
// 1.90 -> 2.53 (the table look-up and the logarithm per droplet in a pass that was bound by memory), the pass itself 0.66: the step
// 8.11 -> 8.41 ms.  The pass stays.
// Also round 5, for crowded cells (C5, 512 per cell, where a pair's members lie anywhere in 4 KB per attribute and this kernel takes
// 14.7 ps per droplet against 10.4 at 64 per cell): one workgroup per cell, the cell's multiplicities, wet radii and velocities
// gathered ONCE into LDS (24 B per droplet) and the pairs served from there -- the same droplets bit for bit, and 16.1 ms against
// 14.2 on C5: the L1 already spares the L2 most of the repeated lines, and 2.1e6 workgroups that each wait at a barrier behind their
// gathers are slower than the walk's independent lanes.)
template <class T, bool ONISHI, bool TAB = false>
__global__ void __launch_bounds__(BS)
k_coal(size_t n_part, const uint32_t *sorted_id, const uint32_t *sorted_ijk, const uint32_t *cell_start,
       n_t *n, T *rw2, T *vt, T *rd3, T *col, const T *dv, T dt, coal_kernel_cfg<T> kc, u01_src<T> rs,
       int pure_const_multi, int *increase_sstp_coal, T *rc2, T *ict, uint32_t *ijk_mark)
{
  if (TAB) { kc.kernel = LCX_KERNEL_HALL; pure_const_multi = 0; rs.arr = nullptr; rc2 = nullptr; ict = nullptr; __builtin_assume(ijk_mark != nullptr); }
  const size_t p0 = 2 * gid();
  if (p0 + 1 >= n_part) return;                       // the reference's range is [0, n_part-1)
  // The three positions a lane may need (2t, 2t+1, 2t+2) and the CSR bounds of their first two cells are fetched up front, in two
  // dependent levels; then the pair's attributes in one batch.  (Written as the decision tree reads, the kernel walked six or seven
  // dependent loads, and 70 % of its wave-cycles were waits.)
  const bool has2 = p0 + 2 < n_part;
  const uint32_t c0 = sorted_ijk[p0], c1 = sorted_ijk[p0 + 1], c2 = has2 ? sorted_ijk[p0 + 2] : DEAD_CELL;
  const uint32_t i0 = sorted_id[p0], i1 = sorted_id[p0 + 1], i2 = has2 ? sorted_id[p0 + 2] : 0u;
  const uint32_t off0 = cell_start[c0], end0 = cell_start[c0 + 1], off1 = cell_start[c1], end1 = cell_start[c1 + 1];
  size_t p = p0;
  uint32_t ca = c0, off = off0, end = end0, a = i0, b = i1;
  {
    const bool even0 = ((uint32_t(p) - off) & 1u) == 0;
    if (!(even0 && c1 == ca)) {
      if (even0 && col) col[p] = T(0);                 // last SD of a cell with an odd count: no partner
      p = p0 + 1;
      if (!has2) return;
      if (c1 != ca) { ca = c1; off = off1; end = end1; }
      if ((uint32_t(p) - off) & 1u) return;            // odd offset: this SD is the b of the previous lane's pair
      if (c2 != ca) { if (col) col[p] = T(0); return; }
      a = i1; b = i2;
    }
  }
  const uint32_t cnt = end - off;
  const n_t nn = cnt;
  const T scl = nn > 1 ? (T(nn * (nn - 1)) / 2) / (nn / 2) : T(0);           // scale_factor, coal.ipp:99-107
  n_t na = n[a], nb = n[b];
  T rw2a = rw2[a], rw2b = rw2[b], vta = vt[a], vtb = vt[b], dvc = dv[ca];
  asm volatile("" : "+v"(rw2a), "+v"(rw2b), "+v"(vta), "+v"(vtb), "+v"(dvc), "+v"(na), "+v"(nb));   // (one batch, not a chain)
  const T prob = dt / dvc * scl * kernel_calc<T, ONISHI>(kc, na, nb, rw2a, rw2b, vta, vtb, ca);
  n_t col_no = n_t(prob);
  if (pure_const_multi && col_no >= 1) *increase_sstp_coal = 1;
  const T u = rs.arr ? rs.arr[p] : philox::u01<T>(p, rs.call, rs.seed);
  if (u < prob - col_no) ++col_no;
  if (col_no == 0) { if (col) { col[p] = T(0); col[p + 1] = T(0); } return; }
  if (na >= nb) {                                                            // collide<>, coal.ipp:110-143
    if (nb > 0) { const n_t q = na / nb; if (q < col_no) col_no = q; }
    n[a] = na - col_no * nb;
    if (ijk_mark && na == col_no * nb) ijk_mark[a] = DEAD_CELL;                // used up: out of the cell histogram at the next re-index (the fused move
                                                                               // then needs no look at n; the sorted order of this step still holds it)
    const T rw_b = cbrt(col_no * rw2a * sqrt(rw2a) + rw2b * sqrt(rw2b));
    rw2[b] = rw_b * rw_b;
    rd3[b] = col_no * rd3[a] + rd3[b];
    vt[b] = T(-1);
    if (rc2) rc2[b] = T(-1);                                                 // invalidator, coal.ipp:33-44,527-545
    if (ict) ict[b] = mx(ict[a], ict[b]);                                    // selector, coal.ipp:17-31,505-525
    if (col) col[p + 1] = T(-2);
  } else {
    if (na > 0) { const n_t q = nb / na; if (q < col_no) col_no = q; }
    n[b] = nb - col_no * na;
    if (ijk_mark && nb == col_no * na) ijk_mark[b] = DEAD_CELL;
    const T rw_a = cbrt(col_no * rw2b * sqrt(rw2b) + rw2a * sqrt(rw2a));
    rw2[a] = rw_a * rw_a;
    rd3[a] = col_no * rd3[b] + rd3[a];
    vt[a] = T(-1);
    if (rc2) rc2[a] = T(-1);
    if (ict) ict[a] = mx(ict[a], ict[b]);
    if (col) col[p + 1] = T(-1);
  }
  if (col) col[p] = T(col_no);
}
// weighted_summator, coal.ipp:57-97,458-480 (only with more than one kappa in the run)
template <class T>
__global__ void k_coal_kappa(size_t n_part, const uint32_t *sorted_id, const T *col, T *kpa, const T *rd3)
{
  const size_t p = gid(); if (p + 1 >= n_part) return;
  const T cn = col[p];
  if (cn <= 0) return;
  const uint32_t a = sorted_id[p], b = sorted_id[p + 1];
  const bool na_ge_nb = col[p + 1] == T(-2);
  const T rd3a = rd3[a], rd3b = rd3[b];
  T rd3_old = na_ge_nb ? rd3b - cn * rd3a : rd3a - cn * rd3b;
  T ka = kpa[a], kb = kpa[b];
  for (int ci = 0; ci < cn; ++ci) {
    if (na_ge_nb) { kb = (ka * rd3a + kb * rd3_old) / (rd3a + rd3_old); rd3_old += rd3a; }
    else          { ka = (kb * rd3b + ka * rd3_old) / (rd3b + rd3_old); rd3_old += rd3b; }
  }
  if (na_ge_nb) kpa[b] = kb; else kpa[a] = ka;
}

// ============================================================================================
// advection + sedimentation + subsidence + boundary conditions in ONE pass
// (adve.ipp:28-165,169-183; sedi.ipp:13-25; subs.ipp:13-25; bcnd.ipp:99-368)
// ============================================================================================
template <class T>
struct move_args {
  size_t n_part; grid_t g;
  T dx, dy, dz, x0, y0, z0, x1, y1, z1, dt;
  T *x, *y, *z; const T *vt, *rw2, *rd3; n_t *n; const uint32_t *ijk;
  const T *courant_x, *courant_y, *courant_z, *w_LS;
  const T *up, *vp, *wp;  // turb_adve (turb_adve.ipp:13-33): x += up dt, y += vp dt, z += wp dt after the advection; else nullptr
  int do_adve, scheme, halo, do_sedi, do_subs, do_bcnd, distmem, bcond_lft, bcond_rgt, open_side_walls, periodic_topbot;
  double *puddle_partial;      // [gridDim][4]: liq_vol, dry_vol, liq_num, prtcl_num
  uint8_t *mig;                // distmem: 1 = left the domain through the left face, 2 = right
  uint32_t *wg_mig;            // distmem: per workgroup, the number of its SDs flagged 1 (bits 0-9), flagged 2 (bits 10-19) and dead
                               // (bits 20-29): what the id lists and the dead count are built from without another pass over the
                               // flags and without an atomic in this kernel (k_mig_tiles3 / k_mig_off / k_mig_ids4)
  // fused re-indexing (single-device runs): the new cell index, the cell histogram with per-SD rank and the number of
  // dead SDs come out of the same pass, so post_copy needs no further sweep over the positions
  int reindex; uint32_t *ijk_out, *cnt, *rank; unsigned int *dead_count;
  int check_n;                 // look at n for SDs that have n == 0 without being marked dead (first move after init / set_particles)
};
template <class T>
__device__ __forceinline__ T adve_1d(int scheme, T x, uint32_t fl, T C_l, T C_r, T dx)
{
  const T f = T(fl);                 // the reference multiplies by a size_t index: same value, native u32 conversion
  if (scheme == LCX_ADVE_IMPLICIT) return (x + dx * (C_l - f * (C_r - C_l))) / (1 - (C_r - C_l));
  return 1 * x + (C_r - C_l) * (x - dx * f) + dx * C_l;
}
// bcnd.ipp:99-110: a + fmod((x - a) + 10 (b - a), b - a).  fmod is exact, so for a non-negative argument it equals
// fma(-q, L, arg) with q = trunc(arg / L) -- computed here with one division and a sign check instead of the ~100
// instruction library loop (the rounded quotient can only be one too large; the remainder is recomputed from the
// original argument, so the result is the exactly representable fmod value in every case)
template <class T> __device__ __forceinline__ T fmod_nonneg(T arg, T L)
{
  if (!(arg >= 0 && arg < L * T(1e6))) return fmod(arg, L);
  const T q = trunc(arg / L);
  T r = fma(-q, L, arg);
  if (r < 0) r = fma(-(q - 1), L, arg);
  else if (r >= L) r = fma(-(q + 1), L, arg);
  return r;
}
template <class T> __device__ __forceinline__ T periodic(T x, T a, T b) { return a + fmod_nonneg((x - a) + 10 * (b - a), b - a); }

// PC / TURB: the predictor-corrector scheme and the SGS velocity perturbations are separate instantiations (their extra
// live state costs the plain first-order pass 8..12 %)
// SPEC: the common configurations compiled for themselves -- a uniform branch costs this kernel far more than its instruction count
// says (k_move on C3: 2.52 ms generic, 2.22 with the dimension tests folded away, see DESIGN.md for the full profile).
//   bit 0: three-dimensional;  bit 1: the full step's pass (advection + sedimentation + boundary + re-indexing, no subsidence, no
//   Courant halo, periodic side walls, closed bottom / top);  with bit 1 -- bit 2: a slab with neighbours, bit 3: implicit scheme (else Euler)
constexpr int MOVE_3D = 1, MOVE_FULL = 2, MOVE_DISTMEM = 4, MOVE_IMPLICIT = 8;
template <class T, bool PC, bool TURB, int SPEC = 0>
__global__ void __launch_bounds__(BS) k_move(move_args<T> a)
{
  if (SPEC & MOVE_3D) { a.g.ndims = 3; __builtin_assume(a.g.nx > 0); __builtin_assume(a.g.ny > 0); __builtin_assume(a.g.nz > 0); }
  if (SPEC & MOVE_FULL) {
    a.do_adve = a.do_sedi = a.do_bcnd = a.reindex = 1; a.do_subs = 0; a.halo = 0; a.open_side_walls = 0; a.periodic_topbot = 0;
    a.distmem = (SPEC & MOVE_DISTMEM) ? 1 : 0; a.scheme = (SPEC & MOVE_IMPLICIT) ? LCX_ADVE_IMPLICIT : LCX_ADVE_EULER;
    if (!(SPEC & MOVE_DISTMEM)) a.wg_mig = nullptr;
  }
  __shared__ double red[4][BS / WAVE];
  const size_t i = gid();
  double pl = 0, pd = 0, pn = 0, pp = 0;
  unsigned int n_dead_wave = 0;
  uint32_t c = i < a.n_part ? a.ijk[i] : DEAD_CELL;
  bool dead_now = false;         // counted in dead_count: was dead already, or dies in this pass
  uint32_t c_new = DEAD_CELL;
  uint8_t mig_stored = 0;        // the migrant flag this lane stores (0 for a dead slot and for a lane behind the end)
  if (a.reindex && i < a.n_part) {
    if (c == DEAD_CELL) dead_now = true;                               // (incl. those that coalescence has just used up: k_coal marks them)
    else if (a.check_n && a.n[i] == 0) { dead_now = true; c = DEAD_CELL; }   // zero multiplicities from the initialisation / set_particles
  }
  if (c != DEAD_CELL) {
    const grid_t &g = a.g;
    const uint32_t nz = g.nz ? g.nz : 1, ny = g.ny ? g.ny : 1;          // 32-bit index arithmetic: n_cell < 2^32
    T x = g.nx ? a.x[i] : T(0), y = g.ny ? a.y[i] : T(0), z = g.nz ? a.z[i] : T(0);
    uint32_t ci = c, cj = 0, ck = 0;
    if (g.ndims >= 2) { const uint32_t cij = c / nz; ck = c - cij * nz; ci = cij; if (g.ndims == 3) { ci = cij / ny; cj = cij - ci * ny; } }
    uint32_t k_subs = ck;                               // subs reads w_LS at the k of `ijk`, which pred_corr leaves at the predictor cell
    if (!PC && a.do_adve && g.ndims > 0) {
      // euler / implicit; with a Courant halo (opts_init.adve_scheme == pred_corr, fallen back to first order for this
      // step) the arrays start `halo` planes to the left: adve_calc(true, halo_x), adve.ipp:169-183
      const uint32_t cih = ci + a.halo;
      const size_t ce = size_t(c) + size_t(a.halo) * (g.ndims == 1 ? 1u : g.ndims == 2 ? nz : nz * ny);
      const size_t rgt = ce + (g.ndims == 3 ? size_t(nz) * ny : size_t(g.nz));            // init_grid.ipp:96-121
      x = adve_1d(a.scheme, x, ci, a.courant_x[ce], a.courant_x[rgt], a.dx);
      if (g.ndims > 2) {
        const size_t fre = ce + size_t(cih) * nz;
        y = adve_1d(a.scheme, y, cj, a.courant_y[fre], a.courant_y[fre + nz], a.dy);
      }
      if (g.ndims > 1) {
        const size_t blw = g.ndims == 2 ? ce + cih : ce + size_t(ny) * cih + cj;
        z = adve_1d(a.scheme, z, ck, a.courant_z[blw], a.courant_z[blw + 1], a.dz);
      }
    } else if (PC && a.do_adve && g.ndims > 0) {
      // predictor-corrector with nearest-neighbour interpolation (adve.ipp:184-304), all in registers: coordinates that
      // start at the halo's left edge; predictor = explicit Euler from the old cell; corrector = the explicit increment at the
      // predicted position averaged with it
      const T shift = T(a.halo) * a.dx;
      x = x + shift;
      auto cell3 = [&](T xx, T yy, T zz, uint32_t &i_, uint32_t &j_, uint32_t &k_) {
        i_ = g.nx ? uint32_t(double(xx) / g.dx) : 0u; j_ = g.ny ? uint32_t(double(yy) / g.dy) : 0u; k_ = g.nz ? uint32_t(double(zz) / g.dz) : 0u;
      };
      auto face_C = [&](uint32_t i_, uint32_t j_, uint32_t k_, T &xl, T &xr, T &yl, T &yr, T &zl, T &zr) {
        const size_t ce = g.ndims == 1 ? size_t(i_) : g.ndims == 2 ? size_t(i_) * nz + k_ : (size_t(i_) * ny + j_) * nz + k_;
        xl = a.courant_x[ce]; xr = a.courant_x[ce + (g.ndims == 3 ? size_t(nz) * ny : size_t(g.nz))];
        if (g.ndims > 2) { const size_t fre = ce + size_t(i_) * nz; yl = a.courant_y[fre]; yr = a.courant_y[fre + nz]; }
        if (g.ndims > 1) { const size_t blw = g.ndims == 2 ? ce + i_ : ce + size_t(ny) * i_ + j_; zl = a.courant_z[blw]; zr = a.courant_z[blw + 1]; }
      };
      uint32_t i1, j1, k1;
      T xl, xr, yl = 0, yr = 0, zl = 0, zr = 0;
      cell3(x, y, z, i1, j1, k1);
      T x_old = x, y_old = y, z_old = z;
      face_C(i1, j1, k1, xl, xr, yl, yr, zl, zr);
      x = 1 * x + (xr - xl) * (x - a.dx * T(i1)) + a.dx * xl;
      if (g.ndims > 2) y = 1 * y + (yr - yl) * (y - a.dy * T(j1)) + a.dy * yl;
      if (g.ndims > 1) z = 1 * z + (zr - zl) * (z - a.dz * T(k1)) + a.dz * zl;
      if (g.ndims > 1) {
        if (z >= a.z1) z = a.z1 - T(1e-8) * a.dz;
        if (z <= a.z0) z = a.z0 + T(1e-8) * a.dz;
      }
      if (g.ndims == 3) {
        if (y >= a.y1) y_old = y_old + (a.y1 - a.y0);
        if (y < a.y0) y_old = y_old - (a.y1 - a.y0);
        y = periodic(y, a.y0, a.y1);
      }
      cell3(x, y, z, i1, j1, k1);
      k_subs = k1;
      x_old = x + x_old;
      if (g.ndims > 2) y_old = y + y_old;
      if (g.ndims > 1) z_old = z + z_old;
      face_C(i1, j1, k1, xl, xr, yl, yr, zl, zr);
      x = 0 * x + (xr - xl) * (x - a.dx * T(i1)) + a.dx * xl;
      if (g.ndims > 2) y = 0 * y + (yr - yl) * (y - a.dy * T(j1)) + a.dy * yl;
      if (g.ndims > 1) z = 0 * z + (zr - zl) * (z - a.dz * T(k1)) + a.dz * zl;
      x = (x + x_old) / T(2.);
      if (g.ndims > 2) y = (y + y_old) / T(2.);
      if (g.ndims > 1) z = (z + z_old) / T(2.);
      x = x - shift;
    }
    if (TURB) {
      if (g.nx) x = x + a.up[i] * a.dt;
      if (g.ny) y = y + a.vp[i] * a.dt;
      if (g.nz) z = z + a.wp[i] * a.dt;
    }
    if (a.do_sedi) z = z - a.dt * a.vt[i];
    if (a.do_subs) z = z - a.dt * a.w_LS[k_subs];
    bool kill = false, emigrant = false;
    uint8_t mig_flag = 0;
    if (a.do_bcnd && g.ndims > 0) {
      if (!a.distmem) {
        if (!a.open_side_walls) x = periodic(x, a.x0, a.x1);
        else if (x >= a.x1 || x < a.x0) kill = true;
      } else {
        uint8_t m = 0;
        if (x < a.x0) { m = 1; if (a.bcond_lft == 3) kill = true; }
        if (x >= a.x1) { m = 2; if (a.bcond_rgt == 3) kill = true; }
        // an SD that an open wall removes on its way out is not a migrant (it would travel with n == 0 only to be dropped
        // by the receiver); one that also precipitates or leaves through the top in this pass is decided below
        emigrant = m != 0;
        mig_flag = m;
      }
      if (g.ndims == 3) {
        if (!a.open_side_walls) y = periodic(y, a.y0, a.y1);
        else if (y >= a.y1 || y < a.y0) kill = true;
      }
      if (g.ndims > 1) {
        if (!a.periodic_topbot) {
          if (z >= a.z1) kill = true;
          if (z < a.z0) {
            // precipitation: the SD's multiplicity as it is AFTER the side-wall / top flags (bcnd.ipp:219-232)
            const double nf = kill ? 0. : double(T(a.n[i]));
            const T r2 = a.rw2[i];
            pl = 4. / 3. * cst<T>::pi * nf * pow(r2, T(3. / 2.));
            pd = 4. / 3. * cst<T>::pi * nf * a.rd3[i];
            pn = (r2 == T(0)) ? 0. : nf;
            pp = nf;
            kill = true;
          }
        } else z = periodic(z, a.z0, a.z1);
      }
    }
    if (g.nx) a.x[i] = x;
    if (g.ny) a.y[i] = y;
    if (g.nz) a.z[i] = z;
    // (every living SD stores its flag byte, dead slots are cleared at the end of the pass: no memset ahead of the launch.)  An SD that dies in this very pass
    // -- open wall, top, precipitation -- is not shipped: the reference sends it with n == 0 and the receiver's
    // hskpng_remove_n0 drops it, so the neighbour never sees it either way, and its puddle contribution stays on this slab
    if (a.distmem && a.do_bcnd) { mig_stored = kill ? uint8_t(0) : mig_flag; a.mig[i] = mig_stored; }
    if (kill) { a.n[i] = 0; dead_now = true; }
    else if (a.reindex) {
      // an emigrant leaves the cell-sorted order at once (it is packed by id and its multiplicity zeroed in migrate_finish)
      if (emigrant) dead_now = true; else c_new = cell_of(g, x, y, z);
    }
  }
  if (c == DEAD_CELL && a.distmem && a.do_bcnd && i < a.n_part) a.mig[i] = 0;       // a dead slot is nobody's migrant
  if (a.reindex) {                                    // every lane of the wave takes part (ballots inside)
    const bool live = c_new != DEAD_CELL;
    const uint32_t r = wave_hist_rank(a.cnt, c_new, live);
    if (i < a.n_part) { a.ijk_out[i] = c_new; if (live) a.rank[i] = r; }
    const unsigned long long db = __ballot(dead_now);
    // a slab with neighbours: the dead count goes into the workgroup's word below and is summed by k_mig_tiles3.  (Every wave of the
    // two boundary planes holds emigrants: 3e4 atomics on ONE address per step, device scope -- 0.1 ms of a 16-plane slab's 0.32 ms pass,
    // whatever the slab's size)
    if (a.wg_mig) n_dead_wave = (unsigned int)__popcll(db);
    else if (db && lane_id() == 0) atomicAdd(a.dead_count, (unsigned int)__popcll(db));
  }
  if (a.wg_mig) {
    __shared__ uint32_t wsum[BS / WAVE];
    const uint32_t packed = uint32_t(__popcll(__ballot(mig_stored == 1))) | (uint32_t(__popcll(__ballot(mig_stored == 2))) << 10) | (n_dead_wave << 20);
    if (lane_id() == 0) wsum[wave_id()] = packed;
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t m = 0;
      for (int w = 0; w < BS / WAVE; ++w) m += wsum[w];      // (at most 256 per field of ten bits: no carry between the fields)
      a.wg_mig[blockIdx.x] = m;
    }
  }
  if (a.puddle_partial) {
    // deterministic block reduction (fixed shuffle tree), one partial per workgroup, summed by the host in order
    if (__ballot(pp != 0.) != 0ull) {                  // (a wave without precipitation has four zeros to contribute: no shuffle tree)
#pragma unroll
      for (int d = WAVE / 2; d > 0; d >>= 1) { pl += __shfl_down(pl, d); pd += __shfl_down(pd, d); pn += __shfl_down(pn, d); pp += __shfl_down(pp, d); }
    }
    if (lane_id() == 0) { red[0][wave_id()] = pl; red[1][wave_id()] = pd; red[2][wave_id()] = pn; red[3][wave_id()] = pp; }
    __syncthreads();
    if (threadIdx.x < 4) {
      double s = 0;
      for (int w = 0; w < BS / WAVE; ++w) s += red[threadIdx.x][w];
      a.puddle_partial[size_t(blockIdx.x) * 4 + threadIdx.x] = s;
    }
  }
}

// fixed-order reduction of the per-workgroup precipitation partials (deterministic for a given launch geometry):
// workgroup g of the first launch reduces the contiguous slice [g*per, (g+1)*per) of the partials into out[g][4];
// the second launch (one workgroup, nblocks = number of slices, per = 1) reduces those.
__global__ void k_accumulate4(const double *s4, double *acc) { if (threadIdx.x < 4) acc[threadIdx.x] = acc[threadIdx.x] + s4[threadIdx.x]; }
// running != nullptr (the final launch): the running totals are advanced by the result as well
__global__ void __launch_bounds__(BS) k_sum_partials(const double *partials, size_t nblocks, size_t per, double *out, double *running = nullptr)
{
  __shared__ double red[4][BS];
  const size_t b0 = size_t(blockIdx.x) * per, b1 = b0 + per < nblocks ? b0 + per : nblocks;
  double acc[4] = {0, 0, 0, 0};
  for (size_t b = b0 + threadIdx.x; b < b1; b += BS)
    for (int k = 0; k < 4; ++k) acc[k] += partials[b * 4 + k];
  for (int k = 0; k < 4; ++k) red[k][threadIdx.x] = acc[k];
  __syncthreads();
  for (int d = BS / 2; d > 0; d >>= 1) {
    if (int(threadIdx.x) < d) for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x < 4) {
    out[size_t(blockIdx.x) * 4 + threadIdx.x] = red[threadIdx.x][0];
    if (running && blockIdx.x == 0) running[threadIdx.x] = running[threadIdx.x] + red[threadIdx.x][0];
  }
}

// ============================================================================================
// stable removal of SDs with n == 0 (hskpng_remove.ipp:20-76), fused with the re-indexing of
// post_copy (post_copy.ipp:18-35): survivors are written to the second buffer set in order, and their
// new cell index and histogram rank are produced in the same pass.
// ============================================================================================
__global__ void __launch_bounds__(BS) k_alive_tiles(const n_t *n, size_t n_part, uint32_t *tile_sums)
{
  __shared__ uint32_t lds[BS / WAVE];
  const size_t base = size_t(blockIdx.x) * SCAN_TILE;
  uint32_t cnt = 0;
  for (int it = 0; it < SCAN_TILE / BS; ++it) {
    const size_t i = base + size_t(it) * BS + threadIdx.x;
    const bool alive = i < n_part && n[i] != 0;
    cnt += __popcll(__ballot(alive));
  }
  if (lane_id() == 0) lds[wave_id()] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) { uint32_t s = 0; for (int w = 0; w < BS / WAVE; ++w) s += lds[w]; tile_sums[blockIdx.x] = s; }
}
// ext[]: further real-valued attributes that travel with a super-droplet (per-particle substepping state, rc2)
constexpr int MAX_EXT = 12;
template <class T> struct attr_set { n_t *n; T *rd3, *rw2, *kpa, *vt, *x, *y, *z; T *ext[MAX_EXT]; int n_ext; };

template <class T>
__global__ void __launch_bounds__(BS)
k_compact(size_t n_part, attr_set<T> src, attr_set<T> dst, const uint32_t *tile_offs, grid_t g,
          uint32_t *ijk, uint32_t *cnt, uint32_t *rank)
{
  __shared__ uint32_t lds[BS / WAVE];
  const size_t base = size_t(blockIdx.x) * SCAN_TILE;
  uint32_t run = tile_offs[blockIdx.x];
  for (int it = 0; it < SCAN_TILE / BS; ++it) {
    const size_t i = base + size_t(it) * BS + threadIdx.x;
    n_t nn = 0;
    if (i < n_part) nn = src.n[i];
    const bool alive = nn != 0;
    const unsigned long long bal = __ballot(alive);
    const uint32_t in_wave = __popcll(bal & ((1ull << lane_id()) - 1ull));
    if (lane_id() == 0) lds[wave_id()] = __popcll(bal);
    __syncthreads();
    uint32_t woff = 0, tot = 0;
    for (unsigned w = 0; w < BS / WAVE; ++w) { const uint32_t s = lds[w]; if (w < wave_id()) woff += s; tot += s; }
    __syncthreads();
    const size_t d = size_t(run) + woff + in_wave;
    uint32_t c = 0;
    if (alive) {
      dst.n[d] = nn; dst.rd3[d] = src.rd3[i]; dst.rw2[d] = src.rw2[i]; dst.kpa[d] = src.kpa[i]; dst.vt[d] = src.vt[i];
      T x = 0, y = 0, z = 0;
      if (g.nx) { x = src.x[i]; dst.x[d] = x; }
      if (g.ny) { y = src.y[i]; dst.y[d] = y; }
      if (g.nz) { z = src.z[i]; dst.z[d] = z; }
      for (int e = 0; e < src.n_ext; ++e) dst.ext[e][d] = src.ext[e][i];
      c = cell_of(g, x, y, z);
      ijk[d] = c;
    }
    const uint32_t r = wave_hist_rank(cnt, c, alive);
    if (alive) rank[d] = r;
    run += tot;
  }
}

// physical re-ordering of the storage into the cell-sorted order (opts_init.reorder_every): position pos of the sorted
// order becomes storage slot pos; dead SDs are not in the order, so this is a compaction as well
template <class T>
__global__ void __launch_bounds__(BS)
k_reorder(size_t n_part, const uint32_t *sorted_id, const uint32_t *sorted_ijk, attr_set<T> src, attr_set<T> dst, grid_t g, uint32_t *ijk_out)
{
  const size_t pos = gid(); if (pos >= n_part) return;
  const uint32_t i = sorted_id[pos];
  dst.n[pos] = src.n[i]; dst.rd3[pos] = src.rd3[i]; dst.rw2[pos] = src.rw2[i]; dst.kpa[pos] = src.kpa[i]; dst.vt[pos] = src.vt[i];
  if (g.nx) dst.x[pos] = src.x[i];
  if (g.ny) dst.y[pos] = src.y[i];
  if (g.nz) dst.z[pos] = src.z[i];
  for (int e = 0; e < src.n_ext; ++e) dst.ext[e][pos] = src.ext[e][i];
  ijk_out[pos] = sorted_ijk[pos];
}

// ============================================================================================
// moments / diagnostics (moms.ipp:50-387, particles_diag.ipp)
// ============================================================================================
// n_filtered = f(n or n_filtered, vec)  mode: 0 all, 1 range [mn,mx), 2 vec > 0
template <class T>
__global__ void k_nfilt(size_t n_part, int mode, int cons, const n_t *n, const T *vec, T vmin, T vmax, T *nf)
{
  const size_t i = gid(); if (i >= n_part) return;
  const T y = cons ? nf[i] : T(n[i]);
  if (mode == 0) nf[i] = y;
  else if (mode == 1) { const T v = vec[i]; nf[i] = (v >= vmin && v < vmax) ? y : T(0); }
  else nf[i] = y * (vec[i] > 0);
}
// selections by activation state (particles_diag.ipp:350-407): which 0: RH[cell] - S_cr >= 0, 1: rw2 >= rc2
template <class T>
__global__ void k_nfilt_act(size_t n_part, int which, const n_t *n, const T *rd3, const T *kpa, const T *rw2, const uint32_t *ijk,
                            const T *Tk, const T *RH, T *nf)
{
  const size_t i = gid(); if (i >= n_part) return;
  const uint32_t c = ijk[i];
  if (c == DEAD_CELL) { nf[i] = 0; return; }
  const T y = T(n[i]);
  if (which == 0) { const T v = RH[c] - S_cr(rd3[i], kpa[i], Tk[c]); nf[i] = y * (v >= 0); }
  else nf[i] = rw2[i] >= rc2_of(rd3[i], kpa[i], Tk[c]) ? y : T(0);
}
// mass_dens_estimator (mass_dens.ipp:14-34) per sorted position; the kernel width uses the SD count of the cell
template <class T>
__global__ void k_massdens_vals(size_t n_part, const uint32_t *sorted_id, const uint32_t *sorted_ijk, const uint32_t *cell_start,
                                const T *nf, const T *rw2, T rad, T sig0, T *out)
{
  const size_t p = gid(); if (p >= n_part) return;
  const uint32_t id = sorted_id[p], c = sorted_ijk[p];
  const T x = rw2[id], sig = sig0 / pow(T(cell_start[c + 1] - cell_start[c]), T(0.2));
  out[p] = nf[id] / sig * pow(x, 3 * T(.5)) * exp(-pow((log(pow(x, T(.5))) - log(rad)) / sig, T(2)) / T(2.));
}
template <class T> __global__ void k_massdens_scale(size_t n_cell, const uint32_t *cell_start, const T *dv, T prefactor, T *v)
{ const size_t c = gid(); if (c < n_cell && cell_start[c + 1] > cell_start[c]) v[c] = prefactor * v[c] / dv[c]; }
// per sorted position: value to be summed.  kind 0: n_f * vec^power (moment_counter, moms.ipp:243-275), 1: n_f > 0,
// 2: n_f * (rw2^(3/2) * vt)  (precip_rate, particles_diag.ipp:56-68 with power 1)
template <class T>
__global__ void k_mom_vals(size_t n_part, const uint32_t *sorted_id, const T *nf, const T *vec, const T *vec2, T power, int kind, T *out)
{
  const size_t p = gid(); if (p >= n_part) return;
  const uint32_t id = sorted_id[p];
  if (kind == 1) { out[p] = nf[id] > T(0) ? T(1) : T(0); return; }
  const T x = kind == 2 ? T(pow(vec[id], T(3. / 2)) * vec2[id]) : vec[id];
  out[p] = x >= 0 ? nf[id] * pow(x, power) : nf[id] * pow(x, T(int(power)));
}
template <class T>
__global__ void __launch_bounds__(BS)
k_cell_seqsum(size_t n_cell, int cfc, const uint32_t *cell_start, const T *vals, const T *dv, const T *rhod, int specific, T *out)
{
  __shared__ T lds[CF_CAP];
  const size_t c0 = size_t(blockIdx.x) * cfc;
  const size_t c1 = c0 + cfc < n_cell ? c0 + cfc : n_cell;
  const uint32_t base = cell_start[c0], end = cell_start[c1];
  const bool staged = (end - base) <= uint32_t(CF_CAP);
  if (staged) for (uint32_t q = base + threadIdx.x; q < end; q += BS) lds[q - base] = vals[q];
  __syncthreads();
  const size_t c = c0 + threadIdx.x;
  if (threadIdx.x >= cfc || c >= n_cell) return;
  const uint32_t s = cell_start[c], e = cell_start[c + 1];
  T acc = 0;
  if (e > s) {
    acc = seg_sum(lds, vals, staged, base, s, e);
    if (specific) { acc = acc / dv[c]; acc = acc / rhod[c]; }
  }
  out[c] = acc;
}
// max over a cell of vec (diag_max_rw)
template <class T>
__global__ void k_cell_max(size_t n_cell, const uint32_t *cell_start, const uint32_t *sorted_id, const T *vec, T *out)
{
  const size_t c = gid(); if (c >= n_cell) return;
  const uint32_t s = cell_start[c], e = cell_start[c + 1];
  T acc = 0;
  for (uint32_t q = s; q < e; ++q) { const T v = sqrt(vec[sorted_id[q]]); if (q == s || acc < v) acc = v; }
  out[c] = acc;
}

// ============================================================================================
// initialisation of super-droplets, sd_conc mode (src/impl/initialization/)
// ============================================================================================
template <class T>
__global__ void k_init_dry(size_t n_new, size_t n_old, n_t per_cell, T log_rd_min, T log_rd_max, u01_src<T> rs,
                           uint32_t *ijk, T *rd3, T *kpa, T kappa, T *vt, T *lnrd_out = nullptr /* the drawn ln(rd) itself, for the host (see init_SD_with_distros) */)
{
  const size_t gI = gid(); if (gI >= n_new) return;
  const size_t c = gI / per_cell;                                            // init_ijk.ipp:36-52 (cell-major)
  const size_t ptr = size_t(per_cell) * c;
  const T u = rs.arr ? rs.arr[gI] : philox::u01<T>(gI, rs.call, rs.seed);
  const T lnrd = log_rd_min + T(T(gI - ptr) + u) * (log_rd_max - log_rd_min) / T(per_cell);   // init_dry_sd_conc.ipp:26-34
  ijk[n_old + gI] = uint32_t(c);
  rd3[n_old + gI] = exp(3 * lnrd);
  if (lnrd_out) lnrd_out[gI] = lnrd;
  kpa[n_old + gI] = kappa;
  vt[n_old + gI] = T(-1);                                                    // resize fills vt with `invalid`
}
// constant-multiplicity / large-tail initialisation (init_ijk.ipp:36-52 with a per-cell count, init_dry_const_multi.ipp:52-80):
// SD g belongs to the cell whose offset range holds g; its dry radius is drawn from the tabulated CDF of the spectrum
// (index = upper_bound(cdf, u01)); multiplicity const_multi
template <class T>
__global__ void k_init_const_multi(size_t n_new, size_t n_old, const uint32_t *cell_off, uint32_t n_cell, const T *cdf, uint32_t n_cdf,
                                   T log_rd_min, T bin, u01_src<T> rs, n_t const_multi, uint32_t *ijk, T *rd3, T *kpa, T kappa, T *vt, n_t *n)
{
  const size_t gI = gid(); if (gI >= n_new) return;
  uint32_t a = 0, b = n_cell;                                  // last cell with cell_off[c] <= g
  while (b - a > 1) { const uint32_t m = a + (b - a) / 2; if (cell_off[m] <= gI) a = m; else b = m; }
  const T u = rs.arr ? rs.arr[gI] : philox::u01<T>(gI, rs.call, rs.seed);
  uint32_t lo = 0, hi = n_cdf;                                 // first index with cdf > u
  while (lo < hi) { const uint32_t m = lo + (hi - lo) / 2; if (!(u < cdf[m])) lo = m + 1; else hi = m; }
  const T lnrd = log_rd_min + T(lo) * bin;
  const size_t p = n_old + gI;
  ijk[p] = a; rd3[p] = exp(3 * lnrd); kpa[p] = kappa; vt[p] = T(-1); n[p] = const_multi;
}
// init_SD_with_sizes.ipp:14-77: `per_cell` SDs of one dry radius in every cell; multiplicity from the STP
// concentration (conc_to_number, init_count_num.ipp:41-70, and init_n.ipp:130-143)
template <class T>
__global__ void k_init_sizes(size_t n_new, size_t n_old, n_t per_cell, T rad3, T kappa, T conc0, const T *dv, const T *rhod,
                             const T *conc_factor, int nz, int indep_rhod, uint32_t *ijk, T *rd3, T *kpa, T *vt, n_t *n)
{
  const size_t gI = gid(); if (gI >= n_new) return;
  const size_t c = gI / per_cell, p = n_old + gI;
  ijk[p] = uint32_t(c); rd3[p] = rad3; kpa[p] = kappa; vt[p] = T(-1);
  T conc = conc0;
  conc = conc * dv[c];
  if (!indep_rhod) conc = rhod[c] / cst<T>::rho_stp * conc;
  if (conc_factor) conc = conc * conc_factor[c % nz];
  n[p] = n_t(conc / size_t(per_cell) + T(.5));
}
struct lognormal_modes { int n; double mean_rd[4], sdev[4], n_stp[4]; };
// init_n.ipp:48-143 with the built-in lognormal spectrum evaluated on the device
template <class T>
__global__ void k_init_n(size_t n_new, size_t n_old, const T *rd3, const uint32_t *ijk, const T *fvals, lognormal_modes lm,
                         T multiplier, const T *rhod, const T *dv, const T *conc_factor, int nz, int indep_rhod, int ndims,
                         T cellvol, n_t *n)
{
  const size_t gI = gid(); if (gI >= n_new) return;
  const size_t p = n_old + gI;
  const uint32_t c = ijk[p];
  T f;
  if (fvals) f = fvals[gI];
  else {
    const T lnrd = log(rd3[p]) / 3.;
    f = 0;
    if (lm.n < 0) {                 // the built-in exponential-in-volume spectrum (lcx_distro_t, n_modes = -1)
      const T r = exp(lnrd), q = pow(r, T(3)) / pow(T(lm.mean_rd[0]), T(3));
      f = T(lm.n_stp[0]) * T(3.) * q * exp(-q);
    }
    for (int m = 0; m < lm.n; ++m)
      f += T(lm.n_stp[m]) / sqrt(2 * cst<T>::pi) / log(T(lm.sdev[m])) *
           exp(-pow((lnrd - log(T(lm.mean_rd[m]))), T(2)) / T(2.) / pow(log(T(lm.sdev[m])), T(2)));
  }
  T v = multiplier * f;
  if (!indep_rhod) v = v * rhod[c] / cst<T>::rho_stp;
  if (conc_factor) v = v * conc_factor[c % nz];
  if (ndims > 0) v = v * dv[c] / cellvol;
  n[p] = n_t(v + T(0.5));
}
// init_wet.ipp:17-78
template <class T>
__global__ void k_init_wet(size_t n_new, size_t n_old, const T *rd3, const T *kpa, const uint32_t *ijk, const T *RH, const T *Tk, T RH_max, T *rw2)
{
  const size_t gI = gid(); if (gI >= n_new) return;
  const size_t p = n_old + gI;
  const uint32_t c = ijk[p];
  rw2[p] = pow(rw3_eq(rd3[p], kpa[p], mn(RH[c], RH_max), Tk[c]), T(2. / 3));
}
// init_xyz.ipp:40-74 for one dimension (dim: 0 x, 1 y, 2 z)
template <class T>
__global__ void k_init_pos(size_t n_new, size_t n_old, int dim, grid_t g, const uint32_t *ijk, u01_src<T> rs, T p0, T p1, T dp, T *pos)
{
  const size_t gI = gid(); if (gI >= n_new) return;
  const size_t p = n_old + gI;
  const size_t c = ijk[p], nz = g.nz ? g.nz : 1, ny = g.ny ? g.ny : 1;
  size_t ii;
  if (g.ndims == 1) ii = c;
  else if (g.ndims == 2) ii = dim == 0 ? c / nz : c % nz;
  else ii = dim == 0 ? c / (nz * ny) : dim == 1 ? (c / nz) % ny : c % nz;
  const T u = rs.arr ? rs.arr[gI] : philox::u01<T>(gI, rs.call, rs.seed);
  pos[p] = u * mn(p1, T((ii + 1) * dp)) + (1. - u) * mx(p0, T(ii * dp));
}

// ============================================================================================
// 1-D domain decomposition: migrant lists, pack, unpack (bcnd.ipp:160-205, pack.ipp:14-133, unpack.ipp:14-143)
// ============================================================================================
// stable list of ids with mig[i] == side (copy_if order): flags -> tile sums -> scan -> ids
__global__ void __launch_bounds__(BS) k_mig_tiles(const uint8_t *mig, size_t n_part, uint8_t side, uint32_t *tile_sums)
{
  __shared__ uint32_t lds[BS / WAVE];
  const size_t base = size_t(blockIdx.x) * SCAN_TILE;
  uint32_t cnt = 0;
  for (int it = 0; it < SCAN_TILE / BS; ++it) {
    const size_t i = base + size_t(it) * BS + threadIdx.x;
    cnt += __popcll(__ballot(i < n_part && mig[i] == side));
  }
  if (lane_id() == 0) lds[wave_id()] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) { uint32_t s = 0; for (int w = 0; w < BS / WAVE; ++w) s += lds[w]; tile_sums[blockIdx.x] = s; }
}
__global__ void __launch_bounds__(BS) k_mig_ids(const uint8_t *mig, size_t n_part, uint8_t side, const uint32_t *tile_offs, uint32_t *ids)
{
  __shared__ uint32_t lds[BS / WAVE];
  const size_t base = size_t(blockIdx.x) * SCAN_TILE;
  uint32_t run = tile_offs[blockIdx.x];
  for (int it = 0; it < SCAN_TILE / BS; ++it) {
    const size_t i = base + size_t(it) * BS + threadIdx.x;
    const bool sel = i < n_part && mig[i] == side;
    const unsigned long long bal = __ballot(sel);
    const uint32_t in_wave = __popcll(bal & ((1ull << lane_id()) - 1ull));
    if (lane_id() == 0) lds[wave_id()] = __popcll(bal);
    __syncthreads();
    uint32_t woff = 0, tot = 0;
    for (unsigned w = 0; w < BS / WAVE; ++w) { const uint32_t s = lds[w]; if (w < wave_id()) woff += s; tot += s; }
    __syncthreads();
    if (sel) ids[size_t(run) + woff + in_wave] = uint32_t(i);
    run += tot;
  }
}
// both directions in one pass each (three launches per step instead of six): tile_sums holds the left-going counts of all tiles,
// then the right-going ones (offset n_tiles); k_scan_sums2 scans the two halves with two workgroups
__global__ void __launch_bounds__(BS) k_mig_tiles2(const uint8_t *mig, size_t n_part, uint32_t n_tiles, uint32_t *tile_sums)
{
  __shared__ uint32_t lds[2][BS / WAVE];
  const size_t base = size_t(blockIdx.x) * SCAN_TILE;
  uint32_t cl = 0, cr = 0;
  for (int it = 0; it < SCAN_TILE / BS; ++it) {
    const size_t i = base + size_t(it) * BS + threadIdx.x;
    const uint8_t m = i < n_part ? mig[i] : uint8_t(0);
    cl += __popcll(__ballot(m == 1)); cr += __popcll(__ballot(m == 2));
  }
  if (lane_id() == 0) { lds[0][wave_id()] = cl; lds[1][wave_id()] = cr; }
  __syncthreads();
  if (threadIdx.x < 2) { uint32_t s = 0; for (int w = 0; w < BS / WAVE; ++w) s += lds[threadIdx.x][w]; tile_sums[threadIdx.x * n_tiles + blockIdx.x] = s; }
}
__global__ void k_scan_sums2(uint32_t *sums, size_t m, uint32_t *total)
{
  __shared__ uint32_t lds[SCAN_MAX_WAVES];
  uint32_t *a = sums + size_t(blockIdx.x) * m;
  uint32_t run = 0;
  for (size_t base = 0; base < m; base += blockDim.x) {
    const size_t i = base + threadIdx.x;
    const uint32_t v = i < m ? a[i] : 0u;
    uint32_t tot;
    const uint32_t ex = block_exclusive_scan(v, tot, lds);
    if (i < m) a[i] = run + ex;
    run += tot;
  }
  if (threadIdx.x == 0) total[blockIdx.x] = run;
}
// The same lists from the per-workgroup words that k_move leaves (wg_mig: left-going, right-going, dead -- ten bits each, one word per
// 256 SDs): the flags are read only where a count says there is something to find -- the boundary planes, a few per cent of a slab.
// (k_mig_tiles2 + k_scan_sums2 + k_mig_ids2 read every flag twice, a byte per lane: 0.36 ms per step on C3, 0.05 on a 16-plane slab;
// these four: 0.05 and 0.024 ms.)  A tile is BS workgroups of k_move, one per thread.  The step's dead count is the sum of the third
// field: one atomic per tile that has any.
__global__ void __launch_bounds__(BS) k_mig_tiles3(const uint32_t *wg_mig, size_t n_wg, uint32_t n_tiles, uint32_t *tile_sums, unsigned int *dead_count)
{
  __shared__ uint32_t lds[3][BS / WAVE];
  const size_t e = size_t(blockIdx.x) * BS + threadIdx.x;
  const uint32_t m = e < n_wg ? wg_mig[e] : 0u;
  uint32_t cl = m & 0x3ffu, cr = (m >> 10) & 0x3ffu, cd = m >> 20;
#pragma unroll
  for (int d = WAVE / 2; d > 0; d >>= 1) { cl += __shfl_down(cl, d); cr += __shfl_down(cr, d); cd += __shfl_down(cd, d); }
  if (lane_id() == 0) { lds[0][wave_id()] = cl; lds[1][wave_id()] = cr; lds[2][wave_id()] = cd; }
  __syncthreads();
  if (threadIdx.x < 3) {
    uint32_t s = 0;
    for (int w = 0; w < BS / WAVE; ++w) s += lds[threadIdx.x][w];
    if (threadIdx.x < 2) tile_sums[threadIdx.x * n_tiles + blockIdx.x] = s;
    else if (s) atomicAdd(dead_count, s);
  }
}
// offsets of every k_move workgroup's migrants in the two lists (behind k_scan_sums2 over the tiles)
__global__ void __launch_bounds__(BS) k_mig_off(const uint32_t *wg_mig, size_t n_wg, uint32_t n_tiles, const uint32_t *tile_offs, uint32_t *off_l, uint32_t *off_r)
{
  __shared__ uint32_t lds[SCAN_MAX_WAVES];
  const size_t e = size_t(blockIdx.x) * BS + threadIdx.x;
  const uint32_t pk = e < n_wg ? wg_mig[e] : 0u;
  uint32_t tot;
  const uint32_t ex_l = block_exclusive_scan(pk & 0x3ffu, tot, lds);
  __syncthreads();
  const uint32_t ex_r = block_exclusive_scan((pk >> 10) & 0x3ffu, tot, lds);
  if (e < n_wg) { off_l[e] = tile_offs[blockIdx.x] + ex_l; off_r[e] = tile_offs[n_tiles + blockIdx.x] + ex_r; }
}
// one WAVE per k_move workgroup: four flags per lane, ids written in ascending order behind the workgroup's offsets; a wave whose
// workgroup has no migrant (nearly all of them) ends after one scalar load
__global__ void __launch_bounds__(BS) k_mig_ids4(const uint8_t *mig, size_t n_part, const uint32_t *wg_mig, size_t n_wg,
                                                 const uint32_t *off_l, const uint32_t *off_r, uint32_t *ids_l, uint32_t *ids_r)
{
  const size_t e = size_t(blockIdx.x) * (BS / WAVE) + wave_id();
  if (e >= n_wg) return;
  if (!(wg_mig[e] & 0xfffffu)) return;
  const unsigned l = lane_id();
  const unsigned long long below = (1ull << l) - 1ull;
  const size_t i4 = e * BS + size_t(l) * 4;
  const uint32_t w = i4 < n_part ? *reinterpret_cast<const uint32_t *>(mig + i4) : 0u;      // (the flag array is allocated in whole workgroups)
  bool is_l[4], is_r[4];
  uint32_t pre_l = 0, pre_r = 0;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const uint32_t m = (w >> (8 * b)) & 0xffu;
    const bool valid = i4 + b < n_part;                // (bytes behind the last SD are stale)
    is_l[b] = valid && m == 1; is_r[b] = valid && m == 2;
    pre_l += __popcll(__ballot(is_l[b]) & below); pre_r += __popcll(__ballot(is_r[b]) & below);
  }
  size_t pl = size_t(off_l[e]) + pre_l, pr = size_t(off_r[e]) + pre_r;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    if (is_l[b]) ids_l[pl++] = uint32_t(i4 + b);
    if (is_r[b]) ids_r[pr++] = uint32_t(i4 + b);
  }
}
__global__ void __launch_bounds__(BS) k_mig_ids2(const uint8_t *mig, size_t n_part, uint32_t n_tiles, const uint32_t *tile_offs, uint32_t *ids_l, uint32_t *ids_r)
{
  __shared__ uint32_t lds[2][BS / WAVE];
  const size_t base = size_t(blockIdx.x) * SCAN_TILE;
  uint32_t run_l = tile_offs[blockIdx.x], run_r = tile_offs[n_tiles + blockIdx.x];
  for (int it = 0; it < SCAN_TILE / BS; ++it) {
    const size_t i = base + size_t(it) * BS + threadIdx.x;
    const uint8_t m = i < n_part ? mig[i] : uint8_t(0);
    const unsigned long long bl = __ballot(m == 1), br = __ballot(m == 2);
    const unsigned long long below = (1ull << lane_id()) - 1ull;
    if (lane_id() == 0) { lds[0][wave_id()] = __popcll(bl); lds[1][wave_id()] = __popcll(br); }
    __syncthreads();
    uint32_t woff_l = 0, tot_l = 0, woff_r = 0, tot_r = 0;
    for (unsigned w = 0; w < BS / WAVE; ++w) {
      const uint32_t sl = lds[0][w], sr = lds[1][w];
      if (w < wave_id()) { woff_l += sl; woff_r += sr; }
      tot_l += sl; tot_r += sr;
    }
    __syncthreads();
    if (m == 1) ids_l[size_t(run_l) + woff_l + __popcll(bl & below)] = uint32_t(i);
    if (m == 2) ids_r[size_t(run_r) + woff_r + __popcll(br & below)] = uint32_t(i);
    run_l += tot_l; run_r += tot_r;
  }
}
// attribute-major record: n[count] | rd3,rw2,kpa,vt,x,(y),(z)[count] ; x re-based to the receiver's frame
template <class T>
__global__ void k_pack(size_t count, const uint32_t *ids, attr_set<T> s, grid_t g, T x_rmt, T x_lcl, n_t *nb, T *rb)
{
  const size_t i = gid(); if (i >= count) return;
  const uint32_t id = ids[i];
  nb[i] = s.n[id];
  size_t slab = 0;
  rb[slab++ * count + i] = s.rd3[id]; rb[slab++ * count + i] = s.rw2[id]; rb[slab++ * count + i] = s.kpa[id]; rb[slab++ * count + i] = s.vt[id];
  if (g.nx) { const T xn = x_rmt + s.x[id] - x_lcl; s.x[id] = xn; rb[slab++ * count + i] = xn; }     // detail::remote, pack.ipp:14-26
  if (g.ny) rb[slab++ * count + i] = s.y[id];
  if (g.nz) rb[slab++ * count + i] = s.z[id];
  for (int e = 0; e < s.n_ext; ++e) rb[slab++ * count + i] = s.ext[e][id];
}
// free_l / free_r: storage slots the emigrants of this step have just vacated (their ids, already packed and flagged n = 0);
// slots [free_used, n_free_l + n_free_r) are still unclaimed.  An immigrant takes the next free slot, or is appended behind
// n_old when they are used up -- so a slab whose inflow balances its outflow keeps its storage extent and is not compacted every
// few steps.  (Only where the storage order is free to differ from the reference's, see opts_init.reorder_every; else n_free = 0.)
// cnt != nullptr: the step's re-indexing is fused into the passes over the positions (k_move for the SDs that stayed): cell index,
// histogram and arrival rank of the immigrant are produced here as well.
template <class T>
__global__ void __launch_bounds__(BS)
k_unpack(size_t count, size_t n_old, attr_set<T> s, grid_t g, const n_t *nb, const T *rb, T x0, T x1, T tol,
         const uint32_t *free_l, uint32_t n_free_l, const uint32_t *free_r, uint32_t n_free_r, uint32_t free_used,
         uint32_t *ijk, uint32_t *cnt, uint32_t *rank)
{
  const size_t i = gid();
  const bool in = i < count;
  uint32_t c = DEAD_CELL;
  size_t d = 0;
  if (in) {
    const size_t j = size_t(free_used) + i, n_free = size_t(n_free_l) + n_free_r;
    d = j < n_free_l ? size_t(free_l[j]) : j < n_free ? size_t(free_r[j - n_free_l]) : n_old + (j - (n_free > free_used ? n_free : size_t(free_used)));
    const n_t nn = nb[i];
    s.n[d] = nn;
    size_t slab = 0;
    s.rd3[d] = rb[slab++ * count + i]; s.rw2[d] = rb[slab++ * count + i]; s.kpa[d] = rb[slab++ * count + i]; s.vt[d] = rb[slab++ * count + i];
    T x = 0, y = 0, z = 0;
    if (g.nx) { x = rb[slab++ * count + i]; x = x >= x1 ? x - tol : x < x0 ? x + tol : x; s.x[d] = x; }   // tolerance_away_from_bcond
    if (g.ny) { y = rb[slab++ * count + i]; s.y[d] = y; }
    if (g.nz) { z = rb[slab++ * count + i]; s.z[d] = z; }
    for (int e = 0; e < s.n_ext; ++e) s.ext[e][d] = rb[slab++ * count + i];
    if (cnt) { c = nn == 0 ? DEAD_CELL : cell_of(g, x, y, z); ijk[d] = c; }
  }
  if (cnt) {                                           // every lane of the wave takes part (ballots inside)
    const bool active = in && c != DEAD_CELL;
    const uint32_t r = wave_hist_rank(cnt, c, active);
    if (active) rank[d] = r;
  }
}
__global__ void k_flag_ids(size_t count, const uint32_t *ids, n_t *n) { const size_t i = gid(); if (i < count) n[ids[i]] = 0; }

// ---- the same three steps driven from the device (multi_HIP, lcx_multi.hpp; one process per GPU: libcloudphxx_amd/multi.py): the
// migrant counts never visit the host.  A message is an "inbox" on the RECEIVING device -- a header {count, overflow flag, the capacity
// the sender will use NEXT step} in its first EXCH_HDR bytes, then the records in TILES of EXCH_TILE super-droplets, each tile
// attribute-major (n[256] | rd3[256] | rw2 | kpa | vt | x (| y | z | ext...)) -- that the sender's pack kernel writes straight through
// the peer mapping (xGMI), or into an outbox that RCCL ships: one message per direction, no count round trip.  Tiles make any prefix
// of a message self-contained, so a transport that must fix the message size before the count is known (RCCL send / recv) ships the
// first K tiles and the rest only in the rare step that needs it.  A tile is one workgroup's worth: 2 KiB contiguous per attribute.
// Kernels are launched over the inbox capacity and read the counts from device memory.
constexpr size_t EXCH_HDR = 256;
constexpr uint32_t EXCH_TILE = 256;
static_assert(EXCH_TILE == uint32_t(BS), "one workgroup packs one tile");
template <class T> __host__ __device__ __forceinline__ size_t exch_rec_bytes(int n_attr) { return sizeof(n_t) + sizeof(T) * size_t(n_attr); }
template <class T> __host__ __device__ __forceinline__ size_t exch_msg_bytes(size_t n_rec, int n_attr)
{ return EXCH_HDR + ((n_rec + EXCH_TILE - 1) / EXCH_TILE) * EXCH_TILE * exch_rec_bytes<T>(n_attr); }
// record r of a message: its multiplicity and its attribute `a`
template <class T> struct exch_tile {
  uint8_t *base; uint32_t j;
  __device__ __forceinline__ exch_tile(uint8_t *msg, size_t r, int n_attr) : base(msg + EXCH_HDR + (r / EXCH_TILE) * (EXCH_TILE * exch_rec_bytes<T>(n_attr))), j(uint32_t(r % EXCH_TILE)) {}
  __device__ __forceinline__ n_t &n() const { return reinterpret_cast<n_t *>(base)[j]; }
  __device__ __forceinline__ T &at(int a) const { return reinterpret_cast<T *>(base + EXCH_TILE * sizeof(n_t))[size_t(a) * EXCH_TILE + j]; }
};
// cap_rec: what the RECEIVER's inbox holds; next_cap: the records the sender will ship in the first part of its NEXT message (header word 2)
template <class T> struct pack_side { const uint32_t *ids; uint8_t *inbox; T x_rmt, x_lcl; uint32_t cap_rec, next_cap; };
// blocks [0, half) pack the left-going emigrants, [half, 2 half) the right-going ones (inbox == nullptr: that face has no neighbour);
// the multiplicity of a packed SD is cleared in the same pass (flag_lft / flag_rgt of the reference)
template <class T>
__global__ void __launch_bounds__(BS)
k_pack_dev(const uint32_t *counts, unsigned half, pack_side<T> L, pack_side<T> R, attr_set<T> s, grid_t g)
{
  const int side = blockIdx.x >= half;
  const pack_side<T> &P = side ? R : L;
  if (!P.inbox) return;
  const uint32_t count = counts[side];
  const size_t i = size_t(blockIdx.x - (side ? half : 0u)) * BS + threadIdx.x;
  if (i == 0) { uint32_t *h = reinterpret_cast<uint32_t *>(P.inbox); h[0] = count; h[1] = count > P.cap_rec ? 1u : 0u; h[2] = P.next_cap; }
  if (count > P.cap_rec || i >= count) return;        // overflow: nothing is shipped, the hosts of both slabs raise
  const int n_attr = 4 + g.ndims + s.n_ext;
  const exch_tile<T> t(P.inbox, i, n_attr);
  const uint32_t id = P.ids[i];
  t.n() = s.n[id];
  s.n[id] = 0;
  int a = 0;
  t.at(a++) = s.rd3[id]; t.at(a++) = s.rw2[id]; t.at(a++) = s.kpa[id]; t.at(a++) = s.vt[id];
  if (g.nx) { const T xn = P.x_rmt + s.x[id] - P.x_lcl; s.x[id] = xn; t.at(a++) = xn; }     // detail::remote, pack.ipp:14-26
  if (g.ny) t.at(a++) = s.y[id];
  if (g.nz) t.at(a++) = s.z[id];
  for (int e = 0; e < s.n_ext; ++e) t.at(a++) = s.ext[e][id];
}
// immigrants of both inboxes in one launch, the left neighbour's first (the reference unpacks lft, then rgt); slots as in k_unpack.
// have_l / have_r: how many records of each message have ARRIVED (a transport that ships a message in two parts, see above): a message
// that is not complete is not touched at all -- flag bit 2 asks the host for the rest and a second call.
// ov (the overlapped re-sort, lcx_core.hip exch_*): cells [c_lo, c_hi) are the slab's interior, where no immigrant may land (flag
// bit 8); flags[1] counts the living immigrants of the left message (the shift of the sorted order), flags[2] = the storage extent
// behind the unpack, big_count / big_mark: the list of crowded cells is marked (first call) or wound back to the mark (second call).
template <class T>
__global__ void __launch_bounds__(BS)
k_unpack_dev(const uint8_t *inbox_l, const uint8_t *inbox_r, uint32_t have_l, uint32_t have_r, size_t n_old, size_t cap, attr_set<T> s, grid_t g, T x0, T x1, T tol,
             const uint32_t *free_l, const uint32_t *free_r, const uint32_t *n_free /* [2] on the device, nullptr: no slot re-use */,
             uint32_t *ijk, uint32_t *cnt, uint32_t *rank, uint32_t *flags, int ov, uint32_t c_lo, uint32_t c_hi,
             uint32_t *big_count, uint32_t *big_mark, int retry, uint8_t *wave_flag = nullptr)
{
  auto hdr_count = [](const uint8_t *b) { const uint32_t *h = reinterpret_cast<const uint32_t *>(b); return b && !h[1] ? h[0] : 0u; };
  const uint32_t cl = hdr_count(inbox_l), cr = hdr_count(inbox_r);
  const bool incomplete = cl > have_l || cr > have_r;
  if (gid() == 0) {
    const uint32_t nf = n_free ? n_free[0] + n_free[1] : 0u, n_in = incomplete ? 0u : cl + cr;
    flags[2] = uint32_t(n_old) + (n_in > nf ? n_in - nf : 0u);
    if (ov) { if (retry) *big_count = *big_mark; else *big_mark = *big_count; }
    if (incomplete) atomicOr(flags, 4u);
  }
  if (incomplete) return;                              // (uniform over the launch)
  const size_t i = gid();
  bool in = i < size_t(cl) + cr;
  uint32_t c = DEAD_CELL;
  size_t d = 0;
  if (in) {
    const bool from_l = i < cl;
    const size_t k = from_l ? i : i - cl;
    const int n_attr = 4 + g.ndims + s.n_ext;
    const exch_tile<T> t(const_cast<uint8_t *>(from_l ? inbox_l : inbox_r), k, n_attr);
    const size_t n_free_l = n_free ? n_free[0] : 0, n_free_all = n_free_l + (n_free ? n_free[1] : 0);
    d = i < n_free_l ? size_t(free_l[i]) : i < n_free_all ? size_t(free_r[i - n_free_l]) : n_old + (i - n_free_all);
    if (d >= cap) { atomicOr(flags, 1u); in = false; }
    else {
      const n_t nn = t.n();
      s.n[d] = nn;
      int a = 0;
      s.rd3[d] = t.at(a++); s.rw2[d] = t.at(a++); s.kpa[d] = t.at(a++); s.vt[d] = t.at(a++);
      T x = 0, y = 0, z = 0;
      if (g.nx) { x = t.at(a++); x = x >= x1 ? x - tol : x < x0 ? x + tol : x; s.x[d] = x; }   // tolerance_away_from_bcond
      if (g.ny) { y = t.at(a++); s.y[d] = y; }
      if (g.nz) { z = t.at(a++); s.z[d] = z; }
      for (int e = 0; e < s.n_ext; ++e) s.ext[e][d] = t.at(a++);
      if (cnt) { c = nn == 0 ? DEAD_CELL : cell_of(g, x, y, z); ijk[d] = c; }
      if (wave_flag) wave_flag[d >> 6] = uint8_t(1);         // (the boundary pass of the re-sort visits this wave of the storage)
    }
  }
  if (cnt) {                                           // every lane of the wave takes part (ballots inside)
    const bool active = in && c != DEAD_CELL;
    const uint32_t r = wave_hist_rank(cnt, c, active);
    if (active) {
      if (ov && c >= c_lo && c < c_hi) atomicOr(flags, 8u);
      rank[d] = r;
    }
    if (ov) {
      const unsigned long long bl = __ballot(active && i < cl);
      if (bl && lane_id() == 0) atomicAdd(flags + 1, uint32_t(__popcll(bl)));
    }
  }
}
// ---- the overlapped re-sort's boundary pass (lcx_core.hip, exch_unpack).  The stayers' scan leaves the histogram in place, the unpack adds
// the immigrants to it (their ranks continue behind the stayers of their cells by themselves) and counts those of the left message: they
// can only land in the left boundary planes, so their number is by how much every later entry of the sorted order moves -- the host's
// arrays begin that far below their nominal start instead of anything being moved.  A second scan of the whole histogram (a few
// microseconds: cells, not super-droplets) then gives the final offsets; boundary SDs are scattered and ranked behind it.
// (A first version redid the two boundary regions' offsets in ONE workgroup: 59 us of latency for 32k cells against 18 for the whole scan.)
// rebuilds the histogram from CSR offsets (the second call of a step whose message arrived in two parts: the first one's scan has cleared it)
__global__ void k_csr_to_counts(const uint32_t *cell_start, uint32_t *cnt, size_t n_cell) { const size_t c = gid(); if (c < n_cell) cnt[c] = cell_start[c + 1] - cell_start[c]; }
// test / measurement only (LCX_TEST_PACK_DELAY_US): holds a stream for so many microseconds -- a neighbour that is late with its message
__global__ void k_spin_us(unsigned us) { const unsigned long long t0 = wall_clock64(); while (wall_clock64() - t0 < 100ull * us) __builtin_amdgcn_s_sleep(32); }

// the step's counts in one small record for ONE host read-back: dead, out_l, out_r, in_l, in_r, flags (1 my inbox overflowed at a
// sender, 2 storage full, 4 a message has not arrived in full yet, 8 an immigrant landed beyond the boundary planes), number of crowded
// cells and the largest occupancy (order_cells), [8] immigrants that joined the left boundary planes (the overlapped re-sort's shift of
// the sorted order, exch_*), [9] / [10] the capacity each neighbour will use for its NEXT message (header word 2; RCCL transport), [11] the
// population of the fuller pair of boundary planes; flag 16: the boundary ranking's grid was too small for it
__global__ void k_collect_counts(const uint32_t *step_cnt /* dead, n_big, max_big */, const uint32_t *out_cnt, const uint8_t *inbox_l, const uint8_t *inbox_r,
                                 const uint32_t *flags, const uint32_t *shift, uint32_t *rec, uint32_t *clear_step_cnt = nullptr,
                                 const uint32_t *cs_lo = nullptr, const uint32_t *cs_hi = nullptr, const uint32_t *cs_end = nullptr, uint32_t planned_pos = 0u)
{
  if (threadIdx.x != 0) return;
  const uint32_t *hl = reinterpret_cast<const uint32_t *>(inbox_l), *hr = reinterpret_cast<const uint32_t *>(inbox_r);
  rec[0] = step_cnt[0]; rec[1] = out_cnt[0]; rec[2] = out_cnt[1]; rec[6] = step_cnt[1]; rec[7] = step_cnt[2];
  rec[3] = hl ? hl[0] : 0u; rec[4] = hr ? hr[0] : 0u;
  uint32_t f = *flags;
  rec[11] = 0u;
  if (cs_lo) {                                       // the larger boundary population; was the boundary ranking's grid large enough for it?
    const uint32_t pop_l = *cs_lo, pop_r = *cs_end - *cs_hi, pop = pop_l > pop_r ? pop_l : pop_r;
    rec[11] = pop;
    if (pop > planned_pos) f |= 16u;
  }
  rec[5] = ((hl && hl[1]) || (hr && hr[1]) ? 1u : 0u) | ((f & 1u) ? 2u : 0u) | (f & 4u) | (f & 8u) | (f & 16u);
  rec[8] = shift ? *shift : 0u;
  rec[9] = hl ? hl[2] : 0u; rec[10] = hr ? hr[2] : 0u;
  // (the overlapped re-sort's scans leave the step's counters alone -- they are read here -- and have them cleared now, unless a
  // message is still incomplete and the boundary pass will run again)
  if (clear_step_cnt && !(f & 4u)) { clear_step_cnt[0] = 0u; clear_step_cnt[1] = 0u; clear_step_cnt[2] = 0u; }
}


// ============================================================================================
// recycling (housekeeping/particles_impl_rcyc.ipp:44-140): SDs with n == 0 become halves of the SDs with the highest
// multiplicities.  The reference sorts ALL multiplicities; here the k largest are found by a radix select (one 256-bin
// histogram pass per significant byte) and two ordered compactions.
// ============================================================================================
// stat[0..2] = number of SDs with n == 0, n == 1, n >= 2; stat64 = max n
__global__ void __launch_bounds__(BS) k_rcyc_stat(const n_t *n, size_t N, unsigned int *stat, unsigned long long *nmax)
{
  const size_t i = gid();
  const n_t v = i < N ? n[i] : n_t(2);
  const bool in = i < N;
  const unsigned long long b0 = __ballot(in && v == 0), b1 = __ballot(in && v == 1), b2 = __ballot(in && v >= 2);
  n_t m = in ? v : 0;
#pragma unroll
  for (int d = WAVE / 2; d > 0; d >>= 1) { const n_t o = __shfl_down(m, d); m = o > m ? o : m; }
  if (lane_id() == 0) {
    if (b0) atomicAdd(stat + 0, (unsigned int)__popcll(b0));
    if (b1) atomicAdd(stat + 1, (unsigned int)__popcll(b1));
    if (b2) atomicAdd(stat + 2, (unsigned int)__popcll(b2));
    atomicMax(nmax, m);
  }
}
// histogram of byte (n >> shift) & 255 over the SDs whose higher bytes equal `prefix`
__global__ void __launch_bounds__(BS) k_rcyc_hist(const n_t *n, size_t N, n_t prefix, int shift, unsigned int *hist)
{
  __shared__ unsigned int lds[256];
  lds[threadIdx.x] = 0;
  __syncthreads();
  const size_t i = gid();
  if (i < N) {
    const n_t v = n[i];
    const bool match = shift + 8 >= 64 ? true : (v >> (shift + 8)) == (prefix >> (shift + 8));
    if (match) atomicAdd(&lds[(v >> shift) & 255u], 1u);
  }
  __syncthreads();
  if (lds[threadIdx.x]) atomicAdd(hist + threadIdx.x, lds[threadIdx.x]);
}
// flag[i] = 1 where  mode 0: n == 0;  1: n == thr;  2: n > thr   (input of the ordered compaction k_mig_tiles / k_mig_ids)
__global__ void k_rcyc_flag(const n_t *n, size_t N, int mode, n_t thr, uint8_t *flag)
{
  const size_t i = gid(); if (i >= N) return;
  const n_t v = n[i];
  flag[i] = uint8_t(mode == 0 ? v == 0 : mode == 1 ? v == thr : v > thr);
}
__global__ void k_gather_n(const uint32_t *ids, size_t m, const n_t *n, n_t *out) { const size_t i = gid(); if (i < m) out[i] = n[ids[i]]; }
// receiver t takes every attribute of donor t and the bigger half of its multiplicity (rcyc.ipp:95-133)
template <class T>
__global__ void k_rcyc_apply(size_t k, const uint32_t *recv, const uint32_t *donor, attr_set<T> s, grid_t g)
{
  const size_t t = gid(); if (t >= k) return;
  const uint32_t r = recv[t], d = donor[t];
  s.rd3[r] = s.rd3[d]; s.rw2[r] = s.rw2[d]; s.kpa[r] = s.kpa[d]; s.vt[r] = s.vt[d];
  if (g.nx) s.x[r] = s.x[d];
  if (g.ny) s.y[r] = s.y[d];
  if (g.nz) s.z[r] = s.z[d];
  for (int e = 0; e < s.n_ext; ++e) s.ext[e][r] = s.ext[e][d];
  const n_t big = s.n[d];
  s.n[r] = big - big / 2;
  s.n[d] = big / 2;
}

// parity hook (lcx_math_probe): the device elementary functions on an array
// diag_vel_div (particles_diag.ipp:499-555): y, then z, then x face differences of the Courant numbers, each divided by dt;
// the face indices are those of k_move (init_grid.ipp:96-121), shifted by the Courant halo
template <class T>
__global__ void k_vel_div(size_t n_cell, grid_t g, int halo, T dt, const T *cx, const T *cy, const T *cz, T *out)
{
  const size_t c = gid(); if (c >= n_cell) return;
  const size_t nz = g.nz ? g.nz : 1, ny = g.ny ? g.ny : 1;
  const size_t plane = g.ndims == 1 ? 1 : g.ndims == 2 ? nz : nz * ny;
  const size_t ce = c + size_t(halo) * plane;
  T d = 0;
  if (g.ndims == 3) { const size_t fre = ce + (ce / (nz * ny)) * nz; d = d + (cy[fre + nz] - cy[fre]) / dt; }
  if (g.ndims >= 2) {
    const size_t blw = g.ndims == 2 ? ce + ce / nz : ce + ny * (ce / (nz * ny)) + (ce - (ce / (nz * ny)) * (nz * ny)) / nz;
    d = d + (cz[blw + 1] - cz[blw]) / dt;
  }
  const size_t rgt = ce + (g.ndims == 3 ? nz * ny : size_t(g.nz));
  d = d + (cx[rgt] - cx[ce]) / dt;
  out[c] = d;
}
// parity hook (LCX_DBG_TAG, lcx_rng_dump): what a coalescence call consumes of the generator, evaluated by the very device functions that
// the consumers call (sort_key: k_cellrank / k_cellsort_*; the u01 expression of k_coal), and the tags and cells by storage index
template <class T>
__global__ void k_rng_record(size_t n_pos, size_t n_store, rng_src r_un, u01_src<T> r_u01, const T *tag, const uint32_t *ijk,
                             T *out_u01, uint32_t *out_un, T *out_tag, uint32_t *out_ijk)
{
  const size_t i = gid();
  if (i < n_pos) out_u01[i] = r_u01.arr ? r_u01.arr[i] : philox::u01<T>(i, r_u01.call, r_u01.seed);
  if (i < n_store) { out_un[i] = uint32_t(sort_key(uint32_t(i), 1, r_un) >> 32); out_tag[i] = tag[i]; out_ijk[i] = ijk[i]; }
}
template <class T> __global__ void k_fill_index(T *a, size_t n) { size_t i = gid(); if (i < n) a[i] = T(i); }
// measurement hook ("raw_collided"): living super-droplets whose terminal velocity carries coalescence's invalid flag (vt = -1: the one
// that grew in a collision of the last step_async, coal.ipp:33-44) -- the number of pairs that collided; one atomic per wave
template <class T> __global__ void k_count_collided(size_t n, const T *vt, const uint32_t *ijk, unsigned long long *out)
{
  const size_t i = gid();
  const bool hit = i < n && ijk[i] != DEAD_CELL && vt[i] == T(-1);
  const unsigned long long bal = __ballot(hit);
  if (bal && lane_id() == unsigned(__ffsll((long long)bal) - 1)) atomicAdd(out + (blockIdx.x & 63u) * 8u, (unsigned long long)__popcll(bal));
}
// parity hook (lcx_philox_probe): raw Philox4x32-10 blocks for given (index, call, seed) triples
__global__ void k_philox_probe(const uint64_t *ics, size_t n, uint32_t *out)
{
  const size_t i = gid(); if (i >= n) return;
  uint32_t r[4];
  philox::gen(ics[3 * i], ics[3 * i + 1], ics[3 * i + 2], r);
  for (int k = 0; k < 4; ++k) out[4 * i + k] = r[k];
}
__global__ void k_math_probe(int which, double *v, size_t n)
{
  const size_t i = gid();
  if (i >= n) return;
  const double x = v[i];
  v[i] = which == 0 ? cbrt_seeded(x) : which == 1 ? exp_reduced(x) : which == 2 ? cbrt(x) : which == 4 ? rcp_refined(x) : which == 5 ? log_lean(x) :
         which == 6 ? rcp_newton1(x) : which == 7 ? cbrt1p<true>(x) : which == 8 ? exp_kelvin(x) : which == 9 ? exp_lib(x) : exp(x);
}

} // namespace lcx
