// lcx_cond_wq.hpp -- the lean condensation kernel with a wave-private queue (k_cond_lean_wq), and what it shares with the kernels of
// lcx_kernels.hpp (launch geometry, cond_args, cond_list).  The kernel is compiled in a translation unit of its own (lcx_cond_wq.hip):
// it LOOPS over batches of droplets, and the compiler's machine-level loop-invariant code motion would move every table address and
// every literal of the inlined growth-rate evaluations in front of that loop and hold them in registers through it (107 vector
// registers, four waves per SIMD); lcx_cond_wq.hip is built with that pass off (96 registers, five waves, no scratch).  The other
// kernels keep the compiler's defaults.
#pragma once
#include "lcx_math.hpp"

namespace lcx {

constexpr int BS = 256;                 // 4 waves per workgroup
constexpr int WAVE = 64;

// (BS, not blockDim.x: every kernel that calls gid() / gid_xcd() is launched with BS threads, and the run-time value is a vector load from
// the dispatch packet with a full memory round trip ahead of the kernel's first own load -- one dependent level per wave, round 5)
__device__ __forceinline__ size_t gid() { return size_t(blockIdx.x) * BS + threadIdx.x; }
// Workgroups are dealt to the 8 XCDs round-robin (workgroup b runs on XCD b % 8; every XCD has its own 4 MiB L2).  gid_xcd hands each
// XCD runs of XCD_GROUP consecutive workgroups of the walk instead of every 8th one, so that the gathers of neighbouring cells
// (droplets that changed cell since the storage was last put in cell order sit in the neighbours' ranges) meet in ONE L2.
// Measured on k_cond_fast, C3 (ms): plain order 7.60; groups of 4: 7.77, 16: 7.31, 64: 7.15, 128..1024: 7.11..7.13, 8192: 7.22, one
// contiguous eighth per XCD: 7.33.  k_coal, k_move and k_scatter_sorted gain nothing (+-1 %; k_coal loses 17 % with eighths).
// The run length follows the cells: xcd_group() = the workgroups of ~2048 cells (C3: 512 workgroups; C5, 512 SDs per cell: 4096).
__host__ __device__ __forceinline__ unsigned xcd_group(size_t n_part, size_t n_cell)
{
  const size_t g = (n_part / (n_cell ? n_cell : 1) + 1) * 2048 / BS;
  return unsigned(g < 64 ? 64 : g > 8192 ? 8192 : g);
}
__device__ __forceinline__ size_t gid_xcd(unsigned group)
{
  const unsigned W = 8u * group;
  const unsigned b = blockIdx.x, w = b / W;
  const unsigned t = (w + 1) * W <= gridDim.x ? w * W + (b % 8u) * group + (b % W) / 8u : b;      // (the ragged tail keeps its order)
  return size_t(t) * BS + threadIdx.x;
}
__device__ __forceinline__ unsigned lane_id() { return threadIdx.x & (WAVE - 1); }
__device__ __forceinline__ unsigned wave_id() { return threadIdx.x / WAVE; }

constexpr uint32_t DEAD_CELL = 0xFFFFFFFFu;   // ijk of a super-droplet with n == 0 that has not been compacted away yet

template <class T>
struct cond_args {
  const uint32_t *sorted_id, *sorted_ijk;
  const n_t *n; const T *rd3, *kpa, *vt; T *rw2;
  const T *rhod, *rv, *Tk, *eta, *RH, *lambda_D, *lambda_K;
  T *m3_before, *m3_after;
  T dt_sub, RH_max, eps, cond_mlt; unsigned n_iter; int first; size_t n_cell;
  unsigned xcd_group;     // workgroups per XCD run (gid_xcd)
  const T *ssp;           // turb_cond: SGS supersaturation perturbation of the SD added to the cell's RH (RH_sgs), else nullptr
  const cond_cell_fast<T> *pre;   // fast arithmetic without turb_cond: the droplet-independent set-up, per cell (k_cond_cellpre)
  const uint32_t *storage_ijk;    // k_cond_lean in storage order (see there), else nullptr
  // the scatter of the re-sort that the end of the previous step left undone (k_scatter_sorted's two loads and two stores per droplet),
  // carried by the storage-order condensation kernel, whose memory pipes idle while its vector ALU is the bottleneck; else sc_rank == nullptr
  const uint32_t *sc_rank, *sc_cell_start; uint32_t *sc_sorted_id, *sc_sorted_ijk;
  unsigned fold_cap;      // k_cond_lean_fold: slots of its LDS stage in use (<= FOLD_CAP; smaller only in tests, opts_init.dbg_cond_budget)
};
// The list of droplets that k_cond_lean / k_cond_lean_wq leave to k_cond_lean_listed, in DEFER_SHARDS parts: workgroup b appends to part
// b % DEFER_SHARDS with the part's own counter (counters 64 B apart; one counter for everybody is one address for every wave of the
// launch).  ent: two words per entry (see k_cond_lean), part s at ent + 2 s shard_cap with room for every droplet of the workgroups that
// feed it -- WHO is listed never depends on the order in which the atomics are served; count[s * DEFER_CNT_STRIDE]: the entries of part s;
// budget: loop trips of the first pass; a droplet that has not converged by then leaves a RECORD -- the loop's state where it stands,
// lean_record -- in part s of `rec` (rcount[s * DEFER_CNT_STRIDE] of them, room for rec_cap: a droplet that finds its part full carries on
// in its own lane, the answer is the same), and k_cond_lean_resume takes the records up 64 to a wave: the first pass's waves no longer
// wait for their slowest droplet
constexpr int DEFER_SHARDS = 64, DEFER_CNT_STRIDE = 16;
struct cond_list { uint32_t *ent, *count; size_t shard_cap; unsigned budget; void *rec; uint32_t *rcount; uint32_t rec_cap; };
// x0 f0 x1 f1 c (lean_state), the far end of the reference's bracket (the near one is rw2_old; which is which rides on the sign bit of
// rd2), the squared dry radius for the two clamps; the kernel's index of the droplet and the position of its change
template <class T> struct lean_record { T v[7]; uint32_t idx, m3_pos; };

// Round 6: the lean kernel with a WAVE-PRIVATE QUEUE for the droplets whose first loop trip has not converged.  k_cond_lean pays for
// its slowest droplet: every lane runs the head and the first loop trip (uniform work: 93 % of the lanes busy), and then the wave makes
// two or three more trips for the fifth of its droplets that have not converged -- a third of the instructions it issues, at a lane use
// of 0.2.  Round 5's fold packed those droplets across the WORKGROUP (two barriers, three of four waves leave: fewer instructions, more
// waiting, not faster).  Here a wave owns n_batch batches of 64 consecutive storage slots and walks them one after the other: head +
// first trip for the batch at full lanes; the droplets that need more push the loop's state where it stands onto a queue in the wave's own
// stretch of LDS (ballot + prefix count: no barrier, nobody waits for anybody); whenever the queue holds FLUSH entries the wave runs
// the remaining trips for (up to) 64 of them at once, one droplet per lane, and goes on with its next batch.  The same operations on every
// droplet's numbers in the same order as k_cond_lean<T, 15, UNI, 0> (lean2_loop is resumable): the same rw2 and the same change of
// n rw^3 bit for bit (tests/test_hip_parity.py).
// An entry is 68 B: the secant's two points and the next iterate (x0 f0 x1 f1 c), the far end of the reference's bracket (the near one is
// rw2_old; which is which rides on the sign bit of rd2), the squared dry radius for the two clamps, the slot, the cell and the position of
// the droplet's change.  The droplet's own attributes and the cell's constants are read again by the lane that takes it up (L2: the wave
// read them a few microseconds ago) and set up with the same expressions -- the same bits as setup_cell's the first time.
// CAP 96 / FLUSH 33: 6.4 KB per wave, six workgroups per CU (the register budget's six waves per SIMD); CAP 128 / FLUSH 64: always full
// waves in the queue's trips, 8.5 KB per wave, four workgroups per CU.
template <int CAP, int PF = 0> struct wq_cfg { static constexpr int flush = CAP - WAVE + 1, waves = PF == 2 ? 3 : CAP <= 96 && !PF ? 5 : 4; };      // (waves per SIMD: the queues' LDS, the registers)
// what a storage slot indexes (the first of the two load levels)
template <class T> struct wq_slot { uint32_t c, rk; T rw2, rd3, vt, kpa; n_t n; };
// what the droplet's cell indexes (the second level)
template <class T> struct wq_cell { uint32_t cs; cond_cell_fast<T> cc; };
// the lanes below this one that are set in a ballot
__device__ __forceinline__ uint32_t mbcnt64(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u)); }
// the workgroup's place in the walk: gid_xcd's permutation of the workgroups (runs of `group` consecutive ones per XCD)
__device__ __forceinline__ unsigned blk_xcd(unsigned group)
{
  const unsigned W = 8u * group;
  const unsigned b = blockIdx.x, w = b / W;
  return (w + 1) * W <= gridDim.x ? w * W + (b % 8u) * group + (b % W) / 8u : b;
}
// The kernel's arguments as ONE structure at the start of the kernel-argument segment: the loop below reads them through a pointer to
// that segment which is a new value in every trip as far as the compiler can tell (an empty asm statement), so that a pointer is a scalar
// load next to its use -- kept across the loop, the fourteen pointers, the tables and the nested branches' masks exceed the 102 scalar
// registers, and the overflow lives in lanes of a vector register (v_readlane: vector issue slots).  The same for what derives from the
// lane index and for the uniform reals (1 - kappa, 2 dt ...): formed once in front of the loop they would each hold vector registers
// through it (the loop's form of the kernel wanted 99-107 of them where the straight-line kernel has 79: four waves per SIMD instead of six).
template <class T> struct wq_params { size_t n_part; cond_args<T> a; T kpa_uniform; cond_list lst; unsigned n_batch; };
// PF 1: the NEXT batch's slot-indexed loads (level one: the ones that come from HBM) are issued before the current batch is computed;
// PF 2: its cell-indexed loads as well, between the current batch's head and its first loop trip (the cell index has arrived by then)
template <class T, bool UNI, int CAP, int PF = 0>
__global__ void __launch_bounds__(BS) __attribute__((amdgpu_waves_per_eu(wq_cfg<CAP, PF>::waves, wq_cfg<CAP, PF>::waves))) k_cond_lean_wq(wq_params<T> prm)
{
  constexpr int NW = BS / WAVE, FLUSH = wq_cfg<CAP>::flush;
  __shared__ T xs_all[NW][7][CAP];
  __shared__ uint32_t xw_all[NW][3][CAP];
  typedef const wq_params<T> __attribute__((address_space(4))) *prm_ptr;
  const unsigned n_batch = prm.n_batch;
  const size_t chunk0 = size_t(blk_xcd(prm.a.xcd_group)) * n_batch;
  uint32_t cnt = 0;                                         // entries in the wave's queue (wave-uniform)
  wq_slot<T> nx{DEAD_CELL, 0u, T(0), T(0), T(0), T(0), n_t(0)};
  auto load_slot = [&](const auto &a, size_t pos, size_t n_part, T kpa_u) {
    wq_slot<T> v{DEAD_CELL, 0u, T(0), T(0), T(0), kpa_u, n_t(0)};
    if (pos < n_part) {
      v.c = a.storage_ijk[pos];
      if (a.sc_rank) v.rk = a.sc_rank[pos];
      v.rw2 = a.rw2[pos]; v.rd3 = a.rd3[pos]; v.vt = a.vt[pos];
      if (!UNI) v.kpa = a.kpa[pos];
      v.n = a.n[pos];
    }
    return v;
  };
  auto load_cell = [&](const auto &a, uint32_t c) {
    wq_cell<T> v;
    v.cs = 0;
    if (c != DEAD_CELL) {
      if (a.sc_rank) v.cs = a.sc_cell_start[c];
      v.cc = a.pre[c];
    }
    return v;
  };
  wq_cell<T> nx2;
  if constexpr (PF != 0) nx = load_slot(prm.a, chunk0 * BS + threadIdx.x, prm.n_part, prm.kpa_uniform);
  if constexpr (PF == 2) nx2 = load_cell(prm.a, nx.c);
  for (unsigned k = 0; k <= n_batch; ++k) {
    prm_ptr pp = (prm_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    unsigned tid = threadIdx.x;
    asm volatile("" : "+s"(pp), "+v"(tid));
    const cond_args<T> __attribute__((address_space(4))) &a = pp->a;
    const cond_list lst{pp->lst.ent, pp->lst.count, pp->lst.shard_cap, pp->lst.budget, nullptr, nullptr, 0u};
    const size_t n_part = pp->n_part;
    const unsigned lane = tid & (WAVE - 1);
    T (*xs)[CAP] = xs_all[tid / WAVE];
    uint32_t (*xw)[CAP] = xw_all[tid / WAVE];
    const T kpa_u = pp->kpa_uniform, dt_sub = a.dt_sub, eps = a.eps, cond_mlt = a.cond_mlt;
    // the growth rate's tables, loaded HERE (lcx_math.hpp lcx_tab: in a block that dominates the evaluations, inside the trip)
    const double *t_expk = nullptr, *t_cbrt = nullptr;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (sizeof(T) == 8) {
      t_expk = lcx_tab<true>(lcx_expk_c); t_cbrt = lcx_tab<true>(lcx_cbrt1p_c);
      lcx_tab_touch<14>(t_expk); lcx_tab_touch<5>(t_cbrt);
    }
#endif
    if (k < n_batch) {
      const size_t pos = (chunk0 + k) * BS + tid;
      bool need = false, several = false;
      uint32_t id = uint32_t(pos), m3_pos = uint32_t(pos);
      lean_state<T> s;
      T rd2 = 0;
      // (the two load levels of k_cond_lean, see there; a slot beyond the storage's end reads as a dead one)
      wq_slot<T> cur;
      wq_cell<T> cur2;
      if constexpr (PF != 0) {
        cur = nx;
        if constexpr (PF == 2) cur2 = nx2;
        if (k + 1 < n_batch) nx = load_slot(a, pos + BS, n_part, kpa_u); else nx.c = DEAD_CELL;
      } else cur = load_slot(a, pos, n_part, kpa_u);
      asm volatile("" ::: "memory");
      const uint32_t c = cur.c, rk = cur.rk;
      T rw2_old = cur.rw2, rd3 = cur.rd3, vt = cur.vt, kpa = cur.kpa;
      const n_t n_raw = cur.n;
      const bool live = c != DEAD_CELL;
      bool go = false;                        // the first loop trip is due
      cond_fun_fast<T, 31> ff;
      T r = 0, nn = 0;
      {
        if (live) {
          if constexpr (PF != 2) cur2 = load_cell(a, c);
          const uint32_t cs = cur2.cs;
          cond_cell_fast<T> cc = cur2.cc;
          asm volatile("" ::: "memory");
          if (a.sc_rank) { m3_pos = cs + rk; a.sc_sorted_id[m3_pos] = id; a.sc_sorted_ijk[m3_pos] = c; }
          nn = T(n_raw);
          asm volatile("" : "+v"(rw2_old), "+v"(rd3), "+v"(vt), "+v"(nn), "+v"(cc.Sc), "+v"(cc.Pr), "+v"(cc.lambda_D), "+v"(cc.lambda_K),
                       "+v"(cc.A), "+v"(cc.RH_eff), "+v"(cc.c1), "+v"(cc.c2_rho), "+v"(cc.RH_rho_w));
          if (!UNI) asm volatile("" : "+v"(kpa));
          asm volatile("" : "+v"(cc.two_rho_eta));
          r = rw2_old;
          if (!(rw2_old <= 0)) {
            ff.t_expk = t_expk; ff.t_cbrt = t_cbrt;
            ff.setup_cell(cc, rw2_old, dt_sub, rd3, kpa, vt);
            go = !lean2_head(ff, rw2_old, rd3, dt_sub, eps, cond_mlt, s, r, rd2, &several, lst.ent != nullptr);
            if constexpr (PF != 2) if (go) {
              need = !lean2_loop(ff, eps, 1u, s, r);
              if (!need) r = lean2_tail(s, r, rd2);
            }
          }
        }
      }
      if constexpr (PF == 2) {
        asm volatile("" ::: "memory"); nx2 = load_cell(a, nx.c); asm volatile("" ::: "memory");
        if (go) {
          need = !lean2_loop(ff, eps, 1u, s, r);
          if (!need) r = lean2_tail(s, r, rd2);
        }
      }
      if (live && !need && !several) {
        T delta = 0;
        if (!(rw2_old <= 0)) {
          a.rw2[id] = r;
          delta = nn * (r * (r * rsqrt_pos(r)) - rw2_old * (rw2_old * rsqrt_pos(rw2_old)));
        }
        a.m3_after[m3_pos] = delta;
      }
      // the droplets for the reference's iterates (cond_list), as in k_cond_lean
      const unsigned long long bsv = __ballot(several);
      if (bsv) {
        const int leader = __ffsll((long long)bsv) - 1;
        uint32_t base = 0;
        const unsigned shard = blockIdx.x % DEFER_SHARDS;
        if (int(lane) == leader) base = atomicAdd(lst.count + shard * DEFER_CNT_STRIDE, uint32_t(__popcll(bsv)));
        base = __shfl(base, leader);
        if (several) {
          const size_t q = 2 * (size_t(shard) * lst.shard_cap + base + mbcnt64(bsv));
          lst.ent[q] = id; lst.ent[q + 1] = m3_pos;
        }
      }
      // the droplets whose loop goes on: onto the wave's queue
      const unsigned long long bal = __ballot(need);
      if (bal) {
        if (need) {
          const uint32_t e = cnt + mbcnt64(bal);
          const bool grows = s.b != rw2_old;                // (shrinking: b = rw2_old + 0, growing: a = max(rd2, rw2_old + 0), see lean2_head)
          xs[0][e] = s.x0; xs[1][e] = s.f0; xs[2][e] = s.x1; xs[3][e] = s.f1; xs[4][e] = s.c;
          xs[5][e] = grows ? s.b : s.a;
          xs[6][e] = grows ? -rd2 : rd2;
          xw[0][e] = id; xw[1][e] = c; xw[2][e] = m3_pos;
        }
        cnt += uint32_t(__popcll(bal));
      }
    }
    if (cnt >= uint32_t(FLUSH) || (k == n_batch && cnt)) {
      const uint32_t m = cnt < uint32_t(WAVE) ? cnt : uint32_t(WAVE);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // (the wave's own LDS stores: in order, and done)
      cnt -= m;
      if (lane < m) {
        const uint32_t e = cnt + lane;
        lean_state<T> s;
        s.x0 = xs[0][e]; s.f0 = xs[1][e]; s.x1 = xs[2][e]; s.f1 = xs[3][e]; s.c = xs[4][e];
        const T far = xs[5][e], rd2s = xs[6][e];
        const uint32_t id = xw[0][e], c = xw[1][e], m3_pos = xw[2][e];
        T rw2_old = a.rw2[id], rd3 = a.rd3[id], vt = a.vt[id], kpa = kpa_u;
        if (!UNI) kpa = a.kpa[id];
        T nn = T(a.n[id]);
        cond_cell_fast<T> cc = a.pre[c];
        asm volatile("" ::: "memory");
        const bool grows = __builtin_signbit(rd2s);
        const T rd2 = fabs(rd2s);
        s.a = grows ? mx(rd2, rw2_old) : far; s.b = grows ? far : rw2_old;
        cond_fun_fast<T, 31> ff;
        ff.t_expk = t_expk; ff.t_cbrt = t_cbrt;
        ff.setup_cell(cc, rw2_old, dt_sub, rd3, kpa, vt);
        T r = s.c;
        lean2_loop(ff, eps, a.n_iter - 1u, s, r);
        r = lean2_tail(s, r, rd2);
        a.rw2[id] = r;
        a.m3_after[m3_pos] = nn * (r * (r * rsqrt_pos(r)) - rw2_old * (rw2_old * rsqrt_pos(rw2_old)));
      }
      asm volatile("" ::: "memory");
    }
  }
}


// the launch (lcx_cond_wq.hip); cap: 96 or 128 entries in a wave's queue
template <class T> void launch_cond_lean_wq(dim3 grid, hipStream_t st, const wq_params<T> &p, bool kpa_uniform, int cap, int prefetch);

}  // namespace lcx
