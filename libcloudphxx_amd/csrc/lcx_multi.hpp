// lcx_multi.hpp -- particles_t<real_t, multi_HIP>: ONE object that drives N devices of this process.
//
// Replaces the reference's multi_CUDA backend (src/particles_multi_gpu_ctor.ipp, _step.ipp:16-83, _diag.ipp,
// src/impl_multi_gpu/particles_multi_gpu_impl.ipp:17-227, ..._impl_step_async_and_copy.ipp:28-206): the domain is cut into x-slabs
// (src/detail/distmem_opts.hpp:10-52), every slab is a Particles<real_t> on its own device, every API call fans out to one host thread
// per slab, the caller hands over the GLOBAL Eulerian arrays (each slab reads / writes its planes, init_e2l.ipp:44-46).
//
// What differs from the reference's orchestration, by design:
//  * persistent worker threads (one per slab, bound to its device once) instead of std::threads spawned per call;
//  * the neighbour exchange of step_async is driven from the DEVICE: the sender's pack kernel reads the migrant count from device
//    memory and writes header + records straight into the receiver's inbox through the peer mapping (xGMI) -- one message per
//    direction, no count round trip through the host, no staging copy, no cudaMemcpyPeerAsync x 4; ordering between devices is by
//    HIP events (sender: packed -> receiver: stream wait), and the host threads meet at ONE barrier per step (so that an event is
//    recorded before a neighbour waits on it) where the reference has five and four cudaEventSynchronize;
//  * the step's counts (dead, emigrants, immigrants) reach the host in one read-back of 24 bytes after the unpack is queued.
#pragma once
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include "lcx_pool.hpp"

namespace lcx {

// rendezvous of the slab threads inside one call; a failing slab breaks it so that nobody waits for ever
class HostBarrier {
  std::mutex m_; std::condition_variable cv_; int n_, count_ = 0; uint64_t gen_ = 0; bool broken_ = false;
public:
  explicit HostBarrier(int n) : n_(n) {}
  void reset() { std::lock_guard<std::mutex> lk(m_); count_ = 0; broken_ = false; }
  void brk() { { std::lock_guard<std::mutex> lk(m_); broken_ = true; } cv_.notify_all(); }
  void wait()
  {
    std::unique_lock<std::mutex> lk(m_);
    if (broken_) throw lcx_error("libcloudph++: another slab of the multi-device object failed");
    const uint64_t g = gen_;
    if (++count_ == n_) { count_ = 0; ++gen_; cv_.notify_all(); return; }
    cv_.wait(lk, [&] { return gen_ != g || broken_; });
    if (gen_ == g) throw lcx_error("libcloudph++: another slab of the multi-device object failed");
  }
};

template <class real_t>
struct MultiParticles : IParticles {
  using T = real_t;
  lcx_opts_init_t glob;                          // global options (pointers inside are NOT valid after the constructor)
  int D = 0, n_dims = 0;
  std::vector<std::unique_ptr<Particles<T>>> slab;
  std::vector<int> dev, nx_loc, n_x_bfr;
  std::vector<hipEvent_t> ev_sent, ev_consumed;
  std::vector<char> peer_ok;                     // slab i can write its neighbours' inboxes from a kernel
  std::unique_ptr<WorkerPool> pool;
  std::unique_ptr<HostBarrier> barrier;
  size_t ncell_tot = 0;
  std::vector<size_t> n_rendezvous, n_rendezvous_hidden;
  std::vector<T> outbuf_glob;
  bool periodic = true;

  static int m1(int n) { return n == 0 ? 1 : n; }
  // distmem_opts.hpp:10-16 (`nx / size + .5` with an integer division)
  static int get_dev_nx(int nx, int rank, int size) { const int per = nx / size; return rank < size - 1 ? per : nx - rank * per; }
  int lft_of(int i) const { return i > 0 ? i - 1 : (periodic ? D - 1 : -1); }
  int rgt_of(int i) const { return i < D - 1 ? i + 1 : (periodic ? 0 : -1); }

  explicit MultiParticles(const lcx_opts_init_t &oi) : glob(oi)
  {
    // particles_multi_gpu_impl.ipp:46-82
    if (oi.chem_switch) throw lcx_error("libcloudph++: multi_CUDA is not yet compatible with chemistry. Use other backend or turn off opts_init.chem_switch.");
    if (oi.nx == 0) throw lcx_error("libcloudph++: multi_CUDA doesn't work for 0D setup.");
    if (!(oi.x1 > oi.x0 && oi.x1 <= oi.nx * oi.dx)) throw lcx_error("libcloudph++: !(x1 > x0 & x1 <= min(1,nx)*dx)");
    int avail = 0;
    HIPCHK(hipGetDeviceCount(&avail));
    // LCX_MULTI_DEVICE_MAP="0,0,0,0": slab -> device, several slabs may share a device (tests on a one-GPU box run the real exchange
    // path that way); without it slab i lives on device i like in the reference
    std::vector<int> map;
    if (const char *m = getenv("LCX_MULTI_DEVICE_MAP")) {
      for (const char *p = m; *p;) { map.push_back(atoi(p)); while (*p && *p != ',') ++p; if (*p == ',') ++p; }
      for (int d : map) if (d < 0 || d >= avail) throw lcx_error("libcloudph++: LCX_MULTI_DEVICE_MAP names a device that is not there");
    }
    D = oi.dev_count > 0 ? oi.dev_count : (map.empty() ? avail : int(map.size()));
    if (map.empty() && avail < D)
      throw lcx_error("number of available GPUs (" + std::to_string(avail) + ") smaller than number of GPUs defined in opts_init (" + std::to_string(D) + ")");
    if (!map.empty() && int(map.size()) < D) throw lcx_error("libcloudph++: LCX_MULTI_DEVICE_MAP lists fewer devices than opts_init.dev_count");
    if (D > oi.nx) throw lcx_error("Number of CUDA devices (" + std::to_string(D) + ") used is greater than nx (" + std::to_string(oi.nx) + ")");
    glob.dev_count = D; glob.dev_id = -1;
    periodic = !oi.open_side_walls;
    n_dims = oi.nx / m1(oi.nx) + oi.ny / m1(oi.ny) + oi.nz / m1(oi.nz);
    ncell_tot = size_t(m1(oi.nx)) * m1(oi.ny) * m1(oi.nz);
    outbuf_glob.assign(ncell_tot, T(0));
    dev.resize(D); nx_loc.resize(D); n_x_bfr.resize(D);
    for (int i = 0; i < D; ++i) dev[i] = map.empty() ? i : map[i];
    // peer access between neighbouring devices (particles_multi_gpu_impl.ipp:100-125)
    peer_ok.assign(D, (oi.dbg_flags & LCX_DBG_MULTI_NO_PEER) ? 0 : 1);      // (LCX_DBG_MULTI_NO_PEER: tests drive the staged path on one device)
    if (D > 1)
      for (int i = 0; i < D; ++i)
        for (int nb : {lft_of(i), rgt_of(i)}) {
          if (nb < 0 || dev[nb] == dev[i]) continue;
          int can = 0;
          HIPCHK(hipDeviceCanAccessPeer(&can, dev[i], dev[nb]));
          if (!can) { peer_ok[i] = 0; continue; }
          HIPCHK(hipSetDevice(dev[i]));
          const hipError_t e = hipDeviceEnablePeerAccess(dev[nb], 0);
          if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) peer_ok[i] = 0;
          (void)hipGetLastError();
        }
    // one inbox capacity for all slabs: the last slab takes the remainder of nx / D and would size its inbox smaller than what its
    // neighbours, with fewer planes, may send (and check against)
    const size_t exch_cap = Particles<T>::exch_capacity(size_t(oi.n_sd_max / D + 1), get_dev_nx(oi.nx, 0, D));
    slab.resize(D); n_rendezvous.assign(D, 0); n_rendezvous_hidden.assign(D, 0);
    ev_sent.assign(D, nullptr); ev_consumed.assign(D, nullptr);
    try {
    for (int i = 0; i < D; ++i) {
      lcx_opts_init_t o = oi;                                                      // distmem_opts.hpp:20-52
      const int bfr = D > 1 ? i * get_dev_nx(oi.nx, 0, D) : 0;
      if (D > 1) {
        o.nx = get_dev_nx(oi.nx, i, D);
        if (i != 0) o.x0 = 0.;
        if (i != D - 1) o.x1 = o.nx * o.dx; else o.x1 = oi.x1 - bfr * oi.dx;
        o.n_sd_max = oi.n_sd_max / D + 1;
        const bool first = i == 0, last = i == D - 1;
        o.bcond_lft = first && !periodic ? 3 : 1;                                  // particles_multi_gpu_impl.ipp:158-179
        o.bcond_rgt = last && !periodic ? 3 : 1;
        // the reference seeds every device alike (particles_multi_gpu_impl.ipp: the same opts_init on every device): distinct streams
        // per slab cost nothing here.  rng_seed_init (the initial sampling's own seed, if switched on) stays common as in the
        // reference: a caller who asks for a reproducible initial condition gets the same draws per slab on any device count
        o.rng_seed = oi.rng_seed + i;
      }
      o.n_x_tot = oi.nx; o.n_x_bfr = bfr; o.dev_id = dev[i]; o.dev_count = D;
      o.stream_ordered = 0;                     // (the slabs of this object are driven from worker threads that join per call)
      nx_loc[i] = o.nx; n_x_bfr[i] = bfr;
      HIPCHK(hipSetDevice(dev[i]));
      slab[i].reset(new Particles<T>(o));
      if (D > 1) {
        slab[i]->exch_alloc(exch_cap);
        HIPCHK(hipEventCreateWithFlags(&ev_sent[i], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&ev_consumed[i], hipEventDisableTiming));
        HIPCHK(hipEventRecord(ev_consumed[i], slab[i]->st));
      }
    }
    } catch (...) { release_devices(); throw; }     // (a slab that could not be built: the ones before it go, each under its own device)
    // an inbox that could not be had as fine-grained memory must not be written by a neighbour's kernel from another device (its
    // receiver could read stale cache lines): those senders pack at home and copy
    if (D > 1)
      for (int i = 0; i < D; ++i)
        for (int nb : {lft_of(i), rgt_of(i)})
          if (nb >= 0 && dev[nb] != dev[i] && !slab[nb]->inbox_finegrained) peer_ok[i] = 0;
    pool.reset(new WorkerPool(D));
    barrier.reset(new HostBarrier(D));
    if (serialize) fprintf(stderr, "libcloudph++ (multi_HIP): LCX_DBG_MULTI_SERIALIZE is set -- the %d slabs take turns (measurement mode)\n", D);
    pool->run([this](int i) { HIPCHK(hipSetDevice(dev[i])); });                    // each worker stays on its device
  }
  // slabs and events are freed with their own device current (the caller's device is put back by the C ABI's guard)
  void release_devices()
  {
    for (int i = 0; i < D; ++i) {
      (void)hipSetDevice(dev[i]);
      if (slab[i]) slab[i].reset();
      if (ev_sent[i]) { (void)hipEventDestroy(ev_sent[i]); ev_sent[i] = nullptr; }
      if (ev_consumed[i]) { (void)hipEventDestroy(ev_consumed[i]); ev_consumed[i] = nullptr; }
    }
  }
  ~MultiParticles() override
  {
    pool.reset();
    release_devices();
  }
  int real_kind() const override { return int(sizeof(T)); }

  // ---- fan-out helpers ----
  // opts_init.dbg_flags & LCX_DBG_MULTI_SERIALIZE (measurement only): the slab threads take turns, so that with all slabs on ONE device the wall time of a
  // call is the SUM of what each slab would spend alone on a device of its own (kernels + its host round trips), not their overlap
  std::mutex serial_mx;
  const bool serialize = (glob.dbg_flags & LCX_DBG_MULTI_SERIALIZE) != 0;
  template <class F> void each(F f)
  {
    barrier->reset();
    pool->run([&](int i) {
      std::unique_lock<std::mutex> lk(serial_mx, std::defer_lock);
      if (serialize) lk.lock();
      try { f(i, *slab[i]); if (serialize) slab[i]->sync(); } catch (...) { barrier->brk(); throw; }
    });
  }
  void rendezvous(int i)
  {
    if (!serialize) { barrier->wait(); return; }
    slab[i]->sync();                              // (this slab's queue drains while it still has the device to itself)
    serial_mx.unlock(); 
    try { barrier->wait(); } catch (...) { serial_mx.lock(); throw; }
    serial_mx.lock();
  }
  // per-slab view of a caller's array: on_device == 2 means `data` is a table of D device pointers, one slab-local array each
  struct Arr {
    lcx_arrinfo_t a; bool null;
    Arr(const lcx_arrinfo_t *src, int i) : null(!src || !src->data || !src->strides)
    {
      if (null) return;
      a = *src;
      if (src->on_device == 2) { a.data = static_cast<void *const *>(src->data)[i]; a.on_device = 3; null = a.data == nullptr; }
    }
    const lcx_arrinfo_t *p() const { return null ? nullptr : &a; }
  };

  void init(const lcx_arrinfo_t *th, const lcx_arrinfo_t *rv, const lcx_arrinfo_t *rhod, const lcx_arrinfo_t *p, const lcx_arrinfo_t *cx,
            const lcx_arrinfo_t *cy, const lcx_arrinfo_t *cz) override
  {
    each([&](int i, Particles<T> &s) {
      s.init(Arr(th, i).p(), Arr(rv, i).p(), Arr(rhod, i).p(), Arr(p, i).p(), Arr(cx, i).p(), Arr(cy, i).p(), Arr(cz, i).p());
    });
    exchange_courant_halo();
  }
  void sync_in(const lcx_arrinfo_t *th, const lcx_arrinfo_t *rv, const lcx_arrinfo_t *rhod, const lcx_arrinfo_t *cx, const lcx_arrinfo_t *cy,
               const lcx_arrinfo_t *cz, const lcx_arrinfo_t *diss) override
  {
    each([&](int i, Particles<T> &s) {
      s.sync_in(Arr(th, i).p(), Arr(rv, i).p(), Arr(rhod, i).p(), Arr(cx, i).p(), Arr(cy, i).p(), Arr(cz, i).p(), Arr(diss, i).p());
    });
    exchange_courant_halo();
  }
  // pred_corr reads Courant numbers up to two x-planes outside the slab (xchng_courants.ipp:15-160): peer copies of the neighbours'
  // edge planes into the halo planes.  (Global arrays fill the halo by themselves through the cyclic index map; per-slab arrays cannot.)
  void exchange_courant_halo()
  {
    if (D == 1 || !slab[0]->halo) return;
    each([&](int, Particles<T> &s) { s.sync(); });                                // the neighbours' arrays are complete
    each([&](int i, Particles<T> &s) {
      for (int which = 0; which < 3; ++which) {
        T *mine; size_t off[4];
        const size_t cnt = s.courant_halo_geom(which, &mine, off);
        if (!cnt) continue;
        const int nb[2] = {lft_of(i), rgt_of(i)};
        for (int side = 0; side < 2; ++side) {
          if (nb[side] < 0) continue;
          T *theirs; size_t offn[4];
          slab[nb[side]]->courant_halo_geom(which, &theirs, offn);
          // my left halo <- the left neighbour's planes next to ITS right edge, and vice versa
          HIPCHK(hipMemcpyPeerAsync(mine + off[2 + side], dev[i], theirs + offn[1 - side], dev[nb[side]], cnt * sizeof(T), s.st));
        }
      }
      s.sync();
    });
  }
  void step_cond(const lcx_opts_t &o, const lcx_arrinfo_t *th, const lcx_arrinfo_t *rv) override
  { each([&](int i, Particles<T> &s) { s.step_cond(o, Arr(th, i).p(), Arr(rv, i).p()); }); }

  // particles_multi_gpu_step.ipp:58-83 + ..._impl_step_async_and_copy.ipp:28-206
  void step_async(const lcx_opts_t &opts) override
  {
    if (opts.rcyc) throw lcx_error("libcloudph++: Particle recycling can't be used in the multi_CUDA backend (it would consume whole memory quickly");
    if (D == 1) { each([&](int, Particles<T> &s) { s.step_async(opts); }); return; }
    each([&](int i, Particles<T> &s) {
      s.step_async(opts);                         // coalescence ... advection + boundary + re-index of those that stay; migrant lists on the device
      const int lft = lft_of(i), rgt = rgt_of(i);
      // 1) pack into the neighbours' inboxes once they have consumed last step's message
      {
        typename Particles<T>::Range r(&s, "exchange_pack");
        if (lft >= 0) HIPCHK(hipStreamWaitEvent(s.st, ev_consumed[lft], 0));
        if (rgt >= 0 && rgt != lft) HIPCHK(hipStreamWaitEvent(s.st, ev_consumed[rgt], 0));
        const double lft_x1 = lft >= 0 ? slab[lft]->o.x1 : 0., rgt_x0 = rgt >= 0 ? slab[rgt]->o.x0 : 0.;
        const size_t cap_l = lft >= 0 ? slab[lft]->inbox_cap_rec : 0, cap_r = rgt >= 0 ? slab[rgt]->inbox_cap_rec : 0;
        if (peer_ok[i]) s.exch_pack(lft >= 0 ? slab[lft]->inbox[1].p : nullptr, lft_x1, cap_l, rgt >= 0 ? slab[rgt]->inbox[0].p : nullptr, rgt_x0, cap_r);
        else {
          // no peer mapping between the two devices: pack at home, then a peer copy of exactly the tiles used (one extra host
          // synchronisation for the two counts -- what the reference always does)
          for (auto &b : s.outbox) b.alloc(s.exch_bytes(std::max(cap_l, cap_r)));
          s.exch_pack(lft >= 0 ? s.outbox[0].p : nullptr, lft_x1, cap_l, rgt >= 0 ? s.outbox[1].p : nullptr, rgt_x0, cap_r);
          uint32_t out[2] = {0, 0};
          s.read_back(out, s.scan_total.p, 2);
          auto bytes = [&](uint32_t c, size_t capn) { return s.exch_bytes(c <= capn ? c : 0); };      // (an overflowing message is its header only)
          if (lft >= 0) HIPCHK(hipMemcpyPeerAsync(slab[lft]->inbox[1].p, dev[lft], s.outbox[0].p, dev[i], bytes(out[0], cap_l), s.st));
          if (rgt >= 0) HIPCHK(hipMemcpyPeerAsync(slab[rgt]->inbox[0].p, dev[rgt], s.outbox[1].p, dev[i], bytes(out[1], cap_r), s.st));
        }
      }
      HIPCHK(hipEventRecord(ev_sent[i], s.st));
      // work that does not depend on the neighbours runs while the messages travel: the precipitation sums, and the re-sort of the
      // slab's interior (scan, scatter, in-cell ranking of every cell that no immigrant can reach)
      s.puddle_reduce_deferred();
      s.exch_sort_interior();
      rendezvous(i);                              // every slab's `sent` event is recorded
      // The host runs ahead of its device (nothing above waits for the GPU): normally the pack kernel has not even started when
      // the threads meet here, i.e. the rendezvous costs the device nothing.  Counted for the record (lcx_timings).
      ++n_rendezvous[i];
      if (hipEventQuery(ev_sent[i]) == hipErrorNotReady) ++n_rendezvous_hidden[i];
      // 2) the neighbours' messages -> append, histogram; 3) counts to the host, scan / scatter / rank
      {
        typename Particles<T>::Range r(&s, "exchange_wait");
        if (lft >= 0) HIPCHK(hipStreamWaitEvent(s.st, ev_sent[lft], 0));
        if (rgt >= 0 && rgt != lft) HIPCHK(hipStreamWaitEvent(s.st, ev_sent[rgt], 0));
      }
      s.exch_unpack(lft >= 0, rgt >= 0);
      HIPCHK(hipEventRecord(ev_consumed[i], s.st));
      (void)s.exch_finish(opts);
    });
  }

  // ---- diagnostics: fan out, gather on the host (particles_multi_gpu_diag.ipp) ----
  void diag_cell(int w) override { each([&](int, Particles<T> &s) { s.diag_cell(w); }); }
  void diag_vel_div() override { each([&](int, Particles<T> &s) { s.diag_vel_div(); }); }
  void diag_sd_conc() override { each([&](int, Particles<T> &s) { s.diag_sd_conc(); }); }
  void diag_select(int mode, int cons, int attr, double a, double b) override { each([&](int, Particles<T> &s) { s.diag_select(mode, cons, attr, a, b); }); }
  void diag_mom(int attr, double power) override { each([&](int, Particles<T> &s) { s.diag_mom(attr, power); }); }
  void diag_precip_rate() override { each([&](int, Particles<T> &s) { s.diag_precip_rate(); }); }
  void diag_act(int w) override { each([&](int, Particles<T> &s) { s.diag_act(w); }); }
  void diag_wet_mass_dens(double rad, double sig0) override { each([&](int, Particles<T> &s) { s.diag_wet_mass_dens(rad, sig0); }); }
  void diag_max_rw() override { each([&](int, Particles<T> &s) { s.diag_max_rw(); }); }
  void outbuf(const void **data, size_t *n) override
  {                                                                                // particles_multi_gpu_diag.ipp:274-307
    const size_t plane = size_t(m1(glob.ny)) * m1(glob.nz);
    each([&](int i, Particles<T> &s) {
      const void *d; size_t m;
      s.outbuf(&d, &m);
      memcpy(outbuf_glob.data() + size_t(n_x_bfr[i]) * plane, d, m * sizeof(T));
    });
    *data = outbuf_glob.data(); *n = ncell_tot;
  }
  void get_attr(const char *, void *, size_t, size_t *) override { throw lcx_error("get_attr doesnt work in multi_CUDA backend."); }
  void diag_puddle(double *out) override
  {                                                                                // particles_multi_gpu_diag.ipp:246-268
    std::vector<std::vector<double>> loc(D, std::vector<double>(LCX_OUT_COUNT, 0.));
    each([&](int i, Particles<T> &s) { s.diag_puddle(loc[i].data()); });
    for (int k = 0; k < LCX_OUT_COUNT; ++k) { out[k] = 0; for (int i = 0; i < D; ++i) out[k] += loc[i][k]; }
  }
  size_t n_part() override { size_t n = 0; for (auto &s : slab) n += s->n_part(); return n; }
  size_t n_cell() override { return ncell_tot; }

  // ---- hooks that address ONE device's storage: use the slab handles (lcx_multi_slab) ----
  [[noreturn]] static void per_slab() { throw lcx_error("libcloudph++: this call addresses one device's storage; take the slab's handle with lcx_multi_slab()"); }
  void get_state_u64(const char *, unsigned long long *, size_t, size_t *) override { per_slab(); }
  void get_state_real(const char *, double *, size_t, size_t *) override { per_slab(); }
  void set_particles(size_t, const unsigned long long *, const double *, const double *, const double *, const double *, const double *,
                     const double *, const double *) override { per_slab(); }
  void rng_replay_push(int, const double *, size_t) override { per_slab(); }
  size_t rng_replay_pending() override { size_t n = 0; for (auto &s : slab) n += s->rng_replay_pending(); return n; }
  void rng_dump(int, int, double *, size_t, size_t *) override { per_slab(); }
  void set_state_real(const char *, const double *, size_t) override { per_slab(); }
  void stage(const char *, const lcx_opts_t *) override { per_slab(); }
  void migrate_counts(size_t *, size_t *) override { per_slab(); }
  size_t migrate_record_bytes() override { return slab[0]->migrate_record_bytes(); }
  void migrate_pack(int, double, void *, size_t) override { per_slab(); }
  void migrate_unpack(const void *, size_t) override { per_slab(); }
  void migrate_finish(const lcx_opts_t &) override { per_slab(); }
  size_t courant_halo_count(int w) override { return slab[0]->courant_halo_count(w); }
  void courant_halo_copy(int, int, void *, bool) override { per_slab(); }
  // per-stage device time: the slowest slab's (what the step waits for)
  std::vector<std::string> tnames; std::vector<double> tms;
  void timings(const char **names, double *ms, size_t capn, size_t *n) override
  {
    std::map<std::string, double> mx; std::vector<std::string> order;
    for (auto &s : slab) {
      const char *nm[64]; double v[64]; size_t k = 0;
      (void)hipSetDevice(s->o.dev_id);
      s->timings(nm, v, 64, &k);
      for (size_t j = 0; j < k; ++j) { if (!mx.count(nm[j])) order.push_back(nm[j]); mx[nm[j]] = std::max(mx[nm[j]], v[j]); }
    }
    // share of the steps in which every slab's thread passed the rendezvous before its own pack kernel had finished on the device
    size_t tot = 0, hid = 0;
    for (int i = 0; i < D; ++i) { tot += n_rendezvous[i]; hid += n_rendezvous_hidden[i]; }
    if (tot) { order.push_back("rendezvous_hidden_share"); mx["rendezvous_hidden_share"] = double(hid) / double(tot); }
    tnames = order; tms.clear();
    size_t k = 0;
    for (auto &nm : tnames) { if (k >= capn) break; names[k] = nm.c_str(); ms[k] = mx[nm]; ++k; }
    *n = k;
  }
  void set_profiling(int on) override
  {
    for (auto &s : slab) { (void)hipSetDevice(s->o.dev_id); s->set_profiling(on); }
    n_rendezvous.assign(D, 0); n_rendezvous_hidden.assign(D, 0);
  }
};

} // namespace lcx
