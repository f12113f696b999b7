// lcx_pool.hpp -- a few persistent host threads that run one job in parallel (the slab threads of the multi-device object,
// csrc/lcx_multi.hpp; the row copies between a caller's host arrays and the page-locked staging area, csrc/lcx_core.hip)
#pragma once
#include <condition_variable>
#include <cstdint>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace lcx {

// n persistent host threads; run(f) calls f(i) on thread i for every i and returns when all are done
class WorkerPool {
  int n_;
  std::vector<std::thread> th_;
  std::mutex m_;
  std::condition_variable go_, done_;
  std::function<void(int)> job_;
  uint64_t gen_ = 0;
  int pending_ = 0;
  bool stop_ = false;
  std::vector<std::exception_ptr> err_;
  void loop(int i)
  {
    uint64_t seen = 0;
    for (;;) {
      std::function<void(int)> f;
      {
        std::unique_lock<std::mutex> lk(m_);
        go_.wait(lk, [&] { return stop_ || gen_ != seen; });
        if (stop_) return;
        seen = gen_;
        f = job_;
      }
      std::exception_ptr e;
      try { f(i); } catch (...) { e = std::current_exception(); }
      {
        std::lock_guard<std::mutex> lk(m_);
        err_[i] = e;
        if (--pending_ == 0) done_.notify_all();
      }
    }
  }
public:
  explicit WorkerPool(int n) : n_(n), err_(n)
  { for (int i = 0; i < n; ++i) th_.emplace_back([this, i] { loop(i); }); }
  ~WorkerPool()
  {
    { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
    go_.notify_all();
    for (auto &t : th_) t.join();
  }
  // f(i) on worker i for every i; returns when all are done; rethrows the first failure (lowest slab index)
  void run(std::function<void(int)> f)
  {
    {
      std::unique_lock<std::mutex> lk(m_);
      job_ = std::move(f); pending_ = n_; ++gen_;
      for (auto &e : err_) e = nullptr;
    }
    go_.notify_all();
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [&] { return pending_ == 0; });
    for (auto &e : err_) if (e) std::rethrow_exception(e);
  }
};

} // namespace lcx
