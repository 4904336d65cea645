// Copy workers of the host entry points (lp_run_batch / lp_detect), see api.cpp upload_images().  Header-only so that
// tests/native/copy_pool_stress.cpp can exercise it on the CPU (also under -fsanitize=thread).
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace lp {

// Host-side upload path of the drop-in entry points (lp_run_batch / lp_detect: the caller's images are ordinary pageable
// NumPy arrays).  A pageable hipMemcpyAsync is staged by the runtime through its own small pinned buffers, one image after
// the other: 24 GB/s of the link's 55.  Here a few worker threads copy groups of images into a pinned staging buffer of the
// handle while the DMA of the previous group runs (hipMemcpyAsync from pinned memory returns at once), so that the upload
// runs at the slower of {parallel memcpy, PCIe} instead of their sum.
class CopyPool {
 public:
  struct Job { const uint8_t* src; uint8_t* dst; size_t bytes; };
  explicit CopyPool(int n) {
    for (int i = 0; i < n; ++i) workers_.emplace_back([this] { loop(); });
  }
  ~CopyPool() {
    { std::lock_guard<std::mutex> l(m_); stop_ = true; }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
  }
  // copies every job (the calling thread takes part); returns when all are done AND no worker still looks at this batch
  void run(const Job* jobs, int n) {
    if (n <= 0) return;
    Batch b;
    b.jobs = jobs; b.n = n;
    { std::lock_guard<std::mutex> l(m_); b.serial = ++serial_; cur_ = &b; }
    cv_.notify_all();
    work(b);
    std::unique_lock<std::mutex> l(m_);
    cur_ = nullptr;   // no worker joins this batch from here on (they pick it up under the lock)
    cv_done_.wait(l, [&] { return active_ == 0 && b.done.load() == b.n; });
  }

 private:
  struct Batch {
    const Job* jobs = nullptr;
    int n = 0;
    unsigned long serial = 0;
    std::atomic<int> next{0}, done{0};
  };
  static void work(Batch& b) {
    int mine = 0;
    for (;;) {
      const int i = b.next.fetch_add(1);
      if (i >= b.n) break;
      memcpy(b.jobs[i].dst, b.jobs[i].src, b.jobs[i].bytes);
      ++mine;
    }
    if (mine) b.done.fetch_add(mine);
  }
  void loop() {
    unsigned long seen = 0;   // serial of the last batch this worker joined (a worker joins a batch once)
    for (;;) {
      Batch* b = nullptr;
      {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return stop_ || (cur_ && cur_->serial != seen); });
        if (stop_) return;
        b = cur_;
        seen = b->serial;
        ++active_;
      }
      work(*b);
      {
        std::lock_guard<std::mutex> l(m_);
        --active_;
      }
      cv_done_.notify_all();
    }
  }
  // (the batch lives on run()'s stack: run() clears cur_ under the lock and waits for active_ == 0 before it returns, so no worker
  //  touches a dead batch; every batch gets a fresh serial because consecutive ones share an address)
  std::vector<std::thread> workers_;
  std::mutex m_;
  std::condition_variable cv_, cv_done_;
  Batch* cur_ = nullptr;
  unsigned long serial_ = 0;
  int active_ = 0;
  bool stop_ = false;
};


}  // namespace lp
