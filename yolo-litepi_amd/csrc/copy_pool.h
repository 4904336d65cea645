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
#if defined(__x86_64__)
#include <emmintrin.h>
#endif

namespace lp {

// Copy into the pinned staging buffer with non-temporal stores: the destination is read next by the DMA engine, never by
// this CPU, so the lines need neither be fetched for ownership first (a plain store stream reads every destination line
// before it overwrites it: a third of the copy's memory traffic) nor be kept in the caches.
inline void stream_copy(uint8_t* dst, const uint8_t* src, size_t n) {
#if defined(__x86_64__)
  const size_t head = (64 - (reinterpret_cast<uintptr_t>(dst) & 63)) & 63;
  if (n < 4096 || head > n) { memcpy(dst, src, n); return; }
  memcpy(dst, src, head);
  dst += head; src += head; n -= head;
  const size_t body = n & ~(size_t)63;
  for (size_t i = 0; i < body; i += 64) {
    const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i));
    const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i + 16));
    const __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i + 32));
    const __m128i d = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i + 48));
    _mm_stream_si128(reinterpret_cast<__m128i*>(dst + i), a);
    _mm_stream_si128(reinterpret_cast<__m128i*>(dst + i + 16), b);
    _mm_stream_si128(reinterpret_cast<__m128i*>(dst + i + 32), c);
    _mm_stream_si128(reinterpret_cast<__m128i*>(dst + i + 48), d);
  }
  _mm_sfence();
  memcpy(dst + body, src + body, n - body);
#else
  memcpy(dst, src, n);
#endif
}

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
      stream_copy(b.jobs[i].dst, b.jobs[i].src, b.jobs[i].bytes);
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
