// CPU stress of the library's copy workers (yolo-litepi_amd/csrc/copy_pool.h): 1500 batches of 1..97 jobs of varying size through
// an 8-thread pool, every batch verified byte for byte.  Built and run by tests/test_host_cpu.py (g++ -O2 -pthread; the same
// file is clean under -fsanitize=thread).
#include <cstdio>

#include "copy_pool.h"

int main() {
  lp::CopyPool pool(7);
  std::vector<uint8_t> src(32u << 20), dst(32u << 20);
  for (size_t i = 0; i < src.size(); ++i) src[i] = (uint8_t)((i * 2654435761u) >> 24);
  for (int it = 0; it < 1500; ++it) {
    std::vector<lp::CopyPool::Job> jobs;
    const int n = 1 + it % 97;
    const size_t sz = 1 + (size_t)(it * 7919) % 300000;
    for (int j = 0; j < n; ++j) jobs.push_back({src.data() + (size_t)j * sz, dst.data() + (size_t)j * sz, sz});
    memset(dst.data(), 0, (size_t)n * sz);
    pool.run(jobs.data(), n);
    if (memcmp(src.data(), dst.data(), (size_t)n * sz) != 0) { printf("MISMATCH at batch %d\n", it); return 1; }
  }
  printf("ok\n");
  return 0;
}
