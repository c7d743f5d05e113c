#include "../../similari_amd/csrc/sa_pool.h"
#include <cstdio>
#include <chrono>
int main() {
  SaPool pool(7);
  std::vector<std::atomic<int>> hits(1000);
  long total = 0;
  for (int rep = 0; rep < 20000; ++rep) {
    uint32_t n = 1 + (rep * 7) % 200;
    for (uint32_t i = 0; i < n; ++i) hits[i] = 0;
    pool.run(n, [&](uint32_t i) { hits[i].fetch_add(1); });
    for (uint32_t i = 0; i < n; ++i) if (hits[i] != 1) { printf("BAD rep %d i %u = %d\n", rep, i, (int)hits[i]); return 1; }
    total += n;
    if (rep % 5000 == 0) std::this_thread::sleep_for(std::chrono::milliseconds(5));
  }
  auto t0 = std::chrono::steady_clock::now();
  for (int rep = 0; rep < 10000; ++rep) pool.run(8, [&](uint32_t i) { hits[i].fetch_add(1, std::memory_order_relaxed); });
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 10000;
  printf("ok %ld jobs; %.2f us per 8-job run\n", total, us);
}
