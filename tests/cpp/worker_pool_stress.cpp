// Host-side stress of csrc/host_common.h's WorkerPool (the class text is pasted in front of this file by
// tests/test_abi.py::test_worker_pool_under_thread_sanitizer, which builds it with g++ -fsanitize=thread: no GPU, no HIP).
// Every run() must execute each of its parts exactly once — also right after the workers have gone to sleep, with the
// number of parts changing from call to call, and with two host threads taking turns at the pool.
#include <cstdio>
#include <cstdlib>

int main() {
  long total = 0;
  for (int it = 0; it < 60000; ++it) {
    const int nt = 2 + it % 7;
    std::atomic<int> hits{0};
    int part_hits[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    WorkerPool::get().run(nt, [&](int t) { part_hits[t] += 1; hits.fetch_add(1); });
    if (hits.load() != nt) { printf("BAD it=%d hits=%d nt=%d\n", it, hits.load(), nt); return 1; }
    for (int t = 0; t < nt; ++t) if (part_hits[t] != 1) { printf("BAD part %d ran %d times\n", t, part_hits[t]); return 1; }
    total += nt;
    if (it % 20000 == 0) std::this_thread::sleep_for(std::chrono::milliseconds(5));   // let them fall asleep
    if (it % 20000 == 1 || it % 3 == 0) WorkerPool::get().nudge();                    // a wake-up without work (asleep or not)
  }
  std::thread a([&] { for (int i = 0; i < 5000; ++i) { std::atomic<int> h{0}; WorkerPool::get().run(8, [&](int) { h.fetch_add(1); }); if (h != 8) { printf("BAD a\n"); _Exit(1); } } });
  std::thread b([&] { for (int i = 0; i < 5000; ++i) { std::atomic<int> h{0}; WorkerPool::get().run(3, [&](int) { h.fetch_add(1); }); if (h != 3) { printf("BAD b\n"); _Exit(1); } } });
  a.join(); b.join();
  printf("ok %ld\n", total);
  return 0;
}
