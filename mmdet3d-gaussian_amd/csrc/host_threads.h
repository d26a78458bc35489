// host_threads.h — the thread team of the `_cpu` twins (csrc/gd3d_cpu.cpp, csrc/rbox_cpu.cpp): contiguous ranges of work units
// over plain std::threads.  No OpenMP runtime is pulled into a process that already has torch's.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <thread>
#include <vector>

namespace gd3d_host {

// threads worth starting for `units` work units of which at least `min_per_thread` should go to each
inline int team_size(int32_t nthreads, int64_t units, int64_t min_per_thread, int64_t inline_below) {
  if (units < inline_below) return 1;
  int64_t t = nthreads > 0 ? nthreads : (int64_t)std::thread::hardware_concurrency();
  t = std::max<int64_t>(1, std::min<int64_t>(t, units / std::max<int64_t>(min_per_thread, 1)));
  return (int)std::min<int64_t>(t, 1024);
}

// body(first, last) over [0, units), one contiguous range per thread; the calling thread takes the first range (and everything
// a failed thread creation left unstarted).  Returns false when a body threw (std::bad_alloc in a worker's scratch vector ...):
// an exception escaping a std::thread would end the process in std::terminate, and nothing may cross the C ABI; the entry point
// then reports GD3D_E_HOST.  The other ranges still run to completion (their outputs are simply not to be used).
template <typename F>
[[nodiscard]] bool parallel_ranges(int64_t units, int team, F&& body) {
  std::atomic<bool> failed{false};
  auto guarded = [&](int64_t first, int64_t last) {
    try {
      body(first, last);
    } catch (...) {
      failed.store(true, std::memory_order_relaxed);
    }
  };
  if (team <= 1) {
    guarded((int64_t)0, units);
    return !failed.load();
  }
  std::vector<std::thread> th;
  int unstarted = 0;
  auto range = [&](int r) { return units * r / team; };
  try {
    th.reserve((size_t)team - 1);
    for (int r = 1; r < team; ++r) th.emplace_back([&, r] { guarded(range(r), range(r + 1)); });
  } catch (...) {
    unstarted = (int)th.size() + 1;
  }
  guarded(range(0), range(1));
  if (unstarted != 0) guarded(range(unstarted), units);
  for (auto& t : th) t.join();
  return !failed.load();
}

}  // namespace gd3d_host
