// host_threads.h — the thread team of the `_cpu` twins (csrc/gd3d_cpu.cpp, csrc/rbox_cpu.cpp): contiguous ranges of work units
// over plain std::threads.  No OpenMP runtime is pulled into a process that already has torch's.
#pragma once
#include <algorithm>
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
// a failed thread creation left unstarted)
template <typename F>
void parallel_ranges(int64_t units, int team, F&& body) {
  if (team <= 1) {
    body((int64_t)0, units);
    return;
  }
  std::vector<std::thread> th;
  th.reserve((size_t)team - 1);
  int unstarted = 0;
  auto range = [&](int r) { return units * r / team; };
  try {
    for (int r = 1; r < team; ++r) th.emplace_back([&, r] { body(range(r), range(r + 1)); });
  } catch (...) {
    unstarted = (int)th.size() + 1;
  }
  body(range(0), range(1));
  if (unstarted != 0) body(range(unstarted), units);
  for (auto& t : th) t.join();
}

}  // namespace gd3d_host
