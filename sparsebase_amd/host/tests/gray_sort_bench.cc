// GrayIntroSort on the three shapes the Gray host stage sorts (packed degree keys with eleven values, (row, key) records
// with few and with distinct keys), with and without the team's partitions (par_min).  CPU only; `make -C
// sparsebase_amd/host stress` builds it as bin/gray_sort_bench.
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "sparsebase/sparsebase.h"
using namespace sparsebase;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
  const size_t n = argc > 1 ? atol(argv[1]) : 3000000;
  // "gpu" as second argument: a device handle first — the sorts then run in a process that holds a HIP context with its
  // runtime threads, as they do inside GrayReorder (does that explain the occasional sort that takes 3 x its time?)
  sbx_handle_t h = nullptr;
  if (argc > 2 && std::string(argv[2]) == "gpu" && sbx_create(0, &h) != SBX_OK) { printf("no device\n"); return 1; }
  unsigned long long state = 88172645463325252ull;
  auto rnd = [&state]() { state ^= state << 13, state ^= state >> 7, state ^= state << 17; return state; };
  auto by_degree = [](uint32_t l, uint32_t r) -> bool { return (l >> 24) < (r >> 24); };
  struct P { int first; unsigned long second; };
  auto asc = [](const P &l, const P &r) -> bool { return l.second < r.second; };
  std::vector<uint32_t> base(n);
  for (size_t i = 0; i < n; i++) base[i] = ((uint32_t)(rnd() % 11) << 24) | (uint32_t)(i & 0xFFFFFF);
  std::vector<P> pb(n), pc(n);
  for (size_t i = 0; i < n; i++) pb[i] = P{(int)i, (unsigned long)(rnd() % 5000)};
  for (size_t i = 0; i < n; i++) pc[i] = P{(int)i, (unsigned long)(rnd() & 0xFFFFFFFF)};
  const int reps = argc > 3 ? atoi(argv[3]) : 3;
  for (int rep = 0; rep < reps; rep++)
    for (int64_t pm : {(int64_t)-1, (int64_t)0, (int64_t)(1 << 18), (int64_t)(1 << 19), (int64_t)(1 << 20)}) {
      if (reps > 3 && pm != 0) continue;  // (many repetitions: the default threshold only)
      std::vector<uint32_t> a = base;
      double t0 = now();
      reorder::detail::GrayIntroSort(a.begin(), a.end(), by_degree, 0, 0, pm);
      double t1 = now();
      std::vector<P> b = pb;
      double t2 = now();
      reorder::detail::GrayIntroSort(b.begin(), b.end(), asc, 0, 0, pm);
      double t3 = now();
      std::vector<P> c = pc;
      double t4 = now();
      reorder::detail::GrayIntroSort(c.begin(), c.end(), asc, 0, 0, pm);
      double t5 = now();
      printf("par_min %8ld: degree keys %.1f ms   pairs(5000 keys) %.1f ms   pairs(distinct) %.1f ms\n", (long)pm, (t1 - t0) * 1e3, (t3 - t2) * 1e3, (t5 - t4) * 1e3);
    }
  std::vector<uint32_t> a = base;
  double t0 = now();
  std::sort(a.begin(), a.end(), by_degree);
  printf("std::sort degree keys %.1f ms\n", (now() - t0) * 1e3);
}
