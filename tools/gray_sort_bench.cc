// Host micro-benchmark (no GPU): std::sort against detail::GrayIntroSort on the three shapes the exact Gray mode sorts —
// 3.4 M (degree << 24 | row) words with a dozen distinct keys, a section of 1 M (row, key) pairs, 4 M pairs with distinct
// keys — by thread count.  Build and run on the GPU box (its 256 cores):
//   g++ -O2 -std=c++17 -pthread -Iinclude -Isparsebase_amd/host/include tools/gray_sort_bench.cc -o /tmp/gsb \
//       -Lsparsebase_amd/lib -lsbx -Wl,-rpath,$PWD/sparsebase_amd/lib && /tmp/gsb
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <sys/resource.h>
#include <vector>
#include "sparsebase/sparsebase.h"
using namespace sparsebase;
using clk = std::chrono::steady_clock;
static double ms(clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); }
int main(int argc, char **argv) {
  unsigned long long state = 88172645463325252ull;
  if (argc > 1 && !strcmp(argv[1], "--starve")) {
    // no new threads for this user (RLIMIT_NPROC = 1; not enforced for root): every std::thread the sort asks for fails
    // with EAGAIN and its share must fall to the calling thread — same permutation, no std::terminate
    struct rlimit rl = {1, 1};
    if (setrlimit(RLIMIT_NPROC, &rl) != 0) std::perror("setrlimit");
    std::vector<uint32_t> a(2000000), e;
    unsigned long long st = 12345;
    for (auto &x : a) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; x = (uint32_t)((st % 11) << 24) | (uint32_t)(st >> 40); }
    e = a;
    auto cmp = [](uint32_t l, uint32_t r) -> bool { return (l >> 24) < (r >> 24); };
    std::sort(e.begin(), e.end(), cmp);
    bool spawned = true;
    try { std::thread t([] {}); t.join(); } catch (const std::system_error &) { spawned = false; }
    reorder::detail::GrayIntroSort(a.begin(), a.end(), cmp, 8, 1000);
    int64_t sum = 0;
    reorder::detail::GrayParallelFor((int64_t)a.size(), [&](int64_t b0, int64_t b1) { int64_t s = 0; for (int64_t i = b0; i < b1; i++) s += a[(size_t)i] >> 24; __atomic_fetch_add(&sum, s, __ATOMIC_RELAXED); });
    std::printf("starved of threads (a probe thread %s): sort %s, parallel-for sum %lld\n", spawned ? "COULD still be created" : "could not be created",
                a == e ? "identical to std::sort" : "DIFFERS", (long long)sum);
    return a == e ? 0 : 1;
  }
  auto rnd = [&state]() { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return state; };
  auto by_degree = [](uint32_t l, uint32_t r) -> bool { return (l >> 24) < (r >> 24); };
  typedef std::pair<int, unsigned long> row_key;
  auto asc = [](const row_key &l, const row_key &r) -> bool { return l.second < r.second; };
  std::vector<uint32_t> words(3400000);
  for (size_t i = 0; i < words.size(); i++) {
    const double u = (rnd() >> 11) * (1.0 / 9007199254740992.0);
    unsigned d = (unsigned)(1.0 / (u + 0.09));  // 1 .. 11, most rows at the small end
    words[i] = (d << 24) | (uint32_t)i;
  }
  std::vector<row_key> section(1000000), dense(4000000);
  for (size_t i = 0; i < section.size(); i++) section[i] = row_key((int)i, (unsigned long)(rnd() % 4096));
  for (size_t i = 0; i < dense.size(); i++) dense[i] = row_key((int)i, (unsigned long)rnd());
  auto run = [&](const char *name, auto &data, auto comp) {
    auto ref = data;
    auto t0 = clk::now();
    std::sort(ref.begin(), ref.end(), comp);
    std::printf("%-28s std::sort %7.1f ms |", name, ms(t0));
    for (unsigned th : {2u, 4u, 8u, 16u, 32u}) {
      auto w = data;
      t0 = clk::now();
      reorder::detail::GrayIntroSort(w.begin(), w.end(), comp, th);
      std::printf(" %u threads %6.1f%s", th, ms(t0), w == ref ? "" : " DIFFERS");
    }
    std::printf("\n");
  };
  for (int rep = 0; rep < 2; rep++) {
    run("3.4 M words, 11 keys", words, by_degree);
    run("1 M pairs, 4096 keys", section, asc);
    run("4 M pairs, distinct keys", dense, asc);
  }
  return 0;
}
