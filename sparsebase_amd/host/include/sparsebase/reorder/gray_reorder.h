// GrayReorder (reference: reorder/gray_reorder.h:13-66, gray_reorder.cc:106-424).
//
// Two stages:
//   device  sbx_gray_row_keys — one pass over the nonzeros: per-row degree,
//           Gray-decoded block-occupancy key and the four band counters
//           (everything in the reference that is O(nnz));
//   host    the ordering of the n row keys.  The reference orders rows with unstable
//           std::sort calls on heavily tied keys (gray_reorder.cc:199,294-299,355-358,
//           404); its result therefore depends on libstdc++'s introsort visiting
//           order, which no other sort reproduces.  To stay bit-exact this stage
//           issues the same std::sort calls, on the same sequences, over the
//           device-computed keys (O(n log n) on rows, nothing touches the nonzeros);
//           calls that do not depend on each other (the sections, the dense rows) run on
//           threads of their own, and a big call runs as that very introsort with the two
//           sides of its partitions on different threads (detail::GrayIntroSort: the
//           library's own routines on disjoint ranges, hence the same permutation).
#ifndef SPARSEBASE_REORDER_GRAY_REORDER_H_
#define SPARSEBASE_REORDER_GRAY_REORDER_H_
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <exception>
#include <memory>
#include <mutex>
#include <system_error>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

#include "sparsebase/reorder/reorderer.h"
#include "sparsebase/utils/logger.h"
#if defined(__linux__)
#include <sched.h>
#include <cstdio>
#endif

namespace sparsebase::reorder {

enum BitMapSize { BitSize16 = 16, BitSize32 = 32, BitSize64 = 64 };

namespace detail {
// The stage's scratch — a dozen arrays of 12 - 70 MB per call on the 4 M-row matrices — comes from a process-wide pool of
// blocks it has used before: fresh memory costs a page fault per 4 KB at its first touch, ~150 MB of them per call, taken
// by two dozen threads at once (and given back to the system by free() behind every call).  Blocks of 1 MB and more,
// sizes rounded up to 2 MB, at most SBX_GRAY_SCRATCH_MB (default 512; 0: plain new / delete) kept between calls.
class GrayScratchPool {
 public:
  static GrayScratchPool &Get() {
    static GrayScratchPool *pool = new GrayScratchPool;  // (never destroyed: worker threads may outlive static destruction)
    return *pool;
  }
  void *Acquire(size_t bytes, size_t *capacity) {
    if (bytes < kMin || limit_ == 0) {
      *capacity = 0;  // (not the pool's)
      return ::operator new(bytes ? bytes : 1);
    }
    const size_t cap = (bytes + kRound - 1) / kRound * kRound;
    {
      std::lock_guard<std::mutex> g(mu_);
      for (size_t i = free_.size(); i-- > 0;)
        if (free_[i].second == cap) {
          void *p = free_[i].first;
          free_.erase(free_.begin() + (std::ptrdiff_t)i);
          held_ -= cap;
          *capacity = cap;
          return p;
        }
    }
    *capacity = cap;
    return ::operator new(cap);
  }
  void Release(void *p, size_t capacity) {
    if (!p) return;
    if (capacity) {
      std::lock_guard<std::mutex> g(mu_);
      if (held_ + capacity <= limit_) {
        free_.emplace_back(p, capacity);
        held_ += capacity;
        return;
      }
    }
    ::operator delete(p);
  }

 private:
  GrayScratchPool() {
    const char *e = std::getenv("SBX_GRAY_SCRATCH_MB");
    limit_ = (size_t)(e ? std::atoll(e) : 512) << 20;
  }
  static constexpr size_t kMin = (size_t)1 << 20, kRound = (size_t)2 << 20;
  std::mutex mu_;
  std::vector<std::pair<void *, size_t>> free_;
  size_t held_ = 0, limit_ = 0;
};

// n elements left uninitialised: the stage writes every element it later reads, and a std::vector would zero-fill
// 14 - 50 MB a piece on one thread first (eight of them: ~20 ms of the call on the 4 M-row matrices).
template <typename T>
class GrayBuffer {
 public:
  GrayBuffer() = default;
  explicit GrayBuffer(size_t n) { reset(n); }
  ~GrayBuffer() { GrayScratchPool::Get().Release(p_, cap_); }
  GrayBuffer(const GrayBuffer &) = delete;
  GrayBuffer &operator=(const GrayBuffer &) = delete;
  void reset(size_t n) {
    static_assert(std::is_trivially_default_constructible<T>::value && std::is_trivially_destructible<T>::value,
                  "raw storage: the elements are neither constructed nor destroyed");
    GrayScratchPool::Get().Release(p_, cap_);
    p_ = nullptr, cap_ = 0, n_ = 0;
    if (n) {
      p_ = static_cast<T *>(GrayScratchPool::Get().Acquire(n * sizeof(T), &cap_));
      n_ = n;
    }
  }
  size_t size() const { return n_; }
  bool empty() const { return n_ == 0; }
  T *data() { return p_; }
  T *begin() { return p_; }
  T *end() { return p_ + n_; }
  T &operator[](size_t i) { return p_[i]; }
  const T &operator[](size_t i) const { return p_[i]; }

 private:
  T *p_ = nullptr;
  size_t n_ = 0, cap_ = 0;
};

// While one of these lives, the calling thread — and every thread created under it, which inherit its mask and are placed
// inside it from the start — stays on the NUMA node it is running on; the destructor gives the thread its mask back.
// The ordering stage's threads share their arrays, barrier words and lists: on the GPU box's host (2 sockets x 64 cores)
// a sort of 3 M records takes 9.4 ms spread over both sockets and 7.2 ms inside one (taskset), and one call in three or four
// was an outlier of +7 ... +18 ms.  (Setting the mask of threads AFTER their creation was tried first and was much worse:
// 32 threads restricted one by one start out stacked on a few CPUs and their spin barriers fight for them — 83 ms for a
// 9 ms sort.)  Linux only; SBX_GRAY_SORT_PIN=0, a single node, a mask that is already inside one node, or any call that
// fails: nothing happens.
class GrayNodeScope {
 public:
  GrayNodeScope() {
#if defined(__linux__)
    const char *e = std::getenv("SBX_GRAY_SORT_PIN");
    if (e && std::atoi(e) == 0) return;
    cpu_set_t cur;
    CPU_ZERO(&cur);
    if (sched_getaffinity(0, sizeof(cur), &cur) != 0) return;
    const int cpu = sched_getcpu();
    if (cpu < 0) return;
    for (int node = 0; node < 64; node++) {
      char path[96];
      std::snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
      FILE *f = std::fopen(path, "r");
      if (!f) break;
      char buf[4096];
      const size_t got = std::fread(buf, 1, sizeof buf - 1, f);
      std::fclose(f);
      buf[got] = 0;
      cpu_set_t m;
      CPU_ZERO(&m);
      bool mine = false;
      for (const char *p = buf; *p;) {  // "0-63,128-191"
        char *end = nullptr;
        const long a = std::strtol(p, &end, 10);
        if (end == p) break;
        long b = a;
        p = end;
        if (*p == '-') {
          b = std::strtol(p + 1, &end, 10);
          p = end;
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++)
          if (c >= 0) {
            mine |= c == cpu;
            if (CPU_ISSET((int)c, &cur)) CPU_SET((int)c, &m);
          }
        while (*p == ',' || *p == '\n' || *p == ' ') p++;
      }
      if (!mine) continue;
      // (a sliver of a node is worse than the whole machine; a mask that already lies inside the node needs nothing)
      if (CPU_COUNT(&m) < 8 || CPU_EQUAL(&m, &cur)) return;
      if (sched_setaffinity(0, sizeof(m), &m) == 0) saved_ = cur, restore_ = true;
      return;
    }
#endif
  }
  ~GrayNodeScope() {
#if defined(__linux__)
    if (restore_) (void)sched_setaffinity(0, sizeof(saved_), &saved_);
#endif
  }
  GrayNodeScope(const GrayNodeScope &) = delete;
  GrayNodeScope &operator=(const GrayNodeScope &) = delete;

 private:
#if defined(__linux__)
  cpu_set_t saved_;
#endif
  bool restore_ = false;
};

// job(w) for w in [0, workers): worker 0 is the calling thread; a thread the system refuses to create (resource limits
// in a container) leaves its job to the caller instead of ending the process through a joinable thread's destructor
template <typename Job>
inline void GrayRunWorkers(unsigned workers, Job job) {
  std::vector<std::thread> pool;
  std::vector<unsigned> inline_jobs;
  pool.reserve(workers);
  for (unsigned w = 1; w < workers; w++) {
    try {
      pool.emplace_back([&job, w]() { job(w); });
    } catch (const std::system_error &) {
      inline_jobs.push_back(w);
    }
  }
  job(0u);
  for (unsigned w : inline_jobs) job(w);
  for (auto &t : pool) t.join();
}

// f(begin, end) over [0, count) in contiguous pieces on up to 16 threads (element-wise loops of the Gray host stage:
// they touch several million rows each and are bound by memory latency, not by the sorts' order)
// grain: the fewest items worth a thread of their own (65 536 rows; 1 where an item is itself a piece of that size)
template <typename F>
inline void GrayParallelFor(int64_t count, F f, int64_t grain = (int64_t)1 << 16) {
  const unsigned hw = std::thread::hardware_concurrency();
  const int64_t workers = std::min<int64_t>(std::min<unsigned>(hw ? hw : 1u, 16u), (count + grain - 1) / grain);
  if (workers <= 1) {
    f((int64_t)0, count);
    return;
  }
  std::vector<std::exception_ptr> errors((size_t)workers);
  GrayRunWorkers((unsigned)workers, [&](unsigned w) {
    try {
      f(count * (int64_t)w / workers, count * ((int64_t)w + 1) / workers);
    } catch (...) {
      errors[(size_t)w] = std::current_exception();
    }
  });
  for (auto &e : errors)
    if (e) std::rethrow_exception(e);
}

// std::sort(first, last, comp) as libstdc++ runs it, with the two sides of its partitions on different threads.
// The reference's orderings are whatever libstdc++'s introsort leaves on heavily tied keys (gray_reorder.cc:199-203 sorts
// 4 M row ids by degree alone), so the exact mode cannot use another algorithm — but introsort's recursion works on
// disjoint ranges: std::__introsort_loop partitions [first, last) around a pivot, recurses into the right part and loops
// on the left one.  Here that loop is the library's own code path by path (std::__unguarded_partition_pivot,
// std::__partial_sort at the depth limit, std::__introsort_loop itself below the grain), only the recursive call goes
// to a pool; the closing insertion sort (std::__final_insertion_sort) runs piecewise between the tasks' cuts.  Same comparisons on the same elements in every range,
// hence the same permutation: compared with std::sort on tied, sorted, reversed and random inputs by
// host/tests/test_host_logic.cc (GraySort.ParallelReplicaOfStdSort).
// The routines called are libstdc++ INTERNALS (std::__introsort_loop, std::__unguarded_partition_pivot,
// std::__partial_sort, std::__insertion_sort, __gnu_cxx::__ops::__iter_comp_iter, _S_threshold = 16), so the replica is
// compiled only inside the window of releases whose <bits/stl_algo.h> was read and tested (GCC 9 - 14: the introsort is
// unchanged there; developed and run on 11.4) — any other standard library or release gets plain std::sort — and the
// first big sort of a process checks the replica against std::sort on 200 K heavily tied keys (GraySortReplicaAgrees):
// a library inside the window that nevertheless sorts differently (another threshold or pivot rule) is noticed, logged,
// and the process falls back to plain std::sort (slower, still exact).  -DSBX_GRAY_NO_SORT_REPLICA forces that.
#if defined(__GLIBCXX__) && defined(_GLIBCXX_RELEASE) && !defined(SBX_GRAY_NO_SORT_REPLICA)
#if _GLIBCXX_RELEASE >= 9 && _GLIBCXX_RELEASE <= 14
#define SBX_GRAY_SORT_REPLICA 1
#endif
#endif

// threads one sort asks for at most (SBX_GRAY_SORT_THREADS: 1 .. 64; default: 32 on hosts of 64 hardware threads and
// more, else 16 — with the team's partitions a sort of 3 M records still gains from 16 -> 32 threads: 12.5 -> 9.3 ms for
// 16-byte records on the 256-thread host of the GPU box; without them nothing beyond 16 did)
inline unsigned GraySortThreadCap() {
  static const unsigned cap = [] {
    const char *e = std::getenv("SBX_GRAY_SORT_THREADS");
    const long v = e ? std::atol(e) : (std::thread::hardware_concurrency() >= 64 ? 32 : 16);
    return (unsigned)(v < 1 ? 1 : (v > 64 ? 64 : v));
  }();
  return cap;
}

// One budget of sort threads per process: the dense rows' sort, the degree sort and up to 16 section workers may each
// ask for a parallel sort at the same time; together they get at most GrayThreadBudget::kMax extra threads (a lease
// that finds the budget spent sorts on its caller's thread).
class GrayThreadBudget {
 public:
  static constexpr int kMax = 64;
  explicit GrayThreadBudget(unsigned want) {
    const unsigned hw = std::thread::hardware_concurrency();
    const int cap = (int)std::min<unsigned>(hw ? hw : 1u, (unsigned)kMax);
    int cur = in_use().load();
    for (;;) {
      const int room = std::max(0, cap - cur);
      got_ = std::min<int>((int)want - 1, room);  // (the caller's own thread is not counted)
      if (got_ <= 0) {
        got_ = 0;
        break;
      }
      if (in_use().compare_exchange_weak(cur, cur + got_)) break;
    }
  }
  ~GrayThreadBudget() {
    if (got_) in_use() -= got_;
  }
  GrayThreadBudget(const GrayThreadBudget &) = delete;
  GrayThreadBudget &operator=(const GrayThreadBudget &) = delete;
  unsigned threads() const { return (unsigned)got_ + 1u; }

 private:
  static std::atomic<int> &in_use() {
    static std::atomic<int> v{0};
    return v;
  }
  int got_ = 0;
};

#if defined(SBX_GRAY_SORT_REPLICA)
// The partitions of the BIG ranges — the first of them walks the whole array on one thread, the next two a half each:
// a third of a sort of 4 M heavily tied keys — are run by all the threads together, to the sequential routine's result:
// std::__unguarded_partition(first, last, pivot) moves a left pointer up to the next element that is not less than the
// pivot, a right pointer down to the next one that is not greater, swaps the two and goes on until the pointers meet.
// Between two swaps the pointers only see elements no swap has touched, so the k-th stop of the left pointer is the k-th
// position (from the left) whose ORIGINAL element is not less than the pivot, L[k], the k-th stop of the right pointer
// the k-th such position from the right for "not greater", R[k]; the swaps are exactly the pairs (L[k], R[k]) with
// L[k] < R[k], k = 0 .. K - 1 (L rises and R falls: a prefix), all 2 K positions distinct; and the routine returns where
// the left pointer stops next: L[K] if that lies in front of R[K - 1], else R[K - 1] (which now holds an element that is
// not less than the pivot).  The threads list both kinds of positions for a stretch of the range each, K is found by a
// binary search over k, and the swaps are shared out.  (Elements equal to the pivot stop both pointers, as they do in
// the library: with eleven distinct keys nearly every position is in both lists.)
template <typename It, typename WrappedCompare>
class GrayIntroSortPool {
 public:
  static constexpr unsigned kMaxTeam = 64;  // (GraySortThreadCap's ceiling)
  // par_min: ranges of at least this many elements are partitioned by the whole team (0: never)
  GrayIntroSortPool(WrappedCompare comp, int64_t grain, int64_t par_min = 0)
      : comp_(comp), grain_(grain), par_min_(par_min > 0 ? std::max<int64_t>(par_min, 64) : 0) {}
  void Run(It first, It last, unsigned threads) {
    const long depth0 = (long)std::__lg(last - first) * 2;
    if (threads > kMaxTeam) threads = kMaxTeam;
    if (par_min_ > 0 && last - first >= par_min_ && (uint64_t)(last - first) < ((uint64_t)1 << 32) && threads > 1) {
      // the team: threads that could not be created are simply not part of it (a barrier counts on every member)
      big_.push_back(Task{first, last, depth0});
      whole_last_ = last;
      std::vector<std::thread> team;
      std::atomic<int> gate{0};  // 0: wait; > 0: the team's size; < 0: leave
      unsigned made = 0;
      for (unsigned w = 1; w < threads; w++) {
        try {
          team.emplace_back([this, &gate, w]() {
            int size;
            while ((size = gate.load(std::memory_order_acquire)) == 0) std::this_thread::yield();
            if (size < 0 || (int)w >= size) return;
            TeamMain(w, (unsigned)size);
          });
          made++;
        } catch (const std::system_error &) {
          break;  // (members are numbered 1 .. made: the ones created so far)
        }
      }
      team_.reset(new TeamState(made + 1));
      gate.store((int)made + 1, std::memory_order_release);
      TeamMain(0u, made + 1);
      for (auto &t : team) t.join();
    } else {
      Push(first, last, depth0);
      GrayRunWorkers(threads, [this](unsigned) { Work(); });
    }
    if (error_) std::rethrow_exception(error_);
    // std::__final_insertion_sort: one stable insertion sort over everything.  No element crosses a partition's cut (what
    // lies left of it does not compare greater than what lies right of it), so the pieces between the cuts the tasks were
    // split at are insertion-sorted on their own, concurrently, to the same result.
    if (insertion_done_) return;  // (the team went on to it itself: one set of threads per sort, not two)
    cuts_.push_back(last);
    std::sort(cuts_.begin(), cuts_.end());
    ins_next_ = 0;
    GrayRunWorkers(threads, [this](unsigned) { InsertionPieces(); });
    if (error_) std::rethrow_exception(error_);
  }

 private:
  struct Task {
    It first, last;
    long depth_limit;
  };
  void InsertionPieces() {
    try {
      for (size_t k = ins_next_++; k + 1 < cuts_.size(); k = ins_next_++) std::__insertion_sort(cuts_[k], cuts_[k + 1], comp_);
    } catch (...) {
      std::lock_guard<std::mutex> g(mu_);
      if (!error_) error_ = std::current_exception();
    }
  }
  void Push(It first, It last, long depth_limit) {
    pending_++;
    {
      std::lock_guard<std::mutex> g(mu_);
      queue_.push_back(Task{first, last, depth_limit});
      cuts_.push_back(first);
    }
    cv_.notify_one();
  }
  void Loop(It first, It last, long depth_limit) {  // std::__introsort_loop, its recursive call handed to the pool
    while (last - first > 16) {                     // (_S_threshold)
      if (last - first <= grain_) {
        std::__introsort_loop(first, last, depth_limit, comp_);
        return;
      }
      if (depth_limit == 0) {
        std::__partial_sort(first, last, last, comp_);
        return;
      }
      --depth_limit;
      It cut = std::__unguarded_partition_pivot(first, last, comp_);
      Push(cut, last, depth_limit);
      last = cut;
    }
  }
  // ---- the team's part: big ranges, one at a time, every partition by all members ----------------------------------
  struct TeamState {
    // a member's lists: offsets (from job_first) whose element stops the left / the right pointer.  The lists themselves
    // are the members' own (locals of TeamMain: vectors whose headers shared cache lines made every push_back a
    // coherence miss — 7 x slower than the sequential partition); here only where they are and how long
    struct alignas(64) Lists {
      const uint32_t *l = nullptr, *r = nullptr;
      size_t nl = 0, nr = 0;
    };
    explicit TeamState(unsigned size_) : size(size_), lists(size_) {}
    const unsigned size;
    alignas(64) std::atomic<unsigned> arrived{0};
    alignas(64) std::atomic<unsigned> phase{0};
    std::vector<Lists> lists;
    It job_first{}, job_last{}, pivot{};      // the range being partitioned: [job_first, job_last), pivot outside it
    bool done = false;
    // sense-reversing; spins briefly, then yields (a team of at most 32 on a busy host).  (A waiter that sleeps on a
    // condition variable instead — tried against the occasional sort that takes 34 ms for 9 — made every barrier a
    // thundering herd on the mutex: 13 -> 46 ms for the 3 M-key sort on 8 cores.)
    void Barrier() {
      const unsigned my = phase.load(std::memory_order_acquire);
      if (arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == size) {
        arrived.store(0, std::memory_order_relaxed);
        phase.store(my + 1, std::memory_order_release);
        return;
      }
      // (pause, not a bare load loop: the waiter's hyperthread sibling may be the member everyone waits for; and a long
      // stretch of pauses before the first sched_yield: thirty waiters yielding in a loop are thirty system calls
      // fighting over the run queues' locks)
      for (unsigned spin = 0; phase.load(std::memory_order_acquire) == my; spin++) {
#if defined(__x86_64__) || defined(__i386__)
        if (spin < 4000) __builtin_ia32_pause();
        else
#endif
          if (spin > 200) std::this_thread::yield();
      }
    }
  };
  // member 0, between two barriers: the next big range, its pivot; false: none left
  bool NextJob() {
    TeamState &t = *team_;
    while (!big_.empty()) {
      Task b = big_.back();
      big_.pop_back();
      if (b.last - b.first < par_min_) {  // (std::__introsort_loop's own test — above 16 elements — is the task's)
        Push(b.first, b.last, b.depth_limit);
        continue;
      }
      if (b.depth_limit == 0) {
        std::__partial_sort(b.first, b.last, b.last, comp_);
        continue;
      }
      cur_ = b;
      cur_.depth_limit--;
      // std::__unguarded_partition_pivot
      It mid = b.first + (b.last - b.first) / 2;
      std::__move_median_to_first(b.first, b.first + 1, mid, b.last - 1, comp_);
      t.job_first = b.first + 1, t.job_last = b.last, t.pivot = b.first;
      return true;
    }
    return false;
  }
  void TeamMain(unsigned w, unsigned size) {
    TeamState &t = *team_;
    GrayBuffer<uint32_t> l, r;  // (not zero-filled: every entry that is read was written by the scan below)
    for (;;) {
      if (w == 0) {
        try {
          t.done = !NextJob();
        } catch (...) {
          t.done = true;
          std::lock_guard<std::mutex> g(mu_);
          if (!error_) error_ = std::current_exception();
        }
      }
      t.Barrier();
      if (t.done) break;
      // 1: the positions of this member's stretch that stop the left / the right pointer
      const int64_t count = t.job_last - t.job_first;
      const int64_t c0 = count * (int64_t)w / size, c1 = count * ((int64_t)w + 1) / size;
      if (l.size() < (size_t)(c1 - c0) + 1) l.reset((size_t)(c1 - c0) + 1), r.reset((size_t)(c1 - c0) + 1);
      size_t nl = 0, nr = 0;
      try {
        uint32_t *lp = l.data(), *rp = r.data();
        for (int64_t i = c0; i < c1; i++) {  // (every position is stored, the counters move for the ones that stop a pointer)
          It it = t.job_first + i;
          lp[nl] = (uint32_t)i;
          nl += comp_(it, t.pivot) ? 0u : 1u;
          rp[nr] = (uint32_t)i;
          nr += comp_(t.pivot, it) ? 0u : 1u;
        }
      } catch (...) {
        std::lock_guard<std::mutex> g(mu_);
        if (!error_) error_ = std::current_exception();
      }
      t.lists[w].l = l.data(), t.lists[w].nl = nl, t.lists[w].r = r.data(), t.lists[w].nr = nr;
      t.Barrier();
      // 2: how many swaps — every member for itself, from the lists' lengths (a barrier less per partition than with
      // member 0 computing it for all: the lists are final, the elements are not looked at)
      uint64_t pl[kMaxTeam + 1], pr[kMaxTeam + 1];
      pl[0] = pr[0] = 0;
      for (unsigned m = 0; m < size; m++) pl[m + 1] = pl[m] + t.lists[m].nl, pr[m + 1] = pr[m] + t.lists[m].nr;
      auto SelL = [&](uint64_t k) -> uint32_t {  // L[k]
        const size_t m = (size_t)(std::upper_bound(pl, pl + size + 1, k) - pl) - 1;
        return t.lists[m].l[(size_t)(k - pl[m])];
      };
      auto SelR = [&](uint64_t k) -> uint32_t {  // R[k]: the k-th from the right
        const uint64_t g = pr[size] - 1 - k;
        const size_t m = (size_t)(std::upper_bound(pr, pr + size + 1, g) - pr) - 1;
        return t.lists[m].r[(size_t)(g - pr[m])];
      };
      uint64_t K;
      {
        uint64_t lo = 0, hi = std::min(pl[size], pr[size]);  // K in [lo, hi]: L[k] < R[k] for k < K
        while (lo < hi) {
          const uint64_t mid = (lo + hi) / 2;
          if (SelL(mid) < SelR(mid)) lo = mid + 1;
          else hi = mid;
        }
        K = lo;
      }
      // 3: the swaps, shared out
      {
        const uint64_t k0 = K * w / size, k1 = K * ((uint64_t)w + 1) / size;
        if (k0 < k1) {
          // cursors instead of a search per swap: L[k] walks its lists upwards, R[k] its lists downwards
          size_t lw = (size_t)(std::upper_bound(pl, pl + size + 1, k0) - pl) - 1;
          size_t li = (size_t)(k0 - pl[lw]);
          const uint64_t g0 = pr[size] - 1 - k0;
          size_t rw = (size_t)(std::upper_bound(pr, pr + size + 1, g0) - pr) - 1;
          size_t ri = (size_t)(g0 - pr[rw]);
          for (uint64_t k = k0; k < k1; k++) {
            while (li >= t.lists[lw].nl) lw++, li = 0;
            std::iter_swap(t.job_first + t.lists[lw].l[li], t.job_first + t.lists[rw].r[ri]);
            li++;
            if (k + 1 < k1) {
              while (ri == 0) {
                rw--;
                ri = t.lists[rw].nr;
              }
              ri--;
            }
          }
        }
      }
      t.Barrier();
      if (w == 0) {
        // where the left pointer stops next (the median-of-three pivot guarantees both lists are non-empty)
        uint32_t at;
        if (K < pl[size] && (K == 0 || SelL(K) < SelR(K - 1))) at = SelL(K);
        else at = SelR(K - 1);
        It cut = t.job_first + at;
        {
          std::lock_guard<std::mutex> g(mu_);
          cuts_.push_back(cut);
        }
        // std::__introsort_loop: the right part is the recursive call's, the left part the loop's next round
        big_.push_back(Task{cut, cur_.last, cur_.depth_limit});
        big_.push_back(Task{cur_.first, cut, cur_.depth_limit});
      }
    }
    // the ranges below par_min are in the queue: the members become the pool's workers
    if (w == 0 && pending_.load() == 0) {
      std::lock_guard<std::mutex> g(mu_);
      done_ = true;
      cv_.notify_all();
    }
    Work();
    // ... and go on to the closing insertion sort (Run): the pieces between the cuts, shared out
    t.Barrier();
    if (w == 0) {
      cuts_.push_back(whole_last_);
      std::sort(cuts_.begin(), cuts_.end());
      ins_next_ = 0;
      insertion_done_ = true;
    }
    t.Barrier();
    InsertionPieces();
  }

  void Work() {
    for (;;) {
      Task t;
      {
        std::unique_lock<std::mutex> g(mu_);
        cv_.wait(g, [this]() { return done_ || !queue_.empty(); });
        if (queue_.empty()) return;
        t = queue_.back();
        queue_.pop_back();
      }
      try {
        Loop(t.first, t.last, t.depth_limit);
      } catch (...) {
        std::lock_guard<std::mutex> g(mu_);
        if (!error_) error_ = std::current_exception();
      }
      if (--pending_ == 0) {
        std::lock_guard<std::mutex> g(mu_);
        done_ = true;
        cv_.notify_all();
      }
    }
  }
  WrappedCompare comp_;
  const int64_t grain_, par_min_;
  std::unique_ptr<TeamState> team_;
  std::vector<Task> big_;  // member 0's: ranges still to be partitioned by the team
  Task cur_{};
  It whole_last_{};
  std::atomic<size_t> ins_next_{0};
  bool insertion_done_ = false;
  std::atomic<int64_t> pending_{0};
  std::mutex mu_;
  std::condition_variable cv_;
  std::vector<Task> queue_;
  std::vector<It> cuts_;  // where a range was split between two tasks (and the start of the whole)
  bool done_ = false;
  std::exception_ptr error_;
};

// once per process: the replica against std::sort on 200 000 words with eleven distinct keys (the degree sort's shape)
// and on (id, key) records in descending order of a tied key, four threads, a grain that forces ~50 tasks
inline bool GraySortReplicaAgrees() {
  static const bool ok = [] {
    try {
      const size_t count = 200000;
      std::vector<uint32_t> a(count), b;
      uint64_t x = 0x9E3779B97F4A7C15ull;
      for (size_t i = 0; i < count; i++) {
        x = x * 6364136223846793005ull + 1442695040888963407ull;
        a[i] = ((uint32_t)((x >> 40) % 11u) << 24) | (uint32_t)i;
      }
      b = a;
      auto by_key = [](uint32_t l, uint32_t r) -> bool { return (l >> 24) < (r >> 24); };
      std::sort(a.begin(), a.end(), by_key);
      {
        auto w = __gnu_cxx::__ops::__iter_comp_iter(by_key);
        GrayIntroSortPool<std::vector<uint32_t>::iterator, decltype(w)> pool(w, 4096, 20000);
        pool.Run(b.begin(), b.end(), 4);
      }
      if (a != b) return false;
      struct Rec { uint32_t first; unsigned long second; };
      std::vector<Rec> c(count), d;
      for (size_t i = 0; i < count; i++) {
        x = x * 6364136223846793005ull + 1442695040888963407ull;
        c[i] = Rec{(uint32_t)i, (unsigned long)((x >> 33) % 977u)};
      }
      d = c;
      auto desc = [](const Rec &l, const Rec &r) -> bool { return l.second > r.second; };
      std::sort(c.begin(), c.end(), desc);
      {
        auto w = __gnu_cxx::__ops::__iter_comp_iter(desc);
        GrayIntroSortPool<typename std::vector<Rec>::iterator, decltype(w)> pool(w, 4096, 20000);
        pool.Run(d.begin(), d.end(), 4);
      }
      for (size_t i = 0; i < count; i++)
        if (c[i].first != d[i].first) return false;
      return true;
    } catch (...) {
      return false;
    }
  }();
  static const bool logged = [] {
    if (!ok)
      utils::Logger(typeid(GrayThreadBudget))
          .Log("the parallel replica of std::sort does not reproduce this libstdc++'s std::sort: GrayReorder falls back "
               "to plain std::sort (exact, slower)", utils::LOG_LVL_WARNING);
    return true;
  }();
  (void)logged;
  return ok;
}
#endif

/// grain: ranges up to this many elements are sorted by the calling task (0: chosen from the size and the threads);
/// par_min: ranges of at least this many elements are partitioned by all threads together (0: 2^18; < 0: never)
template <typename It, typename Compare>
inline void GrayIntroSort(It first, It last, Compare comp, unsigned threads = 0, int64_t grain = 0, int64_t par_min = 0) {
#if defined(SBX_GRAY_SORT_REPLICA)
  const int64_t count = last - first;
  const bool budgeted = threads == 0;  // (an explicit thread count — the tests — is taken as given)
  if (threads == 0) {
    const unsigned hw = std::thread::hardware_concurrency();
    threads = std::min<unsigned>(hw ? hw : 1u, GraySortThreadCap());
  }
  if (threads <= 1 || count <= std::max<int64_t>(grain, budgeted ? ((int64_t)1 << 15) : 1) || !GraySortReplicaAgrees()) {
    std::sort(first, last, comp);
    return;
  }
  GrayThreadBudget lease(budgeted ? threads : 1u);
  if (budgeted) threads = lease.threads();
  if (grain <= 0) grain = std::max<int64_t>((int64_t)1 << 15, count / ((int64_t)threads * 8));
  if (threads <= 1 || count <= grain) {
    std::sort(first, last, comp);
    return;
  }
  auto wrapped = __gnu_cxx::__ops::__iter_comp_iter(comp);
  if (par_min == 0) par_min = (int64_t)1 << 18;
  GrayIntroSortPool<It, decltype(wrapped)> pool(wrapped, grain, par_min < 0 ? 0 : std::max<int64_t>(par_min, grain));
  pool.Run(first, last, threads);
#else
  (void)threads, (void)grain, (void)par_min;
  std::sort(first, last, comp);
#endif
}
}  // namespace detail

struct GrayReorderParams : utils::Parameters {
  BitMapSize resolution;
  int nnz_threshold;
  int sparse_density_group_size;
  // Opt-in, not in the reference: order the row keys on the device with STABLE sorts (sbx_gray_reorder) instead of
  // issuing the reference's std::sort calls on the host.  The result equals the reference's wherever its comparators
  // decide a row's place; tied rows come in stable order instead of libstdc++'s introsort order.  Default: exact.
  bool stable_device_ordering = false;
  explicit GrayReorderParams() : resolution(BitSize32), nnz_threshold(0), sparse_density_group_size(1) {}
  GrayReorderParams(BitMapSize r, int nnz_thresh, int group_size)
      : resolution(r), nnz_threshold(nnz_thresh), sparse_density_group_size(group_size) {}
};

template <typename IDType, typename NNZType, typename ValueType>
class GrayReorder : public Reorderer<IDType> {
  struct row_grey_pair {  // (the reference's std::pair<IDType, unsigned long>, without a constructor that zero-fills arrays)
    IDType first;
    unsigned long second;
  };

 public:
  typedef GrayReorderParams ParamsType;
  GrayReorder(BitMapSize resolution, int nnz_threshold, int sparse_density_group_size) {
    this->params_ = std::make_unique<GrayReorderParams>(resolution, nnz_threshold, sparse_density_group_size);
    this->RegisterFunction({format::CSR<IDType, NNZType, ValueType>::get_id_static()}, GrayReorderingCSR);
    this->RegisterFunction({format::HIPCSR<IDType, NNZType, ValueType>::get_id_static()}, GrayReorderingHIPCSR);
  }
  explicit GrayReorder(GrayReorderParams p)
      : GrayReorder(p.resolution, p.nnz_threshold, p.sparse_density_group_size) {
    static_cast<GrayReorderParams *>(this->params_.get())->stable_device_ordering = p.stable_device_ordering;
  }
  /// Wall time of the last call's stages in this process, in ms: the device key stage (sbx_gray_row_keys, which ends
  /// with a blocking read-back), the copy of degrees and keys to the host, the host ordering stage.  (Not in the
  /// reference: what bench.py reports as Gray end to end.)  [3..6]: the host stage's parts — sparse / dense split, the
  /// sort by degree, the sections (walk and sorts), joining the dense rows' sort and writing the order.
  static double *last_stage_ms() {
    static thread_local double ms[7] = {0, 0, 0, 0, 0, 0, 0};  // (per thread: concurrent reorder calls do not share it)
    return ms;
  }

 protected:
  static bool desc_comparator(const row_grey_pair &l, const row_grey_pair &r) { return l.second > r.second; }
  static bool asc_comparator(const row_grey_pair &l, const row_grey_pair &r) { return l.second < r.second; }

  static IDType *Run(detail::DeviceCsrView<IDType, NNZType, ValueType> v, utils::Parameters *poly) {
    auto *params = static_cast<GrayReorderParams *>(poly);
    const int64_t n = v.n;
    if (params->stable_device_ordering) {  // (opt-in: see GrayReorderParams)
      hip::Staged<IDType> d_inv(*v.dev, (size_t)(n ? n : 1));
      const int rc = sbx_gray_reorder(v.dev->handle(), hip::IndexTag<IDType, NNZType>(), v.n, v.m, v.nnz, v.row_ptr, v.col,
                                      (int)params->resolution, params->nnz_threshold, params->sparse_density_group_size,
                                      0, d_inv.get());
      IDType *inv = nullptr;
      if (rc == SBX_OK) inv = n ? v.dev->Download(d_inv.get(), (size_t)n) : new IDType[1]();
      v.Release();
      v.dev->Check(rc);
      return inv;
    }
    using clock = std::chrono::steady_clock;
    auto ms_since = [](clock::time_point t) { return std::chrono::duration<double, std::milli>(clock::now() - t).count(); };
    // ---- device stage
    hip::HostStaging<IDType> deg(*v.dev, (size_t)n);   // (page-locked and pooled: see hip/device.h)
    hip::HostStaging<uint64_t> key(*v.dev, (size_t)n);
    int64_t counts[4] = {0, 0, 0, 0};
    {
      hip::Staged<IDType> d_deg(*v.dev, (size_t)n);
      hip::Staged<uint64_t> d_key(*v.dev, (size_t)n);
      auto t0 = clock::now();
      const int rc = sbx_gray_row_keys(v.dev->handle(), hip::IndexTag<IDType, NNZType>(), v.n, v.m, v.nnz, v.row_ptr, v.col,
                                       (int)params->resolution, params->nnz_threshold, d_deg.get(), d_key.get(),
                                       counts);  // (returns after its last read-back: the stage is complete)
      last_stage_ms()[0] = ms_since(t0);
      t0 = clock::now();
      if (rc == SBX_OK && n > 0) {
        d_deg.ToHost(deg.data());
        // the keys are read by the sections' sorts (unless the sparse rows are "highly banded", gray_reorder.cc:223) and by
        // the dense rows' sort (unless those are, :369, or there are none: with a threshold >= 0 a dense row has entries):
        // a banded matrix that takes both early-outs never looks at them, and their copy is two thirds of the download
        const bool sparse_early_out = double((int)counts[1]) / (int)counts[0] > 0.3;
        const bool dense_early_out = double((int)counts[3]) / (int)counts[2] > 0.2;
        const bool no_dense_rows = counts[2] == 0 && params->nnz_threshold >= 0;
        if (!(sparse_early_out && (dense_early_out || no_dense_rows))) d_key.ToHost(key.data());
      }
      last_stage_ms()[1] = ms_since(t0);
      v.Release();
      v.dev->Check(rc);
    }
    const auto t_host = clock::now();
    struct HostStageTimer {  // (the ordering stage below returns from several places)
      clock::time_point t;
      ~HostStageTimer() { last_stage_ms()[2] = std::chrono::duration<double, std::milli>(clock::now() - t).count(); }
    } host_stage_timer{t_host};
    // ---- host ordering stage (see header comment)
    detail::GrayNodeScope on_one_node;  // (until the stage returns, by whichever way)
    const int group_size = params->sparse_density_group_size;
    // sparse / dense split in id order (gray_reorder.cc:138-170): counted per piece, then written at the pieces' offsets
    detail::GrayBuffer<IDType> sparse_rows, dense_rows;
    {
      const IDType thr = (IDType)params->nnz_threshold;
      const int64_t pieces = std::max<int64_t>(1, std::min<int64_t>(64, n >> 16));
      std::vector<int64_t> cnt((size_t)pieces + 1, 0);
      detail::GrayParallelFor(pieces, [&](int64_t p0, int64_t p1) {
        for (int64_t p = p0; p < p1; p++) {
          int64_t c = 0;
          for (int64_t i = n * p / pieces; i < n * (p + 1) / pieces; i++) c += deg[i] <= thr;
          cnt[(size_t)p + 1] = c;
        }
      }, 1);  // (an item is a piece of >= 65 536 rows)
      for (int64_t p = 0; p < pieces; p++) cnt[(size_t)p + 1] += cnt[(size_t)p];
      sparse_rows.reset((size_t)cnt[(size_t)pieces]);
      dense_rows.reset((size_t)(n - cnt[(size_t)pieces]));
      detail::GrayParallelFor(pieces, [&](int64_t p0, int64_t p1) {
        for (int64_t p = p0; p < p1; p++) {
          const int64_t b = n * p / pieces;
          int64_t s = cnt[(size_t)p], d = b - s;
          for (int64_t i = b; i < n * (p + 1) / pieces; i++) {
            if (deg[i] <= thr) sparse_rows[(size_t)s++] = (IDType)i;
            else dense_rows[(size_t)d++] = (IDType)i;
          }
        }
      }, 1);
    }
    last_stage_ms()[3] = ms_since(t_host);
    auto t_part = clock::now();
    // the reference keeps these counters in `int` (gray_reorder.cc:134-137)
    const bool sparse_banded = double((int)counts[1]) / (int)counts[0] > 0.3;
    const bool dense_banded = double((int)counts[3]) / (int)counts[2] > 0.2;
    // The std::sort calls below are the reference's, on the same sequences with comparators that answer the same — so
    // each leaves the permutation libstdc++'s introsort leaves in the reference — but they do not run one after the
    // other: the dense rows' sort (gray_reorder.cc:404) depends on nothing and runs beside everything else, and the
    // sections (:293-301,:354-360) are disjoint ranges whose comparator is fixed by their index, so once the walk
    // over the degree-sorted rows has found their bounds they are sorted concurrently.
    std::thread dense_thread;
    std::exception_ptr dense_error;
    bool dense_inline = false;  // (no thread to be had: the dense rows are sorted where the thread would be joined)
    auto sort_dense_rows = [&]() {
        try {
          detail::GrayBuffer<row_grey_pair> d(dense_rows.size());
          detail::GrayParallelFor((int64_t)d.size(), [&](int64_t a0, int64_t a1) {
            for (int64_t a = a0; a < a1; a++) d[(size_t)a] = row_grey_pair{dense_rows[(size_t)a], (unsigned long)key[dense_rows[(size_t)a]]};
          });
          detail::GrayIntroSort(d.begin(), d.end(), asc_comparator);
          detail::GrayParallelFor((int64_t)d.size(), [&](int64_t a0, int64_t a1) {
            for (int64_t a = a0; a < a1; a++) dense_rows[(size_t)a] = d[(size_t)a].first;
          });
        } catch (...) {
          dense_error = std::current_exception();
        }
    };
    if (!dense_banded && !dense_rows.empty()) {
      try {
        dense_thread = std::thread(sort_dense_rows);
      } catch (const std::system_error &) {
        dense_inline = true;
      }
    }
    struct JoinGuard {  // (the ordering stage may leave through an exception)
      std::thread &t;
      ~JoinGuard() { if (t.joinable()) t.join(); }
    } dense_guard{dense_thread};
    detail::GrayBuffer<IDType> sorted_deg;
    {
      // gray_reorder.cc:199-203: std::sort of the row ids by degree.  The (degree, id) pairs are sorted instead, with a
      // comparator that looks at the degree only: every comparison answers what `deg[a] < deg[b]` answers, so the
      // elements make the same moves, without two dependent loads per comparison.
      const int64_t ns0 = (int64_t)sparse_rows.size();
      sorted_deg.reset((size_t)ns0);  // (the section walk below reads the degrees in this order)
      if (params->nnz_threshold < 256 && n <= ((int64_t)1 << 24)) {
        // ... packed into ONE 32-bit word when they fit (degree << 24 | row): half the bytes to move, the same moves
        detail::GrayBuffer<uint32_t> byd((size_t)ns0);
        detail::GrayParallelFor(ns0, [&](int64_t a0, int64_t a1) {
          for (int64_t a = a0; a < a1; a++)
            byd[(size_t)a] = ((uint32_t)deg[sparse_rows[(size_t)a]] << 24) | (uint32_t)sparse_rows[(size_t)a];
        });
        detail::GrayIntroSort(byd.begin(), byd.end(), [](uint32_t l, uint32_t r) -> bool { return (l >> 24) < (r >> 24); });
        detail::GrayParallelFor(ns0, [&](int64_t a0, int64_t a1) {
          for (int64_t a = a0; a < a1; a++) {
            sparse_rows[(size_t)a] = (IDType)(byd[(size_t)a] & 0xFFFFFFu);
            sorted_deg[(size_t)a] = (IDType)(byd[(size_t)a] >> 24);
          }
        });
      } else {
        struct deg_row {  // (degree, row)
          IDType first, second;
        };
        detail::GrayBuffer<deg_row> byd((size_t)ns0);
        detail::GrayParallelFor(ns0, [&](int64_t a0, int64_t a1) {
          for (int64_t a = a0; a < a1; a++) byd[(size_t)a] = deg_row{deg[sparse_rows[(size_t)a]], sparse_rows[(size_t)a]};
        });
        detail::GrayIntroSort(byd.begin(), byd.end(), [](const deg_row &l, const deg_row &r) -> bool { return l.first < r.first; });
        detail::GrayParallelFor(ns0, [&](int64_t a0, int64_t a1) {
          for (int64_t a = a0; a < a1; a++) {
            sparse_rows[(size_t)a] = byd[(size_t)a].second;
            sorted_deg[(size_t)a] = byd[(size_t)a].first;
          }
        });
      }
    }
    last_stage_ms()[4] = ms_since(t_part);
    t_part = clock::now();

    if (!sparse_banded) {
      struct Section { int64_t start, end; bool descending; };
      std::vector<Section> sections;
      bool descending = false;
      int64_t start = 0;
      IDType last_deg = 0;
      int groups = 0;
      const int64_t ns = (int64_t)sparse_rows.size();
      auto flush = [&](int64_t end) {  // rows [start, end) of the degree-sorted list are one section
        if (end > start) sections.push_back(Section{start, end, descending});
        descending = !descending;
      };
      for (int64_t i = 0; i < ns; i++) {
        const IDType d = sorted_deg[(size_t)i];
        if (i == 0) {
          last_deg = d;
          start = 0;
        }
        if (d == 0) {  // empty rows never enter a section
          start = i + 1;
          if (i + 1 < ns) last_deg = sorted_deg[(size_t)i + 1];
          continue;
        }
        if (i != 0 && last_deg != d) {
          groups++;
          last_deg = d;
          if (groups == group_size) {
            flush(i);
            start = i;
            groups = 0;
          }
        }
        if (i == ns - 1) flush(ns);
      }
      // Threads of the sections' sorts: shared out by size BEFORE any of them starts (left to the budget's first come,
      // first served, small sections — whose records are packed sooner — can take the threads of the section of a million
      // rows).
      int64_t big_rows = 0;
      for (const Section &sc : sections)
        if (sc.end - sc.start >= ((int64_t)1 << 18)) big_rows += sc.end - sc.start;
      const unsigned hw_all = std::thread::hardware_concurrency();
      const unsigned share_total = std::min<unsigned>(hw_all ? hw_all : 1u, 2u * detail::GraySortThreadCap());
      auto sort_section = [&](const Section &sc) {
        detail::GrayBuffer<row_grey_pair> section((size_t)(sc.end - sc.start));
        detail::GrayParallelFor(sc.end - sc.start, [&](int64_t a0, int64_t a1) {  // (one piece below 64 K rows)
          for (int64_t a = a0; a < a1; a++)
            section[(size_t)a] = row_grey_pair{sparse_rows[(size_t)(sc.start + a)], (unsigned long)key[sparse_rows[(size_t)(sc.start + a)]]};
        });
        // (a section of a million rows — the rows of one entry of a power-law matrix — is the pool's longest job)
        unsigned th = 1u;  // (1: plain std::sort)
        if (sc.end - sc.start >= ((int64_t)1 << 18))
          th = std::max<unsigned>(2u, std::min<unsigned>(detail::GraySortThreadCap(),
                                                         (unsigned)((int64_t)share_total * (sc.end - sc.start) / big_rows)));
        if (!sc.descending) detail::GrayIntroSort(section.begin(), section.end(), asc_comparator, th);
        else detail::GrayIntroSort(section.begin(), section.end(), desc_comparator, th);
        detail::GrayParallelFor(sc.end - sc.start, [&](int64_t a0, int64_t a1) {
          for (int64_t a = a0; a < a1; a++) sparse_rows[(size_t)(sc.start + a)] = section[(size_t)a].first;
        });
      };
      const unsigned hw = std::thread::hardware_concurrency();
      const size_t workers = std::min<size_t>(sections.size(), std::min<unsigned>(hw ? hw : 1u, 16u));
      if (workers <= 1) {
        for (const Section &sc : sections) sort_section(sc);
      } else {
        std::atomic<size_t> next{0};
        std::vector<std::exception_ptr> errors(workers);
        detail::GrayRunWorkers((unsigned)workers, [&](unsigned w) {
          try {
            for (size_t k = next++; k < sections.size(); k = next++) sort_section(sections[k]);
          } catch (...) {
            errors[w] = std::current_exception();
          }
        });
        for (auto &e : errors)
          if (e) std::rethrow_exception(e);
      }
    }
    last_stage_ms()[5] = ms_since(t_part);
    t_part = clock::now();
    if (dense_thread.joinable()) dense_thread.join();
    if (dense_inline) sort_dense_rows();
    if (dense_error) std::rethrow_exception(dense_error);
    IDType *order = n > 0 ? new IDType[n] : new IDType[1]();  // (every row gets its position below)
    const int64_t ns_all = (int64_t)sparse_rows.size();
    detail::GrayParallelFor(n, [&](int64_t p0, int64_t p1) {  // (every row is written once: disjoint stores)
      for (int64_t pos = p0; pos < p1; pos++)
        order[pos < ns_all ? sparse_rows[(size_t)pos] : dense_rows[(size_t)(pos - ns_all)]] = (IDType)pos;
    });
    last_stage_ms()[6] = ms_since(t_part);
    return order;
  }
  static IDType *GrayReorderingCSR(std::vector<format::Format *> formats, utils::Parameters *params) {
    auto *csr = formats[0]->AsAbsolute<format::CSR<IDType, NNZType, ValueType>>();
    return Run(detail::DeviceCsrView<IDType, NNZType, ValueType>::Stage(csr, false), params);
  }
  static IDType *GrayReorderingHIPCSR(std::vector<format::Format *> formats, utils::Parameters *params) {
    auto *csr = formats[0]->AsAbsolute<format::HIPCSR<IDType, NNZType, ValueType>>();
    return Run(detail::DeviceCsrView<IDType, NNZType, ValueType>::Borrow(csr), params);
  }
};

}  // namespace sparsebase::reorder
#endif
