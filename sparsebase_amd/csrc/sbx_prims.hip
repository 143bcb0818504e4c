// sbx_prims.hip — device-wide building blocks of the reorder/convert path:
// exclusive scan (the "inclusive scan + shift" of converter_order_two.cc:185-192
// and the bucket scans of degree_reorder.cc:36-38) and a stable LSD radix sort
// (stands in for every std::sort on the path: format/coo.cc:133-146,
// format/csr.cc:123-156 for long rows, degree/RCM key orderings).
//
// Radix sort ("onesweep"): one upfront kernel builds the digit histograms of all
// passes, then every pass is ONE kernel: tiles rank their keys in LDS with wave64
// ballots (match-any over the digit bits) and per-wave digit counters, get their
// global bucket offsets by decoupled look-back over per-tile status words, and
// write keys/payloads out through LDS so the stores are coalesced per bucket run.
// Traffic per pass: one read + one write of (key, payload).
#include <stdlib.h>

#include "sbx_device.h"
#include "sbx_internal.h"

// ---------------------------------------------------------------------------
// fill
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_fill(T *__restrict__ dst, T value, int64_t count) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < count; i += stride) dst[i] = value;
}

int sbx_fill_i32(sbx_handle_t h, int32_t *dst, int32_t value, int64_t count) {
  if (count <= 0) return SBX_OK;
  SBX_KLAUNCH(h, SBX_K_MISC, k_fill<int32_t>, dim3(sbx_grid_for(count, 256, 4096)), dim3(256), dst, value, count);
  SBX_LAUNCH_CHECK(h);
  return SBX_OK;
}
int sbx_fill_i64(sbx_handle_t h, int64_t *dst, int64_t value, int64_t count) {
  if (count <= 0) return SBX_OK;
  SBX_KLAUNCH(h, SBX_K_MISC, k_fill<int64_t>, dim3(sbx_grid_for(count, 256, 4096)), dim3(256), dst, value, count);
  SBX_LAUNCH_CHECK(h);
  return SBX_OK;
}

// ---------------------------------------------------------------------------
// exclusive scan: reduce tiles -> scan tile sums (one workgroup) -> scan tiles
// ---------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

template <typename T>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_reduce(const T *__restrict__ in, T *__restrict__ partial,
                                                              int64_t count) {
  __shared__ T lds[SCAN_THREADS / 64 + 1];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
  T s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    const int64_t e = base + (int64_t)i * SCAN_THREADS + threadIdx.x;
    if (e < count) s += in[e];
  }
  s = sbx_block_sum<T, SCAN_THREADS>(s, lds);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

template <typename T>
__global__ __launch_bounds__(1024) void k_scan_partials(T *__restrict__ partial, int64_t nparts,
                                                        T *__restrict__ total_out) {
  __shared__ T lds[1024 / 64 + 1];
  T carry = 0;
  for (int64_t start = 0; start < nparts; start += 1024) {
    const int64_t e = start + threadIdx.x;
    const T v = e < nparts ? partial[e] : (T)0;
    T tot;
    const T ex = sbx_block_exclusive_sum<T, 1024>(v, lds, &tot);
    if (e < nparts) partial[e] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0 && total_out) *total_out = carry;
}

template <typename T>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_final(const T *__restrict__ in, T *__restrict__ out,
                                                             const T *__restrict__ partial, int64_t count,
                                                             T *__restrict__ total_out) {
  __shared__ T lds[SCAN_THREADS / 64 + 1];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  T v[SCAN_ITEMS];
  T s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    v[i] = (base + i < count) ? in[base + i] : (T)0;
    s += v[i];
  }
  T tot;
  T run = sbx_block_exclusive_sum<T, SCAN_THREADS>(s, lds, &tot);
  if (partial) run += partial[blockIdx.x];
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    if (base + i < count) out[base + i] = run;
    run += v[i];
  }
  if (!partial && total_out && threadIdx.x == 0) *total_out = tot;
}

template <typename T>
static int exclusive_scan_impl(sbx_handle_t h, const T *in, T *out, int64_t count, T *total_out) {
  if (count <= 0) {
    if (total_out) SBX_HIP(h, hipMemsetAsync(total_out, 0, sizeof(T), h->stream));
    return SBX_OK;
  }
  const int64_t tiles = (count + SCAN_TILE - 1) / SCAN_TILE;
  if (tiles == 1) {
    SBX_KLAUNCH(h, SBX_K_SCAN, k_scan_final<T>, dim3(1), dim3(SCAN_THREADS), in, out, (const T *)nullptr,
                       count, total_out);
    SBX_LAUNCH_CHECK(h);
    return SBX_OK;
  }
  T *partial = nullptr;
  SBX_TRY(sbx_salloc(h, (size_t)tiles, &partial));
  SBX_KLAUNCH(h, SBX_K_SCAN, k_scan_reduce<T>, dim3((unsigned)tiles), dim3(SCAN_THREADS), in, partial, count);
  SBX_KLAUNCH(h, SBX_K_SCAN, k_scan_partials<T>, dim3(1), dim3(1024), partial, tiles, total_out);
  SBX_KLAUNCH(h, SBX_K_SCAN, k_scan_final<T>, dim3((unsigned)tiles), dim3(SCAN_THREADS), in, out,
                     (const T *)partial, count, (T *)nullptr);
  SBX_LAUNCH_CHECK(h);
  return SBX_OK;
}

int sbx_exclusive_scan_i32(sbx_handle_t h, const int32_t *in, int32_t *out, int64_t count, int32_t *total_out) {
  return exclusive_scan_impl<int32_t>(h, in, out, count, total_out);
}
int sbx_exclusive_scan_i64(sbx_handle_t h, const int64_t *in, int64_t *out, int64_t count, int64_t *total_out) {
  return exclusive_scan_impl<int64_t>(h, in, out, count, total_out);
}
int sbx_exclusive_scan_u32(sbx_handle_t h, const uint32_t *in, uint32_t *out, int64_t count, uint32_t *total_out) {
  return exclusive_scan_impl<uint32_t>(h, in, out, count, total_out);
}

// ---------------------------------------------------------------------------
// radix sort
// ---------------------------------------------------------------------------
int sbx_radix_plan(int lo0, int hi0, int lo1, int hi1, sbx_radix_pass *passes) {
  int np = 0;
  const int lo[2] = {lo0, lo1}, hi[2] = {hi0, hi1};
  for (int r = 0; r < 2; r++) {
    int b = lo[r];
    while (b < hi[r]) {
      // spread the range's bits evenly over ceil(range/8) passes
      const int remaining = hi[r] - b;
      const int passes_left = (remaining + 7) / 8;
      const int take = (remaining + passes_left - 1) / passes_left;
      passes[np].shift = b;
      passes[np].bits = take;
      np++;
      b += take;
    }
  }
  return np;
}

constexpr int RS_THREADS = 256;
constexpr int RS_MAX_PASSES = 8;
#ifndef SBX_RS_LOOKBACK
#define SBX_RS_LOOKBACK 8
#endif
constexpr int RS_LOOKBACK = SBX_RS_LOOKBACK;  // predecessor status words fetched per look-back round
#ifndef RS_STALL_SLEEP
#define RS_STALL_SLEEP 1
#endif

typedef unsigned long long rs_word;  // look-back status word: (count << 2) | flag  (32-bit words, count < 2^30: +2 % only)

struct RadixPlan {
  int n;
  int tied;  // the caller expects heavily tied digits (sbx_handle::rs_tied_hint): the histogram aggregates inside a wave
  int shift[RS_MAX_PASSES];
  int bits[RS_MAX_PASSES];
};

// Upfront digit histograms of every pass in one read of the keys.
// KS: the 64-bit keys are two 32-bit arrays (keys = low words, keys_hi = high words).
#ifndef SBX_RSH_THREADS
#define SBX_RSH_THREADS 256
#endif
constexpr int RSH_THREADS = SBX_RSH_THREADS;  // threads of a histogram workgroup (the first 256 own one digit each)
template <typename K, bool KS = false>
__global__ __launch_bounds__(RSH_THREADS) void k_onesweep_hist(const K *__restrict__ keys, int64_t count, RadixPlan plan,
                                                              unsigned long long *__restrict__ ghist,
                                                              rs_word *__restrict__ state,
                                                              size_t state_words,
                                                              const uint32_t *__restrict__ keys_hi = nullptr) {
  auto key_at = [&](int64_t j) -> K {
    if constexpr (KS) return (K)(((uint64_t)keys_hi[j] << 32) | ((const uint32_t *)keys)[j]);
    else return keys[j];
  };
  __shared__ unsigned lh[RS_MAX_PASSES][256];
  // the look-back status words of all passes are only touched by the pass kernels: clear them here
  for (size_t j = (size_t)blockIdx.x * RSH_THREADS + threadIdx.x; j < state_words; j += (size_t)gridDim.x * RSH_THREADS)
    state[j] = 0;
  for (int i = threadIdx.x; i < RS_MAX_PASSES * 256; i += RSH_THREADS) (&lh[0][0])[i] = 0;
  __syncthreads();
  // HU keys per thread and round, all loads issued before the first counter update: one key per round leaves a
  // thread with one load in flight and the kernel at 1 TB/s
  constexpr int HU = 8;
  const int64_t stride = (int64_t)gridDim.x * RSH_THREADS;
  int64_t i = (int64_t)blockIdx.x * RSH_THREADS + threadIdx.x;
  for (; i + (HU - 1) * stride < count; i += HU * stride) {
    K k[HU];
#pragma unroll
    for (int u = 0; u < HU; u++) k[u] = key_at(i + u * stride);
#pragma unroll
    for (int u = 0; u < HU; u++)
      for (int p = 0; p < plan.n; p++) {
        // Tied keys (class and degree fields, the Gray keys of a power-law matrix) send most lanes of a wave to ONE
        // counter, and LDS adds to one address go one lane at a time: the lanes that share the first lane's digit are
        // counted with a ballot and added once.
        // (only where the caller says so: on keys without ties the ballots cost the 105 M-key sorts 1 - 2 %)
        const unsigned d = (unsigned)(k[u] >> plan.shift[p]) & ((1u << plan.bits[p]) - 1u);
        if (!plan.tied) {
          atomicAdd(&lh[p][d], 1u);
        } else {
          const unsigned d0 = (unsigned)__builtin_amdgcn_readfirstlane((int)d);
          const unsigned long long same = __ballot(d == d0);
          if (d != d0) atomicAdd(&lh[p][d], 1u);
          else if ((same & sbx_lanemask_lt()) == 0) atomicAdd(&lh[p][d0], (unsigned)__popcll(same));
        }
      }
  }
  for (; i < count; i += stride) {
    const K k = key_at(i);
    for (int p = 0; p < plan.n; p++)
      atomicAdd(&lh[p][(unsigned)(k >> plan.shift[p]) & ((1u << plan.bits[p]) - 1u)], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 256)
    for (int p = 0; p < plan.n; p++) {
      const unsigned c = lh[p][threadIdx.x];
      if (c) atomicAdd(&ghist[p * 256 + threadIdx.x], (unsigned long long)c);
    }
}

// diagnostic build only (-DSBX_RADIX_STAMPS, tools/radix_stamps.py): wall-clock cycles from kernel entry to the end of
// each phase of k_onesweep_pass, summed over the tiles by thread 0; [15] counts the tiles
#ifdef SBX_RADIX_STAMPS
__device__ unsigned long long g_rs_stamps[16];
#define RS_STAMP(i, drain)                                                          \
  do {                                                                              \
    if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     \
    if (tid == 0) {                                                                 \
      unsigned long long t_;                                                        \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");    \
      atomicAdd(&g_rs_stamps[i], (i) == 15 ? 1ull : t_ - t_start);                  \
    }                                                                               \
  } while (0)
extern "C" int sbx_debug_radix_stamps(unsigned long long *out, int clear) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rs_stamps), sizeof(g_rs_stamps)) != hipSuccess) return 1;
  if (clear) {
    unsigned long long z[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_rs_stamps), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
#else
#define RS_STAMP(i, drain) do { } while (0)
#endif

// One digit pass in a single kernel.  Tiles take a ticket (so a tile only ever waits
// for tiles that already run), rank their keys in LDS exactly like a classic scatter
// pass, and obtain the number of equal-digit keys in all earlier tiles by decoupled
// look-back over per-(tile,digit) status words: (value << 2) | flag, flag 1 = this
// tile's own count, 2 = inclusive prefix.  Each word is a self-contained granule
// written/read with relaxed agent-scope atomics, so no fence is needed.  `ghist` holds the
// pass's raw digit counts; every tile scans them itself (256 values) instead of paying a
// separate one-workgroup launch per sort.
//
// EMIT (final pass of sbx_radix_sort_emit): the sorted keys are not stored; the pass writes what the caller wanted
// them for — out[position] = value (the key's low word, optionally mapped), value's bit in up to two bitmaps,
// pos_of[value] = position — and saves the caller a kernel that re-reads the sorted keys.
// Workgroup barrier that waits for this wave's LDS operations only.  __syncthreads() also drains vmcnt, i.e. the loads
// of the NEXT tile that the pass kernel keeps in flight while it works on the current one.
constexpr int RSP_THREADS = 512;  // threads of a pass workgroup (the first 256 own one digit each)
constexpr int RSP_WAVES = RSP_THREADS / 64;
__device__ __forceinline__ void rs_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ uint32_t rs_block_exclusive_sum(uint32_t v, uint32_t *lds) {
  const uint32_t inc = sbx_wave_inclusive_sum(v);
  const int w = sbx_wave_in_block();
  if (sbx_lane() == 63) lds[w] = inc;
  rs_lds_barrier();
  uint32_t woff = 0;
#pragma unroll
  for (int i = 0; i < RSP_WAVES; i++) {
    const uint32_t t = lds[i];
    if (i < w) woff += t;
  }
  rs_lds_barrier();
  return woff + inc - v;
}

// The workgroups are persistent: each takes tickets until the tiles run out, and holds the keys and payloads of its
// NEXT tile in registers (loads issued before the current tile is touched) — with one tile per workgroup the phases
// of a tile (load, rank, look-back, store) run one after the other and a CU keeps ~17 KB of loads in flight, a quarter
// of what the HBM latency asks for (measured with tools/radix_stamps.py: 2.1 TB/s per pass whatever the size).
//
// IOM (sbx_radix_sort_io): bit 0 / 1 — the pass LOADS 64-bit keys / payloads from two 32-bit arrays (keys_in = low words,
// io.k_hi_in = high words; vals_in = low, io.p_hi_in = high); bit 2 / 3 — it STORES them that way.
struct RsSplit {
  const uint32_t *k_hi_in, *p_hi_in;
  uint32_t *k_hi_out, *p_hi_out;
};
template <typename K, typename P, int ITEMS, bool HAS_P, bool EMIT, int IOM = 0>
__global__ __launch_bounds__(RSP_THREADS, 4) void k_onesweep_pass(const K *__restrict__ keys_in, K *__restrict__ keys_out,
                                                              const P *__restrict__ vals_in, P *__restrict__ vals_out,
                                                              int64_t count, int shift, int bits,
                                                              const unsigned long long *__restrict__ ghist,
                                                              rs_word *state, unsigned *ticket,
                                                              sbx_radix_emit em, RsSplit io) {
  constexpr int TILE = RSP_THREADS * ITEMS;
  __shared__ K s_keys[TILE];
  __shared__ P s_vals[HAS_P ? TILE : 1];
  __shared__ uint32_t s_whist[RSP_WAVES][256];
  __shared__ uint32_t s_gofs[256];
  __shared__ uint32_t s_cnt[256];
  __shared__ uint32_t s_scan[RSP_WAVES + 1];
  __shared__ unsigned s_tile[2];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const unsigned mask = (1u << bits) - 1u;
  const unsigned tiles = (unsigned)((count + TILE - 1) / TILE);
  const bool owner = tid < 256;                  // thread `tid` owns digit `tid`
  const bool live = owner && (unsigned)tid <= mask;  // digits the pass cannot produce take no part in the look-back
  const uint64_t lt = sbx_lanemask_lt();
  const int e0 = w * 64 * ITEMS + lane;     // this thread's items of a tile: e0 + i * 64
#ifdef SBX_RADIX_STAMPS
  unsigned long long t_start;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_start)::"memory");
#endif
  // tickets are taken two tiles ahead: the returning atomic takes ~3 us, the tile after the current one must be known
  // when the current one starts (its loads are issued then)
  // (a launch with a workgroup per tile takes one ticket per workgroup: the ticket word serves ~88 atomics per
  // microsecond, and the small sorts of the RCM are all of that kind)
  const bool single = tiles <= gridDim.x;
  if (tid == 0) {
    s_tile[0] = atomicAdd(ticket, 1u);
    s_tile[1] = single ? 0xffffffffu : atomicAdd(ticket, 1u);
  }
  __syncthreads();
  unsigned tile = s_tile[0], next = s_tile[1];
  // first output position of digit `tid` = exclusive scan of the global counts (count < 2^32: 32 bits do)
  const uint32_t gbase = rs_block_exclusive_sum(owner ? (uint32_t)ghist[tid] : 0u, s_scan);

  K kn[ITEMS];
  P vn[HAS_P ? ITEMS : 1];
  // every load is issued whatever the tile (clamped address): a load under a condition makes the compiler wait for
  // it at the join
#define RS_PREFETCH(t_)                                                                \
  do {                                                                                 \
    const int64_t last_ = count - 1, b_ = (int64_t)(t_) * TILE + e0;                   \
    _Pragma("unroll") for (int i = 0; i < ITEMS; i++) {                                \
      const int64_t a_ = b_ + i * 64 < last_ ? b_ + i * 64 : last_;                    \
      if constexpr ((IOM & 1) != 0)                                                    \
        kn[i] = (K)(((uint64_t)io.k_hi_in[a_] << 32) | ((const uint32_t *)keys_in)[a_]); \
      else kn[i] = keys_in[a_];                                                        \
      if constexpr (HAS_P && (IOM & 2) != 0)                                           \
        vn[i] = (P)(((uint64_t)io.p_hi_in[a_] << 32) | ((const uint32_t *)vals_in)[a_]); \
      else if (HAS_P) vn[i] = vals_in[a_];                                             \
    }                                                                                  \
  } while (0)
  RS_PREFETCH(tile);
  unsigned after_next = 0;  // thread 0 only
  // A tile's digit counts are published as soon as its keys are here — for the first tile now, for every other one
  // while the workgroup still works on the tile before it: the tiles behind wait for these words, and the slowest
  // workgroup in flight sets for how long.
#define RS_COUNT_PUBLISH(t_, out_)                                                                              \
  do {                                                                                                          \
    const int64_t left_ = count - (int64_t)(t_) * TILE;                                                         \
    const int valid_ = (int)(left_ < TILE ? (left_ > 0 ? left_ : 0) : TILE);                                    \
    _Pragma("unroll") for (int i = 0; i < ITEMS; i++)                                                           \
      if (e0 + i * 64 < valid_) atomicAdd(&s_cnt[(unsigned)(kn[i] >> shift) & mask], 1u);                       \
    rs_lds_barrier();                                                                                           \
    out_ = owner ? s_cnt[tid] : 0u;                                                                             \
    if (live && (t_) < tiles)                                                                                   \
      __hip_atomic_store(state + (size_t)(t_) * 256 + tid, ((rs_word)out_ << 2) | (rs_word)((t_) == 0 ? 2u : 1u), \
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                                           \
  } while (0)
  if (owner) s_cnt[tid] = 0;
  rs_lds_barrier();
  uint32_t tot_next;
  RS_COUNT_PUBLISH(tile, tot_next);
  while (tile < tiles) {
#ifdef SBX_RADIX_STAMPS
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_start)::"memory");
#endif
    const int64_t base = (int64_t)tile * TILE;
    const int valid = (int)((count - base) < TILE ? (count - base) : TILE);
    K k[ITEMS];
    P v[HAS_P ? ITEMS : 1];
    uint32_t rank[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
      k[i] = e0 + i * 64 < valid ? kn[i] : (K) ~(K)0;
      if (HAS_P) v[i] = vn[i];
    }
    if (tid == 0) after_next = single ? 0xffffffffu : atomicAdd(ticket, 1u);  // needed at the end of this tile
    if (owner) {
#pragma unroll
      for (int i = 0; i < RSP_WAVES; i++) s_whist[i][tid] = 0;
      s_cnt[tid] = 0;
    }
    rs_lds_barrier();
    RS_STAMP(0, false);
    RS_PREFETCH(next);
    RS_STAMP(1, false);
    rs_word *mine = state + (size_t)tile * 256 + tid;
    const uint32_t tot_valid = tot_next;  // counted and published one tile ago
    volatile uint32_t *wh = s_whist[w];
    // the lanes holding the same digit as this one, for every item first (independent ballots), then the chain of
    // counter updates (one LDS round trip per item, nothing else left in it)
    uint64_t same[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
      const unsigned d = (unsigned)(k[i] >> shift) & mask;
      uint64_t m = ~(uint64_t)0;
      for (int b = 0; b < bits; b++) {
        const bool bit = (d >> b) & 1u;
        const uint64_t bal = __ballot(bit);
        m &= bit ? bal : ~bal;
      }
      same[i] = m;
    }
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
      const unsigned d = (unsigned)(k[i] >> shift) & mask;
      const uint32_t prev = wh[d];
      const uint32_t r = (uint32_t)__popcll(same[i] & lt);
      __builtin_amdgcn_wave_barrier();
      if (r == 0) wh[d] = prev + (uint32_t)__popcll(same[i]);
      __builtin_amdgcn_wave_barrier();
      rank[i] = prev + r;
    }
    rs_lds_barrier();
    RS_STAMP(2, false);
    // thread `tid` owns digit `tid`
    uint32_t ex0;
    {
      uint32_t c[RSP_WAVES];
      uint32_t tot = 0;
#pragma unroll
      for (int i = 0; i < RSP_WAVES; i++) {
        c[i] = owner ? s_whist[i][tid] : 0u;
        tot += c[i];
      }
      uint32_t ex = rs_block_exclusive_sum(tot, s_scan);
      ex0 = ex;
      if (owner) {
#pragma unroll
        for (int i = 0; i < RSP_WAVES; i++) {
          s_whist[i][tid] = ex;
          ex += c[i];
        }
      }
    }
    rs_lds_barrier();
    RS_STAMP(3, false);
    // sorted by digit inside the tile (LDS) while the predecessors' counts arrive
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
      const unsigned d = (unsigned)(k[i] >> shift) & mask;
      const uint32_t pos = s_whist[w][d] + rank[i];
      s_keys[pos] = k[i];
      if (HAS_P) s_vals[pos] = v[i];
    }
    if (next < tiles) RS_COUNT_PUBLISH(next, tot_next);  // (the same for the whole workgroup)
    {
      // ---- decoupled look-back for digit `tid`
      uint32_t before = 0;
      if (tile != 0 && live) {
        // windowed look-back: fetch up to RS_LOOKBACK predecessor words at once (independent
        // loads), consume them nearest-first; a word that is not published yet is re-polled
        int64_t t = (int64_t)tile - 1;
        bool done = false;
#ifdef SBX_RADIX_STAMPS
        unsigned n_rounds = 0, n_stalls = 0, n_words = 0;
#endif
        while (!done) {
          rs_word wv[RS_LOOKBACK];
#pragma unroll
          for (int i = 0; i < RS_LOOKBACK; i++)
            wv[i] = (t - i >= 0) ? __hip_atomic_load(state + (size_t)(t - i) * 256 + tid, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT)
                                 : (rs_word)2;  // before tile 0: an empty inclusive prefix
          int consumed = 0;
          bool stall = false;
#pragma unroll
          for (int i = 0; i < RS_LOOKBACK; i++) {
            const bool active = !done && !stall;
            const unsigned flag = (unsigned)(wv[i] & 3u);
            if (active && flag == 0u) {
              stall = true;  // not published yet: resume the walk at this tile
            } else if (active) {
              before += (uint32_t)(wv[i] >> 2);
              consumed++;
              if (flag == 2u) done = true;
            }
          }
          t -= consumed;
          if (stall) __builtin_amdgcn_s_sleep(RS_STALL_SLEEP);
#ifdef SBX_RADIX_STAMPS
          n_rounds++; n_stalls += stall; n_words += consumed;
#endif
        }
#ifdef SBX_RADIX_STAMPS
        if (tid == 0) {
          atomicAdd(&g_rs_stamps[8], (unsigned long long)n_rounds);
          atomicAdd(&g_rs_stamps[9], (unsigned long long)n_stalls);
          atomicAdd(&g_rs_stamps[10], (unsigned long long)n_words);
        }
#endif
        __hip_atomic_store(mine, ((rs_word)(before + tot_valid) << 2) | (rs_word)2, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      }
      RS_STAMP(4, false);
      if (owner) s_gofs[tid] = gbase + before - ex0;
    }
    rs_lds_barrier();
    RS_STAMP(5, false);
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
      const int j = i * RSP_THREADS + tid;
      if (j < valid) {
        const K kk = s_keys[j];
        const unsigned d = (unsigned)(kk >> shift) & mask;
        const uint32_t o = s_gofs[d] + (uint32_t)j;
        if (EMIT) {
          const uint32_t lo = em.low_mask ? (uint32_t)kk & em.low_mask : (uint32_t)kk;
          const uint32_t ev = em.map ? em.map[lo] : lo;
          em.out[o] = ev;
          if (em.bits_a) atomicOr(&em.bits_a[ev >> 5], 1u << (ev & 31));
          if (em.bits_b) atomicOr(&em.bits_b[ev >> 5], 1u << (ev & 31));
          if (em.pos_of) em.pos_of[ev] = o;
        } else {
          if constexpr ((IOM & 4) != 0) {
            ((uint32_t *)keys_out)[o] = (uint32_t)kk;
            io.k_hi_out[o] = (uint32_t)((uint64_t)kk >> 32);
          } else {
            keys_out[o] = kk;
          }
          if constexpr (HAS_P && (IOM & 8) != 0) {
            const P pv = s_vals[j];
            ((uint32_t *)vals_out)[o] = (uint32_t)pv;
            io.p_hi_out[o] = (uint32_t)((uint64_t)pv >> 32);
          } else if (HAS_P) {
            vals_out[o] = s_vals[j];
          }
        }
      }
    }
    RS_STAMP(6, false);
    RS_STAMP(15, false);
    if (tid == 0) s_tile[0] = after_next;
    rs_lds_barrier();  // the LDS of this tile has been read: the next one may clear and fill it
    tile = next;
    next = s_tile[0];
  }
#undef RS_PREFETCH
#undef RS_COUNT_PUBLISH
}

static int rs_hist_grid_factor() {  // SBX_RADIX_HIST_GRID: workgroups per CU of the histogram kernel (tuning)
  static const int f = sbx_env_tuning("SBX_RADIX_HIST_GRID") ? atoi(sbx_env_tuning("SBX_RADIX_HIST_GRID")) : 2;
  return f < 1 ? 1 : f;
}

template <typename K, typename P, int ITEMS, bool HAS_P, bool EMIT = false>
static int radix_sort_impl(sbx_handle_t h, K *ka, K *kb, P *va, P *vb, int64_t count, const sbx_radix_pass *passes,
                           int np, int *result_in_b, const sbx_radix_emit *emit = nullptr) {
  constexpr int TILE = RSP_THREADS * ITEMS;
  *result_in_b = 0;
  if (count <= 1 || np == 0) return SBX_OK;
  if (np > RS_MAX_PASSES) SBX_FAIL(h, SBX_ERR_INTERNAL, "radix sort: %d passes requested", np);
  if (count >= ((int64_t)1 << 32)) SBX_FAIL(h, SBX_ERR_UNSUPPORTED, "radix sort: count %lld >= 2^32", (long long)count);
  const int64_t tiles = (count + TILE - 1) / TILE;
  RadixPlan plan;
  plan.n = np;
  plan.tied = h->rs_tied_hint ? 1 : 0;
  for (int p = 0; p < np; p++) {
    plan.shift[p] = passes[p].shift;
    plan.bits[p] = passes[p].bits;
  }
  // zeroed (bins, tickets) slot from the handle's pool; the per-(pass, tile, digit) status words are
  // cleared by the histogram kernel
  static_assert(RS_MAX_PASSES * 256 * 8 + 256 <= SBX_RS_SLOT_BYTES, "radix slot too small");
  void *slot = nullptr;
  SBX_TRY(sbx_radix_slot(h, &slot));
  unsigned long long *ghist = (unsigned long long *)slot;
  unsigned *tickets = (unsigned *)((char *)slot + (size_t)RS_MAX_PASSES * 256 * 8);
  const size_t state_words = (size_t)np * tiles * 256;
  rs_word *state = nullptr;
  SBX_TRY(sbx_salloc(h, state_words, &state));
  // few, fat workgroups: every workgroup ends with passes x 256 adds on the same histogram words, and one word takes
  // only ~88 adds per microsecond whoever issues them
  SBX_KLAUNCH(h, SBX_K_RADIX_HIST, (k_onesweep_hist<K>),
              dim3(sbx_grid_for(count, RSH_THREADS * 16, (int64_t)h->num_cus * rs_hist_grid_factor())), dim3(RSH_THREADS),
              (const K *)ka, count, plan, ghist, state, state_words);
  SBX_PROF_BYTES(h, SBX_K_RADIX_HIST, count * (int64_t)sizeof(K));  // one read of the keys for all passes
  K *src_k = ka, *dst_k = kb;
  P *src_v = va, *dst_v = vb;
  // persistent workgroups: as many as are resident at once (the LDS of a tile allows 160 KB / footprint per CU)
  constexpr size_t lds_bytes = sizeof(K) * TILE + (HAS_P ? sizeof(P) * TILE : 0) + (RSP_WAVES + 2) * 256 * 4 + 64;
  // (two at most: 8 waves of up to 128 VGPRs each)
  const int per_cu = (int)((160 * 1024) / lds_bytes) < 2 ? (int)((160 * 1024) / lds_bytes) : 2;
  const unsigned pass_grid = (unsigned)(tiles < (int64_t)h->num_cus * per_cu ? tiles : (int64_t)h->num_cus * per_cu);
  const sbx_radix_emit no_emit = {nullptr, nullptr, nullptr, nullptr, nullptr, 0u};
  for (int p = 0; p < np; p++) {
    if (EMIT && p == np - 1)
      SBX_KLAUNCH(h, SBX_K_RADIX_SCATTER, (k_onesweep_pass<K, P, ITEMS, HAS_P, EMIT>), dim3(pass_grid),
                  dim3(RSP_THREADS), (const K *)src_k, dst_k, (const P *)src_v, dst_v, count, passes[p].shift,
                  passes[p].bits, (const unsigned long long *)(ghist + (size_t)p * 256),
                  state + (size_t)p * tiles * 256, tickets + p, *emit, RsSplit{});
    else
      SBX_KLAUNCH(h, SBX_K_RADIX_SCATTER, (k_onesweep_pass<K, P, ITEMS, HAS_P, false>), dim3(pass_grid),
                  dim3(RSP_THREADS), (const K *)src_k, dst_k, (const P *)src_v, dst_v, count, passes[p].shift,
                  passes[p].bits, (const unsigned long long *)(ghist + (size_t)p * 256),
                  state + (size_t)p * tiles * 256, tickets + p, no_emit, RsSplit{});
    // a pass reads and writes every (key, payload) record once
    SBX_PROF_BYTES(h, SBX_K_RADIX_SCATTER, 2 * count * (int64_t)(sizeof(K) + (HAS_P ? sizeof(P) : 0)));
    SBX_LAUNCH_CHECK(h);
    K *tk = src_k; src_k = dst_k; dst_k = tk;
    P *tv = src_v; src_v = dst_v; dst_v = tv;
    *result_in_b ^= 1;
  }
  return SBX_OK;
}

// sbx_radix_sort_io: KSPLIT — 64-bit keys split on the caller's sides; PSPLIT — 64-bit payloads split there
template <typename K, typename P, int ITEMS, bool HAS_P, bool KSPLIT, bool PSPLIT>
static int radix_sort_io_impl(sbx_handle_t h, const sbx_radix_side *src, K *ka, K *kb, P *va, P *vb,
                              const sbx_radix_side *dst, int64_t count, const sbx_radix_pass *passes, int np) {
  constexpr int TILE = RSP_THREADS * ITEMS;
  constexpr int IN = (KSPLIT ? 1 : 0) | (PSPLIT ? 2 : 0), OUT = (KSPLIT ? 4 : 0) | (PSPLIT ? 8 : 0);
  if (np > RS_MAX_PASSES) SBX_FAIL(h, SBX_ERR_INTERNAL, "radix sort: %d passes requested", np);
  if (count >= ((int64_t)1 << 32)) SBX_FAIL(h, SBX_ERR_UNSUPPORTED, "radix sort: count %lld >= 2^32", (long long)count);
  const int64_t tiles = (count + TILE - 1) / TILE;
  RadixPlan plan;
  plan.n = np;
  plan.tied = h->rs_tied_hint ? 1 : 0;
  for (int p = 0; p < np; p++) {
    plan.shift[p] = passes[p].shift;
    plan.bits[p] = passes[p].bits;
  }
  void *slot = nullptr;
  SBX_TRY(sbx_radix_slot(h, &slot));
  unsigned long long *ghist = (unsigned long long *)slot;
  unsigned *tickets = (unsigned *)((char *)slot + (size_t)RS_MAX_PASSES * 256 * 8);
  const size_t state_words = (size_t)np * tiles * 256;
  rs_word *state = nullptr;
  SBX_TRY(sbx_salloc(h, state_words, &state));
  SBX_KLAUNCH(h, SBX_K_RADIX_HIST, (k_onesweep_hist<K, KSPLIT>),
              dim3(sbx_grid_for(count, RSH_THREADS * 16, (int64_t)h->num_cus * rs_hist_grid_factor())), dim3(RSH_THREADS),
              (const K *)src->k[0], count, plan, ghist, state, state_words, (const uint32_t *)src->k[1]);
  SBX_PROF_BYTES(h, SBX_K_RADIX_HIST, count * (int64_t)sizeof(K));
  constexpr size_t lds_bytes = sizeof(K) * TILE + (HAS_P ? sizeof(P) * TILE : 0) + (RSP_WAVES + 2) * 256 * 4 + 64;
  const int per_cu = (int)((160 * 1024) / lds_bytes) < 2 ? (int)((160 * 1024) / lds_bytes) : 2;
  const unsigned pass_grid = (unsigned)(tiles < (int64_t)h->num_cus * per_cu ? tiles : (int64_t)h->num_cus * per_cu);
  const sbx_radix_emit no_emit = {nullptr, nullptr, nullptr, nullptr, nullptr, 0u};
  K *bk[2] = {ka, kb};
  P *bv[2] = {va, vb};
  for (int p = 0; p < np; p++) {
    const bool first = p == 0, last = p == np - 1;
    const K *ik = first ? (const K *)src->k[0] : bk[(p - 1) & 1];
    const P *iv = first ? (const P *)src->p[0] : bv[(p - 1) & 1];
    K *ok = last ? (K *)dst->k[0] : bk[p & 1];
    P *ov = last ? (P *)dst->p[0] : bv[p & 1];
    const RsSplit io = {first ? (const uint32_t *)src->k[1] : nullptr, first ? (const uint32_t *)src->p[1] : nullptr,
                        last ? (uint32_t *)dst->k[1] : nullptr, last ? (uint32_t *)dst->p[1] : nullptr};
#define RS_IO_LAUNCH(M)                                                                                            \
  SBX_KLAUNCH(h, SBX_K_RADIX_SCATTER, (k_onesweep_pass<K, P, ITEMS, HAS_P, false, M>), dim3(pass_grid),            \
              dim3(RSP_THREADS), ik, ok, iv, ov, count, passes[p].shift, passes[p].bits,                           \
              (const unsigned long long *)(ghist + (size_t)p * 256), state + (size_t)p * tiles * 256, tickets + p, \
              no_emit, io)
    if (first && last) RS_IO_LAUNCH(IN | OUT);
    else if (first) RS_IO_LAUNCH(IN);
    else if (last) RS_IO_LAUNCH(OUT);
    else RS_IO_LAUNCH(0);
#undef RS_IO_LAUNCH
    SBX_PROF_BYTES(h, SBX_K_RADIX_SCATTER, 2 * count * (int64_t)(sizeof(K) + (HAS_P ? sizeof(P) : 0)));
    SBX_LAUNCH_CHECK(h);
  }
  return SBX_OK;
}

int sbx_radix_sort_io(sbx_handle_t h, int key_bytes, int payload_bytes, const sbx_radix_side *src, void *keys_a,
                      void *keys_b, void *vals_a, void *vals_b, const sbx_radix_side *dst, int64_t count,
                      const sbx_radix_pass *passes, int num_passes) {
  if (!src || !dst || num_passes < 1 || count < 2 || src->k_split != dst->k_split || src->p_split != dst->p_split)
    SBX_FAIL(h, SBX_ERR_INTERNAL, "sbx_radix_sort_io: bad sides");
  if (num_passes == 1) {  // one pass reads the source side and writes the destination side: no array may be on both
    const void *ins[4] = {src->k[0], src->k_split ? src->k[1] : nullptr, payload_bytes ? src->p[0] : nullptr,
                          (payload_bytes && src->p_split) ? src->p[1] : nullptr};
    const void *outs[4] = {dst->k[0], dst->k_split ? dst->k[1] : nullptr, payload_bytes ? dst->p[0] : nullptr,
                           (payload_bytes && dst->p_split) ? dst->p[1] : nullptr};
    for (const void *a : ins)
      for (const void *b : outs)
        if (a && a == b) SBX_FAIL(h, SBX_ERR_INTERNAL, "sbx_radix_sort_io: a single pass cannot sort in place");
  }
  if (key_bytes == 8 && src->k_split && !src->p_split) {
    if (payload_bytes == 0)
      return radix_sort_io_impl<uint64_t, uint32_t, 8, false, true, false>(h, src, (uint64_t *)keys_a, (uint64_t *)keys_b,
                                                                          nullptr, nullptr, dst, count, passes, num_passes);
    if (payload_bytes == 4)
      return radix_sort_io_impl<uint64_t, uint32_t, 8, true, true, false>(h, src, (uint64_t *)keys_a, (uint64_t *)keys_b,
                                                                         (uint32_t *)vals_a, (uint32_t *)vals_b, dst, count,
                                                                         passes, num_passes);
    if (payload_bytes == 8)
      return radix_sort_io_impl<uint64_t, uint64_t, 4, true, true, false>(h, src, (uint64_t *)keys_a, (uint64_t *)keys_b,
                                                                         (uint64_t *)vals_a, (uint64_t *)vals_b, dst, count,
                                                                         passes, num_passes);
  } else if (key_bytes == 4 && !src->k_split && payload_bytes == 8 && src->p_split) {
    return radix_sort_io_impl<uint32_t, uint64_t, 8, true, false, true>(h, src, (uint32_t *)keys_a, (uint32_t *)keys_b,
                                                                       (uint64_t *)vals_a, (uint64_t *)vals_b, dst, count,
                                                                       passes, num_passes);
  } else if (key_bytes == 4 && !src->k_split && payload_bytes == 4 && !src->p_split) {
    return radix_sort_io_impl<uint32_t, uint32_t, 8, true, false, false>(h, src, (uint32_t *)keys_a, (uint32_t *)keys_b,
                                                                        (uint32_t *)vals_a, (uint32_t *)vals_b, dst, count,
                                                                        passes, num_passes);
  }
  SBX_FAIL(h, SBX_ERR_UNSUPPORTED, "sbx_radix_sort_io: key_bytes=%d payload_bytes=%d split=%d/%d", key_bytes,
           payload_bytes, (int)src->k_split, (int)src->p_split);
}

int sbx_radix_sort_emit(sbx_handle_t h, void *keys_a, void *keys_b, int64_t count, const sbx_radix_pass *passes,
                        int num_passes, const sbx_radix_emit *emit) {
  if (!emit || !emit->out || num_passes < 1 || count < 2)
    SBX_FAIL(h, SBX_ERR_INTERNAL, "sbx_radix_sort_emit: needs an output, a pass and two keys");
  int in_b = 0;
  return radix_sort_impl<uint64_t, uint32_t, 8, false, true>(h, (uint64_t *)keys_a, (uint64_t *)keys_b, nullptr, nullptr,
                                                              count, passes, num_passes, &in_b, emit);
}

int sbx_radix_sort(sbx_handle_t h, int key_bytes, int payload_bytes, void *keys_a, void *keys_b, void *vals_a,
                   void *vals_b, int64_t count, const sbx_radix_pass *passes, int num_passes, int *result_in_b) {
  if (key_bytes == 4) {
    if (payload_bytes == 0)
      return radix_sort_impl<uint32_t, uint32_t, 8, false>(h, (uint32_t *)keys_a, (uint32_t *)keys_b, nullptr, nullptr,
                                                            count, passes, num_passes, result_in_b);
    if (payload_bytes == 4)
      return radix_sort_impl<uint32_t, uint32_t, 8, true>(h, (uint32_t *)keys_a, (uint32_t *)keys_b,
                                                           (uint32_t *)vals_a, (uint32_t *)vals_b, count, passes,
                                                           num_passes, result_in_b);
    if (payload_bytes == 8)
      return radix_sort_impl<uint32_t, uint64_t, 8, true>(h, (uint32_t *)keys_a, (uint32_t *)keys_b,
                                                           (uint64_t *)vals_a, (uint64_t *)vals_b, count, passes,
                                                           num_passes, result_in_b);
  } else if (key_bytes == 8) {
    if (payload_bytes == 0)
      return radix_sort_impl<uint64_t, uint32_t, 8, false>(h, (uint64_t *)keys_a, (uint64_t *)keys_b, nullptr, nullptr,
                                                            count, passes, num_passes, result_in_b);
    if (payload_bytes == 4)
      return radix_sort_impl<uint64_t, uint32_t, 8, true>(h, (uint64_t *)keys_a, (uint64_t *)keys_b,
                                                           (uint32_t *)vals_a, (uint32_t *)vals_b, count, passes,
                                                           num_passes, result_in_b);
    if (payload_bytes == 8)
      return radix_sort_impl<uint64_t, uint64_t, 4, true>(h, (uint64_t *)keys_a, (uint64_t *)keys_b,
                                                          (uint64_t *)vals_a, (uint64_t *)vals_b, count, passes,
                                                          num_passes, result_in_b);
  }
  SBX_FAIL(h, SBX_ERR_UNSUPPORTED, "radix sort: key_bytes=%d payload_bytes=%d", key_bytes, payload_bytes);
}
