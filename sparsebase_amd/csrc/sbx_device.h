// sbx_device.h — wave64 / workgroup device helpers (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SBX_WAVE 64

__device__ __forceinline__ int sbx_lane() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ int sbx_wave_in_block() { return (int)(threadIdx.x >> 6); }

__device__ __forceinline__ uint64_t sbx_lanemask_lt() {
  return ((uint64_t)1 << sbx_lane()) - 1;
}

// inclusive wave scan (sum)
template <typename T>
__device__ __forceinline__ T sbx_wave_inclusive_sum(T v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    T o = __shfl_up(v, d, 64);
    if (sbx_lane() >= d) v += o;
  }
  return v;
}

template <typename T>
__device__ __forceinline__ T sbx_wave_inclusive_max(T v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    T o = __shfl_up(v, d, 64);
    if (sbx_lane() >= d) v = o > v ? o : v;
  }
  return v;
}

template <typename T>
__device__ __forceinline__ T sbx_wave_sum(T v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

template <typename T>
__device__ __forceinline__ T sbx_wave_max(T v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    T o = __shfl_xor(v, d, 64);
    v = o > v ? o : v;
  }
  return v;
}

template <typename T>
__device__ __forceinline__ T sbx_wave_min(T v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    T o = __shfl_xor(v, d, 64);
    v = o < v ? o : v;
  }
  return v;
}

// Workgroup exclusive sum over one value per thread.  `lds` needs
// (THREADS/64 + 1) elements.  Returns the exclusive prefix; *total gets the
// workgroup sum (valid in every thread).  Contains two barriers.
template <typename T, int THREADS>
__device__ __forceinline__ T sbx_block_exclusive_sum(T v, T *lds, T *total) {
  constexpr int WAVES = THREADS / 64;
  const T inc = sbx_wave_inclusive_sum(v);
  const int w = sbx_wave_in_block();
  if (sbx_lane() == 63) lds[w] = inc;
  __syncthreads();
  T woff = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < WAVES; i++) {
    const T s = lds[i];
    if (i < w) woff += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return woff + inc - v;
}

template <typename T, int THREADS>
__device__ __forceinline__ T sbx_block_sum(T v, T *lds) {
  constexpr int WAVES = THREADS / 64;
  const T s = sbx_wave_sum(v);
  if (sbx_lane() == 0) lds[sbx_wave_in_block()] = s;
  __syncthreads();
  T tot = 0;
#pragma unroll
  for (int i = 0; i < WAVES; i++) tot += lds[i];
  __syncthreads();
  return tot;
}

// Wave-cooperative upper bound: first index f in [0,len) with arr[f] > target,
// or len.  All 64 lanes must call it with the same arguments; every round is
// one 64-ary probe, so 2^24 entries take 4 dependent loads instead of 24.
template <typename T>
__device__ __forceinline__ int64_t sbx_wave_upper_bound(const T *__restrict__ arr, int64_t len, T target) {
  int64_t lo = 0, hi = len;  // answer in [lo, hi]
  const int lane = sbx_lane();
  while (lo < hi) {
    const int64_t span = hi - lo;
    const int64_t step = (span + 63) / 64;
    const int64_t p = lo + (int64_t)lane * step;
    bool gt = true;  // probes past the end behave as +inf
    if (p < hi) gt = arr[p] > target;
    const uint64_t m = __ballot(gt);
    const int k = m ? __builtin_ctzll(m) : 64;  // first lane whose probe is > target
    if (k == 0) return lo;
    const int64_t pk = lo + (int64_t)k * step;  // may be >= hi
    lo = lo + (int64_t)(k - 1) * step + 1;
    if (pk < hi) hi = pk;
  }
  return lo;
}

// Wave-aggregated append: lanes with `want` get consecutive slots from *counter with a
// single returning atomic per wave (one hot word sustains only ~88 atomics/us on MI355X).
// All 64 lanes must call it.
__device__ __forceinline__ unsigned sbx_wave_append(unsigned *counter, bool want) {
  const uint64_t m = __ballot(want);
  if (!m) return 0;
  const int leader = __builtin_ctzll(m);
  unsigned base = 0;
  if (sbx_lane() == leader) base = atomicAdd(counter, (unsigned)__popcll(m));
  base = __shfl(base, leader, 64);
  return base + (unsigned)__popcll(m & sbx_lanemask_lt());
}
