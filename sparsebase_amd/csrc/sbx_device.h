// sbx_device.h — wave64 / workgroup device helpers (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <limits>

#define SBX_WAVE 64

__device__ __forceinline__ int sbx_lane() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ int sbx_wave_in_block() { return (int)(threadIdx.x >> 6); }

__device__ __forceinline__ uint64_t sbx_lanemask_lt() {
  return ((uint64_t)1 << sbx_lane()) - 1;
}

// ---- wave64 scans and reductions ---------------------------------------------------------------------------
// 32-bit integers go through DPP (data-parallel primitives: a VALU operand read from another lane of the same
// 16-lane row, plus the two row broadcasts gfx9 has): 7 steps for a scan, 4 + four v_readlane for a reduction, no
// LDS traffic.  __shfl* compiles to ds_bpermute_b32 — an LDS-crossbar operation with ~5 VALU instructions of
// address arithmetic per step, ~45 instructions and 6 LDS operations per scan — and stays for 64-bit types only.
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ int sbx_dpp(int old, int src) {
  // lanes switched off by the masks, and lanes whose source lane lies outside the row, get `old`
  return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, BANK_MASK, false);
}
constexpr int SBX_DPP_ROW_SHR = 0x110;       // + n: lane i reads lane i - n of its row
constexpr int SBX_DPP_WAVE_SHR1 = 0x138;     // lane i reads lane i - 1 of the wave
constexpr int SBX_DPP_ROW_MIRROR = 0x140, SBX_DPP_ROW_HALF_MIRROR = 0x141;
constexpr int SBX_DPP_ROW_BCAST15 = 0x142, SBX_DPP_ROW_BCAST31 = 0x143;

template <typename T>
struct sbx_is_dpp {
  static constexpr bool value = sizeof(T) == 4 && (T)0.5 == (T)0;  // 32-bit integer
};

struct SbxOpSum {
  template <typename T> __device__ __forceinline__ T operator()(T a, T b) const { return a + b; }
};
struct SbxOpMax {
  template <typename T> __device__ __forceinline__ T operator()(T a, T b) const { return a > b ? a : b; }
};
struct SbxOpMin {
  template <typename T> __device__ __forceinline__ T operator()(T a, T b) const { return a < b ? a : b; }
};

template <typename T, typename Op>
__device__ __forceinline__ T sbx_dpp_inclusive(T v, T ident, Op op) {
  const int id = (int)ident, s = (int)v;
  T x = v;
  x = op(x, (T)sbx_dpp<SBX_DPP_ROW_SHR + 1>(id, s));
  x = op(x, (T)sbx_dpp<SBX_DPP_ROW_SHR + 2>(id, s));
  x = op(x, (T)sbx_dpp<SBX_DPP_ROW_SHR + 3>(id, s));                    // lane i: lanes i-3..i of its row
  x = op(x, (T)sbx_dpp<SBX_DPP_ROW_SHR + 4, 0xF, 0xE>(id, (int)x));     // lanes 4..15: i-7..i
  x = op(x, (T)sbx_dpp<SBX_DPP_ROW_SHR + 8, 0xF, 0xC>(id, (int)x));     // lanes 8..15: the whole row prefix
  x = op(x, (T)sbx_dpp<SBX_DPP_ROW_BCAST15, 0xA, 0xF>(id, (int)x));     // rows 1, 3 take the total of the row before
  x = op(x, (T)sbx_dpp<SBX_DPP_ROW_BCAST31, 0xC, 0xF>(id, (int)x));     // rows 2, 3 take the total of rows 0..1
  return x;
}

template <typename T, typename Op>
__device__ __forceinline__ T sbx_dpp_reduce(T v, Op op) {  // every lane gets the result
  T x = v;
  // (full permutations of a row: no lane is without a source, so there is no `old` to keep)
  x = op(x, (T)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, false));  // quad_perm [1,0,3,2]
  x = op(x, (T)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xF, 0xF, false));  // quad_perm [2,3,0,1]
  x = op(x, (T)__builtin_amdgcn_mov_dpp((int)x, SBX_DPP_ROW_HALF_MIRROR, 0xF, 0xF, false));  // quads 0<->1, 2<->3
  x = op(x, (T)__builtin_amdgcn_mov_dpp((int)x, SBX_DPP_ROW_MIRROR, 0xF, 0xF, false));  // halves: every row is uniform now
  const T a = (T)__builtin_amdgcn_readlane((int)x, 0), b = (T)__builtin_amdgcn_readlane((int)x, 16);
  const T c = (T)__builtin_amdgcn_readlane((int)x, 32), d = (T)__builtin_amdgcn_readlane((int)x, 48);
  return op(op(a, b), op(c, d));
}

// reduction inside every 16-lane row (all 16 lanes get their row's result): four DPP steps
template <typename T, typename Op>
__device__ __forceinline__ T sbx_row16_reduce(T v, Op op) {
  static_assert(sbx_is_dpp<T>::value, "32-bit integers only");
  T x = v;
  x = op(x, (T)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, false));
  x = op(x, (T)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xF, 0xF, false));
  x = op(x, (T)__builtin_amdgcn_mov_dpp((int)x, SBX_DPP_ROW_HALF_MIRROR, 0xF, 0xF, false));
  x = op(x, (T)__builtin_amdgcn_mov_dpp((int)x, SBX_DPP_ROW_MIRROR, 0xF, 0xF, false));
  return x;
}

// lane i gets lane i - 1's value, lane 0 gets `first`
template <typename T>
__device__ __forceinline__ T sbx_wave_shift_up1(T v, T first) {
  if constexpr (sbx_is_dpp<T>::value) {
    return (T)sbx_dpp<SBX_DPP_WAVE_SHR1>((int)first, (int)v);
  } else {
    const T o = __shfl_up(v, 1, 64);
    return sbx_lane() == 0 ? first : o;
  }
}

// inclusive wave scan (sum)
template <typename T>
__device__ __forceinline__ T sbx_wave_inclusive_sum(T v) {
  if constexpr (sbx_is_dpp<T>::value) {
    return sbx_dpp_inclusive(v, (T)0, SbxOpSum());
  } else {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      T o = __shfl_up(v, d, 64);
      if (sbx_lane() >= d) v += o;
    }
    return v;
  }
}

// inclusive wave scan (max)
template <typename T>
__device__ __forceinline__ T sbx_wave_inclusive_max(T v) {
  if constexpr (sbx_is_dpp<T>::value) {
    return sbx_dpp_inclusive(v, std::numeric_limits<T>::lowest(), SbxOpMax());
  } else {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      T o = __shfl_up(v, d, 64);
      if (sbx_lane() >= d) v = o > v ? o : v;
    }
    return v;
  }
}

template <typename T>
__device__ __forceinline__ T sbx_wave_sum(T v) {
  if constexpr (sbx_is_dpp<T>::value) {
    return sbx_dpp_reduce(v, SbxOpSum());
  } else {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
  }
}

template <typename T>
__device__ __forceinline__ T sbx_wave_max(T v) {
  if constexpr (sbx_is_dpp<T>::value) {
    return sbx_dpp_reduce(v, SbxOpMax());
  } else {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      T o = __shfl_xor(v, d, 64);
      v = o > v ? o : v;
    }
    return v;
  }
}

template <typename T>
__device__ __forceinline__ T sbx_wave_min(T v) {
  if constexpr (sbx_is_dpp<T>::value) {
    return sbx_dpp_reduce(v, SbxOpMin());
  } else {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      T o = __shfl_xor(v, d, 64);
      v = o < v ? o : v;
    }
    return v;
  }
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
  base = (unsigned)__builtin_amdgcn_readlane((int)base, leader);
  return base + (unsigned)__popcll(m & sbx_lanemask_lt());
}
