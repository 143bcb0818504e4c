// sbx_countsort.h — a stable LSD counting sort of (32-bit key, 32-bit value) pairs for inputs of 10^4 .. 10^7 pairs,
// three launches per 9-bit digit and no chained scan:
//   k_cs_count   every workgroup counts the digits of its tile of 1024 pairs (ballot ranks, one LDS add per distinct
//                digit and wave) -> cnt[digit * n_tiles + tile]
//   k_cs_scan    one workgroup per digit: exclusive scan of the tiles' counts, total[digit]
//   k_cs_place   the tiles again: position = pairs of smaller digits + pairs of this digit in the tiles before + in
//                the tile's waves before + in this wave before; the last digit's placement hands (position, key, value)
//                to an emit functor instead of storing the pair.
// The generic radix sort (sbx_prims.hip: one-sweep passes with a decoupled look-back) is built for 10^8 records; on a
// level of 3 * 10^4 .. 10^6 vertices its chained scan over the tiles makes a digit pass cost 10 – 16 us whatever the
// size, where these three kernels take 3 – 8 us together (RCM's level ordering, DegreeReorder's last bucket).
#pragma once
#include "sbx_device.h"
#include "sbx_internal.h"
#include <utility>

namespace sbx_cs {
namespace {  // (kernels in a header shared by several translation units: internal linkage)

constexpr int ROUNDS = 4;                 // pairs per lane
constexpr int TILE = 256 * ROUNDS;        // pairs per workgroup
constexpr int64_t MAX_PAIRS = (int64_t)1 << 20;  // the callers' switch-over: the placement scatters single words, and beyond ~10^6
                                                 // pairs the generic sort's LDS-staged stores win (RCM's 1.7 M-vertex level: 161 vs 147 us)

// the lanes of the wave that hold the same BITS-bit digit as this one (valid lanes only)
template <int BITS>
__device__ __forceinline__ unsigned long long peers_of(unsigned d, bool valid) {
  unsigned long long peers = __ballot(valid);
#pragma unroll
  for (int b = 0; b < BITS; b++) {
    const bool bit = (d >> b) & 1u;
    const unsigned long long m = __ballot(bit);
    peers &= bit ? m : ~m;
  }
  return peers;
}

template <int BITS>
__global__ __launch_bounds__(256) void k_cs_count(const uint32_t *__restrict__ key, int64_t n, int64_t n_tiles, int shift,
                                                  unsigned mask, unsigned *__restrict__ cnt) {
  constexpr int BINS = 1 << BITS, PER = BINS / 256;  // digits per thread
  __shared__ unsigned s_hist[BINS];
#pragma unroll
  for (int k = 0; k < PER; k++) s_hist[threadIdx.x + 256 * k] = 0;
  const int64_t j0 = (int64_t)blockIdx.x * TILE + (threadIdx.x >> 6) * (64 * ROUNDS) + (threadIdx.x & 63);
  unsigned kk[ROUNDS];
#pragma unroll
  for (int r = 0; r < ROUNDS; r++) {
    const int64_t j = j0 + r * 64;
    kk[r] = key[j < n ? j : n - 1];
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < ROUNDS; r++) {
    const bool valid = j0 + r * 64 < n;
    const unsigned d = (kk[r] >> shift) & mask;
    const unsigned long long peers = peers_of<BITS>(d, valid);
    if (valid && (peers & sbx_lanemask_lt()) == 0) atomicAdd(&s_hist[d], (unsigned)__popcll(peers));
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < PER; k++) cnt[(int64_t)(threadIdx.x + 256 * k) * n_tiles + blockIdx.x] = s_hist[threadIdx.x + 256 * k];
}

// workgroup d: cnt[d * n_units + u] -> the number of pairs with digit d in the units before u; total[d]
__device__ __forceinline__ unsigned scan_digit_row(unsigned *__restrict__ row, int64_t n_units, unsigned *s_scan) {
  const bool vec = (n_units & 3) == 0;  // rows of counts start 16-byte aligned: four counts per access
  int64_t per = (n_units + 255) / 256;
  if (vec) per = (per + 3) & ~(int64_t)3;
  const int64_t lo = (int64_t)threadIdx.x * per < n_units ? (int64_t)threadIdx.x * per : n_units;
  const int64_t hi = lo + per < n_units ? lo + per : n_units;
  unsigned sum = 0, tot;
  if (vec) {
    for (int64_t i = lo; i < hi; i += 4) {
      const uint4 c = *(const uint4 *)(row + i);
      sum += c.x + c.y + c.z + c.w;
    }
    unsigned run = sbx_block_exclusive_sum<unsigned, 256>(sum, s_scan, &tot);
    for (int64_t i = lo; i < hi; i += 4) {
      const uint4 c = *(const uint4 *)(row + i);
      *(uint4 *)(row + i) = make_uint4(run, run + c.x, run + c.x + c.y, run + c.x + c.y + c.z);
      run += c.x + c.y + c.z + c.w;
    }
  } else {
    for (int64_t i = lo; i < hi; i++) sum += row[i];
    unsigned run = sbx_block_exclusive_sum<unsigned, 256>(sum, s_scan, &tot);
    for (int64_t i = lo; i < hi; i++) {
      const unsigned c = row[i];
      row[i] = run;
      run += c;
    }
  }
  return tot;
}

__global__ __launch_bounds__(256) void k_cs_scan(unsigned *__restrict__ cnt, int64_t n_tiles, unsigned *__restrict__ total) {
  __shared__ unsigned s_scan[8];
  const unsigned tot = scan_digit_row(cnt + (int64_t)blockIdx.x * n_tiles, n_tiles, s_scan);
  if (threadIdx.x == 0) total[blockIdx.x] = tot;
}

// Emit: struct with  __device__ void operator()(unsigned pos, uint32_t key, uint32_t val) const
template <int BITS, typename Emit>
__global__ __launch_bounds__(256) void k_cs_place(const uint32_t *__restrict__ key, const uint32_t *__restrict__ val,
                                                  int64_t n, int64_t n_tiles, const unsigned *__restrict__ off,
                                                  const unsigned *__restrict__ total, int shift, unsigned mask,
                                                  uint32_t *__restrict__ key_out, uint32_t *__restrict__ val_out, int last,
                                                  Emit emit) {
  constexpr int BINS = 1 << BITS, PER = BINS / 256;  // digits per thread: PER consecutive ones
  __shared__ unsigned s_scan[8];
  __shared__ unsigned s_wave[4][BINS];  // pairs of digit d in wave w; then the position of the wave's first such pair
  const int lane = sbx_lane(), wv = sbx_wave_in_block();
  const int64_t j0 = (int64_t)blockIdx.x * TILE + wv * (64 * ROUNDS) + lane;
  unsigned kk[ROUNDS], vv[ROUNDS];
#pragma unroll
  for (int r = 0; r < ROUNDS; r++) {
    const int64_t j = j0 + r * 64;
    kk[r] = key[j < n ? j : n - 1];
    vv[r] = val[j < n ? j : n - 1];
  }
  // thread t holds digits PER * t .. PER * t + PER - 1: their totals, and this tile's offsets inside them
  unsigned tt[PER], oo[PER], tsum = 0;
#pragma unroll
  for (int k = 0; k < PER; k++) {
    tt[k] = total[PER * threadIdx.x + k];
    oo[k] = off[(int64_t)(PER * threadIdx.x + k) * n_tiles + blockIdx.x];
    tsum += tt[k];
  }
#pragma unroll
  for (int w = 0; w < 4; w++)
#pragma unroll
    for (int k = 0; k < PER; k++) s_wave[w][PER * threadIdx.x + k] = 0;
  unsigned tot;
  const unsigned dbase = sbx_block_exclusive_sum<unsigned, 256>(tsum, s_scan, &tot);  // (two barriers)
  unsigned loc[ROUNDS];  // the pair's rank among the wave's pairs of its digit
#pragma unroll
  for (int r = 0; r < ROUNDS; r++) {
    const bool valid = j0 + r * 64 < n;
    const unsigned d = (kk[r] >> shift) & mask;
    const unsigned long long peers = peers_of<BITS>(d, valid);
    const unsigned rank = (unsigned)__popcll(peers & sbx_lanemask_lt());
    unsigned first = 0;
    if (valid && rank == 0) first = atomicAdd(&s_wave[wv][d], (unsigned)__popcll(peers));
    first = (unsigned)__shfl((int)first, valid ? (int)__builtin_ctzll(peers) : 0, 64);
    loc[r] = first + rank;
  }
  __syncthreads();
  {
    unsigned before = dbase;  // pairs of smaller digits
#pragma unroll
    for (int k = 0; k < PER; k++) {
      unsigned run = before + oo[k];
#pragma unroll
      for (int w = 0; w < 4; w++) {
        const unsigned c = s_wave[w][PER * threadIdx.x + k];
        s_wave[w][PER * threadIdx.x + k] = run;
        run += c;
      }
      before += tt[k];
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < ROUNDS; r++) {
    if (j0 + r * 64 < n) {
      const unsigned d = (kk[r] >> shift) & mask;
      const unsigned pos = s_wave[wv][d] + loc[r];
      if (last) {
        emit(pos, kk[r], vv[r]);
      } else {
        key_out[pos] = kk[r];
        val_out[pos] = vv[r];
      }
    }
  }
}

// Sorts the n pairs (ka, va) by the low `bits` bits of the key (stable), kb / vb as the other side of the ping-pong;
// the last digit's placement calls emit(position, key, value) for every pair instead of storing it.  Digits of up to
// 9 bits.  Scratch: `scratch` (scratch_words(n) words, for callers that sort many times per call) or, when null, from
// the arena.  n >= 1 (any n works; callers stay below MAX_PAIRS).
static inline size_t scratch_words(int64_t n) { return (size_t)((n + TILE - 1) / TILE) * 512 + 512; }
template <int BITS, typename Emit>
static int digit_pass(sbx_handle_t h, int kid, const uint32_t *ka, const uint32_t *va, uint32_t *kb, uint32_t *vb, int64_t n,
                      int shift, int width, int last, unsigned *scratch, Emit emit) {
  constexpr int BINS = 1 << BITS;
  const int64_t n_tiles = (n + TILE - 1) / TILE;
  unsigned *total = scratch, *cnt = scratch + BINS;
  const unsigned mask = (1u << width) - 1u;
  SBX_KLAUNCH(h, kid, k_cs_count<BITS>, dim3((unsigned)n_tiles), dim3(256), ka, n, n_tiles, shift, mask, cnt);
  SBX_KLAUNCH(h, kid, k_cs_scan, dim3(BINS), dim3(256), cnt, n_tiles, total);
  SBX_KLAUNCH(h, kid, (k_cs_place<BITS, Emit>), dim3((unsigned)n_tiles), dim3(256), ka, va, n, n_tiles, (const unsigned *)cnt,
              (const unsigned *)total, shift, mask, kb, vb, last, emit);
  return SBX_OK;
}
template <typename Emit>
static int sort_emit(sbx_handle_t h, int kid, uint32_t *ka, uint32_t *va, uint32_t *kb, uint32_t *vb, int64_t n, int bits,
                     Emit emit, unsigned *scratch = nullptr) {
  if (!scratch) SBX_TRY(sbx_salloc(h, scratch_words(n), &scratch));
  if (bits < 1) bits = 1;  // (all keys equal: one pass that moves nothing but runs the emit)
  const int np = (bits + 8) / 9;  // (digits of 11 bits would save one at 19 - 22 bits, but the kernels slow down by more: measured)
  for (int p = 0, shift = 0; p < np; p++) {
    const int width = (bits - shift + (np - p) - 1) / (np - p);  // the remaining bits in equal shares
    const int last = p == np - 1 ? 1 : 0;
    SBX_TRY((digit_pass<9, Emit>(h, kid, ka, va, kb, vb, n, shift, width, last, scratch, emit)));
    std::swap(ka, kb);
    std::swap(va, vb);
    shift += width;
  }
  SBX_LAUNCH_CHECK(h);
  return SBX_OK;
}

}  // namespace
}  // namespace sbx_cs
