// sbx_gray.hip — device stage of GrayReorder::GrayReorderingCSR
// (reorder/gray_reorder.cc:106-424): everything that touches the nonzeros.
//
// Per row i the kernel produces
//   degree_out[i] = row_ptr[i+1]-row_ptr[i]                                   (:150)
//   key_out[i]    = Gray-decoded (prefix-xor, :38-46) occupancy bitmap over
//                   `resolution` column blocks of width m/resolution (:210,:249-267);
//                   bit b is set iff count_b > thr, thr = 0 for sparse rows and
//                   deg/resolution for dense rows (:384-395)
// and the four band counters of :138-170 (nonzeros within m/128 of the diagonal,
// split by sparse/dense class).  One O(nnz) pass, 4 B/nonzero + 16 B/row of traffic.
// The ordering stage (std::sort calls whose tie order is libstdc++-specific) stays
// above the ABI in the host layer — see DESIGN.md "Gray".
#include "sbx_device.h"
#include "sbx_internal.h"

namespace {

__device__ __forceinline__ uint64_t gray_decode(uint64_t g) {
  // prefix xor from the top: b = g ^ g>>1 ^ g>>2 ...
  g ^= g >> 1; g ^= g >> 2; g ^= g >> 4; g ^= g >> 8; g ^= g >> 16; g ^= g >> 32;
  return g;
}

struct GrayCounts {
  unsigned long long nnz_sparse, diag_sparse, nnz_dense, diag_dense;
};

// The whole stage is nonzero-parallel, so it is balanced under any degree distribution and
// every col load is a coalesced 32 B/lane read: a workgroup owns GT_TILE consecutive
// nonzeros, finds the rows under them (row heads scattered into LDS from a per-tile row
// table), and reduces per row in LDS.  A row owns the LDS words under its own nonzeros:
// two of them hold the 64-bit OR of column-block bits when its threshold deg/resolution
// is 0, `resolution` of them hold per-block counts otherwise (then deg >= resolution, so
// the words exist).  Rows cut by a tile boundary accumulate in a global slot indexed by
// the tile they start in (at most one such row per tile) and are finished by k_gray_finish.
constexpr int GT_THREADS = 256;
constexpr int GT_ITEMS = 8;
constexpr int GT_TILE = GT_THREADS * GT_ITEMS;  // nonzeros per workgroup
constexpr int GT_SLACK = 64;                    // words past the tile end / for the row entering from the left

__device__ __forceinline__ bool gray_counted(int64_t d, int bits, int nnz_threshold) {
  return d > nnz_threshold && d >= bits;  // thr = d / bits > 0
}

// degree_out, keys of empty rows, and tile_row[t] = last row r with row_ptr[r] <= min(t*GT_TILE, nnz)
__global__ __launch_bounds__(256) void k_gray_prep(const int32_t *__restrict__ rp, int64_t n, int64_t nnz,
                                                   int64_t ntiles, int32_t *__restrict__ degree_out,
                                                   unsigned long long *__restrict__ key_out,
                                                   int32_t *__restrict__ tile_row) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t total = n > ntiles + 1 ? n : ntiles + 1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    if (i < n) {
      const int32_t d = rp[i + 1] - rp[i];
      degree_out[i] = d;
      if (d == 0) key_out[i] = 0;
    }
    if (i <= ntiles) {
      const int64_t pos = i * GT_TILE < nnz ? i * GT_TILE : nnz;
      int64_t lo = 0, hi = n;  // last r in [0,n] with rp[r] <= pos
      while (lo < hi) {
        const int64_t mid = (lo + hi + 1) >> 1;
        if ((int64_t)rp[mid] <= pos) lo = mid; else hi = mid - 1;
      }
      tile_row[i] = (int32_t)lo;
    }
  }
}

__global__ __launch_bounds__(GT_THREADS) void k_gray_tile(const int32_t *__restrict__ rp,
                                                          const int32_t *__restrict__ col, int64_t nnz,
                                                          const int32_t *__restrict__ tile_row, uint32_t width,
                                                          uint32_t magic, uint32_t band, int bits, int nnz_threshold,
                                                          unsigned long long *__restrict__ key_out,
                                                          int32_t *__restrict__ fix_row,
                                                          unsigned *__restrict__ slot_acc,
                                                          unsigned *__restrict__ tile_counts) {
  __shared__ unsigned long long s_head[GT_TILE];   // (row - r_lo) << 32 | degree at the row's first nonzero
  __shared__ unsigned s_acc[GT_TILE + 2 * GT_SLACK];
  __shared__ int s_wmax[GT_THREADS / 64];
  __shared__ unsigned long long s_red[GT_THREADS / 64][4];
  const int tid = threadIdx.x;
  const int64_t t0 = (int64_t)blockIdx.x * GT_TILE;
  const int64_t t1 = (t0 + GT_TILE < nnz) ? t0 + GT_TILE : nnz;
  const int cnt = (int)(t1 - t0);
#pragma unroll
  for (int k = 0; k < GT_ITEMS; k++) {
    s_head[k * GT_THREADS + tid] = 0;
    s_acc[k * GT_THREADS + tid] = 0;
  }
  if (tid < 2 * GT_SLACK) s_acc[GT_TILE + tid] = 0;
  const int64_t r_lo = tile_row[blockIdx.x], r_end = tile_row[blockIdx.x + 1];
  const int64_t lo_start = rp[r_lo];
  const int lo_deg = (int)(rp[r_lo + 1] - lo_start);
  // this thread's nonzeros: GT_ITEMS consecutive ones
  const int p0 = tid * GT_ITEMS;
  int32_t c[GT_ITEMS];
  if (p0 + GT_ITEMS <= cnt && ((uintptr_t)col & 15) == 0) {
    const int4 *src = (const int4 *)(col + t0 + p0);
    const int4 a = src[0], b = src[1];
    c[0] = a.x; c[1] = a.y; c[2] = a.z; c[3] = a.w; c[4] = b.x; c[5] = b.y; c[6] = b.z; c[7] = b.w;
  } else {
#pragma unroll
    for (int k = 0; k < GT_ITEMS; k++) c[k] = (p0 + k < cnt) ? col[t0 + p0 + k] : 0;
  }
  __syncthreads();
  if (r_end - r_lo <= 4 * GT_TILE) {
    for (int64_t r = r_lo + 1 + tid; r <= r_end; r += GT_THREADS) {
      const int64_t s = rp[r];
      const int64_t p = s - t0;
      if (p < cnt)  // empty rows share the position of the next non-empty one, which has the largest id
        atomicMax(&s_head[p], ((unsigned long long)(r - r_lo) << 32) | (unsigned)(rp[r + 1] - s));
    }
  } else {
    // mostly empty rows: a position is a head iff the last row with row_ptr <= pos starts exactly there
    for (int k = 0; k < GT_ITEMS; k++) {
      const int p = p0 + k;
      if (p >= cnt || p == 0) continue;
      int64_t lo = r_lo, hi = r_end;
      while (lo < hi) {
        const int64_t mid = (lo + hi + 1) >> 1;
        if ((int64_t)rp[mid] <= t0 + p) lo = mid; else hi = mid - 1;
      }
      if ((int64_t)rp[lo] == t0 + p && lo > r_lo)
        s_head[p] = ((unsigned long long)(lo - r_lo) << 32) | (unsigned)(rp[lo + 1] - rp[lo]);
    }
  }
  __syncthreads();
  // which row is open when this thread starts: position+1 of the last head before p0 (0: r_lo)
  unsigned long long hd[GT_ITEMS];
  int last = 0;
  unsigned headmask = 0;
#pragma unroll
  for (int k = 0; k < GT_ITEMS; k++) {
    hd[k] = s_head[p0 + k];
    if (hd[k]) {
      last = p0 + k + 1;
      headmask |= 1u << k;
    }
  }
  const int inc = sbx_wave_inclusive_max(last);
  int open = sbx_wave_shift_up1(inc, 0);
  if (sbx_lane() == 63) s_wmax[tid >> 6] = inc;
  __syncthreads();
  for (int w = 0; w < (tid >> 6); w++) open = s_wmax[w] > open ? s_wmax[w] : open;
  int row = (int)r_lo;
  int d = lo_deg, base = GT_TILE + GT_SLACK;
  if (open) {
    const unsigned long long hv = s_head[open - 1];
    row = (int)r_lo + (int)(hv >> 32);
    d = (int)(unsigned)hv;
    base = open - 1;
  }
  // The loop is written select-style (the 8 steps are unrolled and every branch that is
  // left costs the whole wave): ~20 VALU ops per nonzero is what 8 TB/s allows.
  unsigned long long acc = 0;
  unsigned ns = 0, ds = 0, dd = 0, nvalid = 0;
  unsigned run = 0, prev = 0xFFFFFFFFu;
#pragma unroll
  for (int k = 0; k < GT_ITEMS; k++) {
    if (p0 + k >= cnt) break;
    nvalid++;
    const bool hk = hd[k] != 0;
    if (hk) {
      row = (int)r_lo + (int)(hd[k] >> 32);
      d = (int)(unsigned)hd[k];
      base = p0 + k;
    }
    const bool counted = d > nnz_threshold && d >= bits;
    const bool closes = (k == GT_ITEMS - 1) || (p0 + k + 1 >= cnt) || (k + 1 < GT_ITEMS && hd[k + 1 < GT_ITEMS ? k + 1 : k] != 0);
    unsigned bkt = __umulhi((unsigned)c[k], magic);  // c / width, one short at most
    bkt += ((unsigned)c[k] - bkt * width) >= width;
    const int diff = c[k] - row;
    const bool inb = (unsigned)(diff < 0 ? -diff : diff) <= band;
    const bool sparse = d <= nnz_threshold;
    ns += sparse;
    ds += sparse && inb;
    dd += !sparse && inb;
    const unsigned long long bit = 1ull << bkt;
    acc = hk ? bit : (acc | bit);
    const bool same = !hk && bkt == prev;
    run = same ? run + 1 : 1;
    prev = bkt;
    if (counted) {
      // sorted rows give long runs of one block: one LDS atomic per run
      bool flush = closes;
      if (!closes) {
        unsigned nb = __umulhi((unsigned)c[k + 1 < GT_ITEMS ? k + 1 : k], magic);
        nb += ((unsigned)c[k + 1 < GT_ITEMS ? k + 1 : k] - nb * width) >= width;
        flush = nb != bkt;
      }
      if (flush) {
        atomicAdd(&s_acc[base + bkt], run);
        prev = 0xFFFFFFFFu;
      }
    } else if (closes) {
      if (d == 1) key_out[row] = gray_decode(acc);
      else {
        atomicOr(&s_acc[base], (unsigned)acc);
        if (bits > 32) atomicOr(&s_acc[base + 1], (unsigned)(acc >> 32));
      }
    }
  }
  unsigned long long c_ns = ns, c_ds = ds, c_nd = nvalid - ns, c_dd = dd;
  __syncthreads();
  // finish the rows whose head this thread owns; thread 0 also the row entering from the left
  auto finish_row = [&](int64_t r, int rd, int rb, bool starts, bool ends) {
    const bool rc = gray_counted(rd, bits, nnz_threshold);
    if (!rc && rd == 1) return;  // written where its nonzero was read
    if (starts && ends) {
      unsigned long long key = 0;
      if (rc) {
        const unsigned thr = (unsigned)(rd / bits);
        for (int b = 0; b < bits; b++) key |= (unsigned long long)(s_acc[rb + b] > thr) << b;
      } else {
        key = (unsigned long long)s_acc[rb] | ((unsigned long long)s_acc[rb + 1] << 32);
      }
      key_out[r] = gray_decode(key);
    } else {
      unsigned *slot = slot_acc + (size_t)(starts ? (int64_t)blockIdx.x : lo_start / GT_TILE) * 64;
      if (rc) {
        for (int b = 0; b < bits; b++) {
          const unsigned v = s_acc[rb + b];
          if (v) atomicAdd(&slot[b], v);
        }
      } else {
        if (s_acc[rb]) atomicOr(&slot[0], s_acc[rb]);
        if (s_acc[rb + 1]) atomicOr(&slot[1], s_acc[rb + 1]);
      }
      if (starts) fix_row[blockIdx.x] = (int32_t)r;
    }
  };
  if (tid == 0) finish_row(r_lo, lo_deg, GT_TILE + GT_SLACK, lo_start >= t0, lo_start + lo_deg <= t1);
  while (headmask) {
    const int k = __builtin_ctz(headmask);
    headmask &= headmask - 1;
    const unsigned long long hv = s_head[p0 + k];
    finish_row(r_lo + (int64_t)(hv >> 32), (int)(unsigned)hv, p0 + k, true, p0 + k + (int)(unsigned)hv <= cnt);
  }
  c_ns = sbx_wave_sum(c_ns); c_ds = sbx_wave_sum(c_ds); c_nd = sbx_wave_sum(c_nd); c_dd = sbx_wave_sum(c_dd);
  if (sbx_lane() == 0) {
    s_red[tid >> 6][0] = c_ns; s_red[tid >> 6][1] = c_ds; s_red[tid >> 6][2] = c_nd; s_red[tid >> 6][3] = c_dd;
  }
  __syncthreads();
  // four hot counter words would serialize ~50 K workgroups: per-tile partials, summed by k_gray_finish
  if (tid < 4) {
    unsigned t = 0;
    for (int w = 0; w < GT_THREADS / 64; w++) t += (unsigned)s_red[w][tid];
    tile_counts[(size_t)blockIdx.x * 4 + tid] = t;
  }
}

// ---- short-row fast path ---------------------------------------------------------------------------------------
// When no row is longer than GR_SHORT_MAX (banded / mesh matrices: BASELINE config 5) the LDS machinery above is
// overhead.  GR_LPR lanes share a row (a wave = 64 / GR_LPR consecutive rows: their nonzeros are one contiguous stretch
// of col[], so the wave's loads cover whole lines between them), no LDS, no barrier.  Per-block counts are kept BIT-SLICED
// and saturating: ge[t] has bit b set iff block b was met more than t times; meeting block b again is
// ge[t+1] |= ge[t] & x from the top down.  The reference's key bit (count_b > thr, thr = 0 for rows below the
// threshold, deg / resolution above: gray_reorder.cc:249-267,384-395) is then bit b of ge[thr].  Lanes of a row are
// merged by a saturating add of the sliced counters over log2(GR_LPR) butterfly steps.  LV = levels needed = largest thr + 1.
constexpr int GR_SHORT_MAX = 64;  // thr <= 64 / 16 = 4: five levels at most
#ifndef GR_LPR_V
#define GR_LPR_V 4
#define GR_VEC_V 2
#endif
// The kernel is VALU-bound (87 % VALU busy, tools/pmc_gray.sh): what is done once per wave step — key selection, Gray
// decode, thresholds, stores, loop control — is shared by 64 / GR_LPR rows, and a row's lanes merge in log2(GR_LPR)
// steps.  Measured at C5 (kernel only): 8 lanes x 1 vector 0.139 ms, 4 x 2 0.107, 2 x 4 0.105, 1 x 8 0.137 (64 VGPRs of
// prefetched entries).  -DGR_LPR_V / -DGR_VEC_V build the variants.
constexpr int GR_LPR = GR_LPR_V;      // lanes per row
constexpr int GR_VEC = GR_VEC_V;      // 16-byte loads in flight per lane and batch
constexpr int GR_BATCH = 4 * GR_VEC;  // entries per lane and batch: rows of up to GR_LPR * GR_BATCH entries take one batch
constexpr int GR_LONG_LIST = 4096;  // rows above GR_SHORT_MAX entries the short-row path lists for k_gray_long_rows

// Exchange with lane ^ m inside a row's GR_LPR lanes.  m is a constant after unrolling: 1 and 2 are DPP quad permutations;
// 4 is row_half_mirror (lane i <-> 7 - i of the 8) — the same partner QUAD, and every value exchanged here is
// uniform inside a quad by then (both partners of a merge step compute the same symmetric result).  A DPP move is one
// VALU instruction; __shfl_xor is a ds_bpermute_b32 plus five instructions of address arithmetic.
__device__ __forceinline__ unsigned gr_xor32(unsigned v, int m) {
  switch (m) {
    case 1: return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, false);
    case 2: return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, false);
    case 4: return (unsigned)__builtin_amdgcn_mov_dpp((int)v, SBX_DPP_ROW_HALF_MIRROR, 0xF, 0xF, false);
    default: return __shfl_xor(v, m, 64);
  }
}
template <typename B>
__device__ __forceinline__ B gr_shfl_xor(B v, int m) {
  if constexpr (sizeof(B) == 8) {
    const unsigned lo = gr_xor32((unsigned)v, m), hi = gr_xor32((unsigned)(v >> 32), m);
    return ((B)hi << 32) | lo;
  } else {
    return (B)gr_xor32((unsigned)v, m);
  }
}

// WSHIFT >= 0: the block width is 2^WSHIFT (the usual case: power-of-two dimensions), block = column >> WSHIFT
// A lane's GR_BATCH = 4 CONSECUTIVE entries [j, j + 4) of its row in one 16-byte load (the address is only 4-byte
// aligned: gfx950 loads unaligned vectors): the lanes of a row fetch one contiguous 128-byte stretch per batch, where
// 4-byte loads at a stride of 8 entries made every load instruction of a wave touch 8 separate 32-byte pieces.  The
// per-row results are sums and ORs over the entries, so which lane sees which entry does not matter.  Entries at or
// past `e` read as `far`; the vector form is not used where it would run past the end of the array.
struct __attribute__((packed, aligned(4))) GrU4 {
  unsigned x, y, z, w;
};
__device__ __forceinline__ int32_t gr_clamp4(int32_t j, int32_t nnz) {  // where the 16 bytes for entry j are read
  return j < nnz - 4 ? j : nnz - 4;  // nnz >= 4 (the caller's business) and j >= 0: never negative
}
// Issued whatever the lane holds (a load under a condition makes the compiler wait for it at the join, which would
// serialise the prefetch): the address is clamped into the array; which of the four words are entries of the lane's
// row, and the shift the clamp caused for the last three entries of the whole array, are sorted out where the words
// are used (gr_take4).
__device__ __forceinline__ void gr_load4(const int32_t *__restrict__ col, int32_t j, int32_t nnz, unsigned *c) {
#pragma unroll
  for (int v4 = 0; v4 < GR_VEC; v4++) {
#if defined(GR_ABLATE) && GR_ABLATE == 4
    const GrU4 v = {(unsigned)j, (unsigned)j + 1u, (unsigned)j + 2u, (unsigned)j + 3u};
#else
    const GrU4 v = *(const GrU4 *)(col + gr_clamp4(j + 4 * v4, nnz));
#endif
    c[4 * v4] = v.x, c[4 * v4 + 1] = v.y, c[4 * v4 + 2] = v.z, c[4 * v4 + 3] = v.w;
  }
}
// c[u] := entry j + u of the array for u < cnt (the lane's share of its row); returns cnt = min(GR_BATCH, e - j)
// clamped at 0
__device__ __forceinline__ int gr_take4(int32_t j, int32_t e, int32_t nnz, unsigned *c) {
  const int left = e - j;
  const int cnt = left < 0 ? 0 : (left > GR_BATCH ? GR_BATCH : left);
  if (__any(j + GR_BATCH - 4 > nnz - 4 && cnt > 0)) {  // the end of the array: a vector was read up to 3 entries early
#pragma unroll
    for (int v4 = 0; v4 < GR_VEC; v4++) {
      const int32_t jv = j + 4 * v4;
      const int sh = jv - gr_clamp4(jv, nnz);  // (a vector wholly past the end holds no entry of the row: any value)
      const unsigned w[4] = {c[4 * v4], c[4 * v4 + 1], c[4 * v4 + 2], c[4 * v4 + 3]};
#pragma unroll
      for (int u = 0; u < 4; u++) {
        unsigned x = w[u];
#pragma unroll
        for (int k = u + 1; k < 4; k++) x = (sh == k - u) ? w[k] : x;
        c[4 * v4 + u] = x;
      }
    }
  }
  return cnt;
}

template <typename B, int LV, bool POW2>
__global__ __launch_bounds__(256) void k_gray_rows_short(const int32_t *__restrict__ rp, const int32_t *__restrict__ col,
                                                         int64_t n, uint32_t width, uint32_t magic, uint32_t band,
                                                         int wshift, int bits, int nnz_threshold,
                                                         int32_t *__restrict__ degree_out,
                                                         unsigned long long *__restrict__ key_out,
                                                         GrayCounts *__restrict__ counts, unsigned *nlong,
                                                         int32_t *__restrict__ long_list) {
  __shared__ unsigned long long s_red[4][4];
  const int tid = threadIdx.x, sub = tid & (GR_LPR - 1);
  const int64_t rows_per_block = 256 / GR_LPR;
  unsigned long long c_ns = 0, c_ds = 0, c_nd = 0, c_dd = 0;
  const int64_t step = (int64_t)gridDim.x * rows_per_block;
  int64_t row = (int64_t)blockIdx.x * rows_per_block + tid / GR_LPR;
  // software pipeline: the bounds of the row two steps ahead and the first batch of entries of the row one step ahead
  // are in flight while this step's row is processed (the kernel is bound by the rp -> col load chain, not by work)
  int32_t s_cur = 0, e_cur = 0, s_nx = 0, e_nx = 0;
  if (row < n) s_cur = rp[row], e_cur = rp[row + 1];
  if (row + step < n) s_nx = rp[row + step], e_nx = rp[row + step + 1];
  const int32_t nnz = rp[n];
  unsigned c_cur[GR_BATCH];
  gr_load4(col, s_cur + GR_BATCH * sub, nnz, c_cur);
  for (; row < n; row += step) {
    const int32_t s = s_cur;
    int32_t e = e_cur;
    // next step: its bounds are here (loaded a step ago), its first batch is issued now; the step after: its bounds
    s_cur = s_nx, e_cur = e_nx;
    if (row + 2 * step < n) s_nx = rp[row + 2 * step], e_nx = rp[row + 2 * step + 1];
    unsigned c_nx[GR_BATCH];
    gr_load4(col, s_cur + GR_BATCH * sub, nnz, c_nx);
    // The kernel runs before anyone knows whether the matrix suits it: rows above GR_SHORT_MAX entries are listed for
    // k_gray_long_rows as they are met, and once there are more of them than the list holds (a power-law matrix) a wave
    // leaves at its next long row — the host then discards the results and takes the tile kernel.
    const bool long_row = e - s > GR_SHORT_MAX;  // k_gray_long_rows' business
    if (__any(long_row)) {  // (nothing on the common path: the counter is only touched by waves that meet a long row)
      const unsigned slot = sbx_wave_append(nlong, long_row && sub == 0);
      if (long_row && sub == 0 && slot < (unsigned)GR_LONG_LIST) long_list[slot] = (int32_t)row;
      if (__any(long_row && sub == 0 && slot >= (unsigned)GR_LONG_LIST)) break;  // list full: this wave stops (it still meets
                                                                                 // the workgroup's barrier below; the host discards the results)
    }
    if (long_row) e = s;
    const int d = e - s;
    const bool sparse = d <= nnz_threshold;
    // d / bits without the division (~35 VALU operations for a runtime divisor; the kernel is VALU-bound): a row of
    // this kernel has at most GR_SHORT_MAX entries and LV - 1 >= GR_SHORT_MAX / bits
    unsigned thr = 0;
#pragma unroll
    for (int t = 1; t < LV; t++) thr += d >= t * bits;
    thr = d > nnz_threshold ? thr : 0u;
    B ge[LV];
#pragma unroll
    for (int t = 0; t < LV; t++) ge[t] = 0;
    unsigned inb = 0;
    // the row's entries, GR_BATCH consecutive ones per lane and batch; the first batch was prefetched, longer rows
    // load further ones here
    const unsigned row_lo = (unsigned)row - band, band2 = 2u * band;  // |c - row| <= band  <=>  c - row_lo <= 2 band (mod 2^32)
    bool first = true;
    for (int32_t j0 = s; first || __any(j0 < e); j0 += GR_BATCH * GR_LPR) {
      unsigned c[GR_BATCH];
      const int32_t j = j0 + GR_BATCH * sub;
      if (first) {
#pragma unroll
        for (int u = 0; u < GR_BATCH; u++) c[u] = c_cur[u];
      } else {
        gr_load4(col, j, nnz, c);
      }
      first = false;
      const int cnt = gr_take4(j, e, nnz, c);
#pragma unroll
      for (int u = 0; u < GR_BATCH; u++) {
        unsigned bkt;
        if (POW2) {
          bkt = c[u] >> wshift;
        } else {
          bkt = __umulhi(c[u], magic);  // c / width, one short at most
          bkt += (c[u] - bkt * width) >= width;
        }
        const bool have = u < cnt;
        const B x = have ? (B)1 << (bkt & (sizeof(B) * 8 - 1)) : (B)0;
#if !defined(GR_ABLATE) || GR_ABLATE != 2  // (timing ablation builds only: tools/build_variant.py)
#pragma unroll
        for (int t = LV - 1; t > 0; t--) ge[t] |= ge[t - 1] & x;
#endif
        ge[0] |= x;
        inb += have && c[u] - row_lo <= band2;
      }
    }
#pragma unroll
    for (int u = 0; u < GR_BATCH; u++) c_cur[u] = c_nx[u];
    // merge the lanes of the row: counts add, saturating at LV
#if defined(GR_ABLATE) && GR_ABLATE == 1
    if (row < 0)
#endif
#pragma unroll
    for (int m = 1; m < GR_LPR; m <<= 1) {
      B o[LV], r[LV];
#pragma unroll
      for (int t = 0; t < LV; t++) o[t] = gr_shfl_xor(ge[t], m);
#pragma unroll
      for (int t = 0; t < LV; t++) {  // more than t in the sum: more than t here, or there, or (i + 1) here and (t - i) there
        B v = ge[t] | o[t];
#pragma unroll
        for (int i = 0; i < t; i++) v |= ge[i] & o[t - 1 - i];
        r[t] = v;
      }
#pragma unroll
      for (int t = 0; t < LV; t++) ge[t] = r[t];
      inb += gr_xor32(inb, m);
    }
    if (sub == 0 && !long_row) {
      B key = ge[0];
#pragma unroll
      for (int t = 1; t < LV; t++) key = thr == (unsigned)t ? ge[t] : key;
#if defined(GR_ABLATE) && GR_ABLATE == 3
      if (key == (B)0x12345678)
#endif
      {
        degree_out[row] = d;
        key_out[row] = gray_decode((unsigned long long)key);
      }
      if (sparse) {
        c_ns += (unsigned)d;
        c_ds += inb;
      } else {
        c_nd += (unsigned)d;
        c_dd += inb;
      }
    }
  }
  c_ns = sbx_wave_sum(c_ns); c_ds = sbx_wave_sum(c_ds); c_nd = sbx_wave_sum(c_nd); c_dd = sbx_wave_sum(c_dd);
  if (sbx_lane() == 0) {
    s_red[tid >> 6][0] = c_ns; s_red[tid >> 6][1] = c_ds; s_red[tid >> 6][2] = c_nd; s_red[tid >> 6][3] = c_dd;
  }
  __syncthreads();
  if (tid < 4) {  // the grid is a few workgroups per CU: one add per counter and workgroup
    const unsigned long long t = s_red[0][tid] + s_red[1][tid] + s_red[2][tid] + s_red[3][tid];
    if (t) atomicAdd(&counts->nnz_sparse + tid, t);
  }
}

// GR_PARTS workgroups per listed long row (a 200 K-entry boundary row must not be one workgroup's job): per-block counts
// in LDS — a thread walks a contiguous piece of the (column-sorted) row, so it adds once per run of equal blocks —
// then into the row's global slot (64 counters + the band count), finished by k_gray_long_finish
constexpr int GR_PARTS = 32;
__global__ __launch_bounds__(256) void k_gray_long_rows(const int32_t *__restrict__ rp, const int32_t *__restrict__ col,
                                                        const int32_t *__restrict__ list, uint32_t width,
                                                        uint32_t magic, uint32_t band, unsigned *__restrict__ slots) {
  __shared__ unsigned s_cnt[64];
  __shared__ unsigned s_inb[4];
  const int tid = threadIdx.x;
  const int64_t li = blockIdx.x / GR_PARTS, part = blockIdx.x % GR_PARTS;
  const int64_t row = list[li];
  const int64_t s = rp[row], len = (int64_t)rp[row + 1] - s;
  const int64_t a = s + len * part / GR_PARTS, b = s + len * (part + 1) / GR_PARTS;  // this workgroup's piece
  if (a >= b) return;
  if (tid < 64) s_cnt[tid] = 0;
  __syncthreads();
  const int64_t per = (b - a + 255) / 256;
  int64_t j = a + per * tid;
  const int64_t jend = j + per < b ? j + per : b;
  unsigned inb = 0, run = 0, cur = 0xFFFFFFFFu;
  for (; j < jend; j++) {
    const int32_t c = col[j];
    unsigned bkt = __umulhi((unsigned)c, magic);
    bkt += ((unsigned)c - bkt * width) >= width;
    if (bkt != cur) {
      if (run) atomicAdd(&s_cnt[cur], run);
      cur = bkt;
      run = 0;
    }
    run++;
    const int diff = c - (int32_t)row;
    inb += (unsigned)(diff < 0 ? -diff : diff) <= band;
  }
  if (run) atomicAdd(&s_cnt[cur], run);
  inb = sbx_wave_sum(inb);
  if (sbx_lane() == 0) s_inb[tid >> 6] = inb;
  __syncthreads();
  unsigned *slot = slots + li * 65;
  if (tid < 64 && s_cnt[tid]) atomicAdd(&slot[tid], s_cnt[tid]);
  if (tid == 64) {
    const unsigned t = s_inb[0] + s_inb[1] + s_inb[2] + s_inb[3];
    if (t) atomicAdd(&slot[64], t);
  }
}

__global__ __launch_bounds__(256) void k_gray_long_finish(const int32_t *__restrict__ rp, const int32_t *__restrict__ list,
                                                          unsigned n_long, const unsigned *__restrict__ slots, int bits,
                                                          int nnz_threshold, int32_t *__restrict__ degree_out,
                                                          unsigned long long *__restrict__ key_out,
                                                          GrayCounts *__restrict__ counts) {
  const unsigned li = blockIdx.x * blockDim.x + threadIdx.x;
  if (li >= n_long) return;
  const int64_t row = list[li];
  const int d = rp[row + 1] - rp[row];
  const unsigned *slot = slots + (size_t)li * 65;
  const unsigned thr = (d > nnz_threshold && d >= bits) ? (unsigned)(d / bits) : 0u;
  unsigned long long key = 0;
  for (int b = 0; b < bits; b++) key |= (unsigned long long)(slot[b] > thr) << b;
  degree_out[row] = d;
  key_out[row] = gray_decode(key);
  if (d <= nnz_threshold) {  // (a long row below the threshold: only with huge thresholds; a handful of adds at most)
    atomicAdd(&counts->nnz_sparse, (unsigned long long)d);
    atomicAdd(&counts->diag_sparse, (unsigned long long)slot[64]);
  } else {
    atomicAdd(&counts->nnz_dense, (unsigned long long)d);
    atomicAdd(&counts->diag_dense, (unsigned long long)slot[64]);
  }
}

// rows cut by a tile boundary: key from the slot of the tile they start in; band counters summed
__global__ __launch_bounds__(256) void k_gray_finish(const int32_t *__restrict__ rp, int bits, int nnz_threshold,
                                                     const int32_t *__restrict__ fix_row,
                                                     const unsigned *__restrict__ slot_acc,
                                                     const unsigned *__restrict__ tile_counts, int64_t ntiles,
                                                     unsigned long long *__restrict__ key_out,
                                                     GrayCounts *__restrict__ counts) {
  __shared__ unsigned long long s_red[4][4];
  unsigned long long part[4] = {0, 0, 0, 0};
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < ntiles; t += (int64_t)gridDim.x * blockDim.x) {
    const uint4 tc = *(const uint4 *)(tile_counts + (size_t)t * 4);
    part[0] += tc.x; part[1] += tc.y; part[2] += tc.z; part[3] += tc.w;
    const int32_t r = fix_row[t];
    if (r < 0) continue;
    const int64_t d = (int64_t)rp[r + 1] - (int64_t)rp[r];
    const unsigned *slot = slot_acc + (size_t)t * 64;
    unsigned long long key = 0;
    if (gray_counted(d, bits, nnz_threshold)) {
      const unsigned thr = (unsigned)(d / bits);
      for (int b = 0; b < bits; b++) key |= (unsigned long long)(slot[b] > thr) << b;
    } else {
      key = (unsigned long long)slot[0] | ((unsigned long long)slot[1] << 32);
    }
    key_out[r] = gray_decode(key);
  }
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const unsigned long long v = sbx_wave_sum(part[q]);
    if (sbx_lane() == 0) s_red[threadIdx.x >> 6][q] = v;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const unsigned long long t = s_red[0][threadIdx.x] + s_red[1][threadIdx.x] + s_red[2][threadIdx.x] + s_red[3][threadIdx.x];
    if (t) atomicAdd(&counts->nnz_sparse + threadIdx.x, t);
  }
}

}  // namespace

extern "C" int sbx_gray_row_keys(sbx_handle_t h, sbx_index_type it, int64_t n, int64_t m, int64_t nnz,
                                 const void *row_ptr, const void *col, int resolution, int nnz_threshold,
                                 void *degree_out, uint64_t *key_out, int64_t *counts_host) {
  if (!h) return SBX_ERR_BAD_ARG;
  if (n < 0 || m < 0 || !row_ptr || !counts_host || (n > 0 && (!degree_out || !key_out)) || (nnz > 0 && !col))
    SBX_FAIL(h, SBX_ERR_BAD_ARG, "sbx_gray_row_keys: bad argument");
  if (it == SBX_I64)
    return sbx_i64_gray_row_keys(h, n, m, nnz, row_ptr, col, resolution, nnz_threshold, degree_out, key_out,
                                 counts_host);
  int bits = resolution;
  if (m < bits) bits = (int)m;  // gray_reorder.cc:206-208
  if (bits <= 0 || bits > 64) SBX_FAIL(h, SBX_ERR_BAD_ARG, "sbx_gray_row_keys: resolution must be in 1..64");
  const int64_t width = m / bits;  // :210
  if (width * bits != m)
    SBX_FAIL(h, SBX_ERR_UNSUPPORTED,
             "sbx_gray_row_keys: m=%lld is not a multiple of the resolution %d (the reference indexes its "
             "bucket array out of bounds there, gray_reorder.cc:251)", (long long)m, bits);
  SBX_TRY(sbx_arena_begin(h));
  counts_host[0] = counts_host[1] = counts_host[2] = counts_host[3] = 0;
  if (n == 0) return SBX_OK;
  struct GrayBoth {  // band counters and the short-row kernel's long-row count: one fill, one read-back
    GrayCounts c;
    unsigned nlong, pad;
  };
  GrayBoth *both = nullptr;
  SBX_TRY(sbx_salloc(h, 1, &both));
  SBX_HIP(h, hipMemsetAsync(both, 0, sizeof(GrayBoth), h->stream));
  GrayCounts *cnt = &both->c;
  const int64_t band = m / 128;  // :138
  // c / width = umulhi(c, magic) or that + 1 (c < 2^31): magic = floor(2^32 / width), saturated for width 1
  const uint64_t mg = ((uint64_t)1 << 32) / (uint64_t)width;
  const uint32_t magic = mg > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)mg;
  const int32_t *rp = (const int32_t *)row_ptr, *cl = (const int32_t *)col;
  unsigned long long *keys = (unsigned long long *)key_out;
  {
    // short-row fast path, tried first: the kernel lists the rows above GR_SHORT_MAX entries it meets; up to
    // GR_LONG_LIST of them (boundary rows of a clamped band, a few hubs) then get GR_PARTS workgroups each
    unsigned *nlong = &both->nlong;
    int32_t *long_list = nullptr;
    SBX_TRY(sbx_salloc(h, (size_t)GR_LONG_LIST, &long_list));
    static const bool allow = !(getenv("SBX_GRAY_SHORT_ROWS") && atoi(getenv("SBX_GRAY_SHORT_ROWS")) == 0);
    const unsigned hmax = (unsigned)GR_SHORT_MAX;  // bound of the rows the kernel handles
    // a row of d <= hmax entries compares its block counts with d / resolution <= hmax / resolution: that many
    // saturating counter slices + 1.  The kernel is built for up to 5 (resolution >= 16) — what the tests of round 2
    // missed and tools/fuzz_ops.py found: below that the tile kernel does the work
    const int lv = (int)(hmax >= (unsigned)bits && (int)hmax > nnz_threshold ? hmax / (unsigned)bits : 0u) + 1;
    if (allow && lv <= (bits <= 32 ? 5 : 2) && nnz >= 4) {  // (gr_load4 reads 16 bytes at a clamped address)
      const unsigned grid = sbx_grid_for(n, 256 / GR_LPR, (int64_t)h->num_cus * 16);
      int wshift = -1;
      if ((width & (width - 1)) == 0)
        for (wshift = 0; ((int64_t)1 << wshift) < width; wshift++) {}
#define GRAY_SHORT(B, LV)                                                                                          \
  do {                                                                                                             \
    if (wshift >= 0)                                                                                               \
      SBX_KLAUNCH(h, SBX_K_GRAY, (k_gray_rows_short<B, LV, true>), dim3(grid), dim3(256), rp, cl, n, (uint32_t)width, \
                  magic, (uint32_t)band, wshift, bits, nnz_threshold, (int32_t *)degree_out, keys, cnt, nlong,   \
                  long_list);                                                                                      \
    else                                                                                                           \
      SBX_KLAUNCH(h, SBX_K_GRAY, (k_gray_rows_short<B, LV, false>), dim3(grid), dim3(256), rp, cl, n,              \
                  (uint32_t)width, magic, (uint32_t)band, 0, bits, nnz_threshold, (int32_t *)degree_out, keys, cnt, \
                  nlong, long_list);                                                                               \
  } while (0)
      if (bits <= 32) {
        if (lv <= 1) GRAY_SHORT(uint32_t, 1);
        else if (lv <= 3) GRAY_SHORT(uint32_t, 3);
        else GRAY_SHORT(uint32_t, 5);
      } else {
        if (lv <= 1) GRAY_SHORT(unsigned long long, 1);
        else GRAY_SHORT(unsigned long long, 2);  // 64 blocks: thr <= 64 / 64
      }
#undef GRAY_SHORT
      SBX_LAUNCH_CHECK(h);
      // one read-back: how many long rows the kernel met and — final if there were none — the band counters
      GrayBoth hb;
      SBX_TRY(sbx_readback(h, &hb, both, sizeof(GrayBoth)));
      const unsigned hlong = hb.nlong;
      if (hlong <= (unsigned)GR_LONG_LIST) {
        if (hlong) {
          unsigned *slots = nullptr;
          SBX_TRY(sbx_salloc(h, (size_t)hlong * 65, &slots));
          SBX_HIP(h, hipMemsetAsync(slots, 0, sizeof(unsigned) * (size_t)hlong * 65, h->stream));
          SBX_KLAUNCH(h, SBX_K_GRAY, k_gray_long_rows, dim3(hlong * GR_PARTS), dim3(256), rp, cl,
                      (const int32_t *)long_list, (uint32_t)width, magic, (uint32_t)band, slots);
          SBX_KLAUNCH(h, SBX_K_GRAY, k_gray_long_finish, dim3((hlong + 255) / 256), dim3(256), rp,
                      (const int32_t *)long_list, hlong, (const unsigned *)slots, bits, nnz_threshold,
                      (int32_t *)degree_out, keys, cnt);
          SBX_LAUNCH_CHECK(h);
          SBX_TRY(sbx_readback(h, &hb, both, sizeof(GrayBoth)));
        }
        SBX_PROF_BYTES(h, SBX_K_GRAY, 4 * nnz + 16 * n + 4);
        counts_host[0] = (int64_t)hb.c.nnz_sparse;
        counts_host[1] = (int64_t)hb.c.diag_sparse;
        counts_host[2] = (int64_t)hb.c.nnz_dense;
        counts_host[3] = (int64_t)hb.c.diag_dense;
        return SBX_OK;
      }
      // too many long rows: the kernel left early; start over with the tile kernel
      SBX_HIP(h, hipMemsetAsync(cnt, 0, sizeof(GrayCounts), h->stream));
    }
  }
  const int64_t ntiles = (nnz + GT_TILE - 1) / GT_TILE;
  int32_t *fix_row = nullptr, *tile_row = nullptr;
  unsigned *slot_acc = nullptr;
  SBX_TRY(sbx_salloc(h, ntiles + 1, &fix_row));
  SBX_TRY(sbx_salloc(h, ntiles + 1, &tile_row));
  SBX_TRY(sbx_salloc(h, (ntiles + 1) * 64, &slot_acc));
  unsigned *tile_counts = nullptr;
  SBX_TRY(sbx_salloc(h, (ntiles + 1) * 4, &tile_counts));
  SBX_HIP(h, hipMemsetAsync(fix_row, 0xFF, (size_t)(ntiles + 1) * sizeof(int32_t), h->stream));
  SBX_HIP(h, hipMemsetAsync(slot_acc, 0, (size_t)(ntiles + 1) * 64 * sizeof(unsigned), h->stream));
  const int64_t prep_items = n > ntiles + 1 ? n : ntiles + 1;
  SBX_KLAUNCH(h, SBX_K_GRAY, k_gray_prep, dim3(sbx_grid_for(prep_items, 256, (int64_t)h->num_cus * 16)), dim3(256),
              rp, n, nnz, ntiles, (int32_t *)degree_out, keys, tile_row);
  if (ntiles > 0) {
    SBX_KLAUNCH(h, SBX_K_GRAY, k_gray_tile, dim3((unsigned)ntiles), dim3(GT_THREADS), rp, cl, nnz,
                (const int32_t *)tile_row, (uint32_t)width, magic, (uint32_t)band, bits, nnz_threshold, keys, fix_row,
                slot_acc, tile_counts);
    SBX_KLAUNCH(h, SBX_K_GRAY, k_gray_finish, dim3(sbx_grid_for(ntiles, 256, (int64_t)h->num_cus)), dim3(256),
                rp, bits, nnz_threshold, (const int32_t *)fix_row, (const unsigned *)slot_acc,
                (const unsigned *)tile_counts, ntiles, keys, cnt);
  }
  SBX_LAUNCH_CHECK(h);
  GrayCounts hc;
  SBX_TRY(sbx_readback(h, &hc, cnt, sizeof(GrayCounts)));
  counts_host[0] = (int64_t)hc.nnz_sparse;
  counts_host[1] = (int64_t)hc.diag_sparse;
  counts_host[2] = (int64_t)hc.nnz_dense;
  counts_host[3] = (int64_t)hc.diag_dense;
  return SBX_OK;
}
