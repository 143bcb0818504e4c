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
// Three kernel families, chosen by what the matrix turns out to be (sbx_gray_row_keys below):
//   k_gray_rows_short (+ k_gray_long_rows/finish, k_gray_list_medium)   banded / mesh matrices: 4 lanes per row
//   k_gray_rows_tiny, k_gray_rows_listed (or k_gray_rows_balanced),    power-law matrices: a lane per row of up to 15
//   k_gray_rows_medium, k_gray_units_finish                            entries, four per row of up to 64, longer rows cut
//                                                                      into 1024-entry units
//   k_gray_prep, k_gray_tile, k_gray_finish                             resolutions below 16: nonzero-parallel tiles
// The ordering stage (std::sort calls whose tie order is libstdc++-specific) stays
// above the ABI in the host layer — see DESIGN.md "Gray".
#include "sbx_device.h"
#include "sbx_internal.h"

namespace {

// element type of the CALLER's arrays — row_ptr, col, degree_out: this file is compiled twice, for 32-bit index arrays
// (sbx_gray.hip) and, through sbx_gray64.hip, for 64-bit ones, which are read and written as they are; every value is a
// 32-bit offset, column or degree once it is in a register (nnz < 2^31, m < 2^31: checked by the entry point)
#ifdef SBX_GRAY_I64
typedef int64_t X;
#define SBX_GRAY_ENTRY sbx_gray_row_keys_x64
#else
typedef int32_t X;
#define SBX_GRAY_ENTRY sbx_gray_row_keys_x32
#endif

__device__ __forceinline__ uint64_t gray_decode(uint64_t g) {
  // prefix xor from the top: b = g ^ g>>1 ^ g>>2 ...
  g ^= g >> 1; g ^= g >> 2; g ^= g >> 4; g ^= g >> 8; g ^= g >> 16; g ^= g >> 32;
  return g;
}

struct GrayCounts {
  unsigned long long nnz_sparse, diag_sparse, nnz_dense, diag_dense;
};

// The lanes of a wave talk through LDS without a barrier in the kernels below (the LDS operations of one wave execute
// in order).  For the language that is a data race, and the compiler does move such accesses past each other
// (k_gray_rows_balanced with its group loop unrolled gave wrong keys): it is told with a wavefront-scope fence, which
// costs no instruction.
#define GR_WAVE_FENCE() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront")

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
__global__ __launch_bounds__(256) void k_gray_prep(const X *__restrict__ rp, int64_t n, int64_t nnz,
                                                   int64_t ntiles, X *__restrict__ degree_out,
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

__global__ __launch_bounds__(GT_THREADS) void k_gray_tile(const X *__restrict__ rp,
                                                          const X *__restrict__ col, int64_t nnz,
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
  if (sizeof(X) == 4 && p0 + GT_ITEMS <= cnt && ((uintptr_t)col & 15) == 0) {
    const int4 *src = (const int4 *)(col + t0 + p0);
    const int4 a = src[0], b = src[1];
    c[0] = a.x; c[1] = a.y; c[2] = a.z; c[3] = a.w; c[4] = b.x; c[5] = b.y; c[6] = b.z; c[7] = b.w;
  } else {
#pragma unroll
    for (int k = 0; k < GT_ITEMS; k++) c[k] = (p0 + k < cnt) ? (int32_t)col[t0 + p0 + k] : 0;
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
constexpr int GR_MED_MAX = 8192;    // rows of GR_SHORT_MAX + 1 .. GR_MED_MAX entries: one wave each (k_gray_rows_medium)
constexpr int GR_LONG_LIST = 4096;  // rows above GR_MED_MAX entries the short-row path lists for k_gray_long_rows
constexpr int GR_FLAGS = 64;     // copies of the stop flag the waves of k_gray_rows_short poll, one per 128-byte line
#ifndef GR_EARLY_V
#define GR_EARLY_V 8
#endif
constexpr int GR_EARLY = GR_EARLY_V;      // copies of the band counters k_gray_rows_short adds to, a line each, right behind GrayBoth's line
constexpr int GR_FLAG_OFF = 24 + GR_EARLY * 32;  // ... starting this many words after the long-row counter (GrayAll below)
constexpr int GR_SPREAD = 64;    // copies of the band counters the power-law kernels add to (4 GrayCounts = a 128-byte line each)
constexpr unsigned GR_POWER_LAW = 2u;  // nlong[1]: 0 no row above GR_SHORT_MAX .. GR_MED_MAX met, 1 some, 2 many (start over)

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
struct __attribute__((packed, aligned(8))) GrU8 {  // four 64-bit columns: the low words are the columns
  unsigned x, xh, y, yh, z, zh, w, wh;
};
// four consecutive columns from entry `at` on (4- resp. 8-byte aligned: gfx950 loads unaligned vectors)
__device__ __forceinline__ GrU4 gr_ld4(const X *__restrict__ col, int32_t at) {
  if constexpr (sizeof(X) == 4) {
    return *(const GrU4 *)(col + at);
  } else {
    const GrU8 v = *(const GrU8 *)(col + at);
    return GrU4{v.x, v.y, v.z, v.w};
  }
}
__device__ __forceinline__ int32_t gr_clamp4(int32_t j, int32_t nnz) {  // where the 16 bytes for entry j are read
  return j < nnz - 4 ? j : nnz - 4;  // nnz >= 4 (the caller's business) and j >= 0: never negative
}
// Issued whatever the lane holds (a load under a condition makes the compiler wait for it at the join, which would
// serialise the prefetch): the address is clamped into the array; which of the four words are entries of the lane's
// row, and the shift the clamp caused for the last three entries of the whole array, are sorted out where the words
// are used (gr_take4).
__device__ __forceinline__ void gr_load4(const X *__restrict__ col, int32_t j, int32_t nnz, unsigned *c) {
#pragma unroll
  for (int v4 = 0; v4 < GR_VEC; v4++) {
#if defined(GR_ABLATE) && GR_ABLATE == 4
    const GrU4 v = {(unsigned)j, (unsigned)j + 1u, (unsigned)j + 2u, (unsigned)j + 3u};
#else
    const GrU4 v = gr_ld4(col, gr_clamp4(j + 4 * v4, nnz));
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
__global__ __launch_bounds__(256) void k_gray_rows_short(const X *__restrict__ rp, const X *__restrict__ col,
                                                         int64_t n, uint32_t width, uint32_t magic, uint32_t band,
                                                         int wshift, int bits, int nnz_threshold,
                                                         X *__restrict__ degree_out,
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
  bool saw_medium = false, stopped = false;
  int steps_done = 0;
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
    const bool long_row = e - s > GR_SHORT_MAX;  // another kernel's business
    // Has some wave found the matrix to be a power-law one (below)?  Looked up in a wave's first three steps — the
    // flag is raised within microseconds of the kernel's start if it is raised at all, and workgroups that start
    // later leave at once — and afterwards only by waves that meet a long row themselves: polling at every step cost
    // the banded matrices 5 – 10 %.  One of GR_FLAGS copies, each on a line of its own (16 K waves polling ONE word
    // queue on its L2 channel: that alone took the kernel from 0.12 to 0.44 ms).
    if (steps_done < 3 || __any(long_row)) {
      if (__atomic_load_n(nlong + GR_FLAG_OFF + (blockIdx.x % GR_FLAGS) * 32, __ATOMIC_RELAXED) == GR_POWER_LAW) {
        stopped = true;
        break;
      }
    }
    steps_done++;
    if (__any(long_row)) {  // (nothing on the common path: the counter is only touched by waves that meet such a row)
      // rows above GR_MED_MAX entries are listed for k_gray_long_rows (a handful of hubs, boundary rows of a clamped
      // band); the rows between only raise a flag: k_gray_list_medium finds them again — a power-law matrix has
      // hundreds of thousands of them, and appending each to a list would queue on one counter word
      const bool heavy = e - s > GR_MED_MAX;
      saw_medium |= long_row && !heavy;  // (published once per workgroup at the end: one word, thousands of waves)
      // two such rows among the 64 / GR_LPR of one wave step: not a banded matrix with a few boundary rows but the body
      // of a power-law degree distribution, where this kernel pays a full batch for rows of four entries.  Every wave
      // stops at its next step, the host discards the results and runs k_gray_rows_balanced.
      if (__popcll(__ballot(long_row && !heavy && sub == 0)) >= 2) {
        // (thousands of waves get here in the kernel's first microseconds: each swaps its OWN flag copy, and only the
        // first one on a copy tells the others and the host — at most GR_FLAGS waves write the shared words)
        unsigned was = 0;
        if (sbx_lane() == 0) was = atomicExch(nlong + GR_FLAG_OFF + (blockIdx.x % GR_FLAGS) * 32, GR_POWER_LAW);
        if (__builtin_amdgcn_readfirstlane((int)was) != (int)GR_POWER_LAW) {
          static_assert(GR_FLAGS == 64, "one flag copy per lane");
          __atomic_store_n(nlong + GR_FLAG_OFF + sbx_lane() * 32, GR_POWER_LAW, __ATOMIC_RELAXED);
          if (sbx_lane() == 0) atomicMax(nlong + 1, GR_POWER_LAW);
        }
        stopped = true;
        break;
      }
      if (__any(heavy)) {
        const unsigned slot = sbx_wave_append(nlong, heavy && sub == 0);
        if (heavy && sub == 0 && slot < (unsigned)GR_LONG_LIST) long_list[slot] = (int32_t)row;
        if (__any(heavy && sub == 0 && slot >= (unsigned)GR_LONG_LIST)) break;  // list full: this wave stops (it still meets
                                                                                // the workgroup's barrier below; the host discards the results)
      }
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
  // (a stopped kernel's results are discarded: nothing to add.  The workgroups of a stopped kernel all end within
  // microseconds of each other, and their 16 K adds to one 128-byte line took 130 us)
  if (__syncthreads_or(stopped)) return;
  if (__syncthreads_or(saw_medium) && tid == 0 && __atomic_load_n(nlong + 1, __ATOMIC_RELAXED) == 0) atomicMax(nlong + 1, 1u);
  if (tid < 4) {
    // one add per counter and workgroup, to one of GR_EARLY copies of the counters, a line each (the host adds them
    // up): the workgroups end together, and their 16 K adds to ONE line were the last 13 us of the kernel
    const unsigned long long t = s_red[0][tid] + s_red[1][tid] + s_red[2][tid] + s_red[3][tid];
    if (t) atomicAdd(&counts[4 + (blockIdx.x % GR_EARLY) * 4].nnz_sparse + tid, t);
  }
}

// ---- power-law matrices ----------------------------------------------------------------------------------------
// On the RMAT bench matrix 41 % of the rows are empty, 45 % hold 1..16 entries, 3.5 % hold 65..8192 entries and 72 % of
// the nonzeros, 0.02 % hold the rest.  k_gray_rows_short pays 32 entry slots for a row of four entries there (270 us
// for a fifth of the nonzeros); it notices (GR_POWER_LAW above) and the host starts over with three kernels that cost
// per ENTRY:
//   k_gray_rows_balanced  rows of up to GR_SHORT_MAX entries, a wave per 64 consecutive rows, a lane per entry;
//                         lists the rows above for the other two
//   k_gray_rows_medium    a wave per listed row of up to GR_MED_MAX entries, 256 entries per step, the next step's
//                         loads in flight across row ends
//   k_gray_long_rows      GR_PARTS workgroups per listed row above GR_MED_MAX entries (as on the banded path)

// The rows above GR_SHORT_MAX entries are cut into UNITS of up to GU_SIZE entries, and k_gray_rows_medium gives every
// wave the same NUMBER of units (one wave per row was built first: rows of 65 and of 8000 entries in one list, the
// average wave was done after 65 us and the last one after 155).  A row of one unit is finished by the wave that
// counts it; the units of a longer row each leave their 64 block counts + band count in a partial slot, and
// k_gray_units_finish adds a row's slots up — plain stores, no atomics, no ordering between units.
//   units[i] = (row, offset of the unit in the row, partial slot or -1, -)
//   mrows[k] = (row, first partial slot, number of units, -)      rows of more than one unit
constexpr int GU_SIZE = 1024;
struct GrayLists {  // (units and slots are reserved together: ONE add per workgroup; the mrows counter on a line of its own)
  alignas(128) unsigned long long units_slots;  // low word: units listed, high word: partial slots handed out
  alignas(128) unsigned n_mrows;
};
// d[i] = length of row row_of(i) + lane if the row is to be listed, 0 otherwise (i < G).  Positions from wave scans and
// one reservation per workgroup on the counters — every thread of the workgroup calls this, once: a workgroup's adds
// to one line take ~7 ns each whatever else it does (4 K workgroups x 3 counters were most of a 109 us kernel).
template <int G, typename RowOf>
__device__ __forceinline__ void gray_emit_units(const int (&d)[G], RowOf row_of, int4 *__restrict__ units,
                                                int4 *__restrict__ mrows, GrayLists *__restrict__ lc,
                                                unsigned (*s_w)[3], unsigned *s_b) {
  const int lane = sbx_lane(), wv = threadIdx.x >> 6;
  unsigned nu = 0, ns = 0, nm = 0;
#pragma unroll
  for (int g = 0; g < G; g++) {
    const unsigned ng = d[g] > 0 ? (unsigned)(d[g] + GU_SIZE - 1) / GU_SIZE : 0u;
    nu += ng;
    ns += ng > 1 ? ng : 0u;
    nm += ng > 1;
  }
  const unsigned iu = sbx_wave_inclusive_sum(nu), is = sbx_wave_inclusive_sum(ns), im = sbx_wave_inclusive_sum(nm);
  if (lane == 63) s_w[wv][0] = iu, s_w[wv][1] = is, s_w[wv][2] = im;
  __syncthreads();
  const unsigned tu = s_w[0][0] + s_w[1][0] + s_w[2][0] + s_w[3][0];
  if (tu == 0) return;  // (workgroup-uniform)
  if (threadIdx.x == 0) {
    const unsigned ts = s_w[0][1] + s_w[1][1] + s_w[2][1] + s_w[3][1];
    const unsigned long long was = atomicAdd(&lc->units_slots, (unsigned long long)tu | ((unsigned long long)ts << 32));
    s_b[0] = (unsigned)was, s_b[1] = (unsigned)(was >> 32);
  }
  if (threadIdx.x == 64) {
    const unsigned tm = s_w[0][2] + s_w[1][2] + s_w[2][2] + s_w[3][2];
    s_b[2] = tm ? atomicAdd(&lc->n_mrows, tm) : 0u;
  }
  __syncthreads();
  unsigned bu = s_b[0] + iu - nu, bs = s_b[1] + is - ns, bm = s_b[2] + im - nm;
  for (int w = 0; w < wv; w++) bu += s_w[w][0], bs += s_w[w][1], bm += s_w[w][2];
#pragma unroll
  for (int g = 0; g < G; g++) {
    const unsigned ng = d[g] > 0 ? (unsigned)(d[g] + GU_SIZE - 1) / GU_SIZE : 0u;
    const int32_t row = (int32_t)(row_of(g) + lane);
    for (unsigned k = 0; k < ng; k++) units[bu + k] = make_int4(row, (int)(k * GU_SIZE), ng > 1 ? (int)(bs + k) : -1, 0);
    if (ng > 1) mrows[bm++] = make_int4(row, (int)bs, (int)ng, 0), bs += ng;
    bu += ng;
  }
}

// Listing for the banded path (k_gray_rows_short met a few rows of GR_SHORT_MAX + 1 .. GR_MED_MAX entries; the ones
// above are on its own list for k_gray_long_rows): 4096 rows per workgroup.
constexpr int GL_ROWS = 4096;
__global__ __launch_bounds__(256) void k_gray_list_medium(const X *__restrict__ rp, int64_t n,
                                                          int4 *__restrict__ units, int4 *__restrict__ mrows,
                                                          GrayLists *__restrict__ lc) {
  __shared__ unsigned s_w[4][3], s_b[3];
  const int lane = sbx_lane(), wv = threadIdx.x >> 6;
  const int64_t r00 = (int64_t)blockIdx.x * GL_ROWS + (int64_t)wv * (GL_ROWS / 4);  // this wave's 1024 rows
  int d[GL_ROWS / 256];
#pragma unroll
  for (int i = 0; i < GL_ROWS / 256; i++) {
    const int64_t r = r00 + i * 64 + lane;
    int32_t len = 0;
    if (r < n) len = rp[r + 1] - rp[r];
    d[i] = len > GR_SHORT_MAX && len <= GR_MED_MAX ? len : 0;
  }
  gray_emit_units(d, [&](int i) { return r00 + i * 64; }, units, mrows, lc, s_w, s_b);
}

// A wave owns GB_GROUPS x 64 consecutive rows.  Per group of 64: lane = row reads the bounds, a wave scan of the short
// rows' lengths numbers their entries 0 .. T-1, and the entries are walked 64 per step with lane = ENTRY: the row of
// an entry is the last row starting at or before it — the rows whose first entry lies in the step's window drop their
// lane number at that slot of a 64-word LDS window, a max-scan spreads it to the right (and a carry brings in the row
// running in from the previous step).  The entry then ORs its block bit into its row's LDS word; rows that compare
// counts with a threshold of 1 .. LV-1 (rows of at least `resolution` entries) keep the same saturating bit-sliced
// counters as k_gray_rows_short, filled by a cascade of returning ORs: the entry that finds its bit already set at
// level t carries it to level t + 1, so level t ends up set iff the block was met more than t times, in any order.
// Cost: a step of ~45 VALU instructions per 64 ENTRIES + ~100 per 64 rows, against ~400 per 16 rows in
// k_gray_rows_short.  (gray_reorder.cc:138-170, :249-267, :384-395)
#ifndef GB_ITERS_V
#define GB_ITERS_V 4
#endif
#ifndef GB_WINDOW_V
#define GB_WINDOW_V 256
#endif
constexpr int GB_GROUPS = 4;
constexpr int GB_ITERS = GB_ITERS_V;      // blocks of 4 x GB_GROUPS x 64 rows per workgroup: one list reservation for all of them
constexpr int GB_WINDOW = GB_WINDOW_V;  // entries per pass: their row lookups, then their loads, then their ORs
template <typename B, int LV, bool POW2>
__global__ __launch_bounds__(256) void k_gray_rows_balanced(const X *__restrict__ rp, const X *__restrict__ col,
                                                            int64_t n, uint32_t width, uint32_t magic, uint32_t band,
                                                            int wshift, int bits, int nnz_threshold,
                                                            X *__restrict__ degree_out,
                                                            unsigned long long *__restrict__ key_out,
                                                            GrayCounts *__restrict__ counts, unsigned *nlong,
                                                            int32_t *__restrict__ long_list, int4 *__restrict__ units,
                                                            int4 *__restrict__ mrows, GrayLists *__restrict__ lc) {
  __shared__ int s_head[4][GB_WINDOW];
  __shared__ int s_base[4][64];       // row_ptr[row] - (number of the row's first entry): entry q is col[s_base + q]
  __shared__ unsigned s_cls[4][64];   // bit 0: row above the threshold ("dense"), bit 1: row counts (thr > 0)
  __shared__ B s_ge[4][LV][64];
  __shared__ unsigned s_w[4][3], s_b[3];
  __shared__ unsigned long long s_red[4][4];
  const int lane = sbx_lane(), wv = threadIdx.x >> 6;
  unsigned long long c_ns = 0, c_nd = 0;
  unsigned c_ds = 0, c_dd = 0;
  int listed[GB_ITERS * GB_GROUPS];  // lengths of the rows above GR_SHORT_MAX entries: k_gray_rows_medium's
#pragma unroll
  for (int i = 0; i < GB_ITERS * GB_GROUPS; i++) listed[i] = 0;
#pragma unroll
  for (int u = 0; u < GB_WINDOW / 64; u++) s_head[wv][u * 64 + lane] = -1;
  // the workgroup's GB_ITERS blocks of 4 x GB_GROUPS x 64 rows lie a grid apart (hubs and leaves in every workgroup)
  auto block_row = [&](int it) { return (((int64_t)blockIdx.x + (int64_t)it * gridDim.x) * 4 + wv) * (GB_GROUPS * 64); };
#pragma unroll 1
  for (int it = 0; it < GB_ITERS; it++) {
  const int64_t r00 = block_row(it);
  int32_t rs_nx = 0, re_nx = 0;
  if (r00 + lane < n) rs_nx = rp[r00 + lane], re_nx = rp[r00 + lane + 1];
  int cur[GB_GROUPS];
#pragma unroll
  for (int g = 0; g < GB_GROUPS; g++) {
    const int64_t r0 = r00 + (int64_t)g * 64, r = r0 + lane;
    const int32_t rs = rs_nx, re = re_nx;
    if (g + 1 < GB_GROUPS && r + 64 < n) rs_nx = rp[r + 64], re_nx = rp[r + 65];
    else rs_nx = re_nx = 0;
    const int d = re - rs;
    cur[g] = d > GR_SHORT_MAX ? d : 0;
    const int ds = d <= GR_SHORT_MAX ? d : 0;
    const int incl = sbx_wave_inclusive_sum(ds);
    const int off = incl - ds;
    const int T = __builtin_amdgcn_readlane(incl, 63);
    unsigned thr = 0;
#pragma unroll
    for (int t = 1; t < LV; t++) thr += ds >= t * bits;
    const bool dense = ds > nnz_threshold;
    thr = dense ? thr : 0u;
    s_base[wv][lane] = rs - off;
    s_cls[wv][lane] = (dense ? 1u : 0u) | (thr ? 2u : 0u);
#pragma unroll
    for (int t = 0; t < LV; t++) s_ge[wv][t][lane] = 0;
    GR_WAVE_FENCE();
    int carry = 0;
    int rl_nx[GB_WINDOW / 64] = {};
    unsigned c_nx[GB_WINDOW / 64] = {};
    // the row lookups of the window at k0 and its loads — always GB_WINDOW / 64 loads, past the end of the group too
    // (dummies): with a path that issues fewer the compiler cannot count the loads behind the ones it waits for
    auto prepare = [&](int k0) {
      if (k0 < T) {  // (wave-uniform)
        // (the LDS operations of one wave execute in order: no barrier between the head writes and the reads)
        if (ds > 0 && off >= k0 && off < k0 + GB_WINDOW) s_head[wv][off - k0] = lane;
        GR_WAVE_FENCE();
#pragma unroll
        for (int u = 0; u < GB_WINDOW / 64; u++) {
          if (k0 + u * 64 < T) {  // (wave-uniform.  Guards, not breaks: see k_gray_rows_medium)
            const int hd = s_head[wv][u * 64 + lane];
            s_head[wv][u * 64 + lane] = -1;
            int m = sbx_wave_inclusive_max(hd);
            m = m > carry ? m : carry;
            carry = __builtin_amdgcn_readlane(m, 63);
            rl_nx[u] = m;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < GB_WINDOW / 64; u++) {
        const int q = k0 + u * 64 + lane;
        const int at = q < T ? s_base[wv][rl_nx[u]] + q : 0;
        c_nx[u] = (unsigned)col[at];
      }
    };
    // the next window's loads are in flight while this one is ORed into its rows' words (a window at a time, a group
    // of 64 rows of 64 entries is 16 load latencies one after the other, and the rows of a power-law matrix with
    // 33..64 entries sit next to each other: those waves were the kernel's last 60 us)
#if defined(GB_ABL) && GB_ABL == 2
    if (r00 < 0)
#endif
    prepare(0);
#if defined(GB_ABL) && GB_ABL == 2
    for (int k0 = 0; k0 < T && r00 < 0; k0 += GB_WINDOW) {
#else
    for (int k0 = 0; k0 < T; k0 += GB_WINDOW) {
#endif
      int rl[GB_WINDOW / 64];
      unsigned c[GB_WINDOW / 64];
#pragma unroll
      for (int u = 0; u < GB_WINDOW / 64; u++) rl[u] = rl_nx[u], c[u] = c_nx[u];
      prepare(k0 + GB_WINDOW);
#pragma unroll
      for (int u = 0; u < GB_WINDOW / 64; u++) {
        if (k0 + u * 64 >= T) continue;
        const bool valid = k0 + u * 64 + lane < T;
        const unsigned cls = s_cls[wv][rl[u]];
        unsigned bkt;
        if (POW2) {
          bkt = c[u] >> wshift;
        } else {
          bkt = __umulhi(c[u], magic);  // c / width, one short at most
          bkt += (c[u] - bkt * width) >= width;
        }
        const B x = (B)1 << (bkt & (sizeof(B) * 8 - 1));
        if (valid) {
          if (LV == 1 || !(cls & 2u)) {
            atomicOr(&s_ge[wv][0][rl[u]], x);
          } else {
#pragma unroll
            for (int t = 0; t < LV; t++) {
              const B old = atomicOr(&s_ge[wv][t][rl[u]], x);
              if (!(old & x)) break;
            }
          }
          const bool inb = c[u] - ((unsigned)(r0 + rl[u]) - band) <= 2u * band;  // |c - row| <= band
          c_dd += inb && (cls & 1u);
          c_ds += inb && !(cls & 1u);
        }
      }
    }
    GR_WAVE_FENCE();
    if (r < n && d <= GR_SHORT_MAX) {
      B key = s_ge[wv][0][lane];
#pragma unroll
      for (int t = 1; t < LV; t++) {
        const B v = s_ge[wv][t][lane];
        key = thr == (unsigned)t ? v : key;
      }
      degree_out[r] = d;
      key_out[r] = gray_decode((unsigned long long)key);
      if (dense) c_nd += (unsigned)d;
      else c_ns += (unsigned)d;
    }
    GR_WAVE_FENCE();
  }
  // (registers are indexed by constants: the block's lengths go to their place under a uniform branch per block)
#pragma unroll
  for (int i = 0; i < GB_ITERS; i++)
    if (it == i) {
#pragma unroll
      for (int g = 0; g < GB_GROUPS; g++) listed[i * GB_GROUPS + g] = cur[g];
    }
  }
  c_ns = sbx_wave_sum(c_ns); c_nd = sbx_wave_sum(c_nd);
  const unsigned long long w_ds = sbx_wave_sum((unsigned long long)c_ds), w_dd = sbx_wave_sum((unsigned long long)c_dd);
  if (lane == 0) {
    s_red[wv][0] = c_ns; s_red[wv][1] = w_ds; s_red[wv][2] = c_nd; s_red[wv][3] = w_dd;
  }
  __syncthreads();
  if (threadIdx.x < 4) {  // (one of GR_SPREAD copies of the counters, a line each; k_gray_units_finish sums them)
    const unsigned long long t = s_red[0][threadIdx.x] + s_red[1][threadIdx.x] + s_red[2][threadIdx.x] + s_red[3][threadIdx.x];
    if (t) atomicAdd(&counts[(blockIdx.x % GR_SPREAD) * 4].nnz_sparse + threadIdx.x, t);
  }
#if !defined(GB_ABL) || GB_ABL != 1  // (timing ablation builds only: tools/build_variant.py)
  gray_emit_units(listed, [&](int i) { return block_row(i / GB_GROUPS) + (i % GB_GROUPS) * 64; }, units, mrows, lc, s_w, s_b);
#endif
}

// ---- power-law matrices, rows of up to GR_SHORT_MAX entries, second form -------------------------------------------
// k_gray_rows_balanced (above) costs ~45 instructions per 64 entries + ~150 per 64 rows and a chain of dependent LDS /
// scan / load steps per group: 123 us on the bench matrix whatever its shape.  Two kernels without LDS do the same work:
//   k_gray_rows_tiny    ONE lane per row for rows of up to GR_TINY entries (87 % of the bench matrix's rows, empty ones
//                       included): four 16-byte loads per lane, the row's block bits ORed in registers.  Such a row is
//                       shorter than `resolution` (>= 16 on this path), so its threshold is 0 and nothing is counted.
//                       Lists the rows of GR_TINY + 1 .. GR_SHORT_MAX entries for the kernel below and cuts the longer
//                       ones into units for k_gray_rows_medium — one reservation per list and workgroup.
//   k_gray_rows_listed  FOUR lanes per listed row, 16 entries each in four 16-byte loads, the bit-sliced saturating
//                       counters and DPP merges of k_gray_rows_short.
constexpr int GR_TINY = 15;
__device__ __forceinline__ void gr_fix4(int32_t j, int cnt, int32_t nnz, unsigned (&c)[4]) {
  if (__any(j > nnz - 4 && cnt > 0)) {  // the end of the array: the vector was read up to 3 entries early
    const int sh = j - gr_clamp4(j, nnz);
    const unsigned w[4] = {c[0], c[1], c[2], c[3]};
#pragma unroll
    for (int t = 0; t < 4; t++) {
      unsigned x = w[t];
#pragma unroll
      for (int k = t + 1; k < 4; k++) x = (sh == k - t) ? w[k] : x;
      c[t] = x;
    }
  }
}

template <typename B, bool POW2>
__global__ __launch_bounds__(256) void k_gray_rows_tiny(const X *__restrict__ rp, const X *__restrict__ col,
                                                        int64_t n, int32_t nnz, uint32_t width, uint32_t magic,
                                                        uint32_t band, int wshift, int nnz_threshold,
                                                        X *__restrict__ degree_out,
                                                        unsigned long long *__restrict__ key_out,
                                                        GrayCounts *__restrict__ counts, int4 *__restrict__ units,
                                                        int4 *__restrict__ mrows, GrayLists *__restrict__ lc,
                                                        int32_t *__restrict__ mid_list, unsigned *__restrict__ mid_count) {
  __shared__ unsigned s_w[4][3], s_b[3];
  __shared__ unsigned s_mid[4], s_mbase;
  __shared__ unsigned long long s_red[4][4];
  const int lane = sbx_lane(), wv = threadIdx.x >> 6;
  constexpr int NG = GB_ITERS * GB_GROUPS;
  unsigned long long c_ns = 0, c_nd = 0;
  unsigned c_ds = 0, c_dd = 0;
  int listed[NG];        // lengths of the rows above GR_SHORT_MAX entries: units of k_gray_rows_medium
  uint64_t midm[NG];     // the rows of GR_TINY + 1 .. GR_SHORT_MAX entries of every group (k_gray_rows_listed)
#pragma unroll
  for (int i = 0; i < NG; i++) listed[i] = 0, midm[i] = 0;
  auto block_row = [&](int it) { return (((int64_t)blockIdx.x + (int64_t)it * gridDim.x) * 4 + wv) * (GB_GROUPS * 64); };
  auto group_row = [&](int i) { return block_row(i / GB_GROUPS) + (int64_t)(i % GB_GROUPS) * 64; };
  // The wave's NG groups of 64 rows as one software pipeline, fully unrolled (every index below is a constant): the row
  // bounds of group i + 2 and the four 16-byte column loads of group i + 1 are issued before group i is worked on.
  // (A group at a time — bounds, then columns, then the ORs — was 108 us: two load latencies per group and wave.)
  int32_t rs[NG + 2], re[NG + 2];
  auto bounds = [&](int i) {  // (unconditional, at a clamped row: loads of a fixed number on every path)
    const int64_t r = group_row(i < NG ? i : NG - 1) + lane;
    const int64_t rc = r < n ? r : n - 1;
    rs[i] = rp[rc], re[i] = rp[rc + 1];
  };
  GrU4 v[2][4];
  // Unconditional loads, but no lane touches a line it has no use for: a vector the row does not reach is read at the
  // row's first vector again, and the lanes of empty and of longer rows read where the group's first row starts (the
  // rows' first lines alone are 2 M x 128 bytes on the bench matrix; four vectors per lane whatever the row's length
  // fetched twice that and the kernel ran at the HBM's pace: 110 us)
  auto vload = [&](int i) {
    const int64_t r = group_row(i) + lane;
    const int d = r < n ? re[i] - rs[i] : 0;
    const int dt = d <= GR_TINY ? d : 0;
    const int32_t first = __builtin_amdgcn_readfirstlane(rs[i]);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int32_t at = dt > 4 * k ? rs[i] + 4 * k : (dt > 0 ? rs[i] : first);
      v[i & 1][k] = gr_ld4(col, gr_clamp4(at, nnz));
    }
  };
  bounds(0);
  bounds(1);
  vload(0);
#pragma unroll
  for (int i = 0; i < NG; i++) {
    if (i + 2 < NG) bounds(i + 2);
    if (i + 1 < NG) vload(i + 1);
    const int64_t r = group_row(i) + lane;
    const int d = r < n ? re[i] - rs[i] : 0;
    listed[i] = d > GR_SHORT_MAX ? d : 0;
    midm[i] = __ballot(d > GR_TINY && d <= GR_SHORT_MAX);
    const int dt = d <= GR_TINY ? d : 0;
    B mask = 0;
    unsigned inb = 0;
    const unsigned row_lo = (unsigned)r - band, band2 = 2u * band;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int left = dt - 4 * k;
      const int cnt = left < 0 ? 0 : (left > 4 ? 4 : left);
      unsigned c[4] = {v[i & 1][k].x, v[i & 1][k].y, v[i & 1][k].z, v[i & 1][k].w};
      gr_fix4(rs[i] + 4 * k, cnt, nnz, c);
#pragma unroll
      for (int t = 0; t < 4; t++) {
        unsigned bkt;
        if (POW2) {
          bkt = c[t] >> wshift;
        } else {
          bkt = __umulhi(c[t], magic);
          bkt += (c[t] - bkt * width) >= width;
        }
        const bool have = t < cnt;
        mask |= have ? (B)1 << (bkt & (sizeof(B) * 8 - 1)) : (B)0;
        inb += have && c[t] - row_lo <= band2;
      }
    }
    if (r < n && d <= GR_TINY) {
      degree_out[r] = d;
      key_out[r] = gray_decode((unsigned long long)mask);
      if (d <= nnz_threshold) c_ns += (unsigned)d, c_ds += inb;
      else c_nd += (unsigned)d, c_dd += inb;
    }
  }
  c_ns = sbx_wave_sum(c_ns); c_nd = sbx_wave_sum(c_nd);
  const unsigned long long w_ds = sbx_wave_sum((unsigned long long)c_ds), w_dd = sbx_wave_sum((unsigned long long)c_dd);
  unsigned mid_mine = 0;
#pragma unroll
  for (int i = 0; i < NG; i++) mid_mine += (unsigned)__popcll(midm[i]);
  if (lane == 0) {
    s_red[wv][0] = c_ns; s_red[wv][1] = w_ds; s_red[wv][2] = c_nd; s_red[wv][3] = w_dd;
    s_mid[wv] = mid_mine;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const unsigned long long t = s_red[0][threadIdx.x] + s_red[1][threadIdx.x] + s_red[2][threadIdx.x] + s_red[3][threadIdx.x];
    if (t) atomicAdd(&counts[(blockIdx.x % GR_SPREAD) * 4].nnz_sparse + threadIdx.x, t);
  }
  {  // the mid-length rows: one reservation per workgroup
    const unsigned tot = s_mid[0] + s_mid[1] + s_mid[2] + s_mid[3];
    if (threadIdx.x == 0) s_mbase = tot ? atomicAdd(mid_count, tot) : 0u;
    __syncthreads();
    unsigned o = s_mbase;
    for (int i = 0; i < wv; i++) o += s_mid[i];
#pragma unroll
    for (int i = 0; i < NG; i++) {
      if ((midm[i] >> lane) & 1ull) mid_list[o + (unsigned)__popcll(midm[i] & sbx_lanemask_lt())] = (int32_t)(group_row(i) + lane);
      o += (unsigned)__popcll(midm[i]);
    }
  }
  gray_emit_units(listed, [&](int i) { return group_row(i); }, units, mrows, lc, s_w, s_b);
}

template <typename B, int LV, bool POW2>
__global__ __launch_bounds__(256) void k_gray_rows_listed(const X *__restrict__ rp, const X *__restrict__ col,
                                                          const int32_t *__restrict__ list,
                                                          const unsigned *__restrict__ count, int32_t nnz,
                                                          uint32_t width, uint32_t magic, uint32_t band, int wshift,
                                                          int bits, int nnz_threshold, X *__restrict__ degree_out,
                                                          unsigned long long *__restrict__ key_out,
                                                          GrayCounts *__restrict__ counts) {
  __shared__ unsigned long long s_red[4][4];
  const int tid = threadIdx.x, lane = sbx_lane(), sub = lane & 3;
  const unsigned total = *count;
  unsigned long long c_ns = 0, c_ds = 0, c_nd = 0, c_dd = 0;
  const unsigned gwave = (blockIdx.x * 256u + (unsigned)tid) >> 6, nwaves = (gridDim.x * 256u) >> 6;
  for (unsigned base = gwave * 16u; base < total; base += nwaves * 16u) {  // 16 rows per wave and step, 4 lanes each
    const unsigned i = base + (unsigned)(lane >> 2);
    const bool valid = i < total;
    const int32_t row = list[valid ? i : total - 1];
    const int32_t rs = rp[row], re = rp[row + 1];
    const int d = valid ? re - rs : 0;
    GrU4 v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {  // (a vector behind the row's end is read at the row's start: no line nobody needs)
      const int32_t at = 16 * sub + 4 * k < d ? rs + 16 * sub + 4 * k : rs;
      v[k] = gr_ld4(col, gr_clamp4(at, nnz));
    }
    const bool sparse = d <= nnz_threshold;
    unsigned thr = 0;
#pragma unroll
    for (int t = 1; t < LV; t++) thr += d >= t * bits;
    thr = d > nnz_threshold ? thr : 0u;
    B ge[LV];
#pragma unroll
    for (int t = 0; t < LV; t++) ge[t] = 0;
    unsigned inb = 0;
    const unsigned row_lo = (unsigned)row - band, band2 = 2u * band;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int left = d - (16 * sub + 4 * k);
      const int cnt = left < 0 ? 0 : (left > 4 ? 4 : left);
      unsigned c[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
      gr_fix4(rs + 16 * sub + 4 * k, cnt, nnz, c);
#pragma unroll
      for (int u = 0; u < 4; u++) {
        unsigned bkt;
        if (POW2) {
          bkt = c[u] >> wshift;
        } else {
          bkt = __umulhi(c[u], magic);
          bkt += (c[u] - bkt * width) >= width;
        }
        const bool have = u < cnt;
        const B x = have ? (B)1 << (bkt & (sizeof(B) * 8 - 1)) : (B)0;
#pragma unroll
        for (int t = LV - 1; t > 0; t--) ge[t] |= ge[t - 1] & x;
        ge[0] |= x;
        inb += have && c[u] - row_lo <= band2;
      }
    }
    // merge the four lanes of the row: counts add, saturating at LV (k_gray_rows_short's merge)
#pragma unroll
    for (int m = 1; m < 4; m <<= 1) {
      B o[LV], r[LV];
#pragma unroll
      for (int t = 0; t < LV; t++) o[t] = gr_shfl_xor(ge[t], m);
#pragma unroll
      for (int t = 0; t < LV; t++) {
        B w = ge[t] | o[t];
#pragma unroll
        for (int q = 0; q < t; q++) w |= ge[q] & o[t - 1 - q];
        r[t] = w;
      }
#pragma unroll
      for (int t = 0; t < LV; t++) ge[t] = r[t];
      inb += gr_xor32(inb, m);
    }
    if (sub == 0 && valid) {
      B key = ge[0];
#pragma unroll
      for (int t = 1; t < LV; t++) key = thr == (unsigned)t ? ge[t] : key;
      degree_out[row] = d;
      key_out[row] = gray_decode((unsigned long long)key);
      if (sparse) c_ns += (unsigned)d, c_ds += inb;
      else c_nd += (unsigned)d, c_dd += inb;
    }
  }
  c_ns = sbx_wave_sum(c_ns); c_ds = sbx_wave_sum(c_ds); c_nd = sbx_wave_sum(c_nd); c_dd = sbx_wave_sum(c_dd);
  if (lane == 0) {
    s_red[tid >> 6][0] = c_ns; s_red[tid >> 6][1] = c_ds; s_red[tid >> 6][2] = c_nd; s_red[tid >> 6][3] = c_dd;
  }
  __syncthreads();
  if (tid < 4) {
    const unsigned long long t = s_red[0][tid] + s_red[1][tid] + s_red[2][tid] + s_red[3][tid];
    if (t) atomicAdd(&counts[(blockIdx.x % GR_SPREAD) * 4].nnz_sparse + tid, t);
  }
}

// One wave per listed unit, round-robin over the list.  (Finding the rows where they are — a wave per 64 consecutive
// rows, or per 64 rows a fixed stride apart — was built first: 8.8 and 9.0 ms on the RMAT bench matrix, whose
// generator puts the hubs next to each other AND at ids with many trailing zero bits: whatever regular map from row
// id to wave is used, some wave gets a few hundred of them.)
// A wave's units are one stream of GM_STEP-entry steps — four consecutive entries per lane, one 16-byte load — with
// GM_DEPTH of them in flight: a step is counted, then the step GM_DEPTH further on, of this unit or of a later one, is
// loaded into its registers.  (The kernel is bound by instruction issue, not by memory: with a 4-byte load per lane and
// step, 41 VALU + 44 scalar instructions per 64 entries kept every SIMD busy, tools/pmc_gray.sh, at 1.3 TB/s; and with
// one load per wave outstanding the list -> row_ptr -> col chain limited the first version to 0.95 TB/s.)  Block
// counts sit in the wave's own 64 LDS words; at the end of a unit lane b holds block b's count: compared with the
// row's threshold, a ballot is the key — or, for a unit of a longer row, stored to the unit's partial slot.
// (gray_reorder.cc:249-267, :384-395)
constexpr int GM_DEPTH = 4;   // steps in flight per wave
constexpr int GM_STEP = 256;  // entries per step
constexpr int GU_SLOT = 65;   // words of a partial slot: 64 block counts, the band count
// (The scalar unit is shared by the four SIMDs of a CU and was the busiest part of the first versions — 86 scalar
// instructions per step, most of them copies of per-step bookkeeping and exec-mask handling: the fetch and the count
// side now each walk the unit table with a cursor of their own, and the adds are unconditional with a value of 0
// where a lane has nothing to add.)
template <bool POW2>
__global__ __launch_bounds__(256) void k_gray_rows_medium(const X *__restrict__ rp, const X *__restrict__ col,
                                                          const int4 *__restrict__ units,
                                                          const GrayLists *__restrict__ lc, uint32_t width,
                                                          uint32_t magic, int wshift, uint32_t band, int bits,
                                                          int nnz_threshold, int32_t nnz,
                                                          X *__restrict__ degree_out,
                                                          unsigned long long *__restrict__ key_out,
                                                          unsigned *__restrict__ partial,
                                                          GrayCounts *__restrict__ counts, unsigned spread) {
  __shared__ unsigned s_cnt[4][64];
  __shared__ unsigned long long s_red[4][4];
  __shared__ int32_t s_tab[4][6][64];  // row, first entry, end, row length, partial slot, threshold
  __shared__ unsigned s_idle[4][64];   // where lanes with nothing to count add
  const int lane = sbx_lane();
  // (readfirstlane: the compiler cannot know that threadIdx.x >> 6 is the same in all lanes of a wave; with it the
  // cursors below are scalars and their branches scalar branches)
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned gwave = blockIdx.x * 4u + (unsigned)wv, nwaves = gridDim.x * 4u;
  const unsigned total = (unsigned)lc->units_slots;
  unsigned long long v_ns = 0, v_ds = 0, v_nd = 0, v_dd = 0;  // (the same in every lane)
  const int mine = gwave < total ? (int)((total - gwave + nwaves - 1) / nwaves) : 0;  // this wave's units: units[gwave + k nwaves]
  // the bounds of 64 of the wave's units at a time, in LDS (list -> row_ptr: two dependent loads, paid once per 64
  // units instead of once per unit); the pipeline drains between such blocks
  for (int kb = 0; kb < mine; kb += 64) {
    const int blk = mine - kb < 64 ? mine - kb : 64;
    {
      int32_t t_row = 0, t_a = 0, t_b = 0, t_dd = 0, t_slot = -1, t_thr = 0;
      if (lane < blk) {
        const int4 un = units[gwave + (unsigned)(kb + lane) * nwaves];
        const int32_t rs = rp[un.x], re = rp[un.x + 1];
        t_row = un.x, t_a = rs + un.y, t_dd = re - rs, t_slot = un.z;
        t_b = re - t_a < GU_SIZE ? re : t_a + GU_SIZE;
        t_thr = (t_dd > nnz_threshold && t_dd >= bits) ? t_dd / bits : 0;
      }
      s_tab[wv][0][lane] = t_row, s_tab[wv][1][lane] = t_a, s_tab[wv][2][lane] = t_b, s_tab[wv][3][lane] = t_dd,
      s_tab[wv][4][lane] = t_slot, s_tab[wv][5][lane] = t_thr;
      GR_WAVE_FENCE();
    }
    auto table = [&](int what, int k) { return __builtin_amdgcn_readfirstlane(s_tab[wv][what][k]); };
    // the fetch cursor: the next step to load.  A load on EVERY call, past the end of the stream too: with a path that
    // skips one the compiler cannot count the loads behind the one it waits for, and waits for all of them
    int f_k = 0;
    int32_t f_j = table(1, 0), f_b = table(2, 0);
    GrU4 R[GM_DEPTH];
    auto fetch = [&](int u) {
      const int32_t jj = f_k < blk ? f_j + 4 * lane : 0;
      R[u] = gr_ld4(col, gr_clamp4(jj, nnz));  // (only 4-byte aligned: gfx950 loads unaligned vectors)
      if (f_k < blk) {
        f_j += GM_STEP;
        if (f_j >= f_b) {  // next unit
          f_k++;
          if (f_k < blk) f_j = table(1, f_k), f_b = table(2, f_k);
        }
      }
    };
#pragma unroll
    for (int u = 0; u < GM_DEPTH; u++) fetch(u);
    // the count cursor: the same walk, GM_DEPTH steps behind
    int c_k = 0;
    int32_t c_row = table(0, 0), c_j = table(1, 0), c_b = table(2, 0);
    s_cnt[wv][lane] = 0;  // (the LDS operations of one wave execute in order: no barrier around the counters)
    GR_WAVE_FENCE();
    unsigned inb = 0;
    while (c_k < blk) {
#pragma unroll
      for (int u = 0; u < GM_DEPTH; u++) {
        // (no break here: with an exit from the middle of the ring the compiler gives up counting the loads in flight
        // and waits for all of them before every step; past the end of the stream the remaining steps are skipped)
        if (c_k < blk) {
          const int32_t j = c_j + 4 * lane;
          const int left = c_b - j;
          const int cnt = left < 0 ? 0 : (left > 4 ? 4 : left);
          unsigned c[4] = {R[u].x, R[u].y, R[u].z, R[u].w};
          if (__any(j > nnz - 4 && cnt > 0)) {  // the end of the array: the vector was read up to 3 entries early
            const int sh = j - gr_clamp4(j, nnz);
            const unsigned w[4] = {c[0], c[1], c[2], c[3]};
#pragma unroll
            for (int t = 0; t < 4; t++) {
              unsigned x = w[t];
#pragma unroll
              for (int k = t + 1; k < 4; k++) x = (sh == k - t) ? w[k] : x;
              c[t] = x;
            }
          }
          unsigned bkt[4];
          const unsigned row_lo = (unsigned)c_row - band, band2 = 2u * band;  // |c - row| <= band  <=>  c - row_lo <= 2 band (mod 2^32)
#pragma unroll
          for (int t = 0; t < 4; t++) {
            if (POW2) {
              bkt[t] = c[t] >> wshift;
            } else {
              bkt[t] = __umulhi(c[t], magic);
              bkt[t] += (c[t] - bkt[t] * width) >= width;
            }
            inb += t < cnt && c[t] - row_lo <= band2;
          }
          // The columns of a row ascend, so its entries come in RUNS of equal blocks: the first lane of a run of lanes
          // whose four entries all fall into one block adds the run's length, in one add; only the lanes a block
          // boundary passes through add their entries one by one.  (Every lane adding its own entries to 64 words
          // kept the LDS busy with bank conflicts for 84 % of its cycles, tools/pmc_gray.sh, and the waves waiting on
          // it for half of theirs.)  Rows whose columns do NOT ascend are counted correctly too, just with more adds.
          const bool active = cnt > 0;
          const bool uni = (cnt < 2 || bkt[1] == bkt[0]) && (cnt < 3 || bkt[2] == bkt[0]) && (cnt < 4 || bkt[3] == bkt[0]);
          const unsigned rk = active && uni ? bkt[0] : 0xFFFFFFFFu;
          const unsigned prev = sbx_wave_shift_up1(rk, 0xFFFFFFFEu);
          const bool head = active && uni && rk != prev;
          const uint64_t stops = __ballot(head || !(active && uni));  // where a run ends: the next run, a mixed lane, the end
          {
            const int left_u = c_b - c_j;  // entries of this step
            const int nvalid = left_u < GM_STEP ? left_u : GM_STEP;
            const uint64_t above = lane < 63 ? stops & ~(((uint64_t)2 << lane) - 1) : 0;
            const int next = above ? __builtin_ctzll(above) : 64;
            const int upto = 4 * next < nvalid ? 4 * next : nvalid;
            // (no add under a condition — a branch on the exec mask is three scalar instructions, and the scalar unit
            // is the busy one: a lane with nothing to add adds to a word of its own)
            atomicAdd(head ? &s_cnt[wv][rk] : &s_idle[wv][lane], (unsigned)(upto - 4 * lane));
          }
          if (__any(active && !uni)) {
#pragma unroll
            for (int t = 0; t < 4; t++) atomicAdd(active && !uni && t < cnt ? &s_cnt[wv][bkt[t]] : &s_idle[wv][lane], 1u);
          }
          if (c_j + GM_STEP >= c_b) {  // the unit's last step
            const int32_t dd = table(3, c_k), slot = table(4, c_k);
            const unsigned thr = (unsigned)table(5, c_k);
            inb = sbx_wave_sum(inb);
            GR_WAVE_FENCE();
            const unsigned cb = s_cnt[wv][lane];
            s_cnt[wv][lane] = 0;
            GR_WAVE_FENCE();
            // (from here on in vector registers although the values are the same in every lane: the scalar unit is
            // shared by four SIMDs and busy with the cursors)
            int row_v = c_row, dd_v = dd;
            asm volatile("" : "+v"(row_v), "+v"(dd_v));
            if (slot >= 0) {  // one of several units of its row: k_gray_units_finish adds them up
              unsigned *ps = partial + (size_t)slot * GU_SLOT;
              ps[lane] = cb;
              if (lane == 0) ps[64] = inb;
            } else {
              unsigned long long key = __ballot(lane < bits && cb > thr);
              asm volatile("" : "+v"(key));
              if (lane == 0) {
                degree_out[row_v] = dd_v;
                key_out[row_v] = gray_decode(key);
              }
              const bool sparse_v = dd_v <= nnz_threshold;
              v_ns += sparse_v ? (unsigned)dd_v : 0u, v_ds += sparse_v ? inb : 0u;
              v_nd += sparse_v ? 0u : (unsigned)dd_v, v_dd += sparse_v ? 0u : inb;
            }
            inb = 0;
            c_k++;
            if (c_k < blk) c_row = table(0, c_k), c_j = table(1, c_k), c_b = table(2, c_k);
          } else {
            c_j += GM_STEP;
          }
        }
        fetch(u);
      }
    }
  }
  if (lane == 0) {
    s_red[wv][0] = v_ns; s_red[wv][1] = v_ds; s_red[wv][2] = v_nd; s_red[wv][3] = v_dd;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const unsigned long long t = s_red[0][threadIdx.x] + s_red[1][threadIdx.x] + s_red[2][threadIdx.x] + s_red[3][threadIdx.x];
    if (t) atomicAdd(&counts[(blockIdx.x % spread) * 4].nnz_sparse + threadIdx.x, t);  // (`spread` copies, a line each)
  }
}

// The rows of more than one unit, a wave each: lane b adds block b's count over the row's partial slots.
__global__ __launch_bounds__(256) void k_gray_units_finish(const X *__restrict__ rp, const int4 *__restrict__ mrows,
                                                           const GrayLists *__restrict__ lc,
                                                           const unsigned *__restrict__ partial, int bits,
                                                           int nnz_threshold, X *__restrict__ degree_out,
                                                           unsigned long long *__restrict__ key_out,
                                                           GrayCounts *__restrict__ counts, unsigned n_spread) {
  __shared__ unsigned long long s_red[4][4];
  const unsigned nm = lc->n_mrows;
  const int lane = sbx_lane(), wv = threadIdx.x >> 6;
  unsigned long long c[4] = {0, 0, 0, 0};  // nnz / band count of the rows below the threshold, of the rows above (lane 0's)
  for (unsigned k = blockIdx.x * 4u + (unsigned)wv; k < nm; k += gridDim.x * 4u) {
    const int4 mr = mrows[k];
    const int32_t row = mr.x;
    const int d = rp[row + 1] - rp[row];
    unsigned cb = 0, inb = 0;
    int u = 0;
    for (; u + 8 <= mr.z; u += 8) {  // (eight loads in flight: a 300 K-entry hub is 300 slots)
      unsigned v[8], w[8];
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const unsigned *ps = partial + (size_t)(mr.y + u + i) * GU_SLOT;
        v[i] = ps[lane], w[i] = ps[64];
      }
#pragma unroll
      for (int i = 0; i < 8; i++) cb += v[i], inb += w[i];
    }
    for (; u < mr.z; u++) {
      const unsigned *ps = partial + (size_t)(mr.y + u) * GU_SLOT;
      cb += ps[lane], inb += ps[64];
    }
    const unsigned thr = (d > nnz_threshold && d >= bits) ? (unsigned)(d / bits) : 0u;
    const unsigned long long key = __ballot(lane < bits && cb > thr);
    if (lane == 0) {
      degree_out[row] = d;
      key_out[row] = gray_decode(key);
    }
    const int cls = d <= nnz_threshold ? 0 : 2;
    c[cls] += (unsigned)d, c[cls + 1] += inb;
  }
  // (one add per counter and workgroup: 23 K rows x 2 adds to one line were 220 us)
  if (lane == 0) s_red[wv][0] = c[0], s_red[wv][1] = c[1], s_red[wv][2] = c[2], s_red[wv][3] = c[3];
  __syncthreads();
  if (threadIdx.x < 4) {  // (to one of the counter copies, as the kernels before: 4 K adds to one line were this kernel's 28 us)
    const unsigned long long t = s_red[0][threadIdx.x] + s_red[1][threadIdx.x] + s_red[2][threadIdx.x] + s_red[3][threadIdx.x];
    if (t) atomicAdd(&counts[(blockIdx.x % n_spread) * 4].nnz_sparse + threadIdx.x, t);
  }
}

// the counter copies of the power-law kernels, summed into the words the host reads
__global__ __launch_bounds__(64) void k_gray_fold(const GrayCounts *__restrict__ spread, GrayCounts *__restrict__ total) {
  const int lane = sbx_lane();
  unsigned long long t[4];
#pragma unroll
  for (int q = 0; q < 4; q++) t[q] = sbx_wave_sum(lane < GR_SPREAD ? (&spread[lane * 4].nnz_sparse)[q] : 0ull);
  static_assert(GR_SPREAD <= 64, "one copy per lane");
  if (lane == 0) total->nnz_sparse = t[0], total->diag_sparse = t[1], total->nnz_dense = t[2], total->diag_dense = t[3];
}

// GR_PARTS workgroups per listed long row (a 200 K-entry boundary row must not be one workgroup's job): per-block counts
// in LDS — a thread walks a contiguous piece of the (column-sorted) row, so it adds once per run of equal blocks —
// then into the row's global slot (64 counters + the band count), finished by k_gray_long_finish.  The number of listed
// rows is read from the device (the power-law path launches this without a read-back in between): the workgroups loop.
constexpr int GR_PARTS = 32;
__global__ __launch_bounds__(256) void k_gray_long_rows(const X *__restrict__ rp, const X *__restrict__ col,
                                                        const int32_t *__restrict__ list,
                                                        const unsigned *__restrict__ n_long, uint32_t width,
                                                        uint32_t magic, uint32_t band, unsigned *__restrict__ slots) {
  __shared__ unsigned s_cnt[64];
  __shared__ unsigned s_inb[4];
  const int tid = threadIdx.x;
  unsigned nl = *n_long;
  if (nl > (unsigned)GR_LONG_LIST) return;  // (the host takes the tile kernel)
  for (unsigned w = blockIdx.x; w < nl * GR_PARTS; w += gridDim.x) {
    const int64_t li = w / GR_PARTS, part = w % GR_PARTS;
    const int64_t row = list[li];
    const int64_t s = rp[row], len = (int64_t)rp[row + 1] - s;
    const int64_t a = s + len * part / GR_PARTS, b = s + len * (part + 1) / GR_PARTS;  // this workgroup's piece
    if (a >= b) continue;
    __syncthreads();  // (the previous piece's counters have been read)
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
}

__global__ __launch_bounds__(256) void k_gray_long_finish(const X *__restrict__ rp, const int32_t *__restrict__ list,
                                                          const unsigned *__restrict__ n_long,
                                                          const unsigned *__restrict__ slots, int bits,
                                                          int nnz_threshold, X *__restrict__ degree_out,
                                                          unsigned long long *__restrict__ key_out,
                                                          GrayCounts *__restrict__ counts) {
  const unsigned nl = *n_long;
  if (nl > (unsigned)GR_LONG_LIST) return;
  const int lane = sbx_lane();
  // a wave per row, lane b compares block b's count (a thread per row read its 64 counters one after the other:
  // 24 us for 800 rows)
  for (unsigned li = blockIdx.x * 4u + (threadIdx.x >> 6); li < nl; li += gridDim.x * 4u) {
    const int64_t row = list[li];
    const int d = rp[row + 1] - rp[row];
    const unsigned *slot = slots + (size_t)li * 65;
    const unsigned thr = (d > nnz_threshold && d >= bits) ? (unsigned)(d / bits) : 0u;
    const unsigned cb = slot[lane];
    const unsigned long long key = __ballot(lane < bits && cb > thr);
    if (lane == 0) {
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
  }
}

// rows cut by a tile boundary: key from the slot of the tile they start in; band counters summed
__global__ __launch_bounds__(256) void k_gray_finish(const X *__restrict__ rp, int bits, int nnz_threshold,
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

int sbx_gray_row_keys_x32(sbx_handle_t h, int64_t n, int64_t m, int64_t nnz, const void *row_ptr, const void *col,
                          int resolution, int nnz_threshold, void *degree_out, uint64_t *key_out, int64_t *counts_host);
int sbx_gray_row_keys_x64(sbx_handle_t h, int64_t n, int64_t m, int64_t nnz, const void *row_ptr, const void *col,
                          int resolution, int nnz_threshold, void *degree_out, uint64_t *key_out, int64_t *counts_host);
#ifndef SBX_GRAY_I64
extern "C" int sbx_gray_row_keys(sbx_handle_t h, sbx_index_type it, int64_t n, int64_t m, int64_t nnz,
                                 const void *row_ptr, const void *col, int resolution, int nnz_threshold,
                                 void *degree_out, uint64_t *key_out, int64_t *counts_host) {
  if (!h) return SBX_ERR_BAD_ARG;
  if (it == SBX_I32_N64)
    return sbx_mixed_gray_row_keys(h, n, m, nnz, row_ptr, col, resolution, nnz_threshold, degree_out, key_out, counts_host);
  if (n < 0 || m < 0 || !row_ptr || !counts_host || (n > 0 && (!degree_out || !key_out)) || (nnz > 0 && !col))
    SBX_FAIL(h, SBX_ERR_BAD_ARG, "sbx_gray_row_keys: bad argument");
  // (offsets, columns and degrees are 32-bit inside, whatever the width of the arrays)
  if (it == SBX_I64 && (n >= ((int64_t)1 << 31) - 1 || m >= ((int64_t)1 << 31) || nnz >= ((int64_t)1 << 31)))
    SBX_FAIL(h, SBX_ERR_UNSUPPORTED, "sbx_gray_row_keys: dimension exceeds int32");
  return it == SBX_I64 ? sbx_gray_row_keys_x64(h, n, m, nnz, row_ptr, col, resolution, nnz_threshold, degree_out, key_out, counts_host)
                       : sbx_gray_row_keys_x32(h, n, m, nnz, row_ptr, col, resolution, nnz_threshold, degree_out, key_out, counts_host);
}
#endif

int SBX_GRAY_ENTRY(sbx_handle_t h, int64_t n, int64_t m, int64_t nnz, const void *row_ptr, const void *col, int resolution,
                   int nnz_threshold, void *degree_out, uint64_t *key_out, int64_t *counts_host) {
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
    unsigned nlong, pad;  // pad: 0 / 1 / GR_POWER_LAW
  };
  struct GrayAll {  // everything the kernels count in: one fill at the start of the call
    GrayBoth b;
    unsigned fill[24 - 2];
    GrayCounts early[GR_EARLY * 4];  // k_gray_rows_short's counter copies (line i at &b.c + 4 + 4 i)
    unsigned flags[GR_FLAGS * 32];
    alignas(128) GrayCounts total2;  // the power-law path's counters (b.c holds what the stopped kernel left)
    GrayLists lists;
    alignas(128) unsigned mid_count;  // rows listed for k_gray_rows_listed
    alignas(128) GrayCounts spread[GR_SPREAD * 4];
  };
  static_assert(offsetof(GrayAll, flags) == offsetof(GrayBoth, nlong) + GR_FLAG_OFF * 4 && offsetof(GrayAll, flags) % 128 == 0, "flag lines");
  static_assert(offsetof(GrayAll, early) == 4 * sizeof(GrayCounts), "the first counter copy sits one line behind the counters");
  GrayAll *all = nullptr;
  SBX_TRY(sbx_salloc(h, 1, &all));
  SBX_HIP(h, hipMemsetAsync(all, 0, sizeof(GrayAll), h->stream));
  GrayBoth *both = &all->b;
  GrayCounts *cnt = &both->c;
  const int64_t band = m / 128;  // :138
  // c / width = umulhi(c, magic) or that + 1 (c < 2^31): magic = floor(2^32 / width), saturated for width 1
  const uint64_t mg = ((uint64_t)1 << 32) / (uint64_t)width;
  const uint32_t magic = mg > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)mg;
  const X *rp = (const X *)row_ptr, *cl = (const X *)col;
  unsigned long long *keys = (unsigned long long *)key_out;
  {
    // short-row fast path, tried first: the kernel lists the rows above GR_SHORT_MAX entries it meets; up to
    // GR_LONG_LIST of them (boundary rows of a clamped band, a few hubs) then get GR_PARTS workgroups each
    unsigned *nlong = &both->nlong;
    int32_t *long_list = nullptr;
    SBX_TRY(sbx_salloc(h, (size_t)GR_LONG_LIST, &long_list));
    static const bool allow = !(sbx_env_tuning("SBX_GRAY_SHORT_ROWS") && atoi(sbx_env_tuning("SBX_GRAY_SHORT_ROWS")) == 0);
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
#define GRAY_ROWS(K, GRID, B, LV, ...)                                                                              \
  do {                                                                                                             \
    if (wshift >= 0)                                                                                               \
      SBX_KLAUNCH(h, SBX_K_GRAY, (K<B, LV, true>), dim3(GRID), dim3(256), rp, cl, n, (uint32_t)width, magic,        \
                  (uint32_t)band, wshift, bits, nnz_threshold, (X *)degree_out, keys, cnt, nlong, long_list,  \
                  ##__VA_ARGS__);                                                                                  \
    else                                                                                                           \
      SBX_KLAUNCH(h, SBX_K_GRAY, (K<B, LV, false>), dim3(GRID), dim3(256), rp, cl, n, (uint32_t)width, magic,       \
                  (uint32_t)band, 0, bits, nnz_threshold, (X *)degree_out, keys, cnt, nlong, long_list,       \
                  ##__VA_ARGS__);                                                                                  \
  } while (0)
#define GRAY_ROWS_BY_LEVELS(K, GRID, ...)                                                                           \
  do {                                                                                                             \
    if (bits <= 32) {                                                                                              \
      if (lv <= 1) GRAY_ROWS(K, GRID, uint32_t, 1, ##__VA_ARGS__);                                                 \
      else if (lv <= 3) GRAY_ROWS(K, GRID, uint32_t, 3, ##__VA_ARGS__);                                            \
      else GRAY_ROWS(K, GRID, uint32_t, 5, ##__VA_ARGS__);                                                         \
    } else {                                                                                                       \
      if (lv <= 1) GRAY_ROWS(K, GRID, unsigned long long, 1, ##__VA_ARGS__);                                       \
      else GRAY_ROWS(K, GRID, unsigned long long, 2, ##__VA_ARGS__); /* 64 blocks: thr <= 64 / 64 */               \
    }                                                                                                              \
  } while (0)
      static const bool try_banded = !(sbx_env_tuning("SBX_GRAY_BANDED_FIRST") && atoi(sbx_env_tuning("SBX_GRAY_BANDED_FIRST")) == 0);
      GrayBoth hb;
      hb.nlong = 0, hb.pad = GR_POWER_LAW;
      // the counters (what the later kernels of the banded path added) + k_gray_rows_short's GR_EARLY copies, summed here
      auto fetch_both = [&]() -> int {
        struct { GrayBoth b; unsigned fill[24 - 2]; GrayCounts early[GR_EARLY * 4]; } head;
        static_assert(sizeof(head) == offsetof(GrayAll, flags), "the head of GrayAll");
        SBX_TRY(sbx_readback(h, &head, all, sizeof(head)));
        hb = head.b;
        for (int i = 0; i < GR_EARLY; i++) {
          hb.c.nnz_sparse += head.early[4 * i].nnz_sparse, hb.c.diag_sparse += head.early[4 * i].diag_sparse;
          hb.c.nnz_dense += head.early[4 * i].nnz_dense, hb.c.diag_dense += head.early[4 * i].diag_dense;
        }
        return SBX_OK;
      };
      if (try_banded) {
        GRAY_ROWS_BY_LEVELS(k_gray_rows_short, grid);
        SBX_LAUNCH_CHECK(h);
        // one read-back: how many long rows the kernel met and — final if there were none — the band counters
        SBX_TRY(fetch_both());
      }
      const int wsh = wshift >= 0 ? wshift : -1;
      // the lists of k_gray_rows_medium: a row above GR_SHORT_MAX entries is at least one unit, every further unit is
      // GU_SIZE entries of it; a row of k > 1 units has more than (k - 1) GU_SIZE entries, and k <= 2 (k - 1)
      const size_t max_units = (size_t)(nnz / (GR_SHORT_MAX + 1) < n ? nnz / (GR_SHORT_MAX + 1) : n) + (size_t)(nnz / GU_SIZE) + 1;
      const size_t max_slots = 2 * (size_t)(nnz / GU_SIZE) + 2, max_mrows = (size_t)(nnz / GU_SIZE) + 1;
      int4 *units = nullptr, *mrows = nullptr;
      GrayLists *lc = nullptr;
      unsigned *partial = nullptr;
#define GRAY_MEDIUM(COUNTS, NSPREAD)                                                                                 \
  do {                                                                                                             \
    if (wshift >= 0)                                                                                               \
      SBX_KLAUNCH(h, SBX_K_GRAY, k_gray_rows_medium<true>, dim3((unsigned)h->num_cus * 8), dim3(256), rp, cl,       \
                  (const int4 *)units, (const GrayLists *)lc, (uint32_t)width, magic, wsh, (uint32_t)band, bits,     \
                  nnz_threshold, (int32_t)nnz, (X *)degree_out, keys, partial, COUNTS, NSPREAD);              \
    else                                                                                                           \
      SBX_KLAUNCH(h, SBX_K_GRAY, k_gray_rows_medium<false>, dim3((unsigned)h->num_cus * 8), dim3(256), rp, cl,      \
                  (const int4 *)units, (const GrayLists *)lc, (uint32_t)width, magic, wsh, (uint32_t)band, bits,     \
                  nnz_threshold, (int32_t)nnz, (X *)degree_out, keys, partial, COUNTS, NSPREAD);              \
  } while (0)
      auto alloc_lists = [&]() -> int {
        SBX_TRY(sbx_salloc(h, max_units, &units));
        SBX_TRY(sbx_salloc(h, max_mrows, &mrows));
        SBX_TRY(sbx_salloc(h, max_slots * GU_SLOT, &partial));
        lc = &all->lists;  // (zeroed with the rest at the start of the call)
        return SBX_OK;
      };
      if (hb.pad == GR_POWER_LAW) {
        // the kernel found the body of a power-law degree distribution and stopped (or was not tried): start over
        // with the kernels that cost per entry; no read-back until the end
        SBX_TRY(alloc_lists());
        GrayCounts *spread = all->spread;
        const int64_t brows = (int64_t)GB_ITERS * 4 * GB_GROUPS * 64;  // rows per workgroup
        const unsigned bgrid = (unsigned)((n + brows - 1) / brows);
        static const bool tiny_on = !(sbx_env_tuning("SBX_GRAY_TINY_ROWS") && atoi(sbx_env_tuning("SBX_GRAY_TINY_ROWS")) == 0);
        // (a row of up to GR_TINY entries has threshold 0 when it is shorter than `resolution` or not above the nnz threshold)
        if (tiny_on && (bits > GR_TINY || nnz_threshold >= GR_TINY)) {
          int32_t *mid_list = nullptr;
          SBX_TRY(sbx_salloc(h, (size_t)n, &mid_list));
#define GRAY_TINY(B)                                                                                                  \
  do {                                                                                                               \
    if (wshift >= 0)                                                                                                 \
      SBX_KLAUNCH(h, SBX_K_GRAY, (k_gray_rows_tiny<B, true>), dim3(bgrid), dim3(256), rp, cl, n, (int32_t)nnz,        \
                  (uint32_t)width, magic, (uint32_t)band, wshift, nnz_threshold, (X *)degree_out, keys, spread, \
                  units, mrows, lc, mid_list, &all->mid_count);                                                      \
    else                                                                                                             \
      SBX_KLAUNCH(h, SBX_K_GRAY, (k_gray_rows_tiny<B, false>), dim3(bgrid), dim3(256), rp, cl, n, (int32_t)nnz,       \
                  (uint32_t)width, magic, (uint32_t)band, 0, nnz_threshold, (X *)degree_out, keys, spread, units, \
                  mrows, lc, mid_list, &all->mid_count);                                                             \
  } while (0)
          if (bits <= 32) GRAY_TINY(uint32_t);
          else GRAY_TINY(unsigned long long);
#undef GRAY_TINY
#define GRAY_LISTED(B, LV)                                                                                            \
  do {                                                                                                               \
    if (wshift >= 0)                                                                                                 \
      SBX_KLAUNCH(h, SBX_K_GRAY, (k_gray_rows_listed<B, LV, true>), dim3((unsigned)h->num_cus * 8), dim3(256), rp, cl, \
                  (const int32_t *)mid_list, (const unsigned *)&all->mid_count, (int32_t)nnz, (uint32_t)width, magic, \
                  (uint32_t)band, wshift, bits, nnz_threshold, (X *)degree_out, keys, spread);                 \
    else                                                                                                             \
      SBX_KLAUNCH(h, SBX_K_GRAY, (k_gray_rows_listed<B, LV, false>), dim3((unsigned)h->num_cus * 8), dim3(256), rp,   \
                  cl, (const int32_t *)mid_list, (const unsigned *)&all->mid_count, (int32_t)nnz, (uint32_t)width,   \
                  magic, (uint32_t)band, 0, bits, nnz_threshold, (X *)degree_out, keys, spread);               \
  } while (0)
          if (bits <= 32) {
            if (lv <= 1) GRAY_LISTED(uint32_t, 1);
            else if (lv <= 3) GRAY_LISTED(uint32_t, 3);
            else GRAY_LISTED(uint32_t, 5);
          } else {
            if (lv <= 1) GRAY_LISTED(unsigned long long, 1);
            else GRAY_LISTED(unsigned long long, 2);
          }
#undef GRAY_LISTED
        } else {
          GrayCounts *cnt = spread;  // (what the launch macro passes as the counters)
          GRAY_ROWS_BY_LEVELS(k_gray_rows_balanced, bgrid, units, mrows, lc);
        }
        GRAY_MEDIUM(spread, (unsigned)GR_SPREAD);
        SBX_KLAUNCH(h, SBX_K_GRAY, k_gray_units_finish, dim3((unsigned)h->num_cus * 4), dim3(256), rp,
                    (const int4 *)mrows, (const GrayLists *)lc, (const unsigned *)partial, bits, nnz_threshold,
                    (X *)degree_out, keys, spread, (unsigned)GR_SPREAD);
        SBX_KLAUNCH(h, SBX_K_GRAY, k_gray_fold, dim3(1), dim3(64), (const GrayCounts *)spread, &all->total2);
        SBX_LAUNCH_CHECK(h);
        SBX_TRY(sbx_readback(h, &hb.c, &all->total2, sizeof(GrayCounts)));
        hb.nlong = 0, hb.pad = 0;
      } else if (hb.nlong <= (unsigned)GR_LONG_LIST && (hb.nlong || hb.pad)) {
        const unsigned hlong = hb.nlong;
        if (hb.pad) {  // a few rows of GR_SHORT_MAX + 1 .. GR_MED_MAX entries: listed, then a wave per unit
          SBX_TRY(alloc_lists());
          SBX_KLAUNCH(h, SBX_K_GRAY, k_gray_list_medium, dim3((unsigned)((n + GL_ROWS - 1) / GL_ROWS)), dim3(256), rp, n,
                      units, mrows, lc);
          GRAY_MEDIUM(cnt, 1u);
          SBX_KLAUNCH(h, SBX_K_GRAY, k_gray_units_finish, dim3(64), dim3(256), rp, (const int4 *)mrows,
                      (const GrayLists *)lc, (const unsigned *)partial, bits, nnz_threshold, (X *)degree_out, keys,
                      cnt, 1u);
        }
        if (hlong) {
          unsigned *slots = nullptr;
          SBX_TRY(sbx_salloc(h, (size_t)hlong * 65, &slots));
          SBX_HIP(h, hipMemsetAsync(slots, 0, sizeof(unsigned) * (size_t)hlong * 65, h->stream));
          SBX_KLAUNCH(h, SBX_K_GRAY, k_gray_long_rows, dim3(hlong * GR_PARTS), dim3(256), rp, cl,
                      (const int32_t *)long_list, (const unsigned *)nlong, (uint32_t)width, magic, (uint32_t)band, slots);
          SBX_KLAUNCH(h, SBX_K_GRAY, k_gray_long_finish, dim3((hlong + 3) / 4), dim3(256), rp,
                      (const int32_t *)long_list, (const unsigned *)nlong, (const unsigned *)slots, bits, nnz_threshold,
                      (X *)degree_out, keys, cnt);
        }
        SBX_LAUNCH_CHECK(h);
        SBX_TRY(fetch_both());
      }
#undef GRAY_MEDIUM
#undef GRAY_ROWS_BY_LEVELS
#undef GRAY_ROWS
      if (hb.nlong <= (unsigned)GR_LONG_LIST) {  // (the power-law path lists nothing there)
        SBX_PROF_BYTES(h, SBX_K_GRAY, 4 * nnz + 16 * n + 4);
        counts_host[0] = (int64_t)hb.c.nnz_sparse;
        counts_host[1] = (int64_t)hb.c.diag_sparse;
        counts_host[2] = (int64_t)hb.c.nnz_dense;
        counts_host[3] = (int64_t)hb.c.diag_dense;
        return SBX_OK;
      }
      // more rows above GR_MED_MAX entries than the list holds: start over with the tile kernel
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
              rp, n, nnz, ntiles, (X *)degree_out, keys, tile_row);
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
