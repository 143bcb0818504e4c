// sbx_convert.hip — COO/CSR constructor checks, COO sort and the COO<->CSR
// conversion functions as HIP kernels for gfx950.
//
//   A1  format/coo.cc:96-157           sbx_coo_is_sorted, sbx_coo_sort
//   A2  converter_order_two.cc:163-246 sbx_coo_to_csr   (copy + move)
//   A3  converter_order_two.cc:72-160  sbx_csr_to_coo   (copy + move)
//   A4  format/csr.cc:102-116          sbx_csr_rows_sorted (the sort itself: sbx_permute.hip)
//   A14 converter_order_two.cc:21-70   sbx_coo_to_csc   (stable counting sort by column)
//   A15 converter_order_two.cc:120-128 sbx_csr_to_csc   (CSR -> COO -> CSC)
//
// All three conversions are single-pass streaming kernels (HBM-bound):
//   COO->CSR  row_ptr comes from row-boundary detection on the row-sorted row[]
//             (identical to exclusive_scan(histogram) there); col/val are copied
//             in the same pass with 16-byte accesses.  Unsorted row[] (only
//             reachable with ignore_sort=true) takes the histogram+scan path.
//   CSR->COO  each 2048-nnz tile finds its row span with a wave-wide 64-ary
//             search, scatters row heads into LDS, max-scans them and streams
//             row ids + col/val copies out.
#include "sbx_device.h"
#include "sbx_internal.h"

namespace {

constexpr int CV_THREADS = 256;

// 16-byte vector for streaming copies
struct alignas(16) vec16 {
  uint32_t x, y, z, w;
};

// Workgroup-cooperative copy of bytes [b0,b1) (both multiples of 4).
template <bool ALIGNED16>
__device__ __forceinline__ void block_copy_bytes(char *__restrict__ dst, const char *__restrict__ src, int64_t b0,
                                                 int64_t b1) {
  if (ALIGNED16) {
    const int64_t nvec = (b1 - b0) >> 4;
    const vec16 *s = (const vec16 *)(src + b0);
    vec16 *d = (vec16 *)(dst + b0);
    for (int64_t i = threadIdx.x; i < nvec; i += blockDim.x) d[i] = s[i];
    const int64_t tail = b0 + (nvec << 4);
    const int64_t nw = (b1 - tail) >> 2;
    if ((int64_t)threadIdx.x < nw)
      ((uint32_t *)(dst + tail))[threadIdx.x] = ((const uint32_t *)(src + tail))[threadIdx.x];
  } else {
    const int64_t nw = (b1 - b0) >> 2;
    const uint32_t *s = (const uint32_t *)(src + b0);
    uint32_t *d = (uint32_t *)(dst + b0);
    for (int64_t i = threadIdx.x; i < nw; i += blockDim.x) d[i] = s[i];
  }
}

// ------------------------------------------------------------------ A1 check
// flags[0] = some record is out of (row, column) order; flags[1] = some coordinate lies outside [0, n) x [0, m) (only
// looked for when n >= 0: the constructor sort's hybrid path packs coordinates into bit fields sized by n and m and
// takes malformed input — which neither this library nor the reference validates — through the plain sort instead)
template <typename I>
__global__ __launch_bounds__(CV_THREADS) void k_coo_is_sorted(const I *__restrict__ row, const I *__restrict__ col,
                                                              int64_t nnz, int *__restrict__ flags, int64_t n, int64_t m) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  bool bad = false, outside = false;
  for (; i < nnz; i += stride) {
    const I pr = i ? row[i - 1] : (I)0, pc = i ? col[i - 1] : (I)0;
    const I r = row[i], c = col[i];
    bad |= (pr > r) || (pr == r && pc > c);
    if (n >= 0) outside |= r < 0 || (int64_t)r >= n || c < 0 || (int64_t)c >= m;
  }
  // (plain stores of the same value: on shuffled input every wave raises the flag, and 10^5 atomics on one word are 0.3 ms)
  if (__any(bad) && sbx_lane() == 0) flags[0] = 1;
  if (__any(outside) && sbx_lane() == 0) flags[1] = 1;
}

// ------------------------------------------------------------------ A1 sort helpers
template <typename I>
__global__ __launch_bounds__(CV_THREADS) void k_pack_rc(const I *__restrict__ row, const I *__restrict__ col,
                                                        uint64_t *__restrict__ key, int64_t nnz) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < nnz; i += stride) key[i] = ((uint64_t)(uint32_t)row[i] << 32) | (uint64_t)(uint32_t)col[i];
}
template <int VB> struct ValT { typedef uint32_t type; };
template <> struct ValT<8> { typedef uint64_t type; };

struct NestGuard {  // the conversions below call other entry points: keep this call's scratch alive
  sbx_handle_t h;
  explicit NestGuard(sbx_handle_t h) : h(h) { h->nest++; }
  ~NestGuard() { h->nest--; }
};

// hybrid COO sort (below): after the records are grouped by row >> s, the rest of the key — the row's low s bits and
// the column — fits 32 bits and is sorted per group in LDS
__global__ __launch_bounds__(CV_THREADS) void k_coo_bucket_keys(const int32_t *__restrict__ row,
                                                                const int32_t *__restrict__ col, int s, int colbits,
                                                                int32_t *__restrict__ hi, int32_t *__restrict__ key,
                                                                int64_t nnz) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const uint32_t smask = (1u << s) - 1u;
  for (; i < nnz; i += stride) {
    const uint32_t r = (uint32_t)row[i];
    hi[i] = (int32_t)(r >> s);
    key[i] = (int32_t)(((r & smask) << colbits) | (uint32_t)col[i]);
  }
}
template <int VB>
__global__ __launch_bounds__(CV_THREADS) void k_coo_bucket_unpack(const int32_t *__restrict__ key,
                                                                  const char *__restrict__ vsorted, int s, int colbits,
                                                                  int32_t *__restrict__ row, int32_t *__restrict__ col,
                                                                  char *__restrict__ val, int64_t nnz) {
  typedef typename ValT<VB>::type V;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const uint32_t cmask = (1u << colbits) - 1u;
  for (; i < nnz; i += stride) {
    const uint32_t k = (uint32_t)key[i];
    // (the group sort moved records inside their group only: the row's high bits at this position are still right)
    row[i] = (int32_t)(((uint32_t)row[i] >> s << s) | (k >> colbits));
    col[i] = (int32_t)(k & cmask);
    if (VB) ((V *)val)[i] = ((const V *)vsorted)[i];
  }
}

// ---- 64-bit index arrays: the same sort on packed keys (row << colbits | column; needs rowbits + colbits <= 64) ----
__global__ __launch_bounds__(CV_THREADS) void k_pack_rc64(const int64_t *__restrict__ row, const int64_t *__restrict__ col,
                                                          int colbits, uint64_t *__restrict__ key, int64_t nnz) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < nnz; i += stride) key[i] = (colbits < 64 ? (uint64_t)row[i] << colbits : 0ull) | (uint64_t)col[i];
}
template <int VB>
__global__ __launch_bounds__(CV_THREADS) void k_unpack_rc64(const uint64_t *__restrict__ key, const char *__restrict__ vsorted,
                                                            int colbits, int64_t *__restrict__ row,
                                                            int64_t *__restrict__ col, char *__restrict__ val, int64_t nnz) {
  typedef typename ValT<VB>::type V;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const uint64_t cmask = colbits >= 64 ? ~0ull : (1ull << colbits) - 1ull;
  for (; i < nnz; i += stride) {
    const uint64_t k = key[i];
    row[i] = (int64_t)(colbits < 64 ? k >> colbits : 0ull);
    col[i] = (int64_t)(k & cmask);
    if (VB) ((V *)val)[i] = ((const V *)vsorted)[i];  // (VB = 0: the values are already where they belong)
  }
}
// hybrid sort, 64-bit arrays: group id and in-group key of a record from its packed key (grouped by the digit passes)
__global__ __launch_bounds__(CV_THREADS) void k_coo_bucket_keys64(const uint64_t *__restrict__ packed, int lowbits,
                                                                  int32_t *__restrict__ hi, int32_t *__restrict__ key,
                                                                  int64_t nnz) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const uint64_t lmask = (1ull << lowbits) - 1ull;
  for (; i < nnz; i += stride) {
    const uint64_t k = packed[i];
    hi[i] = (int32_t)(k >> lowbits);
    key[i] = (int32_t)(k & lmask);
  }
}
template <int VB>
__global__ __launch_bounds__(CV_THREADS) void k_coo_bucket_unpack64(const int32_t *__restrict__ hi,
                                                                    const int32_t *__restrict__ key,
                                                                    const char *__restrict__ vsorted, int s, int colbits,
                                                                    int64_t *__restrict__ row, int64_t *__restrict__ col,
                                                                    char *__restrict__ val, int64_t nnz) {
  typedef typename ValT<VB>::type V;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const uint32_t cmask = (1u << colbits) - 1u;
  for (; i < nnz; i += stride) {
    const uint32_t k = (uint32_t)key[i];
    row[i] = ((int64_t)(uint32_t)hi[i] << s) | (int64_t)(k >> colbits);
    col[i] = (int64_t)(k & cmask);
    if (VB) ((V *)val)[i] = ((const V *)vsorted)[i];
  }
}

template <typename I>
__global__ __launch_bounds__(CV_THREADS) void k_unpack_rc(const uint64_t *__restrict__ key, I *__restrict__ row,
                                                          I *__restrict__ col, int64_t nnz) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < nnz; i += stride) {
    const uint64_t k = key[i];
    row[i] = (I)(uint32_t)(k >> 32);
    col[i] = (I)(uint32_t)k;
  }
}

// ------------------------------------------------------------------ A2 COO -> CSR
constexpr int GAP_INLINE = 64;  // longer runs of empty rows go to the gap queue

struct GapEntry {
  int64_t first, last, value;  // row_ptr[first..last] = value
};

template <typename I>
__device__ __forceinline__ void emit_row_starts(I *__restrict__ rp, int64_t prev, int64_t r, int64_t value,
                                                GapEntry *__restrict__ gaps, unsigned *__restrict__ ngaps,
                                                unsigned gap_cap) {
  // rows prev+1 .. r start at `value`
  const int64_t len = r - prev;
  if (len <= 0) return;
  if (len <= GAP_INLINE) {
    for (int64_t q = prev + 1; q <= r; q++) rp[q] = (I)value;
  } else {
    const unsigned slot = atomicAdd(ngaps, 1u);
    if (slot < gap_cap) {
      gaps[slot].first = prev + 1;
      gaps[slot].last = r;
      gaps[slot].value = value;
    }
  }
}

// One thread = 4 consecutive nonzeros.  MOVE: only row_ptr is produced.  I: the id arrays' word, O: row_ptr's (the
// tuples with sizeof(IDType) != sizeof(NNZType): SBX_I32_N64).
template <typename I, typename O, int VB, bool MOVE, bool ALIGNED16>
__global__ __launch_bounds__(CV_THREADS) void k_coo_to_csr(const I *__restrict__ row, const I *__restrict__ col,
                                                           const char *__restrict__ val, O *__restrict__ rp,
                                                           I *__restrict__ col_out, char *__restrict__ val_out,
                                                           int64_t n, int64_t nnz, GapEntry *__restrict__ gaps,
                                                           unsigned *__restrict__ ngaps, unsigned gap_cap,
                                                           int *__restrict__ unsorted) {
  const int64_t nquads = (nnz + 3) >> 2;
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  bool bad = false;
  for (; q < nquads; q += stride) {
    const int64_t i0 = q << 2;
    I r[4];
    const bool full = (i0 + 4 <= nnz);
    if (full && ALIGNED16 && sizeof(I) == 8) {
      const vec16 v0 = *(const vec16 *)(row + i0), v1 = *(const vec16 *)(row + i0 + 2);
      r[0] = (I)(((uint64_t)v0.y << 32) | v0.x), r[1] = (I)(((uint64_t)v0.w << 32) | v0.z);
      r[2] = (I)(((uint64_t)v1.y << 32) | v1.x), r[3] = (I)(((uint64_t)v1.w << 32) | v1.z);
      if (!MOVE) {
        *(vec16 *)(col_out + i0) = *(const vec16 *)(col + i0);
        *(vec16 *)(col_out + i0 + 2) = *(const vec16 *)(col + i0 + 2);
        if (VB == 4) *(vec16 *)(val_out + i0 * 4) = *(const vec16 *)(val + i0 * 4);
        if (VB == 8) {
          *(vec16 *)(val_out + i0 * 8) = *(const vec16 *)(val + i0 * 8);
          *(vec16 *)(val_out + i0 * 8 + 16) = *(const vec16 *)(val + i0 * 8 + 16);
        }
      }
    } else if (full && ALIGNED16 && sizeof(I) == 4) {
      const vec16 v = *(const vec16 *)(row + i0);
      r[0] = (I)v.x; r[1] = (I)v.y; r[2] = (I)v.z; r[3] = (I)v.w;
      if (!MOVE) {
        *(vec16 *)(col_out + i0) = *(const vec16 *)(col + i0);
        if (VB == 4) *(vec16 *)(val_out + i0 * 4) = *(const vec16 *)(val + i0 * 4);
        if (VB == 8) {
          *(vec16 *)(val_out + i0 * 8) = *(const vec16 *)(val + i0 * 8);
          *(vec16 *)(val_out + i0 * 8 + 16) = *(const vec16 *)(val + i0 * 8 + 16);
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int64_t i = i0 + k;
        r[k] = i < nnz ? row[i] : (I)0;
        if (!MOVE && i < nnz) {
          col_out[i] = col[i];
          if (VB == 4) ((uint32_t *)val_out)[i] = ((const uint32_t *)val)[i];
          if (VB == 8) ((uint64_t *)val_out)[i] = ((const uint64_t *)val)[i];
        }
      }
    }
    int64_t prev = i0 ? (int64_t)row[i0 - 1] : -1;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int64_t i = i0 + k;
      if (i < nnz) {
        const int64_t cur = (int64_t)r[k];
        bad |= cur < prev;
        emit_row_starts<O>(rp, prev, cur, i, gaps, ngaps, gap_cap);
        prev = cur;
        if (i == nnz - 1) emit_row_starts<O>(rp, cur, n, nnz, gaps, ngaps, gap_cap);
      }
    }
  }
  if (__any(bad) && sbx_lane() == 0) *unsorted = 1;
}

template <typename I>
__global__ __launch_bounds__(CV_THREADS) void k_fill_gaps(I *__restrict__ rp, const GapEntry *__restrict__ gaps,
                                                          const unsigned *__restrict__ ngaps, unsigned gap_cap) {
  unsigned cnt = *ngaps;
  if (cnt > gap_cap) cnt = gap_cap;
  for (unsigned g = blockIdx.x; g < cnt; g += gridDim.x) {
    const GapEntry e = gaps[g];
    for (int64_t q = e.first + threadIdx.x; q <= e.last; q += blockDim.x) rp[q] = (I)e.value;
  }
}

// ------------------------------------------------------------------ A3 CSR -> COO
constexpr int EX_ITEMS = 8;
constexpr int EX_TILE = CV_THREADS * EX_ITEMS;  // 2048 nonzeros per workgroup

// first and last row of every EX_TILE-wide tile of the nonzeros, one thread per tile (large inputs: the two searches in
// row_ptr are four dependent rounds of loads when a workgroup does them itself, and with tens of waves of workgroups
// per CU that is most of a tile's life)
template <typename I>
__global__ __launch_bounds__(CV_THREADS) void k_ex_tile_spans(const I *__restrict__ rp, int64_t n, int64_t nnz,
                                                              int64_t tiles, int2 *__restrict__ span) {
  const int64_t t = (int64_t)blockIdx.x * CV_THREADS + threadIdx.x;
  if (t >= tiles) return;
  const int64_t t0 = t * EX_TILE, t1 = (t0 + EX_TILE < nnz) ? t0 + EX_TILE : nnz;
  auto last_le = [&](int64_t v) {  // last row r with rp[r] <= v
    int64_t lo = 0, hi = n + 1;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)rp[mid] > v) hi = mid; else lo = mid + 1;
    }
    return lo - 1;
  };
  span[t] = make_int2((int)last_le(t0), (int)last_le(t1 - 1));
}
constexpr int64_t EX_SPAN_MIN_TILES = 8192;  // from 16 M nonzeros on

template <typename I, typename O, int VB, bool MOVE, bool ALIGNED16>
__global__ __launch_bounds__(CV_THREADS) void k_csr_to_coo(const O *__restrict__ rp, const I *__restrict__ col,
                                                           const char *__restrict__ val, I *__restrict__ row_out,
                                                           I *__restrict__ col_out, char *__restrict__ val_out,
                                                           int64_t n, int64_t nnz, const int2 *__restrict__ span) {
  __shared__ int s_head[EX_TILE];
  __shared__ int64_t s_span[2];
  __shared__ int s_wmax[CV_THREADS / 64];
  const int tid = threadIdx.x;
  const int64_t t0 = (int64_t)blockIdx.x * EX_TILE;
  const int64_t t1 = (t0 + EX_TILE < nnz) ? t0 + EX_TILE : nnz;
  const int cnt = (int)(t1 - t0);
#pragma unroll
  for (int k = 0; k < EX_ITEMS; k++) s_head[k * CV_THREADS + tid] = 0;
  // row span of the tile: r_lo = last row with rp[r] <= t0, r_hi likewise for t1-1 (precomputed for large inputs)
  if (span) {
    if (tid == 0) {
      const int2 sp = span[blockIdx.x];
      s_span[0] = sp.x;
      s_span[1] = sp.y;
    }
  } else if (tid < 64) {
    const int64_t lo = sbx_wave_upper_bound<O>(rp, n + 1, (O)t0) - 1;
    if (tid == 0) s_span[0] = lo;
  } else if (tid < 128) {
    const int64_t hi = sbx_wave_upper_bound<O>(rp, n + 1, (O)(t1 - 1)) - 1;
    if (tid == 64) s_span[1] = hi;
  }
  __syncthreads();
  const int64_t r_lo = s_span[0], r_hi = s_span[1];
  // scatter row heads (relative row number at the row's first position in the tile)
  for (int64_t r = r_lo + 1 + tid; r <= r_hi; r += CV_THREADS) {
    const int p = (int)((int64_t)rp[r] - t0);
    atomicMax(&s_head[p], (int)(r - r_lo));
  }
  // streaming copies overlap with the LDS phase
  if (!MOVE) {
    block_copy_bytes<ALIGNED16>((char *)col_out, (const char *)col, t0 * (int64_t)sizeof(I), t1 * (int64_t)sizeof(I));
    if (VB) block_copy_bytes<ALIGNED16>(val_out, val, t0 * VB, t1 * VB);
  }
  __syncthreads();
  // inclusive max-scan, thread-blocked (8 consecutive positions per thread)
  int h[EX_ITEMS];
  int run = 0;
#pragma unroll
  for (int k = 0; k < EX_ITEMS; k++) {
    h[k] = s_head[tid * EX_ITEMS + k];
    run = h[k] > run ? h[k] : run;
    h[k] = run;
  }
  const int inc = sbx_wave_inclusive_max(run);
  int excl = sbx_wave_shift_up1(inc, 0);
  if (sbx_lane() == 63) s_wmax[tid >> 6] = inc;
  __syncthreads();
  int woff = 0;
  for (int w = 0; w < (tid >> 6); w++) woff = s_wmax[w] > woff ? s_wmax[w] : woff;
  const int before = excl > woff ? excl : woff;
  const int64_t base = t0 + (int64_t)tid * EX_ITEMS;
  I out[EX_ITEMS];
#pragma unroll
  for (int k = 0; k < EX_ITEMS; k++) out[k] = (I)(r_lo + (h[k] > before ? h[k] : before));
  if (tid * EX_ITEMS + EX_ITEMS <= cnt && ALIGNED16 && sizeof(I) == 4) {
    vec16 a, b;
    a.x = (uint32_t)out[0]; a.y = (uint32_t)out[1]; a.z = (uint32_t)out[2]; a.w = (uint32_t)out[3];
    b.x = (uint32_t)out[4]; b.y = (uint32_t)out[5]; b.z = (uint32_t)out[6]; b.w = (uint32_t)out[7];
    *(vec16 *)(row_out + base) = a;
    *(vec16 *)(row_out + base + 4) = b;
  } else {
#pragma unroll
    for (int k = 0; k < EX_ITEMS; k++)
      if (tid * EX_ITEMS + k < cnt) row_out[base + k] = out[k];
  }
}

// A4 check (format/csr.cc:102-116), nnz-parallel so power-law rows stay balanced:
// row heads of the tile are flagged in LDS, then every entry is compared with
// its predecessor (or with 0 at a row head).
template <typename I>
__global__ __launch_bounds__(CV_THREADS) void k_csr_rows_sorted(const I *__restrict__ rp, const I *__restrict__ col,
                                                                int64_t n, int64_t nnz, int *__restrict__ unsorted) {
  __shared__ int s_head[EX_TILE];
  __shared__ int64_t s_span[2];
  const int tid = threadIdx.x;
  const int64_t t0 = (int64_t)blockIdx.x * EX_TILE;
  const int64_t t1 = (t0 + EX_TILE < nnz) ? t0 + EX_TILE : nnz;
#pragma unroll
  for (int k = 0; k < EX_ITEMS; k++) s_head[k * CV_THREADS + tid] = 0;
  if (tid < 64) {
    const int64_t lo = sbx_wave_upper_bound<I>(rp, n + 1, (I)t0) - 1;
    if (tid == 0) s_span[0] = lo;
  } else if (tid < 128) {
    const int64_t hi = sbx_wave_upper_bound<I>(rp, n + 1, (I)(t1 - 1)) - 1;
    if (tid == 64) s_span[1] = hi;
  }
  __syncthreads();
  const int64_t r_lo = s_span[0], r_hi = s_span[1];
  for (int64_t r = r_lo + 1 + tid; r <= r_hi; r += CV_THREADS) s_head[(int)((int64_t)rp[r] - t0)] = 1;
  if (tid == 0 && (int64_t)rp[r_lo] == t0) s_head[0] = 1;
  __syncthreads();
  bool bad = false;
#pragma unroll
  for (int k = 0; k < EX_ITEMS; k++) {
    const int64_t j = t0 + k * CV_THREADS + tid;
    if (j < t1) {
      const I c = col[j];
      const I p = s_head[k * CV_THREADS + tid] ? (I)0 : col[j - 1];
      bad |= c < p;
    }
  }
  if (__any(bad) && sbx_lane() == 0) *unsorted = 1;
}

static bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

// ============================================================================
// C ABI
// ============================================================================
#define SBX_REQUIRE(h, cond, msg)                       \
  do {                                                  \
    if (!(cond)) SBX_FAIL(h, SBX_ERR_BAD_ARG, "%s: %s", __func__, msg); \
  } while (0)

template <typename I>
static int coo_is_sorted_typed(sbx_handle_t h, int64_t nnz, const void *row, const void *col, int *sorted_host,
                               int64_t n = -1, int64_t m = -1, int *in_range_host = nullptr) {
  SBX_TRY(sbx_arena_begin(h));
  *sorted_host = 1;
  if (in_range_host) *in_range_host = 1;
  if (nnz == 0) return SBX_OK;
  int *flag = nullptr;
  SBX_TRY(sbx_salloc(h, 2, &flag));
  SBX_HIP(h, hipMemsetAsync(flag, 0, 2 * sizeof(int), h->stream));
  SBX_KLAUNCH(h, SBX_K_CHECK, k_coo_is_sorted<I>, dim3(sbx_grid_for(nnz, CV_THREADS, 8192)), dim3(CV_THREADS),
              (const I *)row, (const I *)col, nnz, flag, n, m);
  SBX_LAUNCH_CHECK(h);
  int f[2] = {0, 0};
  SBX_TRY(sbx_readback(h, f, flag, 2 * sizeof(int)));
  *sorted_host = !f[0];
  if (in_range_host) *in_range_host = !f[1];
  return SBX_OK;
}

extern "C" int sbx_coo_is_sorted(sbx_handle_t h, sbx_index_type it, int64_t nnz, const void *row, const void *col,
                                 int *sorted_host) {
  if (!h) return SBX_ERR_BAD_ARG;
  if (it == SBX_I32_N64) it = SBX_I32;  // (no offset array)
  SBX_REQUIRE(h, sorted_host && nnz >= 0 && (nnz == 0 || (row && col)), "bad argument");
  if (it == SBX_I64) return coo_is_sorted_typed<int64_t>(h, nnz, row, col, sorted_host);
  return coo_is_sorted_typed<int32_t>(h, nnz, row, col, sorted_host);
}

template <typename I>
static int csr_rows_sorted_typed(sbx_handle_t h, int64_t n, const void *row_ptr, const void *col, int *sorted_host) {
  SBX_TRY(sbx_arena_begin(h));
  *sorted_host = 1;
  if (n == 0) return SBX_OK;
  I nnz_i = 0;
  SBX_TRY(sbx_readback(h, &nnz_i, (const I *)row_ptr + n, sizeof(I)));  // nnz_ = row_ptr[n], csr.cc:86
  const int64_t nnz = (int64_t)nnz_i;
  if (nnz <= 0) return SBX_OK;
  int *flag = nullptr;
  SBX_TRY(sbx_salloc(h, 1, &flag));
  SBX_HIP(h, hipMemsetAsync(flag, 0, sizeof(int), h->stream));
  SBX_KLAUNCH(h, SBX_K_CHECK, k_csr_rows_sorted<I>, dim3((unsigned)((nnz + EX_TILE - 1) / EX_TILE)), dim3(CV_THREADS),
              (const I *)row_ptr, (const I *)col, n, nnz, flag);
  SBX_LAUNCH_CHECK(h);
  int f = 0;
  SBX_TRY(sbx_readback(h, &f, flag, sizeof(int)));
  *sorted_host = !f;
  return SBX_OK;
}

extern "C" int sbx_csr_rows_sorted(sbx_handle_t h, sbx_index_type it, int64_t n, const void *row_ptr,
                                   const void *col, int *sorted_host) {
  if (!h) return SBX_ERR_BAD_ARG;
  if (it == SBX_I32_N64) return sbx_mixed_csr_rows_sorted(h, n, row_ptr, col, sorted_host);
  SBX_REQUIRE(h, sorted_host && n >= 0 && row_ptr, "bad argument");
  SBX_REQUIRE(h, n < ((int64_t)1 << 31) - 1, "row count exceeds int32");
  if (it == SBX_I64) return csr_rows_sorted_typed<int64_t>(h, n, row_ptr, col, sorted_host);
  return csr_rows_sorted_typed<int32_t>(h, n, row_ptr, col, sorted_host);
}

// format/coo.cc:96-157 for 64-bit index arrays, native: coordinates of any size inside [0, n) x [0, m) as long as a
// record's (row, column) packs into 64 bits.  One pack pass (16 -> 8 bytes per record), then the 32-bit sort's plan on
// the packed keys: digit passes over the row's leading bits + the LDS sort of the groups when the rest of the key fits
// 31 bits, the plain LSD sort otherwise; the last kernel writes 64-bit rows and columns back where they came from.
static int coo_sort_i64(sbx_handle_t h, sbx_value_type vt, int64_t n, int64_t m, int64_t nnz, void *row, void *col,
                        void *val) {
  const int vb = val ? sbx_value_bytes(vt) : 0;
  SBX_REQUIRE(h, vb >= 0, "unknown value type");
  if (nnz <= 1) return SBX_OK;
  int sorted = 0, in_range = 1;
  SBX_TRY(coo_is_sorted_typed<int64_t>(h, nnz, row, col, &sorted, n, m, &in_range));
  if (sorted) return SBX_OK;
  // (coordinates outside the matrix: the narrowing path's plain sort takes them if they fit 31 bits, and refuses the rest)
  if (!in_range) return sbx_i64_coo_sort(h, vt, n, m, nnz, row, col, val);
  const int colbits = sbx_bits_for(m > 0 ? (uint64_t)(m - 1) : 0), rowbits = sbx_bits_for(n > 0 ? (uint64_t)(n - 1) : 0);
  if (colbits + rowbits > 64)
    SBX_FAIL(h, SBX_ERR_UNSUPPORTED, "sbx_coo_sort: %d row bits + %d column bits do not fit a 64-bit sort key", rowbits, colbits);
  SBX_TRY(sbx_arena_begin(h));
  NestGuard guard(h);  // (the nested calls below must not rewind the arena)
  uint64_t *ka = nullptr, *kb = nullptr;
  char *vtmp = nullptr;
  SBX_TRY(sbx_salloc(h, (size_t)nnz, &ka));
  SBX_TRY(sbx_salloc(h, (size_t)nnz, &kb));
  if (vb) SBX_TRY(sbx_salloc(h, (size_t)nnz * vb, &vtmp));
  const unsigned grid = sbx_grid_for(nnz, CV_THREADS, 8192);
  SBX_KLAUNCH(h, SBX_K_MISC, k_pack_rc64, dim3(grid), dim3(CV_THREADS), (const int64_t *)row, (const int64_t *)col, colbits,
              ka, nnz);
  SBX_LAUNCH_CHECK(h);
  int p = 2;
  if (rowbits > 16 && nnz / 65536 > 512) p = 3;
  const int s_bits = rowbits > 8 * p ? rowbits - 8 * p : 0;
  sbx_radix_pass msd[16], passes[16];
  const int np = sbx_radix_plan(0, colbits + rowbits, 0, 0, passes);
  const int np_msd = sbx_radix_plan(colbits + s_bits, colbits + rowbits, 0, 0, msd);
  int in_b = 0;
  if (np_msd >= 2 && np >= np_msd + 2 && s_bits + colbits <= 31 && nnz < ((int64_t)1 << 31)) {
    SBX_TRY(sbx_radix_sort(h, 8, vb, ka, kb, val, vtmp, nnz, msd, np_msd, &in_b));
    const uint64_t *grouped = in_b ? kb : ka;
    const char *vcur = in_b ? vtmp : (const char *)val;
    char *vother = in_b ? (char *)val : vtmp;
    int32_t *hk = nullptr, *bptr = nullptr;  // group ids | in-group keys
    SBX_TRY(sbx_salloc(h, (size_t)nnz * 2, &hk));
    int32_t *hi = hk, *key = hk + nnz, *ksorted = (int32_t *)(in_b ? ka : kb);
    const int64_t groups = ((n - 1) >> s_bits) + 1;
    SBX_TRY(sbx_salloc(h, (size_t)groups + 1, &bptr));
    SBX_KLAUNCH(h, SBX_K_MISC, k_coo_bucket_keys64, dim3(grid), dim3(CV_THREADS), grouped, s_bits + colbits, hi, key, nnz);
    SBX_TRY(sbx_coo_to_csr(h, SBX_I32, SBX_V_NONE, groups, groups, nnz, hi, nullptr, nullptr, bptr, nullptr, nullptr,
                           SBX_FLAG_MOVE | SBX_FLAG_ROWS_SORTED));
    SBX_TRY(sbx_sort_segments(h, vb, groups, (int64_t)1 << (s_bits + colbits), nnz, bptr, key, vcur, ksorted, vother));
    const bool copy = vb && vother != (char *)val;
#define UNPACK(VBX)                                                                                                     \
  SBX_KLAUNCH(h, SBX_K_MISC, k_coo_bucket_unpack64<VBX>, dim3(grid), dim3(CV_THREADS), (const int32_t *)hi,             \
              (const int32_t *)ksorted, (const char *)vother, s_bits, colbits, (int64_t *)row, (int64_t *)col, (char *)val, nnz)
    if (!copy) UNPACK(0);
    else if (vb == 4) UNPACK(4);
    else UNPACK(8);
#undef UNPACK
    SBX_LAUNCH_CHECK(h);
    return SBX_OK;
  }
  SBX_TRY(sbx_radix_sort(h, 8, vb, ka, kb, val, vtmp, nnz, passes, np, &in_b));
  const bool copy = vb && in_b;
#define UNPACK(VBX)                                                                                                   \
  SBX_KLAUNCH(h, SBX_K_MISC, k_unpack_rc64<VBX>, dim3(grid), dim3(CV_THREADS), (const uint64_t *)(in_b ? kb : ka),    \
              (const char *)vtmp, colbits, (int64_t *)row, (int64_t *)col, (char *)val, nnz)
  if (!copy) UNPACK(0);
  else if (vb == 4) UNPACK(4);
  else UNPACK(8);
#undef UNPACK
  SBX_LAUNCH_CHECK(h);
  return SBX_OK;
}

extern "C" int sbx_coo_sort(sbx_handle_t h, sbx_index_type it, sbx_value_type vt, int64_t n, int64_t m, int64_t nnz,
                            void *row, void *col, void *val) {
  if (!h) return SBX_ERR_BAD_ARG;
  if (it == SBX_I32_N64) it = SBX_I32;  // (no offset array)
  SBX_REQUIRE(h, nnz >= 0 && n >= 0 && m >= 0 && (nnz == 0 || (row && col)), "bad argument");
  if (it == SBX_I64) return coo_sort_i64(h, vt, n, m, nnz, row, col, val);
  const int vb = val ? sbx_value_bytes(vt) : 0;
  SBX_REQUIRE(h, vb >= 0, "unknown value type");
  if (nnz <= 1) return SBX_OK;
  int sorted = 0, in_range = 1;
  SBX_TRY(coo_is_sorted_typed<int32_t>(h, nnz, row, col, &sorted, n, m, &in_range));  // format/coo.cc:96-108
  if (sorted) return SBX_OK;
  SBX_TRY(sbx_arena_begin(h));
  uint64_t *ka = nullptr, *kb = nullptr;
  char *vtmp = nullptr;
  SBX_TRY(sbx_salloc(h, (size_t)nnz, &ka));
  SBX_TRY(sbx_salloc(h, (size_t)nnz, &kb));
  if (vb) SBX_TRY(sbx_salloc(h, (size_t)nnz * vb, &vtmp));
  sbx_radix_pass passes[16];
  const int colbits = sbx_bits_for(m > 0 ? (uint64_t)(m - 1) : 0), rowbits = sbx_bits_for(n > 0 ? (uint64_t)(n - 1) : 0);
  const int np = sbx_radix_plan(0, colbits, 32, 32 + rowbits, passes);
  {
    // Hybrid: two (three) digit passes group the records by the row's leading 16 (24) bits — stable, so every group
    // holds its records in input order — and the rest of the key, row low bits | column in 32 bits, is sorted per
    // group by the permute's LDS sort stage (stable: (key, position)); rows and columns are then unpacked in place.
    // 2 – 3 global passes + one LDS sort instead of 5 – 6 passes: C2B (10 M uniform records) 0.66 -> 0.52 ms, C3 (105 M
    // power-law records, three passes: groups = rows) 6.1 -> 5.8 ms.  SBX_COO_SORT_HYBRID=0: the plain LSD sort.
    // groups of 2^s rows must stay LDS-sized when the rows are skewed (a group above 8192 records takes the long-row
    // path, six more global passes over its records): two passes only while the AVERAGE group holds <= 512 records
    int p = 2;
    if (rowbits > 16 && nnz / 65536 > 512) p = 3;
    const int s_bits = rowbits > 8 * p ? rowbits - 8 * p : 0;
    sbx_radix_pass msd[16];
    const int np_msd = sbx_radix_plan(0, 0, 32 + s_bits, 32 + rowbits, msd);
    // (coordinates outside [0, n) x [0, m) would overflow the bit fields and the group table: the plain sort takes them)
    if (in_range && np_msd >= 2 && np >= np_msd + 2 && s_bits + colbits <= 31 && nnz < ((int64_t)1 << 31)) {
      NestGuard guard(h);  // (the nested conversions below must not rewind the arena)
      char *vtmp2 = nullptr;
      if (vb) SBX_TRY(sbx_salloc(h, (size_t)nnz * vb, &vtmp2));
      // the digit passes leave rows and columns where they came from and the values in the scratch buffer their last
      // pass does not read, so that the group sort below can write them straight back into the caller's array
      char *vgrouped = (np_msd & 1) ? vtmp : vtmp2;
      const sbx_radix_side side = {{col, row}, {val, nullptr}, true, false};
      if (s_bits == 0) {
        // The groups are the rows themselves (every row bit went through a digit pass): row[] is final behind the passes
        // and the in-group key is the column as it stands — no key kernel in front of the group sort and no unpack
        // behind it (0.40 + 0.35 ms of pure streaming on the 105 M-record sort).  The last pass leaves the columns in
        // scratch, the group sort writes them, sorted, into the caller's array.
        int32_t *clo = nullptr, *bptr = nullptr;
        SBX_TRY(sbx_salloc(h, (size_t)nnz, &clo));
        SBX_TRY(sbx_salloc(h, (size_t)n + 1, &bptr));
        const sbx_radix_side grouped = {{clo, row}, {vb ? vgrouped : nullptr, nullptr}, true, false};
        SBX_TRY(sbx_radix_sort_io(h, 8, vb, &side, ka, kb, vtmp, vtmp2, &grouped, nnz, msd, np_msd));
        SBX_TRY(sbx_coo_to_csr(h, SBX_I32, SBX_V_NONE, n, n, nnz, row, nullptr, nullptr, bptr, nullptr, nullptr,
                               SBX_FLAG_MOVE | SBX_FLAG_ROWS_SORTED));
        return sbx_sort_segments(h, vb, n, (int64_t)1 << colbits, nnz, bptr, clo, (const char *)vgrouped, (int32_t *)col,
                                 (char *)val);
      }
      const sbx_radix_side grouped = {{col, row}, {vb ? vgrouped : nullptr, nullptr}, true, false};
      SBX_TRY(sbx_radix_sort_io(h, 8, vb, &side, ka, kb, vtmp, vtmp2, &grouped, nnz, msd, np_msd));
      // ka / kb are free again: group ids | in-group keys in one, sorted keys in the other
      int32_t *hi = (int32_t *)ka, *key = hi + nnz, *ksorted = (int32_t *)kb, *bptr = nullptr;
      const int64_t groups = ((n - 1) >> s_bits) + 1;
      SBX_TRY(sbx_salloc(h, (size_t)groups + 1, &bptr));
      const unsigned grid = sbx_grid_for(nnz, CV_THREADS, 8192);
      SBX_KLAUNCH(h, SBX_K_MISC, k_coo_bucket_keys, dim3(grid), dim3(CV_THREADS), (const int32_t *)row,
                  (const int32_t *)col, s_bits, colbits, hi, key, nnz);
      SBX_TRY(sbx_coo_to_csr(h, SBX_I32, SBX_V_NONE, groups, groups, nnz, hi, nullptr, nullptr, bptr, nullptr, nullptr,
                             SBX_FLAG_MOVE | SBX_FLAG_ROWS_SORTED));
      SBX_TRY(sbx_sort_segments(h, vb, groups, (int64_t)1 << (s_bits + colbits), nnz, bptr, key, (const char *)vgrouped,
                                ksorted, (char *)val));
      SBX_KLAUNCH(h, SBX_K_MISC, k_coo_bucket_unpack<0>, dim3(grid), dim3(CV_THREADS), (const int32_t *)ksorted,
                  (const char *)nullptr, s_bits, colbits, (int32_t *)row, (int32_t *)col, (char *)nullptr, nnz);
      SBX_LAUNCH_CHECK(h);
      return SBX_OK;
    }
  }
  if (np >= 2) {
    // the first digit pass reads (row, col, val) where they are — the key is (row << 32) | col — and the last one
    // writes them back there: no pack kernel before and no unpack kernel behind the sort
    char *vtmp2 = nullptr;
    if (vb && np >= 3) SBX_TRY(sbx_salloc(h, (size_t)nnz * vb, &vtmp2));
    const sbx_radix_side side = {{col, row}, {val, nullptr}, true, false};
    return sbx_radix_sort_io(h, 8, vb, &side, ka, kb, vtmp, vtmp2, &side, nnz, passes, np);
  }
  const unsigned grid = sbx_grid_for(nnz, CV_THREADS, 8192);
  SBX_KLAUNCH(h, SBX_K_MISC, k_pack_rc<int32_t>, dim3(grid), dim3(CV_THREADS), (const int32_t *)row,
                     (const int32_t *)col, ka, nnz);
  int in_b = 0;
  SBX_TRY(sbx_radix_sort(h, 8, vb, ka, kb, val, vtmp, nnz, passes, np, &in_b));
  if (in_b && vb) SBX_HIP(h, hipMemcpyAsync(val, vtmp, (size_t)nnz * vb, hipMemcpyDeviceToDevice, h->stream));
  SBX_KLAUNCH(h, SBX_K_MISC, k_unpack_rc<int32_t>, dim3(grid), dim3(CV_THREADS),
                     (const uint64_t *)(in_b ? kb : ka), (int32_t *)row, (int32_t *)col, nnz);
  SBX_LAUNCH_CHECK(h);
  return SBX_OK;
}

namespace {

template <typename I, typename O, int VB, bool MOVE>
int launch_coo_to_csr(sbx_handle_t h, int64_t n, int64_t nnz, const I *row, const I *col, const char *val, O *rp,
                      I *col_out, char *val_out, GapEntry *gaps, unsigned *ngaps, unsigned gap_cap, int *unsorted) {
  const bool al = aligned16(row) && (MOVE || (aligned16(col) && aligned16(col_out) && aligned16(val) && aligned16(val_out)));
  const int64_t nquads = (nnz + 3) >> 2;
  const unsigned grid = sbx_grid_for(nquads, CV_THREADS, (int64_t)h->num_cus * 32);
  if (al)
    SBX_KLAUNCH(h, SBX_K_COO_TO_CSR, (k_coo_to_csr<I, O, VB, MOVE, true>), dim3(grid), dim3(CV_THREADS), row, col,
                       val, rp, col_out, val_out, n, nnz, gaps, ngaps, gap_cap, unsorted);
  else
    SBX_KLAUNCH(h, SBX_K_COO_TO_CSR, (k_coo_to_csr<I, O, VB, MOVE, false>), dim3(grid), dim3(CV_THREADS), row, col,
                       val, rp, col_out, val_out, n, nnz, gaps, ngaps, gap_cap, unsorted);
  SBX_LAUNCH_CHECK(h);
  return SBX_OK;
}

template <typename I, typename O, int VB, bool MOVE>
int launch_csr_to_coo(sbx_handle_t h, int64_t n, int64_t nnz, const O *rp, const I *col, const char *val, I *row_out,
                      I *col_out, char *val_out) {
  const bool al = aligned16(row_out) && (MOVE || (aligned16(col) && aligned16(col_out) && aligned16(val) && aligned16(val_out)));
  const unsigned grid = (unsigned)((nnz + EX_TILE - 1) / EX_TILE);
  int2 *span = nullptr;
  if ((int64_t)grid >= EX_SPAN_MIN_TILES) {
    SBX_TRY(sbx_salloc(h, (size_t)grid, &span));
    SBX_KLAUNCH(h, SBX_K_CSR_TO_COO, k_ex_tile_spans<O>, dim3((grid + CV_THREADS - 1) / CV_THREADS), dim3(CV_THREADS),
                rp, n, nnz, (int64_t)grid, span);
  }
  if (al)
    SBX_KLAUNCH(h, SBX_K_CSR_TO_COO, (k_csr_to_coo<I, O, VB, MOVE, true>), dim3(grid), dim3(CV_THREADS), rp, col,
                       val, row_out, col_out, val_out, n, nnz, (const int2 *)span);
  else
    SBX_KLAUNCH(h, SBX_K_CSR_TO_COO, (k_csr_to_coo<I, O, VB, MOVE, false>), dim3(grid), dim3(CV_THREADS), rp, col,
                       val, row_out, col_out, val_out, n, nnz, (const int2 *)span);
  SBX_LAUNCH_CHECK(h);
  return SBX_OK;
}

template <typename I, typename O>
__global__ __launch_bounds__(CV_THREADS) void k_row_hist(const I *__restrict__ row, O *__restrict__ cnt, int64_t nnz) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < nnz; i += stride) {
    if constexpr (sizeof(O) == 4) atomicAdd((int *)&cnt[row[i]], 1);
    else atomicAdd((unsigned long long *)&cnt[row[i]], 1ull);
  }
}

// a largest-value reduction over an index array (the 64-bit entry points check row ids against n this way)
template <typename I>
__global__ __launch_bounds__(CV_THREADS) void k_any_outside(const I *__restrict__ a, int64_t count, int64_t limit,
                                                            int *__restrict__ flag) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  bool bad = false;
  for (; i < count; i += stride) bad |= (int64_t)a[i] < 0 || (int64_t)a[i] >= limit;
  if (__any(bad) && sbx_lane() == 0) *flag = 1;
}

template <typename I>
int fill_index(sbx_handle_t h, I *dst, int64_t value, int64_t count) {
  if constexpr (sizeof(I) == 4) return sbx_fill_i32(h, (int32_t *)dst, (int32_t)value, count);
  else return sbx_fill_i64(h, (int64_t *)dst, value, count);
}
template <typename I>
int scan_index(sbx_handle_t h, I *a, int64_t count) {
  if constexpr (sizeof(I) == 4) return sbx_exclusive_scan_i32(h, (const int32_t *)a, (int32_t *)a, count, nullptr);
  else return sbx_exclusive_scan_i64(h, (const int64_t *)a, (int64_t *)a, count, nullptr);
}

// A2 for every index tuple (64-bit words run natively: values of any size, no narrowed copies)
template <typename I, typename O = I>
int coo_to_csr_typed(sbx_handle_t h, sbx_value_type vt, int64_t n, int64_t nnz, const void *row, const void *col,
                     const void *val, void *row_ptr_out, void *col_out, void *val_out, unsigned flags) {
  const bool move = (flags & SBX_FLAG_MOVE) != 0;
  const int vb = (val && val_out && !move) ? sbx_value_bytes(vt) : 0;
  if (vb < 0) SBX_FAIL(h, SBX_ERR_BAD_ARG, "sbx_coo_to_csr: unknown value type");
  SBX_TRY(sbx_arena_begin(h));
  O *rp = (O *)row_ptr_out;
  if (nnz == 0) return fill_index<O>(h, rp, 0, n + 1);
  const unsigned gap_cap = (unsigned)((n + 1) / GAP_INLINE + 2);
  GapEntry *gaps = nullptr;
  unsigned *ngaps = nullptr;
  int *unsorted = nullptr;
  SBX_TRY(sbx_salloc(h, gap_cap, &gaps));
  SBX_TRY(sbx_salloc(h, 3, &ngaps));
  unsorted = (int *)(ngaps + 1);
  int *outside = (int *)(ngaps + 2);
  SBX_HIP(h, hipMemsetAsync(ngaps, 0, 3 * sizeof(unsigned), h->stream));
  const I *r = (const I *)row, *c = (const I *)col;
  const char *v = (const char *)val;
  I *co = (I *)col_out;
  char *vo = (char *)val_out;
  if (sizeof(I) == 8) {  // row ids index row_ptr: a 64-bit id outside [0, n) must not get that far
    SBX_KLAUNCH(h, SBX_K_CHECK, k_any_outside<I>, dim3(sbx_grid_for(nnz, CV_THREADS, 8192)), dim3(CV_THREADS), r, nnz, n,
                outside);
    SBX_LAUNCH_CHECK(h);
    int f = 0;
    SBX_TRY(sbx_readback(h, &f, outside, sizeof(int)));
    if (f) SBX_FAIL(h, SBX_ERR_BAD_ARG, "sbx_coo_to_csr: a row id lies outside [0, n)");
  }
  int rc;
  if (move) rc = launch_coo_to_csr<I, O, 0, true>(h, n, nnz, r, c, v, rp, co, vo, gaps, ngaps, gap_cap, unsorted);
  else if (vb == 0) rc = launch_coo_to_csr<I, O, 0, false>(h, n, nnz, r, c, v, rp, co, vo, gaps, ngaps, gap_cap, unsorted);
  else if (vb == 4) rc = launch_coo_to_csr<I, O, 4, false>(h, n, nnz, r, c, v, rp, co, vo, gaps, ngaps, gap_cap, unsorted);
  else rc = launch_coo_to_csr<I, O, 8, false>(h, n, nnz, r, c, v, rp, co, vo, gaps, ngaps, gap_cap, unsorted);
  SBX_TRY(rc);
  SBX_KLAUNCH(h, SBX_K_COO_TO_CSR, k_fill_gaps<O>, dim3(256), dim3(CV_THREADS), rp, (const GapEntry *)gaps,
              (const unsigned *)ngaps, gap_cap);
  SBX_LAUNCH_CHECK(h);
  if (!(flags & SBX_FLAG_ROWS_SORTED)) {
    // unsorted row[] is only reachable with ignore_sort=true; the reference then
    // still produces exclusive_scan(histogram(row)) (converter_order_two.cc:180-192)
    int f = 0;
    SBX_TRY(sbx_readback(h, &f, unsorted, sizeof(int)));
    if (f) {
      SBX_TRY(fill_index<O>(h, rp, 0, n + 1));
      SBX_KLAUNCH(h, SBX_K_COO_TO_CSR, (k_row_hist<I, O>), dim3(sbx_grid_for(nnz, CV_THREADS, 8192)), dim3(CV_THREADS), r, rp,
                  nnz);
      SBX_LAUNCH_CHECK(h);
      SBX_TRY(scan_index<O>(h, rp, n + 1));
    }
  }
  return SBX_OK;
}

template <typename I, typename O = I>
int csr_to_coo_typed(sbx_handle_t h, sbx_value_type vt, int64_t n, int64_t nnz, const void *row_ptr, const void *col,
                     const void *val, void *row_out, void *col_out, void *val_out, unsigned flags) {
  const bool move = (flags & SBX_FLAG_MOVE) != 0;
  const int vb = (val && val_out && !move) ? sbx_value_bytes(vt) : 0;
  if (vb < 0) SBX_FAIL(h, SBX_ERR_BAD_ARG, "sbx_csr_to_coo: unknown value type");
  SBX_TRY(sbx_arena_begin(h));
  if (nnz == 0) return SBX_OK;
  const O *rp = (const O *)row_ptr;
  const I *c = (const I *)col;
  const char *v = (const char *)val;
  I *ro = (I *)row_out, *co = (I *)col_out;
  char *vo = (char *)val_out;
  if (move) return launch_csr_to_coo<I, O, 0, true>(h, n, nnz, rp, c, v, ro, co, vo);
  if (vb == 0) return launch_csr_to_coo<I, O, 0, false>(h, n, nnz, rp, c, v, ro, co, vo);
  if (vb == 4) return launch_csr_to_coo<I, O, 4, false>(h, n, nnz, rp, c, v, ro, co, vo);
  return launch_csr_to_coo<I, O, 8, false>(h, n, nnz, rp, c, v, ro, co, vo);
}

// ---------------------------------------------------------------- A14 COO -> CSC helpers
__global__ __launch_bounds__(CV_THREADS) void k_csc_keys(const int32_t *__restrict__ col, uint32_t *__restrict__ key,
                                                         uint32_t *__restrict__ idx, int64_t nnz) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < nnz; i += stride) {
    key[i] = (uint32_t)col[i];
    idx[i] = (uint32_t)i;
  }
}

template <int VB>
__global__ __launch_bounds__(CV_THREADS) void k_csc_gather(const uint32_t *__restrict__ idx,
                                                           const int32_t *__restrict__ row,
                                                           const char *__restrict__ val,
                                                           int32_t *__restrict__ row_out, char *__restrict__ val_out,
                                                           int64_t nnz) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < nnz; i += stride) {
    const uint32_t j = idx[i];
    row_out[i] = row[j];
    if (VB == 4) ((uint32_t *)val_out)[i] = ((const uint32_t *)val)[j];
    if (VB == 8) ((uint64_t *)val_out)[i] = ((const uint64_t *)val)[j];
  }
}

// (column key, payload) records for the stable sort: the payload carries the row (and a
// 4-byte value) itself, so nothing has to be gathered through an index afterwards
template <int VB>
__global__ __launch_bounds__(CV_THREADS) void k_csc_pack(const int32_t *__restrict__ row, const int32_t *__restrict__ col,
                                                         const char *__restrict__ val, uint32_t *__restrict__ key,
                                                         void *__restrict__ pay, int64_t nnz) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < nnz; i += stride) {
    key[i] = (uint32_t)col[i];
    if (VB == 4) ((uint64_t *)pay)[i] = (uint64_t)(uint32_t)row[i] | ((uint64_t)((const uint32_t *)val)[i] << 32);
    else ((uint32_t *)pay)[i] = (uint32_t)row[i];
  }
}
template <int VB>
__global__ __launch_bounds__(CV_THREADS) void k_csc_unpack(const void *__restrict__ pay, int32_t *__restrict__ row_out,
                                                           char *__restrict__ val_out, int64_t nnz) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < nnz; i += stride) {
    if (VB == 4) {
      const uint64_t p = ((const uint64_t *)pay)[i];
      row_out[i] = (int32_t)(uint32_t)p;
      ((uint32_t *)val_out)[i] = (uint32_t)(p >> 32);
    } else {
      row_out[i] = (int32_t)((const uint32_t *)pay)[i];
    }
  }
}

// col_ptr_out[m+1], row_out[nnz], val_out[nnz] from COO arrays (arena already begun, nesting on)
int coo_to_csc_core(sbx_handle_t h, sbx_value_type vt, int64_t n, int64_t m, int64_t nnz, const int32_t *row,
                    const int32_t *col, const char *val, int32_t *col_ptr_out, int32_t *row_out, char *val_out,
                    bool rows_ascend = false) {
  if (nnz == 0) return sbx_fill_i32(h, col_ptr_out, 0, m + 1);
  const int vb = (val && val_out) ? sbx_value_bytes(vt) : 0;
  const unsigned grid = sbx_grid_for(nnz, CV_THREADS, 8192);
  // the reference's placement loop (:53-62) is a stable counting sort on the column: a stable
  // LSD radix sort over the column bits moves the records to the same places
  sbx_radix_pass passes[16];
  const int np = sbx_radix_plan(0, sbx_bits_for(m > 0 ? (uint64_t)(m - 1) : 0), 0, 0, passes);
  uint32_t *ka = nullptr, *kb = nullptr;
  SBX_TRY(sbx_salloc(h, (size_t)nnz, &ka));
  SBX_TRY(sbx_salloc(h, (size_t)nnz, &kb));
  const uint32_t *skey = nullptr;
  int in_b = 0;
  if (vb != 8 && np >= 1 && nnz >= 2) {
    // the first digit pass reads (col; row, val) from the caller's arrays and the last one writes (row_out, val_out)
    // and the sorted column keys: payload = row | value << 32 in between, no pack / unpack kernels around the sort
    const int pb = vb == 4 ? 8 : 4;
    char *pa = nullptr, *pbuf = nullptr;
    if (np >= 2) SBX_TRY(sbx_salloc(h, (size_t)nnz * pb, &pa));
    if (np >= 3) SBX_TRY(sbx_salloc(h, (size_t)nnz * pb, &pbuf));
    uint32_t *kd = ((np - 1) & 1) ? kb : ka;  // the buffer the last pass does not read
    const sbx_radix_side src = {{(void *)col, nullptr}, {(void *)row, vb == 4 ? (void *)val : nullptr}, false, vb == 4};
    const sbx_radix_side dst = {{kd, nullptr}, {row_out, vb == 4 ? (void *)val_out : nullptr}, false, vb == 4};
    SBX_TRY(sbx_radix_sort_io(h, 4, pb, &src, ka, kb, pa, pbuf, &dst, nnz, passes, np));
    in_b = kd == kb;
  } else if (vb != 8) {
    const int pb = vb == 4 ? 8 : 4;
    char *pa = nullptr, *pbuf = nullptr;
    SBX_TRY(sbx_salloc(h, (size_t)nnz * pb, &pa));
    SBX_TRY(sbx_salloc(h, (size_t)nnz * pb, &pbuf));
    if (vb == 4) SBX_KLAUNCH(h, SBX_K_CSC, k_csc_pack<4>, dim3(grid), dim3(CV_THREADS), row, col, val, ka, (void *)pa, nnz);
    else SBX_KLAUNCH(h, SBX_K_CSC, k_csc_pack<0>, dim3(grid), dim3(CV_THREADS), row, col, val, ka, (void *)pa, nnz);
    SBX_LAUNCH_CHECK(h);
    SBX_TRY(sbx_radix_sort(h, 4, pb, ka, kb, pa, pbuf, nnz, passes, np, &in_b));
    const void *sp = in_b ? pbuf : pa;
    if (vb == 4) SBX_KLAUNCH(h, SBX_K_CSC, k_csc_unpack<4>, dim3(grid), dim3(CV_THREADS), sp, row_out, val_out, nnz);
    else SBX_KLAUNCH(h, SBX_K_CSC, k_csc_unpack<0>, dim3(grid), dim3(CV_THREADS), sp, row_out, val_out, nnz);
    SBX_LAUNCH_CHECK(h);
    SBX_PROF_BYTES(h, SBX_K_CSC, 2 * nnz * (int64_t)(8 + vb));
  } else {
    // 8-byte values: 12-byte payloads are not a radix record size, sort (column, source index) and gather
    uint32_t *ia = nullptr, *ib = nullptr;
    SBX_TRY(sbx_salloc(h, (size_t)nnz, &ia));
    SBX_TRY(sbx_salloc(h, (size_t)nnz, &ib));
    SBX_KLAUNCH(h, SBX_K_CSC, k_csc_keys, dim3(grid), dim3(CV_THREADS), col, ka, ia, nnz);
    SBX_LAUNCH_CHECK(h);
    SBX_TRY(sbx_radix_sort(h, 4, 4, ka, kb, ia, ib, nnz, passes, np, &in_b));
    SBX_KLAUNCH(h, SBX_K_CSC, k_csc_gather<8>, dim3(grid), dim3(CV_THREADS), (const uint32_t *)(in_b ? ib : ia), row,
                val, row_out, val_out, nnz);
    SBX_LAUNCH_CHECK(h);
    SBX_PROF_BYTES(h, SBX_K_CSC, nnz * (int64_t)(8 + 2 * (4 + vb)));
  }
  skey = in_b ? kb : ka;
  // col_ptr = exclusive scan of the column histogram = row-pointer construction over the sorted columns
  SBX_TRY(sbx_coo_to_csr(h, SBX_I32, SBX_V_NONE, m, n, nnz, skey, nullptr, nullptr, col_ptr_out, nullptr, nullptr,
                         SBX_FLAG_MOVE));
  // the CSC constructor (format/csc.cc:99-157) then sorts every column's (row, value) pairs if any is out of order —
  // which none is when the entries came in ascending row order (a CSR source): the stable sort kept that order inside
  // every column, and the check (a read of the rows and a host round trip) would find nothing
  if (rows_ascend) return SBX_OK;
  return sbx_csr_sort_rows(h, SBX_I32, vb ? vt : SBX_V_NONE, m, n, nnz, col_ptr_out, row_out, vb ? val_out : nullptr);
}

// ---- 64-bit index arrays, native: (column, source index) pairs through the radix sort — 8-byte keys over the column's
// significant bits, 4-byte payload (nnz < 2^31) — rows and values gathered through the sorted indices, col_ptr from the
// sorted columns.  No narrowed copies; one gather pass more than the 32-bit path, whose payload carries row and value.
__global__ __launch_bounds__(CV_THREADS) void k_csc_keys64(const int64_t *__restrict__ col, uint64_t *__restrict__ key,
                                                           uint32_t *__restrict__ idx, int64_t nnz) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < nnz; i += stride) {
    key[i] = (uint64_t)col[i];
    idx[i] = (uint32_t)i;
  }
}
template <int VB>
__global__ __launch_bounds__(CV_THREADS) void k_csc_gather64(const uint32_t *__restrict__ idx,
                                                             const int64_t *__restrict__ row,
                                                             const char *__restrict__ val,
                                                             int64_t *__restrict__ row_out, char *__restrict__ val_out,
                                                             int64_t nnz) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < nnz; i += stride) {
    const uint32_t j = idx[i];
    row_out[i] = row[j];
    if (VB == 4) ((uint32_t *)val_out)[i] = ((const uint32_t *)val)[j];
    if (VB == 8) ((uint64_t *)val_out)[i] = ((const uint64_t *)val)[j];
  }
}
int coo_to_csc_core64(sbx_handle_t h, sbx_value_type vt, int64_t n, int64_t m, int64_t nnz, const int64_t *row,
                      const int64_t *col, const char *val, int64_t *col_ptr_out, int64_t *row_out, char *val_out,
                      bool rows_ascend = false) {
  if (nnz == 0) return sbx_fill_i64(h, col_ptr_out, 0, m + 1);
  const int vb = (val && val_out) ? sbx_value_bytes(vt) : 0;
  const unsigned grid = sbx_grid_for(nnz, CV_THREADS, 8192);
  sbx_radix_pass passes[16];
  const int np = sbx_radix_plan(0, sbx_bits_for(m > 0 ? (uint64_t)(m - 1) : 0), 0, 0, passes);
  uint64_t *ka = nullptr, *kb = nullptr;
  uint32_t *ia = nullptr, *ib = nullptr;
  SBX_TRY(sbx_salloc(h, (size_t)nnz, &ka));
  SBX_TRY(sbx_salloc(h, (size_t)nnz, &kb));
  SBX_TRY(sbx_salloc(h, (size_t)nnz, &ia));
  SBX_TRY(sbx_salloc(h, (size_t)nnz, &ib));
  SBX_KLAUNCH(h, SBX_K_CSC, k_csc_keys64, dim3(grid), dim3(CV_THREADS), col, ka, ia, nnz);
  SBX_LAUNCH_CHECK(h);
  int in_b = 0;
  if (np > 0 && nnz >= 2) SBX_TRY(sbx_radix_sort(h, 8, 4, ka, kb, ia, ib, nnz, passes, np, &in_b));
  const uint32_t *idx = in_b ? ib : ia;
  if (vb == 0) SBX_KLAUNCH(h, SBX_K_CSC, k_csc_gather64<0>, dim3(grid), dim3(CV_THREADS), idx, row, val, row_out, val_out, nnz);
  else if (vb == 4) SBX_KLAUNCH(h, SBX_K_CSC, k_csc_gather64<4>, dim3(grid), dim3(CV_THREADS), idx, row, val, row_out, val_out, nnz);
  else SBX_KLAUNCH(h, SBX_K_CSC, k_csc_gather64<8>, dim3(grid), dim3(CV_THREADS), idx, row, val, row_out, val_out, nnz);
  SBX_LAUNCH_CHECK(h);
  SBX_PROF_BYTES(h, SBX_K_CSC, nnz * (int64_t)(16 + 2 * (8 + vb)));
  SBX_TRY(sbx_coo_to_csr(h, SBX_I64, SBX_V_NONE, m, n, nnz, in_b ? kb : ka, nullptr, nullptr, col_ptr_out, nullptr, nullptr,
                         SBX_FLAG_MOVE));
  if (rows_ascend) return SBX_OK;  // (see coo_to_csc_core)
  return sbx_csr_sort_rows(h, SBX_I64, vb ? vt : SBX_V_NONE, m, n, nnz, col_ptr_out, row_out, vb ? val_out : nullptr);
}

}  // namespace

extern "C" int sbx_coo_to_csr(sbx_handle_t h, sbx_index_type it, sbx_value_type vt, int64_t n, int64_t m, int64_t nnz,
                              const void *row, const void *col, const void *val, void *row_ptr_out, void *col_out,
                              void *val_out, unsigned flags) {
  if (!h) return SBX_ERR_BAD_ARG;
  (void)m;
  const bool move = (flags & SBX_FLAG_MOVE) != 0;
  SBX_REQUIRE(h, n >= 0 && nnz >= 0 && row_ptr_out && (nnz == 0 || row), "bad argument");
  SBX_REQUIRE(h, move || nnz == 0 || (col && col_out), "col/col_out required for a copy conversion");
  SBX_REQUIRE(h, n < ((int64_t)1 << 31) - 1, "row count exceeds int32");
  if (it == SBX_I64)  // native 64-bit kernels: column ids and nnz of any size
    return coo_to_csr_typed<int64_t>(h, vt, n, nnz, row, col, val, row_ptr_out, col_out, val_out, flags);
  if (it == SBX_I32_N64)  // 32-bit ids, 64-bit row_ptr: native, nnz of any size
    return coo_to_csr_typed<int32_t, int64_t>(h, vt, n, nnz, row, col, val, row_ptr_out, col_out, val_out, flags);
  SBX_REQUIRE(h, nnz < ((int64_t)1 << 31), "nnz exceeds int32");
  return coo_to_csr_typed<int32_t>(h, vt, n, nnz, row, col, val, row_ptr_out, col_out, val_out, flags);
}

extern "C" int sbx_csr_to_coo(sbx_handle_t h, sbx_index_type it, sbx_value_type vt, int64_t n, int64_t m, int64_t nnz,
                              const void *row_ptr, const void *col, const void *val, void *row_out, void *col_out,
                              void *val_out, unsigned flags) {
  if (!h) return SBX_ERR_BAD_ARG;
  (void)m;
  const bool move = (flags & SBX_FLAG_MOVE) != 0;
  SBX_REQUIRE(h, n >= 0 && nnz >= 0 && row_ptr && (nnz == 0 || row_out), "bad argument");
  SBX_REQUIRE(h, move || nnz == 0 || (col && col_out), "col/col_out required for a copy conversion");
  SBX_REQUIRE(h, n < ((int64_t)1 << 31) - 1, "row count exceeds int32");
  if (it == SBX_I64)
    return csr_to_coo_typed<int64_t>(h, vt, n, nnz, row_ptr, col, val, row_out, col_out, val_out, flags);
  if (it == SBX_I32_N64)
    return csr_to_coo_typed<int32_t, int64_t>(h, vt, n, nnz, row_ptr, col, val, row_out, col_out, val_out, flags);
  SBX_REQUIRE(h, nnz < ((int64_t)1 << 31), "nnz exceeds int32");
  return csr_to_coo_typed<int32_t>(h, vt, n, nnz, row_ptr, col, val, row_out, col_out, val_out, flags);
}

// A14: COO -> CSC.  The reference sizes col_ptr by the ROW count (converter_order_two.cc:32-33) and so
// only works for n == m; here col_ptr_out has m + 1 entries (identical for square matrices).
extern "C" int sbx_coo_to_csc(sbx_handle_t h, sbx_index_type it, sbx_value_type vt, int64_t n, int64_t m, int64_t nnz,
                              const void *row, const void *col, const void *val, void *col_ptr_out, void *row_out,
                              void *val_out) {
  if (!h) return SBX_ERR_BAD_ARG;
  if (it == SBX_I32_N64) return sbx_mixed_coo_to_csc(h, vt, n, m, nnz, row, col, val, col_ptr_out, row_out, val_out);
  SBX_REQUIRE(h, n >= 0 && m >= 0 && nnz >= 0 && col_ptr_out && (nnz == 0 || (row && col && row_out)), "bad argument");
  SBX_REQUIRE(h, nnz < ((int64_t)1 << 31) && n < ((int64_t)1 << 31) - 1 && m < ((int64_t)1 << 31) - 1,
              "dimension exceeds int32");
  SBX_REQUIRE(h, !(val && val_out) || sbx_value_bytes(vt) >= 0, "unknown value type");
  SBX_TRY(sbx_arena_begin(h));
  NestGuard guard(h);
  if (it == SBX_I64)  // native 64-bit kernels
    return coo_to_csc_core64(h, vt, n, m, nnz, (const int64_t *)row, (const int64_t *)col, (const char *)val,
                             (int64_t *)col_ptr_out, (int64_t *)row_out, (char *)val_out);
  return coo_to_csc_core(h, vt, n, m, nnz, (const int32_t *)row, (const int32_t *)col, (const char *)val,
                         (int32_t *)col_ptr_out, (int32_t *)row_out, (char *)val_out);
}

// A15: CSR -> CSC = CSR -> COO (row ids expanded into scratch) -> CSC
extern "C" int sbx_csr_to_csc(sbx_handle_t h, sbx_index_type it, sbx_value_type vt, int64_t n, int64_t m, int64_t nnz,
                              const void *row_ptr, const void *col, const void *val, void *col_ptr_out, void *row_out,
                              void *val_out) {
  if (!h) return SBX_ERR_BAD_ARG;
  if (it == SBX_I32_N64) return sbx_mixed_csr_to_csc(h, vt, n, m, nnz, row_ptr, col, val, col_ptr_out, row_out, val_out);
  SBX_REQUIRE(h, n >= 0 && m >= 0 && nnz >= 0 && row_ptr && col_ptr_out && (nnz == 0 || (col && row_out)),
              "bad argument");
  SBX_REQUIRE(h, nnz < ((int64_t)1 << 31) && n < ((int64_t)1 << 31) - 1 && m < ((int64_t)1 << 31) - 1,
              "dimension exceeds int32");
  SBX_REQUIRE(h, !(val && val_out) || sbx_value_bytes(vt) >= 0, "unknown value type");
  SBX_TRY(sbx_arena_begin(h));
  NestGuard guard(h);
  if (it == SBX_I64) {  // native 64-bit kernels
    int64_t *rows64 = nullptr;
    if (nnz > 0) {
      SBX_TRY(sbx_salloc(h, (size_t)nnz, &rows64));
      SBX_TRY(sbx_csr_to_coo(h, SBX_I64, SBX_V_NONE, n, m, nnz, row_ptr, nullptr, nullptr, rows64, nullptr, nullptr,
                             SBX_FLAG_MOVE));
    }
    return coo_to_csc_core64(h, vt, n, m, nnz, rows64, (const int64_t *)col, (const char *)val, (int64_t *)col_ptr_out,
                             (int64_t *)row_out, (char *)val_out, /*rows_ascend=*/true);
  }
  int32_t *rows = nullptr;
  if (nnz > 0) {
    SBX_TRY(sbx_salloc(h, (size_t)nnz, &rows));
    SBX_TRY(sbx_csr_to_coo(h, SBX_I32, SBX_V_NONE, n, m, nnz, row_ptr, nullptr, nullptr, rows, nullptr, nullptr,
                           SBX_FLAG_MOVE));
  }
  return coo_to_csc_core(h, vt, n, m, nnz, rows, (const int32_t *)col, (const char *)val, (int32_t *)col_ptr_out,
                         (int32_t *)row_out, (char *)val_out, /*rows_ascend=*/true);
}
