// sbx_permute.hip — permutation apply and the CSR-constructor row sort.
//
//   A5  permute/permute_order_two.cc:23-79   sbx_permute_csr, sbx_permute_csr_rows
//   A4  format/csr.cc:118-157                sbx_csr_sort_rows (and the sort A5 ends in)
//   A13 bases/reorder_base.h:663-672         sbx_inverse_permutation
//       permute/permute_order_one.cc:18-37   sbx_permute_array
//
// Data flow of one permute (new rows [row_begin,row_end) — the multi-GPU shard):
//   k_rowwise_prep  from the old-row side: rec[row_order[u]] = (length, source offset)   (12n B r, 8n B w)
//   lengths / k_rec_classify -> scan -> row_ptr_out; rows too long for a tile listed by class
//   no column map:  k_permute_copy, a segmented copy (+ in-row order check; unsorted input
//                   rows redo the call through the sorting kernels below)
//   column map:
//   k_permute_tile  one workgroup per 1024 output nonzeros: gathers whole old rows
//                   (col relabelled through col_order) into LDS and sorts every row
//                   of <= 1024 entries there: all-pairs ranking when the tile only
//                   holds rows <= 32, otherwise one tile-wide stable LDS radix sort
//                   on the composite key (local row, column); streams them out.
//   k_permute_block_rows  rows of 1K..16K entries: one workgroup per row in four
//                   capacity classes, LDS radix sort over the column bits.
//   long rows       longer rows: gathered into a compact buffer, sorted by
//                   (row, col) with the device radix sort, scattered back.
//   k_fix_dup_runs  only if some row was unsorted AND duplicate columns exist:
//                   orders equal-column runs by value (std::less<pair<col,val>>).
// HBM traffic per nonzero in the copy, tile and block paths is the compulsory 2*(I+V) bytes.
#include "sbx_device.h"
#include "sbx_internal.h"

namespace {

constexpr int PT_THREADS = 256;
constexpr int PT_TILE = 1024;        // rows up to this many entries are sorted in LDS
constexpr int PT_CAP = 2 * PT_TILE;  // LDS capacity of one tile, in entries
constexpr int PT_ITEMS = PT_CAP / PT_THREADS;
constexpr int PT_SHORT = 32;         // all-pairs rank sort up to this row length
constexpr int PT_MAXMED = PT_CAP / (PT_SHORT + 1) + 2;
// one workgroup sorts one row in LDS; four capacity classes so a 1100-entry row does not pay for 16384 slots
constexpr int BR_CLASSES = 4;
__host__ __device__ constexpr int br_cap(int cls) { return 2048 << cls; }  // 2048, 4096, 8192, 16384
template <int VB> struct BlockRowCap { static constexpr int value = 16384; };
template <> struct BlockRowCap<8> { static constexpr int value = 8192; };

struct PermState {            // device-resident flags/counters of one call
  unsigned any_unsorted;      // some row had col[j] < col[j-1] after relabelling (csr.cc:102-116)
  unsigned any_dup;           // some row holds a duplicate column
  unsigned n_long;            // rows longer than the block-row capacity (global radix path)
  unsigned n_block[BR_CLASSES];  // rows in (PT_TILE, capacity], by capacity class: one workgroup each
  unsigned long long long_nnz;
  unsigned long long total;   // nnz of the shard
  unsigned long_unsorted;     // some row of the global-radix class is out of order
  unsigned pad2;
};

template <int VB> struct ValT { typedef uint32_t type; };
template <> struct ValT<8> { typedef uint64_t type; };

template <typename I>
__global__ __launch_bounds__(256) void k_invert(const I *__restrict__ order, I *__restrict__ inv, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) inv[order[i]] = (I)i;
}

template <typename I, int VB>
__global__ __launch_bounds__(256) void k_permute_array(const I *__restrict__ order, const char *__restrict__ vals,
                                                       char *__restrict__ out, int64_t n) {
  typedef typename ValT<VB>::type V;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) ((V *)out)[order[i]] = ((const V *)vals)[i];
}

// row lengths (for the scan; entry nr zeroed) and the lists of rows too long for the tile kernel.
// The lists are appended to through five counter words; a few thousand long rows spread over millions would
// queue one atomic each on them (~88 atomics/us per word: 100 us for 10 K rows), so every workgroup stages the
// rows it finds in LDS and reserves list space once per class (the order inside a list does not matter).
constexpr int RC_STAGE = 512;  // staged rows per class and workgroup before an early flush
template <typename I>
__global__ __launch_bounds__(256) void k_rec_classify(const int2 *__restrict__ rec, I *__restrict__ rpo, int64_t nr,
                                                      I *__restrict__ long_rows, I *__restrict__ block_rows,
                                                      int64_t block_stride, int block_cap,
                                                      PermState *__restrict__ st) {
  constexpr int NC = BR_CLASSES + 1;  // class BR_CLASSES = rows for the global radix path
  __shared__ unsigned s_cnt[NC], s_base[NC];
  __shared__ unsigned long long s_long_nnz;
  __shared__ int s_full;
  __shared__ I s_rows[NC][RC_STAGE];
  const int tid = threadIdx.x;
  if (tid < NC) s_cnt[tid] = 0;
  if (tid == 0) s_long_nnz = 0;
  __syncthreads();
  auto flush = [&]() {  // all threads; leaves the stage empty
    if (tid < NC && s_cnt[tid]) {
      unsigned *counter = tid < BR_CLASSES ? &st->n_block[tid] : &st->n_long;
      s_base[tid] = atomicAdd(counter, s_cnt[tid]);
    }
    if (tid == 0 && s_long_nnz) {
      atomicAdd(&st->long_nnz, s_long_nnz);
      s_long_nnz = 0;
    }
    __syncthreads();
    for (int c = 0; c < NC; c++) {
      I *list = c < BR_CLASSES ? block_rows + (int64_t)c * block_stride : long_rows;
      for (unsigned k = tid; k < s_cnt[c]; k += 256) list[s_base[c] + k] = s_rows[c][k];
    }
    __syncthreads();
    if (tid < NC) s_cnt[tid] = 0;
    __syncthreads();
  };
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t base = (int64_t)blockIdx.x * 256; base <= nr; base += stride) {  // uniform trip count per workgroup
    const int64_t i = base + tid;
    if (i == nr && rpo) rpo[i] = 0;
    if (i < nr) {
      const I d = (I)rec[i].x;
      if (rpo) rpo[i] = d;
      int cls = -1;
      if (d > block_cap) cls = BR_CLASSES;
      else if (d > PT_TILE) cls = d <= br_cap(0) ? 0 : d <= br_cap(1) ? 1 : d <= br_cap(2) ? 2 : 3;
      if (cls >= 0) {
        s_rows[cls][atomicAdd(&s_cnt[cls], 1u)] = (I)i;
        if (cls == BR_CLASSES) atomicAdd(&s_long_nnz, (unsigned long long)d);
      }
    }
    __syncthreads();
    if (tid == 0) {
      bool full = false;
      for (int c = 0; c < NC; c++) full |= s_cnt[c] + 256u > (unsigned)RC_STAGE;
      s_full = full;
    }
    __syncthreads();
    if (s_full) flush();  // uniform decision
  }
  flush();
}

template <typename I>
__global__ void k_store_total(const I *__restrict__ rpo, int64_t nr, PermState *__restrict__ st) {
  st->total = (unsigned long long)rpo[nr];
}

// ---- row-wise permute: no column relabel, so rows keep their internal order and the
// permute is a segmented copy.  One workgroup per PC_TILE output nonzeros: row heads of
// the tile are scattered into LDS with their (source - destination) offset, a max-scan
// gives every position its offset, then a strided pass copies col/val (reads coalesced
// per row piece, writes fully coalesced) and checks the order inside rows.  If any row
// turns out unsorted (possible only for inputs that never went through a CSR
// constructor) the caller discards the result and runs the sorting pipeline.
// (row length, source offset) in new-row order for rows [rb0, rb0+nr), written from the
// old-row side (sequential reads, one 8-byte scatter per row) so no inverse permutation is needed
template <typename I>
__global__ __launch_bounds__(256) void k_rowwise_prep(const I *__restrict__ rp, const I *__restrict__ row_order,
                                                      int64_t n, int64_t rb0, int64_t nr, int2 *__restrict__ rec) {
  int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; u < n; u += stride) {
    const int64_t r = (row_order ? (int64_t)row_order[u] : u) - rb0;
    if (r < 0 || r >= nr) continue;
    const I s = rp[u];
    rec[r] = make_int2((int)(rp[u + 1] - s), (int)s);
  }
}

template <typename I>
__global__ __launch_bounds__(256) void k_rowwise_lengths(const int2 *__restrict__ rec, int64_t nr, I *__restrict__ deg_out) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; r <= nr; r += stride) deg_out[r] = r < nr ? (I)rec[r].x : (I)0;
}

// tile_row[t] = last row r with rpo[r] <= min(t * tile, total)
template <typename I>
__global__ __launch_bounds__(256) void k_tile_rows(const I *__restrict__ rpo, int64_t nr, int64_t total, int tile,
                                                   int64_t ntiles, I *__restrict__ tile_row) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t > ntiles) return;
  const int64_t pos = t * tile < total ? t * tile : total;
  int64_t lo = 0, hi = nr;
  while (lo < hi) {
    const int64_t mid = (lo + hi + 1) >> 1;
    if ((int64_t)rpo[mid] <= pos) lo = mid; else hi = mid - 1;
  }
  tile_row[t] = (I)lo;
}

constexpr int PC_THREADS = 256;
constexpr int PC_ITEMS = 8;
constexpr int PC_TILE = PC_THREADS * PC_ITEMS;

template <typename I, int VB>
__global__ __launch_bounds__(PC_THREADS) void k_permute_copy(const int2 *__restrict__ rec, const I *__restrict__ tile_row,
                                                             const I *__restrict__ col_in,
                                                             const char *__restrict__ val_in,
                                                             const I *__restrict__ rpo, I *__restrict__ col_out,
                                                             char *__restrict__ val_out, int64_t nr, int64_t total,
                                                             PermState *__restrict__ st) {
  typedef typename ValT<VB>::type V;
  __shared__ unsigned long long s_head[PC_TILE];  // (row - r_lo) << 32 | (source - destination) at the row's first nonzero
  __shared__ int s_delta[PC_TILE];
  __shared__ int s_col[PC_TILE];
  __shared__ int s_wmax[PC_THREADS / 64];
  const int tid = threadIdx.x;
  const int64_t t0 = (int64_t)blockIdx.x * PC_TILE;
  const int64_t t1 = t0 + PC_TILE < total ? t0 + PC_TILE : total;
  const int cnt = (int)(t1 - t0);
#pragma unroll
  for (int k = 0; k < PC_ITEMS; k++) s_head[k * PC_THREADS + tid] = 0;
  // rows under the tile: r_lo owns position 0; rows up to r_hi may start inside (r_hi itself at t1: skipped)
  const int64_t r_lo = tile_row[blockIdx.x], r_hi = tile_row[blockIdx.x + 1];
  __syncthreads();
  const int64_t lo_start = rpo[r_lo];
  const int delta_lo = (int)((int64_t)rec[r_lo].y - lo_start);
  if (r_hi - r_lo <= 4 * PC_TILE) {
    for (int64_t r = r_lo + 1 + tid; r <= r_hi; r += PC_THREADS) {
      const int64_t s = rpo[r];
      if (s >= t1) continue;
      const int delta = (int)((int64_t)rec[r].y - s);
      // empty rows share the position of the next non-empty one, which has the largest id
      atomicMax(&s_head[s - t0], ((unsigned long long)(r - r_lo) << 32) | (unsigned)delta);
    }
  } else {
    for (int k = 0; k < PC_ITEMS; k++) {
      const int p = tid * PC_ITEMS + k;
      if (p >= cnt || p == 0) continue;
      int64_t lo = r_lo, hi = r_hi;  // last row with rpo[r] <= t0 + p
      while (lo < hi) {
        const int64_t mid = (lo + hi + 1) >> 1;
        if ((int64_t)rpo[mid] <= t0 + p) lo = mid; else hi = mid - 1;
      }
      if ((int64_t)rpo[lo] == t0 + p && lo > r_lo)
        s_head[p] = ((unsigned long long)(lo - r_lo) << 32) | (unsigned)(int)((int64_t)rec[lo].y - (t0 + p));
    }
  }
  __syncthreads();
  {  // every position gets the offset of the row it belongs to
    const int p0 = tid * PC_ITEMS;
    unsigned long long hd[PC_ITEMS];
    int last = 0;
#pragma unroll
    for (int k = 0; k < PC_ITEMS; k++) {
      hd[k] = s_head[p0 + k];
      if (hd[k]) last = p0 + k + 1;
    }
    const int inc = sbx_wave_inclusive_max(last);
    int open = __shfl_up(inc, 1, 64);
    if (sbx_lane() == 0) open = 0;
    if (sbx_lane() == 63) s_wmax[tid >> 6] = inc;
    __syncthreads();
    for (int w = 0; w < (tid >> 6); w++) open = s_wmax[w] > open ? s_wmax[w] : open;
    int delta = open ? (int)(unsigned)s_head[open - 1] : delta_lo;
#pragma unroll
    for (int k = 0; k < PC_ITEMS; k++) {
      if (hd[k]) delta = (int)(unsigned)hd[k];
      s_delta[p0 + k] = delta;
    }
  }
  __syncthreads();
  int c[PC_ITEMS];
#pragma unroll
  for (int k = 0; k < PC_ITEMS; k++) {
    const int p = k * PC_THREADS + tid;
    if (p < cnt) {
      const int64_t src = t0 + p + s_delta[p];
      c[k] = col_in[src];
      col_out[t0 + p] = c[k];
      if (VB) ((V *)val_out)[t0 + p] = ((const V *)val_in)[src];
      s_col[p] = c[k];
    }
  }
  __syncthreads();
  bool bad = false;
#pragma unroll
  for (int k = 0; k < PC_ITEMS; k++) {
    const int p = k * PC_THREADS + tid;
    if (p < cnt && s_head[p] == 0) {
      if (p > 0) bad |= s_col[p - 1] > c[k];
      else if (lo_start < t0) bad |= col_in[t0 + delta_lo - 1] > c[k];  // the row began in the previous tile
    }
  }
  if (__any(bad) && sbx_lane() == 0) st->any_unsorted = 1;
}

// ---- the tile kernel ----------------------------------------------------------
// IDENT: csr_sort_rows mode (no row/col maps, input == output arrays allowed).
template <typename I, int VB>
__global__ __launch_bounds__(PT_THREADS) void k_permute_tile(
    const int2 *__restrict__ rec, const I *col_in, const char *val_in, const I *__restrict__ col_order,
    const I *__restrict__ rpo, I *col_out, char *val_out, int64_t nr, PermState *__restrict__ st, int col_bits) {
  typedef typename ValT<VB>::type V;
  constexpr bool HASV = VB != 0;
  __shared__ int s_col[PT_CAP];
  __shared__ V s_val[HASV ? PT_CAP : 1];
  __shared__ int s_row[PT_CAP];  // row heads -> source offset of every entry -> row length -> dense row rank (radix key)
  __shared__ unsigned s_whist[PT_THREADS / 64][256];
  // first position of the entry's row in the tile; dead before the radix passes start, so it lives in the histogram space
  unsigned short *s_hp = (unsigned short *)&s_whist[0][0];
  static_assert(sizeof(unsigned short) * PT_CAP <= sizeof(unsigned) * (PT_THREADS / 64) * 256, "s_hp must fit s_whist");
  __shared__ unsigned s_scan[PT_THREADS / 64 + 1];
  __shared__ int s_nmed;
  __shared__ int s_tile_unsorted;
  __shared__ int64_t s_span[2];
  __shared__ int s_wmax[PT_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t lo_t = (int64_t)blockIdx.x * PT_TILE, hi_t = lo_t + PT_TILE;

  if (tid < 64) {
    const int64_t a = sbx_wave_upper_bound<I>(rpo, nr + 1, (I)(lo_t - 1));  // first row starting at >= lo_t
    if (tid == 0) s_span[0] = a;
  } else if (tid < 128) {
    const int64_t total = (int64_t)rpo[nr];
    int64_t b = nr;
    if (hi_t - 1 < total) b = sbx_wave_upper_bound<I>(rpo, nr + 1, (I)(hi_t - 1));
    if (tid == 64) s_span[1] = b > nr ? nr : b;
  }
  if (tid == 0) {
    s_nmed = 0;
    s_tile_unsorted = 0;
  }
#pragma unroll
  for (int k = 0; k < PT_ITEMS; k++) s_row[k * PT_THREADS + tid] = 0;
  __syncthreads();
  const int64_t ra = s_span[0];
  int64_t rb = s_span[1];
  if (ra >= rb) return;
  if ((int64_t)rpo[rb] - (int64_t)rpo[rb - 1] > PT_TILE) rb--;  // a long row can only be the last one
  if (ra >= rb) return;
  const int64_t e0 = rpo[ra];
  const int cnt = (int)((int64_t)rpo[rb] - e0);
  if (cnt == 0) return;

  // Row heads into LDS (tiles whose row range is dominated by empty rows — e.g. the
  // isolated vertices RCM packs at the end — would walk millions of heads: they test
  // each position by binary search instead), then a max-scan of head positions gives every
  // entry its row; the row's (length, source offset) record is fetched once per row piece.
  const int64_t nrows_range = rb - ra;
  if (nrows_range <= 4 * PT_CAP) {
    for (int64_t r = ra + 1 + tid; r < rb; r += PT_THREADS) {
      const int p = (int)((int64_t)rpo[r] - e0);
      if (p < cnt) atomicMax(&s_row[p], (int)(r - ra));
    }
  } else {
    for (int p = tid; p < cnt; p += PT_THREADS) {
      const int64_t target = e0 + p;  // last row r in [ra, rb) with rpo[r] <= target
      int64_t lo = ra, hi = rb;       // invariant: rpo[lo] <= target, answer in [lo, hi)
      while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)rpo[mid] <= target) lo = mid;
        else hi = mid;
      }
      if ((int64_t)rpo[lo] == target) s_row[p] = (int)(lo - ra);
    }
  }
  __syncthreads();
  {
    const int p0 = tid * PT_ITEMS;
    int hd[PT_ITEMS];
    int last = 0;
#pragma unroll
    for (int k = 0; k < PT_ITEMS; k++) {
      hd[k] = s_row[p0 + k];
      if (hd[k]) last = p0 + k + 1;
    }
    const int inc = sbx_wave_inclusive_max(last);
    int open = __shfl_up(inc, 1, 64);
    if (lane == 0) open = 0;
    if (lane == 63) s_wmax[wv] = inc;
    __syncthreads();
    for (int w = 0; w < wv; w++) open = s_wmax[w] > open ? s_wmax[w] : open;
    int hp = open ? open - 1 : 0;
    int2 rc = rec[ra + (open ? s_row[open - 1] : 0)];
    __syncthreads();  // every carry-in has been read: s_row can be overwritten
#pragma unroll
    for (int k = 0; k < PT_ITEMS; k++) {
      const int p = p0 + k;
      if (hd[k]) {
        hp = p;
        rc = rec[ra + hd[k]];
      }
      if (p < cnt) {
        s_row[p] = (int)((int64_t)rc.y - (e0 + hp));  // source index = e0 + p + this
        s_hp[p] = (unsigned short)hp;
      }
      hd[k] = rc.x;  // row length, stored once the offsets have been consumed
    }
    __syncthreads();
    // gather: whole old rows, columns relabelled (permute_order_two.cc:63-74)
    for (int p = tid; p < cnt; p += PT_THREADS) {
      const int64_t src = e0 + p + s_row[p];
      I c = col_in[src];
      if (col_order) c = col_order[c];
      s_col[p] = (int)c;
      if (HASV) s_val[p] = ((const V *)val_in)[src];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PT_ITEMS; k++)
      if (p0 + k < cnt) s_row[p0 + k] = hd[k];
  }
  __syncthreads();

  // classify the tile: rows <= PT_SHORT only -> all-pairs ranking; otherwise one
  // tile-wide stable LSD radix sort on the composite key (local row, column)
  bool unsorted = false, dup = false, has_medium = false;
  for (int p = tid; p < cnt; p += PT_THREADS) {
    const int s = (int)s_hp[p];
    if (p > s && s_col[p] < s_col[p - 1]) unsorted = true;
    if (p == s && s_row[p] > PT_SHORT) has_medium = true;
  }
  if (__any(unsorted) && lane == 0) {
    st->any_unsorted = 1;
    s_tile_unsorted = 1;
  }
  if (__any(has_medium) && lane == 0) s_nmed = 1;
  __syncthreads();
  if (s_tile_unsorted == 0) {
    // every row of the tile is already in column order (row-wise permutes, identity
    // column maps, orders that preserve locality): a stable sort would not move
    // anything, so stream the gathered rows out as they are
    for (int p = tid; p < cnt; p += PT_THREADS) {
      const int c = s_col[p];
      if (p && c == s_col[p - 1] && s_hp[p] == s_hp[p - 1]) dup = true;
      col_out[e0 + p] = (I)c;
      if (HASV) ((V *)val_out)[e0 + p] = s_val[p];
    }
    if (__any(dup) && lane == 0) st->any_dup = 1;
    return;
  }
  if (s_nmed == 0) {
    for (int p = tid; p < cnt; p += PT_THREADS) {
      const int s = (int)s_hp[p], len = s_row[p];
      const int c = s_col[p];
      int rank = 0;
      for (int j = s; j < s + len; j++) {
        const int cj = s_col[j];
        rank += (cj < c) || (cj == c && j < p);
        dup |= (cj == c) && (j != p);
      }
      const int64_t o = e0 + s + rank;
      col_out[o] = (I)c;
      if (HASV) ((V *)val_out)[o] = s_val[p];
    }
    if (__any(dup) && lane == 0) st->any_dup = 1;
    return;
  }

  // ---- tile-wide LDS radix sort: passes over the column bits, then the row bits.
  // The row part of the key is the DENSE rank of the element's row among the rows
  // present in the tile (prefix count of row heads), so it never exceeds 11 bits
  // however many empty rows the range contains.
  {
    int flag[PT_ITEMS];
    int local = 0;
#pragma unroll
    for (int k = 0; k < PT_ITEMS; k++) {
      const int p = tid * PT_ITEMS + k;
      const bool head = p < cnt && (p == 0 || s_hp[p] != s_hp[p - 1]);
      local += head;
      flag[k] = local;
    }
    int all;
    const int ex = sbx_block_exclusive_sum<int, PT_THREADS>(local, (int *)s_scan, &all);
#pragma unroll
    for (int k = 0; k < PT_ITEMS; k++) {
      const int p = tid * PT_ITEMS + k;
      if (p < cnt) s_row[p] = ex + flag[k] - 1;
    }
    if (tid == 0) s_nmed = all;  // number of rows present
  }
  __syncthreads();
  int row_bits = 0;
  for (int t = s_nmed - 1; t > 0; t >>= 1) row_bits++;
  const int total_bits[2] = {col_bits, row_bits};
  volatile unsigned *wh = s_whist[wv];
  const uint64_t lt = sbx_lanemask_lt();
  for (int part = 0; part < 2; part++) {
    int done = 0;
    while (done < total_bits[part]) {
      const int remaining = total_bits[part] - done;
      const int passes_left = (remaining + 7) >> 3;
      const int bits = (remaining + passes_left - 1) / passes_left;
      const unsigned mask = (1u << bits) - 1u;
      const int shift = done;
      if (tid < 256) {
#pragma unroll
        for (int i = 0; i < PT_THREADS / 64; i++) s_whist[i][tid] = 0;
      }
      int kc[PT_ITEMS], kr[PT_ITEMS];
      V kv[HASV ? PT_ITEMS : 1];
      unsigned rank[PT_ITEMS];
#pragma unroll
      for (int i = 0; i < PT_ITEMS; i++) {
        const int e = wv * 64 * PT_ITEMS + i * 64 + lane;
        const bool ok = e < cnt;
        kc[i] = ok ? s_col[e] : 0x7FFFFFFF;
        kr[i] = ok ? s_row[e] : 0x7FFFFFFF;
        if (HASV) kv[i] = ok ? s_val[e] : (V)0;
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < PT_ITEMS; i++) {
        const unsigned d = ((unsigned)(part ? kr[i] : kc[i]) >> shift) & mask;
        uint64_t m = ~(uint64_t)0;
        for (int bb = 0; bb < bits; bb++) {
          const bool bit = (d >> bb) & 1u;
          const uint64_t bal = __ballot(bit);
          m &= bit ? bal : ~bal;
        }
        const unsigned prev = wh[d];
        const unsigned rk = (unsigned)__popcll(m & lt);
        __builtin_amdgcn_wave_barrier();
        if (rk == 0) wh[d] = prev + (unsigned)__popcll(m);
        __builtin_amdgcn_wave_barrier();
        rank[i] = prev + rk;
      }
      __syncthreads();
      {
        unsigned c4[PT_THREADS / 64];
        unsigned tot = 0;
        if (tid < 256) {  // thread `tid` owns digit `tid`
#pragma unroll
          for (int i = 0; i < PT_THREADS / 64; i++) {
            c4[i] = s_whist[i][tid];
            tot += c4[i];
          }
        }
        unsigned all;
        unsigned ex = sbx_block_exclusive_sum<unsigned, PT_THREADS>(tot, s_scan, &all);
        if (tid < 256) {
#pragma unroll
          for (int i = 0; i < PT_THREADS / 64; i++) {
            s_whist[i][tid] = ex;
            ex += c4[i];
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < PT_ITEMS; i++) {
        const unsigned d = ((unsigned)(part ? kr[i] : kc[i]) >> shift) & mask;
        const unsigned pos = s_whist[wv][d] + rank[i];
        if (pos < (unsigned)PT_CAP) {
          s_col[pos] = kc[i];
          s_row[pos] = kr[i];
          if (HASV) s_val[pos] = kv[i];
        }
      }
      __syncthreads();
      done += bits;
    }
  }
  for (int p = tid; p < cnt; p += PT_THREADS) {
    const int c = s_col[p];
    if (p && c == s_col[p - 1] && s_row[p] == s_row[p - 1]) dup = true;
    col_out[e0 + p] = (I)c;
    if (HASV) ((V *)val_out)[e0 + p] = s_val[p];
  }
  if (__any(dup) && lane == 0) st->any_dup = 1;
}

// ---- rows of (PT_TILE, capacity] entries: one workgroup per row, LDS radix sort ----
// The row is gathered (relabelled) into LDS once, sorted there by a stable LSD radix
// sort over the significant column bits (ballot match-any ranking, per-wave digit
// counters; every thread keeps its items in registers across the in-place scatter)
// and streamed out: HBM sees each nonzero exactly once in and once out.
struct RowPasses {
  int n;
  int shift[4];
  int bits[4];
};

template <typename I, int VB, int CAP, int BR_THREADS>
__global__ __launch_bounds__(BR_THREADS) void k_permute_block_rows(
    const int2 *__restrict__ rec, const I *col_in, const char *val_in, const I *__restrict__ col_order,
    const I *__restrict__ rpo, const I *__restrict__ block_rows, I *col_out, char *val_out, RowPasses passes,
    PermState *__restrict__ st) {
  typedef typename ValT<VB>::type V;
  constexpr bool HASV = VB != 0;
  constexpr int ITEMS = CAP / BR_THREADS;
  constexpr int WAVES = BR_THREADS / 64;
  __shared__ int s_key[CAP];
  __shared__ V s_val[HASV ? CAP : 1];
  __shared__ unsigned s_whist[WAVES][256];
  __shared__ unsigned s_scan[WAVES + 1];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t r = block_rows[blockIdx.x];
  const int64_t e0 = rpo[r];
  const int len = (int)((int64_t)rpo[r + 1] - e0);
  const int64_t src0 = rec[r].y;
  for (int i = tid; i < CAP; i += BR_THREADS) {
    int c = 0x7FFFFFFF;  // padding: sorts last, never written out
    if (i < len) {
      I cc = col_in[src0 + i];
      if (col_order) cc = col_order[cc];
      c = (int)cc;
      if (HASV) s_val[i] = ((const V *)val_in)[src0 + i];
    }
    s_key[i] = c;
  }
  __syncthreads();
  bool unsorted = false;
  for (int i = tid + 1; i < len; i += BR_THREADS) unsorted |= s_key[i] < s_key[i - 1];
  if (tid == 0) s_scan[BR_THREADS / 64] = 0;
  __syncthreads();
  if (__any(unsorted) && lane == 0) {
    st->any_unsorted = 1;
    s_scan[BR_THREADS / 64] = 1;
  }
  __syncthreads();
  const int npass = s_scan[BR_THREADS / 64] ? passes.n : 0;  // an ordered row needs no sort
  __syncthreads();

  volatile unsigned *wh = s_whist[w];
  const uint64_t lt = sbx_lanemask_lt();
  for (int pass = 0; pass < npass; pass++) {
    const int shift = passes.shift[pass], bits = passes.bits[pass];
    const unsigned mask = (1u << bits) - 1u;
    for (int i = tid; i < 256 * WAVES; i += BR_THREADS) (&s_whist[0][0])[i] = 0;
    int k[ITEMS];
    V v[HASV ? ITEMS : 1];
    unsigned rank[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
      const int e = w * 64 * ITEMS + i * 64 + lane;
      k[i] = s_key[e];
      if (HASV) v[i] = s_val[e];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
      const unsigned d = ((unsigned)k[i] >> shift) & mask;
      uint64_t m = ~(uint64_t)0;
      for (int b = 0; b < bits; b++) {
        const bool bit = (d >> b) & 1u;
        const uint64_t bal = __ballot(bit);
        m &= bit ? bal : ~bal;
      }
      const unsigned prev = wh[d];
      const unsigned rk = (unsigned)__popcll(m & lt);
      __builtin_amdgcn_wave_barrier();
      if (rk == 0) wh[d] = prev + (unsigned)__popcll(m);
      __builtin_amdgcn_wave_barrier();
      rank[i] = prev + rk;
    }
    __syncthreads();
    {
      unsigned c[WAVES];
      unsigned tot = 0;
      if (tid < 256) {
#pragma unroll
        for (int i = 0; i < WAVES; i++) {
          c[i] = s_whist[i][tid];
          tot += c[i];
        }
      }
      unsigned all;
      unsigned ex = sbx_block_exclusive_sum<unsigned, BR_THREADS>(tot, s_scan, &all);
      if (tid < 256) {
#pragma unroll
        for (int i = 0; i < WAVES; i++) {
          s_whist[i][tid] = ex;
          ex += c[i];
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
      const unsigned d = ((unsigned)k[i] >> shift) & mask;
      const unsigned pos = s_whist[w][d] + rank[i];
      s_key[pos] = k[i];
      if (HASV) s_val[pos] = v[i];
    }
    __syncthreads();
  }
  bool dup = false;
  for (int i = tid; i < len; i += BR_THREADS) {
    const int c = s_key[i];
    if (i && c == s_key[i - 1]) dup = true;
    col_out[e0 + i] = (I)c;
    if (HASV) ((V *)val_out)[e0 + i] = s_val[i];
  }
  if (__any(dup) && lane == 0) st->any_dup = 1;
}

// ---- long rows ----------------------------------------------------------------
// flat over the nonzeros of all long rows (a workgroup per row would leave the longest row as a straggler)
template <typename I, int VB>
__global__ __launch_bounds__(256) void k_long_gather(const int2 *__restrict__ rec, const I *col_in, const char *val_in,
                                                     const I *__restrict__ col_order,
                                                     const I *__restrict__ long_rows,
                                                     const uint32_t *__restrict__ loff, int n_long, int64_t long_nnz,
                                                     uint64_t *__restrict__ keys, char *__restrict__ pay,
                                                     PermState *__restrict__ st) {
  typedef typename ValT<VB>::type V;
  bool unsorted = false;
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; e < long_nnz; e += stride) {
    int lo = 0, hi = n_long - 1;  // last k with loff[k] <= e
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if ((int64_t)loff[mid] <= e) lo = mid; else hi = mid - 1;
    }
    const int k = lo;
    const int64_t j = e - (int64_t)loff[k];
    const int64_t src0 = rec[long_rows[k]].y;
    I c = col_in[src0 + j];
    if (col_order) c = col_order[c];
    if (j) {
      I pc = col_in[src0 + j - 1];
      if (col_order) pc = col_order[pc];
      unsorted |= c < pc;
    }
    keys[e] = ((uint64_t)(uint32_t)k << 32) | (uint64_t)(uint32_t)c;
    if (VB) ((V *)pay)[e] = ((const V *)val_in)[src0 + j];
  }
  if (__any(unsorted) && sbx_lane() == 0) {
    st->any_unsorted = 1;
    st->long_unsorted = 1;
  }
}

template <typename I, int VB>
__global__ __launch_bounds__(256) void k_long_scatter(const uint64_t *__restrict__ keys, const char *__restrict__ pay,
                                                      const I *__restrict__ rpo, const I *__restrict__ long_rows,
                                                      const uint32_t *__restrict__ loff, int64_t long_nnz,
                                                      I *__restrict__ col_out, char *__restrict__ val_out,
                                                      PermState *__restrict__ st) {
  typedef typename ValT<VB>::type V;
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  bool dup = false;
  for (; p < long_nnz; p += stride) {
    const uint64_t key = keys[p];
    const uint32_t k = (uint32_t)(key >> 32);
    const int64_t o = (int64_t)rpo[long_rows[k]] + (p - (int64_t)loff[k]);
    col_out[o] = (I)(uint32_t)key;
    if (VB) ((V *)val_out)[o] = ((const V *)pay)[p];
    if (p && keys[p - 1] == key) dup = true;
  }
  if (__any(dup) && sbx_lane() == 0) st->any_dup = 1;
}

template <typename I>
__global__ __launch_bounds__(256) void k_long_lengths(const I *__restrict__ rpo, const I *__restrict__ long_rows,
                                                      uint32_t *__restrict__ len, int n_long) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n_long) len[k] = (uint32_t)(rpo[long_rows[k] + 1] - rpo[long_rows[k]]);
}

// ---- duplicate-column fix-up (format/csr.cc:143-156 pair ordering) -------------
template <sbx_value_type VT> struct Typed;
template <> struct Typed<SBX_V_I32> { typedef int32_t T; };
template <> struct Typed<SBX_V_U32> { typedef uint32_t T; };
template <> struct Typed<SBX_V_F32> { typedef float T; };
template <> struct Typed<SBX_V_I64> { typedef int64_t T; };
template <> struct Typed<SBX_V_U64> { typedef uint64_t T; };
template <> struct Typed<SBX_V_F64> { typedef double T; };

template <typename I, typename T>
__global__ __launch_bounds__(256) void k_fix_dup_runs(const I *__restrict__ rpo, const I *__restrict__ col,
                                                      T *__restrict__ val, int64_t nr,
                                                      const PermState *__restrict__ st) {
  if (!(st->any_unsorted && st->any_dup)) return;
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; r < nr; r += stride) {
    const int64_t s = rpo[r], e = rpo[r + 1];
    int64_t a = s;
    while (a < e) {
      int64_t b = a + 1;
      while (b < e && col[b] == col[a]) b++;
      for (int64_t i = a + 1; i < b; i++) {  // insertion sort of the run by value
        const T x = val[i];
        int64_t j = i;
        while (j > a && x < val[j - 1]) {
          val[j] = val[j - 1];
          j--;
        }
        val[j] = x;
      }
      a = b;
    }
  }
}

template <typename I>
int launch_fix(sbx_handle_t h, sbx_value_type vt, const I *rpo, const I *col, void *val, int64_t nr, PermState *st) {
  const unsigned grid = sbx_grid_for(nr, 256, 4096);
#define FIX(VT)                                                                                              \
  case VT:                                                                                                   \
    SBX_KLAUNCH(h, SBX_K_PERMUTE_LONG, (k_fix_dup_runs<I, typename Typed<VT>::T>), dim3(grid), dim3(256), rpo, col, \
                       (typename Typed<VT>::T *)val, nr, (const PermState *)st);                            \
    break;
  switch (vt) {
    FIX(SBX_V_I32) FIX(SBX_V_U32) FIX(SBX_V_F32) FIX(SBX_V_I64) FIX(SBX_V_U64) FIX(SBX_V_F64)
    default: return SBX_OK;
  }
#undef FIX
  SBX_LAUNCH_CHECK(h);
  return SBX_OK;
}

static bool permute_overlap() {
  static const bool on = !(getenv("SBX_PERMUTE_OVERLAP") && atoi(getenv("SBX_PERMUTE_OVERLAP")) == 0);
  return on;
}

// rows of PT_TILE < length <= 16 K: one workgroup per row, by capacity class
template <int VB>
int block_rows_path(sbx_handle_t h, const int2 *rec, const int32_t *col_in, const char *val_in, const int32_t *col_order,
                    const int32_t *rpo, int32_t *col_out, char *val_out, int64_t m, const int32_t *block_rows,
                    const unsigned *n_block, int64_t block_stride, PermState *st) {
  typedef int32_t I;
  RowPasses rpasses;
  sbx_radix_pass pl[16];
  rpasses.n = sbx_radix_plan(0, sbx_bits_for(m > 0 ? (uint64_t)(m - 1) : 0), 0, 0, pl);
  for (int i = 0; i < rpasses.n && i < 4; i++) {
    rpasses.shift[i] = pl[i].shift;
    rpasses.bits[i] = pl[i].bits;
  }
#define BLOCK_ROWS(CLS, THREADS)                                                                                  \
  if (n_block[CLS])                                                                                               \
    SBX_KLAUNCH(h, SBX_K_PERMUTE_BLOCK, (k_permute_block_rows<I, VB, br_cap(CLS), THREADS>), dim3(n_block[CLS]),  \
                dim3(THREADS), rec, col_in, val_in, col_order, rpo, block_rows + (CLS)*block_stride, col_out,     \
                val_out, rpasses, st)
  BLOCK_ROWS(0, 256);
  BLOCK_ROWS(1, 512);
  BLOCK_ROWS(2, 1024);
  if constexpr (VB != 8) BLOCK_ROWS(3, 1024);  // 8-byte values: 16384 entries do not fit LDS, those rows are "long"
#undef BLOCK_ROWS
  SBX_LAUNCH_CHECK(h);
  return SBX_OK;
}

// longer rows: flat gather, device radix sort on (row rank, column), scatter back
template <int VB>
int long_rows_path(sbx_handle_t h, const int2 *rec, const int32_t *col_in, const char *val_in, const int32_t *col_order,
                   const int32_t *rpo, int32_t *col_out, char *val_out, int64_t m, const int32_t *long_rows,
                   unsigned n_long, int64_t long_nnz, PermState *st) {
  typedef int32_t I;
  uint32_t *loff = nullptr;
  uint64_t *ka = nullptr, *kb = nullptr;
  char *pa = nullptr, *pb = nullptr;
  SBX_TRY(sbx_salloc(h, (size_t)n_long + 1, &loff));
  SBX_TRY(sbx_salloc(h, (size_t)long_nnz, &ka));
  SBX_TRY(sbx_salloc(h, (size_t)long_nnz, &kb));
  if (VB) {
    SBX_TRY(sbx_salloc(h, (size_t)long_nnz * VB, &pa));
    SBX_TRY(sbx_salloc(h, (size_t)long_nnz * VB, &pb));
  }
  SBX_KLAUNCH(h, SBX_K_PERMUTE_LONG, k_long_lengths<I>, dim3((n_long + 255) / 256), dim3(256), rpo, long_rows, loff,
                     (int)n_long);
  SBX_TRY(sbx_exclusive_scan_u32(h, loff, loff, n_long, nullptr));
  SBX_KLAUNCH(h, SBX_K_PERMUTE_LONG, (k_long_gather<I, VB>), dim3(sbx_grid_for(long_nnz, 256, 8192)), dim3(256), rec,
                     col_in, val_in, col_order, long_rows, (const uint32_t *)loff, (int)n_long, long_nnz, ka, pa, st);
  SBX_LAUNCH_CHECK(h);
  sbx_radix_pass passes[16];
  const int np = sbx_radix_plan(0, sbx_bits_for(m > 0 ? (uint64_t)(m - 1) : 0), 32,
                                32 + sbx_bits_for((uint64_t)(n_long - 1)), passes);
  int in_b = 0;
  PermState hs2;  // rows that are already ordered (row-wise permutes) skip the sort
  SBX_TRY(sbx_readback(h, &hs2, st, sizeof(PermState)));
  if (hs2.long_unsorted) SBX_TRY(sbx_radix_sort(h, 8, VB, ka, kb, pa, pb, long_nnz, passes, np, &in_b));
  SBX_KLAUNCH(h, SBX_K_PERMUTE_LONG, (k_long_scatter<I, VB>), dim3(sbx_grid_for(long_nnz, 256, 8192)), dim3(256),
                     (const uint64_t *)(in_b ? kb : ka), (const char *)(in_b ? pb : pa), rpo, long_rows,
                     (const uint32_t *)loff, long_nnz, col_out, val_out, st);
  SBX_LAUNCH_CHECK(h);
  return SBX_OK;
}

// Sort stage shared by permute and csr_sort_rows: rows of `rpo` (nr rows, already
// on device) are produced from the source CSR through the row/col maps.
template <int VB>
int sort_stage(sbx_handle_t h, sbx_value_type vt, const int2 *rec, const int32_t *col_in, const char *val_in,
               const int32_t *col_order, const int32_t *rpo, int32_t *col_out, char *val_out, int64_t nr, int64_t m,
               int64_t total, const int32_t *long_rows, unsigned n_long, int64_t long_nnz, const int32_t *block_rows,
               const unsigned *n_block, int64_t block_stride, PermState *st) {
  typedef int32_t I;
  // The three paths write disjoint rows of the output and only read the inputs: with more than one of them
  // present they run on streams of their own (tile kernel on the caller's stream), so the block-row kernels'
  // tails and the long-row path's short, latency-bound launches hide behind the tile kernel.  While the
  // profiler is on they run back to back, so that a kernel's event time is its own.
  const bool has_block = (n_block[0] | n_block[1] | n_block[2] | n_block[3]) != 0;
  const bool fork = !h->prof_on && permute_overlap() && total > 0 && (has_block || n_long);
  hipStream_t main_stream = h->stream;
  if (fork) {
    SBX_TRY(sbx_aux_streams(h));
    SBX_HIP(h, hipEventRecord(h->aux_event[0], main_stream));
    h->aux_dirty = true;
    if (has_block) SBX_HIP(h, hipStreamWaitEvent(h->aux_stream[0], h->aux_event[0], 0));
    if (n_long) SBX_HIP(h, hipStreamWaitEvent(h->aux_stream[1], h->aux_event[0], 0));
  }
  if (total > 0) {
    const unsigned tiles = (unsigned)((total + PT_TILE - 1) / PT_TILE);
    SBX_KLAUNCH(h, SBX_K_PERMUTE_TILE, (k_permute_tile<I, VB>), dim3(tiles), dim3(PT_THREADS), rec, col_in, val_in,
                col_order, rpo, col_out, val_out, nr, st, sbx_bits_for(m > 0 ? (uint64_t)(m - 1) : 0));
    SBX_LAUNCH_CHECK(h);
  }
  if (has_block) {
    if (fork) h->stream = h->aux_stream[0];
    const int rc = block_rows_path<VB>(h, rec, col_in, val_in, col_order, rpo, col_out, val_out, m, block_rows, n_block,
                                       block_stride, st);
    if (fork) {
      if (rc == SBX_OK && hipEventRecord(h->aux_event[1], h->stream) != hipSuccess) { h->stream = main_stream; SBX_FAIL(h, SBX_ERR_HIP, "hipEventRecord failed"); }
      h->stream = main_stream;
    }
    SBX_TRY(rc);
  }
  if (n_long) {
    if (fork) h->stream = h->aux_stream[1];
    const int rc = long_rows_path<VB>(h, rec, col_in, val_in, col_order, rpo, col_out, val_out, m, long_rows, n_long,
                                      long_nnz, st);
    if (fork) {
      if (rc == SBX_OK && hipEventRecord(h->aux_event[2], h->stream) != hipSuccess) { h->stream = main_stream; SBX_FAIL(h, SBX_ERR_HIP, "hipEventRecord failed"); }
      h->stream = main_stream;
    }
    SBX_TRY(rc);
  }
  if (fork) {  // join
    if (has_block) SBX_HIP(h, hipStreamWaitEvent(main_stream, h->aux_event[1], 0));
    if (n_long) SBX_HIP(h, hipStreamWaitEvent(main_stream, h->aux_event[2], 0));
    h->aux_dirty = false;
  }
  if (VB) SBX_TRY(launch_fix<I>(h, vt, rpo, col_out, val_out, nr, st));
  return SBX_OK;
}

}  // namespace

#define SBX_REQUIRE(h, cond, msg)                                       \
  do {                                                                  \
    if (!(cond)) SBX_FAIL(h, SBX_ERR_BAD_ARG, "%s: %s", __func__, msg); \
  } while (0)
#define SBX_ONLY_I32(h, it)                                                                              \
  do {                                                                                                   \
    if ((it) != SBX_I32) SBX_FAIL(h, SBX_ERR_UNSUPPORTED, "%s: 64-bit indices not built yet", __func__); \
  } while (0)

extern "C" int sbx_inverse_permutation(sbx_handle_t h, sbx_index_type it, int64_t n, const void *perm,
                                       void *inv_out) {
  if (!h) return SBX_ERR_BAD_ARG;
  SBX_REQUIRE(h, n >= 0 && (n == 0 || (perm && inv_out)), "bad argument");
  if (it == SBX_I64) return sbx_i64_inverse_permutation(h, n, perm, inv_out);
  SBX_TRY(sbx_arena_begin(h));
  if (n == 0) return SBX_OK;
  SBX_KLAUNCH(h, SBX_K_PERMUTE_PREP, k_invert<int32_t>, dim3(sbx_grid_for(n, 256, 8192)), dim3(256),
                     (const int32_t *)perm, (int32_t *)inv_out, n);
  SBX_LAUNCH_CHECK(h);
  return SBX_OK;
}

extern "C" int sbx_permute_array(sbx_handle_t h, sbx_index_type it, sbx_value_type vt, int64_t n, const void *order,
                                 const void *vals, void *out) {
  if (!h) return SBX_ERR_BAD_ARG;
  SBX_REQUIRE(h, n >= 0 && (n == 0 || (order && vals && out)), "bad argument");
  if (it == SBX_I64) return sbx_i64_permute_array(h, vt, n, order, vals, out);
  const int vb = sbx_value_bytes(vt);
  SBX_REQUIRE(h, vb == 4 || vb == 8, "value type must be 4 or 8 bytes");
  SBX_TRY(sbx_arena_begin(h));
  if (n == 0) return SBX_OK;
  const unsigned grid = sbx_grid_for(n, 256, 8192);
  if (vb == 4)
    SBX_KLAUNCH(h, SBX_K_PERMUTE_PREP, (k_permute_array<int32_t, 4>), dim3(grid), dim3(256), (const int32_t *)order,
                       (const char *)vals, (char *)out, n);
  else
    SBX_KLAUNCH(h, SBX_K_PERMUTE_PREP, (k_permute_array<int32_t, 8>), dim3(grid), dim3(256), (const int32_t *)order,
                       (const char *)vals, (char *)out, n);
  SBX_LAUNCH_CHECK(h);
  return SBX_OK;
}

extern "C" int sbx_permute_csr_rows(sbx_handle_t h, sbx_index_type it, sbx_value_type vt, int64_t n, int64_t m,
                                    int64_t nnz, const void *row_ptr, const void *col, const void *val,
                                    const void *row_order, const void *col_order, int64_t row_begin, int64_t row_end,
                                    void *row_ptr_out, void *col_out, void *val_out, int64_t out_capacity,
                                    int64_t *shard_nnz_host) {
  if (!h) return SBX_ERR_BAD_ARG;
  SBX_REQUIRE(h, n >= 0 && m >= 0 && nnz >= 0 && row_ptr && row_ptr_out, "bad argument");
  SBX_REQUIRE(h, 0 <= row_begin && row_begin <= row_end && row_end <= n, "bad row range");
  SBX_REQUIRE(h, nnz == 0 || (col && col_out), "col/col_out required");
  SBX_REQUIRE(h, nnz < ((int64_t)1 << 31) && n < ((int64_t)1 << 31) - 1, "dimension exceeds int32");
  if (it == SBX_I64)
    return sbx_i64_permute_csr_rows(h, vt, n, m, nnz, row_ptr, col, val, row_order, col_order, row_begin, row_end,
                                    row_ptr_out, col_out, val_out, out_capacity, shard_nnz_host);
  const int vb = (val && val_out) ? sbx_value_bytes(vt) : 0;
  SBX_REQUIRE(h, vb >= 0, "unknown value type");
  SBX_TRY(sbx_arena_begin(h));
  typedef int32_t I;
  const int64_t nr = row_end - row_begin;
  I *rpo = (I *)row_ptr_out;
  if (shard_nnz_host) *shard_nnz_host = 0;
  if (nr == 0) return sbx_fill_i32(h, rpo, 0, 1);

  // (row length, source offset) per new row, written from the old-row side; lengths -> scan -> row_ptr_out
  PermState *st = nullptr;
  int2 *rec = nullptr;
  SBX_TRY(sbx_salloc(h, 1, &st));
  SBX_TRY(sbx_salloc(h, (size_t)nr, &rec));
  SBX_HIP(h, hipMemsetAsync(st, 0, sizeof(PermState), h->stream));
  SBX_KLAUNCH(h, SBX_K_PERMUTE_PREP, k_rowwise_prep<I>, dim3(sbx_grid_for(n, 256, 8192)), dim3(256), (const I *)row_ptr,
              (const I *)row_order, n, row_begin, nr, rec);
  I *long_rows = nullptr, *block_rows = nullptr;
  const int block_cap = vb == 8 ? BlockRowCap<8>::value : BlockRowCap<4>::value;
  int64_t block_stride = 0;
  if (col_order) {  // the sorting pipeline needs the rows that do not fit a tile listed by class
    int64_t cap_long = nnz / PT_TILE + 1;
    if (cap_long > nr) cap_long = nr;
    SBX_TRY(sbx_salloc(h, (size_t)cap_long, &long_rows));
    SBX_TRY(sbx_salloc(h, (size_t)cap_long * BR_CLASSES, &block_rows));
    block_stride = cap_long;
    SBX_KLAUNCH(h, SBX_K_PERMUTE_PREP, k_rec_classify<I>, dim3(sbx_grid_for(nr + 1, 256, 2048)), dim3(256),
                (const int2 *)rec, rpo, nr, long_rows, block_rows, block_stride, block_cap, st);
  } else {
    SBX_KLAUNCH(h, SBX_K_PERMUTE_PREP, k_rowwise_lengths<I>, dim3(sbx_grid_for(nr + 1, 256, 8192)), dim3(256),
                (const int2 *)rec, nr, rpo);
  }
  SBX_LAUNCH_CHECK(h);
  SBX_TRY(sbx_exclusive_scan_i32(h, rpo, rpo, nr + 1, nullptr));
  int64_t total = nnz;  // the full permute keeps every nonzero; a shard has to ask
  PermState hs;
  memset(&hs, 0, sizeof(hs));
  if (nr != n || col_order) {
    SBX_KLAUNCH(h, SBX_K_PERMUTE_PREP, k_store_total<I>, dim3(1), dim3(1), (const I *)rpo, nr, st);
    SBX_TRY(sbx_readback(h, &hs, st, sizeof(PermState)));
    total = (int64_t)hs.total;
  }
  if (shard_nnz_host) *shard_nnz_host = total;
  if (total > out_capacity)
    SBX_FAIL(h, SBX_ERR_BAD_ARG, "sbx_permute_csr_rows: shard needs %lld entries, capacity %lld", (long long)total,
             (long long)out_capacity);
  if (total == 0) return SBX_OK;
  if (!col_order) {
    // row-wise: a segmented copy; sorted input rows (every CSR that went through a constructor) end here
    const unsigned tiles = (unsigned)((total + PC_TILE - 1) / PC_TILE);
    I *tile_row = nullptr;
    SBX_TRY(sbx_salloc(h, (size_t)tiles + 1, &tile_row));
    SBX_KLAUNCH(h, SBX_K_PERMUTE_PREP, k_tile_rows<I>, dim3(tiles / 256 + 1), dim3(256), (const I *)rpo, nr, total,
                PC_TILE, (int64_t)tiles, tile_row);
#define COPY(VBX)                                                                                              \
  SBX_KLAUNCH(h, SBX_K_PERMUTE_TILE, (k_permute_copy<I, VBX>), dim3(tiles), dim3(PC_THREADS), (const int2 *)rec, \
              (const I *)tile_row, (const I *)col, (const char *)val, (const I *)rpo, (I *)col_out, (char *)val_out, \
              nr, total, st)
    if (vb == 0) COPY(0);
    else if (vb == 4) COPY(4);
    else COPY(8);
#undef COPY
    SBX_LAUNCH_CHECK(h);
    SBX_PROF_BYTES(h, SBX_K_PERMUTE_TILE, total * (int64_t)(2 * (sizeof(I) + vb)));
    PermState hc;
    SBX_TRY(sbx_readback(h, &hc, st, sizeof(PermState)));
    if (!hc.any_unsorted) return SBX_OK;
    // some input row is out of order: redo with the sorting pipeline (row_ptr_out is already final)
    int64_t cap_long = nnz / PT_TILE + 1;
    if (cap_long > nr) cap_long = nr;
    SBX_TRY(sbx_salloc(h, (size_t)cap_long, &long_rows));
    SBX_TRY(sbx_salloc(h, (size_t)cap_long * BR_CLASSES, &block_rows));
    block_stride = cap_long;
    SBX_HIP(h, hipMemsetAsync(st, 0, sizeof(PermState), h->stream));
    SBX_KLAUNCH(h, SBX_K_PERMUTE_PREP, k_rec_classify<I>, dim3(sbx_grid_for(nr + 1, 256, 2048)), dim3(256),
                (const int2 *)rec, (I *)nullptr, nr, long_rows, block_rows, block_stride, block_cap, st);
    SBX_LAUNCH_CHECK(h);
    SBX_TRY(sbx_readback(h, &hs, st, sizeof(PermState)));
  }
  int rc;
#define STAGE(VBX)                                                                                                  \
  rc = sort_stage<VBX>(h, vt, (const int2 *)rec, (const I *)col, (const char *)val, (const I *)col_order, rpo,      \
                       (I *)col_out, (char *)val_out, nr, m, total, long_rows, hs.n_long, (int64_t)hs.long_nnz,     \
                       block_rows, hs.n_block, block_stride, st)
  if (vb == 0) STAGE(0);
  else if (vb == 4) STAGE(4);
  else STAGE(8);
#undef STAGE
  return rc;
}

extern "C" int sbx_permute_csr(sbx_handle_t h, sbx_index_type it, sbx_value_type vt, int64_t n, int64_t m, int64_t nnz,
                               const void *row_ptr, const void *col, const void *val, const void *row_order,
                               const void *col_order, void *row_ptr_out, void *col_out, void *val_out) {
  return sbx_permute_csr_rows(h, it, vt, n, m, nnz, row_ptr, col, val, row_order, col_order, 0, n, row_ptr_out,
                              col_out, val_out, nnz, nullptr);
}

// A4: the CSR constructor's "if any row is unsorted, sort every row" in place.
extern "C" int sbx_csr_sort_rows(sbx_handle_t h, sbx_index_type it, sbx_value_type vt, int64_t n, int64_t m,
                                 int64_t nnz, const void *row_ptr, void *col, void *val) {
  if (!h) return SBX_ERR_BAD_ARG;
  SBX_REQUIRE(h, n >= 0 && nnz >= 0 && row_ptr && (nnz == 0 || col), "bad argument");
  if (it == SBX_I64) return sbx_i64_csr_sort_rows(h, vt, n, m, nnz, row_ptr, col, val);
  if (nnz <= 1 || n == 0) return SBX_OK;
  int sorted = 1;
  SBX_TRY(sbx_csr_rows_sorted(h, it, n, row_ptr, col, &sorted));  // csr.cc:102-116
  if (sorted) return SBX_OK;
  // out-of-place into scratch, then copied back (tiles may read rows another
  // tile has already rewritten if the sort ran in place across tiles)
  const int vb = val ? sbx_value_bytes(vt) : 0;
  SBX_REQUIRE(h, vb >= 0, "unknown value type");
  SBX_TRY(sbx_arena_begin(h));
  typedef int32_t I;
  PermState *st = nullptr;
  I *long_rows = nullptr, *block_rows = nullptr, *ctmp = nullptr;
  int2 *rec = nullptr;
  const int block_cap = vb == 8 ? BlockRowCap<8>::value : BlockRowCap<4>::value;
  char *vtmp = nullptr;
  SBX_TRY(sbx_salloc(h, 1, &st));
  SBX_TRY(sbx_salloc(h, (size_t)n, &rec));
  SBX_TRY(sbx_salloc(h, (size_t)n, &long_rows));
  SBX_TRY(sbx_salloc(h, (size_t)n * BR_CLASSES, &block_rows));
  SBX_TRY(sbx_salloc(h, (size_t)nnz, &ctmp));
  if (vb) SBX_TRY(sbx_salloc(h, (size_t)nnz * vb, &vtmp));
  SBX_HIP(h, hipMemsetAsync(st, 0, sizeof(PermState), h->stream));
  SBX_KLAUNCH(h, SBX_K_PERMUTE_PREP, k_rowwise_prep<I>, dim3(sbx_grid_for(n, 256, 8192)), dim3(256), (const I *)row_ptr,
              (const I *)nullptr, n, (int64_t)0, n, rec);
  SBX_KLAUNCH(h, SBX_K_PERMUTE_PREP, k_rec_classify<I>, dim3(sbx_grid_for(n + 1, 256, 2048)), dim3(256),
              (const int2 *)rec, (I *)nullptr, n, long_rows, block_rows, n, block_cap, st);
  SBX_LAUNCH_CHECK(h);
  PermState hs;
  SBX_TRY(sbx_readback(h, &hs, st, sizeof(PermState)));
  int rc;
#define STAGE(VBX)                                                                                               \
  rc = sort_stage<VBX>(h, vt, (const int2 *)rec, (const I *)col, (const char *)val, (const I *)nullptr,          \
                       (const I *)row_ptr, ctmp, vtmp, n, m, nnz, long_rows, hs.n_long, (int64_t)hs.long_nnz,    \
                       block_rows, hs.n_block, (int64_t)n, st)
  if (vb == 0) STAGE(0);
  else if (vb == 4) STAGE(4);
  else STAGE(8);
#undef STAGE
  SBX_TRY(rc);
  SBX_HIP(h, hipMemcpyAsync(col, ctmp, (size_t)nnz * sizeof(I), hipMemcpyDeviceToDevice, h->stream));
  if (vb) SBX_HIP(h, hipMemcpyAsync(val, vtmp, (size_t)nnz * vb, hipMemcpyDeviceToDevice, h->stream));
  return SBX_OK;
}
