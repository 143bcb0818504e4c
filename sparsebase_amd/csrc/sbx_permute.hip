// sbx_permute.hip — permutation apply and the CSR-constructor row sort.
//
//   A5  permute/permute_order_two.cc:23-79   sbx_permute_csr, sbx_permute_csr_rows
//   A4  format/csr.cc:118-157                sbx_csr_sort_rows (and the sort A5 ends in)
//   A13 bases/reorder_base.h:663-672         sbx_inverse_permutation
//       permute/permute_order_one.cc:18-37   sbx_permute_array
//
// Data flow of one permute (new rows [row_begin,row_end) — the multi-GPU shard):
//   k_rowwise_prep  from the old-row side: rec[row_order[u]] = (length, source offset)   (12n B r, 8n B w)
//   k_classify_scan lengths -> row_ptr_out (single-pass scan), the short rows' prefix sums, long rows listed by class
//   no column map:  k_permute_copy, a segmented copy (+ in-row order check; unsorted input
//                   rows redo the call through the sorting kernels below)
//   column map:
//   k_permute_tile  one workgroup per PT_W output nonzeros: gathers whole old rows (col relabelled
//                   through col_order) and sorts every row of <= PT_LMAX entries in LDS with ONE
//                   bucket-rank pass (below); streams them out.
//   k_permute_block_rows  rows of PT_LMAX..8K entries: one workgroup per row in three capacity
//                   classes, the same bucket-rank sort.
//   long rows       longer rows: gathered into a compact buffer, sorted by
//                   (row, col) with the device radix sort, scattered back.
//   k_fix_dup_runs  only if some row was unsorted AND duplicate columns exist:
//                   orders equal-column runs by value (std::less<pair<col,val>>).
// HBM traffic per nonzero in the copy, tile and block paths is the compulsory 2*(I+V) bytes.
//
// Bucket-rank sort (what replaces the five LDS radix passes of round 1).  A row of L relabelled
// columns with smallest / largest value mn / mx gets NB = 2^ceil(log2 L) order-preserving buckets
// of width 2^shift, shift = bits(mx - mn) - log2 NB: bucket = (col - mn) >> shift.  Rows sit back to
// back in the tile, so the row whose first entry is at tile position hp owns the counter words
// [2 hp, 2 hp + NB) and every row is sorted by the same four sweeps over the whole tile: count
// (one LDS atomic per entry), inclusive scan of the counters, placement (one returning atomic:
// the entry's word = low `shift` bits of the column | position in the row goes to its bucket's
// slot range) and ranking (an entry counts the smaller words of its own bucket, ~1 on average).
// Final position = bucket start + rank; entries move in place through registers.  The word order
// is (column, original position): the sort is stable.  A bucket of more than BK_MAX entries (a row
// whose columns cluster far more tightly than its range) sends the tile to the LSD radix sort kept
// below as the distribution-independent path.
#include "sbx_device.h"
#include "sbx_internal.h"

namespace {

constexpr int PT_THREADS = 64;
#ifndef SBX_PT_ITEMS
#define SBX_PT_ITEMS 8
#endif
#ifndef SBX_PT_W
#define SBX_PT_W 384
#endif
#ifndef SBX_PT_LMAX
#define SBX_PT_LMAX 128
#endif
constexpr int PT_ITEMS = SBX_PT_ITEMS;  // a multiple of 4: the thread-consecutive sweeps move 16 bytes per LDS access
constexpr int PT_CAP = PT_THREADS * PT_ITEMS;  // LDS capacity of one tile, in entries
constexpr int PT_W = SBX_PT_W;                      // a tile owns the rows that start in a window of PT_W output positions
// waves per SIMD the persistent tile kernel is compiled for: 4 = at most 128 VGPRs, where the kernels that carry values
// do not spill (uncapped they take 169 and lose occupancy: Permute2D 1.56 vs 1.49 ms); the pattern-only kernel spills 9
// registers under the cap and is 7 % faster without it
#ifndef PT_MIN_WAVES
#define PT_MIN_WAVES 4
#endif
constexpr int PT_LMAX = SBX_PT_LMAX;                   // rows up to this many entries are sorted by the tile kernel
static_assert(PT_W + PT_LMAX - 1 <= PT_CAP, "a tile must hold its window plus the tail of its last row");
constexpr int BK_SHORT = 8;   // rows up to this length are one bucket (plain all-pairs ranking)
constexpr int BK_MAX = 96;    // a larger bucket sends the tile / row to the radix sort
constexpr int BK_REFINE = 6;  // a larger bucket after the first level: every bucket is split again in proportion to its count
// one workgroup sorts one row in LDS; capacity classes so a 1100-entry row does not pay for 8192 slots
constexpr int BR_CLASSES = 6;
__host__ __device__ constexpr int br_cap(int cls) { return 256 << cls; }  // 256, 512, 1024, 2048, 4096, 8192
template <int VB> struct BlockRowCap { static constexpr int value = 8192; };
template <> struct BlockRowCap<8> { static constexpr int value = 4096; };   // 8-byte values: 8192 entries do not fit LDS

struct PermState {            // device-resident flags/counters of one call
  unsigned any_unsorted;      // some row had col[j] < col[j-1] after relabelling (csr.cc:102-116)
  unsigned any_dup;           // some row holds a duplicate column
  unsigned n_long;            // rows longer than the block-row capacity (global radix path)
  unsigned n_block[BR_CLASSES];  // rows in (PT_LMAX, capacity], by capacity class: one workgroup each
  unsigned long long long_nnz;
  unsigned long long block_nnz;  // nonzeros of the one-workgroup-per-row classes
  unsigned long long total;   // nnz of the shard
  unsigned long_unsorted;     // some row of the global-radix class is out of order
  unsigned n_fb_rows;         // rows / tiles whose columns cluster: listed for the radix kernels
  unsigned n_fb_tiles;
  unsigned n_seg_fb_rows;     // long-row segments whose columns cluster (listed for the radix kernel)
  unsigned n_seg[2];          // long-row segments of <= 4096 / <= 8192 entries (virtual rows of the one-workgroup classes)
  unsigned n_long_fb;         // long rows the segment path hands to the global radix sort (already ordered, or a segment too full)
  unsigned long long long_fb_nnz;
};

// The counters every workgroup of k_classify_scan adds to, one per 128-byte line (words of one line queue behind each
// other at the L2's atomic unit: 2048 workgroups x 9 counters on one line were half of that kernel's 65 us), behind the
// PermState of the call in ONE allocation, so that one read-back brings both; perm_fetch() folds them into the state.
constexpr int PH_STRIDE = 32;                      // words per line
constexpr int PH_LONG = BR_CLASSES, PH_LONG_NNZ = BR_CLASSES + 1, PH_BLOCK_NNZ = BR_CLASSES + 2, PH_SLOTS = BR_CLASSES + 3;
struct PermAll {
  PermState st;
  alignas(128) unsigned hot[PH_SLOTS * PH_STRIDE];
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

// The lists of rows too long for the tile kernel are appended to through a handful of counter words; a few thousand
// long rows spread over millions would queue one atomic each on them (~88 atomics/us per word: 100 us for 10 K rows),
// so every workgroup stages the rows it finds in LDS and reserves list space once per class (the order inside a list
// does not matter).
constexpr int RC_STAGE = 512;  // staged rows per class and workgroup before an early flush

// ---- classification + both prefix sums ----------------------------------------------------------------------
// rpo = exclusive scan of the new rows' lengths, sp = the same over the lengths of the rows the tile kernel sorts
// (<= PT_LMAX), and the lists of longer rows by class.  Three launches: the tiles' sums (k_classify_reduce), their
// exclusive prefix by one workgroup (k_classify_prefix), and the pass that scans, writes and lists (k_classify_scan).
// (Rounds 2 - 4 did it in ONE launch with a decoupled look-back over the tiles: 55 us for 4.2 M rows — its 1024 tiles wait
// for each other in a chain — where the three plain launches take ~32: two reads of the records instead of one, no chain.)
#ifndef SBX_CS_ITEMS
#define SBX_CS_ITEMS 16
#endif
constexpr int CS_ITEMS = SBX_CS_ITEMS;
constexpr int CS_TILE = 256 * CS_ITEMS;
// a thread's CS_ITEMS row lengths: two (length, source) records per 16-byte load
__device__ __forceinline__ void cs_load_lengths(const int2 *__restrict__ rec, int64_t base, int64_t nr, int (&d)[CS_ITEMS]) {
#pragma unroll
  for (int k = 0; k < CS_ITEMS; k += 2) {
    int4 q = make_int4(0, 0, 0, 0);
    if (base + k + 1 < nr) q = *(const int4 *)(rec + base + k);
    else if (base + k < nr) q.x = rec[base + k].x;
    d[k] = q.x;
    d[k + 1] = q.z;
  }
}

// tile_sum[2 t] / [2 t + 1] = sum of all / of the short rows' lengths of tile t
__global__ __launch_bounds__(256) void k_classify_reduce(const int2 *__restrict__ rec, int64_t nr,
                                                         unsigned long long *__restrict__ tile_sum) {
  __shared__ unsigned long long s_red[2][4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int d[CS_ITEMS];
  cs_load_lengths(rec, (int64_t)blockIdx.x * CS_TILE + (int64_t)tid * CS_ITEMS, nr, d);
  unsigned long long a = 0, sh = 0;
#pragma unroll
  for (int k = 0; k < CS_ITEMS; k++) {
    a += (unsigned)d[k];
    sh += d[k] <= PT_LMAX ? (unsigned)d[k] : 0u;
  }
  a = sbx_wave_sum(a), sh = sbx_wave_sum(sh);
  if (lane == 0) s_red[0][wv] = a, s_red[1][wv] = sh;
  __syncthreads();
  if (tid == 0) {
    tile_sum[2 * (size_t)blockIdx.x] = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
    tile_sum[2 * (size_t)blockIdx.x + 1] = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
  }
}

// in place: tile_sum -> the exclusive prefixes of both sums; the grand totals close rpo / sp and go to st->total
template <typename I>
__global__ __launch_bounds__(1024) void k_classify_prefix(unsigned long long *__restrict__ tile_sum, int64_t tiles,
                                                          I *__restrict__ rpo, I *__restrict__ sp, int64_t nr,
                                                          PermState *__restrict__ st) {
  __shared__ unsigned long long s_w[2][16];
  __shared__ unsigned long long s_carry[2];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid < 2) s_carry[tid] = 0;
  __syncthreads();
  for (int64_t t0 = 0; t0 < tiles; t0 += 1024) {
    const int64_t t = t0 + tid;
    const unsigned long long a = t < tiles ? tile_sum[2 * t] : 0ull, sh = t < tiles ? tile_sum[2 * t + 1] : 0ull;
    const unsigned long long ia = sbx_wave_inclusive_sum(a), is = sbx_wave_inclusive_sum(sh);
    if (lane == 63) s_w[0][wv] = ia, s_w[1][wv] = is;
    __syncthreads();
    unsigned long long ba = s_carry[0], bs = s_carry[1], ta = 0, ts = 0;
    for (int i = 0; i < 16; i++) {
      if (i < wv) ba += s_w[0][i], bs += s_w[1][i];
      ta += s_w[0][i], ts += s_w[1][i];
    }
    if (t < tiles) tile_sum[2 * t] = ba + ia - a, tile_sum[2 * t + 1] = bs + is - sh;
    __syncthreads();
    if (tid == 0) s_carry[0] += ta, s_carry[1] += ts;
    __syncthreads();
  }
  if (tid == 0) {
    if (rpo) rpo[nr] = (I)s_carry[0];
    sp[nr] = (I)s_carry[1];
    st->total = s_carry[0];
  }
}

template <typename I>
__global__ __launch_bounds__(256) void k_classify_scan(const int2 *__restrict__ rec, I *__restrict__ rpo,
                                                       I *__restrict__ sp, int64_t nr, I *__restrict__ long_rows,
                                                       I *__restrict__ block_rows, int64_t block_stride, int block_cap,
                                                       PermState *__restrict__ st,
                                                       const unsigned long long *__restrict__ tile_prefix, int rpo_aligned,
                                                       I *__restrict__ tile_first) {
  constexpr int NC = BR_CLASSES + 1;  // class BR_CLASSES = rows for the global radix path
  __shared__ unsigned s_base[NC];
  __shared__ unsigned long long s_red[2][4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t tile = blockIdx.x;
  const int64_t base = tile * CS_TILE + (int64_t)tid * CS_ITEMS;  // this thread's first row
  int d[CS_ITEMS];
  cs_load_lengths(rec, base, nr, d);
  unsigned long long sum_all = 0, sum_short = 0;
  unsigned long long loc_all[CS_ITEMS], loc_short[CS_ITEMS];
#pragma unroll
  for (int k = 0; k < CS_ITEMS; k++) {
    loc_all[k] = sum_all;
    loc_short[k] = sum_short;
    sum_all += (unsigned)d[k];
    sum_short += d[k] <= PT_LMAX ? (unsigned)d[k] : 0u;
  }
  // exclusive prefix of the thread sums inside the tile
  const unsigned long long inc_a = sbx_wave_inclusive_sum(sum_all), inc_s = sbx_wave_inclusive_sum(sum_short);
  if (lane == 63) {
    s_red[0][wv] = inc_a;
    s_red[1][wv] = inc_s;
  }
  __syncthreads();
  unsigned long long off_a = inc_a - sum_all, off_s = inc_s - sum_short;
#pragma unroll
  for (int i = 0; i < 4; i++)
    if (i < wv) off_a += s_red[0][i], off_s += s_red[1][i];
  off_a += tile_prefix[2 * tile];
  off_s += tile_prefix[2 * tile + 1];
  if (base < nr) {
    I oa[CS_ITEMS], os[CS_ITEMS];
#pragma unroll
    for (int k = 0; k < CS_ITEMS; k++) {
      oa[k] = (I)(off_a + loc_all[k]);
      os[k] = (I)(off_s + loc_short[k]);
    }
    if (base + CS_ITEMS <= nr) {
#pragma unroll
      for (int k = 0; k < CS_ITEMS; k += 4) {
        if (sizeof(I) == 4 && rpo && rpo_aligned) {
          *(int4 *)(void *)(rpo + base + k) = make_int4((int)oa[k], (int)oa[k + 1], (int)oa[k + 2], (int)oa[k + 3]);
        } else if (rpo) {  // a caller's row_ptr_out at an odd offset, or 64-bit indices
          rpo[base + k] = oa[k], rpo[base + k + 1] = oa[k + 1], rpo[base + k + 2] = oa[k + 2], rpo[base + k + 3] = oa[k + 3];
        }
        if (sizeof(I) == 4)
          *(int4 *)(void *)(sp + base + k) = make_int4((int)os[k], (int)os[k + 1], (int)os[k + 2], (int)os[k + 3]);
        else
          sp[base + k] = os[k], sp[base + k + 1] = os[k + 1], sp[base + k + 2] = os[k + 2], sp[base + k + 3] = os[k + 3];
      }
    } else {
      for (int k = 0; base + k < nr; k++) {
        if (rpo) rpo[base + k] = oa[k];
        sp[base + k] = os[k];
      }
    }
  }
  // tile_first[t] = first row whose range in the short rows' entry space starts at or behind t * PT_W (what the tile
  // kernel's tiles are cut by; k_tile_first finds the same by binary search): the row in front of it is the short row
  // that reaches or crosses that position — at most one t per row, a row being shorter than a window.  The entries
  // behind the last tile keep the caller's fill (nr).
  if (tile_first) {
    if (base == 0 && tid == 0) tile_first[0] = 0;
#pragma unroll
    for (int k = 0; k < CS_ITEMS; k++) {
      const unsigned len = (base + k < nr && d[k] > 0 && d[k] <= PT_LMAX) ? (unsigned)d[k] : 0u;
      if (len) {
        const unsigned long long s0 = off_s + loc_short[k], t = (s0 + len) / (unsigned)PT_W;
        if (t * (unsigned)PT_W > s0) tile_first[t] = (I)(base + k + 1);
      }
    }
  }
  // the lists of rows too long for the tile kernel: every thread counts its rows per class (seven 16-bit counters in two
  // words: a workgroup holds 4096 rows), ONE scan of the packed counters gives each thread its place inside the
  // workgroup's stretch of every list, one reservation per class and workgroup, and the rows are written where they
  // belong.  (Until round 5 the rows were staged in LDS sixteen rounds of barriers long: 25 of the kernel's 40 us.)
  if (block_rows == nullptr) return;
  static_assert(NC <= 8 && CS_TILE < 65536, "seven 16-bit counters in two words");
  unsigned long long c_lo = 0, c_hi = 0, lnz = 0, bnz = 0;
#pragma unroll
  for (int k = 0; k < CS_ITEMS; k++) {
    const int dd = d[k];
    if (base + k < nr && dd > PT_LMAX) {
      int cls = BR_CLASSES;
      if (dd <= block_cap) {
        cls = 0;
        while (dd > br_cap(cls)) cls++;
      }
      if (cls < 4) c_lo += 1ull << (16 * cls);
      else c_hi += 1ull << (16 * (cls - 4));
      if (cls == BR_CLASSES) lnz += (unsigned)dd;
      else bnz += (unsigned)dd;
    }
  }
  __shared__ unsigned long long s_c[2][4], s_nz[2][4];
  const unsigned long long i_lo = sbx_wave_inclusive_sum(c_lo), i_hi = sbx_wave_inclusive_sum(c_hi);
  lnz = sbx_wave_sum(lnz), bnz = sbx_wave_sum(bnz);
  __syncthreads();  // (s_red of the scan above has been read)
  if (lane == 63) s_c[0][wv] = i_lo, s_c[1][wv] = i_hi;
  if (lane == 0) s_nz[0][wv] = lnz, s_nz[1][wv] = bnz;
  __syncthreads();
  unsigned long long e_lo = i_lo - c_lo, e_hi = i_hi - c_hi, t_lo = 0, t_hi = 0;
#pragma unroll
  for (int w2 = 0; w2 < 4; w2++) {
    if (w2 < wv) e_lo += s_c[0][w2], e_hi += s_c[1][w2];
    t_lo += s_c[0][w2], t_hi += s_c[1][w2];
  }
  if (t_lo == 0 && t_hi == 0) return;  // (uniform: nothing to list in this tile)
  if (tid < NC) {
    const unsigned cnt = (unsigned)((tid < 4 ? t_lo >> (16 * tid) : t_hi >> (16 * (tid - 4))) & 0xFFFFull);
    unsigned *const hot = ((PermAll *)st)->hot;  // (tid == BR_CLASSES: the long rows' slot, PH_LONG)
    s_base[tid] = cnt ? atomicAdd(&hot[tid * PH_STRIDE], cnt) : 0u;
  }
  if (tid == 32) {
    const unsigned long long all = s_nz[0][0] + s_nz[0][1] + s_nz[0][2] + s_nz[0][3];
    if (all) atomicAdd((unsigned long long *)&((PermAll *)st)->hot[PH_LONG_NNZ * PH_STRIDE], all);
  }
  if (tid == 33) {
    const unsigned long long all = s_nz[1][0] + s_nz[1][1] + s_nz[1][2] + s_nz[1][3];
    if (all) atomicAdd((unsigned long long *)&((PermAll *)st)->hot[PH_BLOCK_NNZ * PH_STRIDE], all);
  }
  __syncthreads();
  if (c_lo | c_hi) {
#pragma unroll
    for (int k = 0; k < CS_ITEMS; k++) {
      const int dd = d[k];
      if (base + k < nr && dd > PT_LMAX) {
        int cls = BR_CLASSES;
        if (dd <= block_cap) {
          cls = 0;
          while (dd > br_cap(cls)) cls++;
        }
        unsigned at;
        if (cls < 4) {
          at = (unsigned)((e_lo >> (16 * cls)) & 0xFFFFull);
          e_lo += 1ull << (16 * cls);
        } else {
          at = (unsigned)((e_hi >> (16 * (cls - 4))) & 0xFFFFull);
          e_hi += 1ull << (16 * (cls - 4));
        }
        I *list = cls < BR_CLASSES ? block_rows + (int64_t)cls * block_stride : long_rows;
        list[s_base[cls] + at] = (I)(base + k);
      }
    }
  }
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
  // four consecutive old rows per thread: their bounds and new indices are loaded together (16-byte loads where the
  // arrays allow), then the four scatters are in flight together — one row per thread and round trip left the kernel
  // waiting for its own loads (65 us for 4 M rows; the traffic is 80 MB)
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, nq = (n + 3) >> 2;
  const bool al = sizeof(I) == 4 && (((uintptr_t)rp | (uintptr_t)row_order) & 15) == 0;  // (64-bit indices: the scalar form)
  for (; q < nq; q += stride) {
    const int64_t u0 = q << 2;
    int64_t b[5], r[4];
    if (u0 + 4 <= n && al) {
      const int4 p4 = *(const int4 *)(const void *)(rp + u0);
      b[0] = p4.x, b[1] = p4.y, b[2] = p4.z, b[3] = p4.w, b[4] = rp[u0 + 4];
      if (row_order) {
        const int4 o4 = *(const int4 *)(const void *)(row_order + u0);
        r[0] = o4.x, r[1] = o4.y, r[2] = o4.z, r[3] = o4.w;
      } else {
        r[0] = u0, r[1] = u0 + 1, r[2] = u0 + 2, r[3] = u0 + 3;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 5; k++) b[k] = rp[u0 + k <= n ? u0 + k : n];
#pragma unroll
      for (int k = 0; k < 4; k++) r[k] = u0 + k < n ? (row_order ? (int64_t)row_order[u0 + k] : u0 + k) : -1;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int64_t rr = r[k] - rb0;
      if (u0 + k < n && rr >= 0 && rr < nr && b[k + 1] > b[k])
        rec[rr] = make_int2((int)(b[k + 1] - b[k]), (int)b[k]);  // (rec is zeroed: empty rows — half of a power-law graph's — cost no store)
    }
  }
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
  __shared__ I s_col[PC_TILE];
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
    int open = sbx_wave_shift_up1(inc, 0);
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
  // all loads of a thread first, then its stores.  Written as "load, store, next position" under `p < cnt`, the
  // compiler waited for every load before the store behind it and the copy was 16 dependent round trips per tile
  // (tools/isa_waits.py); positions past the tile's end read the entry of position 0 (the loads are unconditional).
  I c[PC_ITEMS];
  V v[VB ? PC_ITEMS : 1];
  {
    int64_t src[PC_ITEMS];
    const int64_t src0 = t0 + s_delta[0];
#pragma unroll
    for (int k = 0; k < PC_ITEMS; k++) {
      const int p = k * PC_THREADS + tid;
      src[k] = p < cnt ? t0 + p + s_delta[p < cnt ? p : 0] : src0;
    }
#pragma unroll
    for (int k = 0; k < PC_ITEMS; k++) c[k] = col_in[src[k]];
    if (VB) {
#pragma unroll
      for (int k = 0; k < PC_ITEMS; k++) v[k] = ((const V *)val_in)[src[k]];
    }
  }
#pragma unroll
  for (int k = 0; k < PC_ITEMS; k++) {
    const int p = k * PC_THREADS + tid;
    if (p < cnt) {
      col_out[t0 + p] = c[k];
      if (VB) ((V *)val_out)[t0 + p] = v[k];
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

// ---- shared pieces of the two sorting kernels -----------------------------------
__device__ __forceinline__ int bits_u32(unsigned x) { return x ? 32 - __clz((int)x) : 0; }  // bits needed for 0..x

// Stable LSD radix sort of `count` (<= THREADS * ITEMS) LDS records by (krow, kcol) — column bits first, then
// the row bits — with ballot match-any ranking and per-wave digit counters.  The distribution-independent
// path behind the bucket-rank sort (clustered rows), and the round-1 sort of this file.  Padding records
// must carry 0x7FFFFFFF keys.  `whist` needs (THREADS / 64) * 256 words.  All threads call it.
template <typename V, bool HASV, bool HASROW, int THREADS, int ITEMS>
__device__ void lds_radix_sort(int *s_col, int *s_row, V *s_val, unsigned *whist_base, unsigned *s_scan, int count,
                               int col_bits, int row_bits) {
  constexpr int WAVES = THREADS / 64;
  constexpr int CAP = THREADS * ITEMS;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  volatile unsigned *wh = whist_base + wv * 256;
  const uint64_t lt = sbx_lanemask_lt();
  const int total_bits[2] = {col_bits, HASROW ? row_bits : 0};
  for (int part = 0; part < (HASROW ? 2 : 1); part++) {
    int done = 0;
    while (done < total_bits[part]) {
      const int remaining = total_bits[part] - done;
      const int passes_left = (remaining + 7) >> 3;
      const int bits = (remaining + passes_left - 1) / passes_left;
      const unsigned mask = (1u << bits) - 1u;
      const int shift = done;
      for (int i = tid; i < 256 * WAVES; i += THREADS) whist_base[i] = 0;
      int kc[ITEMS], kr[HASROW ? ITEMS : 1];
      V kv[HASV ? ITEMS : 1];
      unsigned rank[ITEMS];
#pragma unroll
      for (int i = 0; i < ITEMS; i++) {
        const int e = wv * 64 * ITEMS + i * 64 + lane;
        const bool ok = e < count;
        kc[i] = ok ? s_col[e] : 0x7FFFFFFF;
        if (HASROW) kr[i] = ok ? s_row[e] : 0x7FFFFFFF;
        if (HASV) kv[i] = ok ? s_val[e] : (V)0;
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < ITEMS; i++) {
        const unsigned d = ((unsigned)((HASROW && part) ? kr[i] : kc[i]) >> shift) & mask;
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
        // digit totals over the waves -> exclusive scan over the digits -> per-(wave, digit) bases; a thread owns
        // DPT consecutive digits (1 when the workgroup has 256 threads or more)
        constexpr int DPT = THREADS >= 256 ? 1 : 256 / THREADS;
        unsigned c4[DPT][WAVES];
        unsigned tot = 0;
        if (tid * DPT < 256) {
#pragma unroll
          for (int j = 0; j < DPT; j++)
#pragma unroll
            for (int i = 0; i < WAVES; i++) {
              c4[j][i] = whist_base[i * 256 + tid * DPT + j];
              tot += c4[j][i];
            }
        }
        unsigned all;
        unsigned ex = sbx_block_exclusive_sum<unsigned, THREADS>(tot, s_scan, &all);
        if (tid * DPT < 256) {
#pragma unroll
          for (int j = 0; j < DPT; j++)
#pragma unroll
            for (int i = 0; i < WAVES; i++) {
              whist_base[i * 256 + tid * DPT + j] = ex;
              ex += c4[j][i];
            }
        }
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < ITEMS; i++) {
        const unsigned d = ((unsigned)((HASROW && part) ? kr[i] : kc[i]) >> shift) & mask;
        const unsigned pos = whist_base[wv * 256 + d] + rank[i];
        if (pos < (unsigned)CAP) {
          s_col[pos] = kc[i];
          if (HASROW) s_row[pos] = kr[i];
          if (HASV) s_val[pos] = kv[i];
        }
      }
      __syncthreads();
      done += bits;
    }
  }
}

// In-place inclusive scan of the bucket counters c[0, nc) (thread t owns CPT consecutive words, moved 16 bytes at a
// time; the array must extend to THREADS * CPT words); returns this thread's largest count.  Two barriers inside;
// the caller adds the one that publishes the result.
template <int THREADS, int CPT>
__device__ __forceinline__ unsigned scan_bucket_counts(unsigned *c, int nc, unsigned *s_scan) {
  static_assert(CPT % 4 == 0, "16-byte LDS accesses");
  const int base = (int)threadIdx.x * CPT;
  unsigned v[CPT];
  unsigned sum = 0, mx = 0;
  if (base < nc) {
#pragma unroll
    for (int i = 0; i < CPT; i += 4) {
      const uint4 q = *(const uint4 *)(c + base + i);
      v[i] = q.x, v[i + 1] = q.y, v[i + 2] = q.z, v[i + 3] = q.w;
    }
#pragma unroll
    for (int i = 0; i < CPT; i++) {
      const unsigned y = base + i < nc ? v[i] : 0u;  // words past nc are stale
      mx = y > mx ? y : mx;
      sum += y;
      v[i] = sum;
    }
  }
  unsigned tot;
  const unsigned ex = sbx_block_exclusive_sum<unsigned, THREADS>(sum, s_scan, &tot);
  if (base < nc) {
#pragma unroll
    for (int i = 0; i < CPT; i += 4)
      *(uint4 *)(c + base + i) = make_uint4(v[i] + ex, v[i + 1] + ex, v[i + 2] + ex, v[i + 3] + ex);
  }
  return mx;
}

// diagnostic only (SBX_DEBUG_TILE_STOP=9): wall-clock cycles from kernel entry to the end of each phase, summed over
// the tiles by thread 0; [31] counts the tiles.  Printed and cleared by sort_stage.
__device__ unsigned long long g_tile_stamps[32];
#define TILE_STAMP(i)                                                                    \
  do {                                                                                   \
    if (dbg_stop == 9 && tid == 0) {                                                     \
      unsigned long long t_;                                                             \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");         \
      atomicAdd(&g_tile_stamps[i], t_ - t_start);                                        \
    }                                                                                    \
  } while (0)

// ---- the tile kernel ----------------------------------------------------------
// col_order == nullptr: csr_sort_rows mode (no relabel).  tile_first[t] = first row whose output range starts at
// or after position t * PT_W (k_tile_first).  Entries live in registers (position p = k * THREADS + tid); LDS is one
// pool of 16 bytes per entry whose regions change hands between the phases (a row longer than PT_LMAX that ends
// inside the window keeps its positions in the tile as a hole: counted, never gathered or written):
//   r0  row-head marks -> (source offset - position) -> gathered columns -> placed words -> sorted columns
//   r1  first position of the entry's row | row length << 16            (radix path: dense row rank)
//   c   per-head (source offset, length) -> per-row (min, max) -> bucket counters -> sorted values
// What a tile needs before it can touch its entries, three dependent rounds of loads deep: its row range
// (tile_first), the range's bounds in the short-row entry space (sp) and, per lane, the record of the lane's first row.
// k_permute_tile keeps these in flight for the tiles behind the one it works on.
template <typename I>
struct TileIn {
  I ra, rb;      // rows [ra, rb)
  I e0, e1;      // sp[ra], sp[rb]
  int2 rc;       // rec[ra + lane] (length, source offset)
  I sp_r, rpo_r; // sp / rpo of that row
};
// The tile kernels are one wave per workgroup: LDS operations of a wave execute in order, so the points where the
// phases hand LDS regions over need no barrier — and must not wait for vector memory (__syncthreads() would: the
// loads of the next tiles are in flight across them).
#define TILE_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

template <typename I, int VB, bool RADIX>
__device__ __forceinline__ void permute_tile_body(
    const int64_t tile, const TileIn<I> in, const int2 *__restrict__ rec, const I *col_in, const char *val_in,
    const I *__restrict__ col_order, const I *__restrict__ rpo, const I *__restrict__ sp,
    I *col_out, char *val_out, int64_t nr, PermState *__restrict__ st, int col_bits,
    int force_radix, unsigned *__restrict__ fb_tiles) {
  typedef typename ValT<VB>::type V;
  constexpr bool HASV = VB != 0;
  constexpr int THREADS = PT_THREADS, ITEMS = PT_ITEMS, CAP = PT_CAP, WAVES = THREADS / 64;
  static_assert(sizeof(V) * CAP <= sizeof(unsigned) * 2 * CAP, "the sorted values are staged in the counter region");
  __shared__ __attribute__((aligned(16))) unsigned s_pool[4 * CAP + 4];
  __shared__ unsigned s_whist[RADIX ? WAVES * 256 : 1];
  __shared__ unsigned s_scan[WAVES + 1];
  __shared__ int s_wmax[WAVES];
  __shared__ int s_ob[CAP];  // at a row's first position: where the row starts in the output
  __shared__ int s_flag[4];  // [0] some row of the tile is out of order, [1] radix path, [2] refine, [3] overfull level-0 bucket
  int *const s_a = (int *)s_pool;  // r0 under its successive names
  int *const s_key = (int *)s_pool;
  unsigned *const s_hl = s_pool + CAP;
  unsigned *const s_c = s_pool + 2 * CAP;
  V *const s_val = (V *)(s_pool + 2 * CAP);
  // output index of entry k (position p): the row's start in the output + the entry's position inside the row
#define OUTPOS(k, p) ((int64_t)s_ob[hl[k] & 0xFFFFu] + ((p) - (int)(hl[k] & 0xFFFFu)))
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int dbg_stop = force_radix >> 8;  // timing ablation only (SBX_DEBUG_TILE_STOP): leave after a phase, output junk
  force_radix &= 0xFF;
  unsigned long long t_start = 0;
  if (dbg_stop == 9) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_start)::"memory");

  // the tile's rows: those of at most PT_LMAX entries whose range in the SHORT-row entry space (prefix sums `sp`
  // over the lengths of such rows only) starts in the window; longer rows between them belong to other kernels
  const int64_t ra = in.ra;
  const int64_t rb = in.rb;
  if (ra >= rb) return;
  const int64_t e0 = in.e0;
  const int cnt = (int)((int64_t)in.e1 - e0);
  if (cnt == 0) return;
  TILE_STAMP(0);
  if (dbg_stop == 9 && tid == 0) {
    atomicAdd(&g_tile_stamps[31], 1ull);
    atomicAdd(&g_tile_stamps[30], (unsigned long long)cnt);
  }

  // ---- row map.  Non-empty rows mark their first position and leave (source offset, length) there; a max-scan
  // of head positions then gives every entry its row.  (Windows dominated by empty rows — the isolated vertices
  // RCM packs at the end — would walk millions of rows: they search each position instead.)
#pragma unroll
  for (int k = 0; k < ITEMS; k += 4) *(int4 *)&s_a[k * THREADS + 4 * tid] = make_int4(0, 0, 0, 0);
  if (tid < 4) s_flag[tid] = 0;
  TILE_SYNC();
  TILE_STAMP(1);
  if (rb - ra <= 4 * CAP) {
    int64_t r = ra + tid;
    if (r < rb) {  // the first row of every lane arrived with the tile's bounds (TileIn)
      const int2 rc = in.rc;
      if (rc.x > 0 && rc.x <= PT_LMAX) {
        const int p = (int)((int64_t)in.sp_r - e0);
        s_a[p] = 1;
        s_ob[p] = (int)in.rpo_r;
        *(uint2 *)&s_c[2 * p] = make_uint2((unsigned)rc.y, (unsigned)rc.x);
      }
      r += THREADS;
    }
    for (; r < rb; r += THREADS) {
      const int2 rc = rec[r];
      if (rc.x > 0 && rc.x <= PT_LMAX) {
        const int p = (int)((int64_t)sp[r] - e0);
        s_a[p] = 1;
        s_ob[p] = (int)rpo[r];
        *(uint2 *)&s_c[2 * p] = make_uint2((unsigned)rc.y, (unsigned)rc.x);
      }
    }
  } else {
    for (int p = tid; p < cnt; p += THREADS) {
      const int64_t target = e0 + p;  // last row r in [ra, rb) with rpo[r] <= target: the row that owns the position
      int64_t lo = ra, hi = rb;
      while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)sp[mid] <= target) lo = mid;
        else hi = mid;
      }
      if ((int64_t)sp[lo] == target) {  // (rows that add nothing to sp share the value of the tile row behind them)
        const int2 rc = rec[lo];
        s_a[p] = 1;
        s_ob[p] = (int)rpo[lo];
        *(uint2 *)&s_c[2 * p] = make_uint2((unsigned)rc.y, (unsigned)rc.x);
      }
    }
  }
  TILE_SYNC();
  TILE_STAMP(2);
  {
    const int p0 = tid * ITEMS;
    int hd[ITEMS];
    int last = 0;
#pragma unroll
    for (int k = 0; k < ITEMS; k += 4) {
      const int4 q = *(const int4 *)&s_a[p0 + k];
      hd[k] = q.x, hd[k + 1] = q.y, hd[k + 2] = q.z, hd[k + 3] = q.w;
    }
#pragma unroll
    for (int k = 0; k < ITEMS; k++)
      if (hd[k]) last = p0 + k + 1;
    const int inc = sbx_wave_inclusive_max(last);
    int open = sbx_wave_shift_up1(inc, 0);
    if (lane == 63) s_wmax[wv] = inc;
    TILE_SYNC();
    for (int w = 0; w < wv; w++) open = s_wmax[w] > open ? s_wmax[w] : open;
    int hp = open ? open - 1 : 0;  // position 0 is a head
    uint2 sl = *(const uint2 *)&s_c[2 * hp];  // (source offset, length) of the row open at this thread's first entry
    unsigned ohl[ITEMS];
    int oa[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      const int p = p0 + k;
      if (hd[k]) {
        hp = p;
        sl = *(const uint2 *)&s_c[2 * p];
      }
      ohl[k] = (unsigned)hp | (sl.y << 16);  // entries past cnt inherit the last row: never used
      oa[k] = (int)sl.x - hp;                // source index of the entry = this + p
    }
#pragma unroll
    for (int k = 0; k < ITEMS; k += 4) {
      *(uint4 *)&s_hl[p0 + k] = make_uint4(ohl[k], ohl[k + 1], ohl[k + 2], ohl[k + 3]);
      *(int4 *)&s_a[p0 + k] = make_int4(oa[k], oa[k + 1], oa[k + 2], oa[k + 3]);
    }
  }
  TILE_SYNC();
  TILE_STAMP(3);
  if (dbg_stop == 1) {
    for (int p = tid; p < cnt; p += THREADS) col_out[(int64_t)s_ob[s_hl[p] & 0xFFFFu] + (p - (int)(s_hl[p] & 0xFFFFu))] = (I)(s_a[p] + (int)s_hl[p]);
    return;
  }

  // ---- gather: whole old rows, columns relabelled (permute_order_two.cc:63-74); entries stay in registers
  int kc[ITEMS];
  V kv[HASV ? ITEMS : 1];
  unsigned hl[ITEMS];
  unsigned live = 0;  // bit k: the tile holds an entry at position k * THREADS + tid
  {
    I c[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      const int p = k * THREADS + tid;
      c[k] = 0;
      hl[k] = 0;
      if (HASV) kv[k] = (V)0;
      if (p < cnt) {
        hl[k] = s_hl[p];
        live |= 1u << k;
        const int64_t s = (int64_t)s_a[p] + p;
        c[k] = __builtin_nontemporal_load(col_in + s);
        if (HASV) kv[k] = __builtin_nontemporal_load((const V *)val_in + s);
      }
    }
    if (dbg_stop == 9) {
      int acc = 0;
#pragma unroll
      for (int k = 0; k < ITEMS; k++) acc += (int)c[k];
      if (acc == 0x12345678) s_flag[1] = 1;  // forces the wait for the column loads in front of the stamp
      TILE_STAMP(4);
    }
    // the relabel gathers, all ITEMS of them in flight together: issued for every position (a dead one gathers entry 0
    // of the table) — under `if (live...)` each gather sat in a block of its own with the LDS store of its result,
    // and the compiler's wait at every join made them eight dependent round trips (37 % of a tile's life)
    // (one opaque use of all the columns: the compiler waits for their loads HERE, once — they were issued under
    // `p < cnt`, and without this it waits before every gather for all but the latest outstanding operation, i.e. for
    // the gather before the last one)
    static_assert(ITEMS == 8 || ITEMS == 4, "operand list below");
    if constexpr (ITEMS == 8) asm volatile("" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4 % ITEMS]), "+v"(c[5 % ITEMS]), "+v"(c[6 % ITEMS]), "+v"(c[7 % ITEMS]));
    else asm volatile("" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]));
#pragma unroll
    for (int k = 0; k < ITEMS; k++) kc[k] = (int)((col_order && !(force_radix & 4)) ? col_order[c[k]] : c[k]);  // (bit 2: timing ablation without the gathers)
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      const int p = k * THREADS + tid;
      if (live >> k & 1) s_key[p] = kc[k];  // the word this thread read its source offset from
      else kc[k] = 0x7FFFFFFF;
    }
  }
  TILE_SYNC();
  TILE_STAMP(5);
  if (dbg_stop == 2) {
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      const int p = k * THREADS + tid;
      if (live >> k & 1) {
        col_out[OUTPOS(k, p)] = (I)kc[k];
        if (HASV) ((V *)val_out)[OUTPOS(k, p)] = kv[k];
      }
    }
    return;
  }
  {
    bool unsorted = false;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      const int p = k * THREADS + tid;
      if (live >> k & 1) {
        const int hp = (int)(hl[k] & 0xFFFFu);
        if (p > hp && kc[k] < s_key[p - 1]) unsorted = true;
        if (p == hp)  // the row's (min, max) accumulators, where its (source, length) record was
          *(uint2 *)&s_c[2 * p] = make_uint2(0xFFFFFFFFu, 0u);
      }
    }
    if (__any(unsorted) && lane == 0 && !(force_radix & 2)) {  // (bit 1: timing ablation, rows stream out unsorted)
      st->any_unsorted = 1;
      s_flag[0] = 1;
    }
  }
  TILE_SYNC();
  TILE_STAMP(6);
  if (s_flag[0] == 0) {
    // every row of the tile is already in column order (identity column maps, orders that preserve
    // locality): a stable sort would not move anything, so stream the gathered rows out as they are
    bool dup = false;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      const int p = k * THREADS + tid;
      if (live >> k & 1) {
        if (p > (int)(hl[k] & 0xFFFFu) && kc[k] == s_key[p - 1]) dup = true;
        col_out[OUTPOS(k, p)] = (I)kc[k];
        if (HASV) ((V *)val_out)[OUTPOS(k, p)] = kv[k];
      }
    }
    if (__any(dup) && lane == 0) st->any_dup = 1;
    return;
  }

  // ---- bucket-rank sort.  (min, max) per row: every thread reduces the runs of its own consecutive entries and
  // adds one LDS atomic pair per run
  {
    const int p0 = tid * ITEMS;
    unsigned cur = 0xFFFFFFFFu, mn = 0, mx = 0;
    unsigned key[ITEMS], hpv[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; k += 4) {
      const int4 q = *(const int4 *)&s_key[p0 + k];
      const uint4 g = *(const uint4 *)&s_hl[p0 + k];
      key[k] = (unsigned)q.x, key[k + 1] = (unsigned)q.y, key[k + 2] = (unsigned)q.z, key[k + 3] = (unsigned)q.w;
      hpv[k] = g.x, hpv[k + 1] = g.y, hpv[k + 2] = g.z, hpv[k + 3] = g.w;
    }
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      hpv[k] &= 0xFFFFu;
      if (p0 + k < cnt) {
        if (hpv[k] != cur) {
          if (cur != 0xFFFFFFFFu) {
            atomicMin(&s_c[2 * cur], mn);
            atomicMax(&s_c[2 * cur + 1], mx);
          }
          cur = hpv[k];
          mn = mx = key[k];
        } else {
          mn = key[k] < mn ? key[k] : mn;
          mx = key[k] > mx ? key[k] : mx;
        }
      }
    }
    if (cur != 0xFFFFFFFFu) {
      atomicMin(&s_c[2 * cur], mn);
      atomicMax(&s_c[2 * cur + 1], mx);
    }
  }
  TILE_SYNC();
  TILE_STAMP(7);
  unsigned bk[ITEMS], wd[ITEMS];
  unsigned char sh[ITEMS];
  {
    bool bad = false;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      const int p = k * THREADS + tid;
      bk[k] = 0;
      wd[k] = 0;
      sh[k] = 0;
      if (live >> k & 1) {
        const unsigned hp = hl[k] & 0xFFFFu, len = hl[k] >> 16;
        const uint2 mm = *(const uint2 *)&s_c[2 * hp];
        const unsigned mn = mm.x, mx = mm.y;
        const int rbits = bits_u32(mx - mn);
        const int ib = bits_u32(len - 1);                 // bits of the position inside the row
        const int lg = len > (unsigned)BK_SHORT ? ib : 0; // 2^lg buckets, 2^lg < 2 * len
        const int shift = rbits > lg ? rbits - lg : 0;
        const unsigned rel = (unsigned)kc[k] - mn;
        const unsigned bo = shift >= 32 ? 0u : rel >> shift;
        const unsigned low = shift >= 32 ? rel : rel & ((1u << shift) - 1u);
        if (shift + ib > 32) bad = true;  // word does not fit (short rows spread over > 2^27 columns)
        wd[k] = (low << ib) | ((unsigned)p - hp);
        bk[k] = 2 * hp + bo;
        sh[k] = (unsigned char)(shift >= 32 ? 32 : shift);
      }
    }
    if ((__any(bad) || (force_radix & 1) || RADIX) && lane == 0) s_flag[1] = 1;
  }
  TILE_SYNC();  // the (min, max) words and the column copies in r0 have been read
  TILE_STAMP(8);
  if (dbg_stop == 3) {
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      const int p = k * THREADS + tid;
      if (live >> k & 1) {
        col_out[OUTPOS(k, p)] = (I)(bk[k] + wd[k]);
        if (HASV) ((V *)val_out)[OUTPOS(k, p)] = kv[k];
      }
    }
    return;
  }
  for (int i = 4 * tid; i < 2 * cnt; i += 4 * THREADS) *(uint4 *)&s_c[i] = make_uint4(0, 0, 0, 0);
  TILE_SYNC();
  TILE_STAMP(9);
#pragma unroll
  for (int k = 0; k < ITEMS; k++)
    if (live >> k & 1) atomicAdd(&s_c[bk[k]], 1u);
  TILE_SYNC();
  TILE_STAMP(10);
  {
    const unsigned mxc = scan_bucket_counts<THREADS, 2 * ITEMS>(s_c, 2 * cnt, s_scan);
    if (__any(mxc > (unsigned)BK_REFINE) && lane == 0) s_flag[2] = 1;
    if (__any(mxc > (unsigned)BK_MAX) && lane == 0) s_flag[3] = 1;
  }
  TILE_SYNC();
  TILE_STAMP(11);
  if (s_flag[2] && !s_flag[1]) {
    // level 1 (clustered columns): every bucket is split into as many sub-buckets as it holds entries, by interpolation
    // inside the bucket; the sub-buckets of the row that starts at hp take the counter words from 2 hp on
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      if (live >> k & 1) {
        const unsigned b = bk[k], hp = hl[k] & 0xFFFFu;
        const unsigned start = b ? s_c[b - 1] : 0u, cntb = s_c[b] - start;
        const unsigned ib = bits_u32((hl[k] >> 16) - 1);
        const unsigned low = wd[k] >> ib;
        const unsigned sub = (sh[k] && sh[k] < 32) ? (unsigned)(((unsigned long long)low * cntb) >> sh[k]) : 0u;
        bk[k] = hp + start + sub;  // = 2 hp + (start - hp) + sub, below 2 hp + length
      }
    }
    TILE_SYNC();  // the level-0 bounds have been read
    for (int i = 4 * tid; i < 2 * cnt; i += 4 * THREADS) *(uint4 *)&s_c[i] = make_uint4(0, 0, 0, 0);
    TILE_SYNC();
#pragma unroll
    for (int k = 0; k < ITEMS; k++)
      if (live >> k & 1) atomicAdd(&s_c[bk[k]], 1u);
    TILE_SYNC();
    const unsigned mxc = scan_bucket_counts<THREADS, 2 * ITEMS>(s_c, 2 * cnt, s_scan);
    if (__any(mxc > (unsigned)BK_MAX) && lane == 0) s_flag[1] = 1;
    TILE_SYNC();
  } else if (s_flag[3]) {
    TILE_SYNC();
    if (tid == 0) s_flag[1] = 1;
    TILE_SYNC();
  }
  if (dbg_stop == 4) {
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      const int p = k * THREADS + tid;
      if (live >> k & 1) {
        col_out[OUTPOS(k, p)] = (I)(s_c[bk[k]] + wd[k]);
        if (HASV) ((V *)val_out)[OUTPOS(k, p)] = kv[k];
      }
    }
    return;
  }
  if (s_flag[1] == 0) {
    // placement fills every bucket from its end: afterwards s_c[b] is the bucket's first slot and s_c[b + 1] its end
#pragma unroll
    for (int k = 0; k < ITEMS; k++)
      if (live >> k & 1) s_a[atomicSub(&s_c[bk[k]], 1u) - 1u] = (int)wd[k];
    TILE_SYNC();
    TILE_STAMP(12);
    int fin[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      fin[k] = 0;
      if (live >> k & 1) {
        const unsigned b0 = s_c[bk[k]], b1 = s_c[bk[k] + 1];
        unsigned r = 0;
        for (unsigned j = b0; j < b1; j++) r += (unsigned)s_a[j] < wd[k];
        fin[k] = (int)(b0 + r);
      }
    }
    if (dbg_stop == 5) {
#pragma unroll
      for (int k = 0; k < ITEMS; k++) {
        const int p = k * THREADS + tid;
        if (live >> k & 1) {
          col_out[OUTPOS(k, p)] = (I)(fin[k] + kc[k]);
          if (HASV) ((V *)val_out)[OUTPOS(k, p)] = kv[k];
        }
      }
      return;
    }
    TILE_SYNC();  // the placed words (r0) and the bucket bounds (c) are dead: the sorted entries move in
    TILE_STAMP(13);
#pragma unroll
    for (int k = 0; k < ITEMS; k++)
      if (live >> k & 1) {
        s_key[fin[k]] = kc[k];
        if (HASV) s_val[fin[k]] = kv[k];
      }
    TILE_SYNC();
    TILE_STAMP(14);
  } else if constexpr (!RADIX) {
    // the columns of some row cluster: the tile goes on the list of the radix kernel (same kernel body, RADIX = true)
    if (tid == 0) fb_tiles[atomicAdd(&st->n_fb_tiles, 1u)] = (unsigned)tile;
    return;
  } else {
    // radix path: composite key (dense rank of the row among the rows present, column); r0 still holds the columns
    int flag[ITEMS];
    int local = 0;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      const int p = tid * ITEMS + k;
      const bool head = p < cnt && (s_hl[p] & 0xFFFFu) == (unsigned)p;
      local += head;
      flag[k] = local;
    }
    int all;
    const int ex = sbx_block_exclusive_sum<int, THREADS>(local, (int *)s_scan, &all);  // its barriers also retire the reads above
    int *const s_rank = (int *)s_hl;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      const int p = tid * ITEMS + k;
      if (p < cnt) s_rank[p] = ex + flag[k] - 1;
    }
    if (HASV) {
#pragma unroll
      for (int k = 0; k < ITEMS; k++)
        if (k * THREADS + tid < cnt) s_val[k * THREADS + tid] = kv[k];
    }
    TILE_SYNC();
    int row_bits = 0;
    for (int t = all - 1; t > 0; t >>= 1) row_bits++;
    lds_radix_sort<V, HASV, true, THREADS, ITEMS>(s_key, s_rank, s_val, s_whist, s_scan, cnt, col_bits, row_bits);
  }
  {
    bool dup = false;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      const int p = k * THREADS + tid;
      if (live >> k & 1) {
        const int c = s_key[p];
        if (p > (int)(hl[k] & 0xFFFFu) && c == s_key[p - 1]) dup = true;  // rows keep their position ranges
        col_out[OUTPOS(k, p)] = (I)c;
        if (HASV) ((V *)val_out)[OUTPOS(k, p)] = s_val[p];
      }
    }
    if (__any(dup) && lane == 0) st->any_dup = 1;
  }
  if (dbg_stop == 9) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TILE_STAMP(15);
  }
}
#undef OUTPOS

#undef TILE_SYNC

template <typename I>
__device__ __forceinline__ TileIn<I> tile_in_now(int64_t tile, const int2 *__restrict__ rec, const I *__restrict__ rpo,
                                                const I *__restrict__ sp, const I *__restrict__ tile_first, int64_t nr) {
  TileIn<I> in;
  in.ra = tile_first[tile];
  in.rb = tile_first[tile + 1];
  in.e0 = sp[in.ra];
  in.e1 = sp[in.rb];
  const int64_t r = (int64_t)in.ra + threadIdx.x < nr ? (int64_t)in.ra + threadIdx.x : nr - 1;
  in.rc = rec[r];
  in.sp_r = sp[r];
  in.rpo_r = rpo[r];
  return in;
}

// Persistent waves: a wave walks the tiles blockIdx.x, blockIdx.x + gridDim.x, ... and keeps the inputs of the tiles
// behind the current one in flight — the row range of the tile after next, the bounds and the lanes' first row records
// of the next one — so that a tile starts from registers instead of three dependent rounds of loads (18 % of a tile's
// life, tools/tile_stamps.py).  Every load is issued unconditionally with a clamped index.
template <typename I, int VB>
__global__ __launch_bounds__(PT_THREADS, VB == 0 ? 1 : PT_MIN_WAVES) void k_permute_tile(
    const int2 *__restrict__ rec, const I *col_in, const char *val_in, const I *__restrict__ col_order,
    const I *__restrict__ rpo, const I *__restrict__ sp, const I *__restrict__ tile_first, I *col_out, char *val_out,
    int64_t nr, PermState *__restrict__ st, int col_bits, int force_radix, unsigned *__restrict__ fb_tiles,
    int64_t tiles) {
  const int64_t G = gridDim.x, last = tiles - 1;
  int64_t t = blockIdx.x;
  auto clampt = [&](int64_t x) { return x < last ? x : last; };
  auto clampr = [&](int64_t x) { return x < nr - 1 ? x : nr - 1; };
  TileIn<I> cur = tile_in_now<I>(clampt(t), rec, rpo, sp, tile_first, nr);
  I ra1 = tile_first[clampt(t + G)], rb1 = tile_first[clampt(t + G) + 1];  // row range of the next tile
  for (; t < tiles; t += G) {
    // issued now, read after the current tile: range of the tile after next, bounds and row records of the next one
    const I ra2 = tile_first[clampt(t + 2 * G)], rb2 = tile_first[clampt(t + 2 * G) + 1];
    TileIn<I> nx;
    nx.ra = ra1, nx.rb = rb1;
    nx.e0 = sp[ra1];
    nx.e1 = sp[rb1];
    const int64_t r1 = clampr((int64_t)ra1 + threadIdx.x);
    nx.rc = rec[r1];
    nx.sp_r = sp[r1];
    nx.rpo_r = rpo[r1];
    permute_tile_body<I, VB, false>(t, cur, rec, col_in, val_in, col_order, rpo, sp, col_out, val_out, nr, st, col_bits,
                                    force_radix, fb_tiles);
    cur = nx;
    ra1 = ra2, rb1 = rb2;
  }
}

// the tiles the kernel above listed (clustered columns): same body, LSD radix sort instead of the bucket-rank pass
template <typename I, int VB>
__global__ __launch_bounds__(PT_THREADS) void k_permute_tile_radix(
    const int2 *__restrict__ rec, const I *col_in, const char *val_in, const I *__restrict__ col_order,
    const I *__restrict__ rpo, const I *__restrict__ sp, const I *__restrict__ tile_first, I *col_out, char *val_out,
    int64_t nr, PermState *__restrict__ st, int col_bits, const unsigned *__restrict__ fb_tiles) {
  const unsigned n = st->n_fb_tiles;
  for (unsigned i = blockIdx.x; i < n; i += gridDim.x) {
    const int64_t tile = (int64_t)fb_tiles[i];
    permute_tile_body<I, VB, true>(tile, tile_in_now<I>(tile, rec, rpo, sp, tile_first, nr), rec, col_in, val_in,
                                   col_order, rpo, sp, col_out, val_out, nr, st, col_bits, 0, nullptr);
    __syncthreads();  // the next tile reuses the LDS pool
  }
}

// ---- the tile kernel of the permutes that relabel (round 6) ---------------------------------------------------------
// What the kernel above spends its instructions on is not the sort alone: with sort and gathers switched off it streams
// its 458 MB at 1.85 TB/s (the row classes: 4) — the row map (LDS scan, two more LDS round trips), 64-bit address
// arithmetic for every load and store, a predicate around every per-entry statement, the rows' min / max.  This kernel
// keeps the formulation (a wave per tile of up to 512 entries, 8 per lane at position k * 64 + lane, bucket-rank sort)
// and removes those:
//   * row map from BALLOTS: a row marks its first position in a 512-bit mask (one LDS atomic) and leaves ONE 16-byte
//     record there (source - position, output - position, length, log2 buckets); the mask's 16 words become scalars, an
//     entry finds its row's head with a count-leading-zeros on the lanes below it, and reads that one record;
//   * loads, gathers and stores through BUFFER descriptors with 32-bit offsets (the caller checks that the arrays fit
//     4 GB): one multiply per address, and a dead position is an out-of-range offset — loads return 0, stores are
//     dropped — so nothing in the kernel carries a predicate except the counter index of the dead positions;
//   * the neighbour key for the order and duplicate checks comes through DPP, not LDS;
//   * buckets from the key-distribution map F (below: k_cdf_sample) instead of the row's min / max: no min / max pass,
//     and no second level — under an ordering like RCM's 61 % of the entries shared an equal-width bucket with more
//     than four others and every tile took the interpolating level; with F 1 % do, and a fuller bucket is ranked by a
//     loop (a bucket holds at most a row: 128);
//   * the count pass returns the arrival number, placement = bucket start + arrival (no second atomic sweep); four
//     neighbour words ranked unrolled; sorted (column, value) pairs through LDS as 8-byte accesses.
// Words are key << 7 | arrival (distinct; duplicates fall in arrival order and are ordered by value afterwards,
// k_fix_dup_runs), so the ids must fit CDF_MAX_COL_BITS bits.  Anything else — wider ids, arrays beyond 4 GB, callers
// that sort without relabelling (they need stable sorts) — keeps the kernel above.
//
// The key-distribution map.  Relabelled columns are not spread evenly: a column turns up in proportion to its degree,
// and an ordering that means something packs the heavy columns together (RCM: the hubs of a power-law graph sit in a
// few BFS levels).  What all rows have in common is the GLOBAL distribution of the new ids, and that is cheap to
// estimate: k_cdf_sample relabels 2^18 entries taken at regular distances from the column array and counts them in
// CDF_K bins of the new id space; k_cdf_table turns the counts (mixed with 1/9 of a uniform distribution: no bin is
// flat) into a monotone piecewise-linear F: [0, m) -> [0, 2^32), (base, slope / 2^16) per bin, which the tile kernel
// keeps in LDS: bucket = F(key) >> (32 - log2 buckets).  The map only balances buckets — any monotone F sorts right.
constexpr int CDF_K = 256;
constexpr int CDF_LOG_K = 8;
constexpr int CDF_MAX_COL_BITS = 25;
constexpr int CDF_SAMPLE_BLOCKS = 64, CDF_SAMPLE_ITEMS = 16;  // x 256 threads = 2^18 samples
constexpr uint64_t PT2_MAX_BYTES = 0xFFFFFFC0ull;             // arrays addressed by 32-bit byte offsets
constexpr unsigned PT2_DEAD_ELEM = 0x3FFFFFFCu;               // x 4 or x 8 (mod 2^32) lies behind every such array

struct CdfMap {
  const uint2 *table;  // CDF_K x (base, slope >> 16); nullptr: no map (the equal-width kernels run)
  int shift;           // bin = key >> shift
  int fsh;             // position inside the bin, 16 bits: (key << fsh) >> 16
};

template <typename I>
__global__ __launch_bounds__(256) void k_cdf_sample(const I *__restrict__ col_in, const I *__restrict__ col_order,
                                                    int64_t nnz, int64_t m, int shift, unsigned *__restrict__ hist) {
  static_assert(CDF_K == 256, "one bin per thread");
  __shared__ unsigned s_h[CDF_K];
  const int tid = threadIdx.x;
  s_h[tid] = 0;
  __syncthreads();
  const int64_t S = (int64_t)gridDim.x * 256 * CDF_SAMPLE_ITEMS;
  const int64_t stride = nnz >= S ? nnz / S : 1;
  unsigned c[CDF_SAMPLE_ITEMS], k[CDF_SAMPLE_ITEMS];
#pragma unroll
  for (int u = 0; u < CDF_SAMPLE_ITEMS; u++) {
    const int64_t j = ((int64_t)blockIdx.x * CDF_SAMPLE_ITEMS + u) * 256 + tid;
    const int64_t pos = j * stride;
    c[u] = pos < nnz ? (unsigned)col_in[pos] : 0xFFFFFFFFu;
  }
#pragma unroll
  for (int u = 0; u < CDF_SAMPLE_ITEMS; u++)
    k[u] = (uint64_t)c[u] < (uint64_t)m ? (unsigned)col_order[c[u]] : 0xFFFFFFFFu;
#pragma unroll
  for (int u = 0; u < CDF_SAMPLE_ITEMS; u++) {
    if ((uint64_t)k[u] < (uint64_t)m) {
      const unsigned bin = k[u] >> shift;
      atomicAdd(&s_h[bin < (unsigned)CDF_K ? bin : (unsigned)CDF_K - 1u], 1u);
    }
  }
  __syncthreads();
  if (s_h[tid]) atomicAdd(&hist[tid], s_h[tid]);
}

__global__ __launch_bounds__(CDF_K) void k_cdf_table(const unsigned *__restrict__ hist, uint2 *__restrict__ table,
                                                     int flat) {
  __shared__ unsigned long long s_scan[CDF_K / 64 + 1];
  const int tid = threadIdx.x;
  const unsigned long long hcnt = hist[tid];
  const unsigned long long total = sbx_block_sum<unsigned long long, CDF_K>(hcnt, s_scan);
  const unsigned long long w = hcnt * (8ull * CDF_K) + total + 1ull;  // eight parts sample, one part uniform
  unsigned long long W = 0;
  const unsigned long long cum = sbx_block_exclusive_sum<unsigned long long, CDF_K>(w, s_scan, &W);
  const unsigned long long b0 = (cum << 32) / W, b1 = tid == CDF_K - 1 ? (1ull << 32) : ((cum + w) << 32) / W;
  // F(key) = base + (slope >> 16) * (16-bit position inside the bin) < base + slope = the next bin's base
  // (flat: fewer ids than bins — every id is a bin of its own)
  table[tid] = make_uint2((unsigned)b0, flat ? 0u : (unsigned)((b1 - b0) >> 16));
}

// enqueues the map's kernels on h->stream; map->table == nullptr afterwards: not applicable
template <typename I>
static int build_cdf_map(sbx_handle_t h, const I *col_in, const I *col_order, int64_t nnz, int64_t m, CdfMap *map) {
  map->table = nullptr, map->shift = 0, map->fsh = 0;
  const int col_bits = sbx_bits_for(m > 0 ? (uint64_t)(m - 1) : 0);
  if (col_bits > CDF_MAX_COL_BITS || nnz <= 0 || !col_order) return SBX_OK;
  unsigned *hist = nullptr;
  uint2 *table = nullptr;
  SBX_TRY(sbx_salloc(h, (size_t)CDF_K, &hist));
  SBX_TRY(sbx_salloc(h, (size_t)CDF_K, &table));
  SBX_HIP(h, hipMemsetAsync(hist, 0, sizeof(unsigned) * CDF_K, h->stream));
  const int shift = col_bits > CDF_LOG_K ? col_bits - CDF_LOG_K : 0;
  const int64_t want = (nnz + 256 * CDF_SAMPLE_ITEMS - 1) / (256 * CDF_SAMPLE_ITEMS);
  const unsigned blocks = (unsigned)(want < CDF_SAMPLE_BLOCKS ? want : CDF_SAMPLE_BLOCKS);
  SBX_KLAUNCH(h, SBX_K_PERMUTE_PREP, k_cdf_sample<I>, dim3(blocks), dim3(256), col_in, col_order, nnz, m, shift, hist);
  SBX_KLAUNCH(h, SBX_K_PERMUTE_PREP, k_cdf_table, dim3(1), dim3(CDF_K), (const unsigned *)hist, table, shift == 0 ? 1 : 0);
  SBX_LAUNCH_CHECK(h);
  map->table = table, map->shift = shift, map->fsh = shift ? 32 - shift : 0;
  return SBX_OK;
}

#define PT2_RSRC_FLAGS 0x00020000  // raw buffer, 32-bit data format (gfx9 descriptor word 3)
typedef unsigned pt2_u2 __attribute__((ext_vector_type(2)));

template <typename I, int VB>
__global__ __launch_bounds__(PT_THREADS, 4) void k_permute_tile2(
    const int2 *__restrict__ rec, const I *col_in, const char *val_in, const I *__restrict__ col_order,
    const I *__restrict__ rpo, const I *__restrict__ sp, const I *__restrict__ tile_first, I *col_out, char *val_out,
    int64_t nr, PermState *__restrict__ st, int64_t tiles, unsigned in_bytes_c, unsigned in_bytes_v, unsigned out_bytes_c,
    unsigned out_bytes_v, unsigned table_bytes, const CdfMap cdf) {
  typedef typename ValT<VB>::type V;
  constexpr bool HASV = VB != 0;
  constexpr int IB = (int)sizeof(I);
  constexpr int ITEMS = 8, CAP = 512;
  static_assert(PT_THREADS == 64 && PT_CAP == CAP && PT_ITEMS == ITEMS && PT_LMAX <= 128, "one wave, 8 entries per lane, rows of at most 128");
  // LDS words of the wave's pool, by phase (6.2 KB: the waves per CU follow from it):
  //   row map   [0, 2 CAP)  the rows' records, 8 bytes per head position
  //   sort      [0, W_WRD)  counters (2 per position, the dead positions' word, pad); [W_WRD, POOL) the placed words
  //   out       [0, ...)    the sorted keys (VB 4: (key, value) pairs; VB 8: keys, then the 8-byte values)
  constexpr int W_WRD = 2 * CAP + 8, POOL = W_WRD + CAP + 8;
  constexpr int JUNK = 2 * CAP + 4;
  static_assert(2 * (CAP + 1) <= W_WRD && (CAP + 8) + 2 * (CAP + 1) <= POOL, "the sorted entries fit where the sort's arrays were");
  __shared__ __attribute__((aligned(16))) unsigned s_pool[POOL];
  __shared__ unsigned s_mask[16];
  __shared__ uint2 s_cdf[CDF_K];
  uint2 *const s_rr = (uint2 *)s_pool;
  unsigned *const s_c = s_pool;
  unsigned *const s_w = s_pool + W_WRD;
  unsigned *const s_key = s_pool;
  uint2 *const s_pair = (uint2 *)s_pool;
  uint64_t *const s_v8 = (uint64_t *)(s_pool + CAP + 8);
#define T2_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
  const int lane = threadIdx.x;
  const uint64_t le = ((uint64_t)2 << lane) - 1;  // lanes 0 .. lane
  const __amdgpu_buffer_rsrc_t b_ci = __builtin_amdgcn_make_buffer_rsrc((void *)col_in, 0, (int)in_bytes_c, PT2_RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t b_vi = __builtin_amdgcn_make_buffer_rsrc((void *)val_in, 0, (int)in_bytes_v, PT2_RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t b_tab = __builtin_amdgcn_make_buffer_rsrc((void *)col_order, 0, (int)table_bytes, PT2_RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t b_co = __builtin_amdgcn_make_buffer_rsrc((void *)col_out, 0, (int)out_bytes_c, PT2_RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t b_vo = __builtin_amdgcn_make_buffer_rsrc((void *)val_out, 0, (int)out_bytes_v, PT2_RSRC_FLAGS);
  for (int i = lane; i < CDF_K; i += 64) s_cdf[i] = cdf.table[i];
  const int cdf_shift = cdf.shift, cdf_fsh = cdf.fsh;

  const int64_t G = gridDim.x, last = tiles - 1;
  auto clampt = [&](int64_t x) { return x < last ? x : last; };
  auto clampr = [&](int64_t x) { return x < nr - 1 ? x : nr - 1; };
  int64_t t = blockIdx.x;
  TileIn<I> cur = tile_in_now<I>(clampt(t), rec, rpo, sp, tile_first, nr);
  I ra1 = tile_first[clampt(t + G)], rb1 = tile_first[clampt(t + G) + 1];
  bool any_uns = false, any_dup = false;
  for (; t < tiles; t += G) {
    // issued now, read after the current tile: range of the tile after next, bounds and row records of the next one
    const I ra2 = tile_first[clampt(t + 2 * G)], rb2 = tile_first[clampt(t + 2 * G) + 1];
    TileIn<I> nx;
    nx.ra = ra1, nx.rb = rb1;
    nx.e0 = sp[ra1];
    nx.e1 = sp[rb1];
    const int64_t r1 = clampr((int64_t)ra1 + lane);
    nx.rc = rec[r1];
    nx.sp_r = sp[r1];
    nx.rpo_r = rpo[r1];
    const TileIn<I> in = cur;
    cur = nx;
    ra1 = ra2, rb1 = rb2;

    const int64_t ra = in.ra, rb = in.rb;
    if (ra >= rb) continue;
    const int64_t e0 = in.e0;
    const int cnt = (int)((int64_t)in.e1 - e0);
    if (cnt <= 0) continue;
    // ---- row map: heads mark the mask and leave their record
    if (lane < 16) s_mask[lane] = 0;
    auto put_head = [&](const int p, const int2 rc, const I rpo_r) {
      atomicOr(&s_mask[p >> 5], 1u << (p & 31));
      const unsigned len = (unsigned)rc.x;
      const unsigned lg = len > (unsigned)BK_SHORT ? (unsigned)bits_u32(len - 1u) : 0u;  // 2^lg buckets, 2^lg < 2 len; lg <= 7
      // (source - position) << 3 | lg: the source array holds fewer than 2^28 entries (the caller checked)
      s_rr[p] = make_uint2(((unsigned)(rc.y - p) << 3) | lg, (unsigned)((int)rpo_r - p));
    };
    if (rb - ra <= 4 * CAP) {
      int64_t r = ra + lane;
      if (r < rb) {  // the first row of every lane arrived with the tile's bounds
        if (in.rc.x > 0 && in.rc.x <= PT_LMAX) put_head((int)((int64_t)in.sp_r - e0), in.rc, in.rpo_r);
        r += 64;
      }
      for (; r < rb; r += 64) {
        const int2 rc = rec[r];
        if (rc.x > 0 && rc.x <= PT_LMAX) put_head((int)((int64_t)sp[r] - e0), rc, rpo[r]);
      }
    } else {  // a window of mostly empty rows: every position looks for the row that starts there
      for (int p = lane; p < cnt; p += 64) {
        const int64_t target = e0 + p;
        int64_t lo = ra, hi = rb;
        while (hi - lo > 1) {
          const int64_t mid = (lo + hi) >> 1;
          if ((int64_t)sp[mid] <= target) lo = mid;
          else hi = mid;
        }
        if ((int64_t)sp[lo] == target) {
          const int2 rc = rec[lo];
          if (rc.x > 0 && rc.x <= PT_LMAX) put_head(p, rc, rpo[lo]);
        }
      }
    }
    T2_SYNC();
    const unsigned mw = s_mask[lane & 15];
    T2_SYNC();
    // ---- per entry: its row's head (the highest mask bit at or below its position), the row's record, the loads
    unsigned c[ITEMS], oidx[ITEMS], meta[ITEMS];  // meta: position in the row | (31 - lg) << 8
    unsigned hpv[ITEMS];
    V kv[HASV ? ITEMS : 1];
    int carry = 0;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      const uint64_t mk = (uint64_t)(unsigned)__builtin_amdgcn_readlane((int)mw, 2 * k) |
                          ((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)mw, 2 * k + 1) << 32);
      const uint64_t x = mk & le;
      const int hp = x ? k * 64 + 63 - __builtin_clzll(x) : carry;
      if (mk) carry = k * 64 + 63 - __builtin_clzll(mk);
      const int p = k * 64 + lane;
      const bool live = p < cnt;
      const uint2 rr = s_rr[hp];
      const unsigned pin = (unsigned)(p - hp);
      hpv[k] = (unsigned)hp;
      meta[k] = pin | ((31u - (rr.x & 7u)) << 8);
      const unsigned sidx = live ? (unsigned)(((int)rr.x >> 3) + p) : PT2_DEAD_ELEM;
      oidx[k] = live ? rr.y + (unsigned)p : PT2_DEAD_ELEM;
      c[k] = __builtin_amdgcn_raw_buffer_load_b32(b_ci, sidx * (unsigned)IB, 0, 2);
      if (HASV) {
        if (VB == 4) {
          kv[k] = (V)__builtin_amdgcn_raw_buffer_load_b32(b_vi, sidx * 4u, 0, 2);
        } else {
          const pt2_u2 v2 = __builtin_amdgcn_raw_buffer_load_b64(b_vi, sidx * 8u, 0, 2);
          kv[k] = (V)(((uint64_t)v2.y << 32) | v2.x);
        }
      }
    }
    unsigned kc[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; k++) kc[k] = __builtin_amdgcn_raw_buffer_load_b32(b_tab, c[k] * (unsigned)IB, 0, 0);
    // ---- order inside the rows (csr.cc:102-116): the neighbour in front through DPP
    bool uns = false, dupq = false;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      const unsigned first = k ? (unsigned)__builtin_amdgcn_readlane((int)kc[k - 1], 63) : 0u;
      const unsigned prev = (unsigned)sbx_wave_shift_up1((int)kc[k], (int)first);
      const bool inrow = (meta[k] & 0xFFu) != 0u && k * 64 + lane < cnt;
      uns |= inrow & (kc[k] < prev);
      dupq |= inrow & (kc[k] == prev);
    }
    if (!__any(uns)) {
      // every row of the tile is in column order already: a sort would not move anything
#pragma unroll
      for (int k = 0; k < ITEMS; k++) {
        if (IB == 4) {
          __builtin_amdgcn_raw_buffer_store_b32(kc[k], b_co, oidx[k] * 4u, 0, 0);
        } else {
          pt2_u2 w2;
          w2.x = kc[k], w2.y = 0u;
          __builtin_amdgcn_raw_buffer_store_b64(w2, b_co, oidx[k] * 8u, 0, 0);
        }
        if (HASV) {
          if (VB == 4) {
            __builtin_amdgcn_raw_buffer_store_b32((unsigned)kv[k], b_vo, oidx[k] * 4u, 0, 0);
          } else {
            pt2_u2 w2;
            w2.x = (unsigned)kv[k], w2.y = (unsigned)((uint64_t)kv[k] >> 32);
            __builtin_amdgcn_raw_buffer_store_b64(w2, b_vo, oidx[k] * 8u, 0, 0);
          }
        }
      }
      any_dup |= dupq;
      continue;
    }
    any_uns = true;
    // ---- bucket-rank sort.  Counters: the row whose head is at hp owns the words [2 hp, 2 hp + 2^lg)
    T2_SYNC();  // (the records have been read: the counters take their place)
#pragma unroll
    for (int i = 0; i < (2 * CAP) / 256; i++) *(uint4 *)&s_c[i * 256 + 4 * lane] = make_uint4(0, 0, 0, 0);
    if (lane < 2) *(uint4 *)&s_c[2 * CAP + 4 * lane] = make_uint4(0, 0, 0, 0);
    unsigned ci[ITEMS], ar[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      const uint2 tb = s_cdf[kc[k] >> cdf_shift];
      const unsigned f = tb.x + __umul24(tb.y, (kc[k] << cdf_fsh) >> 16);
      const unsigned b = (f >> 1) >> (meta[k] >> 8);  // = f >> (32 - lg); lg = 0: 0
      ci[k] = k * 64 + lane < cnt ? 2u * hpv[k] + b : (unsigned)JUNK;
    }
    T2_SYNC();
#pragma unroll
    for (int k = 0; k < ITEMS; k++) ar[k] = atomicAdd(&s_c[ci[k]], 1u);
    T2_SYNC();
    {  // exclusive scan of the 2 cnt counters in place: a lane owns 16 consecutive ones
      unsigned v[16];
#pragma unroll
      for (int i = 0; i < 16; i += 4) {
        const uint4 q = *(const uint4 *)&s_c[16 * lane + i];
        v[i] = q.x, v[i + 1] = q.y, v[i + 2] = q.z, v[i + 3] = q.w;
      }
      unsigned sum = 0;
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const unsigned y = v[i];
        v[i] = sum;
        sum += y;
      }
      const unsigned ex = sbx_wave_inclusive_sum(sum) - sum;
#pragma unroll
      for (int i = 0; i < 16; i += 4)
        *(uint4 *)&s_c[16 * lane + i] = make_uint4(v[i] + ex, v[i + 1] + ex, v[i + 2] + ex, v[i + 3] + ex);
      // (the words behind the counters: s_c[2 CAP] must read as the total for the last bucket's end)
      if (lane == 63) s_c[2 * CAP] = ex + sum;
    }
    T2_SYNC();
    // placement: slot = bucket start + arrival; the word orders a bucket by (column, arrival)
    unsigned s0[ITEMS], cb[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      const unsigned a0 = s_c[ci[k]], a1 = s_c[ci[k] + 1];
      const bool live = k * 64 + lane < cnt;
      s0[k] = a0;
      cb[k] = live ? a1 - a0 : 0u;
      s_w[live ? a0 + ar[k] : (unsigned)CAP + 4u] = (kc[k] << 7) | ar[k];  // (dead positions: a word of the pad)
    }
    T2_SYNC();
    unsigned fin[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      const unsigned me = (kc[k] << 7) | ar[k];
      const unsigned w0 = s_w[s0[k]], w1 = s_w[s0[k] + 1], w2 = s_w[s0[k] + 2], w3 = s_w[s0[k] + 3];
      unsigned r = (unsigned)(w0 < me) + ((w1 < me) & (cb[k] > 1u)) + ((w2 < me) & (cb[k] > 2u)) + ((w3 < me) & (cb[k] > 3u));
      if (__any(cb[k] > 4u)) {
        for (unsigned j = 4; __any(j < cb[k]); j++) r += (j < cb[k]) & (s_w[s0[k] + (j < cb[k] ? j : 0u)] < me);
      }
      fin[k] = k * 64 + lane < cnt ? s0[k] + r : (unsigned)CAP;  // (dead positions: the slot behind the last one)
    }
    T2_SYNC();  // (the placed words and the bounds have been read)
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      if (VB == 4) {
        s_pair[fin[k]] = make_uint2(kc[k], (unsigned)kv[k]);
      } else {
        s_key[fin[k]] = kc[k];
        if (VB == 8) s_v8[fin[k]] = (uint64_t)kv[k];
      }
    }
    T2_SYNC();
    // ---- out: position p holds the p-th entry of the tile in (row, column) order; rows keep their position ranges
    bool dup = false;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      const int p = k * 64 + lane;
      unsigned key;
      V val = (V)0;
      if (VB == 4) {
        const uint2 pr = s_pair[p];
        key = pr.x, val = (V)pr.y;
      } else {
        key = s_key[p];
        if (VB == 8) val = (V)s_v8[p];
      }
      c[k] = key;
      if (HASV) kv[k] = val;
    }
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      const unsigned first = k ? (unsigned)__builtin_amdgcn_readlane((int)c[k - 1], 63) : 0u;
      const unsigned prev = (unsigned)sbx_wave_shift_up1((int)c[k], (int)first);
      dup |= ((meta[k] & 0xFFu) != 0u) & (k * 64 + lane < cnt) & (c[k] == prev);
      if (IB == 4) {
        __builtin_amdgcn_raw_buffer_store_b32(c[k], b_co, oidx[k] * 4u, 0, 0);
      } else {
        pt2_u2 w2;
        w2.x = c[k], w2.y = 0u;
        __builtin_amdgcn_raw_buffer_store_b64(w2, b_co, oidx[k] * 8u, 0, 0);
      }
      if (HASV) {
        if (VB == 4) {
          __builtin_amdgcn_raw_buffer_store_b32((unsigned)kv[k], b_vo, oidx[k] * 4u, 0, 0);
        } else {
          pt2_u2 w2;
          w2.x = (unsigned)kv[k], w2.y = (unsigned)((uint64_t)kv[k] >> 32);
          __builtin_amdgcn_raw_buffer_store_b64(w2, b_vo, oidx[k] * 8u, 0, 0);
        }
      }
    }
    any_dup |= dup;
    T2_SYNC();  // (the sorted entries have been read: the next tile's records take their place)
  }
  if (__any(any_uns) && lane == 0) st->any_unsorted = 1;
  if (__any(any_dup) && lane == 0) st->any_dup = 1;
#undef T2_SYNC
}

template <typename I>
__global__ __launch_bounds__(256) void k_fill_index(I *__restrict__ a, int64_t count, I value) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < count; i += stride) a[i] = value;
}

// tile_first[t] = first row r in [0, nr] whose output range starts at or after position t * PT_W
template <typename I>
__global__ __launch_bounds__(256) void k_tile_first(const I *__restrict__ rpo, int64_t nr, int64_t ntiles,
                                                    I *__restrict__ tile_first) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t > ntiles) return;
  const int64_t pos = t * PT_W;
  int64_t lo = 0, hi = nr;  // first r with rpo[r] >= pos, nr if none
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((int64_t)rpo[mid] >= pos) hi = mid;
    else lo = mid + 1;
  }
  tile_first[t] = (I)lo;
}

// ---- rows of (PT_LMAX, capacity] entries: one workgroup per row ------------------------
// LDS-only barrier: waits for this wave's LDS traffic, not for its global loads — the prefetches of the rows
// to come stay in flight across it (__syncthreads() would drain them: vmcnt counts loads and stores alike).
template <int THREADS>
__device__ __forceinline__ void lds_barrier() {
  if (THREADS > 64) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // one wave: LDS operations execute in order
}

// A workgroup walks the rows blockIdx.x, blockIdx.x + gridDim.x, ... of its class list as a software pipeline:
// while row i is sorted in LDS and streamed out, the relabel gathers of row i + 1, the column / value loads of
// row i + 2, the row record of row i + 3 and the list entry of row i + 4 are in flight (registers only).  The
// sort itself is the bucket-rank pass (or, if the row's columns cluster, the LSD radix sort over the significant
// column bits); HBM sees each nonzero once in and once out.
template <typename I, int VB, int CAP, int BR_THREADS>
__global__ __launch_bounds__(BR_THREADS) void k_permute_block_rows(
    const int2 *__restrict__ rec, const I *col_in, const char *val_in, const I *__restrict__ col_order,
    const I *__restrict__ rpo, const I *__restrict__ block_rows, int n_rows, I *col_out, char *val_out,
    PermState *__restrict__ st, int force_radix, unsigned *__restrict__ fb_rows, unsigned *__restrict__ fb_count,
    const unsigned *__restrict__ n_rows_dev) {
  typedef typename ValT<VB>::type V;
  constexpr bool HASV = VB != 0;
  constexpr int ITEMS = CAP / BR_THREADS;
  if (n_rows_dev) n_rows = (int)*n_rows_dev;  // (segments of long rows: the host does not know how many there are)
  constexpr int WAVES = BR_THREADS / 64;
  static_assert(sizeof(V) <= 8, "sorted values are staged in the placed-word + counter regions");
  __shared__ __attribute__((aligned(16))) unsigned s_pool[3 * CAP + 4];
  __shared__ unsigned s_scan[WAVES + 1];
  __shared__ unsigned s_mm[2 * WAVES];
  __shared__ int s_flag[4];  // [0] row out of order, [1] radix list, [2] refine, [3] overfull bucket at level 0
  int *const s_key = (int *)s_pool;             // gathered columns -> sorted columns
  int *const s_a = (int *)s_pool + CAP;         // placed words
  unsigned *const s_c = s_pool + 2 * CAP;       // bucket counters (CAP + 4 words)
  V *const s_val = (V *)(s_pool + CAP);         // sorted values: over the placed words (and, 8-byte values, the counters)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int G = (int)gridDim.x;

  // pipeline state.  Row-level quantities are the same in every lane but travel in vector registers so that their
  // loads are vector-memory loads (scalar loads share lgkmcnt with LDS: every LDS wait would wait for them too).
  // Every load of an iteration is issued unconditionally (clamped addresses) at its top and nothing reads the
  // results before the rotation that follows the sort: the compiler's waits then sit behind the sort, where the
  // loads have landed, and the stores of the sorted row are issued after the rotation, behind nothing.
  int rid_d = 0;                                  // stage D: list entry loaded
  int e0_c = 0, len_c = -1, src_c = 0, rid_c = 0; // stage C: row record loaded
  I c_b[ITEMS];                                   // stage B: columns / values loaded
  V v_b[HASV ? ITEMS : 1];
  int e0_b = 0, len_b = -1, rid_b = 0;
  int k_a[ITEMS];                                 // stage A: relabelled columns loaded -> sorted this iteration
  V v_a[HASV ? ITEMS : 1];
  int e0_a = 0, len_a = -1, rid_a = 0;
#pragma unroll
  for (int k = 0; k < ITEMS; k++) {
    c_b[k] = 0;
    k_a[k] = 0;
    if (HASV) v_b[k] = (V)0, v_a[k] = (V)0;
  }
  int zero = 0;
  asm volatile("" : "+v"(zero));  // a vector-register zero the compiler cannot fold: keeps the row-level loads on the vector path
  const int my_rows = ((int)blockIdx.x < n_rows) ? (n_rows - 1 - (int)blockIdx.x) / G + 1 : 0;
  bool valid_d = false;

  for (int it = 0; it < my_rows + 4; it++) {
    // E: list entry of row it
    const bool valid_e = it < my_rows;
    const int rid_e = (int)block_rows[(int64_t)blockIdx.x + (int64_t)(valid_e ? it : 0) * G + zero];
    // D: row record of row it - 1
    const int rid_s = valid_d ? rid_d : 0;
    const int r0_d = (int)rpo[rid_s], r1_d = (int)rpo[rid_s + 1], src_d = rec[rid_s].y;
    // C: columns / values of row it - 2
    I c_c[ITEMS];
    V v_c[HASV ? ITEMS : 1];
    {
      const int64_t base = len_c > 0 ? (int64_t)src_c : 0;
#pragma unroll
      for (int k = 0; k < ITEMS; k++) {
        const int p = k * BR_THREADS + tid;
        const int64_t o = base + (p < len_c ? p : 0);
        c_c[k] = __builtin_nontemporal_load(col_in + o);
        if (HASV) v_c[k] = __builtin_nontemporal_load((const V *)val_in + o);
      }
    }
    // B: relabel gathers of row it - 3 (permute_order_two.cc:68)
    int k_b[ITEMS];
    if (col_order && !(force_radix & 4)) {  // (bit 2: timing ablation without the relabel gathers)
#pragma unroll
      for (int k = 0; k < ITEMS; k++) k_b[k] = (int)col_order[k * BR_THREADS + tid < len_b ? c_b[k] : 0];
    } else {
#pragma unroll
      for (int k = 0; k < ITEMS; k++) k_b[k] = (int)c_b[k];
    }

    // A: sort row it - 4 in LDS
    const int len = __builtin_amdgcn_readfirstlane(len_a);
    const int64_t e0 = e0_a;
    int out_mode = 0;  // 1: sorted row in LDS, to be streamed out after the rotation
    if (len >= 0) {
      if (tid < 4) s_flag[tid] = 0;
#pragma unroll
      for (int k = 0; k < ITEMS; k++) {
        if (k * BR_THREADS + tid >= len) k_a[k] = 0x7FFFFFFF;  // padding: sorts last, never written out
        s_key[k * BR_THREADS + tid] = k_a[k];
      }
      lds_barrier<BR_THREADS>();
      {
        bool unsorted = false;
        unsigned mn = 0xFFFFFFFFu, mx = 0;
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
          const int p = k * BR_THREADS + tid;
          if (p < len) {
            if (p > 0 && k_a[k] < s_key[p - 1]) unsorted = true;
            mn = (unsigned)k_a[k] < mn ? (unsigned)k_a[k] : mn;
            mx = (unsigned)k_a[k] > mx ? (unsigned)k_a[k] : mx;
          }
        }
        mn = sbx_wave_min(mn);
        mx = sbx_wave_max(mx);
        if (lane == 0) {
          s_mm[2 * w] = mn;
          s_mm[2 * w + 1] = mx;
        }
        if (__any(unsorted) && lane == 0 && !(force_radix & 2)) {  // (bit 1: timing ablation, rows stream out unsorted)
          st->any_unsorted = 1;
          s_flag[0] = 1;
        }
      }
      lds_barrier<BR_THREADS>();
      if (__builtin_amdgcn_readfirstlane(s_flag[0]) == 0) {  // an ordered row needs no sort
        bool dup = false;
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
          const int p = k * BR_THREADS + tid;
          if (p < len) {
            if (p > 0 && k_a[k] == s_key[p - 1]) dup = true;
            col_out[e0 + p] = (I)k_a[k];
            if (HASV) ((V *)val_out)[e0 + p] = v_a[k];
          }
        }
        if (__any(dup) && lane == 0) st->any_dup = 1;
      } else {
        unsigned mn = 0xFFFFFFFFu, mx = 0;
#pragma unroll
        for (int i = 0; i < WAVES; i++) {
          mn = s_mm[2 * i] < mn ? s_mm[2 * i] : mn;
          mx = s_mm[2 * i + 1] > mx ? s_mm[2 * i + 1] : mx;
        }
        const int rbits = bits_u32(mx - mn);
        const int ib = bits_u32((unsigned)len - 1);  // 2^ib buckets, len <= 2^ib <= CAP
        const int shift = rbits > ib ? rbits - ib : 0;
        const unsigned lowmask = shift >= 32 ? 0xFFFFFFFFu : (1u << shift) - 1u;  // shift + ib = max(rbits, ib) <= 32
        unsigned bk[ITEMS], wd[ITEMS];
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
          const int p = k * BR_THREADS + tid;
          const unsigned rel = (unsigned)k_a[k] - mn;
          bk[k] = shift >= 32 ? 0u : rel >> shift;
          wd[k] = ((rel & lowmask) << ib) | (unsigned)p;
        }
        // level 0: buckets of equal width.  If one of them holds more than BK_REFINE entries (clustered columns, e.g. a
        // banded order), level 1 splits every bucket into as many sub-buckets as it has entries, again by interpolation.
        for (int level = 0;; level++) {
#pragma unroll
          for (int k = 0; k < ITEMS; k++) s_c[k * BR_THREADS + tid] = 0;
          if (tid < 4) s_c[CAP + tid] = (unsigned)len;  // end of the last bucket
          lds_barrier<BR_THREADS>();
#pragma unroll
          for (int k = 0; k < ITEMS; k++)
            if (k * BR_THREADS + tid < len) atomicAdd(&s_c[bk[k]], 1u);
          lds_barrier<BR_THREADS>();
          {
            // in-place inclusive scan of the counters (as scan_bucket_counts, with LDS-only barriers)
            const int base = tid * ITEMS;
            unsigned v[ITEMS];
            unsigned sum = 0, mxc = 0;
#pragma unroll
            for (int i = 0; i < ITEMS; i += 4) {
              const uint4 q = *(const uint4 *)(s_c + base + i);
              v[i] = q.x, v[i + 1] = q.y, v[i + 2] = q.z, v[i + 3] = q.w;
            }
#pragma unroll
            for (int i = 0; i < ITEMS; i++) {
              mxc = v[i] > mxc ? v[i] : mxc;
              sum += v[i];
              v[i] = sum;
            }
            const unsigned inc = sbx_wave_inclusive_sum(sum);
            if (lane == 63) s_scan[w] = inc;
            if (level == 0 && __any(mxc > (unsigned)BK_REFINE) && lane == 0) s_flag[2] = 1;
            if ((__any(mxc > (unsigned)BK_MAX) || (force_radix & 1)) && lane == 0) s_flag[1 + 2 * (1 - level)] = 1;
            lds_barrier<BR_THREADS>();
            unsigned ex = inc - sum;
#pragma unroll
            for (int i = 0; i < WAVES; i++)
              if (i < w) ex += s_scan[i];
#pragma unroll
            for (int i = 0; i < ITEMS; i += 4)
              *(uint4 *)(s_c + base + i) = make_uint4(v[i] + ex, v[i + 1] + ex, v[i + 2] + ex, v[i + 3] + ex);
          }
          lds_barrier<BR_THREADS>();
          if (level == 1 || __builtin_amdgcn_readfirstlane(s_flag[2]) == 0) {
            if (level == 0 && __builtin_amdgcn_readfirstlane(s_flag[3])) {  // not refined: an overfull level-0 bucket stands
              if (tid == 0) s_flag[1] = 1;
              lds_barrier<BR_THREADS>();
            }
            break;
          }
#pragma unroll
          for (int k = 0; k < ITEMS; k++) {
            if (k * BR_THREADS + tid < len) {
              const unsigned b = bk[k];
              const unsigned start = b ? s_c[b - 1] : 0u, cntb = s_c[b] - start;
              const unsigned low = ((unsigned)k_a[k] - mn) & lowmask;
              const unsigned sub = shift ? (unsigned)(((unsigned long long)low * cntb) >> shift) : 0u;
              bk[k] = start + sub;  // < start + cntb: sub-buckets of different buckets do not meet
            }
          }
          lds_barrier<BR_THREADS>();  // the level-0 bounds have been read
        }
        if (__builtin_amdgcn_readfirstlane(s_flag[1]) != 0) {  // (rare) the row goes on the list of k_permute_rows_radix
          if (tid == 0) fb_rows[atomicAdd(fb_count, 1u)] = (unsigned)rid_a;
        } else {
#pragma unroll
          for (int k = 0; k < ITEMS; k++)
            if (k * BR_THREADS + tid < len) s_a[atomicSub(&s_c[bk[k]], 1u) - 1u] = (int)wd[k];
          lds_barrier<BR_THREADS>();
          int fin[ITEMS];
#pragma unroll
          for (int k = 0; k < ITEMS; k++) {
            fin[k] = 0;
            if (k * BR_THREADS + tid < len) {
              const unsigned b0 = s_c[bk[k]], b1 = s_c[bk[k] + 1];
              unsigned rk = 0;
              for (unsigned j = b0; j < b1; j++) rk += (unsigned)s_a[j] < wd[k];
              fin[k] = (int)(b0 + rk);
            }
          }
          lds_barrier<BR_THREADS>();  // placed words and bucket bounds are dead: the sorted entries move in
#pragma unroll
          for (int k = 0; k < ITEMS; k++)
            if (k * BR_THREADS + tid < len) {
              s_key[fin[k]] = k_a[k];
              if (HASV) s_val[fin[k]] = v_a[k];
            }
          out_mode = 1;
        }
      }
    }

    // rotate the pipeline (the first reads of this iteration's loads)
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      k_a[k] = k_b[k];
      if (HASV) v_a[k] = v_b[k];
      c_b[k] = c_c[k];
      if (HASV) v_b[k] = v_c[k];
    }
    e0_a = e0_b, len_a = len_b, rid_a = rid_b;
    e0_b = e0_c, len_b = len_c, rid_b = rid_c;
    e0_c = r0_d, len_c = valid_d ? r1_d - r0_d : -1, src_c = src_d, rid_c = rid_d;
    rid_d = rid_e, valid_d = valid_e;

    if (len >= 0) {
      if (out_mode) {  // stream the sorted row out of LDS
        lds_barrier<BR_THREADS>();
        bool dup = false;
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
          const int p = k * BR_THREADS + tid;
          if (p < len) {
            const int c = s_key[p];
            if (p > 0 && c == s_key[p - 1]) dup = true;
            col_out[e0 + p] = (I)c;
            if (HASV) ((V *)val_out)[e0 + p] = s_val[p];
          }
        }
        if (__any(dup) && lane == 0) st->any_dup = 1;
      }
      lds_barrier<BR_THREADS>();  // the next row's columns overwrite s_key
    }
  }
}

// the rows the kernel above listed (clustered columns): one workgroup per row, LSD radix sort over the column bits
template <typename I, int VB>
__global__ __launch_bounds__(1024) void k_permute_rows_radix(
    const int2 *__restrict__ rec, const I *col_in, const char *val_in, const I *__restrict__ col_order,
    const I *__restrict__ rpo, const unsigned *__restrict__ fb_rows, I *col_out, char *val_out, int col_bits,
    PermState *__restrict__ st, const unsigned *__restrict__ fb_count) {
  typedef typename ValT<VB>::type V;
  constexpr bool HASV = VB != 0;
  constexpr int THREADS = 1024, CAP = BlockRowCap<VB>::value, ITEMS = CAP / THREADS, WAVES = THREADS / 64;
  __shared__ int s_key[CAP];
  __shared__ V s_val[HASV ? CAP : 1];
  __shared__ unsigned s_whist[WAVES * 256];
  __shared__ unsigned s_scan[WAVES + 1];
  const int tid = threadIdx.x, lane = tid & 63;
  const unsigned n = *fb_count;
  for (unsigned i = blockIdx.x; i < n; i += gridDim.x) {
    const int64_t r = fb_rows[i];
    const int64_t e0 = rpo[r];
    const int len = (int)((int64_t)rpo[r + 1] - e0);
    const int64_t src0 = rec[r].y;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      const int p = k * THREADS + tid;
      int c = 0x7FFFFFFF;  // padding: sorts last, never written out
      if (p < len) {
        const I cc = col_in[src0 + p];
        c = (int)(col_order ? col_order[cc] : cc);
        if (HASV) s_val[p] = ((const V *)val_in)[src0 + p];
      }
      s_key[p] = c;
    }
    __syncthreads();
    bool unsorted = false;  // an ordered row is left alone
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      const int p = k * THREADS + tid;
      if (p > 0 && p < len && s_key[p] < s_key[p - 1]) unsorted = true;
    }
    if (__syncthreads_or(unsorted)) {
      if (tid == 0) st->any_unsorted = 1;
      lds_radix_sort<V, HASV, false, THREADS, ITEMS>(s_key, nullptr, s_val, s_whist, s_scan, CAP, col_bits, 0);
    }
    bool dup = false;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      const int p = k * THREADS + tid;
      if (p < len) {
        const int c = s_key[p];
        if (p > 0 && c == s_key[p - 1]) dup = true;
        col_out[e0 + p] = (I)c;
        if (HASV) ((V *)val_out)[e0 + p] = s_val[p];
      }
    }
    if (__any(dup) && lane == 0) st->any_dup = 1;
    __syncthreads();
  }
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

// ---- long rows, column-relabelled permutes: split into segments, sort the segments in LDS ------------------
// A row of L > capacity relabelled columns is cut into column ranges holding about LS_TARGET entries each; its entries
// are partitioned by range through two compact buffers and every range — a segment — is then a "virtual row" of the
// one-workgroup-per-row kernel above (bucket-rank sort in LDS), written straight to its place in the output.  The
// ranges adapt to the row: a histogram over F = 64 .. 1024 equal bins between the row's smallest and largest column,
// consecutive bins grouped while their running count stays inside a window of T entries (segment <= T + fullest bin).
// Three passes over the long rows' entries plus one over their columns (gather, histogram, partition, sort) against
// the ~ 6 of the global radix sort on (row rank, column), which stays for what this does not cover: a row that is
// already ordered (the constructor leaves such a row alone, so its equal columns must keep their order: the partition
// is not stable), a row with a bin too full for a segment (columns clustered below 1/F of the row's range), rows of
// more than 2 M entries, and the row-wise permutes.
constexpr int LS_MAXLG = 10;      // at most 1024 histogram bins per row
constexpr int LS_MAXSEG = 2048;   // segment slots per row held in LDS by the partition pass (two rows per chunk)
constexpr int LS_CHUNK = 4096;    // compact entries per workgroup: spans at most two long rows (every one is longer)
constexpr int LS_THREADS = 256;
constexpr int LS_ITEMS = LS_CHUNK / LS_THREADS;

__host__ __device__ __forceinline__ int ls_lg_bins(int64_t len) {  // 2^lg bins: ~ 128 .. 256 entries each, at most 1024
  int lg = 0;
  for (int64_t t = (len - 1) / 256; t > 0; t >>= 1) lg++;
  return lg > LS_MAXLG ? LS_MAXLG : lg;
}
__host__ __device__ __forceinline__ unsigned ls_slots(int64_t len, int target) { return (unsigned)(len / target) + 1u; }

// exclusive prefix sums over the long rows of their lengths, bin counts and segment slots: one workgroup walks the
// list (a few hundred to a few ten thousand rows) instead of three scans of three launches each.
// Its 1024 threads also decide WHEN the long rows' chain (eight dependent kernels on a side stream) runs: in a kernel
// trace this one-workgroup kernel completes 0.58 ms after its dispatch, when the tile kernel is running out, so the
// chain runs beside the big row classes instead of beside the tile kernel.  Measured on the bench matrix (A/B in one
// process pair, round 3): as is 1.40 / 1.45 ms (random / RCM order); with 256 threads — the chain beside the tile
// kernel from the start, its segment sort then fighting the two big classes for the CUs (three persistent kernels of
// one 1024-thread workgroup per CU) — 1.53 / 1.59; the chain behind an event recorded after the tile kernel
// 1.48 / 1.54.  An explicit 1024-thread gate kernel in front of a 256-thread version did NOT reproduce the delay
// (1.53), so the mechanism is not simply "16 free wave slots on one CU"; it is kept because it measures best, and
// NOTES §4.4-r3 has the traces.
template <typename I>
__global__ __launch_bounds__(1024) void k_long_seg_offsets(const I *__restrict__ rpo, const I *__restrict__ long_rows,
                                                           uint32_t *__restrict__ loff, uint32_t *__restrict__ foff,
                                                           uint32_t *__restrict__ soff, int n_long, int target) {
  __shared__ unsigned s_scan[1024 / 64 + 1];
  unsigned run_l = 0, run_f = 0, run_s = 0;
  for (int k0 = 0; k0 < n_long; k0 += 1024) {
    const int k = k0 + (int)threadIdx.x;
    unsigned l = 0, f = 0, sl = 0;
    if (k < n_long) {
      const int64_t len = (int64_t)rpo[long_rows[k] + 1] - (int64_t)rpo[long_rows[k]];
      l = (unsigned)len;
      f = 1u << ls_lg_bins(len);
      sl = ls_slots(len, target);
    }
    unsigned tl, tf, ts;
    const unsigned el = sbx_block_exclusive_sum<unsigned, 1024>(l, s_scan, &tl);
    const unsigned ef = sbx_block_exclusive_sum<unsigned, 1024>(f, s_scan, &tf);
    const unsigned es = sbx_block_exclusive_sum<unsigned, 1024>(sl, s_scan, &ts);
    if (k < n_long) {
      loff[k] = run_l + el;
      foff[k] = run_f + ef;
      soff[k] = run_s + es;
    }
    run_l += tl, run_f += tf, run_s += ts;
  }
}

// the (at most two) long rows a chunk of the compact entry space touches: the row of the chunk's first entry
__device__ __forceinline__ int ls_first_row(const uint32_t *__restrict__ loff, int n_long, int64_t e) {
  int lo = 0, hi = n_long - 1;  // last k with loff[k] <= e
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((int64_t)loff[mid] <= e) lo = mid; else hi = mid - 1;
  }
  return lo;
}

struct LsChunk {  // what a workgroup knows about its chunk [base, base + LS_CHUNK) of the compact entry space
  int k0;                  // row of the first entry; the chunk's other row, if any, is k0 + 1
  int64_t start0, end0;    // row k0 = compact entries [start0, end0); entries from end0 on belong to row k0 + 1
  int64_t lenB;
  bool hasB;
};
__device__ __forceinline__ LsChunk ls_chunk(const uint32_t *__restrict__ loff, int n_long, int64_t long_nnz, int64_t base) {
  LsChunk c;
  c.k0 = ls_first_row(loff, n_long, base);
  c.start0 = loff[c.k0];
  c.end0 = c.k0 + 1 < n_long ? (int64_t)loff[c.k0 + 1] : long_nnz;
  c.hasB = base + LS_CHUNK > c.end0 && c.k0 + 1 < n_long;
  c.lenB = c.hasB ? (c.k0 + 2 < n_long ? (int64_t)loff[c.k0 + 2] : long_nnz) - c.end0 : 0;
  return c;
}
// bin of a column inside its row's range [mn, mx]: equal widths 2^shift, fewer than 2^lg of them
__device__ __forceinline__ int ls_shift(unsigned mn, unsigned mx, int lg) {
  const int rb = bits_u32(mx - mn);
  return rb > lg ? rb - lg : 0;
}

// pass 1: gather + relabel into the first compact buffer; per-row "out of order" flags and column range
// (rowmm[2 k] = ~min, rowmm[2 k + 1] = max: both grow from the zero fill)
template <typename I, int VB>
__global__ __launch_bounds__(LS_THREADS) void k_long_seg_gather(
    const int2 *__restrict__ rec, const I *col_in, const char *val_in, const I *__restrict__ col_order,
    const I *__restrict__ long_rows, const uint32_t *__restrict__ loff, int n_long, int64_t long_nnz,
    I *__restrict__ c1, char *__restrict__ v1, unsigned *__restrict__ rowmm, unsigned *__restrict__ row_unsorted,
    PermState *__restrict__ st) {
  typedef typename ValT<VB>::type V;
  __shared__ int s_col[LS_CHUNK + 1];  // [0] = the entry in front of the chunk
  const int tid = threadIdx.x;
  const int64_t base = (int64_t)blockIdx.x * LS_CHUNK;
  const LsChunk ch = ls_chunk(loff, n_long, long_nnz, base);
  const int64_t srcA = rec[long_rows[ch.k0]].y, srcB = ch.hasB ? (int64_t)rec[long_rows[ch.k0 + 1]].y : 0;
  if (tid == 0) {  // the relabelled column in front of the chunk (same row only)
    int pc = 0;
    if (base > ch.start0) {
      const I c = col_in[srcA + (base - 1 - ch.start0)];
      pc = (int)(col_order ? col_order[c] : c);
    }
    s_col[0] = pc;
  }
  int cc[LS_ITEMS];
#pragma unroll
  for (int i = 0; i < LS_ITEMS; i++) {
    const int64_t e = base + i * LS_THREADS + tid;
    cc[i] = 0;
    if (e < long_nnz) {
      const bool b = e >= ch.end0;
      const int64_t s = b ? srcB + (e - ch.end0) : srcA + (e - ch.start0);
      const I c = __builtin_nontemporal_load(col_in + s);
      cc[i] = (int)(col_order ? col_order[c] : c);
      c1[e] = (I)cc[i];
      if (VB) ((V *)v1)[e] = __builtin_nontemporal_load((const V *)val_in + s);
    }
    s_col[1 + i * LS_THREADS + tid] = cc[i];
  }
  __syncthreads();
  bool unsA = false, unsB = false;
  unsigned nmnA = 0, mxA = 0, nmnB = 0, mxB = 0;
#pragma unroll
  for (int i = 0; i < LS_ITEMS; i++) {
    const int p = i * LS_THREADS + tid;
    const int64_t e = base + p;
    if (e < long_nnz) {
      const bool b = e >= ch.end0;
      const bool first = b ? e == ch.end0 : e == ch.start0;
      const unsigned c = (unsigned)cc[i];
      if (!first && cc[i] < s_col[p]) (b ? unsB : unsA) = true;
      if (b) {
        nmnB = ~c > nmnB ? ~c : nmnB;
        mxB = c > mxB ? c : mxB;
      } else {
        nmnA = ~c > nmnA ? ~c : nmnA;
        mxA = c > mxA ? c : mxA;
      }
    }
  }
  nmnA = sbx_wave_max(nmnA), mxA = sbx_wave_max(mxA);
  if (sbx_lane() == 0) {
    if (nmnA) atomicMax(&rowmm[2 * ch.k0], nmnA);
    if (mxA) atomicMax(&rowmm[2 * ch.k0 + 1], mxA);
  }
  if (ch.hasB) {
    nmnB = sbx_wave_max(nmnB), mxB = sbx_wave_max(mxB);
    if (sbx_lane() == 0) {
      if (nmnB) atomicMax(&rowmm[2 * ch.k0 + 2], nmnB);
      if (mxB) atomicMax(&rowmm[2 * ch.k0 + 3], mxB);
    }
  }
  if (__any(unsA) && sbx_lane() == 0) row_unsorted[ch.k0] = 1;
  if (__any(unsB) && sbx_lane() == 0) row_unsorted[ch.k0 + 1] = 1;
  if (__any(unsA || unsB) && sbx_lane() == 0) st->any_unsorted = 1;
}

// pass 2: per-row histograms of the relabelled columns (one streaming read of the first compact buffer)
template <typename I>
__global__ __launch_bounds__(LS_THREADS) void k_long_seg_hist(const I *__restrict__ c1, const uint32_t *__restrict__ loff,
                                                              const uint32_t *__restrict__ foff, int n_long,
                                                              int64_t long_nnz, const unsigned *__restrict__ rowmm,
                                                              unsigned *__restrict__ fine) {
  __shared__ unsigned s_hist[2][1 << LS_MAXLG];
  const int tid = threadIdx.x;
  const int64_t base = (int64_t)blockIdx.x * LS_CHUNK;
  const LsChunk ch = ls_chunk(loff, n_long, long_nnz, base);
  const int lgA = ls_lg_bins(ch.end0 - ch.start0), lgB = ch.hasB ? ls_lg_bins(ch.lenB) : 0;
  const unsigned mnA = ~rowmm[2 * ch.k0], mnB = ch.hasB ? ~rowmm[2 * ch.k0 + 2] : 0u;
  const int shA = ls_shift(mnA, rowmm[2 * ch.k0 + 1], lgA);
  const int shB = ch.hasB ? ls_shift(mnB, rowmm[2 * ch.k0 + 3], lgB) : 0;
  for (int i = tid; i < 2 * (1 << LS_MAXLG); i += LS_THREADS) (&s_hist[0][0])[i] = 0;
  __syncthreads();
  unsigned cc[LS_ITEMS];  // all loads first (clamped index): under `e < long_nnz` each waited for the one before it
#pragma unroll
  for (int i = 0; i < LS_ITEMS; i++) {
    const int64_t e = base + i * LS_THREADS + tid;
    cc[i] = (unsigned)c1[e < long_nnz ? e : long_nnz - 1];
  }
#pragma unroll
  for (int i = 0; i < LS_ITEMS; i++) {
    const int64_t e = base + i * LS_THREADS + tid;
    if (e < long_nnz) {
      const bool b = e >= ch.end0;
      const unsigned c = cc[i];
      atomicAdd(&s_hist[b][b ? (c - mnB) >> shB : (c - mnA) >> shA], 1u);
    }
  }
  __syncthreads();
  for (int i = tid; i < (1 << lgA); i += LS_THREADS)
    if (s_hist[0][i]) atomicAdd(&fine[foff[ch.k0] + i], s_hist[0][i]);
  if (ch.hasB)
    for (int i = tid; i < (1 << lgB); i += LS_THREADS)
      if (s_hist[1][i]) atomicAdd(&fine[foff[ch.k0 + 1] + i], s_hist[1][i]);
}

// pass 3 (one workgroup per long row): bins -> segments.  Bin b with exclusive prefix P belongs to segment P / T, so a
// segment holds fewer than T + (its last bin) entries; fine[] becomes the bin -> segment map, segstart[] where a
// segment starts inside the row.  Virtual row 2 v of the row kernel: vrec = (length, source in the second compact
// buffer), vrpo[2 v], vrpo[2 v + 1] = its output range.  Or the row is left to the global radix sort.
template <typename I>
__global__ __launch_bounds__(256) void k_long_seg_scan(const I *__restrict__ rpo, const I *__restrict__ long_rows,
                                                       const uint32_t *__restrict__ loff,
                                                       const uint32_t *__restrict__ foff,
                                                       const uint32_t *__restrict__ soff, int n_long, int64_t long_nnz,
                                                       int target, int seg_cap, unsigned *__restrict__ fine,
                                                       unsigned *__restrict__ segstart,
                                                       const unsigned *__restrict__ row_unsorted,
                                                       unsigned *__restrict__ row_skip, int2 *__restrict__ vrec,
                                                       I *__restrict__ vrpo, I *__restrict__ vlist, int64_t vstride,
                                                       I *__restrict__ fb_list, PermState *__restrict__ st) {
  constexpr int PER = (1 << LS_MAXLG) / 256;
  __shared__ unsigned s_scan[256 / 64 + 1];
  __shared__ unsigned s_cnt[LS_MAXSEG];
  __shared__ int s_skip;
  const int k = blockIdx.x, tid = threadIdx.x;
  const int64_t start = loff[k], len = (k + 1 < n_long ? (int64_t)loff[k + 1] : long_nnz) - start;
  const int F = 1 << ls_lg_bins(len);
  const unsigned slots = ls_slots(len, target);
  const unsigned fo = foff[k], so = soff[k];
  unsigned c[PER], sum = 0, mx = 0;
#pragma unroll
  for (int i = 0; i < PER; i++) {
    const int b = tid * PER + i;
    c[i] = b < F ? fine[fo + b] : 0u;
    sum += c[i];
    mx = c[i] > mx ? c[i] : mx;
  }
  if (tid == 0) s_skip = (row_unsorted[k] == 0 || slots > (unsigned)LS_MAXSEG) ? 1 : 0;  // (an ordered row: stable path)
  for (unsigned g = tid; g < slots && g < (unsigned)LS_MAXSEG; g += 256) s_cnt[g] = 0;
  __syncthreads();
  if (mx + (unsigned)target > (unsigned)seg_cap) s_skip = 1;  // a segment may reach target - 1 + its last bin
  unsigned all;
  unsigned ex = sbx_block_exclusive_sum<unsigned, 256>(sum, s_scan, &all);  // (its barriers publish s_skip)
  const int skip = s_skip;
  if (tid == 0) {
    row_skip[k] = (unsigned)skip;
    if (skip) {
      fb_list[atomicAdd(&st->n_long_fb, 1u)] = long_rows[k];
      atomicAdd(&st->long_fb_nnz, (unsigned long long)len);
    }
  }
  if (skip) return;
#pragma unroll
  for (int i = 0; i < PER; i++) {
    const int b = tid * PER + i;
    if (b < F) {
      const unsigned g = ex / (unsigned)target;
      fine[fo + b] = g;
      if (c[i]) atomicAdd(&s_cnt[g], c[i]);
    }
    ex += c[i];
  }
  __syncthreads();
  // segment starts: exclusive prefix of the segment counts
  const int64_t out0 = rpo[long_rows[k]];
  unsigned run = 0;
  for (unsigned g0 = 0; g0 < slots; g0 += 256) {
    const unsigned g = g0 + tid;
    const unsigned n = g < slots ? s_cnt[g] : 0u;
    unsigned tot;
    const unsigned e = run + sbx_block_exclusive_sum<unsigned, 256>(n, s_scan, &tot);
    run += tot;
    const bool emit = n > 0;
    const int cls = n > 4096u ? 1 : 0;
    const unsigned s0 = sbx_wave_append(&st->n_seg[0], emit && cls == 0);
    const unsigned s1 = sbx_wave_append(&st->n_seg[1], emit && cls == 1);
    if (g < slots) segstart[so + g] = e;
    if (emit) {
      const int64_t v = (int64_t)so + g;
      vrec[2 * v] = make_int2((int)n, (int)(start + e));
      vrpo[2 * v] = (I)(out0 + e);
      vrpo[2 * v + 1] = (I)(out0 + e + n);
      vlist[(int64_t)cls * vstride + (cls ? s1 : s0)] = (I)(2 * v);
    }
  }
}

// pass 4: the first compact buffer partitioned by segment into the second one (the order of the entries inside a
// segment is whatever the atomics give: the segment is sorted next)
template <typename I, int VB>
__global__ __launch_bounds__(LS_THREADS) void k_long_seg_partition(
    const I *__restrict__ c1, const char *__restrict__ v1, const uint32_t *__restrict__ loff,
    const uint32_t *__restrict__ foff, const uint32_t *__restrict__ soff, int n_long, int64_t long_nnz, int target,
    const unsigned *__restrict__ rowmm, const unsigned *__restrict__ fine, const unsigned *__restrict__ segstart,
    unsigned *__restrict__ cursor, const unsigned *__restrict__ row_skip, I *__restrict__ c2, char *__restrict__ v2) {
  typedef typename ValT<VB>::type V;
  __shared__ unsigned short s_map[2][1 << LS_MAXLG];  // bin -> segment
  __shared__ unsigned s_hist[2][LS_MAXSEG];           // counts, then: where this chunk's entries of the segment go
  const int tid = threadIdx.x;
  const int64_t base = (int64_t)blockIdx.x * LS_CHUNK;
  const LsChunk ch = ls_chunk(loff, n_long, long_nnz, base);
  const bool skipA = row_skip[ch.k0] != 0, skipB = ch.hasB ? row_skip[ch.k0 + 1] != 0 : true;
  if (skipA && skipB) return;
  const int lgA = ls_lg_bins(ch.end0 - ch.start0), lgB = ch.hasB ? ls_lg_bins(ch.lenB) : 0;
  const unsigned slotsA = skipA ? 0u : ls_slots(ch.end0 - ch.start0, target);
  const unsigned slotsB = skipB ? 0u : ls_slots(ch.lenB, target);
  const unsigned mnA = ~rowmm[2 * ch.k0], mnB = ch.hasB ? ~rowmm[2 * ch.k0 + 2] : 0u;
  const int shA = ls_shift(mnA, rowmm[2 * ch.k0 + 1], lgA);
  const int shB = ch.hasB ? ls_shift(mnB, rowmm[2 * ch.k0 + 3], lgB) : 0;
  if (!skipA)
    for (int i = tid; i < (1 << lgA); i += LS_THREADS) s_map[0][i] = (unsigned short)fine[foff[ch.k0] + i];
  if (!skipB)
    for (int i = tid; i < (1 << lgB); i += LS_THREADS) s_map[1][i] = (unsigned short)fine[foff[ch.k0 + 1] + i];
  for (unsigned i = tid; i < slotsA; i += LS_THREADS) s_hist[0][i] = 0;
  for (unsigned i = tid; i < slotsB; i += LS_THREADS) s_hist[1][i] = 0;
  __syncthreads();
  int cc[LS_ITEMS];
  unsigned sg[LS_ITEMS];
  // (all loads of a thread before anything uses them, at clamped indices: issued one by one under the conditions
  // below, each waited for the one before it)
#pragma unroll
  for (int i = 0; i < LS_ITEMS; i++) {
    const int64_t e = base + i * LS_THREADS + tid;
    cc[i] = (int)c1[e < long_nnz ? e : long_nnz - 1];
  }
#pragma unroll
  for (int i = 0; i < LS_ITEMS; i++) {
    const int64_t e = base + i * LS_THREADS + tid;
    sg[i] = 0xFFFFFFFFu;
    if (e < long_nnz) {
      const bool b = e >= ch.end0;
      if (!(b ? skipB : skipA)) {
        const unsigned c = (unsigned)cc[i];
        const unsigned g = b ? s_map[1][(c - mnB) >> shB] : s_map[0][(c - mnA) >> shA];
        sg[i] = (b ? (unsigned)LS_MAXSEG : 0u) + g;
        atomicAdd(&(&s_hist[0][0])[sg[i]], 1u);
      }
    }
  }
  __syncthreads();
  // one reservation per (chunk, segment): where the chunk's entries of the segment start in the second buffer
  for (unsigned i = tid; i < slotsA; i += LS_THREADS) {
    const unsigned n = s_hist[0][i];
    if (n) s_hist[0][i] = (unsigned)ch.start0 + segstart[soff[ch.k0] + i] + atomicAdd(&cursor[soff[ch.k0] + i], n);
  }
  for (unsigned i = tid; i < slotsB; i += LS_THREADS) {
    const unsigned n = s_hist[1][i];
    if (n) s_hist[1][i] = (unsigned)ch.end0 + segstart[soff[ch.k0 + 1] + i] + atomicAdd(&cursor[soff[ch.k0 + 1] + i], n);
  }
  __syncthreads();
  V vv[VB ? LS_ITEMS : 1];
  if (VB) {
#pragma unroll
    for (int i = 0; i < LS_ITEMS; i++) {
      const int64_t e = base + i * LS_THREADS + tid;
      vv[i] = ((const V *)v1)[e < long_nnz ? e : long_nnz - 1];
    }
  }
#pragma unroll
  for (int i = 0; i < LS_ITEMS; i++) {
    if (sg[i] != 0xFFFFFFFFu) {
      const unsigned d = atomicAdd(&(&s_hist[0][0])[sg[i]], 1u);
      c2[d] = (I)cc[i];
      if (VB) ((V *)v2)[d] = vv[i];
    }
  }
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
  static const bool on = !(sbx_env_test("SBX_PERMUTE_OVERLAP") && atoi(sbx_env_test("SBX_PERMUTE_OVERLAP")) == 0);
  return on;
}

static int permute_grid_factor() {  // SBX_PERMUTE_GRID_FACTOR: workgroups launched per resident slot (tuning)
  static const int f = sbx_env_tuning("SBX_PERMUTE_GRID_FACTOR") ? atoi(sbx_env_tuning("SBX_PERMUTE_GRID_FACTOR")) : 1;
  return f < 1 ? 1 : f;
}

// SBX_PERMUTE_FORCE_RADIX=1: every tile / row takes the radix path (tests; a path production takes for clustered columns).
// The timing ablations — bits 1 - 3 of the same variable (rows stream out unsorted / no relabel gathers / ...) and
// SBX_DEBUG_TILE_STOP (leave the tile kernel behind phase k) — produce WRONG matrices and exist in the tuning build only.
static int permute_force_radix() {
  static const int on =
      (sbx_env_test("SBX_PERMUTE_FORCE_RADIX") ? atoi(sbx_env_test("SBX_PERMUTE_FORCE_RADIX")) & 0x01 : 0) |
      (sbx_env_tuning("SBX_PERMUTE_FORCE_RADIX") ? atoi(sbx_env_tuning("SBX_PERMUTE_FORCE_RADIX")) & 0xFE : 0) |
      (sbx_env_tuning("SBX_DEBUG_TILE_STOP") ? atoi(sbx_env_tuning("SBX_DEBUG_TILE_STOP")) << 8 : 0);
  return on;
}

// SBX_PERMUTE_ROW_WAVES: resident waves per CU the grids of the row classes up to 2048 entries are sized for.  Measured on
// the bench matrix (tools/kt_rowwaves.sh; classes of 256 + 512 + 1024 slots): with a random order 8 / 12 / 16 waves per
// CU take 399 / 401 / 440 us, with the RCM order (every row takes the second, interpolating level: more LDS round trips
// to hide) 538 / 484 / 463 us; 4 waves are 25 - 60 % slower, more than the 16 the registers admit queue workgroups
// behind each other.  Orders that come out of a reordering are the common case: 16.
// Side stream of a row class (0: the stage's class stream itself; 1 belongs to the long rows).  HIP feeds its streams
// through four hardware queues by default, so some of the stage's eight streams share a queue with each other or with
// the caller's stream, and kernels that look concurrent wait behind unrelated ones (a kernel trace of the stage,
// tools/permute_span.py, shows it).  More queues do not help — GPU_MAX_HW_QUEUES=8 / 16: the stage is 3 - 6 % SLOWER,
// persistent kernels that all run at once fight for the CUs' LDS and wave slots — and fewer streams are a coin toss:
// all classes on one stream 1.32 / 1.49 ms (random / RCM order) on one box and 1.34 / 1.44 on another, where the six
// streams below gave 1.31 / 1.42 and 1.35 / 1.46; big and small classes on two streams 1.34 / 1.40 and 1.40 / 1.50
// (tools/permute_streams.sh).  Left at one stream per class.  SBX_PERMUTE_CLASS_STREAMS: six digits, class 0 ... 5.
static int class_stream(int cls) {
  static int map[BR_CLASSES] = {-1};
  if (map[0] < 0) {
    const char *e = sbx_env_tuning("SBX_PERMUTE_CLASS_STREAMS");
    const char *d = (e && strlen(e) == BR_CLASSES) ? e : "023456";
    for (int c = 0; c < BR_CLASSES; c++) {
      const int v = d[c] - '0';
      map[c] = (v < 0 || v >= SBX_AUX_STREAMS || v == 1) ? 0 : v;
    }
  }
  return map[cls];
}

static int rq_waves_per_cu() {
  static const int f = sbx_env_tuning("SBX_PERMUTE_ROW_WAVES") ? atoi(sbx_env_tuning("SBX_PERMUTE_ROW_WAVES")) : 16;
  return f < 1 ? 1 : f;
}

static int tile_grid_factor() {  // SBX_PERMUTE_TILE_GRID: persistent tile waves per resident slot (tuning)
  static const int f = sbx_env_tuning("SBX_PERMUTE_TILE_GRID") ? atoi(sbx_env_tuning("SBX_PERMUTE_TILE_GRID")) : 12;
  return f < 1 ? 1 : f;
}

#include "sbx_rowsort.h"


// rows of PT_LMAX < length <= 8 K: one workgroup per row, by capacity class
template <typename I, int VB>
int block_rows_path(sbx_handle_t h, const int2 *rec, const I *col_in, const char *val_in, const I *col_order,
                    const I *rpo, I *col_out, char *val_out, int64_t m, const I *block_rows,
                    const unsigned *n_block, int64_t block_stride, int64_t block_nnz, PermState *st, bool fork) {
  // fork: h->stream is side stream 0, which already waits for the fork event; the classes spread over side streams
  // 0, 2, 3, ... (1 belongs to the long rows) and the radix kernel behind them joins them again on stream 0
  hipStream_t base = h->stream;
  const int col_bits = sbx_bits_for(m > 0 ? (uint64_t)(m - 1) : 0);
  const int force = permute_force_radix() & 0xFF;
  size_t n_all = 0;
  for (int c = 0; c < BR_CLASSES; c++) n_all += n_block[c];
  unsigned *fb_rows = nullptr;  // rows whose columns cluster (listed by the class kernels, sorted by the radix kernel)
  SBX_TRY(sbx_salloc(h, n_all + 1, &fb_rows));
  // persistent grids: enough workgroups to fill the CUs' LDS a few times over, each walking its rows as a pipeline
#define BLOCK_ROWS(CLS, THREADS)                                                                                  \
  if (n_block[CLS]) {                                                                                             \
    /* resident workgroups per CU: LDS (12 B per slot) and ~24 waves (the kernels need 64..100 VGPRs) */          \
    const unsigned by_lds = (unsigned)(160 * 1024 / (br_cap(CLS) * 12 + 256)), by_waves = 24u / ((THREADS) / 64); \
    const unsigned per_cu = by_lds < by_waves ? by_lds : by_waves;                                                \
    unsigned grid = (unsigned)h->num_cus * (per_cu < 1 ? 1 : per_cu) * (unsigned)permute_grid_factor();           \
    if (grid > n_block[CLS]) grid = n_block[CLS];                                                                 \
    const int si = class_stream(CLS);                                                                             \
    if (fork && si) {                                                                                             \
      h->stream = h->aux_stream[si];                                                                              \
      SBX_HIP(h, hipStreamWaitEvent(h->stream, h->aux_event[0], 0));                                              \
    }                                                                                                             \
    SBX_KLAUNCH(h, SBX_K_PERMUTE_BLOCK, (k_permute_block_rows<I, VB, br_cap(CLS), THREADS>), dim3(grid),          \
                dim3(THREADS), rec, col_in, val_in, col_order, rpo, block_rows + (CLS)*block_stride,              \
                (int)n_block[CLS], col_out, val_out, st, force, fb_rows, &st->n_fb_rows,                          \
                (const unsigned *)nullptr);                                                                       \
    if (fork && si) {                                                                                             \
      const hipError_t e1_ = hipEventRecord(h->aux_event[1 + si], h->stream);                                     \
      const hipError_t e2_ = hipStreamWaitEvent(base, h->aux_event[1 + si], 0);                                   \
      h->stream = base;                                                                                           \
      if (e1_ != hipSuccess || e2_ != hipSuccess) SBX_FAIL(h, SBX_ERR_HIP, "side stream hand-over failed");       \
    }                                                                                                             \
  }
  // k_rows_quad: T threads x Q quads = the class capacity; the column map must fit a buffer descriptor (4 GB)
  // (without a column map — csr_sort_rows, the hybrid COO sort's groups — there is no gather to hide the sort behind and
  // round 3's kernels, lighter in registers, measure 6 % better on C2B: 0.536 against 0.570 ms)
  // (64-bit index arrays: the same kernel, two 16-byte accesses per quad of columns, the map's low words gathered)
  const bool quad = col_order && (uint64_t)m * sizeof(I) <= 0xFFFFFFFCull;
  const unsigned table_bytes = col_order ? (unsigned)((uint64_t)m * sizeof(I)) : 0u;
#define QUAD_ROWS(CLS, THREADS, QUADS, MINW)                                                                      \
  if (n_block[CLS]) {                                                                                             \
    static_assert(4 * (THREADS) * (QUADS) == br_cap(CLS), "class capacity");                                      \
    const unsigned by_lds = (unsigned)(160 * 1024 / RqLds<VB, THREADS, QUADS>::BYTES);                            \
    const unsigned by_waves = (unsigned)((THREADS) >= 512 ? 16 : rq_waves_per_cu()) / ((THREADS) / 64);           \
    const unsigned per_cu = by_lds < by_waves ? by_lds : by_waves;                                                \
    unsigned grid = (unsigned)h->num_cus * (per_cu < 1 ? 1 : per_cu) * (unsigned)permute_grid_factor();           \
    if (grid > n_block[CLS]) grid = n_block[CLS];                                                                 \
    const int si = class_stream(CLS);                                                                             \
    if (fork && si) {                                                                                             \
      h->stream = h->aux_stream[si];                                                                              \
      SBX_HIP(h, hipStreamWaitEvent(h->stream, h->aux_event[0], 0));                                              \
    }                                                                                                             \
    SBX_KLAUNCH(h, SBX_K_PERMUTE_BLOCK, (k_rows_quad<I, VB, THREADS, QUADS, MINW>), dim3(grid), dim3(THREADS), rec, \
                col_in, val_in, col_order, rpo, block_rows + (CLS)*block_stride, (int)n_block[CLS], col_out,      \
                val_out, st, force, fb_rows, &st->n_fb_rows, (const unsigned *)nullptr, table_bytes);             \
    if (fork && si) {                                                                                             \
      const hipError_t e1_ = hipEventRecord(h->aux_event[1 + si], h->stream);                                     \
      const hipError_t e2_ = hipStreamWaitEvent(base, h->aux_event[1 + si], 0);                                   \
      h->stream = base;                                                                                           \
      if (e1_ != hipSuccess || e2_ != hipSuccess) SBX_FAIL(h, SBX_ERR_HIP, "side stream hand-over failed");       \
    }                                                                                                             \
  }
  if (quad) {
    QUAD_ROWS(0, 64, 1, 1);
    // the 512-slot class keeps round 3's kernel: one wave per row and two quad steps need 112 registers (16 waves per
    // CU) where that kernel runs 24 waves — with the RCM order, whose rows all take the second sort level, 146 us against
    // its 118 on the bench matrix (random order: 128 / 112; measured again with the interleaved gathers of round 5: level)
    BLOCK_ROWS(1, 64);
    QUAD_ROWS(2, 128, 2, (sizeof(I) == 8 ? 4 : 1));  // (64-bit index arrays: held to 128 registers, four waves per SIMD)
    QUAD_ROWS(3, 256, 2, (sizeof(I) == 8 ? 4 : 1));
    QUAD_ROWS(4, 512, 2, (VB == 8 ? 2 : 4));  // two workgroups per CU: at most 128 registers (8-byte values: LDS allows one)
    if constexpr (VB != 8) QUAD_ROWS(5, 1024, 2, 1);
  }
  if (!quad) {
    // threads per class, measured on the bench matrix: one wave (no s_barrier at all) up to 512 entries, 8 entries per
    // thread up to 2048, 4 for 4096 (8 per thread there: +5 %)
    BLOCK_ROWS(0, 64);
    BLOCK_ROWS(1, 64);
    BLOCK_ROWS(2, 128);
    BLOCK_ROWS(3, 256);
    BLOCK_ROWS(4, 1024);
    if constexpr (VB != 8) BLOCK_ROWS(5, 1024);  // 8-byte values: 8192 entries do not fit LDS, those rows are "long"
  }
#undef BLOCK_ROWS
#undef QUAD_ROWS
  SBX_KLAUNCH(h, SBX_K_PERMUTE_BLOCK, (k_permute_rows_radix<I, VB>), dim3((unsigned)(n_all < 512 ? n_all : 512)),
              dim3(1024), rec, col_in, val_in, col_order, rpo, (const unsigned *)fb_rows, col_out, val_out, col_bits, st,
              (const unsigned *)&st->n_fb_rows);
  SBX_LAUNCH_CHECK(h);
  SBX_PROF_BYTES(h, SBX_K_PERMUTE_BLOCK, block_nnz * (int64_t)(2 * (sizeof(I) + VB)));
  return SBX_OK;
}

// longer rows: flat gather, device radix sort on (row rank, column), scatter back
template <typename I, int VB>
int long_rows_radix_path(sbx_handle_t h, const int2 *rec, const I *col_in, const char *val_in, const I *col_order,
                   const I *rpo, I *col_out, char *val_out, int64_t m, const I *long_rows,
                   unsigned n_long, int64_t long_nnz, PermState *st) {
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



// longer rows of a permute that relabels columns: segments sorted in LDS (kernels above); what that path declines
// (rows already in order, overfull segments) and every long row of the other callers goes through the radix path
template <typename I, int VB>
int long_rows_path(sbx_handle_t h, const int2 *rec, const I *col_in, const char *val_in, const I *col_order,
                   const I *rpo, I *col_out, char *val_out, int64_t m, const I *long_rows,
                   unsigned n_long, int64_t long_nnz, PermState *st) {
  if (!col_order || (permute_force_radix() & 1) || long_nnz >= ((int64_t)1 << 31))
    return long_rows_radix_path<I, VB>(h, rec, col_in, val_in, col_order, rpo, col_out, val_out, m, long_rows, n_long,
                                    long_nnz, st);
  const int col_bits = sbx_bits_for(m > 0 ? (uint64_t)(m - 1) : 0);
  const int seg_cap = BlockRowCap<VB>::value;
  const int target = seg_cap / 4;  // a segment: fewer than target + (one bin) entries
  const size_t seg_max = (size_t)(long_nnz / target) + (size_t)n_long + 2;  // sum of the rows' segment slots
  const size_t bin_max = (size_t)(long_nnz / 128) + (size_t)n_long * 2 + 2;  // sum of the rows' bins (each < L / 128 + 2)
  uint32_t *loff = nullptr, *foff = nullptr, *soff = nullptr;
  unsigned *zero = nullptr;
  I *c1 = nullptr, *c2 = nullptr, *vrpo = nullptr, *vlist = nullptr, *fb_list = nullptr;
  char *v1 = nullptr, *v2 = nullptr;
  int2 *vrec = nullptr;
  unsigned *fb_rows = nullptr;
  SBX_TRY(sbx_salloc(h, (size_t)n_long + 1, &loff));
  SBX_TRY(sbx_salloc(h, (size_t)n_long + 1, &foff));
  SBX_TRY(sbx_salloc(h, (size_t)n_long + 1, &soff));
  // fine | cursor | row range | row_unsorted | row_skip: one fill
  const size_t zero_words = bin_max + seg_max + 4 * (size_t)n_long;
  SBX_TRY(sbx_salloc(h, zero_words + seg_max, &zero));
  unsigned *fine = zero, *cursor = fine + bin_max, *rowmm = cursor + seg_max, *row_unsorted = rowmm + 2 * (size_t)n_long;
  unsigned *row_skip = row_unsorted + n_long, *segstart = row_skip + n_long;
  SBX_TRY(sbx_salloc(h, (size_t)long_nnz, &c1));
  SBX_TRY(sbx_salloc(h, (size_t)long_nnz, &c2));
  if (VB) {
    SBX_TRY(sbx_salloc(h, (size_t)long_nnz * VB, &v1));
    SBX_TRY(sbx_salloc(h, (size_t)long_nnz * VB, &v2));
  }
  SBX_TRY(sbx_salloc(h, 2 * seg_max, &vrec));
  SBX_TRY(sbx_salloc(h, 2 * seg_max + 1, &vrpo));
  SBX_TRY(sbx_salloc(h, 2 * seg_max, &vlist));
  SBX_TRY(sbx_salloc(h, (size_t)n_long, &fb_list));
  SBX_TRY(sbx_salloc(h, seg_max, &fb_rows));
  SBX_HIP(h, hipMemsetAsync(zero, 0, sizeof(unsigned) * zero_words, h->stream));
  SBX_KLAUNCH(h, SBX_K_PERMUTE_LONG, k_long_seg_offsets<I>, dim3(1), dim3(1024), rpo, long_rows, loff, foff, soff,
              (int)n_long, target);
  const unsigned chunks = (unsigned)((long_nnz + LS_CHUNK - 1) / LS_CHUNK);
  SBX_KLAUNCH(h, SBX_K_PERMUTE_LONG, (k_long_seg_gather<I, VB>), dim3(chunks), dim3(LS_THREADS), rec, col_in, val_in,
              col_order, long_rows, (const uint32_t *)loff, (int)n_long, long_nnz, c1, v1, rowmm, row_unsorted, st);
  SBX_KLAUNCH(h, SBX_K_PERMUTE_LONG, k_long_seg_hist<I>, dim3(chunks), dim3(LS_THREADS), (const I *)c1,
              (const uint32_t *)loff, (const uint32_t *)foff, (int)n_long, long_nnz, (const unsigned *)rowmm, fine);
  SBX_KLAUNCH(h, SBX_K_PERMUTE_LONG, k_long_seg_scan<I>, dim3(n_long), dim3(256), rpo, long_rows, (const uint32_t *)loff,
              (const uint32_t *)foff, (const uint32_t *)soff, (int)n_long, long_nnz, target, seg_cap, fine, segstart,
              (const unsigned *)row_unsorted, row_skip, vrec, vrpo, vlist, (int64_t)seg_max, fb_list, st);
  SBX_KLAUNCH(h, SBX_K_PERMUTE_LONG, (k_long_seg_partition<I, VB>), dim3(chunks), dim3(LS_THREADS), (const I *)c1,
              (const char *)v1, (const uint32_t *)loff, (const uint32_t *)foff, (const uint32_t *)soff, (int)n_long,
              long_nnz, target, (const unsigned *)rowmm, (const unsigned *)fine, (const unsigned *)segstart, cursor,
              (const unsigned *)row_skip, c2, v2);
  // the segments: virtual rows of the one-workgroup-per-row kernel (columns already relabelled: no column map)
  const int force = permute_force_radix() & 0xFE;
  {
    // (512 threads, two workgroups per CU for segments of up to 4096 entries; 1024 threads for the longer ones)
    SBX_KLAUNCH(h, SBX_K_PERMUTE_LONG, (k_rows_quad<I, VB, 512, 2, (VB == 8 ? 2 : 4)>), dim3(2 * (unsigned)h->num_cus), dim3(512),
                (const int2 *)vrec, (const I *)c2, (const char *)v2, (const I *)nullptr, (const I *)vrpo, (const I *)vlist,
                0, col_out, val_out, st, force, fb_rows, &st->n_seg_fb_rows, (const unsigned *)&st->n_seg[0], 0u);
    if constexpr (VB != 8) {
      SBX_KLAUNCH(h, SBX_K_PERMUTE_LONG, (k_rows_quad<I, VB, 1024, 2, 1>), dim3((unsigned)h->num_cus), dim3(1024),
                  (const int2 *)vrec, (const I *)c2, (const char *)v2, (const I *)nullptr, (const I *)vrpo,
                  (const I *)(vlist + seg_max), 0, col_out, val_out, st, force, fb_rows, &st->n_seg_fb_rows,
                  (const unsigned *)&st->n_seg[1], 0u);
    }
  }
  SBX_KLAUNCH(h, SBX_K_PERMUTE_LONG, (k_permute_rows_radix<I, VB>), dim3(256), dim3(1024), (const int2 *)vrec,
              (const I *)c2, (const char *)v2, (const I *)nullptr, (const I *)vrpo, (const unsigned *)fb_rows, col_out,
              val_out, col_bits, st, (const unsigned *)&st->n_seg_fb_rows);
  SBX_LAUNCH_CHECK(h);
  SBX_PROF_BYTES(h, SBX_K_PERMUTE_LONG, long_nnz * (int64_t)(2 * (sizeof(I) + VB)));
  PermState hs2;
  SBX_TRY(sbx_readback(h, &hs2, st, sizeof(PermState)));
  if (sbx_env_tuning("SBX_DEBUG_LONG"))
    fprintf(stderr, "long rows %u (%lld entries): segments %u + %u, clustered segments %u, rows left to the radix sort %u (%llu entries)\n",
            n_long, (long long)long_nnz, hs2.n_seg[0], hs2.n_seg[1], hs2.n_seg_fb_rows, hs2.n_long_fb, hs2.long_fb_nnz);
  if (hs2.n_long_fb)
    SBX_TRY((long_rows_radix_path<I, VB>(h, rec, col_in, val_in, col_order, rpo, col_out, val_out, m, (const I *)fb_list,
                                         hs2.n_long_fb, (int64_t)hs2.long_fb_nnz, st)));
  return SBX_OK;
}

// ---- the wide path --------------------------------------------------------------------------------------------------
// 64-bit index arrays whose column ids do not fit the 32-bit keys of the LDS sorts (m beyond 2^31): every entry becomes
// the key (new row << col_bits | new column), one stable radix sort over all of them, and the sorted position of a key IS
// its place in the output (rows are contiguous there).  8 - 9 digit passes over 8 + VB bytes per entry instead of one
// pass through LDS: the price of generality, paid only by matrices with more than two billion columns.
constexpr int64_t KEY32_MAX_COLS = 0x7FFFFFF0ll;  // the LDS sorts keep a column in an int; ids up to here leave room for their sentinels

template <typename I, int VB>
__global__ __launch_bounds__(256) void k_wide_gather(const int2 *__restrict__ rec, const I *__restrict__ col_in,
                                                     const char *__restrict__ val_in, const I *__restrict__ col_order,
                                                     const I *__restrict__ rpo, int64_t nr, int64_t total, int col_bits,
                                                     uint64_t *__restrict__ keys, char *__restrict__ pay,
                                                     PermState *__restrict__ st) {
  typedef typename ValT<VB>::type V;
  bool unsorted = false;
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; p < total; p += stride) {
    int64_t lo = 0, hi = nr - 1;  // the last row with rpo[r] <= p: the non-empty one that owns the position
    while (lo < hi) {
      const int64_t mid = (lo + hi + 1) >> 1;
      if ((int64_t)rpo[mid] <= p) lo = mid; else hi = mid - 1;
    }
    const int64_t j = p - (int64_t)rpo[lo], src = (int64_t)rec[lo].y + j;
    I c = col_in[src];
    if (col_order) c = col_order[c];
    if (j) {
      I pc = col_in[src - 1];
      if (col_order) pc = col_order[pc];
      unsorted |= c < pc;
    }
    keys[p] = ((uint64_t)lo << col_bits) | (uint64_t)c;
    if (VB) ((V *)pay)[p] = ((const V *)val_in)[src];
  }
  if (__any(unsorted) && sbx_lane() == 0) {
    st->any_unsorted = 1;
    st->long_unsorted = 1;
  }
}

template <typename I, int VB>
__global__ __launch_bounds__(256) void k_wide_scatter(const uint64_t *__restrict__ keys, const char *__restrict__ pay,
                                                      int64_t total, int col_bits, I *__restrict__ col_out,
                                                      char *__restrict__ val_out, PermState *__restrict__ st) {
  typedef typename ValT<VB>::type V;
  const uint64_t mask = col_bits >= 64 ? ~0ull : ((1ull << col_bits) - 1ull);
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  bool dup = false;
  for (; p < total; p += stride) {
    const uint64_t key = keys[p];
    col_out[p] = (I)(key & mask);
    if (VB) ((V *)val_out)[p] = ((const V *)pay)[p];
    if (p && keys[p - 1] == key) dup = true;
  }
  if (__any(dup) && sbx_lane() == 0) st->any_dup = 1;
}

template <typename I, int VB>
int wide_rows_path(sbx_handle_t h, sbx_value_type vt, const int2 *rec, const I *col_in, const char *val_in,
                   const I *col_order, const I *rpo, I *col_out, char *val_out, int64_t nr, int64_t m, int64_t total,
                   PermState *st) {
  const int col_bits = sbx_bits_for(m > 0 ? (uint64_t)(m - 1) : 0), row_bits = sbx_bits_for(nr > 0 ? (uint64_t)(nr - 1) : 0);
  if (col_bits + row_bits > 64)
    SBX_FAIL(h, SBX_ERR_UNSUPPORTED, "permute: %d row bits + %d column bits do not fit a 64-bit sort key", row_bits, col_bits);
  uint64_t *ka = nullptr, *kb = nullptr;
  char *pa = nullptr, *pb = nullptr;
  SBX_TRY(sbx_salloc(h, (size_t)total, &ka));
  SBX_TRY(sbx_salloc(h, (size_t)total, &kb));
  if (VB) {
    SBX_TRY(sbx_salloc(h, (size_t)total * VB, &pa));
    SBX_TRY(sbx_salloc(h, (size_t)total * VB, &pb));
  }
  const unsigned grid = sbx_grid_for(total, 256, 8192);
  SBX_KLAUNCH(h, SBX_K_PERMUTE_LONG, (k_wide_gather<I, VB>), dim3(grid), dim3(256), rec, col_in, val_in, col_order, rpo, nr,
              total, col_bits, ka, pa, st);
  SBX_LAUNCH_CHECK(h);
  PermState hs2;  // rows that are already ordered skip the sort
  SBX_TRY(sbx_readback(h, &hs2, st, sizeof(PermState)));
  int in_b = 0;
  if (hs2.long_unsorted && col_bits + row_bits > 0) {
    sbx_radix_pass passes[16];
    const int np = sbx_radix_plan(0, col_bits + row_bits, 0, 0, passes);
    SBX_TRY(sbx_radix_sort(h, 8, VB, ka, kb, pa, pb, total, passes, np, &in_b));
  }
  SBX_KLAUNCH(h, SBX_K_PERMUTE_LONG, (k_wide_scatter<I, VB>), dim3(grid), dim3(256), (const uint64_t *)(in_b ? kb : ka),
              (const char *)(in_b ? pb : pa), total, col_bits, col_out, val_out, st);
  SBX_LAUNCH_CHECK(h);
  SBX_PROF_BYTES(h, SBX_K_PERMUTE_LONG, total * (int64_t)(2 * (sizeof(I) + VB)));
  if (VB) SBX_TRY(launch_fix<I>(h, vt, rpo, (const I *)col_out, val_out, nr, st));
  return SBX_OK;
}

// Sort stage shared by permute and csr_sort_rows: rows of `rpo` (nr rows, already
// on device) are produced from the source CSR through the row/col maps.
template <typename I, int VB>
int sort_stage(sbx_handle_t h, sbx_value_type vt, const int2 *rec, const I *col_in, const char *val_in,
               const I *col_order, const I *rpo, I *col_out, char *val_out, int64_t nr, int64_t m,
               int64_t total, const I *long_rows, const I *block_rows, int64_t block_stride,
               const PermState &hs, PermState *st, const I *sp, const I *tile_first_ready = nullptr,
               int64_t nnz_in = 0 /* entries of col_in / val_in */, const CdfMap *cdf = nullptr, bool cdf_on_side = false) {
  if (cdf_on_side) {  // the map was built on side stream 1: everything from here on comes behind it
    SBX_HIP(h, hipStreamWaitEvent(h->stream, h->aux_event[2], 0));
    h->aux_dirty = false;
  }
  if constexpr (sizeof(I) == 8)
    if (m > KEY32_MAX_COLS)
      return wide_rows_path<I, VB>(h, vt, rec, col_in, val_in, col_order, rpo, col_out, val_out, nr, m, total, st);
  const unsigned n_long = hs.n_long;
  const unsigned *n_block = hs.n_block;
  const int64_t long_nnz = (int64_t)hs.long_nnz, block_nnz = (int64_t)hs.block_nnz;
  // The three paths write disjoint rows of the output and only read the inputs: with more than one of them
  // present they run on streams of their own (tile kernel on the caller's stream), so the block-row kernels'
  // tails and the long-row path's short, latency-bound launches hide behind the tile kernel.  While the
  // profiler is on they run back to back, so that a kernel's event time is its own.
  bool has_block = false;
  for (int c = 0; c < BR_CLASSES; c++) has_block |= n_block[c] != 0;
  const bool fork = !h->prof_on && permute_overlap() && total > 0 && (has_block || n_long);
  (void)total;
  hipStream_t main_stream = h->stream;
  if (fork) {
    SBX_TRY(sbx_aux_streams(h));
    SBX_HIP(h, hipEventRecord(h->aux_event[0], main_stream));
    h->aux_dirty = true;
    if (has_block) SBX_HIP(h, hipStreamWaitEvent(h->aux_stream[0], h->aux_event[0], 0));
    if (n_long) SBX_HIP(h, hipStreamWaitEvent(h->aux_stream[1], h->aux_event[0], 0));
  }
  const int64_t short_nnz = total - long_nnz - block_nnz;  // entries of the rows the tile kernel sorts
  if (short_nnz > 0) {
    const int64_t tiles = (short_nnz + PT_W - 1) / PT_W;
    const I *tile_first = tile_first_ready;  // (from the classification pass, where it had the bounds to allocate it)
    if (!tile_first) {
      I *tf = nullptr;
      SBX_TRY(sbx_salloc(h, (size_t)tiles + 1, &tf));
      SBX_KLAUNCH(h, SBX_K_PERMUTE_PREP, k_tile_first<I>, dim3((unsigned)(tiles / 256 + 1)), dim3(256), sp, nr, tiles, tf);
      tile_first = tf;
    }
    unsigned *fb_tiles = nullptr;  // tiles whose rows cluster (listed by the tile kernel, sorted by its radix twin)
    SBX_TRY(sbx_salloc(h, (size_t)tiles, &fb_tiles));
    const int col_bits = sbx_bits_for(m > 0 ? (uint64_t)(m - 1) : 0);
    // persistent waves, 15 per CU (the LDS of a tile) times 12 (measured: 1 ... 2 per slot lose 10 % to imbalance — a
    // tile is 30 ... 512 entries —, 8 ... 16 are level, one wave per tile is 7 % slower): each walks its tiles as a pipeline
    const int64_t tile_grid = (int64_t)h->num_cus * 15 * tile_grid_factor();
    // the relabelling permutes whose arrays fit 32-bit byte offsets and whose ids fit the map's words: k_permute_tile2
    constexpr uint64_t EB = sizeof(I) > (size_t)VB ? sizeof(I) : (size_t)VB;
    const bool tile2 = cdf && cdf->table && col_order && permute_force_radix() == 0 && nnz_in < ((int64_t)1 << 28) &&
                       (uint64_t)nnz_in * EB <= PT2_MAX_BYTES && (uint64_t)total * EB <= PT2_MAX_BYTES &&
                       (uint64_t)m * sizeof(I) <= PT2_MAX_BYTES;
    if (tile2) {
      SBX_KLAUNCH(h, SBX_K_PERMUTE_TILE, (k_permute_tile2<I, VB>), dim3((unsigned)(tiles < tile_grid ? tiles : tile_grid)),
                  dim3(PT_THREADS), rec, col_in, val_in, col_order, rpo, sp, (const I *)tile_first, col_out, val_out, nr,
                  st, (int64_t)tiles, (unsigned)((uint64_t)nnz_in * sizeof(I)), (unsigned)((uint64_t)nnz_in * VB),
                  (unsigned)((uint64_t)total * sizeof(I)), (unsigned)((uint64_t)total * VB),
                  (unsigned)((uint64_t)m * sizeof(I)), *cdf);
    } else {
    SBX_KLAUNCH(h, SBX_K_PERMUTE_TILE, (k_permute_tile<I, VB>), dim3((unsigned)(tiles < tile_grid ? tiles : tile_grid)),
                dim3(PT_THREADS), rec, col_in, val_in, col_order, rpo, sp, (const I *)tile_first, col_out, val_out, nr,
                st, col_bits, permute_force_radix(), fb_tiles, (int64_t)tiles);
    SBX_KLAUNCH(h, SBX_K_PERMUTE_TILE, (k_permute_tile_radix<I, VB>), dim3((unsigned)(tiles < 2048 ? tiles : 2048)),
                dim3(PT_THREADS), rec, col_in, val_in, col_order, rpo, sp, (const I *)tile_first, col_out, val_out, nr,
                st, col_bits, (const unsigned *)fb_tiles);
    }
    SBX_LAUNCH_CHECK(h);
    SBX_PROF_BYTES(h, SBX_K_PERMUTE_TILE, short_nnz * (int64_t)(2 * (sizeof(I) + VB)));
    if ((permute_force_radix() >> 8) == 9) {  // diagnostic: print and clear the phase stamps
      unsigned long long hs_[32];
      SBX_HIP(h, hipStreamSynchronize(h->stream));
      SBX_HIP(h, hipMemcpyFromSymbol(hs_, HIP_SYMBOL(g_tile_stamps), sizeof(hs_)));
      if (hs_[31]) {
        fprintf(stderr, "tile stamps: tiles %llu avg entries %.0f | cycles to end of phase:", hs_[31],
                (double)hs_[30] / (double)hs_[31]);
        for (int i = 0; i < 16; i++) fprintf(stderr, " [%d] %.0f", i, (double)hs_[i] / (double)hs_[31]);
        fprintf(stderr, "\n");
      }
      memset(hs_, 0, sizeof(hs_));
      SBX_HIP(h, hipMemcpyToSymbol(HIP_SYMBOL(g_tile_stamps), hs_, sizeof(hs_)));
    }
  }
  if (has_block) {
    if (fork) h->stream = h->aux_stream[0];
    const int rc = block_rows_path<I, VB>(h, rec, col_in, val_in, col_order, rpo, col_out, val_out, m, block_rows, n_block,
                                       block_stride, block_nnz, st, fork);
    if (fork) {
      if (rc == SBX_OK && hipEventRecord(h->aux_event[1], h->stream) != hipSuccess) { h->stream = main_stream; SBX_FAIL(h, SBX_ERR_HIP, "hipEventRecord failed"); }
      h->stream = main_stream;
    }
    SBX_TRY(rc);
  }
  if (n_long) {
    if (fork) h->stream = h->aux_stream[1];
    const int rc = long_rows_path<I, VB>(h, rec, col_in, val_in, col_order, rpo, col_out, val_out, m, long_rows, n_long,
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

// The call's state, the classification's look-back words and the row records in ONE allocation, zeroed by ONE fill
// (three fills cost three launches of ~4.5 us in front of the first kernel).  status: for classify_and_scan.
static int64_t cs_tiles(int64_t nr) { return (nr + CS_TILE - 1) / CS_TILE > 0 ? (nr + CS_TILE - 1) / CS_TILE : 1; }
// The call's state, the classification's tile sums and the row records in ONE allocation, zeroed by ONE fill (three
// fills cost three launches of ~4.5 us in front of the first kernel; the records of rows a malformed row order never
// names must read as empty).  status: for classify_and_scan.
static int perm_prep_alloc(sbx_handle_t h, int64_t nr, PermState **st, unsigned long long **status, int2 **rec) {
  const size_t status_bytes = sizeof(unsigned long long) * (size_t)(2 * cs_tiles(nr) + 2);
  const size_t bytes = sizeof(PermAll) + status_bytes + sizeof(int2) * (size_t)nr;
  char *base = nullptr;
  SBX_TRY(sbx_salloc(h, bytes + 128, &base));
  base = (char *)(((uintptr_t)base + 127) & ~(uintptr_t)127);  // (PermAll keeps its counters on lines of their own)
  *st = &((PermAll *)base)->st;
  *status = (unsigned long long *)(base + sizeof(PermAll));
  *rec = (int2 *)(base + sizeof(PermAll) + status_bytes);
  SBX_HIP(h, hipMemsetAsync(base, 0, bytes, h->stream));
  return SBX_OK;
}
static int perm_state_zero(sbx_handle_t h, PermState *st) {
  SBX_HIP(h, hipMemsetAsync((PermAll *)st, 0, sizeof(PermAll), h->stream));
  return SBX_OK;
}
// the state of the call on the host, the classification's counters folded in
static int perm_fetch(sbx_handle_t h, PermState *hs, const PermState *st) {
  PermAll all;
  SBX_TRY(sbx_readback(h, &all, (const PermAll *)st, sizeof(PermAll)));
  *hs = all.st;
  for (int c = 0; c < BR_CLASSES; c++) hs->n_block[c] = all.hot[c * PH_STRIDE];
  hs->n_long = all.hot[PH_LONG * PH_STRIDE];
  memcpy(&hs->long_nnz, &all.hot[PH_LONG_NNZ * PH_STRIDE], sizeof(unsigned long long));
  memcpy(&hs->block_nnz, &all.hot[PH_BLOCK_NNZ * PH_STRIDE], sizeof(unsigned long long));
  return SBX_OK;
}

// rpo (may be nullptr: lengths already scanned), sp and the class lists (block_rows nullptr: none) in one launch;
// st must have been zeroed by the caller
template <typename I>
int classify_and_scan(sbx_handle_t h, const int2 *rec, I *rpo, I *sp, int64_t nr, I *long_rows, I *block_rows,
                      int64_t block_stride, int block_cap, PermState *st, unsigned long long *status = nullptr,
                      I **tile_first_out = nullptr, int64_t entries_upper = 0) {
  const int64_t tiles = cs_tiles(nr);
  if (!status) SBX_TRY(sbx_salloc(h, (size_t)(2 * tiles + 2), &status));  // (else: the words from perm_prep_alloc)
  I *tile_first = nullptr;
  if (tile_first_out) {  // the tile kernel's row ranges come out of the same pass (entries_upper >= the short rows' entries)
    const int64_t slots = entries_upper / PT_W + 3;
    SBX_TRY(sbx_salloc(h, (size_t)slots, &tile_first));
    SBX_KLAUNCH(h, SBX_K_PERMUTE_PREP, k_fill_index<I>, dim3(sbx_grid_for(slots, 256, 1024)), dim3(256), tile_first, slots,
                (I)nr);
    *tile_first_out = tile_first;
  }
  SBX_KLAUNCH(h, SBX_K_PERMUTE_PREP, k_classify_reduce, dim3((unsigned)tiles), dim3(256), rec, nr, status);
  SBX_KLAUNCH(h, SBX_K_PERMUTE_PREP, k_classify_prefix<I>, dim3(1), dim3(1024), status, tiles, rpo, sp, nr, st);
  SBX_KLAUNCH(h, SBX_K_PERMUTE_PREP, k_classify_scan<I>, dim3((unsigned)tiles), dim3(256), rec, rpo, sp, nr, long_rows,
              block_rows, block_stride, block_cap, st, (const unsigned long long *)status,
              (int)(((uintptr_t)rpo & 15) == 0), tile_first);
  SBX_LAUNCH_CHECK(h);
  return SBX_OK;
}

}  // namespace

#define SBX_REQUIRE(h, cond, msg)                                       \
  do {                                                                  \
    if (!(cond)) SBX_FAIL(h, SBX_ERR_BAD_ARG, "%s: %s", __func__, msg); \
  } while (0)

extern "C" int sbx_inverse_permutation(sbx_handle_t h, sbx_index_type it, int64_t n, const void *perm,
                                       void *inv_out) {
  if (!h) return SBX_ERR_BAD_ARG;
  if (it == SBX_I32_N64) it = SBX_I32;  // (no offset array)
  SBX_REQUIRE(h, n >= 0 && (n == 0 || (perm && inv_out)), "bad argument");
  SBX_TRY(sbx_arena_begin(h));
  if (n == 0) return SBX_OK;
  if (it == SBX_I64)  // (native: no narrowed copies)
    SBX_KLAUNCH(h, SBX_K_PERMUTE_PREP, k_invert<int64_t>, dim3(sbx_grid_for(n, 256, 8192)), dim3(256),
                (const int64_t *)perm, (int64_t *)inv_out, n);
  else
    SBX_KLAUNCH(h, SBX_K_PERMUTE_PREP, k_invert<int32_t>, dim3(sbx_grid_for(n, 256, 8192)), dim3(256),
                (const int32_t *)perm, (int32_t *)inv_out, n);
  SBX_LAUNCH_CHECK(h);
  return SBX_OK;
}

extern "C" int sbx_permute_array(sbx_handle_t h, sbx_index_type it, sbx_value_type vt, int64_t n, const void *order,
                                 const void *vals, void *out) {
  if (!h) return SBX_ERR_BAD_ARG;
  if (it == SBX_I32_N64) it = SBX_I32;  // (no offset array)
  SBX_REQUIRE(h, n >= 0 && (n == 0 || (order && vals && out)), "bad argument");
  const int vb = sbx_value_bytes(vt);
  SBX_REQUIRE(h, vb == 4 || vb == 8, "value type must be 4 or 8 bytes");
  SBX_TRY(sbx_arena_begin(h));
  if (n == 0) return SBX_OK;
  const unsigned grid = sbx_grid_for(n, 256, 8192);
  if (it == SBX_I64) {  // (native: no narrowed copy of the order)
    if (vb == 4)
      SBX_KLAUNCH(h, SBX_K_PERMUTE_PREP, (k_permute_array<int64_t, 4>), dim3(grid), dim3(256), (const int64_t *)order,
                  (const char *)vals, (char *)out, n);
    else
      SBX_KLAUNCH(h, SBX_K_PERMUTE_PREP, (k_permute_array<int64_t, 8>), dim3(grid), dim3(256), (const int64_t *)order,
                  (const char *)vals, (char *)out, n);
  } else if (vb == 4)
    SBX_KLAUNCH(h, SBX_K_PERMUTE_PREP, (k_permute_array<int32_t, 4>), dim3(grid), dim3(256), (const int32_t *)order,
                       (const char *)vals, (char *)out, n);
  else
    SBX_KLAUNCH(h, SBX_K_PERMUTE_PREP, (k_permute_array<int32_t, 8>), dim3(grid), dim3(256), (const int32_t *)order,
                       (const char *)vals, (char *)out, n);
  SBX_LAUNCH_CHECK(h);
  return SBX_OK;
}

// One shard of the permute for either index width.  The records (length, source offset) stay 32-bit — nnz < 2^31, checked
// by the caller — while row_ptr, the orders and the columns are read and written as they are: no narrowed copies.
template <typename I>
static int permute_csr_rows_typed(sbx_handle_t h, sbx_value_type vt, int64_t n, int64_t m, int64_t nnz, const void *row_ptr,
                                  const void *col, const void *val, const void *row_order, const void *col_order,
                                  int64_t row_begin, int64_t row_end, void *row_ptr_out, void *col_out, void *val_out,
                                  int64_t out_capacity, int64_t *shard_nnz_host) {
  const int vb = (val && val_out) ? sbx_value_bytes(vt) : 0;
  SBX_REQUIRE(h, vb >= 0, "unknown value type");
  SBX_TRY(sbx_arena_begin(h));
  const int64_t nr = row_end - row_begin;
  I *rpo = (I *)row_ptr_out;
  if (shard_nnz_host) *shard_nnz_host = 0;
  if (nr == 0) {
    if constexpr (sizeof(I) == 8) return sbx_fill_i64(h, (int64_t *)rpo, 0, 1);
    else return sbx_fill_i32(h, (int32_t *)rpo, 0, 1);
  }

  // (row length, source offset) per new row, written from the old-row side; lengths -> scan -> row_ptr_out
  PermState *st = nullptr;
  int2 *rec = nullptr;
  unsigned long long *status = nullptr;
  SBX_TRY(perm_prep_alloc(h, nr, &st, &status, &rec));
  // the key-distribution map of the relabelled columns (k_cdf_sample): needs nothing but the caller's arrays, so it is
  // built beside the preparation — on side stream 1, behind an event that orders it after the caller's earlier work
  CdfMap cdf;
  cdf.table = nullptr, cdf.shift = 0, cdf.fsh = 0;
  bool cdf_on_side = false;
  static const bool tile2_off = sbx_env_test("SBX_PERMUTE_NO_TILE2") != nullptr;  // (tests: the equal-width tile kernel)
  if (col_order && nnz > 0 && !tile2_off) {
    if (!h->prof_on && permute_overlap() && sbx_aux_streams(h) == SBX_OK) {
      hipStream_t main_stream = h->stream;
      SBX_HIP(h, hipEventRecord(h->aux_event[0], main_stream));
      SBX_HIP(h, hipStreamWaitEvent(h->aux_stream[1], h->aux_event[0], 0));
      h->aux_dirty = true;
      h->stream = h->aux_stream[1];
      const int rc_cdf = build_cdf_map<I>(h, (const I *)col, (const I *)col_order, nnz, m, &cdf);
      const hipError_t e_cdf = rc_cdf == SBX_OK ? hipEventRecord(h->aux_event[2], h->stream) : hipSuccess;
      h->stream = main_stream;
      SBX_TRY(rc_cdf);
      if (e_cdf != hipSuccess) SBX_FAIL(h, SBX_ERR_HIP, "hipEventRecord failed");
      cdf_on_side = true;
    } else {
      SBX_TRY(build_cdf_map<I>(h, (const I *)col, (const I *)col_order, nnz, m, &cdf));
    }
  }
  SBX_KLAUNCH(h, SBX_K_PERMUTE_PREP, k_rowwise_prep<I>, dim3(sbx_grid_for((n + 3) / 4, 256, 8192)), dim3(256),
              (const I *)row_ptr, (const I *)row_order, n, row_begin, nr, rec);
  I *long_rows = nullptr, *block_rows = nullptr, *sp = nullptr;
  const int block_cap = vb == 8 ? BlockRowCap<8>::value : BlockRowCap<4>::value;
  int64_t block_stride = 0;
  SBX_TRY(sbx_salloc(h, (size_t)nr + 1, &sp));
  if (col_order) {  // the sorting pipeline needs the rows that do not fit a tile listed by class
    int64_t cap_long = nnz / PT_LMAX + 1;
    if (cap_long > nr) cap_long = nr;
    SBX_TRY(sbx_salloc(h, (size_t)cap_long, &long_rows));
    SBX_TRY(sbx_salloc(h, (size_t)cap_long * BR_CLASSES, &block_rows));
    block_stride = cap_long;
  }
  // lengths -> row_ptr_out, the short rows' prefix sums and the class lists: one launch
  I *tile_first = nullptr;
  SBX_TRY(classify_and_scan<I>(h, (const int2 *)rec, rpo, sp, nr, long_rows, block_rows, block_stride, block_cap, st,
                               status, col_order ? &tile_first : nullptr, nnz));
  int64_t total = nnz;  // the full permute keeps every nonzero; a shard has to ask
  PermState hs;
  memset(&hs, 0, sizeof(hs));
  if (nr != n || col_order) {
    SBX_TRY(perm_fetch(h, &hs, st));
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
    int64_t cap_long = nnz / PT_LMAX + 1;
    if (cap_long > nr) cap_long = nr;
    SBX_TRY(sbx_salloc(h, (size_t)cap_long, &long_rows));
    SBX_TRY(sbx_salloc(h, (size_t)cap_long * BR_CLASSES, &block_rows));
    block_stride = cap_long;
    SBX_TRY(perm_state_zero(h, st));
    SBX_TRY(classify_and_scan<I>(h, (const int2 *)rec, (I *)nullptr, sp, nr, long_rows, block_rows, block_stride,
                                 block_cap, st));
    SBX_TRY(perm_fetch(h, &hs, st));
    hs.total = (unsigned long long)total;
  }
  int rc;
#define STAGE(VBX)                                                                                                  \
  rc = sort_stage<I, VBX>(h, vt, (const int2 *)rec, (const I *)col, (const char *)val, (const I *)col_order, rpo,      \
                       (I *)col_out, (char *)val_out, nr, m, total, long_rows, block_rows, block_stride, hs, st, sp, \
                       (const I *)tile_first, nnz, &cdf, cdf_on_side)
  if (vb == 0) STAGE(0);
  else if (vb == 4) STAGE(4);
  else STAGE(8);
#undef STAGE
  return rc;
}

extern "C" int sbx_permute_csr_rows(sbx_handle_t h, sbx_index_type it, sbx_value_type vt, int64_t n, int64_t m,
                                    int64_t nnz, const void *row_ptr, const void *col, const void *val,
                                    const void *row_order, const void *col_order, int64_t row_begin, int64_t row_end,
                                    void *row_ptr_out, void *col_out, void *val_out, int64_t out_capacity,
                                    int64_t *shard_nnz_host) {
  if (!h) return SBX_ERR_BAD_ARG;
  if (it == SBX_I32_N64)
    return sbx_mixed_permute_csr_rows(h, vt, n, m, nnz, row_ptr, col, val, row_order, col_order, row_begin, row_end,
                                      row_ptr_out, col_out, val_out, out_capacity, shard_nnz_host);
  SBX_REQUIRE(h, n >= 0 && m >= 0 && nnz >= 0 && row_ptr && row_ptr_out, "bad argument");
  SBX_REQUIRE(h, 0 <= row_begin && row_begin <= row_end && row_end <= n, "bad row range");
  SBX_REQUIRE(h, nnz == 0 || (col && col_out), "col/col_out required");
  SBX_REQUIRE(h, nnz < ((int64_t)1 << 31) && n < ((int64_t)1 << 31) - 1, "dimension exceeds int32");
  if (it == SBX_I64)
    return permute_csr_rows_typed<int64_t>(h, vt, n, m, nnz, row_ptr, col, val, row_order, col_order, row_begin, row_end,
                                           row_ptr_out, col_out, val_out, out_capacity, shard_nnz_host);
  return permute_csr_rows_typed<int32_t>(h, vt, n, m, nnz, row_ptr, col, val, row_order, col_order, row_begin, row_end,
                                         row_ptr_out, col_out, val_out, out_capacity, shard_nnz_host);
}

extern "C" int sbx_permute_csr(sbx_handle_t h, sbx_index_type it, sbx_value_type vt, int64_t n, int64_t m, int64_t nnz,
                               const void *row_ptr, const void *col, const void *val, const void *row_order,
                               const void *col_order, void *row_ptr_out, void *col_out, void *val_out) {
  return sbx_permute_csr_rows(h, it, vt, n, m, nnz, row_ptr, col, val, row_order, col_order, 0, n, row_ptr_out,
                              col_out, val_out, nnz, nullptr);
}

// A4: the CSR constructor's "if any row is unsorted, sort every row" in place.
// Stable sort of every segment of (key, value) pairs by its 32-bit key, out of place: the sort stage of the permute
// with identity maps — segments are "rows", keys "columns" below `key_limit` — and without the reference's value
// ordering of duplicate columns (a CSR-constructor rule): equal keys keep their input order (tiles and row classes
// order (key, original position); long segments take the stable radix path).  The hybrid COO sort's second half
// (sbx_convert.hip).  The caller has begun the arena.
int sbx_sort_segments(sbx_handle_t h, int vb, int64_t nseg, int64_t key_limit, int64_t nnz, const int32_t *seg_ptr,
                      const int32_t *key_in, const char *val_in, int32_t *key_out, char *val_out) {
  typedef int32_t I;
  PermState *st = nullptr;
  I *long_rows = nullptr, *block_rows = nullptr;
  int2 *rec = nullptr;
  const int block_cap = vb == 8 ? BlockRowCap<8>::value : BlockRowCap<4>::value;
  unsigned long long *status = nullptr;
  SBX_TRY(perm_prep_alloc(h, nseg, &st, &status, &rec));
  SBX_TRY(sbx_salloc(h, (size_t)nseg, &long_rows));
  SBX_TRY(sbx_salloc(h, (size_t)nseg * BR_CLASSES, &block_rows));
  SBX_KLAUNCH(h, SBX_K_PERMUTE_PREP, k_rowwise_prep<I>, dim3(sbx_grid_for((nseg + 3) / 4, 256, 8192)), dim3(256), seg_ptr,
              (const I *)nullptr, nseg, (int64_t)0, nseg, rec);
  I *sp = nullptr;
  SBX_TRY(sbx_salloc(h, (size_t)nseg + 1, &sp));
  SBX_TRY(classify_and_scan<I>(h, (const int2 *)rec, (I *)nullptr, sp, nseg, long_rows, block_rows, nseg, block_cap, st,
                               status));
  PermState hs;
  SBX_TRY(perm_fetch(h, &hs, st));
  // (SBX_V_NONE: no value ordering of equal keys behind the sort)
  if (vb == 0)
    return sort_stage<I, 0>(h, SBX_V_NONE, (const int2 *)rec, key_in, val_in, (const I *)nullptr, seg_ptr, key_out, val_out,
                         nseg, key_limit, nnz, long_rows, block_rows, nseg, hs, st, sp);
  if (vb == 4)
    return sort_stage<I, 4>(h, SBX_V_NONE, (const int2 *)rec, key_in, val_in, (const I *)nullptr, seg_ptr, key_out, val_out,
                         nseg, key_limit, nnz, long_rows, block_rows, nseg, hs, st, sp);
  return sort_stage<I, 8>(h, SBX_V_NONE, (const int2 *)rec, key_in, val_in, (const I *)nullptr, seg_ptr, key_out, val_out,
                       nseg, key_limit, nnz, long_rows, block_rows, nseg, hs, st, sp);
}

template <typename I>
static int csr_sort_rows_typed(sbx_handle_t h, sbx_value_type vt, int64_t n, int64_t m, int64_t nnz, const void *row_ptr,
                               void *col, void *val) {
  if (nnz <= 1 || n == 0) return SBX_OK;
  int sorted = 1;
  SBX_TRY(sbx_csr_rows_sorted(h, sizeof(I) == 8 ? SBX_I64 : SBX_I32, n, row_ptr, col, &sorted));  // csr.cc:102-116
  if (sorted) return SBX_OK;
  // out-of-place into scratch, then copied back (tiles may read rows another
  // tile has already rewritten if the sort ran in place across tiles)
  const int vb = val ? sbx_value_bytes(vt) : 0;
  SBX_REQUIRE(h, vb >= 0, "unknown value type");
  SBX_TRY(sbx_arena_begin(h));
  PermState *st = nullptr;
  I *long_rows = nullptr, *block_rows = nullptr, *ctmp = nullptr;
  int2 *rec = nullptr;
  const int block_cap = vb == 8 ? BlockRowCap<8>::value : BlockRowCap<4>::value;
  char *vtmp = nullptr;
  unsigned long long *status = nullptr;
  SBX_TRY(perm_prep_alloc(h, n, &st, &status, &rec));
  SBX_TRY(sbx_salloc(h, (size_t)n, &long_rows));
  SBX_TRY(sbx_salloc(h, (size_t)n * BR_CLASSES, &block_rows));
  SBX_TRY(sbx_salloc(h, (size_t)nnz, &ctmp));
  if (vb) SBX_TRY(sbx_salloc(h, (size_t)nnz * vb, &vtmp));
  SBX_KLAUNCH(h, SBX_K_PERMUTE_PREP, k_rowwise_prep<I>, dim3(sbx_grid_for((n + 3) / 4, 256, 8192)), dim3(256),
              (const I *)row_ptr, (const I *)nullptr, n, (int64_t)0, n, rec);
  I *sp = nullptr;
  SBX_TRY(sbx_salloc(h, (size_t)n + 1, &sp));
  SBX_TRY(classify_and_scan<I>(h, (const int2 *)rec, (I *)nullptr, sp, n, long_rows, block_rows, (int64_t)n, block_cap, st,
                               status));
  PermState hs;
  SBX_TRY(perm_fetch(h, &hs, st));
  int rc;
#define STAGE(VBX)                                                                                               \
  rc = sort_stage<I, VBX>(h, vt, (const int2 *)rec, (const I *)col, (const char *)val, (const I *)nullptr,          \
                       (const I *)row_ptr, ctmp, vtmp, n, m, nnz, long_rows, block_rows, (int64_t)n, hs, st, sp)
  if (vb == 0) STAGE(0);
  else if (vb == 4) STAGE(4);
  else STAGE(8);
#undef STAGE
  SBX_TRY(rc);
  SBX_HIP(h, hipMemcpyAsync(col, ctmp, (size_t)nnz * sizeof(I), hipMemcpyDeviceToDevice, h->stream));
  if (vb) SBX_HIP(h, hipMemcpyAsync(val, vtmp, (size_t)nnz * vb, hipMemcpyDeviceToDevice, h->stream));
  return SBX_OK;
}

extern "C" int sbx_csr_sort_rows(sbx_handle_t h, sbx_index_type it, sbx_value_type vt, int64_t n, int64_t m,
                                 int64_t nnz, const void *row_ptr, void *col, void *val) {
  if (!h) return SBX_ERR_BAD_ARG;
  if (it == SBX_I32_N64) return sbx_mixed_csr_sort_rows(h, vt, n, m, nnz, row_ptr, col, val);
  SBX_REQUIRE(h, n >= 0 && nnz >= 0 && row_ptr && (nnz == 0 || col), "bad argument");
  SBX_REQUIRE(h, nnz < ((int64_t)1 << 31) && n < ((int64_t)1 << 31) - 1, "dimension exceeds int32");
  if (it == SBX_I64) return csr_sort_rows_typed<int64_t>(h, vt, n, m, nnz, row_ptr, col, val);
  return csr_sort_rows_typed<int32_t>(h, vt, n, m, nnz, row_ptr, col, val);
}
