// sbx_degree.hip — DegreeReorder::CalculateReorderCSR (reorder/degree_reorder.cc:22-62).
//
// The reference is a counting sort that fills each degree bucket from its END in row-id order, i.e. the sequence
// (degree ascending, id descending); "descending" reverses the whole sequence.
//
// Degrees are small numbers with a thin tail (a power-law graph: half the rows empty, 98 % below 255 entries), so the
// rows are taken in descending id order and placed by ONE stable counting pass on min(degree, 255) — that already is
// the final position of every row below 255 entries.  The pass is three kernels that never materialise a key: every
// workgroup owns a tile of 4096 consecutive rows and counts their digits (k_degree_count), one workgroup per digit
// scans the tiles' counts (k_degree_scan), and the tiles walk their rows again, rank each among the wave's equal
// digits with ballots and write inv[id] = position — a coalesced store, the rows come in id order (k_degree_place).
// Two reads of row_ptr and one write of the result: 12n bytes of traffic, 16 + 5 + 17 us for 4.2 M rows (as 64-bit
// keys through the generic radix sort — keys written, histogram, one digit pass — the same pass took 95).  The rows
// of the last bucket leave k_degree_place as (degree, id) pairs in id order; they are sorted among themselves by
// their full degree with the same three kernels per 9-bit digit, the last placement writing inv[id] (76 K rows, two
// digits: 6 launches of 3 - 7 us; the generic sort's four launches took 46).  One read-back (rows in the last bucket,
// largest degree) sits behind the big pass.  8n + 4 algorithmic bytes; n = 4.2 M: 0.075 ms.  Templated on the index
// type: 64-bit row_ptr arrays are read as they are and the inverse permutation is written in 64 bits.
#include "sbx_device.h"
#include "sbx_internal.h"
#include "sbx_countsort.h"

namespace {

constexpr unsigned DG_TOP = 255;   // digit of every row with at least this many entries
constexpr int DG_BINS = 256;
constexpr int DG_WAVE_ROWS = 1024;  // rows per wave (16 rounds of 64), the unit of the counting pass
constexpr int DG_ROUNDS = DG_WAVE_ROWS / 64;

struct DegState {
  unsigned n_top;    // rows with degree >= DG_TOP
  unsigned max_deg;
};

// the lanes of the wave that hold the same 8-bit digit as this one (valid lanes only)
__device__ __forceinline__ unsigned long long dg_peers(unsigned d, bool valid) {
  unsigned long long peers = __ballot(valid);
#pragma unroll
  for (int b = 0; b < 8; b++) {
    const bool bit = (d >> b) & 1u;
    const unsigned long long m = __ballot(bit);
    peers &= bit ? m : ~m;
  }
  return peers;
}

// Workgroup t of the launch owns the tile of rows j = t * 4096 .. + 4095 of the descending-id order (id = n - 1 - j), a
// wave 1024 consecutive ones (16 rounds of 64, all their row_ptr loads issued before the first is used): the counts of
// the tile's digits go to cnt[digit * n_tiles + t], its largest degree to tmax[t].
// (IDASC: the walk goes through the rows in ascending id order instead — the degree ranks of the RCM, sbx_degree_ranks)
template <typename I, bool IDASC>
__device__ __forceinline__ void dg_load_degrees(const I *__restrict__ rp, int64_t n, int64_t j0, unsigned (&deg)[DG_ROUNDS]) {
#pragma unroll
  for (int r = 0; r < DG_ROUNDS; r++) {
    const int64_t j = j0 + (int64_t)r * 64;
    const int64_t id = j < n ? (IDASC ? j : n - 1 - j) : 0;
    deg[r] = (unsigned)(rp[id + 1] - rp[id]);
  }
}

template <typename I, bool IDASC = false>
__global__ __launch_bounds__(256) void k_degree_count(const I *__restrict__ rp, int64_t n, int64_t n_tiles,
                                                      unsigned *__restrict__ cnt, unsigned *__restrict__ tmax) {
  __shared__ unsigned s_hist[DG_BINS];
  __shared__ unsigned s_mx[4];
  const int lane = sbx_lane(), wv = sbx_wave_in_block();
  s_hist[threadIdx.x] = 0;
  const int64_t j0 = ((int64_t)blockIdx.x * 4 + wv) * DG_WAVE_ROWS + lane;
  unsigned deg[DG_ROUNDS];
  dg_load_degrees<I, IDASC>(rp, n, j0, deg);
  __syncthreads();
  unsigned mx = 0;
#pragma unroll
  for (int r = 0; r < DG_ROUNDS; r++) {
    const bool valid = j0 + (int64_t)r * 64 < n;
    const unsigned d = deg[r] < DG_TOP ? deg[r] : DG_TOP;
    const unsigned long long peers = dg_peers(d, valid);
    if (valid) {
      mx = deg[r] > mx ? deg[r] : mx;
      if ((peers & sbx_lanemask_lt()) == 0) atomicAdd(&s_hist[d], (unsigned)__popcll(peers));  // the digit's first lane
    }
  }
  mx = sbx_wave_max(mx);
  if (lane == 0) s_mx[wv] = mx;
  __syncthreads();
  cnt[(int64_t)threadIdx.x * n_tiles + blockIdx.x] = s_hist[threadIdx.x];
  if (threadIdx.x == 0) {
    for (int i = 1; i < 4; i++) mx = s_mx[i] > mx ? s_mx[i] : mx;
    tmax[blockIdx.x] = mx;
  }
}

// workgroup d: cnt[d * n_waves + w] -> the number of rows with digit d in the units (tiles of the first pass, waves of
// the tail's passes) before w; total[d]; the first pass (wmax != nullptr) also leaves the state: rows in the last
// bucket, largest degree
__global__ __launch_bounds__(256) void k_degree_scan(unsigned *__restrict__ cnt, const unsigned *__restrict__ wmax,
                                                     int64_t n_waves, unsigned *__restrict__ total,
                                                     DegState *__restrict__ st) {
  __shared__ unsigned s_scan[8];
  unsigned *row = cnt + (int64_t)blockIdx.x * n_waves;
  const bool vec = (n_waves & 3) == 0;  // rows of counts start 16-byte aligned: four counts per access
  int64_t per = (n_waves + 255) / 256;
  if (vec) per = (per + 3) & ~(int64_t)3;
  const int64_t lo = (int64_t)threadIdx.x * per < n_waves ? (int64_t)threadIdx.x * per : n_waves;
  const int64_t hi = lo + per < n_waves ? lo + per : n_waves;
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
  if (threadIdx.x == 0) {
    total[blockIdx.x] = tot;
    if (wmax && blockIdx.x == DG_TOP) st->n_top = tot;
  }
  if (wmax && blockIdx.x == 0) {
    unsigned mx = 0;
    for (int64_t i = threadIdx.x; i < n_waves; i += 256) mx = wmax[i] > mx ? wmax[i] : mx;
    mx = sbx_wave_max(mx);
    __syncthreads();
    if (sbx_lane() == 0) s_scan[sbx_wave_in_block()] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int i = 1; i < 4; i++) mx = s_scan[i] > mx ? s_scan[i] : mx;
      st->max_deg = mx;
    }
  }
}

// the rows again: position = rows of smaller digits + rows of this digit in the tiles before + in the tile's waves
// before + in this wave before it; inv[id] = position (or its mirror); rows of the last bucket also leave (degree, id)
// at their place in it
// (RANKS: the degree ranks of the RCM — rows in ascending id order, empty rows left out: rank = position - empty rows,
// written both ways, inv[id] = rank and order[rank] = id)
template <typename I, bool RANKS = false, typename O = I>
__global__ __launch_bounds__(256) void k_degree_place(const I *__restrict__ rp, int64_t n, int64_t n_tiles,
                                                      const unsigned *__restrict__ off, const unsigned *__restrict__ total,
                                                      O *__restrict__ inv, uint32_t *__restrict__ tail_key,
                                                      uint32_t *__restrict__ tail_id, int ascending,
                                                      uint32_t *__restrict__ order) {
  __shared__ unsigned s_scan[8];
  __shared__ unsigned s_wave[4][DG_BINS];  // rows of digit d in wave w; then the position of the wave's first such row
  __shared__ unsigned s_top_base;
  const int lane = sbx_lane(), wv = sbx_wave_in_block();
  const int64_t j0 = ((int64_t)blockIdx.x * 4 + wv) * DG_WAVE_ROWS + lane;
  unsigned deg[DG_ROUNDS];
  dg_load_degrees<I, RANKS>(rp, n, j0, deg);
  const unsigned tile_off = off[(int64_t)threadIdx.x * n_tiles + blockIdx.x];  // thread = digit
#pragma unroll
  for (int k = 0; k < 4; k++) s_wave[k][threadIdx.x] = 0;
  unsigned tot;
  const unsigned dbase = sbx_block_exclusive_sum<unsigned, 256>(total[threadIdx.x], s_scan, &tot);  // (two barriers)
  if (threadIdx.x == DG_TOP) s_top_base = dbase;
  unsigned loc[DG_ROUNDS];  // the row's rank among the wave's rows of its digit
#pragma unroll
  for (int r = 0; r < DG_ROUNDS; r++) {
    const bool valid = j0 + (int64_t)r * 64 < n;
    const unsigned d = deg[r] < DG_TOP ? deg[r] : DG_TOP;
    const unsigned long long peers = dg_peers(d, valid);
    const unsigned rank = (unsigned)__popcll(peers & sbx_lanemask_lt());
    unsigned first = 0;
    if (valid && rank == 0) first = atomicAdd(&s_wave[wv][d], (unsigned)__popcll(peers));
    first = (unsigned)__shfl((int)first, valid ? (int)__builtin_ctzll(peers) : 0, 64);
    loc[r] = first + rank;
  }
  __syncthreads();
  {
    unsigned run = dbase + tile_off;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const unsigned c = s_wave[k][threadIdx.x];
      s_wave[k][threadIdx.x] = run;
      run += c;
    }
  }
  __syncthreads();
  const unsigned top_base = s_top_base;
  const unsigned n_empty = RANKS ? total[0] : 0u;
#pragma unroll
  for (int r = 0; r < DG_ROUNDS; r++) {
    const int64_t j = j0 + (int64_t)r * 64;
    if (j < n) {
      const int64_t id = RANKS ? j : n - 1 - j;
      const unsigned d = deg[r] < DG_TOP ? deg[r] : DG_TOP;
      const unsigned pos = s_wave[wv][d] + loc[r];
      if (RANKS) {
        if (d != 0 && d != DG_TOP) {  // (the last bucket's rows get theirs from the tail sort)
          inv[id] = (O)(pos - n_empty);
          order[pos - n_empty] = (uint32_t)id;
        }
      } else {
        inv[id] = (O)(ascending ? (int64_t)pos : n - 1 - (int64_t)pos);
      }
      if (d == DG_TOP) {
        tail_key[pos - top_base] = deg[r];
        tail_id[pos - top_base] = (uint32_t)id;
      }
    }
  }
}

// ---- the last bucket: a stable counting sort of its (degree, id) pairs by the full degree (sbx_countsort.h: three
// small launches per 9-bit digit; a few ten thousand rows, where a digit pass of the generic sort — a chained scan
// over its tiles — took 11 us); the last digit's placement writes inv[id] itself.
template <typename I>
struct DegreeTailEmit {
  I *inv;
  int64_t first, n;  // the bucket's first position; the dimension
  int ascending;
  __device__ void operator()(unsigned pos, uint32_t, uint32_t id) const {
    inv[id] = (I)(ascending ? first + (int64_t)pos : n - 1 - (first + (int64_t)pos));
  }
};

template <typename I>
__global__ __launch_bounds__(256) void k_degree_tail_emit(const uint32_t *__restrict__ sorted_id, I *__restrict__ inv,
                                                          int64_t count, int64_t first, int64_t n, int ascending) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; t < count; t += stride) inv[sorted_id[t]] = (I)(ascending ? first + t : n - 1 - (first + t));
}

template <typename I>
__global__ void k_degree_one(I *inv) { inv[0] = 0; }

}  // namespace

// I: row_ptr's word; P: the word of the inverse permutation (SBX_I32_N64: 64-bit offsets, 32-bit ids)
template <typename I, typename P = I>
static int degree_reorder_typed(sbx_handle_t h, int64_t n, const void *row_ptr, int ascending, void *inv_perm_out) {
  SBX_TRY(sbx_arena_begin(h));
  if (n == 0) return SBX_OK;
  if (n == 1) {  // (the radix sort wants two keys)
    SBX_KLAUNCH(h, SBX_K_DEGREE, k_degree_one<P>, dim3(1), dim3(1), (P *)inv_perm_out);
    SBX_LAUNCH_CHECK(h);
    return SBX_OK;
  }
  const I *rp = (const I *)row_ptr;
  const int64_t n_tiles = (n + 4 * DG_WAVE_ROWS - 1) / (4 * DG_WAVE_ROWS);
  unsigned *cnt, *wmax, *total;
  uint32_t *ta, *ia;
  DegState *st;
  SBX_TRY(sbx_salloc(h, (size_t)n_tiles * DG_BINS, &cnt));
  SBX_TRY(sbx_salloc(h, (size_t)n_tiles, &wmax));
  SBX_TRY(sbx_salloc(h, (size_t)DG_BINS, &total));
  SBX_TRY(sbx_salloc(h, (size_t)n, &ta));  // the last bucket's (degree, id), in id order
  SBX_TRY(sbx_salloc(h, (size_t)n, &ia));
  SBX_TRY(sbx_salloc(h, 1, &st));
  const unsigned grid = (unsigned)n_tiles;
  SBX_KLAUNCH(h, SBX_K_DEGREE, k_degree_count<I>, dim3(grid), dim3(256), rp, n, n_tiles, cnt, wmax);
  SBX_KLAUNCH(h, SBX_K_DEGREE, k_degree_scan, dim3(DG_BINS), dim3(256), cnt, (const unsigned *)wmax, n_tiles, total, st);
  SBX_KLAUNCH(h, SBX_K_DEGREE, (k_degree_place<I, false, P>), dim3(grid), dim3(256), rp, n, n_tiles, (const unsigned *)cnt,
              (const unsigned *)total, (P *)inv_perm_out, ta, ia, ascending, (uint32_t *)nullptr);
  SBX_LAUNCH_CHECK(h);
  DegState hs;
  SBX_TRY(sbx_readback(h, &hs, st, sizeof(hs)));
  const int64_t top = hs.n_top;
  if (top < 2) return SBX_OK;
  // the last bucket by full degree (stable: equal degrees keep their descending id order)
  uint32_t *tb, *ib;
  SBX_TRY(sbx_salloc(h, (size_t)top, &tb));
  SBX_TRY(sbx_salloc(h, (size_t)top, &ib));
  if (top <= sbx_cs::MAX_PAIRS) {
    const DegreeTailEmit<P> emit = {(P *)inv_perm_out, n - top, n, ascending};
    SBX_TRY(sbx_cs::sort_emit(h, SBX_K_DEGREE, ta, ia, tb, ib, top, sbx_bits_for(hs.max_deg), emit));
  } else {  // (millions of rows of 255+ entries: the generic sort's staged stores win; then a scatter)
    sbx_radix_pass passes[16];
    const int np = sbx_radix_plan(0, sbx_bits_for(hs.max_deg), 0, 0, passes);
    int in_b = 0;
    SBX_TRY(sbx_radix_sort(h, 4, 4, ta, tb, ia, ib, top, passes, np, &in_b));
    SBX_KLAUNCH(h, SBX_K_DEGREE, k_degree_tail_emit<P>, dim3(sbx_grid_for(top, 256, 2048)), dim3(256),
                (const uint32_t *)(in_b ? ib : ia), (P *)inv_perm_out, top, n - top, n, ascending);
    SBX_LAUNCH_CHECK(h);
  }
  return SBX_OK;
}

// Degree ranks of the vertices with a non-empty row, for the Cuthill-McKee keys of sbx_rcm.hip (rcm_reorder.cc:101-118
// orders a vertex's children by degree; equal degrees keep ascending ids): rank[v] = position of v in the sequence
// (degree ascending, id ascending) of the non-empty rows, order[rank] = v.  The counting pass above with the rows walked
// in ascending id order — three launches and, for the rows of 255 entries and more, three per 9-bit digit — instead of a
// generic radix sort of (degree, id) pairs (keys written, histogram, three digit passes: 0.31 ms for 4.2 M rows against
// 0.09).  The caller knows how many rows the last bucket holds (n_top) and the largest degree: nothing is read back; the
// scratch comes from the arena of the call that is running; everything is enqueued on h->stream.
namespace {
struct RankTailEmit {
  uint32_t *rank, *order;
  uint32_t first;  // rank of the last bucket's first row
  __device__ void operator()(unsigned pos, uint32_t, uint32_t id) const {
    rank[id] = first + pos;
    order[first + pos] = id;
  }
};
__global__ __launch_bounds__(256) void k_rank_tail_emit(const uint32_t *__restrict__ sorted_id, uint32_t *__restrict__ rank,
                                                        uint32_t *__restrict__ order, int64_t count, uint32_t first) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; t < count; t += stride) {
    rank[sorted_id[t]] = first + (uint32_t)t;
    order[first + t] = sorted_id[t];
  }
}
}  // namespace

template <typename I>
static int degree_ranks_typed(sbx_handle_t h, const I *rp, int64_t n, int64_t n_nonempty, int64_t n_top, unsigned max_deg,
                              uint32_t *rank, uint32_t *order) {
  if (n_nonempty <= 0) return SBX_OK;
  const int64_t n_tiles = (n + 4 * DG_WAVE_ROWS - 1) / (4 * DG_WAVE_ROWS);
  unsigned *cnt, *wmax, *total;
  uint32_t *ta, *ia, *tb, *ib;
  DegState *st;
  SBX_TRY(sbx_salloc(h, (size_t)n_tiles * DG_BINS, &cnt));
  SBX_TRY(sbx_salloc(h, (size_t)n_tiles, &wmax));
  SBX_TRY(sbx_salloc(h, (size_t)DG_BINS, &total));
  SBX_TRY(sbx_salloc(h, (size_t)n_top + 1, &ta));
  SBX_TRY(sbx_salloc(h, (size_t)n_top + 1, &ia));
  SBX_TRY(sbx_salloc(h, (size_t)n_top + 1, &tb));
  SBX_TRY(sbx_salloc(h, (size_t)n_top + 1, &ib));
  SBX_TRY(sbx_salloc(h, 1, &st));
  const unsigned grid = (unsigned)n_tiles;
  SBX_KLAUNCH(h, SBX_K_RCM_MISC, (k_degree_count<I, true>), dim3(grid), dim3(256), rp, n, n_tiles, cnt, wmax);
  SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_degree_scan, dim3(DG_BINS), dim3(256), cnt, (const unsigned *)wmax, n_tiles, total, st);
  SBX_KLAUNCH(h, SBX_K_RCM_MISC, (k_degree_place<I, true, uint32_t>), dim3(grid), dim3(256), rp, n, n_tiles,
              (const unsigned *)cnt, (const unsigned *)total, rank, ta, ia, 1, order);
  SBX_LAUNCH_CHECK(h);
  if (n_top == 0) return SBX_OK;
  const uint32_t first = (uint32_t)(n_nonempty - n_top);
  if (n_top == 1 || n_top <= sbx_cs::MAX_PAIRS) {
    const RankTailEmit emit = {rank, order, first};
    SBX_TRY(sbx_cs::sort_emit(h, SBX_K_RCM_MISC, ta, ia, tb, ib, n_top, sbx_bits_for(max_deg), emit));
  } else {
    sbx_radix_pass passes[16];
    const int np = sbx_radix_plan(0, sbx_bits_for(max_deg), 0, 0, passes);
    int in_b = 0;
    SBX_TRY(sbx_radix_sort(h, 4, 4, ta, tb, ia, ib, n_top, passes, np, &in_b));
    SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_rank_tail_emit, dim3(sbx_grid_for(n_top, 256, 2048)), dim3(256),
                (const uint32_t *)(in_b ? ib : ia), rank, order, n_top, first);
    SBX_LAUNCH_CHECK(h);
  }
  return SBX_OK;
}

// rank[id] / order[rank] of the non-empty rows by (degree, id): the degree ranks of the RCM (row_ptr of either width)
int sbx_degree_ranks(sbx_handle_t h, sbx_index_type it, const void *rp, int64_t n, int64_t n_nonempty, int64_t n_top,
                     unsigned max_deg, uint32_t *rank, uint32_t *order) {
  return it != SBX_I32 ? degree_ranks_typed<int64_t>(h, (const int64_t *)rp, n, n_nonempty, n_top, max_deg, rank, order)
                       : degree_ranks_typed<int32_t>(h, (const int32_t *)rp, n, n_nonempty, n_top, max_deg, rank, order);
}

extern "C" int sbx_degree_reorder(sbx_handle_t h, sbx_index_type it, int64_t n, const void *row_ptr, int ascending,
                                  void *inv_perm_out) {
  if (!h) return SBX_ERR_BAD_ARG;
  if (n < 0 || !row_ptr || (n > 0 && !inv_perm_out)) SBX_FAIL(h, SBX_ERR_BAD_ARG, "sbx_degree_reorder: bad argument");
  // (vertex ids travel in the low word of the sort keys and degrees in 32 bits: n and every row length below 2^32;
  // row_ptr VALUES — nnz — are not limited for 64-bit indices)
  if (n >= ((int64_t)1 << 31)) SBX_FAIL(h, SBX_ERR_UNSUPPORTED, "sbx_degree_reorder: dimension exceeds int32");
  if (it == SBX_I32_N64) return degree_reorder_typed<int64_t, int32_t>(h, n, row_ptr, ascending, inv_perm_out);
  return it == SBX_I64 ? degree_reorder_typed<int64_t>(h, n, row_ptr, ascending, inv_perm_out)
                       : degree_reorder_typed<int32_t>(h, n, row_ptr, ascending, inv_perm_out);
}
