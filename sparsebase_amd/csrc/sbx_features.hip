// sbx_features.hip — the reorder-quality features of SURVEY §8(f).2: what a reordering
// bought, computed where the permuted CSR already lives.
//
//   feature/degrees.cc:93-105               sbx_csr_degrees
//   feature/degree_distribution.cc:152-167  sbx_csr_degree_distribution
//   feature/bandwidth.cc:93-112             sbx_csr_bandwidth
//   feature/profile.cc:91-105               sbx_csr_profile
//
// All four are single-pass reductions.  Bandwidth and profile are nonzero-parallel so
// power-law rows stay balanced: the row of every nonzero is derived per tile from row_ptr
// (tile_rows), never from a per-row loop.
#include <type_traits>

#include "sbx_device.h"
#include "sbx_internal.h"

namespace {

constexpr int FT_THREADS = 256;
constexpr int FT_ITEMS = 8;

template <typename I, typename D = I>  // (D: the degrees' word — an id type: SBX_I32_N64 reads 64-bit offsets, writes 32-bit degrees)
__global__ __launch_bounds__(FT_THREADS) void k_degrees(const I *__restrict__ rp, D *__restrict__ out, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = (D)(rp[i + 1] - rp[i]);
}

// dist[i] = degree / (FeatureType)num_edges, one IEEE division per row (degree_distribution.cc:163)
template <typename I, typename F>
__global__ __launch_bounds__(FT_THREADS) void k_degree_distribution(const I *__restrict__ rp, F *__restrict__ out,
                                                                    int64_t n, F nnz) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = (F)(rp[i + 1] - rp[i]) / nnz;
}

struct FeatureAcc {
  unsigned long long max_dist;  // max |row - col| over the nonzeros
  unsigned long long profile;   // sum over rows of (row - min(row, smallest column))
};

// ---- one pass over the CSR itself, nonzero-parallel -------------------------------------------------------------
// A workgroup takes FT_TILE consecutive nonzeros: the rows the tile touches come from two wave-wide binary searches
// in row_ptr, their first positions are marked in LDS (an empty row shares its position with the next non-empty one,
// which wins the atomicMax) and a max-scan gives every nonzero its row — the CSR -> COO expansion without the 4 bytes
// per nonzero written and read again.
constexpr int FT_TILE = FT_THREADS * FT_ITEMS;
constexpr int FT_SLOTS = 1024;  // result words of the tile kernels (zeroed by the caller, reduced by k_feature_finish)

struct TileRows {
  int64_t r_lo;         // row of the tile's first nonzero
  int h[FT_ITEMS];      // row - r_lo of this thread's FT_ITEMS consecutive nonzeros
  bool head[FT_ITEMS];  // the nonzero is the first one of its row
};

// first and last row of every tile, one thread per tile: the two searches in row_ptr are four dependent rounds of
// loads when a workgroup does them for itself, and 25 waves of workgroups per CU then spend half their lives in them
template <typename I>
__global__ __launch_bounds__(FT_THREADS) void k_tile_spans(const I *__restrict__ rp, int64_t n, int64_t nnz,
                                                           int64_t tiles, int2 *__restrict__ span) {
  const int64_t t = (int64_t)blockIdx.x * FT_THREADS + threadIdx.x;
  if (t >= tiles) return;
  const int64_t t0 = t * FT_TILE, t1 = (t0 + FT_TILE < nnz) ? t0 + FT_TILE : nnz;
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

template <typename I>
__device__ __forceinline__ TileRows tile_rows(const I *__restrict__ rp, const int2 *__restrict__ span, int64_t tile,
                                              int64_t t0, int *s_head, int *s_wmax) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int k = 0; k < FT_ITEMS; k++) s_head[k * FT_THREADS + tid] = 0;
  const int2 sp = span[tile];
  __syncthreads();
  TileRows t;
  t.r_lo = sp.x;
  const int64_t r_hi = sp.y;
  for (int64_t r = t.r_lo + 1 + tid; r <= r_hi; r += FT_THREADS) atomicMax(&s_head[(int)((int64_t)rp[r] - t0)], (int)(r - t.r_lo));
  __syncthreads();
  const bool first_is_head = (int64_t)rp[t.r_lo] == t0;
  int run = 0;
#pragma unroll
  for (int k = 0; k < FT_ITEMS; k++) {
    const int v = s_head[tid * FT_ITEMS + k];
    t.head[k] = v != 0 || (first_is_head && tid == 0 && k == 0);
    run = v > run ? v : run;
    t.h[k] = run;
  }
  const int inc = sbx_wave_inclusive_max(run);
  const int excl = sbx_wave_shift_up1(inc, 0);
  if (sbx_lane() == 63) s_wmax[tid >> 6] = inc;
  __syncthreads();
  int before = excl;
  for (int w = 0; w < (tid >> 6); w++) before = s_wmax[w] > before ? s_wmax[w] : before;
#pragma unroll
  for (int k = 0; k < FT_ITEMS; k++) t.h[k] = t.h[k] > before ? t.h[k] : before;
  return t;
}

// this thread's FT_ITEMS consecutive columns: two 16-byte loads where the array allows it (eight 4-byte loads at a
// stride of 32 bytes across the lanes make eight times the requests)
template <typename I>
__device__ __forceinline__ void load_items(const I *__restrict__ col, int64_t base, int64_t t1, bool vec_ok, I *c) {
  static_assert(FT_ITEMS == 8, "two 16-byte loads per thread (four for 64-bit columns)");
  if (vec_ok && base + FT_ITEMS <= t1) {
    if (sizeof(I) == 4) {
      const int4 a = *(const int4 *)(col + base), b = *(const int4 *)(col + base + 4);
      c[0] = (I)a.x; c[1] = (I)a.y; c[2] = (I)a.z; c[3] = (I)a.w;
      c[4] = (I)b.x; c[5] = (I)b.y; c[6] = (I)b.z; c[7] = (I)b.w;
    } else {
#pragma unroll
      for (int k = 0; k < FT_ITEMS; k += 2) {
        const longlong2 a = *(const longlong2 *)(col + base + k);
        c[k] = (I)a.x; c[k + 1] = (I)a.y;
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < FT_ITEMS; k++) c[k] = col[base + k < t1 ? base + k : t1 - 1];
  }
}

// bandwidth.cc:100-107: max |row - col| over the nonzeros
template <typename I, typename P>  // P: unsigned for 32-bit columns, unsigned long long for 64-bit ones
__global__ __launch_bounds__(FT_THREADS) void k_bandwidth_csr(const I *__restrict__ rp, const I *__restrict__ col, int64_t n,
                                                              int64_t nnz, P *__restrict__ partial, bool vec_ok,
                                                              const int2 *__restrict__ span) {
  __shared__ int s_head[FT_TILE];
  __shared__ int s_wmax[FT_THREADS / 64];
  __shared__ P s_mx[FT_THREADS / 64];
  const int tid = threadIdx.x;
  const int64_t tiles = (nnz + FT_TILE - 1) / FT_TILE;
  P mx = 0;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {  // (one tile per workgroup as launched)
    const int64_t t0 = tile * FT_TILE;
    const int64_t t1 = (t0 + FT_TILE < nnz) ? t0 + FT_TILE : nnz;
    const int64_t base = t0 + (int64_t)tid * FT_ITEMS;
    I c[FT_ITEMS];  // in flight during the row search
    load_items(col, base, t1, vec_ok, c);
    const TileRows t = tile_rows(rp, span, tile, t0, s_head, s_wmax);
#pragma unroll
    for (int k = 0; k < FT_ITEMS; k++) {
      if (base + k < t1) {
        const int64_t d = t.r_lo + t.h[k] - (int64_t)c[k];
        const P a = (P)(d < 0 ? -d : d);
        mx = a > mx ? a : mx;
      }
    }
    __syncthreads();  // the LDS of tile_rows is reused
  }
  mx = sbx_wave_max(mx);
  if (sbx_lane() == 0) s_mx[sbx_wave_in_block()] = mx;
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < FT_THREADS / 64; w++) mx = s_mx[w] > mx ? s_mx[w] : mx;
    // FT_SLOTS result words shared by all workgroups: a word per workgroup leaves the single finishing workgroup with
    // 51 K loads on the bench matrix (0.14 ms), one word for all would take 88 atomics per microsecond
    if (mx) atomicMax(&partial[blockIdx.x & (FT_SLOTS - 1)], mx);
  }
}

// profile.cc:95-104 for column-sorted rows, and the check that they are (format/csr.cc:102-116) in the same read of
// the columns: the smallest column of a row is its first one; `*unsorted` is raised if some row is out of order and
// the caller then takes the general path.
template <typename I>
__global__ __launch_bounds__(FT_THREADS) void k_profile_csr(const I *__restrict__ rp, const I *__restrict__ col, int64_t n,
                                                            int64_t nnz,
                                                            unsigned long long *__restrict__ partial,
                                                            int *__restrict__ unsorted, bool vec_ok,
                                                            const int2 *__restrict__ span) {
  __shared__ int s_head[FT_TILE];
  __shared__ int s_wmax[FT_THREADS / 64];
  __shared__ unsigned long long s_sum[FT_THREADS / 64];
  const int tid = threadIdx.x;
  const int64_t tiles = (nnz + FT_TILE - 1) / FT_TILE;
  unsigned long long sum = 0;
  bool bad = false;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t t0 = tile * FT_TILE;
    const int64_t t1 = (t0 + FT_TILE < nnz) ? t0 + FT_TILE : nnz;
    const int64_t base = t0 + (int64_t)tid * FT_ITEMS;
    I c[FT_ITEMS + 1];  // c[0]: the nonzero before this thread's first one
    c[0] = base > 0 ? col[base - 1 < t1 ? base - 1 : t1 - 1] : 0;
    load_items(col, base, t1, vec_ok, c + 1);
    const TileRows t = tile_rows(rp, span, tile, t0, s_head, s_wmax);
#pragma unroll
    for (int k = 0; k < FT_ITEMS; k++) {
      if (base + k < t1) {
        if (t.head[k]) {
          const int64_t row = t.r_lo + t.h[k];
          if ((int64_t)c[k + 1] < row) sum += (unsigned long long)(row - (int64_t)c[k + 1]);
        } else {
          bad |= c[k + 1] < c[k];
        }
      }
    }
    __syncthreads();  // the LDS of tile_rows is reused
  }
  if (__any(bad) && sbx_lane() == 0) *unsorted = 1;
  sum = sbx_wave_sum(sum);
  if (sbx_lane() == 0) s_sum[sbx_wave_in_block()] = sum;
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < FT_THREADS / 64; w++) sum += s_sum[w];
    if (sum) atomicAdd(&partial[blockIdx.x & (FT_SLOTS - 1)], sum);
  }
}

__device__ __forceinline__ void ft_atomic_min(int32_t *p, int32_t v) { atomicMin(p, v); }
__device__ __forceinline__ void ft_atomic_min(int64_t *p, int64_t v) { atomicMin((long long *)p, (long long)v); }

// smallest column of every row: each thread owns FT_ITEMS consecutive nonzeros and issues one
// atomicMin per run of equal rows (rows are contiguous, so ~nnz/FT_ITEMS + n atomics in total)
template <typename I>
__global__ __launch_bounds__(FT_THREADS) void k_row_min_col(const I *__restrict__ row, const I *__restrict__ col, int64_t nnz,
                                                            I *__restrict__ rowmin) {
  const int64_t p0 = ((int64_t)blockIdx.x * FT_THREADS + threadIdx.x) * FT_ITEMS;
  if (p0 >= nnz) return;
  I cur = row[p0], m = col[p0];
#pragma unroll
  for (int k = 1; k < FT_ITEMS; k++) {
    if (p0 + k >= nnz) break;
    const I r = row[p0 + k], c = col[p0 + k];
    if (r != cur) {
      ft_atomic_min(&rowmin[cur], m);
      cur = r;
      m = c;
    } else {
      m = c < m ? c : m;
    }
  }
  ft_atomic_min(&rowmin[cur], m);
}

template <typename I>
__global__ __launch_bounds__(FT_THREADS) void k_iota(I *__restrict__ out, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = (I)i;
}

template <typename I>
__global__ __launch_bounds__(FT_THREADS) void k_profile(const I *__restrict__ rowmin, int64_t n,
                                                        unsigned long long *__restrict__ partial) {
  __shared__ unsigned long long s_sum[FT_THREADS / 64];
  unsigned long long sum = 0;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) sum += (unsigned long long)(i - (int64_t)rowmin[i]);  // rowmin <= i by construction
  sum = sbx_wave_sum(sum);
  if (sbx_lane() == 0) s_sum[sbx_wave_in_block()] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < FT_THREADS / 64; w++) sum += s_sum[w];
    partial[blockIdx.x] = sum;
  }
}

// single workgroup: reduce the per-workgroup partials
template <typename P>
__global__ __launch_bounds__(FT_THREADS) void k_feature_finish(const P *__restrict__ pmax,
                                                               const unsigned long long *__restrict__ psum, int count,
                                                               FeatureAcc *__restrict__ acc) {
  __shared__ unsigned long long s_a[FT_THREADS / 64], s_b[FT_THREADS / 64];
  unsigned long long mx = 0, sum = 0;
  for (int i = threadIdx.x; i < count; i += FT_THREADS) {
    if (pmax) mx = (unsigned long long)pmax[i] > mx ? (unsigned long long)pmax[i] : mx;
    if (psum) sum += psum[i];
  }
  mx = sbx_wave_max(mx);
  sum = sbx_wave_sum(sum);
  if (sbx_lane() == 0) {
    s_a[sbx_wave_in_block()] = mx;
    s_b[sbx_wave_in_block()] = sum;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < FT_THREADS / 64; w++) {
      mx = s_a[w] > mx ? s_a[w] : mx;
      sum += s_b[w];
    }
    acc->max_dist = mx;
    acc->profile = sum;
  }
}

struct NestGuard {
  sbx_handle_t h;
  explicit NestGuard(sbx_handle_t h) : h(h) { h->nest++; }
  ~NestGuard() { h->nest--; }
};

// row ids of every nonzero into scratch (the CSR -> COO move conversion)
template <typename I>
int expand_rows(sbx_handle_t h, int64_t n, int64_t nnz, const I *rp, I **rows) {
  SBX_TRY(sbx_salloc(h, (size_t)nnz, rows));
  return sbx_csr_to_coo(h, sizeof(I) == 4 ? SBX_I32 : SBX_I64, SBX_V_NONE, n, n, nnz, rp, nullptr, nullptr, *rows, nullptr,
                        nullptr, SBX_FLAG_MOVE);
}

}  // namespace

#define SBX_REQUIRE(h, cond, msg)                                       \
  do {                                                                  \
    if (!(cond)) SBX_FAIL(h, SBX_ERR_BAD_ARG, "%s: %s", __func__, msg); \
  } while (0)

template <typename I, typename D = I>
static int csr_degrees_typed(sbx_handle_t h, int64_t n, const void *row_ptr, void *degrees_out) {
  SBX_TRY(sbx_arena_begin(h));
  if (n == 0) return SBX_OK;
  SBX_KLAUNCH(h, SBX_K_FEATURE, (k_degrees<I, D>), dim3(sbx_grid_for(n, FT_THREADS, 8192)), dim3(FT_THREADS), (const I *)row_ptr,
              (D *)degrees_out, n);
  SBX_LAUNCH_CHECK(h);
  SBX_PROF_BYTES(h, SBX_K_FEATURE, 2 * (int64_t)sizeof(I) * n + (int64_t)sizeof(I));
  return SBX_OK;
}

extern "C" int sbx_csr_degrees(sbx_handle_t h, sbx_index_type it, int64_t n, const void *row_ptr, void *degrees_out) {
  if (!h) return SBX_ERR_BAD_ARG;
  SBX_REQUIRE(h, n >= 0 && row_ptr && (n == 0 || degrees_out), "bad argument");
  if (it == SBX_I32_N64) return csr_degrees_typed<int64_t, int32_t>(h, n, row_ptr, degrees_out);
  return it == SBX_I64 ? csr_degrees_typed<int64_t>(h, n, row_ptr, degrees_out)
                       : csr_degrees_typed<int32_t>(h, n, row_ptr, degrees_out);
}

template <typename I>
static int csr_degree_distribution_typed(sbx_handle_t h, int64_t n, int64_t nnz, const void *row_ptr, int feature_bytes,
                                         void *dist_out) {
  SBX_TRY(sbx_arena_begin(h));
  if (n == 0) return SBX_OK;
  const unsigned grid = sbx_grid_for(n, FT_THREADS, 8192);
  if (feature_bytes == 4)
    SBX_KLAUNCH(h, SBX_K_FEATURE, (k_degree_distribution<I, float>), dim3(grid), dim3(FT_THREADS), (const I *)row_ptr,
                (float *)dist_out, n, (float)nnz);
  else
    SBX_KLAUNCH(h, SBX_K_FEATURE, (k_degree_distribution<I, double>), dim3(grid), dim3(FT_THREADS), (const I *)row_ptr,
                (double *)dist_out, n, (double)nnz);
  SBX_LAUNCH_CHECK(h);
  SBX_PROF_BYTES(h, SBX_K_FEATURE, ((int64_t)sizeof(I) + feature_bytes) * n + (int64_t)sizeof(I));
  return SBX_OK;
}

extern "C" int sbx_csr_degree_distribution(sbx_handle_t h, sbx_index_type it, int64_t n, int64_t nnz,
                                           const void *row_ptr, int feature_bytes, void *dist_out) {
  if (!h) return SBX_ERR_BAD_ARG;
  SBX_REQUIRE(h, n >= 0 && nnz >= 0 && row_ptr && (n == 0 || dist_out), "bad argument");
  SBX_REQUIRE(h, feature_bytes == 4 || feature_bytes == 8, "feature type must be float or double");
  return it != SBX_I32 ? csr_degree_distribution_typed<int64_t>(h, n, nnz, row_ptr, feature_bytes, dist_out)  // (SBX_I32_N64: 64-bit offsets, no id array)
                       : csr_degree_distribution_typed<int32_t>(h, n, nnz, row_ptr, feature_bytes, dist_out);
}

// (rows are numbered in 32 bits inside a tile: n < 2^31 for either index type; 64-bit columns and row_ptr values —
// nnz >= 2^31, column ids >= 2^31 — are read as they are)
template <typename I>
static int csr_bandwidth_typed(sbx_handle_t h, int64_t n, int64_t nnz, const void *row_ptr, const void *col,
                               int64_t *bandwidth_host) {
  typedef typename std::conditional<sizeof(I) == 4, unsigned, unsigned long long>::type P;
  SBX_TRY(sbx_arena_begin(h));
  if (nnz == 0) return SBX_OK;  // bandwidth.cc:100: stays 0 without nonzeros
  const int64_t tiles = (nnz + FT_TILE - 1) / FT_TILE;
  const unsigned grid = FT_SLOTS;  // result words for k_feature_finish
  P *partial = nullptr;
  FeatureAcc *acc = nullptr;
  SBX_TRY(sbx_salloc(h, grid, &partial));
  SBX_TRY(sbx_salloc(h, 1, &acc));
  SBX_HIP(h, hipMemsetAsync(partial, 0, grid * sizeof(P), h->stream));
  int2 *span = nullptr;
  SBX_TRY(sbx_salloc(h, (size_t)tiles, &span));
  SBX_KLAUNCH(h, SBX_K_FEATURE, k_tile_spans<I>, dim3((unsigned)((tiles + FT_THREADS - 1) / FT_THREADS)), dim3(FT_THREADS),
              (const I *)row_ptr, n, nnz, tiles, span);
  SBX_KLAUNCH(h, SBX_K_FEATURE, (k_bandwidth_csr<I, P>), dim3((unsigned)tiles), dim3(FT_THREADS), (const I *)row_ptr,
              (const I *)col, n, nnz, partial, ((uintptr_t)col & 15) == 0, (const int2 *)span);
  SBX_KLAUNCH(h, SBX_K_FEATURE, k_feature_finish<P>, dim3(1), dim3(FT_THREADS), (const P *)partial,
              (const unsigned long long *)nullptr, (int)grid, acc);
  SBX_LAUNCH_CHECK(h);
  SBX_PROF_BYTES(h, SBX_K_FEATURE, (int64_t)sizeof(I) * (nnz + n + 1));
  FeatureAcc ha;
  SBX_TRY(sbx_readback(h, &ha, acc, sizeof(FeatureAcc)));
  *bandwidth_host = (int64_t)ha.max_dist + 1;  // |i - j| + 1 (:104-107)
  return SBX_OK;
}

extern "C" int sbx_csr_bandwidth(sbx_handle_t h, sbx_index_type it, int64_t n, int64_t nnz, const void *row_ptr,
                                 const void *col, int64_t *bandwidth_host) {
  if (!h) return SBX_ERR_BAD_ARG;
  if (it == SBX_I32_N64) return sbx_mixed_csr_bandwidth(h, n, nnz, row_ptr, col, bandwidth_host);
  SBX_REQUIRE(h, n >= 0 && nnz >= 0 && row_ptr && bandwidth_host && (nnz == 0 || col), "bad argument");
  SBX_REQUIRE(h, n < ((int64_t)1 << 31) - 1 && (it == SBX_I64 || nnz < ((int64_t)1 << 31)) &&
                     nnz / FT_TILE < ((int64_t)1 << 31), "dimension exceeds what the index type holds");
  *bandwidth_host = 0;
  return it == SBX_I64 ? csr_bandwidth_typed<int64_t>(h, n, nnz, row_ptr, col, bandwidth_host)
                       : csr_bandwidth_typed<int32_t>(h, n, nnz, row_ptr, col, bandwidth_host);
}

template <typename I>
static int csr_profile_typed(sbx_handle_t h, int64_t n, int64_t nnz, const void *row_ptr, const void *col,
                             int64_t *profile_host) {
  SBX_TRY(sbx_arena_begin(h));
  if (nnz == 0 || n == 0) return SBX_OK;
  NestGuard guard(h);
  const unsigned grid_n = sbx_grid_for(n, FT_THREADS, (int64_t)h->num_cus * 8);
  unsigned long long *partial = nullptr;
  FeatureAcc *acc = nullptr;
  SBX_TRY(sbx_salloc(h, grid_n, &partial));
  SBX_TRY(sbx_salloc(h, 1, &acc));
  {
    // column-sorted rows (every CSR that went through a constructor): checked and summed in ONE read of the columns;
    // unsorted rows only exist for ignore_sort CSRs and take the general path below
    struct { FeatureAcc a; int unsorted; int pad; } hs;
    const int64_t tiles = (nnz + FT_TILE - 1) / FT_TILE;
    const unsigned grid = FT_SLOTS;
    unsigned long long *psum = nullptr, *res = nullptr;
    SBX_TRY(sbx_salloc(h, grid + 3, &psum));
    res = psum + grid;
    SBX_HIP(h, hipMemsetAsync(psum, 0, (grid + 3) * sizeof(unsigned long long), h->stream));
    int2 *span = nullptr;
    SBX_TRY(sbx_salloc(h, (size_t)tiles, &span));
    SBX_KLAUNCH(h, SBX_K_FEATURE, k_tile_spans<I>, dim3((unsigned)((tiles + FT_THREADS - 1) / FT_THREADS)), dim3(FT_THREADS),
                (const I *)row_ptr, n, nnz, tiles, span);
    SBX_KLAUNCH(h, SBX_K_FEATURE, k_profile_csr<I>, dim3((unsigned)tiles), dim3(FT_THREADS), (const I *)row_ptr,
                (const I *)col, n, nnz, psum, (int *)(res + 2), ((uintptr_t)col & 15) == 0, (const int2 *)span);
    SBX_KLAUNCH(h, SBX_K_FEATURE, k_feature_finish<unsigned>, dim3(1), dim3(FT_THREADS), (const unsigned *)nullptr,
                (const unsigned long long *)psum, (int)grid, (FeatureAcc *)res);
    SBX_LAUNCH_CHECK(h);
    SBX_PROF_BYTES(h, SBX_K_FEATURE, (int64_t)sizeof(I) * (nnz + n + 1));
    SBX_TRY(sbx_readback(h, &hs, res, sizeof(hs)));
    if (!hs.unsorted) {
      *profile_host = (int64_t)hs.a.profile;
      return SBX_OK;
    }
  }
  I *rows = nullptr, *rowmin = nullptr;
  SBX_TRY(expand_rows<I>(h, n, nnz, (const I *)row_ptr, &rows));
  SBX_TRY(sbx_salloc(h, (size_t)n, &rowmin));
  SBX_KLAUNCH(h, SBX_K_FEATURE, k_iota<I>, dim3(grid_n), dim3(FT_THREADS), rowmin, n);  // j starts at i (:99)
  SBX_KLAUNCH(h, SBX_K_FEATURE, k_row_min_col<I>, dim3((unsigned)((nnz + FT_THREADS * FT_ITEMS - 1) / (FT_THREADS * FT_ITEMS))),
              dim3(FT_THREADS), (const I *)rows, (const I *)col, nnz, rowmin);
  SBX_KLAUNCH(h, SBX_K_FEATURE, k_profile<I>, dim3(grid_n), dim3(FT_THREADS), (const I *)rowmin, n, partial);
  SBX_KLAUNCH(h, SBX_K_FEATURE, k_feature_finish<unsigned>, dim3(1), dim3(FT_THREADS), (const unsigned *)nullptr,
              (const unsigned long long *)partial, (int)grid_n, acc);
  SBX_LAUNCH_CHECK(h);
  SBX_PROF_BYTES(h, SBX_K_FEATURE, (int64_t)sizeof(I) * (nnz + n + 1));
  FeatureAcc ha;
  SBX_TRY(sbx_readback(h, &ha, acc, sizeof(FeatureAcc)));
  *profile_host = (int64_t)ha.profile;
  return SBX_OK;
}

extern "C" int sbx_csr_profile(sbx_handle_t h, sbx_index_type it, int64_t n, int64_t nnz, const void *row_ptr,
                               const void *col, int64_t *profile_host) {
  if (!h) return SBX_ERR_BAD_ARG;
  if (it == SBX_I32_N64) return sbx_mixed_csr_profile(h, n, nnz, row_ptr, col, profile_host);
  SBX_REQUIRE(h, n >= 0 && nnz >= 0 && row_ptr && profile_host && (nnz == 0 || col), "bad argument");
  SBX_REQUIRE(h, n < ((int64_t)1 << 31) - 1 && (it == SBX_I64 || nnz < ((int64_t)1 << 31)) &&
                     nnz / FT_TILE < ((int64_t)1 << 31), "dimension exceeds what the index type holds");
  *profile_host = 0;
  return it == SBX_I64 ? csr_profile_typed<int64_t>(h, n, nnz, row_ptr, col, profile_host)
                       : csr_profile_typed<int32_t>(h, n, nnz, row_ptr, col, profile_host);
}
