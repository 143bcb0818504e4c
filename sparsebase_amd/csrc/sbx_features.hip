// sbx_features.hip — the reorder-quality features of SURVEY §8(f).2: what a reordering
// bought, computed where the permuted CSR already lives.
//
//   feature/degrees.cc:93-105               sbx_csr_degrees
//   feature/degree_distribution.cc:152-167  sbx_csr_degree_distribution
//   feature/bandwidth.cc:93-112             sbx_csr_bandwidth
//   feature/profile.cc:91-105               sbx_csr_profile
//
// All four are single-pass reductions.  Bandwidth and profile are nonzero-parallel so
// power-law rows stay balanced: the row of every nonzero comes from the CSR->COO
// expansion kernel (row ids into scratch), never from a per-row loop.
#include "sbx_device.h"
#include "sbx_internal.h"

namespace {

constexpr int FT_THREADS = 256;
constexpr int FT_ITEMS = 8;

__global__ __launch_bounds__(FT_THREADS) void k_degrees(const int32_t *__restrict__ rp, int32_t *__restrict__ out,
                                                        int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = rp[i + 1] - rp[i];
}

// dist[i] = degree / (FeatureType)num_edges, one IEEE division per row (degree_distribution.cc:163)
template <typename F>
__global__ __launch_bounds__(FT_THREADS) void k_degree_distribution(const int32_t *__restrict__ rp, F *__restrict__ out,
                                                                    int64_t n, F nnz) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = (F)(rp[i + 1] - rp[i]) / nnz;
}

struct FeatureAcc {
  unsigned long long max_dist;  // max |row - col| over the nonzeros
  unsigned long long profile;   // sum over rows of (row - min(row, smallest column))
};

__global__ __launch_bounds__(FT_THREADS) void k_bandwidth(const int32_t *__restrict__ row,
                                                          const int32_t *__restrict__ col, int64_t nnz,
                                                          unsigned *__restrict__ partial) {
  __shared__ unsigned s_mx[FT_THREADS / 64];
  unsigned mx = 0;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < nnz; i += stride) {
    const int d = row[i] - col[i];
    const unsigned a = (unsigned)(d < 0 ? -d : d);
    mx = a > mx ? a : mx;
  }
  mx = sbx_wave_max(mx);
  if (sbx_lane() == 0) s_mx[sbx_wave_in_block()] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < FT_THREADS / 64; w++) mx = s_mx[w] > mx ? s_mx[w] : mx;
    partial[blockIdx.x] = mx;  // one word per workgroup, reduced by k_feature_finish (no hot atomic)
  }
}

// smallest column of every row: each thread owns FT_ITEMS consecutive nonzeros and issues one
// atomicMin per run of equal rows (rows are contiguous, so ~nnz/FT_ITEMS + n atomics in total)
__global__ __launch_bounds__(FT_THREADS) void k_row_min_col(const int32_t *__restrict__ row,
                                                            const int32_t *__restrict__ col, int64_t nnz,
                                                            int32_t *__restrict__ rowmin) {
  const int64_t p0 = ((int64_t)blockIdx.x * FT_THREADS + threadIdx.x) * FT_ITEMS;
  if (p0 >= nnz) return;
  int32_t cur = row[p0], m = col[p0];
#pragma unroll
  for (int k = 1; k < FT_ITEMS; k++) {
    if (p0 + k >= nnz) break;
    const int32_t r = row[p0 + k], c = col[p0 + k];
    if (r != cur) {
      atomicMin(&rowmin[cur], m);
      cur = r;
      m = c;
    } else {
      m = c < m ? c : m;
    }
  }
  atomicMin(&rowmin[cur], m);
}

__global__ __launch_bounds__(FT_THREADS) void k_iota(int32_t *__restrict__ out, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = (int32_t)i;
}

__global__ __launch_bounds__(FT_THREADS) void k_profile(const int32_t *__restrict__ rowmin, int64_t n,
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

// column-sorted rows (every CSR that went through a constructor): the smallest column is the first one
__global__ __launch_bounds__(FT_THREADS) void k_profile_sorted(const int32_t *__restrict__ rp,
                                                               const int32_t *__restrict__ col, int64_t n,
                                                               unsigned long long *__restrict__ partial) {
  __shared__ unsigned long long s_sum[FT_THREADS / 64];
  unsigned long long sum = 0;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const int32_t s0 = rp[i];
    if (rp[i + 1] > s0) {
      const int64_t c = col[s0];
      if (c < i) sum += (unsigned long long)(i - c);
    }
  }
  sum = sbx_wave_sum(sum);
  if (sbx_lane() == 0) s_sum[sbx_wave_in_block()] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < FT_THREADS / 64; w++) sum += s_sum[w];
    partial[blockIdx.x] = sum;
  }
}

// single workgroup: reduce the per-workgroup partials
__global__ __launch_bounds__(FT_THREADS) void k_feature_finish(const unsigned *__restrict__ pmax,
                                                               const unsigned long long *__restrict__ psum, int count,
                                                               FeatureAcc *__restrict__ acc) {
  __shared__ unsigned long long s_a[FT_THREADS / 64], s_b[FT_THREADS / 64];
  unsigned long long mx = 0, sum = 0;
  for (int i = threadIdx.x; i < count; i += FT_THREADS) {
    if (pmax) mx = pmax[i] > mx ? pmax[i] : mx;
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
int expand_rows(sbx_handle_t h, int64_t n, int64_t nnz, const int32_t *rp, int32_t **rows) {
  SBX_TRY(sbx_salloc(h, (size_t)nnz, rows));
  return sbx_csr_to_coo(h, SBX_I32, SBX_V_NONE, n, n, nnz, rp, nullptr, nullptr, *rows, nullptr, nullptr, SBX_FLAG_MOVE);
}

}  // namespace

#define SBX_REQUIRE(h, cond, msg)                                       \
  do {                                                                  \
    if (!(cond)) SBX_FAIL(h, SBX_ERR_BAD_ARG, "%s: %s", __func__, msg); \
  } while (0)

extern "C" int sbx_csr_degrees(sbx_handle_t h, sbx_index_type it, int64_t n, const void *row_ptr, void *degrees_out) {
  if (!h) return SBX_ERR_BAD_ARG;
  SBX_REQUIRE(h, n >= 0 && row_ptr && (n == 0 || degrees_out), "bad argument");
  if (it == SBX_I64) return sbx_i64_csr_degrees(h, n, row_ptr, degrees_out);
  SBX_TRY(sbx_arena_begin(h));
  if (n == 0) return SBX_OK;
  SBX_KLAUNCH(h, SBX_K_FEATURE, k_degrees, dim3(sbx_grid_for(n, FT_THREADS, 8192)), dim3(FT_THREADS),
              (const int32_t *)row_ptr, (int32_t *)degrees_out, n);
  SBX_LAUNCH_CHECK(h);
  SBX_PROF_BYTES(h, SBX_K_FEATURE, 8 * n + 4);
  return SBX_OK;
}

extern "C" int sbx_csr_degree_distribution(sbx_handle_t h, sbx_index_type it, int64_t n, int64_t nnz,
                                           const void *row_ptr, int feature_bytes, void *dist_out) {
  if (!h) return SBX_ERR_BAD_ARG;
  SBX_REQUIRE(h, n >= 0 && nnz >= 0 && row_ptr && (n == 0 || dist_out), "bad argument");
  SBX_REQUIRE(h, feature_bytes == 4 || feature_bytes == 8, "feature type must be float or double");
  if (it == SBX_I64) return sbx_i64_csr_degree_distribution(h, n, nnz, row_ptr, feature_bytes, dist_out);
  SBX_TRY(sbx_arena_begin(h));
  if (n == 0) return SBX_OK;
  const unsigned grid = sbx_grid_for(n, FT_THREADS, 8192);
  if (feature_bytes == 4)
    SBX_KLAUNCH(h, SBX_K_FEATURE, k_degree_distribution<float>, dim3(grid), dim3(FT_THREADS), (const int32_t *)row_ptr,
                (float *)dist_out, n, (float)nnz);
  else
    SBX_KLAUNCH(h, SBX_K_FEATURE, k_degree_distribution<double>, dim3(grid), dim3(FT_THREADS),
                (const int32_t *)row_ptr, (double *)dist_out, n, (double)nnz);
  SBX_LAUNCH_CHECK(h);
  SBX_PROF_BYTES(h, SBX_K_FEATURE, (4 + feature_bytes) * n + 4);
  return SBX_OK;
}

extern "C" int sbx_csr_bandwidth(sbx_handle_t h, sbx_index_type it, int64_t n, int64_t nnz, const void *row_ptr,
                                 const void *col, int64_t *bandwidth_host) {
  if (!h) return SBX_ERR_BAD_ARG;
  SBX_REQUIRE(h, n >= 0 && nnz >= 0 && row_ptr && bandwidth_host && (nnz == 0 || col), "bad argument");
  SBX_REQUIRE(h, nnz < ((int64_t)1 << 31) && n < ((int64_t)1 << 31) - 1, "dimension exceeds int32");
  *bandwidth_host = 0;
  if (it == SBX_I64) return sbx_i64_csr_bandwidth(h, n, nnz, row_ptr, col, bandwidth_host);
  SBX_TRY(sbx_arena_begin(h));
  if (nnz == 0) return SBX_OK;  // bandwidth.cc:100: stays 0 without nonzeros
  NestGuard guard(h);
  int32_t *rows = nullptr;
  SBX_TRY(expand_rows(h, n, nnz, (const int32_t *)row_ptr, &rows));
  const unsigned grid = sbx_grid_for(nnz, FT_THREADS * 8, (int64_t)h->num_cus * 8);
  unsigned *partial = nullptr;
  FeatureAcc *acc = nullptr;
  SBX_TRY(sbx_salloc(h, grid, &partial));
  SBX_TRY(sbx_salloc(h, 1, &acc));
  SBX_KLAUNCH(h, SBX_K_FEATURE, k_bandwidth, dim3(grid), dim3(FT_THREADS), (const int32_t *)rows, (const int32_t *)col,
              nnz, partial);
  SBX_KLAUNCH(h, SBX_K_FEATURE, k_feature_finish, dim3(1), dim3(FT_THREADS), (const unsigned *)partial,
              (const unsigned long long *)nullptr, (int)grid, acc);
  SBX_LAUNCH_CHECK(h);
  SBX_PROF_BYTES(h, SBX_K_FEATURE, 4 * nnz + 4 * (n + 1));
  FeatureAcc ha;
  SBX_TRY(sbx_readback(h, &ha, acc, sizeof(FeatureAcc)));
  *bandwidth_host = (int64_t)ha.max_dist + 1;  // |i - j| + 1 (:104-107)
  return SBX_OK;
}

extern "C" int sbx_csr_profile(sbx_handle_t h, sbx_index_type it, int64_t n, int64_t nnz, const void *row_ptr,
                               const void *col, int64_t *profile_host) {
  if (!h) return SBX_ERR_BAD_ARG;
  SBX_REQUIRE(h, n >= 0 && nnz >= 0 && row_ptr && profile_host && (nnz == 0 || col), "bad argument");
  SBX_REQUIRE(h, nnz < ((int64_t)1 << 31) && n < ((int64_t)1 << 31) - 1, "dimension exceeds int32");
  *profile_host = 0;
  if (it == SBX_I64) return sbx_i64_csr_profile(h, n, nnz, row_ptr, col, profile_host);
  SBX_TRY(sbx_arena_begin(h));
  if (nnz == 0 || n == 0) return SBX_OK;
  NestGuard guard(h);
  const unsigned grid_n = sbx_grid_for(n, FT_THREADS, (int64_t)h->num_cus * 8);
  unsigned long long *partial = nullptr;
  FeatureAcc *acc = nullptr;
  SBX_TRY(sbx_salloc(h, grid_n, &partial));
  SBX_TRY(sbx_salloc(h, 1, &acc));
  int sorted = 0;  // one streaming pass over col; unsorted rows only exist for ignore_sort CSRs
  SBX_TRY(sbx_csr_rows_sorted(h, SBX_I32, n, row_ptr, col, &sorted));
  if (sorted) {
    SBX_KLAUNCH(h, SBX_K_FEATURE, k_profile_sorted, dim3(grid_n), dim3(FT_THREADS), (const int32_t *)row_ptr,
                (const int32_t *)col, n, partial);
    SBX_KLAUNCH(h, SBX_K_FEATURE, k_feature_finish, dim3(1), dim3(FT_THREADS), (const unsigned *)nullptr,
                (const unsigned long long *)partial, (int)grid_n, acc);
    SBX_LAUNCH_CHECK(h);
    SBX_PROF_BYTES(h, SBX_K_FEATURE, 4 * nnz + 4 * (n + 1));
    FeatureAcc hs;
    SBX_TRY(sbx_readback(h, &hs, acc, sizeof(FeatureAcc)));
    *profile_host = (int64_t)hs.profile;
    return SBX_OK;
  }
  int32_t *rows = nullptr, *rowmin = nullptr;
  SBX_TRY(expand_rows(h, n, nnz, (const int32_t *)row_ptr, &rows));
  SBX_TRY(sbx_salloc(h, (size_t)n, &rowmin));
  SBX_KLAUNCH(h, SBX_K_FEATURE, k_iota, dim3(grid_n), dim3(FT_THREADS), rowmin, n);  // j starts at i (:99)
  SBX_KLAUNCH(h, SBX_K_FEATURE, k_row_min_col, dim3((unsigned)((nnz + FT_THREADS * FT_ITEMS - 1) / (FT_THREADS * FT_ITEMS))),
              dim3(FT_THREADS), (const int32_t *)rows, (const int32_t *)col, nnz, rowmin);
  SBX_KLAUNCH(h, SBX_K_FEATURE, k_profile, dim3(grid_n), dim3(FT_THREADS), (const int32_t *)rowmin, n, partial);
  SBX_KLAUNCH(h, SBX_K_FEATURE, k_feature_finish, dim3(1), dim3(FT_THREADS), (const unsigned *)nullptr,
              (const unsigned long long *)partial, (int)grid_n, acc);
  SBX_LAUNCH_CHECK(h);
  SBX_PROF_BYTES(h, SBX_K_FEATURE, 4 * nnz + 4 * (n + 1));
  FeatureAcc ha;
  SBX_TRY(sbx_readback(h, &ha, acc, sizeof(FeatureAcc)));
  *profile_host = (int64_t)ha.profile;
  return SBX_OK;
}
