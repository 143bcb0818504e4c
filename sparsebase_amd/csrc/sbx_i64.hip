// sbx_i64.hip — SBX_I64 support: 64-bit IDType/NNZType arrays (the reference's
// <int64,int64,double> tuple).  NATIVE 64-bit kernels (no copies): the conversions COO <-> CSR and the two sortedness
// checks (sbx_convert.hip), the four features (sbx_features.hip), DegreeReorder (sbx_degree.hip), InversePermutation,
// PermuteArray, the CSR permute with and without a column map, the CSR constructor's row sort (sbx_permute.hip) and the
// COO constructor's sort and the CSC conversions (sbx_convert.hip).  The entry points in THIS file — RCM, Gray keys, the text parsers, and
// the COO sort of coordinates that lie outside their matrix — narrow their index arrays to int32 scratch copies (with
// an overflow check), run the int32 kernels and widen the index outputs back; values are opaque payload and pass
// through untouched; arrays with entries >= 2^31 return SBX_ERR_UNSUPPORTED there.
#include "sbx_device.h"
#include "sbx_internal.h"

namespace {

__global__ __launch_bounds__(256) void k_narrow(const int64_t *__restrict__ in, int32_t *__restrict__ out,
                                                int64_t count, int *__restrict__ overflow) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  bool bad = false;
  for (; i < count; i += stride) {
    const int64_t v = in[i];
    bad |= (v < 0) || (v > 0x7FFFFFFFll);
    out[i] = (int32_t)v;
  }
  if (__any(bad) && sbx_lane() == 0) *overflow = 1;
}

__global__ __launch_bounds__(256) void k_widen(const int32_t *__restrict__ in, int64_t *__restrict__ out,
                                               int64_t count) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < count; i += stride) out[i] = (int64_t)in[i];
}

}  // namespace

int sbx_i64_begin(sbx_handle_t h, int **overflow_flag_dev) {
  SBX_TRY(sbx_arena_begin(h));
  h->nest++;
  int rc = sbx_salloc(h, 1, overflow_flag_dev);
  if (rc == SBX_OK && hipMemsetAsync(*overflow_flag_dev, 0, sizeof(int), h->stream) != hipSuccess) rc = SBX_ERR_HIP;
  if (rc != SBX_OK) h->nest--;
  return rc;
}

void sbx_i64_end(sbx_handle_t h) {
  if (h->nest > 0) h->nest--;
}

int sbx_narrow_i64(sbx_handle_t h, const void *src_i64, int64_t count, int32_t **out, int *overflow_flag_dev) {
  *out = nullptr;
  if (!src_i64) return SBX_OK;
  SBX_TRY(sbx_salloc(h, (size_t)(count > 0 ? count : 1), out));
  if (count > 0) {
    SBX_KLAUNCH(h, SBX_K_MISC, k_narrow, dim3(sbx_grid_for(count, 256, 8192)), dim3(256), (const int64_t *)src_i64, *out,
                count, overflow_flag_dev);
    SBX_LAUNCH_CHECK(h);
  }
  return SBX_OK;
}

int sbx_widen_i32(sbx_handle_t h, const int32_t *src, void *dst_i64, int64_t count) {
  if (!src || !dst_i64 || count <= 0) return SBX_OK;
  SBX_KLAUNCH(h, SBX_K_MISC, k_widen, dim3(sbx_grid_for(count, 256, 8192)), dim3(256), src, (int64_t *)dst_i64, count);
  SBX_LAUNCH_CHECK(h);
  return SBX_OK;
}

int sbx_i64_check(sbx_handle_t h, const int *overflow_flag_dev) {
  int f = 0;
  SBX_TRY(sbx_readback(h, &f, overflow_flag_dev, sizeof(int)));
  if (f)
    SBX_FAIL(h, SBX_ERR_UNSUPPORTED,
             "64-bit index array holds a value outside [0, 2^31) on a path that sorts 32-bit keys (coordinates outside the matrix, text ingest)");
  return SBX_OK;
}

// ---------------------------------------------------------------------------
// 64-bit entry points: narrow -> int32 entry point -> widen
// ---------------------------------------------------------------------------
namespace {
struct Scope {  // leaves nesting mode on every return path
  sbx_handle_t h;
  explicit Scope(sbx_handle_t h) : h(h) {}
  ~Scope() { sbx_i64_end(h); }
};
}  // namespace

#define I64_BEGIN()                       \
  int *ovf = nullptr;                     \
  SBX_TRY(sbx_i64_begin(h, &ovf));        \
  Scope scope_(h)
#define NARROW(name, src, count) \
  int32_t *name = nullptr;       \
  SBX_TRY(sbx_narrow_i64(h, src, count, &name, ovf))
#define SCRATCH32(name, count, want) \
  int32_t *name = nullptr;           \
  if (want) SBX_TRY(sbx_salloc(h, (size_t)((count) > 0 ? (count) : 1), &name))

int sbx_i64_coo_sort(sbx_handle_t h, sbx_value_type vt, int64_t n, int64_t m, int64_t nnz, void *row, void *col,
                     void *val) {
  I64_BEGIN();
  NARROW(r, row, nnz);
  NARROW(c, col, nnz);
  SBX_TRY(sbx_i64_check(h, ovf));
  SBX_TRY(sbx_coo_sort(h, SBX_I32, vt, n, m, nnz, r, c, val));
  SBX_TRY(sbx_widen_i32(h, r, row, nnz));
  return sbx_widen_i32(h, c, col, nnz);
}

static int read_nnz_i64(sbx_handle_t h, const void *row_ptr, int64_t n, int64_t *nnz) {
  return sbx_readback(h, nnz, (const int64_t *)row_ptr + n, sizeof(int64_t));
}

int sbx_i64_mtx_parse_coordinate(sbx_handle_t h, sbx_value_type vt, const void *text_dev, int64_t bytes, int64_t n_rows,
                                 int64_t n_cols, int64_t entries, int fields, int symmetry, unsigned flags,
                                 int64_t capacity, void *row_out, void *col_out, void *val_out, int64_t *nnz_host) {
  I64_BEGIN();
  (void)ovf;
  SCRATCH32(r, capacity, true);
  SCRATCH32(c, capacity, true);
  SBX_TRY(sbx_mtx_parse_coordinate(h, SBX_I32, vt, text_dev, bytes, n_rows, n_cols, entries, fields, symmetry, flags,
                                   capacity, r, c, val_out, nnz_host));
  SBX_TRY(sbx_widen_i32(h, r, row_out, *nnz_host));
  return sbx_widen_i32(h, c, col_out, *nnz_host);
}

int sbx_i64_edge_list_parse(sbx_handle_t h, sbx_value_type vt, const void *text_dev, int64_t bytes, int64_t entries,
                            int weighted, unsigned flags, int64_t capacity, void *row_out, void *col_out, void *val_out,
                            int64_t *dims_nnz_host) {
  I64_BEGIN();
  (void)ovf;
  SCRATCH32(r, capacity, true);
  SCRATCH32(c, capacity, true);
  SBX_TRY(sbx_edge_list_parse(h, SBX_I32, vt, text_dev, bytes, entries, weighted, flags, capacity, r, c, val_out,
                              dims_nnz_host));
  SBX_TRY(sbx_widen_i32(h, r, row_out, dims_nnz_host[2]));
  return sbx_widen_i32(h, c, col_out, dims_nnz_host[2]);
}


// ---------------------------------------------------------------------------
// SBX_I32_N64 — 32-bit ids with 64-bit offsets (the reference's <int | unsigned int, long long | unsigned long long, V>
// tuples, CMakeLists.txt:15-17).  COO <-> CSR, DegreeReorder and the degree features read and write the 64-bit offsets
// themselves (sbx_convert.hip, sbx_degree.hip, sbx_features.hip).  Every other entry point that takes an offset array
// keeps 32-bit offsets inside: the adapters below narrow row_ptr / col_ptr (n + 1 words; nnz < 2^31 is required and
// checked where the call does not state nnz), run the SBX_I32 entry point and widen the offset outputs.  The id arrays
// are 32-bit already and pass through untouched.
// ---------------------------------------------------------------------------
#define MIXED_NNZ_LIMIT(name)                                                                                        \
  if (nnz >= ((int64_t)1 << 31))                                                                                     \
  SBX_FAIL(h, SBX_ERR_UNSUPPORTED, name ": 64-bit offsets with nnz >= 2^31 (the offsets inside this operation are 32-bit)")

int sbx_mixed_csr_rows_sorted(sbx_handle_t h, int64_t n, const void *row_ptr, const void *col, int *sorted_host) {
  I64_BEGIN();
  NARROW(rp, row_ptr, n + 1);
  SBX_TRY(sbx_i64_check(h, ovf));  // (the call does not state nnz)
  return sbx_csr_rows_sorted(h, SBX_I32, n, rp, col, sorted_host);
}

int sbx_mixed_csr_sort_rows(sbx_handle_t h, sbx_value_type vt, int64_t n, int64_t m, int64_t nnz, const void *row_ptr,
                            void *col, void *val) {
  MIXED_NNZ_LIMIT("sbx_csr_sort_rows");
  I64_BEGIN();
  NARROW(rp, row_ptr, n + 1);
  return sbx_csr_sort_rows(h, SBX_I32, vt, n, m, nnz, rp, col, val);
}

int sbx_mixed_coo_to_csc(sbx_handle_t h, sbx_value_type vt, int64_t n, int64_t m, int64_t nnz, const void *row,
                         const void *col, const void *val, void *col_ptr_out, void *row_out, void *val_out) {
  MIXED_NNZ_LIMIT("sbx_coo_to_csc");
  I64_BEGIN();
  (void)ovf;
  SCRATCH32(cp, m + 1, true);
  SBX_TRY(sbx_coo_to_csc(h, SBX_I32, vt, n, m, nnz, row, col, val, cp, row_out, val_out));
  return sbx_widen_i32(h, cp, col_ptr_out, m + 1);
}

int sbx_mixed_csr_to_csc(sbx_handle_t h, sbx_value_type vt, int64_t n, int64_t m, int64_t nnz, const void *row_ptr,
                         const void *col, const void *val, void *col_ptr_out, void *row_out, void *val_out) {
  MIXED_NNZ_LIMIT("sbx_csr_to_csc");
  I64_BEGIN();
  NARROW(rp, row_ptr, n + 1);
  SCRATCH32(cp, m + 1, true);
  SBX_TRY(sbx_csr_to_csc(h, SBX_I32, vt, n, m, nnz, rp, col, val, cp, row_out, val_out));
  return sbx_widen_i32(h, cp, col_ptr_out, m + 1);
}

int sbx_mixed_csr_bandwidth(sbx_handle_t h, int64_t n, int64_t nnz, const void *row_ptr, const void *col,
                            int64_t *bandwidth_host) {
  MIXED_NNZ_LIMIT("sbx_csr_bandwidth");
  I64_BEGIN();
  NARROW(rp, row_ptr, n + 1);
  return sbx_csr_bandwidth(h, SBX_I32, n, nnz, rp, col, bandwidth_host);
}

int sbx_mixed_csr_profile(sbx_handle_t h, int64_t n, int64_t nnz, const void *row_ptr, const void *col,
                          int64_t *profile_host) {
  MIXED_NNZ_LIMIT("sbx_csr_profile");
  I64_BEGIN();
  NARROW(rp, row_ptr, n + 1);
  return sbx_csr_profile(h, SBX_I32, n, nnz, rp, col, profile_host);
}

int sbx_mixed_rcm_reorder(sbx_handle_t h, int64_t n, int64_t nnz, const void *row_ptr, const void *col, void *inv_perm_out,
                          sbx_rcm_stats *stats_host) {
  MIXED_NNZ_LIMIT("sbx_rcm_reorder");
  I64_BEGIN();
  NARROW(rp, row_ptr, n + 1);
  return sbx_rcm_reorder(h, SBX_I32, n, nnz, rp, col, inv_perm_out, stats_host);
}

int sbx_mixed_gray_row_keys(sbx_handle_t h, int64_t n, int64_t m, int64_t nnz, const void *row_ptr, const void *col,
                            int resolution, int nnz_threshold, void *degree_out, uint64_t *key_out, int64_t *counts_host) {
  MIXED_NNZ_LIMIT("sbx_gray_row_keys");
  I64_BEGIN();
  NARROW(rp, row_ptr, n + 1);
  return sbx_gray_row_keys(h, SBX_I32, n, m, nnz, rp, col, resolution, nnz_threshold, degree_out, key_out, counts_host);
}

int sbx_mixed_gray_reorder(sbx_handle_t h, int64_t n, int64_t m, int64_t nnz, const void *row_ptr, const void *col,
                           int resolution, int nnz_threshold, int group_size, int exact_ties, void *inv_perm_out) {
  MIXED_NNZ_LIMIT("sbx_gray_reorder");
  I64_BEGIN();
  NARROW(rp, row_ptr, n + 1);
  return sbx_gray_reorder(h, SBX_I32, n, m, nnz, rp, col, resolution, nnz_threshold, group_size, exact_ties, inv_perm_out);
}

int sbx_mixed_permute_csr_rows(sbx_handle_t h, sbx_value_type vt, int64_t n, int64_t m, int64_t nnz, const void *row_ptr,
                               const void *col, const void *val, const void *row_order, const void *col_order,
                               int64_t row_begin, int64_t row_end, void *row_ptr_out, void *col_out, void *val_out,
                               int64_t out_capacity, int64_t *shard_nnz_host) {
  MIXED_NNZ_LIMIT("sbx_permute_csr_rows");
  if (row_begin < 0 || row_end < row_begin || row_end > n) SBX_FAIL(h, SBX_ERR_BAD_ARG, "sbx_permute_csr_rows: bad row range");
  I64_BEGIN();
  NARROW(rp, row_ptr, n + 1);
  const int64_t nr = row_end - row_begin;
  SCRATCH32(rpo, nr + 1, true);
  SBX_TRY(sbx_permute_csr_rows(h, SBX_I32, vt, n, m, nnz, rp, col, val, row_order, col_order, row_begin, row_end, rpo,
                               col_out, val_out, out_capacity, shard_nnz_host));
  return sbx_widen_i32(h, rpo, row_ptr_out, nr + 1);
}

int sbx_mixed_permute_csr_rows_nnz(sbx_handle_t h, int64_t n, const void *row_ptr, const void *row_order, int64_t row_begin,
                                   int64_t row_end, int64_t *nnz_host) {
  I64_BEGIN();
  NARROW(rp, row_ptr, n + 1);
  SBX_TRY(sbx_i64_check(h, ovf));  // (the call does not state nnz)
  return sbx_permute_csr_rows_nnz(h, SBX_I32, n, rp, row_order, row_begin, row_end, nnz_host);
}
