// sbx_degree.hip — DegreeReorder::CalculateReorderCSR (reorder/degree_reorder.cc:22-62).
//
// The reference is a counting sort that fills each degree bucket from its END in
// row-id order, i.e. the sequence (degree ascending, id descending); "descending"
// reverses the whole sequence.  Here: keys = degrees presented in descending id
// order, one stable LSD radix sort over the significant degree bits, then the
// inversion inv[sorted[k]] = k (or n-1-k) is fused into the final scatter.
#include "sbx_device.h"
#include "sbx_internal.h"

namespace {

template <typename I>
__global__ __launch_bounds__(256) void k_degree_keys(const I *__restrict__ rp, uint32_t *__restrict__ key,
                                                     uint32_t *__restrict__ id, int64_t n,
                                                     unsigned *__restrict__ max_deg) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned mx = 0;
  for (; j < n; j += stride) {
    const int64_t u = n - 1 - j;  // descending id order
    const unsigned d = (unsigned)(rp[u + 1] - rp[u]);
    key[j] = d;
    id[j] = (uint32_t)u;
    mx = d > mx ? d : mx;
  }
  __shared__ unsigned s_mx[4];  // one atomic per workgroup: the result word is hot
  mx = sbx_wave_max(mx);
  if (sbx_lane() == 0) s_mx[sbx_wave_in_block()] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 4; i++) mx = s_mx[i] > mx ? s_mx[i] : mx;
    if (mx) atomicMax(max_deg, mx);
  }
}

template <typename I>
__global__ __launch_bounds__(256) void k_degree_invert(const uint32_t *__restrict__ sorted_id, I *__restrict__ inv,
                                                       int64_t n, int ascending) {
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; k < n; k += stride) inv[sorted_id[k]] = (I)(ascending ? k : n - 1 - k);
}

}  // namespace

extern "C" int sbx_degree_reorder(sbx_handle_t h, sbx_index_type it, int64_t n, const void *row_ptr, int ascending,
                                  void *inv_perm_out) {
  if (!h) return SBX_ERR_BAD_ARG;
  if (n < 0 || !row_ptr || (n > 0 && !inv_perm_out)) SBX_FAIL(h, SBX_ERR_BAD_ARG, "sbx_degree_reorder: bad argument");
  if (it == SBX_I64) return sbx_i64_degree_reorder(h, n, row_ptr, ascending, inv_perm_out);
  SBX_TRY(sbx_arena_begin(h));
  if (n == 0) return SBX_OK;
  uint32_t *ka, *kb, *ia, *ib;
  unsigned *mx;
  SBX_TRY(sbx_salloc(h, (size_t)n, &ka));
  SBX_TRY(sbx_salloc(h, (size_t)n, &kb));
  SBX_TRY(sbx_salloc(h, (size_t)n, &ia));
  SBX_TRY(sbx_salloc(h, (size_t)n, &ib));
  SBX_TRY(sbx_salloc(h, 1, &mx));
  SBX_HIP(h, hipMemsetAsync(mx, 0, sizeof(unsigned), h->stream));
  const unsigned grid = sbx_grid_for(n, 256, 1024);
  SBX_KLAUNCH(h, SBX_K_DEGREE, k_degree_keys<int32_t>, dim3(grid), dim3(256), (const int32_t *)row_ptr, ka, ia, n,
                     mx);
  SBX_LAUNCH_CHECK(h);
  unsigned max_deg = 0;
  SBX_TRY(sbx_readback(h, &max_deg, mx, sizeof(unsigned)));
  sbx_radix_pass passes[16];
  const int np = sbx_radix_plan(0, sbx_bits_for(max_deg), 0, 0, passes);
  int in_b = 0;
  SBX_TRY(sbx_radix_sort(h, 4, 4, ka, kb, ia, ib, n, passes, np, &in_b));
  SBX_KLAUNCH(h, SBX_K_DEGREE, k_degree_invert<int32_t>, dim3(grid), dim3(256), (const uint32_t *)(in_b ? ib : ia),
                     (int32_t *)inv_perm_out, n, ascending);
  SBX_LAUNCH_CHECK(h);
  return SBX_OK;
}
