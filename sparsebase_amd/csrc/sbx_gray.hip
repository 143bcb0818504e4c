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

constexpr int GR_INLINE = 16;  // rows up to this length are done by one lane

__device__ __forceinline__ uint64_t gray_decode(uint64_t g) {
  // prefix xor from the top: b = g ^ g>>1 ^ g>>2 ...
  g ^= g >> 1; g ^= g >> 2; g ^= g >> 4; g ^= g >> 8; g ^= g >> 16; g ^= g >> 32;
  return g;
}

struct GrayCounts {
  unsigned long long nnz_sparse, diag_sparse, nnz_dense, diag_dense;
};

// short rows: one lane per row (threshold is always 0 there because deg < resolution)
template <typename I>
__global__ __launch_bounds__(256) void k_gray_short(const I *__restrict__ rp, const I *__restrict__ col, int64_t n,
                                                    int64_t width, int64_t band, int bits, int nnz_threshold,
                                                    I *__restrict__ degree_out, uint64_t *__restrict__ key_out,
                                                    GrayCounts *__restrict__ counts, I *__restrict__ long_list,
                                                    unsigned *__restrict__ n_long) {
  __shared__ unsigned long long lds[256 / 64 + 1];
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned long long c_ns = 0, c_ds = 0, c_nd = 0, c_dd = 0;
  // wave-uniform trip count so that the aggregated append below sees all 64 lanes
  for (int64_t ib = i - sbx_lane(); ib < n; ib += stride) {
    i = ib + sbx_lane();
    const bool in = i < n;
    const I s = in ? rp[i] : 0, e = in ? rp[i + 1] : 0;
    const int64_t d = (int64_t)e - (int64_t)s;
    if (in) degree_out[i] = (I)d;
    const bool is_long = in && d > GR_INLINE;
    const unsigned slot = sbx_wave_append(n_long, is_long);
    if (is_long) long_list[slot] = (I)i;
    if (!in || is_long) continue;
    uint64_t bm = 0;
    unsigned in_band = 0;
    const int64_t thr = (d <= nnz_threshold) ? 0 : d / bits;  // == 0 unless bits < GR_INLINE
    if (thr == 0) {
      for (I j = s; j < e; j++) {
        const int64_t c = col[j];
        bm |= (uint64_t)1 << (c / width);
        const int64_t dist = c >= i ? c - i : i - c;
        in_band += dist <= band;
      }
    } else {
      // tiny resolution (m < 16 clamps it): count per block the slow way
      for (int b = 0; b < bits; b++) {
        int64_t cnt = 0;
        for (I j = s; j < e; j++) cnt += ((int64_t)col[j] / width) == b;
        if (cnt > thr) bm |= (uint64_t)1 << b;
      }
      for (I j = s; j < e; j++) {
        const int64_t c = col[j];
        const int64_t dist = c >= i ? c - i : i - c;
        in_band += dist <= band;
      }
    }
    key_out[i] = gray_decode(bm);
    if (d <= nnz_threshold) { c_ns += d; c_ds += in_band; }
    else { c_nd += d; c_dd += in_band; }
  }
  c_ns = sbx_block_sum<unsigned long long, 256>(c_ns, lds);
  c_ds = sbx_block_sum<unsigned long long, 256>(c_ds, lds);
  c_nd = sbx_block_sum<unsigned long long, 256>(c_nd, lds);
  c_dd = sbx_block_sum<unsigned long long, 256>(c_dd, lds);
  if (threadIdx.x == 0) {
    if (c_ns) atomicAdd(&counts->nnz_sparse, c_ns);
    if (c_ds) atomicAdd(&counts->diag_sparse, c_ds);
    if (c_nd) atomicAdd(&counts->nnz_dense, c_nd);
    if (c_dd) atomicAdd(&counts->diag_dense, c_dd);
  }
}

// longer rows: one wave per row, coalesced reads, per-wave block counters in LDS
template <typename I>
__global__ __launch_bounds__(256) void k_gray_long(const I *__restrict__ rp, const I *__restrict__ col, int64_t n,
                                                   int64_t width, int64_t band, int bits, int nnz_threshold,
                                                   uint64_t *__restrict__ key_out, GrayCounts *__restrict__ counts,
                                                   const I *__restrict__ long_list,
                                                   const unsigned *__restrict__ n_long) {
  __shared__ unsigned s_cnt[4][64];
  const int lane = sbx_lane(), wv = sbx_wave_in_block();
  const int64_t wave = (int64_t)blockIdx.x * 4 + wv;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  unsigned long long c_ns = 0, c_ds = 0, c_nd = 0, c_dd = 0;
  const int64_t total = *n_long;
  for (int64_t k = wave; k < total; k += nwaves) {
    const int64_t i = long_list[k];
    const I s = rp[i], e = rp[i + 1];
    const int64_t d = (int64_t)e - (int64_t)s;
    s_cnt[wv][lane] = 0;
    __builtin_amdgcn_wave_barrier();
    unsigned in_band = 0;
    for (int64_t j = (int64_t)s + lane; j < e; j += 64) {
      const int64_t c = col[j];
      atomicAdd(&s_cnt[wv][c / width], 1u);
      const int64_t dist = c >= i ? c - i : i - c;
      in_band += dist <= band;
    }
    __builtin_amdgcn_wave_barrier();
    const int64_t thr = (d <= nnz_threshold) ? 0 : d / bits;
    const bool set = lane < bits && (int64_t)s_cnt[wv][lane] > thr;
    const uint64_t bm = __ballot(set);
    in_band = sbx_wave_sum(in_band);
    if (lane == 0) {
      key_out[i] = gray_decode(bm);
      if (d <= nnz_threshold) { c_ns += d; c_ds += in_band; }
      else { c_nd += d; c_dd += in_band; }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (lane == 0) {
    if (c_ns) atomicAdd(&counts->nnz_sparse, c_ns);
    if (c_ds) atomicAdd(&counts->diag_sparse, c_ds);
    if (c_nd) atomicAdd(&counts->nnz_dense, c_nd);
    if (c_dd) atomicAdd(&counts->diag_dense, c_dd);
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
  GrayCounts *cnt = nullptr;
  int32_t *long_list = nullptr;
  unsigned *n_long = nullptr;
  SBX_TRY(sbx_salloc(h, 2, &cnt));  // second slot holds the long-row counter
  n_long = (unsigned *)(cnt + 1);
  {
    int64_t cap = nnz / (GR_INLINE + 1) + 1;
    if (cap > n) cap = n;
    SBX_TRY(sbx_salloc(h, (size_t)cap, &long_list));
  }
  SBX_HIP(h, hipMemsetAsync(cnt, 0, 2 * sizeof(GrayCounts), h->stream));
  const int64_t band = m / 128;  // :138
  SBX_KLAUNCH(h, SBX_K_GRAY, k_gray_short<int32_t>, dim3(sbx_grid_for(n, 256, 8192)), dim3(256),
                     (const int32_t *)row_ptr, (const int32_t *)col, n, width, band, bits, nnz_threshold,
                     (int32_t *)degree_out, key_out, cnt, long_list, n_long);
  SBX_KLAUNCH(h, SBX_K_GRAY, k_gray_long<int32_t>, dim3(sbx_grid_for(n, 4, (int64_t)h->num_cus * 8)), dim3(256),
                     (const int32_t *)row_ptr, (const int32_t *)col, n, width, band, bits, nnz_threshold, key_out, cnt,
                     (const int32_t *)long_list, (const unsigned *)n_long);
  SBX_LAUNCH_CHECK(h);
  GrayCounts hc;
  SBX_TRY(sbx_readback(h, &hc, cnt, sizeof(GrayCounts)));
  counts_host[0] = (int64_t)hc.nnz_sparse;
  counts_host[1] = (int64_t)hc.diag_sparse;
  counts_host[2] = (int64_t)hc.nnz_dense;
  counts_host[3] = (int64_t)hc.diag_dense;
  return SBX_OK;
}
