// sbx_degree.hip — DegreeReorder::CalculateReorderCSR (reorder/degree_reorder.cc:22-62).
//
// The reference is a counting sort that fills each degree bucket from its END in row-id order, i.e. the sequence
// (degree ascending, id descending); "descending" reverses the whole sequence.
//
// Degrees are small numbers with a thin tail (a power-law graph: half the rows empty, 98 % below 255 entries), so the
// rows are presented in descending id order and sorted by ONE stable 8-bit radix pass on min(degree, 255): that pass
// already is the final order of every row below 255 entries, and its scatter writes the answer itself
// (inv[id] = position, sbx_radix_sort_emit) instead of a sorted array.  The rows of the last bucket — in id order
// after the stable pass — are then sorted among themselves by their full degree (a few ten thousand rows: a small
// radix sort) and get their positions from a second scatter.  One read-back (rows in the last bucket, largest degree)
// sits behind the big pass, where the host would wait anyway.  8n + 4 algorithmic bytes; n = 4.2 M: 0.157 ms.  Templated on
// the index type: 64-bit row_ptr arrays are read as they are and the inverse permutation is written in 64 bits.
#include "sbx_device.h"
#include "sbx_internal.h"

namespace {

constexpr unsigned DG_TOP = 255;  // digit of every row with at least this many entries

struct DegState {
  unsigned n_top;    // rows with degree >= DG_TOP
  unsigned max_deg;
};

// key[j] = min(degree, DG_TOP) << 32 | id for id = n - 1 - j (descending id order).  Four consecutive keys per thread:
// the five row_ptr words behind them in one 16-byte load (4-byte aligned only: gfx950 loads unaligned vectors) and a
// 4-byte one, the keys out in two 16-byte stores (one key per thread and iteration, two dependent 4-byte loads each:
// 36 us for 4.2 M rows; this form 15).
template <typename I>
struct __attribute__((packed, aligned(4))) DgQuad {
  I a, b, c, d;
};
template <typename I>
__global__ __launch_bounds__(256) void k_degree_keys(const I *__restrict__ rp, uint64_t *__restrict__ key, int64_t n,
                                                     DegState *__restrict__ st) {
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // keys 4 q .. 4 q + 3
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned mx = 0, top = 0;
  for (; 4 * q < n; q += stride) {
    const int64_t j0 = 4 * q, u0 = n - 1 - j0;  // ids u0, u0 - 1, u0 - 2, u0 - 3
    unsigned d[4];
    const int cnt = n - j0 < 4 ? (int)(n - j0) : 4;
    if (cnt == 4) {
      const DgQuad<I> w = *(const DgQuad<I> *)(rp + u0 - 3);  // rp[u0 - 3 .. u0]
      const I hi = rp[u0 + 1];
      d[0] = (unsigned)(hi - w.d), d[1] = (unsigned)(w.d - w.c), d[2] = (unsigned)(w.c - w.b), d[3] = (unsigned)(w.b - w.a);
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) d[k] = k < cnt ? (unsigned)(rp[u0 - k + 1] - rp[u0 - k]) : 0u;
    }
    uint64_t kk[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      kk[k] = ((uint64_t)(d[k] < DG_TOP ? d[k] : DG_TOP) << 32) | (uint64_t)(uint32_t)(u0 - k);
      if (k < cnt) {
        mx = d[k] > mx ? d[k] : mx;
        top += d[k] >= DG_TOP;
      }
    }
    if (cnt == 4) {  // (key is 256-byte aligned scratch: 32-byte aligned stores)
      *(ulonglong2 *)(key + j0) = make_ulonglong2(kk[0], kk[1]);
      *(ulonglong2 *)(key + j0 + 2) = make_ulonglong2(kk[2], kk[3]);
    } else {
      for (int k = 0; k < cnt; k++) key[j0 + k] = kk[k];
    }
  }
  __shared__ unsigned s_mx[4], s_top[4];  // one atomic per workgroup: the result words are hot
  mx = sbx_wave_max(mx);
  top = sbx_wave_sum(top);
  if (sbx_lane() == 0) s_mx[sbx_wave_in_block()] = mx, s_top[sbx_wave_in_block()] = top;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 4; i++) mx = s_mx[i] > mx ? s_mx[i] : mx, top += s_top[i];
    if (mx >= DG_TOP) atomicMax(&st->max_deg, mx);
    if (top) atomicAdd(&st->n_top, top);
  }
}

// the last bucket: (degree, id) of its rows, in the order the stable pass left them (descending id)
template <typename I>
__global__ __launch_bounds__(256) void k_degree_tail_keys(const I *__restrict__ rp, const uint32_t *__restrict__ ids,
                                                          int64_t count, uint32_t *__restrict__ key,
                                                          uint32_t *__restrict__ id) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; t < count; t += stride) {
    const uint32_t u = ids[t];
    key[t] = (uint32_t)(rp[u + 1] - rp[u]);
    id[t] = u;
  }
}
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

template <typename I>
static int degree_reorder_typed(sbx_handle_t h, int64_t n, const void *row_ptr, int ascending, void *inv_perm_out) {
  SBX_TRY(sbx_arena_begin(h));
  if (n == 0) return SBX_OK;
  if (n == 1) {  // (the radix sort wants two keys)
    SBX_KLAUNCH(h, SBX_K_DEGREE, k_degree_one<I>, dim3(1), dim3(1), (I *)inv_perm_out);
    SBX_LAUNCH_CHECK(h);
    return SBX_OK;
  }
  const I *rp = (const I *)row_ptr;
  uint64_t *ka, *kb;
  uint32_t *sorted_id;
  DegState *st;
  SBX_TRY(sbx_salloc(h, (size_t)n, &ka));
  SBX_TRY(sbx_salloc(h, (size_t)n, &kb));
  SBX_TRY(sbx_salloc(h, (size_t)n, &sorted_id));
  SBX_TRY(sbx_salloc(h, 1, &st));
  SBX_HIP(h, hipMemsetAsync(st, 0, sizeof(DegState), h->stream));
  // (few workgroups: each ends with two adds on one line of DegState, ~7 ns apiece whoever issues them — with 4096
  // workgroups those adds, not the 64 MB the kernel moves, were its 52 us)
  const unsigned grid = sbx_grid_for((n + 3) / 4, 256, 512);
  SBX_KLAUNCH(h, SBX_K_DEGREE, k_degree_keys<I>, dim3(grid), dim3(256), rp, ka, n, st);
  SBX_LAUNCH_CHECK(h);
  // one pass over the digit in bits [32, 40); its scatter leaves the ids in order and inv[id] = position (in the
  // caller's index width: no 32-bit copy of a 64-bit row_ptr, no widening pass over the result)
  const sbx_radix_pass pass = {32, 8};
  sbx_radix_emit em;
  memset(&em, 0, sizeof(em));
  em.out = sorted_id;
  if (sizeof(I) == 4) em.pos_of = (unsigned *)inv_perm_out;
  else em.pos_of64 = (unsigned long long *)inv_perm_out;
  em.pos_flip = ascending ? 0u : (uint32_t)n;
  SBX_TRY(sbx_radix_sort_emit(h, ka, kb, n, &pass, 1, &em));
  DegState hs;
  SBX_TRY(sbx_readback(h, &hs, st, sizeof(hs)));
  const int64_t top = hs.n_top;
  if (top < 2) return SBX_OK;
  // the last bucket by full degree (stable: equal degrees keep their descending id order)
  uint32_t *ta, *tb, *ia, *ib;
  SBX_TRY(sbx_salloc(h, (size_t)top, &ta));
  SBX_TRY(sbx_salloc(h, (size_t)top, &tb));
  SBX_TRY(sbx_salloc(h, (size_t)top, &ia));
  SBX_TRY(sbx_salloc(h, (size_t)top, &ib));
  const unsigned tgrid = sbx_grid_for(top, 256, 2048);
  SBX_KLAUNCH(h, SBX_K_DEGREE, k_degree_tail_keys<I>, dim3(tgrid), dim3(256), rp, (const uint32_t *)(sorted_id + (n - top)),
              top, ta, ia);
  SBX_LAUNCH_CHECK(h);
  sbx_radix_pass passes[16];
  const int np = sbx_radix_plan(0, sbx_bits_for(hs.max_deg), 0, 0, passes);
  int in_b = 0;
  SBX_TRY(sbx_radix_sort(h, 4, 4, ta, tb, ia, ib, top, passes, np, &in_b));
  SBX_KLAUNCH(h, SBX_K_DEGREE, k_degree_tail_emit<I>, dim3(tgrid), dim3(256), (const uint32_t *)(in_b ? ib : ia),
              (I *)inv_perm_out, top, n - top, n, ascending);
  SBX_LAUNCH_CHECK(h);
  return SBX_OK;
}

extern "C" int sbx_degree_reorder(sbx_handle_t h, sbx_index_type it, int64_t n, const void *row_ptr, int ascending,
                                  void *inv_perm_out) {
  if (!h) return SBX_ERR_BAD_ARG;
  if (n < 0 || !row_ptr || (n > 0 && !inv_perm_out)) SBX_FAIL(h, SBX_ERR_BAD_ARG, "sbx_degree_reorder: bad argument");
  // (vertex ids travel in the low word of the sort keys and degrees in 32 bits: n and every row length below 2^32;
  // row_ptr VALUES — nnz — are not limited for 64-bit indices)
  if (n >= ((int64_t)1 << 31)) SBX_FAIL(h, SBX_ERR_UNSUPPORTED, "sbx_degree_reorder: dimension exceeds int32");
  return it == SBX_I64 ? degree_reorder_typed<int64_t>(h, n, row_ptr, ascending, inv_perm_out)
                       : degree_reorder_typed<int32_t>(h, n, row_ptr, ascending, inv_perm_out);
}
