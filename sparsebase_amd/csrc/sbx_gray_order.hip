// sbx_gray_order.hip — sbx_gray_reorder: GrayReorder with its ordering stage on the device (opt-in, stable ties).
//
//   A8  reorder/gray_reorder.cc:106-424
//
// The reference orders the row keys with unstable std::sort calls on heavily tied keys (:199-203 by degree, :293-301 and
// :354-360 the sections by decoded key — ascending and descending in turn —, :404 the dense rows), so its output depends
// on libstdc++'s introsort; the host layer's GrayReorder therefore issues those very calls over the device-computed keys
// (exact mode, the default: DESIGN.md section 5).  This entry point is the other mode SURVEY section 8(b) sketches
// (`exact_ties = 0`): every one of those sorts as a STABLE sort, which makes the whole ordering one lexicographic order
//
//     (class, section, +-key, degree, row id)         class: sparse rows (degree <= nnz_threshold) in front of dense ones
//
// and that is what three stable radix sorts of (key, row id) pairs produce, last criterion first: by degree, by the
// signed key, by (class, section).  Sections are what the reference's walk over the degree-sorted sparse rows finds
// (:271-329): empty rows in front, then runs of `group_size` distinct degrees; section k sorts ascending for even k and
// descending for odd k (a descending stable sort is an ascending one on the complemented key).  A "highly banded" class
// (:181-190) keeps the reference's early-outs: sparse rows by degree only, dense rows in their original order.
// Wherever the reference's comparators decide the order — no two rows of a section with equal keys — the result is the
// reference's; among tied rows it is the stable order instead of introsort's.
//
// ONE sort where the criteria fit one word (round 6): class / section, signed key and degree are fields of a single
// 64-bit key built in row order — no gathers through a half-sorted id list between the sorts, one histogram, one
// read-back less (the number of sections only sizes the class field: its bound, ceil(threshold / group_size), does) —
// and (key, row id) pairs go through the radix sort once: five 8-bit passes for the bench line's parameters (3 + 32 + 4
// bits) where the three sorts took six, with three histograms and three gather kernels around them.  Parameters whose
// fields need more than 64 bits (resolution 64; a threshold in the millions) keep the three sorts.
#include "sbx_device.h"
#include "sbx_internal.h"

namespace {

struct NestGuard {  // the nested entry point must not rewind the caller's scratch
  sbx_handle_t h;
  explicit NestGuard(sbx_handle_t h_) : h(h_) { h->nest++; }
  ~NestGuard() { if (h->nest > 0) h->nest--; }
};

// pass 1: (degree key, row id) pairs — the dense rows' degree is no criterion — and the degrees the sparse rows have
template <typename I>
__global__ __launch_bounds__(256) void k_go_degree_keys(const I *__restrict__ deg, int64_t n, int64_t thr,
                                                        uint32_t *__restrict__ key, uint32_t *__restrict__ id,
                                                        uint32_t *__restrict__ present) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const int64_t d = (int64_t)deg[i];
    const bool sparse = d <= thr;
    key[i] = sparse ? (uint32_t)d : 0u;
    id[i] = (uint32_t)i;
    if (sparse && d > 0 && present[d] == 0) present[d] = 1;  // (a benign race: every writer stores 1)
  }
}

// section of a sparse degree d >= 1: 1 + (number of smaller degrees present) / group_size  (0: the empty rows);
// `rank` = exclusive scan of `present`
__global__ __launch_bounds__(256) void k_go_sections(const uint32_t *__restrict__ present, const uint32_t *__restrict__ rank,
                                                     int64_t count, uint32_t group_size, uint32_t *__restrict__ section) {
  const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= count) return;
  section[d] = (d > 0 && present[d]) ? 1u + rank[d] / group_size : 0u;
}

// pass 2: the signed key of the row at every position of the degree-sorted list
template <typename I>
__global__ __launch_bounds__(256) void k_go_signed_keys(const uint32_t *__restrict__ id, const I *__restrict__ deg,
                                                        const uint64_t *__restrict__ gkey, int64_t n, int64_t thr,
                                                        const uint32_t *__restrict__ section, int sparse_banded,
                                                        int dense_banded, uint64_t mask, uint64_t *__restrict__ key) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; p < n; p += stride) {
    const uint32_t r = id[p];
    const int64_t d = (int64_t)deg[r];
    uint64_t k = 0;
    if (d > thr) {
      if (!dense_banded) k = gkey[r] & mask;
    } else if (d > 0 && !sparse_banded) {
      const uint32_t s = section[d] - 1u;  // index of the section's sort call
      k = (s & 1u) ? (~gkey[r]) & mask : gkey[r] & mask;
    }
    key[p] = k;
  }
}

// pass 3: (class, section)
template <typename I>
__global__ __launch_bounds__(256) void k_go_class_keys(const uint32_t *__restrict__ id, const I *__restrict__ deg,
                                                       int64_t n, int64_t thr, const uint32_t *__restrict__ section,
                                                       int sparse_banded, uint32_t dense_class,
                                                       uint32_t *__restrict__ key) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; p < n; p += stride) {
    const int64_t d = (int64_t)deg[id[p]];
    key[p] = d > thr ? dense_class : (sparse_banded ? 0u : section[d]);
  }
}

template <typename I>
__global__ __launch_bounds__(256) void k_go_emit(const uint32_t *__restrict__ id, int64_t n, I *__restrict__ inv) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; p < n; p += stride) inv[id[p]] = (I)p;
}

// the degrees the sparse rows have (the one-sort path: the sections must be known before any key is built)
template <typename I>
__global__ __launch_bounds__(256) void k_go_present(const I *__restrict__ deg, int64_t n, int64_t thr,
                                                    uint32_t *__restrict__ present) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const int64_t d = (int64_t)deg[i];
    if (d <= thr && d > 0 && present[d] == 0) present[d] = 1;  // (a benign race: every writer stores 1)
  }
}

// one key per row, in row order: class / section << (kbits + dbits) | signed key << dbits | degree (sparse rows)
template <typename I>
__global__ __launch_bounds__(256) void k_go_composite(const I *__restrict__ deg, const uint64_t *__restrict__ gkey, int64_t n,
                                                      int64_t thr, const uint32_t *__restrict__ section, int sparse_banded,
                                                      int dense_banded, uint64_t mask, uint32_t dense_class, int kbits,
                                                      int dbits, int idbits, uint64_t *__restrict__ key,
                                                      uint32_t *__restrict__ id) {
  // id == nullptr: the row id is the key's low field (idbits wide) and the sort has no payload
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const int64_t d = (int64_t)deg[i];
    uint64_t cls, k = 0, dk = 0;
    if (d > thr) {
      cls = dense_class;
      if (!dense_banded) k = gkey[i] & mask;
    } else {
      const uint32_t sec = sparse_banded ? 0u : section[d];
      cls = sec;
      dk = (uint64_t)d;
      if (d > 0 && !sparse_banded) k = ((sec - 1u) & 1u) ? (~gkey[i]) & mask : gkey[i] & mask;
    }
    if (kbits == 0) k = 0;  // (both classes "highly banded": the keys are no criterion)
    const uint64_t comp = (cls << (kbits + dbits)) | (k << dbits) | dk;
    if (id) {
      key[i] = comp;
      id[i] = (uint32_t)i;
    } else {
      key[i] = (comp << idbits) | (uint64_t)i;
    }
  }
}

template <typename I>
__global__ __launch_bounds__(256) void k_go_widen(const uint32_t *__restrict__ in, int64_t n, I *__restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = (I)in[i];
}

static bool gray_three_sorts() {  // SBX_GRAY_ORDER_THREE_SORTS=1: the ordering keeps its three sorts whatever the parameters (tests, A/B)
  static const bool on = sbx_env_test("SBX_GRAY_ORDER_THREE_SORTS") && atoi(sbx_env_test("SBX_GRAY_ORDER_THREE_SORTS")) != 0;
  return on;
}

template <typename I>
int gray_order_one_sort(sbx_handle_t h, int64_t n, const I *deg, const uint64_t *gkey, bool sparse_banded, bool dense_banded,
                        int kbits, int dbits, int cbits, uint32_t dense_class, int64_t dmax, int64_t thr, int group_size,
                        uint64_t mask, I *inv_out) {
  const unsigned grid = sbx_grid_for(n, 256, (int64_t)h->num_cus * 16);
  uint32_t *ia = nullptr, *ib = nullptr, *present = nullptr, *rank = nullptr, *section = nullptr;
  uint64_t *qa = nullptr, *qb = nullptr;
  SBX_TRY(sbx_salloc(h, (size_t)n, &ia));
  SBX_TRY(sbx_salloc(h, (size_t)n, &ib));
  SBX_TRY(sbx_salloc(h, (size_t)n, &qa));
  SBX_TRY(sbx_salloc(h, (size_t)n, &qb));
  SBX_TRY(sbx_salloc(h, (size_t)dmax + 2, &section));
  if (!sparse_banded && dmax > 0) {  // the sections of the degrees (a "highly banded" sparse class has none)
    SBX_TRY(sbx_salloc(h, (size_t)dmax + 2, &present));
    SBX_TRY(sbx_salloc(h, (size_t)dmax + 2, &rank));
    SBX_HIP(h, hipMemsetAsync(present, 0, sizeof(uint32_t) * (size_t)(dmax + 2), h->stream));
    SBX_KLAUNCH(h, SBX_K_GRAY, k_go_present<I>, dim3(grid), dim3(256), deg, n, thr, present);
    SBX_TRY(sbx_exclusive_scan_u32(h, present, rank, dmax + 1, nullptr));
    SBX_KLAUNCH(h, SBX_K_GRAY, k_go_sections, dim3((unsigned)((dmax + 1 + 255) / 256)), dim3(256), (const uint32_t *)present,
                (const uint32_t *)rank, dmax + 1, (uint32_t)group_size, section);
  } else {
    SBX_HIP(h, hipMemsetAsync(section, 0, sizeof(uint32_t) * (size_t)(dmax + 2), h->stream));
  }
  sbx_radix_pass passes[16];
  // Where the row id fits below the criteria (bench line: 39 + 22 bits) it is the key's low field: the sort moves 8-byte
  // keys without a payload, and its last pass writes inv[id] = position itself (sbx_radix_emit::pos_of) — no emit kernel
  const int idbits = sbx_bits_for((uint64_t)(n - 1));
  if (cbits + kbits + dbits + idbits <= 64) {
    SBX_KLAUNCH(h, SBX_K_GRAY, k_go_composite<I>, dim3(grid), dim3(256), deg, gkey, n, thr, (const uint32_t *)section,
                sparse_banded ? 1 : 0, dense_banded ? 1 : 0, mask, dense_class, kbits, dbits, idbits, qa, (uint32_t *)nullptr);
    SBX_LAUNCH_CHECK(h);
    const int np = sbx_radix_plan(idbits, idbits + cbits + kbits + dbits, 0, 0, passes);
    const sbx_radix_emit em = {nullptr, ia, nullptr, nullptr, sizeof(I) == 4 ? (unsigned *)inv_out : (unsigned *)ib,
                               idbits < 32 ? (1u << idbits) - 1u : 0u};
    h->rs_tied_hint = true;  // (class and degree fields of a handful of values, Gray keys shared by many rows)
    const int rc_sort = sbx_radix_sort_emit(h, qa, qb, n, passes, np, &em);
    h->rs_tied_hint = false;
    SBX_TRY(rc_sort);
    if (sizeof(I) != 4) {
      SBX_KLAUNCH(h, SBX_K_GRAY, k_go_widen<I>, dim3(grid), dim3(256), (const uint32_t *)ib, n, inv_out);
      SBX_LAUNCH_CHECK(h);
    }
    return SBX_OK;
  }
  SBX_KLAUNCH(h, SBX_K_GRAY, k_go_composite<I>, dim3(grid), dim3(256), deg, gkey, n, thr, (const uint32_t *)section,
              sparse_banded ? 1 : 0, dense_banded ? 1 : 0, mask, dense_class, kbits, dbits, 0, qa, ia);
  SBX_LAUNCH_CHECK(h);
  int in_b = 0;
  const int np = sbx_radix_plan(0, cbits + kbits + dbits, 0, 0, passes);
  h->rs_tied_hint = true;
  const int rc_sort = sbx_radix_sort(h, 8, 4, qa, qb, ia, ib, n, passes, np, &in_b);
  h->rs_tied_hint = false;
  SBX_TRY(rc_sort);
  SBX_KLAUNCH(h, SBX_K_GRAY, k_go_emit<I>, dim3(grid), dim3(256), (const uint32_t *)(in_b ? ib : ia), n, inv_out);
  SBX_LAUNCH_CHECK(h);
  return SBX_OK;
}

template <typename I>
int gray_order_typed(sbx_handle_t h, int64_t n, int64_t nnz, const I *deg, const uint64_t *gkey, const int64_t *counts,
                     int bits, int64_t thr, int group_size, I *inv_out) {
  // gray_reorder.cc:181-190 (the reference keeps the counters in `int`)
  const bool sparse_banded = double((int)counts[1]) / (int)counts[0] > 0.3;
  const bool dense_banded = double((int)counts[3]) / (int)counts[2] > 0.2;
  if (thr < 0) thr = -1;  // no row is sparse
  // a sparse row's degree is at most the threshold (and, without duplicate columns, m; the tables cover min(threshold, nnz))
  const int64_t dmax = thr < 0 ? 0 : (thr < nnz ? thr : nnz);
  if (dmax > ((int64_t)1 << 28))
    SBX_FAIL(h, SBX_ERR_UNSUPPORTED, "sbx_gray_reorder: nnz_threshold %lld: the degree tables would need %lld entries",
             (long long)thr, (long long)dmax);
  {
    // the one-sort form: do the three fields fit a word?  (at most dmax degrees are present: that many sections at most)
    const int64_t gs = group_size > 0 ? group_size : 1;
    const uint32_t dense_class_1 = sparse_banded ? 1u : (uint32_t)((dmax + gs - 1) / gs) + 1u;
    const int cbits = sbx_bits_for((uint64_t)dense_class_1), dbits1 = sbx_bits_for((uint64_t)dmax);
    const int kbits = (sparse_banded && dense_banded) ? 0 : bits;
    if (cbits + kbits + dbits1 <= 64 && !gray_three_sorts())
      return gray_order_one_sort<I>(h, n, deg, gkey, sparse_banded, dense_banded, kbits, dbits1, cbits, dense_class_1, dmax,
                                    thr, (int)gs, bits >= 64 ? ~0ull : ((1ull << bits) - 1ull), inv_out);
  }
  const unsigned grid = sbx_grid_for(n, 256, (int64_t)h->num_cus * 16);
  uint32_t *ka = nullptr, *kb = nullptr, *ia = nullptr, *ib = nullptr, *present = nullptr, *rank = nullptr, *section = nullptr;
  uint64_t *qa = nullptr, *qb = nullptr;
  SBX_TRY(sbx_salloc(h, (size_t)n, &ka));
  SBX_TRY(sbx_salloc(h, (size_t)n, &kb));
  SBX_TRY(sbx_salloc(h, (size_t)n, &ia));
  SBX_TRY(sbx_salloc(h, (size_t)n, &ib));
  SBX_TRY(sbx_salloc(h, (size_t)n, &qa));
  SBX_TRY(sbx_salloc(h, (size_t)n, &qb));
  SBX_TRY(sbx_salloc(h, (size_t)dmax + 2, &present));
  SBX_TRY(sbx_salloc(h, (size_t)dmax + 2, &rank));
  SBX_TRY(sbx_salloc(h, (size_t)dmax + 2, &section));
  SBX_HIP(h, hipMemsetAsync(present, 0, sizeof(uint32_t) * (size_t)(dmax + 2), h->stream));
  sbx_radix_pass passes[16];
  int in_b = 0;
  // 1: by degree (sparse rows; stable: ids ascending inside a degree)
  SBX_KLAUNCH(h, SBX_K_GRAY, k_go_degree_keys<I>, dim3(grid), dim3(256), deg, n, thr, ka, ia, present);
  SBX_LAUNCH_CHECK(h);
  uint32_t *id = ia, *id_tmp = ib;
  const int dbits = sbx_bits_for((uint64_t)dmax);
  if (dbits > 0) {
    const int np = sbx_radix_plan(0, dbits, 0, 0, passes);
    SBX_TRY(sbx_radix_sort(h, 4, 4, ka, kb, ia, ib, n, passes, np, &in_b));
    if (in_b) id = ib, id_tmp = ia;
  }
  // the sections of the degrees
  uint32_t n_present = 0, *tot = nullptr;
  SBX_TRY(sbx_salloc(h, 1, &tot));
  SBX_TRY(sbx_exclusive_scan_u32(h, present, rank, dmax + 1, tot));
  SBX_TRY(sbx_readback(h, &n_present, tot, sizeof(uint32_t)));
  SBX_KLAUNCH(h, SBX_K_GRAY, k_go_sections, dim3((unsigned)((dmax + 1 + 255) / 256)), dim3(256), (const uint32_t *)present,
              (const uint32_t *)rank, dmax + 1, (uint32_t)(group_size > 0 ? group_size : 1), section);
  const uint32_t n_sections = sparse_banded ? 0u : (n_present + (uint32_t)(group_size > 0 ? group_size : 1) - 1u) / (uint32_t)(group_size > 0 ? group_size : 1);
  // 2: by the signed key
  const uint64_t mask = bits >= 64 ? ~0ull : ((1ull << bits) - 1ull);
  if (!(sparse_banded && dense_banded)) {
    SBX_KLAUNCH(h, SBX_K_GRAY, k_go_signed_keys<I>, dim3(grid), dim3(256), (const uint32_t *)id, deg, gkey, n, thr,
                (const uint32_t *)section, sparse_banded ? 1 : 0, dense_banded ? 1 : 0, mask, qa);
    SBX_LAUNCH_CHECK(h);
    const int np = sbx_radix_plan(0, bits, 0, 0, passes);
    SBX_TRY(sbx_radix_sort(h, 8, 4, qa, qb, id, id_tmp, n, passes, np, &in_b));
    if (in_b) { uint32_t *t = id; id = id_tmp; id_tmp = t; }
  }
  // 3: by (class, section)
  const uint32_t dense_class = n_sections + 1u;
  SBX_KLAUNCH(h, SBX_K_GRAY, k_go_class_keys<I>, dim3(grid), dim3(256), (const uint32_t *)id, deg, n, thr,
              (const uint32_t *)section, sparse_banded ? 1 : 0, dense_class, ka);
  SBX_LAUNCH_CHECK(h);
  {
    const int np = sbx_radix_plan(0, sbx_bits_for((uint64_t)dense_class), 0, 0, passes);
    SBX_TRY(sbx_radix_sort(h, 4, 4, ka, kb, id, id_tmp, n, passes, np, &in_b));
    if (in_b) { uint32_t *t = id; id = id_tmp; id_tmp = t; }
  }
  SBX_KLAUNCH(h, SBX_K_GRAY, k_go_emit<I>, dim3(grid), dim3(256), (const uint32_t *)id, n, inv_out);
  SBX_LAUNCH_CHECK(h);
  return SBX_OK;
}

}  // namespace

extern "C" int sbx_gray_reorder(sbx_handle_t h, sbx_index_type it, int64_t n, int64_t m, int64_t nnz,
                                const void *row_ptr, const void *col, int resolution, int nnz_threshold,
                                int group_size, int exact_ties, void *inv_perm_out) {
  if (!h) return SBX_ERR_BAD_ARG;
  if (it == SBX_I32_N64)
    return sbx_mixed_gray_reorder(h, n, m, nnz, row_ptr, col, resolution, nnz_threshold, group_size, exact_ties, inv_perm_out);
  if (n < 0 || m < 0 || !row_ptr || (n > 0 && !inv_perm_out) || (nnz > 0 && !col) || group_size < 1)
    SBX_FAIL(h, SBX_ERR_BAD_ARG, "sbx_gray_reorder: bad argument");
  if (exact_ties)
    SBX_FAIL(h, SBX_ERR_UNSUPPORTED,
             "sbx_gray_reorder: the reference's tie order is libstdc++'s introsort visiting order and is reproduced by "
             "the host layer (reorder::GrayReorder over sbx_gray_row_keys); this entry point orders ties stably");
  if (n >= ((int64_t)1 << 32)) SBX_FAIL(h, SBX_ERR_UNSUPPORTED, "sbx_gray_reorder: more than 2^32 rows");
  SBX_TRY(sbx_arena_begin(h));
  if (n == 0) return SBX_OK;
  if (n == 1) {
    SBX_HIP(h, hipMemsetAsync(inv_perm_out, 0, it == SBX_I64 ? 8 : 4, h->stream));
    return SBX_OK;
  }
  NestGuard guard(h);
  int bits = resolution;
  if (m < bits) bits = (int)m;  // gray_reorder.cc:206-208
  void *deg = nullptr;
  uint64_t *gkey = nullptr;
  SBX_TRY(sbx_arena_alloc(h, (size_t)n * (it == SBX_I64 ? 8 : 4), &deg));
  SBX_TRY(sbx_salloc(h, (size_t)n, &gkey));
  int64_t counts[4] = {0, 0, 0, 0};
  SBX_TRY(sbx_gray_row_keys(h, it, n, m, nnz, row_ptr, col, resolution, nnz_threshold, deg, gkey, counts));
  if (it == SBX_I64)
    return gray_order_typed<int64_t>(h, n, nnz, (const int64_t *)deg, gkey, counts, bits, (int64_t)nnz_threshold, group_size,
                                     (int64_t *)inv_perm_out);
  return gray_order_typed<int32_t>(h, n, nnz, (const int32_t *)deg, gkey, counts, bits, (int64_t)nnz_threshold, group_size,
                                   (int32_t *)inv_perm_out);
}
