// sbx_internal.h — handle, scratch arena and launch helpers shared by the HIP
// translation units of libsbx.  gfx950 (MI355X) only: wave64, 160 KiB LDS/CU.
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "sbx.h"

#define SBX_WAVE 64

// kernel groups for the optional HIP-event profiler (sbx_profile_*)
enum sbx_kernel_id {
  SBX_K_SCAN = 0,
  SBX_K_RADIX_HIST,
  SBX_K_RADIX_SCATTER,
  SBX_K_COO_TO_CSR,
  SBX_K_CSR_TO_COO,
  SBX_K_PERMUTE_TILE,
  SBX_K_PERMUTE_LONG,
  SBX_K_PERMUTE_BLOCK,
  SBX_K_PERMUTE_PREP,
  SBX_K_BFS_EXPAND,
  SBX_K_BFS_HEAVY,
  SBX_K_BFS_BOTTOMUP,
  SBX_K_BFS_SMALL,
  SBX_K_LEVEL_ORDER,
  SBX_K_CC,
  SBX_K_RCM_SMALL,
  SBX_K_RCM_MISC,
  SBX_K_GRAY,
  SBX_K_DEGREE,
  SBX_K_CHECK,
  SBX_K_CSC,
  SBX_K_FEATURE,
  SBX_K_MTX,
  SBX_K_MISC,
  SBX_K_COUNT
};
extern const char *const sbx_kernel_names[SBX_K_COUNT];

struct sbx_prof_rec {
  int kid;
  hipEvent_t start, stop;
};

struct sbx_block {
  char *ptr;
  size_t cap;
};

#define SBX_AUX_STREAMS 7  /* permute: [0] row classes (serial mode) / class 0, [1] long rows, [2..6] classes 1..5 */

struct sbx_handle_s {
  int device;
  hipStream_t stream;
  // grow-only scratch arena: a list of device blocks, bump-allocated per call.
  // After a call that needed more than one block the arena is consolidated
  // into a single block, so steady-state calls never reach hipMalloc.
  std::vector<sbx_block> blocks;
  size_t cur_block;
  size_t cur_off;
  size_t call_bytes;  // bytes handed out during the current call
  size_t high_water;
  int nest;  // > 0 while an API entry point calls another one: the arena is not rewound
  void *pinned;  // small pinned host buffer for device->host read-backs
  size_t pinned_bytes;
  // pool of pre-zeroed (digit histogram + ticket) slots for the radix sort: one memset per
  // SBX_RS_SLOTS sorts instead of one per sort (every memset is a launch on the critical path)
  void *rs_pool;
  int rs_next;
  void *pow5;       // device table of 5^k for the exact decimal conversion (sbx_mtx.hip), built on first use
  unsigned rb_seq;  // sequence number of the last polled read-back (sbx_readback)
  bool rb_poll;     // SBX_READBACK_POLL=0 selects the copy-engine path
  sbx_oom_hook oom_hook;  // asked once when a device allocation of the library's own fails (sbx_set_oom_hook)
  void *oom_user;
  int rcm_gb_backoff;  // RCM calls left to run without the persistent (grid-barrier) kernels after one of them gave up
  int num_cus;
  // side streams for independent stages of one call (permute: tile / block-row / long-row paths), created on
  // first use; aux_event[0] marks the fork point on `stream`, [1 + i] the end of side stream i
  hipStream_t aux_stream[SBX_AUX_STREAMS];
  hipEvent_t aux_event[SBX_AUX_STREAMS + 1];
  bool aux_ready;
  bool aux_dirty;     // a side stream may still be running work of a call that returned early (error path)
  bool rs_tied_hint;  // set around a sort whose keys are heavily tied (Gray's composite keys): k_onesweep_hist aggregates per wave
  void *rs_override;  // next radix sort takes this zeroed slot instead of one from the pool (sorts on a side stream
                      // must not share the pool with the main stream: the pool is re-zeroed in stream order)
  // profiler: when on, every kernel launch is bracketed by HIP events on the
  // handle's stream; sbx_profile_query drains them into the accumulators
  bool prof_on;
  std::vector<sbx_prof_rec> prof_pending;
  std::vector<hipEvent_t> prof_pool;
  double prof_ms[SBX_K_COUNT];
  long long prof_launches[SBX_K_COUNT];
  long long prof_bytes[SBX_K_COUNT];  // algorithmic bytes declared by the launch sites (sbx_prof_bytes)
  char err[512];
};

int sbx_aux_streams(sbx_handle_t h);  // creates aux_stream / aux_event if needed
void sbx_prof_begin(sbx_handle_t h, int kid);
void sbx_prof_end(sbx_handle_t h);

// every kernel launch of the library goes through this macro
#define SBX_KLAUNCH(h, kid, kernel, grid, block, ...)                        \
  do {                                                                       \
    if ((h)->prof_on) sbx_prof_begin((h), (kid));                            \
    hipLaunchKernelGGL(kernel, grid, block, 0, (h)->stream, __VA_ARGS__);    \
    if ((h)->prof_on) sbx_prof_end((h));                                     \
  } while (0)

// algorithmic bytes of the launches just issued for kernel group `kid` (profiling runs only)
#define SBX_PROF_BYTES(h, kid, bytes)                          \
  do {                                                         \
    if ((h)->prof_on) (h)->prof_bytes[(kid)] += (long long)(bytes); \
  } while (0)

#define SBX_FAIL(h, code, ...)                       \
  do {                                               \
    snprintf((h)->err, sizeof((h)->err), __VA_ARGS__); \
    return (code);                                   \
  } while (0)

#define SBX_HIP(h, expr)                                                                  \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) {                                                               \
      snprintf((h)->err, sizeof((h)->err), "%s:%d: %s -> %s", __FILE__, __LINE__, #expr,  \
               hipGetErrorString(e_));                                                    \
      return e_ == hipErrorOutOfMemory ? SBX_ERR_OOM : SBX_ERR_HIP;                       \
    }                                                                                     \
  } while (0)

#define SBX_TRY(expr)            \
  do {                           \
    int rc_ = (expr);            \
    if (rc_ != SBX_OK) return rc_; \
  } while (0)

#define SBX_LAUNCH_CHECK(h) SBX_HIP(h, hipGetLastError())

// ---- arena -----------------------------------------------------------------
int sbx_arena_begin(sbx_handle_t h);  // start of an API call: rewinds (and consolidates)
int sbx_arena_alloc(sbx_handle_t h, size_t bytes, void **out);

template <typename T>
static inline int sbx_salloc(sbx_handle_t h, size_t count, T **out) {
  void *p = nullptr;
  int rc = sbx_arena_alloc(h, count * sizeof(T), &p);
  *out = (T *)p;
  return rc;
}

#define SBX_RS_SLOTS 32
#define SBX_RS_SLOT_BYTES (8 * 256 * 8 + 256)  /* RS_MAX_PASSES x 256 u64 counts + tickets */
// next zeroed radix slot (device pointer); re-zeroes the pool in stream order when it wraps
int sbx_radix_slot(sbx_handle_t h, void **slot);

// blocking read-back of `bytes` (<= pinned buffer) from device memory
int sbx_readback(sbx_handle_t h, void *dst_host, const void *src_dev, size_t bytes);

// ---- environment switches ------------------------------------------------------------------------------------------------
// Two kinds.  sbx_env_test(): what the test-suite needs to reach a code path that production takes only under conditions a
// test cannot arrange (a grid barrier giving up, a level beyond a size limit, every row on the radix fallback) — always
// read; every one of them selects code that also runs without the switch.  sbx_env_tuning(): tuning ranges and
// diagnostics — read only in builds with -DSBX_TUNING (python -m sparsebase_amd.build --tuning -> libsbx_tuning.so); the
// product library does not look at them.
static inline const char *sbx_env_test(const char *name) { return getenv(name); }
static inline const char *sbx_env_tuning(const char *name) {
#ifdef SBX_TUNING
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

static inline int sbx_value_bytes(sbx_value_type vt) {
  switch (vt) {
    case SBX_V_NONE: return 0;
    case SBX_V_I32: case SBX_V_U32: case SBX_V_F32: return 4;
    case SBX_V_I64: case SBX_V_U64: case SBX_V_F64: return 8;
  }
  return -1;
}
static inline int sbx_index_bytes(sbx_index_type it) { return it == SBX_I32 ? 4 : 8; }

static inline int sbx_bits_for(uint64_t max_value) {  // bits needed to represent values 0..max_value
  int b = 0;
  while (max_value) { b++; max_value >>= 1; }
  return b;
}

static inline unsigned sbx_grid_for(int64_t work_items, int per_block, int64_t cap) {
  int64_t g = (work_items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (unsigned)g;
}

// ---- primitives implemented in sbx_prims.hip --------------------------------
// exclusive prefix sum; in == out allowed.  If total_out != nullptr the grand
// total is stored there (device pointer).  T in {int32,int64,uint32,uint64}.
int sbx_exclusive_scan_i32(sbx_handle_t h, const int32_t *in, int32_t *out, int64_t count,
                           int32_t *total_out);
int sbx_exclusive_scan_i64(sbx_handle_t h, const int64_t *in, int64_t *out, int64_t count,
                           int64_t *total_out);
int sbx_exclusive_scan_u32(sbx_handle_t h, const uint32_t *in, uint32_t *out, int64_t count,
                           uint32_t *total_out);

// stable sort of every segment [seg_ptr[i], seg_ptr[i+1]) of (32-bit key < key_limit, vb-byte value) pairs by key, out
// of place (sbx_permute.hip: the permute's LDS sort stage with identity maps); the caller has begun the arena
int sbx_sort_segments(sbx_handle_t h, int vb, int64_t nseg, int64_t key_limit, int64_t nnz, const int32_t *seg_ptr,
                      const int32_t *key_in, const char *val_in, int32_t *key_out, char *val_out);

// A radix pass covers key bits [shift, shift+bits), bits <= 8.
struct sbx_radix_pass {
  int shift;
  int bits;
};
// builds the pass list for significant bit ranges [lo0,hi0) u [lo1,hi1) (second may be empty)
int sbx_radix_plan(int lo0, int hi0, int lo1, int hi1, sbx_radix_pass *passes /*>=16*/);

// Stable LSD radix sort.  keys_a/vals_a hold the input; *_b are same-sized
// temporaries.  payload_bytes in {0,4,8}; key_bytes in {4,8}.  On return
// *result_in_b tells which buffer holds the sorted sequence.
int sbx_radix_sort(sbx_handle_t h, int key_bytes, int payload_bytes, void *keys_a, void *keys_b,
                   void *vals_a, void *vals_b, int64_t count, const sbx_radix_pass *passes,
                   int num_passes, int *result_in_b);

// The same sort reading its records from, and leaving them in, the caller's own arrays: the first pass loads from
// `src`, the last one stores to `dst` (which may be the arrays of `src`: needs num_passes >= 2), the passes between
// use (keys_a, vals_a) / (keys_b, vals_b) — no pack kernel before and no unpack kernel behind the sort.  A side is
//   k_split: a 64-bit key kept as two 32-bit arrays, k[0] = low word, k[1] = high word (else k[0] = the key array)
//   p_split: a 64-bit payload kept as two 32-bit arrays, p[0] = low, p[1] = high    (else p[0] = the payload array)
// Supported: k_split with 8-byte keys (payload 0 / 4 / 8 bytes, not split), p_split with 4-byte keys and 8-byte payloads.
struct sbx_radix_side {
  void *k[2];
  void *p[2];
  bool k_split, p_split;
};
int sbx_radix_sort_io(sbx_handle_t h, int key_bytes, int payload_bytes, const sbx_radix_side *src, void *keys_a,
                      void *keys_b, void *vals_a, void *vals_b, const sbx_radix_side *dst, int64_t count,
                      const sbx_radix_pass *passes, int num_passes);

// Same sort of 64-bit keys without payload (count >= 2, num_passes >= 1) whose final pass, instead of storing the
// sorted keys, emits per key at sorted position p: value = map ? map[low32(key)] : low32(key); out[p] = value;
// bit `value` set in bits_a / bits_b (each optional); pos_of[value] = p (optional).
struct sbx_radix_emit {
  const uint32_t *map;
  uint32_t *out;
  unsigned *bits_a, *bits_b;
  unsigned *pos_of;
  uint32_t low_mask;  // low32(key) is ANDed with this first (0: all 32 bits) — keys whose fields are packed tightly
};
int sbx_radix_sort_emit(sbx_handle_t h, void *keys_a, void *keys_b, int64_t count, const sbx_radix_pass *passes,
                        int num_passes, const sbx_radix_emit *emit);

int sbx_fill_i32(sbx_handle_t h, int32_t *dst, int32_t value, int64_t count);

// degree ranks of the non-empty rows, (degree, id) ascending, both ways (sbx_degree.hip; for the RCM's Cuthill-McKee keys);
// n_top = rows of 255 entries and more; enqueued on h->stream, scratch from the running call's arena, no read-back
int sbx_degree_ranks(sbx_handle_t h, sbx_index_type it, const void *rp, int64_t n, int64_t n_nonempty, int64_t n_top,
                     unsigned max_deg, uint32_t *rank, uint32_t *order);

// ---- 64-bit index arrays (sbx_i64.hip): narrowed to the int32 kernels when every value fits
struct sbx_narrowed {
  int32_t *ptr;  // int32 copy in the arena (nullptr if the source was nullptr)
};
int sbx_narrow_i64(sbx_handle_t h, const void *src_i64, int64_t count, int32_t **out, int *overflow_flag_dev);
int sbx_widen_i32(sbx_handle_t h, const int32_t *src, void *dst_i64, int64_t count);
int sbx_i64_begin(sbx_handle_t h, int **overflow_flag_dev);          // arena_begin + nesting on
int sbx_i64_check(sbx_handle_t h, const int *overflow_flag_dev);     // synchronous overflow check
void sbx_i64_end(sbx_handle_t h);
int sbx_i64_coo_sort(sbx_handle_t h, sbx_value_type vt, int64_t n, int64_t m, int64_t nnz, void *row, void *col,
                     void *val);
int sbx_i64_mtx_parse_coordinate(sbx_handle_t h, sbx_value_type vt, const void *text_dev, int64_t bytes, int64_t n_rows,
                                 int64_t n_cols, int64_t entries, int fields, int symmetry, unsigned flags,
                                 int64_t capacity, void *row_out, void *col_out, void *val_out, int64_t *nnz_host);
int sbx_i64_edge_list_parse(sbx_handle_t h, sbx_value_type vt, const void *text_dev, int64_t bytes, int64_t entries,
                            int weighted, unsigned flags, int64_t capacity, void *row_out, void *col_out, void *val_out,
                            int64_t *dims_nnz_host);
int sbx_fill_i64(sbx_handle_t h, int64_t *dst, int64_t value, int64_t count);

// ---- SBX_I32_N64 (32-bit ids, 64-bit offsets): adapters of the entry points whose offsets are 32-bit inside (sbx_i64.hip)
int sbx_mixed_csr_rows_sorted(sbx_handle_t h, int64_t n, const void *row_ptr, const void *col, int *sorted_host);
int sbx_mixed_csr_sort_rows(sbx_handle_t h, sbx_value_type vt, int64_t n, int64_t m, int64_t nnz, const void *row_ptr,
                            void *col, void *val);
int sbx_mixed_coo_to_csc(sbx_handle_t h, sbx_value_type vt, int64_t n, int64_t m, int64_t nnz, const void *row,
                         const void *col, const void *val, void *col_ptr_out, void *row_out, void *val_out);
int sbx_mixed_csr_to_csc(sbx_handle_t h, sbx_value_type vt, int64_t n, int64_t m, int64_t nnz, const void *row_ptr,
                         const void *col, const void *val, void *col_ptr_out, void *row_out, void *val_out);
int sbx_mixed_csr_bandwidth(sbx_handle_t h, int64_t n, int64_t nnz, const void *row_ptr, const void *col,
                            int64_t *bandwidth_host);
int sbx_mixed_csr_profile(sbx_handle_t h, int64_t n, int64_t nnz, const void *row_ptr, const void *col,
                          int64_t *profile_host);
int sbx_mixed_rcm_reorder(sbx_handle_t h, int64_t n, int64_t nnz, const void *row_ptr, const void *col, void *inv_perm_out,
                          sbx_rcm_stats *stats_host);
int sbx_mixed_gray_row_keys(sbx_handle_t h, int64_t n, int64_t m, int64_t nnz, const void *row_ptr, const void *col,
                            int resolution, int nnz_threshold, void *degree_out, uint64_t *key_out, int64_t *counts_host);
int sbx_mixed_gray_reorder(sbx_handle_t h, int64_t n, int64_t m, int64_t nnz, const void *row_ptr, const void *col,
                           int resolution, int nnz_threshold, int group_size, int exact_ties, void *inv_perm_out);
int sbx_mixed_permute_csr_rows(sbx_handle_t h, sbx_value_type vt, int64_t n, int64_t m, int64_t nnz, const void *row_ptr,
                               const void *col, const void *val, const void *row_order, const void *col_order,
                               int64_t row_begin, int64_t row_end, void *row_ptr_out, void *col_out, void *val_out,
                               int64_t out_capacity, int64_t *shard_nnz_host);
int sbx_mixed_permute_csr_rows_nnz(sbx_handle_t h, int64_t n, const void *row_ptr, const void *row_order, int64_t row_begin,
                                   int64_t row_end, int64_t *nnz_host);
