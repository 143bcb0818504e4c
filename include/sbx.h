/*
 * sbx.h — C ABI of the MI355X-native reorder / convert / permute hot path.
 *
 * This is the drop-in boundary: every entry point below is what SparseBase's
 * plug-in points would bind for this path.  Each function
 * cites the reference interface (file:line under the SparseBase tree) whose
 * work it replaces.
 *
 * Conventions
 *  - Every array pointer is a DEVICE pointer on the handle's GPU unless the
 *    parameter name ends in `_host`.
 *  - No entry point allocates memory it hands back: the caller owns inputs and
 *    outputs.  Scratch comes from the handle's grow-only arena (sbx_reserve
 *    pre-sizes it, so steady-state calls never call hipMalloc).
 *  - Work is enqueued on the handle's stream, and DEVICE outputs of every
 *    entry point are complete IN STREAM ORDER: anything enqueued on the
 *    handle's stream afterwards (sbx_memcpy_d2h and sbx_sync included), or
 *    ordered behind an event recorded on it, sees the finished arrays; a
 *    consumer on another stream, or a host copy that does not go through the
 *    handle's stream, must order itself that way first.  Functions documented
 *    as "synchronous" need device->host read-backs and wait for those
 *    internally: their status and everything they return through `_host`
 *    pointers are final when they return — their device outputs keep the
 *    stream-order guarantee and no more (sbx_rcm_reorder and
 *    sbx_gray_reorder return with their last kernel enqueued, not finished).
 *    All other functions return as soon as the work is enqueued.
 *  - Return value: SBX_OK (0) or an sbx_status error code; nothing throws
 *    across this boundary.  sbx_last_error() gives a message for the last
 *    failing call on the handle.
 *  - Index type: SBX_I32 / SBX_I64 — IDType and NNZType share one width — or
 *    SBX_I32_N64, the tuple a matrix with fewer than 2^31 rows and columns and
 *    64-bit offsets uses (the reference pre-instantiates it, CMakeLists.txt:15-17:
 *    <int | unsigned int ids, long long | unsigned long long offsets>): every
 *    OFFSET array — row_ptr, col_ptr, their outputs — holds 64-bit words, every
 *    ID array (row, col, orders, inverse permutations, degrees) 32-bit words.
 *    COO <-> CSR, the degree order and the row-pointer features read and write
 *    the 64-bit offsets natively (nnz of any size); the other entry points that
 *    take offsets (permutes, constructor sort of CSR, CSC conversions, RCM, Gray)
 *    keep 32-bit offsets inside: nnz < 2^31, the offsets are narrowed to scratch
 *    and widened back (n + 1 words each way), larger nnz is SBX_ERR_UNSUPPORTED.
 *    Entry points without an offset array treat SBX_I32_N64 as SBX_I32; the
 *    sharded (multi-GPU) entry points do not take it.
 *    uint32 index arrays alias SBX_I32 (all dimensions must be < 2^31).
 *    SBX_I64: the conversions (CSC included), checks, features, degree order,
 *    permutes, both constructor sorts, RCM and the Gray keys read and write
 *    the 64-bit arrays themselves (index values of any size where the
 *    operation has room for them; nnz < 2^31 for the permutes, sorts, CSC,
 *    RCM and the Gray keys, whose ids, offsets and positions are 32-bit inside:
 *    n < 2^31 - 1, m < 2^31 there — larger dimensions return
 *    SBX_ERR_UNSUPPORTED); only the text parsers write int32 scratch outputs
 *    that are widened afterwards.
 *  - Value type: the 0/4/8-byte payload that follows each nonzero.  The
 *    arithmetic type matters only where the reference compares values
 *    (std::less<pair<col,val>> between duplicate columns, format/csr.cc:143-156).
 */
#ifndef SBX_H_
#define SBX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SBX_VERSION 102 /* 0.1.2: sbx_set_oom_hook, sbx_host_alloc / sbx_host_free, SBX_I32_N64 added (101: sbx_rcm_stats.unordered_sweeps, sbx_gray_reorder) */

typedef struct sbx_handle_s *sbx_handle_t;

typedef enum sbx_status {
  SBX_OK = 0,
  SBX_ERR_BAD_ARG = 1,     /* null pointer / negative size / unknown enum   */
  SBX_ERR_NO_DEVICE = 2,   /* no usable HIP device (fails loudly, no fallback) */
  SBX_ERR_HIP = 3,         /* a HIP runtime call failed                      */
  SBX_ERR_OOM = 4,         /* device allocation failed                       */
  SBX_ERR_UNSUPPORTED = 5, /* type tuple / shape not built                   */
  SBX_ERR_INTERNAL = 6
} sbx_status;

typedef enum sbx_index_type {
  SBX_I32 = 0,    /* 32-bit ids, 32-bit offsets */
  SBX_I64 = 1,    /* 64-bit ids, 64-bit offsets */
  SBX_I32_N64 = 2 /* 32-bit ids, 64-bit offsets (row_ptr / col_ptr arrays) */
} sbx_index_type;

typedef enum sbx_value_type {
  SBX_V_NONE = 0, /* ValueType = void, or vals == nullptr */
  SBX_V_I32 = 1,
  SBX_V_U32 = 2,
  SBX_V_F32 = 3,
  SBX_V_I64 = 4,
  SBX_V_U64 = 5,
  SBX_V_F64 = 6
} sbx_value_type;

/* flags for the conversion entry points */
#define SBX_FLAG_MOVE 0x1u        /* move-conversion: only the index array that
                                     changes shape is produced (col/val are
                                     handed over by pointer on the host side) */
#define SBX_FLAG_ROWS_SORTED 0x2u /* caller guarantees row[] is non-decreasing */

/* ------------------------------------------------------------------ *
 * Handle, memory and device utilities                                  *
 * (replaces the raw cudaMalloc/cudaMemcpy calls of                     *
 *  converter/converter_order_two_cuda.cu:11-105,                       *
 *  context/cuda_context_cuda.cu:9-21, converter/converter_cuda.cu:12-21)  *
 *                                                                      *
 * Streams: every entry point enqueues its work on the handle's stream  *
 * (sbx_set_stream; default: the null stream).  Some entry points run    *
 * independent stages on up to two private side streams of the handle;  *
 * those fork from and are joined back into the handle's stream by      *
 * events before the call returns, so callers order against the         *
 * handle's stream only.  A handle serves one host thread at a time.    */
/* ------------------------------------------------------------------ */
int sbx_version(void);
const char *sbx_status_string(int status);
int sbx_device_count(int *count_host);
int sbx_can_access_peer(int device, int peer_device, int *can_host);

int sbx_create(int device, sbx_handle_t *out);
int sbx_destroy(sbx_handle_t h);
int sbx_set_stream(sbx_handle_t h, void *hip_stream);
int sbx_get_device(sbx_handle_t h, int *device_host);
int sbx_reserve(sbx_handle_t h, size_t scratch_bytes);
int sbx_sync(sbx_handle_t h);
const char *sbx_last_error(sbx_handle_t h);
/* A caller that caches freed device blocks (the C++ host layer's pool, hip/device.h) registers a hook: when a device
 * allocation of the library's own (scratch arena, radix slots) fails for lack of memory the hook is called ONCE with
 * the size wanted; if it returns non-zero (it gave memory back to the driver) the allocation is tried again before
 * the entry point returns SBX_ERR_OOM.  The hook runs on the calling thread and must not call into the handle. */
typedef int (*sbx_oom_hook)(void *user, size_t bytes_wanted);
int sbx_set_oom_hook(sbx_handle_t h, sbx_oom_hook hook, void *user);

/* Optional per-kernel profiler: when enabled every kernel the library launches is
 * bracketed by HIP events on the handle's stream.  sbx_profile_query(index) drains
 * the events (synchronous) and returns the accumulated device time and launch
 * count of kernel group `index` (names via sbx_profile_kernel_name). */
int sbx_profile_enable(sbx_handle_t h, int on);
int sbx_profile_kernel_count(void);
const char *sbx_profile_kernel_name(int index);
int sbx_profile_query(sbx_handle_t h, int index, double *total_ms_host, int64_t *launches_host);
/* algorithmic bytes the launch sites of group `index` declared while profiling was on (0 if the
 * group declares none): what the kernel has to move, not what it did move. */
int sbx_profile_query_bytes(sbx_handle_t h, int index, int64_t *alg_bytes_host);

int sbx_malloc(sbx_handle_t h, size_t bytes, void **dev_ptr_host);
int sbx_free(sbx_handle_t h, void *dev_ptr);
/* Page-locked host memory for the blocking copies below (the reference stages through pageable arrays; a pageable
 * target is pinned and unpinned by the runtime around every copy, and on this platform the unpinning of a target that
 * is then freed stalls the process's next GPU submission by milliseconds: tools/gray_kt2.sh).  Plain host pointers,
 * readable and writable by the host like any other; freed with sbx_host_free only. */
int sbx_host_alloc(sbx_handle_t h, size_t bytes, void **host_ptr_host);
int sbx_host_free(sbx_handle_t h, void *host_ptr);
/* blocking copies (stream-ordered after prior work on the handle's stream) */
int sbx_memcpy_h2d(sbx_handle_t h, void *dst_dev, const void *src_host, size_t bytes);
int sbx_memcpy_d2h(sbx_handle_t h, void *dst_host, const void *src_dev, size_t bytes);
int sbx_memcpy_d2d(sbx_handle_t h, void *dst_dev, const void *src_dev, size_t bytes);
int sbx_memcpy_peer(sbx_handle_t h, void *dst_dev, int dst_device, const void *src_dev,
                    int src_device, size_t bytes);

/* ------------------------------------------------------------------ *
 * A1  COO constructor semantics — format/coo.cc:76-158                 *
 * ------------------------------------------------------------------ */
/* Lexicographic (row,col) non-decreasing test, format/coo.cc:96-108.  Synchronous. */
int sbx_coo_is_sorted(sbx_handle_t h, sbx_index_type it, int64_t nnz, const void *row,
                      const void *col, int *sorted_host);
/* In-place sort by (row,col), payload follows; format/coo.cc:110-157.
 * Stable (the reference's std::sort leaves the order of duplicate coordinates
 * unspecified).  val may be NULL.  Synchronous (one max-reduction read-back). */
int sbx_coo_sort(sbx_handle_t h, sbx_index_type it, sbx_value_type vt, int64_t n, int64_t m,
                 int64_t nnz, void *row, void *col, void *val);

/* ------------------------------------------------------------------ *
 * A4  CSR constructor semantics — format/csr.cc:78-159                 *
 * ------------------------------------------------------------------ */
/* 1 iff no row has col[j] < col[j-1]; format/csr.cc:102-116.  Synchronous. */
int sbx_csr_rows_sorted(sbx_handle_t h, sbx_index_type it, int64_t n, const void *row_ptr,
                        const void *col, int *sorted_host);
/* If any row is unsorted, sort EVERY row by (col,val) in place
 * (format/csr.cc:118-157); otherwise leave the arrays untouched. */
int sbx_csr_sort_rows(sbx_handle_t h, sbx_index_type it, sbx_value_type vt, int64_t n,
                      int64_t m, int64_t nnz, const void *row_ptr, void *col, void *val);

/* ------------------------------------------------------------------ *
 * A2/A2m  COO -> CSR — converter/converter_order_two.cc:163-212, :215-246
 * row_ptr_out = exclusive scan of the row histogram; col/val copied verbatim.
 * With SBX_FLAG_MOVE col_out/val_out are ignored (may be NULL).         */
/* ------------------------------------------------------------------ */
int sbx_coo_to_csr(sbx_handle_t h, sbx_index_type it, sbx_value_type vt, int64_t n, int64_t m,
                   int64_t nnz, const void *row, const void *col, const void *val,
                   void *row_ptr_out, void *col_out, void *val_out, unsigned flags);

/* ------------------------------------------------------------------ *
 * A3  CSR -> COO — converter/converter_order_two.cc:72-118, :131-160   *
 * ------------------------------------------------------------------ */
int sbx_csr_to_coo(sbx_handle_t h, sbx_index_type it, sbx_value_type vt, int64_t n, int64_t m,
                   int64_t nnz, const void *row_ptr, const void *col, const void *val,
                   void *row_out, void *col_out, void *val_out, unsigned flags);

/* ------------------------------------------------------------------ *
 * A14  COO -> CSC — converter/converter_order_two.cc:21-70: stable counting sort of the
 * nonzeros by column (within a column they keep their input order), followed by what the
 * CSC constructor does (format/csc.cc:99-157: if any column's rows are out of order, every
 * column's (row, value) pairs are sorted).  col_ptr_out has m + 1 entries; the reference
 * sizes it by the ROW count (:32-33) and is only safe for n == m, where the two agree.
 * A15  CSR -> CSC — :120-128: CSR -> COO -> CSC (the transpose when rows are sorted). */
/* ------------------------------------------------------------------ */
int sbx_coo_to_csc(sbx_handle_t h, sbx_index_type it, sbx_value_type vt, int64_t n, int64_t m,
                   int64_t nnz, const void *row, const void *col, const void *val,
                   void *col_ptr_out, void *row_out, void *val_out);
int sbx_csr_to_csc(sbx_handle_t h, sbx_index_type it, sbx_value_type vt, int64_t n, int64_t m,
                   int64_t nnz, const void *row_ptr, const void *col, const void *val,
                   void *col_ptr_out, void *row_out, void *val_out);

/* ------------------------------------------------------------------ *
 * Matrix Market ingest — io/mtx_reader.cc:307-495 MTXReader::ReadCoordinateIntoCOO.
 * text_dev: the bytes of the file AFTER the size line ("M N L"), resident on the device; the banner
 * and the size line are parsed by the caller.  `entries` = L, `fields` = tokens per entry (2 for
 * pattern files, 3 otherwise; with fields == 3 and val_out == NULL the values are skipped).  Tokens are
 * whitespace separated exactly like the reference's `fin >> m >> n >> w`; values of type vt are parsed as
 * decimal integers (I32/U32/I64/U64) or converted exactly to float/double (the result of strtof/strtod).
 * symmetry: 0 general, 1 symmetric, 2 skew-symmetric: unless SBX_MTX_UPPER_TRIANGLE every entry is
 * followed by its mirror (symmetric: off-diagonal entries only, :441-447; skew: negated value, :452-454);
 * with SBX_MTX_UPPER_TRIANGLE the entry is stored as (min, max) instead (:368-384).  SBX_MTX_ZERO_INDEX
 * subtracts 1 from both indices (:334-337).  Outputs need capacity >= entries (2 * entries when mirrors
 * are produced); *nnz_host receives the number of nonzeros written.  The COO constructor's sort is the
 * caller's next step (sbx_coo_sort).  Malformed tokens -> SBX_ERR_BAD_ARG; values with more than 38
 * significant digits and a non-zero tail -> SBX_ERR_UNSUPPORTED.  Synchronous.                 */
/* ------------------------------------------------------------------ */
#define SBX_MTX_ZERO_INDEX 1u
#define SBX_MTX_UPPER_TRIANGLE 2u
int sbx_mtx_parse_coordinate(sbx_handle_t h, sbx_index_type it, sbx_value_type vt, const void *text_dev,
                             int64_t bytes, int64_t n_rows, int64_t n_cols, int64_t entries, int fields,
                             int symmetry, unsigned flags, int64_t capacity, void *row_out, void *col_out,
                             void *val_out, int64_t *nnz_host);

/* Number of whitespace-separated tokens in a device text buffer (to size the parsers' outputs). */
int sbx_text_count_tokens(sbx_handle_t h, const void *text_dev, int64_t bytes, int64_t *tokens_host);

/* ------------------------------------------------------------------ *
 * Edge-list ingest — io/edge_list_reader.cc:19-158 EdgeListReader::ReadCOO.  text_dev: the whole file on
 * the device; entries = tokens / (weighted ? 3 : 2) (sbx_text_count_tokens).  Vertices are 0-based (:31).
 * Every edge "u v [w]" is kept unless SBX_EDGES_REMOVE_SELF and u == v (:35); with SBX_EDGES_UNDIRECTED
 * its reverse follows it (:38-39); n = max(u) + 1, m = max(v) + 1 over the kept edges, made equal by
 * SBX_EDGES_SQUARE or SBX_EDGES_UNDIRECTED (:41-49); the edges are sorted by (row, col) (:51-56; stable
 * here, the reference's std::sort leaves the order of equal coordinates unspecified) and, with
 * SBX_EDGES_REMOVE_DUPLICATES, reduced to the first of every run of equal coordinates (:58-66).
 * Outputs need capacity >= entries (2 * entries with SBX_EDGES_UNDIRECTED); dims_nnz_host[3] receives
 * n, m, nnz.  Synchronous.                                              */
/* ------------------------------------------------------------------ */
#define SBX_EDGES_REMOVE_DUPLICATES 1u
#define SBX_EDGES_REMOVE_SELF 2u
#define SBX_EDGES_UNDIRECTED 4u
#define SBX_EDGES_SQUARE 8u
int sbx_edge_list_parse(sbx_handle_t h, sbx_index_type it, sbx_value_type vt, const void *text_dev,
                        int64_t bytes, int64_t entries, int weighted, unsigned flags, int64_t capacity,
                        void *row_out, void *col_out, void *val_out, int64_t *dims_nnz_host);

/* ------------------------------------------------------------------ *
 * Reorder-quality features: single-pass reductions over a device CSR.
 *   sbx_csr_degrees              feature/degrees.cc:93-105      degrees_out[i] = row_ptr[i+1] - row_ptr[i]
 *   sbx_csr_degree_distribution  feature/degree_distribution.cc:152-167
 *                                dist_out[i] = degree / (FeatureType)nnz, FeatureType float (4) or double (8)
 *   sbx_csr_bandwidth            feature/bandwidth.cc:93-112    max over nonzeros of |i - j| + 1 (0 if none)
 *   sbx_csr_profile              feature/profile.cc:91-105      sum over rows of i - min(i, smallest column);
 *                                returned exactly in 64 bits (the reference accumulates in IDType)
 * The two scalar results are written to host memory; the calls are synchronous.            */
/* ------------------------------------------------------------------ */
int sbx_csr_degrees(sbx_handle_t h, sbx_index_type it, int64_t n, const void *row_ptr, void *degrees_out);
int sbx_csr_degree_distribution(sbx_handle_t h, sbx_index_type it, int64_t n, int64_t nnz,
                                const void *row_ptr, int feature_bytes, void *dist_out);
int sbx_csr_bandwidth(sbx_handle_t h, sbx_index_type it, int64_t n, int64_t nnz, const void *row_ptr,
                      const void *col, int64_t *bandwidth_host);
int sbx_csr_profile(sbx_handle_t h, sbx_index_type it, int64_t n, int64_t nnz, const void *row_ptr,
                    const void *col, int64_t *profile_host);

/* ------------------------------------------------------------------ *
 * A6  DegreeReorder::CalculateReorderCSR — reorder/degree_reorder.cc:22-62
 * inv_perm_out[old_row] = new_row; ascending: (deg asc, id desc),        *
 * descending: the exact reverse.                                         */
/* ------------------------------------------------------------------ */
int sbx_degree_reorder(sbx_handle_t h, sbx_index_type it, int64_t n, const void *row_ptr,
                       int ascending, void *inv_perm_out);

/* ------------------------------------------------------------------ *
 * A7  RCMReorder::GetReorderCSR — reorder/rcm_reorder.cc:83-166 (+ :22-81)
 * Parity is defined for structurally symmetric patterns with column-sorted
 * rows (what the CSR constructor guarantees).  Synchronous (status and
 * statistics final on return); inv_perm_out is complete in stream order (see
 * Conventions): when one component holds all the work the host drives, the
 * kernel that writes its positions is enqueued behind the call's last
 * read-back and may still run when the call returns.  The host layer's
 * RCMReorder::GetReorder (reorder/reorderer.h:53-117: a host array, synchronous)
 * downloads through sbx_memcpy_d2h on the handle's stream and is synchronous.
 *
 * Grid barriers.  Three of the call's kernels (the small-level runs of the
 * pseudo-peripheral sweeps and the two tie-break walks, sbx_rcm.hip: gb_wait)
 * are persistent grids of 64 workgroups that meet at counter barriers in
 * global memory instead of returning to the host after every BFS level.  A
 * counter barrier only completes while every workgroup of the grid is
 * resident, which a plain launch does not promise when ANOTHER process or
 * stream keeps the GPU's CUs busy.  Every wait is therefore bounded (2^16
 * polls, a few milliseconds; SBX_DEBUG_GB_SPINS overrides the bound for
 * tests): a workgroup that runs out of polls raises a flag and leaves, the
 * others follow, nothing such a kernel computed is committed, and the host —
 * which finds the flag in its next read-back — redoes that sweep with the
 * one-launch-per-level kernels.  What the caller sees:
 *   - the RESULT is the same bit-exact ordering either way, and the call still
 *     returns SBX_OK: a given-up barrier is not an error;
 *   - that call takes a few milliseconds longer (the bound), and the next 16
 *     sbx_rcm_reorder calls on the same handle stay away from the persistent
 *     kernels (one launch + one read-back per level: about 2x the time on a
 *     power-law graph), after which they are tried again;
 *   - nothing is reported through stats_host; sbx_last_error() is untouched.
 * Sharing one GPU between processes that all run RCM is therefore safe but
 * slow (plain time slicing); one handle per process and GPU is the intended use. */
/* ------------------------------------------------------------------ */
typedef struct sbx_rcm_stats {
  int64_t components;       /* connected components incl. isolated vertices */
  int64_t isolated;         /* vertices with an empty row                   */
  int64_t small_components; /* components ordered by the batched kernel     */
  int64_t large_components; /* components ordered by level-synchronous BFS  */
  int64_t bfs_sweeps;       /* full BFS sweeps over the largest component (speculative ones included) */
  int64_t bfs_levels;       /* levels summed over those sweeps              */
  int64_t edges_scanned;    /* adjacency entries visited (all sweeps)       */
  int64_t edges_scanned_bottom_up; /* ... of which by the bottom-up kernel   */
  int64_t largest_component;
  int64_t reference_sweeps; /* sweeps the reference's serial algorithm runs over the largest component:
                               pseudo-peripheral iterations (rcm_reorder.cc:34) + 1 — the B of the roofline figure */
  int64_t unordered_sweeps; /* sweeps of the pseudo-peripheral search (over all components the host orders) that kept
                               the level SETS only; 0 when every sweep kept the order inside its levels */
} sbx_rcm_stats;
int sbx_rcm_reorder(sbx_handle_t h, sbx_index_type it, int64_t n, int64_t nnz,
                    const void *row_ptr, const void *col, void *inv_perm_out,
                    sbx_rcm_stats *stats_host /* may be NULL */);

/* ------------------------------------------------------------------ *
 * A8  GrayReorder::GrayReorderingCSR — reorder/gray_reorder.cc:106-424 *
 * Device stage: per-row degree, band count and Gray-decoded bitmap key. *
 * key_out[i] (uint64) = decoded bitmap of row i computed with the row's  *
 * class threshold (sparse rows: 0; dense rows: deg/resolution).          *
 * counts_host[4] = {nnz_sparse, diag_sparse, nnz_dense, diag_dense}.     *
 * Synchronous.  The ordering stage lives above the ABI.
 * Three families of kernels serve the call, chosen by what the matrix turns
 * out to be: banded / mesh matrices one sweep of 4 lanes per
 * row; power-law matrices (the first kernel notices and stops within ~30 us)
 * per-entry kernels over rows cut into 1024-entry units; resolutions below 16
 * a nonzero-parallel tile kernel.  The results are identical.              */
/* ------------------------------------------------------------------ */
int sbx_gray_row_keys(sbx_handle_t h, sbx_index_type it, int64_t n, int64_t m, int64_t nnz,
                      const void *row_ptr, const void *col, int resolution, int nnz_threshold,
                      void *degree_out, uint64_t *key_out, int64_t *counts_host);

/* GrayReorder with its ordering stage on the device — the opt-in mode SURVEY.md section 8(b) sketches (`exact_ties`).  *
 * exact_ties = 0: every sort of the reference (gray_reorder.cc:199-203 by degree, :293-301 / :354-360 the sections by
 * decoded key, ascending and descending in turn, :404 the dense rows) runs as a STABLE device sort: the ordering is
 * (class, section, +-key, degree, row id), three radix sorts of (key, row) pairs behind sbx_gray_row_keys.  It equals
 * the reference's wherever the reference's comparators decide the order; tied rows come in stable order instead of
 * libstdc++'s introsort order.  exact_ties != 0 is refused with SBX_ERR_UNSUPPORTED: the exact mode — the default of
 * reorder::GrayReorder — issues the reference's own std::sort calls in the host layer over the device-computed keys
 * (DESIGN.md section 5).  inv_perm_out[n] (index type `it`) on the device.  m must be a multiple of the resolution
 * (see sbx_gray_row_keys).  Synchronous (status); inv_perm_out complete in stream order. */
int sbx_gray_reorder(sbx_handle_t h, sbx_index_type it, int64_t n, int64_t m, int64_t nnz,
                     const void *row_ptr, const void *col, int resolution, int nnz_threshold,
                     int group_size, int exact_ties, void *inv_perm_out);

/* ------------------------------------------------------------------ *
 * A13 ReorderBase::InversePermutation — bases/reorder_base.h:663-672   *
 * ------------------------------------------------------------------ */
int sbx_inverse_permutation(sbx_handle_t h, sbx_index_type it, int64_t n, const void *perm,
                            void *inv_out);

/* ------------------------------------------------------------------ *
 * A5 (+A4) PermuteOrderTwo::PermuteOrderTwoCSR — permute/permute_order_two.cc:23-79
 * row_order / col_order are inverse permutations (order[old] = new) or NULL
 * (identity).  Output rows are sorted by (col,val) exactly when the CSR
 * constructor would sort them (format/csr.cc:99-157).
 * sbx_permute_csr_rows produces only new rows [row_begin,row_end): row_ptr_out
 * receives row_end-row_begin+1 entries rebased to 0, col_out/val_out the
 * shard's nonzeros; *shard_nnz_host gets the shard's nnz.  The row-range
 * split is the multi-GPU decomposition.  Synchronous only
 * when shard_nnz_host != NULL.                                           */
/* ------------------------------------------------------------------ */
int sbx_permute_csr(sbx_handle_t h, sbx_index_type it, sbx_value_type vt, int64_t n, int64_t m,
                    int64_t nnz, const void *row_ptr, const void *col, const void *val,
                    const void *row_order, const void *col_order, void *row_ptr_out,
                    void *col_out, void *val_out);
int sbx_permute_csr_rows(sbx_handle_t h, sbx_index_type it, sbx_value_type vt, int64_t n,
                         int64_t m, int64_t nnz, const void *row_ptr, const void *col,
                         const void *val, const void *row_order, const void *col_order,
                         int64_t row_begin, int64_t row_end, void *row_ptr_out, void *col_out,
                         void *val_out, int64_t out_capacity, int64_t *shard_nnz_host);

/* PermuteOrderOne::PermuteArray — permute/permute_order_one.cc:18-37:
 * out[order[i]] = vals[i]. */
int sbx_permute_array(sbx_handle_t h, sbx_index_type it, sbx_value_type vt, int64_t n,
                      const void *order, const void *vals, void *out);

/* ------------------------------------------------------------------ *
 * §8e  The steps that shard over the GPUs of one node (one process per GPU): permutation
 * apply by new-row range and COO -> CSR by row range.  Inputs (CSR / row-sorted COO, order
 * vectors) are replicated; a rank computes its own slab with no communication, then two
 * all-gathers (8 bytes per rank of nnz totals; the rebased row_ptr segments in padded equal
 * chunks, Z * n bytes in total) give every rank the complete row_ptr.  col / val stay
 * row-sharded.  The reference has no multi-GPU operator; its device-to-device edge is
 * converter/converter_order_two_cuda.cu:41-76 (predicate converter/converter_cuda.cu:12-21).
 *
 * Communicator: RCCL (sbx_comm_create_rccl; rank 0 makes the id with sbx_comm_unique_id and
 * hands it to the other ranks by any means) or a caller-supplied all-gather hook
 * (sbx_comm_create) — the hook gets device pointers and must leave `recv` (world * bytes, in
 * rank order) complete for work enqueued afterwards on `stream` (a hipStream_t).            */
/* ------------------------------------------------------------------ */
#define SBX_COMM_ID_BYTES 128
typedef struct sbx_comm_s *sbx_comm_t;
typedef int (*sbx_allgather_fn)(void *user, const void *send_dev, void *recv_dev, size_t bytes,
                                void *stream);
int sbx_comm_create(int rank, int world, sbx_allgather_fn allgather, void *user, sbx_comm_t *out);
int sbx_comm_unique_id(void *id_out /* SBX_COMM_ID_BYTES */);
int sbx_comm_create_rccl(int device, int rank, int world, const void *unique_id, sbx_comm_t *out);
int sbx_comm_rank(sbx_comm_t comm, int *rank, int *world);
int sbx_comm_destroy(sbx_comm_t comm);

/* Entries of the new rows [row_begin,row_end): what a rank's col_out / val_out must hold. */
int sbx_permute_csr_rows_nnz(sbx_handle_t h, sbx_index_type it, int64_t n, const void *row_ptr,
                             const void *row_order, int64_t row_begin, int64_t row_end,
                             int64_t *nnz_host);

/* A5 sharded.  row_splits: world + 1 new-row boundaries (NULL: equal ranges).  row_ptr_out
 * (n + 1 entries) is complete on every rank; col_out / val_out receive THIS rank's rows
 * (out_capacity entries); shard_offsets_host (world + 1, may be NULL) are the positions of the
 * shards in the global entry space.  Synchronous; the collectives run on the handle's stream and
 * the host waits once, behind them (the shards' offsets are computed on the device from the
 * gathered totals).  A rank whose own part fails (a slab beyond out_capacity, ...) still takes
 * part in both collectives with a status word, so every rank of the call returns an error
 * instead of waiting for ever (SBX_ERR_INTERNAL "rank r reported a failure" on the others); a
 * launch or a collective that reports an error between the two all-gathers is remembered and the
 * second one is still entered.  Not covered: a rank that cannot allocate the call's few KB +
 * 2 x (largest range + 1) x world index words of scratch before the first collective, or whose
 * device has stopped executing — its peers then wait inside the collective (the RCCL watchdog's
 * business, as for any collective). */
int sbx_permute_csr_sharded(sbx_handle_t h, sbx_comm_t comm, sbx_index_type it, sbx_value_type vt,
                            int64_t n, int64_t m, int64_t nnz, const void *row_ptr,
                            const void *col, const void *val, const void *row_order,
                            const void *col_order, const int64_t *row_splits, void *row_ptr_out,
                            void *col_out, void *val_out, int64_t out_capacity,
                            int64_t *shard_offsets_host);

/* A2 sharded: row-sorted COO (replicated) -> this rank's rows of the CSR + the whole row_ptr. */
int sbx_coo_to_csr_sharded(sbx_handle_t h, sbx_comm_t comm, sbx_index_type it, sbx_value_type vt,
                           int64_t n, int64_t m, int64_t nnz, const void *row, const void *col,
                           const void *val, const int64_t *row_splits, void *row_ptr_out,
                           void *col_out, void *val_out, int64_t out_capacity,
                           int64_t *shard_offsets_host);

/* A3 sharded (converter/converter_order_two.cc:72-160 by row range; the reference's
 * device-to-device edge: converter/converter_order_two_cuda.cu:41-76): CSR (replicated) -> this
 * rank's slab of the COO, rows [row_splits[rank], row_splits[rank + 1]) with their global row
 * ids.  No collective: shard_offsets_host (world + 1, may be NULL) = row_ptr at the split
 * points.  Synchronous. */
int sbx_csr_to_coo_sharded(sbx_handle_t h, sbx_comm_t comm, sbx_index_type it, sbx_value_type vt,
                           int64_t n, int64_t m, int64_t nnz, const void *row_ptr, const void *col,
                           const void *val, const int64_t *row_splits, void *row_out, void *col_out,
                           void *val_out, int64_t out_capacity, int64_t *shard_offsets_host);

/* Row ranges of (nearly) equal entry counts for the sharded permute (balance by
 * nnz, not rows): splits_host[0 .. world] over the NEW rows of sbx_permute_csr (row_order may be
 * NULL: the rows as they are).  Every rank computes the same splits from the replicated input.
 * world <= 64.  Synchronous. */
int sbx_balanced_row_splits(sbx_handle_t h, sbx_index_type it, int64_t n, const void *row_ptr,
                            const void *row_order, int world, int64_t *splits_host);

#ifdef __cplusplus
}
#endif
#endif /* SBX_H_ */
