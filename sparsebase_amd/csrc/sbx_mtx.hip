// sbx_mtx.hip — Matrix Market coordinate section -> COO on the device.
//
//   io/mtx_reader.cc:307-495  MTXReader::ReadCoordinateIntoCOO   sbx_mtx_parse_coordinate
//
// The reference reads the entries with `fin >> m >> n [>> w]` — whitespace-separated tokens,
// not lines.  Here: token starts are found and compacted in file order (count, scan,
// write), one thread per entry parses its tokens (indices: decimal integers; values:
// integers, or decimal floating point converted exactly by sbx_dec2bin.h, i.e. the same
// result as the stream extraction), then the symmetric / skew-symmetric expansion of
// :403-470 (entry, then its mirror unless it lies on the diagonal of a symmetric matrix)
// is placed by an exclusive scan of the mirror flags.  The COO constructor's sort is the
// caller's next step (sbx_coo_sort).  Tokens the stream extraction would choke on (hex,
// inf/nan, garbage, a '.' in an integer field) are reported as an error instead of
// reproducing the stream's fail state; values with more than 38 significant digits and a
// non-zero tail are refused (SBX_ERR_UNSUPPORTED).
#include "sbx_dec2bin.h"
#include "sbx_device.h"
#include "sbx_internal.h"

namespace {

constexpr int MX_THREADS = 256;
constexpr int MX_BPT = 16;                       // text bytes per thread
constexpr int MX_TILE = MX_THREADS * MX_BPT;     // text bytes per workgroup

enum : unsigned { MX_BAD_INDEX = 1u, MX_BAD_VALUE = 2u, MX_TOO_MANY_DIGITS = 4u, MX_INDEX_RANGE = 8u };

__device__ __forceinline__ bool mx_space(char c) {
  return c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '\v' || c == '\f';
}

// token starts of this thread's MX_BPT bytes as a bit mask
__device__ __forceinline__ unsigned mx_starts(const char *__restrict__ text, int64_t bytes, int64_t p0) {
  static_assert(MX_BPT == 16, "one 16-byte load per thread");
  unsigned mask = 0;
  const char before = p0 == 0 ? ' ' : text[p0 - 1];
  if (p0 + MX_BPT <= bytes && (((uintptr_t)text + (uintptr_t)p0) & 15) == 0) {
    // the thread's 16 bytes in one load (byte by byte, under the end-of-text test, the compiler made them 17 loads
    // that wait for one another: tools/isa_waits.py)
    const uint4 v = *(const uint4 *)(text + p0);
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    bool prev_space = mx_space(before);
#pragma unroll
    for (int k = 0; k < MX_BPT; k++) {
      const bool sp = mx_space((char)((w[k >> 2] >> (8 * (k & 3))) & 0xFFu));
      if (!sp && prev_space) mask |= 1u << k;
      prev_space = sp;
    }
    return mask;
  }
  bool prev_space = mx_space(before);
#pragma unroll
  for (int k = 0; k < MX_BPT; k++) {
    if (p0 + k >= bytes) break;
    const bool sp = mx_space(text[p0 + k]);
    if (!sp && prev_space) mask |= 1u << k;
    prev_space = sp;
  }
  return mask;
}

__global__ __launch_bounds__(MX_THREADS) void k_mtx_count(const char *__restrict__ text, int64_t bytes,
                                                          unsigned *__restrict__ tile_tokens) {
  __shared__ unsigned s_red[MX_THREADS / 64 + 1];
  const int64_t p0 = (int64_t)blockIdx.x * MX_TILE + (int64_t)threadIdx.x * MX_BPT;
  const unsigned c = p0 < bytes ? (unsigned)__popc(mx_starts(text, bytes, p0)) : 0u;
  const unsigned tot = sbx_block_sum<unsigned, MX_THREADS>(c, s_red);
  if (threadIdx.x == 0) tile_tokens[blockIdx.x] = tot;
}

__global__ __launch_bounds__(MX_THREADS) void k_mtx_offsets(const char *__restrict__ text, int64_t bytes,
                                                            const unsigned *__restrict__ tile_base,
                                                            int64_t max_tokens, unsigned *__restrict__ tok_off) {
  __shared__ unsigned s_scan[MX_THREADS / 64 + 1];
  const int64_t p0 = (int64_t)blockIdx.x * MX_TILE + (int64_t)threadIdx.x * MX_BPT;
  unsigned mask = p0 < bytes ? mx_starts(text, bytes, p0) : 0u;
  unsigned all;
  unsigned t = tile_base[blockIdx.x] + sbx_block_exclusive_sum<unsigned, MX_THREADS>((unsigned)__popc(mask), s_scan, &all);
  while (mask) {
    const int k = __ffs(mask) - 1;
    mask &= mask - 1;
    if ((int64_t)t < max_tokens) tok_off[t] = (unsigned)(p0 + k);
    t++;
  }
}

__device__ __forceinline__ int64_t mx_token_len(const char *__restrict__ text, int64_t bytes, int64_t start) {
  int64_t e = start;
  while (e < bytes && !mx_space(text[e])) e++;
  return e - start;
}

// one thread per entry: indices, value, mirror flag
template <int VKIND /*0 none, 1 integer, 2 float, 3 double*/, int VB>
__global__ __launch_bounds__(MX_THREADS) void k_mtx_parse(const char *__restrict__ text, int64_t bytes,
                                                          const unsigned *__restrict__ tok_off, int64_t entries,
                                                          int fields, int symmetry, int zero_index, int upper,
                                                          int64_t n_rows, int64_t n_cols, int value_signed,
                                                          const uint64_t *__restrict__ pow5,
                                                          int32_t *__restrict__ row, int32_t *__restrict__ col,
                                                          char *__restrict__ val, unsigned *__restrict__ mirror,
                                                          unsigned *__restrict__ status) {
  const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= entries) return;
  unsigned bad = 0;
  long long idx[2] = {0, 0};
#pragma unroll
  for (int f = 0; f < 2; f++) {
    const int64_t s = tok_off[l * fields + f];
    if (sbx_parse_integer(text + s, mx_token_len(text, bytes, s), &idx[f])) bad |= MX_BAD_INDEX;
  }
  long long m = idx[0], n = idx[1];
  if (zero_index) { m--; n--; }
  if (m < 0 || n < 0 || m > 0x7FFFFFFFll || n > 0x7FFFFFFFll) bad |= MX_INDEX_RANGE;
  (void)n_rows; (void)n_cols;  // the reference does not check indices against the size line either
  uint64_t vbits = 0;
  if (VKIND != 0) {
    const int64_t s = tok_off[l * fields + 2];
    const int64_t len = mx_token_len(text, bytes, s);
    if (VKIND == 1) {
      long long v = 0;
      if (sbx_parse_integer(text + s, len, &v)) bad |= MX_BAD_VALUE;
      if (VB == 4) {
        if (value_signed ? (v < -2147483648ll || v > 2147483647ll) : (v < 0 || v > 4294967295ll)) bad |= MX_BAD_VALUE;
      } else if (!value_signed && v < 0) {
        bad |= MX_BAD_VALUE;
      }
      vbits = (uint64_t)v;
    } else {
      const sbx_decimal d = sbx_parse_decimal(text + s, len);
      if (d.status == 1) bad |= MX_BAD_VALUE;
      if (d.status == 2) bad |= MX_TOO_MANY_DIGITS;
      if (VKIND == 2) vbits = (uint64_t)(sbx_decimal_to_float_bits(d, pow5) | ((uint32_t)d.neg << 31));
      else vbits = sbx_decimal_to_double_bits(d, pow5) | ((uint64_t)d.neg << 63);
    }
  }
  if (upper && symmetry != 0) {  // :368-384: keep the entry in the upper triangle, no mirror
    const long long a = m < n ? m : n, b = m < n ? n : m;
    m = a;
    n = b;
  }
  row[l] = (int32_t)m;
  col[l] = (int32_t)n;
  if (VKIND != 0) {
    if (VB == 4) ((uint32_t *)val)[l] = (uint32_t)vbits;
    else ((uint64_t *)val)[l] = vbits;
  }
  if (mirror) mirror[l] = (symmetry == 2 || m != n) ? 1u : 0u;  // skew: always; symmetric: off-diagonal (:441-447)
  if (bad) atomicOr(status, bad);
}

// entry l goes to l + (mirrors before it); its mirror (if any) right behind it
template <int VKIND, int VB>
__global__ __launch_bounds__(MX_THREADS) void k_mtx_expand(const int32_t *__restrict__ row, const int32_t *__restrict__ col,
                                                           const char *__restrict__ val,
                                                           const unsigned *__restrict__ mirror_before,
                                                           const unsigned *__restrict__ mirror, int64_t entries,
                                                           int skew, int value_signed_or_float,
                                                           int32_t *__restrict__ row_out, int32_t *__restrict__ col_out,
                                                           char *__restrict__ val_out) {
  const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= entries) return;
  const int64_t o = l + (int64_t)mirror_before[l];
  const int32_t m = row[l], n = col[l];
  row_out[o] = m;
  col_out[o] = n;
  uint64_t v = 0;
  if (VKIND != 0) {
    v = VB == 4 ? (uint64_t)((const uint32_t *)val)[l] : ((const uint64_t *)val)[l];
    if (VB == 4) ((uint32_t *)val_out)[o] = (uint32_t)v;
    else ((uint64_t *)val_out)[o] = v;
  }
  if (mirror[l]) {
    row_out[o + 1] = n;
    col_out[o + 1] = m;
    if (VKIND != 0) {
      uint64_t mv = v;
      if (skew) {  // vals[mirror] = -vals[entry] (:452-454)
        if (VKIND == 1) mv = VB == 4 ? (uint64_t)(uint32_t)(0u - (uint32_t)v) : (uint64_t)(0ull - v);
        else mv = v ^ (VB == 4 ? 0x80000000ull : 0x8000000000000000ull);
      }
      if (VB == 4) ((uint32_t *)val_out)[o + 1] = (uint32_t)mv;
      else ((uint64_t *)val_out)[o + 1] = mv;
    }
  }
  (void)value_signed_or_float;
}

// ---- edge lists (io/edge_list_reader.cc:19-158): self-edge filter, reverse edges, dimensions, duplicates
struct EdgeDims {
  unsigned n, m;  // max(u) + 1, max(v) + 1 over the edges that are kept (:41-42)
};

__global__ __launch_bounds__(MX_THREADS) void k_edge_flags(const int32_t *__restrict__ u, const int32_t *__restrict__ v,
                                                           int64_t entries, int remove_self, int undirected,
                                                           unsigned *__restrict__ outputs, EdgeDims *__restrict__ dims) {
  __shared__ unsigned s_n[MX_THREADS / 64], s_m[MX_THREADS / 64];
  const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned mn = 0, mm = 0;
  if (l < entries) {
    const bool keep = !(remove_self && u[l] == v[l]);
    outputs[l] = keep ? (undirected ? 2u : 1u) : 0u;
    if (keep) {
      mn = (unsigned)u[l] + 1u;
      mm = (unsigned)v[l] + 1u;
    }
  }
  mn = sbx_wave_max(mn);
  mm = sbx_wave_max(mm);
  if (sbx_lane() == 0) {
    s_n[threadIdx.x >> 6] = mn;
    s_m[threadIdx.x >> 6] = mm;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < MX_THREADS / 64; w++) {
      mn = s_n[w] > mn ? s_n[w] : mn;
      mm = s_m[w] > mm ? s_m[w] : mm;
    }
    if (mn) atomicMax(&dims->n, mn);  // one atomic pair per workgroup
    if (mm) atomicMax(&dims->m, mm);
  }
}

template <int VB>
__global__ __launch_bounds__(MX_THREADS) void k_edge_emit(const int32_t *__restrict__ u, const int32_t *__restrict__ v,
                                                          const char *__restrict__ w, const unsigned *__restrict__ before,
                                                          const unsigned *__restrict__ outputs, int64_t entries,
                                                          int32_t *__restrict__ row, int32_t *__restrict__ col,
                                                          char *__restrict__ val) {
  const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= entries || outputs[l] == 0) return;
  const int64_t o = before[l];
  row[o] = u[l];
  col[o] = v[l];
  if (VB == 4) ((uint32_t *)val)[o] = ((const uint32_t *)w)[l];
  if (VB == 8) ((uint64_t *)val)[o] = ((const uint64_t *)w)[l];
  if (outputs[l] == 2) {  // the reverse edge right behind (:38-39)
    row[o + 1] = v[l];
    col[o + 1] = u[l];
    if (VB == 4) ((uint32_t *)val)[o + 1] = ((const uint32_t *)w)[l];
    if (VB == 8) ((uint64_t *)val)[o + 1] = ((const uint64_t *)w)[l];
  }
}

// first entry of every run of equal (row, col) in the sorted list (std::unique, :58-66)
__global__ __launch_bounds__(MX_THREADS) void k_edge_first(const int32_t *__restrict__ row, const int32_t *__restrict__ col,
                                                           int64_t count, unsigned *__restrict__ first) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) first[i] = (i == 0 || row[i] != row[i - 1] || col[i] != col[i - 1]) ? 1u : 0u;
}

template <int VB>
__global__ __launch_bounds__(MX_THREADS) void k_edge_compact(const int32_t *__restrict__ row, const int32_t *__restrict__ col,
                                                             const char *__restrict__ val, const unsigned *__restrict__ first,
                                                             const unsigned *__restrict__ before, int64_t count,
                                                             int32_t *__restrict__ row_out, int32_t *__restrict__ col_out,
                                                             char *__restrict__ val_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count || !first[i]) return;
  const int64_t o = before[i];
  row_out[o] = row[i];
  col_out[o] = col[i];
  if (VB == 4) ((uint32_t *)val_out)[o] = ((const uint32_t *)val)[i];
  if (VB == 8) ((uint64_t *)val_out)[o] = ((const uint64_t *)val)[i];
}

struct NestGuard {
  sbx_handle_t h;
  explicit NestGuard(sbx_handle_t h) : h(h) { h->nest++; }
  ~NestGuard() { h->nest--; }
};

}  // namespace

#define SBX_REQUIRE(h, cond, msg)                                       \
  do {                                                                  \
    if (!(cond)) SBX_FAIL(h, SBX_ERR_BAD_ARG, "%s: %s", __func__, msg); \
  } while (0)

// tables for the decimal conversion, built on first use (62 KB per handle)
static int mx_pow5(sbx_handle_t h, const uint64_t **out) {
  if (!h->pow5) {
    const size_t words = (size_t)SBX_TABLE_WORDS;  // 5^k limbs + the Eisel-Lemire significands
    uint64_t *host = new uint64_t[words];
    sbx_pow5_table_fill(host);
    void *dev = nullptr;
    hipError_t e = hipMalloc(&dev, words * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMemcpy(dev, host, words * sizeof(uint64_t), hipMemcpyHostToDevice);
    delete[] host;
    if (e != hipSuccess) {
      if (dev) (void)hipFree(dev);
      SBX_FAIL(h, SBX_ERR_HIP, "pow5 table: %s", hipGetErrorString(e));
    }
    h->pow5 = dev;
  }
  *out = (const uint64_t *)h->pow5;
  return SBX_OK;
}

extern "C" int sbx_mtx_parse_coordinate(sbx_handle_t h, sbx_index_type it, sbx_value_type vt, const void *text_dev,
                                        int64_t bytes, int64_t n_rows, int64_t n_cols, int64_t entries, int fields,
                                        int symmetry, unsigned flags, int64_t capacity, void *row_out, void *col_out,
                                        void *val_out, int64_t *nnz_host) {
  if (!h) return SBX_ERR_BAD_ARG;
  if (it == SBX_I32_N64) it = SBX_I32;  // (no offset array)
  SBX_REQUIRE(h, nnz_host && bytes >= 0 && entries >= 0 && (entries == 0 || (text_dev && row_out && col_out)),
              "bad argument");
  SBX_REQUIRE(h, fields == 2 || fields == 3, "an entry has 2 (pattern) or 3 tokens");
  SBX_REQUIRE(h, symmetry >= 0 && symmetry <= 2, "symmetry: 0 general, 1 symmetric, 2 skew-symmetric");
  SBX_REQUIRE(h, bytes < ((int64_t)1 << 32), "text sections of 4 GiB and more are not supported (32-bit token offsets)");
  SBX_REQUIRE(h, entries * fields < ((int64_t)1 << 32) && 2 * entries < ((int64_t)1 << 31), "too many entries for int32 indices");
  *nnz_host = 0;
  const bool zero_index = (flags & SBX_MTX_ZERO_INDEX) != 0, upper = (flags & SBX_MTX_UPPER_TRIANGLE) != 0;
  const bool expand = symmetry != 0 && !upper;
  SBX_REQUIRE(h, capacity >= (expand ? 2 : 1) * entries, "output capacity: entries (2 * entries for a symmetric expansion)");
  if (it == SBX_I64)
    return sbx_i64_mtx_parse_coordinate(h, vt, text_dev, bytes, n_rows, n_cols, entries, fields, symmetry, flags,
                                        capacity, row_out, col_out, val_out, nnz_host);
  const int vb = (val_out && fields == 3) ? sbx_value_bytes(vt) : 0;
  SBX_REQUIRE(h, vb >= 0, "unknown value type");
  SBX_TRY(sbx_arena_begin(h));
  if (entries == 0) return SBX_OK;
  NestGuard guard(h);
  const uint64_t *pow5 = nullptr;
  SBX_TRY(mx_pow5(h, &pow5));
  const char *text = (const char *)text_dev;
  const int64_t need = entries * fields;
  const unsigned tiles = (unsigned)((bytes + MX_TILE - 1) / MX_TILE);
  unsigned *tile_tokens = nullptr, *tok_off = nullptr, *status = nullptr, *mirror = nullptr;
  SBX_TRY(sbx_salloc(h, (size_t)tiles + 1, &tile_tokens));
  SBX_TRY(sbx_salloc(h, (size_t)need, &tok_off));
  SBX_TRY(sbx_salloc(h, 2, &status));
  SBX_HIP(h, hipMemsetAsync(status, 0, 2 * sizeof(unsigned), h->stream));
  SBX_HIP(h, hipMemsetAsync(tile_tokens + tiles, 0, sizeof(unsigned), h->stream));
  SBX_KLAUNCH(h, SBX_K_MTX, k_mtx_count, dim3(tiles), dim3(MX_THREADS), text, bytes, tile_tokens);
  SBX_LAUNCH_CHECK(h);
  unsigned total_tokens_dev_unused = 0;
  (void)total_tokens_dev_unused;
  SBX_TRY(sbx_exclusive_scan_u32(h, tile_tokens, tile_tokens, (int64_t)tiles + 1, nullptr));
  unsigned total_tokens = 0;
  SBX_TRY(sbx_readback(h, &total_tokens, tile_tokens + tiles, sizeof(unsigned)));
  if ((int64_t)total_tokens < need)
    SBX_FAIL(h, SBX_ERR_BAD_ARG, "sbx_mtx_parse_coordinate: the text holds %u tokens, %lld entries of %d need %lld",
             total_tokens, (long long)entries, fields, (long long)need);
  SBX_KLAUNCH(h, SBX_K_MTX, k_mtx_offsets, dim3(tiles), dim3(MX_THREADS), text, bytes, (const unsigned *)tile_tokens,
              need, tok_off);
  // staging arrays when the entries are expanded, the outputs themselves otherwise
  int32_t *r0 = (int32_t *)row_out, *c0 = (int32_t *)col_out;
  char *v0 = (char *)val_out;
  if (expand) {
    SBX_TRY(sbx_salloc(h, (size_t)entries, &r0));
    SBX_TRY(sbx_salloc(h, (size_t)entries, &c0));
    if (vb) SBX_TRY(sbx_salloc(h, (size_t)entries * vb, &v0));
    SBX_TRY(sbx_salloc(h, (size_t)entries + 1, &mirror));
    SBX_HIP(h, hipMemsetAsync(mirror + entries, 0, sizeof(unsigned), h->stream));
  }
  const int vkind = vb == 0 ? 0 : (vt == SBX_V_F32 ? 2 : vt == SBX_V_F64 ? 3 : 1);
  const int vsigned = (vt == SBX_V_I32 || vt == SBX_V_I64) ? 1 : 0;
  const unsigned grid = (unsigned)((entries + MX_THREADS - 1) / MX_THREADS);
#define PARSE(VK, VBX)                                                                                              \
  SBX_KLAUNCH(h, SBX_K_MTX, (k_mtx_parse<VK, VBX>), dim3(grid), dim3(MX_THREADS), text, bytes,                      \
              (const unsigned *)tok_off, entries, fields, symmetry, zero_index ? 1 : 0, upper ? 1 : 0, n_rows, n_cols, \
              vsigned, pow5, r0, c0, v0, mirror, status)
  if (vkind == 0) PARSE(0, 4);
  else if (vkind == 1 && vb == 4) PARSE(1, 4);
  else if (vkind == 1) PARSE(1, 8);
  else if (vkind == 2) PARSE(2, 4);
  else PARSE(3, 8);
#undef PARSE
  SBX_LAUNCH_CHECK(h);
  int64_t nnz = entries;
  if (expand) {
    unsigned *before = nullptr;
    SBX_TRY(sbx_salloc(h, (size_t)entries + 1, &before));
    SBX_TRY(sbx_exclusive_scan_u32(h, mirror, before, entries + 1, nullptr));
    unsigned mirrors = 0;
    SBX_TRY(sbx_readback(h, &mirrors, before + entries, sizeof(unsigned)));
    nnz = entries + (int64_t)mirrors;
#define EXPAND(VK, VBX)                                                                                           \
  SBX_KLAUNCH(h, SBX_K_MTX, (k_mtx_expand<VK, VBX>), dim3(grid), dim3(MX_THREADS), (const int32_t *)r0,           \
              (const int32_t *)c0, (const char *)v0, (const unsigned *)before, (const unsigned *)mirror, entries,   \
              symmetry == 2 ? 1 : 0, vsigned, (int32_t *)row_out, (int32_t *)col_out, (char *)val_out)
    if (vkind == 0) EXPAND(0, 4);
    else if (vkind == 1 && vb == 4) EXPAND(1, 4);
    else if (vkind == 1) EXPAND(1, 8);
    else if (vkind == 2) EXPAND(2, 4);
    else EXPAND(3, 8);
#undef EXPAND
    SBX_LAUNCH_CHECK(h);
  }
  SBX_PROF_BYTES(h, SBX_K_MTX, bytes + nnz * (int64_t)(8 + vb));
  unsigned st[2] = {0, 0};
  SBX_TRY(sbx_readback(h, st, status, sizeof(st)));
  if (st[0] & MX_TOO_MANY_DIGITS)
    SBX_FAIL(h, SBX_ERR_UNSUPPORTED, "sbx_mtx_parse_coordinate: a value has more than 38 significant digits");
  if (st[0])
    SBX_FAIL(h, SBX_ERR_BAD_ARG, "sbx_mtx_parse_coordinate: malformed %s%s%s token in the coordinate section",
             (st[0] & MX_BAD_INDEX) ? "index " : "", (st[0] & MX_INDEX_RANGE) ? "(index out of range) " : "",
             (st[0] & MX_BAD_VALUE) ? "value" : "");
  *nnz_host = nnz;
  return SBX_OK;
}

// number of whitespace-separated tokens of a text buffer (sizes the outputs of the parsers)
extern "C" int sbx_text_count_tokens(sbx_handle_t h, const void *text_dev, int64_t bytes, int64_t *tokens_host) {
  if (!h) return SBX_ERR_BAD_ARG;
  SBX_REQUIRE(h, tokens_host && bytes >= 0 && (bytes == 0 || text_dev), "bad argument");
  SBX_REQUIRE(h, bytes < ((int64_t)1 << 32), "text sections of 4 GiB and more are not supported (32-bit token offsets)");
  *tokens_host = 0;
  SBX_TRY(sbx_arena_begin(h));
  if (bytes == 0) return SBX_OK;
  NestGuard guard(h);
  const unsigned tiles = (unsigned)((bytes + MX_TILE - 1) / MX_TILE);
  unsigned *tile_tokens = nullptr;
  SBX_TRY(sbx_salloc(h, (size_t)tiles + 1, &tile_tokens));
  SBX_HIP(h, hipMemsetAsync(tile_tokens + tiles, 0, sizeof(unsigned), h->stream));
  SBX_KLAUNCH(h, SBX_K_MTX, k_mtx_count, dim3(tiles), dim3(MX_THREADS), (const char *)text_dev, bytes, tile_tokens);
  SBX_LAUNCH_CHECK(h);
  SBX_TRY(sbx_exclusive_scan_u32(h, tile_tokens, tile_tokens, (int64_t)tiles + 1, nullptr));
  unsigned total = 0;
  SBX_TRY(sbx_readback(h, &total, tile_tokens + tiles, sizeof(unsigned)));
  *tokens_host = total;
  return SBX_OK;
}

// Edge list -> sorted COO: io/edge_list_reader.cc:19-158 (EdgeListReader::ReadCOO).
extern "C" int sbx_edge_list_parse(sbx_handle_t h, sbx_index_type it, sbx_value_type vt, const void *text_dev,
                                   int64_t bytes, int64_t entries, int weighted, unsigned flags, int64_t capacity,
                                   void *row_out, void *col_out, void *val_out, int64_t *dims_nnz_host) {
  if (!h) return SBX_ERR_BAD_ARG;
  if (it == SBX_I32_N64) it = SBX_I32;  // (no offset array)
  SBX_REQUIRE(h, dims_nnz_host && bytes >= 0 && entries >= 0 && (entries == 0 || (text_dev && row_out && col_out)),
              "bad argument");
  const bool undirected = (flags & SBX_EDGES_UNDIRECTED) != 0, remove_self = (flags & SBX_EDGES_REMOVE_SELF) != 0;
  const bool dedup = (flags & SBX_EDGES_REMOVE_DUPLICATES) != 0, square = (flags & SBX_EDGES_SQUARE) != 0;
  SBX_REQUIRE(h, capacity >= (undirected ? 2 : 1) * entries, "output capacity: entries (2 * entries for undirected reads)");
  SBX_REQUIRE(h, 2 * entries < ((int64_t)1 << 31), "too many entries for int32 indices");
  dims_nnz_host[0] = dims_nnz_host[1] = dims_nnz_host[2] = 0;
  if (it == SBX_I64)
    return sbx_i64_edge_list_parse(h, vt, text_dev, bytes, entries, weighted, flags, capacity, row_out, col_out, val_out,
                                   dims_nnz_host);
  const int vb = (weighted && val_out) ? sbx_value_bytes(vt) : 0;
  SBX_REQUIRE(h, vb >= 0, "unknown value type");
  SBX_TRY(sbx_arena_begin(h));
  if (entries == 0) return SBX_OK;
  NestGuard guard(h);
  // (1) tokens -> (u, v, w) in file order: the coordinate parser, 0-based indices, no expansion
  int32_t *u = nullptr, *v = nullptr;
  char *w = nullptr;
  SBX_TRY(sbx_salloc(h, (size_t)entries, &u));
  SBX_TRY(sbx_salloc(h, (size_t)entries, &v));
  if (vb) SBX_TRY(sbx_salloc(h, (size_t)entries * vb, &w));
  int64_t parsed = 0;
  SBX_TRY(sbx_mtx_parse_coordinate(h, SBX_I32, vt, text_dev, bytes, 0, 0, entries, weighted ? 3 : 2, 0, 0u, entries, u, v,
                                   vb ? w : nullptr, &parsed));
  // (2) self-edge filter, reverse edges, dimensions
  unsigned *outputs = nullptr, *before = nullptr;
  EdgeDims *dims = nullptr;
  SBX_TRY(sbx_salloc(h, (size_t)entries + 1, &outputs));
  SBX_TRY(sbx_salloc(h, (size_t)entries + 1, &before));
  SBX_TRY(sbx_salloc(h, 1, &dims));
  SBX_HIP(h, hipMemsetAsync(dims, 0, sizeof(EdgeDims), h->stream));
  SBX_HIP(h, hipMemsetAsync(outputs + entries, 0, sizeof(unsigned), h->stream));
  const unsigned grid = (unsigned)((entries + MX_THREADS - 1) / MX_THREADS);
  SBX_KLAUNCH(h, SBX_K_MTX, k_edge_flags, dim3(grid), dim3(MX_THREADS), (const int32_t *)u, (const int32_t *)v, entries,
              remove_self ? 1 : 0, undirected ? 1 : 0, outputs, dims);
  SBX_TRY(sbx_exclusive_scan_u32(h, outputs, before, entries + 1, nullptr));
  int32_t *r1 = (int32_t *)row_out, *c1 = (int32_t *)col_out;
  char *v1 = (char *)val_out;
  if (dedup) {  // stage the sorted list, compact into the outputs
    SBX_TRY(sbx_salloc(h, (size_t)capacity, &r1));
    SBX_TRY(sbx_salloc(h, (size_t)capacity, &c1));
    if (vb) SBX_TRY(sbx_salloc(h, (size_t)capacity * vb, &v1));
  }
#define EMIT(VBX)                                                                                                       \
  SBX_KLAUNCH(h, SBX_K_MTX, k_edge_emit<VBX>, dim3(grid), dim3(MX_THREADS), (const int32_t *)u, (const int32_t *)v,     \
              (const char *)w, (const unsigned *)before, (const unsigned *)outputs, entries, r1, c1, v1)
  if (vb == 0) EMIT(0);
  else if (vb == 4) EMIT(4);
  else EMIT(8);
#undef EMIT
  SBX_LAUNCH_CHECK(h);
  unsigned count = 0;
  SBX_TRY(sbx_readback(h, &count, before + entries, sizeof(unsigned)));
  EdgeDims hd;
  SBX_TRY(sbx_readback(h, &hd, dims, sizeof(EdgeDims)));
  int64_t n = hd.n, m = hd.m;
  if (square || undirected) {  // :46-49
    n = n > m ? n : m;
    m = n;
  }
  // (3) sort by (row, col) (:51-56; stable here: duplicates keep their file order), (4) std::unique (:58-66)
  int64_t nnz = count;
  if (count > 1) SBX_TRY(sbx_coo_sort(h, SBX_I32, vb ? vt : SBX_V_NONE, n, m, count, r1, c1, vb ? v1 : nullptr));
  if (dedup && count > 0) {
    unsigned *first = nullptr, *fbefore = nullptr;
    SBX_TRY(sbx_salloc(h, (size_t)count + 1, &first));
    SBX_TRY(sbx_salloc(h, (size_t)count + 1, &fbefore));
    SBX_HIP(h, hipMemsetAsync(first + count, 0, sizeof(unsigned), h->stream));
    const unsigned g2 = (unsigned)((count + MX_THREADS - 1) / MX_THREADS);
    SBX_KLAUNCH(h, SBX_K_MTX, k_edge_first, dim3(g2), dim3(MX_THREADS), (const int32_t *)r1, (const int32_t *)c1,
                (int64_t)count, first);
    SBX_TRY(sbx_exclusive_scan_u32(h, first, fbefore, (int64_t)count + 1, nullptr));
#define COMPACT(VBX)                                                                                                   \
  SBX_KLAUNCH(h, SBX_K_MTX, k_edge_compact<VBX>, dim3(g2), dim3(MX_THREADS), (const int32_t *)r1, (const int32_t *)c1, \
              (const char *)v1, (const unsigned *)first, (const unsigned *)fbefore, (int64_t)count, (int32_t *)row_out, \
              (int32_t *)col_out, (char *)val_out)
    if (vb == 0) COMPACT(0);
    else if (vb == 4) COMPACT(4);
    else COMPACT(8);
#undef COMPACT
    SBX_LAUNCH_CHECK(h);
    unsigned uniq = 0;
    SBX_TRY(sbx_readback(h, &uniq, fbefore + count, sizeof(unsigned)));
    nnz = uniq;
  }
  dims_nnz_host[0] = n;
  dims_nnz_host[1] = m;
  dims_nnz_host[2] = nnz;
  return SBX_OK;
}
