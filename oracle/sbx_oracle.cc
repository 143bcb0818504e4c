// sbx_oracle.cc — CPU restatement of SparseBase's reorder / convert / permute path.
//
// TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the HIP path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// load it; nothing under sparsebase_amd/ links, imports or executes it.
//
// Pinning: every function below is checked (tests/test_oracle_*.py) against
//   (1) the known-answer vectors held by the reference's own test-suite
//       (tests/golden/reference_tests.json, transcribed from
//       tests/suites/sparsebase/{functionality_common.inc,converter/common.inc,
//       format/common.inc,format/coo_tests.cc,format/csr_tests.cc}),
//   (2) outputs of the real reference compiled from /root/reference by
//       oracle/Makefile into oracle/_ref/libsbref.so (fixtures under
//       tests/golden/ made by oracle/make_golden.py, plus live randomized
//       comparison whenever that library is present).
//
// Written in C++ (g++) rather than C because GrayReorder's result depends on
// the tie order of libstdc++'s std::sort (reorder/gray_reorder.cc:199,294-299,
// 355-358,404); the same calls are issued here on the same sequences.
//
// All paths are /root/reference/src/sparsebase/… unless noted.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <queue>
#include <sstream>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

namespace {

enum VType { V_NONE = 0, V_I32, V_U32, V_F32, V_I64, V_U64, V_F64 };

inline int vbytes(int vt) {
  switch (vt) {
    case V_NONE: return 0;
    case V_I32: case V_U32: case V_F32: return 4;
    default: return 8;
  }
}

// strict "a < b" on the value payload with the arithmetic type's ordering
// (std::less<std::pair<IDType,ValueType>> second member, format/csr.cc:148)
inline bool val_less(int vt, const void *a, const void *b) {
  switch (vt) {
    case V_I32: { int32_t x, y; memcpy(&x, a, 4); memcpy(&y, b, 4); return x < y; }
    case V_U32: { uint32_t x, y; memcpy(&x, a, 4); memcpy(&y, b, 4); return x < y; }
    case V_F32: { float x, y; memcpy(&x, a, 4); memcpy(&y, b, 4); return x < y; }
    case V_I64: { int64_t x, y; memcpy(&x, a, 8); memcpy(&y, b, 8); return x < y; }
    case V_U64: { uint64_t x, y; memcpy(&x, a, 8); memcpy(&y, b, 8); return x < y; }
    case V_F64: { double x, y; memcpy(&x, a, 8); memcpy(&y, b, 8); return x < y; }
    default: return false;
  }
}

// ---------------------------------------------------------------------------
// A1  COO constructor: sortedness test + sort by (row,col)   format/coo.cc:96-157
// ---------------------------------------------------------------------------
template <typename I>
int coo_is_sorted(int64_t nnz, const I *row, const I *col) {
  I pr = 0, pc = 0;  // coo.cc:97-98: the scan starts from (0,0)
  for (int64_t i = 0; i < nnz; i++) {
    if (pr > row[i] || (pr == row[i] && pc > col[i])) return 0;
    pr = row[i];
    pc = col[i];
  }
  return 1;
}

// The reference's std::sort compares (row,col) only and is unstable, so the
// relative order of duplicate coordinates is unspecified (coo.cc:133-146).
// The oracle (and the HIP path) define it as stable: duplicates keep their
// input order.  Parity with the reference is claimed on duplicate-free input.
template <typename I>
void coo_sort(int vt, int64_t nnz, I *row, I *col, void *val) {
  if (coo_is_sorted(nnz, row, col)) return;
  std::vector<int64_t> idx(nnz);
  for (int64_t i = 0; i < nnz; i++) idx[i] = i;
  std::stable_sort(idx.begin(), idx.end(), [&](int64_t a, int64_t b) {
    if (row[a] != row[b]) return row[a] < row[b];
    return col[a] < col[b];
  });
  std::vector<I> r(row, row + nnz), c(col, col + nnz);
  const int vb = (val != nullptr) ? vbytes(vt) : 0;
  std::vector<char> v;
  if (vb) v.assign((char *)val, (char *)val + nnz * vb);
  for (int64_t i = 0; i < nnz; i++) {
    row[i] = r[idx[i]];
    col[i] = c[idx[i]];
    if (vb) memcpy((char *)val + i * vb, v.data() + idx[i] * vb, vb);
  }
}

// ---------------------------------------------------------------------------
// A4  CSR constructor: any-row-unsorted test, then sort every row by (col,val)
//     format/csr.cc:99-157
// ---------------------------------------------------------------------------
template <typename I>
int csr_rows_sorted(int64_t n, const I *rp, const I *col) {
  for (int64_t i = 0; i < n; i++) {
    I prev = 0;  // csr.cc:107: first entry is compared with 0
    for (I j = rp[i]; j < rp[i + 1]; j++) {
      if (col[j] < prev) return 0;
      prev = col[j];
    }
  }
  return 1;
}

template <typename I>
void csr_sort_rows(int vt, int64_t n, const I *rp, I *col, void *val) {
  if (csr_rows_sorted(n, rp, col)) return;  // csr.cc:118
  const int vb = (val != nullptr) ? vbytes(vt) : 0;
  std::vector<int64_t> idx;
  std::vector<I> c;
  std::vector<char> v;
  for (int64_t i = 0; i < n; i++) {
    const int64_t s = rp[i], e = rp[i + 1], len = e - s;
    if (len <= 1) continue;  // csr.cc:127
    idx.resize(len);
    for (int64_t k = 0; k < len; k++) idx[k] = k;
    // (col,val) lexicographic; entries that compare equal are bit-identical in
    // col and equal in val, so stability is immaterial to the arrays produced
    // (except -0.0 / +0.0, documented in DESIGN.md).
    std::stable_sort(idx.begin(), idx.end(), [&](int64_t a, int64_t b) {
      if (col[s + a] != col[s + b]) return col[s + a] < col[s + b];
      if (!vb) return false;
      return val_less(vt, (char *)val + (s + a) * vb, (char *)val + (s + b) * vb);
    });
    c.assign(col + s, col + e);
    if (vb) v.assign((char *)val + s * vb, (char *)val + e * vb);
    for (int64_t k = 0; k < len; k++) {
      col[s + k] = c[idx[k]];
      if (vb) memcpy((char *)val + (s + k) * vb, v.data() + idx[k] * vb, vb);
    }
  }
}

// ---------------------------------------------------------------------------
// A2  COO -> CSR    converter/converter_order_two.cc:163-212 (move: :215-246)
// ---------------------------------------------------------------------------
template <typename I>
void coo_to_csr(int vt, int64_t n, int64_t nnz, const I *row, const I *col, const void *val,
                I *rp_out, I *col_out, void *val_out) {
  // row_ptr = exclusive scan of the row histogram (the :180-192 three loops)
  std::vector<int64_t> cnt(n + 1, 0);
  for (int64_t i = 0; i < nnz; i++) cnt[row[i]]++;
  int64_t run = 0;
  for (int64_t r = 0; r <= n; r++) {
    rp_out[r] = (I)run;
    if (r < n) run += cnt[r];
  }
  if (col_out) memcpy(col_out, col, nnz * sizeof(I));
  const int vb = vbytes(vt);
  if (val_out && val && vb) memcpy(val_out, val, nnz * vb);
}

// ---------------------------------------------------------------------------
// A3  CSR -> COO    converter/converter_order_two.cc:72-118 (move: :131-160)
// ---------------------------------------------------------------------------
template <typename I>
void csr_to_coo(int vt, int64_t n, int64_t nnz, const I *rp, const I *col, const void *val,
                I *row_out, I *col_out, void *val_out) {
  int64_t w = 0;
  for (int64_t r = 0; r < n; r++)
    for (I j = rp[r]; j < rp[r + 1]; j++) row_out[w++] = (I)r;
  if (col_out) memcpy(col_out, col, nnz * sizeof(I));
  const int vb = vbytes(vt);
  if (val_out && val && vb) memcpy(val_out, val, nnz * vb);
}

// ---------------------------------------------------------------------------
// A14 COO -> CSC    converter/converter_order_two.cc:21-70
// Column histogram + inclusive scan (:45-50), then every nonzero in input order goes to
// the next free slot of its column (:53-62): a stable counting sort by column.  The CSC
// constructor that receives the arrays (:66-68 -> format/csc.cc:99-157) checks every
// column and, if any has decreasing rows, sorts every column's (row, value) pairs — the
// same procedure as the CSR constructor with the roles of the dimensions swapped.
// The reference sizes col_ptr and its counters by n, the ROW count (:32-33), which is
// only memory-safe for n == m; this restatement uses m + 1 entries (equal when square).
// A15 CSR -> CSC    :120-128 = CSR -> COO -> CSC.
// ---------------------------------------------------------------------------
template <typename I>
void coo_to_csc(int vt, int64_t n, int64_t m, int64_t nnz, const I *row, const I *col, const void *val,
                I *cp_out, I *row_out, void *val_out) {
  (void)n;
  const int vb = (val && val_out) ? vbytes(vt) : 0;
  std::vector<int64_t> start(m + 1, 0);
  for (int64_t i = 0; i < nnz; i++) start[col[i] + 1]++;
  for (int64_t c = 1; c <= m; c++) start[c] += start[c - 1];
  for (int64_t c = 0; c <= m; c++) cp_out[c] = (I)start[c];
  std::vector<int64_t> next(start.begin(), start.end() - 1);
  for (int64_t i = 0; i < nnz; i++) {
    const int64_t o = next[col[i]]++;
    row_out[o] = row[i];
    if (vb) memcpy((char *)val_out + o * vb, (const char *)val + i * vb, vb);
  }
  csr_sort_rows<I>(vb ? vt : V_NONE, m, cp_out, row_out, vb ? val_out : nullptr);  // csc.cc:99-157
}

template <typename I>
void csr_to_csc(int vt, int64_t n, int64_t m, int64_t nnz, const I *rp, const I *col, const void *val,
                I *cp_out, I *row_out, void *val_out) {
  std::vector<I> rows(nnz > 0 ? nnz : 1);
  csr_to_coo<I>(V_NONE, n, nnz, rp, col, nullptr, rows.data(), nullptr, nullptr);
  coo_to_csc<I>(vt, n, m, nnz, rows.data(), col, val, cp_out, row_out, val_out);
}

// ---------------------------------------------------------------------------
// Matrix Market coordinate section -> COO   io/mtx_reader.cc:307-495 (ReadCoordinateIntoCOO)
// The same stream extractions as the reference (`fin >> m >> n [>> w]` into IDType / ValueType
// variables), on the bytes that follow the size line.  Returns 0, or 1 when the stream failed.
// ---------------------------------------------------------------------------
template <typename I, typename V>
int mtx_parse(const char *text, int64_t bytes, int64_t L, int fields, int symmetry, int zero_index, int upper,
              I *row, I *col, V *val, int64_t *nnz_out) {
  std::istringstream fin(std::string(text, (size_t)bytes));
  int64_t nnz = 0;
  const bool weighted = fields == 3;
  for (int64_t l = 0; l < L; l++) {
    I m, n;
    fin >> m >> n;
    V w = V();
    if (weighted) {
      if (val) fin >> w;
      else { std::string skip; fin >> skip; }
    }
    if (!fin) return 1;
    if (zero_index) { n--; m--; }
    if (symmetry == 0) {                       // :322-366
      row[nnz] = m; col[nnz] = n;
      if (val) val[nnz] = w;
      nnz++;
    } else if (upper) {                        // :368-384
      row[nnz] = std::min(m, n); col[nnz] = std::max(m, n);
      if (val) val[nnz] = w;
      nnz++;
    } else {                                   // :403-470
      row[nnz] = m; col[nnz] = n;
      if (val) val[nnz] = w;
      nnz++;
      const bool check_diagonal = symmetry != 2;
      if (!check_diagonal || m != n) {
        row[nnz] = n; col[nnz] = m;
        if (val) val[nnz] = symmetry == 2 ? (V)(-val[nnz - 1]) : val[nnz - 1];
        nnz++;
      }
    }
  }
  *nnz_out = nnz;
  return 0;
}

// ---------------------------------------------------------------------------
// Edge list -> sorted COO   io/edge_list_reader.cc:19-158 (EdgeListReader::ReadCOO), same stream
// extractions and the same std::sort / std::unique calls.  dims_nnz = {n, m, nnz}.
// ---------------------------------------------------------------------------
template <typename I, typename V>
int edge_list_parse(const char *text, int64_t bytes, int weighted, int remove_duplicates, int remove_self_edges,
                    int read_undirected, int square, I *row, I *col, V *val, int64_t *dims_nnz) {
  std::istringstream infile(std::string(text, (size_t)bytes));
  I u, v;
  V w = 0;
  I m = 0, n = 0;
  std::vector<std::tuple<I, I, V>> edges;
  while (infile >> u >> v) {
    if (weighted) infile >> w;
    if (u != v || !remove_self_edges) {
      edges.push_back(std::tuple<I, I, V>(u, v, w));
      if (read_undirected) edges.push_back(std::tuple<I, I, V>(v, u, w));
      n = std::max(n, (I)(u + 1));
      m = std::max(m, (I)(v + 1));
    }
  }
  if (square || read_undirected) {
    n = std::max(n, m);
    m = n;
  }
  std::sort(edges.begin(), edges.end(), [](const std::tuple<I, I, V> &t1, const std::tuple<I, I, V> t2) {
    if (std::get<0>(t1) == std::get<0>(t2)) return std::get<1>(t1) < std::get<1>(t2);
    return std::get<0>(t1) < std::get<0>(t2);
  });
  if (remove_duplicates) {
    auto it = std::unique(edges.begin(), edges.end(), [](const std::tuple<I, I, V> &t1, const std::tuple<I, I, V> t2) {
      return std::get<0>(t1) == std::get<0>(t2) && std::get<1>(t1) == std::get<1>(t2);
    });
    edges.erase(it, edges.end());
  }
  for (size_t i = 0; i < edges.size(); i++) {
    row[i] = std::get<0>(edges[i]);
    col[i] = std::get<1>(edges[i]);
    if (weighted && val) val[i] = std::get<2>(edges[i]);
  }
  dims_nnz[0] = n;
  dims_nnz[1] = m;
  dims_nnz[2] = (int64_t)edges.size();
  return 0;
}

// ---------------------------------------------------------------------------
// Features
//   bandwidth   feature/bandwidth.cc:93-112: max over nonzeros of |i - j| + 1, 0 without nonzeros
//   profile     feature/profile.cc:91-105:   sum over rows of i - min(i, smallest column of the row); the
//               reference accumulates in IDType — this returns the exact sum, callers narrow it
//   degrees     feature/degrees.cc:93-105
//   degree distribution  feature/degree_distribution.cc:152-167: degree / (FeatureType)num_edges
// ---------------------------------------------------------------------------
template <typename I>
int64_t csr_bandwidth(int64_t n, const I *rp, const I *col) {
  int64_t bw = 0;
  for (int64_t i = 0; i < n; i++)
    for (I k = rp[i]; k < rp[i + 1]; k++) {
      const int64_t j = col[k];
      const int64_t d = (i >= j ? i - j : j - i) + 1;
      if (bw < d) bw = d;
    }
  return bw;
}
template <typename I>
int64_t csr_profile(int64_t n, const I *rp, const I *col) {
  int64_t sum = 0;
  for (int64_t i = 0; i < n; i++) {
    int64_t j = i;
    for (I k = rp[i]; k < rp[i + 1]; k++)
      if (j > col[k]) j = col[k];
    sum += i - j;
  }
  return sum;
}
template <typename I, typename F>
void csr_degree_distribution(int64_t n, int64_t nnz, const I *rp, F *out) {
  for (int64_t i = 0; i < n; i++) out[i] = (rp[i + 1] - rp[i]) / (F)nnz;
}

// ---------------------------------------------------------------------------
// A6  DegreeReorder   reorder/degree_reorder.cc:22-62
// Intended semantics (the reference indexes `mr` one past its end, :41-45):
// rows placed from the END of their degree bucket in id order, i.e. the final
// sequence is (degree ascending, id descending); !ascending reverses it all.
// ---------------------------------------------------------------------------
template <typename I>
void degree_reorder(int64_t n, const I *rp, int ascending, I *inv) {
  int64_t top = n;  // the reference sizes its buckets by n (:29); wider rows are UB there
  for (int64_t u = 0; u < n; u++) top = std::max<int64_t>(top, rp[u + 1] - rp[u]);
  std::vector<int64_t> bucket_end(top + 2, 0);
  for (int64_t u = 0; u < n; u++) bucket_end[(int64_t)(rp[u + 1] - rp[u])]++;
  for (int64_t d = 1; d <= top; d++) bucket_end[d] += bucket_end[d - 1];
  std::vector<I> seq(n);
  std::vector<int64_t> used(top + 2, 0);
  for (int64_t u = 0; u < n; u++) {
    const int64_t d = rp[u + 1] - rp[u];
    seq[bucket_end[d] - used[d] - 1] = (I)u;  // fill bucket back to front
    used[d]++;
  }
  if (!ascending) std::reverse(seq.begin(), seq.end());
  for (int64_t k = 0; k < n; k++) inv[seq[k]] = (I)k;
}

// ---------------------------------------------------------------------------
// A7  RCMReorder    reorder/rcm_reorder.cc:22-81 (peripheral), :83-166
// ---------------------------------------------------------------------------
template <typename I>
struct RcmStats {
  int64_t components = 0, isolated = 0, sweeps_max = 0, levels = 0, edges = 0, largest = 0;
};

// pseudo-peripheral search: repeated FIFO BFS; `ecc` persists across sweeps
// exactly like `qlevel` (never reset, rcm_reorder.cc:31-56).
template <typename I>
I pseudo_peripheral(const I *xadj, const I *adj, I start, std::vector<int64_t> &dist,
                    std::vector<I> &bfsq, int64_t *sweeps, int64_t *levels, int64_t *edges) {
  I root = start;
  int64_t prev_ecc = -1, ecc = 0;
  while (prev_ecc != ecc) {
    prev_ecc = ecc;
    int64_t head = 0, tail = 0;
    dist[root] = 0;
    bfsq[tail++] = root;
    while (head < tail) {
      const I u = bfsq[head++];
      for (I p = xadj[u]; p < xadj[u + 1]; p++) {
        const I v = adj[p];
        (*edges)++;
        if (dist[v] < 0) {
          dist[v] = dist[u] + 1;
          bfsq[tail++] = v;
          if (dist[v] > ecc) ecc = dist[v];
        }
      }
    }
    (*sweeps)++;
    (*levels) += ecc + 1;
    if (head == ecc + 1) return root;  // one vertex per level: a path end (:58)
    if (prev_ecc != ecc) {
      // among the deepest vertices, in queue order, the strictly smallest degree
      bool have = false;
      int64_t best = 0;
      for (int64_t i = 0; i < head; i++) {
        const I w = bfsq[i];
        if (dist[w] == ecc) {
          const int64_t d = xadj[w + 1] - xadj[w];
          if (!have) { best = d + 1; have = true; }
          if (d < best) { best = d; root = w; }
        }
        dist[w] = -1;
      }
    }
  }
  return root;
}

template <typename I>
void rcm_reorder(int64_t n, const I *xadj, const I *adj, I *inv, int64_t *stats8) {
  std::vector<I> order(n);      // Cuthill-McKee visiting order, all components
  std::vector<I> bfsq(n);
  std::vector<int64_t> dist(n, -1);
  std::vector<char> seen(n, 0);
  typedef std::pair<I, I> DegId;
  std::priority_queue<DegId, std::vector<DegId>, std::greater<DegId>> heap;
  int64_t tail = 0;
  int64_t comps = 0, isolated = 0, sw_max = 0, lv_tot = 0, edges = 0, largest = 0;
  for (int64_t i = 0; i < n; i++) {
    if (seen[i]) continue;
    comps++;
    if (xadj[i] == xadj[i + 1]) {  // empty row: placed as is (:111-116)
      order[tail] = (I)i;
      inv[i] = (I)tail;
      tail++;
      seen[i] = 1;
      isolated++;
      continue;
    }
    int64_t sw = 0, lv = 0;
    const I r = pseudo_peripheral<I>(xadj, adj, (I)i, dist, bfsq, &sw, &lv, &edges);
    const int64_t seg = tail;
    int64_t head = tail;
    seen[r] = 1;
    order[tail++] = r;
    while (head < tail) {
      const I u = order[head++];
      for (I p = xadj[u]; p < xadj[u + 1]; p++) {
        const I v = adj[p];
        edges++;
        if (!seen[v]) {
          seen[v] = 1;
          heap.push(DegId((I)(xadj[v + 1] - xadj[v]), v));
        }
      }
      while (!heap.empty()) {  // children of u in (degree,id) order (:139-143)
        order[tail++] = heap.top().second;
        heap.pop();
      }
    }
    const int64_t sz = tail - seg;
    // reversed within the component (:146-153), then inverted (:158-160)
    for (int64_t k = 0; k < sz; k++) inv[order[seg + k]] = (I)(seg + sz - 1 - k);
    if (sz > largest) { largest = sz; sw_max = sw + 1; lv_tot = lv; }
  }
  if (stats8) {
    stats8[0] = comps; stats8[1] = isolated; stats8[2] = 0; stats8[3] = 0;
    stats8[4] = sw_max; stats8[5] = lv_tot; stats8[6] = edges; stats8[7] = largest;
  }
}

// ---------------------------------------------------------------------------
// A8  GrayReorder   reorder/gray_reorder.cc:106-424
// Returns 0, or -1 when the reference itself is undefined on the shape
// (bucket index col/row_split >= resolution, :251/:382).
// ---------------------------------------------------------------------------
inline uint64_t gray_decode(uint64_t g) {  // :38-46, prefix-xor
  uint64_t b = 0;
  for (; g; g >>= 1) b ^= g;
  return b;
}

template <typename I>
int gray_reorder(int64_t n_rows, int64_t n_cols, const I *rp, const I *col, int resolution,
                 int nnz_threshold, int group_size, I *inv) {
  typedef std::pair<I, unsigned long> RowKey;
  auto deg = [&](int64_t r) -> int64_t { return (int64_t)(rp[r + 1] - rp[r]); };
  int bits = resolution;
  if (n_cols < bits) bits = (int)n_cols;  // :206-208
  if (bits <= 0) return -1;
  const int64_t width = n_cols / bits;  // :210
  if (width * bits != n_cols) return -1;  // reference overflows its bucket array
  const int64_t band = n_cols / 128;      // :138

  std::vector<I> sparse_rows, dense_rows;
  int64_t nnz_s = 0, diag_s = 0, nnz_d = 0, diag_d = 0;
  for (int64_t i = 0; i < n_rows; i++) {
    const bool is_sparse = deg(i) <= nnz_threshold;  // :150
    int64_t in_band = 0;
    for (I j = rp[i]; j < rp[i + 1]; j++) {
      const int64_t c = col[j];
      const int64_t dist = (c >= i) ? (c - i) : (i - c);
      if (dist <= band) in_band++;
    }
    if (is_sparse) { sparse_rows.push_back((I)i); nnz_s += deg(i); diag_s += in_band; }
    else { dense_rows.push_back((I)i); nnz_d += deg(i); diag_d += in_band; }
  }
  // the counters are `int` in the reference (:134-137)
  const bool sparse_banded = double((int)diag_s) / (int)nnz_s > 0.3;  // :181
  const bool dense_banded = double((int)diag_d) / (int)nnz_d > 0.2;   // :186

  // :199-203 — unstable sort by degree; the SAME std::sort call on the same sequence
  std::sort(sparse_rows.begin(), sparse_rows.end(),
            [&](int a, int b) -> bool { return deg(a) < deg(b); });

  auto asc = [](const RowKey &l, const RowKey &r) { return l.second < r.second; };
  auto desc = [](const RowKey &l, const RowKey &r) { return l.second > r.second; };

  auto bitmap_key = [&](int64_t r, int64_t thr) -> unsigned long {
    std::vector<int64_t> cnt(bits, 0);
    for (I j = rp[r]; j < rp[r + 1]; j++) cnt[col[j] / width]++;
    uint64_t bm = 0;
    for (int b = 0; b < bits; b++)
      if (cnt[b] > thr) bm += (uint64_t)1 << b;  // pow(2,b), exact (:263,:393)
    return (unsigned long)gray_decode(bm);
  };

  std::vector<RowKey> section;
  section.reserve(n_rows);
  if (!sparse_banded) {  // :223
    bool descending = false;
    int64_t start = 0, last_deg = 0;
    int groups = 0;
    const int64_t ns = (int64_t)sparse_rows.size();
    auto flush = [&](int64_t end) {
      if (!descending) std::sort(section.begin(), section.end(), asc);
      else std::sort(section.begin(), section.end(), desc);
      descending = !descending;
      for (int64_t a = start; a < end; a++) sparse_rows[a] = section[a - start].first;
    };
    for (int64_t i = 0; i < ns; i++) {
      const int64_t d = deg(sparse_rows[i]);
      if (i == 0) { last_deg = d; start = 0; }
      if (d == 0) {  // :235-242 — empty rows never enter a section
        start = i + 1;
        if (i + 1 < ns) last_deg = deg(sparse_rows[i + 1]);
        continue;
      }
      const unsigned long key = bitmap_key(sparse_rows[i], 0);
      if (i != 0 && last_deg != d) {  // :271-329
        groups++;
        last_deg = d;
        if (groups == group_size) {
          flush(i);
          start = i;
          section.clear();
          groups = 0;
        }
      }
      section.push_back(RowKey(sparse_rows[i], key));
      if (i == ns - 1) flush(ns);  // :348-364
    }
    section.clear();
  }
  if (!dense_banded) {  // :369-410
    const int64_t nd = (int64_t)dense_rows.size();
    for (int64_t i = 0; i < nd; i++) {
      const int64_t r = dense_rows[i];
      section.push_back(RowKey((I)r, bitmap_key(r, deg(r) / bits)));
    }
    std::sort(section.begin(), section.end(), asc);
    for (int64_t a = 0; a < nd; a++) dense_rows[a] = section[a].first;
    section.clear();
  }
  int64_t pos = 0;
  for (I r : sparse_rows) inv[r] = (I)pos++;
  for (I r : dense_rows) inv[r] = (I)pos++;
  return 0;
}

// Per-row Gray keys exactly as the device stage exports them (sbx_gray_row_keys).
template <typename I>
int gray_row_keys(int64_t n_rows, int64_t n_cols, const I *rp, const I *col, int resolution,
                  int nnz_threshold, I *degree_out, uint64_t *key_out, int64_t *counts4) {
  int bits = resolution;
  if (n_cols < bits) bits = (int)n_cols;
  if (bits <= 0) return -1;
  const int64_t width = n_cols / bits;
  if (width * bits != n_cols) return -1;
  const int64_t band = n_cols / 128;
  counts4[0] = counts4[1] = counts4[2] = counts4[3] = 0;
  std::vector<int64_t> cnt(bits);
  for (int64_t i = 0; i < n_rows; i++) {
    const int64_t d = rp[i + 1] - rp[i];
    const bool is_sparse = d <= nnz_threshold;
    const int64_t thr = is_sparse ? 0 : d / bits;
    std::fill(cnt.begin(), cnt.end(), 0);
    int64_t in_band = 0;
    for (I j = rp[i]; j < rp[i + 1]; j++) {
      const int64_t c = col[j];
      cnt[c / width]++;
      if (((c >= i) ? (c - i) : (i - c)) <= band) in_band++;
    }
    uint64_t bm = 0;
    for (int b = 0; b < bits; b++)
      if (cnt[b] > thr) bm |= (uint64_t)1 << b;
    degree_out[i] = (I)d;
    key_out[i] = gray_decode(bm);
    counts4[is_sparse ? 0 : 2] += d;
    counts4[is_sparse ? 1 : 3] += in_band;
  }
  return 0;
}

// ---------------------------------------------------------------------------
// A5  PermuteOrderTwoCSR + the CSR constructor it ends in
//     permute/permute_order_two.cc:23-79, format/csr.cc:99-157
// ---------------------------------------------------------------------------
template <typename I>
void permute_csr(int vt, int64_t n, const I *rp, const I *col, const void *val,
                 const I *row_order, const I *col_order, I *rp_out, I *col_out, void *val_out) {
  const int vb = (val && val_out) ? vbytes(vt) : 0;
  std::vector<I> old_of_new(n);
  for (int64_t i = 0; i < n; i++) old_of_new[row_order ? row_order[i] : i] = (I)i;  // :46-48
  int64_t w = 0;
  rp_out[0] = 0;
  for (int64_t i = 0; i < n; i++) {
    const int64_t u = old_of_new[i];
    for (I p = rp[u]; p < rp[u + 1]; p++) {
      col_out[w] = col_order ? col_order[col[p]] : col[p];  // :68
      if (vb) memcpy((char *)val_out + w * vb, (const char *)val + (int64_t)p * vb, vb);
      w++;
    }
    rp_out[i + 1] = (I)w;
  }
  csr_sort_rows<I>(vb ? vt : V_NONE, n, rp_out, col_out, vb ? val_out : nullptr);  // :76-77
}

template <typename I>
void inverse_permutation(int64_t n, const I *perm, I *inv) {  // bases/reorder_base.h:663-672
  for (int64_t i = 0; i < n; i++) inv[perm[i]] = (I)i;
}

// PermuteOrderOne::PermuteArray   permute/permute_order_one.cc:18-37
template <typename I>
void permute_array(int vt, int64_t n, const I *order, const void *vals, void *out) {
  const int vb = vbytes(vt);
  for (int64_t i = 0; i < n; i++)
    memcpy((char *)out + (int64_t)order[i] * vb, (const char *)vals + i * vb, vb);
}

}  // namespace

// ---------------------------------------------------------------------------
// C entry points (ctypes): it = 0 -> int32 indices, 1 -> int64 indices
// ---------------------------------------------------------------------------
#define DISPATCH(it, CALL32, CALL64) \
  do { if ((it) == 0) { CALL32; } else { CALL64; } } while (0)

extern "C" {

int orc_coo_is_sorted(int it, int64_t nnz, const void *row, const void *col) {
  if (it == 0) return coo_is_sorted<int32_t>(nnz, (const int32_t *)row, (const int32_t *)col);
  return coo_is_sorted<int64_t>(nnz, (const int64_t *)row, (const int64_t *)col);
}
void orc_coo_sort(int it, int vt, int64_t nnz, void *row, void *col, void *val) {
  DISPATCH(it, coo_sort<int32_t>(vt, nnz, (int32_t *)row, (int32_t *)col, val),
           coo_sort<int64_t>(vt, nnz, (int64_t *)row, (int64_t *)col, val));
}
int orc_csr_rows_sorted(int it, int64_t n, const void *rp, const void *col) {
  if (it == 0) return csr_rows_sorted<int32_t>(n, (const int32_t *)rp, (const int32_t *)col);
  return csr_rows_sorted<int64_t>(n, (const int64_t *)rp, (const int64_t *)col);
}
void orc_csr_sort_rows(int it, int vt, int64_t n, const void *rp, void *col, void *val) {
  DISPATCH(it, csr_sort_rows<int32_t>(vt, n, (const int32_t *)rp, (int32_t *)col, val),
           csr_sort_rows<int64_t>(vt, n, (const int64_t *)rp, (int64_t *)col, val));
}
void orc_coo_to_csr(int it, int vt, int64_t n, int64_t nnz, const void *row, const void *col,
                    const void *val, void *rp_out, void *col_out, void *val_out) {
  DISPATCH(it,
           coo_to_csr<int32_t>(vt, n, nnz, (const int32_t *)row, (const int32_t *)col, val,
                               (int32_t *)rp_out, (int32_t *)col_out, val_out),
           coo_to_csr<int64_t>(vt, n, nnz, (const int64_t *)row, (const int64_t *)col, val,
                               (int64_t *)rp_out, (int64_t *)col_out, val_out));
}
void orc_csr_to_coo(int it, int vt, int64_t n, int64_t nnz, const void *rp, const void *col,
                    const void *val, void *row_out, void *col_out, void *val_out) {
  DISPATCH(it,
           csr_to_coo<int32_t>(vt, n, nnz, (const int32_t *)rp, (const int32_t *)col, val,
                               (int32_t *)row_out, (int32_t *)col_out, val_out),
           csr_to_coo<int64_t>(vt, n, nnz, (const int64_t *)rp, (const int64_t *)col, val,
                               (int64_t *)row_out, (int64_t *)col_out, val_out));
}
void orc_coo_to_csc(int it, int vt, int64_t n, int64_t m, int64_t nnz, const void *row, const void *col,
                    const void *val, void *cp_out, void *row_out, void *val_out) {
  DISPATCH(it,
           coo_to_csc<int32_t>(vt, n, m, nnz, (const int32_t *)row, (const int32_t *)col, val,
                               (int32_t *)cp_out, (int32_t *)row_out, val_out),
           coo_to_csc<int64_t>(vt, n, m, nnz, (const int64_t *)row, (const int64_t *)col, val,
                               (int64_t *)cp_out, (int64_t *)row_out, val_out));
}
void orc_csr_to_csc(int it, int vt, int64_t n, int64_t m, int64_t nnz, const void *rp, const void *col,
                    const void *val, void *cp_out, void *row_out, void *val_out) {
  DISPATCH(it,
           csr_to_csc<int32_t>(vt, n, m, nnz, (const int32_t *)rp, (const int32_t *)col, val,
                               (int32_t *)cp_out, (int32_t *)row_out, val_out),
           csr_to_csc<int64_t>(vt, n, m, nnz, (const int64_t *)rp, (const int64_t *)col, val,
                               (int64_t *)cp_out, (int64_t *)row_out, val_out));
}
int orc_mtx_parse(int it, int vt, const char *text, int64_t bytes, int64_t entries, int fields, int symmetry,
                  int zero_index, int upper, void *row, void *col, void *val, int64_t *nnz) {
#define MTX_CALL(I, V) \
  return mtx_parse<I, V>(text, bytes, entries, fields, symmetry, zero_index, upper, (I *)row, (I *)col, (V *)val, nnz)
  if (it == 0) {
    switch (vt) {
      case V_NONE: MTX_CALL(int32_t, int32_t);
      case V_I32: MTX_CALL(int32_t, int32_t);
      case V_U32: MTX_CALL(int32_t, uint32_t);
      case V_F32: MTX_CALL(int32_t, float);
      case V_I64: MTX_CALL(int32_t, int64_t);
      case V_U64: MTX_CALL(int32_t, uint64_t);
      case V_F64: MTX_CALL(int32_t, double);
    }
  } else {
    switch (vt) {
      case V_NONE: MTX_CALL(int64_t, int32_t);
      case V_I32: MTX_CALL(int64_t, int32_t);
      case V_U32: MTX_CALL(int64_t, uint32_t);
      case V_F32: MTX_CALL(int64_t, float);
      case V_I64: MTX_CALL(int64_t, int64_t);
      case V_U64: MTX_CALL(int64_t, uint64_t);
      case V_F64: MTX_CALL(int64_t, double);
    }
  }
#undef MTX_CALL
  return 2;
}
int orc_edge_list_parse(int it, int vt, const char *text, int64_t bytes, int weighted, int remove_duplicates,
                        int remove_self_edges, int read_undirected, int square, void *row, void *col, void *val,
                        int64_t *dims_nnz) {
#define EL_CALL(I, V)                                                                                              \
  return edge_list_parse<I, V>(text, bytes, weighted, remove_duplicates, remove_self_edges, read_undirected, square, \
                               (I *)row, (I *)col, (V *)val, dims_nnz)
  if (it == 0) {
    if (vt == V_F32) EL_CALL(int32_t, float);
    if (vt == V_F64) EL_CALL(int32_t, double);
    if (vt == V_I64) EL_CALL(int32_t, int64_t);
    EL_CALL(int32_t, int32_t);
  }
  if (vt == V_F32) EL_CALL(int64_t, float);
  if (vt == V_F64) EL_CALL(int64_t, double);
  if (vt == V_I64) EL_CALL(int64_t, int64_t);
  EL_CALL(int64_t, int32_t);
#undef EL_CALL
}
int64_t orc_csr_bandwidth(int it, int64_t n, const void *rp, const void *col) {
  if (it == 0) return csr_bandwidth<int32_t>(n, (const int32_t *)rp, (const int32_t *)col);
  return csr_bandwidth<int64_t>(n, (const int64_t *)rp, (const int64_t *)col);
}
int64_t orc_csr_profile(int it, int64_t n, const void *rp, const void *col) {
  if (it == 0) return csr_profile<int32_t>(n, (const int32_t *)rp, (const int32_t *)col);
  return csr_profile<int64_t>(n, (const int64_t *)rp, (const int64_t *)col);
}
void orc_csr_degrees(int it, int64_t n, const void *rp, void *out) {
  if (it == 0) for (int64_t i = 0; i < n; i++) ((int32_t *)out)[i] = ((const int32_t *)rp)[i + 1] - ((const int32_t *)rp)[i];
  else for (int64_t i = 0; i < n; i++) ((int64_t *)out)[i] = ((const int64_t *)rp)[i + 1] - ((const int64_t *)rp)[i];
}
void orc_csr_degree_distribution(int it, int fbytes, int64_t n, int64_t nnz, const void *rp, void *out) {
  if (it == 0 && fbytes == 4) csr_degree_distribution<int32_t, float>(n, nnz, (const int32_t *)rp, (float *)out);
  else if (it == 0) csr_degree_distribution<int32_t, double>(n, nnz, (const int32_t *)rp, (double *)out);
  else if (fbytes == 4) csr_degree_distribution<int64_t, float>(n, nnz, (const int64_t *)rp, (float *)out);
  else csr_degree_distribution<int64_t, double>(n, nnz, (const int64_t *)rp, (double *)out);
}
void orc_degree_reorder(int it, int64_t n, const void *rp, int ascending, void *inv) {
  DISPATCH(it, degree_reorder<int32_t>(n, (const int32_t *)rp, ascending, (int32_t *)inv),
           degree_reorder<int64_t>(n, (const int64_t *)rp, ascending, (int64_t *)inv));
}
void orc_rcm_reorder(int it, int64_t n, const void *rp, const void *col, void *inv,
                     int64_t *stats8) {
  DISPATCH(it,
           rcm_reorder<int32_t>(n, (const int32_t *)rp, (const int32_t *)col, (int32_t *)inv,
                                stats8),
           rcm_reorder<int64_t>(n, (const int64_t *)rp, (const int64_t *)col, (int64_t *)inv,
                                stats8));
}
int orc_gray_reorder(int it, int64_t n, int64_t m, const void *rp, const void *col,
                     int resolution, int nnz_threshold, int group_size, void *inv) {
  if (it == 0)
    return gray_reorder<int32_t>(n, m, (const int32_t *)rp, (const int32_t *)col, resolution,
                                 nnz_threshold, group_size, (int32_t *)inv);
  return gray_reorder<int64_t>(n, m, (const int64_t *)rp, (const int64_t *)col, resolution,
                               nnz_threshold, group_size, (int64_t *)inv);
}
int orc_gray_row_keys(int it, int64_t n, int64_t m, const void *rp, const void *col,
                      int resolution, int nnz_threshold, void *degree_out, uint64_t *key_out,
                      int64_t *counts4) {
  if (it == 0)
    return gray_row_keys<int32_t>(n, m, (const int32_t *)rp, (const int32_t *)col, resolution,
                                  nnz_threshold, (int32_t *)degree_out, key_out, counts4);
  return gray_row_keys<int64_t>(n, m, (const int64_t *)rp, (const int64_t *)col, resolution,
                                nnz_threshold, (int64_t *)degree_out, key_out, counts4);
}
void orc_permute_csr(int it, int vt, int64_t n, const void *rp, const void *col,
                     const void *val, const void *row_order, const void *col_order,
                     void *rp_out, void *col_out, void *val_out) {
  DISPATCH(it,
           permute_csr<int32_t>(vt, n, (const int32_t *)rp, (const int32_t *)col, val,
                                (const int32_t *)row_order, (const int32_t *)col_order,
                                (int32_t *)rp_out, (int32_t *)col_out, val_out),
           permute_csr<int64_t>(vt, n, (const int64_t *)rp, (const int64_t *)col, val,
                                (const int64_t *)row_order, (const int64_t *)col_order,
                                (int64_t *)rp_out, (int64_t *)col_out, val_out));
}
void orc_inverse_permutation(int it, int64_t n, const void *perm, void *inv) {
  DISPATCH(it, inverse_permutation<int32_t>(n, (const int32_t *)perm, (int32_t *)inv),
           inverse_permutation<int64_t>(n, (const int64_t *)perm, (int64_t *)inv));
}
void orc_permute_array(int it, int vt, int64_t n, const void *order, const void *vals,
                       void *out) {
  DISPATCH(it, permute_array<int32_t>(vt, n, (const int32_t *)order, vals, out),
           permute_array<int64_t>(vt, n, (const int64_t *)order, vals, out));
}

}  // extern "C"
