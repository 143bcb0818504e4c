// ref_driver.cc — thin C entry points around the REAL SparseBase reference.
//
// TEST INFRASTRUCTURE ONLY.  Compiled by oracle/Makefile against the headers
// and sources where they lie under /root/reference/src (header-only mode, no
// reference file is copied into this repository); the output goes to
// oracle/_ref/libsbref.so, which is git-ignored.  It exists to (a) pin the
// CPU restatement in sbx_oracle.cc, (b) generate tests/golden fixtures
// (oracle/make_golden.py) and (c) optionally serve as bench.py's
// cpu_baseline of kind "reference".
//
// Signatures mirror the orc_* functions of sbx_oracle.cc so that the same
// ctypes prototypes drive both.  it/vt combinations built:
//   it=0: <int,int,void|int|float>, plus <unsigned,unsigned,unsigned> (vt=2)
//   it=1: <int64,int64,void|double>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>

#include "sparsebase/bases/reorder_base.h"
#include "sparsebase/context/cpu_context.h"
#include "sparsebase/feature/bandwidth.h"
#include "sparsebase/feature/degree_distribution.h"
#include "sparsebase/feature/degrees.h"
#include "sparsebase/feature/profile.h"
#include "sparsebase/format/coo.h"
#include "sparsebase/io/binary_reader_order_one.h"
#include "sparsebase/io/binary_reader_order_two.h"
#include "sparsebase/io/binary_writer_order_one.h"
#include "sparsebase/io/binary_writer_order_two.h"
#include "sparsebase/io/edge_list_reader.h"
#include "sparsebase/io/mtx_reader.h"
#include "sparsebase/format/csc.h"
#include "sparsebase/format/csr.h"
#include "sparsebase/permute/permute_order_two.h"
#include "sparsebase/reorder/degree_reorder.h"
#include "sparsebase/reorder/gray_reorder.h"
#include "sparsebase/reorder/rcm_reorder.h"
#include "sparsebase/utils/logger.h"

// Padded, zeroing allocator: DegreeReorder indexes one element past `mr`
// (reorder/degree_reorder.cc:41-45) and GrayReorder reads one element past
// sparse_v_order (gray_reorder.cc:239-240).  64 zero bytes of slack make the
// reference deterministic and give its intended result.
// Bound locally with -Wl,-Bsymbolic so only this library uses it.
void *operator new(std::size_t sz) {
  void *p = std::calloc(1, sz + 64);
  if (!p) throw std::bad_alloc();
  return p;
}
void *operator new[](std::size_t sz) {
  void *p = std::calloc(1, sz + 64);
  if (!p) throw std::bad_alloc();
  return p;
}
void operator delete(void *p) noexcept { std::free(p); }
void operator delete[](void *p) noexcept { std::free(p); }
void operator delete(void *p, std::size_t) noexcept { std::free(p); }
void operator delete[](void *p, std::size_t) noexcept { std::free(p); }

using namespace sparsebase;

namespace {

struct Quiet {
  Quiet() { utils::Logger::set_level(utils::LOG_LVL_NONE); }
} quiet_;

template <typename I, typename V>
void t_coo_sort(int64_t n, int64_t m, int64_t nnz, I *row, I *col, V *val) {
  // the constructor sorts the caller's arrays in place (format/coo.cc:147-155)
  format::COO<I, I, V> coo((I)n, (I)m, (I)nnz, row, col, val, format::kNotOwned);
}
template <typename I, typename V>
void t_csr_sort(int64_t n, int64_t m, I *rp, I *col, V *val) {
  format::CSR<I, I, V> csr((I)n, (I)m, rp, col, val, format::kNotOwned);
}
template <typename I, typename V>
void t_coo_to_csr(int64_t n, int64_t m, int64_t nnz, I *row, I *col, V *val, I *rp_out,
                  I *col_out, V *val_out) {
  context::CPUContext cpu;
  format::COO<I, I, V> coo((I)n, (I)m, (I)nnz, row, col, val, format::kNotOwned, true);
  auto *csr = coo.template Convert<format::CSR>(&cpu);
  memcpy(rp_out, csr->get_row_ptr(), (n + 1) * sizeof(I));
  memcpy(col_out, csr->get_col(), nnz * sizeof(I));
  if constexpr (!std::is_same_v<V, void>)
    if (val && val_out) memcpy(val_out, csr->get_vals(), nnz * sizeof(V));
  delete csr;
}
template <typename I, typename V>
void t_csr_to_coo(int64_t n, int64_t m, int64_t nnz, I *rp, I *col, V *val, I *row_out,
                  I *col_out, V *val_out) {
  context::CPUContext cpu;
  format::CSR<I, I, V> csr((I)n, (I)m, rp, col, val, format::kNotOwned, true);
  auto *coo = csr.template Convert<format::COO>(&cpu);
  memcpy(row_out, coo->get_row(), nnz * sizeof(I));
  memcpy(col_out, coo->get_col(), nnz * sizeof(I));
  if constexpr (!std::is_same_v<V, void>)
    if (val && val_out) memcpy(val_out, coo->get_vals(), nnz * sizeof(V));
  delete coo;
}
// square matrices only: the reference sizes col_ptr by the row count (converter_order_two.cc:32-33)
template <typename I, typename V, typename Src>
void csc_out(Src *src, int64_t n, int64_t nnz, bool has_val, I *cp_out, I *row_out, V *val_out) {
  context::CPUContext cpu;
  auto *csc = src->template Convert<format::CSC>(&cpu);
  memcpy(cp_out, csc->get_col_ptr(), (n + 1) * sizeof(I));
  memcpy(row_out, csc->get_row(), nnz * sizeof(I));
  if constexpr (!std::is_same_v<V, void>)
    if (has_val && val_out) memcpy(val_out, csc->get_vals(), nnz * sizeof(V));
  delete csc;
}
template <typename I, typename V>
void t_coo_to_csc(int64_t n, int64_t nnz, I *row, I *col, V *val, I *cp_out, I *row_out, V *val_out) {
  format::COO<I, I, V> coo((I)n, (I)n, (I)nnz, row, col, val, format::kNotOwned, true);
  csc_out<I, V>(&coo, n, nnz, val != nullptr, cp_out, row_out, val_out);
}
template <typename I, typename V>
void t_csr_to_csc(int64_t n, int64_t nnz, I *rp, I *col, V *val, I *cp_out, I *row_out, V *val_out) {
  format::CSR<I, I, V> csr((I)n, (I)n, rp, col, val, format::kNotOwned, true);
  csc_out<I, V>(&csr, n, nnz, val != nullptr, cp_out, row_out, val_out);
}
template <typename I, typename V>
void t_features(int64_t n, I *rp, I *col, int64_t *bandwidth, int64_t *profile, I *degrees, float *dist_f,
                double *dist_d) {
  context::CPUContext cpu;
  format::CSR<I, I, V> csr((I)n, (I)n, rp, col, nullptr, format::kNotOwned, true);
  feature::Bandwidth<I, I, V> bw;
  int *b = bw.GetBandwidth(&csr, {&cpu}, false);
  *bandwidth = *b;
  delete b;
  feature::Profile<I, I, V> pf;
  I *p = pf.GetProfile(&csr, {&cpu}, false);
  *profile = (int64_t)*p;
  delete p;
  feature::Degrees<I, I, V> dg;
  I *d = dg.GetDegrees(&csr, {&cpu}, false);
  memcpy(degrees, d, n * sizeof(I));
  delete[] d;
  feature::DegreeDistribution<I, I, V, float> df;
  float *f = df.GetDistribution(&csr, {&cpu}, false);
  memcpy(dist_f, f, n * sizeof(float));
  delete[] f;
  feature::DegreeDistribution<I, I, V, double> dd;
  double *g = dd.GetDistribution(&csr, {&cpu}, false);
  memcpy(dist_d, g, n * sizeof(double));
  delete[] g;
}
template <typename I, typename V>
void t_degree(int64_t n, int64_t m, I *rp, I *col, int ascending, I *inv) {
  context::CPUContext cpu;
  format::CSR<I, I, V> csr((I)n, (I)m, rp, col, nullptr, format::kNotOwned, true);
  reorder::DegreeReorder<I, I, V> r(ascending != 0);
  I *o = r.GetReorder(&csr, {&cpu}, false);
  memcpy(inv, o, n * sizeof(I));
  delete[] o;
}
template <typename I, typename V>
void t_rcm(int64_t n, I *rp, I *col, I *inv) {
  context::CPUContext cpu;
  format::CSR<I, I, V> csr((I)n, (I)n, rp, col, nullptr, format::kNotOwned, true);
  reorder::RCMReorder<I, I, V> r;
  I *o = r.GetReorder(&csr, {&cpu}, false);
  memcpy(inv, o, n * sizeof(I));
  delete[] o;
}
template <typename I, typename V>
void t_gray(int64_t n, int64_t m, I *rp, I *col, int res, int thr, int grp, I *inv) {
  context::CPUContext cpu;
  format::CSR<I, I, V> csr((I)n, (I)m, rp, col, nullptr, format::kNotOwned, true);
  reorder::GrayReorder<I, I, V> r((reorder::BitMapSize)res, thr, grp);
  I *o = r.GetReorder(&csr, {&cpu}, false);
  memcpy(inv, o, n * sizeof(I));
  delete[] o;
}
template <typename I, typename V>
void t_permute(int64_t n, int64_t m, I *rp, I *col, V *val, I *row_order, I *col_order,
               I *rp_out, I *col_out, V *val_out) {
  context::CPUContext cpu;
  format::CSR<I, I, V> csr((I)n, (I)m, rp, col, val, format::kNotOwned, true);
  permute::PermuteOrderTwo<I, I, V> p(row_order, col_order);
  auto *out = p.GetPermutation(&csr, {&cpu}, false);
  auto *ocsr = out->template As<format::CSR>();
  const int64_t nnz = rp[n];
  memcpy(rp_out, ocsr->get_row_ptr(), (n + 1) * sizeof(I));
  memcpy(col_out, ocsr->get_col(), nnz * sizeof(I));
  if constexpr (!std::is_same_v<V, void>)
    if (val && val_out) memcpy(val_out, ocsr->get_vals(), nnz * sizeof(V));
  // the permuted CSR does not own its arrays (permute_order_two.cc:76-77)
  I *a = ocsr->get_row_ptr();
  I *b = ocsr->get_col();
  V *c = ocsr->get_vals();
  delete ocsr;
  delete[] a;
  delete[] b;
  if constexpr (!std::is_same_v<V, void>) delete[] c;
}

// the real EdgeListReader on a file
template <typename I, typename V>
int t_edge_list_read(const char *path, int weighted, int dedup, int no_self, int undirected, int square, int64_t cap,
                     I *row, I *col, V *val, int64_t *dims) {
  io::EdgeListReader<I, I, V> reader(path, weighted != 0, dedup != 0, no_self != 0, undirected != 0, square != 0);
  format::COO<I, I, V> *coo = reader.ReadCOO();
  const int64_t nnz = coo->get_num_nnz();
  dims[0] = coo->get_dimensions()[0];
  dims[1] = coo->get_dimensions()[1];
  dims[2] = nnz;
  int rc = 0;
  if (nnz > cap) rc = -3;
  else {
    memcpy(row, coo->get_row(), nnz * sizeof(I));
    memcpy(col, coo->get_col(), nnz * sizeof(I));
    if constexpr (!std::is_same_v<V, void>)
      if (val && coo->get_vals()) memcpy(val, coo->get_vals(), nnz * sizeof(V));
  }
  delete coo;
  return rc;
}

// the real MTXReader on a file: COO arrays as its ReadCOO() returns them (constructor sort applied)
template <typename I, typename V>
int t_mtx_read(const char *path, int zero_index, int upper, int64_t cap, I *row, I *col, V *val, int64_t *dims) {
  io::MTXReader<I, I, V> reader(path, zero_index != 0, upper != 0);
  format::COO<I, I, V> *coo = reader.ReadCOO();
  const int64_t nnz = coo->get_num_nnz();
  dims[0] = coo->get_dimensions()[0];
  dims[1] = coo->get_dimensions()[1];
  dims[2] = nnz;
  int rc = 0;
  if (nnz > cap) rc = -3;
  else {
    memcpy(row, coo->get_row(), nnz * sizeof(I));
    memcpy(col, coo->get_col(), nnz * sizeof(I));
    if constexpr (!std::is_same_v<V, void>)
      if (val && coo->get_vals()) memcpy(val, coo->get_vals(), nnz * sizeof(V));
  }
  delete coo;
  return rc;
}
}  // namespace

// (it, vt) -> concrete tuple.  vt: 0 void, 1 int32, 2 uint32 (unsigned tuple), 3 float, 6 double
#define TUPLE_SWITCH(it, vt, F, ...)                                                     \
  do {                                                                                   \
    if ((it) == 0 && (vt) == 0) { F(int, void, __VA_ARGS__); }                           \
    else if ((it) == 0 && (vt) == 1) { F(int, int, __VA_ARGS__); }                       \
    else if ((it) == 0 && (vt) == 2) { F(unsigned, unsigned, __VA_ARGS__); }             \
    else if ((it) == 0 && (vt) == 3) { F(int, float, __VA_ARGS__); }                     \
    else if ((it) == 1 && (vt) == 0) { F(long long, void, __VA_ARGS__); }                \
    else if ((it) == 1 && (vt) == 6) { F(long long, double, __VA_ARGS__); }              \
    else return -2;                                                                      \
  } while (0)

extern "C" {

#define F_COO_SORT(I, V, ...) t_coo_sort<I, V>(n, m, nnz, (I *)row, (I *)col, (V *)val)
int ref_coo_sort(int it, int vt, int64_t n, int64_t m, int64_t nnz, void *row, void *col,
                 void *val) {
  TUPLE_SWITCH(it, vt, F_COO_SORT, 0);
  return 0;
}
#define F_CSR_SORT(I, V, ...) t_csr_sort<I, V>(n, m, (I *)rp, (I *)col, (V *)val)
int ref_csr_sort_rows(int it, int vt, int64_t n, int64_t m, void *rp, void *col, void *val) {
  TUPLE_SWITCH(it, vt, F_CSR_SORT, 0);
  return 0;
}
#define F_COO_CSR(I, V, ...)                                                          \
  t_coo_to_csr<I, V>(n, m, nnz, (I *)row, (I *)col, (V *)val, (I *)rp_out, (I *)col_out, \
                     (V *)val_out)
int ref_coo_to_csr(int it, int vt, int64_t n, int64_t m, int64_t nnz, void *row, void *col,
                   void *val, void *rp_out, void *col_out, void *val_out) {
  TUPLE_SWITCH(it, vt, F_COO_CSR, 0);
  return 0;
}
#define F_CSR_COO(I, V, ...)                                                          \
  t_csr_to_coo<I, V>(n, m, nnz, (I *)rp, (I *)col, (V *)val, (I *)row_out, (I *)col_out, \
                     (V *)val_out)
int ref_csr_to_coo(int it, int vt, int64_t n, int64_t m, int64_t nnz, void *rp, void *col,
                   void *val, void *row_out, void *col_out, void *val_out) {
  TUPLE_SWITCH(it, vt, F_CSR_COO, 0);
  return 0;
}
#define F_COO_CSC(I, V, ...) \
  t_coo_to_csc<I, V>(n, nnz, (I *)row, (I *)col, (V *)val, (I *)cp_out, (I *)row_out, (V *)val_out)
int ref_coo_to_csc(int it, int vt, int64_t n, int64_t nnz, void *row, void *col, void *val, void *cp_out,
                   void *row_out, void *val_out) {
  TUPLE_SWITCH(it, vt, F_COO_CSC, 0);
  return 0;
}
#define F_CSR_CSC(I, V, ...) \
  t_csr_to_csc<I, V>(n, nnz, (I *)rp, (I *)col, (V *)val, (I *)cp_out, (I *)row_out, (V *)val_out)
int ref_csr_to_csc(int it, int vt, int64_t n, int64_t nnz, void *rp, void *col, void *val, void *cp_out,
                   void *row_out, void *val_out) {
  TUPLE_SWITCH(it, vt, F_CSR_CSC, 0);
  return 0;
}
int ref_edge_list_read(int it, int vt, const char *path, int weighted, int dedup, int no_self, int undirected, int square,
                       int64_t cap, void *row, void *col, void *val, int64_t *dims) {
  try {
    if (it == 0 && vt == 0) return t_edge_list_read<int, void>(path, weighted, dedup, no_self, undirected, square, cap, (int *)row, (int *)col, (void *)nullptr, dims);
    if (it == 0 && vt == 3) return t_edge_list_read<int, float>(path, weighted, dedup, no_self, undirected, square, cap, (int *)row, (int *)col, (float *)val, dims);
    if (it == 0 && vt == 6) return t_edge_list_read<int, double>(path, weighted, dedup, no_self, undirected, square, cap, (int *)row, (int *)col, (double *)val, dims);
    if (it == 1 && vt == 0) return t_edge_list_read<long long, void>(path, weighted, dedup, no_self, undirected, square, cap, (long long *)row, (long long *)col, (void *)nullptr, dims);
  } catch (const std::exception &e) {
    return -4;
  }
  return -2;
}
int ref_mtx_read(int it, int vt, const char *path, int zero_index, int upper, int64_t cap, void *row, void *col,
                 void *val, int64_t *dims) {
  try {
    if (it == 0 && vt == 0) return t_mtx_read<int, void>(path, zero_index, upper, cap, (int *)row, (int *)col, (void *)nullptr, dims);
    if (it == 0 && vt == 1) return t_mtx_read<int, int>(path, zero_index, upper, cap, (int *)row, (int *)col, (int *)val, dims);
    if (it == 0 && vt == 3) return t_mtx_read<int, float>(path, zero_index, upper, cap, (int *)row, (int *)col, (float *)val, dims);
    if (it == 0 && vt == 6) return t_mtx_read<int, double>(path, zero_index, upper, cap, (int *)row, (int *)col, (double *)val, dims);
    if (it == 1 && vt == 6) return t_mtx_read<long long, double>(path, zero_index, upper, cap, (long long *)row, (long long *)col, (double *)val, dims);
  } catch (const std::exception &e) {
    return -4;
  }
  return -2;
}

// all four features of one square CSR (profile is returned as the reference's IDType value, widened)
int ref_features(int it, int64_t n, void *rp, void *col, int64_t *bandwidth, int64_t *profile, void *degrees,
                 float *dist_f, double *dist_d) {
  if (it == 0) t_features<int, int>(n, (int *)rp, (int *)col, bandwidth, profile, (int *)degrees, dist_f, dist_d);
  else t_features<long long, double>(n, (long long *)rp, (long long *)col, bandwidth, profile, (long long *)degrees,
                                     dist_f, dist_d);
  return 0;
}
#define F_DEGREE(I, V, ...) t_degree<I, V>(n, m, (I *)rp, (I *)col, ascending, (I *)inv)
int ref_degree_reorder(int it, int vt, int64_t n, int64_t m, void *rp, void *col, int ascending,
                       void *inv) {
  TUPLE_SWITCH(it, vt, F_DEGREE, 0);
  return 0;
}
#define F_RCM(I, V, ...) t_rcm<I, V>(n, (I *)rp, (I *)col, (I *)inv)
int ref_rcm_reorder(int it, int vt, int64_t n, void *rp, void *col, void *inv) {
  TUPLE_SWITCH(it, vt, F_RCM, 0);
  return 0;
}
#define F_GRAY(I, V, ...) t_gray<I, V>(n, m, (I *)rp, (I *)col, res, thr, grp, (I *)inv)
int ref_gray_reorder(int it, int vt, int64_t n, int64_t m, void *rp, void *col, int res, int thr,
                     int grp, void *inv) {
  TUPLE_SWITCH(it, vt, F_GRAY, 0);
  return 0;
}
#define F_PERMUTE(I, V, ...)                                                              \
  t_permute<I, V>(n, m, (I *)rp, (I *)col, (V *)val, (I *)row_order, (I *)col_order,      \
                  (I *)rp_out, (I *)col_out, (V *)val_out)
int ref_permute_csr(int it, int vt, int64_t n, int64_t m, void *rp, void *col, void *val,
                    void *row_order, void *col_order, void *rp_out, void *col_out,
                    void *val_out) {
  TUPLE_SWITCH(it, vt, F_PERMUTE, 0);
  return 0;
}

// SbFF binary container, <int,int,float> and Array<float> (io/binary_{reader,writer}_order_{one,two}.cc).
// The reference's COO reader takes nnz from dimensions[1] and its CSR writer stores dimensions[1]
// entries of col/vals, so callers keep nnz == column count where the reference must read or write.
int ref_sbff_write_coo(const char *path, int n, int m, int nnz, int *row, int *col, float *vals) {
  try {
    sparsebase::format::COO<int, int, float> coo(n, m, nnz, row, col, vals, sparsebase::format::kNotOwned, true);
    sparsebase::io::BinaryWriterOrderTwo<int, int, float>(path).WriteCOO(&coo);
  } catch (const std::exception &e) {
    return -4;
  }
  return 0;
}
int ref_sbff_write_csr(const char *path, int n, int m, int *row_ptr, int *col, float *vals) {
  try {
    sparsebase::format::CSR<int, int, float> csr(n, m, row_ptr, col, vals, sparsebase::format::kNotOwned, true);
    sparsebase::io::BinaryWriterOrderTwo<int, int, float>(path).WriteCSR(&csr);
  } catch (const std::exception &e) {
    return -4;
  }
  return 0;
}
int ref_sbff_read_coo(const char *path, int64_t cap, int *row, int *col, float *vals, int64_t *dims) {
  try {
    auto *coo = sparsebase::io::BinaryReaderOrderTwo<int, int, float>(path).ReadCOO();
    dims[0] = coo->get_dimensions()[0];
    dims[1] = coo->get_dimensions()[1];
    dims[2] = coo->get_num_nnz();
    dims[3] = coo->get_vals() != nullptr;
    if (dims[2] > cap) return -3;
    std::memcpy(row, coo->get_row(), dims[2] * sizeof(int));
    std::memcpy(col, coo->get_col(), dims[2] * sizeof(int));
    if (coo->get_vals()) std::memcpy(vals, coo->get_vals(), dims[2] * sizeof(float));
    delete coo;
  } catch (const std::exception &e) {
    return -4;
  }
  return 0;
}
int ref_sbff_read_csr(const char *path, int64_t cap_rows, int64_t cap, int *row_ptr, int *col, float *vals, int64_t *dims) {
  try {
    auto *csr = sparsebase::io::BinaryReaderOrderTwo<int, int, float>(path).ReadCSR();
    dims[0] = csr->get_dimensions()[0];
    dims[1] = csr->get_dimensions()[1];
    dims[2] = csr->get_num_nnz();
    dims[3] = csr->get_vals() != nullptr;
    if (dims[2] > cap || dims[0] > cap_rows) return -3;
    std::memcpy(row_ptr, csr->get_row_ptr(), (dims[0] + 1) * sizeof(int));
    std::memcpy(col, csr->get_col(), dims[2] * sizeof(int));
    if (csr->get_vals()) std::memcpy(vals, csr->get_vals(), dims[2] * sizeof(float));
    delete csr;
  } catch (const std::exception &e) {
    return -4;
  }
  return 0;
}
int ref_sbff_write_array(const char *path, int n, float *vals) {
  try {
    sparsebase::format::Array<float> arr(n, vals, sparsebase::format::kNotOwned);
    sparsebase::io::BinaryWriterOrderOne<float>(path).WriteArray(&arr);
  } catch (const std::exception &e) {
    return -4;
  }
  return 0;
}
int ref_sbff_read_array(const char *path, int64_t cap, float *vals, int64_t *n) {
  try {
    auto *arr = sparsebase::io::BinaryReaderOrderOne<float>(path).ReadArray();
    *n = arr->get_dimensions()[0];
    if (*n > cap) return -3;
    std::memcpy(vals, arr->get_vals(), *n * sizeof(float));
    delete arr;
  } catch (const std::exception &e) {
    return -4;
  }
  return 0;
}
// ---- the tuple with sizeof(IDType) != sizeof(NNZType) the reference pre-instantiates (CMakeLists.txt:15-17):
// <int ids, long long offsets, float values> — COO -> CSR -> COO, Permute2D, DegreeReorder, RCMReorder.  What the
// SBX_I32_N64 tests assume — the same results as <int, int, float> with the offsets widened — is pinned against this
// (tests/test_oracle.py::test_reference_mixed_width_tuple_equals_the_32_bit_one).
int ref_mixed_pipeline(int64_t n, int64_t m, int64_t nnz, int *row, int *col, float *val, int *row_order, int *col_order,
                       long long *rp_out, int *row_back, long long *prp_out, int *pcol_out, float *pval_out, int *deg_asc,
                       int *rcm_out /* may be NULL: RCM wants a symmetric pattern */) {
  typedef int I;
  typedef long long N;
  typedef float V;
  try {
    context::CPUContext cpu;
    format::COO<I, N, V> coo((I)n, (I)m, (N)nnz, row, col, val, format::kNotOwned, true);
    auto *csr = coo.Convert<format::CSR>(&cpu);
    memcpy(rp_out, csr->get_row_ptr(), (n + 1) * sizeof(N));
    auto *back = csr->Convert<format::COO>(&cpu);
    memcpy(row_back, back->get_row(), nnz * sizeof(I));
    delete back;
    permute::PermuteOrderTwo<I, N, V> p(row_order, col_order);
    auto *out = p.GetPermutation(csr, {&cpu}, false);
    auto *ocsr = out->As<format::CSR>();
    memcpy(prp_out, ocsr->get_row_ptr(), (n + 1) * sizeof(N));
    memcpy(pcol_out, ocsr->get_col(), nnz * sizeof(I));
    memcpy(pval_out, ocsr->get_vals(), nnz * sizeof(V));
    N *a = ocsr->get_row_ptr();
    I *b = ocsr->get_col();
    V *c = ocsr->get_vals();
    delete ocsr;
    delete[] a;
    delete[] b;
    delete[] c;
    reorder::DegreeReorder<I, N, V> d(true);
    I *o = d.GetReorder(csr, {&cpu}, false);
    memcpy(deg_asc, o, n * sizeof(I));
    delete[] o;
    if (rcm_out) {
      reorder::RCMReorder<I, N, V> r;
      I *q = r.GetReorder(csr, {&cpu}, false);
      memcpy(rcm_out, q, n * sizeof(I));
      delete[] q;
    }
    delete csr;
  } catch (const std::exception &e) {
    return -4;
  }
  return 0;
}
}  // extern "C"
