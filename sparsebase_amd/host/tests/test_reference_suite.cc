// GPU tests of the host layer: the hot-path cases of the reference's own suites,
// re-expressed against this library (same fixtures, same assertions), plus the device
// formats.  Sources of the cases (under /root/reference/tests/suites/sparsebase/):
//   converter/converter_order_two_tests.cc:9-48,205-252,303-351   COO<->CSR copy / move / self
//   converter/converter_order_two_tests.cc:49-160                 CSR->CSC direct, cached, multi-step
//   format/coo_tests.cc:77-115, format/csr_tests.cc:80-115        constructor sorts
//   permute/permute_order_two_tests.cc:27-91                      row / row+col / inverse
//   reorder/{degree,rcm,gray}_reorder_tests.cc, reorder_tests.cc:27-125, bases/reorder_base_tests.cc
//   converter/converter_order_two_cuda_tests.cu:11-49             host<->device round trips
//   feature/{bandwidth,profile,degrees,degree_distribution}_tests.cc  reorder-quality features
//   io/mtx_reader_tests.cc:47-260 (coordinate files of io/reader_data.inc)   Matrix Market ingest
//   io/edge_list_reader_tests.cc:7-80                               edge-list ingest
//   io/binary_{reader,writer}_order_{one,two}_tests.cc                SbFF binary containers
// Fixtures: functionality_common.inc:6-44, converter/common.inc:5-16, format/common.inc:4-12.
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <fstream>
#include <memory>
#include <numeric>

#include "minitest.h"
#include "sparsebase/sparsebase.h"

using namespace sparsebase;
typedef format::CSR<int, int, int> CSR3;
typedef format::COO<int, int, int> COO3;
typedef format::HIPCSR<int, int, int> DCSR3;
typedef format::HIPCOO<int, int, int> DCOO3;
typedef format::CSC<int, int, int> CSC3;
typedef format::HIPCSC<int, int, int> DCSC3;

// functionality_common.inc
static const int n = 3, nnz = 4;
static int row_ptr[n + 1] = {0, 2, 3, 4}, cols[nnz] = {1, 2, 0, 0}, rows[nnz] = {0, 0, 1, 2}, vals[nnz] = {1, 2, 3, 4};
static int r_reorder_vector[3] = {1, 2, 0}, r_row_ptr[n + 1] = {0, 1, 3, 4}, r_cols[nnz] = {0, 1, 2, 0},
           r_vals[nnz] = {4, 1, 2, 3};
static int c_reorder_vector[3] = {2, 0, 1};
static int rc_row_ptr[n + 1] = {0, 1, 3, 4}, rc_cols[nnz] = {2, 0, 1, 2}, rc_vals[nnz] = {4, 1, 2, 3};
static int inverse_perm_array[3] = {2, 0, 1}, perm_array[3] = {1, 2, 0};
static float original_array[3] = {0.0f, 0.1f, 0.2f}, reordered_array[3] = {0.1f, 0.2f, 0.0f};
// converter/common.inc
static const int cn = 12, cm = 9, cnnz = 7;
static int coo_row[7]{0, 0, 1, 3, 5, 10, 11}, coo_col[7]{0, 2, 1, 3, 3, 8, 7}, coo_vals[7]{3, 5, 7, 9, 15, 11, 13};
static int csr_row_ptr[13]{0, 2, 3, 3, 4, 4, 5, 5, 5, 5, 5, 6, 7}, csr_col[7]{0, 2, 1, 3, 3, 8, 7},
    csr_vals[7]{3, 5, 7, 9, 15, 11, 13};
static int csc_col_ptr[13]{0, 1, 2, 3, 5, 5, 5, 5, 6, 7, 7, 7, 7}, csc_row[7]{0, 1, 0, 3, 5, 11, 10},
    csc_vals[7]{3, 7, 5, 9, 15, 13, 11};

static context::CPUContext cpu_context;
static std::unique_ptr<context::HIPContext> hip_context;

template <typename A, typename B>
static bool same(const A *a, const B *b, size_t count) {
  for (size_t i = 0; i < count; i++)
    if (!(a[i] == (A)b[i])) return false;
  return true;
}
template <typename T>
static bool is_permutation_of_iota(const T *p, size_t count) {
  std::vector<char> seen(count, 0);
  for (size_t i = 0; i < count; i++) {
    if ((size_t)p[i] >= count || seen[p[i]]) return false;
    seen[p[i]] = 1;
  }
  return true;
}
template <typename T>
static std::vector<T> fetch(hip::Device &dev, const T *d, size_t count) {
  std::vector<T> h(count);
  if (count) dev.ToHost(h.data(), d, count * sizeof(T));
  return h;
}

// ------------------------------------------------------------------ converter_order_two_tests.cc
TEST(ConverterOrderTwo, COOToCSR) {
  COO3 coo(cn, cm, cnnz, coo_row, coo_col, coo_vals, format::kNotOwned);
  converter::ConverterOrderTwo<int, int, int> conv;
  auto *csr = conv.Convert<CSR3>(&coo, &cpu_context);  // copy
  EXPECT_TRUE(same(csr->get_row_ptr(), csr_row_ptr, cn + 1));
  EXPECT_TRUE(same(csr->get_col(), csr_col, cnnz));
  EXPECT_TRUE(same(csr->get_vals(), csr_vals, cnnz));
  EXPECT_NE(csr->get_col(), coo.get_col());
  EXPECT_NE(csr->get_vals(), coo.get_vals());
  delete csr;
  // move: col/vals are handed over by pointer (:236-240)
  int *r = new int[7], *c = new int[7], *v = new int[7];
  std::copy(coo_row, coo_row + 7, r);
  std::copy(coo_col, coo_col + 7, c);
  std::copy(coo_vals, coo_vals + 7, v);
  COO3 owned(cn, cm, cnnz, r, c, v, format::kOwned);
  auto *moved = conv.Convert<CSR3>(&owned, &cpu_context, true);
  EXPECT_TRUE(same(moved->get_row_ptr(), csr_row_ptr, cn + 1));
  EXPECT_EQ(moved->get_col(), c);
  EXPECT_EQ(moved->get_vals(), v);
  EXPECT_FALSE(owned.ColIsOwned());
  delete moved;
}

TEST(ConverterOrderTwo, CSRToCOO) {
  CSR3 csr(cn, cm, csr_row_ptr, csr_col, csr_vals, format::kNotOwned);
  auto *coo = csr.Convert<format::COO>(&cpu_context);  // member syntax (format_conversion.cc:18-19)
  EXPECT_TRUE(same(coo->get_row(), coo_row, cnnz));
  EXPECT_TRUE(same(coo->get_col(), coo_col, cnnz));
  EXPECT_TRUE(same(coo->get_vals(), coo_vals, cnnz));
  EXPECT_NE(coo->get_col(), csr.get_col());
  delete coo;
  int *rp = new int[13], *c = new int[7], *v = new int[7];
  std::copy(csr_row_ptr, csr_row_ptr + 13, rp);
  std::copy(csr_col, csr_col + 7, c);
  std::copy(csr_vals, csr_vals + 7, v);
  CSR3 owned(cn, cm, rp, c, v, format::kOwned);
  auto *moved = owned.Convert<format::COO>(&cpu_context, true);
  EXPECT_TRUE(same(moved->get_row(), coo_row, cnnz));
  EXPECT_EQ(moved->get_col(), c);
  EXPECT_EQ(moved->get_vals(), v);
  delete moved;
}

TEST(ConverterOrderTwo, SelfConversionReturnsTheSource) {
  CSR3 csr(cn, cm, csr_row_ptr, csr_col, csr_vals, format::kNotOwned);
  EXPECT_EQ(csr.Convert<format::CSR>(&cpu_context), &csr);  // :303-351
  COO3 coo(cn, cm, cnnz, coo_row, coo_col, coo_vals, format::kNotOwned);
  EXPECT_EQ(coo.Convert<format::COO>(&cpu_context), &coo);
}

TEST(ConverterOrderTwo, VoidValuesAndOtherTuples) {
  format::COO<int, int, void> coo(cn, cm, cnnz, coo_row, coo_col, nullptr, format::kNotOwned);
  auto *csr = coo.Convert<format::CSR>(&cpu_context);
  EXPECT_TRUE(same(csr->get_row_ptr(), csr_row_ptr, cn + 1));
  EXPECT_TRUE(same(csr->get_col(), csr_col, cnnz));
  EXPECT_EQ(csr->get_vals(), (void *)nullptr);
  delete csr;
  unsigned ur[7], uc[7];
  float fv[7];
  for (int i = 0; i < 7; i++) { ur[i] = coo_row[i]; uc[i] = coo_col[i]; fv[i] = (float)coo_vals[i]; }
  format::COO<unsigned, unsigned, float> ucoo(cn, cm, cnnz, ur, uc, fv, format::kNotOwned);
  auto *ucsr = ucoo.Convert<format::CSR>(&cpu_context);
  EXPECT_TRUE(same(ucsr->get_row_ptr(), csr_row_ptr, cn + 1));
  EXPECT_TRUE(same(ucsr->get_vals(), csr_vals, cnnz));
  delete ucsr;
}

// The tuples with sizeof(IDType) != sizeof(NNZType) the reference pre-instantiates (CMakeLists.txt:15-17): 32-bit ids,
// 64-bit offsets — <int, long long, V> and <unsigned int, unsigned long long, V> — through the whole path
// (SBX_I32_N64: row_ptr / col_ptr are 64-bit arrays, every id array 32-bit).
template <typename I, typename N>
static void mixed_tuple_path() {
  typedef float V;
  I r[7], c[7];
  V v[7];
  for (int i = 0; i < 7; i++) r[i] = (I)coo_row[i], c[i] = (I)coo_col[i], v[i] = (V)coo_vals[i];
  format::COO<I, N, V> coo(cn, cm, cnnz, r, c, v, format::kNotOwned);
  std::unique_ptr<format::CSR<I, N, V>> csr(coo.template Convert<format::CSR>(&cpu_context));
  EXPECT_TRUE(same(csr->get_row_ptr(), csr_row_ptr, cn + 1));
  EXPECT_TRUE(same(csr->get_col(), csr_col, cnnz) && same(csr->get_vals(), csr_vals, cnnz));
  std::unique_ptr<format::COO<I, N, V>> back(csr->template Convert<format::COO>(&cpu_context));
  EXPECT_TRUE(same(back->get_row(), coo_row, cnnz) && same(back->get_col(), coo_col, cnnz));
  std::unique_ptr<format::CSC<I, N, V>> csc(csr->template Convert<format::CSC>(&cpu_context));
  EXPECT_TRUE(same(csc->get_col_ptr(), csc_col_ptr, cn + 1) && same(csc->get_row(), csc_row, cnnz) &&
              same(csc->get_vals(), csc_vals, cnnz));
  // device formats, reorderers, permute: the 3 x 3 fixture of functionality_common.inc
  N rp3[4];
  I c3[4];
  V v3[4];
  for (int i = 0; i < 4; i++) rp3[i] = (N)row_ptr[i], c3[i] = (I)cols[i], v3[i] = (V)vals[i];
  format::CSR<I, N, V> small(n, n, rp3, c3, v3, format::kNotOwned);
  std::unique_ptr<format::HIPCSR<I, N, V>> dsmall(small.template Convert<format::HIPCSR>(hip_context.get()));
  auto &dev = hip::Device::Get(hip_context->device_id);
  EXPECT_TRUE(same(fetch(dev, dsmall->get_row_ptr(), n + 1).data(), row_ptr, n + 1));
  reorder::DegreeReorder<I, N, V> deg(true);
  std::unique_ptr<I[]> o_deg(deg.GetReorder(dsmall.get(), {hip_context.get()}, false));
  const int want_deg[3] = {2, 1, 0};
  EXPECT_TRUE(same(o_deg.get(), want_deg, n));
  reorder::RCMReorder<I, N, V> rcm;
  std::unique_ptr<I[]> o_rcm(rcm.GetReorder(&small, {hip_context.get()}, true));
  const int want_rcm[3] = {1, 2, 0};
  EXPECT_TRUE(same(o_rcm.get(), want_rcm, n));
  reorder::GrayReorder<I, N, V> gray(reorder::BitSize16, 100, 10);
  std::unique_ptr<I[]> o_gray(gray.GetReorder(dsmall.get(), {hip_context.get()}, false));
  const int want_gray[3] = {2, 0, 1};
  EXPECT_TRUE(same(o_gray.get(), want_gray, n));
  I ro[3], co[3];
  for (int i = 0; i < 3; i++) ro[i] = (I)r_reorder_vector[i], co[i] = (I)c_reorder_vector[i];
  permute::PermuteOrderTwo<I, N, V> perm(ro, co);
  std::unique_ptr<format::FormatOrderTwo<I, N, V>> out(perm.GetPermutation(dsmall.get(), {hip_context.get()}, false));
  auto *pc = out->template AsAbsolute<format::HIPCSR<I, N, V>>();
  EXPECT_TRUE(same(fetch(dev, pc->get_row_ptr(), n + 1).data(), rc_row_ptr, n + 1));
  EXPECT_TRUE(same(fetch(dev, pc->get_col(), nnz).data(), rc_cols, nnz) && same(fetch(dev, pc->get_vals(), nnz).data(), rc_vals, nnz));
  std::unique_ptr<format::FormatOrderTwo<I, N, V>> hout(perm.GetPermutation(&small, {hip_context.get()}, true));
  auto *hc = hout->template AsAbsolute<format::CSR<I, N, V>>();
  EXPECT_TRUE(same(hc->get_row_ptr(), rc_row_ptr, n + 1) && same(hc->get_col(), rc_cols, nnz));
  static_assert(sizeof(*hc->get_row_ptr()) == 8 && sizeof(*hc->get_col()) == 4, "64-bit offsets over 32-bit ids");
}
TEST(ConverterOrderTwo, MixedWidthTuples) {
  mixed_tuple_path<int, long long>();
  mixed_tuple_path<unsigned int, unsigned long long>();
}

// converter_order_two_tests.cc:49-160: compare_cscs checks n + 1 entries of col_ptr (common.inc:68)
static void expect_csc(format::Format *f) {
  auto *csc = f->AsAbsolute<CSC3>();
  EXPECT_EQ((int)csc->get_num_nnz(), cnnz);
  EXPECT_EQ((int)csc->get_dimensions()[0], cn);
  EXPECT_EQ((int)csc->get_dimensions()[1], cm);
  EXPECT_TRUE(same(csc->get_col_ptr(), csc_col_ptr, cn + 1));
  EXPECT_TRUE(same(csc->get_row(), csc_row, cnnz));
  EXPECT_TRUE(same(csc->get_vals(), csc_vals, cnnz));
}

TEST(ConverterOrderTwo, CSRToCSCMultipleContextsAndMultiStep) {
  CSR3 csr(cn, cm, csr_row_ptr, csr_col, csr_vals, format::kNotOwned);
  context::CPUContext cpu1, cpu2;
  converter::ConverterOrderTwo<int, int, int> conv;
  CSC3 correct(cn, cm, csc_col_ptr, csc_row, csc_vals, format::kNotOwned);  // the fixture is a valid CSC (sorted)
  EXPECT_EQ((int)correct.get_num_nnz(), cnnz);
  auto *a = conv.Convert<CSC3>(&csr, {&cpu1, &cpu2}, false);  // templated
  expect_csc(a);
  delete a;
  auto *b = conv.Convert(&csr, CSC3::get_id_static(), {&cpu1, &cpu2}, false);  // non-templated
  expect_csc(b);
  delete b;
  auto *c = csr.Convert<format::CSC>({&cpu1, &cpu2}, false);  // member
  expect_csc(c);
  delete c;
  // remove the direct CSR->CSC function: the chain CSR -> COO -> CSC must be found (:77-99)
  conv.ClearConversionFunctions(CSR3::get_id_static(), CSC3::get_id_static(), false);
  auto *d = conv.Convert<CSC3>(&csr, {&cpu1, &cpu2}, false);
  expect_csc(d);
  delete d;
  auto chain = conv.ConvertCached(&csr, CSC3::get_id_static(), {&cpu1}, false);  // :142-150
  EXPECT_EQ(chain.size(), (size_t)2);
  if (chain.size() == 2) {
    expect_csc(chain[1]);
    auto *mid = chain[0]->AsAbsolute<COO3>();
    EXPECT_TRUE(same(mid->get_row(), coo_row, cnnz));
    EXPECT_TRUE(same(mid->get_col(), coo_col, cnnz));
  }
  for (auto *f : chain) delete f;
}

TEST(ConverterOrderTwo, CSRToCSCCached) {  // :101-133
  CSR3 csr(cn, cm, csr_row_ptr, csr_col, csr_vals, format::kNotOwned);
  converter::ConverterOrderTwo<int, int, int> conv;
  auto out = conv.ConvertCached(&csr, CSC3::get_id_static(), {&cpu_context}, false);
  EXPECT_EQ(out.size(), (size_t)1);
  if (!out.empty()) expect_csc(out[0]);
  for (auto *f : out) delete f;
}

TEST(ConverterOrderTwo, COOToCSCAndUnsortedColumns) {
  COO3 coo(cn, cm, cnnz, coo_row, coo_col, coo_vals, format::kNotOwned);
  auto *csc = coo.Convert<format::CSC>(&cpu_context);
  expect_csc(csc);
  delete csc;
  // the CSC constructor sorts every column's (row, value) pairs when one is out of order (csc.cc:99-157)
  int cp[4] = {0, 2, 3, 4}, r[4] = {2, 0, 1, 0}, v[4] = {7, 8, 9, 10};
  format::CSC<int, int, int> fixed(3, 3, cp, r, v, format::kNotOwned);
  const int want_r[4] = {0, 2, 1, 0}, want_v[4] = {8, 7, 9, 10};
  EXPECT_TRUE(same(r, want_r, 4));
  EXPECT_TRUE(same(v, want_v, 4));
  // rectangular the other way (m > n), where the reference overflows: col_ptr has m + 1 entries
  int wr[3] = {0, 1, 1}, wc[3] = {4, 0, 4}, wv[3] = {1, 2, 3};
  COO3 wide(2, 5, 3, wr, wc, wv, format::kNotOwned);
  auto *wcsc = wide.Convert<format::CSC>(&cpu_context);
  const int want_cp[6] = {0, 1, 1, 1, 1, 3}, want_row[3] = {1, 0, 1}, want_val[3] = {2, 1, 3};
  EXPECT_EQ((int)wcsc->get_num_nnz(), 3);
  EXPECT_TRUE(same(wcsc->get_col_ptr(), want_cp, 6));
  EXPECT_TRUE(same(wcsc->get_row(), want_row, 3));
  EXPECT_TRUE(same(wcsc->get_vals(), want_val, 3));
  delete wcsc;
}

TEST(HIPFormats, CSCOnDevice) {
  CSR3 csr(cn, cm, csr_row_ptr, csr_col, csr_vals, format::kNotOwned);
  auto *dcsr = csr.Convert<format::HIPCSR>(hip_context.get());
  auto *dcsc = dcsr->Convert<format::HIPCSC>(hip_context.get());  // sbx_csr_to_csc in HBM
  auto &dev = dcsc->device();
  EXPECT_TRUE(same(fetch(dev, dcsc->get_col_ptr(), cn + 1).data(), csc_col_ptr, cn + 1));
  EXPECT_TRUE(same(fetch(dev, dcsc->get_row(), cnnz).data(), csc_row, cnnz));
  EXPECT_TRUE(same(fetch(dev, dcsc->get_vals(), cnnz).data(), csc_vals, cnnz));
  auto *back = dcsc->Convert<format::CSC>(&cpu_context);  // D2H
  expect_csc(back);
  delete back;
  auto *copy = static_cast<DCSC3 *>(dcsc->Clone());
  EXPECT_TRUE(same(fetch(dev, copy->get_row(), cnnz).data(), csc_row, cnnz));
  delete copy;
  // host CSR straight to a device CSC: CSR -> HIPCSR -> HIPCSC (or CSR -> CSC -> HIPCSC), either chain is two hops
  auto *direct = csr.Convert<format::HIPCSC>(hip_context.get());
  EXPECT_TRUE(same(fetch(dev, direct->get_row(), cnnz).data(), csc_row, cnnz));
  delete direct;
  COO3 coo(cn, cm, cnnz, coo_row, coo_col, coo_vals, format::kNotOwned);
  auto *dcoo = coo.Convert<format::HIPCOO>(hip_context.get());
  auto *dcsc2 = dcoo->Convert<format::HIPCSC>(hip_context.get());
  EXPECT_TRUE(same(fetch(dev, dcsc2->get_col_ptr(), cn + 1).data(), csc_col_ptr, cn + 1));
  EXPECT_TRUE(same(fetch(dev, dcsc2->get_vals(), cnnz).data(), csc_vals, cnnz));
  delete dcsc2;
  delete dcoo;
  delete dcsc;
  delete dcsr;
}

// ------------------------------------------------------------------ io/mtx_reader_tests.cc (coordinate files), reader_data.inc
static std::string write_tmp(const std::string &name, const std::string &text) {
  const std::string path = std::string("/tmp/sbx_test_") + std::to_string((long)getpid()) + "_" + name;
  std::ofstream f(path, std::ios::binary);
  f << text;
  return path;
}
static int m_row_ptr[6]{0, 0, 1, 2, 3, 5}, m_row[5]{1, 2, 3, 4, 4}, m_col[5]{0, 1, 0, 2, 3};
static float m_vals[5]{0.1f, 0.3f, 0.2f, 0.4f, 0.5f};
static int m_row_ptr_symm[6]{0, 3, 5, 7, 9, 11}, m_row_symm[11]{0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4},
    m_col_symm[11]{0, 1, 3, 0, 2, 1, 4, 0, 4, 2, 3};
static float m_vals_symm[11]{0.7f, 0.1f, 0.2f, 0.1f, 0.3f, 0.3f, 0.4f, 0.2f, 0.5f, 0.4f, 0.5f};
static int m_row_skew[10]{0, 0, 1, 1, 2, 2, 3, 3, 4, 4}, m_col_skew[10]{1, 3, 0, 2, 1, 4, 0, 4, 2, 3};
static float m_vals_skew[10]{-0.1f, -0.2f, 0.1f, -0.3f, 0.3f, -0.4f, 0.2f, -0.5f, 0.4f, 0.5f};

TEST(MTXReader, CoordinateFilesOfTheReferenceSuite) {
  const std::string general = write_tmp("g.mtx", "%%MatrixMarket matrix coordinate pattern general\n%This is a comment\n5 5 5\n2 1\n4 1\n3 2\n5 3\n5 4\n");
  const std::string general_v = write_tmp("gv.mtx", "%%MatrixMarket matrix coordinate real general\n%This is a comment\n5 5 5\n2 1 0.1\n4 1 0.2\n3 2 0.3\n5 3 0.4\n5 4 0.5\n");
  const std::string symm_v = write_tmp("sv.mtx", "%%MatrixMarket matrix coordinate real symmetric\n%This is a comment\n5 5 6\n1 1 0.7\n2 1 0.1\n4 1 0.2\n3 2 0.3\n5 3 0.4\n5 4 0.5\n");
  const std::string skew_v = write_tmp("kv.mtx", "%%MatrixMarket matrix coordinate real skew-symmetric\n%This is a comment\n5 5 5\n2 1 0.1\n4 1 0.2\n3 2 0.3\n5 3 0.4\n5 4 0.5\n");
  {  // BasicsGeneral (:58-108)
    io::MTXReader<int, int, int> reader(general);
    auto *coo = reader.ReadCOO();
    EXPECT_EQ((int)coo->get_num_nnz(), 5);
    EXPECT_TRUE(same(coo->get_row(), m_row, 5));
    EXPECT_TRUE(same(coo->get_col(), m_col, 5));
    EXPECT_EQ(coo->get_vals(), (int *)nullptr);
    delete coo;
    auto *csr = reader.ReadCSR();
    EXPECT_TRUE(same(csr->get_row_ptr(), m_row_ptr, 6));
    EXPECT_TRUE(same(csr->get_col(), m_col, 5));
    delete csr;
    io::MTXReader<int, int, float> reader_v(general_v);
    auto *coo_v = reader_v.ReadCOO();
    EXPECT_TRUE(same(coo_v->get_row(), m_row, 5));
    EXPECT_TRUE(same(coo_v->get_vals(), m_vals, 5));
    delete coo_v;
    auto *csr_v = bases::IOBase::ReadMTXToCSR<int, int, float>(general_v, true);
    EXPECT_TRUE(same(csr_v->get_row_ptr(), m_row_ptr, 6));
    EXPECT_TRUE(same(csr_v->get_vals(), m_vals, 5));
    delete csr_v;
    io::MTXReader<int, int, double> one_based(general_v, false);  // convert_to_zero_index = false
    auto *coo1 = one_based.ReadCOO();
    EXPECT_EQ(coo1->get_row()[0], 2);
    EXPECT_EQ(coo1->get_vals()[0], 0.1);
    delete coo1;
  }
  {  // BasicsSymmetric (:162-213)
    io::MTXReader<int, int, float> reader(symm_v);
    auto *coo = reader.ReadCOO();
    EXPECT_EQ((int)coo->get_num_nnz(), 11);
    EXPECT_TRUE(same(coo->get_row(), m_row_symm, 11));
    EXPECT_TRUE(same(coo->get_col(), m_col_symm, 11));
    EXPECT_TRUE(same(coo->get_vals(), m_vals_symm, 11));
    delete coo;
    auto *csr = reader.ReadCSR();
    EXPECT_TRUE(same(csr->get_row_ptr(), m_row_ptr_symm, 6));
    EXPECT_TRUE(same(csr->get_vals(), m_vals_symm, 11));
    delete csr;
    io::MTXReader<int, int, float> upper(symm_v, true, true);  // upper triangle only: 6 entries, (min, max)
    auto *ut = upper.ReadCOO();
    EXPECT_EQ((int)ut->get_num_nnz(), 6);
    for (int i = 0; i < 6; i++) EXPECT_TRUE(ut->get_row()[i] <= ut->get_col()[i]);
    delete ut;
    std::unique_ptr<format::HIPCOO<int, int, float>> d(reader.ReadHIPCOO(*hip_context));  // stays in HBM
    EXPECT_TRUE(same(fetch(d->device(), d->get_col(), 11).data(), m_col_symm, 11));
  }
  {  // BasicsSkewSymmetric (:215-260)
    io::MTXReader<int, int, float> reader(skew_v);
    auto *coo = reader.ReadCOO();
    EXPECT_EQ((int)coo->get_num_nnz(), 10);
    EXPECT_TRUE(same(coo->get_row(), m_row_skew, 10));
    EXPECT_TRUE(same(coo->get_col(), m_col_skew, 10));
    EXPECT_TRUE(same(coo->get_vals(), m_vals_skew, 10));
    delete coo;
  }
  // ReadingWeightedIntoVoidValues (:47-56) and the other header errors of ParseHeader
  EXPECT_THROW((io::MTXReader<int, int, void>(general_v)), utils::ReaderException);
  EXPECT_THROW((io::MTXReader<int, int, int>("/nonexistent/file.mtx")), utils::ReaderException);
  const std::string bad = write_tmp("bad.mtx", "%%MatrixMarket matrix coordinate real hermitian\n1 1 1\n1 1 1.0\n");
  EXPECT_THROW((io::MTXReader<int, int, float>(bad)), utils::ReaderException);
  const std::string garbage = write_tmp("garbage.mtx", "%%MatrixMarket matrix coordinate real general\n2 2 2\n1 1 1.0\n2 x 3\n");
  io::MTXReader<int, int, float> greader(garbage);
  EXPECT_THROW(greader.ReadCOO(), utils::ReaderException);
  for (const auto &f : {general, general_v, symm_v, skew_v, bad, garbage}) std::remove(f.c_str());
}

TEST(EdgeListReader, Basics) {  // io/edge_list_reader_tests.cc:7-80, fixtures reader_data.inc:388-400
  const std::string edges = write_tmp("e.edges", "1 0\n3 0\n2 1\n4 2\n4 3\n");
  const std::string edges_v = write_tmp("ev.edges", "1 0 0.1\n3 0 0.2\n2 1 0.3\n4 2 0.4\n4 3 0.5\n");
  io::EdgeListReader<int, int, int> reader1(edges);  // undirected: every edge twice
  auto *coo = reader1.ReadCOO();
  EXPECT_EQ((int)coo->get_num_nnz(), 10);
  EXPECT_EQ((int)coo->get_dimensions()[0], 5);
  EXPECT_EQ((int)coo->get_dimensions()[1], 5);
  for (int i = 0; i < 10; i++) {
    bool found = false;
    for (int k = 0; k < 5; k++)
      found |= (coo->get_row()[i] == m_row[k] && coo->get_col()[i] == m_col[k]) ||
               (coo->get_row()[i] == m_col[k] && coo->get_col()[i] == m_row[k]);
    EXPECT_TRUE(found);
    if (i) EXPECT_TRUE(coo->get_row()[i - 1] < coo->get_row()[i] ||
                       (coo->get_row()[i - 1] == coo->get_row()[i] && coo->get_col()[i - 1] < coo->get_col()[i]));
  }
  delete coo;
  io::EdgeListReader<int, int, int> reader2(edges, false, false, false, false);  // directed
  auto *coo2 = reader2.ReadCOO();
  EXPECT_EQ((int)coo2->get_num_nnz(), 5);
  EXPECT_EQ((int)coo2->get_dimensions()[1], 4);  // m = max(v) + 1 when neither square nor undirected
  EXPECT_TRUE(same(coo2->get_row(), m_row, 5));
  EXPECT_TRUE(same(coo2->get_col(), m_col, 5));
  delete coo2;
  io::EdgeListReader<int, int, float> reader3(edges_v, true, false, false, false);  // weighted, directed
  auto *coo3 = reader3.ReadCOO();
  EXPECT_EQ((int)coo3->get_num_nnz(), 5);
  EXPECT_TRUE(same(coo3->get_row(), m_row, 5));
  EXPECT_TRUE(same(coo3->get_col(), m_col, 5));
  EXPECT_TRUE(same(coo3->get_vals(), m_vals, 5));
  delete coo3;
  auto *csr = bases::IOBase::ReadEdgeListToCSR<int, int, float>(edges_v, true, false, false, true);  // square
  EXPECT_TRUE(same(csr->get_row_ptr(), m_row_ptr, 6));
  EXPECT_TRUE(same(csr->get_vals(), m_vals, 5));
  EXPECT_EQ((int)csr->get_dimensions()[1], 5);
  delete csr;
  EXPECT_THROW((io::EdgeListReader<int, int, void>(edges_v, true).ReadCOO()), utils::ReaderException);
  EXPECT_THROW((io::EdgeListReader<int, int, int>("/nonexistent.edges").ReadCOO()), utils::ReaderException);
  std::remove(edges.c_str());
  std::remove(edges_v.c_str());
}

// ------------------------------------------------------------------ io/binary_{reader,writer}_order_{one,two}_tests.cc
TEST(BinaryOrderTwo, COO) {  // binary_reader_order_two_tests.cc:7-36, binary_writer_order_two_tests.cc:5-34
  int row[4]{1, 2, 3, 4}, col[4]{5, 6, 7, 8};
  float fv[4]{0.1f, 0.2f, 0.3f, 0.4f};
  format::COO<int, int, float> coo(4, 4, 4, row, col, fv, format::kNotOwned);
  const std::string path = write_tmp("coo.bin", "");
  io::BinaryWriterOrderTwo<int, int, float>(path).WriteCOO(&coo);
  std::unique_ptr<format::COO<int, int, float>> coo2(io::BinaryReaderOrderTwo<int, int, float>(path).ReadCOO());
  EXPECT_TRUE(coo.get_dimensions() == coo2->get_dimensions());
  EXPECT_EQ(coo.get_num_nnz(), coo2->get_num_nnz());
  EXPECT_TRUE(same(coo2->get_row(), row, 4) && same(coo2->get_col(), col, 4) && same(coo2->get_vals(), fv, 4));
  // straight to the device; nnz is the length of `row`, not the column count
  int row6[6]{2, 0, 1, 0, 2, 1}, col6[6]{1, 2, 0, 0, 2, 1};
  float fv6[6]{1, 2, 3, 4, 5, 6};
  format::COO<int, int, float> wide(3, 3, 6, row6, col6, fv6, format::kNotOwned, true);  // written unsorted
  bases::IOBase::WriteCOOToBinary(&wide, path);
  std::unique_ptr<format::HIPCOO<int, int, float>> d(io::BinaryReaderOrderTwo<int, int, float>(path).ReadHIPCOO(*hip_context));
  EXPECT_EQ((int)d->get_num_nnz(), 6);
  std::unique_ptr<format::COO<int, int, float>> back(d->Convert<format::COO>(&cpu_context));
  const int srow[6]{0, 0, 1, 1, 2, 2}, scol[6]{0, 2, 0, 1, 1, 2};  // the constructor sorted it on the device
  const float sval[6]{4, 2, 3, 6, 1, 5};
  EXPECT_TRUE(same(back->get_row(), srow, 6) && same(back->get_col(), scol, 6) && same(back->get_vals(), sval, 6));
  std::unique_ptr<format::COO<int, int, float>> host(bases::IOBase::ReadBinaryToCOO<int, int, float>(path));
  EXPECT_TRUE(same(host->get_row(), srow, 6) && same(host->get_col(), scol, 6) && same(host->get_vals(), sval, 6));
  // pattern files; values cannot go into ValueType void (binary_reader_order_two.cc:61-68)
  format::COO<int, int, void> pattern(3, 3, 6, row6, col6, nullptr, format::kNotOwned, true);
  io::BinaryWriterOrderTwo<int, int, void>(path).WriteCOO(&pattern);
  std::unique_ptr<format::COO<int, int, void>> p2(io::BinaryReaderOrderTwo<int, int, void>(path).ReadCOO());
  EXPECT_TRUE(same(p2->get_row(), srow, 6) && same(p2->get_col(), scol, 6) && p2->get_vals() == nullptr);
  std::unique_ptr<format::COO<int, int, float>> p3(io::BinaryReaderOrderTwo<int, int, float>(path).ReadCOO());
  EXPECT_TRUE(p3->get_vals() == nullptr);
  bases::IOBase::WriteCOOToBinary(&wide, path);
  EXPECT_THROW((io::BinaryReaderOrderTwo<int, int, void>(path).ReadCOO()), utils::ReaderException);
  EXPECT_THROW((io::BinaryReaderOrderTwo<int, int, float>(path).ReadCSR()), utils::ReaderException);  // not a CSR file
  EXPECT_THROW((io::BinaryReaderOrderTwo<int, int, double>(path).ReadCOO()), utils::ReaderException);  // element size
  EXPECT_THROW((io::BinaryReaderOrderTwo<long long, long long, float>(path).ReadCOO()), utils::ReaderException);
  std::remove(path.c_str());
}

TEST(BinaryOrderTwo, CSR) {  // binary_reader_order_two_tests.cc:38-70, binary_writer_order_two_tests.cc:36-70
  int rp[5]{0, 2, 3, 3, 4}, col[4]{0, 2, 1, 3};
  float fv[4]{0.1f, 0.2f, 0.3f, 0.4f};
  format::CSR<int, int, float> csr(4, 4, rp, col, fv, format::kNotOwned);
  const std::string path = write_tmp("csr.bin", "");
  io::BinaryWriterOrderTwo<int, int, float>(path).WriteCSR(&csr);
  std::unique_ptr<format::CSR<int, int, float>> csr2(io::BinaryReaderOrderTwo<int, int, float>(path).ReadCSR());
  EXPECT_TRUE(csr.get_dimensions() == csr2->get_dimensions());
  EXPECT_EQ(csr.get_num_nnz(), csr2->get_num_nnz());
  EXPECT_TRUE(same(csr2->get_row_ptr(), rp, 5) && same(csr2->get_col(), col, 4) && same(csr2->get_vals(), fv, 4));
  // more nonzeros than columns (outside what the reference's writer can store), rows written unsorted
  int rp9[4]{0, 3, 5, 9}, col9[9]{2, 0, 1, 2, 0, 3, 1, 0, 2};
  int iv9[9]{1, 2, 3, 4, 5, 6, 7, 8, 9};
  format::CSR<int, int, int> wide(3, 4, rp9, col9, iv9, format::kNotOwned, true);
  bases::IOBase::WriteCSRToBinary(&wide, path);
  const int scol[9]{0, 1, 2, 0, 2, 0, 1, 2, 3}, sval[9]{2, 3, 1, 5, 4, 8, 7, 9, 6};
  std::unique_ptr<format::CSR<int, int, int>> host(bases::IOBase::ReadBinaryToCSR<int, int, int>(path));
  EXPECT_EQ((int)host->get_num_nnz(), 9);
  EXPECT_TRUE(same(host->get_row_ptr(), rp9, 4) && same(host->get_col(), scol, 9) && same(host->get_vals(), sval, 9));
  std::unique_ptr<format::HIPCSR<int, int, int>> d(io::BinaryReaderOrderTwo<int, int, int>(path).ReadHIPCSR(*hip_context));
  std::unique_ptr<CSR3> back(d->Convert<format::CSR>(&cpu_context));
  EXPECT_TRUE(same(back->get_row_ptr(), rp9, 4) && same(back->get_col(), scol, 9) && same(back->get_vals(), sval, 9));
  // the file feeds the path directly: read to the device, reorder there
  reorder::DegreeReorder<int, int, int> by_degree(true);
  std::unique_ptr<int[]> order(by_degree.GetReorder(d.get(), {hip_context.get()}, false));
  EXPECT_TRUE(is_permutation_of_iota(order.get(), 3));
  EXPECT_THROW((io::BinaryReaderOrderTwo<int, int, int>(path).ReadCOO()), utils::ReaderException);
  EXPECT_THROW((io::BinaryReaderOrderTwo<int, int, void>(path).ReadCSR()), utils::ReaderException);
  EXPECT_THROW((io::BinaryReaderOrderTwo<int, int, double>(path).ReadCSR()), utils::ReaderException);  // element size
  std::remove(path.c_str());
}

TEST(BinaryOrderOne, Array) {  // binary_reader_order_one_tests.cc:6-28
  int array[5]{1, 2, 3, 4, 5};
  format::Array<int> sb_array(5, array, format::kNotOwned);
  const std::string path = write_tmp("arr.bin", "");
  io::BinaryWriterOrderOne<int>(path).WriteArray(&sb_array);
  std::unique_ptr<format::Array<int>> a2(io::BinaryReaderOrderOne<int>(path).ReadArray());
  EXPECT_EQ((int)a2->get_dimensions()[0], 5);
  EXPECT_TRUE(same(a2->get_vals(), array, 5));
  std::unique_ptr<format::HIPArray<int>> d(io::BinaryReaderOrderOne<int>(path).ReadHIPArray(*hip_context));
  std::unique_ptr<format::Array<int>> back(d->Convert<format::Array>(&cpu_context));
  EXPECT_TRUE(same(back->get_vals(), array, 5));
  std::unique_ptr<format::Array<int>> a3(bases::IOBase::ReadBinaryToArray<int>(path));
  EXPECT_TRUE(same(a3->get_vals(), array, 5));
  EXPECT_THROW((io::BinaryReaderOrderOne<double>(path).ReadArray()), utils::ReaderException);    // element size
  EXPECT_THROW((io::BinaryReaderOrderOne<unsigned>(path).ReadArray()), utils::ReaderException);  // signed file
  EXPECT_THROW((io::BinaryReaderOrderTwo<int, int, int>(path).ReadCOO()), utils::ReaderException);
  std::remove(path.c_str());
}

// ------------------------------------------------------------------ feature/{bandwidth,profile,degrees,degree_distribution}_tests.cc
TEST(Features, BandwidthProfileDegreesDistribution) {
  const int fn = 7, fnnz = 12;  // bandwidth_tests.cc:33-48, profile_tests.cc:33-48
  int frp[fn + 1] = {0, 2, 2, 5, 7, 9, 11, 12}, fcol[fnnz] = {2, 3, 0, 3, 4, 0, 2, 2, 5, 4, 6, 5};
  format::CSR<int, int, void> csr(fn, fn, frp, fcol, nullptr, format::kNotOwned);
  feature::Bandwidth<int, int, void> bw;
  EXPECT_EQ(bw.get_sub_ids().size(), (size_t)1);
  EXPECT_TRUE(bw.get_sub_ids()[0] == std::type_index(typeid(bw)));
  auto subs = bw.get_subs();
  EXPECT_EQ(subs.size(), (size_t)1);
  EXPECT_TRUE(std::type_index(typeid(*subs[0])) == std::type_index(typeid(bw)));
  EXPECT_NE(subs[0], (utils::Extractable *)&bw);
  delete subs[0];
  utils::Parameters p1;
  int *b = feature::Bandwidth<int, int, void>::GetBandwidthCSR({&csr}, &p1);
  EXPECT_EQ(*b, 4);
  delete b;
  for (bool convert : {true, false}) {
    b = bw.GetBandwidth(&csr, {&cpu_context}, convert);
    EXPECT_EQ(*b, 4);
    delete b;
  }
  auto fmap = bw.Extract(&csr, {&cpu_context}, true);
  EXPECT_EQ(fmap.size(), (size_t)1);
  EXPECT_EQ(*std::any_cast<int *>(fmap[bw.get_id()]), 4);
  delete std::any_cast<int *>(fmap[bw.get_id()]);
  feature::Profile<int, int, void> pf;
  int *p = pf.GetProfile(&csr, {&cpu_context}, true);
  EXPECT_EQ(*p, 9);
  delete p;
  auto cached = pf.GetProfileCached(&csr, {&cpu_context}, true);
  EXPECT_EQ(*std::get<1>(cached), 9);
  delete std::get<1>(cached);
  // degrees + distribution on the converter fixture; a COO input needs a conversion (degrees_tests.cc:75)
  CSR3 c12(cn, cm, csr_row_ptr, csr_col, csr_vals, format::kNotOwned);
  feature::Degrees<int, int, int> dg;
  int *deg = dg.GetDegrees(&c12, {&cpu_context}, true);
  for (int i = 0; i < cn; i++) EXPECT_EQ(deg[i], csr_row_ptr[i + 1] - csr_row_ptr[i]);
  delete[] deg;
  COO3 coo(cn, cm, cnnz, coo_row, coo_col, coo_vals, format::kNotOwned);
  EXPECT_THROW(dg.GetDegrees(&coo, {&cpu_context}, false), utils::DirectExecutionNotAvailableException<std::vector<std::type_index>>);
  deg = dg.GetDegrees(&coo, {&cpu_context}, true);
  for (int i = 0; i < cn; i++) EXPECT_EQ(deg[i], csr_row_ptr[i + 1] - csr_row_ptr[i]);
  delete[] deg;
  feature::DegreeDistribution<int, int, int, float> dd;
  float *dist = dd.GetDistribution(&c12, {&cpu_context}, true);
  for (int i = 0; i < cn; i++) EXPECT_EQ(dist[i], (csr_row_ptr[i + 1] - csr_row_ptr[i]) / (float)cnnz);
  delete[] dist;
  // device-resident input: the {HIPCSR} implementation, nothing staged
  auto *dcsr = csr.Convert<format::HIPCSR>(hip_context.get());
  b = bw.GetBandwidth(dcsr, {hip_context.get()}, false);
  EXPECT_EQ(*b, 4);
  delete b;
  p = pf.GetProfile(dcsr, {hip_context.get()}, false);
  EXPECT_EQ(*p, 9);
  delete p;
  delete dcsr;
  // what the features are for: RCM lowers the bandwidth of a shuffled band matrix
  {
    const int gn = 400;
    std::vector<int> perm(gn), grp(gn + 1, 0), gcol;
    std::iota(perm.begin(), perm.end(), 0);
    for (int i = gn - 1; i > 0; i--) std::swap(perm[i], perm[(i * 7919 + 13) % (i + 1)]);
    std::vector<std::vector<int>> adj(gn);
    for (int i = 0; i < gn; i++)
      for (int d = -2; d <= 2; d++)
        if (d != 0 && i + d >= 0 && i + d < gn) adj[perm[i]].push_back(perm[i + d]);
    for (int i = 0; i < gn; i++) {
      std::sort(adj[i].begin(), adj[i].end());
      gcol.insert(gcol.end(), adj[i].begin(), adj[i].end());
      grp[i + 1] = (int)gcol.size();
    }
    format::CSR<int, int, void> g(gn, gn, grp.data(), gcol.data(), nullptr, format::kNotOwned);
    int *before = bw.GetBandwidth(&g, {&cpu_context}, true);
    int *order = bases::ReorderBase::Reorder<reorder::RCMReorder>({}, &g, {&cpu_context}, true);
    auto *pg = bases::ReorderBase::Permute2D(order, &g, {&cpu_context}, true);
    int *after = bw.GetBandwidth(pg, {&cpu_context}, true);
    EXPECT_TRUE(*after <= 5);
    EXPECT_TRUE(*after < *before);
    delete before;
    delete after;
    delete[] order;
    delete pg;
  }
}

// ------------------------------------------------------------------ coo_tests.cc / csr_tests.cc (Sort)
TEST(COO, Sort) {
  const int want_row[4]{0, 0, 1, 3}, want_col[4]{0, 2, 1, 3}, want_vals[4]{4, 5, 7, 9};
  int r[4]{0, 0, 3, 1}, c[4]{2, 0, 3, 1}, v[4]{5, 4, 9, 7};
  COO3 coo(4, 4, 4, r, c, v, format::kNotOwned);
  EXPECT_TRUE(same(coo.get_row(), want_row, 4) && same(coo.get_col(), want_col, 4) && same(coo.get_vals(), want_vals, 4));
  EXPECT_TRUE(same(r, want_row, 4));  // sorted in place on the caller's arrays
  int r2[4]{0, 0, 3, 1}, c2[4]{2, 0, 3, 1}, v2[4]{5, 4, 9, 7};
  COO3 keep(4, 4, 4, r2, c2, v2, format::kNotOwned, true);  // ignore_sort
  const int orig_r[4]{0, 0, 3, 1}, orig_c[4]{2, 0, 3, 1};
  EXPECT_TRUE(same(keep.get_row(), orig_r, 4) && same(keep.get_col(), orig_c, 4));
  int r3[4]{0, 0, 3, 1}, c3[4]{2, 0, 3, 1};
  format::COO<int, int, void> pattern(4, 4, 4, r3, c3, nullptr, format::kNotOwned);
  EXPECT_EQ(pattern.get_vals(), (void *)nullptr);
  EXPECT_TRUE(same(pattern.get_row(), want_row, 4) && same(pattern.get_col(), want_col, 4));
}

TEST(CSR, Sort) {
  const int want_col[4]{0, 2, 1, 3}, want_vals[4]{4, 5, 7, 9};
  int rp[5]{0, 2, 3, 3, 4}, c[4]{2, 0, 1, 3}, v[4]{5, 4, 7, 9};
  CSR3 csr(4, 4, rp, c, v, format::kNotOwned);
  EXPECT_TRUE(same(csr.get_col(), want_col, 4) && same(csr.get_vals(), want_vals, 4));
  int c2[4]{2, 0, 1, 3}, v2[4]{5, 4, 7, 9};
  CSR3 keep(4, 4, rp, c2, v2, format::kNotOwned, true);
  const int orig_c[4]{2, 0, 1, 3};
  EXPECT_TRUE(same(keep.get_col(), orig_c, 4));
  int c3[4]{2, 0, 1, 3};
  format::CSR<int, int, void> pattern(4, 4, rp, c3, nullptr, format::kNotOwned);
  EXPECT_TRUE(same(pattern.get_col(), want_col, 4));
}

// ------------------------------------------------------------------ permute_order_two_tests.cc
TEST(PermuteOrderTwo, RowWise) {
  CSR3 global_csr(n, n, row_ptr, cols, vals, format::kNotOwned);
  COO3 global_coo(n, n, nnz, rows, cols, vals, format::kNotOwned);
  permute::PermuteOrderTwo<int, int, int> transformer(r_reorder_vector, nullptr);
  EXPECT_THROW(transformer.GetPermutation(&global_coo, {&cpu_context}, false),
               utils::DirectExecutionNotAvailableException<std::vector<std::type_index>>);
  auto *out = transformer.GetPermutation(&global_csr, {&cpu_context}, false)->As<format::CSR>();
  EXPECT_TRUE(same(out->get_row_ptr(), r_row_ptr, n + 1) && same(out->get_col(), r_cols, nnz) &&
              same(out->get_vals(), r_vals, nnz));
  delete out;
  // a COO input is converted when allowed
  auto *via = transformer.GetPermutation(&global_coo, {&cpu_context}, true)->As<format::CSR>();
  EXPECT_TRUE(same(via->get_row_ptr(), r_row_ptr, n + 1) && same(via->get_col(), r_cols, nnz));
  delete via;
}

TEST(PermuteOrderTwo, RowColWiseAndInverse) {
  CSR3 global_csr(n, n, row_ptr, cols, vals, format::kNotOwned);
  permute::PermuteOrderTwo<int, int, int> transformer(r_reorder_vector, c_reorder_vector);
  auto *perm = transformer.GetPermutation(&global_csr, {&cpu_context}, false)->As<format::CSR>();
  EXPECT_TRUE(same(perm->get_row_ptr(), rc_row_ptr, n + 1) && same(perm->get_col(), rc_cols, nnz) &&
              same(perm->get_vals(), rc_vals, nnz));
  auto *inv_r = bases::ReorderBase::InversePermutation(r_reorder_vector, n);
  auto *inv_c = bases::ReorderBase::InversePermutation(c_reorder_vector, n);
  permute::PermuteOrderTwo<int, int, int> back(inv_r, inv_c);
  auto *orig = back.GetPermutation(perm, {&cpu_context}, false)->As<format::CSR>();
  EXPECT_TRUE(same(orig->get_row_ptr(), row_ptr, n + 1) && same(orig->get_col(), cols, nnz) &&
              same(orig->get_vals(), vals, nnz));
  delete orig;
  delete perm;
  delete[] inv_r;
  delete[] inv_c;
}

// the sharded form through the C++ API: a world of one rank over RCCL, and two "ranks" of this process
// (on this box's one GPU) whose all-gather hook exchanges through host memory — both against the plain Permute2D
struct TwoRankHook {  // the two ranks run one after the other: what each contributed to exchange c is recorded in
  std::vector<char> sent[2][2];  // pass c and replayed from pass c + 1 on (an operator call makes two exchanges)
  size_t call = 0;
  int rank = 0, replay = 0;  // exchanges [0, replay) are real in this pass
};
static int two_rank_allgather(void *user, const void *send, void *recv, size_t bytes, void *) {
  auto *hk = static_cast<TwoRankHook *>(user);
  auto &dev = hip::Device::Get(hip::DefaultDevice());
  const size_t c = hk->call++;
  if ((int)c < hk->replay) {
    for (int r = 0; r < 2; r++) dev.ToDevice((char *)recv + bytes * r, hk->sent[r][c].data(), bytes);
    return SBX_OK;
  }
  hk->sent[hk->rank][c].resize(bytes);
  dev.ToHost(hk->sent[hk->rank][c].data(), send, bytes);
  std::vector<char> zeros(bytes, 0);  // the other rank's part is not known yet
  for (int r = 0; r < 2; r++)
    dev.ToDevice((char *)recv + bytes * r, r == hk->rank ? hk->sent[r][c].data() : zeros.data(), bytes);
  return SBX_OK;
}

TEST(PermuteOrderTwo, ShardedMatchesWhole) {
  CSR3 global_csr(n, n, row_ptr, cols, vals, format::kNotOwned);
  context::HIPContext gpu(hip::DefaultDevice());
  auto *dcsr = global_csr.Convert<format::HIPCSR>(&gpu);
  auto check_rows = [&](permute::ShardedHIPCSR<int, int, int> *sh, int rank) {
    auto &dev = hip::Device::Get(gpu.device_id);
    int *rp = dev.Download(sh->row_ptr->get_vals(), (size_t)n + 1);
    EXPECT_TRUE(same(rp, rc_row_ptr, n + 1));
    const int64_t a = rc_row_ptr[sh->row_begin], k = sh->local_nnz(rank);
    EXPECT_TRUE(sh->entry_offsets[rank] == a && k == rc_row_ptr[sh->row_end] - a);
    int *c = dev.Download(sh->col->get_vals(), (size_t)k), *v = dev.Download(sh->vals->get_vals(), (size_t)k);
    EXPECT_TRUE(same(c, rc_cols + a, (int)k) && same(v, rc_vals + a, (int)k));
    delete[] rp;
    delete[] c;
    delete[] v;
  };
  {  // one rank, RCCL
    context::HIPCommunicator comm(gpu, 0, 1, context::HIPCommunicator::NewId());
    permute::PermuteOrderTwo<int, int, int> t(r_reorder_vector, c_reorder_vector);
    auto *sh = t.GetPermutationSharded(dcsr, comm);
    EXPECT_TRUE(sh->row_begin == 0 && sh->row_end == n);
    check_rows(sh, 0);
    delete sh;
  }
  {  // two ranks played one after the other; pass 2 sees both exchanges (nnz totals, row_ptr segments) for real
    TwoRankHook hk;
    const int64_t splits[3] = {0, n / 3, n};
    for (int pass = 0; pass < 3; pass++) {
      hk.replay = pass;
      for (int rank = 0; rank < 2; rank++) {
        hk.rank = rank;
        hk.call = 0;
        context::HIPCommunicator comm(gpu, rank, 2, two_rank_allgather, &hk);
        auto *sh = bases::ReorderBase::Permute2DRowColumnWiseSharded(r_reorder_vector, c_reorder_vector, dcsr, comm, splits);
        if (pass == 2) check_rows(sh, rank);
        delete sh;
      }
    }
  }
  delete dcsr;
}

TEST(PermuteOrderOne, ArrayAndInversePermutation) {
  auto *inv = bases::ReorderBase::InversePermutation(perm_array, 3);
  EXPECT_TRUE(same(inv, inverse_perm_array, 3));
  delete[] inv;
  format::Array<float> arr(3, original_array, format::kNotOwned);
  auto *out = bases::ReorderBase::Permute1D(inverse_perm_array, &arr, {&cpu_context}, true)->As<format::Array>();
  EXPECT_TRUE(same(out->get_vals(), reordered_array, 3));
  delete out;
}

// ------------------------------------------------------------------ reorderers
template <typename I>
static void check_degree_ordering(I *order, I nrows, const I *rp, bool ascending) {  // functionality_common.inc:67-90
  EXPECT_TRUE(is_permutation_of_iota(order, nrows));
  std::vector<I> perm(nrows);
  for (I i = 0; i < nrows; i++) perm[order[i]] = i;
  for (I i = 0; i + 1 < nrows; i++) {
    const I a = rp[perm[i] + 1] - rp[perm[i]], b = rp[perm[i + 1] + 1] - rp[perm[i + 1]];
    EXPECT_TRUE(ascending ? a <= b : a >= b);
  }
}

TEST(Reorderers, DegreeRCMGrayOnTheFixture) {
  CSR3 global_csr(n, n, row_ptr, cols, vals, format::kNotOwned);
  COO3 global_coo(n, n, nnz, rows, cols, vals, format::kNotOwned);
  for (bool asc : {true, false}) {
    reorder::DegreeReorder<int, int, int> reorder(asc);
    auto *order = reorder.GetReorder(&global_csr, {&cpu_context}, true);
    check_degree_ordering(order, n, row_ptr, asc);
    const int want_asc[3] = {2, 1, 0}, want_desc[3] = {0, 1, 2};
    EXPECT_TRUE(same(order, asc ? want_asc : want_desc, 3));
    delete[] order;
    EXPECT_THROW(reorder.GetReorder(&global_coo, {&cpu_context}, false),
                 utils::DirectExecutionNotAvailableException<std::vector<std::type_index>>);
    auto *via = reorder.GetReorder(&global_coo, {&cpu_context}, true);
    check_degree_ordering(via, n, row_ptr, asc);
    delete[] via;
  }
  reorder::RCMReorder<int, int, int> rcm;
  auto *order = rcm.GetReorder(&global_csr, {&cpu_context}, true);
  const int want_rcm[3] = {1, 2, 0};
  EXPECT_TRUE(same(order, want_rcm, 3));
  delete[] order;
  reorder::GrayReorder<int, int, int> gray(reorder::BitSize16, 100, 10);
  order = gray.GetReorder(&global_csr, {&cpu_context}, true);
  const int want_gray[3] = {2, 0, 1};
  EXPECT_TRUE(same(order, want_gray, 3));
  delete[] order;
}

TEST(Reorderers, CachedReturnsTheConvertedInput) {  // reorder_tests.cc:27-125
  COO3 global_coo(n, n, nnz, rows, cols, vals, format::kNotOwned);
  CSR3 global_csr(n, n, row_ptr, cols, vals, format::kNotOwned);
  reorder::DegreeReorder<int, int, int> reorder(true);
  auto cached = reorder.GetReorderCached(&global_coo, {&cpu_context}, true);
  EXPECT_EQ(std::get<0>(cached).size(), (size_t)1);
  EXPECT_EQ(std::get<0>(cached)[0].size(), (size_t)1);
  auto *conv = std::get<0>(cached)[0][0]->AsAbsolute<CSR3>();
  EXPECT_TRUE(same(conv->get_row_ptr(), row_ptr, n + 1) && same(conv->get_col(), cols, nnz));
  delete conv;
  delete[] std::get<1>(cached);
  auto direct = reorder.GetReorderCached(&global_csr, {&cpu_context}, true);
  EXPECT_EQ(std::get<0>(direct)[0].size(), (size_t)0);
  delete[] std::get<1>(direct);
  // explicit params override the instance's (reorderer.h:62-72)
  reorder::DegreeReorderParams desc(false);
  auto *o = reorder.GetReorder(&global_csr, &desc, {&cpu_context}, true);
  check_degree_ordering(o, n, row_ptr, false);
  delete[] o;
}

TEST(ReorderBase, Facade) {  // bases/reorder_base_tests.cc:31-92
  CSR3 global_csr(n, n, row_ptr, cols, vals, format::kNotOwned);
  COO3 global_coo(n, n, nnz, rows, cols, vals, format::kNotOwned);
  auto *o = bases::ReorderBase::Reorder<reorder::RCMReorder>({}, &global_csr, {&cpu_context}, true);
  EXPECT_TRUE(is_permutation_of_iota(o, n));
  delete[] o;
  o = bases::ReorderBase::Reorder<reorder::DegreeReorder>({true}, &global_coo, {&cpu_context}, true);
  check_degree_ordering(o, n, row_ptr, true);
  delete[] o;
  o = bases::ReorderBase::Reorder<reorder::GrayReorder>({reorder::BitSize16, 10, 5}, &global_csr, {&cpu_context}, true);
  EXPECT_TRUE(is_permutation_of_iota(o, n));
  delete[] o;
  auto cached = bases::ReorderBase::ReorderCached<reorder::DegreeReorder>({true}, &global_coo, {&cpu_context});
  EXPECT_EQ(cached.first.size(), (size_t)1);
  delete cached.first[0];
  delete[] cached.second;
  auto *p = bases::ReorderBase::Permute2DRowWise<format::CSR>(r_reorder_vector, &global_csr, {&cpu_context}, true);
  EXPECT_TRUE(same(p->get_row_ptr(), r_row_ptr, n + 1) && same(p->get_col(), r_cols, nnz) && same(p->get_vals(), r_vals, nnz));
  delete p;
  auto *q = bases::ReorderBase::Permute2DRowColumnWise(r_reorder_vector, c_reorder_vector, &global_coo,
                                                       {&cpu_context}, true);
  EXPECT_TRUE(same(q->As<format::CSR>()->get_col(), rc_cols, nnz));
  delete q;
  EXPECT_THROW(bases::ReorderBase::Permute2D(r_reorder_vector, &global_coo, {&cpu_context}, false),
               utils::DirectExecutionNotAvailableException<std::vector<std::type_index>>);
  auto *as_coo = bases::ReorderBase::Permute2DRowWise<format::COO>(r_reorder_vector, &global_csr, {&cpu_context},
                                                                  true, true);
  const int want_rows[4] = {0, 1, 1, 2};
  EXPECT_TRUE(same(as_coo->get_row(), want_rows, nnz) && same(as_coo->get_col(), r_cols, nnz));
  delete as_coo;
}

// ------------------------------------------------------------------ device formats
TEST(HIPFormats, RoundTripsAndDeviceOperators) {  // converter_order_two_cuda_tests.cu:11-49
  CSR3 global_csr(n, n, row_ptr, cols, vals, format::kNotOwned);
  auto *dcsr = global_csr.Convert<format::HIPCSR>(hip_context.get());
  EXPECT_TRUE(dcsr->get_context()->IsEquivalent(hip_context.get()));
  auto &dev = dcsr->device();
  EXPECT_TRUE(same(fetch(dev, dcsr->get_row_ptr(), n + 1).data(), row_ptr, n + 1));
  auto *back = dcsr->Convert<format::CSR>(&cpu_context);
  EXPECT_TRUE(same(back->get_row_ptr(), row_ptr, n + 1) && same(back->get_col(), cols, nnz) && same(back->get_vals(), vals, nnz));
  delete back;
  // a host CSR cannot become an HIPCSR when only the CPU is offered
  EXPECT_THROW(global_csr.Convert<format::HIPCSR>(&cpu_context), utils::ConversionException);
  // device CSR -> device COO -> device CSR, copy and move, all in HBM
  auto *dcoo = dcsr->Convert<format::HIPCOO>(hip_context.get());
  EXPECT_TRUE(same(fetch(dev, dcoo->get_row(), nnz).data(), rows, nnz));
  EXPECT_NE(dcoo->get_col(), dcsr->get_col());
  auto *dcsr2 = dcoo->Convert<format::HIPCSR>(hip_context.get(), true);  // move: col pointer handed over
  int *moved_col = dcsr2->get_col();
  EXPECT_EQ(moved_col, dcoo->get_col());
  EXPECT_TRUE(same(fetch(dev, dcsr2->get_row_ptr(), n + 1).data(), row_ptr, n + 1));
  // COO on the host -> CSR on the device: a two-hop chain found by the graph search
  COO3 global_coo(n, n, nnz, rows, cols, vals, format::kNotOwned);
  auto *chain = global_coo.Convert<format::HIPCSR>(hip_context.get());
  EXPECT_TRUE(same(fetch(dev, chain->get_col(), nnz).data(), cols, nnz));
  delete chain;
  // operators on device-resident input
  reorder::RCMReorder<int, int, int> rcm;
  auto *o = rcm.GetReorder(dcsr, {hip_context.get()}, false);
  const int want_rcm[3] = {1, 2, 0};
  EXPECT_TRUE(same(o, want_rcm, 3));
  delete[] o;
  permute::PermuteOrderTwo<int, int, int> perm(r_reorder_vector, c_reorder_vector);
  auto *pd = perm.GetPermutation(dcsr, {hip_context.get()}, false)->As<format::HIPCSR>();
  EXPECT_TRUE(same(fetch(dev, pd->get_col(), nnz).data(), rc_cols, nnz));
  EXPECT_TRUE(same(fetch(dev, pd->get_vals(), nnz).data(), rc_vals, nnz));
  delete pd;
  // an HIPCOO input reaches a {HIPCSR} operator through the device conversion
  auto *o2 = rcm.GetReorder(dcoo, {hip_context.get()}, true);
  EXPECT_TRUE(same(o2, want_rcm, 3));
  delete[] o2;
  // device constructor sorts like the host one
  int rp4[5]{0, 2, 3, 3, 4}, c4[4]{2, 0, 1, 3}, v4[4]{5, 4, 7, 9};
  auto &d0 = hip::Device::Get(hip_context->device_id);
  DCSR3 sorted(4, 4, 4, d0.Upload(rp4, 5), d0.Upload(c4, 4), d0.Upload(v4, 4), *hip_context, format::kOwned);
  const int want_col[4]{0, 2, 1, 3}, want_vals[4]{4, 5, 7, 9};
  EXPECT_TRUE(same(fetch(d0, sorted.get_col(), 4).data(), want_col, 4));
  EXPECT_TRUE(same(fetch(d0, sorted.get_vals(), 4).data(), want_vals, 4));
  std::unique_ptr<format::Format> clone(sorted.Clone());
  EXPECT_NE(clone->AsAbsolute<DCSR3>()->get_col(), sorted.get_col());
  delete dcsr2;
  delete dcoo;
  delete dcsr;
  EXPECT_THROW(context::HIPContext bad(hip::DeviceCount() + 3), utils::HIPDeviceException);
}

// The device-resident overloads (additive; the reference trades the order vector as a host array,
// bases/reorder_base.h:50-66,:145-150): same results as the host-array signatures, nothing crosses PCIe in between.
TEST(ReorderBase, DeviceResidentOrderVector) {
  CSR3 global_csr(n, n, row_ptr, cols, vals, format::kNotOwned);
  std::unique_ptr<DCSR3> dcsr(global_csr.Convert<format::HIPCSR>(hip_context.get()));
  auto &dev = dcsr->device();
  // Reorder -> HIPArray on the format's device; RCM and Degree return what the kernels wrote, Gray uploads its host result
  std::unique_ptr<format::HIPArray<int>> d_rcm(bases::ReorderBase::Reorder<reorder::RCMReorder>({}, dcsr.get(), *hip_context));
  const int want_rcm[3] = {1, 2, 0};
  EXPECT_EQ(d_rcm->get_num_nnz(), (format::DimensionType)n);
  EXPECT_TRUE(d_rcm->get_context()->IsEquivalent(hip_context.get()));
  EXPECT_TRUE(same(fetch(dev, d_rcm->get_vals(), n).data(), want_rcm, n));
  for (bool asc : {true, false}) {
    std::unique_ptr<format::HIPArray<int>> d_deg(
        bases::ReorderBase::Reorder<reorder::DegreeReorder>({asc}, dcsr.get(), *hip_context));
    const int want_asc[3] = {2, 1, 0}, want_desc[3] = {0, 1, 2};
    EXPECT_TRUE(same(fetch(dev, d_deg->get_vals(), n).data(), asc ? want_asc : want_desc, n));
  }
  std::unique_ptr<format::HIPArray<int>> d_gray(
      bases::ReorderBase::Reorder<reorder::GrayReorder>({reorder::BitSize16, 100, 10}, dcsr.get(), *hip_context));
  const int want_gray[3] = {2, 0, 1};
  EXPECT_TRUE(same(fetch(dev, d_gray->get_vals(), n).data(), want_gray, n));
  // opt-in device ordering of GrayReorder (stable ties): on this fixture every key is unique, so it is the reference's order
  reorder::GrayReorderParams stable(reorder::BitSize16, 100, 10);
  stable.stable_device_ordering = true;
  std::unique_ptr<format::HIPArray<int>> d_gray_stable(
      bases::ReorderBase::Reorder<reorder::GrayReorder>(stable, dcsr.get(), *hip_context));
  EXPECT_TRUE(same(fetch(dev, d_gray_stable->get_vals(), n).data(), want_gray, n));
  // a host CSR is converted on the way (convert_input) and the result still lands on the device
  std::unique_ptr<format::HIPArray<int>> d_from_host(
      bases::ReorderBase::Reorder<reorder::RCMReorder>({}, &global_csr, *hip_context, true));
  EXPECT_TRUE(same(fetch(dev, d_from_host->get_vals(), n).data(), want_rcm, n));
  // Permute2D with device-resident order vectors == the host-array call
  format::HIPArray<int> d_r(n, dev.Upload(r_reorder_vector, n), *hip_context, format::kOwned);
  format::HIPArray<int> d_c(n, dev.Upload(c_reorder_vector, n), *hip_context, format::kOwned);
  auto *rc = bases::ReorderBase::Permute2DRowColumnWise<format::HIPCSR>(&d_r, &d_c, dcsr.get(), {hip_context.get()}, false);
  EXPECT_TRUE(same(fetch(dev, rc->get_row_ptr(), n + 1).data(), rc_row_ptr, n + 1));
  EXPECT_TRUE(same(fetch(dev, rc->get_col(), nnz).data(), rc_cols, nnz));
  EXPECT_TRUE(same(fetch(dev, rc->get_vals(), nnz).data(), rc_vals, nnz));
  delete rc;
  auto *rw = bases::ReorderBase::Permute2DRowWise<format::HIPCSR>(&d_r, dcsr.get(), {hip_context.get()}, false);
  EXPECT_TRUE(same(fetch(dev, rw->get_col(), nnz).data(), r_cols, nnz) && same(fetch(dev, rw->get_vals(), nnz).data(), r_vals, nnz));
  delete rw;
  auto *both = bases::ReorderBase::Permute2D<format::HIPCSR>(d_rcm.get(), dcsr.get(), {hip_context.get()}, false);
  auto *both_host = bases::ReorderBase::Permute2D<format::HIPCSR>(const_cast<int *>(want_rcm), dcsr.get(), {hip_context.get()}, false);
  EXPECT_TRUE(fetch(dev, both->get_col(), nnz) == fetch(dev, both_host->get_col(), nnz));
  EXPECT_TRUE(fetch(dev, both->get_vals(), nnz) == fetch(dev, both_host->get_vals(), nnz));
  EXPECT_TRUE(fetch(dev, both->get_row_ptr(), n + 1) == fetch(dev, both_host->get_row_ptr(), n + 1));
  delete both;
  delete both_host;
  // a host CSR with device-resident orders: staged through the same device, host result
  auto *hp = bases::ReorderBase::Permute2DRowColumnWise<format::CSR>(&d_r, &d_c, &global_csr, {&cpu_context}, true);
  EXPECT_TRUE(same(hp->get_col(), rc_cols, nnz) && same(hp->get_vals(), rc_vals, nnz));
  delete hp;
  // the order vectors stay the caller's: still readable afterwards
  EXPECT_TRUE(same(fetch(dev, d_r.get_vals(), n).data(), r_reorder_vector, n));
}

// The host layer's block pool (hip/device.h): a released block serves the next request of its size; a block whose
// ownership left the layer (release_*()) is forgotten and never handed out again.
TEST(HIPDevice, BlockPool) {
  auto &dev = hip::Device::Get(hip_context->device_id);
  if (hip::Device::pool_limit() == 0) return;
  dev.TrimPool(0);
  void *a = dev.Malloc(1 << 20);
  dev.Free(a);
  void *b = dev.Malloc(1 << 20);
  EXPECT_EQ(a, b);                       // served from the pool
  void *c = dev.Malloc(1 << 20);
  EXPECT_NE(c, b);
  dev.Free(c);
  dev.Free(b);
  EXPECT_TRUE(dev.TrimPool(0));          // both blocks go back to the driver
  EXPECT_FALSE(dev.TrimPool(0));
  int host[4] = {1, 2, 3, 4};
  auto *arr = new format::HIPArray<int>(4, dev.Upload(host, 4), *hip_context, format::kOwned);
  int *raw = arr->release_vals();        // ownership leaves the layer
  delete arr;
  EXPECT_TRUE(same(fetch(dev, raw, 4).data(), host, 4));
  EXPECT_EQ(sbx_free(dev.handle(), raw), SBX_OK);
  EXPECT_FALSE(dev.TrimPool(0));         // nothing of it stayed behind in the pool
}

int main() {
  if (hip::DeviceCount() < 1) {
    std::printf("test_reference_suite needs a GPU (the path has no CPU fallback)\n");
    return 2;
  }
  utils::Logger::set_level(utils::LOG_LVL_NONE);
  hip_context.reset(new context::HIPContext(0));
  return minitest::run_all();
}
