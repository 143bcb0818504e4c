// CPU-only tests of the host layer's plumbing: conversion graph, function matcher,
// converter store, ownership, exceptions.  No GPU work is issued (formats are built
// with ignore_sort=true and every registered function is a local stand-in), so this
// runs in the build container.  Behavioural spec: the reference's
// tests/suites/sparsebase/utils/function_matcher_mixin_tests.cc:99-349,
// converter/converter_tests.cc and format/{csr,coo}_tests.cc (ownership parts).
#include "minitest.h"
#include "sparsebase/sparsebase.h"

using namespace sparsebase;
typedef format::CSR<int, int, int> CSR3;
typedef format::COO<int, int, int> COO3;

static int g_row_ptr[4] = {0, 2, 3, 4}, g_cols[4] = {1, 2, 0, 0}, g_vals[4] = {1, 2, 3, 4}, g_rows[4] = {0, 0, 1, 2};
static context::CPUContext cpu;

// ---- a user-defined format + converter edges (docs: "custom_format" example) ----
template <typename I, typename N, typename V>
class Tagged : public utils::IdentifiableImplementation<Tagged<I, N, V>, format::FormatOrderTwo<I, N, V>> {
 public:
  explicit Tagged(int tag) : tag(tag) {
    this->order_ = 2;
    this->dimension_ = {1, 1};
    this->nnz_ = 0;
    this->context_ = std::unique_ptr<context::Context>(new context::CPUContext);
  }
  format::Format *Clone() const override { return new Tagged(tag); }
  int tag;
};
typedef Tagged<int, int, int> Tag3;

static format::Format *CsrToTag(format::Format *, context::Context *) { return new Tag3(1); }
static format::Format *CooToCsrStub(format::Format *, context::Context *) {
  return new CSR3(3, 3, g_row_ptr, g_cols, g_vals, format::kNotOwned, true);
}
static bool Always(context::Context *, context::Context *) { return true; }
static bool Never(context::Context *, context::Context *) { return false; }

class TestConverter : public converter::ConverterImpl<TestConverter> {
 public:
  converter::Converter *Clone() const override { return new TestConverter(*this); }
  void Reset() override {}
};

TEST(Converter, ChainSearchAndApply) {
  TestConverter c;
  COO3 coo(3, 3, 4, g_rows, g_cols, g_vals, format::kNotOwned, true);
  EXPECT_FALSE(c.CanConvert(COO3::get_id_static(), &cpu, Tag3::get_id_static(), &cpu));
  c.RegisterConversionFunction(COO3::get_id_static(), CSR3::get_id_static(), CooToCsrStub, Always);
  c.RegisterConversionFunction(CSR3::get_id_static(), Tag3::get_id_static(), CsrToTag, Always);
  auto chain = c.GetConversionChain(COO3::get_id_static(), &cpu, Tag3::get_id_static(), {&cpu});
  EXPECT_TRUE(chain.has_value());
  EXPECT_EQ(std::get<1>(*chain), 2u);              // two hops, unit cost each
  EXPECT_EQ(std::get<0>(*chain).size(), (size_t)2);
  // cached: returns every created format; plain: only the last
  auto all = c.ConvertCached(&coo, Tag3::get_id_static(), &cpu);
  EXPECT_EQ(all.size(), (size_t)2);
  EXPECT_TRUE(all[0]->IsAbsolute<CSR3>());
  EXPECT_TRUE(all[1]->IsAbsolute<Tag3>());
  for (auto *f : all) delete f;
  format::Format *last = c.Convert(&coo, Tag3::get_id_static(), &cpu);
  EXPECT_EQ(last->AsAbsolute<Tag3>()->tag, 1);
  delete last;
  // same type + equivalent context -> the source itself
  EXPECT_EQ(c.Convert(&coo, COO3::get_id_static(), &cpu), (format::Format *)&coo);
  // move map is separate
  EXPECT_FALSE(c.CanConvert(COO3::get_id_static(), &cpu, CSR3::get_id_static(), &cpu, true));
  // clearing
  c.ClearConversionFunctions(CSR3::get_id_static(), Tag3::get_id_static());
  EXPECT_FALSE(c.CanConvert(COO3::get_id_static(), &cpu, Tag3::get_id_static(), &cpu));
  EXPECT_TRUE(c.CanConvert(COO3::get_id_static(), &cpu, CSR3::get_id_static(), &cpu));
  c.ClearConversionFunctions();
  EXPECT_FALSE(c.CanConvert(COO3::get_id_static(), &cpu, CSR3::get_id_static(), &cpu));
  EXPECT_THROW(c.Convert(&coo, CSR3::get_id_static(), &cpu), utils::ConversionException);
}

TEST(Converter, ConditionsSelectEdges) {
  TestConverter c;
  c.RegisterConversionFunction(COO3::get_id_static(), CSR3::get_id_static(), CooToCsrStub, Never);
  EXPECT_FALSE(c.CanConvert(COO3::get_id_static(), &cpu, CSR3::get_id_static(), &cpu));
  // a second edge between the same pair with a passing condition is found
  c.RegisterConversionFunction(COO3::get_id_static(), CSR3::get_id_static(), CooToCsrStub, Always);
  EXPECT_TRUE(c.CanConvert(COO3::get_id_static(), &cpu, CSR3::get_id_static(), &cpu));
}

TEST(Converter, StoreIsASingletonPerType) {
  auto a = converter::ConverterStore::GetStore().get_converter<converter::ConverterOrderTwo<int, int, int>>();
  auto b = converter::ConverterStore::GetStore().get_converter<converter::ConverterOrderTwo<int, int, int>>();
  auto c = converter::ConverterStore::GetStore().get_converter<converter::ConverterOrderTwo<int, int, float>>();
  EXPECT_EQ(a.get(), b.get());
  EXPECT_NE((void *)a.get(), (void *)c.get());
  CSR3 csr(3, 3, g_row_ptr, g_cols, g_vals, format::kNotOwned, true);
  EXPECT_EQ(csr.get_converter().get(), (const converter::Converter *)a.get());
}

TEST(Converter, RegisteredGraphOfTheHotPath) {
  // which conversions exist for which target context (no function is executed here)
  auto conv = converter::ConverterStore::GetStore().get_converter<converter::ConverterOrderTwo<int, int, int>>();
  typedef format::HIPCSR<int, int, int> DCSR;
  typedef format::HIPCOO<int, int, int> DCOO;
  EXPECT_TRUE(conv->CanConvert(COO3::get_id_static(), &cpu, CSR3::get_id_static(), &cpu));
  EXPECT_TRUE(conv->CanConvert(CSR3::get_id_static(), &cpu, COO3::get_id_static(), &cpu));
  EXPECT_TRUE(conv->CanConvert(COO3::get_id_static(), &cpu, CSR3::get_id_static(), &cpu, true));
  EXPECT_TRUE(conv->CanConvert(CSR3::get_id_static(), &cpu, COO3::get_id_static(), &cpu, true));
  // device formats are unreachable with only a CPU context on offer
  EXPECT_FALSE(conv->CanConvert(CSR3::get_id_static(), &cpu, DCSR::get_id_static(), &cpu));
  EXPECT_FALSE(conv->CanConvert(COO3::get_id_static(), &cpu, DCOO::get_id_static(), &cpu));
}

// ---- FunctionMatcherMixin -----------------------------------------------------
static int FnOne(std::vector<format::Format *>, utils::Parameters *) { return 1; }
static int FnTwo(std::vector<format::Format *>, utils::Parameters *) { return 2; }
static int FnTag(std::vector<format::Format *> f, utils::Parameters *) { return 10 + f[0]->AsAbsolute<Tag3>()->tag; }

class Matcher : public utils::FunctionMatcherMixin<int> {
 public:
  int Run(format::Format *f, std::vector<context::Context *> ctxs, bool convert) {
    return this->Execute(nullptr, ctxs, convert, f);
  }
  std::tuple<std::vector<std::vector<format::Format *>>, int> RunCached(format::Format *f,
                                                                        std::vector<context::Context *> ctxs) {
    return this->CachedExecute(nullptr, ctxs, true, false, f);
  }
  int Run2(format::Format *a, format::Format *b) { return this->Execute(nullptr, {&cpu}, true, a, b); }
};

TEST(FunctionMatcher, RegisterOverrideUnregister) {
  Matcher m;
  CSR3 csr(3, 3, g_row_ptr, g_cols, g_vals, format::kNotOwned, true);
  EXPECT_THROW(m.Run(&csr, {&cpu}, true), utils::FunctionNotFoundException);  // empty map
  EXPECT_TRUE(m.RegisterFunctionNoOverride({CSR3::get_id_static()}, FnOne));
  EXPECT_FALSE(m.RegisterFunctionNoOverride({CSR3::get_id_static()}, FnTwo));
  EXPECT_EQ(m.Run(&csr, {&cpu}, false), 1);
  m.RegisterFunction({CSR3::get_id_static()}, FnTwo);  // overrides
  EXPECT_EQ(m.Run(&csr, {&cpu}, false), 2);
  EXPECT_EQ(m.GetAvailableFormats().size(), (size_t)1);
  EXPECT_TRUE(m.UnregisterFunction({CSR3::get_id_static()}));
  EXPECT_FALSE(m.UnregisterFunction({CSR3::get_id_static()}));
  EXPECT_THROW(m.Run(&csr, {&cpu}, true), utils::FunctionNotFoundException);
}

TEST(FunctionMatcher, ConvertsInputsWhenAllowed) {
  // give the shared <int,int,int> converter a CSR->Tagged edge, then ask an operator
  // that only knows Tagged to run on a CSR
  auto conv = converter::ConverterStore::GetStore().get_converter<converter::ConverterOrderTwo<int, int, int>>();
  conv->RegisterConversionFunction(CSR3::get_id_static(), Tag3::get_id_static(), CsrToTag, Always);
  Matcher m;
  m.RegisterFunction({Tag3::get_id_static()}, FnTag);
  CSR3 csr(3, 3, g_row_ptr, g_cols, g_vals, format::kNotOwned, true);
  EXPECT_EQ(m.Run(&csr, {&cpu}, true), 11);
  EXPECT_THROW(m.Run(&csr, {&cpu}, false),
               utils::DirectExecutionNotAvailableException<std::vector<std::type_index>>);
  auto cached = m.RunCached(&csr, {&cpu});
  EXPECT_EQ(std::get<1>(cached), 11);
  EXPECT_EQ(std::get<0>(cached).size(), (size_t)1);
  EXPECT_EQ(std::get<0>(cached)[0].size(), (size_t)1);  // the converted input is handed back
  EXPECT_TRUE(std::get<0>(cached)[0][0]->IsAbsolute<Tag3>());
  delete std::get<0>(cached)[0][0];
  // a direct hit returns no intermediates
  Tag3 t(5);
  auto direct = m.RunCached(&t, {&cpu});
  EXPECT_EQ(std::get<1>(direct), 15);
  EXPECT_EQ(std::get<0>(direct)[0].size(), (size_t)0);
  conv->ClearConversionFunctions(CSR3::get_id_static(), Tag3::get_id_static());
}

TEST(FunctionMatcher, MultiFormatKeys) {
  Matcher m;
  m.RegisterFunction({CSR3::get_id_static(), Tag3::get_id_static()}, FnTwo);
  CSR3 csr(3, 3, g_row_ptr, g_cols, g_vals, format::kNotOwned, true);
  Tag3 t(0);
  EXPECT_EQ(m.Run2(&csr, &t), 2);
  EXPECT_THROW(m.Run2(&t, &csr), utils::FunctionNotFoundException);
}

TEST(Operators, RegisterHostAndDeviceKeys) {
  // every operator of the path offers {CSR} (staged through the GPU) and {HIPCSR}
  reorder::RCMReorder<int, int, int> rcm;
  reorder::DegreeReorder<int, int, int> deg(true);
  reorder::GrayReorder<int, int, int> gray(reorder::BitSize16, 10, 4);
  int order[3] = {0, 1, 2};
  permute::PermuteOrderTwo<int, int, int> perm(order, order);
  EXPECT_EQ(rcm.GetAvailableFormats().size(), (size_t)2);
  EXPECT_EQ(deg.GetAvailableFormats().size(), (size_t)2);
  EXPECT_EQ(gray.GetAvailableFormats().size(), (size_t)2);
  EXPECT_EQ(perm.GetAvailableFormats().size(), (size_t)2);
  // the params constructor registers too (the reference's registers nothing, permute_order_two.cc:17-20)
  permute::PermuteOrderTwo<int, int, int> perm2(permute::PermuteOrderTwoParams<int>(order, nullptr));
  EXPECT_EQ(perm2.GetAvailableFormats().size(), (size_t)2);
  // a COO input without permission to convert is refused before anything runs
  COO3 coo(3, 3, 4, g_rows, g_cols, g_vals, format::kNotOwned, true);
  EXPECT_THROW(perm.GetPermutation(&coo, {&cpu}, false),
               utils::DirectExecutionNotAvailableException<std::vector<std::type_index>>);
  EXPECT_THROW(rcm.GetReorder(&coo, {&cpu}, false),
               utils::DirectExecutionNotAvailableException<std::vector<std::type_index>>);
}

TEST(Format, OwnershipAndCasting) {
  int *rp = new int[4]{0, 2, 3, 4};
  int *col = new int[4]{1, 2, 0, 0};
  int *val = new int[4]{1, 2, 3, 4};
  {
    CSR3 owned(3, 3, rp, col, val, format::kOwned, true);
    EXPECT_TRUE(owned.RowPtrIsOwned() && owned.ColIsOwned() && owned.ValsIsOwned());
    EXPECT_EQ(owned.get_num_nnz(), (format::DimensionType)4);
    int *released = owned.release_col();  // hand-off: the format stops owning it
    EXPECT_EQ(released, col);
    EXPECT_FALSE(owned.ColIsOwned());
    EXPECT_EQ(owned.get_col(), col);
    format::Format *base = &owned;
    EXPECT_NO_THROW(base->AsAbsolute<CSR3>());
    EXPECT_THROW(base->AsAbsolute<COO3>(), utils::TypeException);
    EXPECT_TRUE(owned.Is<format::CSR>());
    EXPECT_FALSE(owned.Is<format::COO>());
    std::unique_ptr<format::Format> copy(owned.Clone());
    EXPECT_NE((void *)copy->AsAbsolute<CSR3>()->get_col(), (void *)col);
    EXPECT_EQ(copy->AsAbsolute<CSR3>()->get_col()[1], 2);
  }
  delete[] col;  // released above, so still ours
  CSR3 borrowed(3, 3, g_row_ptr, g_cols, g_vals, format::kNotOwned, true);
  EXPECT_FALSE(borrowed.ColIsOwned());
  format::CSR<int, int, void> pattern(3, 3, g_row_ptr, g_cols, nullptr, format::kNotOwned, true);
  EXPECT_EQ(pattern.get_vals(), (void *)nullptr);
  EXPECT_TRUE(cpu.IsEquivalent(borrowed.get_context()));
}

// The exact Gray mode's sorts must leave what libstdc++'s std::sort leaves (ties included): the threaded form against
// the library call on the kinds of input the reorderer sorts, with grains small enough to split every range many times.
TEST(GraySort, ReplicaSelfCheckPassesOnThisLibrary) {
  // (the once-per-process check GrayIntroSort runs before its first parallel sort; compiled out — and trivially absent —
  // outside the window of libstdc++ releases the replica is built for)
#if defined(SBX_GRAY_SORT_REPLICA)
  EXPECT_TRUE(reorder::detail::GraySortReplicaAgrees());
#endif
  // the shared budget: leases never hand out more than kMax extra threads together, and give them back
  {
    reorder::detail::GrayThreadBudget a(16), b(16), c(16);
    EXPECT_TRUE(a.threads() >= 1 && b.threads() >= 1 && c.threads() >= 1);
    EXPECT_TRUE(a.threads() + b.threads() + c.threads() - 3 <= (unsigned)reorder::detail::GrayThreadBudget::kMax);
  }
  reorder::detail::GrayThreadBudget again(2);
  EXPECT_TRUE(again.threads() >= 1);
}

#if defined(__linux__)
TEST(GraySort, NodeScopeGivesTheCallersMaskBack) {
  // the ordering stage keeps its threads on the caller's NUMA node by narrowing the calling thread's affinity mask for the
  // stage's duration (threads created meanwhile inherit it): whatever it did, the caller has its own mask afterwards
  cpu_set_t before, during, after;
  CPU_ZERO(&before);
  CPU_ZERO(&during);
  CPU_ZERO(&after);
  EXPECT_EQ(sched_getaffinity(0, sizeof(before), &before), 0);
  {
    reorder::detail::GrayNodeScope scope;
    EXPECT_EQ(sched_getaffinity(0, sizeof(during), &during), 0);
    EXPECT_TRUE(CPU_COUNT(&during) >= 1 && CPU_COUNT(&during) <= CPU_COUNT(&before));
    for (int c = 0; c < CPU_SETSIZE; c++)
      if (CPU_ISSET(c, &during)) EXPECT_TRUE(CPU_ISSET(c, &before));  // (never a CPU the caller could not use)
    std::thread t([&]() {  // a thread created inside the scope starts with the narrowed mask
      cpu_set_t child;
      CPU_ZERO(&child);
      EXPECT_EQ(sched_getaffinity(0, sizeof(child), &child), 0);
      EXPECT_TRUE(CPU_EQUAL(&child, &during));
    });
    t.join();
  }
  EXPECT_EQ(sched_getaffinity(0, sizeof(after), &after), 0);
  EXPECT_TRUE(CPU_EQUAL(&before, &after));
}
#endif

TEST(GraySort, ParallelReplicaOfStdSort) {
  unsigned long long state = 88172645463325252ull;
  auto rnd = [&state]() {
    state ^= state << 13, state ^= state >> 7, state ^= state << 17;
    return state;
  };
  auto by_degree = [](uint32_t l, uint32_t r) -> bool { return (l >> 24) < (r >> 24); };
  typedef std::pair<int, unsigned long> row_key;
  auto asc = [](const row_key &l, const row_key &r) -> bool { return l.second < r.second; };
  auto desc = [](const row_key &l, const row_key &r) -> bool { return l.second > r.second; };
  int differing = 0;
  for (int round = 0; round < 60; round++) {
    const size_t count = round < 4 ? (size_t)round * 8 : 1 + (size_t)(rnd() % 200000);
    const unsigned distinct = 1u + (unsigned)(rnd() % (round % 3 == 0 ? 3 : round % 3 == 1 ? 200 : 1u << 20));
    const int64_t grain = round % 2 ? 64 : 1 + (int64_t)(rnd() % 5000);
    const unsigned threads = 2 + (unsigned)(rnd() % 7);
    std::vector<uint32_t> packed(count);
    for (size_t i = 0; i < count; i++) packed[i] = ((uint32_t)(rnd() % distinct % 256) << 24) | (uint32_t)(i & 0xFFFFFF);
    if (round % 7 == 0) std::sort(packed.begin(), packed.end());
    if (round % 11 == 0) std::reverse(packed.begin(), packed.end());
    std::vector<uint32_t> expect = packed;
    std::sort(expect.begin(), expect.end(), by_degree);
    // (par_min: every third round the ranges above a few thousand elements are partitioned by the whole team)
    const int64_t par_min = round % 3 == 0 ? 200 + (int64_t)(rnd() % 20000) : (round % 3 == 1 ? -1 : 0);
    reorder::detail::GrayIntroSort(packed.begin(), packed.end(), by_degree, threads, grain, par_min);
    differing += packed != expect;
    std::vector<row_key> pairs(count);
    for (size_t i = 0; i < count; i++) pairs[i] = row_key((int)i, (unsigned long)(rnd() % distinct));
    std::vector<row_key> expect_pairs = pairs;
    if (round % 2) {
      std::sort(expect_pairs.begin(), expect_pairs.end(), asc);
      reorder::detail::GrayIntroSort(pairs.begin(), pairs.end(), asc, threads, grain, par_min);
    } else {
      std::sort(expect_pairs.begin(), expect_pairs.end(), desc);
      reorder::detail::GrayIntroSort(pairs.begin(), pairs.end(), desc, threads, grain, par_min);
    }
    differing += pairs != expect_pairs;
  }
  EXPECT_EQ(differing, 0);
  // Musser's adversary for the median-of-three pivot drives introsort to its depth limit (99 ranges of this input end in
  // the heap sort of std::__partial_sort): above the grain that is the pool's own branch, below it the library's
  {
    const int count = 1 << 16, half = count / 2;
    std::vector<uint32_t> killer((size_t)count, 0u), killer_expect;
    for (int i = 1; i <= half; i++) {
      if (i % 2) killer[(size_t)i - 1] = (uint32_t)i, killer[(size_t)i] = (uint32_t)(half + i);
      killer[(size_t)(half + i - 1)] = (uint32_t)(2 * i);
    }
    auto less = [](uint32_t l, uint32_t r) -> bool { return l < r; };
    for (int64_t grain : {(int64_t)32, (int64_t)4096}) {
      for (int64_t par_min : {(int64_t)-1, (int64_t)300}) {  // (300: the depth limit is reached inside the team's part)
        std::vector<uint32_t> work = killer;
        killer_expect = killer;
        std::sort(killer_expect.begin(), killer_expect.end(), less);
        reorder::detail::GrayIntroSort(work.begin(), work.end(), less, 4, grain, par_min);
        EXPECT_TRUE(work == killer_expect);
      }
    }
  }
}

TEST(Device, FailsLoudlyWithoutAGpu) {
  if (hip::DeviceCount() > 0) return;  // only meaningful in the CPU-only container
  EXPECT_THROW(context::HIPContext bad(0), utils::HIPDeviceException);
  reorder::DegreeReorder<int, int, int> deg(true);
  CSR3 csr(3, 3, g_row_ptr, g_cols, g_vals, format::kNotOwned, true);
  // the {CSR} implementation stages through the GPU: no device -> exception, never a CPU result
  EXPECT_THROW(deg.GetReorder(&csr, {&cpu}, true), utils::HIPDeviceException);
  int r[4] = {0, 0, 3, 1}, c[4] = {2, 0, 3, 1}, v[4] = {5, 4, 9, 7};
  EXPECT_THROW(COO3 unsorted(4, 4, 4, r, c, v, format::kNotOwned), utils::HIPDeviceException);
}

int main(int argc, char **argv) { return minitest::run_all(argc > 1 ? argv[1] : nullptr); }
