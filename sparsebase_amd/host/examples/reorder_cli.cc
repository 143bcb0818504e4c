// reorder_cli — runs one reorderer of the host layer on a CSR given as raw int32 binary
// files and writes the inverse permutation; used by tests/test_host_layer.py to check the
// C++ API end to end (Gray's device keys + host ordering stage in particular) against
// fixtures produced by the real reference.
// Usage: reorder_cli <rcm|degree_asc|degree_desc|gray> <row_ptr.bin> <col.bin> <out.bin> <n> <m> [res thr grp] [--device] [--time] [--stable]
// --stable (gray): the opt-in ordering on the device with stable ties (GrayReorderParams::stable_device_ordering)
// --time repeats the call five times and prints the last (warm) call's wall time in seconds on stdout (gray: and, on a
// second line, the ms of its device key stage, of the copy of the keys to the host and of the host ordering stage)
//        reorder_cli pipeline <row_ptr.bin> <col.bin> <out.bin> <n> <m>
// the canonical pipeline of the reference's experiment helper (experiment/experiment_helper.h:81-97) on a device-
// resident HIPCSR<int, int, float>: ReorderBase::Reorder<RCMReorder> -> ReorderBase::Permute2D<HIPCSR> ->
// Convert<HIPCSR>; warm calls, prints "reorder_ms permute2d_ms convert_ms" of the best of five rounds twice: first
// line through the reference's own signatures (the order vector is a host array between the calls: two trips over
// PCIe), second line through the device-resident overloads (Reorder -> HIPArray<int> -> Permute2D: what a C++ caller
// of the boundary pays per call is dispatch, handle lookup and the result objects); writes the order
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <vector>

#include "sparsebase/sparsebase.h"

using namespace sparsebase;

static std::vector<int> read_bin(const char *path) {
  std::ifstream f(path, std::ios::binary | std::ios::ate);
  const size_t bytes = (size_t)f.tellg();
  f.seekg(0);
  std::vector<int> v(bytes / sizeof(int));
  f.read((char *)v.data(), bytes);
  return v;
}

static int pipeline(const std::vector<int> &rp, const std::vector<int> &col, int n, int m, const char *out_path) {
  using clock = std::chrono::steady_clock;
  // (device outputs are complete in stream order, and some calls return with their last kernels still running: the
  // clock is read behind a sync, so that every kernel is charged to the call that enqueued it)
  auto ms = [](clock::time_point a) {
    hip::Device::Get(0).Sync();
    return std::chrono::duration<double, std::milli>(clock::now() - a).count();
  };
  utils::Logger::set_level(utils::LOG_LVL_NONE);
  std::vector<float> val(col.size());
  for (size_t i = 0; i < val.size(); i++) val[i] = (float)(i % 1021);
  format::CSR<int, int, float> csr(n, m, const_cast<int *>(rp.data()), const_cast<int *>(col.data()), val.data(),
                                   format::kNotOwned, true);
  context::HIPContext gpu(0);
  std::unique_ptr<format::HIPCSR<int, int, float>> dcsr(csr.Convert<format::HIPCSR>(&gpu));
  std::vector<context::Context *> ctxs{&gpu};
  // (a) the reference's own signatures: the order vector is a host array between the two calls
  double best[3] = {1e30, 1e30, 1e30};
  int *order = nullptr;
  for (int round = 0; round < 7; round++) {  // (two rounds to settle the scratch arena, five measured)
    delete[] order;
    auto t = clock::now();
    order = bases::ReorderBase::Reorder<reorder::RCMReorder>({}, dcsr.get(), ctxs, true);
    const double t_reorder = ms(t);
    t = clock::now();
    auto *perm = bases::ReorderBase::Permute2D<format::HIPCSR>(order, dcsr.get(), ctxs, true);
    const double t_permute = ms(t);
    t = clock::now();
    auto *conv = perm->Convert<format::HIPCSR>(&gpu);  // (the experiment helper's third step; same format: no copy is due)
    const double t_convert = ms(t);
    if (round >= 2) {
      best[0] = std::min(best[0], t_reorder), best[1] = std::min(best[1], t_permute), best[2] = std::min(best[2], t_convert);
    }
    if ((void *)conv != (void *)perm) delete conv;
    delete perm;
  }
  // (b) the device-resident overloads: the order vector stays in HBM as an HIPArray<int>
  double dbest[3] = {1e30, 1e30, 1e30};
  std::vector<int> dorder((size_t)n);
  for (int round = 0; round < 7; round++) {
    auto t = clock::now();
    std::unique_ptr<format::HIPArray<int>> d_order(bases::ReorderBase::Reorder<reorder::RCMReorder>({}, dcsr.get(), gpu));
    const double t_reorder = ms(t);
    t = clock::now();
    auto *perm = bases::ReorderBase::Permute2D<format::HIPCSR>(d_order.get(), dcsr.get(), ctxs, true);
    const double t_permute = ms(t);
    t = clock::now();
    auto *conv = perm->Convert<format::HIPCSR>(&gpu);
    const double t_convert = ms(t);
    if (round >= 2) {
      dbest[0] = std::min(dbest[0], t_reorder), dbest[1] = std::min(dbest[1], t_permute), dbest[2] = std::min(dbest[2], t_convert);
    }
    if (round == 6) hip::Device::Get(0).ToHost(dorder.data(), d_order->get_vals(), (size_t)n * sizeof(int));
    if ((void *)conv != (void *)perm) delete conv;
    delete perm;
  }
  if (memcmp(dorder.data(), order, (size_t)n * sizeof(int)) != 0) {
    std::fprintf(stderr, "pipeline: the device-resident order differs from the host-array one\n");
    return 2;
  }
  std::printf("%.4f %.4f %.4f\n", best[0], best[1], best[2]);
  std::printf("%.4f %.4f %.4f\n", dbest[0], dbest[1], dbest[2]);
  std::ofstream out(out_path, std::ios::binary);
  out.write((const char *)order, (size_t)n * sizeof(int));
  delete[] order;
  return 0;
}

int main(int argc, char **argv) {
  if (argc < 7) return 1;
  const std::string kind = argv[1];
  auto rp = read_bin(argv[2]);
  auto col = read_bin(argv[3]);
  const int n = atoi(argv[5]), m = atoi(argv[6]);
  if (kind == "pipeline") return pipeline(rp, col, n, m, argv[4]);
  bool on_device = false, timed = false, stable = false;
  int idle_ms = 0, reps = 1;
  for (int i = 7; i < argc; i++) {
    on_device |= !strcmp(argv[i], "--device");
    timed |= !strcmp(argv[i], "--time");
    stable |= !strcmp(argv[i], "--stable");  // gray: GrayReorderParams::stable_device_ordering (sbx_gray_reorder)
    if (!strcmp(argv[i], "--idle-ms") && i + 1 < argc) idle_ms = atoi(argv[i + 1]);  // (diagnostic: busy host between calls)
    // --reps N: N timed calls; the one of MEDIAN duration is the one reported (all durations go to stderr): the exact Gray
    // mode's threaded host stage has calls that take twice the usual time
    if (!strcmp(argv[i], "--reps") && i + 1 < argc) reps = atoi(argv[i + 1]) > 0 ? atoi(argv[i + 1]) : 1;
  }
  utils::Logger::set_level(utils::LOG_LVL_NONE);
  context::CPUContext cpu;
  format::CSR<int, int, void> csr(n, m, rp.data(), col.data(), nullptr, format::kNotOwned, true);
  std::unique_ptr<context::HIPContext> gpu;
  format::Format *input = &csr;
  std::vector<context::Context *> ctxs{&cpu};
  std::unique_ptr<format::Format> dcsr;
  if (on_device) {  // same call on a device-resident HIPCSR
    gpu.reset(new context::HIPContext(0));
    dcsr.reset(csr.Convert<format::HIPCSR>(gpu.get()));
    input = dcsr.get();
    ctxs = {gpu.get()};
  }
  auto run = [&]() -> int * {
    if (kind == "rcm") {
      reorder::RCMReorder<int, int, void> r;
      return r.GetReorder(input, ctxs, false);
    }
    if (kind == "degree_asc" || kind == "degree_desc") {
      reorder::DegreeReorder<int, int, void> r(kind == "degree_asc");
      return r.GetReorder(input, ctxs, false);
    }
    if (kind == "gray") {
      reorder::GrayReorderParams gp((reorder::BitMapSize)atoi(argv[7]), atoi(argv[8]), atoi(argv[9]));
      gp.stable_device_ordering = stable;
      reorder::GrayReorder<int, int, void> r(gp);
      return r.GetReorder(input, ctxs, false);
    }
    return nullptr;
  };
  int *order = run();
  if (!order) return 1;
  if (timed) {
    delete[] order;
    for (int warm = 0; warm < 4; warm++) {  // (the library's scratch arena settles into one block over the first calls)
      order = run();
      delete[] order;
    }
    if (idle_ms > 0) {
      const auto until = std::chrono::steady_clock::now() + std::chrono::milliseconds(idle_ms);
      while (std::chrono::steady_clock::now() < until) {}
    }
    struct Timed {
      double seconds, st[7];
    };
    std::vector<Timed> calls;
    for (int rpt = 0; rpt < reps; rpt++) {
      if (rpt) delete[] order;
      const auto t0 = std::chrono::steady_clock::now();
      order = run();
      Timed t;
      t.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      for (int i = 0; i < 7; i++) t.st[i] = 0;
      if (kind == "gray" && !stable) {
        const double *st = reorder::GrayReorder<int, int, void>::last_stage_ms();
        for (int i = 0; i < 7; i++) t.st[i] = st[i];
      }
      calls.push_back(t);
    }
    if (reps > 1) {
      std::fprintf(stderr, "timed calls (ms):");
      for (const Timed &t : calls) std::fprintf(stderr, " %.2f", t.seconds * 1e3);
      std::fprintf(stderr, "\n");
    }
    std::sort(calls.begin(), calls.end(), [](const Timed &a, const Timed &b) { return a.seconds < b.seconds; });
    const Timed &mid = calls[(calls.size() - 1) / 2];
    std::printf("%.6f\n", mid.seconds);
    if (kind == "gray" && !stable) {  // the stages of that call: device key stage, keys to the host, host ordering (ms)
      std::fprintf(stderr, "gray host stage (ms): split %.2f, sort by degree %.2f, sections %.2f, dense rows + order %.2f\n",
                   mid.st[3], mid.st[4], mid.st[5], mid.st[6]);
      std::printf("%.4f %.4f %.4f\n", mid.st[0], mid.st[1], mid.st[2]);
    }
  }
  std::ofstream out(argv[4], std::ios::binary);
  out.write((const char *)order, (size_t)n * sizeof(int));
  delete[] order;
  return 0;
}
