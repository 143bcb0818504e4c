// reorder_cli — runs one reorderer of the host layer on a CSR given as raw int32 binary
// files and writes the inverse permutation; used by tests/test_host_layer.py to check the
// C++ API end to end (Gray's device keys + host ordering stage in particular) against
// fixtures produced by the real reference.
// Usage: reorder_cli <rcm|degree_asc|degree_desc|gray> <row_ptr.bin> <col.bin> <out.bin> <n> <m> [res thr grp] [--device] [--time]
// --time repeats the call five times and prints the last (warm) call's wall time in seconds on stdout (gray: and, on a
// second line, the ms of its device key stage, of the copy of the keys to the host and of the host ordering stage)
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

int main(int argc, char **argv) {
  if (argc < 7) return 1;
  const std::string kind = argv[1];
  auto rp = read_bin(argv[2]);
  auto col = read_bin(argv[3]);
  const int n = atoi(argv[5]), m = atoi(argv[6]);
  bool on_device = false, timed = false;
  for (int i = 7; i < argc; i++) {
    on_device |= !strcmp(argv[i], "--device");
    timed |= !strcmp(argv[i], "--time");
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
      reorder::GrayReorder<int, int, void> r((reorder::BitMapSize)atoi(argv[7]), atoi(argv[8]), atoi(argv[9]));
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
    const auto t0 = std::chrono::steady_clock::now();
    order = run();
    std::printf("%.6f\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    if (kind == "gray") {  // the stages of that call: device key stage, keys to the host, host ordering (ms)
      const double *st = reorder::GrayReorder<int, int, void>::last_stage_ms();
      std::printf("%.4f %.4f %.4f\n", st[0], st[1], st[2]);
    }
  }
  std::ofstream out(argv[4], std::ios::binary);
  out.write((const char *)order, (size_t)n * sizeof(int));
  delete[] order;
  return 0;
}
