// Counterpart of examples/gray_order/gray_order.cc:36-83 (caller H2): GrayReorder with the
// example's parameters (BitSize16, 20, (nnz/n)/16) and a permutation check.
// Usage: gray_order <file.mtx>   (column count must be a multiple of 16)
#include <iostream>

#include "sparsebase/sparsebase.h"


using namespace sparsebase;

int main(int argc, char *argv[]) {
  if (argc < 2) {
    std::cout << "Usage: ./gray_order <matrix_market_format>\n";
    return 1;
  }
  context::CPUContext cpu_context;
  auto *coo = bases::IOBase::ReadMTXToCOO<int, int, void>(argv[1], true);
  auto *csr = coo->Convert<format::CSR>(&cpu_context);
  const int n = csr->get_dimensions()[0];
  const int avg = (int)(csr->get_num_nnz() / (n ? n : 1));
  reorder::GrayReorder<int, int, void> orderer(reorder::BitSize16, 20, avg / 16);
  int *order = orderer.GetReorder(csr, {&cpu_context}, true);
  std::vector<char> seen(n, 0);
  bool ok = true;
  for (int i = 0; i < n; i++) {
    if (order[i] < 0 || order[i] >= n || seen[order[i]]) ok = false;
    else seen[order[i]] = 1;
  }
  std::cout << (ok ? "Order is correct" : "Order is NOT correct") << std::endl;
  delete[] order;
  delete csr;
  delete coo;
  return ok ? 0 : 3;
}
