// Counterpart of examples/rcm_order/rcm_order.cc:40-74 (caller H2): RCM-reorder a graph
// and check that the result is a permutation; then permute with it (experiment::ReorderCSR,
// experiment/experiment_helper.h:81-97) keeping everything on the device.
// Usage: rcm_order <symmetric.mtx>
#include <iostream>

#include "sparsebase/sparsebase.h"


using namespace sparsebase;
using vertex_type = unsigned int;
using edge_type = unsigned int;

int main(int argc, char *argv[]) {
  if (argc < 2) {
    std::cout << "Usage: ./rcm_order <matrix_market_format>\n";
    return 1;
  }
  context::CPUContext cpu_context;
  context::HIPContext gpu(0);
  auto *coo = bases::IOBase::ReadMTXToCOO<vertex_type, edge_type, void>(argv[1], true);
  // host COO -> device CSR in one call: the graph search chains COO->HIPCOO->HIPCSR
  auto *dcsr = coo->Convert<format::HIPCSR>(&gpu);
  const vertex_type n = dcsr->get_dimensions()[0];
  std::cout << "Number of vertices: " << n << "\nNumber of edges: " << dcsr->get_num_nnz() << std::endl;
  vertex_type *order = bases::ReorderBase::Reorder<reorder::RCMReorder>({}, dcsr, {&gpu}, true);
  std::vector<char> seen(n, 0);
  bool ok = true;
  for (vertex_type i = 0; i < n; i++) {
    if (order[i] >= n || seen[order[i]]) ok = false;
    else seen[order[i]] = 1;
  }
  std::cout << (ok ? "Order is correct" : "Order is NOT correct") << std::endl;
  auto *permuted = bases::ReorderBase::Permute2D<format::HIPCSR>(order, dcsr, {&gpu}, true);
  auto *host = permuted->Convert<format::CSR>(&cpu_context);
  unsigned long long bandwidth = 0;
  for (vertex_type i = 0; i < n; i++)
    for (edge_type j = host->get_row_ptr()[i]; j < host->get_row_ptr()[i + 1]; j++) {
      const long long d = (long long)host->get_col()[j] - (long long)i;
      bandwidth = std::max<unsigned long long>(bandwidth, (unsigned long long)(d < 0 ? -d : d));
    }
  std::cout << "bandwidth after RCM: " << bandwidth << std::endl;
  delete host;
  delete permuted;
  delete[] order;
  delete dcsr;
  delete coo;
  return ok ? 0 : 3;
}
