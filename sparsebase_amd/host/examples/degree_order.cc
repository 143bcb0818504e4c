// Counterpart of the reference's examples/degree_order/degree_order.cc:20-134 (caller H1):
// read a graph, DegreeReorder it, verify the degrees are monotone, permute, verify the
// permuted row lengths, permute back with the inverse and compare with the original.
// Usage: degree_order <file.mtx>     (reference input: examples/data/ash958.mtx)
#include <iostream>

#include "sparsebase/sparsebase.h"


using namespace sparsebase;
using vertex_type = unsigned int;
using edge_type = unsigned int;

int main(int argc, char *argv[]) {
  if (argc < 2) {
    std::cout << "Usage: ./degree_order <matrix_market_format>\n";
    return 1;
  }
  context::CPUContext cpu_context;
  auto *coo = bases::IOBase::ReadMTXToCOO<vertex_type, edge_type, void>(argv[1], true);
  auto *csr = coo->Convert<format::CSR>(&cpu_context);
  const vertex_type n = csr->get_dimensions()[0];
  auto *row_ptr = csr->get_row_ptr();
  std::cout << "Number of vertices: " << n << "\nNumber of edges: " << csr->get_num_nnz() << std::endl;

  reorder::DegreeReorder<vertex_type, edge_type, void> orderer(true);
  vertex_type *permutation = orderer.GetReorder(csr, {&cpu_context}, true);
  auto *order = new vertex_type[n];
  for (vertex_type i = 0; i < n; i++) order[permutation[i]] = i;
  bool order_is_correct = true;
  std::vector<char> seen(n, 0);
  for (vertex_type i = 0; i < n; i++) {
    if (seen[order[i]]) order_is_correct = false;
    seen[order[i]] = 1;
    if (i + 1 < n) {
      const vertex_type u = order[i], v = order[i + 1];
      if (row_ptr[u + 1] - row_ptr[u] > row_ptr[v + 1] - row_ptr[v]) order_is_correct = false;
    }
  }
  std::cout << (order_is_correct ? "Order is correct." : "Order is NOT correct.") << std::endl;

  permute::PermuteOrderTwo<vertex_type, edge_type, void> transformer(permutation, permutation);
  // rectangular inputs (ash958 is 958x292): permute the rows only, as the column order has another length
  const bool square = csr->get_dimensions()[0] == csr->get_dimensions()[1];
  permute::PermuteOrderTwo<vertex_type, edge_type, void> row_only(permutation, nullptr);
  auto *permuted = (square ? transformer : row_only).GetPermutation(csr, {&cpu_context}, true)->As<format::CSR>();
  bool transform_is_correct = true;
  for (vertex_type i = 0; i + 1 < n; i++)
    if (permuted->get_row_ptr()[i + 2] - permuted->get_row_ptr()[i + 1] <
        permuted->get_row_ptr()[i + 1] - permuted->get_row_ptr()[i])
      transform_is_correct = false;
  std::cout << (transform_is_correct ? "Transformation is correct." : "Transformation is NOT correct.") << std::endl;

  auto *inv = bases::ReorderBase::InversePermutation(permutation, n);
  permute::PermuteOrderTwo<vertex_type, edge_type, void> inverse(inv, square ? inv : nullptr);
  auto *restored = inverse.GetPermutation(permuted, {&cpu_context}, true)->As<format::CSR>();
  bool inverse_is_correct = true;
  for (vertex_type i = 0; i <= n; i++) inverse_is_correct &= restored->get_row_ptr()[i] == row_ptr[i];
  for (edge_type j = 0; j < csr->get_num_nnz(); j++) inverse_is_correct &= restored->get_col()[j] == csr->get_col()[j];
  std::cout << (inverse_is_correct ? "Inversion is correct." : "Inversion is NOT correct.") << std::endl;
  std::cout << "first/last of permutation: " << permutation[0] << " " << permutation[n - 1] << std::endl;

  delete restored;
  delete[] inv;
  delete permuted;
  delete[] order;
  delete[] permutation;
  delete csr;
  delete coo;
  return (order_is_correct && transform_is_correct && inverse_is_correct) ? 0 : 3;
}
