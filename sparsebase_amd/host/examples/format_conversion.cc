// Counterpart of examples/format_conversion/format_conversion.cc:10-54 (caller H3):
// COO -> CSR -> CSR (identity) -> COO through the member Convert<> syntax.
#include <iostream>

#include "sparsebase/sparsebase.h"

using namespace sparsebase;

int main() {
  int row[6] = {0, 0, 1, 1, 2, 2};
  int col[6] = {0, 1, 1, 2, 3, 3};
  int vals[6] = {10, 20, 30, 40, 50, 60};
  context::CPUContext cpu_context;
  auto *coo = new format::COO<int, int, int>(6, 6, 6, row, col, vals);
  auto *csr = coo->Convert<format::CSR>(&cpu_context);
  auto *csr2 = csr->Convert<format::CSR>(&cpu_context);
  const int n = csr2->get_dimensions()[0], nnz = csr->get_num_nnz();
  std::cout << "CSR" << std::endl;
  for (int i = 0; i < nnz; i++) std::cout << csr2->get_vals()[i] << ",";
  std::cout << std::endl;
  for (int i = 0; i < nnz; i++) std::cout << csr2->get_col()[i] << ",";
  std::cout << std::endl;
  for (int i = 0; i < n + 1; i++) std::cout << csr2->get_row_ptr()[i] << ",";
  std::cout << std::endl << std::endl;
  auto *coo2 = csr->Convert<format::COO>(&cpu_context);
  std::cout << "COO" << std::endl;
  for (int i = 0; i < nnz; i++) std::cout << coo2->get_vals()[i] << ",";
  std::cout << std::endl;
  for (int i = 0; i < nnz; i++) std::cout << coo2->get_row()[i] << ",";
  std::cout << std::endl;
  for (int i = 0; i < nnz; i++) std::cout << coo2->get_col()[i] << ",";
  std::cout << std::endl;
  const bool same_obj = (csr2 == csr);
  delete coo2;
  delete csr;
  delete coo;
  return same_obj ? 0 : 3;
}
