// Minimal Matrix Market coordinate reader for the example programs (config C1
// plumbing: .mtx -> COO -> CSR).  Text parsing is host/IO work and is NOT part of the
// accelerated path; the reference's full reader is io/mtx_reader.cc (out of scope).
#ifndef EXAMPLES_MTX_IO_H_
#define EXAMPLES_MTX_IO_H_
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "sparsebase/sparsebase.h"

namespace examples {

// Reads a "coordinate" .mtx (pattern / real / integer, general or symmetric) into a COO
// with 0-based indices and no values (pattern), the way ReadMTXToCSR(file, true) is used
// by examples/degree_order/degree_order.cc:26-33.
template <typename IDType, typename NNZType>
sparsebase::format::COO<IDType, NNZType, void> *ReadMTXToCOO(const std::string &path) {
  std::ifstream in(path);
  if (!in) throw std::runtime_error("cannot open " + path);
  std::string line;
  bool symmetric = false;
  if (!std::getline(in, line) || line.rfind("%%MatrixMarket", 0) != 0) throw std::runtime_error("not a MatrixMarket file");
  if (line.find("coordinate") == std::string::npos) throw std::runtime_error("only coordinate format is supported");
  if (line.find("symmetric") != std::string::npos) symmetric = true;
  while (std::getline(in, line) && !line.empty() && line[0] == '%') {}
  std::istringstream hdr(line);
  long long n, m, entries;
  hdr >> n >> m >> entries;
  std::vector<IDType> r, c;
  r.reserve(entries * (symmetric ? 2 : 1));
  c.reserve(entries * (symmetric ? 2 : 1));
  for (long long k = 0; k < entries && std::getline(in, line); k++) {
    std::istringstream es(line);
    long long i, j;
    es >> i >> j;
    r.push_back((IDType)(i - 1));
    c.push_back((IDType)(j - 1));
    if (symmetric && i != j) {
      r.push_back((IDType)(j - 1));
      c.push_back((IDType)(i - 1));
    }
  }
  const size_t nnz = r.size();
  IDType *row = new IDType[nnz], *col = new IDType[nnz];
  std::copy(r.begin(), r.end(), row);
  std::copy(c.begin(), c.end(), col);
  // the COO constructor puts the entries in (row,col) order — on the GPU
  return new sparsebase::format::COO<IDType, NNZType, void>((IDType)n, (IDType)m, (NNZType)nnz, row, col, nullptr,
                                                           sparsebase::format::kOwned);
}

}  // namespace examples
#endif
