// sparsebase/io/binary_writer_order_two.h — SbFF writer for COO / CSR (reference:
// io/binary_writer_order_two.h, io/binary_writer_order_two.cc:16-53).
// One deliberate difference: WriteCSR stores all nnz entries of `col` / `vals`.  The reference
// stores dimensions[1] (= the column count) entries (:43, :47), which loses data whenever a
// matrix has more nonzeros than columns and reads past the arrays when it has fewer; the two
// writers produce the same arrays exactly when nnz == column count (the case the reference's
// tests use), and the reference's reader accepts either file.
#ifndef SPARSEBASE_IO_BINARY_WRITER_ORDER_TWO_H_
#define SPARSEBASE_IO_BINARY_WRITER_ORDER_TWO_H_
#include <string>

#include "sparsebase/format/coo.h"
#include "sparsebase/format/csr.h"
#include "sparsebase/io/sparse_file_format.h"

namespace sparsebase::io {

template <typename IDType, typename NNZType, typename ValueType>
class BinaryWriterOrderTwo {
 public:
  explicit BinaryWriterOrderTwo(std::string filename) : filename_(std::move(filename)) {}

  void WriteCOO(format::COO<IDType, NNZType, ValueType> *coo) const {
    SbffWriter sbff("coo");
    sbff.AddDimensions(coo->get_dimensions());
    const size_t nnz = (size_t)coo->get_num_nnz();
    sbff.AddArray("row", coo->get_row(), nnz);
    sbff.AddArray("col", coo->get_col(), nnz);
    if constexpr (!std::is_same_v<ValueType, void>)
      if (coo->get_vals() != nullptr) sbff.AddArray("vals", coo->get_vals(), nnz);
    sbff.Write(filename_);
  }
  void WriteCSR(format::CSR<IDType, NNZType, ValueType> *csr) const {
    SbffWriter sbff("csr");
    const auto dims = csr->get_dimensions();
    sbff.AddDimensions(dims);
    const size_t nnz = (size_t)csr->get_num_nnz();
    sbff.AddArray("row_ptr", csr->get_row_ptr(), (size_t)dims[0] + 1);
    sbff.AddArray("col", csr->get_col(), nnz);
    if constexpr (!std::is_same_v<ValueType, void>)
      if (csr->get_vals() != nullptr) sbff.AddArray("vals", csr->get_vals(), nnz);
    sbff.Write(filename_);
  }

 private:
  std::string filename_;
};

}  // namespace sparsebase::io
#endif
