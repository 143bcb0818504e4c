// sparsebase/io/binary_writer_order_one.h — SbFF writer for Array (reference:
// io/binary_writer_order_one.cc:13-24).
#ifndef SPARSEBASE_IO_BINARY_WRITER_ORDER_ONE_H_
#define SPARSEBASE_IO_BINARY_WRITER_ORDER_ONE_H_
#include <string>

#include "sparsebase/format/array.h"
#include "sparsebase/io/sparse_file_format.h"

namespace sparsebase::io {

template <typename T>
class BinaryWriterOrderOne {
 public:
  explicit BinaryWriterOrderOne(std::string filename) : filename_(std::move(filename)) {}
  void WriteArray(format::Array<T> *arr) const {
    static_assert(!std::is_same_v<T, void>, "Cannot write an Array with void ValueType");
    SbffWriter sbff("array");
    sbff.AddDimensions(arr->get_dimensions());
    sbff.AddArray("array", arr->get_vals(), (size_t)arr->get_dimensions()[0]);
    sbff.Write(filename_);
  }

 private:
  std::string filename_;
};

}  // namespace sparsebase::io
#endif
