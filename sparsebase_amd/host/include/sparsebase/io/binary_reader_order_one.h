// sparsebase/io/binary_reader_order_one.h — SbFF reader for Array (reference:
// io/binary_reader_order_one.cc:16-34); ReadHIPArray uploads the payload directly.
#ifndef SPARSEBASE_IO_BINARY_READER_ORDER_ONE_H_
#define SPARSEBASE_IO_BINARY_READER_ORDER_ONE_H_
#include <memory>
#include <string>

#include "sparsebase/format/array.h"
#include "sparsebase/format/hip_formats.h"
#include "sparsebase/io/sparse_file_format.h"

namespace sparsebase::io {

template <typename T>
class BinaryReaderOrderOne {
  static_assert(!std::is_same_v<T, void>, "A BinaryReaderOrderOne cannot read an Array of type void");

 public:
  explicit BinaryReaderOrderOne(std::string filename) : filename_(std::move(filename)) {}
  format::Array<T> *ReadArray() const {
    size_t size = 0;
    std::unique_ptr<T[]> v = Load(&size);
    return new format::Array<T>((format::DimensionType)size, v.release(), format::kOwned);
  }
  format::HIPArray<T> *ReadHIPArray(context::HIPContext ctx) const {
    size_t size = 0;
    std::unique_ptr<T[]> v = Load(&size);
    auto &dev = hip::Device::Get(ctx.device_id);
    return new format::HIPArray<T>((format::DimensionType)size, dev.Upload(v.get(), size ? size : 1), ctx, format::kOwned);
  }

 private:
  std::unique_ptr<T[]> Load(size_t *size) const {
    SbffFile f(filename_);
    if (f.name() != "array") throw utils::ReaderException("SBFF file is not in Array format");
    if (f.dimensions().empty() || f.dimensions()[0] < 0) throw utils::ReaderException("SBFF file holds no dimension");
    *size = (size_t)f.dimensions()[0];
    const SbffEntry &e = f.template Typed<T>("array");
    std::unique_ptr<T[]> v(new T[*size ? *size : 1]);
    f.ReadPayload(e, v.get(), *size);
    return v;
  }
  std::string filename_;
};

}  // namespace sparsebase::io
#endif
