// sparsebase/io/binary_reader_order_two.h — SbFF reader for COO / CSR (reference:
// io/binary_reader_order_two.h, io/binary_reader_order_two.cc:15-72).  ReadHIPCSR / ReadHIPCOO
// upload the file's arrays directly; ReadCSR / ReadCOO build the host formats, whose
// constructors sort on the device like every other host format of this layer.
// Where the reference is only defined for nnz == column count, this reader takes the counts
// from the file: COO nnz = length of `row` (reference: dimensions[1], :71); a CSR file must
// hold at least row_ptr[n] entries of `col` (a longer array, as the reference's writer produces
// when nnz < column count, is cut at nnz).  `vals` is optional for both formats (reference:
// COO only, :61-68); a file with values cannot be read into ValueType void (:64-66).
#ifndef SPARSEBASE_IO_BINARY_READER_ORDER_TWO_H_
#define SPARSEBASE_IO_BINARY_READER_ORDER_TWO_H_
#include <memory>
#include <string>

#include "sparsebase/format/coo.h"
#include "sparsebase/format/csr.h"
#include "sparsebase/format/hip_formats.h"
#include "sparsebase/io/sparse_file_format.h"

namespace sparsebase::io {

template <typename IDType, typename NNZType, typename ValueType>
class BinaryReaderOrderTwo {
 public:
  explicit BinaryReaderOrderTwo(std::string filename) : filename_(std::move(filename)) {}

  format::CSR<IDType, NNZType, ValueType> *ReadCSR() const {
    CsrArrays a = LoadCSR();
    return new format::CSR<IDType, NNZType, ValueType>((IDType)a.n, (IDType)a.m, a.row_ptr.release(), a.col.release(),
                                                       a.Vals(), format::kOwned);
  }
  format::COO<IDType, NNZType, ValueType> *ReadCOO() const {
    CooArrays a = LoadCOO();
    return new format::COO<IDType, NNZType, ValueType>((IDType)a.n, (IDType)a.m, (NNZType)a.nnz, a.row.release(),
                                                       a.col.release(), a.Vals(), format::kOwned);
  }
  format::HIPCSR<IDType, NNZType, ValueType> *ReadHIPCSR(context::HIPContext ctx) const {
    CsrArrays a = LoadCSR();
    auto &dev = hip::Device::Get(ctx.device_id);
    NNZType *row_ptr = dev.Upload(a.row_ptr.get(), (size_t)a.n + 1);
    IDType *col = dev.Upload(a.col.get(), a.nnz ? a.nnz : 1);
    ValueType *vals = UploadVals(dev, a.vals.get(), a.nnz);
    return new format::HIPCSR<IDType, NNZType, ValueType>((IDType)a.n, (IDType)a.m, (NNZType)a.nnz, row_ptr, col, vals, ctx,
                                                          format::kOwned);
  }
  format::HIPCOO<IDType, NNZType, ValueType> *ReadHIPCOO(context::HIPContext ctx) const {
    CooArrays a = LoadCOO();
    auto &dev = hip::Device::Get(ctx.device_id);
    IDType *row = dev.Upload(a.row.get(), a.nnz ? a.nnz : 1);
    IDType *col = dev.Upload(a.col.get(), a.nnz ? a.nnz : 1);
    ValueType *vals = UploadVals(dev, a.vals.get(), a.nnz);
    return new format::HIPCOO<IDType, NNZType, ValueType>((IDType)a.n, (IDType)a.m, (NNZType)a.nnz, row, col, vals, ctx,
                                                          format::kOwned);
  }

 private:
  // `void` has no array type: values are carried as bytes and cast at the end
  using Stored = std::conditional_t<std::is_same_v<ValueType, void>, char, ValueType>;
  struct CsrArrays {
    long long n = 0, m = 0;
    size_t nnz = 0;
    std::unique_ptr<NNZType[]> row_ptr;
    std::unique_ptr<IDType[]> col;
    std::unique_ptr<Stored[]> vals;
    ValueType *Vals() { return reinterpret_cast<ValueType *>(vals.release()); }
  };
  struct CooArrays {
    long long n = 0, m = 0;
    size_t nnz = 0;
    std::unique_ptr<IDType[]> row, col;
    std::unique_ptr<Stored[]> vals;
    ValueType *Vals() { return reinterpret_cast<ValueType *>(vals.release()); }
  };
  static ValueType *UploadVals(const hip::Device &dev, const Stored *host, size_t nnz) {
    if constexpr (std::is_same_v<ValueType, void>) return nullptr;
    else return host ? dev.Upload(host, nnz ? nnz : 1) : nullptr;
  }
  static void Dimensions(const SbffFile &f, long long *n, long long *m) {
    if (f.dimensions().size() < 2) throw utils::ReaderException("SBFF file does not hold two dimensions");
    *n = f.dimensions()[0];
    *m = f.dimensions()[1];
    if (*n < 0 || *m < 0) throw utils::ReaderException("SBFF file holds a negative dimension");
  }
  static std::unique_ptr<Stored[]> Values(SbffFile &f, size_t nnz) {
    if (!f.Has("vals")) return nullptr;
    if constexpr (std::is_same_v<ValueType, void>) {
      throw utils::ReaderException("Cannot read a weighted COO into a format with void ValueType");
    } else {
      const SbffEntry &e = f.template Typed<ValueType>("vals");
      std::unique_ptr<Stored[]> v(new Stored[nnz ? nnz : 1]);
      f.ReadPayload(e, v.get(), nnz);
      return v;
    }
  }
  CsrArrays LoadCSR() const {
    SbffFile f(filename_);
    if (f.name() != "csr") throw utils::ReaderException("SBFF file is not in CSR format");
    CsrArrays a;
    Dimensions(f, &a.n, &a.m);
    const SbffEntry &rp = f.template Typed<NNZType>("row_ptr");
    if (rp.array_size < (size_t)a.n + 1) throw utils::ReaderException("SBFF row_ptr is shorter than rows + 1");
    a.row_ptr.reset(new NNZType[(size_t)a.n + 1]);
    f.ReadPayload(rp, a.row_ptr.get(), (size_t)a.n + 1);
    if (a.row_ptr[a.n] < 0) throw utils::ReaderException("SBFF row_ptr ends in a negative count");
    a.nnz = (size_t)a.row_ptr[a.n];
    const SbffEntry &ce = f.template Typed<IDType>("col");
    a.col.reset(new IDType[a.nnz ? a.nnz : 1]);
    f.ReadPayload(ce, a.col.get(), a.nnz);
    a.vals = Values(f, a.nnz);
    return a;
  }
  CooArrays LoadCOO() const {
    SbffFile f(filename_);
    if (f.name() != "coo") throw utils::ReaderException("SBFF file is not in COO format");
    CooArrays a;
    Dimensions(f, &a.n, &a.m);
    const SbffEntry &re = f.template Typed<IDType>("row");
    const SbffEntry &ce = f.template Typed<IDType>("col");
    a.nnz = re.array_size;
    a.row.reset(new IDType[a.nnz ? a.nnz : 1]);
    a.col.reset(new IDType[a.nnz ? a.nnz : 1]);
    f.ReadPayload(re, a.row.get(), a.nnz);
    f.ReadPayload(ce, a.col.get(), a.nnz);
    a.vals = Values(f, a.nnz);
    return a;
  }
  std::string filename_;
};

}  // namespace sparsebase::io
#endif
