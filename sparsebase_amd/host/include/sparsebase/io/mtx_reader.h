// sparsebase/io/mtx_reader.h — Matrix Market reader (reference: io/mtx_reader.h:16-63,
// io/mtx_reader.cc:11-585) for the coordinate format.  The banner and the size line are
// parsed here exactly like the reference's ParseHeader (:29-120, same exceptions); the
// entry section — where the time goes — is shipped to the GPU and parsed there
// (sbx_mtx_parse_coordinate: tokenization, exact decimal -> binary conversion, symmetric
// expansion), followed by the COO constructor's sort, also on the GPU.
// Array-format files (dense) are not part of this path and throw ReaderException.
#ifndef SPARSEBASE_IO_MTX_READER_H_
#define SPARSEBASE_IO_MTX_READER_H_
#include <fstream>
#include <sstream>
#include <string>

#include "sparsebase/converter/converter_order_two.h"
#include "sparsebase/format/coo.h"
#include "sparsebase/format/csr.h"
#include "sparsebase/format/hip_formats.h"
#include "sparsebase/utils/exception.h"

namespace sparsebase::io {

template <typename IDType, typename NNZType, typename ValueType>
class MTXReader {
 public:
  explicit MTXReader(std::string filename, bool convert_to_zero_index = true, bool upper_triangle = false)
      : filename_(std::move(filename)), convert_to_zero_index_(convert_to_zero_index), upper_triangle_(upper_triangle) {
    std::ifstream fin(filename_);
    if (!fin.is_open()) throw utils::ReaderException("Wrong matrix market file name\n");
    std::string header_line;
    std::getline(fin, header_line);
    options_ = ParseHeader(header_line);
  }

  // host COO, entries in (row, col) order as the reference's COO constructor leaves them
  format::COO<IDType, NNZType, ValueType> *ReadCOO() const {
    std::unique_ptr<format::HIPCOO<IDType, NNZType, ValueType>> d(ReadHIPCOO(context::HIPContext(hip::DefaultDevice())));
    context::CPUContext cpu;
    return static_cast<format::COO<IDType, NNZType, ValueType> *>(
        converter::HIPCooCooConditionalFunction<IDType, NNZType, ValueType>(d.get(), &cpu));
  }
  format::CSR<IDType, NNZType, ValueType> *ReadCSR() const {  // mtx_reader.cc:517-523: ReadCOO + convert
    std::unique_ptr<format::HIPCOO<IDType, NNZType, ValueType>> d(ReadHIPCOO(context::HIPContext(hip::DefaultDevice())));
    context::HIPContext gpu(d->get_hip_context()->device_id);
    std::unique_ptr<format::Format> dcsr(converter::HIPCooHIPCsrFunction<IDType, NNZType, ValueType, true>(d.get(), &gpu));
    context::CPUContext cpu;
    return static_cast<format::CSR<IDType, NNZType, ValueType> *>(
        converter::HIPCsrCsrConditionalFunction<IDType, NNZType, ValueType>(dcsr.get(), &cpu));
  }
  // the same matrix left in HBM (what a GPU pipeline wants: no host copy of the entries at all)
  format::HIPCOO<IDType, NNZType, ValueType> *ReadHIPCOO(context::HIPContext ctx) const {
    (void)hip::IndexTag<IDType, NNZType>();  // (a COO holds id arrays only: any tuple the device path takes)
    if (options_.format != kCoordinate)
      throw utils::ReaderException("array-format Matrix Market files are not read by this library");
    std::ifstream fin(filename_, std::ios::binary);
    if (!fin.is_open()) throw utils::ReaderException("file does not exists!!");
    std::string text((std::istreambuf_iterator<char>(fin)), std::istreambuf_iterator<char>());
    // skip the banner and the comment lines (:318-319), read the size line (:321)
    size_t pos = 0;
    while (pos < text.size() && text[pos] == '%') {
      const size_t eol = text.find('\n', pos);
      pos = eol == std::string::npos ? text.size() : eol + 1;
    }
    const size_t size_end = text.find('\n', pos);
    std::istringstream size_line(text.substr(pos, size_end == std::string::npos ? std::string::npos : size_end - pos));
    long long M = 0, N = 0, L = 0;
    size_line >> M >> N >> L;
    if (!size_line) throw utils::ReaderException("malformed size line in matrix market file");
    const size_t body = size_end == std::string::npos ? text.size() : size_end + 1;
    const bool weighted = options_.field != kPattern;
    const int symmetry = options_.symmetry == kGeneral ? 0 : options_.symmetry == kSymmetric ? 1 : 2;
    // the reference honours upper_triangle for symmetric files only (:201-214)
    const bool upper = upper_triangle_ && options_.symmetry == kSymmetric;
    const bool expand = symmetry != 0 && !upper;
    auto &dev = hip::Device::Get(ctx.device_id);
    const size_t cap = (size_t)L * (expand ? 2 : 1) + 1;
    hip::Staged<char> d_text(dev, text.data() + body, text.size() - body + 1);
    IDType *row = (IDType *)dev.Malloc(cap * sizeof(IDType)), *col = (IDType *)dev.Malloc(cap * sizeof(IDType));
    void *val = nullptr;
    constexpr size_t vb = hip::ValueBytes<ValueType>();
    if (weighted && vb) val = dev.Malloc(cap * vb);
    int64_t nnz = 0;
    unsigned flags = (convert_to_zero_index_ ? SBX_MTX_ZERO_INDEX : 0u) | (upper ? SBX_MTX_UPPER_TRIANGLE : 0u);
    const int rc = sbx_mtx_parse_coordinate(dev.handle(), hip::IndexTag<IDType>(), hip::ValueTag<ValueType>(), d_text.get(),
                                            (int64_t)(text.size() - body), M, N, L, weighted ? 3 : 2, symmetry, flags,
                                            (int64_t)cap, row, col, val, &nnz);
    if (rc != SBX_OK) {
      dev.Free(row);
      dev.Free(col);
      if (val) dev.Free(val);
      throw utils::ReaderException(std::string("matrix market coordinate section: ") + sbx_last_error(dev.handle()));
    }
    // the COO constructor checks / sorts on the device (format/coo.cc:96-157)
    return new format::HIPCOO<IDType, NNZType, ValueType>((IDType)M, (IDType)N, (NNZType)nnz, row, col, (ValueType *)val,
                                                         ctx, format::kOwned, false);
  }

 private:
  enum Format { kCoordinate, kArray };
  enum Field { kReal, kDouble, kComplex, kInteger, kPattern };
  enum Symmetry { kGeneral, kSymmetric, kSkewSymmetric };
  struct Options {
    Format format;
    Field field;
    Symmetry symmetry;
  };
  static void NoVoidValues() {
    if constexpr (std::is_same_v<void, ValueType>)
      throw utils::ReaderException("You are reading the values of the matrix market file into a void array");
  }
  static Options ParseHeader(const std::string &header_line) {  // mtx_reader.cc:29-120
    std::stringstream line_ss(header_line);
    Options o;
    std::string prefix, object, format, field, symmetry;
    line_ss >> prefix >> object >> format >> field >> symmetry;
    if (prefix != "%%MatrixMarket") throw utils::ReaderException("Wrong prefix in a matrix market file");
    if (object == "vector")
      throw utils::ReaderException("Matrix market reader does not currently support reading vectors.");
    if (object != "matrix") throw utils::ReaderException("Illegal value for the 'object' option in matrix market header");
    if (format == "array") o.format = kArray;
    else if (format == "coordinate") o.format = kCoordinate;
    else throw utils::ReaderException("Illegal value for the 'format' option in matrix market header");
    if (field == "real") { o.field = kReal; NoVoidValues(); }
    else if (field == "double") { o.field = kDouble; NoVoidValues(); }
    else if (field == "complex") { o.field = kComplex; NoVoidValues(); }
    else if (field == "integer") { o.field = kInteger; NoVoidValues(); }
    else if (field == "pattern") o.field = kPattern;
    else throw utils::ReaderException("Illegal value for the 'field' option in matrix market header");
    if (symmetry == "general") o.symmetry = kGeneral;
    else if (symmetry == "symmetric") o.symmetry = kSymmetric;
    else if (symmetry == "skew-symmetric") o.symmetry = kSkewSymmetric;
    else if (symmetry == "hermitian")
      throw utils::ReaderException("Matrix market reader does not currently support hermitian symmetry.");
    else throw utils::ReaderException("Illegal value for the 'symmetry' option in matrix market header");
    return o;
  }

  std::string filename_;
  bool convert_to_zero_index_, upper_triangle_;
  Options options_;
};

}  // namespace sparsebase::io
#endif
