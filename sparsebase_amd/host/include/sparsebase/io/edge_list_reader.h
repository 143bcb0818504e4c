// sparsebase/io/edge_list_reader.h — edge list reader (reference: io/edge_list_reader.h:20-45,
// io/edge_list_reader.cc:9-158).  The file goes to the GPU as it is: sbx_edge_list_parse
// tokenizes it, parses the vertex ids (and weights, converted exactly), drops self edges, adds the
// reverse edges, finds the dimensions, sorts by (row, col) and removes duplicates — the whole of
// ReadCOO.  Duplicates with DIFFERENT weights: the reference's unstable std::sort leaves the
// survivor unspecified; here the first one in file order survives.
#ifndef SPARSEBASE_IO_EDGE_LIST_READER_H_
#define SPARSEBASE_IO_EDGE_LIST_READER_H_
#include <fstream>
#include <string>

#include "sparsebase/converter/converter_order_two.h"
#include "sparsebase/format/coo.h"
#include "sparsebase/format/csr.h"
#include "sparsebase/format/hip_formats.h"
#include "sparsebase/utils/exception.h"

namespace sparsebase::io {

template <typename IDType, typename NNZType, typename ValueType>
class EdgeListReader {
 public:
  explicit EdgeListReader(std::string filename, bool weighted = false, bool remove_duplicates = false,
                          bool remove_self_edges = false, bool read_undirected = true, bool square = false)
      : filename_(std::move(filename)), weighted_(weighted), remove_duplicates_(remove_duplicates),
        remove_self_edges_(remove_self_edges), read_undirected_(read_undirected), square_(square) {}

  format::COO<IDType, NNZType, ValueType> *ReadCOO() const {
    std::unique_ptr<format::HIPCOO<IDType, NNZType, ValueType>> d(ReadHIPCOO(context::HIPContext(hip::DefaultDevice())));
    context::CPUContext cpu;
    return static_cast<format::COO<IDType, NNZType, ValueType> *>(
        converter::HIPCooCooConditionalFunction<IDType, NNZType, ValueType>(d.get(), &cpu));
  }
  format::CSR<IDType, NNZType, ValueType> *ReadCSR() const {  // edge_list_reader.cc:160-167
    std::unique_ptr<format::HIPCOO<IDType, NNZType, ValueType>> d(ReadHIPCOO(context::HIPContext(hip::DefaultDevice())));
    context::HIPContext gpu(d->get_hip_context()->device_id);
    std::unique_ptr<format::Format> dcsr(converter::HIPCooHIPCsrFunction<IDType, NNZType, ValueType, true>(d.get(), &gpu));
    context::CPUContext cpu;
    return static_cast<format::CSR<IDType, NNZType, ValueType> *>(
        converter::HIPCsrCsrConditionalFunction<IDType, NNZType, ValueType>(dcsr.get(), &cpu));
  }
  format::HIPCOO<IDType, NNZType, ValueType> *ReadHIPCOO(context::HIPContext ctx) const {
    (void)hip::IndexTag<IDType, NNZType>();  // (a COO holds id arrays only: any tuple the device path takes)
    if constexpr (std::is_same_v<ValueType, void>)
      if (weighted_) throw utils::ReaderException("Cannot read weights into ValueType void");
    std::ifstream fin(filename_, std::ios::binary);
    if (!fin.is_open()) throw utils::ReaderException("file does not exist!");
    std::string text((std::istreambuf_iterator<char>(fin)), std::istreambuf_iterator<char>());
    auto &dev = hip::Device::Get(ctx.device_id);
    hip::Staged<char> d_text(dev, text.data(), text.size() + 1);
    int64_t tokens = 0;
    dev.Check(sbx_text_count_tokens(dev.handle(), d_text.get(), (int64_t)text.size(), &tokens));
    const int fields = weighted_ ? 3 : 2;
    if (tokens % fields) throw utils::ReaderException("edge list does not hold whole edges");
    const int64_t entries = tokens / fields;
    const size_t cap = (size_t)entries * (read_undirected_ ? 2 : 1) + 1;
    IDType *row = (IDType *)dev.Malloc(cap * sizeof(IDType)), *col = (IDType *)dev.Malloc(cap * sizeof(IDType));
    void *val = nullptr;
    constexpr size_t vb = hip::ValueBytes<ValueType>();
    if (weighted_ && vb) val = dev.Malloc(cap * vb);
    int64_t dims[3] = {0, 0, 0};
    const unsigned flags = (remove_duplicates_ ? SBX_EDGES_REMOVE_DUPLICATES : 0u) |
                           (remove_self_edges_ ? SBX_EDGES_REMOVE_SELF : 0u) |
                           (read_undirected_ ? SBX_EDGES_UNDIRECTED : 0u) | (square_ ? SBX_EDGES_SQUARE : 0u);
    const int rc = sbx_edge_list_parse(dev.handle(), hip::IndexTag<IDType>(), hip::ValueTag<ValueType>(), d_text.get(),
                                       (int64_t)text.size(), entries, weighted_ ? 1 : 0, flags, (int64_t)cap, row, col, val,
                                       dims);
    if (rc != SBX_OK) {
      dev.Free(row);
      dev.Free(col);
      if (val) dev.Free(val);
      throw utils::ReaderException(std::string("edge list: ") + sbx_last_error(dev.handle()));
    }
    // already in (row, col) order: the constructor's check would find nothing to do
    return new format::HIPCOO<IDType, NNZType, ValueType>((IDType)dims[0], (IDType)dims[1], (NNZType)dims[2], row, col,
                                                         (ValueType *)val, ctx, format::kOwned, true);
  }

 private:
  std::string filename_;
  bool weighted_, remove_duplicates_, remove_self_edges_, read_undirected_, square_;
};

}  // namespace sparsebase::io
#endif
