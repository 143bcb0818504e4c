// sparsebase/bases/iobase.h — the reading facade the examples use (reference:
// bases/iobase.h ReadMTXToCSR / ReadMTXToCOO); the Matrix Market entries are parsed on the GPU.
#ifndef SPARSEBASE_BASES_IOBASE_H_
#define SPARSEBASE_BASES_IOBASE_H_
#include "sparsebase/io/mtx_reader.h"

namespace sparsebase::bases {

class IOBase {
 public:
  template <typename IDType, typename NNZType, typename ValueType>
  static format::CSR<IDType, NNZType, ValueType> *ReadMTXToCSR(std::string filename, bool convert_index_to_zero = true) {
    io::MTXReader<IDType, NNZType, ValueType> reader(filename, convert_index_to_zero);
    return reader.ReadCSR();
  }
  template <typename IDType, typename NNZType, typename ValueType>
  static format::COO<IDType, NNZType, ValueType> *ReadMTXToCOO(std::string filename, bool convert_index_to_zero = true) {
    io::MTXReader<IDType, NNZType, ValueType> reader(filename, convert_index_to_zero);
    return reader.ReadCOO();
  }
};

}  // namespace sparsebase::bases
#endif
