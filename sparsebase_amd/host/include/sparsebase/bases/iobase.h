// sparsebase/bases/iobase.h — the reading facade the examples use (reference:
// bases/iobase.h:46-90 ReadMTXToCSR / ReadMTXToCOO, :161-195 ReadEdgeListToCSR / ReadEdgeListToCOO);
// the files are parsed on the GPU.
#ifndef SPARSEBASE_BASES_IOBASE_H_
#define SPARSEBASE_BASES_IOBASE_H_
#include "sparsebase/io/edge_list_reader.h"
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
  // like the reference, the facade always removes duplicate edges (iobase.h:165-166)
  template <typename IDType, typename NNZType, typename ValueType>
  static format::CSR<IDType, NNZType, ValueType> *ReadEdgeListToCSR(std::string filename, bool weighted = false,
                                                                    bool remove_self_edges = false,
                                                                    bool read_undirected = true, bool square = false) {
    io::EdgeListReader<IDType, NNZType, ValueType> reader(filename, weighted, true, remove_self_edges, read_undirected, square);
    return reader.ReadCSR();
  }
  template <typename IDType, typename NNZType, typename ValueType>
  static format::COO<IDType, NNZType, ValueType> *ReadEdgeListToCOO(std::string filename, bool weighted = false,
                                                                    bool remove_self_edges = false,
                                                                    bool read_undirected = true, bool square = false) {
    io::EdgeListReader<IDType, NNZType, ValueType> reader(filename, weighted, true, remove_self_edges, read_undirected, square);
    return reader.ReadCOO();
  }
};

}  // namespace sparsebase::bases
#endif
