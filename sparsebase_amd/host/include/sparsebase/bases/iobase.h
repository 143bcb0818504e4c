// sparsebase/bases/iobase.h — the reading facade the examples use (reference:
// bases/iobase.h:46-90 ReadMTXToCSR / ReadMTXToCOO, :161-195 ReadEdgeListToCSR / ReadEdgeListToCOO);
// the text files are parsed on the GPU.  Binary (SbFF) facade: :195-295.
#ifndef SPARSEBASE_BASES_IOBASE_H_
#define SPARSEBASE_BASES_IOBASE_H_
#include "sparsebase/io/binary_reader_order_one.h"
#include "sparsebase/io/binary_reader_order_two.h"
#include "sparsebase/io/binary_writer_order_one.h"
#include "sparsebase/io/binary_writer_order_two.h"
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
  template <typename IDType, typename NNZType, typename ValueType>
  static format::CSR<IDType, NNZType, ValueType> *ReadBinaryToCSR(std::string filename) {
    return io::BinaryReaderOrderTwo<IDType, NNZType, ValueType>(filename).ReadCSR();
  }
  template <typename IDType, typename NNZType, typename ValueType>
  static format::COO<IDType, NNZType, ValueType> *ReadBinaryToCOO(std::string filename) {
    return io::BinaryReaderOrderTwo<IDType, NNZType, ValueType>(filename).ReadCOO();
  }
  template <typename ValueType>
  static format::Array<ValueType> *ReadBinaryToArray(std::string filename) {
    return io::BinaryReaderOrderOne<ValueType>(filename).ReadArray();
  }
  template <typename IDType, typename NNZType, typename ValueType>
  static void WriteCOOToBinary(format::COO<IDType, NNZType, ValueType> *coo, std::string filename) {
    io::BinaryWriterOrderTwo<IDType, NNZType, ValueType>(filename).WriteCOO(coo);
  }
  template <typename IDType, typename NNZType, typename ValueType>
  static void WriteCSRToBinary(format::CSR<IDType, NNZType, ValueType> *csr, std::string filename) {
    io::BinaryWriterOrderTwo<IDType, NNZType, ValueType>(filename).WriteCSR(csr);
  }
  template <typename ValueType>
  static void WriteArrayToBinary(format::Array<ValueType> *array, std::string filename) {
    io::BinaryWriterOrderOne<ValueType>(filename).WriteArray(array);
  }
};

}  // namespace sparsebase::bases
#endif
