// sbff_tool — looks into and writes SparseBase binary (SbFF) containers; needs no GPU.
//
//   sbff_tool dump  <file.bin>             prints name, dimensions and every array as text
//   sbff_tool write <file.bin> <desc.txt>  writes the container a text description names
//
// Description (what `dump` prints, minus the array headers' type columns):
//   kind coo|csr|array
//   dims <n> [<m>]
//   row_ptr|row|col <int> ...       (one line per array; `vals` / `array` hold floats)
// `write` goes through BinaryWriterOrderTwo / BinaryWriterOrderOne with <int,int,float> formats
// whose constructors are told not to sort (that is the only step of the host formats that runs
// on the device), so the tests can exchange files with the reference on a machine without a GPU.
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>

#include "sparsebase/sparsebase.h"

using namespace sparsebase;

template <typename T>
static void print_array(io::SbffFile &f, const io::SbffEntry &e, const char *fmt) {
  std::vector<T> v(e.array_size ? e.array_size : 1);
  f.ReadPayload(e, v.data(), e.array_size);
  for (size_t i = 0; i < e.array_size; i++) {
    std::printf(" ");
    std::printf(fmt, v[i]);
  }
}

static int dump(const std::string &path) {
  io::SbffFile f(path);
  std::printf("kind %s\ndims", f.name().c_str());
  for (long long d : f.dimensions()) std::printf(" %lld", d);
  std::printf("\nendian %s\narrays %zu\n", f.endian().c_str(), f.array_count());
  for (const char *name : {"row_ptr", "row", "col", "vals", "array"}) {
    if (!f.Has(name)) continue;
    std::printf("%s", name);
    const bool real = std::string(name) == "vals" || std::string(name) == "array";
    if (real) {
      const io::SbffEntry &e = f.Typed<float>(name);
      std::printf(" [%s %zu x%zu]", e.type.c_str(), e.array_size, e.type_size);
      print_array<float>(f, e, "%.9g");
    } else {
      const io::SbffEntry &e = f.Typed<int>(name);
      std::printf(" [%s %zu x%zu]", e.type.c_str(), e.array_size, e.type_size);
      print_array<int>(f, e, "%d");
    }
    std::printf("\n");
  }
  return 0;
}

static int write(const std::string &path, const std::string &desc) {
  std::ifstream in(desc);
  if (!in.is_open()) throw utils::ReaderException("cannot open " + desc);
  std::string kind, line;
  std::vector<int> dims, row_ptr, row, col;
  std::vector<float> vals;
  bool have_vals = false;
  while (std::getline(in, line)) {
    std::istringstream ls(line);
    std::string key;
    if (!(ls >> key)) continue;
    if (key == "kind") ls >> kind;
    else if (key == "dims") for (int d; ls >> d;) dims.push_back(d);
    else if (key == "row_ptr") for (int x; ls >> x;) row_ptr.push_back(x);
    else if (key == "row") for (int x; ls >> x;) row.push_back(x);
    else if (key == "col") for (int x; ls >> x;) col.push_back(x);
    else if (key == "vals" || key == "array") {
      have_vals = true;
      for (float x; ls >> x;) vals.push_back(x);
    }
  }
  if (kind == "coo") {
    format::COO<int, int, float> coo(dims.at(0), dims.at(1), (int)row.size(), row.data(), col.data(),
                                     have_vals ? vals.data() : nullptr, format::kNotOwned, true);
    bases::IOBase::WriteCOOToBinary(&coo, path);
  } else if (kind == "csr") {
    format::CSR<int, int, float> csr(dims.at(0), dims.at(1), row_ptr.data(), col.data(), have_vals ? vals.data() : nullptr,
                                     format::kNotOwned, true);
    bases::IOBase::WriteCSRToBinary(&csr, path);
  } else if (kind == "array") {
    format::Array<float> arr((format::DimensionType)vals.size(), vals.data(), format::kNotOwned);
    bases::IOBase::WriteArrayToBinary(&arr, path);
  } else {
    throw utils::ReaderException("unknown kind " + kind);
  }
  return 0;
}

int main(int argc, char **argv) {
  try {
    if (argc == 3 && std::string(argv[1]) == "dump") return dump(argv[2]);
    if (argc == 4 && std::string(argv[1]) == "write") return write(argv[2], argv[3]);
  } catch (std::exception &e) {
    std::fprintf(stderr, "sbff_tool: %s\n", e.what());
    return 2;
  }
  std::fprintf(stderr, "usage: sbff_tool dump <file.bin> | sbff_tool write <file.bin> <description.txt>\n");
  return 1;
}
