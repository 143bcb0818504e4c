// sparsebase/io/sparse_file_format.h — the SparseBase binary container "SbFF" (reference:
// io/sparse_file_format.h:75-327).  Layout, as the reference writes it:
//
//   [1024 B]  object header   {"array_count":A,"dimensions":[..],"endian":"little","name":"coo"}
//   A times:  [1024 B] array header {"array_size":S,"name":"row","type":"signed","type_size":4}
//             [S * type_size B] raw elements
//
// Headers are compact JSON with the keys in alphabetical order, padded with spaces to exactly
// 1024 bytes.  The reference builds them with a JSON library; the four keys of each header are
// flat (strings, unsigned integers, one integer list), so this file carries its own writer and
// a small scanner instead.  Arrays may come in any order (the reference iterates an
// unordered_map, :274-277) and are looked up by name.
//
// Reading is lazy: the headers are scanned once and an array's payload is read on demand into
// a buffer the caller supplies (SbffFile::ReadPayload), so the binary readers can upload a
// file's arrays to the device without building a host format first.
#ifndef SPARSEBASE_IO_SPARSE_FILE_FORMAT_H_
#define SPARSEBASE_IO_SPARSE_FILE_FORMAT_H_
#include <cstdint>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <type_traits>
#include <vector>

#include "sparsebase/utils/exception.h"

namespace sparsebase::io {

constexpr size_t kSbffHeaderBytes = 1024;

inline std::string SbffHostEndian() {
  const uint16_t probe = 1;
  unsigned char first;
  std::memcpy(&first, &probe, 1);
  return first == 1 ? "little" : "big";
}

// the element class the reference records for a C++ type (:96-108)
template <typename T>
std::string SbffTypeName() {
  static_assert(!std::is_same_v<T, void>, "void arrays cannot be stored");
  if constexpr (std::is_floating_point_v<T>) return "float";
  else if constexpr (std::is_signed_v<T>) return "signed";
  else return "unsigned";
}

// ---- flat JSON: values are a string, an unsigned integer or a list of integers ----
struct SbffValue {
  enum Kind { kString, kNumber, kList } kind = kNumber;
  std::string text;
  uint64_t number = 0;
  std::vector<long long> list;
};
using SbffFields = std::map<std::string, SbffValue>;  // std::map: alphabetical, like the reference's dump

inline std::string SbffQuote(const std::string &s) {
  std::string out = "\"";
  for (char c : s) {
    if (c == '"' || c == '\\') out.push_back('\\');
    out.push_back(c);
  }
  out.push_back('"');
  return out;
}

inline std::vector<char> SbffHeaderBlock(const SbffFields &fields) {
  std::string js = "{";
  bool first = true;
  for (const auto &[key, v] : fields) {
    if (!first) js += ",";
    first = false;
    js += SbffQuote(key) + ":";
    if (v.kind == SbffValue::kString) js += SbffQuote(v.text);
    else if (v.kind == SbffValue::kNumber) js += std::to_string(v.number);
    else {
      js += "[";
      for (size_t i = 0; i < v.list.size(); i++) js += (i ? "," : "") + std::to_string(v.list[i]);
      js += "]";
    }
  }
  js += "}";
  if (js.size() > kSbffHeaderBytes) throw utils::WriterException("Header size exceeds 1 KB");  // :148-150
  std::vector<char> block(js.begin(), js.end());
  block.resize(kSbffHeaderBytes, ' ');
  return block;
}

class SbffScanner {
 public:
  SbffScanner(const char *p, size_t n) : p_(p), end_(p + n) {}
  SbffFields Object() {
    SbffFields out;
    Expect('{');
    Blank();
    if (Peek() == '}') return out;
    while (true) {
      const std::string key = String();
      Expect(':');
      Blank();
      SbffValue v;
      if (Peek() == '"') {
        v.kind = SbffValue::kString;
        v.text = String();
      } else if (Peek() == '[') {
        v.kind = SbffValue::kList;
        ++p_;
        Blank();
        if (Peek() == ']') ++p_;
        else
          while (true) {
            v.list.push_back(Integer());
            Blank();
            if (Peek() == ',') { ++p_; continue; }
            Expect(']');
            break;
          }
      } else {
        v.kind = SbffValue::kNumber;
        const long long x = Integer();
        if (x < 0) Bad();
        v.number = (uint64_t)x;
      }
      out[key] = std::move(v);
      Blank();
      if (Peek() == ',') { ++p_; continue; }
      Expect('}');
      return out;
    }
  }

 private:
  [[noreturn]] static void Bad() { throw utils::ReaderException("Unknown SBFF ReadArray Error"); }  // :135, :323
  char Peek() const { return p_ < end_ ? *p_ : '\0'; }
  void Blank() { while (p_ < end_ && (*p_ == ' ' || *p_ == '\n' || *p_ == '\t' || *p_ == '\r')) ++p_; }
  void Expect(char c) {
    Blank();
    if (Peek() != c) Bad();
    ++p_;
  }
  std::string String() {
    Expect('"');
    std::string s;
    while (p_ < end_ && *p_ != '"') {
      if (*p_ == '\\' && p_ + 1 < end_) ++p_;
      s.push_back(*p_++);
    }
    if (p_ >= end_) Bad();
    ++p_;
    return s;
  }
  long long Integer() {
    Blank();
    bool neg = false;
    if (Peek() == '-') { neg = true; ++p_; }
    if (Peek() < '0' || Peek() > '9') Bad();
    long long x = 0;
    while (Peek() >= '0' && Peek() <= '9') x = x * 10 + (*p_++ - '0');
    return neg ? -x : x;
  }
  const char *p_, *end_;
};

inline const SbffValue &SbffField(const SbffFields &f, const char *key, SbffValue::Kind kind) {
  auto it = f.find(key);
  if (it == f.end() || it->second.kind != kind) throw utils::ReaderException("Unknown SBFF ReadArray Error");
  return it->second;
}

// One stored array: where its payload sits in the file and how it is typed.
struct SbffEntry {
  std::string name, type;
  size_t array_size = 0, type_size = 0;
  std::streamoff payload = 0;
  size_t bytes() const { return array_size * type_size; }
};

// A container opened for reading: the headers are scanned once, payloads are read on demand.
class SbffFile {
 public:
  explicit SbffFile(const std::string &filename) : in_(filename, std::ios::in | std::ios::binary) {
    if (!in_.is_open()) throw utils::ReaderException("file does not exist!");
    in_.seekg(0, std::ios::end);
    const std::streamoff file_bytes = in_.tellg();
    in_.seekg(0);
    const SbffFields head = Header();
    name_ = SbffField(head, "name", SbffValue::kString).text;
    endian_ = SbffField(head, "endian", SbffValue::kString).text;
    const uint64_t count = SbffField(head, "array_count", SbffValue::kNumber).number;
    for (long long d : SbffField(head, "dimensions", SbffValue::kList).list) dimensions_.push_back(d);
    std::streamoff at = (std::streamoff)kSbffHeaderBytes;
    for (uint64_t i = 0; i < count; i++) {
      in_.seekg(at);
      const SbffFields ah = Header();
      SbffEntry e;
      e.name = SbffField(ah, "name", SbffValue::kString).text;
      e.type = SbffField(ah, "type", SbffValue::kString).text;
      e.array_size = (size_t)SbffField(ah, "array_size", SbffValue::kNumber).number;
      e.type_size = (size_t)SbffField(ah, "type_size", SbffValue::kNumber).number;
      e.payload = at + (std::streamoff)kSbffHeaderBytes;
      at = e.payload + (std::streamoff)e.bytes();
      if (at > file_bytes) throw utils::ReaderException("SBFF file is truncated");
      entries_[e.name] = e;
    }
  }
  const std::string &name() const { return name_; }
  const std::string &endian() const { return endian_; }
  const std::vector<long long> &dimensions() const { return dimensions_; }
  size_t array_count() const { return entries_.size(); }
  bool Has(const std::string &array_name) const { return entries_.count(array_name) != 0; }

  // the entry of an array that must hold elements of type T (same checks and messages as :219-236)
  template <typename T>
  const SbffEntry &Typed(const std::string &array_name) const {
    auto it = entries_.find(array_name);
    if (it == entries_.end()) throw utils::ReaderException("Unknown SBFF ReadArray Error");
    const SbffEntry &e = it->second;
    if (e.type == "float" && !std::is_floating_point_v<T>)
      throw utils::ReaderException("Type mismatch, array type is float");
    if (e.type == "signed" && !std::is_signed_v<T>) throw utils::ReaderException("Type mismatch, array type is signed");
    if (e.type == "unsigned" && !std::is_unsigned_v<T>)
      throw utils::ReaderException("Type mismatch, array type is unsigned");
    if (e.type_size != sizeof(T))
      throw utils::ReaderException(std::string("Type mismatch, array type has size ") + std::to_string(e.type_size));
    return e;
  }
  // reads the first `count` elements of the array into dst, in host byte order
  template <typename T>
  void ReadPayload(const SbffEntry &e, T *dst, size_t count) {
    if (count > e.array_size) throw utils::ReaderException("SBFF array is shorter than the format needs");
    in_.seekg(e.payload);
    in_.read(reinterpret_cast<char *>(dst), (std::streamsize)(count * sizeof(T)));
    if ((size_t)in_.gcount() != count * sizeof(T)) throw utils::ReaderException("SBFF file is truncated");
    if (endian_ != SbffHostEndian()) {  // :240-245
      for (size_t i = 0; i < count; i++) {
        unsigned char *b = reinterpret_cast<unsigned char *>(dst + i);
        for (size_t k = 0; k < sizeof(T) / 2; k++) std::swap(b[k], b[sizeof(T) - 1 - k]);
      }
    }
  }
  template <typename T>
  T *ReadNew(const std::string &array_name, size_t *count_out = nullptr) {  // new T[array_size]; caller owns
    const SbffEntry &e = Typed<T>(array_name);
    T *p = new T[e.array_size ? e.array_size : 1];
    try {
      ReadPayload(e, p, e.array_size);
    } catch (...) {
      delete[] p;
      throw;
    }
    if (count_out) *count_out = e.array_size;
    return p;
  }

 private:
  SbffFields Header() {
    char block[kSbffHeaderBytes];
    in_.read(block, kSbffHeaderBytes);
    if ((size_t)in_.gcount() != kSbffHeaderBytes) throw utils::ReaderException("SBFF file is truncated");
    return SbffScanner(block, kSbffHeaderBytes).Object();
  }
  std::ifstream in_;
  std::string name_, endian_;
  std::vector<long long> dimensions_;
  std::map<std::string, SbffEntry> entries_;
};

// A container being written: arrays are recorded by pointer and streamed out by Write().
class SbffWriter {
 public:
  explicit SbffWriter(std::string object_name) : name_(std::move(object_name)) {}
  template <typename D>
  void AddDimensions(const std::vector<D> &dims) {
    for (auto d : dims) dimensions_.push_back((long long)d);
  }
  template <typename T>
  void AddArray(const std::string &array_name, const T *data, size_t count) {
    Pending p;
    p.fields["name"] = {SbffValue::kString, array_name, 0, {}};
    p.fields["type"] = {SbffValue::kString, SbffTypeName<T>(), 0, {}};
    p.fields["type_size"] = {SbffValue::kNumber, "", sizeof(T), {}};
    p.fields["array_size"] = {SbffValue::kNumber, "", count, {}};
    p.data = reinterpret_cast<const char *>(data);
    p.bytes = count * sizeof(T);
    arrays_.push_back(std::move(p));
  }
  void Write(const std::string &filename) const {
    std::ofstream out(filename, std::ios::out | std::ios::binary);
    if (!out.is_open()) throw utils::WriterException("cannot open " + filename + " for writing");
    SbffFields head;
    head["name"] = {SbffValue::kString, name_, 0, {}};
    head["array_count"] = {SbffValue::kNumber, "", arrays_.size(), {}};
    head["dimensions"] = {SbffValue::kList, "", 0, dimensions_};
    head["endian"] = {SbffValue::kString, SbffHostEndian(), 0, {}};
    out.write(SbffHeaderBlock(head).data(), kSbffHeaderBytes);
    for (const Pending &p : arrays_) {
      out.write(SbffHeaderBlock(p.fields).data(), kSbffHeaderBytes);
      out.write(p.data, (std::streamsize)p.bytes);
    }
    if (!out.good()) throw utils::WriterException("writing " + filename + " failed");
  }

 private:
  struct Pending {
    SbffFields fields;
    const char *data = nullptr;
    size_t bytes = 0;
  };
  std::string name_;
  std::vector<long long> dimensions_;
  std::vector<Pending> arrays_;
};

}  // namespace sparsebase::io
#endif
