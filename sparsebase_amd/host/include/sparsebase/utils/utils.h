// Type identity, once-settable members and small helpers of the host layer
// (counterpart of the reference's utils/utils.h:130-199).
#ifndef SPARSEBASE_UTILS_UTILS_H_
#define SPARSEBASE_UTILS_UTILS_H_
#include <cxxabi.h>

#include <cstdlib>
#include <limits>
#include <string>
#include <type_traits>
#include <typeindex>
#include <typeinfo>
#include <vector>

#include "sparsebase/config.h"
#include "sparsebase/utils/exception.h"

namespace sparsebase::utils {

typedef unsigned int CostType;

template <typename T>
inline constexpr bool always_false = false;

inline std::string demangle(const std::string &name) {
  int status = 0;
  char *res = abi::__cxa_demangle(name.c_str(), nullptr, nullptr, &status);
  std::string out = (status == 0 && res) ? std::string(res) : name;
  std::free(res);
  return out;
}
inline std::string demangle(std::type_index type) { return demangle(std::string(type.name())); }

struct TypeIndexVectorHash {
  std::size_t operator()(const std::vector<std::type_index> &v) const {
    std::size_t h = 0x9e3779b97f4a7c15ull;
    for (const auto &t : v) h ^= t.hash_code() + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
    return h;
  }
};

// value that may be assigned exactly once (Format::context_)
template <typename T>
class OnceSettable {
 public:
  OnceSettable() : set_(false) {}
  OnceSettable(const OnceSettable &) = delete;
  OnceSettable(OnceSettable &&) = delete;
  OnceSettable &operator=(T &&v) {
    if (set_) throw AttemptToReset<T>();
    data_ = std::move(v);
    set_ = true;
    return *this;
  }
  const T &get() const { return data_; }

 private:
  T data_;
  bool set_;
};

class Identifiable {
 public:
  virtual ~Identifiable() = default;
  virtual std::type_index get_id() const = 0;
  virtual std::string get_name() const = 0;
};

template <typename IdentifiableType, typename Base>
class IdentifiableImplementation : public Base {
 public:
  std::type_index get_id() const override { return typeid(IdentifiableType); }
  std::string get_name() const override { return demangle(get_id()); }
  static std::type_index get_id_static() { return typeid(IdentifiableType); }
  static std::string get_name_static() { return demangle(get_id_static()); }
};

// checked element-wise array conversion (utils.h:130-149)
template <typename ToType, typename FromType, typename SizeType>
ToType *ConvertArrayType(FromType *from, SizeType size) {
  if constexpr (std::is_same_v<ToType, void> || std::is_same_v<FromType, void>) {
    return nullptr;
  } else {
    if (from == nullptr) return nullptr;
    auto *to = new ToType[size];
    for (SizeType i = 0; i < size; i++) {
      to[i] = static_cast<ToType>(from[i]);
      if (static_cast<FromType>(to[i]) != from[i] || ((from[i] < FromType(0)) != (to[i] < ToType(0)))) {
        delete[] to;
        throw TypeException("Could not convert array from type " + demangle(typeid(FromType)) + " to type " +
                            demangle(typeid(ToType)) + ". Overflow detected");
      }
    }
    return to;
  }
}

}  // namespace sparsebase::utils
#endif
