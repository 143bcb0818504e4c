// sparsebase/format/format.h — Format base, ownership tags and the common
// implementation mix-in (reference: format/format.h:41-163, format_implementation.h).
#ifndef SPARSEBASE_FORMAT_FORMAT_H_
#define SPARSEBASE_FORMAT_FORMAT_H_
#include <algorithm>
#include <functional>
#include <memory>
#include <typeindex>
#include <vector>

#include "sparsebase/config.h"
#include "sparsebase/context/context.h"
#include "sparsebase/converter/converter.h"
#include "sparsebase/utils/exception.h"
#include "sparsebase/utils/utils.h"

namespace sparsebase::format {

enum Ownership { kNotOwned = 0, kOwned = 1 };
typedef unsigned long long DimensionType;

// host arrays handed to a format with kOwned are released with scalar delete in the
// reference (format.h:50-56); arrays come from new[] everywhere, so delete[] is used here
template <typename T>
struct Deleter {
  void operator()(T *p) const {
    if constexpr (!std::is_same_v<T, void>) delete[] p;
  }
};
template <typename T>
struct BlankDeleter {
  void operator()(T *) const {}
};

namespace detail {
template <typename T>
using OwnedPtr = std::unique_ptr<T, std::function<void(T *)>>;
template <typename T>
OwnedPtr<T> Hold(T *p, Ownership own) {
  if (own == kOwned) return OwnedPtr<T>(p, Deleter<T>());
  return OwnedPtr<T>(p, BlankDeleter<T>());
}
template <typename T>
T *CloneArray(const T *src, size_t count) {
  if constexpr (std::is_same_v<T, void>) {
    return nullptr;
  } else {
    if (!src) return nullptr;
    T *dst = new T[count];
    std::copy(src, src + count, dst);
    return dst;
  }
}
}  // namespace detail

class Format : public utils::Identifiable {
 public:
  ~Format() override = default;
  virtual Format *Clone() const = 0;
  virtual std::vector<DimensionType> get_dimensions() const = 0;
  virtual DimensionType get_num_nnz() const = 0;
  virtual DimensionType get_order() const = 0;
  virtual context::Context *get_context() const = 0;
  virtual std::shared_ptr<converter::Converter const> get_converter() const = 0;

  template <typename T>
  typename std::remove_pointer<T>::type *AsAbsolute() {
    using TBase = typename std::remove_pointer<T>::type;
    static_assert(std::is_base_of_v<Format, TBase>, "Cannot cast a non-Format class using AsAbsolute");
    if (this->get_id() == std::type_index(typeid(TBase))) return static_cast<TBase *>(this);
    throw utils::TypeException(get_name(), utils::demangle(typeid(TBase)));
  }
  template <typename T>
  bool IsAbsolute() {
    using TBase = typename std::remove_pointer<T>::type;
    return this->get_id() == std::type_index(typeid(TBase));
  }
};

class FormatImplementation : public Format {
 public:
  std::vector<DimensionType> get_dimensions() const override { return dimension_; }
  DimensionType get_num_nnz() const override { return nnz_; }
  DimensionType get_order() const override { return order_; }
  context::Context *get_context() const override { return context_.get().get(); }
  std::shared_ptr<converter::Converter const> get_converter() const override { return converter_; }
  void set_converter(std::shared_ptr<converter::Converter> c) { converter_ = std::move(c); }

 protected:
  DimensionType order_ = 0;
  std::vector<DimensionType> dimension_;
  DimensionType nnz_ = 0;
  utils::OnceSettable<std::unique_ptr<context::Context>> context_;
  std::shared_ptr<converter::Converter> converter_;
};

}  // namespace sparsebase::format

// ---- Converter members that need the complete Format type -------------------
namespace sparsebase::converter {

inline std::vector<format::Format *> Converter::ApplyConversionChain(const ConversionChain &chain,
                                                                     format::Format *input,
                                                                     bool clear_intermediate) {
  std::vector<format::Format *> out{input};
  if (!chain) return out;
  const auto &steps = std::get<0>(*chain);
  format::Format *cur = input;
  for (size_t i = 0; i < steps.size(); i++) {
    format::Format *next = std::get<0>(steps[i])(cur, std::get<1>(steps[i]));
    const bool last = (i + 1 == steps.size());
    if (!clear_intermediate || last) out.push_back(next);
    if (clear_intermediate && i != 0) delete cur;  // cur is an intermediate nobody else holds
    cur = next;
  }
  return out;
}

inline std::vector<format::Format *> Converter::ConvertCached(format::Format *source, std::type_index to_type,
                                                              std::vector<context::Context *> to_contexts,
                                                              bool is_move_conversion) const {
  if (to_type == source->get_id()) {
    for (auto *c : to_contexts)
      if (c->IsEquivalent(source->get_context())) return {source};  // same type, equivalent place
  }
  ConversionChain chain =
      GetConversionChain(source->get_id(), source->get_context(), to_type, to_contexts, is_move_conversion);
  if (!chain) throw utils::ConversionException(source->get_name(), utils::demangle(to_type));
  auto all = ApplyConversionChain(chain, source, false);
  return std::vector<format::Format *>(all.begin() + 1, all.end());
}

inline format::Format *Converter::Convert(format::Format *source, std::type_index to_type,
                                          std::vector<context::Context *> to_contexts,
                                          bool is_move_conversion) const {
  auto outs = ConvertCached(source, to_type, std::move(to_contexts), is_move_conversion);
  for (size_t i = 0; i + 1 < outs.size(); i++) delete outs[i];
  return outs.back();
}

template <typename FormatType>
FormatType *Converter::Convert(format::Format *source, context::Context *to_context, bool is_move_conversion) const {
  return this->Convert(source, FormatType::get_id_static(), to_context, is_move_conversion)
      ->template AsAbsolute<FormatType>();
}
template <typename FormatType>
FormatType *Converter::Convert(format::Format *source, std::vector<context::Context *> to_contexts,
                               bool is_move_conversion) const {
  return this->Convert(source, FormatType::get_id_static(), std::move(to_contexts), is_move_conversion)
      ->template AsAbsolute<FormatType>();
}

}  // namespace sparsebase::converter
#endif
