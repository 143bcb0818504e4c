// FormatOrderOne / Array (reference: format/format_order_one.h, format/array.h).
#ifndef SPARSEBASE_FORMAT_FORMAT_ORDER_ONE_H_
#define SPARSEBASE_FORMAT_FORMAT_ORDER_ONE_H_
#include "sparsebase/context/cpu_context.h"
#include "sparsebase/converter/converter_store.h"
#include "sparsebase/format/format.h"

namespace sparsebase::converter {
template <typename ValueType>
class ConverterOrderOne;
}

namespace sparsebase::format {

template <typename ValueType>
class FormatOrderOne : public FormatImplementation {
 public:
  FormatOrderOne() {
    this->set_converter(
        converter::ConverterStore::GetStore().get_converter<converter::ConverterOrderOne<ValueType>>());
  }
  template <template <typename> class ToType>
  ToType<ValueType> *Convert(context::Context *to_context = nullptr, bool is_move_conversion = false) {
    context::Context *ctx = to_context == nullptr ? this->get_context() : to_context;
    return this->get_converter()
        ->Convert(this, ToType<ValueType>::get_id_static(), ctx, is_move_conversion)
        ->template AsAbsolute<ToType<ValueType>>();
  }
  template <template <typename> typename T>
  T<ValueType> *As() {
    if (this->get_id() == std::type_index(typeid(T<ValueType>))) return static_cast<T<ValueType> *>(this);
    throw utils::TypeException(this->get_name(), utils::demangle(typeid(T<ValueType>)));
  }
  template <template <typename> typename T>
  bool Is() {
    return this->get_id() == std::type_index(typeid(T<ValueType>));
  }
};

template <typename ValueType>
class Array : public utils::IdentifiableImplementation<Array<ValueType>, FormatOrderOne<ValueType>> {
 public:
  Array(DimensionType nnz, ValueType *vals, Ownership own = kNotOwned)
      : vals_(vals, BlankDeleter<ValueType>()) {
    if (own == kOwned) vals_ = detail::OwnedPtr<ValueType>(vals, Deleter<ValueType>());
    this->order_ = 1;
    this->dimension_ = {nnz};
    this->nnz_ = nnz;
    this->context_ = std::unique_ptr<context::Context>(new context::CPUContext);
  }
  Array(const Array &rhs) : vals_(nullptr, BlankDeleter<ValueType>()) {
    ValueType *v = new ValueType[rhs.nnz_];
    std::copy(rhs.get_vals(), rhs.get_vals() + rhs.nnz_, v);
    vals_ = detail::OwnedPtr<ValueType>(v, Deleter<ValueType>());
    this->order_ = 1;
    this->dimension_ = rhs.dimension_;
    this->nnz_ = rhs.nnz_;
    this->context_ = std::unique_ptr<context::Context>(new context::CPUContext);
  }
  Format *Clone() const override { return new Array(*this); }
  ValueType *get_vals() const { return vals_.get(); }
  ValueType *release_vals() {
    ValueType *raw = vals_.release();
    vals_ = detail::OwnedPtr<ValueType>(raw, BlankDeleter<ValueType>());
    return raw;
  }
  void set_vals(ValueType *p, Ownership own = kNotOwned) {
    if (own == kOwned) vals_ = detail::OwnedPtr<ValueType>(p, Deleter<ValueType>());
    else vals_ = detail::OwnedPtr<ValueType>(p, BlankDeleter<ValueType>());
  }
  virtual bool ValsIsOwned() { return vals_.get_deleter().target_type() != typeid(BlankDeleter<ValueType>); }

 protected:
  detail::OwnedPtr<ValueType> vals_;
};

}  // namespace sparsebase::format
#include "sparsebase/converter/converter_order_one.h"
#endif
