// FormatOrderTwo — common base of the 2-D formats (reference: format/format_order_two.h:22-178).
#ifndef SPARSEBASE_FORMAT_FORMAT_ORDER_TWO_H_
#define SPARSEBASE_FORMAT_FORMAT_ORDER_TWO_H_
#include "sparsebase/context/cpu_context.h"
#include "sparsebase/converter/converter_store.h"
#include "sparsebase/format/format.h"

namespace sparsebase::converter {
template <typename IDType, typename NNZType, typename ValueType>
class ConverterOrderTwo;
}

namespace sparsebase::format {

template <typename IDType, typename NNZType, typename ValueType>
class FormatOrderTwo : public FormatImplementation {
 public:
  FormatOrderTwo() {
    // every order-two format of one type tuple shares one conversion graph
    this->set_converter(converter::ConverterStore::GetStore()
                            .get_converter<converter::ConverterOrderTwo<IDType, NNZType, ValueType>>());
  }

  template <template <typename, typename, typename> class ToType>
  ToType<IDType, NNZType, ValueType> *Convert(context::Context *to_context = nullptr,
                                              bool is_move_conversion = false) {
    static_assert(std::is_base_of_v<FormatOrderTwo, ToType<IDType, NNZType, ValueType>>,
                  "T must be an order two format");
    context::Context *ctx = to_context == nullptr ? this->get_context() : to_context;
    return this->get_converter()
        ->Convert(this, ToType<IDType, NNZType, ValueType>::get_id_static(), ctx, is_move_conversion)
        ->template AsAbsolute<ToType<IDType, NNZType, ValueType>>();
  }
  template <template <typename, typename, typename> class ToType>
  ToType<IDType, NNZType, ValueType> *Convert(const std::vector<context::Context *> &to_contexts,
                                              bool is_move_conversion = false) {
    std::vector<context::Context *> ctxs = to_contexts;
    if (ctxs.empty()) ctxs.push_back(this->get_context());
    return this->get_converter()
        ->Convert(this, ToType<IDType, NNZType, ValueType>::get_id_static(), ctxs, is_move_conversion)
        ->template AsAbsolute<ToType<IDType, NNZType, ValueType>>();
  }

  template <template <typename, typename, typename> typename T>
  T<IDType, NNZType, ValueType> *As() {
    using TBase = T<IDType, NNZType, ValueType>;
    static_assert(std::is_base_of_v<FormatOrderTwo, TBase>, "Cannot cast to a non-FormatOrderTwo class");
    if (this->get_id() == std::type_index(typeid(TBase))) return static_cast<TBase *>(this);
    throw utils::TypeException(this->get_name(), utils::demangle(typeid(TBase)));
  }
  template <template <typename, typename, typename> typename T>
  bool Is() {
    return this->get_id() == std::type_index(typeid(T<IDType, NNZType, ValueType>));
  }
};

}  // namespace sparsebase::format
#endif
