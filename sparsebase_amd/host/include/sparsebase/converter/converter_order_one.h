// Array <-> HIPArray copies (reference: converter/converter_order_one_cuda.cu:8-42).
#ifndef SPARSEBASE_CONVERTER_CONVERTER_ORDER_ONE_H_
#define SPARSEBASE_CONVERTER_CONVERTER_ORDER_ONE_H_
#include "sparsebase/converter/converter.h"
#include "sparsebase/format/format_order_one.h"
#include "sparsebase/format/hip_formats.h"

namespace sparsebase::converter {

template <typename V>
format::Format *ArrayHIPArrayConditionalFunction(format::Format *source, context::Context *to) {
  auto *a = source->AsAbsolute<format::Array<V>>();
  const int did = static_cast<context::HIPContext *>(to)->device_id;
  auto &dev = hip::Device::Get(did);
  V *d = dev.Upload(a->get_vals(), (size_t)a->get_num_nnz());
  return new format::HIPArray<V>(a->get_num_nnz(), d, context::HIPContext(did), format::kOwned);
}
template <typename V>
format::Format *HIPArrayArrayConditionalFunction(format::Format *source, context::Context *) {
  auto *d = source->AsAbsolute<format::HIPArray<V>>();
  auto &dev = hip::Device::Get(d->get_hip_context()->device_id);
  V *h = dev.Download(d->get_vals(), (size_t)d->get_num_nnz());
  return new format::Array<V>(d->get_num_nnz(), h, format::kOwned);
}

template <typename ValueType>
class ConverterOrderOne : public ConverterImpl<ConverterOrderOne<ValueType>> {
 public:
  ConverterOrderOne() { ResetConverterOrderOne(); }
  Converter *Clone() const override { return new ConverterOrderOne(*this); }
  void Reset() override { ResetConverterOrderOne(); }
  void ResetConverterOrderOne() {
    const auto arr = format::Array<ValueType>::get_id_static(), darr = format::HIPArray<ValueType>::get_id_static();
    auto to_hip = [](context::Context *, context::Context *to) {
      return to->get_id() == context::HIPContext::get_id_static();
    };
    auto to_cpu = [](context::Context *, context::Context *to) {
      return to->get_id() == context::CPUContext::get_id_static();
    };
    for (bool move : {false, true}) {
      this->RegisterConversionFunction(arr, darr, ArrayHIPArrayConditionalFunction<ValueType>, to_hip, move);
      this->RegisterConversionFunction(darr, arr, HIPArrayArrayConditionalFunction<ValueType>, to_cpu, move);
    }
  }
};

}  // namespace sparsebase::converter
#endif
