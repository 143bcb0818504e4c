// PermuteOrderOne (reference: permute/permute_order_one.h, permute_order_one.cc:18-37):
// out[order[i]] = vals[i].
#ifndef SPARSEBASE_PERMUTE_PERMUTE_ORDER_ONE_H_
#define SPARSEBASE_PERMUTE_PERMUTE_ORDER_ONE_H_
#include "sparsebase/format/format_order_one.h"
#include "sparsebase/permute/permuter.h"

namespace sparsebase::permute {

template <typename IDType>
struct PermuteOrderOneParams : utils::Parameters {
  IDType *order;
  explicit PermuteOrderOneParams(IDType *order) : order(order) {}
};

template <typename IDType, typename ValueType>
class PermuteOrderOne : public Permuter<format::FormatOrderOne<ValueType>, format::FormatOrderOne<ValueType>> {
 public:
  typedef PermuteOrderOneParams<IDType> ParamsType;
  explicit PermuteOrderOne(IDType *order) {
    this->RegisterFunction({format::Array<ValueType>::get_id_static()}, PermuteArray);
    this->RegisterFunction({format::HIPArray<ValueType>::get_id_static()}, PermuteHIPArray);
    this->params_ = std::make_unique<PermuteOrderOneParams<IDType>>(order);
  }
  explicit PermuteOrderOne(ParamsType params) : PermuteOrderOne(params.order) {}

 protected:
  static ValueType *Run(hip::Device &dev, int64_t n, const ValueType *d_vals, utils::Parameters *poly) {
    static_assert(sizeof(ValueType) == 4 || sizeof(ValueType) == 8, "PermuteOrderOne: 4- or 8-byte values");
    auto *order = static_cast<PermuteOrderOneParams<IDType> *>(poly)->order;
    hip::Staged<IDType> d_order(dev, order, (size_t)n);
    ValueType *d_out = (ValueType *)dev.Malloc((size_t)(n ? n : 1) * sizeof(ValueType));
    const int rc = sbx_permute_array(dev.handle(), hip::IndexTag<IDType>(), hip::ValueTag<ValueType>(), n,
                                     d_order.get(), d_vals, d_out);
    if (rc != SBX_OK) {
      dev.Free(d_out);
      dev.Check(rc);
    }
    return d_out;
  }
  static format::FormatOrderOne<ValueType> *PermuteArray(std::vector<format::Format *> formats,
                                                         utils::Parameters *params) {
    auto *a = formats[0]->AsAbsolute<format::Array<ValueType>>();
    auto &dev = hip::Device::Get(hip::DefaultDevice());
    const int64_t n = (int64_t)a->get_dimensions()[0];
    hip::Staged<ValueType> d_vals(dev, a->get_vals(), (size_t)n);
    ValueType *d_out = Run(dev, n, d_vals.get(), params);
    ValueType *out = dev.Download(d_out, (size_t)n);
    dev.Free(d_out);
    return new format::Array<ValueType>(n, out, format::kOwned);
  }
  static format::FormatOrderOne<ValueType> *PermuteHIPArray(std::vector<format::Format *> formats,
                                                            utils::Parameters *params) {
    auto *a = formats[0]->AsAbsolute<format::HIPArray<ValueType>>();
    const int did = a->get_hip_context()->device_id;
    auto &dev = hip::Device::Get(did);
    const int64_t n = (int64_t)a->get_dimensions()[0];
    ValueType *d_out = Run(dev, n, a->get_vals(), params);
    return new format::HIPArray<ValueType>(n, d_out, context::HIPContext(did), format::kOwned);
  }
};

}  // namespace sparsebase::permute
#endif
