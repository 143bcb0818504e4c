// DegreeReorder (reference: reorder/degree_reorder.h:15-40, degree_reorder.cc:9-62):
// rows ordered by (degree ascending, id descending); !ascending reverses the order.
// Both registered implementations run sbx_degree_reorder on the GPU: {HIPCSR} in
// place in HBM, {CSR} by staging the host arrays through the default device.
#ifndef SPARSEBASE_REORDER_DEGREE_REORDER_H_
#define SPARSEBASE_REORDER_DEGREE_REORDER_H_
#include "sparsebase/reorder/reorderer.h"

namespace sparsebase::reorder {

struct DegreeReorderParams : utils::Parameters {
  bool ascending;
  DegreeReorderParams(bool ascending) : ascending(ascending) {}
};

template <typename IDType, typename NNZType, typename ValueType>
class DegreeReorder : public Reorderer<IDType> {
 public:
  typedef DegreeReorderParams ParamsType;
  explicit DegreeReorder(bool ascending) {
    this->RegisterFunction({format::CSR<IDType, NNZType, ValueType>::get_id_static()}, CalculateReorderCSR);
    this->RegisterFunction({format::HIPCSR<IDType, NNZType, ValueType>::get_id_static()}, CalculateReorderHIPCSR);
    this->params_ = std::make_unique<DegreeReorderParams>(ascending);
  }
  explicit DegreeReorder(DegreeReorderParams params) : DegreeReorder(params.ascending) {}
  // the order vector stays where sbx_degree_reorder writes it (see Reorderer::GetReorderDevice)
  format::HIPArray<IDType> *GetReorderDevice(format::Format *format, context::HIPContext *context,
                                             bool convert_input) override {
    typedef format::HIPCSR<IDType, NNZType, ValueType> D;
    if (!format->template IsAbsolute<D>() || format->template AsAbsolute<D>()->get_hip_context()->device_id != context->device_id)
      return Reorderer<IDType>::GetReorderDevice(format, context, convert_input);
    auto v = detail::DeviceCsrView<IDType, NNZType, ValueType>::Borrow(format->template AsAbsolute<D>());
    const bool ascending = static_cast<DegreeReorderParams *>(this->params_.get())->ascending;
    IDType *d_inv = (IDType *)v.dev->Malloc((size_t)(v.n ? v.n : 1) * sizeof(IDType));
    const int rc = sbx_degree_reorder(v.dev->handle(), hip::IndexTag<IDType, NNZType>(), v.n, v.row_ptr, ascending ? 1 : 0, d_inv);
    if (rc != SBX_OK) {
      v.dev->Free(d_inv);
      v.dev->Check(rc);
    }
    return new format::HIPArray<IDType>((format::DimensionType)v.n, d_inv, *context, format::kOwned);
  }

 protected:
  static IDType *Run(detail::DeviceCsrView<IDType, NNZType, ValueType> v, utils::Parameters *params) {
    const bool ascending = static_cast<DegreeReorderParams *>(params)->ascending;
    hip::Staged<IDType> d_inv(*v.dev, (size_t)v.n);
    const int rc =
        sbx_degree_reorder(v.dev->handle(), hip::IndexTag<IDType, NNZType>(), v.n, v.row_ptr, ascending ? 1 : 0, d_inv.get());
    IDType *inv = nullptr;
    if (rc == SBX_OK) inv = v.dev->Download(d_inv.get(), (size_t)v.n);
    v.Release();
    v.dev->Check(rc);
    return inv;
  }
  static IDType *CalculateReorderCSR(std::vector<format::Format *> formats, utils::Parameters *params) {
    auto *csr = formats[0]->AsAbsolute<format::CSR<IDType, NNZType, ValueType>>();
    return Run(detail::DeviceCsrView<IDType, NNZType, ValueType>::Stage(csr, false), params);
  }
  static IDType *CalculateReorderHIPCSR(std::vector<format::Format *> formats, utils::Parameters *params) {
    auto *csr = formats[0]->AsAbsolute<format::HIPCSR<IDType, NNZType, ValueType>>();
    return Run(detail::DeviceCsrView<IDType, NNZType, ValueType>::Borrow(csr), params);
  }
};

}  // namespace sparsebase::reorder
#endif
