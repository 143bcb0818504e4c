// RCMReorder (reference: reorder/rcm_reorder.h:14-45, rcm_reorder.cc:9-166).
// Parity holds for structurally symmetric, column-sorted patterns.
#ifndef SPARSEBASE_REORDER_RCM_REORDER_H_
#define SPARSEBASE_REORDER_RCM_REORDER_H_
#include "sparsebase/reorder/reorderer.h"

namespace sparsebase::reorder {

struct RCMReorderParams : utils::Parameters {};

template <typename IDType, typename NNZType, typename ValueType>
class RCMReorder : public Reorderer<IDType> {
 public:
  typedef RCMReorderParams ParamsType;
  RCMReorder() {
    this->RegisterFunction({format::CSR<IDType, NNZType, ValueType>::get_id_static()}, GetReorderCSR);
    this->RegisterFunction({format::HIPCSR<IDType, NNZType, ValueType>::get_id_static()}, GetReorderHIPCSR);
  }
  explicit RCMReorder(RCMReorderParams) : RCMReorder() {}
  // the order vector stays where sbx_rcm_reorder writes it (see Reorderer::GetReorderDevice)
  format::HIPArray<IDType> *GetReorderDevice(format::Format *format, context::HIPContext *context,
                                             bool convert_input) override {
    typedef format::HIPCSR<IDType, NNZType, ValueType> D;
    if (!format->template IsAbsolute<D>() || format->template AsAbsolute<D>()->get_hip_context()->device_id != context->device_id)
      return Reorderer<IDType>::GetReorderDevice(format, context, convert_input);
    auto v = detail::DeviceCsrView<IDType, NNZType, ValueType>::Borrow(format->template AsAbsolute<D>());
    IDType *d_inv = (IDType *)v.dev->Malloc((size_t)(v.n ? v.n : 1) * sizeof(IDType));
    const int rc = sbx_rcm_reorder(v.dev->handle(), hip::IndexTag<IDType, NNZType>(), v.n, v.nnz, v.row_ptr, v.col, d_inv, nullptr);
    if (rc != SBX_OK) {
      v.dev->Free(d_inv);
      v.dev->Check(rc);
    }
    return new format::HIPArray<IDType>((format::DimensionType)v.n, d_inv, *context, format::kOwned);
  }

 protected:
  static IDType *Run(detail::DeviceCsrView<IDType, NNZType, ValueType> v) {
    hip::Staged<IDType> d_inv(*v.dev, (size_t)v.n);
    const int rc = sbx_rcm_reorder(v.dev->handle(), hip::IndexTag<IDType, NNZType>(), v.n, v.nnz, v.row_ptr, v.col,
                                   d_inv.get(), nullptr);
    IDType *inv = nullptr;
    if (rc == SBX_OK) inv = v.dev->Download(d_inv.get(), (size_t)v.n);
    v.Release();
    v.dev->Check(rc);
    return inv;
  }
  static IDType *GetReorderCSR(std::vector<format::Format *> formats, utils::Parameters *) {
    auto *csr = formats[0]->AsAbsolute<format::CSR<IDType, NNZType, ValueType>>();
    return Run(detail::DeviceCsrView<IDType, NNZType, ValueType>::Stage(csr, false));
  }
  static IDType *GetReorderHIPCSR(std::vector<format::Format *> formats, utils::Parameters *) {
    auto *csr = formats[0]->AsAbsolute<format::HIPCSR<IDType, NNZType, ValueType>>();
    return Run(detail::DeviceCsrView<IDType, NNZType, ValueType>::Borrow(csr));
  }
};

}  // namespace sparsebase::reorder
#endif
