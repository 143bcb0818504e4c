// sparsebase/reorder/reorderer.h — Reorderer<IDType>: "format in, inverse permutation out"
// (reference: reorder/reorderer.h:36-119, reorderer.cc:21-51).  The returned array is
// a host `new IDType[n]`; the caller releases it with delete[].
#ifndef SPARSEBASE_REORDER_REORDERER_H_
#define SPARSEBASE_REORDER_REORDERER_H_
#include <tuple>
#include <vector>

#include "sparsebase/context/cpu_context.h"
#include "sparsebase/context/hip_context.h"
#include "sparsebase/format/coo.h"
#include "sparsebase/format/csr.h"
#include "sparsebase/format/hip_formats.h"
#include "sparsebase/utils/function_matcher_mixin.h"

namespace sparsebase::reorder {

template <typename IDType>
class Reorderer : public utils::FunctionMatcherMixin<IDType *> {
 public:
  IDType *GetReorder(format::Format *format, std::vector<context::Context *> contexts, bool convert_input) {
    return this->Execute(this->params_.get(), contexts, convert_input, format);
  }
  IDType *GetReorder(format::Format *format, utils::Parameters *params, std::vector<context::Context *> contexts,
                     bool convert_input) {
    return this->Execute(params, contexts, convert_input, format);
  }
  std::tuple<std::vector<std::vector<format::Format *>>, IDType *> GetReorderCached(
      format::Format *format, std::vector<context::Context *> contexts, bool convert_input) {
    return this->CachedExecute(this->params_.get(), contexts, convert_input, false, format);
  }
  std::tuple<std::vector<std::vector<format::Format *>>, IDType *> GetReorderCached(
      format::Format *format, utils::Parameters *params, std::vector<context::Context *> contexts,
      bool convert_input) {
    return this->CachedExecute(params, contexts, convert_input, false, format);
  }
  // Device-resident result (additive; the reference hands the order back as a host IDType*, bases/reorder_base.h:145-150,
  // and keeps device arrays in format/cuda_array_cuda.cuh): the inverse permutation stays in the HBM of `context`'s
  // device as an HIPArray<IDType> the caller owns, ready for PermuteOrderTwo / ReorderBase::Permute2D without a trip
  // over PCIe.  Reorderers whose result is born on the device (RCM, Degree) override this and return it as it is;
  // the default runs GetReorder and uploads the result.
  virtual format::HIPArray<IDType> *GetReorderDevice(format::Format *format, context::HIPContext *context,
                                                     bool convert_input) {
    IDType *host = GetReorder(format, {context}, convert_input);
    const size_t n = (size_t)format->get_dimensions()[0];
    auto &dev = hip::Device::Get(context->device_id);
    IDType *d = nullptr;
    try {
      d = dev.Upload(host, n ? n : 1);
    } catch (...) {
      delete[] host;
      throw;
    }
    delete[] host;
    return new format::HIPArray<IDType>((format::DimensionType)n, d, *context, format::kOwned);
  }
  virtual ~Reorderer() = default;
};

namespace detail {
// A CSR's index arrays on a device: borrowed from an HIPCSR, or staged from a host CSR.
template <typename I, typename N, typename V>
struct DeviceCsrView {
  hip::Device *dev = nullptr;
  I n = 0, m = 0;
  int64_t nnz = 0;
  N *row_ptr = nullptr;
  I *col = nullptr;
  V *vals = nullptr;
  bool staged = false;

  static DeviceCsrView Borrow(format::HIPCSR<I, N, V> *d) {
    DeviceCsrView v;
    v.dev = &d->device();
    v.n = (I)d->get_dimensions()[0];
    v.m = (I)d->get_dimensions()[1];
    v.nnz = (int64_t)d->get_num_nnz();
    v.row_ptr = d->get_row_ptr();
    v.col = d->get_col();
    v.vals = d->get_vals();
    return v;
  }
  static DeviceCsrView Stage(format::CSR<I, N, V> *h, bool with_vals) {
    DeviceCsrView v;
    v.dev = &hip::Device::Get(hip::DefaultDevice());
    v.n = (I)h->get_dimensions()[0];
    v.m = (I)h->get_dimensions()[1];
    v.nnz = (int64_t)h->get_num_nnz();
    v.row_ptr = v.dev->Upload(h->get_row_ptr(), (size_t)v.n + 1);
    v.col = v.dev->Upload(h->get_col(), (size_t)v.nnz);
    if constexpr (!std::is_same_v<V, void>)
      if (with_vals && h->get_vals()) v.vals = v.dev->Upload(h->get_vals(), (size_t)v.nnz);
    v.staged = true;
    return v;
  }
  void Release() {
    if (!staged) return;
    dev->Free(row_ptr);
    dev->Free(col);
    if (vals) dev->Free((void *)vals);
    staged = false;
  }
};
}  // namespace detail

}  // namespace sparsebase::reorder
#endif
