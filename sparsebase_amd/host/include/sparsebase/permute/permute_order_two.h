// PermuteOrderTwo (reference: permute/permute_order_two.h:16-50, permute_order_two.cc:8-79).
// row_order / col_order are inverse permutations (order[old] = new) on the HOST, or
// nullptr for identity on that axis.  The result's rows are sorted exactly when the
// reference's CSR constructor would sort them.  {CSR} stages through the default
// device and returns a host CSR; {HIPCSR} permutes in HBM and returns an HIPCSR.
// Unlike the reference (:76-77) the returned format OWNS its arrays.
#ifndef SPARSEBASE_PERMUTE_PERMUTE_ORDER_TWO_H_
#define SPARSEBASE_PERMUTE_PERMUTE_ORDER_TWO_H_
#include "sparsebase/permute/permuter.h"

namespace sparsebase::permute {

template <typename IDType>
struct PermuteOrderTwoParams : utils::Parameters {
  IDType *row_order;
  IDType *col_order;
  explicit PermuteOrderTwoParams(IDType *r_order, IDType *c_order) : row_order(r_order), col_order(c_order) {}
};

template <typename IDType, typename NNZType, typename ValueType>
class PermuteOrderTwo : public Permuter<format::FormatOrderTwo<IDType, NNZType, ValueType>,
                                        format::FormatOrderTwo<IDType, NNZType, ValueType>> {
  typedef format::FormatOrderTwo<IDType, NNZType, ValueType> Out;

 public:
  typedef PermuteOrderTwoParams<IDType> ParamsType;
  PermuteOrderTwo(IDType *row_order, IDType *col_order) {
    this->RegisterFunction({format::CSR<IDType, NNZType, ValueType>::get_id_static()}, PermuteOrderTwoCSR);
    this->RegisterFunction({format::HIPCSR<IDType, NNZType, ValueType>::get_id_static()}, PermuteOrderTwoHIPCSR);
    this->params_ = std::make_unique<PermuteOrderTwoParams<IDType>>(row_order, col_order);
  }
  // the reference's params constructor builds a temporary and registers nothing
  // (permute_order_two.cc:17-20); here it delegates properly
  explicit PermuteOrderTwo(PermuteOrderTwoParams<IDType> params)
      : PermuteOrderTwo(params.row_order, params.col_order) {}

 protected:
  struct DeviceResult {
    NNZType *row_ptr;
    IDType *col;
    ValueType *vals;
  };
  static DeviceResult Run(reorder::detail::DeviceCsrView<IDType, NNZType, ValueType> &v, utils::Parameters *poly) {
    auto *params = static_cast<PermuteOrderTwoParams<IDType> *>(poly);
    auto &dev = *v.dev;
    IDType *d_ro = params->row_order ? dev.Upload(params->row_order, (size_t)v.n) : nullptr;
    IDType *d_co = nullptr;
    if (params->col_order) d_co = (params->col_order == params->row_order && v.n == v.m)
                                      ? d_ro
                                      : dev.Upload(params->col_order, (size_t)v.m);
    DeviceResult out;
    out.row_ptr = (NNZType *)dev.Malloc(((size_t)v.n + 1) * sizeof(NNZType));
    out.col = (IDType *)dev.Malloc((size_t)(v.nnz ? v.nnz : 1) * sizeof(IDType));
    out.vals = nullptr;
    if (v.vals) out.vals = (ValueType *)dev.Malloc((size_t)v.nnz * hip::ValueBytes<ValueType>());
    const int rc = sbx_permute_csr(dev.handle(), hip::IndexTag<IDType>(), hip::ValueTag<ValueType>(), v.n, v.m, v.nnz,
                                   v.row_ptr, v.col, v.vals, d_ro, d_co, out.row_ptr, out.col, out.vals);
    if (d_co && d_co != d_ro) dev.Free(d_co);
    if (d_ro) dev.Free(d_ro);
    if (rc != SBX_OK) {
      dev.Free(out.row_ptr);
      dev.Free(out.col);
      if (out.vals) dev.Free((void *)out.vals);
      v.Release();
      dev.Check(rc);
    }
    return out;
  }
  static Out *PermuteOrderTwoCSR(std::vector<format::Format *> formats, utils::Parameters *params) {
    auto *csr = formats[0]->AsAbsolute<format::CSR<IDType, NNZType, ValueType>>();
    auto v = reorder::detail::DeviceCsrView<IDType, NNZType, ValueType>::Stage(csr, true);
    DeviceResult r = Run(v, params);
    auto &dev = *v.dev;
    NNZType *rp = dev.Download(r.row_ptr, (size_t)v.n + 1);
    IDType *col = dev.Download(r.col, (size_t)v.nnz);
    ValueType *vals = nullptr;
    if constexpr (!std::is_same_v<ValueType, void>)
      if (r.vals) vals = dev.Download(r.vals, (size_t)v.nnz);
    dev.Free(r.row_ptr);
    dev.Free(r.col);
    if (r.vals) dev.Free((void *)r.vals);
    v.Release();
    return new format::CSR<IDType, NNZType, ValueType>(v.n, v.m, rp, col, vals, format::kOwned, true);
  }
  static Out *PermuteOrderTwoHIPCSR(std::vector<format::Format *> formats, utils::Parameters *params) {
    auto *csr = formats[0]->AsAbsolute<format::HIPCSR<IDType, NNZType, ValueType>>();
    auto v = reorder::detail::DeviceCsrView<IDType, NNZType, ValueType>::Borrow(csr);
    DeviceResult r = Run(v, params);
    return new format::HIPCSR<IDType, NNZType, ValueType>(v.n, v.m, (NNZType)v.nnz, r.row_ptr, r.col, r.vals,
                                                          context::HIPContext(v.dev->id()), format::kOwned, true);
  }
};

}  // namespace sparsebase::permute
#endif
