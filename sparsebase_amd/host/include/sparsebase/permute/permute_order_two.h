// PermuteOrderTwo (reference: permute/permute_order_two.h:16-50, permute_order_two.cc:8-79).
// row_order / col_order are inverse permutations (order[old] = new) on the HOST, or
// nullptr for identity on that axis.  The result's rows are sorted exactly when the
// reference's CSR constructor would sort them.  {CSR} stages through the default
// device and returns a host CSR; {HIPCSR} permutes in HBM and returns an HIPCSR.
// Unlike the reference (:76-77) the returned format OWNS its arrays.
#ifndef SPARSEBASE_PERMUTE_PERMUTE_ORDER_TWO_H_
#define SPARSEBASE_PERMUTE_PERMUTE_ORDER_TWO_H_
#include <vector>

#include "sparsebase/context/hip_communicator.h"
#include "sparsebase/permute/permuter.h"

namespace sparsebase::permute {

// One rank's part of a row-sharded permuted CSR: the WHOLE row_ptr (n + 1 entries) plus the col / vals
// of the rank's own new rows [row_begin, row_end), all in the rank's HBM; entry_offsets[r] is where rank r's entries
// start in the global entry space (entry_offsets[world] = nnz).
template <typename IDType, typename NNZType, typename ValueType>
struct ShardedHIPCSR {
  IDType n = 0, m = 0;
  int64_t row_begin = 0, row_end = 0;
  std::vector<int64_t> entry_offsets;
  format::HIPArray<NNZType> *row_ptr = nullptr;
  format::HIPArray<IDType> *col = nullptr;
  format::HIPArray<typename std::conditional<std::is_same<ValueType, void>::value, char, ValueType>::type> *vals = nullptr;
  int64_t local_nnz(int rank) const { return entry_offsets[rank + 1] - entry_offsets[rank]; }
  ~ShardedHIPCSR() {
    delete row_ptr;
    delete col;
    delete vals;
  }
};

template <typename IDType>
struct PermuteOrderTwoParams : utils::Parameters {
  IDType *row_order;
  IDType *col_order;
  // the order vectors are device arrays on device `orders_device` (>= 0) instead of host arrays: set by the
  // PermuteOrderTwo constructor that takes HIPArrays (additive; the reference trades host arrays only)
  int orders_device = -1;
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
      : PermuteOrderTwo(params.row_order, params.col_order) {
    static_cast<PermuteOrderTwoParams<IDType> *>(this->params_.get())->orders_device = params.orders_device;
  }
  // Order vectors that already live on the device (what Reorderer::GetReorderDevice returns): nothing is uploaded, and
  // an HIPCSR on the same device is permuted with them in place.  nullptr: identity on that axis.  Both arrays must be
  // on one device; they stay the caller's.
  PermuteOrderTwo(format::HIPArray<IDType> *row_order, format::HIPArray<IDType> *col_order)
      : PermuteOrderTwo(row_order ? row_order->get_vals() : nullptr, col_order ? col_order->get_vals() : nullptr) {
    auto *a = row_order ? row_order : col_order;
    if (row_order && col_order && row_order->get_hip_context()->device_id != col_order->get_hip_context()->device_id)
      throw utils::HIPDeviceException("PermuteOrderTwo: the two order vectors live on different devices");
    static_cast<PermuteOrderTwoParams<IDType> *>(this->params_.get())->orders_device =
        a ? a->get_hip_context()->device_id : -1;
  }

 protected:
  struct DeviceResult {
    NNZType *row_ptr;
    IDType *col;
    ValueType *vals;
  };
  static DeviceResult Run(reorder::detail::DeviceCsrView<IDType, NNZType, ValueType> &v, utils::Parameters *poly) {
    auto *params = static_cast<PermuteOrderTwoParams<IDType> *>(poly);
    auto &dev = *v.dev;
    const bool on_device = params->orders_device >= 0 && (params->row_order || params->col_order);
    if (on_device && params->orders_device != dev.id()) {
      v.Release();
      throw utils::HIPDeviceException("PermuteOrderTwo: the order vectors live on device " +
                                      std::to_string(params->orders_device) + ", the matrix on device " +
                                      std::to_string(dev.id()));
    }
    IDType *d_ro = nullptr, *d_co = nullptr;
    if (on_device) {
      d_ro = params->row_order, d_co = params->col_order;
    } else {
      d_ro = params->row_order ? dev.Upload(params->row_order, (size_t)v.n) : nullptr;
      if (params->col_order) d_co = (params->col_order == params->row_order && v.n == v.m)
                                        ? d_ro
                                        : dev.Upload(params->col_order, (size_t)v.m);
    }
    DeviceResult out;
    out.row_ptr = (NNZType *)dev.Malloc(((size_t)v.n + 1) * sizeof(NNZType));
    out.col = (IDType *)dev.Malloc((size_t)(v.nnz ? v.nnz : 1) * sizeof(IDType));
    out.vals = nullptr;
    if (v.vals) out.vals = (ValueType *)dev.Malloc((size_t)v.nnz * hip::ValueBytes<ValueType>());
    const int rc = sbx_permute_csr(dev.handle(), hip::IndexTag<IDType, NNZType>(), hip::ValueTag<ValueType>(), v.n, v.m, v.nnz,
                                   v.row_ptr, v.col, v.vals, d_ro, d_co, out.row_ptr, out.col, out.vals);
    if (!on_device) {
      if (d_co && d_co != d_ro) dev.Free(d_co);
      if (d_ro) dev.Free(d_ro);
    }
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

 public:
  // The sharded form of GetPermutation (one call per rank, every rank holding the same HIPCSR and order vectors):
  // this rank permutes its own new-row range and the ranks all-gather row_ptr (sbx_permute_csr_sharded).
  // row_splits: world + 1 new-row boundaries, or nullptr: ranges of equal ENTRY counts, computed on the device
  // (sbx_balanced_row_splits; SURVEY §8e: balance by nnz, not rows).  Caller owns the result.
  ShardedHIPCSR<IDType, NNZType, ValueType> *GetPermutationSharded(format::HIPCSR<IDType, NNZType, ValueType> *csr,
                                                                    context::HIPCommunicator &comm,
                                                                    const int64_t *row_splits = nullptr) {
    auto *params = static_cast<PermuteOrderTwoParams<IDType> *>(this->params_.get());
    auto v = reorder::detail::DeviceCsrView<IDType, NNZType, ValueType>::Borrow(csr);
    auto &dev = *v.dev;
    const int rank = comm.rank(), world = comm.world();
    std::vector<int64_t> splits(world + 1);
    IDType *d_ro = params->row_order ? dev.Upload(params->row_order, (size_t)v.n) : nullptr;
    if (row_splits) {
      for (int r = 0; r <= world; r++) splits[r] = row_splits[r];
    } else {
      const int rs = sbx_balanced_row_splits(dev.handle(), hip::IndexTag<IDType, NNZType>(), v.n, v.row_ptr, d_ro, world, splits.data());
      if (rs != SBX_OK) {
        if (d_ro) dev.Free(d_ro);
        dev.Check(rs);
      }
    }
    IDType *d_co = nullptr;
    if (params->col_order) d_co = (params->col_order == params->row_order && v.n == v.m)
                                      ? d_ro
                                      : dev.Upload(params->col_order, (size_t)v.m);
    int64_t cap = 0;
    int rc = sbx_permute_csr_rows_nnz(dev.handle(), hip::IndexTag<IDType, NNZType>(), v.n, v.row_ptr, d_ro, splits[rank],
                                      splits[rank + 1], &cap);
    typedef typename std::conditional<std::is_same<ValueType, void>::value, char, ValueType>::type Stored;
    NNZType *rp = nullptr;
    IDType *col = nullptr;
    Stored *vals = nullptr;
    auto *out = new ShardedHIPCSR<IDType, NNZType, ValueType>();
    out->entry_offsets.assign(world + 1, 0);
    if (rc == SBX_OK) {
      rp = (NNZType *)dev.Malloc(((size_t)v.n + 1) * sizeof(NNZType));
      col = (IDType *)dev.Malloc((size_t)(cap ? cap : 1) * sizeof(IDType));
      if (v.vals) vals = (Stored *)dev.Malloc((size_t)(cap ? cap : 1) * hip::ValueBytes<ValueType>());
      rc = sbx_permute_csr_sharded(dev.handle(), comm.handle(), hip::IndexTag<IDType, NNZType>(), hip::ValueTag<ValueType>(), v.n,
                                   v.m, v.nnz, v.row_ptr, v.col, v.vals, d_ro, d_co, splits.data(), rp, col, vals, cap,
                                   out->entry_offsets.data());
    }
    if (d_co && d_co != d_ro) dev.Free(d_co);
    if (d_ro) dev.Free(d_ro);
    if (rc != SBX_OK) {
      if (rp) dev.Free(rp);
      if (col) dev.Free(col);
      if (vals) dev.Free(vals);
      delete out;
      dev.Check(rc);
    }
    context::HIPContext ctx(dev.id());
    out->n = v.n, out->m = v.m;
    out->row_begin = splits[rank], out->row_end = splits[rank + 1];
    out->row_ptr = new format::HIPArray<NNZType>((size_t)v.n + 1, rp, ctx, format::kOwned);
    out->col = new format::HIPArray<IDType>((size_t)cap, col, ctx, format::kOwned);
    if (vals) out->vals = new format::HIPArray<Stored>((size_t)cap, vals, ctx, format::kOwned);
    return out;
  }
};

}  // namespace sparsebase::permute
#endif
