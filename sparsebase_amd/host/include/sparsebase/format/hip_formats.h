// sparsebase/format/hip_formats.h — device-resident formats HIPCSR / HIPCOO / HIPCSC / HIPArray
// (MI355X counterparts of the reference's CUDACSR / CUDAArray,
// format/cuda_csr_cuda.cuh:20-59, cuda_array_cuda.cuh).  They own (or borrow) raw
// device pointers on the GPU named by their HIPContext.  Constructors keep the
// reference's sort-on-construct contract, executed by the HIP kernels behind the C ABI.
#ifndef SPARSEBASE_FORMAT_HIP_FORMATS_H_
#define SPARSEBASE_FORMAT_HIP_FORMATS_H_
#include "sparsebase/context/hip_context.h"
#include "sparsebase/format/format_order_one.h"
#include "sparsebase/format/format_order_two.h"
#include "sparsebase/hip/device.h"

namespace sparsebase::format {

// frees with sbx_free on the owning device (reference: utils/utils_cuda.cuh:6-9)
template <typename T>
struct HIPDeleter {
  int device_id;
  void operator()(T *p) const {
    if (p) hip::Device::Get(device_id).Free((void *)p);
  }
};

namespace detail {
template <typename T>
OwnedPtr<T> HoldDevice(T *p, Ownership own, int device_id) {
  if (own == kOwned) return OwnedPtr<T>(p, HIPDeleter<T>{device_id});
  return OwnedPtr<T>(p, BlankDeleter<T>());
}
template <typename T>
T *CloneDevice(const hip::Device &d, const T *src, size_t bytes) {
  if (!src) return nullptr;
  void *dst = d.Malloc(bytes);
  d.Copy(dst, src, bytes);
  return (T *)dst;
}
}  // namespace detail

template <typename IDType, typename NNZType, typename ValueType>
class HIPCSR : public utils::IdentifiableImplementation<HIPCSR<IDType, NNZType, ValueType>,
                                                        FormatOrderTwo<IDType, NNZType, ValueType>> {
 public:
  // nnz is passed explicitly (row_ptr lives on the device)
  HIPCSR(IDType n, IDType m, NNZType nnz, NNZType *row_ptr, IDType *col, ValueType *vals,
         context::HIPContext context, Ownership own = kOwned, bool ignore_sort = false)
      : row_ptr_(detail::HoldDevice(row_ptr, own, context.device_id)),
        col_(detail::HoldDevice(col, own, context.device_id)),
        vals_(detail::HoldDevice(vals, own, context.device_id)) {
    this->order_ = 2;
    this->dimension_ = {(DimensionType)n, (DimensionType)m};
    this->nnz_ = (DimensionType)nnz;
    this->context_ = std::unique_ptr<context::Context>(new context::HIPContext(context));
    if (!ignore_sort && nnz > 1) {  // format/csr.cc:99-157 on the device
      auto &dev = device();
      dev.Check(sbx_csr_sort_rows(dev.handle(), hip::IndexTag<IDType, NNZType>(), hip::ValueTag<ValueType>(), n, m,
                                  (int64_t)nnz, row_ptr, col, vals));
    }
  }
  HIPCSR(const HIPCSR &rhs) : HIPCSR(rhs, CloneTag{}) {}
  Format *Clone() const override { return new HIPCSR(*this); }
  ~HIPCSR() override = default;

  NNZType *get_row_ptr() const { return row_ptr_.get(); }
  IDType *get_col() const { return col_.get(); }
  ValueType *get_vals() const { return vals_.get(); }
  NNZType *release_row_ptr() { return Release(row_ptr_); }
  IDType *release_col() { return Release(col_); }
  ValueType *release_vals() { return Release(vals_); }
  context::HIPContext *get_hip_context() const { return static_cast<context::HIPContext *>(this->get_context()); }
  hip::Device &device() const { return hip::Device::Get(get_hip_context()->device_id); }

 protected:
  detail::OwnedPtr<NNZType> row_ptr_;
  detail::OwnedPtr<IDType> col_;
  detail::OwnedPtr<ValueType> vals_;

 private:
  struct CloneTag {};
  HIPCSR(const HIPCSR &rhs, CloneTag)
      : row_ptr_(nullptr, BlankDeleter<NNZType>()), col_(nullptr, BlankDeleter<IDType>()),
        vals_(nullptr, BlankDeleter<ValueType>()) {
    const int did = rhs.get_hip_context()->device_id;
    auto &dev = hip::Device::Get(did);
    const size_t n = rhs.dimension_[0], nnz = rhs.nnz_;
    row_ptr_ = detail::HoldDevice(detail::CloneDevice(dev, rhs.get_row_ptr(), (n + 1) * sizeof(NNZType)), kOwned, did);
    col_ = detail::HoldDevice(detail::CloneDevice(dev, rhs.get_col(), nnz * sizeof(IDType)), kOwned, did);
    vals_ = detail::HoldDevice(
        (ValueType *)detail::CloneDevice(dev, (const char *)rhs.get_vals(), nnz * hip::ValueBytes<ValueType>()), kOwned,
        did);
    this->order_ = 2;
    this->dimension_ = rhs.dimension_;
    this->nnz_ = rhs.nnz_;
    this->context_ = std::unique_ptr<context::Context>(new context::HIPContext(did));
  }
  template <typename T>
  T *Release(detail::OwnedPtr<T> &p) {
    T *raw = p.release();
    p = detail::OwnedPtr<T>(raw, BlankDeleter<T>());
    device().Forget((void *)raw);  // (whoever frees it now does so past the host layer's block pool)
    return raw;
  }
};

template <typename IDType, typename NNZType, typename ValueType>
class HIPCOO : public utils::IdentifiableImplementation<HIPCOO<IDType, NNZType, ValueType>,
                                                        FormatOrderTwo<IDType, NNZType, ValueType>> {
 public:
  HIPCOO(IDType n, IDType m, NNZType nnz, IDType *row, IDType *col, ValueType *vals, context::HIPContext context,
         Ownership own = kOwned, bool ignore_sort = false)
      : col_(detail::HoldDevice(col, own, context.device_id)),
        row_(detail::HoldDevice(row, own, context.device_id)),
        vals_(detail::HoldDevice(vals, own, context.device_id)),
        rows_sorted_(false) {
    this->order_ = 2;
    this->dimension_ = {(DimensionType)n, (DimensionType)m};
    this->nnz_ = (DimensionType)nnz;
    this->context_ = std::unique_ptr<context::Context>(new context::HIPContext(context));
    if (!ignore_sort) {  // format/coo.cc:96-157 on the device
      auto &dev = device();
      if (nnz > 1)
        dev.Check(sbx_coo_sort(dev.handle(), hip::IndexTag<IDType, NNZType>(), hip::ValueTag<ValueType>(), n, m, (int64_t)nnz,
                               row, col, vals));
      rows_sorted_ = true;
    }
  }
  HIPCOO(const HIPCOO &rhs)
      : col_(nullptr, BlankDeleter<IDType>()), row_(nullptr, BlankDeleter<IDType>()),
        vals_(nullptr, BlankDeleter<ValueType>()), rows_sorted_(rhs.rows_sorted_) {
    const int did = rhs.get_hip_context()->device_id;
    auto &dev = hip::Device::Get(did);
    const size_t nnz = rhs.nnz_;
    row_ = detail::HoldDevice(detail::CloneDevice(dev, rhs.get_row(), nnz * sizeof(IDType)), kOwned, did);
    col_ = detail::HoldDevice(detail::CloneDevice(dev, rhs.get_col(), nnz * sizeof(IDType)), kOwned, did);
    vals_ = detail::HoldDevice(
        (ValueType *)detail::CloneDevice(dev, (const char *)rhs.get_vals(), nnz * hip::ValueBytes<ValueType>()), kOwned,
        did);
    this->order_ = 2;
    this->dimension_ = rhs.dimension_;
    this->nnz_ = rhs.nnz_;
    this->context_ = std::unique_ptr<context::Context>(new context::HIPContext(did));
  }
  Format *Clone() const override { return new HIPCOO(*this); }
  ~HIPCOO() override = default;

  IDType *get_row() const { return row_.get(); }
  IDType *get_col() const { return col_.get(); }
  ValueType *get_vals() const { return vals_.get(); }
  IDType *release_row() { return Release(row_); }
  IDType *release_col() { return Release(col_); }
  ValueType *release_vals() { return Release(vals_); }
  bool rows_known_sorted() const { return rows_sorted_; }
  context::HIPContext *get_hip_context() const { return static_cast<context::HIPContext *>(this->get_context()); }
  hip::Device &device() const { return hip::Device::Get(get_hip_context()->device_id); }

 protected:
  detail::OwnedPtr<IDType> col_;
  detail::OwnedPtr<IDType> row_;
  detail::OwnedPtr<ValueType> vals_;
  bool rows_sorted_;

 private:
  template <typename T>
  T *Release(detail::OwnedPtr<T> &p) {
    T *raw = p.release();
    p = detail::OwnedPtr<T>(raw, BlankDeleter<T>());
    device().Forget((void *)raw);  // (whoever frees it now does so past the host layer's block pool)
    return raw;
  }
};

// Device-resident CSC.  col_ptr has max(n, m) + 1 entries like the host CSC (csc.h).
template <typename IDType, typename NNZType, typename ValueType>
class HIPCSC : public utils::IdentifiableImplementation<HIPCSC<IDType, NNZType, ValueType>,
                                                        FormatOrderTwo<IDType, NNZType, ValueType>> {
 public:
  static size_t PtrCount(DimensionType n, DimensionType m) { return (size_t)(n > m ? n : m); }
  HIPCSC(IDType n, IDType m, NNZType nnz, NNZType *col_ptr, IDType *row, ValueType *vals,
         context::HIPContext context, Ownership own = kOwned, bool ignore_sort = false)
      : col_ptr_(detail::HoldDevice(col_ptr, own, context.device_id)),
        row_(detail::HoldDevice(row, own, context.device_id)),
        vals_(detail::HoldDevice(vals, own, context.device_id)) {
    this->order_ = 2;
    this->dimension_ = {(DimensionType)n, (DimensionType)m};
    this->nnz_ = (DimensionType)nnz;
    this->context_ = std::unique_ptr<context::Context>(new context::HIPContext(context));
    if (!ignore_sort && nnz > 1) {  // format/csc.cc:99-157 on the device
      auto &dev = device();
      dev.Check(sbx_csr_sort_rows(dev.handle(), hip::IndexTag<IDType, NNZType>(), hip::ValueTag<ValueType>(),
                                  (int64_t)PtrCount(n, m), n, (int64_t)nnz, col_ptr, row, vals));
    }
  }
  HIPCSC(const HIPCSC &rhs)
      : col_ptr_(nullptr, BlankDeleter<NNZType>()), row_(nullptr, BlankDeleter<IDType>()),
        vals_(nullptr, BlankDeleter<ValueType>()) {
    const int did = rhs.get_hip_context()->device_id;
    auto &dev = hip::Device::Get(did);
    const size_t nnz = rhs.nnz_;
    col_ptr_ = detail::HoldDevice(detail::CloneDevice(dev, rhs.get_col_ptr(), (rhs.ptr_count() + 1) * sizeof(NNZType)),
                                  kOwned, did);
    row_ = detail::HoldDevice(detail::CloneDevice(dev, rhs.get_row(), nnz * sizeof(IDType)), kOwned, did);
    vals_ = detail::HoldDevice(
        (ValueType *)detail::CloneDevice(dev, (const char *)rhs.get_vals(), nnz * hip::ValueBytes<ValueType>()), kOwned,
        did);
    this->order_ = 2;
    this->dimension_ = rhs.dimension_;
    this->nnz_ = rhs.nnz_;
    this->context_ = std::unique_ptr<context::Context>(new context::HIPContext(did));
  }
  Format *Clone() const override { return new HIPCSC(*this); }
  ~HIPCSC() override = default;

  size_t ptr_count() const { return PtrCount(this->dimension_[0], this->dimension_[1]); }
  NNZType *get_col_ptr() const { return col_ptr_.get(); }
  IDType *get_row() const { return row_.get(); }
  ValueType *get_vals() const { return vals_.get(); }
  context::HIPContext *get_hip_context() const { return static_cast<context::HIPContext *>(this->get_context()); }
  hip::Device &device() const { return hip::Device::Get(get_hip_context()->device_id); }

 protected:
  detail::OwnedPtr<NNZType> col_ptr_;
  detail::OwnedPtr<IDType> row_;
  detail::OwnedPtr<ValueType> vals_;
};

template <typename ValueType>
class HIPArray
    : public utils::IdentifiableImplementation<HIPArray<ValueType>, FormatOrderOne<ValueType>> {
 public:
  HIPArray(DimensionType nnz, ValueType *vals, context::HIPContext context, Ownership own = kOwned)
      : vals_(detail::HoldDevice(vals, own, context.device_id)) {
    this->order_ = 1;
    this->dimension_ = {nnz};
    this->nnz_ = nnz;
    this->context_ = std::unique_ptr<context::Context>(new context::HIPContext(context));
  }
  HIPArray(const HIPArray &rhs) : vals_(nullptr, BlankDeleter<ValueType>()) {
    const int did = rhs.get_hip_context()->device_id;
    vals_ = detail::HoldDevice(detail::CloneDevice(hip::Device::Get(did), rhs.get_vals(), rhs.nnz_ * sizeof(ValueType)),
                               kOwned, did);
    this->order_ = 1;
    this->dimension_ = rhs.dimension_;
    this->nnz_ = rhs.nnz_;
    this->context_ = std::unique_ptr<context::Context>(new context::HIPContext(did));
  }
  Format *Clone() const override { return new HIPArray(*this); }
  ValueType *get_vals() const { return vals_.get(); }
  ValueType *release_vals() {
    ValueType *raw = vals_.release();
    vals_ = detail::OwnedPtr<ValueType>(raw, BlankDeleter<ValueType>());
    hip::Device::Get(get_hip_context()->device_id).Forget((void *)raw);
    return raw;
  }
  context::HIPContext *get_hip_context() const { return static_cast<context::HIPContext *>(this->get_context()); }

 protected:
  detail::OwnedPtr<ValueType> vals_;
};

}  // namespace sparsebase::format
#endif
