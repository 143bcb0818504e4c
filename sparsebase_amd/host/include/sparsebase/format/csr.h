// sparsebase/format/csr.h — host-resident CSR with the reference's constructor
// contract (format/csr.cc:78-159): nnz = row_ptr[n]; unless ignore_sort, if ANY row is
// out of order EVERY row is sorted by (col,val) IN PLACE on the caller's arrays.
// The sort itself runs on the GPU: the arrays are staged to the default device and
// sbx_csr_sort_rows does check + sort (no host sort exists in this library).
#ifndef SPARSEBASE_FORMAT_CSR_H_
#define SPARSEBASE_FORMAT_CSR_H_
#include "sparsebase/format/format_order_two.h"
#include "sparsebase/hip/device.h"
#include "sparsebase/utils/logger.h"

namespace sparsebase::format {


template <typename IDType, typename NNZType, typename ValueType>
class CSR : public utils::IdentifiableImplementation<CSR<IDType, NNZType, ValueType>,
                                                     FormatOrderTwo<IDType, NNZType, ValueType>> {
 public:
  CSR(IDType n, IDType m, NNZType *row_ptr, IDType *col, ValueType *vals, Ownership own = kNotOwned,
      bool ignore_sort = false)
      : row_ptr_(detail::Hold(row_ptr, own)), col_(detail::Hold(col, own)), vals_(detail::Hold(vals, own)) {
    this->order_ = 2;
    this->dimension_ = {(DimensionType)n, (DimensionType)m};
    this->nnz_ = (DimensionType)row_ptr[n];
    this->context_ = std::unique_ptr<context::Context>(new context::CPUContext);
    if (!ignore_sort) SortOnDevice(n, m, row_ptr, col, vals);
  }
  CSR(const CSR &rhs)
      : row_ptr_(detail::Hold(detail::CloneArray(rhs.get_row_ptr(), rhs.get_dimensions()[0] + 1), kOwned)),
        col_(detail::Hold(detail::CloneArray(rhs.get_col(), rhs.get_num_nnz()), kOwned)),
        vals_(detail::Hold(detail::CloneArray(rhs.get_vals(), rhs.get_num_nnz()), kOwned)) {
    this->order_ = 2;
    this->dimension_ = rhs.dimension_;
    this->nnz_ = rhs.nnz_;
    this->context_ = std::unique_ptr<context::Context>(new context::CPUContext);
  }
  CSR &operator=(const CSR &rhs) {
    if (this == &rhs) return *this;
    this->dimension_ = rhs.dimension_;
    this->nnz_ = rhs.nnz_;
    row_ptr_ = detail::Hold(detail::CloneArray(rhs.get_row_ptr(), rhs.get_dimensions()[0] + 1), kOwned);
    col_ = detail::Hold(detail::CloneArray(rhs.get_col(), rhs.get_num_nnz()), kOwned);
    vals_ = detail::Hold(detail::CloneArray(rhs.get_vals(), rhs.get_num_nnz()), kOwned);
    return *this;
  }
  Format *Clone() const override { return new CSR(*this); }
  ~CSR() override = default;

  NNZType *get_row_ptr() const { return row_ptr_.get(); }
  IDType *get_col() const { return col_.get(); }
  ValueType *get_vals() const { return vals_.get(); }

  // hand the array out and stop owning it (csr.cc:178-197)
  NNZType *release_row_ptr() { return Release(row_ptr_); }
  IDType *release_col() { return Release(col_); }
  ValueType *release_vals() { return Release(vals_); }

  void set_row_ptr(NNZType *p, Ownership own = kNotOwned) { row_ptr_ = detail::Hold(p, own); }
  void set_col(IDType *p, Ownership own = kNotOwned) { col_ = detail::Hold(p, own); }
  void set_vals(ValueType *p, Ownership own = kNotOwned) { vals_ = detail::Hold(p, own); }

  virtual bool RowPtrIsOwned() { return Owns(row_ptr_); }
  virtual bool ColIsOwned() { return Owns(col_); }
  virtual bool ValsIsOwned() { return Owns(vals_); }

 protected:
  detail::OwnedPtr<NNZType> row_ptr_;
  detail::OwnedPtr<IDType> col_;
  detail::OwnedPtr<ValueType> vals_;

 private:
  template <typename T>
  static T *Release(detail::OwnedPtr<T> &p) {
    T *raw = p.release();
    p = detail::OwnedPtr<T>(raw, BlankDeleter<T>());
    return raw;
  }
  template <typename T>
  static bool Owns(detail::OwnedPtr<T> &p) {
    return p.get_deleter().target_type() != typeid(BlankDeleter<T>);
  }
  void SortOnDevice(IDType n, IDType m, NNZType *row_ptr, IDType *col, ValueType *vals) {
    const size_t nnz = (size_t)row_ptr[n];
    if (nnz <= 1) return;
    auto &dev = hip::Device::Get(hip::DefaultDevice());
    hip::Staged<NNZType> d_rp(dev, row_ptr, (size_t)n + 1);
    hip::Staged<IDType> d_col(dev, col, nnz);
    int sorted = 1;
    dev.Check(sbx_csr_rows_sorted(dev.handle(), hip::IndexTag<IDType, NNZType>(), n, d_rp.get(), d_col.get(), &sorted));
    if (sorted) return;
    utils::Logger(typeid(this)).Log("CSR column array must be sorted. Sorting...", utils::LOG_LVL_WARNING);
    constexpr size_t vb = hip::ValueBytes<ValueType>();
    void *d_val = nullptr;
    if (vb && vals) {
      d_val = dev.Malloc(nnz * vb);
      dev.ToDevice(d_val, vals, nnz * vb);
    }
    const int rc = sbx_csr_sort_rows(dev.handle(), hip::IndexTag<IDType, NNZType>(), hip::ValueTag<ValueType>(), n, m,
                                     (int64_t)nnz, d_rp.get(), d_col.get(), d_val);
    if (rc == SBX_OK) {
      d_col.ToHost(col);
      if (d_val) dev.ToHost(vals, d_val, nnz * vb);
    }
    if (d_val) dev.Free(d_val);
    dev.Check(rc);
  }
};

}  // namespace sparsebase::format
#include "sparsebase/format/coo.h"
#endif
