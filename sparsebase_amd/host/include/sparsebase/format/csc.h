// sparsebase/format/csc.h — host-resident CSC with the reference's constructor contract
// (format/csc.cc:79-159): unless ignore_sort, if ANY column's rows are out of order EVERY
// column is sorted by (row,val) IN PLACE on the caller's arrays — on the GPU, through the
// same kernels as the CSR constructor with the roles of the dimensions swapped.
//
// Reference quirk kept on purpose: the reference treats dimension[0] (the ROW count n) as
// the number of columns of the col_ptr array — nnz = col_ptr[n] (csc.cc:87), the sort loop
// runs i < n (:104,:125), COO->CSC allocates n + 1 entries (converter_order_two.cc:32) and
// the reference's own tests compare n + 1 entries (converter/common.inc:14, :68).  That is
// well defined for n >= m (entries past m repeat nnz) and a heap overflow for m > n.  Here
// col_ptr has max(n, m) + 1 entries: identical to the reference wherever the reference is
// defined, and correct (m + 1 entries) where it is not.
#ifndef SPARSEBASE_FORMAT_CSC_H_
#define SPARSEBASE_FORMAT_CSC_H_
#include <algorithm>

#include "sparsebase/format/format_order_two.h"
#include "sparsebase/hip/device.h"
#include "sparsebase/utils/logger.h"

namespace sparsebase::format {

template <typename IDType, typename NNZType, typename ValueType>
class CSC : public utils::IdentifiableImplementation<CSC<IDType, NNZType, ValueType>,
                                                     FormatOrderTwo<IDType, NNZType, ValueType>> {
 public:
  // number of col_ptr intervals (see the header comment)
  static size_t PtrCount(DimensionType n, DimensionType m) { return (size_t)std::max(n, m); }

  CSC(IDType n, IDType m, NNZType *col_ptr, IDType *row, ValueType *vals, Ownership own = kNotOwned,
      bool ignore_sort = false)
      : col_ptr_(detail::Hold(col_ptr, own)), row_(detail::Hold(row, own)), vals_(detail::Hold(vals, own)) {
    this->order_ = 2;
    this->dimension_ = {(DimensionType)n, (DimensionType)m};
    this->nnz_ = (DimensionType)col_ptr[PtrCount(n, m)];
    this->context_ = std::unique_ptr<context::Context>(new context::CPUContext);
    if (!ignore_sort) SortOnDevice(col_ptr, row, vals);
  }
  CSC(const CSC &rhs)
      : col_ptr_(detail::Hold(detail::CloneArray(rhs.get_col_ptr(), rhs.ptr_count() + 1), kOwned)),
        row_(detail::Hold(detail::CloneArray(rhs.get_row(), rhs.get_num_nnz()), kOwned)),
        vals_(detail::Hold(detail::CloneArray(rhs.get_vals(), rhs.get_num_nnz()), kOwned)) {
    this->order_ = 2;
    this->dimension_ = rhs.dimension_;
    this->nnz_ = rhs.nnz_;
    this->context_ = std::unique_ptr<context::Context>(new context::CPUContext);
  }
  CSC &operator=(const CSC &rhs) {
    if (this == &rhs) return *this;
    this->dimension_ = rhs.dimension_;
    this->nnz_ = rhs.nnz_;
    col_ptr_ = detail::Hold(detail::CloneArray(rhs.get_col_ptr(), rhs.ptr_count() + 1), kOwned);
    row_ = detail::Hold(detail::CloneArray(rhs.get_row(), rhs.get_num_nnz()), kOwned);
    vals_ = detail::Hold(detail::CloneArray(rhs.get_vals(), rhs.get_num_nnz()), kOwned);
    return *this;
  }
  Format *Clone() const override { return new CSC(*this); }
  ~CSC() override = default;

  size_t ptr_count() const { return PtrCount(this->dimension_[0], this->dimension_[1]); }
  NNZType *get_col_ptr() const { return col_ptr_.get(); }
  IDType *get_row() const { return row_.get(); }
  ValueType *get_vals() const { return vals_.get(); }

  NNZType *release_col_ptr() { return Release(col_ptr_); }
  IDType *release_row() { return Release(row_); }
  ValueType *release_vals() { return Release(vals_); }

  void set_col_ptr(NNZType *p, Ownership own = kNotOwned) { col_ptr_ = detail::Hold(p, own); }
  void set_row(IDType *p, Ownership own = kNotOwned) { row_ = detail::Hold(p, own); }
  void set_vals(ValueType *p, Ownership own = kNotOwned) { vals_ = detail::Hold(p, own); }

  virtual bool ColPtrIsOwned() { return Owns(col_ptr_); }
  virtual bool RowIsOwned() { return Owns(row_); }
  virtual bool ValsIsOwned() { return Owns(vals_); }

 protected:
  detail::OwnedPtr<NNZType> col_ptr_;
  detail::OwnedPtr<IDType> row_;
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
  void SortOnDevice(NNZType *col_ptr, IDType *row, ValueType *vals) {
    const size_t nnz = (size_t)this->nnz_, ncols = ptr_count();
    if (nnz <= 1) return;
    auto &dev = hip::Device::Get(hip::DefaultDevice());
    hip::Staged<NNZType> d_cp(dev, col_ptr, ncols + 1);
    hip::Staged<IDType> d_row(dev, row, nnz);
    int sorted = 1;
    dev.Check(sbx_csr_rows_sorted(dev.handle(), hip::IndexTag<IDType, NNZType>(), (int64_t)ncols, d_cp.get(), d_row.get(),
                                  &sorted));
    if (sorted) return;
    utils::Logger(typeid(this)).Log("CSC column array must be sorted. Sorting...", utils::LOG_LVL_WARNING);
    constexpr size_t vb = hip::ValueBytes<ValueType>();
    void *d_val = nullptr;
    if (vb && vals) {
      d_val = dev.Malloc(nnz * vb);
      dev.ToDevice(d_val, vals, nnz * vb);
    }
    const int rc = sbx_csr_sort_rows(dev.handle(), hip::IndexTag<IDType, NNZType>(), hip::ValueTag<ValueType>(),
                                     (int64_t)ncols, (int64_t)this->dimension_[0], (int64_t)nnz, d_cp.get(),
                                     d_row.get(), d_val);
    if (rc == SBX_OK) {
      d_row.ToHost(row);
      if (d_val) dev.ToHost(vals, d_val, nnz * vb);
    }
    if (d_val) dev.Free(d_val);
    dev.Check(rc);
  }
};

}  // namespace sparsebase::format
#endif
