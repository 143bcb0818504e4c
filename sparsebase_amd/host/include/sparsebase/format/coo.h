// sparsebase/format/coo.h — host-resident COO with the reference's constructor
// contract (format/coo.cc:76-158): unless ignore_sort, entries that are not in
// (row,col) order are sorted IN PLACE on the caller's arrays.  Check and sort run on
// the GPU (sbx_coo_sort); duplicates keep their input order (the reference leaves
// their order unspecified).
#ifndef SPARSEBASE_FORMAT_COO_H_
#define SPARSEBASE_FORMAT_COO_H_
#include "sparsebase/format/csr.h"

namespace sparsebase::format {

template <typename IDType, typename NNZType, typename ValueType>
class COO : public utils::IdentifiableImplementation<COO<IDType, NNZType, ValueType>,
                                                     FormatOrderTwo<IDType, NNZType, ValueType>> {
 public:
  COO(IDType n, IDType m, NNZType nnz, IDType *row, IDType *col, ValueType *vals, Ownership own = kNotOwned,
      bool ignore_sort = false)
      : col_(detail::Hold(col, own)), row_(detail::Hold(row, own)), vals_(detail::Hold(vals, own)) {
    this->nnz_ = (DimensionType)nnz;
    this->order_ = 2;
    this->dimension_ = {(DimensionType)n, (DimensionType)m};
    this->context_ = std::unique_ptr<context::Context>(new context::CPUContext);
    if (!ignore_sort) SortOnDevice(n, m, nnz, row, col, vals);
  }
  COO(const COO &rhs)
      : col_(detail::Hold(detail::CloneArray(rhs.get_col(), rhs.get_num_nnz()), kOwned)),
        row_(detail::Hold(detail::CloneArray(rhs.get_row(), rhs.get_num_nnz()), kOwned)),
        vals_(detail::Hold(detail::CloneArray(rhs.get_vals(), rhs.get_num_nnz()), kOwned)) {
    this->order_ = 2;
    this->dimension_ = rhs.dimension_;
    this->nnz_ = rhs.nnz_;
    this->context_ = std::unique_ptr<context::Context>(new context::CPUContext);
  }
  Format *Clone() const override { return new COO(*this); }
  ~COO() override = default;

  IDType *get_col() const { return col_.get(); }
  IDType *get_row() const { return row_.get(); }
  ValueType *get_vals() const { return vals_.get(); }
  IDType *release_col() { return Release(col_); }
  IDType *release_row() { return Release(row_); }
  ValueType *release_vals() { return Release(vals_); }
  void set_row(IDType *p, Ownership own = kNotOwned) { row_ = detail::Hold(p, own); }
  void set_col(IDType *p, Ownership own = kNotOwned) { col_ = detail::Hold(p, own); }
  void set_vals(ValueType *p, Ownership own = kNotOwned) { vals_ = detail::Hold(p, own); }
  virtual bool RowIsOwned() { return Owns(row_); }
  virtual bool ColIsOwned() { return Owns(col_); }
  virtual bool ValsIsOwned() { return Owns(vals_); }

 protected:
  detail::OwnedPtr<IDType> col_;
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
  void SortOnDevice(IDType n, IDType m, NNZType nnz, IDType *row, IDType *col, ValueType *vals) {
    if (nnz <= 1) return;
    auto &dev = hip::Device::Get(hip::DefaultDevice());
    hip::Staged<IDType> d_row(dev, row, (size_t)nnz), d_col(dev, col, (size_t)nnz);
    int sorted = 1;
    dev.Check(sbx_coo_is_sorted(dev.handle(), hip::IndexTag<IDType>(), (int64_t)nnz, d_row.get(), d_col.get(),
                                &sorted));
    if (sorted) return;
    utils::Logger(typeid(this)).Log("COO arrays must be sorted. Sorting...", utils::LOG_LVL_WARNING);
    constexpr size_t vb = hip::ValueBytes<ValueType>();
    void *d_val = nullptr;
    if (vb && vals) {
      d_val = dev.Malloc((size_t)nnz * vb);
      dev.ToDevice(d_val, vals, (size_t)nnz * vb);
    }
    const int rc = sbx_coo_sort(dev.handle(), hip::IndexTag<IDType>(), hip::ValueTag<ValueType>(), n, m,
                                (int64_t)nnz, d_row.get(), d_col.get(), d_val);
    if (rc == SBX_OK) {
      d_row.ToHost(row);
      d_col.ToHost(col);
      if (d_val) dev.ToHost(vals, d_val, (size_t)nnz * vb);
    }
    if (d_val) dev.Free(d_val);
    dev.Check(rc);
  }
};

}  // namespace sparsebase::format
#include "sparsebase/converter/converter_order_two.h"
#endif
