// sparsebase/converter/converter_order_two.h — conversion functions between the 2-D
// formats and their registration (reference: converter/converter_order_two.cc:21-345,
// converter_order_two_cuda.cu:11-105).
//
// Every function below ends in a HIP kernel behind the C ABI:
//   COO -> CSR   sbx_coo_to_csr  (converter_order_two.cc:163-212, move :215-246)
//   CSR -> COO   sbx_csr_to_coo  (converter_order_two.cc:72-118,  move :131-160)
//   COO -> CSC   sbx_coo_to_csc  (converter_order_two.cc:21-70)
//   CSR -> CSC   sbx_csr_to_csc  (converter_order_two.cc:120-128)
// Host-resident formats (CPUContext) are staged through the default device and the
// result is delivered back to host arrays, so existing call sites such as
// coo->Convert<format::CSR>(&cpu_context) keep working unchanged; device-resident
// formats (HIPCSR/HIPCOO) convert in place in HBM.  The constructor semantics of the
// destination format (sort check, format/csr.cc:99-157, format/coo.cc:96-157) are
// applied on the device before anything is copied back.
#ifndef SPARSEBASE_CONVERTER_CONVERTER_ORDER_TWO_H_
#define SPARSEBASE_CONVERTER_CONVERTER_ORDER_TWO_H_
#include <type_traits>
#include <vector>

#include "sparsebase/context/cpu_context.h"
#include "sparsebase/context/hip_context.h"
#include "sparsebase/converter/converter.h"
#include "sparsebase/format/coo.h"
#include "sparsebase/format/csc.h"
#include "sparsebase/format/csr.h"
#include "sparsebase/format/hip_formats.h"

namespace sparsebase::converter {

namespace detail {
inline bool ToCPU(context::Context *, context::Context *to) {
  return to->get_id() == context::CPUContext::get_id_static();
}
inline bool ToHIP(context::Context *, context::Context *to) {
  return to->get_id() == context::HIPContext::get_id_static();
}
// device -> device edges: the BFS hands every hop the ORIGINAL source context
// (converter.cc:163), so `from` may still be the CPU context of the first hop
inline bool OnHIP(context::Context *from, context::Context *to) {
  if (to->get_id() != context::HIPContext::get_id_static()) return false;
  if (from->get_id() != context::HIPContext::get_id_static()) return true;
  return from->IsEquivalent(to);
}
inline int DeviceOf(context::Context *ctx) { return static_cast<context::HIPContext *>(ctx)->device_id; }

template <typename V>
void *UploadValues(const hip::Device &dev, const V *vals, size_t nnz) {
  if constexpr (std::is_same_v<V, void>) {
    return nullptr;
  } else {
    return vals ? (void *)dev.Upload(vals, nnz) : nullptr;
  }
}
template <typename V>
V *DownloadValues(const hip::Device &dev, const void *d_vals, size_t nnz) {
  if constexpr (std::is_same_v<V, void>) {
    return nullptr;
  } else {
    return d_vals ? dev.Download((const V *)d_vals, nnz) : nullptr;
  }
}
}  // namespace detail

// ------------------------------------------------------------------ host formats (staged)
template <typename I, typename N, typename V>
format::Format *CooCsrFunctionConditional(format::Format *source, context::Context *) {
  auto *coo = source->AsAbsolute<format::COO<I, N, V>>();
  const auto dims = coo->get_dimensions();
  const I n = (I)dims[0], m = (I)dims[1];
  const size_t nnz = coo->get_num_nnz();
  auto &dev = hip::Device::Get(hip::DefaultDevice());
  hip::Staged<I> d_row(dev, coo->get_row(), nnz), d_col(dev, coo->get_col(), nnz);
  hip::Staged<N> d_rp(dev, (size_t)n + 1);
  hip::Staged<I> d_col_out(dev, nnz);
  void *d_val = detail::UploadValues<V>(dev, coo->get_vals(), nnz);
  void *d_val_out = d_val ? dev.Malloc(nnz * hip::ValueBytes<V>()) : nullptr;
  int rc = sbx_coo_to_csr(dev.handle(), hip::IndexTag<I, N>(), hip::ValueTag<V>(), n, m, (int64_t)nnz, d_row.get(),
                          d_col.get(), d_val, d_rp.get(), d_col_out.get(), d_val_out, 0u);
  if (rc == SBX_OK)  // destination constructor semantics on the device
    rc = sbx_csr_sort_rows(dev.handle(), hip::IndexTag<I, N>(), hip::ValueTag<V>(), n, m, (int64_t)nnz, d_rp.get(),
                           d_col_out.get(), d_val_out);
  N *row_ptr = nullptr;
  I *col = nullptr;
  V *vals = nullptr;
  if (rc == SBX_OK) {
    row_ptr = dev.Download(d_rp.get(), (size_t)n + 1);
    col = dev.Download(d_col_out.get(), nnz);
    vals = detail::DownloadValues<V>(dev, d_val_out, nnz);
  }
  if (d_val) dev.Free(d_val);
  if (d_val_out) dev.Free(d_val_out);
  dev.Check(rc);
  return new format::CSR<I, N, V>(n, m, row_ptr, col, vals, format::kOwned, true);
}

template <typename I, typename N, typename V>
format::Format *CooCsrMoveConditionalFunction(format::Format *source, context::Context *) {
  auto *coo = source->AsAbsolute<format::COO<I, N, V>>();
  const auto dims = coo->get_dimensions();
  const I n = (I)dims[0], m = (I)dims[1];
  const size_t nnz = coo->get_num_nnz();
  auto &dev = hip::Device::Get(hip::DefaultDevice());
  hip::Staged<I> d_row(dev, coo->get_row(), nnz);
  hip::Staged<N> d_rp(dev, (size_t)n + 1);
  dev.Check(sbx_coo_to_csr(dev.handle(), hip::IndexTag<I, N>(), hip::ValueTag<V>(), n, m, (int64_t)nnz, d_row.get(),
                           nullptr, nullptr, d_rp.get(), nullptr, nullptr, SBX_FLAG_MOVE));
  N *row_ptr = dev.Download(d_rp.get(), (size_t)n + 1);
  I *col = coo->release_col();    // pointer hand-off (converter_order_two.cc:225-226)
  V *vals = coo->release_vals();
  // CSR constructor: sort check on the handed-over arrays (in place if needed)
  return new format::CSR<I, N, V>(n, m, row_ptr, col, vals, format::kOwned, false);
}

template <typename I, typename N, typename V>
format::Format *CsrCooFunctionConditional(format::Format *source, context::Context *) {
  auto *csr = source->AsAbsolute<format::CSR<I, N, V>>();
  const auto dims = csr->get_dimensions();
  const I n = (I)dims[0], m = (I)dims[1];
  const size_t nnz = csr->get_num_nnz();
  auto &dev = hip::Device::Get(hip::DefaultDevice());
  hip::Staged<N> d_rp(dev, csr->get_row_ptr(), (size_t)n + 1);
  hip::Staged<I> d_col(dev, csr->get_col(), nnz), d_row_out(dev, nnz), d_col_out(dev, nnz);
  void *d_val = detail::UploadValues<V>(dev, csr->get_vals(), nnz);
  void *d_val_out = d_val ? dev.Malloc(nnz * hip::ValueBytes<V>()) : nullptr;
  int rc = sbx_csr_to_coo(dev.handle(), hip::IndexTag<I, N>(), hip::ValueTag<V>(), n, m, (int64_t)nnz, d_rp.get(),
                          d_col.get(), d_val, d_row_out.get(), d_col_out.get(), d_val_out, 0u);
  if (rc == SBX_OK)  // COO constructor semantics (check, sort if a row was out of order)
    rc = sbx_coo_sort(dev.handle(), hip::IndexTag<I, N>(), hip::ValueTag<V>(), n, m, (int64_t)nnz, d_row_out.get(),
                      d_col_out.get(), d_val_out);
  I *row = nullptr, *col = nullptr;
  V *vals = nullptr;
  if (rc == SBX_OK) {
    row = dev.Download(d_row_out.get(), nnz);
    col = dev.Download(d_col_out.get(), nnz);
    vals = detail::DownloadValues<V>(dev, d_val_out, nnz);
  }
  if (d_val) dev.Free(d_val);
  if (d_val_out) dev.Free(d_val_out);
  dev.Check(rc);
  return new format::COO<I, N, V>(n, m, (N)nnz, row, col, vals, format::kOwned, true);
}

template <typename I, typename N, typename V>
format::Format *CsrCooMoveConditionalFunction(format::Format *source, context::Context *) {
  auto *csr = source->AsAbsolute<format::CSR<I, N, V>>();
  const auto dims = csr->get_dimensions();
  const I n = (I)dims[0], m = (I)dims[1];
  const size_t nnz = csr->get_num_nnz();
  auto &dev = hip::Device::Get(hip::DefaultDevice());
  hip::Staged<N> d_rp(dev, csr->get_row_ptr(), (size_t)n + 1);
  hip::Staged<I> d_row_out(dev, nnz);
  dev.Check(sbx_csr_to_coo(dev.handle(), hip::IndexTag<I, N>(), hip::ValueTag<V>(), n, m, (int64_t)nnz, d_rp.get(),
                           nullptr, nullptr, d_row_out.get(), nullptr, nullptr, SBX_FLAG_MOVE));
  I *row = dev.Download(d_row_out.get(), nnz);
  I *col = csr->release_col();    // converter_order_two.cc:134-135
  V *vals = csr->release_vals();
  return new format::COO<I, N, V>(n, m, (N)nnz, row, col, vals, format::kOwned, false);
}

// ------------------------------------------------------------------ CSC (host formats, staged)
namespace detail {
// Device col_ptr arrays have max(n, m) + 1 entries (format/csc.h): the C ABI fills the first
// m + 1, the rest repeat nnz exactly like the reference's n + 1-entry array does for n > m.
template <typename N>
void PadColPtr(const hip::Device &dev, N *d_col_ptr, size_t m, size_t ptr_count, size_t nnz) {
  if (ptr_count <= m) return;
  std::vector<N> tail(ptr_count - m, (N)nnz);
  dev.ToDevice(d_col_ptr + m + 1, tail.data(), tail.size() * sizeof(N));
}
// runs sbx_coo_to_csc / sbx_csr_to_csc on device-resident inputs; returns owned device arrays
template <typename I, typename N, typename V, bool FROM_CSR>
void ToCscOnDevice(const hip::Device &dev, I n, I m, size_t nnz, const void *d_first, const I *d_col,
                   const void *d_val, N **cp_out, I **row_out, void **val_out) {
  const size_t pc = format::CSC<I, N, V>::PtrCount(n, m);
  N *cp = (N *)dev.Malloc((pc + 1) * sizeof(N));
  I *row = (I *)dev.Malloc((nnz ? nnz : 1) * sizeof(I));
  void *val = d_val ? dev.Malloc(nnz * hip::ValueBytes<V>()) : nullptr;
  const int rc = FROM_CSR ? sbx_csr_to_csc(dev.handle(), hip::IndexTag<I, N>(), hip::ValueTag<V>(), n, m, (int64_t)nnz,
                                           d_first, d_col, d_val, cp, row, val)
                          : sbx_coo_to_csc(dev.handle(), hip::IndexTag<I, N>(), hip::ValueTag<V>(), n, m, (int64_t)nnz,
                                           d_first, d_col, d_val, cp, row, val);
  if (rc != SBX_OK) {
    dev.Free(cp);
    dev.Free(row);
    if (val) dev.Free(val);
    dev.Check(rc);
  }
  PadColPtr<N>(dev, cp, (size_t)m, pc, nnz);
  *cp_out = cp;
  *row_out = row;
  *val_out = val;
}
template <typename I, typename N, typename V, bool FROM_CSR>
format::Format *StagedToCsc(I n, I m, size_t nnz, const void *first, size_t first_count, const I *col, const V *vals) {
  auto &dev = hip::Device::Get(hip::DefaultDevice());
  typedef typename std::conditional<FROM_CSR, N, I>::type F;  // (first: row_ptr — offsets — or the COO's row ids)
  hip::Staged<F> d_first(dev, (const F *)first, first_count);
  hip::Staged<I> d_col(dev, col, nnz);
  void *d_val = UploadValues<V>(dev, vals, nnz);
  N *d_cp = nullptr;
  I *d_row = nullptr;
  void *d_val_out = nullptr;
  try {
    ToCscOnDevice<I, N, V, FROM_CSR>(dev, n, m, nnz, d_first.get(), d_col.get(), d_val, &d_cp, &d_row, &d_val_out);
  } catch (...) {
    if (d_val) dev.Free(d_val);
    throw;
  }
  const size_t pc = format::CSC<I, N, V>::PtrCount(n, m);
  N *col_ptr = dev.Download(d_cp, pc + 1);
  I *row = dev.Download(d_row, nnz);
  V *out_vals = DownloadValues<V>(dev, d_val_out, nnz);
  dev.Free(d_cp);
  dev.Free(d_row);
  if (d_val_out) dev.Free(d_val_out);
  if (d_val) dev.Free(d_val);
  // the constructor's sort already ran on the device (sbx_*_to_csc ends with it)
  return new format::CSC<I, N, V>(n, m, col_ptr, row, out_vals, format::kOwned, true);
}
}  // namespace detail

template <typename I, typename N, typename V>
format::Format *CooCscFunctionConditional(format::Format *source, context::Context *) {
  auto *coo = source->AsAbsolute<format::COO<I, N, V>>();
  const auto dims = coo->get_dimensions();
  const size_t nnz = coo->get_num_nnz();
  return detail::StagedToCsc<I, N, V, false>((I)dims[0], (I)dims[1], nnz, coo->get_row(), nnz, coo->get_col(),
                                             coo->get_vals());
}
template <typename I, typename N, typename V>
format::Format *CsrCscFunctionConditional(format::Format *source, context::Context *) {
  auto *csr = source->AsAbsolute<format::CSR<I, N, V>>();
  const auto dims = csr->get_dimensions();
  return detail::StagedToCsc<I, N, V, true>((I)dims[0], (I)dims[1], csr->get_num_nnz(), csr->get_row_ptr(),
                                            (size_t)dims[0] + 1, csr->get_col(), csr->get_vals());
}

// CSC <-> HIPCSC copies and the in-HBM conversions
template <typename I, typename N, typename V>
format::Format *CscHIPCscConditionalFunction(format::Format *source, context::Context *to) {
  auto *csc = source->AsAbsolute<format::CSC<I, N, V>>();
  const auto dims = csc->get_dimensions();
  const size_t nnz = csc->get_num_nnz();
  const int did = detail::DeviceOf(to);
  auto &dev = hip::Device::Get(did);
  N *d_cp = dev.Upload(csc->get_col_ptr(), csc->ptr_count() + 1);
  I *d_row = dev.Upload(csc->get_row(), nnz);
  V *d_val = (V *)detail::UploadValues<V>(dev, csc->get_vals(), nnz);
  return new format::HIPCSC<I, N, V>((I)dims[0], (I)dims[1], (N)nnz, d_cp, d_row, d_val, context::HIPContext(did),
                                     format::kOwned, true);
}
template <typename I, typename N, typename V>
format::Format *HIPCscCscConditionalFunction(format::Format *source, context::Context *) {
  auto *d = source->AsAbsolute<format::HIPCSC<I, N, V>>();
  const auto dims = d->get_dimensions();
  const size_t nnz = d->get_num_nnz();
  auto &dev = d->device();
  N *cp = dev.Download(d->get_col_ptr(), d->ptr_count() + 1);
  I *row = dev.Download(d->get_row(), nnz);
  V *val = detail::DownloadValues<V>(dev, d->get_vals(), nnz);
  return new format::CSC<I, N, V>((I)dims[0], (I)dims[1], cp, row, val, format::kOwned, true);
}
template <typename I, typename N, typename V>
format::Format *HIPCooHIPCscFunction(format::Format *source, context::Context *) {
  auto *coo = source->AsAbsolute<format::HIPCOO<I, N, V>>();
  const auto dims = coo->get_dimensions();
  auto &dev = coo->device();
  N *cp = nullptr;
  I *row = nullptr;
  void *val = nullptr;
  detail::ToCscOnDevice<I, N, V, false>(dev, (I)dims[0], (I)dims[1], coo->get_num_nnz(), coo->get_row(), coo->get_col(),
                                        coo->get_vals(), &cp, &row, &val);
  return new format::HIPCSC<I, N, V>((I)dims[0], (I)dims[1], (N)coo->get_num_nnz(), cp, row, (V *)val,
                                     context::HIPContext(dev.id()), format::kOwned, true);
}
template <typename I, typename N, typename V>
format::Format *HIPCsrHIPCscFunction(format::Format *source, context::Context *) {
  auto *csr = source->AsAbsolute<format::HIPCSR<I, N, V>>();
  const auto dims = csr->get_dimensions();
  auto &dev = csr->device();
  N *cp = nullptr;
  I *row = nullptr;
  void *val = nullptr;
  detail::ToCscOnDevice<I, N, V, true>(dev, (I)dims[0], (I)dims[1], csr->get_num_nnz(), csr->get_row_ptr(),
                                       csr->get_col(), csr->get_vals(), &cp, &row, &val);
  return new format::HIPCSC<I, N, V>((I)dims[0], (I)dims[1], (N)csr->get_num_nnz(), cp, row, (V *)val,
                                     context::HIPContext(dev.id()), format::kOwned, true);
}

// ------------------------------------------------------------------ host <-> device copies
template <typename I, typename N, typename V>
format::Format *CsrHIPCsrConditionalFunction(format::Format *source, context::Context *to) {
  auto *csr = source->AsAbsolute<format::CSR<I, N, V>>();
  const auto dims = csr->get_dimensions();
  const size_t n = dims[0], nnz = csr->get_num_nnz();
  const int did = detail::DeviceOf(to);
  auto &dev = hip::Device::Get(did);
  N *d_rp = dev.Upload(csr->get_row_ptr(), n + 1);
  I *d_col = dev.Upload(csr->get_col(), nnz);
  V *d_val = (V *)detail::UploadValues<V>(dev, csr->get_vals(), nnz);
  // host CSR already satisfied its constructor contract: no second check
  return new format::HIPCSR<I, N, V>((I)n, (I)dims[1], (N)nnz, d_rp, d_col, d_val, context::HIPContext(did),
                                     format::kOwned, true);
}
template <typename I, typename N, typename V>
format::Format *HIPCsrCsrConditionalFunction(format::Format *source, context::Context *) {
  auto *d = source->AsAbsolute<format::HIPCSR<I, N, V>>();
  const auto dims = d->get_dimensions();
  const size_t n = dims[0], nnz = d->get_num_nnz();
  auto &dev = d->device();
  N *rp = dev.Download(d->get_row_ptr(), n + 1);
  I *col = dev.Download(d->get_col(), nnz);
  V *val = detail::DownloadValues<V>(dev, d->get_vals(), nnz);
  return new format::CSR<I, N, V>((I)n, (I)dims[1], rp, col, val, format::kOwned, true);
}
template <typename I, typename N, typename V>
format::Format *CooHIPCooConditionalFunction(format::Format *source, context::Context *to) {
  auto *coo = source->AsAbsolute<format::COO<I, N, V>>();
  const auto dims = coo->get_dimensions();
  const size_t nnz = coo->get_num_nnz();
  const int did = detail::DeviceOf(to);
  auto &dev = hip::Device::Get(did);
  I *d_row = dev.Upload(coo->get_row(), nnz);
  I *d_col = dev.Upload(coo->get_col(), nnz);
  V *d_val = (V *)detail::UploadValues<V>(dev, coo->get_vals(), nnz);
  // the source may have been built with ignore_sort: let the device constructor check
  return new format::HIPCOO<I, N, V>((I)dims[0], (I)dims[1], (N)nnz, d_row, d_col, d_val, context::HIPContext(did),
                                     format::kOwned, false);
}
template <typename I, typename N, typename V>
format::Format *HIPCooCooConditionalFunction(format::Format *source, context::Context *) {
  auto *d = source->AsAbsolute<format::HIPCOO<I, N, V>>();
  const auto dims = d->get_dimensions();
  const size_t nnz = d->get_num_nnz();
  auto &dev = d->device();
  I *row = dev.Download(d->get_row(), nnz);
  I *col = dev.Download(d->get_col(), nnz);
  V *val = detail::DownloadValues<V>(dev, d->get_vals(), nnz);
  return new format::COO<I, N, V>((I)dims[0], (I)dims[1], (N)nnz, row, col, val, format::kOwned, true);
}

// ------------------------------------------------------------------ device formats (in HBM)
template <typename I, typename N, typename V, bool MOVE>
format::Format *HIPCooHIPCsrFunction(format::Format *source, context::Context *) {
  auto *coo = source->AsAbsolute<format::HIPCOO<I, N, V>>();
  const auto dims = coo->get_dimensions();
  const I n = (I)dims[0], m = (I)dims[1];
  const size_t nnz = coo->get_num_nnz();
  auto &dev = coo->device();
  const int did = dev.id();
  N *rp = (N *)dev.Malloc(((size_t)n + 1) * sizeof(N));
  I *col = nullptr;
  V *val = nullptr;
  unsigned flags = coo->rows_known_sorted() ? SBX_FLAG_ROWS_SORTED : 0u;
  int rc;
  if (MOVE) {
    rc = sbx_coo_to_csr(dev.handle(), hip::IndexTag<I, N>(), hip::ValueTag<V>(), n, m, (int64_t)nnz, coo->get_row(),
                        nullptr, nullptr, rp, nullptr, nullptr, flags | SBX_FLAG_MOVE);
    if (rc == SBX_OK) {
      col = coo->release_col();
      val = coo->release_vals();
    }
  } else {
    col = (I *)dev.Malloc(nnz * sizeof(I));
    if (coo->get_vals()) val = (V *)dev.Malloc(nnz * hip::ValueBytes<V>());
    rc = sbx_coo_to_csr(dev.handle(), hip::IndexTag<I, N>(), hip::ValueTag<V>(), n, m, (int64_t)nnz, coo->get_row(),
                        coo->get_col(), coo->get_vals(), rp, col, val, flags);
  }
  if (rc != SBX_OK) {
    dev.Free(rp);
    if (!MOVE) {
      dev.Free(col);
      dev.Free((void *)val);
    }
    dev.Check(rc);
  }
  return new format::HIPCSR<I, N, V>(n, m, (N)nnz, rp, col, val, context::HIPContext(did), format::kOwned, false);
}

template <typename I, typename N, typename V, bool MOVE>
format::Format *HIPCsrHIPCooFunction(format::Format *source, context::Context *) {
  auto *csr = source->AsAbsolute<format::HIPCSR<I, N, V>>();
  const auto dims = csr->get_dimensions();
  const I n = (I)dims[0], m = (I)dims[1];
  const size_t nnz = csr->get_num_nnz();
  auto &dev = csr->device();
  const int did = dev.id();
  I *row = (I *)dev.Malloc(nnz * sizeof(I));
  I *col = nullptr;
  V *val = nullptr;
  int rc;
  if (MOVE) {
    rc = sbx_csr_to_coo(dev.handle(), hip::IndexTag<I, N>(), hip::ValueTag<V>(), n, m, (int64_t)nnz, csr->get_row_ptr(),
                        nullptr, nullptr, row, nullptr, nullptr, SBX_FLAG_MOVE);
    if (rc == SBX_OK) {
      col = csr->release_col();
      val = csr->release_vals();
    }
  } else {
    col = (I *)dev.Malloc(nnz * sizeof(I));
    if (csr->get_vals()) val = (V *)dev.Malloc(nnz * hip::ValueBytes<V>());
    rc = sbx_csr_to_coo(dev.handle(), hip::IndexTag<I, N>(), hip::ValueTag<V>(), n, m, (int64_t)nnz, csr->get_row_ptr(),
                        csr->get_col(), csr->get_vals(), row, col, val, 0u);
  }
  if (rc != SBX_OK) {
    dev.Free(row);
    if (!MOVE) {
      dev.Free(col);
      dev.Free((void *)val);
    }
    dev.Check(rc);
  }
  return new format::HIPCOO<I, N, V>(n, m, (N)nnz, row, col, val, context::HIPContext(did), format::kOwned, false);
}

// ------------------------------------------------------------------ registration
template <typename IDType, typename NNZType, typename ValueType>
class ConverterOrderTwo : public ConverterImpl<ConverterOrderTwo<IDType, NNZType, ValueType>> {
 public:
  ConverterOrderTwo() { ResetConverterOrderTwo(); }
  Converter *Clone() const override { return new ConverterOrderTwo(*this); }
  void Reset() override { ResetConverterOrderTwo(); }
  void ResetConverterOrderTwo() {
    using namespace format;
    typedef IDType I;
    typedef NNZType N;
    typedef ValueType V;
    const auto csr = CSR<I, N, V>::get_id_static(), coo = COO<I, N, V>::get_id_static();
    const auto dcsr = HIPCSR<I, N, V>::get_id_static(), dcoo = HIPCOO<I, N, V>::get_id_static();
    const auto csc = CSC<I, N, V>::get_id_static(), dcsc = HIPCSC<I, N, V>::get_id_static();
    // host formats, results on the host (reference converter_order_two.cc:258-340)
    this->RegisterConversionFunction(coo, csr, CooCsrFunctionConditional<I, N, V>, detail::ToCPU);
    this->RegisterConversionFunction(csr, coo, CsrCooFunctionConditional<I, N, V>, detail::ToCPU);
    this->RegisterConversionFunction(coo, csr, CooCsrMoveConditionalFunction<I, N, V>, detail::ToCPU, true);
    this->RegisterConversionFunction(csr, coo, CsrCooMoveConditionalFunction<I, N, V>, detail::ToCPU, true);
    // the reference registers the two CSC functions as copy AND as move conversions (:273-286, :326-340)
    for (bool move : {false, true}) {
      this->RegisterConversionFunction(coo, csc, CooCscFunctionConditional<I, N, V>, detail::ToCPU, move);
      this->RegisterConversionFunction(csr, csc, CsrCscFunctionConditional<I, N, V>, detail::ToCPU, move);
    }
    // host <-> device copies (reference: CsrCUDACsr / CUDACsrCsr, converter_order_two_cuda.cu)
    for (bool move : {false, true}) {
      this->RegisterConversionFunction(csc, dcsc, CscHIPCscConditionalFunction<I, N, V>, detail::ToHIP, move);
      this->RegisterConversionFunction(dcsc, csc, HIPCscCscConditionalFunction<I, N, V>, detail::ToCPU, move);
      this->RegisterConversionFunction(csr, dcsr, CsrHIPCsrConditionalFunction<I, N, V>, detail::ToHIP, move);
      this->RegisterConversionFunction(dcsr, csr, HIPCsrCsrConditionalFunction<I, N, V>, detail::ToCPU, move);
      this->RegisterConversionFunction(coo, dcoo, CooHIPCooConditionalFunction<I, N, V>, detail::ToHIP, move);
      this->RegisterConversionFunction(dcoo, coo, HIPCooCooConditionalFunction<I, N, V>, detail::ToCPU, move);
    }
    // device formats, converted in HBM
    this->RegisterConversionFunction(dcoo, dcsr, HIPCooHIPCsrFunction<I, N, V, false>, detail::OnHIP);
    this->RegisterConversionFunction(dcsr, dcoo, HIPCsrHIPCooFunction<I, N, V, false>, detail::OnHIP);
    this->RegisterConversionFunction(dcoo, dcsr, HIPCooHIPCsrFunction<I, N, V, true>, detail::OnHIP, true);
    this->RegisterConversionFunction(dcsr, dcoo, HIPCsrHIPCooFunction<I, N, V, true>, detail::OnHIP, true);
    for (bool move : {false, true}) {
      this->RegisterConversionFunction(dcoo, dcsc, HIPCooHIPCscFunction<I, N, V>, detail::OnHIP, move);
      this->RegisterConversionFunction(dcsr, dcsc, HIPCsrHIPCscFunction<I, N, V>, detail::OnHIP, move);
    }
  }
};

}  // namespace sparsebase::converter
#endif
