// GrayReorder (reference: reorder/gray_reorder.h:13-66, gray_reorder.cc:106-424).
//
// Two stages:
//   device  sbx_gray_row_keys — one pass over the nonzeros: per-row degree,
//           Gray-decoded block-occupancy key and the four band counters
//           (everything in the reference that is O(nnz));
//   host    the ordering of the n row keys.  The reference orders rows with unstable
//           std::sort calls on heavily tied keys (gray_reorder.cc:199,294-299,355-358,
//           404); its result therefore depends on libstdc++'s introsort visiting
//           order, which no parallel sort reproduces.  To stay bit-exact this stage
//           issues the same std::sort calls, on the same sequences, over the
//           device-computed keys (O(n log n) on rows, nothing touches the nonzeros).
#ifndef SPARSEBASE_REORDER_GRAY_REORDER_H_
#define SPARSEBASE_REORDER_GRAY_REORDER_H_
#include <chrono>
#include <algorithm>
#include <utility>
#include <vector>

#include "sparsebase/reorder/reorderer.h"

namespace sparsebase::reorder {

enum BitMapSize { BitSize16 = 16, BitSize32 = 32, BitSize64 = 64 };

struct GrayReorderParams : utils::Parameters {
  BitMapSize resolution;
  int nnz_threshold;
  int sparse_density_group_size;
  explicit GrayReorderParams() : resolution(BitSize32), nnz_threshold(0), sparse_density_group_size(1) {}
  GrayReorderParams(BitMapSize r, int nnz_thresh, int group_size)
      : resolution(r), nnz_threshold(nnz_thresh), sparse_density_group_size(group_size) {}
};

template <typename IDType, typename NNZType, typename ValueType>
class GrayReorder : public Reorderer<IDType> {
  typedef std::pair<IDType, unsigned long> row_grey_pair;

 public:
  typedef GrayReorderParams ParamsType;
  GrayReorder(BitMapSize resolution, int nnz_threshold, int sparse_density_group_size) {
    this->params_ = std::make_unique<GrayReorderParams>(resolution, nnz_threshold, sparse_density_group_size);
    this->RegisterFunction({format::CSR<IDType, NNZType, ValueType>::get_id_static()}, GrayReorderingCSR);
    this->RegisterFunction({format::HIPCSR<IDType, NNZType, ValueType>::get_id_static()}, GrayReorderingHIPCSR);
  }
  explicit GrayReorder(GrayReorderParams p)
      : GrayReorder(p.resolution, p.nnz_threshold, p.sparse_density_group_size) {}
  /// Wall time of the last call's stages in this process, in ms: the device key stage (sbx_gray_row_keys, which ends
  /// with a blocking read-back), the copy of degrees and keys to the host, the host ordering stage.  (Not in the
  /// reference: what bench.py reports as Gray end to end.)
  static double *last_stage_ms() {
    static thread_local double ms[3] = {0, 0, 0};  // (per thread: concurrent reorder calls do not share it)
    return ms;
  }

 protected:
  static bool desc_comparator(const row_grey_pair &l, const row_grey_pair &r) { return l.second > r.second; }
  static bool asc_comparator(const row_grey_pair &l, const row_grey_pair &r) { return l.second < r.second; }

  static IDType *Run(detail::DeviceCsrView<IDType, NNZType, ValueType> v, utils::Parameters *poly) {
    auto *params = static_cast<GrayReorderParams *>(poly);
    const int64_t n = v.n;
    using clock = std::chrono::steady_clock;
    auto ms_since = [](clock::time_point t) { return std::chrono::duration<double, std::milli>(clock::now() - t).count(); };
    // ---- device stage
    std::vector<IDType> deg((size_t)n);
    std::vector<uint64_t> key((size_t)n);
    int64_t counts[4] = {0, 0, 0, 0};
    {
      hip::Staged<IDType> d_deg(*v.dev, (size_t)n);
      hip::Staged<uint64_t> d_key(*v.dev, (size_t)n);
      auto t0 = clock::now();
      const int rc = sbx_gray_row_keys(v.dev->handle(), hip::IndexTag<IDType>(), v.n, v.m, v.nnz, v.row_ptr, v.col,
                                       (int)params->resolution, params->nnz_threshold, d_deg.get(), d_key.get(),
                                       counts);  // (returns after its last read-back: the stage is complete)
      last_stage_ms()[0] = ms_since(t0);
      t0 = clock::now();
      if (rc == SBX_OK && n > 0) {
        d_deg.ToHost(deg.data());
        d_key.ToHost(key.data());
      }
      last_stage_ms()[1] = ms_since(t0);
      v.Release();
      v.dev->Check(rc);
    }
    const auto t_host = clock::now();
    struct HostStageTimer {  // (the ordering stage below returns from several places)
      clock::time_point t;
      ~HostStageTimer() { last_stage_ms()[2] = std::chrono::duration<double, std::milli>(clock::now() - t).count(); }
    } host_stage_timer{t_host};
    // ---- host ordering stage (see header comment)
    const int group_size = params->sparse_density_group_size;
    std::vector<IDType> sparse_rows, dense_rows;
    sparse_rows.reserve((size_t)n);
    for (int64_t i = 0; i < n; i++) {
      if (deg[i] <= (IDType)params->nnz_threshold) sparse_rows.push_back((IDType)i);
      else dense_rows.push_back((IDType)i);
    }
    // the reference keeps these counters in `int` (gray_reorder.cc:134-137)
    const bool sparse_banded = double((int)counts[1]) / (int)counts[0] > 0.3;
    const bool dense_banded = double((int)counts[3]) / (int)counts[2] > 0.2;
    std::sort(sparse_rows.begin(), sparse_rows.end(), [&](int a, int b) -> bool { return deg[a] < deg[b]; });

    std::vector<row_grey_pair> section;
    section.reserve((size_t)n);
    if (!sparse_banded) {
      bool descending = false;
      int64_t start = 0;
      IDType last_deg = 0;
      int groups = 0;
      const int64_t ns = (int64_t)sparse_rows.size();
      auto flush = [&](int64_t end) {
        if (!descending) std::sort(section.begin(), section.end(), asc_comparator);
        else std::sort(section.begin(), section.end(), desc_comparator);
        descending = !descending;
        for (int64_t a = start; a < end; a++) sparse_rows[a] = section[a - start].first;
      };
      for (int64_t i = 0; i < ns; i++) {
        const IDType d = deg[sparse_rows[i]];
        if (i == 0) {
          last_deg = d;
          start = 0;
        }
        if (d == 0) {  // empty rows never enter a section
          start = i + 1;
          if (i + 1 < ns) last_deg = deg[sparse_rows[i + 1]];
          continue;
        }
        if (i != 0 && last_deg != d) {
          groups++;
          last_deg = d;
          if (groups == group_size) {
            flush(i);
            start = i;
            section.clear();
            groups = 0;
          }
        }
        section.push_back(row_grey_pair(sparse_rows[i], (unsigned long)key[sparse_rows[i]]));
        if (i == ns - 1) flush(ns);
      }
      section.clear();
    }
    if (!dense_banded) {
      for (IDType r : dense_rows) section.push_back(row_grey_pair(r, (unsigned long)key[r]));
      std::sort(section.begin(), section.end(), asc_comparator);
      for (size_t a = 0; a < dense_rows.size(); a++) dense_rows[a] = section[a].first;
    }
    IDType *order = new IDType[n > 0 ? n : 1]();
    int64_t pos = 0;
    for (IDType r : sparse_rows) order[r] = (IDType)pos++;
    for (IDType r : dense_rows) order[r] = (IDType)pos++;
    return order;
  }
  static IDType *GrayReorderingCSR(std::vector<format::Format *> formats, utils::Parameters *params) {
    auto *csr = formats[0]->AsAbsolute<format::CSR<IDType, NNZType, ValueType>>();
    return Run(detail::DeviceCsrView<IDType, NNZType, ValueType>::Stage(csr, false), params);
  }
  static IDType *GrayReorderingHIPCSR(std::vector<format::Format *> formats, utils::Parameters *params) {
    auto *csr = formats[0]->AsAbsolute<format::HIPCSR<IDType, NNZType, ValueType>>();
    return Run(detail::DeviceCsrView<IDType, NNZType, ValueType>::Borrow(csr), params);
  }
};

}  // namespace sparsebase::reorder
#endif
