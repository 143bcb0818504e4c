// sparsebase/feature/bandwidth.h — feature::Bandwidth (reference: feature/bandwidth.h:12-72, bandwidth.cc:13-112)
// Every feature class registers two implementations: {CSR} stages the host arrays through the
// default device, {HIPCSR} runs in place in HBM; both end in the same sbx_csr_* entry point.
#ifndef SPARSEBASE_FEATURE_BANDWIDTH_H_
#define SPARSEBASE_FEATURE_BANDWIDTH_H_
#include <tuple>

#include "sparsebase/feature/feature_preprocess_type.h"
#include "sparsebase/format/csr.h"
#include "sparsebase/format/hip_formats.h"
#include "sparsebase/reorder/reorderer.h"

namespace sparsebase::feature {

template <typename IDType, typename NNZType, typename ValueType>
class Bandwidth : public FeaturePreprocessType<int *> {
  typedef reorder::detail::DeviceCsrView<IDType, NNZType, ValueType> View;

 public:
  typedef utils::Parameters ParamsType;
  Bandwidth() {
    Register();
    this->params_ = std::shared_ptr<ParamsType>(new ParamsType());
    this->pmap_.insert({get_id_static(), this->params_});
  }
  Bandwidth(ParamsType) : Bandwidth() {}
  Bandwidth(const Bandwidth &d) {
    Register();
    this->params_ = d.params_;
    this->pmap_ = d.pmap_;
  }
  Bandwidth(std::shared_ptr<ParamsType> p) {
    Register();
    this->params_ = p;
    this->pmap_[get_id_static()] = p;
  }
  ~Bandwidth() override = default;

  std::unordered_map<std::type_index, std::any> Extract(format::Format *format, std::vector<context::Context *> c,
                                                        bool convert_input) override {
    return {{this->get_id(), std::forward<int *>(GetBandwidth(format, c, convert_input))}};
  }
  std::vector<std::type_index> get_sub_ids() override { return {typeid(Bandwidth<IDType, NNZType, ValueType>)}; }
  std::vector<utils::Extractable *> get_subs() override { return {new Bandwidth<IDType, NNZType, ValueType>(*this)}; }
  static std::type_index get_id_static() { return typeid(Bandwidth<IDType, NNZType, ValueType>); }

  int * GetBandwidth(format::Format *format, std::vector<context::Context *> c, bool convert_input) {
    return this->Execute(this->params_.get(), c, convert_input, format);
  }
  std::tuple<std::vector<std::vector<format::Format *>>, int *> GetBandwidthCached(format::Format *format,
      std::vector<context::Context *> c, bool convert_input) {
    return this->CachedExecute(this->params_.get(), c, convert_input, false, format);
  }

  // max over the nonzeros of |i - j| + 1 (bandwidth.cc:93-112); caller frees with delete
  static int *Run(View v) {
    int64_t bw = 0;
    const int rc = sbx_csr_bandwidth(v.dev->handle(), hip::IndexTag<IDType, NNZType>(), v.n, v.nnz, v.row_ptr, v.col, &bw);
    v.Release();
    v.dev->Check(rc);
    return new int((int)bw);
  }
  // the reference's name for the {CSR} implementation (bandwidth.h:66-67)
  static int *GetBandwidthCSR(std::vector<format::Format *> formats, utils::Parameters *p) { return OnHostCSR(formats, p); }

 protected:
  void Register() {
    this->RegisterFunction({format::CSR<IDType, NNZType, ValueType>::get_id_static()}, OnHostCSR);
    this->RegisterFunction({format::HIPCSR<IDType, NNZType, ValueType>::get_id_static()}, OnDeviceCSR);
  }
  static int * OnHostCSR(std::vector<format::Format *> formats, utils::Parameters *) {
    return Run(View::Stage(formats[0]->AsAbsolute<format::CSR<IDType, NNZType, ValueType>>(), false));
  }
  static int * OnDeviceCSR(std::vector<format::Format *> formats, utils::Parameters *) {
    return Run(View::Borrow(formats[0]->AsAbsolute<format::HIPCSR<IDType, NNZType, ValueType>>()));
  }
};

}  // namespace sparsebase::feature
#endif
