// sparsebase/feature/degree_distribution.h — feature::DegreeDistribution (reference: feature/degree_distribution.h:20-105, .cc:13-167)
// Every feature class registers two implementations: {CSR} stages the host arrays through the
// default device, {HIPCSR} runs in place in HBM; both end in the same sbx_csr_* entry point.
#ifndef SPARSEBASE_FEATURE_DEGREE_DISTRIBUTION_H_
#define SPARSEBASE_FEATURE_DEGREE_DISTRIBUTION_H_
#include <tuple>

#include "sparsebase/feature/feature_preprocess_type.h"
#include "sparsebase/format/csr.h"
#include "sparsebase/format/hip_formats.h"
#include "sparsebase/reorder/reorderer.h"

namespace sparsebase::feature {
struct DegreeDistributionParams : utils::Parameters {};

template <typename IDType, typename NNZType, typename ValueType, typename FeatureType>
class DegreeDistribution : public FeaturePreprocessType<FeatureType *> {
  typedef reorder::detail::DeviceCsrView<IDType, NNZType, ValueType> View;

 public:
  typedef DegreeDistributionParams ParamsType;
  DegreeDistribution() {
    Register();
    this->params_ = std::shared_ptr<ParamsType>(new ParamsType());
    this->pmap_.insert({get_id_static(), this->params_});
  }
  DegreeDistribution(ParamsType) : DegreeDistribution() {}
  DegreeDistribution(const DegreeDistribution &d) {
    Register();
    this->params_ = d.params_;
    this->pmap_ = d.pmap_;
  }
  DegreeDistribution(std::shared_ptr<ParamsType> p) {
    Register();
    this->params_ = p;
    this->pmap_[get_id_static()] = p;
  }
  ~DegreeDistribution() override = default;

  std::unordered_map<std::type_index, std::any> Extract(format::Format *format, std::vector<context::Context *> c,
                                                        bool convert_input) override {
    return {{this->get_id(), std::forward<FeatureType *>(GetDistribution(format, c, convert_input))}};
  }
  std::vector<std::type_index> get_sub_ids() override { return {typeid(DegreeDistribution<IDType, NNZType, ValueType, FeatureType>)}; }
  std::vector<utils::Extractable *> get_subs() override { return {new DegreeDistribution<IDType, NNZType, ValueType, FeatureType>(*this)}; }
  static std::type_index get_id_static() { return typeid(DegreeDistribution<IDType, NNZType, ValueType, FeatureType>); }

  FeatureType * GetDistribution(format::Format *format, std::vector<context::Context *> c, bool convert_input) {
    return this->Execute(this->params_.get(), c, convert_input, format);
  }
  std::tuple<std::vector<std::vector<format::Format *>>, FeatureType *> GetDistributionCached(format::Format *format,
      std::vector<context::Context *> c, bool convert_input) {
    return this->CachedExecute(this->params_.get(), c, convert_input, false, format);
  }

  // dist[i] = degree(i) / (FeatureType)nnz (degree_distribution.cc:152-167); new FeatureType[n], caller frees
  // with delete[].  The object::Graph overload of the reference (:135-150) is outside this library's scope.
  static FeatureType *Run(View v) {
    static_assert(std::is_same_v<FeatureType, float> || std::is_same_v<FeatureType, double>,
                  "FeatureType must be float or double");
    hip::Staged<FeatureType> d_out(*v.dev, (size_t)v.n);
    const int rc = sbx_csr_degree_distribution(v.dev->handle(), hip::IndexTag<IDType, NNZType>(), v.n, v.nnz, v.row_ptr,
                                               (int)sizeof(FeatureType), d_out.get());
    FeatureType *out = nullptr;
    if (rc == SBX_OK) out = v.dev->Download(d_out.get(), (size_t)v.n);
    v.Release();
    v.dev->Check(rc);
    return out;
  }
  static FeatureType *GetDegreeDistributionCSR(std::vector<format::Format *> formats, utils::Parameters *p) {
    return OnHostCSR(formats, p);
  }

 protected:
  void Register() {
    this->RegisterFunction({format::CSR<IDType, NNZType, ValueType>::get_id_static()}, OnHostCSR);
    this->RegisterFunction({format::HIPCSR<IDType, NNZType, ValueType>::get_id_static()}, OnDeviceCSR);
  }
  static FeatureType * OnHostCSR(std::vector<format::Format *> formats, utils::Parameters *) {
    return Run(View::Stage(formats[0]->AsAbsolute<format::CSR<IDType, NNZType, ValueType>>(), false));
  }
  static FeatureType * OnDeviceCSR(std::vector<format::Format *> formats, utils::Parameters *) {
    return Run(View::Borrow(formats[0]->AsAbsolute<format::HIPCSR<IDType, NNZType, ValueType>>()));
  }
};

}  // namespace sparsebase::feature
#endif
