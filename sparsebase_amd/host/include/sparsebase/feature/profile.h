// sparsebase/feature/profile.h — feature::Profile (reference: feature/profile.h, profile.cc:13-105)
// Every feature class registers two implementations: {CSR} stages the host arrays through the
// default device, {HIPCSR} runs in place in HBM; both end in the same sbx_csr_* entry point.
#ifndef SPARSEBASE_FEATURE_PROFILE_H_
#define SPARSEBASE_FEATURE_PROFILE_H_
#include <tuple>

#include "sparsebase/feature/feature_preprocess_type.h"
#include "sparsebase/format/csr.h"
#include "sparsebase/format/hip_formats.h"
#include "sparsebase/reorder/reorderer.h"

namespace sparsebase::feature {

template <typename IDType, typename NNZType, typename ValueType>
class Profile : public FeaturePreprocessType<IDType *> {
  typedef reorder::detail::DeviceCsrView<IDType, NNZType, ValueType> View;

 public:
  typedef utils::Parameters ParamsType;
  Profile() {
    Register();
    this->params_ = std::shared_ptr<ParamsType>(new ParamsType());
    this->pmap_.insert({get_id_static(), this->params_});
  }
  Profile(ParamsType) : Profile() {}
  Profile(const Profile &d) {
    Register();
    this->params_ = d.params_;
    this->pmap_ = d.pmap_;
  }
  Profile(std::shared_ptr<ParamsType> p) {
    Register();
    this->params_ = p;
    this->pmap_[get_id_static()] = p;
  }
  ~Profile() override = default;

  std::unordered_map<std::type_index, std::any> Extract(format::Format *format, std::vector<context::Context *> c,
                                                        bool convert_input) override {
    return {{this->get_id(), std::forward<IDType *>(GetProfile(format, c, convert_input))}};
  }
  std::vector<std::type_index> get_sub_ids() override { return {typeid(Profile<IDType, NNZType, ValueType>)}; }
  std::vector<utils::Extractable *> get_subs() override { return {new Profile<IDType, NNZType, ValueType>(*this)}; }
  static std::type_index get_id_static() { return typeid(Profile<IDType, NNZType, ValueType>); }

  IDType * GetProfile(format::Format *format, std::vector<context::Context *> c, bool convert_input) {
    return this->Execute(this->params_.get(), c, convert_input, format);
  }
  std::tuple<std::vector<std::vector<format::Format *>>, IDType *> GetProfileCached(format::Format *format,
      std::vector<context::Context *> c, bool convert_input) {
    return this->CachedExecute(this->params_.get(), c, convert_input, false, format);
  }

  // sum over rows of i - min(i, smallest column) (profile.cc:91-105), accumulated exactly on the device and
  // narrowed to IDType here, which is where the reference's IDType accumulator wraps; caller frees with delete
  static IDType *Run(View v) {
    int64_t sum = 0;
    const int rc = sbx_csr_profile(v.dev->handle(), hip::IndexTag<IDType, NNZType>(), v.n, v.nnz, v.row_ptr, v.col, &sum);
    v.Release();
    v.dev->Check(rc);
    return new IDType((IDType)sum);
  }
  static IDType *GetProfileCSR(std::vector<format::Format *> formats, utils::Parameters *p) { return OnHostCSR(formats, p); }

 protected:
  void Register() {
    this->RegisterFunction({format::CSR<IDType, NNZType, ValueType>::get_id_static()}, OnHostCSR);
    this->RegisterFunction({format::HIPCSR<IDType, NNZType, ValueType>::get_id_static()}, OnDeviceCSR);
  }
  static IDType * OnHostCSR(std::vector<format::Format *> formats, utils::Parameters *) {
    return Run(View::Stage(formats[0]->AsAbsolute<format::CSR<IDType, NNZType, ValueType>>(), false));
  }
  static IDType * OnDeviceCSR(std::vector<format::Format *> formats, utils::Parameters *) {
    return Run(View::Borrow(formats[0]->AsAbsolute<format::HIPCSR<IDType, NNZType, ValueType>>()));
  }
};

}  // namespace sparsebase::feature
#endif
