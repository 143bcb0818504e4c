// sparsebase/feature/degrees.h — feature::Degrees (reference: feature/degrees.h, degrees.cc:13-105)
// Every feature class registers two implementations: {CSR} stages the host arrays through the
// default device, {HIPCSR} runs in place in HBM; both end in the same sbx_csr_* entry point.
#ifndef SPARSEBASE_FEATURE_DEGREES_H_
#define SPARSEBASE_FEATURE_DEGREES_H_
#include <tuple>

#include "sparsebase/feature/feature_preprocess_type.h"
#include "sparsebase/format/csr.h"
#include "sparsebase/format/hip_formats.h"
#include "sparsebase/reorder/reorderer.h"

namespace sparsebase::feature {
struct DegreesParams : utils::Parameters {};

template <typename IDType, typename NNZType, typename ValueType>
class Degrees : public FeaturePreprocessType<IDType *> {
  typedef reorder::detail::DeviceCsrView<IDType, NNZType, ValueType> View;

 public:
  typedef DegreesParams ParamsType;
  Degrees() {
    Register();
    this->params_ = std::shared_ptr<ParamsType>(new ParamsType());
    this->pmap_.insert({get_id_static(), this->params_});
  }
  Degrees(ParamsType) : Degrees() {}
  Degrees(const Degrees &d) {
    Register();
    this->params_ = d.params_;
    this->pmap_ = d.pmap_;
  }
  Degrees(std::shared_ptr<ParamsType> p) {
    Register();
    this->params_ = p;
    this->pmap_[get_id_static()] = p;
  }
  ~Degrees() override = default;

  std::unordered_map<std::type_index, std::any> Extract(format::Format *format, std::vector<context::Context *> c,
                                                        bool convert_input) override {
    return {{this->get_id(), std::forward<IDType *>(GetDegrees(format, c, convert_input))}};
  }
  std::vector<std::type_index> get_sub_ids() override { return {typeid(Degrees<IDType, NNZType, ValueType>)}; }
  std::vector<utils::Extractable *> get_subs() override { return {new Degrees<IDType, NNZType, ValueType>(*this)}; }
  static std::type_index get_id_static() { return typeid(Degrees<IDType, NNZType, ValueType>); }

  IDType * GetDegrees(format::Format *format, std::vector<context::Context *> c, bool convert_input) {
    return this->Execute(this->params_.get(), c, convert_input, format);
  }
  std::tuple<std::vector<std::vector<format::Format *>>, IDType *> GetDegreesCached(format::Format *format,
      std::vector<context::Context *> c, bool convert_input) {
    return this->CachedExecute(this->params_.get(), c, convert_input, false, format);
  }

  // degrees[i] = row_ptr[i+1] - row_ptr[i] (degrees.cc:93-105); new IDType[n], caller frees with delete[]
  static IDType *Run(View v) {
    hip::Staged<IDType> d_out(*v.dev, (size_t)v.n);
    const int rc = sbx_csr_degrees(v.dev->handle(), hip::IndexTag<IDType, NNZType>(), v.n, v.row_ptr, d_out.get());
    IDType *out = nullptr;
    if (rc == SBX_OK) out = v.dev->Download(d_out.get(), (size_t)v.n);
    v.Release();
    v.dev->Check(rc);
    return out;
  }
  static IDType *GetDegreesCSR(std::vector<format::Format *> formats, utils::Parameters *p) { return OnHostCSR(formats, p); }

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
