// sparsebase/feature/feature_preprocess_type.h — an Extractable with function matching
// (reference: feature/feature_preprocess_type.h:17-27, .cc:9-40).
#ifndef SPARSEBASE_FEATURE_FEATURE_PREPROCESS_TYPE_H_
#define SPARSEBASE_FEATURE_FEATURE_PREPROCESS_TYPE_H_
#include <algorithm>
#include <memory>

#include "sparsebase/utils/exception.h"
#include "sparsebase/utils/extractable.h"
#include "sparsebase/utils/function_matcher_mixin.h"

namespace sparsebase::feature {

template <typename FeatureType>
class FeaturePreprocessType : public utils::FunctionMatcherMixin<FeatureType, utils::Extractable> {
 public:
  std::shared_ptr<utils::Parameters> get_params() override { return this->params_; }
  std::shared_ptr<utils::Parameters> get_params(std::type_index t) override {
    auto it = this->pmap_.find(t);
    if (it == this->pmap_.end()) throw utils::FeatureParamsException(get_id().name(), t.name());
    return it->second;
  }
  void set_params(std::type_index t, std::shared_ptr<utils::Parameters> p) override {
    auto ids = this->get_sub_ids();
    if (std::find(ids.begin(), ids.end(), t) == ids.end())
      throw utils::FeatureParamsException(get_id().name(), t.name());
    this->pmap_[t] = p;
  }
  std::type_index get_id() override { return typeid(*this); }
  ~FeaturePreprocessType() override = default;
};

}  // namespace sparsebase::feature
#endif
