// sparsebase/utils/extractable.h — what a feature extractor has to offer so that several of them can be
// fused and driven through one interface (reference: utils/extractable.h:32-104): its identity, the
// identities and fresh instances of the extractors fused into it, their parameter objects, and the
// extraction itself, which returns {type_index of the feature class -> std::any holding the result}.
#ifndef SPARSEBASE_UTILS_EXTRACTABLE_H_
#define SPARSEBASE_UTILS_EXTRACTABLE_H_
#include <any>
#include <memory>
#include <typeindex>
#include <unordered_map>
#include <vector>

#include "sparsebase/context/context.h"
#include "sparsebase/format/format.h"
#include "sparsebase/utils/parameterizable.h"

namespace sparsebase::utils {

class Extractable {
 public:
  typedef std::unordered_map<std::type_index, std::any> FeatureMap;
  typedef std::shared_ptr<utils::Parameters> ParamsPtr;

  virtual ~Extractable() = default;
  // identity
  virtual std::type_index get_id() = 0;
  virtual std::vector<std::type_index> get_sub_ids() = 0;
  virtual std::vector<Extractable *> get_subs() = 0;  // caller owns the instances
  // parameters: of this object, of one fused extractor
  virtual ParamsPtr get_params() = 0;
  virtual ParamsPtr get_params(std::type_index feature_extractor) = 0;
  virtual void set_params(std::type_index feature_extractor, ParamsPtr params) = 0;
  // the work
  virtual FeatureMap Extract(format::Format *format, std::vector<context::Context *> contexts, bool convert_input) = 0;

 protected:
  ParamsPtr params_;
  std::unordered_map<std::type_index, ParamsPtr> pmap_;
};

}  // namespace sparsebase::utils
#endif
