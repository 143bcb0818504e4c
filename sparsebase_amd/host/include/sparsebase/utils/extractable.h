// sparsebase/utils/extractable.h — interface of feature extractors that can be fused by an
// Extractor (reference: utils/extractable.h:32-104).
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
  // features of `format` as {type_index of the feature class -> std::any holding the result}
  virtual std::unordered_map<std::type_index, std::any> Extract(format::Format *format,
                                                                std::vector<context::Context *> contexts,
                                                                bool convert_input) = 0;
  virtual std::type_index get_id() = 0;
  virtual std::vector<std::type_index> get_sub_ids() = 0;  // the classes fused into this one
  virtual std::vector<Extractable *> get_subs() = 0;       // fresh instances of them (caller owns)
  virtual std::shared_ptr<utils::Parameters> get_params() = 0;
  virtual std::shared_ptr<utils::Parameters> get_params(std::type_index feature_extractor) = 0;
  virtual void set_params(std::type_index feature_extractor, std::shared_ptr<utils::Parameters> params) = 0;
  virtual ~Extractable() = default;

 protected:
  std::shared_ptr<utils::Parameters> params_;
  std::unordered_map<std::type_index, std::shared_ptr<utils::Parameters>> pmap_;
};

}  // namespace sparsebase::utils
#endif
