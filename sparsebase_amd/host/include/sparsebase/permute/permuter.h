// Permuter<In,Out> (reference: permute/permuter.h:23-102, permuter.cc:7-48).
#ifndef SPARSEBASE_PERMUTE_PERMUTER_H_
#define SPARSEBASE_PERMUTE_PERMUTER_H_
#include "sparsebase/reorder/reorderer.h"

namespace sparsebase::permute {
template <typename InputFormatType, typename ReturnFormatType>
class Permuter : public utils::FunctionMatcherMixin<ReturnFormatType *> {
 public:
  Permuter() {
    static_assert(std::is_base_of<format::Format, InputFormatType>::value, "Permuter must take a Format");
    static_assert(std::is_base_of<format::Format, ReturnFormatType>::value, "Permuter must return a Format");
  }
  ReturnFormatType *GetPermutation(format::Format *f, std::vector<context::Context *> contexts, bool convert_input) {
    return this->Execute(this->params_.get(), contexts, convert_input, f);
  }
  ReturnFormatType *GetPermutation(format::Format *f, utils::Parameters *params,
                                   std::vector<context::Context *> contexts, bool convert_input) {
    return this->Execute(params, contexts, convert_input, f);
  }
  std::tuple<std::vector<std::vector<format::Format *>>, ReturnFormatType *> GetPermutationCached(
      format::Format *f, std::vector<context::Context *> contexts, bool convert_input) {
    return this->CachedExecute(this->params_.get(), contexts, convert_input, false, f);
  }
  std::tuple<std::vector<std::vector<format::Format *>>, ReturnFormatType *> GetPermutationCached(
      format::Format *f, utils::Parameters *params, std::vector<context::Context *> contexts, bool convert_input) {
    return this->CachedExecute(params, contexts, convert_input, false, f);
  }
  virtual ~Permuter() = default;
};
}  // namespace sparsebase::permute
#endif
