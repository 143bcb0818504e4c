// GenericReorder: an empty reorderer users register their own functions into
// (reference: reorder/generic_reorder.h:13-18).
#ifndef SPARSEBASE_REORDER_GENERIC_REORDER_H_
#define SPARSEBASE_REORDER_GENERIC_REORDER_H_
#include "sparsebase/reorder/reorderer.h"
namespace sparsebase::reorder {
template <typename IDType, typename NNZType, typename ValueType>
class GenericReorder : public Reorderer<IDType> {
 public:
  typedef utils::Parameters ParamsType;
  GenericReorder() = default;
  // ReorderBase::Reorder<GenericReorder>(params, ...) constructs the operator from its parameter object
  explicit GenericReorder(ParamsType) {}
};
}  // namespace sparsebase::reorder
#endif
