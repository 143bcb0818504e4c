#ifndef SPARSEBASE_CONTEXT_CPU_CONTEXT_H_
#define SPARSEBASE_CONTEXT_CPU_CONTEXT_H_
#include "sparsebase/context/context.h"
namespace sparsebase::context {
struct CPUContext : utils::IdentifiableImplementation<CPUContext, Context> {
  bool IsEquivalent(Context *rhs) const override { return dynamic_cast<CPUContext *>(rhs) != nullptr; }
};
}  // namespace sparsebase::context
#endif
