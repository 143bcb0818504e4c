// Execution-place tags (reference: context/context.h:18-21, cpu_context.h:12-14,
// cuda_context_cuda.cuh:15-19 -> HIPContext).
#ifndef SPARSEBASE_CONTEXT_CONTEXT_H_
#define SPARSEBASE_CONTEXT_CONTEXT_H_
#include "sparsebase/config.h"
#include "sparsebase/utils/utils.h"

namespace sparsebase::context {
struct Context : public utils::Identifiable {
  virtual bool IsEquivalent(Context *) const = 0;
  ~Context() override = default;
};
}  // namespace sparsebase::context
#endif
