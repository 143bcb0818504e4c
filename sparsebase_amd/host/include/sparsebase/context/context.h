// Execution-place tags.  A context says WHERE a format lives: the host (CPUContext, below) or one
// GPU (HIPContext, hip_context.h).  Reference counterparts: context/context.h:18-21,
// cpu_context.h:12-14, cuda_context_cuda.cuh:15-19.
#ifndef SPARSEBASE_CONTEXT_CONTEXT_H_
#define SPARSEBASE_CONTEXT_CONTEXT_H_
#include "sparsebase/config.h"
#include "sparsebase/utils/utils.h"

namespace sparsebase::context {

// Two contexts are equivalent when data in one is directly usable in the other (same place).
struct Context : public utils::Identifiable {
  ~Context() override = default;
  virtual bool IsEquivalent(Context *other) const = 0;
};

// every host context is the same place
struct CPUContext : utils::IdentifiableImplementation<CPUContext, Context> {
  bool IsEquivalent(Context *other) const override { return dynamic_cast<CPUContext *>(other) != nullptr; }
};

}  // namespace sparsebase::context
#endif
