// HIPContext — the device execution place (reference: CUDAContext,
// context/cuda_context_cuda.cuh:15-19, .cu:9-21: validity check + same-device equivalence).
#ifndef SPARSEBASE_CONTEXT_HIP_CONTEXT_H_
#define SPARSEBASE_CONTEXT_HIP_CONTEXT_H_
#include "sparsebase/context/context.h"
#include "sparsebase/hip/device.h"
namespace sparsebase::context {
struct HIPContext : utils::IdentifiableImplementation<HIPContext, Context> {
  int device_id;
  explicit HIPContext(int did) : device_id(did) {
    const int count = hip::DeviceCount();
    if (did < 0 || did >= count) throw utils::HIPDeviceException(count, did);
  }
  bool IsEquivalent(Context *rhs) const override {
    auto *other = dynamic_cast<HIPContext *>(rhs);
    return other != nullptr && other->device_id == device_id;
  }
};
}  // namespace sparsebase::context
#endif
