// HIPCommunicator — the ranks of one node that share a sharded operator, one process and one HIPContext per GPU.
// (The reference has no multi-GPU operator; its device-to-device edge is the peer copy of
// converter/converter_order_two_cuda.cu:41-76.)  RAII over the C ABI's sbx_comm_t: RCCL (ncclAllGather over xGMI,
// called inside libsbx) or a caller-supplied all-gather hook.
#ifndef SPARSEBASE_CONTEXT_HIP_COMMUNICATOR_H_
#define SPARSEBASE_CONTEXT_HIP_COMMUNICATOR_H_
#include <array>

#include "sparsebase/context/hip_context.h"

namespace sparsebase::context {

class HIPCommunicator {
 public:
  typedef std::array<char, SBX_COMM_ID_BYTES> Id;
  // rank 0 creates the id and hands it to the other ranks (file, socket, MPI, environment ...)
  static Id NewId() {
    Id id;
    const int rc = sbx_comm_unique_id(id.data());
    if (rc != SBX_OK) throw utils::HIPDeviceException(std::string("sbx_comm_unique_id: ") + sbx_status_string(rc));
    return id;
  }
  HIPCommunicator(const HIPContext &ctx, int rank, int world, const Id &id) : device_(ctx.device_id) {
    const int rc = sbx_comm_create_rccl(ctx.device_id, rank, world, id.data(), &comm_);
    if (rc != SBX_OK) throw utils::HIPDeviceException(std::string("sbx_comm_create_rccl: ") + sbx_status_string(rc));
  }
  HIPCommunicator(const HIPContext &ctx, int rank, int world, sbx_allgather_fn allgather, void *user)
      : device_(ctx.device_id) {
    const int rc = sbx_comm_create(rank, world, allgather, user, &comm_);
    if (rc != SBX_OK) throw utils::HIPDeviceException(std::string("sbx_comm_create: ") + sbx_status_string(rc));
  }
  ~HIPCommunicator() { sbx_comm_destroy(comm_); }
  HIPCommunicator(const HIPCommunicator &) = delete;
  HIPCommunicator &operator=(const HIPCommunicator &) = delete;
  sbx_comm_t handle() const { return comm_; }
  int device() const { return device_; }
  int rank() const {
    int r = 0, w = 0;
    sbx_comm_rank(comm_, &r, &w);
    return r;
  }
  int world() const {
    int r = 0, w = 0;
    sbx_comm_rank(comm_, &r, &w);
    return w;
  }

 private:
  sbx_comm_t comm_ = nullptr;
  int device_;
};

}  // namespace sparsebase::context
#endif
