// sparsebase/hip/device.h — the only place of the host layer that touches the C
// ABI's handle/memory entry points (include/sbx.h).  One sbx handle per device,
// created on first use; every failure becomes a utils::HIPDeviceException — there
// is no CPU fallback anywhere behind these calls.
#ifndef SPARSEBASE_HIP_DEVICE_H_
#define SPARSEBASE_HIP_DEVICE_H_
#include <cstdlib>
#include <map>
#include <type_traits>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "sbx.h"
#include "sparsebase/utils/exception.h"

namespace sparsebase::hip {

inline int DeviceCount() {
  int c = 0;
  if (sbx_device_count(&c) != SBX_OK) return 0;
  return c;
}

// device used when host-resident formats are staged through the GPU
inline int DefaultDevice() {
  const char *e = std::getenv("SBX_DEVICE");
  return e ? std::atoi(e) : 0;
}

class Device {
 public:
  static Device &Get(int device_id) {
    static std::mutex mu;
    static std::map<int, Device *> devices;
    std::lock_guard<std::mutex> lock(mu);
    auto it = devices.find(device_id);
    if (it != devices.end()) return *it->second;
    sbx_handle_t h = nullptr;
    const int rc = sbx_create(device_id, &h);
    if (rc != SBX_OK)
      throw utils::HIPDeviceException(std::string("cannot open HIP device ") + std::to_string(device_id) + ": " +
                                      sbx_status_string(rc) + " (the reorder/convert path has no CPU fallback)");
    auto *d = new Device(device_id, h);
    devices[device_id] = d;
    return *d;
  }
  sbx_handle_t handle() const { return h_; }
  int id() const { return id_; }
  void Check(int rc) const {
    if (rc != SBX_OK)
      throw utils::HIPDeviceException(std::string("sbx: ") + sbx_status_string(rc) + ": " + sbx_last_error(h_));
  }
  // Device blocks released by the host layer's formats and staging buffers are kept (up to pool_limit() bytes) and handed
  // out again to a request of the same size: a pipeline that produces a permuted HIPCSR per call (Reorder -> Permute2D,
  // experiment/experiment_helper.h:81-97) otherwise pays three hipMalloc / hipFree pairs of hundreds of megabytes per
  // call.  SBX_HOST_POOL_MB sets the limit (default 4096; 0: every block goes straight back to the driver).
  void *Malloc(size_t bytes) const {
    if (bytes == 0) bytes = 1;
    {
      std::lock_guard<std::mutex> lock(pool_mu_);
      auto it = pool_free_.find(bytes);
      if (it != pool_free_.end() && !it->second.empty()) {
        void *p = it->second.back();
        it->second.pop_back();
        pooled_bytes_ -= bytes;
        pool_live_[p] = bytes;
        return p;
      }
    }
    void *p = nullptr;
    int rc = sbx_malloc(h_, bytes, &p);
    if (rc != SBX_OK && TrimPool(0)) rc = sbx_malloc(h_, bytes, &p);  // (the pool may hold what the request needs)
    Check(rc);
    std::lock_guard<std::mutex> lock(pool_mu_);
    pool_live_[p] = bytes;
    return p;
  }
  void Free(void *p) const {
    if (!p) return;
    {
      std::lock_guard<std::mutex> lock(pool_mu_);
      auto it = pool_live_.find(p);
      if (it != pool_live_.end()) {
        const size_t bytes = it->second;
        pool_live_.erase(it);
        if (bytes <= pool_limit() && pooled_bytes_ + bytes <= pool_limit()) {
          pool_free_[bytes].push_back(p);
          pooled_bytes_ += bytes;
          return;
        }
      }
    }
    sbx_free(h_, p);
  }
  // a block leaves the host layer's ownership (release_*() of a device format): whoever frees it does so directly
  void Forget(void *p) const {
    std::lock_guard<std::mutex> lock(pool_mu_);
    pool_live_.erase(p);
  }
  // returns every pooled block beyond `keep_bytes` to the driver; true if anything was released
  bool TrimPool(size_t keep_bytes) const {
    std::vector<void *> drop;
    {
      std::lock_guard<std::mutex> lock(pool_mu_);
      for (auto it = pool_free_.begin(); it != pool_free_.end() && pooled_bytes_ > keep_bytes;) {
        while (!it->second.empty() && pooled_bytes_ > keep_bytes) {
          drop.push_back(it->second.back());
          it->second.pop_back();
          pooled_bytes_ -= it->first;
        }
        it = it->second.empty() ? pool_free_.erase(it) : std::next(it);
      }
    }
    for (void *p : drop) sbx_free(h_, p);
    return !drop.empty();
  }
  static size_t pool_limit() {
    static const size_t lim = [] {
      const char *e = std::getenv("SBX_HOST_POOL_MB");
      return (size_t)(e ? std::atoll(e) : 4096) << 20;
    }();
    return lim;
  }
  // Page-locked host blocks for downloads the host layer consumes itself (GrayReorder's degrees and keys), pooled like
  // the device blocks (same limit variable, counted apart): a pageable target is pinned and unpinned by the runtime
  // around every copy, and freeing it afterwards stalled the process's next GPU submission by 14 - 22 ms
  // (tools/gray_kt2.sh).  SBX_HOST_PINNED_STAGING=0: pageable arrays as before (diagnostic).
  static bool pinned_staging() {
    static const bool on = [] {
      const char *e = std::getenv("SBX_HOST_PINNED_STAGING");
      return !(e && e[0] == '0');
    }();
    return on;
  }
  void *HostMalloc(size_t bytes) const {
    if (bytes == 0) bytes = 1;
    {
      std::lock_guard<std::mutex> lock(pool_mu_);
      auto it = host_free_.find(bytes);
      if (it != host_free_.end() && !it->second.empty()) {
        void *p = it->second.back();
        it->second.pop_back();
        host_pooled_bytes_ -= bytes;
        host_live_[p] = bytes;
        return p;
      }
    }
    void *p = nullptr;
    Check(sbx_host_alloc(h_, bytes, &p));
    std::lock_guard<std::mutex> lock(pool_mu_);
    host_live_[p] = bytes;
    return p;
  }
  void HostFree(void *p) const {
    if (!p) return;
    {
      std::lock_guard<std::mutex> lock(pool_mu_);
      auto it = host_live_.find(p);
      if (it != host_live_.end()) {
        const size_t bytes = it->second;
        host_live_.erase(it);
        if (host_pooled_bytes_ + bytes <= pool_limit() / 4) {  // (at most a quarter of the device pool's limit: 1 GB)
          host_free_[bytes].push_back(p);
          host_pooled_bytes_ += bytes;
          return;
        }
      }
    }
    sbx_host_free(h_, p);
  }
  void ToDevice(void *dst, const void *src, size_t bytes) const { Check(sbx_memcpy_h2d(h_, dst, src, bytes)); }
  void ToHost(void *dst, const void *src, size_t bytes) const { Check(sbx_memcpy_d2h(h_, dst, src, bytes)); }
  void Copy(void *dst, const void *src, size_t bytes) const { Check(sbx_memcpy_d2d(h_, dst, src, bytes)); }
  void Sync() const { Check(sbx_sync(h_)); }

  template <typename T>
  T *Upload(const T *host, size_t count) const {
    if (host == nullptr) return nullptr;
    T *d = static_cast<T *>(Malloc(count * sizeof(T)));
    ToDevice(d, host, count * sizeof(T));
    return d;
  }
  template <typename T>
  T *Download(const T *dev, size_t count) const {  // returns new T[count]; caller owns
    if (dev == nullptr) return nullptr;
    T *hptr = new T[count];
    ToHost(hptr, dev, count * sizeof(T));
    return hptr;
  }

 private:
  Device(int id, sbx_handle_t h) : id_(id), h_(h) {
    // the library's own allocations (scratch arena growth, radix slots) must not fail while this pool sits on idle blocks
    sbx_set_oom_hook(h_, [](void *self, size_t) -> int { return static_cast<Device *>(self)->TrimPool(0) ? 1 : 0; }, this);
  }
  int id_;
  sbx_handle_t h_;
  mutable std::mutex pool_mu_;
  mutable std::unordered_map<void *, size_t> pool_live_;       // blocks handed out by Malloc: their sizes
  mutable std::map<size_t, std::vector<void *>> pool_free_;    // released blocks by size
  mutable size_t pooled_bytes_ = 0;
  mutable std::unordered_map<void *, size_t> host_live_;       // page-locked host blocks, the same way
  mutable std::map<size_t, std::vector<void *>> host_free_;
  mutable size_t host_pooled_bytes_ = 0;
};

// n elements of page-locked (or, with SBX_HOST_PINNED_STAGING=0, plain) host memory, left uninitialised: the target of a
// download whose contents the host layer reads itself
template <typename T>
class HostStaging {
 public:
  HostStaging(const Device &d, size_t n) : d_(d), n_(n), pinned_(Device::pinned_staging()) {
    static_assert(std::is_trivially_default_constructible<T>::value && std::is_trivially_destructible<T>::value, "raw storage");
    p_ = n ? (pinned_ ? static_cast<T *>(d.HostMalloc(n * sizeof(T))) : new T[n]) : nullptr;
  }
  ~HostStaging() {
    if (pinned_) d_.HostFree(p_);
    else delete[] p_;
  }
  HostStaging(const HostStaging &) = delete;
  HostStaging &operator=(const HostStaging &) = delete;
  size_t size() const { return n_; }
  T *data() { return p_; }
  T &operator[](size_t i) { return p_[i]; }
  const T &operator[](size_t i) const { return p_[i]; }

 private:
  const Device &d_;
  size_t n_;
  bool pinned_;
  T *p_;
};

// RAII device allocation used for staging host-resident formats through the GPU
template <typename T>
class Staged {
 public:
  Staged(const Device &d, size_t count) : d_(d), count_(count), p_(static_cast<T *>(d.Malloc(count * sizeof(T)))) {}
  Staged(const Device &d, const T *host, size_t count) : Staged(d, count) {
    if (host) d_.ToDevice(p_, host, count * sizeof(T));
  }
  ~Staged() { d_.Free(p_); }
  Staged(const Staged &) = delete;
  Staged &operator=(const Staged &) = delete;
  T *get() const { return p_; }
  void ToHost(T *host) const { d_.ToHost(host, p_, count_ * sizeof(T)); }

 private:
  const Device &d_;
  size_t count_;
  T *p_;
};

// C ABI type tags for a C++ type tuple
template <typename IDType>
constexpr sbx_index_type IndexTag() {
  static_assert(sizeof(IDType) == 4 || sizeof(IDType) == 8, "index type must be 32 or 64 bit");
  return sizeof(IDType) == 4 ? SBX_I32 : SBX_I64;
}
// ... for a format's (IDType, NNZType) pair: equal widths, or 32-bit ids with 64-bit offsets (SBX_I32_N64: the id arrays
// — row, col, orders — hold 32-bit words, the offset arrays — row_ptr, col_ptr — 64-bit ones)
template <typename IDType, typename NNZType>
constexpr sbx_index_type IndexTag() {
  static_assert(sizeof(IDType) == sizeof(NNZType) || (sizeof(IDType) == 4 && sizeof(NNZType) == 8),
                "index tuples of the device path: IDType and NNZType of one width, or 32-bit ids with 64-bit offsets");
  return sizeof(IDType) == sizeof(NNZType) ? IndexTag<IDType>() : SBX_I32_N64;
}
template <typename V>
constexpr sbx_value_type ValueTag() {
  if constexpr (std::is_same_v<V, void>) return SBX_V_NONE;
  else if constexpr (std::is_floating_point_v<V>) return sizeof(V) == 4 ? SBX_V_F32 : SBX_V_F64;
  else if constexpr (std::is_signed_v<V>) return sizeof(V) == 4 ? SBX_V_I32 : SBX_V_I64;
  else return sizeof(V) == 4 ? SBX_V_U32 : SBX_V_U64;
}
template <typename V>
constexpr size_t ValueBytes() {
  if constexpr (std::is_same_v<V, void>) return 0;
  else return sizeof(V);
}

}  // namespace sparsebase::hip
#endif
