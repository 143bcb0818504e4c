// sbx_handle.hip — handle life-cycle, scratch arena, memory/device utilities.
// Replaces the raw CUDA runtime calls of the reference's device formats
// (converter/converter_order_two_cuda.cu:11-105, context/cuda_context_cuda.cu:9-21,
// converter/converter_cuda.cu:12-21) behind the C ABI of include/sbx.h.
#include "sbx_internal.h"

#include <new>

static const size_t kArenaAlign = 256;
static const size_t kPinnedBytes = 1 << 16;

extern "C" int sbx_version(void) { return SBX_VERSION; }

extern "C" const char *sbx_status_string(int status) {
  switch (status) {
    case SBX_OK: return "ok";
    case SBX_ERR_BAD_ARG: return "bad argument";
    case SBX_ERR_NO_DEVICE: return "no usable HIP device";
    case SBX_ERR_HIP: return "HIP runtime error";
    case SBX_ERR_OOM: return "out of device memory";
    case SBX_ERR_UNSUPPORTED: return "unsupported type tuple or shape";
    case SBX_ERR_INTERNAL: return "internal error";
  }
  return "unknown status";
}

extern "C" int sbx_device_count(int *count_host) {
  if (!count_host) return SBX_ERR_BAD_ARG;
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) {
    *count_host = 0;
    (void)hipGetLastError();
    return SBX_ERR_NO_DEVICE;
  }
  *count_host = c;
  return SBX_OK;
}

extern "C" int sbx_can_access_peer(int device, int peer_device, int *can_host) {
  if (!can_host) return SBX_ERR_BAD_ARG;
  if (device == peer_device) {
    *can_host = 1;
    return SBX_OK;
  }
  int can = 0;
  hipError_t e = hipDeviceCanAccessPeer(&can, device, peer_device);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    *can_host = 0;
    return SBX_ERR_HIP;
  }
  *can_host = can;
  return SBX_OK;
}

extern "C" int sbx_create(int device, sbx_handle_t *out) {
  if (!out) return SBX_ERR_BAD_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    (void)hipGetLastError();
    return SBX_ERR_NO_DEVICE;  // loud: there is no CPU fallback behind this ABI
  }
  if (device < 0 || device >= count) return SBX_ERR_BAD_ARG;
  sbx_handle_s *h = new (std::nothrow) sbx_handle_s();
  if (!h) return SBX_ERR_INTERNAL;
  h->device = device;
  h->stream = nullptr;
  h->cur_block = 0;
  h->cur_off = 0;
  h->call_bytes = 0;
  h->high_water = 0;
  h->nest = 0;
  h->pinned = nullptr;
  h->pinned_bytes = 0;
  h->rcm_gb_backoff = 0;
  h->err[0] = 0;
  h->prof_on = false;
  for (int i = 0; i < SBX_K_COUNT; i++) {
    h->prof_ms[i] = 0.0;
    h->prof_launches[i] = 0;
    h->prof_bytes[i] = 0;
  }
  if (hipSetDevice(device) != hipSuccess) {
    delete h;
    return SBX_ERR_HIP;
  }
  hipDeviceProp_t prop;
  h->num_cus = 256;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) h->num_cus = prop.multiProcessorCount;
  // coherent (fine-grained) so that a kernel's system-scope stores are visible to a polling host
  if (hipHostMalloc(&h->pinned, kPinnedBytes, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) {
    delete h;
    return SBX_ERR_OOM;
  }
  h->pinned_bytes = kPinnedBytes;
  h->rb_seq = 0;
  h->pow5 = nullptr;
  h->aux_ready = false;
  h->aux_dirty = false;
  h->rs_tied_hint = false;
  h->rs_override = nullptr;
  h->rs_pool = nullptr;
  h->rs_next = 0;
  h->oom_hook = nullptr;
  h->oom_user = nullptr;
  {
    const char *e = sbx_env_tuning("SBX_READBACK_POLL");
    h->rb_poll = !(e && e[0] == '0');
  }
  memset(h->pinned, 0, kPinnedBytes);
  *out = h;
  return SBX_OK;
}

extern "C" int sbx_destroy(sbx_handle_t h) {
  if (!h) return SBX_OK;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);  // may be a caller-owned stream that is already gone: the error is ignored
  if (h->aux_ready)  // side-stream work of a call that failed between fork and join may still use the scratch
    for (int i = 0; i < SBX_AUX_STREAMS; i++) (void)hipStreamSynchronize(h->aux_stream[i]);
  for (auto &b : h->blocks) (void)hipFree(b.ptr);
  if (h->pinned) (void)hipHostFree(h->pinned);
  if (h->rs_pool) (void)hipFree(h->rs_pool);
  if (h->pow5) (void)hipFree(h->pow5);
  if (h->aux_ready) {
    for (int i = 0; i < SBX_AUX_STREAMS; i++) (void)hipStreamDestroy(h->aux_stream[i]);
    for (int i = 0; i < SBX_AUX_STREAMS + 1; i++) (void)hipEventDestroy(h->aux_event[i]);
  }
  for (auto &r : h->prof_pending) {
    (void)hipEventDestroy(r.start);
    (void)hipEventDestroy(r.stop);
  }
  for (auto &e : h->prof_pool) (void)hipEventDestroy(e);
  delete h;
  return SBX_OK;
}

int sbx_aux_streams(sbx_handle_t h) {
  if (h->aux_ready) return SBX_OK;
  for (int i = 0; i < SBX_AUX_STREAMS; i++) SBX_HIP(h, hipStreamCreateWithFlags(&h->aux_stream[i], hipStreamNonBlocking));
  for (int i = 0; i < SBX_AUX_STREAMS + 1; i++) SBX_HIP(h, hipEventCreateWithFlags(&h->aux_event[i], hipEventDisableTiming));
  h->aux_ready = true;
  return SBX_OK;
}

extern "C" int sbx_set_stream(sbx_handle_t h, void *hip_stream) {
  if (!h) return SBX_ERR_BAD_ARG;
  if (h->stream != (hipStream_t)hip_stream) {
    // scratch (arena, radix slot pool, pinned read-back buffer) is reused in stream order: drain the old stream
    SBX_HIP(h, hipSetDevice(h->device));
    SBX_HIP(h, hipStreamSynchronize(h->stream));
  }
  h->stream = (hipStream_t)hip_stream;
  return SBX_OK;
}

extern "C" int sbx_get_device(sbx_handle_t h, int *device_host) {
  if (!h || !device_host) return SBX_ERR_BAD_ARG;
  *device_host = h->device;
  return SBX_OK;
}

extern "C" const char *sbx_last_error(sbx_handle_t h) { return h ? h->err : "null handle"; }

extern "C" int sbx_sync(sbx_handle_t h) {
  if (!h) return SBX_ERR_BAD_ARG;
  SBX_HIP(h, hipSetDevice(h->device));
  SBX_HIP(h, hipStreamSynchronize(h->stream));
  return SBX_OK;
}

// hipMalloc for the library's own blocks (arena, radix slot pool): when the driver is out of memory the caller's hook
// (sbx_set_oom_hook: e.g. the host layer's pool of idle device blocks) is asked once to give some back
static hipError_t internal_malloc(sbx_handle_t h, void **p, size_t bytes) {
  hipError_t e = hipMalloc(p, bytes);
  if (e == hipErrorOutOfMemory && h->oom_hook) {
    (void)hipGetLastError();
    if (h->oom_hook(h->oom_user, bytes)) e = hipMalloc(p, bytes);
  }
  return e;
}

extern "C" int sbx_set_oom_hook(sbx_handle_t h, sbx_oom_hook hook, void *user) {
  if (!h) return SBX_ERR_BAD_ARG;
  h->oom_hook = hook;
  h->oom_user = user;
  return SBX_OK;
}

static int arena_add_block(sbx_handle_t h, size_t bytes) {
  sbx_block b;
  b.cap = (bytes + kArenaAlign - 1) / kArenaAlign * kArenaAlign;
  b.ptr = nullptr;
  hipError_t e = internal_malloc(h, (void **)&b.ptr, b.cap);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    SBX_FAIL(h, SBX_ERR_OOM, "scratch arena: hipMalloc(%zu) failed: %s", b.cap, hipGetErrorString(e));
  }
  h->blocks.push_back(b);
  return SBX_OK;
}

extern "C" int sbx_reserve(sbx_handle_t h, size_t scratch_bytes) {
  if (!h) return SBX_ERR_BAD_ARG;
  SBX_HIP(h, hipSetDevice(h->device));
  size_t have = 0;
  if (h->blocks.size() == 1) have = h->blocks[0].cap;
  if (have >= scratch_bytes) return SBX_OK;
  SBX_HIP(h, hipStreamSynchronize(h->stream));
  for (auto &b : h->blocks) (void)hipFree(b.ptr);
  h->blocks.clear();
  h->cur_block = 0;
  h->cur_off = 0;
  return arena_add_block(h, scratch_bytes);
}

int sbx_arena_begin(sbx_handle_t h) {
  SBX_HIP(h, hipSetDevice(h->device));
  if (h->nest > 0) return SBX_OK;  // nested entry point: keep the caller's scratch alive
  if (h->aux_dirty) {  // a call gave up between fork and join: its side work still owns scratch
    for (int i = 0; i < SBX_AUX_STREAMS; i++) SBX_HIP(h, hipStreamSynchronize(h->aux_stream[i]));
    h->aux_dirty = false;
    h->rs_override = nullptr;
  }
  if (h->blocks.size() > 1) {
    // the previous call overflowed the first block: consolidate to one block
    size_t total = 0;
    for (auto &b : h->blocks) total += b.cap;
    SBX_HIP(h, hipStreamSynchronize(h->stream));
    for (auto &b : h->blocks) (void)hipFree(b.ptr);
    h->blocks.clear();
    SBX_TRY(arena_add_block(h, total + total / 8));
  }
  h->cur_block = 0;
  h->cur_off = 0;
  h->call_bytes = 0;
  return SBX_OK;
}

int sbx_arena_alloc(sbx_handle_t h, size_t bytes, void **out) {
  bytes = (bytes + kArenaAlign - 1) / kArenaAlign * kArenaAlign;
  if (bytes == 0) bytes = kArenaAlign;
  while (true) {
    if (h->cur_block < h->blocks.size()) {
      sbx_block &b = h->blocks[h->cur_block];
      if (h->cur_off + bytes <= b.cap) {
        *out = b.ptr + h->cur_off;
        h->cur_off += bytes;
        h->call_bytes += bytes;
        if (h->call_bytes > h->high_water) h->high_water = h->call_bytes;
        return SBX_OK;
      }
      h->cur_block++;
      h->cur_off = 0;
      continue;
    }
    size_t want = bytes;
    if (!h->blocks.empty() && h->blocks.back().cap > want) want = h->blocks.back().cap;
    if (want < ((size_t)1 << 20)) want = (size_t)1 << 20;
    SBX_TRY(arena_add_block(h, want));
  }
}

int sbx_radix_slot(sbx_handle_t h, void **slot) {
  if (h->rs_override) {
    *slot = h->rs_override;
    h->rs_override = nullptr;
    return SBX_OK;
  }
  if (!h->rs_pool) {
    SBX_HIP(h, internal_malloc(h, &h->rs_pool, (size_t)SBX_RS_SLOTS * SBX_RS_SLOT_BYTES));
    h->rs_next = SBX_RS_SLOTS;  // forces the first memset
  }
  if (h->rs_next >= SBX_RS_SLOTS) {
    // every earlier user of the pool was enqueued on this stream, so the memset is ordered after them
    SBX_HIP(h, hipMemsetAsync(h->rs_pool, 0, (size_t)SBX_RS_SLOTS * SBX_RS_SLOT_BYTES, h->stream));
    h->rs_next = 0;
  }
  *slot = (char *)h->rs_pool + (size_t)h->rs_next * SBX_RS_SLOT_BYTES;
  h->rs_next++;
  return SBX_OK;
}

// Small device->host read-backs sit on the critical path of every BFS level.  A copy engine
// transfer + hipStreamSynchronize costs ~20 us; instead one wave copies the words into the
// coherent pinned buffer, fences at system scope and stores a sequence number the host
// polls (word 0 of the buffer).  The poll is bounded: after ~1 ms without the number the host
// falls back to hipStreamSynchronize, which also surfaces a failed kernel.
static const size_t kReadbackHeader = 64;  // sequence word + padding to keep the payload 64-byte aligned

__global__ void k_readback(unsigned *__restrict__ dst, const unsigned *__restrict__ src, unsigned words,
                           unsigned *seq_word, unsigned seq) {
  if (((words & 3u) | ((uintptr_t)dst & 15) | ((uintptr_t)src & 15)) == 0) {  // 16-byte stores: a third of the bus transactions
    for (unsigned i = threadIdx.x; i < words / 4; i += blockDim.x) ((uint4 *)dst)[i] = ((const uint4 *)src)[i];
  } else {
    for (unsigned i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(seq_word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int sbx_readback(sbx_handle_t h, void *dst_host, const void *src_dev, size_t bytes) {
  if (bytes + kReadbackHeader > h->pinned_bytes) SBX_FAIL(h, SBX_ERR_INTERNAL, "readback of %zu bytes too large", bytes);
  char *payload = (char *)h->pinned + kReadbackHeader;
  if (!h->rb_poll || (bytes & 3) || ((uintptr_t)src_dev & 3)) {
    SBX_HIP(h, hipMemcpyAsync(payload, src_dev, bytes, hipMemcpyDeviceToHost, h->stream));
    SBX_HIP(h, hipStreamSynchronize(h->stream));
    memcpy(dst_host, payload, bytes);
    return SBX_OK;
  }
  volatile unsigned *seq_word = (volatile unsigned *)h->pinned;
  const unsigned seq = ++h->rb_seq;
  hipLaunchKernelGGL(k_readback, dim3(1), dim3(64), 0, h->stream, (unsigned *)payload, (const unsigned *)src_dev,
                     (unsigned)(bytes >> 2), (unsigned *)h->pinned, seq);
  SBX_HIP(h, hipGetLastError());
  bool seen = false;
  for (int spin = 0; spin < 200000; spin++) {
    if (*seq_word == seq) {
      seen = true;
      break;
    }
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
  if (!seen) {
    SBX_HIP(h, hipStreamSynchronize(h->stream));
    if (*seq_word != seq) SBX_FAIL(h, SBX_ERR_INTERNAL, "read-back kernel did not publish its sequence number");
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  memcpy(dst_host, payload, bytes);
  return SBX_OK;
}

extern "C" int sbx_malloc(sbx_handle_t h, size_t bytes, void **dev_ptr_host) {
  if (!h || !dev_ptr_host) return SBX_ERR_BAD_ARG;
  SBX_HIP(h, hipSetDevice(h->device));
  *dev_ptr_host = nullptr;
  if (bytes == 0) bytes = 1;
  SBX_HIP(h, hipMalloc(dev_ptr_host, bytes));
  return SBX_OK;
}

extern "C" int sbx_free(sbx_handle_t h, void *dev_ptr) {
  if (!h) return SBX_ERR_BAD_ARG;
  if (!dev_ptr) return SBX_OK;
  SBX_HIP(h, hipSetDevice(h->device));
  SBX_HIP(h, hipFree(dev_ptr));
  return SBX_OK;
}

extern "C" int sbx_host_alloc(sbx_handle_t h, size_t bytes, void **host_ptr_host) {
  if (!h || !host_ptr_host) return SBX_ERR_BAD_ARG;
  SBX_HIP(h, hipSetDevice(h->device));
  *host_ptr_host = nullptr;
  if (bytes == 0) bytes = 1;
  SBX_HIP(h, hipHostMalloc(host_ptr_host, bytes, hipHostMallocDefault));
  return SBX_OK;
}

extern "C" int sbx_host_free(sbx_handle_t h, void *host_ptr) {
  if (!h) return SBX_ERR_BAD_ARG;
  if (!host_ptr) return SBX_OK;
  SBX_HIP(h, hipSetDevice(h->device));
  SBX_HIP(h, hipHostFree(host_ptr));
  return SBX_OK;
}

static int copy_blocking(sbx_handle_t h, void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
  if (!h || (bytes && (!dst || !src))) return SBX_ERR_BAD_ARG;
  if (bytes == 0) return SBX_OK;
  SBX_HIP(h, hipSetDevice(h->device));
  SBX_HIP(h, hipMemcpyAsync(dst, src, bytes, kind, h->stream));
  SBX_HIP(h, hipStreamSynchronize(h->stream));
  return SBX_OK;
}

extern "C" int sbx_memcpy_h2d(sbx_handle_t h, void *dst_dev, const void *src_host, size_t bytes) {
  return copy_blocking(h, dst_dev, src_host, bytes, hipMemcpyHostToDevice);
}
extern "C" int sbx_memcpy_d2h(sbx_handle_t h, void *dst_host, const void *src_dev, size_t bytes) {
  return copy_blocking(h, dst_host, src_dev, bytes, hipMemcpyDeviceToHost);
}
extern "C" int sbx_memcpy_d2d(sbx_handle_t h, void *dst_dev, const void *src_dev, size_t bytes) {
  return copy_blocking(h, dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice);
}
extern "C" int sbx_memcpy_peer(sbx_handle_t h, void *dst_dev, int dst_device, const void *src_dev,
                               int src_device, size_t bytes) {
  if (!h || (bytes && (!dst_dev || !src_dev))) return SBX_ERR_BAD_ARG;
  if (bytes == 0) return SBX_OK;
  SBX_HIP(h, hipSetDevice(h->device));
  SBX_HIP(h, hipMemcpyPeerAsync(dst_dev, dst_device, src_dev, src_device, bytes, h->stream));
  SBX_HIP(h, hipStreamSynchronize(h->stream));
  return SBX_OK;
}

// ---------------------------------------------------------------------------
// HIP-event profiler (used by bench.py for the live per-kernel roofline figures)
// ---------------------------------------------------------------------------
const char *const sbx_kernel_names[SBX_K_COUNT] = {
    "scan",          "radix_hist",   "radix_scatter", "coo_to_csr", "csr_to_coo", "permute_tile",
    "permute_long",  "permute_block", "permute_prep", "bfs_expand",    "bfs_heavy",  "bfs_bottom_up", "bfs_small_levels",  "level_order", "cc",
    "rcm_small",     "rcm_misc",     "gray",          "degree",     "check",       "csc",  "feature", "mtx", "misc"};

static hipEvent_t prof_event(sbx_handle_t h) {
  if (!h->prof_pool.empty()) {
    hipEvent_t e = h->prof_pool.back();
    h->prof_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

void sbx_prof_begin(sbx_handle_t h, int kid) {
  sbx_prof_rec r;
  r.kid = kid;
  r.start = prof_event(h);
  r.stop = prof_event(h);
  (void)hipEventRecord(r.start, h->stream);
  h->prof_pending.push_back(r);
}

void sbx_prof_end(sbx_handle_t h) { (void)hipEventRecord(h->prof_pending.back().stop, h->stream); }

static void prof_drain(sbx_handle_t h) {
  for (auto &r : h->prof_pending) {
    float ms = 0.f;
    if (hipEventSynchronize(r.stop) == hipSuccess && hipEventElapsedTime(&ms, r.start, r.stop) == hipSuccess) {
      h->prof_ms[r.kid] += ms;
      h->prof_launches[r.kid] += 1;
    }
    h->prof_pool.push_back(r.start);
    h->prof_pool.push_back(r.stop);
  }
  h->prof_pending.clear();
}

extern "C" int sbx_profile_enable(sbx_handle_t h, int on) {
  if (!h) return SBX_ERR_BAD_ARG;
  SBX_HIP(h, hipSetDevice(h->device));
  prof_drain(h);
  if (on) {
    for (int i = 0; i < SBX_K_COUNT; i++) {
      h->prof_ms[i] = 0.0;
      h->prof_launches[i] = 0;
      h->prof_bytes[i] = 0;
    }
  }
  h->prof_on = on != 0;
  return SBX_OK;
}

extern "C" int sbx_profile_kernel_count(void) { return SBX_K_COUNT; }

extern "C" const char *sbx_profile_kernel_name(int index) {
  return (index >= 0 && index < SBX_K_COUNT) ? sbx_kernel_names[index] : nullptr;
}

extern "C" int sbx_profile_query(sbx_handle_t h, int index, double *total_ms_host, int64_t *launches_host) {
  if (!h || index < 0 || index >= SBX_K_COUNT || !total_ms_host || !launches_host) return SBX_ERR_BAD_ARG;
  SBX_HIP(h, hipSetDevice(h->device));
  prof_drain(h);
  *total_ms_host = h->prof_ms[index];
  *launches_host = h->prof_launches[index];
  return SBX_OK;
}

extern "C" int sbx_profile_query_bytes(sbx_handle_t h, int index, int64_t *alg_bytes_host) {
  if (!h || index < 0 || index >= SBX_K_COUNT || !alg_bytes_host) return SBX_ERR_BAD_ARG;
  *alg_bytes_host = (int64_t)h->prof_bytes[index];
  return SBX_OK;
}
