// Cache-policy sweep of Permute2D's whole memory side (diagnostic, not part of the library): col in, val in, one table
// gather, col out, val out — the replay of tools/gather_replay.hip — with the streaming loads and the streaming stores
// issued under each cache policy (inline asm).  Question: do the 1.7 GB of streamed lines push the 16 MB relabel table
// out of the XCDs' L2s (every table miss is a 128-byte line over the fabric), and does a store that drops its line help?
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int P>
__device__ __forceinline__ int sld(const int *p) {  // streaming load
  int v;
  if (P == 0) asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  if (P == 1) asm volatile("global_load_dword %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
  if (P == 2) asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  if (P == 3) asm volatile("global_load_dword %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
  if (P == 4) asm volatile("global_load_dword %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int P>
__device__ __forceinline__ void sst(int *p, int v) {  // streaming store
  if (P == 0) asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
  if (P == 1) asm volatile("global_store_dword %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
  if (P == 2) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  if (P == 3) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  if (P == 4) asm volatile("global_store_dword %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
}

template <int LP, int SP, int U>
__global__ __launch_bounds__(256) void k_replay_policy(const int *__restrict__ idx, const int *table,
                                                       int *__restrict__ out, const int *__restrict__ val,
                                                       int *__restrict__ val_out, int64_t n) {
  int64_t base = (int64_t)blockIdx.x * 256 * U + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256 * U;
  for (; base < n; base += stride) {
    int ix[U], v[U], w[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t j = base + u * 256 < n ? base + u * 256 : n - 1;
      ix[u] = sld<LP>(idx + j);
      w[u] = sld<LP>(val + j);
    }
    // (the loads above are invisible to the compiler's own wait counting: everything that reads their results must
    // depend on this wait)
    static_assert(U == 8, "operand list");
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(ix[0]), "+v"(ix[1]), "+v"(ix[2]), "+v"(ix[3]), "+v"(ix[4]), "+v"(ix[5]), "+v"(ix[6]), "+v"(ix[7]),
                   "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7])
                 :
                 : "memory");
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = table[ix[u]];
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (base + u * 256 < n) {
        sst<SP>(out + base + u * 256, v[u]);
        sst<SP>(val_out + base + u * 256, w[u]);
      }
    }
  }
}

template <int LP, int SP>
static float run1(const int *idx, const int *table, int *out, const int *val, int *val_out, int64_t n, int grid, int reps) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a), (void)hipEventCreate(&b);
  hipLaunchKernelGGL((k_replay_policy<LP, SP, 8>), dim3(grid), dim3(256), 0, 0, idx, table, out, val, val_out, n);
  (void)hipEventRecord(a, 0);
  for (int r = 0; r < reps; r++)
    hipLaunchKernelGGL((k_replay_policy<LP, SP, 8>), dim3(grid), dim3(256), 0, 0, idx, table, out, val, val_out, n);
  (void)hipEventRecord(b, 0);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  (void)hipEventDestroy(a), (void)hipEventDestroy(b);
  return ms / reps;
}

extern "C" float replay_policy(const int *idx, const int *table, int *out, const int *val, int *val_out, int64_t n,
                               int lp, int sp, int waves_per_cu, int reps) {
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int grid = cus * waves_per_cu / 4;
#define ROW(L)                                                                        \
  if (lp == L) {                                                                      \
    if (sp == 0) return run1<L, 0>(idx, table, out, val, val_out, n, grid, reps);     \
    if (sp == 1) return run1<L, 1>(idx, table, out, val, val_out, n, grid, reps);     \
    if (sp == 2) return run1<L, 2>(idx, table, out, val, val_out, n, grid, reps);     \
    if (sp == 3) return run1<L, 3>(idx, table, out, val, val_out, n, grid, reps);     \
    if (sp == 4) return run1<L, 4>(idx, table, out, val, val_out, n, grid, reps);     \
  }
  ROW(0) ROW(1) ROW(2) ROW(3) ROW(4)
#undef ROW
  return -1.f;
}
