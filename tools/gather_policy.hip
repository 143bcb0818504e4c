// Cache-policy variants of the relabel gather (diagnostic): out[j] = table[col[j]] over a real column stream with the
// table load issued as plain / nt / sc1 / sc0 sc1 / sc0 sc1 nt (inline asm), 8 independent gathers per lane.
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int POLICY>
__device__ __forceinline__ int gld(const int *table, int i) {
  int v;
  const unsigned off = (unsigned)i * 4u;
  if (POLICY == 0) asm volatile("global_load_dword %0, %1, %2" : "=v"(v) : "v"(off), "s"(table) : "memory");
  if (POLICY == 1) asm volatile("global_load_dword %0, %1, %2 nt" : "=v"(v) : "v"(off), "s"(table) : "memory");
  if (POLICY == 2) asm volatile("global_load_dword %0, %1, %2 sc1" : "=v"(v) : "v"(off), "s"(table) : "memory");
  if (POLICY == 3) asm volatile("global_load_dword %0, %1, %2 sc0 sc1" : "=v"(v) : "v"(off), "s"(table) : "memory");
  if (POLICY == 4) asm volatile("global_load_dword %0, %1, %2 sc0 sc1 nt" : "=v"(v) : "v"(off), "s"(table) : "memory");
  if (POLICY == 5) asm volatile("global_load_dword %0, %1, %2 sc0" : "=v"(v) : "v"(off), "s"(table) : "memory");
  if (POLICY == 6) asm volatile("global_load_ushort %0, %1, %2" : "=v"(v) : "v"(off), "s"(table) : "memory");
  return v;
}

template <int POLICY, int U>
__global__ __launch_bounds__(256) void k_policy(const int *__restrict__ idx, const int *table, int *__restrict__ out,
                                                int64_t n) {
  int64_t base = (int64_t)blockIdx.x * 256 * U + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256 * U;
  for (; base < n; base += stride) {
    int ix[U], v[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t j = base + u * 256 < n ? base + u * 256 : n - 1;
      ix[u] = __builtin_nontemporal_load(idx + j);
    }
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = gld<POLICY>(table, ix[u]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < U; u++)
      if (base + u * 256 < n) __builtin_nontemporal_store(v[u], out + base + u * 256);
  }
}

template <int POLICY>
static float run1(const int *idx, const int *table, int *out, int64_t n, int grid, int reps) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a), (void)hipEventCreate(&b);
  hipLaunchKernelGGL((k_policy<POLICY, 8>), dim3(grid), dim3(256), 0, 0, idx, table, out, n);
  (void)hipEventRecord(a, 0);
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL((k_policy<POLICY, 8>), dim3(grid), dim3(256), 0, 0, idx, table, out, n);
  (void)hipEventRecord(b, 0);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  (void)hipEventDestroy(a), (void)hipEventDestroy(b);
  return ms / reps;
}

extern "C" float gather_policy(const int *idx, const int *table, int *out, int64_t n, int policy, int waves_per_cu, int reps) {
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int grid = cus * waves_per_cu / 4;
  switch (policy) {
    case 0: return run1<0>(idx, table, out, n, grid, reps);
    case 1: return run1<1>(idx, table, out, n, grid, reps);
    case 2: return run1<2>(idx, table, out, n, grid, reps);
    case 3: return run1<3>(idx, table, out, n, grid, reps);
    case 4: return run1<4>(idx, table, out, n, grid, reps);
    case 5: return run1<5>(idx, table, out, n, grid, reps);
    case 6: return run1<6>(idx, table, out, n, grid, reps);
  }
  return -1.f;
}
