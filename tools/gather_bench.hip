// Scattered-gather / scattered-atomic ceilings on MI355X (diagnostic, not part of the library).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

template <int U, typename T>
__global__ __launch_bounds__(256) void k_gather(const uint32_t* __restrict__ idx, const T* __restrict__ table, T* __restrict__ out, int64_t n) {
  int64_t base = ((int64_t)blockIdx.x * 256 * U) + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256 * U;
  for (; base < n; base += stride) {
    uint32_t ix[U]; T v[U];
#pragma unroll
    for (int u = 0; u < U; u++) ix[u] = (base + u * 256 < n) ? idx[base + u * 256] : 0;
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = table[ix[u]];
#pragma unroll
    for (int u = 0; u < U; u++) if (base + u * 256 < n) out[base + u * 256] = v[u];
  }
}
template <int U>
__global__ __launch_bounds__(256) void k_gather_sc1(const uint32_t* __restrict__ idx, const unsigned long long* table, unsigned long long* __restrict__ out, int64_t n) {
  int64_t base = ((int64_t)blockIdx.x * 256 * U) + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256 * U;
  for (; base < n; base += stride) {
    uint32_t ix[U]; unsigned long long v[U];
#pragma unroll
    for (int u = 0; u < U; u++) ix[u] = (base + u * 256 < n) ? idx[base + u * 256] : 0;
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = __hip_atomic_load(&table[ix[u]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int u = 0; u < U; u++) if (base + u * 256 < n) out[base + u * 256] = v[u];
  }
}
__global__ __launch_bounds__(256) void k_atomic_min(const uint32_t* __restrict__ idx, unsigned long long* table, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (; i < n; i += stride) atomicMin(&table[idx[i]], (unsigned long long)i);
}
__global__ __launch_bounds__(256) void k_atomic_min32(const uint32_t* __restrict__ idx, unsigned* table, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (; i < n; i += stride) atomicMin(&table[idx[i]], (unsigned)i);
}
__global__ __launch_bounds__(256) void k_scatter(const uint32_t* __restrict__ idx, uint32_t* table, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (; i < n; i += stride) table[idx[i]] = (uint32_t)i;
}

template <typename F> float timeit(F f, int reps = 5) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int r = 0; r < reps; r++) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}
int main() {
  const int64_t n = 100000000;
  uint32_t* idx; CK(hipMalloc(&idx, n * 4));
  void* out; CK(hipMalloc(&out, n * 8));
  void* table; CK(hipMalloc(&table, (size_t)512 << 20));
  CK(hipMemset(table, 0xFF, (size_t)512 << 20));
  std::vector<uint32_t> h(n);
  for (int lg = 16; lg <= 26; lg += 2) {
    const uint32_t entries = 1u << lg;
    uint64_t s = 88172645463325252ull;
    for (int64_t i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint32_t)(s >> 20) & (entries - 1); }
    CK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice));
    const int grid = 256 * 8;
    float g1 = timeit([&] { hipLaunchKernelGGL((k_gather<1, uint32_t>), dim3(grid), dim3(256), 0, 0, idx, (const uint32_t*)table, (uint32_t*)out, n); });
    float g4 = timeit([&] { hipLaunchKernelGGL((k_gather<4, uint32_t>), dim3(grid), dim3(256), 0, 0, idx, (const uint32_t*)table, (uint32_t*)out, n); });
    float g8 = timeit([&] { hipLaunchKernelGGL((k_gather<8, uint32_t>), dim3(grid), dim3(256), 0, 0, idx, (const uint32_t*)table, (uint32_t*)out, n); });
    float g8q = timeit([&] { hipLaunchKernelGGL((k_gather<8, unsigned long long>), dim3(grid), dim3(256), 0, 0, idx, (const unsigned long long*)table, (unsigned long long*)out, n); });
    float s8 = timeit([&] { hipLaunchKernelGGL((k_gather_sc1<8>), dim3(grid), dim3(256), 0, 0, idx, (const unsigned long long*)table, (unsigned long long*)out, n); });
    float am = timeit([&] { hipLaunchKernelGGL(k_atomic_min, dim3(grid), dim3(256), 0, 0, idx, (unsigned long long*)table, n); });
    float am32 = timeit([&] { hipLaunchKernelGGL(k_atomic_min32, dim3(grid), dim3(256), 0, 0, idx, (unsigned*)table, n); });
    float sc = timeit([&] { hipLaunchKernelGGL(k_scatter, dim3(grid), dim3(256), 0, 0, idx, (uint32_t*)table, n); });
    printf("entries 2^%d (4B table %6.1f MB): gather4B U1 %.2f ms (%.0f G/s) U4 %.2f (%.0f) U8 %.2f (%.0f) | 8B U8 %.2f (%.0f) | sc1 8B U8 %.2f (%.0f) | atomicMin64 %.2f (%.0f) atomicMin32 %.2f (%.0f) | scatter4B %.2f (%.0f)\n",
           lg, entries * 4.0 / 1e6, g1, n / g1 / 1e6, g4, n / g4 / 1e6, g8, n / g8 / 1e6, g8q, n / g8q / 1e6, s8, n / s8 / 1e6, am, n / am / 1e6, am32, n / am32 / 1e6, sc, n / sc / 1e6);
    CK(hipMemset(table, 0xFF, (size_t)512 << 20));
  }
  return 0;
}
