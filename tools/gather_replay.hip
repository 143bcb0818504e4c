// Relabel ceiling of Permute2D (diagnostic, not part of the library): out[j] = table[col[j]] over the bench matrix's real
// column stream, with U independent gathers in flight per lane, streaming (nt) index loads / result stores, and a
// grid of `waves_per_cu` resident waves per CU.  Built by tools/gather_ceiling2.py into tools/libgather_replay.so.
#include <hip/hip_runtime.h>
#include <stdint.h>

// the whole memory side of Permute2D and nothing else: col in, val in, one table gather, col out, val out (16 B + a gather
// per entry), everything streamed nt, U entries per lane in flight
template <int U>
__global__ __launch_bounds__(256) void k_replay_full(const int *__restrict__ idx, const int *__restrict__ table,
                                                     int *__restrict__ out, const int *__restrict__ val,
                                                     int *__restrict__ val_out, int64_t n) {
  int64_t base = (int64_t)blockIdx.x * 256 * U + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256 * U;
  for (; base < n; base += stride) {
    int ix[U], v[U], w[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t j = base + u * 256 < n ? base + u * 256 : n - 1;
      ix[u] = __builtin_nontemporal_load(idx + j);
      w[u] = __builtin_nontemporal_load(val + j);
    }
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = table[ix[u]];
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (base + u * 256 < n) {
        __builtin_nontemporal_store(v[u], out + base + u * 256);
        __builtin_nontemporal_store(w[u], val_out + base + u * 256);
      }
    }
  }
}

template <int U, bool NT, bool STORE>
__global__ __launch_bounds__(256) void k_replay(const int *__restrict__ idx, const int *__restrict__ table,
                                                int *__restrict__ out, int64_t n) {
  int64_t base = (int64_t)blockIdx.x * 256 * U + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256 * U;
  int acc = 0;
  for (; base < n; base += stride) {
    int ix[U], v[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t j = base + u * 256 < n ? base + u * 256 : n - 1;
      ix[u] = NT ? __builtin_nontemporal_load(idx + j) : idx[j];
    }
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = table[ix[u]];
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (STORE) {
        if (base + u * 256 < n) {
          if (NT) __builtin_nontemporal_store(v[u], out + base + u * 256);
          else out[base + u * 256] = v[u];
        }
      } else {
        acc ^= v[u];
      }
    }
  }
  if (!STORE && acc == 0x7FEDCBA9) out[0] = acc;
}

template <int U, bool NT, bool STORE>
static float run1(const int *idx, const int *table, int *out, int64_t n, int grid, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a), hipEventCreate(&b);
  hipLaunchKernelGGL((k_replay<U, NT, STORE>), dim3(grid), dim3(256), 0, 0, idx, table, out, n);
  hipEventRecord(a, 0);
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL((k_replay<U, NT, STORE>), dim3(grid), dim3(256), 0, 0, idx, table, out, n);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  hipEventDestroy(a), hipEventDestroy(b);
  return ms / reps;
}

extern "C" float gather_replay_full(const int *idx, const int *table, int *out, const int *val, int *val_out, int64_t n,
                                    int waves_per_cu, int reps) {
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int grid = cus * waves_per_cu / 4;
  hipEvent_t a, b;
  (void)hipEventCreate(&a), (void)hipEventCreate(&b);
  hipLaunchKernelGGL((k_replay_full<8>), dim3(grid), dim3(256), 0, 0, idx, table, out, val, val_out, n);
  (void)hipEventRecord(a, 0);
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL((k_replay_full<8>), dim3(grid), dim3(256), 0, 0, idx, table, out, val, val_out, n);
  (void)hipEventRecord(b, 0);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

extern "C" float gather_replay(const int *idx, const int *table, int *out, int64_t n, int U, int nt, int store,
                               int waves_per_cu, int reps) {
  int dev = 0, cus = 256;
  hipGetDevice(&dev);
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int grid = cus * waves_per_cu / 4;
#define GO(UU)                                                                                          \
  if (U == UU) {                                                                                        \
    if (nt && store) return run1<UU, true, true>(idx, table, out, n, grid, reps);                       \
    if (nt && !store) return run1<UU, true, false>(idx, table, out, n, grid, reps);                     \
    if (!nt && store) return run1<UU, false, true>(idx, table, out, n, grid, reps);                     \
    return run1<UU, false, false>(idx, table, out, n, grid, reps);                                      \
  }
  GO(1) GO(2) GO(4) GO(8) GO(16)
#undef GO
  return -1.f;
}
