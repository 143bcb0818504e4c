// Segmented row copy ceilings (diagnostic): nrows rows of L 4-byte entries from random source offsets to packed
// destinations; one wave per row-chunk of 256 entries.  mode 0: dword per lane; 1: 16 bytes per lane through
// registers; 2: 16 bytes per lane by LDS-DMA, then LDS -> registers -> 16-byte stores.  misalign: source / destination
// offsets are 16-byte aligned (0) or 4 bytes off (1).  Prints TB/s of read + written bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
struct __attribute__((packed, aligned(4))) i4u { int x, y, z, w; };

template <int MODE>
__global__ __launch_bounds__(64) void k_copy(const int *__restrict__ src, int *__restrict__ dst, const unsigned *__restrict__ soff,
                                             const unsigned *__restrict__ doff, int L, int nchunks, int chunks_per_row) {
  __shared__ __attribute__((aligned(16))) int s_buf[2][256];
  const int lane = threadIdx.x;
  int it = 0;
  for (int c = blockIdx.x; c < nchunks; c += gridDim.x, it++) {
    const int row = c / chunks_per_row, part = c % chunks_per_row;
    const unsigned s0 = soff[row] + part * 256, d0 = doff[row] + part * 256;
    const int n = L - part * 256 < 256 ? L - part * 256 : 256;
    if (MODE == 0) {
      int v[4];
#pragma unroll
      for (int k = 0; k < 4; k++) v[k] = k * 64 + lane < n ? __builtin_nontemporal_load(src + s0 + k * 64 + lane) : 0;
#pragma unroll
      for (int k = 0; k < 4; k++) if (k * 64 + lane < n) dst[d0 + k * 64 + lane] = v[k];
    } else if (MODE == 1) {
      if (4 * lane + 4 <= n) {
        const i4u q = *(const i4u *)(src + s0 + 4 * lane);
        *(i4u *)(dst + d0 + 4 * lane) = q;
      } else {
        for (int k = 0; k < 4; k++) if (4 * lane + k < n) dst[d0 + 4 * lane + k] = src[s0 + 4 * lane + k];
      }
    } else {
      int *buf = s_buf[it & 1];
      if (4 * lane + 4 <= n)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + s0 + 4 * lane),
                                         (__attribute__((address_space(3))) void *)buf, 16, 0, 0);
      else
        for (int k = 0; k < 4; k++) if (4 * lane + k < n) buf[4 * lane + k] = src[s0 + 4 * lane + k];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const int4 q = *(const int4 *)(buf + 4 * lane);
      if (4 * lane + 4 <= n) {
        i4u o; o.x = q.x, o.y = q.y, o.z = q.z, o.w = q.w;
        *(i4u *)(dst + d0 + 4 * lane) = o;
      } else {
        const int qq[4] = {q.x, q.y, q.z, q.w};
        for (int k = 0; k < 4; k++) if (4 * lane + k < n) dst[d0 + 4 * lane + k] = qq[k];
      }
    }
  }
}

int main() {
  const int64_t N = 72000000;  // entries
  int *src, *dst; CK(hipMalloc(&src, (N + 64) * 4)); CK(hipMalloc(&dst, (N + 64) * 4));
  CK(hipMemset(src, 1, (N + 64) * 4));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int L : {269, 804, 2482}) {
    const int nrows = (int)(N / (L + 8));
    const int cpr = (L + 255) / 256;
    for (int mis_s = 0; mis_s < 2; mis_s++) for (int mis_d = 0; mis_d < 2; mis_d++) {
      std::vector<unsigned> so(nrows), dof(nrows), perm(nrows);
      for (int i = 0; i < nrows; i++) perm[i] = i;
      uint64_t s = 88172645463325252ull;
      for (int i = nrows - 1; i > 0; i--) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; int j = (int)(s % (uint64_t)(i + 1)); std::swap(perm[i], perm[j]); }
      const unsigned pitch = ((L + 3) / 4) * 4 + 4;
      for (int i = 0; i < nrows; i++) { so[i] = perm[i] * pitch + mis_s; dof[i] = i * pitch + mis_d; }
      unsigned *dso, *ddo; CK(hipMalloc(&dso, nrows * 4)); CK(hipMalloc(&ddo, nrows * 4));
      CK(hipMemcpy(dso, so.data(), nrows * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(ddo, dof.data(), nrows * 4, hipMemcpyHostToDevice));
      const int nchunks = nrows * cpr;
      for (int wpc : {16, 32}) {
        const int grid = 256 * wpc;
        float ms[3];
        for (int mode = 0; mode < 3; mode++) {
          auto go = [&] {
            if (mode == 0) hipLaunchKernelGGL(k_copy<0>, dim3(grid), dim3(64), 0, 0, src, dst, dso, ddo, L, nchunks, cpr);
            if (mode == 1) hipLaunchKernelGGL(k_copy<1>, dim3(grid), dim3(64), 0, 0, src, dst, dso, ddo, L, nchunks, cpr);
            if (mode == 2) hipLaunchKernelGGL(k_copy<2>, dim3(grid), dim3(64), 0, 0, src, dst, dso, ddo, L, nchunks, cpr);
          };
          go(); CK(hipDeviceSynchronize());
          CK(hipEventRecord(a)); for (int r = 0; r < 5; r++) go(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
          CK(hipEventElapsedTime(&ms[mode], a, b)); ms[mode] /= 5;
        }
        const double bytes = 8.0 * nrows * L;
        printf("L=%4d src+%d dst+%d waves/CU=%2d: dword %.3f ms %.2f TB/s | 16B regs %.3f ms %.2f TB/s | 16B DMA %.3f ms %.2f TB/s\n", L, mis_s, mis_d, wpc,
               ms[0], bytes / ms[0] / 1e9, ms[1], bytes / ms[1] / 1e9, ms[2], bytes / ms[2] / 1e9);
      }
      CK(hipFree(dso)); CK(hipFree(ddo));
    }
  }
  return 0;
}
