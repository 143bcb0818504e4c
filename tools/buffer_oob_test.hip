// Do raw-buffer dwordx4 accesses range-check every dword on its own (gfx950)?  k_rows_quad relies on it: a row's
// descriptor ends at the row's last entry and the quad that straddles it is loaded / stored with one instruction.
// Prints one line per record count; exit code 1 if any dword behaves differently.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned *in, unsigned *out, unsigned *got, int nrec_bytes) {
  const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, nrec_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)out, 0, nrec_bytes, 0x00020000);
  const u4 x = __builtin_amdgcn_raw_buffer_load_b128(ri, threadIdx.x * 16, 0, 0);
  got[4 * threadIdx.x] = x.x, got[4 * threadIdx.x + 1] = x.y, got[4 * threadIdx.x + 2] = x.z, got[4 * threadIdx.x + 3] = x.w;
  u4 y;
  y.x = 1000 + 4 * threadIdx.x, y.y = y.x + 1, y.z = y.x + 2, y.w = y.x + 3;
  __builtin_amdgcn_raw_buffer_store_b128(y, ro, threadIdx.x * 16, 0, 0);
}
int main() {
  const int N = 64 * 4;
  unsigned *in, *out, *got, h_in[N], h_out[N], h_got[N];
  hipMalloc(&in, N * 4), hipMalloc(&out, N * 4), hipMalloc(&got, N * 4);
  for (int i = 0; i < N; i++) h_in[i] = 7 + i;
  hipMemcpy(in, h_in, N * 4, hipMemcpyHostToDevice);
  int bad = 0;
  for (int nrec = 0; nrec <= 23; nrec++) {
    hipMemset(out, 0xEE, N * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, in, out, got, nrec * 4);
    hipMemcpy(h_out, out, N * 4, hipMemcpyDeviceToHost);
    hipMemcpy(h_got, got, N * 4, hipMemcpyDeviceToHost);
    int lbad = 0, sbad = 0;
    for (int i = 0; i < N; i++) {
      const unsigned wl = i < nrec ? 7u + i : 0u, ws = i < nrec ? 1000u + i : 0xEEEEEEEEu;
      lbad += h_got[i] != wl;
      sbad += h_out[i] != ws;
    }
    printf("records %2d: load mismatches %d, store mismatches %d\n", nrec, lbad, sbad);
    bad += lbad + sbad;
  }
  printf(bad ? "PER-DWORD RANGE CHECK: NO\n" : "PER-DWORD RANGE CHECK: YES\n");
  return bad ? 1 : 0;
}
