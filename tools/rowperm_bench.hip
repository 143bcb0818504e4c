// Which side of a row permute should be the random one (diagnostic)?  Rows of L entries (column + value) are moved
// with a relabel gather per entry and no sort, either READ at random source offsets and written back to back (what the
// permute kernels do: they walk the NEW rows) or read back to back and WRITTEN to random destinations (walking the OLD
// rows).  Rows of up to 64 entries: 64 / G rows per wave (G lanes x 4 entries); longer rows: a wave per 256-entry chunk.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#include <algorithm>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
struct __attribute__((packed, aligned(4))) i4u { int x, y, z, w; };

template <int G>  // lanes per row (rows of at most 4 G entries); G = 64: chunks of long rows
__global__ __launch_bounds__(256) void k_perm(const int *__restrict__ col, const int *__restrict__ val, const int *__restrict__ table,
                                              int *__restrict__ col_out, int *__restrict__ val_out,
                                              const unsigned *__restrict__ soff, const unsigned *__restrict__ doff, int L, int nunits,
                                              int chunks_per_row) {
  const int lane = threadIdx.x & 63, lig = lane & (G - 1);
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = gridDim.x * 4;
  for (int u = wave; u < nunits; u += nwaves) {
    int row, q0;
    if (G == 64) {
      row = u / chunks_per_row;
      q0 = (u % chunks_per_row) * 256 + 4 * lane;
    } else {
      row = u * (64 / G) + lane / G;
      q0 = 4 * lig;
    }
    const unsigned s0 = soff[row] + q0, d0 = doff[row] + q0;
    const int n = L - q0;
    if (n >= 4) {
      const i4u c = *(const i4u *)(col + s0);
      const i4u v = *(const i4u *)(val + s0);
      i4u k;
      k.x = table[c.x], k.y = table[c.y], k.z = table[c.z], k.w = table[c.w];
      *(i4u *)(col_out + d0) = k;
      *(i4u *)(val_out + d0) = v;
    } else {
      for (int j = 0; j < n; j++) {
        col_out[d0 + j] = table[col[s0 + j]];
        val_out[d0 + j] = val[s0 + j];
      }
    }
  }
}

int main() {
  const int64_t N = 64000000;  // entries
  const int M = 1 << 22;
  int *col, *val, *co, *vo, *table;
  CK(hipMalloc(&col, (N + 64) * 4)); CK(hipMalloc(&val, (N + 64) * 4)); CK(hipMalloc(&co, (N + 64) * 4)); CK(hipMalloc(&vo, (N + 64) * 4));
  CK(hipMalloc(&table, M * 4));
  {
    std::vector<int> h(N + 64), t(M);
    uint64_t s = 88172645463325252ull;
    for (int64_t i = 0; i < N + 64; i++) {  // power-law-ish column ids: product of two uniforms
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;
      const double a = (double)(s & 0xFFFFFF) / 16777216.0, b = (double)((s >> 24) & 0xFFFFFF) / 16777216.0;
      h[i] = (int)(a * a * b * b * (M - 1));
    }
    for (int i = 0; i < M; i++) t[i] = i;
    for (int i = M - 1; i > 0; i--) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; std::swap(t[i], t[(int)(s % (uint64_t)(i + 1))]); }
    CK(hipMemcpy(col, h.data(), (N + 64) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(table, t.data(), M * 4, hipMemcpyHostToDevice));
    CK(hipMemset(val, 1, (N + 64) * 4));
  }
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int L : {6, 13, 28, 60, 269, 804, 2482}) {
    const int G = L <= 8 ? 2 : L <= 16 ? 4 : L <= 32 ? 8 : L <= 64 ? 16 : 64;
    const unsigned pitch = (unsigned)L;  // rows back to back, as in a CSR
    const int nrows = (int)(N / pitch) / 64 * 64;
    const int cpr = G == 64 ? (L + 255) / 256 : 1;
    const int nunits = G == 64 ? nrows * cpr : nrows / (64 / G);
    std::vector<unsigned> seq(nrows), rnd(nrows), perm(nrows);
    for (int i = 0; i < nrows; i++) perm[i] = i;
    uint64_t s = 1234567891234567ull;
    for (int i = nrows - 1; i > 0; i--) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; std::swap(perm[i], perm[(int)(s % (uint64_t)(i + 1))]); }
    for (int i = 0; i < nrows; i++) { seq[i] = i * pitch; rnd[i] = perm[i] * pitch; }
    unsigned *dseq, *drnd; CK(hipMalloc(&dseq, nrows * 4)); CK(hipMalloc(&drnd, nrows * 4));
    CK(hipMemcpy(dseq, seq.data(), nrows * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(drnd, rnd.data(), nrows * 4, hipMemcpyHostToDevice));
    for (int wpc : {16, 32}) {
      const int grid = 256 * wpc / 4;
      float ms[3];
      for (int mode = 0; mode < 3; mode++) {  // 0: random reads, 1: random writes, 2: neither
        const unsigned *so = mode == 0 ? drnd : dseq, *dof = mode == 1 ? drnd : dseq;
        auto go = [&] {
          switch (G) {
            case 2: hipLaunchKernelGGL(k_perm<2>, dim3(grid), dim3(256), 0, 0, col, val, table, co, vo, so, dof, L, nunits, cpr); break;
            case 4: hipLaunchKernelGGL(k_perm<4>, dim3(grid), dim3(256), 0, 0, col, val, table, co, vo, so, dof, L, nunits, cpr); break;
            case 8: hipLaunchKernelGGL(k_perm<8>, dim3(grid), dim3(256), 0, 0, col, val, table, co, vo, so, dof, L, nunits, cpr); break;
            case 16: hipLaunchKernelGGL(k_perm<16>, dim3(grid), dim3(256), 0, 0, col, val, table, co, vo, so, dof, L, nunits, cpr); break;
            default: hipLaunchKernelGGL(k_perm<64>, dim3(grid), dim3(256), 0, 0, col, val, table, co, vo, so, dof, L, nunits, cpr); break;
          }
        };
        go(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(a)); for (int r = 0; r < 5; r++) go(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms[mode], a, b)); ms[mode] /= 5;
      }
      const double ent = (double)nrows * L;
      printf("L=%4d G=%2d waves/CU=%2d: random READS %.3f ms %.2f ps/entry | random WRITES %.3f ms %.2f ps/entry | sequential %.3f ms %.2f ps/entry\n", L, G, wpc,
             ms[0], ms[0] * 1e9 / ent, ms[1], ms[1] * 1e9 / ent, ms[2], ms[2] * 1e9 / ent);
    }
    CK(hipFree(dseq)); CK(hipFree(drnd));
  }
  return 0;
}
