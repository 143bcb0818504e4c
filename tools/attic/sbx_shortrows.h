// sbx_shortrows.h — k_short_rows: the rows of 1 .. 64 entries of the CSR permute (round 4), included by
// sbx_permute.hip inside its anonymous namespace.
//
//   A5  permute/permute_order_two.cc:63-77 (relabel every column of a row through col_order) followed by
//   A4  format/csr.cc:118-157 (the constructor's per-row sort by (column, value))
//
// 87 % of a power-law matrix's rows have fewer than 16 entries; round 3's tile kernel spent ~430 wave instructions and
// 33 LDS operations per 64 entries on them (row map by max-scan, per-row min / max by LDS atomics, bucket counters,
// placement, ranking — all in LDS).  Here a row is sorted in REGISTERS:
//   * a workgroup takes SR_TILE consecutive new rows, reads their (length, source) records and output offsets once,
//     coalesced, and bins the rows of 1 .. 64 entries by length class in LDS (no global lists, no second pass over
//     the records, no prefix sums over the short rows);
//   * a class is served by groups of G = 2 / 4 / 8 / 16 lanes per row (rows of up to 8 / 16 / 32 / 64 entries), a lane
//     owns one QUAD — four consecutive entries, 16-byte loads and stores through bounds-checked buffer descriptors —
//     and every entry finds its rank by comparing itself with all entries of the row: the other lanes' quads come through
//     DPP (quad permutes, row shifts and rotates — xor patterns inside 16-lane rows), two instructions per pair, no LDS
//     and no waits; the word compared is column << 6 | position, so equal columns keep their input order (a stable sort,
//     as csr.cc's pair sort followed by k_fix_dup_runs needs);
//   * the sorted row goes through a 2 KB LDS patch of its wave (one ds_write_b64 per entry at its rank, two ds_read_b128
//     per quad) and leaves as 16-byte stores.
// Requires column ids below 2^25 and source arrays of less than 4 GB each (32-bit buffer offsets); otherwise the caller
// leaves these rows to the tile kernel.
#pragma once

constexpr int SR_THREADS = 256, SR_WAVES = SR_THREADS / 64;
constexpr int SR_TILE = 1024;   // new rows per workgroup
constexpr int SR_CLASSES = 4;   // G = 2, 4, 8, 16 lanes per row
constexpr int SR_MAX = 64;      // longest row served here
constexpr int SR_KEY_BITS = 25; // column << 6 | position stays below 2^31: the words of dead slots lie above every live one

__device__ __forceinline__ unsigned sr_q1(unsigned v) { return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, false); }  // lane ^ 1
__device__ __forceinline__ unsigned sr_q2(unsigned v) { return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, false); }  // lane ^ 2
__device__ __forceinline__ unsigned sr_q3(unsigned v) { return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x1B, 0xF, 0xF, false); }  // lane ^ 3
__device__ __forceinline__ unsigned sr_x4(unsigned v) {  // lane ^ 4: banks 0, 2 read four lanes up, banks 1, 3 four lanes down
  int t = __builtin_amdgcn_update_dpp((int)v, (int)v, 0x104, 0xF, 0x5, false);  // row_shl:4
  return (unsigned)__builtin_amdgcn_update_dpp(t, (int)v, 0x114, 0xF, 0xA, false);  // row_shr:4
}
__device__ __forceinline__ unsigned sr_x8(unsigned v) { return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xF, 0xF, false); }  // row_ror:8 = lane ^ 8

// rank of each of the lane's four words among the 4 G words of its group (all words distinct)
template <int G>
__device__ __forceinline__ void sr_rank(const unsigned (&w)[4], unsigned (&r)[4]) {
#pragma unroll
  for (int j = 0; j < 4; j++) {
    r[j] = 0;
#pragma unroll
    for (int i = 0; i < 4; i++)
      if (i != j) r[j] += w[i] < w[j];
  }
  auto acc = [&](const unsigned (&o)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) r[j] += o[i] < w[j];
  };
  auto via = [&](const unsigned (&b)[4], auto f) __attribute__((always_inline)) {
    unsigned o[4];
#pragma unroll
    for (int i = 0; i < 4; i++) o[i] = f(b[i]);
    acc(o);
  };
  auto quads = [&](const unsigned (&b)[4]) __attribute__((always_inline)) {  // the three other lanes of the quad, seen from b
    via(b, sr_q1);
    if (G >= 4) {
      via(b, sr_q2);
      via(b, sr_q3);
    }
  };
  quads(w);
  if (G >= 8) {
    unsigned u[4];
#pragma unroll
    for (int i = 0; i < 4; i++) u[i] = sr_x4(w[i]);
    acc(u);
    quads(u);
  }
  if (G >= 16) {
    unsigned t[4], u[4];
#pragma unroll
    for (int i = 0; i < 4; i++) t[i] = sr_x8(w[i]);
    acc(t);
    quads(t);
#pragma unroll
    for (int i = 0; i < 4; i++) u[i] = sr_x4(t[i]);
    acc(u);
    quads(u);
  }
}

template <int VB> struct SrLds {
  // per wave: 256 sorted (key, value) slots
  static constexpr int PATCH_WORDS = VB == 0 ? 256 : (VB == 4 ? 512 : 768);
};

template <typename I, int VB>
__global__ __launch_bounds__(SR_THREADS, 4) void k_short_rows(
    const int2 *__restrict__ rec, const I *col_in, const char *val_in, const I *__restrict__ col_order,
    const I *__restrict__ rpo, I *col_out, char *val_out, int64_t nr, PermState *__restrict__ st, unsigned col_bytes,
    unsigned val_bytes, unsigned table_bytes) {
  static_assert(sizeof(I) == 4, "32-bit indices");
  typedef typename RqVal<VB>::type V;
  constexpr bool HASV = VB != 0;
  __shared__ int2 s_rec[SR_TILE];
  __shared__ int s_out[SR_TILE];
  __shared__ unsigned short s_list[SR_CLASSES][SR_TILE];
  __shared__ unsigned s_cnt[SR_CLASSES], s_next[SR_CLASSES];
  __shared__ __attribute__((aligned(16))) unsigned s_patch[SR_WAVES][SrLds<VB>::PATCH_WORDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t r0 = (int64_t)blockIdx.x * SR_TILE;
  if (tid < SR_CLASSES) s_cnt[tid] = 0, s_next[tid] = 0;
  __syncthreads();
  // ---- the tile's records, read once; rows of 1 .. 64 entries binned by length class
#pragma unroll
  for (int k = 0; k < SR_TILE / SR_THREADS; k++) {
    const int i = k * SR_THREADS + tid;
    const int64_t r = r0 + i;
    int2 rc = make_int2(0, 0);
    int o = 0;
    if (r < nr) {
      rc = rec[r];
      o = (int)rpo[r];
    }
    s_rec[i] = rc;
    s_out[i] = o;
    if (rc.x > 0 && rc.x <= SR_MAX) {
      const int c = rc.x <= 8 ? 0 : (rc.x <= 16 ? 1 : (rc.x <= 32 ? 2 : 3));
      s_list[c][atomicAdd(&s_cnt[c], 1u)] = (unsigned short)i;
    }
  }
  __syncthreads();
  const __amdgpu_buffer_rsrc_t cb = __builtin_amdgcn_make_buffer_rsrc((void *)col_in, 0, (int)col_bytes, RQ_RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t vbuf = __builtin_amdgcn_make_buffer_rsrc((void *)val_in, 0, (int)val_bytes, RQ_RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t tab = __builtin_amdgcn_make_buffer_rsrc((void *)col_order, 0, (int)table_bytes, RQ_RSRC_FLAGS);
  bool any_uns = false, any_dup = false;

  // One class = a queue of rows in LDS; a wave claims batches of 64 / G rows and runs them as a three-stage software
  // pipeline: while batch i is ranked and stored, the relabel gathers of batch i + 1 and the column / value loads of
  // batch i + 2 are in flight (one batch at a time left a wave with a single dependent chain of three memory round
  // trips and the CU's memory path idle most of the time: 2.3 x the time of this form).
  struct Stage {
    unsigned x[4];  // columns of the lane's quad
    V v[HASV ? 4 : 1];
    int nv;         // live entries of the quad
    int outp;       // where the quad goes in the output
  };
  auto serve = [&](auto gtag, const int c) __attribute__((always_inline)) {
    constexpr int G = decltype(gtag)::value;
    const unsigned cnt = s_cnt[c];
    const int lig = lane & (G - 1), grp = lane / G, q0 = 4 * lig;
    auto claim = [&]() __attribute__((always_inline)) {
      unsigned b = 0;
      if (lane == 0) b = atomicAdd(&s_next[c], (unsigned)(64 / G));
      return (unsigned)__builtin_amdgcn_readfirstlane((int)b);
    };
    auto load = [&](const unsigned b) __attribute__((always_inline)) {
      Stage sg;
      const unsigned slot = b + (unsigned)grp;
      const bool valid = slot < cnt;
      const int ri = valid ? (int)s_list[c][slot] : 0;
      const int2 rc = s_rec[ri];
      const int len = valid ? rc.x : 0;
      sg.nv = len - q0 < 0 ? 0 : (len - q0 > 4 ? 4 : len - q0);
      sg.outp = s_out[ri] + q0;
      // the quad's columns and values: one 16-byte buffer load each (a dead quad reads out of range: zeros, no traffic)
      const unsigned off = sg.nv > 0 ? (unsigned)(rc.y + q0) * 4u : 0xFFFFFFF0u;
      const sbx_u4 x = __builtin_amdgcn_raw_buffer_load_b128(cb, off, 0, RQ_NT);
      sg.x[0] = x.x, sg.x[1] = x.y, sg.x[2] = x.z, sg.x[3] = x.w;
      if (VB == 4) {
        const sbx_u4 y = __builtin_amdgcn_raw_buffer_load_b128(vbuf, off, 0, RQ_NT);
        sg.v[0] = (V)y.x, sg.v[1] = (V)y.y, sg.v[2] = (V)y.z, sg.v[3] = (V)y.w;
      } else if (VB == 8) {
        const unsigned off8 = sg.nv > 0 ? (unsigned)(rc.y + q0) * 8u : 0xFFFFFFF0u;
        const sbx_u4 y = __builtin_amdgcn_raw_buffer_load_b128(vbuf, off8, 0, RQ_NT);
        const sbx_u4 z = __builtin_amdgcn_raw_buffer_load_b128(vbuf, off8 + 16u, 0, RQ_NT);
        sg.v[0] = (V)(((uint64_t)y.y << 32) | y.x), sg.v[1] = (V)(((uint64_t)y.w << 32) | y.z);
        sg.v[2] = (V)(((uint64_t)z.y << 32) | z.x), sg.v[3] = (V)(((uint64_t)z.w << 32) | z.z);
      }
      return sg;
    };
    // relabel (permute_order_two.cc:68); entries behind the row's end hold a neighbour row's column or 0: never used
    auto gather = [&](const Stage &sg, unsigned (&k)[4]) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 4; j++) k[j] = col_order ? __builtin_amdgcn_raw_buffer_load_b32(tab, sg.x[j] << 2, 0, 0) : sg.x[j];
    };
    auto process = [&](const unsigned (&k)[4], const Stage &sg) __attribute__((always_inline)) {
      const int nv = sg.nv;
      // order of the row as it stands (csr.cc:102-116 decides on it whether anything is sorted at all)
      unsigned w[4];
      {
        const unsigned prev = (unsigned)sbx_dpp<SBX_DPP_ROW_SHR + 1>((int)k[0], (int)k[3]);  // the lane in front (lane 0 of a row: itself)
        bool uns = false;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const unsigned y = j ? k[j - 1] : prev;
          uns |= (j < nv) & (j > 0 || lig > 0) & (k[j] < y);
          w[j] = j < nv ? (k[j] << 6) | (unsigned)(q0 + j) : 0xFFFFFFFFu;  // (dead slots: behind every live word)
        }
        any_uns |= uns;
      }
      unsigned r[4];
      sr_rank<G>(w, r);
      // through the wave's LDS patch: every entry to its rank, every lane reads its quad of the sorted row back
      unsigned *const pw = s_patch[wv];
      const int base = 4 * (lane - lig);  // = grp * 4 G
      unsigned ck[4];
      V cv[4];
      if (VB == 4) {
        uint2 *const pp = (uint2 *)pw;
#pragma unroll
        for (int j = 0; j < 4; j++) pp[base + (int)(r[j] & (4 * G - 1))] = make_uint2(k[j], (unsigned)sg.v[j]);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        const sbx_u4 a = *(const sbx_u4 *)&pp[base + q0], bq = *(const sbx_u4 *)&pp[base + q0 + 2];
        ck[0] = a.x, ck[1] = a.z, ck[2] = bq.x, ck[3] = bq.z;
        cv[0] = (V)a.y, cv[1] = (V)a.w, cv[2] = (V)bq.y, cv[3] = (V)bq.w;
      } else {
        uint64_t *const p8 = (uint64_t *)(pw + 256);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          pw[base + (int)(r[j] & (4 * G - 1))] = k[j];
          if (VB == 8) p8[base + (int)(r[j] & (4 * G - 1))] = sg.v[j];
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        const sbx_u4 a = *(const sbx_u4 *)&pw[base + q0];
        ck[0] = a.x, ck[1] = a.y, ck[2] = a.z, ck[3] = a.w;
        if (VB == 8) {
#pragma unroll
          for (int j = 0; j < 4; j++) cv[j] = (V)p8[base + q0 + j];
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // (the next batch rewrites the patch)
      {  // duplicate columns (csr.cc:143-156 orders them by value afterwards, k_fix_dup_runs)
        const unsigned prev = (unsigned)sbx_dpp<SBX_DPP_ROW_SHR + 1>((int)ck[0], (int)ck[3]);
        bool dup = false;
#pragma unroll
        for (int j = 0; j < 4; j++) dup |= (j < nv) & (j > 0 || lig > 0) & (ck[j] == (j ? ck[j - 1] : prev));
        any_dup |= dup;
      }
      const int64_t o = (int64_t)sg.outp;
      if (nv == 4) {
        sbx_i4a xo;
        xo.x = (int)ck[0], xo.y = (int)ck[1], xo.z = (int)ck[2], xo.w = (int)ck[3];
        *(sbx_i4a *)(col_out + o) = xo;
        if (VB == 4) {
          sbx_i4a yo;
          yo.x = (int)cv[0], yo.y = (int)cv[1], yo.z = (int)cv[2], yo.w = (int)cv[3];
          *(sbx_i4a *)((uint32_t *)val_out + o) = yo;
        } else if (VB == 8) {
          sbx_l2a yo, zo;
          yo.x = cv[0], yo.y = cv[1], zo.x = cv[2], zo.y = cv[3];
          *(sbx_l2a *)((uint64_t *)val_out + o) = yo;
          *(sbx_l2a *)((uint64_t *)val_out + o + 2) = zo;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 3; j++)
          if (j < nv) {
            col_out[o + j] = (I)ck[j];
            if (HASV) ((V *)val_out)[o + j] = cv[j];
          }
      }
    };
    unsigned bA = claim();
    if (bA >= cnt) return;
    unsigned bB = claim();
    Stage A = load(bA);
    unsigned kA[4];
    gather(A, kA);
    Stage B = load(bB);
    while (bA < cnt) {
      const unsigned bC = claim();
      unsigned kB[4];
      gather(B, kB);
      const Stage C = load(bC);
      process(kA, A);
      A = B, B = C;
#pragma unroll
      for (int j = 0; j < 4; j++) kA[j] = kB[j];
      bA = bB, bB = bC;
    }
  };
  // (longest rows first: the batches of the last class to run are the cheapest to be left waiting for)
  serve(std::integral_constant<int, 16>(), 3);
  serve(std::integral_constant<int, 8>(), 2);
  serve(std::integral_constant<int, 4>(), 1);
  serve(std::integral_constant<int, 2>(), 0);
  if (__any(any_uns) && lane == 0) st->any_unsorted = 1;
  if (__any(any_dup) && lane == 0) st->any_dup = 1;
}
