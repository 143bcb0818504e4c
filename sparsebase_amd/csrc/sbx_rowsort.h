// sbx_rowsort.h — k_rows_quad: the row-class kernel of the CSR permute (round 4), included by sbx_permute.hip inside its
// anonymous namespace.
//
//   A5  permute/permute_order_two.cc:63-77 (relabel every column of a row through col_order) followed by
//   A4  format/csr.cc:118-157 (the constructor's per-row sort by (column, value))
//
// One workgroup of T threads sorts one row of up to CAP = 4 T Q relabelled columns in LDS; the workgroups are persistent
// and walk their class list as a software pipeline (column loads of row i + 2 and relabel gathers + value loads of row
// i + 1 in flight while row i is sorted).  What is different from round 3's k_permute_block_rows is the instruction
// count, not the algorithm:
//   * a lane owns QUADS — four consecutive entries — so columns, values and the sorted row move 16 bytes per
//     instruction (global_load/store_dwordx4, ds_read_b128), min / max / order checks stay in registers (one DPP shift
//     per quad for the neighbour lane) and the per-entry predicates disappear from every quad step that lies inside
//     the row (a wave-uniform test); only the wave that holds the row's end runs the predicated form;
//   * the count pass takes the entry's arrival number from a RETURNING LDS add, so placement needs no second atomic
//     sweep: slot = bucket start + arrival;
//   * ranking inside a bucket reads four neighbour words unrolled and loops only for buckets of more than four;
//   * the sorted (column, value) pairs go to LDS with one ds_write_b64 and come back as two ds_read_b128 per quad;
//   * rows are read and written through BUFFER instructions with a per-row resource descriptor (base = the row's first
//     entry, num_records = its byte length): the hardware drops every dword outside the row — loads return 0, stores
//     write nothing — so no load or store of the kernel carries a predicate or a clamped address, dead lanes cost no
//     memory traffic, and an address is one 32-bit offset that never changes; the relabel gather is one shift and one
//     bounds-checked buffer_load_dword through a descriptor of the column map;
//   * six barriers per row (none for T = 64) instead of about twelve; the pipeline registers ping-pong between two
//     unrolled copies of the loop body instead of being moved; the row-level flags live in two parity sets so that no
//     barrier is needed just to reset them.
// The bucket-rank sort itself (order-preserving buckets from the row's own min / max, a second interpolation level
// for clustered columns, the LSD radix kernel for what stays overfull) is described at the top of sbx_permute.hip.
#pragma once

typedef int sbx_i4a __attribute__((ext_vector_type(4), aligned(4)));  // 16-byte access at 4-byte alignment
typedef unsigned sbx_u4 __attribute__((ext_vector_type(4)));
typedef unsigned long long sbx_l2a __attribute__((ext_vector_type(2), aligned(8)));

#ifndef RQ_GPOINTS
#define RQ_GPOINTS 8  // issue points of a row's relabel gathers inside the sort of the row in front of it
#endif
#define RQ_NEIGHBOURS 8  // words of its own bucket an entry ranks itself against without a loop; a fuller bucket at level 0 sends the row to level 1

template <int VB> struct RqVal { typedef uint32_t type; };
template <> struct RqVal<8> { typedef uint64_t type; };

// LDS words of one workgroup (T threads, Q quads per thread)
template <int VB, int T, int Q>
struct RqLds {
  static constexpr int CAP = 4 * T * Q;
  static constexpr int CNT = CAP + 8;                  // 4 words in front (the one before bucket 0 stays 0), CAP counters, pad
  static constexpr int WRD = CAP + 12;                 // placed words (+ what the ranking step may read behind the last one)
  static constexpr int KEY = VB == 4 ? 0 : CAP + 12;   // sorted keys (VB 4: inside the pair array)
  static constexpr int PAY = VB == 0 ? 0 : 2 * (CAP + 12);  // (key, value) pairs / 8-byte values
  static constexpr int WORDS = CNT + WRD + KEY + PAY;
  static constexpr int BYTES = 4 * WORDS + 1024;       // + the small arrays
};

#define RQ_INLINE __attribute__((always_inline))
#define RQ_RSRC_FLAGS 0x00020000  // raw buffer, 32-bit data format (gfx9 descriptor word 3)
#define RQ_NT 2                   // cache policy of the streamed loads: nt
// all vector-memory counters drained (vmcnt 0, the other counters untouched): placed where every load of the step has
// landed long ago, so that the waits the compiler derives for the next step never include this step's row stores
#define RQ_WAIT_LOADS() __builtin_amdgcn_s_waitcnt(0x0F70)

template <typename I, int VB, int T, int Q, int MINW = 1>
__global__ __launch_bounds__(T, MINW) void k_rows_quad(
    const int2 *__restrict__ rec, const I *col_in, const char *val_in, const I *__restrict__ col_order,
    const I *__restrict__ rpo, const I *__restrict__ rows, int n_rows, I *col_out, char *val_out,
    PermState *__restrict__ st, int force_radix, unsigned *__restrict__ fb_rows, unsigned *__restrict__ fb_count,
    const unsigned *__restrict__ n_rows_dev, unsigned table_bytes) {
  static_assert(sizeof(I) == 4 || sizeof(I) == 8, "32- or 64-bit index arrays (a column is a 32-bit key in LDS either way: m < 2^31)");
  constexpr int IB = (int)sizeof(I);  // bytes of a column / row-pointer word in memory
  typedef typename RqVal<VB>::type V;
  constexpr bool HASV = VB != 0;
  constexpr int W = T / 64, E = 4 * Q, CAP = 4 * T * Q, CHUNKS = Q * W;
  static_assert(CHUNKS <= 64, "one lane per 256-entry chunk in the boundary step");
  typedef RqLds<VB, T, Q> L;
  __shared__ __attribute__((aligned(16))) unsigned s_pool[L::WORDS];
  __shared__ unsigned s_scan[W + 1];
  __shared__ unsigned s_first[2][CHUNKS + 1], s_last[2][CHUNKS + 1];
  // per row parity: [0] min, [1] max of the relabelled columns (LDS atomics of the waves), [2] row out of order,
  // [4] fullest bucket if it holds more than RQ_NEIGHBOURS entries
  __shared__ unsigned s_row[2][8];
  unsigned *const s_cnt = s_pool + 4;                // s_cnt[-1] == 0
  unsigned *const s_w = s_pool + L::CNT;
  unsigned *const s_key = s_pool + L::CNT + L::WRD;  // VB 0 / 8
  uint2 *const s_pair = (uint2 *)(s_pool + L::CNT + L::WRD);                // VB 4: (key, value)
  uint64_t *const s_v8 = (uint64_t *)(s_pool + L::CNT + L::WRD + L::KEY);   // VB 8
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // (a scalar: the tests on it are branches, not exec masks)
  const int G = (int)gridDim.x;
  if (n_rows_dev) n_rows = (int)*n_rows_dev;
  const int my_rows = ((int)blockIdx.x < n_rows) ? (n_rows - 1 - (int)blockIdx.x) / G + 1 : 0;
  if (my_rows == 0) return;
  // the column map as a bounds-checked buffer (an index outside it reads 0)
  const __amdgpu_buffer_rsrc_t tab = __builtin_amdgcn_make_buffer_rsrc((void *)col_order, 0, (int)table_bytes, RQ_RSRC_FLAGS);

  auto rq_barrier = [&]() RQ_INLINE {  // LDS-only barrier: the prefetches of the rows to come stay in flight across it
    if (T > 64) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  auto row_reset = [&](int par) RQ_INLINE {
    s_row[par][0] = 0xFFFFFFFFu;
    s_row[par][1] = 0, s_row[par][2] = 0, s_row[par][4] = 0;
  };

  // pipeline state (row-level values travel in vector registers: their loads stay off lgkmcnt)
  int rid_d = 0;
  bool valid_d = false;
  int e0_c = 0, len_c = -1, src_c = 0, rid_c = 0;  // stage C: record loaded -> its columns are loaded this step
  int e0_b = 0, len_b = -1, src_b = 0, rid_b = 0;  // stage B: columns loaded -> gathers + value loads this step
  int e0_a = 0, len_a = -1, src_a = 0, rid_a = 0;  // stage A: sorted this step (its values are loaded at the top of the step)
  int cq[Q][4];
  int k0[Q][4], k1[Q][4];
#pragma unroll
  for (int q = 0; q < Q; q++)
#pragma unroll
    for (int j = 0; j < 4; j++) cq[q][j] = 0, k0[q][j] = 0, k1[q][j] = 0;
  int zero = 0;
  asm volatile("" : "+v"(zero));
  if (tid == 0) {
    s_pool[3] = 0;
    row_reset(0), row_reset(1);
  }

  // one pipeline step: ka = the relabelled columns of the row sorted now; kb = the target of this step's gathers
  auto step = [&](const int it, int (&ka)[Q][4], int (&kb)[Q][4]) RQ_INLINE {
    const int par = it & 1;
    // ---- E / D: list entry of row `it`, record of row it - 1
    const bool valid_e = it < my_rows;
    const int rid_e = (int)rows[(int64_t)blockIdx.x + (int64_t)(valid_e ? it : 0) * G + zero];
    const int rid_s = valid_d ? rid_d : 0;
    const int r0_d = (int)rpo[rid_s], r1_d = (int)rpo[rid_s + 1], src_d = rec[rid_s].y;
    // ---- B: relabel gathers (permute_order_two.cc:68) of row it - 3; its columns are in cq (entries past the row's
    // end hold 0: they gather map entry 0 and are never used).  They are NOT issued here in one burst: a wave that
    // has handed the memory pipeline a dozen instructions waits at the next one until the queue has room, and would
    // start its sort only then; gather_pt(k) issues the k-th eighth of them, between the phases of the sort
    auto gather_span = [&](const int lo, const int hi) RQ_INLINE {
#pragma unroll
      for (int i = 0; i < 4 * Q; i++)
        if (i >= lo && i < hi)
          kb[i >> 2][i & 3] = (col_order && !(force_radix & 4)) ? (int)__builtin_amdgcn_raw_buffer_load_b32(tab, (unsigned)cq[i >> 2][i & 3] * (unsigned)IB, 0, 0)  /* (64-bit map: the low word) */ : cq[i >> 2][i & 3];  // (bit 2: timing ablation without the gathers)
    };
    auto gather_pt = [&](const int k) RQ_INLINE { gather_span(k * (4 * Q) / RQ_GPOINTS, (k + 1) * (4 * Q) / RQ_GPOINTS); };
    auto gather_rest = [&](const int k) RQ_INLINE { gather_span(k * (4 * Q) / RQ_GPOINTS, 4 * Q); };
    // ---- the values of row it - 4, the row sorted in this step: they are needed when the sort ends
    V va[HASV ? Q : 1][4];
    if (HASV) {
      const int lena = __builtin_amdgcn_readfirstlane(len_a), srca = __builtin_amdgcn_readfirstlane(src_a);
      const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
          (void *)(val_in + (int64_t)(lena > 0 ? srca : 0) * VB), 0, (lena > 0 ? lena : 0) * VB, RQ_RSRC_FLAGS);
#pragma unroll
      for (int q = 0; q < Q; q++) {
        const unsigned off = (unsigned)(q * 4 * T + 4 * tid) * VB;
        if (VB == 4) {
          const sbx_u4 x = __builtin_amdgcn_raw_buffer_load_b128(rv, off, 0, RQ_NT);
          va[q][0] = (V)x.x, va[q][1] = (V)x.y, va[q][2] = (V)x.z, va[q][3] = (V)x.w;
        } else {
          const sbx_u4 x = __builtin_amdgcn_raw_buffer_load_b128(rv, off, 0, RQ_NT);
          const sbx_u4 y = __builtin_amdgcn_raw_buffer_load_b128(rv, off + 16, 0, RQ_NT);
          va[q][0] = (V)(((uint64_t)x.y << 32) | x.x), va[q][1] = (V)(((uint64_t)x.w << 32) | x.z);
          va[q][2] = (V)(((uint64_t)y.y << 32) | y.x), va[q][3] = (V)(((uint64_t)y.w << 32) | y.z);
        }
      }
    }
    // ---- C: columns of row it - 2, loaded behind the last gather of the step (which reads cq)
    auto load_cols = [&]() RQ_INLINE {
      const int lenc = __builtin_amdgcn_readfirstlane(len_c), srcc = __builtin_amdgcn_readfirstlane(src_c);
      const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(
          (void *)(col_in + (int64_t)(lenc > 0 ? srcc : 0)), 0, (lenc > 0 ? lenc : 0) * IB, RQ_RSRC_FLAGS);
#pragma unroll
      for (int q = 0; q < Q; q++) {
        const unsigned off = (unsigned)(q * 4 * T + 4 * tid) * (unsigned)IB;
        const sbx_u4 x = __builtin_amdgcn_raw_buffer_load_b128(rc, off, 0, RQ_NT);
        if (IB == 4) {
          cq[q][0] = (int)x.x, cq[q][1] = (int)x.y, cq[q][2] = (int)x.z, cq[q][3] = (int)x.w;
        } else {  // 64-bit columns: two entries per 16 bytes, the low words are the keys
          const sbx_u4 y = __builtin_amdgcn_raw_buffer_load_b128(rc, off + 16, 0, RQ_NT);
          cq[q][0] = (int)x.x, cq[q][1] = (int)x.z, cq[q][2] = (int)y.x, cq[q][3] = (int)y.z;
        }
      }
    };

    // ---- A: sort row it - 4
    const int len = __builtin_amdgcn_readfirstlane(len_a);
    const int e0 = __builtin_amdgcn_readfirstlane(e0_a);
    // body(q, j, p, live) for every entry slot of the quad steps this wave has entries in (a wave-uniform test); `live`
    // is false for the slots behind the row's end in the wave that holds it
    auto for_slots = [&](auto &&body) RQ_INLINE {
#pragma unroll
      for (int q = 0; q < Q; q++) {
        const int p0 = q * 4 * T + 4 * tid;
        if (q * 4 * T + w * 256 < len) {
#pragma unroll
          for (int j = 0; j < 4; j++) body(q, j, p0 + j, p0 + j < len);
        }
      }
    };
    // writes the row: fill(q, c, v) supplies the four entries of quad q; the stores carry no predicate (the row's
    // descriptors drop what lies past its end)
    auto store_row = [&](auto &&fill) RQ_INLINE {
      const __amdgpu_buffer_rsrc_t oc = __builtin_amdgcn_make_buffer_rsrc((void *)(col_out + (int64_t)e0), 0, len * IB, RQ_RSRC_FLAGS);
      const __amdgpu_buffer_rsrc_t ov = __builtin_amdgcn_make_buffer_rsrc((void *)(val_out + (int64_t)e0 * VB), 0, len * VB, RQ_RSRC_FLAGS);
#pragma unroll
      for (int q = 0; q < Q; q++) {
        if (q * 4 * T + w * 256 < len) {
          unsigned c[4];
          V v[4];
          fill(q, c, v);
          const unsigned off = (unsigned)(q * 4 * T + 4 * tid);
          sbx_u4 x;
          if (IB == 4) {
            x.x = c[0], x.y = c[1], x.z = c[2], x.w = c[3];
            __builtin_amdgcn_raw_buffer_store_b128(x, oc, off * 4, 0, 0);
          } else {  // widened: (c, 0) pairs
            sbx_u4 x2;
            x.x = c[0], x.y = 0u, x.z = c[1], x.w = 0u;
            x2.x = c[2], x2.y = 0u, x2.z = c[3], x2.w = 0u;
            __builtin_amdgcn_raw_buffer_store_b128(x, oc, off * 8, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(x2, oc, off * 8 + 16, 0, 0);
          }
          if (VB == 4) {
            sbx_u4 y;
            y.x = (unsigned)v[0], y.y = (unsigned)v[1], y.z = (unsigned)v[2], y.w = (unsigned)v[3];
            __builtin_amdgcn_raw_buffer_store_b128(y, ov, off * 4, 0, 0);
          } else if (VB == 8) {
            sbx_u4 y, z;
            y.x = (unsigned)v[0], y.y = (unsigned)((uint64_t)v[0] >> 32), y.z = (unsigned)v[1], y.w = (unsigned)((uint64_t)v[1] >> 32);
            z.x = (unsigned)v[2], z.y = (unsigned)((uint64_t)v[2] >> 32), z.z = (unsigned)v[3], z.w = (unsigned)((uint64_t)v[3] >> 32);
            __builtin_amdgcn_raw_buffer_store_b128(y, ov, off * 8, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(z, ov, off * 8 + 16, 0, 0);
          }
        }
      }
    };
    int out_mode = 0;  // 1: sorted row in LDS, streamed out after the rotation
    if (len >= 0) {
      // A0: min / max / order inside the lane's quads; the lane in front through DPP, the chunk in front through LDS
      unsigned mn = 0xFFFFFFFFu, mx = 0;
      bool uns = false, dupq = false;
#pragma unroll
      for (int q = 0; q < Q; q++) {
        const int p0 = q * 4 * T + 4 * tid;
        // (a wave's first lane gets its own first key: its neighbour is compared in the boundary step below)
        const unsigned prev = (unsigned)sbx_wave_shift_up1(ka[q][3], ka[q][0]);
        if (q * 4 * T + (w + 1) * 256 <= len) {
          const unsigned a = (unsigned)ka[q][0], b = (unsigned)ka[q][1], c = (unsigned)ka[q][2], d = (unsigned)ka[q][3];
          mn = min(mn, min(min(a, b), min(c, d)));
          mx = max(mx, max(max(a, b), max(c, d)));
          uns |= (b < a) | (c < b) | (d < c) | (a < prev);
          dupq |= (b == a) | (c == b) | (d == c) | ((a == prev) & (lane != 0));
        } else if (q * 4 * T + w * 256 < len) {
#pragma unroll
          for (int j = 0; j < 4; j++) {
            if (p0 + j < len) {
              const unsigned x = (unsigned)ka[q][j], y = j ? (unsigned)ka[q][j - 1] : prev;
              mn = min(mn, x), mx = max(mx, x);
              uns |= x < y;
              dupq |= (x == y) & (j > 0 || lane != 0);
            }
          }
        }
        if (lane == 63) s_last[par][q * W + w] = (unsigned)ka[q][3];
        if (lane == 0) s_first[par][q * W + w] = (unsigned)ka[q][0];
      }
      mn = sbx_wave_min(mn);
      mx = sbx_wave_max(mx);
      if (lane == 0) {
        atomicMin(&s_row[par][0], mn);
        atomicMax(&s_row[par][1], mx);
      }
      if (__any(uns) && lane == 0) s_row[par][2] = 1;
      // the counters of this row (the scan reads all CAP of them)
#pragma unroll
      for (int q = 0; q < Q; q++) *(sbx_u4 *)&s_cnt[q * 4 * T + 4 * tid] = (sbx_u4)(0u);
      if (tid < 4) s_cnt[CAP + tid] = 0;
      rq_barrier();  // B1
      if (tid == 0) row_reset(par ^ 1);  // every wave is through with the row before; the next row's waves come after B2
      bool row_uns = s_row[par][2] != 0, row_dup = __any(dupq);
      mn = s_row[par][0], mx = s_row[par][1];
      {  // chunk boundaries: chunk i (256 entries) against the last entry of chunk i - 1
        const int i = lane < CHUNKS ? lane : 0;
        const unsigned f = s_first[par][i], l = s_last[par][i > 0 ? i - 1 : 0];
        const bool live = lane > 0 && lane < CHUNKS && lane * 256 < len;
        row_uns |= __any(live && f < l);
        row_dup |= __any(live && f == l);
      }
      if (force_radix & 2) row_uns = false;  // (timing ablation: rows stream out unsorted)
      if (!row_uns) {
        // an ordered row needs no sort (csr.cc:102-116): straight out of the registers
        gather_rest(0);
        load_cols();
        RQ_WAIT_LOADS();
        store_row([&](int q, unsigned (&c)[4], V (&v)[4]) RQ_INLINE {
#pragma unroll
          for (int j = 0; j < 4; j++) {
            c[j] = (unsigned)ka[q][j];
            if (HASV) v[j] = va[q][j];
          }
        });
        if (row_dup && tid == 0) st->any_dup = 1;
        rq_barrier();  // (B2: the reset above is in front of the next row's atomics)
      } else {
        if (tid == 0) st->any_unsorted = 1;
        const int rbits = bits_u32(mx - mn);
        const int ib = bits_u32((unsigned)len - 1);  // 2^ib buckets, len <= 2^ib <= CAP
        const int shift = rbits > ib ? rbits - ib : 0;
        const unsigned lowmask = shift >= 32 ? 0xFFFFFFFFu : (1u << shift) - 1u;  // shift + ib = max(rbits, ib) <= 32
        // bucket and arrival number inside the bucket; the slots behind the row's end count into a bucket of their own
        // behind the counters the scan covers (s_cnt[CAP]): they land behind the row and are never stored
        unsigned bk[Q][4], ar[Q][4];
        bool to_radix = false;
        // level 0: buckets of equal width.  Level 1 (clustered columns): every bucket is split into as many sub-buckets
        // as it holds entries, by interpolation.
        for (int level = 0;; level++) {
          if (level == 0) {
            for_slots([&](int q, int j, int p, bool live) RQ_INLINE {
              const unsigned rel = (unsigned)ka[q][j] - mn;
              bk[q][j] = live ? (shift >= 32 ? 0u : rel >> shift) : (unsigned)CAP;
            });
          }
          for_slots([&](int q, int j, int p, bool live) RQ_INLINE { ar[q][j] = atomicAdd(&s_cnt[bk[q][j]], 1u); });
          if (level == 0) gather_pt(0);
          rq_barrier();  // B2
          if (level == 0) gather_pt(1);
          {
            // in-place inclusive scan of the counters; a thread owns E consecutive ones
            unsigned v[E];
            unsigned sum = 0, mxc = 0;
#pragma unroll
            for (int q = 0; q < Q; q++) {
              const sbx_u4 x = *(const sbx_u4 *)&s_cnt[E * tid + 4 * q];
              v[4 * q] = x.x, v[4 * q + 1] = x.y, v[4 * q + 2] = x.z, v[4 * q + 3] = x.w;
            }
#pragma unroll
            for (int i = 0; i < E; i++) {
              mxc = max(mxc, v[i]);
              sum += v[i];
              v[i] = sum;
            }
            const unsigned inc = sbx_wave_inclusive_sum(sum);
            if (lane == 63) s_scan[w] = inc;
            const unsigned wmx = sbx_wave_max(mxc);
            if (wmx > (unsigned)RQ_NEIGHBOURS && lane == 0) atomicMax(&s_row[par][4], wmx);
            if (level == 0) gather_pt(2);
            rq_barrier();  // B3
            unsigned ex = inc - sum;
#pragma unroll
            for (int i = 0; i < W; i++)
              if (i < w) ex += s_scan[i];
#pragma unroll
            for (int q = 0; q < Q; q++) {
              sbx_u4 x;
              x.x = v[4 * q] + ex, x.y = v[4 * q + 1] + ex, x.z = v[4 * q + 2] + ex, x.w = v[4 * q + 3] + ex;
              *(sbx_u4 *)&s_cnt[E * tid + 4 * q] = x;
            }
          }
          if (level == 0) gather_pt(3);
          rq_barrier();  // B4
          const unsigned big = s_row[par][4];  // 0, or the fullest bucket if it holds more than the ranking step unrolls
          if (level == 1 || (big == 0 && !(force_radix & 8))) {  // (bit 3: timing ablation, level 1 always)
            to_radix = big > (unsigned)BK_MAX || (force_radix & 1);
            break;
          }
          for_slots([&](int q, int j, int p, bool live) RQ_INLINE {
            const unsigned b = bk[q][j];
            const unsigned start = s_cnt[(int)b - 1], cntb = s_cnt[b] - start;
            const unsigned low = ((unsigned)ka[q][j] - mn) & lowmask;
            const unsigned sub = shift ? (unsigned)(((unsigned long long)low * cntb) >> shift) : 0u;
            bk[q][j] = live ? start + sub : (unsigned)CAP;  // < start + cntb: sub-buckets of different buckets do not meet
          });
          rq_barrier();  // the level-0 bounds and the fullest-bucket word have been read
#pragma unroll
          for (int q = 0; q < Q; q++) *(sbx_u4 *)&s_cnt[q * 4 * T + 4 * tid] = (sbx_u4)(0u);
          if (tid < 4) s_cnt[CAP + tid] = 0;
          if (tid == 0) s_row[par][4] = 0;
          rq_barrier();
        }
        if (to_radix) {
          // (rare) the row's columns cluster below what two levels resolve: the LSD radix kernel sorts it
          if (tid == 0) fb_rows[atomicAdd(fb_count, 1u)] = (unsigned)rid_a;
          gather_rest(4);
          load_cols();
        } else {
          gather_pt(4);
          // placement: slot = bucket start + arrival; the word orders the bucket by (low column bits, position)
          for_slots([&](int q, int j, int p, bool live) RQ_INLINE {
            const unsigned start = s_cnt[(int)bk[q][j] - 1];
            s_w[start + ar[q][j]] = ((((unsigned)ka[q][j] - mn) & lowmask) << ib) | (unsigned)p;
          });
          gather_pt(5);
          rq_barrier();  // B5
          gather_rest(6);
          load_cols();
          // rank among the words of the own bucket: four unrolled, the next four only in waves that hold a bucket of more
          // than four, a loop for what level 1 left fuller than eight (more than BK_MAX went to the radix list)
          for_slots([&](int q, int j, int p, bool live) RQ_INLINE {
            const unsigned b = bk[q][j];
            const unsigned s0 = s_cnt[(int)b - 1], cntb = live ? s_cnt[b] - s0 : 0u;
            const unsigned me = ((((unsigned)ka[q][j] - mn) & lowmask) << ib) | (unsigned)p;
            const unsigned w0 = s_w[s0], w1 = s_w[s0 + 1], w2 = s_w[s0 + 2], w3 = s_w[s0 + 3];
            unsigned r = (w0 < me) + ((w1 < me) & (cntb > 1u)) + ((w2 < me) & (cntb > 2u)) + ((w3 < me) & (cntb > 3u));
            if (__any(cntb > 4u)) {
              const unsigned w4 = s_w[s0 + 4], w5 = s_w[s0 + 5], w6 = s_w[s0 + 6], w7 = s_w[s0 + 7];
              r += ((w4 < me) & (cntb > 4u)) + ((w5 < me) & (cntb > 5u)) + ((w6 < me) & (cntb > 6u)) + ((w7 < me) & (cntb > 7u));
              if (__any(cntb > (unsigned)RQ_NEIGHBOURS)) {  // (a bucket level 1 left fuller than that: e.g. the end of a run of consecutive columns)
                for (unsigned t = RQ_NEIGHBOURS; t < cntb; t++) r += s_w[s0 + t] < me;
              }
            }
            const unsigned fin = s0 + r;
            if (VB == 4) {
              s_pair[fin] = make_uint2((unsigned)ka[q][j], (unsigned)va[q][j]);
            } else {
              s_key[fin] = (unsigned)ka[q][j];
              if (VB == 8) s_v8[fin] = va[q][j];
            }
          });
          out_mode = 1;
        }
        RQ_WAIT_LOADS();
      }
    } else {
      gather_rest(0);
      load_cols();
      RQ_WAIT_LOADS();
    }

    // rotate the row-level pipeline registers (the first reads of this step's loads: every path above ends in
    // RQ_WAIT_LOADS, behind its sort and in front of its stores)
    e0_a = e0_b, len_a = len_b, src_a = src_b, rid_a = rid_b;
    e0_b = e0_c, len_b = len_c, src_b = src_c, rid_b = rid_c;
    e0_c = r0_d, len_c = valid_d ? r1_d - r0_d : -1, src_c = src_d, rid_c = rid_d;
    rid_d = rid_e, valid_d = valid_e;

    if (out_mode) {  // stream the sorted row out of LDS
      rq_barrier();  // B6
      bool dup = false;
      store_row([&](int q, unsigned (&c)[4], V (&v)[4]) RQ_INLINE {
        const int p0 = q * 4 * T + 4 * tid;
        if (VB == 4) {
          const sbx_u4 x = *(const sbx_u4 *)&s_pair[p0], y = *(const sbx_u4 *)&s_pair[p0 + 2];
          c[0] = x.x, c[1] = x.z, c[2] = y.x, c[3] = y.z;
          v[0] = (V)x.y, v[1] = (V)x.w, v[2] = (V)y.y, v[3] = (V)y.w;
        } else {
          const sbx_u4 x = *(const sbx_u4 *)&s_key[p0];
          c[0] = x.x, c[1] = x.y, c[2] = x.z, c[3] = x.w;
          if (VB == 8) {
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = (V)s_v8[p0 + j];
          }
        }
        // duplicate columns (csr.cc:143-156 orders them by value afterwards, k_fix_dup_runs)
        const unsigned prev = VB == 4 ? s_pair[p0 > 0 ? p0 - 1 : 0].x : s_key[p0 > 0 ? p0 - 1 : 0];
        if (p0 + 4 <= len) {
          dup |= (c[1] == c[0]) | (c[2] == c[1]) | (c[3] == c[2]) | ((c[0] == prev) & (p0 > 0));
        } else {
#pragma unroll
          for (int j = 0; j < 4; j++) dup |= (p0 + j < len) & (c[j] == (j ? c[j - 1] : prev)) & (p0 + j > 0);
        }
      });
      if (__any(dup) && lane == 0) st->any_dup = 1;
    }
  };

  rq_barrier();
  for (int it = 0; it < my_rows + 4; it += 2) {
    step(it, k0, k1);
    step(it + 1, k1, k0);
  }
}
