// sbx_permute_rows5.h — rows of (PT_LMAX, 8192] entries of a column-relabelling permute: one workgroup per row.
// Included by sbx_permute.hip inside its anonymous namespace (permute/permute_order_two.cc:63-74 + format/csr.cc:123-156).
//
// What round 3's measurements said about this stage (tools/permute_v3_probe.py, -DR5_STAMPS builds,
// tools/gather_replay.hip, tools/gather_policy.hip):
//   * the chip relabels the bench matrix's column stream at 184 - 220 G gathers/s whatever the cache policy of the table
//     load and however many gathers a lane keeps in flight beyond four — IF every wave has gathers in flight all the
//     time (8 K per CU in the replay); the row kernels of rounds 2 and 3 had ~1 K in flight per CU (a wave gathers,
//     waits, then sorts) and relabelled at 90 G/s, and every other vector-memory instruction of theirs queued behind
//     those gathers (8 000 cycles from the issue of a row's loads to their use, thousands to ISSUE a store);
//   * occupancy (8 waves per SIMD, k_rows3) and prefetching the next row's columns / values (k_rows4) each left the
//     stage where round 2 had it: 0.8 ms for 70 M entries.
// So this kernel keeps the memory pipeline full and never waits for it in the middle of a row.  Everything that comes
// from memory travels by LDS-DMA (`global_load_lds_*`: no destination registers, in flight whatever the wave does),
// two rows ahead of the sort:
//     iteration i:   wait (everything issued an iteration ago)        row i + 1's raw columns / values, row i's keys
//                    GATHER row i + 1: raw columns LDS -> lanes, one `global_load_lds_dword` per 64 entries with the
//                        lane's table address as source: the relabelled columns land in LDS position by position
//                    store row i - 1 (sorted, from LDS through registers, 16 bytes per lane)
//                    RAW row i + 2: 1 KB LDS-DMA pieces of its columns and values
//                    SORT row i in LDS (bucket rank: returning count atomics = arrival index, plain
//                        placement, R5_UNROLL unconditional mate compares, 0xFFFFFFFF = "duplicate columns")
// Row-level scalars come by scalar loads into scalar registers; no ordinary vector load is left in the loop, so the
// compiler inserts no vector-memory wait of its own.  LDS per slot, 4-byte values: 8 - 11 B counters / placed words,
// 2 key buffers, 1 raw-column buffer, 3 value buffers = 32 - 35 B (16 - 19 waves per CU); DEEP = false (the largest
// classes) keeps one buffer of each and runs the stages of a row one after the other.
#pragma once

constexpr int R5_UNROLL = 6;   // bucket mates compared without a loop
constexpr int R5_PAD = 8;      // words behind both counter regions: the end-of-row word of the scans and the
                               // 0xFFFFFFFF words behind the placed ones (>= R5_UNROLL)
constexpr int R5_MISC = 32;    // words: [0..15] wave totals of the scan, [16] min, [17] max, [18] flags, [19] dup listed
constexpr unsigned R5_F_UNSORTED = 1u, R5_F_REFINE = 2u, R5_F_FALLBACK = 4u;


// diagnostic build only (-DR5_STAMPS): wall-clock cycles between phases of a row, summed over the rows by thread 0
#ifdef R5_STAMPS
__device__ unsigned long long g_r5_stamps[32];
#define R5_STAMP(i)                                                                 \
  do {                                                                              \
    if (tid == 0) {                                                                 \
      unsigned long long t_;                                                        \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");    \
      s_st[i] += t_ - t_prev;                                                       \
      t_prev = t_;                                                                  \
    }                                                                               \
  } while (0)
#else
#define R5_STAMP(i)
#endif

#ifndef R5_STREAM_AUX
#define R5_STREAM_AUX 2  // cache policy of the streamed (read-once) rows: 2 = nt, 0 = default
#endif
template <int AUX = 0>
__device__ __forceinline__ void r5_dma16(const char *g, unsigned *lds_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                   (__attribute__((address_space(3))) void *)lds_base, 16, 0, AUX);
}
template <int AUX = 0>
__device__ __forceinline__ void r5_dma4(const char *g, unsigned *lds_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                   (__attribute__((address_space(3))) void *)lds_base, 4, 0, AUX);
}

// this wave's share of a row's bytes: pieces of 1 KB (whole 16-byte units only), then wave 0 brings the last < 16
// bytes in dwords.  `row` = the row's first byte, `rb` = its length in bytes, `lds_row` = where byte 0 goes,
// `wave_bytes` = bytes a wave owns (64 lanes x ITEMS elements).
template <int PIECES>
__device__ __forceinline__ void r5_dma_row(const char *row, unsigned rb, unsigned *lds_row, int w, int lane,
                                           unsigned wave_bytes) {
  const unsigned wb = (unsigned)w * wave_bytes;
#pragma unroll
  for (int j = 0; j < PIECES; j++) {
    const unsigned off = wb + (unsigned)j * 1024u + (unsigned)lane * 16u;
    if (off + 16u <= rb) r5_dma16<R5_STREAM_AUX>(row + off, lds_row + ((wb + (unsigned)j * 1024u) >> 2));
  }
  if (w == 0 && (rb & 15u)) {
    const unsigned t0 = rb & ~15u;
    if ((unsigned)lane * 4u < (rb & 15u)) r5_dma4<R5_STREAM_AUX>(row + t0 + (unsigned)lane * 4u, lds_row + (t0 >> 2));
  }
}

struct __attribute__((packed, aligned(4))) r5_i4u { int x, y, z, w; };            // 16 bytes at a 4-byte aligned address
struct __attribute__((packed, aligned(4))) r5_l2u { unsigned long long x, y; };


// in-place exclusive scan of cnt[0, n) by the whole workgroup (thread t owns CPT consecutive words, n <= CPT *
// blockDim.x); cnt[n] = `total` (the caller knows it: the row's length).  Returns the largest count this thread saw.
// One barrier inside (workgroups of more than one wave); the caller adds the one that publishes the result.
template <int CPT>
__device__ __forceinline__ unsigned r5_scan(unsigned *cnt, int n, unsigned total, unsigned *s_tot, bool multi) {
  static_assert(CPT % 4 == 0, "16-byte LDS accesses");
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int base = tid * CPT;
  unsigned c[CPT], sum = 0, mxc = 0;
  const bool mine = base < n;
  if (mine) {
#pragma unroll
    for (int i = 0; i < CPT; i += 4) {
      const uint4 q = *(const uint4 *)(cnt + base + i);
      c[i] = q.x, c[i + 1] = q.y, c[i + 2] = q.z, c[i + 3] = q.w;
    }
#pragma unroll
    for (int i = 0; i < CPT; i++) {
      mxc = c[i] > mxc ? c[i] : mxc;
      sum += c[i];
    }
  }
  const unsigned inc = sbx_wave_inclusive_sum(sum);
  unsigned ex = inc - sum;
  if (multi) {
    if (lane == 63) s_tot[w] = inc;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const unsigned t = sbx_wave_inclusive_sum(lane < 16 ? s_tot[lane] : 0u);  // (words of absent waves are zero)
    if (w > 0) ex += (unsigned)__builtin_amdgcn_readlane((int)t, w - 1);
  }
  if (mine) {
#pragma unroll
    for (int i = 0; i < CPT; i += 4) {
      uint4 q;
      q.x = ex, ex += c[i];
      q.y = ex, ex += c[i + 1];
      q.z = ex, ex += c[i + 2];
      q.w = ex, ex += c[i + 3];
      *(uint4 *)(cnt + base + i) = q;
    }
  }
  if (tid == 0) cnt[n] = total;  // (a word this scan's last thread may have covered: written behind its stores)
  return mxc;
}


__host__ __device__ constexpr size_t r5_lds_bytes(int threads, int items, int vb, int nbw, bool deep) {
  return sizeof(unsigned) * (size_t)(2 * (nbw + R5_PAD) + R5_MISC + 16) +
         (size_t)((deep ? 3 : 1) * 4 + (deep ? 3 : 1) * vb) * (size_t)(threads * items);
}

template <int VB, int ITEMS, int MINW, bool DEEP>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(MINW, 8))) void k_rows5(
    const int2 *__restrict__ rec, const int *col_in, const char *val_in, const int *__restrict__ col_order,
    const int *__restrict__ rpo, const int *__restrict__ list, int n_rows, int *col_out, char *val_out,
    PermState *__restrict__ st, int force, unsigned *__restrict__ fb_rows, unsigned *__restrict__ fb_count,
    const unsigned *__restrict__ n_rows_dev, int nbw) {
  typedef typename ValT<VB>::type V;
  constexpr bool HASV = VB != 0;
  constexpr int VW = VB / 4;
  constexpr int NK = DEEP ? 2 : 1, NV = DEEP ? 3 : 1;
  static_assert(ITEMS == 4 || ITEMS == 8, "16-byte units");
  extern __shared__ __attribute__((aligned(16))) unsigned s_dyn[];
  const int T = (int)blockDim.x, S = ITEMS * T;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool multi = T > 64;
  unsigned *const s_A = s_dyn;                      // level-0 counters / (second level) placed words
  unsigned *const s_B = s_A + nbw + R5_PAD;         // placed words / (second level) counters
  unsigned *const s_misc = s_B + nbw + R5_PAD;      // R5_MISC words, then one edge word per wave
  unsigned *const s_edge = s_misc + R5_MISC;
  unsigned *const s_key = s_edge + 16;              // [NK][S] relabelled columns (gather DMA), then the sorted row
  unsigned *const s_raw = s_key + NK * S;           // [S] raw columns (DEEP; otherwise they pass through s_key)
  unsigned *const s_val = DEEP ? s_raw + S : s_raw; // [NV][S] values (VW words each): raw, then sorted
#define R5_BARRIER()                                                              \
  do {                                                                            \
    if (multi) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");    \
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       \
  } while (0)
#define R5_VMWAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#ifndef R5_NT_STORES
#define R5_NT_STORES 1
#endif
#if R5_NT_STORES
#define R5_STORE16(p, q)                                                                                         \
  asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"((const void *)(p)), "v"(*(const __attribute__((ext_vector_type(4))) int *)&(q)) : "memory")
#else
#define R5_STORE16(p, q) (*(r5_i4u *)(p) = (q))
#endif
  if (n_rows_dev) n_rows = (int)*n_rows_dev;  // (segments of long rows: the host does not know how many there are)
  const int G = (int)gridDim.x;
  const int my_rows = ((int)blockIdx.x < n_rows) ? (n_rows - 1 - (int)blockIdx.x) / G + 1 : 0;
  if (my_rows == 0) return;

  for (int i = 4 * tid; i < 2 * (nbw + R5_PAD); i += 4 * T) *(uint4 *)(s_A + i) = make_uint4(0, 0, 0, 0);
  if (tid < R5_MISC) s_misc[tid] = tid == 16 ? 0xFFFFFFFFu : 0u;

  const int p0 = w * (64 * ITEMS) + ITEMS * lane;  // this lane's first entry of a row (sort / store layout)
  struct Row {
    int rid, e0, len, src;
  };
  auto load_row = [&](int it) {  // scalar loads (uniform addresses); rows past the workgroup's last one are empty
    Row r;
    r.rid = 0, r.e0 = 0, r.len = 0, r.src = 0;
    if (it < my_rows) {
      r.rid = list[(int64_t)blockIdx.x + (int64_t)it * G];
      r.e0 = rpo[r.rid];
      r.len = rpo[r.rid + 1] - r.e0;
      r.src = rec[r.rid].y;
    }
    return r;
  };
  // RAW: a row's columns and values, global -> LDS in 1 KB pieces
  auto stage_raw = [&](const Row &r, unsigned *cdst, unsigned *vdst) {
    r5_dma_row<ITEMS / 4>((const char *)(col_in + r.src), (unsigned)r.len * 4u, cdst, w, lane, 256u * ITEMS);
    if (HASV) r5_dma_row<ITEMS * VB / 16>(val_in + (int64_t)r.src * VB, (unsigned)r.len * VB, vdst, w, lane, 64u * ITEMS * VB);
  };
  // GATHER: raw columns (LDS, one per lane and 64-entry piece) -> table addresses -> relabelled columns land in kdst
  auto stage_gather = [&](const Row &r, const unsigned *csrc, unsigned *kdst) {
    const int wbase = w * (64 * ITEMS);
    if (col_order && !(force & 4)) {  // (bit 2: timing ablation without the relabel gathers)
      unsigned c[ITEMS];
#pragma unroll
      for (int j = 0; j < ITEMS; j++) c[j] = csrc[wbase + j * 64 + lane];
#pragma unroll
      for (int j = 0; j < ITEMS; j++)
        if (wbase + j * 64 + lane < r.len)
          r5_dma4((const char *)col_order + (size_t)c[j] * 4u, kdst + wbase + j * 64);
    } else if (csrc != kdst) {
#pragma unroll
      for (int k = 0; k < ITEMS; k += 4) *(uint4 *)(kdst + p0 + k) = *(const uint4 *)(csrc + p0 + k);
    }
  };

  Row cur = load_row(0), nxt = load_row(1), nn = load_row(2), prev;
  prev.rid = 0, prev.e0 = 0, prev.len = -1, prev.src = 0;
  bool prev_in_lds = false;
  R5_BARRIER();
  if (DEEP) {
    // prologue: row 0's raw data, its gathers, row 1's raw data
    stage_raw(cur, s_raw, s_val);
    R5_VMWAIT();
    R5_BARRIER();
    stage_gather(cur, s_raw, s_key);
    R5_BARRIER();  // (every wave has read row 0's raw columns)
    stage_raw(nxt, s_raw, s_val + 1 * S * VW);
  }

  for (int it = 0; it <= my_rows; it++) {
    const int kb = DEEP ? (it & 1) : 0, vb_i = DEEP ? it % 3 : 0;
    unsigned *const kbuf = s_key + kb * S;
    unsigned *const vbuf = s_val + vb_i * S * VW;
    const int len = it < my_rows ? cur.len : -1;
    // ---- the previous row's sorted entries: LDS -> registers
    int oc[ITEMS];
    V ov[HASV ? ITEMS : 1];
    const bool have_prev = prev.len >= 0 && prev_in_lds;
    if (have_prev) {
      const unsigned *pc = s_key + (DEEP ? ((it - 1) & 1) : 0) * S + p0;
#pragma unroll
      for (int k = 0; k < ITEMS; k += 4) {
        const uint4 q = *(const uint4 *)(pc + k);
        oc[k] = (int)q.x, oc[k + 1] = (int)q.y, oc[k + 2] = (int)q.z, oc[k + 3] = (int)q.w;
      }
      if (HASV) {
        const V *pv = (const V *)(s_val + (DEEP ? (it + 2) % 3 : 0) * S * VW) + p0;
#pragma unroll
        for (int k = 0; k < ITEMS; k++) ov[k] = pv[k];
      }
    }
    if (DEEP) {
      R5_VMWAIT();   // row it + 1's raw data and row it's relabelled columns have landed (this wave's pieces)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // B0: ... and every other wave's; every wave has
                                                                      // read the previous row out of its buffers
      if (it + 1 < my_rows) stage_gather(nxt, s_raw, s_key + (kb ^ 1) * S);
    } else {
      R5_BARRIER();  // every wave has read the previous row out of the buffers
      if (len >= 0) {
        stage_raw(cur, kbuf, vbuf);
        R5_VMWAIT();
        R5_BARRIER();
        if (col_order && !(force & 4)) {
          unsigned c[ITEMS];
          const int wbase = w * (64 * ITEMS);
#pragma unroll
          for (int j = 0; j < ITEMS; j++) c[j] = kbuf[wbase + j * 64 + lane];
          R5_BARRIER();  // the raw columns are in registers: the relabelled ones may land on them
#pragma unroll
          for (int j = 0; j < ITEMS; j++)
            if (wbase + j * 64 + lane < len) r5_dma4((const char *)col_order + (size_t)c[j] * 4u, kbuf + wbase + j * 64);
        }
      }
    }
    // ---- the previous row's stores (behind the gathers in the memory pipeline)
    if (have_prev) {
      bool dup = false;
#pragma unroll
      for (int k = 0; k < ITEMS; k++) dup |= p0 + k < prev.len && oc[k] == -1;  // a slot nobody wrote: two entries ranked equal
#pragma unroll
      for (int k = 0; k < ITEMS; k += 4) {
        int *o = col_out + (unsigned)prev.e0 + (unsigned)(p0 + k);
        if (p0 + k + 4 <= prev.len) {
          r5_i4u q;
          q.x = oc[k], q.y = oc[k + 1], q.z = oc[k + 2], q.w = oc[k + 3];
          R5_STORE16(o, q);
        } else {
#pragma unroll
          for (int j = 0; j < 4; j++)
            if (p0 + k + j < prev.len) o[j] = oc[k + j];
        }
      }
      if (HASV) {
        V *o = (V *)val_out + (unsigned)prev.e0 + (unsigned)p0;
#pragma unroll
        for (int k = 0; k < ITEMS; k += 4) {
          if (p0 + k + 4 <= prev.len) {
            if constexpr (VB == 4) {
              r5_i4u q;
              q.x = (int)ov[k], q.y = (int)ov[k + 1], q.z = (int)ov[k + 2], q.w = (int)ov[k + 3];
              R5_STORE16(o + k, q);
            } else {
              r5_l2u q0, q1;
              q0.x = ov[k], q0.y = ov[k + 1], q1.x = ov[k + 2], q1.y = ov[k + 3];
              *(r5_l2u *)(o + k) = q0;
              *(r5_l2u *)(o + k + 2) = q1;
            }
          } else {
#pragma unroll
            for (int j = 0; j < 4; j++)
              if (p0 + k + j < prev.len) o[k + j] = ov[k + j];
          }
        }
      }
      if (__any(dup) && lane == 0 && atomicOr(&s_misc[19], 1u) == 0u)  // duplicate columns: the stable kernel redoes the row
        fb_rows[atomicAdd(fb_count, 1u)] = (unsigned)prev.rid;
    }
    prev_in_lds = false;
    if (len < 0) break;
    const Row far = load_row(it + 3);
    if (!DEEP) {
      R5_VMWAIT();   // the relabelled columns have landed
      R5_BARRIER();
    }

    // ---- SORT row `it`: its relabelled columns sit in kbuf, its values in vbuf
    int key[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; k += 4) {
      const uint4 q = *(const uint4 *)(kbuf + p0 + k);
      key[k] = (int)q.x, key[k + 1] = (int)q.y, key[k + 2] = (int)q.z, key[k + 3] = (int)q.w;
    }
    bool uns = false, dupc = false;
    unsigned mn = 0x7FFFFFFFu, mx = 0;
    {
      int prevk = sbx_wave_shift_up1(key[ITEMS - 1], (int)0x80000000);
#pragma unroll
      for (int k = 0; k < ITEMS; k++) {
        const bool valid = p0 + k < len;
        if (!valid) key[k] = 0x7FFFFFFF;
        uns |= valid && key[k] < prevk;
        dupc |= valid && key[k] == prevk;
        mn = (unsigned)key[k] < mn ? (unsigned)key[k] : mn;
        mx = valid && (unsigned)key[k] > mx ? (unsigned)key[k] : mx;
        prevk = key[k];
      }
      if (lane == 63) s_edge[w] = (unsigned)key[ITEMS - 1];
    }
    mn = sbx_wave_min(mn);
    mx = sbx_wave_max(mx);
    if (lane == 0) {
      atomicMin(&s_misc[16], mn);
      atomicMax(&s_misc[17], mx);
    }
    if (__any(uns) && lane == 0) atomicOr(&s_misc[18], R5_F_UNSORTED);
    R5_BARRIER();  // 1: edges, min / max; (DEEP) every wave has read row it + 1's raw columns
    if (DEEP && it + 2 < my_rows) stage_raw(nn, s_raw, s_val + ((it + 2) % 3) * S * VW);  // in flight for the whole sort
    if (lane == 0 && w > 0 && p0 < len) {  // the entry in front of this wave's first one
      const int pk = (int)s_edge[w - 1];
      if (key[0] < pk) atomicOr(&s_misc[18], R5_F_UNSORTED);
      dupc |= key[0] == pk;
    }
    const uint2 mm = *(const uint2 *)&s_misc[16];
    mn = mm.x, mx = mm.y;
    const int rbits = bits_u32(mx - mn);
    const int ib = bits_u32((unsigned)len - 1u);
    const int shift = rbits > ib ? rbits - ib : 0;
    const int n_scan = (int)((mx - mn) >> shift) + 1;  // buckets in use
    unsigned b[ITEMS], arr[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      b[k] = ((unsigned)key[k] - mn) >> shift;
      arr[k] = 0;
      if (p0 + k < len) arr[k] = atomicAdd(&s_A[b[k]], 1u);
    }
    // (meanwhile: the second level's counters get their fill)
    for (int i = 4 * tid; i < nbw + R5_PAD; i += 4 * T) *(uint4 *)(s_B + i) = make_uint4(0, 0, 0, 0);
    R5_BARRIER();  // 2: counts, flags
    unsigned flags = s_misc[18];
    if (force & 8) flags |= R5_F_UNSORTED;   // (test hook: sort rows that are in order too)
    if (force & 2) flags &= ~R5_F_UNSORTED;  // (bit 1: timing ablation, rows stream out unsorted)
    bool in_lds = false;
    if (!(flags & R5_F_UNSORTED)) {
      // an ordered row needs no sort (and its equal columns keep their order): it is stored as it stands
      if (__any(dupc) && lane == 0) st->any_dup = 1;
      in_lds = true;
      R5_BARRIER();  // (every wave has read the flags)
    } else {
      if (tid == 0) st->any_unsorted = 1;
      unsigned *cnt = s_A, *words = s_B;
      {
        const unsigned mxc = (ITEMS == 4 && nbw <= 4 * T) ? r5_scan<4>(s_A, n_scan, (unsigned)len, s_misc, multi)
                                                          : r5_scan<8>(s_A, n_scan, (unsigned)len, s_misc, multi);
        if (__any(mxc > (unsigned)BK_REFINE) && lane == 0) atomicOr(&s_misc[18], R5_F_REFINE);
        if ((force & 1) && tid == 0) atomicOr(&s_misc[18], R5_F_FALLBACK);
      }
      R5_BARRIER();  // 4: bucket starts, refine flag
      flags = s_misc[18];
      if (flags & R5_F_REFINE) {
        // second level (clustered columns): every bucket is split into as many sub-buckets as it holds entries, by
        // interpolation on the 16 leading bits of the column's offset inside the bucket
        const int sh16 = shift > 16 ? shift - 16 : 0, shl = shift > 16 ? 16 : shift;
        const unsigned lowmask = (1u << shift) - 1u;
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
          const unsigned bb = p0 + k < len ? b[k] : 0u;
          const unsigned start = s_A[bb], cntb = s_A[bb + 1] - start;
          const unsigned low = (((unsigned)key[k] - mn) & lowmask) >> sh16;
          b[k] = start + ((low * cntb) >> shl);  // < start + cntb
          if (p0 + k < len) arr[k] = atomicAdd(&s_B[b[k]], 1u);
        }
        R5_BARRIER();  // level-1 counts
        const unsigned mxc = ITEMS == 4 ? r5_scan<4>(s_B, len, (unsigned)len, s_misc, multi)
                                        : r5_scan<8>(s_B, len, (unsigned)len, s_misc, multi);
        if (__any(mxc > (unsigned)BK_MAX) && lane == 0) atomicOr(&s_misc[18], R5_F_FALLBACK);
        R5_BARRIER();
        flags = s_misc[18];
        cnt = s_B, words = s_A;
      }
      if (flags & R5_F_FALLBACK) {  // (rare) the row goes on the list of k_permute_rows_radix
        if (tid == 0) fb_rows[atomicAdd(fb_count, 1u)] = (unsigned)cur.rid;
        R5_BARRIER();  // (every wave has read the flags)
      } else {
        V v[HASV ? ITEMS : 1];
        if (HASV) {  // the raw values leave their buffer before the sorted ones move in
          const V *pv = (const V *)vbuf + p0;
#pragma unroll
          for (int k = 0; k < ITEMS; k++) v[k] = pv[k];
        }
#pragma unroll
        for (int k = 0; k < ITEMS; k++)
          if (p0 + k < len) words[cnt[b[k]] + arr[k]] = (unsigned)key[k] - mn;
        if (tid < R5_PAD) words[len + tid] = 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < ITEMS; k += 4) *(uint4 *)(kbuf + p0 + k) = make_uint4(~0u, ~0u, ~0u, ~0u);
        R5_BARRIER();  // 5: placed words; the row's columns / values are in registers
        unsigned fin[ITEMS];
        bool big = false;
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
          const unsigned bb = p0 + k < len ? b[k] : 0u;
          const unsigned start = cnt[bb], end = cnt[bb + 1];
          const unsigned rel = (unsigned)key[k] - mn;
          unsigned r = 0;
#pragma unroll
          for (int j = 0; j < R5_UNROLL; j++) r += words[start + j] < rel;
          big |= end - start > (unsigned)R5_UNROLL;
          fin[k] = start + r;
        }
        if (__any(big)) {
#pragma unroll
          for (int k = 0; k < ITEMS; k++) {
            if (p0 + k < len) {
              const unsigned end = cnt[b[k] + 1], rel = (unsigned)key[k] - mn;
              for (unsigned j = cnt[b[k]] + R5_UNROLL; j < end; j++) fin[k] += words[j] < rel;
            }
          }
        }
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
          if (p0 + k < len) {
            kbuf[fin[k]] = (unsigned)key[k];
            if (HASV) ((V *)vbuf)[fin[k]] = v[k];
          }
        }
        in_lds = true;
      }
    }
    if (tid == 0) {
      s_misc[16] = 0xFFFFFFFFu, s_misc[17] = 0, s_misc[18] = 0, s_misc[19] = 0;
    }
    R5_BARRIER();  // 6: the row sits sorted in its buffers
    // the level-0 counters of the next row
    for (int i = 4 * tid; i < nbw + R5_PAD; i += 4 * T) *(uint4 *)(s_A + i) = make_uint4(0, 0, 0, 0);
    prev = cur, prev_in_lds = in_lds;
    cur = nxt, nxt = nn, nn = far;
  }
#undef R5_STORE16
#undef R5_VMWAIT
#undef R5_BARRIER
}
