// sbx_rcm.hip — RCMReorder::GetReorderCSR (reorder/rcm_reorder.cc:83-166) and its
// pseudo-peripheral search (:22-81) as a bit-exact level-synchronous GPU algorithm.
//
// The reference is a serial queue algorithm; its output is reproduced exactly from
// these order-free statements (validated against the reference):
//   * components are laid out in order of their smallest vertex id, each occupying a
//     contiguous block of positions (prefix sum of component sizes); isolated
//     vertices are singleton components (:111-116);
//   * a plain FIFO BFS visits level k+1 in the order (position of the first parent
//     in level k, vertex id) — adjacency lists are column-sorted;
//   * the Cuthill-McKee BFS visits level k+1 in the order (position of the first
//     parent, degree, vertex id) — the (degree,id) min-heap of :125-144;
//   * pseudo-peripheral: restart from the deepest vertex with the strictly smallest
//     degree, first in queue order on ties, until the eccentricity stops growing or
//     the BFS visits one vertex per level (:34-80);
//   * each component's visiting order is reversed (:146-153) and inverted (:158-160).
//
// GPU mapping (wave64):
//   components      ECL-CC style union-find (hook larger root under smaller, so the
//                   root IS the smallest vertex id), sizes by wave-aggregated atomics,
//                   bases by one device scan;
//   small comps     (<= RCM_SMALL vertices; <= RCM_MID when many) one lane per component runs the serial
//                   algorithm in its own slice of the output;
//   large comps     level-synchronous BFS: wave-per-frontier-vertex expansion with
//                   ballot-aggregated frontier append, atomicCAS claim + atomicMin of
//                   the parent position; hub vertices (> 256 neighbours) are split
//                   into 1024-neighbour chunks across workgroups; each level is put
//                   in reference order by sorting 64-bit keys (parent_pos<<32 | id or
//                   global (degree,id) rank): one-workgroup LDS bitonic sort for small
//                   levels, the device radix sort for large ones.
#include "sbx_device.h"
#include "sbx_internal.h"
#include "sbx_countsort.h"

#include <utility>
#include <algorithm>
#include <vector>
#include <functional>
#include <string>

namespace {

typedef int32_t I;  // vertex ids, offsets and every internal list: 32 bits (n < 2^31 - 1, nnz < 2^31 in either build)
// element type of the CALLER's arrays — row_ptr, col, the inverse permutation: this file is compiled twice, for 32-bit
// index arrays (sbx_rcm.hip) and, through sbx_rcm64.hip, for 64-bit ones, which are read and written as they are (no
// narrowed copies); every value is used as a 32-bit id or offset once it is in a register
#ifdef SBX_RCM_I64
typedef int64_t X;
#define SBX_RCM_IT SBX_I64
#define SBX_RCM_ENTRY sbx_rcm_reorder_x64
#else
typedef int32_t X;
#define SBX_RCM_IT SBX_I32
#define SBX_RCM_ENTRY sbx_rcm_reorder_x32
#endif
constexpr int RCM_SMALL = 64;        // components up to this size: one lane each
constexpr int RCM_MID = 2048;        // ... and up to this size too when there are RCM_MID_BATCH or more of them: a mesh
constexpr int RCM_MID_BATCH = 64;    // collection with 10^5 components would otherwise be ordered one by one from the host
constexpr int RCM_LIGHT = 256;       // neighbours expanded by the discovering wave itself
constexpr int RCM_CHUNK = 1024;      // hub neighbours per workgroup chunk
constexpr int RCM_LDS_SORT = 4096;   // levels up to this size are sorted by one workgroup
constexpr int RCM_REBUILD_BITS = 32768;  // levels from max(this, n/128) vertices on rebuild the visited bitmap from ppos
constexpr unsigned UNSEEN = 0xFFFFFFFFu;

// Device-resident scalars.  The five counters every expansion workgroup adds to sit in cache lines of their
// own: atomics on one line queue behind each other whichever word they target.
struct RcmDev {
  alignas(128) unsigned nf;                  // size of the frontier being built
  alignas(128) unsigned n_heavy;             // hub chunk descriptors queued for the chunked kernel
  unsigned hub_overflow;                     // some workgroup reserved descriptors outside its directory entry
  alignas(128) unsigned long long fedges;    // sum of degrees of the level being built (direction heuristic)
  alignas(128) unsigned long long edges;     // adjacency entries scanned (statistics)
  alignas(128) unsigned long long edges_bu;  // adjacency entries scanned by the bottom-up kernel
  alignas(128) unsigned n_small;             // small components listed
  unsigned n_large;             // large components listed
  unsigned n_mid;               // components of RCM_SMALL + 1 .. RCM_MID vertices listed
  unsigned n_cc_big;            // high-degree vertices queued by the CC hook kernel
  unsigned max_deg;
  unsigned root;                // current BFS root of the component being ordered
  unsigned first_vertex;        // smallest vertex id with a non-empty row (UNSEEN if none)
  unsigned long long best;      // (degree<<32 | position) minimum over the deepest level
  // hand-off state of the persistent small-level kernel
  unsigned sl_off, sl_fsize, sl_level, sl_total;
  unsigned sl_status;           // SL_DONE / SL_STOP_READY / SL_STOP_EXPANDED
  unsigned unsym;               // the pattern is not structurally symmetric (a BFS cannot reach its component)
  unsigned long long sl_edges;  // degree sum of the levels the kernel ordered
  unsigned n_components;        // statistics: union-find roots (isolated vertices included) / vertices with an empty row
  unsigned n_empty_rows;
  unsigned n_nonempty;          // vertices with a non-empty row: what the degree-rank sort sorts
  unsigned n_top;               // ... of them with 255 entries and more (the counting pass's last bucket, sbx_degree_ranks)
  // unordered sweeps: the deepest level's smallest degree, how many vertices have it, the smallest id among them
  unsigned tie_deg, tie_count, tie_min_id;
  unsigned desc[3];             // tie-break walk root -> w_1 -> ... : w_k in desc[k % 3]
  unsigned cone_begin, cone_end;  // tie-break: the marked vertices of the level being expanded are list[begin, end)
  unsigned cone_k, cone_status;   // k_ubfs_cone_run: the level it stopped at (its list is too long for it) / UR_DONE
  unsigned bar;                 // grid barrier of k_ubfs_descend_all
  unsigned gb_abort;            // a grid barrier gave up waiting (see gb_wait): the host redoes the sweep the safe way
  unsigned root_next;           // what the tie-break found; becomes `root` when the next sweep starts (root_pending) —
  unsigned root_pending;        //   unless a grid barrier gave up on the way: k_gb_reset drops it and the old root stands
  unsigned tie_done;            // the last level held one candidate: the cone / descend kernels have nothing to do
  unsigned gb_spins;            // how many polls a barrier waits (k_ubfs_start sets it; SBX_DEBUG_GB_SPINS for tests)
  unsigned start_done;          // workgroups of a sweep's start kernel that have cleared their share (start_clear_elect)
  // k_ubfs_small_run: its grid barrier (arrivals, exits), the state it hands back and the frontier's degree sum
  unsigned ur_bar, ur_exit;
  unsigned ur_off, ur_size, ur_level, ur_total, ur_status;
  unsigned long long ur_fe, ur_esum;
  unsigned ur_nf[3], ur_nh[3];          // per level (slot = level % 3): vertices appended, hubs queued
  unsigned long long ur_deg[3];         // ... and the degree sum of the level
  // unordered sweeps, levels chained on the device (run_ubfs): the frontier that is current for the next kernel of the
  // chain, the unvisited edges, what that kernel is to be (UC_*), and how often the chain swapped the frontier bitmaps
  unsigned uc_off, uc_size, uc_level, uc_total, uc_mode, uc_flips, uc_small_ran, uc_done;
  // The sweep behind a tie walk is enqueued before the host has seen how the walk ended: its first kernels (the start
  // kernel and the small-level kernel of either kind) leave at once while this is set and tie_done is not — nothing of the
  // finished sweep is then touched and the persistent cone kernels can still take the tie-break over (ubfs_pick_root_slow).
  // Set by k_ubfs_ties_small, cleared by a start kernel that goes ahead.
  unsigned spec_guard;
  unsigned tie_walk_exit;  // where tie_walk left (0: it named the root; SBX_DEBUG_TIE_WALK=1 prints it)
  unsigned tie_walk_arg;
  unsigned uc_head;  // a chain was begun on the device behind a sweep's head small-level run (k_ubfs_chain_from_small)
  unsigned long long uc_fe;
  long long uc_remaining;
  // unordered sweeps: size and degree sum of level l in slot l & 1 (the collect kernel of level l clears the other one)
  alignas(128) unsigned unf[2];
  alignas(128) unsigned long long ufedges[2];
};

// release / acquire fences of the grid barriers and elections: only in the checking build (see gb_wait)
#ifdef SBX_GB_FENCED
#define SBX_GB_RELEASE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent")
#define SBX_GB_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
#else
#define SBX_GB_RELEASE() ((void)0)
#define SBX_GB_ACQUIRE() ((void)0)
#endif


// ------------------------------------------------------------------ degree rank
// Only vertices with a non-empty row ever meet a BFS, so only they get a (degree, id) rank: on a power-law graph that
// halves the sort.  DEG_UNITS waves own contiguous vertex ranges; the first kernel counts each unit's non-empty rows
// (and finds the largest degree and the first non-empty vertex), the second writes (degree, id) compacted in id
// order — a unit's base is the sum of the counts before it — which is what the stable sort needs.
constexpr int DEG_UNITS = 4096;
__device__ __forceinline__ int64_t deg_unit_len(int64_t n) { return ((n + DEG_UNITS - 1) / DEG_UNITS + 63) / 64 * 64; }

// (The call's three n-sized fills ride along — component sizes 0, distances and parent positions UNSEEN: as
// hipMemsetAsync calls they cost the host ~25 us of enqueueing at the head of the call, with the GPU idle.)
__global__ __launch_bounds__(256) void k_deg_count(const X *__restrict__ rp, int64_t n, unsigned *__restrict__ ucnt,
                                                   I *__restrict__ csize, unsigned *__restrict__ dist,
                                                   unsigned *__restrict__ ppos) {
  const int unit = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = sbx_lane();
  const int64_t len = deg_unit_len(n), v0 = (int64_t)unit * len;
  unsigned mx = 0, fv = UNSEEN, cnt = 0, top = 0;
  for (int64_t v = v0 + lane; v < v0 + len && v < n; v += 64) {
    const unsigned d = (unsigned)(rp[v + 1] - rp[v]);
    mx = d > mx ? d : mx;
    top += d >= 255u;
    if (d) {
      fv = (unsigned)v < fv ? (unsigned)v : fv;
      cnt++;
    }
    csize[v] = 0;
    dist[v] = UNSEEN;
    ppos[v] = UNSEEN;
  }
  if (unit == 0 && lane == 0) csize[n] = 0;
  cnt = sbx_wave_sum(cnt);
  top = sbx_wave_sum(top);
  mx = sbx_wave_max(mx);
  fv = sbx_wave_min(fv);
  // per-unit partials, reduced by k_deg_reduce: three adds per workgroup on one line of RcmDev — 3 K adds at ~7 ns
  // apiece whoever issues them — were 20 of this kernel's 31 us, at the head of every RCM call
  if (lane == 0 && unit < DEG_UNITS)
    ucnt[unit] = cnt, ucnt[DEG_UNITS + unit] = mx, ucnt[2 * DEG_UNITS + unit] = fv, ucnt[3 * DEG_UNITS + unit] = top;
}

__global__ __launch_bounds__(1024) void k_deg_reduce(const unsigned *__restrict__ ucnt, RcmDev *__restrict__ dv) {
  __shared__ unsigned s_mx[16], s_fv[16], s_cnt[16], s_top[16];
  for (unsigned i = threadIdx.x; i < sizeof(RcmDev) / sizeof(unsigned); i += 1024) ((unsigned *)dv)[i] = 0;  // the call's state
  unsigned mx = 0, fv = UNSEEN, cnt = 0, top = 0;
  for (int u = threadIdx.x; u < DEG_UNITS; u += 1024) {
    cnt += ucnt[u];
    top += ucnt[3 * DEG_UNITS + u];
    mx = ucnt[DEG_UNITS + u] > mx ? ucnt[DEG_UNITS + u] : mx;
    fv = ucnt[2 * DEG_UNITS + u] < fv ? ucnt[2 * DEG_UNITS + u] : fv;
  }
  cnt = sbx_wave_sum(cnt);
  top = sbx_wave_sum(top);
  mx = sbx_wave_max(mx);
  fv = sbx_wave_min(fv);
  if (sbx_lane() == 0) s_mx[threadIdx.x >> 6] = mx, s_fv[threadIdx.x >> 6] = fv, s_cnt[threadIdx.x >> 6] = cnt, s_top[threadIdx.x >> 6] = top;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 16; i++) {
      mx = s_mx[i] > mx ? s_mx[i] : mx;
      fv = s_fv[i] < fv ? s_fv[i] : fv;
      cnt += s_cnt[i];
      top += s_top[i];
    }
    dv->max_deg = mx;
    dv->first_vertex = fv;
    dv->root = fv == UNSEEN ? 0u : fv;  // the first sweep starts here without the host having seen it (sbx_rcm_reorder)
    dv->n_nonempty = cnt;
    dv->n_top = top;
  }
}

// One non-trivial component (the usual power-law input: a giant component and isolated vertices): every vertex with an
// empty row is a component of its own, placed by the count of components in front of it — the empty rows with a smaller
// id, plus the whole first component if its root v0 lies in front (format of k_classify / cbase: rcm_reorder.cc visits
// the components in the order of their smallest vertex).  Same unit walk as k_deg_count; replaces the union-find, its
// size scan and the classification when the first sweep already reached every non-empty row.
__global__ __launch_bounds__(256) void k_iso_positions(const X *__restrict__ rp, int64_t n, const unsigned *__restrict__ ucnt,
                                                       I v0, I comp_size, X *__restrict__ inv) {
  const int unit = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = sbx_lane();
  if (unit >= DEG_UNITS) return;
  unsigned base = 0;  // non-empty rows in front of the unit
  for (int j = lane; j < unit; j += 64) base += ucnt[j];
  base = sbx_wave_sum(base);
  const int64_t len = deg_unit_len(n), u0 = (int64_t)unit * len;
  for (int64_t vb = u0; vb < u0 + len && vb < n; vb += 64) {
    const int64_t v = vb + lane;
    const bool full = v < n && rp[v + 1] != rp[v];
    const uint64_t m = __ballot(full);
    if (v < n && !full) {
      const int64_t nonempty_before = (int64_t)base + __popcll(m & sbx_lanemask_lt());
      inv[v] = (I)(v - nonempty_before + (v > (int64_t)v0 ? (int64_t)comp_size : 0));
    }
    base += (unsigned)__popcll(m);
  }
}

// ------------------------------------------------------------------ connected components
__device__ __forceinline__ I cc_find(I *parent, I v) {
  I curr = parent[v];
  if (curr != v) {
    I prev = v, next;
    while (curr > (next = parent[curr])) {  // parent[x] <= x always: the chain descends
      parent[prev] = next;                  // path halving
      prev = curr;
      curr = next;
    }
  }
  return curr;
}

__device__ __forceinline__ void cc_hook(I *parent, I a, I b) {
  I ra = cc_find(parent, a), rb = cc_find(parent, b);
  bool again;
  do {
    again = false;
    if (ra != rb) {
      if (ra < rb) {
        const I got = atomicCAS(&parent[rb], rb, ra);
        if (got != rb) { rb = got; again = true; }
      } else {
        const I got = atomicCAS(&parent[ra], ra, rb);
        if (got != ra) { ra = got; again = true; }
      }
    }
  } while (again);
}

__global__ __launch_bounds__(256) void k_cc_init(const X *__restrict__ rp, const X *__restrict__ col,
                                                 I *__restrict__ parent, int64_t n,
                                                 const unsigned *__restrict__ cbits) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; v < n; v += stride) {
    I p = (I)v;
    // (the members of the component the first sweep covered — nearly every non-empty row of a power-law graph — are
    // labelled by k_cc_finalize whatever stands here: no gather of their first neighbour)
    if (rp[v] < rp[v + 1] && !((cbits[v >> 5] >> (v & 31)) & 1u)) {
      const I first = col[rp[v]];  // smallest neighbour (rows are column-sorted)
      if (first < p) p = first;
    }
    parent[v] = p;
  }
}

// low-degree vertices: one lane per vertex; others queued for the wave kernel
__global__ __launch_bounds__(256) void k_cc_hook_small(const X *__restrict__ rp, const X *__restrict__ col,
                                                       I *parent, int64_t n, I *__restrict__ big_list,
                                                       const unsigned *__restrict__ cbits, RcmDev *__restrict__ dv) {
  const int64_t v0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t vb = v0 - sbx_lane(); vb < n; vb += stride) {  // wave-uniform trip count
    const int64_t v = vb + sbx_lane();
    // members of the component the first sweep covered are already labelled
    const bool todo = v < n && !((cbits[v >> 5] >> (v & 31)) & 1u);
    const I s = todo ? rp[v] : 0, e = todo ? rp[v + 1] : 0;
    const bool big = todo && (e - s > 16);
    const unsigned slot = sbx_wave_append(&dv->n_cc_big, big);
    if (big) big_list[slot] = (I)v;
    if (!todo || big) continue;
    for (I j = s; j < e; j++) {
      const I u = col[j];
      if (u < (I)v) cc_hook(parent, (I)v, u);
    }
  }
}

__global__ __launch_bounds__(256) void k_cc_hook_big(const X *__restrict__ rp, const X *__restrict__ col, I *parent,
                                                     const I *__restrict__ big_list,
                                                     const RcmDev *__restrict__ dv) {
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int lane = sbx_lane();
  const int64_t cnt = dv->n_cc_big;
  for (int64_t k = wave; k < cnt; k += nwaves) {
    const I v = big_list[k];
    const I s = rp[v], e = rp[v + 1];
    for (I j = s + lane; j < e; j += 64) {
      const I u = col[j];
      if (u < v) cc_hook(parent, v, u);
    }
  }
}

// label[v] = root; component sizes.  Equal roots are combined inside the wave, and the
// workgroup's dominant root (the giant component) is accumulated in LDS and flushed
// once, so the giant's counter word is not hammered by one atomic per wave.
__global__ __launch_bounds__(256) void k_cc_finalize(const X *__restrict__ rp, I *parent, I *__restrict__ csize,
                                                     int64_t n, const unsigned *__restrict__ cbits, I first_root,
                                                     I first_size, int phase, RcmDev *__restrict__ dv) {
  __shared__ I s_major;
  __shared__ unsigned s_major_cnt;
  __shared__ unsigned s_empty[4];
  int64_t v0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (threadIdx.x == 0) {
    s_major = -1;
    s_major_cnt = 0;
  }
  __syncthreads();
  if (phase == 0) {
    // Phase 0: a vertex with an empty row that is its own root starts its component's count with a plain
    // (coalesced) store.  On a symmetric pattern it is isolated and that is all; on an unsymmetric one other
    // vertices may point at it and add themselves in phase 1 — a launch later, so the store cannot race them.
    unsigned empty = 0;
    for (int64_t v = v0; v < n; v += stride) {
      const bool in_first = (cbits[v >> 5] >> (v & 31)) & 1u;
      if (rp[v] == rp[v + 1]) {
        empty++;
        if (!in_first && parent[v] == (I)v) csize[v] = 1;
      }
    }
    empty = sbx_block_sum<unsigned, 256>(empty, s_empty);
    if (threadIdx.x == 0 && empty) atomicAdd(&dv->n_empty_rows, empty);
    return;
  }
  bool unsym = false;
  for (int64_t vb = v0 - sbx_lane(); vb < n; vb += stride) {
    const int64_t v = vb + sbx_lane();
    I root = -1;
    if (v < n) {
      if ((cbits[v >> 5] >> (v & 31)) & 1u) {
        parent[v] = first_root;  // labelled by the first sweep; its size is known
      } else if (rp[v] == rp[v + 1] && parent[v] == (I)v) {
        // counted in phase 0
      } else {
        root = parent[v];
        while (root != parent[root]) root = parent[root];
        parent[v] = root;
        // a vertex the first sweep did not reach cannot hang under one it did reach unless some edge has no mirror
        if ((cbits[root >> 5] >> (root & 31)) & 1u) unsym = true;
      }
    }
    uint64_t todo = __ballot(root >= 0);
    while (todo) {
      const int leader = __builtin_ctzll(todo);
      const I lr = (I)__builtin_amdgcn_readlane((int)root, leader);
      const uint64_t same = __ballot(root == lr) & todo;
      if (sbx_lane() == leader) {
        const unsigned cnt = (unsigned)__popcll(same);
        if (cnt >= 32) {  // a dominant root: try to own the workgroup's LDS accumulator
          const I prev = atomicCAS(&s_major, (I)-1, lr);
          if (prev == -1 || prev == lr) atomicAdd(&s_major_cnt, cnt);
          else atomicAdd(&csize[lr], (I)cnt);
        } else {
          atomicAdd(&csize[lr], (I)cnt);
        }
      }
      todo &= ~same;
    }
  }
  if (__any(unsym) && sbx_lane() == 0) dv->unsym = 1;
  __syncthreads();
  if (threadIdx.x == 0 && s_major_cnt) atomicAdd(&csize[s_major], (I)s_major_cnt);
  if (blockIdx.x == 0 && threadIdx.x == 0 && first_size > 0) csize[first_root] = first_size;
}

// classify components: singletons are final here; small / large roots are listed
__global__ __launch_bounds__(256) void k_classify(const I *__restrict__ label, const I *__restrict__ csize,
                                                  const I *__restrict__ cbase, X *__restrict__ inv,
                                                  I *__restrict__ small_list, I *__restrict__ mid_list,
                                                  I *__restrict__ large_list, int64_t n, RcmDev *__restrict__ dv) {
  __shared__ unsigned s_roots[4];
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned roots = 0;
  for (; v < n; v += stride) {
    if (label[v] != (I)v) continue;  // not a root
    roots++;
    const I sz = csize[v];
    if (sz == 1) inv[v] = cbase[v];
    else if (sz <= RCM_SMALL) small_list[atomicAdd(&dv->n_small, 1u)] = (I)v;
    else if (sz <= RCM_MID) mid_list[atomicAdd(&dv->n_mid, 1u)] = (I)v;
    else large_list[atomicAdd(&dv->n_large, 1u)] = (I)v;
  }
  roots = sbx_block_sum<unsigned, 256>(roots, s_roots);  // one add per workgroup: the counter word is hot otherwise
  if (threadIdx.x == 0 && roots) atomicAdd(&dv->n_components, roots);
}

// ------------------------------------------------------------------ small components
// One lane runs the reference's serial algorithm for one component.  q = the
// component's own slice of the order array, dist = BFS distances (UNSEEN = unvisited).
// (degree, id) order of the children a parent discovered (rcm_reorder.cc:125-144 drains a min-heap of such pairs)
__device__ __forceinline__ bool rcm_child_less(const X *__restrict__ rp, I a, I b) {
  const I da = rp[a + 1] - rp[a], db = rp[b + 1] - rp[b];
  return da < db || (da == db && a < b);
}
// in-place heapsort of a parent's children: hubs of mid-size components (a star of 2000 leaves) would cost an
// insertion sort millions of moves in one lane
__device__ void rcm_sort_children(const X *__restrict__ rp, I *a, int cnt) {
  auto sift = [&](int root, int end) {
    while (2 * root + 1 < end) {
      int child = 2 * root + 1;
      if (child + 1 < end && rcm_child_less(rp, a[child], a[child + 1])) child++;
      if (!rcm_child_less(rp, a[root], a[child])) return;
      const I t = a[root];
      a[root] = a[child];
      a[child] = t;
      root = child;
    }
  };
  for (int i = cnt / 2 - 1; i >= 0; i--) sift(i, cnt);
  for (int end = cnt - 1; end > 0; end--) {
    const I t = a[0];
    a[0] = a[end];
    a[end] = t;
    sift(0, end);
  }
}

// One component (its root `start`), one lane.
__device__ void rcm_small_component(const X *__restrict__ rp, const X *__restrict__ col, const I start,
                                    const I *__restrict__ csize, const I *__restrict__ cbase, unsigned *dist, I *order,
                                    X *__restrict__ inv, RcmDev *__restrict__ dv) {
  const I base = cbase[start], sz = csize[start];
  I *q = order + base;
  // --- pseudo-peripheral search (rcm_reorder.cc:22-81)
  I root = start;
  int prev_ecc = -1, ecc = 0;
  int reached = 0;  // vertices in q after the last sweep (all of them carry a distance until reset)
  while (prev_ecc != ecc) {
    prev_ecc = ecc;
    int head = 0, tail = 0;
    dist[root] = 0;
    q[tail++] = root;
    while (head < tail) {
      const I u = q[head++];
      const unsigned du = dist[u];
      for (I j = rp[u]; j < rp[u + 1]; j++) {
        const I v = col[j];
        if (dist[v] == UNSEEN) {
          dist[v] = du + 1;
          q[tail++] = v;
          if ((int)(du + 1) > ecc) ecc = (int)(du + 1);
        }
      }
    }
    reached = tail;
    if (tail != (int)sz) break;  // not the whole (weakly connected) component: the pattern has an edge without a mirror
    if (head == ecc + 1) break;  // one vertex per level
    if (prev_ecc != ecc) {
      reached = 0;  // the loop below resets every distance
      bool have = false;
      I best = 0;
      for (int i = 0; i < head; i++) {
        const I w = q[i];
        if ((int)dist[w] == ecc) {
          const I d = rp[w + 1] - rp[w];
          if (!have) { best = d + 1; have = true; }
          if (d < best) { best = d; root = w; }
        }
        dist[w] = UNSEEN;
      }
    }
  }
  for (int i = 0; i < reached; i++) dist[q[i]] = UNSEEN;
  // --- Cuthill-McKee BFS (rcm_reorder.cc:118-144); dist doubles as the visited mark
  int head = 0, tail = 0;
  dist[root] = 0;
  q[tail++] = root;
  while (head < tail) {
    const I u = q[head++];
    const int first = tail;
    for (I j = rp[u]; j < rp[u + 1]; j++) {
      const I v = col[j];
      if (dist[v] == UNSEEN) {
        dist[v] = 1;
        q[tail++] = v;
      }
    }
    const int kids = tail - first;
    if (kids > 16) {
      rcm_sort_children(rp, q + first, kids);
    } else {
      for (int i = first + 1; i < tail; i++) {  // insertion into the (degree,id)-sorted run of u's children
        const I v = q[i];
        int p = i;
        while (p > first && rcm_child_less(rp, v, q[p - 1])) {
          q[p] = q[p - 1];
          p--;
        }
        q[p] = v;
      }
    }
  }
  if (tail != (int)sz) {  // a BFS that cannot reach its component: report, leave inv alone (q beyond tail is stale)
    dv->unsym = 1;
    for (int i = 0; i < tail; i++) dist[q[i]] = UNSEEN;
    return;
  }
  for (int i = 0; i < sz; i++) inv[q[i]] = base + (sz - 1 - i);  // reverse + invert (:146-160)
}

// count_dev: the number of listed components, still on the device (the launch then covers an upper bound: a lane takes
// every (lanes of the grid)-th component)
__global__ __launch_bounds__(64) void k_rcm_small(const X *__restrict__ rp, const X *__restrict__ col,
                                                  const I *__restrict__ list, unsigned count,
                                                  const I *__restrict__ csize, const I *__restrict__ cbase,
                                                  unsigned *dist, I *order, X *__restrict__ inv,
                                                  RcmDev *__restrict__ dv, const unsigned *__restrict__ count_dev) {
  if (count_dev) count = *count_dev;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < count; k += stride)
    rcm_small_component(rp, col, list[k], csize, cbase, dist, order, inv, dv);
}

// ------------------------------------------------------------------ large components: BFS machinery
// BFS bookkeeping (sized by the measured scatter ceilings of the chip: scattered
// atomics run at ~27 G/s, scattered 4-byte loads at ~70 G/s from a 16 MB table and
// ~180 G/s from an L2-resident one):
//   vbits  one bit per vertex, set for every vertex of the levels already ORDERED
//          (updated only between expansions, so plain loads are exact); n/8 bytes,
//          L2-resident, filters every edge that points backwards;
//   ppos   position of the smallest-position parent seen so far for a vertex of the
//          level being built (UNSEEN = untouched).  A relaxed agent-scope load shows
//          whether this edge can still lower it; only then the atomicMin is issued.
//          The winner of the UNSEEN -> p transition appends the vertex to the frontier.
// The start kernels of a sweep clear the sweep's bitmaps themselves (hipMemsetAsync costs the host ~6 us apiece to
// enqueue, and a sweep is latency-bound): every workgroup clears its share, and the LAST one to finish — a counter in
// RcmDev elects it — does the single-threaded part on cleared memory.  Returns true for that workgroup's thread 0.
constexpr unsigned RCM_START_GRID = 256;
struct StartClear {
  unsigned *a;  // up to three word ranges
  unsigned long long na;
  unsigned *b;
  unsigned long long nb;
  unsigned *c;
  unsigned long long nc;
};
// (see RcmDev::spec_guard; both words were written by kernels in front: the same answer in every workgroup)
#define SBX_SPEC_GUARD_LEAVE(dv) \
  if ((dv)->spec_guard && !(dv)->tie_done) return
__device__ __forceinline__ bool start_clear_elect(const StartClear &sc, RcmDev *__restrict__ dv) {
  const unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = t; i < sc.na; i += stride) sc.a[i] = 0;
  for (unsigned long long i = t; i < sc.nb; i += stride) sc.b[i] = 0;
  for (unsigned long long i = t; i < sc.nc; i += stride) sc.c[i] = 0;
  __syncthreads();  // (every wave's stores have reached L2 ...)
  if (threadIdx.x != 0) return false;
  __threadfence();  // (... and ONE release per workgroup writes them back: a fence per thread made this kernel 20 us)
  const unsigned done = atomicAdd(&dv->start_done, 1u);
  if (done != gridDim.x - 1) return false;
  __threadfence();
  dv->start_done = 0;
  return true;
}

__global__ __launch_bounds__(256) void k_bfs_start(const X *__restrict__ rp, unsigned *__restrict__ vbits,
                                                   unsigned *__restrict__ fbits, unsigned *__restrict__ lpos,
                                                   unsigned *__restrict__ ppos, I *__restrict__ q,
                                                   RcmDev *__restrict__ dv, I fixed_root, StartClear sc) {
  SBX_SPEC_GUARD_LEAVE(dv);
  if (!start_clear_elect(sc, dv)) return;
  dv->spec_guard = 0;
  const I r = fixed_root >= 0 ? fixed_root : (dv->root_pending ? (I)dv->root_next : (I)dv->root);
  dv->root_pending = 0;
  dv->root = (unsigned)r;
  vbits[r >> 5] = 1u << (r & 31);  // both bitmaps were cleared above
  fbits[r >> 5] = 1u << (r & 31);
  lpos[r] = 0;
  ppos[r] = 0;  // every vertex of the sweep has a parent position: "ppos set" == "visited" (k_visited_from_ppos)
  q[0] = r;
  dv->nf = 0;
  dv->n_heavy = 0;
  dv->hub_overflow = 0;
  dv->best = ~0ull;
  dv->fedges = (unsigned long long)(rp[r + 1] - rp[r]);  // degree sum of level 0
}

// Winners are staged per wave in LDS and appended to the frontier in batches: one
// hot counter word sustains only ~88 returning atomics per microsecond, so the
// append must not cost one atomic per ballot.
constexpr int RCM_STAGE = 512;  // staged vertices per wave

// bottom-up is chosen when the frontier owns more than this many times the edges of the unvisited rest
static double bu_ratio() {
  static const double r = sbx_env_test("SBX_DEBUG_BU_RATIO") ? atof(sbx_env_test("SBX_DEBUG_BU_RATIO")) : 4.0;
  return r;
}

// calls a handle keeps away from the persistent kernels after a grid barrier gave up (SBX_DEBUG_GB_BACKOFF: stress tests
// set 0 so that every call tries them again)
static int gb_backoff_calls() {
  static const int k = sbx_env_test("SBX_DEBUG_GB_BACKOFF") ? atoi(sbx_env_test("SBX_DEBUG_GB_BACKOFF")) : 16;
  return k;
}

static bool rcm_split_expand() {  // SBX_RCM_SPLIT_EXPAND=0: a wide frontier's light rows are expanded in front of its hubs
  static const bool on = !(sbx_env_test("SBX_RCM_SPLIT_EXPAND") && atoi(sbx_env_test("SBX_RCM_SPLIT_EXPAND")) == 0);
  return on;
}

static bool rcm_cc_overlap() {  // SBX_RCM_CC_OVERLAP=0: the labelling of the other components runs in line (see sbx_rcm_reorder)
  static const bool on = !(sbx_env_test("SBX_RCM_CC_OVERLAP") && atoi(sbx_env_test("SBX_RCM_CC_OVERLAP")) == 0);
  return on;
}

static bool rcm_overlap() {
  static const bool on = !(sbx_env_test("SBX_RCM_OVERLAP") && atoi(sbx_env_test("SBX_RCM_OVERLAP")) == 0);
  return on;
}

// first candidate root of the pseudo-peripheral search whose Cuthill-McKee sweep is run speculatively
static int64_t rcm_speculate_from() {
  static const int64_t k = sbx_env_tuning("SBX_DEBUG_RCM_SPECULATE") ? atoll(sbx_env_tuning("SBX_DEBUG_RCM_SPECULATE")) : 2;
  return k;
}

struct WaveStage {
  I *buf;                  // this wave's LDS slice
  unsigned cnt;            // wave-uniform fill level
  unsigned long long deg;  // per-lane: degrees of the vertices this lane appended
  unsigned *nf_ctr;             // the list's counter (dv->nf, or a level slot of the unordered sweeps)
  unsigned long long *fe_ctr;   // the degree-sum counter that goes with it
};

__device__ __forceinline__ void stage_flush(WaveStage &st, I *__restrict__ nf_list, RcmDev *__restrict__ dv) {
  if (st.cnt == 0) return;
  unsigned base = 0;
  if (sbx_lane() == 0) base = atomicAdd(st.nf_ctr, st.cnt);
  base = __shfl(base, 0, 64);  // (a v_readfirstlane here makes the bottom-up kernel 28 % slower: measured)
  __builtin_amdgcn_wave_barrier();
  for (unsigned i = sbx_lane(); i < st.cnt; i += 64) nf_list[base + i] = st.buf[i];
  __builtin_amdgcn_wave_barrier();
  st.cnt = 0;
}

// Kernel epilogue for a 256-thread workgroup: what is still staged is appended with one
// returning atomic per workgroup, and the degree / edge counters get one atomic each
// (8192 waves x 3 atomics on one cache line cost more than a small level's expansion).
// `scanned` is a per-lane partial.  Every thread of the workgroup must call it.
__device__ __forceinline__ void stage_end_block(WaveStage &st, I *__restrict__ nf_list, RcmDev *__restrict__ dv,
                                                unsigned long long scanned, bool bottom_up) {
  __shared__ unsigned s_cnt[4];
  __shared__ unsigned long long s_deg[4], s_scan[4];
  __shared__ unsigned s_base;
  const int w = sbx_wave_in_block();
  const unsigned long long d = sbx_wave_sum(st.deg);
  scanned = sbx_wave_sum(scanned);
  if (sbx_lane() == 0) {
    s_cnt[w] = st.cnt;
    s_deg[w] = d;
    s_scan[w] = scanned;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned tot = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    const unsigned long long dsum = s_deg[0] + s_deg[1] + s_deg[2] + s_deg[3];
    const unsigned long long ssum = s_scan[0] + s_scan[1] + s_scan[2] + s_scan[3];
    s_base = tot ? atomicAdd(st.nf_ctr, tot) : 0u;
    if (dsum) atomicAdd(st.fe_ctr, dsum);
    if (ssum) {
      atomicAdd(&dv->edges, ssum);
      if (bottom_up) atomicAdd(&dv->edges_bu, ssum);
    }
  }
  __syncthreads();
  unsigned base = s_base;
  for (int q = 0; q < w; q++) base += s_cnt[q];
  for (unsigned i = sbx_lane(); i < st.cnt; i += 64) nf_list[base + i] = st.buf[i];
  st.cnt = 0;
}

// appends v (for lanes with `won`) to the staged frontier; all lanes of the wave call it
__device__ __forceinline__ void stage_push(I v, bool won, const X *__restrict__ rp, WaveStage &st,
                                           I *__restrict__ nf_list, RcmDev *__restrict__ dv) {
  const uint64_t winners = __ballot(won);
  if (winners) {
    if (won) {
      st.buf[st.cnt + __popcll(winners & sbx_lanemask_lt())] = v;
      st.deg += (unsigned long long)(rp[v + 1] - rp[v]);
    }
    st.cnt += (unsigned)__popcll(winners);
    if (st.cnt > RCM_STAGE - 64) stage_flush(st, nf_list, dv);
  }
}

// Visits up to 4 neighbours per lane in phases (bitmap tests, then coherent ppos
// pre-checks, then the atomics) so that 4 independent loads per lane are in flight
// instead of one 3-deep dependent chain per neighbour.
struct UnorderedSweep {  // state of a sweep that only needs the level SETS (see run_ubfs)
  unsigned char *claim8;  // MODE 1: one byte per vertex, set when an edge reaches it (plain stores: any parent will do)
  unsigned *nbits;        // MODE 2: the cone bitmap
  unsigned *dist;         // level of every reached vertex
  unsigned level;         // the level being built (MODE 2: the level being marked)
};

// MODE 0: ordered sweep (smallest parent position wins).  MODE 1: unordered sweep (the first edge to reach a vertex
// claims its bit in us.nbits).  MODE 2: cone marking of the tie-break (run_ubfs / ubfs_pick_root): the "frontier" is
// the marked vertices of level us.level + 1, visited are the REACHED vertices of level us.level not marked yet.
template <int MODE>
__device__ __forceinline__ void bfs_visit4(const I (&v)[4], unsigned actmask, unsigned p, const X *__restrict__ rp,
                                           const unsigned *__restrict__ vbits, unsigned *ppos, WaveStage &st,
                                           I *__restrict__ nf_list, RcmDev *__restrict__ dv, const UnorderedSweep &us) {
  unsigned seen = 0;  // bit k: v[k] is in the sweep's visited bitmap
  // (MODE 0 does not look: every visited vertex has a parent position, and that position lies in front of the frontier
  // being expanded — `ppos[v] > p` below is false for it.  One gather per edge instead of two for the unvisited ends.)
#ifdef SBX_RCM_ORDERED_VBITS
  constexpr bool LOOK = true;
#else
  constexpr bool LOOK = MODE != 0;
#endif
  if (LOOK) {
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (((actmask >> k) & 1u) && ((vbits[v[k] >> 5] >> (v[k] & 31)) & 1u)) seen |= 1u << k;
  }
  unsigned unv = MODE == 2 ? seen : (actmask & ~seen & 0xFu);
  unsigned won = 0;
  if (MODE == 2) {
    unsigned dk[4];
#pragma unroll
    for (int k = 0; k < 4; k++) dk[k] = ((unv >> k) & 1u) ? us.dist[v[k]] : 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (dk[k] != us.level) unv &= ~(1u << k);
  }
  if (MODE == 1) {
    // no atomics and no winner: the level is collected from the bytes afterwards (k_ubfs_collect_*)
    unsigned char cur[4];
#pragma unroll
    for (int k = 0; k < 4; k++) cur[k] = ((unv >> k) & 1u) ? us.claim8[v[k]] : (unsigned char)1;
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (!cur[k]) us.claim8[v[k]] = 1;  // (storing without the pre-check load measured 1 % slower)
    return;
  } else if (MODE == 2) {
    // bits only ever get set, so a stale word from the pre-check costs an atomic, not a result
    unsigned cur[4];
#pragma unroll
    for (int k = 0; k < 4; k++) cur[k] = ((unv >> k) & 1u) ? us.nbits[v[k] >> 5] : ~0u;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const unsigned bit = 1u << (v[k] & 31);
      if (((unv >> k) & 1u) && !(cur[k] & bit) && !(atomicOr(&us.nbits[v[k] >> 5], bit) & bit)) won |= 1u << k;
    }
  } else {
    unsigned cur[4];
#pragma unroll
    for (int k = 0; k < 4; k++)
      cur[k] = ((unv >> k) & 1u) ? __hip_atomic_load(&ppos[v[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (((unv >> k) & 1u) && cur[k] > p && atomicMin(&ppos[v[k]], p) == UNSEEN) won |= 1u << k;
  }
#pragma unroll
  for (int k = 0; k < 4; k++) stage_push(v[k], (won >> k) & 1u, rp, st, nf_list, dv);
}

__device__ __forceinline__ void bfs_visit(I v, unsigned p, const X *__restrict__ rp,
                                          const unsigned *__restrict__ vbits, unsigned *ppos, WaveStage &st,
                                          I *__restrict__ nf_list, RcmDev *__restrict__ dv, bool active) {
  bool won = false;
  if (active && !((vbits[v >> 5] >> (v & 31)) & 1u)) {
    const unsigned cur = __hip_atomic_load(&ppos[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur > p) won = (atomicMin(&ppos[v], p) == UNSEEN);
  }
  stage_push(v, won, rp, st, nf_list, dv);
}

// Light expansion: RCM_GROUP lanes per frontier vertex (4 vertices per wave) because
// the kernel is bound by the dependent-load chain frontier -> row_ptr -> col -> bitmap,
// not by lanes; hubs (> RCM_LIGHT neighbours) are queued as 1024-neighbour chunk
// descriptors for k_bfs_expand_heavy.
constexpr int RCM_GROUP = 16;
constexpr int RCM_VPW = 64 / RCM_GROUP;  // frontier vertices per wave
constexpr int RCM_BU_INLINE = 16;        // bottom-up: candidates up to this degree get one lane each
constexpr int RCM_BU_HEAVY = 256;        // ... above this one their rows are queued as chunks of
constexpr int RCM_BU_CHUNK = 1024;       //     this many entries, a wave each (k_bfs_bottom_up_heavy)

constexpr int RCM_HUB_STAGE = 256;  // hubs a workgroup stages before it reserves their chunk descriptors
constexpr int RCM_DIR_MAX = 2048;    // workgroups of k_bfs_expand the hub kernel can follow through the directory

template <int U>
__global__ __launch_bounds__(256) void k_bfs_expand(const X *__restrict__ rp, const X *__restrict__ col,
                                                    const I *__restrict__ frontier, unsigned fsize,
                                                    unsigned next_level, const unsigned *__restrict__ vbits,
                                                    unsigned *ppos, I *__restrict__ nf_list, uint64_t *__restrict__ heavy,
                                                    uint2 *__restrict__ hub_dir, RcmDev *__restrict__ dv,
                                                    UnorderedSweep us, int part) {
  // part 0: everything; 1: only the hubs' chunk descriptors (so that the hub kernel can start at once), 2: only the
  // light rows (beside the hub kernel, on a stream of its own) — how a wide frontier is expanded (run_bfs)
  __shared__ I s_stage[4][RCM_STAGE];
  // hubs found by this workgroup: (frontier position, number of chunks).  A frontier of the RMAT input holds
  // ~20 K hubs; one returning atomic each on dv->n_heavy is 0.2 ms of queueing on that word.  (The hub kernel
  // is ~10 % faster when it meets the descriptors in position order — a parent position is then rarely
  // lowered twice.  Workgroups therefore own CONTIGUOUS position ranges and leave (first descriptor, count) in
  // a directory indexed by workgroup: the hub kernel walks the directory, i.e. the positions, front to back.)
  __shared__ unsigned s_hub_p[RCM_HUB_STAGE], s_hub_n[RCM_HUB_STAGE];
  __shared__ unsigned s_hub_cnt, s_hub_base;
  __shared__ unsigned s_hub_scan[256 / 64 + 1];
  if (threadIdx.x == 0) s_hub_cnt = 0;
  __syncthreads();
  const int lane = sbx_lane();
  const int grp = lane / RCM_GROUP, gl = lane % RCM_GROUP;
  WaveStage st{s_stage[sbx_wave_in_block()], 0u, 0ull, &dv->nf, &dv->fedges};
  unsigned long long scanned = 0;
  if (U == 2) {  // cone marking: the range of the list to expand is on the device (no host round trip per level)
    frontier += dv->cone_begin;
    fsize = dv->cone_end - dv->cone_begin;
  }
  // positions [blockIdx.x * per, +per) belong to this workgroup; its four waves interleave inside the range
  const int64_t per = (((int64_t)fsize + gridDim.x - 1) / gridDim.x + 4 * RCM_VPW - 1) / (4 * RCM_VPW) * (4 * RCM_VPW);
  const int64_t pb = (int64_t)blockIdx.x * per;
  const int64_t pe = pb + per < (int64_t)fsize ? pb + per : (int64_t)fsize;
  for (int64_t p0 = pb + (int64_t)sbx_wave_in_block() * RCM_VPW; p0 < pe; p0 += 4 * RCM_VPW) {
    const int64_t p = p0 + grp;
    I s = 0, e = 0;
    if (p < pe) {
      const I u = frontier[p];
      s = rp[u];
      e = rp[u + 1];
    }
    if (part == 2 && e - s > RCM_LIGHT) e = s;  // (queued by the launch in front)
    if (part == 1 && e - s <= RCM_LIGHT) e = s;  // (the other launch's)
    if (e - s > RCM_LIGHT) {
      // hub: one descriptor (position, chunk) per 1024-neighbour chunk, queued at the end of the kernel
      const unsigned nchunks = (unsigned)((e - s + RCM_CHUNK - 1) / RCM_CHUNK);
      unsigned hs = 0;
      if (gl == 0) hs = atomicAdd(&s_hub_cnt, 1u);
      hs = __shfl(hs, grp * RCM_GROUP, 64);
      if (hs < (unsigned)RCM_HUB_STAGE) {
        if (gl == 0) {
          s_hub_p[hs] = (unsigned)p;
          s_hub_n[hs] = nchunks;
        }
      } else {  // stage full: reserve this hub's descriptors directly (the hub kernel then ignores the directory)
        unsigned slot = 0;
        if (gl == 0) {
          slot = atomicAdd(&dv->n_heavy, nchunks);
          dv->hub_overflow = 1;
        }
        slot = __shfl(slot, grp * RCM_GROUP, 64);
        for (unsigned c = gl; c < nchunks; c += RCM_GROUP) heavy[slot + c] = ((uint64_t)p << 32) | c;
      }
      e = s;  // nothing to do inline
    }
    if (gl == 0) scanned += (unsigned)(e - s);
    I j = s + gl;
    while (__any(j < e)) {
      I v[4];
      unsigned act = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const I jk = j + k * RCM_GROUP;
        v[k] = jk < e ? col[jk] : 0;
        if (jk < e) act |= 1u << k;
      }
      bfs_visit4<U>(v, act, (unsigned)p, rp, vbits, ppos, st, nf_list, dv, us);
      j += 4 * RCM_GROUP;
    }
  }
  __syncthreads();
  if (part != 2) {  // the staged hubs: one reservation for the whole workgroup
    const unsigned nh = s_hub_cnt < (unsigned)RCM_HUB_STAGE ? s_hub_cnt : (unsigned)RCM_HUB_STAGE;
    const unsigned mine = threadIdx.x < nh ? s_hub_n[threadIdx.x] : 0u;
    unsigned all;
    const unsigned ex = sbx_block_exclusive_sum<unsigned, 256>(mine, s_hub_scan, &all);
    if (threadIdx.x == 0) {
      s_hub_base = all ? atomicAdd(&dv->n_heavy, all) : 0u;
      hub_dir[blockIdx.x] = make_uint2(s_hub_base, all);
    }
    if (threadIdx.x < nh) s_hub_n[threadIdx.x] = ex;  // now: first descriptor of the hub (its chunk count is `mine`)
    __syncthreads();
    // descriptors of hub h = [base + ex_h, base + ex_{h+1}): all threads share the writing
    for (unsigned hh = 0; hh < nh; hh++) {
      const unsigned first = s_hub_n[hh];
      const unsigned cnt = (hh + 1 < nh ? s_hub_n[hh + 1] : all) - first;
      const uint64_t hi = (uint64_t)s_hub_p[hh] << 32;
      for (unsigned c = threadIdx.x; c < cnt; c += 256) heavy[s_hub_base + first + c] = hi | c;
    }
  }
  stage_end_block(st, nf_list, dv, scanned, false);
}

#ifndef RCM_HEAVY_MINW
#define RCM_HEAVY_MINW 1
#endif
template <int U>
__global__ __launch_bounds__(256, RCM_HEAVY_MINW) void k_bfs_expand_heavy(const X *__restrict__ rp, const X *__restrict__ col,
                                                          const I *__restrict__ frontier, unsigned next_level,
                                                          const unsigned *__restrict__ vbits, unsigned *ppos,
                                                          I *__restrict__ nf_list,
                                                          const uint64_t *__restrict__ heavy,
                                                          const uint2 *__restrict__ hub_dir, unsigned ndir,
                                                          RcmDev *__restrict__ dv, UnorderedSweep us) {
  __shared__ I s_stage[4][RCM_STAGE];
  __shared__ unsigned s_dfirst[RCM_DIR_MAX + 1];  // directory: how many descriptors came before a workgroup's run
  __shared__ unsigned s_dscan[256 / 64 + 1];
  const unsigned nd = dv->n_heavy;  // chunk descriptors queued by k_bfs_expand
  if (nd == 0) return;              // a level without hubs: nothing staged, nothing to add to the counters
  if (U == 2) frontier += dv->cone_begin;
  // Without overflow the descriptors are visited in directory (= frontier position) order: the d-th one overall
  // is entry d - first[g] of workgroup g's run.  Otherwise: in queue order.
  const bool ordered = dv->hub_overflow == 0 && ndir <= (unsigned)RCM_DIR_MAX;
  if (ordered) {
    constexpr int PER = RCM_DIR_MAX / 256;
    unsigned cnt[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) {
      const unsigned g = threadIdx.x * PER + k;
      uint2 e = make_uint2(0u, 0u);
      if (g < ndir) e = hub_dir[g];
      cnt[k] = e.y;
      sum += e.y;
    }
    unsigned all;
    unsigned run = sbx_block_exclusive_sum<unsigned, 256>(sum, s_dscan, &all);
#pragma unroll
    for (int k = 0; k < PER; k++) {
      s_dfirst[threadIdx.x * PER + k] = run;
      run += cnt[k];
    }
    if (threadIdx.x == 255) s_dfirst[RCM_DIR_MAX] = run;
    __syncthreads();
  }
  WaveStage st{s_stage[sbx_wave_in_block()], 0u, 0ull, &dv->nf, &dv->fedges};
  unsigned long long scanned = 0;
  unsigned g = 0;
  for (unsigned d = blockIdx.x; d < nd; d += gridDim.x) {
    unsigned at = d;
    if (ordered) {
      // last g with first[g] <= d (first[] is non-decreasing; d only grows, so the search resumes at g)
      unsigned lo = g, hi = (unsigned)RCM_DIR_MAX;
      while (lo < hi) {
        const unsigned mid = (lo + hi + 1) >> 1;
        if (s_dfirst[mid] <= d) lo = mid; else hi = mid - 1;
      }
      g = lo;
      at = hub_dir[g].x + (d - s_dfirst[g]);
    }
    const uint64_t desc = heavy[at];
    const unsigned p = (unsigned)(desc >> 32), c = (unsigned)desc;
    const I u = frontier[p];
    const I s = rp[u], e = rp[u + 1];
    const I cs = s + (I)c * RCM_CHUNK;
    const I ce = (cs + RCM_CHUNK < e) ? cs + RCM_CHUNK : e;
    {
      I v[4];
      unsigned act = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const I j = cs + k * 256 + (I)threadIdx.x;
        v[k] = j < ce ? col[j] : 0;
        if (j < ce) act |= 1u << k;
      }
      bfs_visit4<U>(v, act, p, rp, vbits, ppos, st, nf_list, dv, us);
    }
    scanned += (unsigned long long)(ce - cs);
  }
  stage_end_block(st, nf_list, dv, threadIdx.x == 0 ? scanned : 0ull, false);
}

// Bottom-up expansion (direction-optimising BFS): instead of the frontier pushing along
// its edges with atomics, every still-unvisited vertex of the component pulls — it scans
// its own adjacency, keeps the smallest level position among neighbours that are in the
// current frontier (fbits / lpos) and, if it found one, joins the next level.  No atomics
// on vertex state, no hot words; chosen by the host when the frontier owns more edges
// than the unvisited remainder.
// (eight waves per SIMD: the chunk reservation below, which hardly ever runs, would otherwise cost the scans 30 registers)
__global__ __launch_bounds__(256, 8) void k_bfs_bottom_up(const X *__restrict__ rp, const X *__restrict__ col,
                                                       const I *__restrict__ label, I comp_label,
                                                       const unsigned *__restrict__ vbits,
                                                       const unsigned *__restrict__ fbits,
                                                       const unsigned *__restrict__ lpos, unsigned *__restrict__ ppos,
                                                       I *__restrict__ nf_list, int64_t n, RcmDev *__restrict__ dv,
                                                       uint64_t *__restrict__ heavy, uint64_t heavy_cap) {
  __shared__ I s_stage[4][RCM_STAGE];
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int lane = sbx_lane();
  const int grp = lane / RCM_GROUP, gl = lane % RCM_GROUP;
  WaveStage st{s_stage[sbx_wave_in_block()], 0u, 0ull, &dv->nf, &dv->fedges};
  unsigned long long scanned = 0;
  for (int64_t base = wave * 64; base < n; base += nwaves * 64) {
    const int64_t v = base + lane;
    bool cand = false;
    I s = 0, e = 0;
    if (v < n && !((vbits[v >> 5] >> (v & 31)) & 1u)) {
      s = rp[v];
      e = rp[v + 1];
      cand = (e > s) && (label == nullptr || label[v] == comp_label);
    }
    // hubs that are still unvisited (a sweep from the periphery meets the largest ones late) would keep one 16-lane
    // group busy for milliseconds — 64 entries per step of three dependent loads; forcing the bench matrix's widest level
    // bottom-up showed it: 6.1 ms for that level, 8000 waves waiting for a handful.  Rows above RCM_BU_HEAVY entries are
    // queued as chunks of RCM_BU_CHUNK for k_bfs_bottom_up_heavy, a wave each (the queue holds nnz / 256 + nnz / 1024
    // descriptors and more: enough for every such row of the graph)
    const bool hv = cand && (e - s) > RCM_BU_HEAVY;
    if (__any(hv)) {  // one reservation per wave: a lane's slots follow those of the lanes before it
      const unsigned nch = hv ? (unsigned)((e - s + RCM_BU_CHUNK - 1) / RCM_BU_CHUNK) : 0u;
      const unsigned incl = sbx_wave_inclusive_sum(nch);
      unsigned at = 0;
      if (lane == 63) at = atomicAdd(&dv->n_heavy, incl);
      at = __shfl(at, 63, 64) + incl - nch;
      if (hv) {
        if ((uint64_t)at + nch <= heavy_cap) {
          for (unsigned c = 0; c < nch; c++) heavy[at + c] = ((uint64_t)(uint32_t)v << 32) | c;
          cand = false;
        } else {  // a full queue: the row stays with its 16-lane group below, the slots it got in front of the end are voided
          for (uint64_t q = at; q < heavy_cap; q++) heavy[q] = ~0ull;
        }
      }
    }
    // low-degree candidates: one lane each (64 independent load chains per wave; the
    // rows of consecutive vertices are adjacent in col[], so the lanes share lines)
    const bool small = cand && (e - s) <= RCM_BU_INLINE;
    if (__any(small)) {
      // three unrolled phases keep RCM_BU_INLINE independent loads in flight per lane
      // instead of a 3-deep dependent chain per neighbour
      const int dg = small ? (int)(e - s) : 0;
      I us[RCM_BU_INLINE];
#pragma unroll
      for (int k = 0; k < RCM_BU_INLINE; k++) us[k] = k < dg ? col[s + k] : (I)-1;
      unsigned hit = 0;
#pragma unroll
      for (int k = 0; k < RCM_BU_INLINE; k++)
        if (us[k] >= 0 && ((fbits[us[k] >> 5] >> (us[k] & 31)) & 1u)) hit |= 1u << k;
      unsigned best = UNSEEN;
#pragma unroll
      for (int k = 0; k < RCM_BU_INLINE; k++) {
        const unsigned lp = ((hit >> k) & 1u) ? lpos[us[k]] : UNSEEN;
        best = lp < best ? lp : best;
      }
      if (small) scanned += (unsigned)dg;
      const bool found = small && best != UNSEEN;
      if (found) ppos[v] = best;
      stage_push((I)v, found, rp, st, nf_list, dv);
    }
    uint64_t todo = __ballot(cand && !small);
    while (todo) {
      // the next RCM_VPW candidates, one per 16-lane group
      uint64_t t = todo;
      int pick = -1;
      for (int g = 0; g < RCM_VPW; g++) {
        const int c = t ? __builtin_ctzll(t) : -1;
        if (g == grp) pick = c;
        if (t) t &= t - 1;
      }
      todo = t;
      const I cs = __shfl(s, pick < 0 ? 0 : pick, 64), ce = __shfl(e, pick < 0 ? 0 : pick, 64);
      unsigned best = UNSEEN;
      I j = (pick < 0 ? 0 : cs) + gl;
      const I jend = pick < 0 ? 0 : ce;
      while (__any(j < jend)) {
        I us[4];
#pragma unroll
        for (int k = 0; k < 4; k++) us[k] = (j + k * RCM_GROUP) < jend ? col[j + k * RCM_GROUP] : (I)-1;
        unsigned hit = 0;
#pragma unroll
        for (int k = 0; k < 4; k++)
          if (us[k] >= 0 && ((fbits[us[k] >> 5] >> (us[k] & 31)) & 1u)) hit |= 1u << k;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const unsigned lp = ((hit >> k) & 1u) ? lpos[us[k]] : UNSEEN;
          best = lp < best ? lp : best;
        }
        j += 4 * RCM_GROUP;
      }
      if (gl == 0 && pick >= 0) scanned += (unsigned)(ce - cs);
      static_assert(RCM_GROUP == 16, "a group is one DPP row");
      best = sbx_row16_reduce(best, SbxOpMin());
      const bool found = (gl == 0) && pick >= 0 && best != UNSEEN;
      const I nv = (I)(base + (pick < 0 ? 0 : pick));
      if (found) ppos[nv] = best;
      stage_push(nv, found, rp, st, nf_list, dv);
    }
  }
  stage_end_block(st, nf_list, dv, scanned, true);
}

// The queued row chunks of a bottom-up level: a wave scans the RCM_BU_CHUNK entries of one (four loads per lane in
// flight), the chunks of a row meet in an atomicMin on the candidate's parent position, and the chunk that finds it
// UNSEEN appends the vertex to the level.
__global__ __launch_bounds__(256) void k_bfs_bottom_up_heavy(const X *__restrict__ rp, const X *__restrict__ col,
                                                             const unsigned *__restrict__ fbits,
                                                             const unsigned *__restrict__ lpos, unsigned *ppos,
                                                             I *__restrict__ nf_list, RcmDev *dv,
                                                             const uint64_t *__restrict__ heavy, uint64_t heavy_cap) {
  // (the counter may have run past the queue's end: a reservation that did not fit voided its slots)
  const unsigned nh = dv->n_heavy < heavy_cap ? dv->n_heavy : (unsigned)heavy_cap;
  __shared__ I s_stage[4][RCM_STAGE];
  const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const int lane = sbx_lane();
  // (the level's counter is a hot word: winners are staged per wave and appended in batches, as in the kernel above)
  WaveStage st{s_stage[sbx_wave_in_block()], 0u, 0ull, &dv->nf, &dv->fedges};
  unsigned long long scanned = 0;
  for (unsigned i = wave; i < nh; i += nwaves) {
    const uint64_t d = heavy[i];
    if (d == ~0ull) continue;
    const I v = (I)(uint32_t)(d >> 32);
    const I r0 = rp[v];
    const I s = r0 + (I)(uint32_t)d * RCM_BU_CHUNK;
    const I r1 = rp[v + 1];
    const I e = (r1 - s) > RCM_BU_CHUNK ? s + RCM_BU_CHUNK : r1;
    unsigned best = UNSEEN;
    for (I j = s + lane; j < e; j += 4 * 64) {
      I us[4];
#pragma unroll
      for (int k = 0; k < 4; k++) us[k] = (j + k * 64) < e ? col[j + k * 64] : (I)-1;
      unsigned hit = 0;
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (us[k] >= 0 && ((fbits[us[k] >> 5] >> (us[k] & 31)) & 1u)) hit |= 1u << k;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const unsigned lp = ((hit >> k) & 1u) ? lpos[us[k]] : UNSEEN;
        best = lp < best ? lp : best;
      }
    }
    best = sbx_wave_min(best);
    bool won = false;
    if (lane == 0) {
      scanned += (unsigned long long)(e - s);
      // a row of one chunk — most of them — has nobody to meet
      if (best != UNSEEN) won = (r1 - r0) <= RCM_BU_CHUNK ? (ppos[v] = best, true) : atomicMin(&ppos[v], best) == UNSEEN;
    }
    stage_push(v, won, rp, st, nf_list, dv);
  }
  stage_end_block(st, nf_list, dv, scanned, true);
}

// the host has read the expansion's counters by the time a level is ordered: the key kernels clear them for the next one
__device__ __forceinline__ void reset_level_counters(RcmDev *__restrict__ dv) {
  dv->nf = 0;
  dv->n_heavy = 0;
  dv->hub_overflow = 0;
  dv->fedges = 0;
}

template <bool CM>
__global__ __launch_bounds__(256) void k_level_keys(const I *__restrict__ nf_list, unsigned nf,
                                                    const unsigned *__restrict__ ppos,
                                                    const uint32_t *__restrict__ drank, uint64_t *__restrict__ key,
                                                    RcmDev *__restrict__ dv, int low_bits) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; j < nf; j += stride) {  // the parent position sits right above the low field: no dead digit in between
    const I v = nf_list[j];
    key[j] = ((uint64_t)ppos[v] << low_bits) | (uint64_t)(CM ? drank[v] : (uint32_t)v);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) reset_level_counters(dv);
}

template <bool CM>
__global__ __launch_bounds__(256) void k_level_emit(const uint64_t *__restrict__ key, unsigned nf,
                                                    const uint32_t *__restrict__ dorder, I *__restrict__ q_level,
                                                    unsigned *__restrict__ vbits, unsigned *__restrict__ fbits,
                                                    unsigned *__restrict__ lpos, int mark_frontier, int set_bits,
                                                    RcmDev *__restrict__ dv) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; j < nf; j += stride) {
    const uint32_t lo = (uint32_t)key[j];
    const I v = (I)(CM ? dorder[lo] : lo);
    q_level[j] = v;
    if (set_bits) atomicOr(&vbits[v >> 5], 1u << (v & 31));  // this level is now ordered
    if (mark_frontier) {  // only a bottom-up expansion reads the frontier bitmap / positions
      if (set_bits) atomicOr(&fbits[v >> 5], 1u << (v & 31));
      lpos[v] = (unsigned)j;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    dv->nf = 0;
    dv->n_heavy = 0;
    dv->hub_overflow = 0;
    dv->fedges = 0;
  }
}

// Big levels do not set their bits one atomic per vertex (1.4 M single-bit atomics on a 512 KB bitmap take
// ~150 us): every vertex of the sweep has a parent position and nothing else has, so the visited bitmap IS
// (ppos != UNSEEN) — one coalesced pass over ppos, a ballot per 64 vertices, no atomics.  The frontier
// bitmap is what that pass adds to the old one.
__global__ __launch_bounds__(256) void k_visited_from_ppos(const unsigned *__restrict__ ppos,
                                                           unsigned long long *__restrict__ vbits64,
                                                           unsigned long long *__restrict__ fbits64, int64_t n) {
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int lane = sbx_lane();
  for (int64_t base = wave * 64; base < n; base += nwaves * 64) {
    const int64_t v = base + lane;
    const bool seen = v < n && ppos[v] != UNSEEN;
    const unsigned long long now = __ballot(seen);
    if (lane == 0) {
      const unsigned long long before = vbits64[base >> 6];
      vbits64[base >> 6] = now;
      if (fbits64) fbits64[base >> 6] = now & ~before;
    }
  }
}

// Plain sweeps, big levels: the vertices the level discovered are exactly what the pass above adds to the
// visited bitmap, and reading them off that bitmap yields them in ascending id order for free.  The sort key
// (parent position, id) then only needs its parent-position digits sorted (stable): half the radix passes.
constexpr int RCM_FW_WORDS = 64;     // 64-vertex words per workgroup of the two kernels below (4096 vertices)
constexpr int RCM_FW_INLINE = 4096;  // up to this many workgroups each one sums its predecessors' totals itself

__global__ __launch_bounds__(256) void k_fresh_words(const unsigned *__restrict__ ppos,
                                                     unsigned long long *__restrict__ vbits64,
                                                     unsigned long long *__restrict__ fbits64,
                                                     unsigned long long *__restrict__ fresh64, int *__restrict__ cnt,
                                                     int *__restrict__ btot, int64_t n) {
  constexpr int WPW = RCM_FW_WORDS / 4;  // words per wave: 16 consecutive ones, loaded together
  __shared__ int s_tot;
  __shared__ unsigned long long s_now[4][WPW];
  if (threadIdx.x == 0) s_tot = 0;
  __syncthreads();
  const int lane = sbx_lane(), wv = threadIdx.x >> 6;
  const int64_t w0 = (int64_t)blockIdx.x * RCM_FW_WORDS + (int64_t)wv * WPW;
  unsigned pv[WPW];
#pragma unroll
  for (int i = 0; i < WPW; i++) {
    const int64_t v = (w0 + i) * 64 + lane;
    pv[i] = v < n ? ppos[v] : UNSEEN;
  }
#pragma unroll
  for (int i = 0; i < WPW; i++) {
    const unsigned long long now = __ballot(pv[i] != UNSEEN);
    if (lane == 0) s_now[wv][i] = now;
  }
  __builtin_amdgcn_wave_barrier();
  int mine = 0;
  if (lane < WPW && (w0 + lane) * 64 < n) {  // lane = word: the 16 bitmap words of the wave in one go
    const int64_t w = w0 + lane;
    const unsigned long long now = s_now[wv][lane];
    const unsigned long long fresh = now & ~vbits64[w];
    vbits64[w] = now;
    if (fbits64) fbits64[w] = fresh;
    fresh64[w] = fresh;
    mine = __popcll(fresh);
    cnt[w] = mine;
  }
  mine = sbx_wave_sum(mine);
  if (lane == 0 && mine) atomicAdd(&s_tot, mine);
  __syncthreads();
  if (threadIdx.x == 0) btot[blockIdx.x] = s_tot;
}

// The Cuthill-McKee sweep's big levels, the same idea in RANK space: bit k of the words written here says that the
// vertex of (degree, id) rank k — dorder[k] — was discovered by the current level (it has a parent position and is not
// in the visited bitmap yet, which k_visited_from_ppos only updates after the level is ordered).  Read off in bit order
// the level's vertices come out in ascending degree rank — the low field of their sort keys — so only the
// parent-position digits are left for the (stable) radix sort: 2 digit passes instead of 5 for the bench matrix's
// 1.4 M-vertex level.  The price is two 4-byte gathers per ranked vertex (ppos, the visited word) instead of a
// coalesced pass: worth it from a level of n_ranked / 3 vertices on.
__global__ __launch_bounds__(256) void k_fresh_words_ranked(const unsigned *__restrict__ ppos,
                                                            const unsigned *__restrict__ vbits,
                                                            const uint32_t *__restrict__ dorder,
                                                            unsigned long long *__restrict__ fresh64,
                                                            int *__restrict__ cnt, int *__restrict__ btot,
                                                            int64_t n_ranked) {
  constexpr int WPW = RCM_FW_WORDS / 4;  // words per wave: 16 consecutive ones
  __shared__ int s_tot;
  if (threadIdx.x == 0) s_tot = 0;
  __syncthreads();
  const int lane = sbx_lane(), wv = threadIdx.x >> 6;
  const int64_t w0 = (int64_t)blockIdx.x * RCM_FW_WORDS + (int64_t)wv * WPW;
  uint32_t vv[WPW];
#pragma unroll
  for (int i = 0; i < WPW; i++) {  // (all loads of a round together, at clamped indices: none waits for another)
    const int64_t k = (w0 + i) * 64 + lane;
    vv[i] = dorder[k < n_ranked ? k : n_ranked - 1];
  }
  unsigned pv[WPW], vw[WPW];
#pragma unroll
  for (int i = 0; i < WPW; i++) pv[i] = ppos[vv[i]], vw[i] = vbits[vv[i] >> 5];
  int mine = 0;
#pragma unroll
  for (int i = 0; i < WPW; i++) {
    const int64_t k = (w0 + i) * 64 + lane;
    const bool fresh = k < n_ranked && pv[i] != UNSEEN && !((vw[i] >> (vv[i] & 31)) & 1u);
    const unsigned long long f = __ballot(fresh);
    if (lane == 0 && (w0 + i) * 64 < n_ranked) {
      fresh64[w0 + i] = f;
      cnt[w0 + i] = __popcll(f);
      mine += __popcll(f);
    }
  }
  if (lane == 0 && mine) atomicAdd(&s_tot, mine);
  __syncthreads();
  if (threadIdx.x == 0) btot[blockIdx.x] = s_tot;
}

// One workgroup per RCM_FW_WORDS words.  Its base = totals of the workgroups before it (summed here when there
// are few of them, taken from their scan otherwise); every wave scans the 64 word counts itself (lane = word) and
// then writes the keys of its 16 words with lane = vertex: parent positions read and keys written in runs.
__global__ __launch_bounds__(256) void k_keys_from_fresh(unsigned long long *__restrict__ fresh64,
                                                         const int *__restrict__ cnt, const int *__restrict__ btot,
                                                         int btot_is_scanned, const unsigned *__restrict__ ppos,
                                                         const uint32_t *__restrict__ dorder,  // ranked form: bit k = rank k
                                                         uint64_t *__restrict__ key, uint32_t *__restrict__ key32,
                                                         uint32_t *__restrict__ val32, int64_t words,
                                                         RcmDev *__restrict__ dv) {
  constexpr int WPW = RCM_FW_WORDS / 4;
  const int lane = sbx_lane(), wv = threadIdx.x >> 6;
  if (blockIdx.x == 0 && threadIdx.x == 0) reset_level_counters(dv);
  int base = 0;
  if (btot_is_scanned) {
    base = btot[blockIdx.x];
  } else {
    for (int i = lane; i < (int)blockIdx.x; i += 64) base += btot[i];
    base = sbx_wave_sum(base);
  }
  const int64_t wb = (int64_t)blockIdx.x * RCM_FW_WORDS;
  const int c = wb + lane < words ? cnt[wb + lane] : 0;
  const int inc = sbx_wave_inclusive_sum(c);  // over the workgroup's 64 words
  const int excl = base + inc - c;
  const bool own = lane >= wv * WPW && lane < (wv + 1) * WPW && wb + lane < words;
  const unsigned long long mine = own ? fresh64[wb + lane] : 0ull;
  if (key32 && mine) fresh64[wb + lane] = 0;  // (the scattered form of the bitmap wants it clear: k_rank_scatter)
  // (the gathers of the wave's 16 words in three rounds — rank -> vertex, vertex -> parent position, stores — instead
  // of 16 dependent chains one after the other: 17 -> 6 us on a 28 K-vertex level)
  uint32_t vtx[WPW], low[WPW];
  int at[WPW];
  unsigned long long onmask = 0;  // bit i: this lane holds a vertex of word i
#pragma unroll
  for (int i = 0; i < WPW; i++) {
    const int src = wv * WPW + i;
    const unsigned long long f = __shfl(mine, src, 64);
    const int o = __shfl(excl, src, 64);
    const bool on = (f >> lane) & 1ull;
    low[i] = (uint32_t)((wb + src) * 64 + lane);  // vertex id, or degree rank in the ranked form
    at[i] = o + __popcll(f & sbx_lanemask_lt());
    if (on) onmask |= 1ull << i;
    vtx[i] = (on && dorder) ? dorder[low[i]] : low[i];
  }
  unsigned pp[WPW];
#pragma unroll
  for (int i = 0; i < WPW; i++) pp[i] = ((onmask >> i) & 1ull) ? ppos[vtx[i]] : 0u;
#pragma unroll
  for (int i = 0; i < WPW; i++) {
    if ((onmask >> i) & 1ull) {
      if (key32) {
        key32[at[i]] = pp[i];
        val32[at[i]] = low[i];
      } else {
        key[at[i]] = ((uint64_t)pp[i] << 32) | (uint64_t)low[i];
      }
    }
  }
}

// Levels too small for a pass over every (ranked) vertex: the level's own vertices set their bits — bit drank[v] in
// rank space (Cuthill-McKee) or bit v (plain sweeps) — in a bitmap that is clear between uses (k_bfs_start clears it,
// k_keys_from_fresh clears what it reads); k_rank_counts then leaves the per-word counts and per-workgroup totals
// k_fresh_words_ranked would have.
__global__ __launch_bounds__(256) void k_rank_scatter(const I *__restrict__ nf_list, unsigned nf,
                                                      const uint32_t *__restrict__ drank,
                                                      unsigned long long *__restrict__ fresh64) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; j < nf; j += stride) {
    const I v = nf_list[j];
    const uint32_t k = drank ? drank[v] : (uint32_t)v;
    atomicOr(&fresh64[k >> 6], 1ull << (k & 63));
  }
}
__global__ __launch_bounds__(64) void k_rank_counts(const unsigned long long *__restrict__ fresh64, int *__restrict__ cnt,
                                                    int *__restrict__ btot, int64_t words) {
  const int64_t w = (int64_t)blockIdx.x * RCM_FW_WORDS + threadIdx.x;
  const int c = w < words ? __popcll(fresh64[w]) : 0;
  if (w < words) cnt[w] = c;
  const int t = sbx_wave_sum(c);
  if (threadIdx.x == 0) btot[blockIdx.x] = t;
}

// what the last digit's placement of a level's counting sort does with a vertex (sbx_countsort.h): the queue entry,
// the bitmaps (small levels only: big ones rebuild them from ppos), the level position for a bottom-up expansion
struct LevelEmit {
  const uint32_t *map;  // rank -> vertex (Cuthill-McKee) or nullptr
  I *out;
  unsigned *bits_a, *bits_b, *pos_of;
  __device__ void operator()(unsigned pos, uint32_t, uint32_t val) const {
    const uint32_t v = map ? map[val] : val;
    out[pos] = (I)v;
    if (bits_a) atomicOr(&bits_a[v >> 5], 1u << (v & 31));
    if (bits_b) atomicOr(&bits_b[v >> 5], 1u << (v & 31));
    if (pos_of) pos_of[v] = pos;
  }
};

// small level: keys, sort in LDS and emit in one workgroup.  The sort is a bucket rank: the keys (all different)
// go to 4096 order-preserving buckets by their leading bits above the smallest key, the bucket counts are scanned,
// every key is placed in its bucket's range and ranked there among a handful of neighbours — 8 barriers where the
// bitonic network over 4096 keys takes 78 (77 us -> ~10 us for the bench matrix's 3.4 K-vertex level).  A bucket
// holding more than RCM_BR_MAX keys (keys clustered at several scales) sends the level to the bitonic network.
constexpr int RCM_BR_MAX = 48;
template <bool CM>
__global__ __launch_bounds__(1024) void k_level_sort_small(const I *__restrict__ nf_list, unsigned nf,
                                                           const unsigned *__restrict__ ppos,
                                                           const uint32_t *__restrict__ drank,
                                                           const uint32_t *__restrict__ dorder,
                                                           I *__restrict__ q_level, unsigned *__restrict__ vbits,
                                                           unsigned *__restrict__ fbits, unsigned *__restrict__ lpos,
                                                           int mark_frontier, RcmDev *__restrict__ dv) {
  constexpr int IPT = RCM_LDS_SORT / 1024;
  __shared__ uint64_t s_key[RCM_LDS_SORT];
  __shared__ uint64_t s_tmp[RCM_LDS_SORT];
  __shared__ unsigned s_end[RCM_LDS_SORT];
  __shared__ uint64_t s_mn[16], s_mx[16];
  __shared__ unsigned s_wsum[16], s_big;
  const int lane = sbx_lane(), wv = threadIdx.x >> 6;
  uint64_t mk[IPT];
  uint64_t mn = ~0ull, mx = 0;
#pragma unroll
  for (int i = 0; i < IPT; i++) {
    const unsigned j = threadIdx.x + i * 1024;
    mk[i] = ~0ull;
    if (j < nf) {
      const I v = nf_list[j];
      mk[i] = ((uint64_t)ppos[v] << 32) | (uint64_t)(CM ? drank[v] : (uint32_t)v);
      mn = mk[i] < mn ? mk[i] : mn;
      mx = mk[i] > mx ? mk[i] : mx;
    }
    s_end[j] = 0;
  }
  mn = sbx_wave_min(mn);
  mx = sbx_wave_max(mx);
  if (lane == 0) {
    s_mn[wv] = mn;
    s_mx[wv] = mx;
  }
  if (threadIdx.x == 0) s_big = 0;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; i++) {
    mn = s_mn[i] < mn ? s_mn[i] : mn;
    mx = s_mx[i] > mx ? s_mx[i] : mx;
  }
  const uint64_t range = mx - mn;  // nf >= 1
  int shift = 0;
  while (shift < 64 && (range >> shift) >= (uint64_t)RCM_LDS_SORT) shift++;
  unsigned bk[IPT];
#pragma unroll
  for (int i = 0; i < IPT; i++) {
    bk[i] = 0;
    if (threadIdx.x + i * 1024 < nf) {
      bk[i] = (unsigned)((mk[i] - mn) >> shift);
      atomicAdd(&s_end[bk[i]], 1u);
    }
  }
  __syncthreads();
  {  // inclusive scan of the bucket counts: thread t owns buckets 4t..4t+3
    const uint4 c = *(const uint4 *)&s_end[threadIdx.x * IPT];
    const unsigned big = c.x > c.y ? (c.x > c.z ? (c.x > c.w ? c.x : c.w) : (c.z > c.w ? c.z : c.w))
                                   : (c.y > c.z ? (c.y > c.w ? c.y : c.w) : (c.z > c.w ? c.z : c.w));
    if (big > (unsigned)RCM_BR_MAX) s_big = 1;
    const unsigned tsum = c.x + c.y + c.z + c.w;
    const unsigned inc = sbx_wave_inclusive_sum(tsum);
    if (lane == 63) s_wsum[wv] = inc;
    __syncthreads();
    unsigned base = inc - tsum;
#pragma unroll
    for (int i = 0; i < 16; i++)
      if (i < wv) base += s_wsum[i];
    uint4 e;
    e.x = base + c.x;
    e.y = e.x + c.y;
    e.z = e.y + c.z;
    e.w = e.z + c.w;
    *(uint4 *)&s_end[threadIdx.x * IPT] = e;
  }
  __syncthreads();
  if (!s_big) {
#pragma unroll
    for (int i = 0; i < IPT; i++)
      if (threadIdx.x + i * 1024 < nf) s_tmp[atomicSub(&s_end[bk[i]], 1u) - 1u] = mk[i];
    __syncthreads();  // s_end[b] is now where bucket b starts
#pragma unroll
    for (int i = 0; i < IPT; i++)
      if (threadIdx.x + i * 1024 < nf) {
        const unsigned lo = s_end[bk[i]], hi = bk[i] + 1 < (unsigned)RCM_LDS_SORT ? s_end[bk[i] + 1] : nf;
        unsigned r = lo;
        for (unsigned t = lo; t < hi; t++) r += s_tmp[t] < mk[i] ? 1u : 0u;
        s_key[r] = mk[i];
      }
    __syncthreads();
  } else {
    unsigned p2 = 1;
    while (p2 < nf) p2 <<= 1;
#pragma unroll
    for (int i = 0; i < IPT; i++)
      if (threadIdx.x + i * 1024 < p2) s_key[threadIdx.x + i * 1024] = mk[i];
    __syncthreads();
    for (unsigned k = 2; k <= p2; k <<= 1) {
      for (unsigned j = k >> 1; j > 0; j >>= 1) {
        for (unsigned t = threadIdx.x; t < p2; t += blockDim.x) {
          const unsigned l = t ^ j;
          if (l > t) {
            const uint64_t a = s_key[t], b = s_key[l];
            const bool up = (t & k) == 0;
            if ((a > b) == up) {
              s_key[t] = b;
              s_key[l] = a;
            }
          }
        }
        __syncthreads();
      }
    }
  }
  for (unsigned j = threadIdx.x; j < nf; j += blockDim.x) {
    const uint32_t lo = (uint32_t)s_key[j];
    const I v = (I)(CM ? dorder[lo] : lo);
    q_level[j] = v;
    atomicOr(&vbits[v >> 5], 1u << (v & 31));
    if (mark_frontier) {
      atomicOr(&fbits[v >> 5], 1u << (v & 31));
      lpos[v] = j;
    }
  }
  if (threadIdx.x == 0) {
    dv->nf = 0;
    dv->n_heavy = 0;
    dv->hub_overflow = 0;
    dv->fedges = 0;
  }
}

// ---- persistent small-level kernel ------------------------------------------------
// Deep, narrow BFS (banded / grid / road-like graphs, and the first and last levels of
// every sweep) is launch-latency bound when each level costs a host round trip.  One
// 1024-thread workgroup therefore runs consecutive levels on its own for as long as
// the frontier stays small: expand (atomicMin on ppos, winners listed in LDS and in
// nf_list), order the level with an LDS bitonic sort of the 64-bit keys, publish it
// (q, visited/frontier bitmaps, level positions) and go on.  It hands back to the host
// loop at a level boundary when the frontier or its edge count outgrows the budget
// (SL_STOP_READY), or right after an expansion whose result does not fit LDS
// (SL_STOP_EXPANDED: nf_list / dv->nf / dv->fedges are exactly what the general
// expansion kernels would have left).
constexpr unsigned SL_DONE = 0, SL_STOP_READY = 1, SL_STOP_EXPANDED = 2;
constexpr int SL_MAXF = 4096;    // frontier vertices handled in-kernel
constexpr int SL_MAXE = 32768;   // frontier adjacency entries handled in-kernel (32 per thread)
constexpr int SL_CAP = 8192;     // next-level vertices that fit the LDS sort
constexpr int SL_FPT = SL_MAXF / 1024;  // frontier entries per thread in the degree scan
constexpr int SL_GROUP = 64;            // children of one parent ordered by counting (longer: bitonic sort)
constexpr int SL_CH = 8;                // adjacency entries a thread loads together
static_assert(SL_MAXE <= 32 * 1024, "the plain sweep keeps one winner bit per entry of a thread's run");

template <bool CM>
__global__ __launch_bounds__(1024) void k_bfs_small_levels(const X *__restrict__ rp, const X *__restrict__ col,
                                                           I *q, unsigned *vbits, unsigned *fbits, unsigned *lpos,
                                                           unsigned *ppos, I *nf_list,
                                                           const uint32_t *__restrict__ drank,
                                                           const uint32_t *__restrict__ dorder, unsigned off,
                                                           unsigned fsize, unsigned level, unsigned total,
                                                           RcmDev *dv) {
  SBX_SPEC_GUARD_LEAVE(dv);  // (the start kernel in front left: so does this one)
  __shared__ uint64_t s_key[SL_CAP];
  __shared__ I s_front[SL_MAXF];
  __shared__ I s_start[SL_MAXF];          // rp[u] of every frontier vertex
  __shared__ unsigned s_eoff[SL_MAXF];    // exclusive prefix of the frontier degrees
  __shared__ unsigned s_escan[1024 / 64 + 1];
  __shared__ uint16_t s_rank[SL_CAP];     // Cuthill-McKee sweep: where an entry moves inside its parent's group
  __shared__ I s_nstart[CM ? 1 : SL_MAXF];        // plain sweep: rp[v] and degree of the vertices just discovered, in their
  __shared__ unsigned s_ndeg[CM ? 1 : SL_MAXF];   // final order - the next level starts without a row-pointer round trip
  __shared__ unsigned s_gfirst[CM ? SL_MAXF : 1];  // Cuthill-McKee sweep: first slot of every parent's group of children
  unsigned *s_gcnt = reinterpret_cast<unsigned *>(s_front);  // ... and its size (s_front is idle between the degree
                                                             // scan and the publication of the next frontier)
  __shared__ unsigned s_cnt;
  __shared__ unsigned long long s_deg[1024 / 64 + 1];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  unsigned long long ordered_edges = 0, scanned = 0;
  unsigned status = SL_STOP_READY;
  bool carried = false;  // s_nstart / s_ndeg describe the current frontier
  if (fsize <= SL_MAXF)
    for (unsigned i = tid; i < fsize; i += 1024) s_front[i] = q[off + i];
  __syncthreads();
  while (true) {
    if (fsize > SL_MAXF) break;  // status stays SL_STOP_READY
    // adjacency offsets of the frontier: s_start[i] = rp[u_i], s_eoff[i] = degrees of u_0..u_{i-1}
    unsigned dl[SL_FPT];
    unsigned mine = 0;
#pragma unroll
    for (int k = 0; k < SL_FPT; k++) {
      const unsigned i = (unsigned)tid * SL_FPT + k;
      dl[k] = 0;
      if (i < fsize) {
        if (!CM && carried) {
          s_start[i] = s_nstart[i];
          dl[k] = s_ndeg[i];
        } else {
          const I u = s_front[i];
          const I s0 = rp[u];
          s_start[i] = s0;
          dl[k] = (unsigned)(rp[u + 1] - s0);
        }
      }
      mine += dl[k];
    }
    unsigned etotal;
    unsigned run = sbx_block_exclusive_sum<unsigned, 1024>(mine, s_escan, &etotal);
#pragma unroll
    for (int k = 0; k < SL_FPT; k++) {
      const unsigned i = (unsigned)tid * SL_FPT + k;
      if (i < fsize) s_eoff[i] = run;
      run += dl[k];
    }
    const unsigned long long dsum = etotal;
    if (tid == 0) {
      s_cnt = 0;
      dv->fedges = dsum;  // what the host's direction heuristic expects for this frontier
    }
    if (dsum > (unsigned long long)SL_MAXE) break;  // SL_STOP_READY
    __syncthreads();
    if (CM)
      for (unsigned i = tid; i < fsize; i += 1024) {
        s_gfirst[i] = ~0u;
        s_gcnt[i] = 0;
      }
    // ---- expand, flattened over the frontier's adjacency entries: thread -> entry e, its frontier
    // position by binary search in the LDS prefix (a wave per vertex left 60 of 64 lanes idle on meshes)
    unsigned long long wdeg = 0;
    {
      // Plain BFS order is (parent position, id) = the order of the WINNING adjacency entries in
      // the flattened entry numbering (rows are column-sorted), so the new level needs no sort:
      // pass 1 settles the smallest parent position of every neighbour, pass 2 walks the entries
      // in order (a contiguous run per thread), keeps those that won and compacts them.  The
      // Cuthill-McKee sweep gets its children grouped by parent the same way and then only
      // orders each parent's group by degree rank.
      // Every thread owns a contiguous run of <= 32 entries and works through it in chunks of SL_CH with
      // the loads of a chunk issued together (col, then the bitmap words, then ppos): a level costs a
      // fixed handful of dependent memory round trips instead of a handful per entry.  The first chunk
      // (the whole run whenever the frontier has <= 8 K entries) stays in registers across both passes.
      const unsigned per = (etotal + 1023u) / 1024u;  // <= SL_MAXE / 1024 = 32 entries per thread
      const unsigned e0 = (unsigned)tid * per;
      const unsigned e1 = e0 + per < etotal ? e0 + per : etotal;
      unsigned pfirst = 0;
      if (e0 < etotal) {
        unsigned lo = 0, hi = fsize - 1;
        while (lo < hi) {
          const unsigned mid = (lo + hi + 1) >> 1;
          if (s_eoff[mid] <= e0) lo = mid; else hi = mid - 1;
        }
        pfirst = lo;
      }
      I v0[SL_CH];          // first chunk: neighbour ids,
      unsigned p0[SL_CH];   // frontier positions of their parents,
      unsigned unv0 = 0;    // not-yet-visited mask,
      I jfirst = 0;         // index in col[] of the run's first entry
      // (re)builds one chunk: parents by walking the LDS prefix, then col and the bitmap words in two batches
      auto load_chunk = [&](unsigned eb, unsigned &pcur, I (&v)[SL_CH], unsigned (&pp)[SL_CH], I &j0, unsigned &unv) {
        I jj[SL_CH];
#pragma unroll
        for (int k = 0; k < SL_CH; k++) {
          const unsigned e = eb + k;
          jj[k] = -1;
          pp[k] = 0;
          if (e < e1) {
            while (pcur + 1 < fsize && s_eoff[pcur + 1] <= e) pcur++;
            pp[k] = pcur;
            jj[k] = s_start[pcur] + (I)(e - s_eoff[pcur]);
          }
        }
        j0 = jj[0];
#pragma unroll
        for (int k = 0; k < SL_CH; k++) v[k] = jj[k] >= 0 ? col[jj[k]] : (I)0;
        unsigned word[SL_CH];
#pragma unroll
        for (int k = 0; k < SL_CH; k++)
          word[k] = jj[k] >= 0 ? __hip_atomic_load(&vbits[v[k] >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0u;
        unv = 0;
#pragma unroll
        for (int k = 0; k < SL_CH; k++)
          if (jj[k] >= 0 && !((word[k] >> (v[k] & 31)) & 1u)) unv |= 1u << k;
      };
      {  // pass 1: smallest parent position of every unvisited neighbour
        unsigned pcur = pfirst;
        for (unsigned eb = e0; eb < e1; eb += SL_CH) {
          I v[SL_CH];
          unsigned pp[SL_CH], unv;
          I j0;
          load_chunk(eb, pcur, v, pp, j0, unv);
#pragma unroll
          for (int k = 0; k < SL_CH; k++)
            if ((unv >> k) & 1u) atomicMin(&ppos[v[k]], pp[k]);
          if (eb == e0) {
#pragma unroll
            for (int k = 0; k < SL_CH; k++) {
              v0[k] = v[k];
              p0[k] = pp[k];
            }
            unv0 = unv;
            jfirst = j0;
          }
        }
      }
      __syncthreads();
      unsigned won = 0, cntw = 0;
      {  // pass 2: the entries that hold the winning (parent, neighbour) pairs, in entry order
        unsigned pcur = pfirst;
        for (unsigned eb = e0; eb < e1; eb += SL_CH) {
          I v[SL_CH];
          unsigned pp[SL_CH], unv;
          I j0;
          if (eb == e0) {
#pragma unroll
            for (int k = 0; k < SL_CH; k++) {
              v[k] = v0[k];
              pp[k] = p0[k];
            }
            unv = unv0;
            j0 = jfirst;
            pcur = p0[SL_CH - 1] > pcur ? p0[SL_CH - 1] : pcur;
          } else {
            load_chunk(eb, pcur, v, pp, j0, unv);
          }
          unsigned cur[SL_CH];
#pragma unroll
          for (int k = 0; k < SL_CH; k++)
            cur[k] = ((unv >> k) & 1u) ? __hip_atomic_load(&ppos[v[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : UNSEEN;
          // a duplicate entry of a row does not count: its predecessor in the same row holds the same id
          const bool has_prev = eb > s_eoff[pp[0]];
          const I prev0 = has_prev ? col[j0 - 1] : (I)-1;
#pragma unroll
          for (int k = 0; k < SL_CH; k++) {
            bool w1 = ((unv >> k) & 1u) && cur[k] == pp[k];
            if (k == 0) w1 = w1 && !(has_prev && prev0 == v[0]);
            else w1 = w1 && !(pp[k - 1] == pp[k] && v[k - 1] == v[k]);
            if (w1) {
              won |= 1u << (eb - e0 + k);
              cntw++;
            }
          }
        }
      }
      unsigned nfw;
      unsigned slot = sbx_block_exclusive_sum<unsigned, 1024>(cntw, s_escan, &nfw);
      if (tid == 0) s_cnt = nfw;
      {  // emission: keys of the winners in entry order, their degrees for the host's bookkeeping
        unsigned pcur = pfirst;
        for (unsigned eb = e0; eb < e1 && (won >> (eb - e0)); eb += SL_CH) {
          const unsigned wm = (won >> (eb - e0)) & ((1u << SL_CH) - 1u);
          I v[SL_CH];
          unsigned pp[SL_CH], unv;
          I j0;
          if (eb == e0) {
#pragma unroll
            for (int k = 0; k < SL_CH; k++) {
              v[k] = v0[k];
              pp[k] = p0[k];
            }
            pcur = p0[SL_CH - 1] > pcur ? p0[SL_CH - 1] : pcur;
          } else {
            load_chunk(eb, pcur, v, pp, j0, unv);
          }
          uint32_t dr[SL_CH];
          I ra[SL_CH], rb[SL_CH];
#pragma unroll
          for (int k = 0; k < SL_CH; k++) {
            const bool on = (wm >> k) & 1u;
            dr[k] = (CM && on) ? drank[v[k]] : 0u;
            ra[k] = on ? rp[v[k]] : (I)0;
            rb[k] = on ? rp[v[k] + 1] : (I)0;
          }
#pragma unroll
          for (int k = 0; k < SL_CH; k++)
            if ((wm >> k) & 1u) {
              if (slot < (unsigned)SL_CAP)
                s_key[slot] = CM ? (((uint64_t)pp[k] << 32) | (uint64_t)dr[k]) : (uint64_t)(uint32_t)v[k];
              nf_list[slot] = v[k];
              if (CM) {
                atomicMin(&s_gfirst[pp[k]], slot);
                atomicAdd(&s_gcnt[pp[k]], 1u);
              } else if (slot < (unsigned)SL_MAXF) {
                s_nstart[slot] = ra[k];
                s_ndeg[slot] = (unsigned)(rb[k] - ra[k]);
              }
              wdeg += (unsigned long long)(rb[k] - ra[k]);
              slot++;
            }
        }
      }
      __syncthreads();
    }
    scanned += dsum;
    const unsigned nf = s_cnt;
    if (nf == 0) {
      status = SL_DONE;
      break;
    }
    if (nf > (unsigned)SL_CAP) {  // cannot order it here: leave it to the host path
      wdeg = sbx_block_sum<unsigned long long, 1024>(wdeg, s_deg);
      if (tid == 0) {
        dv->nf = nf;
        dv->fedges = wdeg;
      }
      status = SL_STOP_EXPANDED;
      break;
    }
    ordered_edges += wdeg;  // degrees of the vertices discovered (and ordered) in-kernel: per-thread partial, summed at exit
    // ---- Cuthill-McKee sweep: s_key holds (parent position << 32 | degree rank), grouped by parent.
    // Groups of <= SL_GROUP children (every mesh) are ordered by counting smaller ranks inside the
    // group; a longer group (a hub) falls back to a bitonic sort of the whole level.
    if (CM) {
      unsigned maxc = 0;
      for (unsigned j = tid; j < nf; j += 1024) {
        const uint64_t kj = s_key[j];
        const uint32_t pj = (uint32_t)(kj >> 32);
        const unsigned a = s_gfirst[pj], c = s_gcnt[pj], bnd = a + c;  // the group's slots, recorded by the emission
        maxc = c > maxc ? c : maxc;
        unsigned r = 0;
        if (c <= (unsigned)SL_GROUP)
          for (unsigned i = a; i < bnd; i++) r += (uint32_t)s_key[i] < (uint32_t)kj;
        s_rank[j] = (uint16_t)(a + r - j + SL_GROUP);  // destination - source, biased to stay non-negative
      }
      maxc = sbx_wave_max(maxc);
      if (lane == 0) s_escan[w] = maxc;
      __syncthreads();
      maxc = 0;
      for (int i = 0; i < 1024 / 64; i++) maxc = s_escan[i] > maxc ? s_escan[i] : maxc;
      __syncthreads();
      if (maxc <= (unsigned)SL_GROUP) {
        uint64_t mine[SL_CAP / 1024];
#pragma unroll
        for (int k = 0; k < SL_CAP / 1024; k++) {
          const unsigned j = (unsigned)tid + k * 1024u;
          mine[k] = j < nf ? s_key[j] : 0;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SL_CAP / 1024; k++) {
          const unsigned j = (unsigned)tid + k * 1024u;
          if (j < nf) s_key[j + s_rank[j] - SL_GROUP] = mine[k];
        }
        __syncthreads();
      } else {
        unsigned p2 = 1;
        while (p2 < nf) p2 <<= 1;
        for (unsigned j = nf + tid; j < p2; j += 1024) s_key[j] = ~0ull;
        __syncthreads();
        for (unsigned k = 2; k <= p2; k <<= 1) {
          for (unsigned j = k >> 1; j > 0; j >>= 1) {
            for (unsigned t = tid; t < p2; t += 1024) {
              const unsigned l = t ^ j;
              if (l > t) {
                const uint64_t x = s_key[t], y = s_key[l];
                const bool up = (t & k) == 0;
                if ((x > y) == up) {
                  s_key[t] = y;
                  s_key[l] = x;
                }
              }
            }
            __syncthreads();
          }
        }
      }
    }
    // ---- publish: q and the visited bitmap.  The frontier bitmap / level positions are only read by
    // the bottom-up kernel, which cannot follow a frontier this small: the host marks them
    // (k_mark_frontier) in the rare case it wants bottom-up right after this kernel stops.
    const unsigned noff = off + fsize;
    for (unsigned j = tid; j < nf; j += 1024) {
      const uint32_t lo = (uint32_t)s_key[j];
      const I v = (I)(CM ? dorder[lo] : lo);
      q[noff + j] = v;
      atomicOr(&vbits[v >> 5], 1u << (v & 31));
      if (j < (unsigned)SL_MAXF) s_front[j] = v;
    }
    off = noff;
    fsize = nf;
    total += nf;
    level++;
    carried = true;
    __syncthreads();
  }
  ordered_edges = sbx_block_sum<unsigned long long, 1024>(ordered_edges, s_deg);
  if (tid == 0) {
    dv->sl_off = off;
    dv->sl_fsize = fsize;
    dv->sl_level = level;
    dv->sl_total = total;
    dv->sl_status = status;
    dv->sl_edges = ordered_edges;
    if (status != SL_STOP_EXPANDED) {
      dv->nf = 0;
      dv->n_heavy = 0;
      dv->hub_overflow = 0;
    }
    atomicAdd(&dv->edges, scanned);
  }
}

// frontier bitmap + level positions of q[0..fsize) for a bottom-up expansion (fbits cleared by the host)
__global__ __launch_bounds__(256) void k_mark_frontier(const I *__restrict__ frontier, unsigned fsize,
                                                       unsigned *__restrict__ fbits, unsigned *__restrict__ lpos) {
  for (unsigned j = blockIdx.x * blockDim.x + threadIdx.x; j < fsize; j += gridDim.x * blockDim.x) {
    const I v = frontier[j];
    atomicOr(&fbits[v >> 5], 1u << (v & 31));
    lpos[v] = j;
  }
}

// deepest level: vertex of strictly smallest degree, first in queue order (:64-75)
__global__ __launch_bounds__(256) void k_pick_root(const X *__restrict__ rp, const I *__restrict__ level,
                                                   unsigned lsize, RcmDev *__restrict__ dv) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned long long best = ~0ull;
  for (; j < lsize; j += stride) {
    const I w = level[j];
    const unsigned long long k = ((unsigned long long)(unsigned)(rp[w + 1] - rp[w]) << 32) | (unsigned)j;
    best = k < best ? k : best;
  }
  best = sbx_wave_min(best);
  if (sbx_lane() == 0 && best != ~0ull) atomicMin(&dv->best, best);
}

__global__ void k_set_root_from_best(const I *__restrict__ level, RcmDev *__restrict__ dv) {
  dv->root = (unsigned)level[(unsigned)dv->best];
}

__global__ __launch_bounds__(256) void k_reset_visited(const I *__restrict__ q, unsigned cnt, unsigned *ppos) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; j < cnt; j += stride) ppos[q[j]] = UNSEEN;
}

// (size, base) of the listed components, for the host loop
__global__ __launch_bounds__(256) void k_comp_info(const I *__restrict__ roots, unsigned cnt, const I *__restrict__ csize,
                                                   const I *__restrict__ cbase, I *__restrict__ sizes,
                                                   I *__restrict__ bases) {
  const unsigned c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cnt) return;
  sizes[c] = csize[roots[c]];
  bases[c] = cbase[roots[c]];
}

__global__ __launch_bounds__(256) void k_write_component(const I *__restrict__ q, unsigned cnt, I base,
                                                         X *__restrict__ inv, const I *__restrict__ base_dev) {
  if (base_dev) base = *base_dev;  // (the component's place, still on the device: the size scan's entry of its root)
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; j < cnt; j += stride) inv[q[j]] = base + (I)(cnt - 1 - j);
}

// SBX_RCM_COUNT_SORT=0: levels above 4096 vertices are ordered by the generic radix sort over (parent position, low field)
static bool rcm_count_sort() {
  static const bool on = !(sbx_env_test("SBX_RCM_COUNT_SORT") && atoi(sbx_env_test("SBX_RCM_COUNT_SORT")) == 0);
  return on;
}

constexpr int64_t RCM_COUNT_SORT_MAX = (int64_t)1 << 20;  // levels above this: the generic sort's staged stores win

static bool rcm_ranked_keys() {  // SBX_RCM_RANKED_KEYS=0: big Cuthill-McKee levels sort their full (parent position, rank) keys
  static const bool on = !(sbx_env_test("SBX_RCM_RANKED_KEYS") && atoi(sbx_env_test("SBX_RCM_RANKED_KEYS")) == 0);
  return on;
}

static int rcm_ranked_div() {  // ... for levels of at least n_ranked / this many vertices (SBX_DEBUG_RCM_RANKED_DIV)
  static const int v = sbx_env_test("SBX_DEBUG_RCM_RANKED_DIV") ? atoi(sbx_env_test("SBX_DEBUG_RCM_RANKED_DIV")) : 3;
  return v > 0 ? v : 1;
}

struct BfsBuffers {
  bool *claim_clean;  // the claim bytes of the unordered sweeps are all zero (a finished sweep leaves them that way)
  const X *rp, *col;
  unsigned *vbits, *fbits, *lpos, *ppos;
  const I *label;
  int64_t nnz;
  I *q;         // visiting order of the current BFS (levels concatenated)
  I *nf_list;   // unordered next frontier
  uint64_t *heavy;
  uint64_t heavy_cap;  // descriptors the queue holds
  uint2 *hub_dir;  // per workgroup of k_bfs_expand: (first chunk descriptor, count)
  uint64_t *ka, *kb;
  unsigned long long *fresh64;  // per 64 vertices: the bits the current level added to the visited bitmap
  int *wcnt, *woff;             // ... their popcounts, and the totals of every RCM_FW_WORDS of them
  unsigned *cs_scratch;         // the level ordering's counting sort (sbx_cs::scratch_words(n) words)
  const uint32_t *drank, *dorder;
  int64_t n_ranked;  // entries of dorder (vertices with a non-empty row)
  RcmDev *dv;
  int64_t n;
  unsigned max_deg;  // largest degree of the graph: no hub kernel launches when nothing exceeds RCM_LIGHT
  // the call's first read-back rides on the first sweep's (saves a round trip at the head of every call): when set, the
  // first read-back of run_ubfs also delivers the degree counts k_deg_reduce left (max_deg above is not known before)
  RcmDev *late_counts;
  bool *late_counts_ready;
  // work the host still has to enqueue elsewhere (the degree ranks on their side stream): run once, right after a sweep's
  // first launch and before the host waits for it, so that its enqueue time hides behind the kernel (bfs_first_launch)
  std::function<int()> *after_first_launch;
  // the hook enqueues a dozen launches or more (~5 us of host time each): it is only run where the GPU has that much
  // work in front of the next read-back — a chain of big levels (run_ubfs) or an ordered sweep's launches — and not
  // behind a 40 us kernel, which it would outlast with the caller's stream idle
  bool hook_wants_long_cover;
  // an unordered sweep enqueues its first chain of big levels right behind its head small-level run (run_ubfs): set for
  // the call's first sweep only — it starts at the first non-empty row, the later ones at a vertex of smallest degree in
  // a deepest level, whose first big frontier is a thin one for the top-down kernels (the chain's launches would all
  // leave at once, ~35 us of empty launches; measured on the bench matrix, NOTES section 4.5-r6)
  bool head_chain;
  // the tie-break in front of this sweep was left unverified (ubfs_pick_root): the sweep's first read-back says whether
  // the walk named the root (tie_done) — if not, the sweep's first kernels have left (RcmDev::spec_guard), the sweep
  // returns with *spec_failed set and the caller lets the persistent cone kernels finish the tie-break
  bool *tie_unverified;
  // the labelling of the other components runs on a side stream (sbx_rcm_reorder); its counters live in *dv.  Once both
  // of its halves are enqueued (*side_stage >= 2) a Cuthill-McKee sweep joins that stream at its start — the work
  // finished long before — so that the sweep's own last read-back of *dv also delivers those counters (*last_read,
  // *last_read_joined) and the call needs no read-back of its own for them
  const int *side_stage;
  hipEvent_t side_event;
  bool *side_joined;
  RcmDev *last_read;
  bool *last_read_joined;
};

// cover_us: roughly how long the kernels just enqueued keep the GPU busy before the host's next read-back returns
static int bfs_first_launch(const BfsBuffers &b, int cover_us = 1000) {  // (the hook ignores every call after its last stage)
  if (!b.after_first_launch) return SBX_OK;
  if (b.hook_wants_long_cover && cover_us < 100) return SBX_OK;
  return (*b.after_first_launch)();
}

struct BfsResult {
  unsigned count;        // vertices reached
  unsigned levels;       // number of levels (eccentricity + 1)
  unsigned last_offset;  // offset of the deepest level in q
  unsigned last_size;
};

// One ordered BFS over the component containing the root (fixed_root >= 0, or the
// device-resident dv->root).  CM selects Cuthill-McKee child order.  Direction per
// level: top-down (frontier pushes) unless the frontier is large and owns more than
// four times as many edges as the still-unvisited remainder — then bottom-up (pull).
template <bool CM>
int run_bfs(sbx_handle_t h, const BfsBuffers &b, I fixed_root, I comp_label, BfsResult *out, bool *spec_failed = nullptr) {
  if (spec_failed) *spec_failed = false;
  const size_t bm_bytes = (size_t)((b.n + 31) / 32) * sizeof(unsigned);
  // vbits and fbits are adjacent pieces of one allocation (sbx_rcm_reorder): one fill clears both
  // (also the bitmap the level ordering scatters into: k_rank_scatter)
  const StartClear sc = {b.vbits, (unsigned long long)((b.fbits - b.vbits) + bm_bytes / sizeof(unsigned)),
                         (unsigned *)b.fresh64, (unsigned long long)(2 * ((b.n + 63) / 64 + 1)), nullptr, 0};
  if (CM && b.side_stage && *b.side_stage >= 2 && !*b.side_joined) {
    SBX_HIP(h, hipStreamWaitEvent(h->stream, b.side_event, 0));
    *b.side_joined = true;
  }
  SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_bfs_start, dim3(RCM_START_GRID), dim3(256), b.rp, b.vbits, b.fbits, b.lpos, b.ppos, b.q,
              b.dv, fixed_root, sc);
  unsigned off = 0, fsize = 1, level = 0, total = 1;
  const unsigned max_grid = (unsigned)h->num_cus * 8;
  // the hub kernel runs as exactly one resident wave of workgroups: a partial second wave (x8 on a kernel that
  // fits 5 per CU) left a tail that cost 15 %
  static int heavy_per_cu = 0;
  if (heavy_per_cu == 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_bfs_expand_heavy<0>, 256, 0) != hipSuccess || nb < 1) nb = 4;
    heavy_per_cu = nb;
  }
  const unsigned heavy_grid = (unsigned)h->num_cus * (unsigned)heavy_per_cu;
  int64_t remaining = b.nnz;      // adjacency entries owned by vertices not yet in any level
  int64_t frontier_edges = -1;    // degree sum of the current frontier (-1: level 0, read lazily)
  bool try_small = true;  // false right after the small-level kernel declined this very frontier
  bool frontier_unmarked = false;  // the current frontier was published by the small-level kernel (no fbits/lpos)
  while (true) {
    bool expanded_by_small = false;
    RcmDev hd;
    if (try_small && fsize <= (unsigned)SL_MAXF) {
      SBX_KLAUNCH(h, SBX_K_BFS_SMALL, (k_bfs_small_levels<CM>), dim3(1), dim3(1024), b.rp, b.col, b.q, b.vbits, b.fbits,
                  b.lpos, b.ppos, b.nf_list, b.drank, b.dorder, off, fsize, level, total, b.dv);
      SBX_LAUNCH_CHECK(h);
      SBX_TRY(bfs_first_launch(b));
      SBX_TRY(sbx_readback(h, &hd, b.dv, sizeof(RcmDev)));
      if (b.tie_unverified && *b.tie_unverified) {  // (the sweep's first read-back: see BfsBuffers::tie_unverified)
        *b.tie_unverified = false;
        if (!hd.tie_done) {
          if (!spec_failed) SBX_FAIL(h, SBX_ERR_INTERNAL, "run_bfs: an unverified tie-break in front of a sweep that cannot be redone");
          *spec_failed = true;
          return SBX_OK;
        }
      }
      if (b.last_read) *b.last_read = hd, *b.last_read_joined = b.side_joined && *b.side_joined;
      remaining -= (int64_t)hd.sl_edges;
      if (remaining < 0) remaining = 0;
      off = hd.sl_off;
      fsize = hd.sl_fsize;
      level = hd.sl_level;
      total = hd.sl_total;
      if (hd.sl_status == SL_DONE) break;
      if (hd.sl_status == SL_STOP_READY) {
        frontier_edges = (int64_t)hd.fedges;
        try_small = false;  // expand this frontier with the general kernels
        frontier_unmarked = true;
        continue;
      }
      expanded_by_small = true;  // SL_STOP_EXPANDED: nf_list / nf / fedges are ready for ordering
    }
    try_small = true;
    if (!expanded_by_small) {
    // (frontier_unmarked is consumed by the expansion right below)
    const bool bottom_up = frontier_edges >= 0 && fsize >= 8192 && (double)frontier_edges > bu_ratio() * (double)remaining;
    if (bottom_up) {
      if (frontier_unmarked) {
        SBX_HIP(h, hipMemsetAsync(b.fbits, 0, bm_bytes, h->stream));
        SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_mark_frontier, dim3(sbx_grid_for(fsize, 256, 1024)), dim3(256),
                    (const I *)(b.q + off), fsize, b.fbits, b.lpos);
      }
      SBX_KLAUNCH(h, SBX_K_BFS_BOTTOMUP, k_bfs_bottom_up, dim3(max_grid), dim3(256), b.rp, b.col, b.label, comp_label,
                  (const unsigned *)b.vbits, (const unsigned *)b.fbits, (const unsigned *)b.lpos, b.ppos, b.nf_list,
                  b.n, b.dv, b.heavy, b.heavy_cap);
      if (b.max_deg > (unsigned)RCM_BU_HEAVY)
        SBX_KLAUNCH(h, SBX_K_BFS_BOTTOMUP, k_bfs_bottom_up_heavy, dim3(8 * (unsigned)h->num_cus), dim3(256), b.rp, b.col,
                    (const unsigned *)b.fbits, (const unsigned *)b.lpos, b.ppos, b.nf_list, b.dv,
                    (const uint64_t *)b.heavy, b.heavy_cap);
    } else {
      const unsigned waves_needed = (fsize + RCM_VPW - 1) / RCM_VPW;
      unsigned grid = (waves_needed + 3) / 4;
      if (grid > max_grid) grid = max_grid;
      if (grid < 1) grid = 1;
      const UnorderedSweep none = {nullptr, nullptr, nullptr, 0u};
      // A wide frontier with hubs: the light rows' kernel — tens of thousands of short dependent chains, the GPU mostly
      // idle — used to run in front of the hub kernel because it also queues the hubs' chunks.  Now a first launch only
      // queues (part 1: ~10 us), the hub kernel follows at once and the light rows (part 2) run beside it on a side
      // stream: 85 us off the widest level of the bench matrix's Cuthill-McKee sweep.
      const bool split = b.max_deg > (unsigned)RCM_LIGHT && fsize >= 4096 && !h->prof_on && rcm_split_expand() && h->aux_ready;
      if (split) {
        hipStream_t main_s = h->stream;
        const bool was_dirty = h->aux_dirty;
        SBX_KLAUNCH(h, SBX_K_BFS_EXPAND, k_bfs_expand<0>, dim3(grid), dim3(256), b.rp, b.col, (const I *)(b.q + off),
                    fsize, level + 1, (const unsigned *)b.vbits, b.ppos, b.nf_list, b.heavy, b.hub_dir, b.dv, none, 1);
        SBX_HIP(h, hipEventRecord(h->aux_event[4], main_s));
        SBX_HIP(h, hipStreamWaitEvent(h->aux_stream[2], h->aux_event[4], 0));
        h->aux_dirty = true;
        h->stream = h->aux_stream[2];
        hipLaunchKernelGGL(k_bfs_expand<0>, dim3(grid), dim3(256), 0, h->stream, b.rp, b.col, (const I *)(b.q + off), fsize,
                           level + 1, (const unsigned *)b.vbits, b.ppos, b.nf_list, b.heavy, b.hub_dir, b.dv, none, 2);
        const hipError_t e1 = hipGetLastError(), e2 = hipEventRecord(h->aux_event[5], h->stream);
        h->stream = main_s;
        if (e1 != hipSuccess || e2 != hipSuccess) SBX_FAIL(h, SBX_ERR_HIP, "run_bfs: side launch failed");
        SBX_KLAUNCH(h, SBX_K_BFS_HEAVY, k_bfs_expand_heavy<0>, dim3(heavy_grid), dim3(256), b.rp, b.col,
                    (const I *)(b.q + off), level + 1, (const unsigned *)b.vbits, b.ppos, b.nf_list,
                    (const uint64_t *)b.heavy, (const uint2 *)b.hub_dir, grid, b.dv, none);
        SBX_HIP(h, hipStreamWaitEvent(main_s, h->aux_event[5], 0));
        h->aux_dirty = was_dirty;
      } else {
      SBX_KLAUNCH(h, SBX_K_BFS_EXPAND, k_bfs_expand<0>, dim3(grid), dim3(256), b.rp, b.col, (const I *)(b.q + off),
                  fsize, level + 1, (const unsigned *)b.vbits, b.ppos, b.nf_list, b.heavy, b.hub_dir, b.dv, none, 0);
      if (b.max_deg > (unsigned)RCM_LIGHT)  // mesh-like inputs have no hubs: one launch less per level
        SBX_KLAUNCH(h, SBX_K_BFS_HEAVY, k_bfs_expand_heavy<0>, dim3(heavy_grid), dim3(256), b.rp, b.col,
                    (const I *)(b.q + off), level + 1, (const unsigned *)b.vbits, b.ppos, b.nf_list,
                    (const uint64_t *)b.heavy, (const uint2 *)b.hub_dir, grid, b.dv, none);
      }
    }
    SBX_LAUNCH_CHECK(h);
    SBX_TRY(bfs_first_launch(b));
    SBX_TRY(sbx_readback(h, &hd, b.dv, sizeof(RcmDev)));
    if (b.last_read) *b.last_read = hd, *b.last_read_joined = b.side_joined && *b.side_joined;
    }
    frontier_unmarked = false;
    const unsigned nf = hd.nf;
    static const bool trace_levels = sbx_env_tuning("SBX_DEBUG_RCM_LEVELS") && atoi(sbx_env_tuning("SBX_DEBUG_RCM_LEVELS")) != 0;
    if (trace_levels)
      fprintf(stderr, "[rcm %s] level %u: frontier %u vertices / %lld edges, unvisited edges %lld -> %u new vertices / %llu edges\n",
              CM ? "cm" : "plain", level, fsize, (long long)frontier_edges, (long long)remaining, nf,
              (unsigned long long)hd.fedges);
    if (nf == 0) break;
    if (frontier_edges < 0) remaining -= (int64_t)0;  // level 0's degree is part of hd.fedges history below
    frontier_edges = (int64_t)hd.fedges;              // degree sum of the level just discovered
    remaining -= frontier_edges;
    if (remaining < 0) remaining = 0;
    I *q_next = b.q + off + fsize;
    // direction of the NEXT expansion is already decidable: the frontier marks (bitmap +
    // level positions) are only written when it will be bottom-up
    const bool next_bottom_up = nf >= 8192 && (double)frontier_edges > bu_ratio() * (double)remaining;
    const int mark_frontier = next_bottom_up ? 1 : 0;
    // levels wide enough for the bitmap pass (k_fresh_words / k_visited_from_ppos) get every word of fbits rewritten
    const bool bitmap_pass = nf > (unsigned)RCM_LDS_SORT && (int64_t)nf >= std::max<int64_t>(RCM_REBUILD_BITS, b.n / 128);
    if (mark_frontier && !bitmap_pass) SBX_HIP(h, hipMemsetAsync(b.fbits, 0, bm_bytes, h->stream));
    if (nf <= RCM_LDS_SORT) {
      SBX_KLAUNCH(h, SBX_K_LEVEL_ORDER, (k_level_sort_small<CM>), dim3(1), dim3(1024), (const I *)b.nf_list, nf,
                  (const unsigned *)b.ppos, b.drank, b.dorder, q_next, b.vbits, b.fbits, b.lpos, mark_frontier, b.dv);
    } else {
      const unsigned g = sbx_grid_for(nf, 256, 4096);
      // one pass over ppos (n words) against one single-bit atomic per vertex of the level
      const int set_bits = bitmap_pass ? 0 : 1;
      sbx_radix_pass passes[16];
      int np;
      int low_bits = 32;  // width of the key's low field (vertex id or degree rank); 32: the parent position starts at bit 32
      if (rcm_count_sort() && (int64_t)nf <= RCM_COUNT_SORT_MAX) {
        // The level's vertices in the order of their low key field — ascending degree rank (Cuthill-McKee) or id
        // (plain) — read off a bitmap, so that only the parent positions are left to sort, stably, by the counting
        // sort of sbx_countsort.h: three small launches per 9-bit digit.  The bitmap comes from a pass over all
        // (ranked) vertices when the level holds a good part of them, from the level's own list otherwise.
        const int64_t span = CM ? b.n_ranked : b.n;
        const int64_t words = (span + 63) / 64;
        const int64_t fw_blocks = (words + RCM_FW_WORDS - 1) / RCM_FW_WORDS;
        if (!CM && !set_bits) {
          SBX_KLAUNCH(h, SBX_K_LEVEL_ORDER, k_fresh_words, dim3((unsigned)fw_blocks), dim3(256), (const unsigned *)b.ppos,
                      (unsigned long long *)b.vbits,
                      mark_frontier ? (unsigned long long *)b.fbits : (unsigned long long *)nullptr, b.fresh64, b.wcnt,
                      b.woff, b.n);
        } else if (CM && (int64_t)nf * 16 >= b.n_ranked) {  // (a scattered bit costs ~16 gathered ones)
          SBX_KLAUNCH(h, SBX_K_LEVEL_ORDER, k_fresh_words_ranked, dim3((unsigned)fw_blocks), dim3(256),
                      (const unsigned *)b.ppos, (const unsigned *)b.vbits, b.dorder, b.fresh64, b.wcnt, b.woff, b.n_ranked);
        } else {
          SBX_KLAUNCH(h, SBX_K_LEVEL_ORDER, k_rank_scatter, dim3(g), dim3(256), (const I *)b.nf_list, nf,
                      CM ? b.drank : (const uint32_t *)nullptr, b.fresh64);
          SBX_KLAUNCH(h, SBX_K_LEVEL_ORDER, k_rank_counts, dim3((unsigned)fw_blocks), dim3(64),
                      (const unsigned long long *)b.fresh64, b.wcnt, b.woff, words);
        }
        const int scanned = fw_blocks > RCM_FW_INLINE ? 1 : 0;
        if (scanned) SBX_TRY(sbx_exclusive_scan_i32(h, b.woff, b.woff, fw_blocks, nullptr));
        uint32_t *k32 = (uint32_t *)b.ka, *v32 = k32 + b.n, *k32b = (uint32_t *)b.kb, *v32b = k32b + b.n;
        SBX_KLAUNCH(h, SBX_K_LEVEL_ORDER, k_keys_from_fresh, dim3((unsigned)fw_blocks), dim3(256), b.fresh64,
                    (const int *)b.wcnt, (const int *)b.woff, scanned, (const unsigned *)b.ppos,
                    CM ? b.dorder : (const uint32_t *)nullptr, (uint64_t *)nullptr, k32, v32, words, b.dv);
        const LevelEmit em = {CM ? b.dorder : nullptr, q_next, set_bits ? b.vbits : nullptr,
                              (set_bits && mark_frontier) ? b.fbits : nullptr, mark_frontier ? b.lpos : nullptr};
        SBX_TRY(sbx_cs::sort_emit(h, SBX_K_LEVEL_ORDER, k32, v32, k32b, v32b, (int64_t)nf,
                                  sbx_bits_for((uint64_t)(fsize - 1)), em, b.cs_scratch));
        np = -1;  // ordered
      } else if (!CM && !set_bits) {
        // keys in ascending id order straight from the bitmap pass: only the parent positions are left to sort
        const int64_t words = (b.n + 63) / 64;
        const int64_t fw_blocks = (words + RCM_FW_WORDS - 1) / RCM_FW_WORDS;
        SBX_KLAUNCH(h, SBX_K_LEVEL_ORDER, k_fresh_words, dim3((unsigned)fw_blocks), dim3(256), (const unsigned *)b.ppos,
                    (unsigned long long *)b.vbits,
                    mark_frontier ? (unsigned long long *)b.fbits : (unsigned long long *)nullptr, b.fresh64, b.wcnt,
                    b.woff, b.n);
        const int scanned = fw_blocks > RCM_FW_INLINE ? 1 : 0;
        if (scanned) SBX_TRY(sbx_exclusive_scan_i32(h, b.woff, b.woff, fw_blocks, nullptr));
        SBX_KLAUNCH(h, SBX_K_LEVEL_ORDER, k_keys_from_fresh, dim3((unsigned)fw_blocks), dim3(256),
                    b.fresh64, (const int *)b.wcnt, (const int *)b.woff, scanned,
                    (const unsigned *)b.ppos, (const uint32_t *)nullptr, b.ka, (uint32_t *)nullptr, (uint32_t *)nullptr, words,
                    b.dv);
        np = sbx_radix_plan(0, 0, 32, 32 + sbx_bits_for((uint64_t)(fsize - 1)), passes);
      } else if (CM && !set_bits && rcm_ranked_keys() && (int64_t)nf * rcm_ranked_div() >= b.n_ranked) {
        // keys in ascending degree-rank order from a bitmap in rank space: only the parent positions are left to sort
        const int64_t words = (b.n_ranked + 63) / 64;
        const int64_t fw_blocks = (words + RCM_FW_WORDS - 1) / RCM_FW_WORDS;
        SBX_KLAUNCH(h, SBX_K_LEVEL_ORDER, k_fresh_words_ranked, dim3((unsigned)fw_blocks), dim3(256),
                    (const unsigned *)b.ppos, (const unsigned *)b.vbits, b.dorder, b.fresh64, b.wcnt, b.woff, b.n_ranked);
        const int scanned = fw_blocks > RCM_FW_INLINE ? 1 : 0;
        if (scanned) SBX_TRY(sbx_exclusive_scan_i32(h, b.woff, b.woff, fw_blocks, nullptr));
        SBX_KLAUNCH(h, SBX_K_LEVEL_ORDER, k_keys_from_fresh, dim3((unsigned)fw_blocks), dim3(256),
                    b.fresh64, (const int *)b.wcnt, (const int *)b.woff, scanned,
                    (const unsigned *)b.ppos, b.dorder, b.ka, (uint32_t *)nullptr, (uint32_t *)nullptr, words, b.dv);
        np = sbx_radix_plan(0, 0, 32, 32 + sbx_bits_for((uint64_t)(fsize - 1)), passes);
      } else {
        low_bits = sbx_bits_for((uint64_t)(b.n - 1));
        SBX_KLAUNCH(h, SBX_K_LEVEL_ORDER, (k_level_keys<CM>), dim3(g), dim3(256), (const I *)b.nf_list, nf,
                    (const unsigned *)b.ppos, b.drank, b.ka, b.dv, low_bits);
        np = sbx_radix_plan(0, low_bits + sbx_bits_for((uint64_t)(fsize - 1)), 0, 0, passes);
      }
      if (np < 0) {
      } else if (np > 0) {
        // the last digit pass writes the queue, the bits and the level positions itself (no pass over sorted keys)
        const sbx_radix_emit em = {CM ? b.dorder : nullptr, (uint32_t *)q_next, set_bits ? b.vbits : nullptr,
                                   (set_bits && mark_frontier) ? b.fbits : nullptr, mark_frontier ? b.lpos : nullptr,
                                   low_bits < 32 ? (1u << low_bits) - 1u : 0u};
        SBX_TRY(sbx_radix_sort_emit(h, b.ka, b.kb, nf, passes, np, &em));
      } else {  // one parent and keys already in id order: nothing to sort
        SBX_KLAUNCH(h, SBX_K_LEVEL_ORDER, (k_level_emit<CM>), dim3(g), dim3(256), (const uint64_t *)b.ka, nf, b.dorder,
                    q_next, b.vbits, b.fbits, b.lpos, mark_frontier, set_bits, b.dv);
      }
      if (CM && !set_bits)
        SBX_KLAUNCH(h, SBX_K_LEVEL_ORDER, k_visited_from_ppos, dim3(sbx_grid_for(b.n, 256 * 4, 4096)), dim3(256),
                    (const unsigned *)b.ppos, (unsigned long long *)b.vbits,
                    mark_frontier ? (unsigned long long *)b.fbits : (unsigned long long *)nullptr, b.n);
    }
    SBX_LAUNCH_CHECK(h);
    off += fsize;
    fsize = nf;
    total += nf;
    level++;
  }
  out->count = total;
  out->levels = level + 1;
  out->last_offset = off;
  out->last_size = fsize;
  return SBX_OK;
}

// ---- unordered sweeps ---------------------------------------------------------------------------------------
// A sweep of the pseudo-peripheral search is only asked three things: how many levels, how many vertices, and which
// vertex of the deepest level comes FIRST in queue order among those of smallest degree (rcm_reorder.cc:58-76).  The
// first two do not depend on the order inside a level at all, so these sweeps run as a plain direction-optimising BFS
// over level SETS: an edge that reaches an unvisited vertex sets its byte in `claim8` (a plain store: any parent will
// do, so there is nothing to win and no atomic), the bottom-up kernel stops at the first frontier neighbour, and the
// level — bitmaps, distances, the list in ascending id order, its size and degree sum — is collected from the bytes
// by one streaming kernel over n (k_ubfs_collect).  Nothing is sorted, no parent position is kept.
// The third is settled afterwards, exactly, without knowing any position: let T_L be the candidates (deepest level,
// smallest degree) and T_{k-1} the level-(k-1) neighbours of T_k.  A vertex's queue position is ordered by (position
// of its first parent, id); all parents of T_k lie in T_{k-1}, and every member of T_{k-1} is a parent of some member
// of T_k — so the first member of T_k in queue order is the smallest id among the members of T_k adjacent to the first
// member of T_{k-1}.  Marking the T_k upwards (the expansion kernels in their cone mode) and walking root -> w_1 -> ...
// -> w_L downwards (k_ubfs_descend_step) costs a few adjacency scans where the ordered sweep sorted every level.
// Deep, narrow graphs (a sweep of more than 64 levels) keep the ordered sweep, whose small levels run in one
// persistent workgroup.
static unsigned ub_max_levels() {  // SBX_DEBUG_UB_MAX_LEVELS: deeper sweeps fall back to the ordered kind (tests raise it)
  static const unsigned v = sbx_env_test("SBX_DEBUG_UB_MAX_LEVELS") ? (unsigned)atoll(sbx_env_test("SBX_DEBUG_UB_MAX_LEVELS")) : 64u;
  return v;
}

// an unordered bottom-up step stops at a vertex's first frontier neighbour, so it pays off much earlier than the ordered
// one, which must see every neighbour: bottom-up when the frontier owns more than this many times the unvisited edges
static double ubu_ratio() {
  static const double r = sbx_env_tuning("SBX_DEBUG_UBU_RATIO") ? atof(sbx_env_tuning("SBX_DEBUG_UBU_RATIO")) : 0.5;
  return r;
}

#ifndef SBX_UR_GRID
#define SBX_UR_GRID 64  // (128: +1 %, 256: +4 % on the bench matrix's RCM, measured again in round 6 with tools/build_variant.py)
#endif
constexpr unsigned UR_GRID = SBX_UR_GRID;
constexpr unsigned UR_CONTINUE = 0, UR_DONE = 1, UR_STOP = 2, UR_DEEP = 3;  // how a persistent run ended (DEEP: too many levels in one launch)
constexpr unsigned UR_MAX_E = 1u << 18;   // a frontier owning more adjacency entries than this is the host loop's
constexpr unsigned UR_MAX_N = 1024;       // ... or holding more vertices (a wave takes a vertex: 256 waves)
// what the next kernel of a device-driven chain is to be: nothing more (the sweep is over), a bottom-up level, the
// persistent small-level kernel, or a level the host has to launch
constexpr unsigned UC_DONE = 1, UC_BU = 2, UC_SMALL = 3, UC_HOST = 4;
struct ChainInit {  // what the host knows when it launches the level a chain starts behind
  unsigned off, size, level, total;
  long long remaining;
  double bu_ratio;
};
// Level uc_level + 1 has just been built (its size and degree sum are in the level's slot): the chain's frontier moves
// on to it and the next kernel's kind is decided by the host loop's own rules.  One thread, after the level's kernel
// has finished (its last workgroup out, or k_ubfs_chain_next behind it).
__device__ __forceinline__ void uc_advance(RcmDev *dv, double bu_ratio) {
  const unsigned built = dv->uc_level + 1;
  const unsigned nf = __hip_atomic_load(&dv->unf[built & 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long fe = __hip_atomic_load(&dv->ufedges[built & 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (nf == 0) {  // the frontier just expanded was the deepest level: the chain's state stays on it
    dv->uc_mode = UC_DONE;
    return;
  }
  long long rem = dv->uc_remaining - (long long)fe;
  if (rem < 0) rem = 0;
  dv->uc_remaining = rem;
  dv->uc_off += dv->uc_size;
  dv->uc_size = nf;
  dv->uc_total += nf;
  dv->uc_level = built;
  dv->uc_fe = fe;
  if (fe <= (unsigned long long)UR_MAX_E && nf <= UR_MAX_N) dv->uc_mode = UC_SMALL;
  else if (nf >= 1024 && (double)fe > bu_ratio * (double)rem) dv->uc_mode = UC_BU;
  else dv->uc_mode = UC_HOST;
}
__device__ __forceinline__ void uc_begin(RcmDev *dv, const ChainInit &ci) {
  dv->uc_off = ci.off, dv->uc_size = ci.size, dv->uc_level = ci.level, dv->uc_total = ci.total;
  dv->uc_remaining = ci.remaining;
  dv->uc_flips = 0;
  dv->uc_small_ran = 0;
}
// levels an unordered sweep enqueues behind a big one without waiting for it (run_ubfs); SBX_RCM_UBFS_CHAIN=0: none
static int ubfs_chain() {
  // (3: a sweep of the bench matrix has three bottom-up levels in a row — behind a bottom-up level of the host's the
  // last link finds nothing to do, 5 us; behind a top-down level all three run.  With 2 the second sweep needs another
  // round trip and leaves two links idle)
  static const int k = sbx_env_test("SBX_RCM_UBFS_CHAIN") ? atoi(sbx_env_test("SBX_RCM_UBFS_CHAIN")) : 3;
  return k < 0 ? 0 : (k > 8 ? 8 : k);
}

static bool rcm_unordered() {  // SBX_RCM_UNORDERED=0: every sweep of the search keeps the order inside its levels
  static const bool on = !(sbx_env_test("SBX_RCM_UNORDERED") && atoi(sbx_env_test("SBX_RCM_UNORDERED")) == 0);
  return on;
}

__global__ __launch_bounds__(256) void k_ubfs_start(const X *__restrict__ rp, unsigned *__restrict__ vbits,
                                                    unsigned *__restrict__ fbits, unsigned *__restrict__ dist,
                                                    I *__restrict__ q, RcmDev *__restrict__ dv, I fixed_root,
                                                    unsigned gb_spins, StartClear sc) {
  SBX_SPEC_GUARD_LEAVE(dv);
  if (!start_clear_elect(sc, dv)) return;
  dv->spec_guard = 0;
  const I r = fixed_root >= 0 ? fixed_root : (dv->root_pending ? (I)dv->root_next : (I)dv->root);
  dv->root_pending = 0;
  dv->root = (unsigned)r;
  vbits[r >> 5] = 1u << (r & 31);  // the bitmaps were cleared above
  fbits[r >> 5] = 1u << (r & 31);
  dist[r] = 0;
  q[0] = r;
  dv->nf = 0;
  dv->n_heavy = 0;
  dv->hub_overflow = 0;
  dv->fedges = (unsigned long long)(rp[r + 1] - rp[r]);
  dv->unf[0] = dv->unf[1] = 0;
  dv->ufedges[0] = dv->ufedges[1] = 0;
  dv->ur_bar = 0;
  dv->ur_exit = 0;
  dv->uc_done = 0;
  dv->uc_mode = 0;
  for (int i = 0; i < 3; i++) dv->ur_nf[i] = 0, dv->ur_nh[i] = 0, dv->ur_deg[i] = 0;
  dv->gb_spins = gb_spins;
}

// Level collection in one launch (64 bitmap words = 4096 vertices per workgroup): claimed bytes -> the frontier word,
// visited |= it, bytes cleared, distances written, and the level's vertices appended to the queue — a workgroup that
// holds any reserves its stretch with one atomic (the order of an unordered level is free; empty workgroups, most of
// them on a small level, touch nothing).  Size and degree sum of level l accumulate in slot l & 1; the other slot is
// cleared for level l + 1.  (Two launches — totals, then a list written at bases summed from them — took 19 us per
// level; a version with 16 K vertices per workgroup to spare the counter word took 24.)
__global__ __launch_bounds__(256) void k_ubfs_collect(unsigned char *__restrict__ claim8,
                                                      unsigned long long *__restrict__ vbits64,
                                                      unsigned long long *__restrict__ fbits64,
                                                      unsigned *__restrict__ dist, unsigned level,
                                                      const X *__restrict__ rp, I *__restrict__ q_next, int64_t n,
                                                      RcmDev *__restrict__ dv) {
  constexpr int WPW = RCM_FW_WORDS / 4;
  __shared__ unsigned s_wtot[4], s_base;
  __shared__ unsigned long long s_deg[4];
  const int lane = sbx_lane(), wv = threadIdx.x >> 6;
  const int slot = (int)(level & 1u);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    dv->unf[slot ^ 1] = 0;
    dv->ufedges[slot ^ 1] = 0;
    dv->n_heavy = 0;
    dv->hub_overflow = 0;
  }
  const int64_t w0 = (int64_t)blockIdx.x * RCM_FW_WORDS + (int64_t)wv * WPW;
  unsigned char cb[WPW];
#pragma unroll
  for (int i = 0; i < WPW; i++) {
    const int64_t v = (w0 + i) * 64 + lane;
    cb[i] = v < n ? claim8[v] : (unsigned char)0;
  }
  unsigned long long now[WPW];
  unsigned long long deg = 0;
  unsigned mine = 0;  // wave-uniform: the wave's vertices in this level
#pragma unroll
  for (int i = 0; i < WPW; i++) {
    const int64_t v = (w0 + i) * 64 + lane;
    now[i] = __ballot(cb[i] != 0);
    if (cb[i]) {
      claim8[v] = 0;
      dist[v] = level;
      deg += (unsigned long long)(rp[v + 1] - rp[v]);
    }
    if ((w0 + i) * 64 < n && lane == 0) {
      fbits64[w0 + i] = now[i];
      if (now[i]) vbits64[w0 + i] |= now[i];
    }
    mine += (unsigned)__popcll(now[i]);
  }
  deg = sbx_wave_sum(deg);
  if (lane == 0) {
    s_wtot[wv] = mine;
    s_deg[wv] = deg;
  }
  __syncthreads();
  const unsigned tot = s_wtot[0] + s_wtot[1] + s_wtot[2] + s_wtot[3];
  if (tot == 0) return;
  if (threadIdx.x == 0) {
    s_base = atomicAdd(&dv->unf[slot], tot);
    atomicAdd(&dv->ufedges[slot], s_deg[0] + s_deg[1] + s_deg[2] + s_deg[3]);
  }
  __syncthreads();
  unsigned o = s_base;
  for (int i = 0; i < wv; i++) o += s_wtot[i];
#pragma unroll
  for (int i = 0; i < WPW; i++) {
    if ((now[i] >> lane) & 1ull) q_next[o + (unsigned)__popcll(now[i] & sbx_lanemask_lt())] = (I)((w0 + i) * 64 + lane);
    o += (unsigned)__popcll(now[i]);
  }
}

// bottom-up without positions: an unvisited vertex joins the level at its FIRST neighbour in the frontier
// candidates up to this degree get one lane each: with the early exit a lane rarely reads more than its first batch of
// four entries, where a 16-lane group fetches 64 at a time (16 -> 1024: 33 M -> 12 M entries scanned bottom-up on the
// bench matrix, the kernel 0.64 -> 0.36 ms per RCM; beyond 1024 nothing changes)
constexpr int UB_INLINE = 1024;
__global__ __launch_bounds__(256) void k_ubfs_bottom_up(const X *__restrict__ rp, const X *__restrict__ col,
                                                        const I *__restrict__ label, I comp_label,
                                                        unsigned *vbits, const unsigned *__restrict__ fbits,
                                                        unsigned *__restrict__ nbits, unsigned *__restrict__ dist,
                                                        unsigned level, I *__restrict__ nf_list, int64_t n,
                                                        RcmDev *dv, int chain, ChainInit ci) {
  // chain 1: a link of a device-driven chain (run_ubfs) — nf_list is the queue's base, level and place come from the
  // chain, and the link runs only if the chain asks for a bottom-up level; chain 2: the level the host launched in front
  // of a chain (its state in ci).  Either way the last workgroup out moves the chain on (uc_advance).
  if (chain == 1) {
    if (dv->uc_mode != UC_BU) return;
    level = dv->uc_level + 1;
    nf_list += dv->uc_off + dv->uc_size;
  }
  __shared__ I s_stage[4][RCM_STAGE];
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int lane = sbx_lane();
  const int grp = lane / RCM_GROUP, gl = lane % RCM_GROUP;
  const int slot = (int)(level & 1u);
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // (as k_ubfs_collect: the other level slot is cleared for the next level)
    dv->unf[slot ^ 1] = 0;
    dv->ufedges[slot ^ 1] = 0;
    dv->n_heavy = 0;
    dv->hub_overflow = 0;
  }
  WaveStage st{s_stage[sbx_wave_in_block()], 0u, 0ull, &dv->unf[slot], &dv->ufedges[slot]};
  unsigned long long scanned = 0;
  for (int64_t base = wave * 64; base < n; base += nwaves * 64) {
    const int64_t v = base + lane;
    bool cand = false;
    I s = 0, e = 0;
    if (v < n && !((vbits[v >> 5] >> (v & 31)) & 1u)) {
      s = rp[v];
      e = rp[v + 1];
      cand = (e > s) && (label == nullptr || label[v] == comp_label);
    }
    const bool small = cand && (e - s) <= UB_INLINE;
    bool found = false;
    if (__any(small)) {
      // four entries at a time, every load of a batch in flight together; a wave stops as soon as every one of its small
      // candidates has either found a frontier neighbour or run out of entries (most have one to four of them)
      const int dg = small ? (int)(e - s) : 0;
      int done = 0;
      for (int k0 = 0; k0 < UB_INLINE; k0 += 4) {
        if (!__any(small && !found && dg > k0)) break;
        I us[4];
#pragma unroll
        for (int k = 0; k < 4; k++) us[k] = (small && !found && k0 + k < dg) ? col[s + k0 + k] : (I)-1;
#pragma unroll
        for (int k = 0; k < 4; k++)
          if (us[k] >= 0) {
            done++;
            if ((fbits[us[k] >> 5] >> (us[k] & 31)) & 1u) found = true;
          }
      }
      scanned += (unsigned)done;
    }
    uint64_t todo = __ballot(cand && !small);
    uint64_t big_found = 0;  // lanes (vertices) of this wave's 64 that a group found a frontier neighbour for
    while (todo) {
      uint64_t t = todo;
      int pick = -1;
      for (int g = 0; g < RCM_VPW; g++) {
        const int c = t ? __builtin_ctzll(t) : -1;
        if (g == grp) pick = c;
        if (t) t &= t - 1;
      }
      todo = t;
      const I cs = __shfl(s, pick < 0 ? 0 : pick, 64), ce = __shfl(e, pick < 0 ? 0 : pick, 64);
      bool hit = false;
      I j = (pick < 0 ? 0 : cs) + gl;
      const I jend = pick < 0 ? 0 : ce;
      unsigned seen = 0;
      while (__any(j < jend)) {
        I us[4];
#pragma unroll
        for (int k = 0; k < 4; k++) us[k] = (j + k * RCM_GROUP) < jend ? col[j + k * RCM_GROUP] : (I)-1;
#pragma unroll
        for (int k = 0; k < 4; k++)
          if (us[k] >= 0) {
            seen++;
            if ((fbits[us[k] >> 5] >> (us[k] & 31)) & 1u) hit = true;
          }
        // a group whose vertex has been found stops (its 16 lanes agree through the row-wide OR)
        const unsigned any_hit = sbx_row16_reduce(hit ? 1u : 0u, SbxOpMax());
        j = any_hit ? jend : j + 4 * RCM_GROUP;
      }
      scanned += seen;
      const unsigned ghit = sbx_row16_reduce(hit ? 1u : 0u, SbxOpMax());
      // the lane that holds a group's vertex is lane `pick` of the wave: collect the groups' results for their owners
#pragma unroll
      for (int g = 0; g < RCM_VPW; g++) {
        const int pk = __shfl(pick, g * RCM_GROUP, 64);
        const unsigned gh = __shfl(ghit, g * RCM_GROUP, 64);
        if (gh && pk >= 0) big_found |= (uint64_t)1 << pk;
      }
    }
    found = found || ((big_found >> lane) & 1ull);
    // lane = vertex and the wave owns the two bitmap words of its 64 vertices: the level is published right here —
    // visited bit (only this wave ever tests it), next frontier word (every word is written, so the array needs no
    // clearing), distance, queue — and needs no collection pass
    const uint64_t fm = __ballot(found);
    if ((lane & 31) == 0 && base + lane < n) {
      const unsigned wbits = (unsigned)(fm >> (lane & 32));
      nbits[(base >> 5) + (lane >> 5)] = wbits;
      if (wbits) vbits[(base >> 5) + (lane >> 5)] |= wbits;
    }
    if (found) dist[v] = level;
    stage_push((I)v, found, rp, st, nf_list, dv);
  }
  stage_end_block(st, nf_list, dv, scanned, true);
  if (chain && threadIdx.x == 0) {
    // This workgroup's additions to the level's counters were returning atomics or thread 0's own: waiting for the
    // latter is all the ordering the election needs (the counters live in L2 and are read there).  A release fence
    // here — a write-back of the L2 per workgroup, 2048 of them — doubled the kernel's time.
    // Words under the invariant (see gb_wait): the level's counters unf[] / ufedges[] / n_heavy (returning atomics of the
    // waves, or thread 0's own adds), uc_done (the election), and what uc_advance / uc_begin write for the next link
    // of the chain (uc_mode, uc_level, uc_off, ...: read by kernels launched behind this one, i.e. behind a kernel boundary).
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    SBX_GB_RELEASE();
    if (atomicAdd(&dv->uc_done, 1u) == gridDim.x - 1) {
      SBX_GB_ACQUIRE();
      dv->uc_done = 0;
      if (chain == 2) uc_begin(dv, ci);
      else dv->uc_flips++;
      uc_advance(dv, ci.bu_ratio);
    }
  }
}

// deepest level: smallest degree, then the vertices that have it — marked in the cone bitmap, listed (the list counter
// is dv->nf), their number and smallest id.  A level of up to UB_TIES_SMALL vertices (the usual case: a handful) is one
// workgroup's job; larger ones take three launches.
// (agent-scope relaxed loads and stores: what the persistent kernels below exchange their words with)
template <typename T>
__device__ __forceinline__ T ur_load(const T *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T>
__device__ __forceinline__ void ur_store(T *p, T v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
constexpr unsigned UB_TIES_SMALL = 8192;

// ---- the whole tie-break in the candidates' workgroup, where the cone is small ------------------------------------
// On the bench matrix the cones ARE small — sweep 1: |T_k| = 6, 6, 6, 8, ~1200 for k = 5 .. 1; sweep 2 (from a vertex of
// degree 1: levels of 1, 1, 1, 1, 850, 622 K, 1.6 M, 43 K, 189, 2 vertices): 2, 2, 2, 2, 42, ~400, 1, 1, 1 — and the
// persistent kernels spent 146 + 212 us per call on them: grid barriers level by level, a hand-over to the expansion
// kernels where T_4's hubs own 2.4 M adjacency entries, the walk down scanning the root's 140 K entries on 64
// workgroups, two or three round trips.  This routine does marking and walk in ONE workgroup, with the direction of
// every step chosen by its cost:
//   * the leading levels that hold ONE vertex (the path out of a peripheral root) are found first, walking down from the
//     root (chain[k]): T_k of such a level is that vertex, whatever T_{k+1} owns, and under such a level every member of
//     T_{k+1} hangs — w_{k+1} is simply the smallest id in T_{k+1}, so of the first level that holds more than one
//     vertex only the SMALLEST member of its T is ever needed (no set, no limit on its size);
//   * T_{k-1} from T_k: the members' adjacency entries are shared out flat over the threads (prefix sums of the degrees,
//     a branch-free binary search per entry, eight entries in flight per thread), members are deduplicated through an
//     LDS hash set;
//   * w_k from w_{k-1}: either w_{k-1}'s adjacency is scanned for members of T_k (the hash set), or the members'
//     adjacency for w_{k-1} — whichever owns fewer entries.
// One workgroup gathers at one CU's rate: a step over 73 K entries (T_5's 42 hubs in the bench matrix's second sweep,
// looking for the smallest level-4 neighbour) took it 124 us.  Such a step — more than TS_SINGLE entries, and of the
// smallest-member kind — is handed to a grid (k_tie_heavy_min, enqueued behind this kernel either way: 6 us for the 73 K
// entries), and k_tie_walk_resume, one workgroup again, takes the walk up from the state left in TieWalkState.
// Anything beyond the limits (TS_CAP members in a level, TS_GATHER entries in an ordinary step, TS_EDGES in a handed-over
// one or a scan of one vertex's entries, TS_LEVELS levels, a second step to hand over) and anything odd (an empty T_{k-1}, no
// candidate under w_{k-1}: a pattern that is not symmetric) makes it LEAVE with nothing changed — the T_k live in the
// scratch behind the candidates' list and in LDS, not in the cone bitmap — and the persistent kernels, which are enqueued
// behind it either way and leave at once when it has named the root (dv->tie_done), take over.
constexpr unsigned TS_CAP = 1024, TS_LEVELS = 64, TS_EDGES = 1u << 17, TS_HASH = 4096;
constexpr unsigned TS_SINGLE = 1u << 13;  // entries of a smallest-member step the one workgroup still scans itself
constexpr unsigned TS_GATHER = 1u << 14;  // entries of an ordinary marking step (visited word + level gathered per entry)
struct TieWalkState {  // between k_ubfs_ties_small, k_tie_heavy_min and k_tie_walk_resume (device memory)
  unsigned req;        // 1: the step k -> k - 1 is the grid's; anything else: nothing to resume
  unsigned k, P, cursor, L, n_ties, begin, count, edges;
  unsigned min_id;     // the grid's result (atomicMin): the smallest member of T_{k-1}, i.e. w_{k-1}
  unsigned next_id;    // ... and w_k: the smallest member of T_k adjacent to it (k_tie_heavy_adj)
  unsigned chain[TS_LEVELS + 2], lvl_begin[TS_LEVELS + 2], lvl_count[TS_LEVELS + 2];
};
struct TieWalkLds {
  unsigned pref[TS_CAP + 1];  // exclusive prefix of the members' degrees
  unsigned mem[TS_CAP];       // the members of the level at hand
  X start[TS_CAP];            // rp[member]
  unsigned hash[TS_HASH];     // open addressing, 0xFFFFFFFF = empty
  unsigned lvl_begin[TS_LEVELS + 2], lvl_count[TS_LEVELS + 2];  // T_k in the scratch list
  unsigned chain[TS_LEVELS + 2];
  unsigned red[16], red2[16];
  unsigned new_count;
};
__device__ __forceinline__ bool tw_visited_at(const unsigned *__restrict__ vbits, const unsigned *__restrict__ dist, unsigned c,
                                              unsigned k) {
  return ((vbits[c >> 5] >> (c & 31)) & 1u) && dist[c] == k;
}
template <int THREADS>
__device__ __forceinline__ unsigned tw_block_min(unsigned v, unsigned *red) {
  v = sbx_wave_min(v);
  if (sbx_lane() == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  unsigned m = 0xFFFFFFFFu;
  for (int i = 0; i < THREADS / 64; i++) m = red[i] < m ? red[i] : m;
  __syncthreads();
  return m;
}
__device__ __forceinline__ unsigned tw_block_max(unsigned v, unsigned *red) {  // (1024 threads)
  v = sbx_wave_max(v);
  if (sbx_lane() == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  unsigned m = 0;
  for (int i = 0; i < 16; i++) m = red[i] > m ? red[i] : m;
  __syncthreads();
  return m;
}
// members of list[begin, begin + count) into LDS with their degree prefix; returns the degree sum (same for all threads)
__device__ __forceinline__ unsigned tw_load_members(TieWalkLds &t, const X *__restrict__ rp, const I *list, unsigned begin,
                                                    unsigned count) {
  unsigned d = 0;
  if (threadIdx.x < count) {
    const unsigned v = (unsigned)ur_load(&list[begin + threadIdx.x]);
    const X s0 = rp[v];
    t.mem[threadIdx.x] = v;
    t.start[threadIdx.x] = s0;
    d = (unsigned)(rp[v + 1] - s0);
  }
  unsigned total;
  const unsigned ex = sbx_block_exclusive_sum<unsigned, 1024>(d, t.red, &total);
  if (threadIdx.x < count) t.pref[threadIdx.x] = ex;
  if (threadIdx.x == 0) t.pref[count] = total;
  __syncthreads();
  return total;
}
__device__ __forceinline__ void tw_hash_clear(TieWalkLds &t) {
  for (unsigned i = threadIdx.x; i < TS_HASH; i += 1024) t.hash[i] = 0xFFFFFFFFu;
  __syncthreads();
}
// true: v was not in the set (a set that is full — the caller stops inserting long before — takes nothing more)
__device__ __forceinline__ bool tw_hash_insert(TieWalkLds &t, unsigned v) {
  unsigned slot = (v * 2654435761u) >> 20;  // 12 bits
  for (unsigned probes = 0; probes < TS_HASH; probes++) {
    const unsigned old = atomicCAS(&t.hash[slot], 0xFFFFFFFFu, v);
    if (old == 0xFFFFFFFFu) return true;
    if (old == v) return false;
    slot = (slot + 1) & (TS_HASH - 1);
  }
  return false;
}
__device__ __forceinline__ bool tw_hash_has(const TieWalkLds &t, unsigned v) {
  unsigned slot = (v * 2654435761u) >> 20;
  for (unsigned probes = 0; probes < TS_HASH; probes++) {
    const unsigned cur = t.hash[slot];
    if (cur == v) return true;
    if (cur == 0xFFFFFFFFu) return false;
    slot = (slot + 1) & (TS_HASH - 1);
  }
  return false;
}
// entry e of the members' concatenated adjacency lists: the index of its member (last j < count with pref[j] <= e).
// (a fixed number of steps and no branch: eight of these per thread are in flight together)
__device__ __forceinline__ unsigned tw_member_of(const unsigned *pref, unsigned count, unsigned e) {
  unsigned lo = 0;
#pragma unroll
  for (unsigned step = TS_CAP / 2; step >= 1; step >>= 1) {
    const unsigned probe = lo + step;
    const unsigned at = probe < count ? probe : 0u;
    lo = (probe < count && pref[at] <= e) ? probe : lo;
  }
  return lo;
}
// phase 0: from the candidates on; phase 1: behind the grid's step (k_tie_walk_resume).  Returns true when the next root
// is in dv->root_next (dv->tie_done = 1).
__device__ bool tie_walk(TieWalkLds &t, const X *__restrict__ rp, const X *__restrict__ col, const unsigned *__restrict__ vbits,
                         const unsigned *__restrict__ dist, I *list, unsigned n_ties, unsigned levels, RcmDev *dv,
                         TieWalkState *st, unsigned max_edges, unsigned max_cap, unsigned max_single, int phase) {
  // (max_edges <= TS_EDGES, max_cap <= TS_CAP: the limits, lowered by the tests so that small graphs reach every exit)
#define TW_LEAVE(code, arg)                                                               \
  do {                                                                                    \
    if (threadIdx.x == 0) dv->tie_walk_exit = (code), dv->tie_walk_arg = (unsigned)(arg); \
    return false;                                                                         \
  } while (0)
  constexpr int TW_U = 8;  // entries in flight per thread
  const unsigned single = max_edges < max_single ? max_edges : max_single;  // (max_single <= TS_SINGLE)
  const unsigned gather = max_edges < TS_GATHER ? max_edges : TS_GATHER;
  const unsigned root = dv->root;
  unsigned L, P, cursor, k_from;
  // (behind the grid's step the walk's step at that level is known as well: the same members' entries, scanned for
  // w_{k-1} by the same grid — on one workgroup that step alone took 35 us)
  unsigned walk_known_level = 0, walk_known = 0xFFFFFFFFu;
  if (phase == 0) {
    L = levels - 1;  // the deepest level
    if (threadIdx.x == 0) st->req = 0;
    if (levels < 2 || L > TS_LEVELS || n_ties > max_cap) TW_LEAVE(1, n_ties);
    // ---- the leading levels of one vertex
    if (threadIdx.x == 0) t.chain[0] = root;
    __syncthreads();
    P = 1;  // levels 0 .. P - 1 hold one vertex each: chain[]
    for (unsigned k = 1; k <= L; k++) {
      const unsigned u = t.chain[k - 1];
      const X s0 = rp[u], e0 = rp[u + 1];
      if ((uint64_t)(e0 - s0) > (uint64_t)single) break;
      unsigned lo = 0xFFFFFFFFu, hi = 0;
      for (X a = s0 + (X)threadIdx.x; a < e0; a += 1024) {
        const unsigned c = (unsigned)col[a];
        if (tw_visited_at(vbits, dist, c, k)) lo = c < lo ? c : lo, hi = c > hi ? c : hi;
      }
      lo = tw_block_min<1024>(lo, t.red);
      hi = tw_block_max(hi, t.red2);
      if (lo == 0xFFFFFFFFu || lo != hi) break;
      if (threadIdx.x == 0) t.chain[k] = lo;
      __syncthreads();
      P = k + 1;
    }
    if (threadIdx.x == 0) t.lvl_begin[L] = 0, t.lvl_count[L] = n_ties;  // T_L = the candidates, list[0, n_ties)
    __syncthreads();
    cursor = n_ties;  // the scratch behind the candidates
    k_from = L;
  } else {
    if (st->req != 1) return false;  // (nothing was handed over: the walk is finished or has left)
    L = st->L, P = st->P, cursor = st->cursor;
    const unsigned k = st->k, m = st->min_id;
    walk_known_level = k, walk_known = st->next_id;
    for (unsigned i = threadIdx.x; i < TS_LEVELS + 2; i += 1024)
      t.chain[i] = st->chain[i], t.lvl_begin[i] = st->lvl_begin[i], t.lvl_count[i] = st->lvl_count[i];
    __syncthreads();
    if (threadIdx.x == 0) st->req = 2;
    if (m == 0xFFFFFFFFu) TW_LEAVE(7 | (k << 8), st->edges);
    if (threadIdx.x == 0) {
      ur_store(&list[cursor], (I)m);
      t.lvl_begin[k - 1] = cursor, t.lvl_count[k - 1] = 1;
    }
    __syncthreads();
    cursor += 1;
    k_from = k - 1;
  }
  // ---- marking: T_{k-1} from T_k, k = L .. 2
  for (unsigned k = k_from; k >= 2; k--) {
    if (k - 1 < P) {
      if (threadIdx.x == 0) {
        ur_store(&list[cursor], (I)t.chain[k - 1]);
        t.lvl_begin[k - 1] = cursor, t.lvl_count[k - 1] = 1;
      }
      __syncthreads();
      cursor += 1;
      continue;
    }
    const unsigned count = t.lvl_count[k];
    const unsigned edges = tw_load_members(t, rp, list, t.lvl_begin[k], count);
    // Under a level of one vertex only the SMALLEST member of T_{k-1} is ever asked for (w_{k-1}; T_{k-2} is that one
    // vertex whatever T_{k-1} holds): no set, no limit on its size
    const bool min_only = k - 2 < P;
    if (min_only && edges > single) {
      // the grid's step: the state goes to device memory, k_tie_heavy_min finds the smallest member, k_tie_walk_resume goes on
      if (phase != 0 || edges > max_edges) TW_LEAVE(2 | (k << 8), edges);
      for (unsigned i = threadIdx.x; i < TS_LEVELS + 2; i += 1024)
        st->chain[i] = t.chain[i], st->lvl_begin[i] = t.lvl_begin[i], st->lvl_count[i] = t.lvl_count[i];
      if (threadIdx.x == 0) {
        st->k = k, st->P = P, st->cursor = cursor, st->L = L, st->n_ties = n_ties;
        st->begin = t.lvl_begin[k], st->count = count, st->edges = edges, st->min_id = 0xFFFFFFFFu;
        st->next_id = 0xFFFFFFFFu;
        st->req = 1;
        dv->tie_walk_exit = 8 | (k << 8), dv->tie_walk_arg = edges;  // (overwritten by whoever finishes or leaves)
      }
      return false;
    }
    if (edges > (min_only ? single : gather)) TW_LEAVE(3 | (k << 8), edges);
    tw_hash_clear(t);
    if (threadIdx.x == 0) t.new_count = 0;
    __syncthreads();
    unsigned my_min = 0xFFFFFFFFu;
    for (unsigned e0 = threadIdx.x; e0 < edges; e0 += 1024 * TW_U) {
      // (every search and every load is issued whatever the entry: under a condition each would be a block of its own and
      // the eight would run one after the other; entries past the end read the last one and are dropped afterwards)
      unsigned c[TW_U], vb[TW_U], dd[TW_U], jj[TW_U];
#pragma unroll
      for (int u = 0; u < TW_U; u++) {
        const unsigned e = e0 + (unsigned)u * 1024u;
        jj[u] = tw_member_of(t.pref, count, e < edges ? e : edges - 1u);
      }
#pragma unroll
      for (int u = 0; u < TW_U; u++) {
        const unsigned e = e0 + (unsigned)u * 1024u, ee = e < edges ? e : edges - 1u;
        c[u] = (unsigned)col[t.start[jj[u]] + (X)(ee - t.pref[jj[u]])];
      }
#pragma unroll
      for (int u = 0; u < TW_U; u++) vb[u] = vbits[c[u] >> 5];
#pragma unroll
      for (int u = 0; u < TW_U; u++) dd[u] = ((vb[u] >> (c[u] & 31)) & 1u) ? dist[c[u]] : 0xFFFFFFFFu;
#pragma unroll
      for (int u = 0; u < TW_U; u++) {
        if (dd[u] != k - 1 || e0 + (unsigned)u * 1024u >= edges) continue;
        if (min_only) {
          my_min = c[u] < my_min ? c[u] : my_min;
        } else if (__hip_atomic_load(&t.new_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= max_cap &&
                   tw_hash_insert(t, c[u])) {
          // (the count is re-read every time: past the cap nothing is inserted any more, the set of 4096 never fills)
          const unsigned pos = atomicAdd(&t.new_count, 1u);
          if (pos < max_cap) ur_store(&list[cursor + pos], (I)c[u]);
        }
      }
    }
    if (min_only) {
      const unsigned m = tw_block_min<1024>(my_min, t.red);
      if (m == 0xFFFFFFFFu) TW_LEAVE(7 | (k << 8), edges);
      if (threadIdx.x == 0) {
        ur_store(&list[cursor], (I)m);
        t.lvl_begin[k - 1] = cursor, t.lvl_count[k - 1] = 1;
      }
      __syncthreads();
      cursor += 1;
      continue;
    }
    __syncthreads();
    const unsigned got = t.new_count;
    if (got == 0 || got > max_cap) TW_LEAVE(4 | (k << 8), got);
    if (threadIdx.x == 0) t.lvl_begin[k - 1] = cursor, t.lvl_count[k - 1] = got;
    __syncthreads();
    cursor += got;
  }
  // ---- the walk down: w_0 = root, w_k = the smallest id among the members of T_k adjacent to w_{k-1}
  unsigned w = root;
  for (unsigned k = 1; k <= L; k++) {
    const unsigned count = t.lvl_count[k], begin = t.lvl_begin[k];
    unsigned best = 0xFFFFFFFFu;
    if (k == walk_known_level && walk_known != 0xFFFFFFFFu) {
      w = walk_known;
      continue;
    }
    if (k - 1 < P) {  // (under a level of one vertex hangs the whole next level)
      if (threadIdx.x < count) best = (unsigned)ur_load(&list[begin + threadIdx.x]);
    } else {
      const unsigned edges = tw_load_members(t, rp, list, begin, count);
      const X ws = rp[w], we = rp[w + 1];
      if ((uint64_t)(we - ws) <= (uint64_t)edges) {
        if ((uint64_t)(we - ws) > (uint64_t)max_edges) TW_LEAVE(5 | (k << 8), we - ws);
        tw_hash_clear(t);
        if (threadIdx.x < count) (void)tw_hash_insert(t, t.mem[threadIdx.x]);
        __syncthreads();
        for (X a0 = ws + (X)threadIdx.x; a0 < we; a0 += (X)1024 * TW_U) {
          unsigned c[TW_U], first[TW_U];
#pragma unroll
          for (int u = 0; u < TW_U; u++) {
            const X a = a0 + (X)u * 1024;
            c[u] = a < we ? (unsigned)col[a] : 0xFFFFFFFEu;  // (no vertex id: it is in no set)
          }
#pragma unroll
          for (int u = 0; u < TW_U; u++) first[u] = t.hash[(c[u] * 2654435761u) >> 20];  // the first probe of all eight
#pragma unroll
          for (int u = 0; u < TW_U; u++) {
            const bool hit = first[u] == c[u] || (first[u] != 0xFFFFFFFFu && tw_hash_has(t, c[u]));
            if (hit) best = c[u] < best ? c[u] : best;
          }
        }
      } else {
        if (edges > max_edges) TW_LEAVE(6 | (k << 8), edges);
        for (unsigned e0 = threadIdx.x; e0 < edges; e0 += 1024 * TW_U) {
          unsigned cc[TW_U], jj[TW_U];
#pragma unroll
          for (int u = 0; u < TW_U; u++) {
            const unsigned e = e0 + (unsigned)u * 1024u;
            jj[u] = tw_member_of(t.pref, count, e < edges ? e : edges - 1u);
          }
#pragma unroll
          for (int u = 0; u < TW_U; u++) {
            const unsigned e = e0 + (unsigned)u * 1024u, ee = e < edges ? e : edges - 1u;
            cc[u] = (unsigned)col[t.start[jj[u]] + (X)(ee - t.pref[jj[u]])];
          }
#pragma unroll
          for (int u = 0; u < TW_U; u++)
            if (cc[u] == w && e0 + (unsigned)u * 1024u < edges) best = t.mem[jj[u]] < best ? t.mem[jj[u]] : best;
        }
      }
    }
    best = tw_block_min<1024>(best, t.red);
    if (best == 0xFFFFFFFFu) TW_LEAVE(9 | (k << 8), count);
    w = best;
  }
  if (threadIdx.x == 0) {
    dv->root_next = w;
    dv->root_pending = 1;
    dv->tie_done = 1;
    dv->tie_walk_exit = 0;
  }
  return true;
#undef TW_LEAVE
}

// the step the one workgroup handed over: the smallest level-(k - 1) neighbour of T_k's members, st->min_id
__global__ __launch_bounds__(256) void k_tie_heavy_min(const X *__restrict__ rp, const X *__restrict__ col,
                                                       const unsigned *__restrict__ vbits, const unsigned *__restrict__ dist,
                                                       const I *list, TieWalkState *st) {
  if (st->req != 1) return;
  __shared__ unsigned s_pref[TS_CAP + 1], s_red[4], s_scan[4];
  __shared__ X s_start[TS_CAP];
  const unsigned count = st->count, begin = st->begin, edges = st->edges, want = st->k - 1;
  unsigned carry = 0;
  for (unsigned base = 0; base < count; base += 256) {  // (count <= TS_CAP: four rounds at most)
    const unsigned i = base + threadIdx.x;
    unsigned d = 0;
    if (i < count) {
      const unsigned v = (unsigned)list[begin + i];
      const X s0 = rp[v];
      s_start[i] = s0;
      d = (unsigned)(rp[v + 1] - s0);
    }
    unsigned total;
    const unsigned ex = sbx_block_exclusive_sum<unsigned, 256>(d, s_scan, &total);
    if (i < count) s_pref[i] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0) s_pref[count] = carry;
  __syncthreads();
  constexpr int U = 4;
  unsigned my_min = 0xFFFFFFFFu;
  const unsigned stride = gridDim.x * 256u;
  for (unsigned e0 = blockIdx.x * 256u + threadIdx.x; e0 < edges; e0 += stride * U) {
    unsigned c[U], vb[U], jj[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const unsigned e = e0 + (unsigned)u * stride;
      jj[u] = tw_member_of(s_pref, count, e < edges ? e : edges - 1u);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const unsigned e = e0 + (unsigned)u * stride, ee = e < edges ? e : edges - 1u;
      c[u] = (unsigned)col[s_start[jj[u]] + (X)(ee - s_pref[jj[u]])];
    }
#pragma unroll
    for (int u = 0; u < U; u++) vb[u] = vbits[c[u] >> 5];
#pragma unroll
    for (int u = 0; u < U; u++)
      if (((vb[u] >> (c[u] & 31)) & 1u) && dist[c[u]] == want && e0 + (unsigned)u * stride < edges)
        my_min = c[u] < my_min ? c[u] : my_min;
  }
  const unsigned m = tw_block_min<256>(my_min, s_red);
  if (threadIdx.x == 0 && m != 0xFFFFFFFFu) atomicMin(&st->min_id, m);
}

static bool rcm_tie_walk() {  // SBX_RCM_TIE_WALK=0: every tie-break through the persistent kernels (tests, A/B)
  static const bool on = !(sbx_env_test("SBX_RCM_TIE_WALK") && atoi(sbx_env_test("SBX_RCM_TIE_WALK")) == 0);
  return on;
}
// SBX_DEBUG_TIE_EDGES / SBX_DEBUG_TIE_CAP: the walk's limits, lowered (tests: small graphs then leave it at every exit)
static unsigned tie_walk_edges() {
  static const unsigned v = sbx_env_test("SBX_DEBUG_TIE_EDGES") ? (unsigned)atoll(sbx_env_test("SBX_DEBUG_TIE_EDGES")) : TS_EDGES;
  return v < TS_EDGES ? v : TS_EDGES;
}
static unsigned tie_walk_single() {  // SBX_DEBUG_TIE_SINGLE: entries above which a smallest-member step goes to the grid
  static const unsigned v = sbx_env_test("SBX_DEBUG_TIE_SINGLE") ? (unsigned)atoll(sbx_env_test("SBX_DEBUG_TIE_SINGLE")) : TS_SINGLE;
  return v < TS_SINGLE ? v : TS_SINGLE;
}
static unsigned tie_walk_cap() {
  static const unsigned v = sbx_env_test("SBX_DEBUG_TIE_CAP") ? (unsigned)atoll(sbx_env_test("SBX_DEBUG_TIE_CAP")) : TS_CAP;
  return v < TS_CAP ? v : TS_CAP;
}

// ... and, behind it, the walk's step at the same level: the smallest member of T_k adjacent to w_{k-1} = st->min_id (the
// members' entries again — consecutive ones, coalesced — compared with one vertex: no gathers)
__global__ __launch_bounds__(256) void k_tie_heavy_adj(const X *__restrict__ rp, const X *__restrict__ col, const I *list,
                                                       TieWalkState *st) {
  if (st->req != 1 || st->min_id == 0xFFFFFFFFu) return;
  __shared__ unsigned s_pref[TS_CAP + 1], s_mem[TS_CAP], s_red[4], s_scan[4];
  __shared__ X s_start[TS_CAP];
  const unsigned count = st->count, begin = st->begin, edges = st->edges, w = st->min_id;
  unsigned carry = 0;
  for (unsigned base = 0; base < count; base += 256) {
    const unsigned i = base + threadIdx.x;
    unsigned d = 0;
    if (i < count) {
      const unsigned v = (unsigned)list[begin + i];
      const X s0 = rp[v];
      s_mem[i] = v;
      s_start[i] = s0;
      d = (unsigned)(rp[v + 1] - s0);
    }
    unsigned total;
    const unsigned ex = sbx_block_exclusive_sum<unsigned, 256>(d, s_scan, &total);
    if (i < count) s_pref[i] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0) s_pref[count] = carry;
  __syncthreads();
  constexpr int U = 4;
  unsigned my_min = 0xFFFFFFFFu;
  const unsigned stride = gridDim.x * 256u;
  for (unsigned e0 = blockIdx.x * 256u + threadIdx.x; e0 < edges; e0 += stride * U) {
    unsigned c[U], jj[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const unsigned e = e0 + (unsigned)u * stride;
      jj[u] = tw_member_of(s_pref, count, e < edges ? e : edges - 1u);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const unsigned e = e0 + (unsigned)u * stride, ee = e < edges ? e : edges - 1u;
      c[u] = (unsigned)col[s_start[jj[u]] + (X)(ee - s_pref[jj[u]])];
    }
#pragma unroll
    for (int u = 0; u < U; u++)
      if (c[u] == w && e0 + (unsigned)u * stride < edges) my_min = s_mem[jj[u]] < my_min ? s_mem[jj[u]] : my_min;
  }
  const unsigned m = tw_block_min<256>(my_min, s_red);
  if (threadIdx.x == 0 && m != 0xFFFFFFFFu) atomicMin(&st->next_id, m);
}

// walk: 0, or the sweep's number of levels — the workgroup then goes on to tie_walk (col / vbits / dist are its inputs)
__global__ __launch_bounds__(1024) void k_ubfs_ties_small(const X *__restrict__ rp, const I *__restrict__ level,
                                                          unsigned count, unsigned *__restrict__ cone,
                                                          I *__restrict__ list, RcmDev *__restrict__ dv,
                                                          const X *__restrict__ col, const unsigned *__restrict__ vbits,
                                                          const unsigned *__restrict__ dist, unsigned walk,
                                                          unsigned walk_edges, unsigned walk_cap, unsigned walk_single,
                                                          TieWalkState *st, unsigned spec) {
  // (spec: the next sweep is enqueued behind this tie-break without a look at its outcome — RcmDev::spec_guard)
  __shared__ TieWalkLds s_walk;
  __shared__ unsigned s_red[16], s_cnt;
  if (threadIdx.x == 0) s_cnt = 0;
  unsigned best = 0xFFFFFFFFu;
  for (unsigned j = threadIdx.x; j < count; j += 1024) {
    const I v = level[j];
    const unsigned d = (unsigned)(rp[v + 1] - rp[v]);
    best = d < best ? d : best;
  }
  best = sbx_wave_min(best);
  if (sbx_lane() == 0) s_red[threadIdx.x >> 6] = best;
  __syncthreads();
  unsigned dmin = 0xFFFFFFFFu;
  for (int i = 0; i < 16; i++) dmin = s_red[i] < dmin ? s_red[i] : dmin;
  __syncthreads();
  unsigned mid = 0xFFFFFFFFu;
  for (unsigned j = threadIdx.x; j < count; j += 1024) {
    const I v = level[j];
    if ((unsigned)(rp[v + 1] - rp[v]) == dmin) {
      atomicOr(&cone[v >> 5], 1u << (v & 31));
      list[atomicAdd(&s_cnt, 1u)] = v;
      mid = (unsigned)v < mid ? (unsigned)v : mid;
    }
  }
  mid = sbx_wave_min(mid);
  if (sbx_lane() == 0) s_red[threadIdx.x >> 6] = mid;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned m = 0xFFFFFFFFu;
    for (int i = 0; i < 16; i++) m = s_red[i] < m ? s_red[i] : m;
    dv->tie_deg = dmin;
    dv->tie_min_id = m;
    dv->tie_count = s_cnt;
    dv->nf = s_cnt;
    dv->n_heavy = 0;
    dv->hub_overflow = 0;
    dv->cone_begin = 0;
    dv->cone_end = s_cnt;
    dv->bar = 0;
    dv->spec_guard = spec;
    dv->tie_done = s_cnt <= 1 ? 1u : 0u;  // one candidate: it is the next root, nothing to walk
    if (s_cnt <= 1) {
      dv->root_next = m;
      dv->root_pending = 1;
    }
  }
  __syncthreads();
  if (walk >= 2) {
    if (s_cnt > 1) (void)tie_walk(s_walk, rp, col, vbits, dist, list, s_cnt, walk, dv, st, walk_edges, walk_cap, walk_single, 0);
    else if (threadIdx.x == 0) st->req = 0;  // (one candidate: nothing for the kernels behind)
  }
}
// the walk taken up behind the grid's step (k_tie_heavy_min); leaves at once unless a step was handed over
__global__ __launch_bounds__(1024) void k_tie_walk_resume(const X *__restrict__ rp, const X *__restrict__ col,
                                                          const unsigned *__restrict__ vbits, const unsigned *__restrict__ dist,
                                                          I *list, RcmDev *dv, TieWalkState *st, unsigned walk_edges,
                                                          unsigned walk_cap, unsigned walk_single) {
  __shared__ TieWalkLds s_walk;
  if (st->req != 1) return;
  (void)tie_walk(s_walk, rp, col, vbits, dist, list, 0u, 0u, dv, st, walk_edges, walk_cap, walk_single, 1);
}
__global__ void k_ubfs_ties_init(RcmDev *__restrict__ dv) {
  dv->tie_done = 0;
  dv->tie_deg = 0xFFFFFFFFu;
  dv->tie_min_id = 0xFFFFFFFFu;
  dv->tie_count = 0;
  dv->nf = 0;
  dv->n_heavy = 0;
  dv->hub_overflow = 0;
}
__global__ __launch_bounds__(256) void k_ubfs_min_degree(const X *__restrict__ rp, const I *__restrict__ level,
                                                         unsigned count, RcmDev *__restrict__ dv) {
  unsigned best = 0xFFFFFFFFu;
  for (unsigned j = blockIdx.x * blockDim.x + threadIdx.x; j < count; j += gridDim.x * blockDim.x) {
    const I v = level[j];
    const unsigned d = (unsigned)(rp[v + 1] - rp[v]);
    best = d < best ? d : best;
  }
  best = sbx_wave_min(best);
  if (sbx_lane() == 0 && best != 0xFFFFFFFFu) atomicMin(&dv->tie_deg, best);
}
__global__ __launch_bounds__(256) void k_ubfs_mark_ties(const X *__restrict__ rp, const I *__restrict__ level,
                                                        unsigned count, unsigned *__restrict__ cone,
                                                        I *__restrict__ list, RcmDev *__restrict__ dv) {
  const unsigned dmin = dv->tie_deg;
  unsigned mid = 0xFFFFFFFFu;
  const unsigned rounds = (count + gridDim.x * blockDim.x - 1) / (gridDim.x * blockDim.x);
  for (unsigned r = 0; r < rounds; r++) {  // (all lanes stay in the loop: the append is wave-wide)
    const unsigned j = (r * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
    I v = 0;
    bool tie = false;
    if (j < count) {
      v = level[j];
      tie = (unsigned)(rp[v + 1] - rp[v]) == dmin;
    }
    if (tie) {
      atomicOr(&cone[v >> 5], 1u << (v & 31));
      mid = (unsigned)v < mid ? (unsigned)v : mid;
    }
    const unsigned slot = sbx_wave_append(&dv->nf, tie);
    if (tie) list[slot] = v;
  }
  mid = sbx_wave_min(mid);
  if (sbx_lane() == 0 && mid != 0xFFFFFFFFu) atomicMin(&dv->tie_min_id, mid);
}
__global__ void k_ubfs_root_from_single_tie(RcmDev *__restrict__ dv) { dv->root = dv->tie_min_id; }
// between two cone levels: what the expansion appended is the next level's range; hub queue cleared
__global__ void k_ubfs_cone_next(RcmDev *__restrict__ dv, int first) {
  if (first) {
    dv->cone_begin = 0;
    dv->cone_end = dv->nf;
    dv->bar = 0;
  } else {
    dv->cone_begin = dv->cone_end;
    dv->cone_end = dv->nf;
    if (dv->cone_end == dv->cone_begin) dv->unsym = 1;  // a level without a path to the level below
  }
  dv->n_heavy = 0;
  dv->hub_overflow = 0;
}

// w_0 = root, w_k = smallest id among the marked level-k neighbours of w_{k-1}; w_L is the next root.  One launch for
// the whole walk: UB_DESC_GRID workgroups (all resident: 64 of them on 256 CUs) scan adj(w_{k-1}) together — it may be
// a hub with 10^5 entries — and meet at a grid barrier (one arrival counter, dv->bar) after every level.  w_k lives in
// dv->desc[k % 3]; the step that fills slot k % 3 finds it cleared by the step before the previous barrier.
constexpr unsigned UB_DESC_GRID = 64;
// ---- grid barriers that give up --------------------------------------------------------------------------
// The persistent kernels below synchronise their workgroups through a counter in device memory.  That only works while
// every workgroup of the grid is running; a plain launch does not promise it, and with a second process on the same GPU
// one of two such kernels was seen to wait for ever for workgroups that never started (a cooperative launch does
// promise it, at ~170 us per launch: 1.7 ms per RCM).  So a wait is bounded: after GB_SPINS polls (a few
// milliseconds; a barrier normally takes a microsecond or two) the waiter raises dv->gb_abort and leaves, every other
// waiter sees the flag and leaves too, late workgroups leave at their first barrier, and the host — which finds the flag
// in its next read-back — throws the sweep away and runs it again with the one-launch-per-level kernels.
constexpr unsigned GB_SPINS = 1u << 16;
// THE INVARIANT the barriers rest on: every word one workgroup hands to another between two barriers is written and
// read with agent-scope atomics (relaxed: they go past the per-XCD L2s, the barrier orders them), and a wave drains its
// vector-memory counter (vmcnt 0) before its workgroup's arrival is counted.  There is no release / acquire fence in
// the product build: an agent-scope fence writes a whole L2 back.  The words under the invariant are listed at each
// persistent kernel (k_ubfs_descend_all, k_ubfs_small_run, k_ubfs_cone_run) and at the chained bottom-up level's
// election (k_ubfs_bottom_up).  A build with -DSBX_GB_FENCED puts a release fence in front of every arrival and an
// acquire fence behind every wait, so that ordinary stores and loads would be enough: if the product build ever
// returns something the fenced build does not, a word has escaped the invariant
// (tests/test_gpu_parity.py::test_rcm_fenced_build_agrees runs both on the stress graphs, several processes at once).
__device__ __forceinline__ bool gb_wait(RcmDev *dv, unsigned *word, unsigned target) {
  __shared__ int s_ok;
  // Everything this workgroup stored for the others (queue entries, hub queue, counters: agent-scope stores and atomics,
  // written through to the memory side) must have ARRIVED before the arrival below is counted.  __syncthreads() does
  // not wait for that — it compiles to s_barrier with no vmcnt wait in front — and the arrival, an atomic on another
  // address, can overtake a store that is still on its way: a workgroup on another XCD then passes the barrier and
  // reads the old word.  Seen only with several processes on the GPU (the memory side congested): one call in ~5000
  // lost part of a level — a hub's neighbours scanned from a stale hub queue — and with it vertices of its component.
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): every wave, for its own stores
  SBX_GB_RELEASE();                    // (SBX_GB_FENCED builds: the checking build of the invariant below)
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(word, 1u);
    int ok = 1;
    unsigned spins = 0;
    while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (__hip_atomic_load(&dv->gb_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || ++spins > dv->gb_spins) {
        __hip_atomic_store(&dv->gb_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = 0;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    s_ok = ok;
  }
  __syncthreads();
  SBX_GB_ACQUIRE();
  return s_ok != 0;
}

// (No __threadfence() around a barrier: on a multi-XCD part an agent-scope fence writes the L2 back — ~5 us a piece — and
// everything the workgroups exchange between two barriers moves through agent-scope atomics anyway.)
// Words under the barrier invariant (gb_wait) in k_ubfs_descend_all: dv->desc[0..2] (agent-scope stores, atomicMin, agent-
// scope loads), dv->bar, dv->gb_abort.  cone / vbits / dist were written by the kernels in front (a kernel boundary).
__global__ __launch_bounds__(256) void k_ubfs_descend_all(const X *__restrict__ rp, const X *__restrict__ col,
                                                          const unsigned *__restrict__ vbits,
                                                          const unsigned *__restrict__ dist,
                                                          const unsigned *__restrict__ cone, unsigned levels,
                                                          RcmDev *dv, int behind_cone_run) {
  if (dv->tie_done) return;  // (one candidate: k_ubfs_ties_small has named the next root)
  // launched straight behind k_ubfs_cone_run, before the host has seen how that ended: only a finished cone is walked
  // (written by the kernel in front or, gb_abort, through agent-scope atomics: the same for every workgroup)
  if (behind_cone_run && (dv->cone_status != UR_DONE || dv->unsym ||
                          __hip_atomic_load(&dv->gb_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))
    return;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    __hip_atomic_store(&dv->desc[0], dv->root, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&dv->desc[1], 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&dv->desc[2], 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (!gb_wait(dv, &dv->bar, gridDim.x)) return;
  unsigned w = __hip_atomic_load(&dv->desc[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (unsigned k = 1; k < levels; k++) {
    if (blockIdx.x == 0 && threadIdx.x == 0)  // idle during this step
      __hip_atomic_store(&dv->desc[(k + 1) % 3], 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned best = 0xFFFFFFFFu;
    if (w != 0xFFFFFFFFu) {
      const I s = rp[w], e = rp[w + 1];
      for (int64_t a = (int64_t)s + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; a < e;
           a += (int64_t)gridDim.x * blockDim.x) {
        const I c = col[a];
        if (((cone[c >> 5] >> (c & 31)) & 1u) && ((vbits[c >> 5] >> (c & 31)) & 1u) && dist[c] == k)
          best = (unsigned)c < best ? (unsigned)c : best;
      }
    }
    best = sbx_wave_min(best);
    if (sbx_lane() == 0 && best != 0xFFFFFFFFu) atomicMin(&dv->desc[k % 3], best);
    if (!gb_wait(dv, &dv->bar, gridDim.x * (k + 1))) return;
    w = __hip_atomic_load(&dv->desc[k % 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    // (not dv->root: if the LAST barrier gave up on the other workgroups while this one passed it, the host redoes the
    // sweep from the root it still expects there — k_gb_reset drops the pending one)
    if (w == 0xFFFFFFFFu) {
      // cannot happen on a symmetric pattern — unless a workgroup gave up at a barrier (gb_wait) and its share of a
      // later level was never scanned: then the host discards this walk, and the flag must not outlive it
      if (!__hip_atomic_load(&dv->gb_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) dv->unsym = 1;
    } else {
      dv->root_next = w;
      dv->root_pending = 1;
    }
  }
}

// after a grid barrier gave up: the words the persistent kernels leave zero when they finish normally
__global__ void k_gb_reset(RcmDev *__restrict__ dv) {
  dv->gb_abort = 0;
  dv->spec_guard = 0;
  dv->root_pending = 0;
  dv->tie_done = 0;
  dv->bar = 0;
  dv->ur_bar = 0;
  dv->ur_exit = 0;
  for (int i = 0; i < 3; i++) dv->ur_nf[i] = 0, dv->ur_nh[i] = 0, dv->ur_deg[i] = 0;
  dv->nf = 0;
  dv->fedges = 0;
  dv->n_heavy = 0;
  dv->hub_overflow = 0;
  dv->unf[0] = dv->unf[1] = 0;
  dv->ufedges[0] = dv->ufedges[1] = 0;
  dv->uc_mode = 0;
  dv->uc_done = 0;
}

// ---- unordered sweeps, small levels: one launch for as many consecutive levels as stay small ------------------
// The first levels of a sweep and its last ones hold a handful of vertices: an expansion + an n-sized collection + a
// host round trip per level is all overhead there.  UR_GRID workgroups (all resident) run such levels in one launch:
// vertices are claimed directly in the visited bitmap (atomicOr: nothing contends on a small level), appended to the
// queue through one wave-aggregated counter, hubs of the frontier are queued and scanned by the whole grid, and the
// workgroups meet at a grid barrier three times per level.  No fences (an agent-scope fence is an L2 write-back here):
// everything one workgroup writes for another — queue entries, hub queue, the hand-over state — is stored and loaded
// with agent-scope atomics, which go past the per-XCD L2s.  The kernel returns when the sweep is over (UR_DONE) or
// the level it just built is too big for it (UR_STOP: the host continues with the byte-claim / bottom-up kernels; the
// frontier bitmap is not kept here and is rebuilt if a bottom-up step wants it).
#ifndef SBX_UR_HEAVY
#define SBX_UR_HEAVY 2048
#endif
constexpr unsigned UR_HEAVY = SBX_UR_HEAVY;  // frontier vertices above this degree are scanned by the whole grid

__device__ __forceinline__ bool ur_barrier(RcmDev *dv, unsigned &epoch) {
  epoch++;
  return gb_wait(dv, &dv->ur_bar, epoch * gridDim.x);
}

// Words under the barrier invariant (gb_wait) in k_ubfs_small_run: the queue entries q[] and the hub queue hq[]
// (ur_store / ur_load), the visited words vbits[] (ur_load, atomicOr), the level slots dv->ur_nf / ur_nh / ur_deg[0..2]
// (atomicAdd, ur_store, ur_load), dv->ur_bar, dv->ur_exit, dv->gb_abort.  dist[] is written with plain stores: no
// workgroup of THIS kernel reads it.  The hand-over state (dv->ur_off ...) is thread 0's alone and read behind the
// kernel's end.
__global__ __launch_bounds__(256) void k_ubfs_small_run(const X *__restrict__ rp, const X *__restrict__ col,
                                                        unsigned *vbits, unsigned *dist, I *q, I *hq, RcmDev *dv,
                                                        unsigned off, unsigned size, unsigned level, unsigned total,
                                                        long long fe_in, unsigned max_levels, int from_dev) {
  SBX_SPEC_GUARD_LEAVE(dv);  // (the start kernel in front left: so does this one)
  if (from_dev) {  // the tail of a device-driven chain (run_ubfs): the frontier is the chain's, if it asks for this kernel at all
    if (dv->uc_mode != UC_SMALL) return;
    // (tests, SBX_DEBUG_CHAIN_TAIL_ABORT: the grid barriers of this launch give up — after the workgroups have claimed
    // their first vertices, which is what a barrier that gives up on a shared GPU leaves behind)
    if (from_dev == 2 && blockIdx.x == 0 && threadIdx.x == 0)
      __hip_atomic_store(&dv->gb_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    off = dv->uc_off, size = dv->uc_size, level = dv->uc_level, total = dv->uc_total;
    fe_in = (long long)dv->uc_fe;
  }
  // Counters live in three slots used in rotation (slot = level being built % 3): every workgroup reads a level's
  // totals right after its barrier and derives the next state itself — no broadcast, no reset in between; the slot
  // the NEXT level will use is cleared during this one (its last readers passed the previous barrier).  One barrier
  // per level, two when the frontier held hubs.  All slots and the barrier words are zero on entry (k_ubfs_start,
  // and the last workgroup out leaves them so).
  const int lane = sbx_lane();
  const unsigned gwave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const unsigned level_in = level;
  unsigned epoch = 0;
  unsigned long long scanned = 0, esum = 0;
  unsigned long long fe_cur = fe_in >= 0 ? (unsigned long long)fe_in : ur_load(&dv->fedges);  // level 0: k_ubfs_start left it
  unsigned status = (fe_cur > (unsigned long long)UR_MAX_E || size > UR_MAX_N) ? UR_STOP : UR_CONTINUE;
  while (status == UR_CONTINUE) {
    const unsigned slot = (level + 1) % 3, next_slot = (level + 2) % 3;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      ur_store(&dv->ur_nf[next_slot], 0u);
      ur_store(&dv->ur_nh[next_slot], 0u);
      ur_store(&dv->ur_deg[next_slot], 0ull);
    }
    I *q_next = q + off + size;
    unsigned long long degacc = 0;
    // 256 adjacency entries per wave and step, in phases — columns, bitmap words, claims, ONE reservation, stores —
    // so that a step costs one chain of dependent memory round trips instead of four (a 2048-entry vertex: 8 steps
    // of ~2.5 us instead of 32)
    auto visit4 = [&](const I (&c)[4]) {
      unsigned w[4];
#pragma unroll
      for (int k = 0; k < 4; k++) w[k] = c[k] >= 0 ? ur_load(&vbits[c[k] >> 5]) : ~0u;
      unsigned long long m[4];
      unsigned tot = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const unsigned bit = 1u << (c[k] & 31);
        const bool won = c[k] >= 0 && !(w[k] & bit) && !(atomicOr(&vbits[c[k] >> 5], bit) & bit);
        m[k] = __ballot(won);
        tot += (unsigned)__popcll(m[k]);
      }
      if (tot == 0) return;
      unsigned base = 0;
      if (lane == 0) base = atomicAdd(&dv->ur_nf[slot], tot);
      base = (unsigned)__shfl((int)base, 0, 64);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if ((m[k] >> lane) & 1ull) {
          ur_store(&q_next[base + (unsigned)__popcll(m[k] & sbx_lanemask_lt())], c[k]);
          dist[c[k]] = level + 1;  // (read by later kernels only)
          degacc += (unsigned long long)(rp[c[k] + 1] - rp[c[k]]);
        }
        base += (unsigned)__popcll(m[k]);
      }
    };
    auto flush_degrees = [&]() {
      degacc = sbx_wave_sum(degacc);
      if (lane == 0 && degacc) atomicAdd(&dv->ur_deg[slot], degacc);
      degacc = 0;
    };
    // light vertices: one wave each; hubs of the frontier are queued
    for (unsigned p = gwave; p < size; p += nwaves) {
      const I u = ur_load(&q[off + p]);
      const I s = rp[u], e = rp[u + 1];
      if ((unsigned)(e - s) > UR_HEAVY) {
        if (lane == 0) ur_store(&hq[atomicAdd(&dv->ur_nh[slot], 1u)], u);
        continue;
      }
      for (int64_t a0 = s; a0 < e; a0 += 256) {
        I c[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int64_t a = a0 + k * 64 + lane;
          c[k] = a < e ? col[a] : (I)-1;
          scanned += a < e ? 1u : 0u;
        }
        visit4(c);
      }
    }
    flush_degrees();
    if (!ur_barrier(dv, epoch)) return;
    const unsigned nh = ur_load(&dv->ur_nh[slot]);
    if (nh) {  // the whole grid scans each hub
      for (unsigned i = 0; i < nh; i++) {
        const I u = ur_load(&hq[i]);
        const I s = rp[u], e = rp[u + 1];
        for (int64_t a0 = (int64_t)s + (int64_t)gwave * 256; a0 < e; a0 += (int64_t)nwaves * 256) {
          I c[4];
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int64_t a = a0 + k * 64 + lane;
            c[k] = a < e ? col[a] : (I)-1;
            scanned += a < e ? 1u : 0u;
          }
          visit4(c);
        }
      }
      flush_degrees();
      if (!ur_barrier(dv, epoch)) return;
    }
    // every workgroup derives the same next state from the level's totals
    const unsigned nf = ur_load(&dv->ur_nf[slot]);
    const unsigned long long fe = ur_load(&dv->ur_deg[slot]);
    if (nf == 0) {
      status = UR_DONE;  // the frontier just expanded was the deepest level: off / size stay on it
    } else {
      esum += fe;
      fe_cur = fe;
      off += size;
      size = nf;
      level++;
      total += nf;
      status = (fe > (unsigned long long)UR_MAX_E || nf > UR_MAX_N) ? UR_STOP : UR_CONTINUE;
      // a deep, narrow component: the one-workgroup ordered kernel walks such levels at half the cost of this one
      if (status == UR_CONTINUE && level - level_in >= max_levels) status = UR_DEEP;
    }
  }
  scanned = sbx_wave_sum(scanned);
  if (lane == 0 && scanned) atomicAdd(&dv->edges, scanned);
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // the hand-over state (plain stores: the host reads after the kernel)
    dv->ur_off = off, dv->ur_size = size, dv->ur_level = level, dv->ur_total = total;
    dv->ur_status = status;
    dv->ur_fe = fe_cur;
    dv->ur_esum = esum;
    if (from_dev) {  // (the host reads the unvisited edges back with the rest)
      const long long rem = dv->uc_remaining - (long long)esum;
      dv->uc_remaining = rem < 0 ? 0 : rem;
      dv->uc_small_ran = 1;  // (uc_mode itself stays: workgroups that start late still have to read it)
    }
    dv->unf[0] = dv->unf[1] = 0;  // (the level slots of the other kernels: levels may change parity in here)
    dv->ufedges[0] = dv->ufedges[1] = 0;
    dv->n_heavy = 0;
    dv->hub_overflow = 0;
  }
  // the last workgroup out leaves the barrier words and the slots zero for the next launch
  __syncthreads();
  if (threadIdx.x == 0 && atomicAdd(&dv->ur_exit, 1u) == gridDim.x - 1) {
    ur_store(&dv->ur_bar, 0u);
    ur_store(&dv->ur_exit, 0u);
    for (int i = 0; i < 3; i++) {
      ur_store(&dv->ur_nf[i], 0u);
      ur_store(&dv->ur_nh[i], 0u);
      ur_store(&dv->ur_deg[i], 0ull);
    }
  }
}

// The frontier bitmap of an unordered sweep rebuilt from what every level kernel leaves behind — frontier = visited
// vertices at distance `level` — in one streaming pass that writes every word (a fill of the bitmap + a scatter of the
// frontier's bits were four launches: hipMemsetAsync splits into three).  Unvisited vertices may hold stale distances.
// (a 512 KB device-to-device hipMemcpyAsync costs the host ~25 us of enqueueing around a 4 us blit: on the call's
// latency chain, between the first sweep and its tie-break)
__global__ __launch_bounds__(256) void k_copy_words(unsigned *__restrict__ dst, const unsigned *__restrict__ src,
                                                    int64_t words) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < words; i += stride) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void k_ubfs_fbits_from_dist(const unsigned *__restrict__ vbits,
                                                              const unsigned *__restrict__ dist, unsigned level,
                                                              int64_t n, unsigned *__restrict__ fbits,
                                                              const RcmDev *__restrict__ chain = nullptr) {
  if (chain) {  // in front of a chain begun on the device: only if its first link is a bottom-up level, for its frontier
    if (chain->uc_mode != UC_BU) return;
    level = chain->uc_level;
  }
  const int64_t words = (n + 31) / 32;
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;  // (a multiple of 64: a wave covers two whole words)
  for (; v < words * 32; v += stride) {
    const bool in = v < n && ((vbits[v >> 5] >> (v & 31)) & 1u) && dist[v] == level;
    const unsigned long long m = __ballot(in);
    const int lane = sbx_lane();
    if (lane == 0) fbits[v >> 5] = (unsigned)m;
    if (lane == 32) fbits[v >> 5] = (unsigned)(m >> 32);
  }
}

// Cone marking, small levels: the same persistent scheme as k_ubfs_small_run.  T_{k-1} = the level-(k-1) neighbours of
// T_k, for k = k_start down to 2, while the list of T_k stays within CONE_SMALL members; a longer list hands the level
// back to the host (UR_STOP, dv->cone_k = k: the expansion kernels in their cone mode take it), a finished walk returns
// UR_DONE.  The list is one growing array with one counter (dv->nf); members and hub queue move through agent-scope
// atomics, the cone bits through atomicOr.
// Words under the barrier invariant (gb_wait) in k_ubfs_cone_run: the member list list[] and the hub queue hq[] (ur_store /
// ur_load), the cone words cone[] (ur_load, atomicOr), dv->nf (atomicAdd), dv->ur_nf / ur_nh[0..2], dv->cone_begin /
// cone_end (ur_load: written by the one-thread kernel in front), dv->ur_bar, dv->ur_exit, dv->gb_abort.  vbits[] and
// dist[] are the finished sweep's (a kernel boundary): plain loads.
constexpr unsigned CONE_SMALL = 256;  // (a wave takes a member: 256 waves; 1024 measured 0.3 ms slower per RCM)
__global__ __launch_bounds__(256) void k_ubfs_cone_run(const X *__restrict__ rp, const X *__restrict__ col,
                                                       const unsigned *__restrict__ vbits,
                                                       const unsigned *__restrict__ dist, unsigned *cone, I *list, I *hq,
                                                       RcmDev *dv, unsigned k_start) {
  const int lane = sbx_lane();
  const unsigned gwave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  if (dv->tie_done) {  // (written by the kernel in front: the same for every workgroup)
    if (blockIdx.x == 0 && threadIdx.x == 0) dv->cone_status = UR_DONE;
    return;
  }
  unsigned epoch = 0;
  unsigned begin = ur_load(&dv->cone_begin), end = ur_load(&dv->cone_end);
  unsigned k = k_start, status = UR_DONE;
  bool broken = false;
  for (; k >= 2; k--) {
    if (end - begin > CONE_SMALL) {
      status = UR_STOP;
      break;
    }
    const unsigned slot = k % 3, next_slot = (k + 2) % 3;  // (k counts down: the next level uses (k - 1) % 3)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      ur_store(&dv->ur_nh[next_slot], 0u);
      ur_store(&dv->ur_nf[next_slot], 0u);
    }
    auto visit4 = [&](const I (&c)[4]) {  // (phases as in k_ubfs_small_run)
      unsigned vb[4];
#pragma unroll
      for (int j = 0; j < 4; j++) vb[j] = c[j] >= 0 ? vbits[c[j] >> 5] : 0u;
      unsigned dd[4], cw[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const bool reached = (vb[j] >> (c[j] & 31)) & 1u;
        dd[j] = reached ? dist[c[j]] : 0xFFFFFFFFu;
        cw[j] = reached ? ur_load(&cone[c[j] >> 5]) : ~0u;
      }
      unsigned long long m[4];
      unsigned tot = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const unsigned bit = 1u << (c[j] & 31);
        const bool won = dd[j] == k - 1 && !(cw[j] & bit) && !(atomicOr(&cone[c[j] >> 5], bit) & bit);
        m[j] = __ballot(won);
        tot += (unsigned)__popcll(m[j]);
      }
      if (tot == 0) return;
      unsigned base = 0;
      if (lane == 0) {
        base = atomicAdd(&dv->nf, tot);        // the member's place in the one growing list ...
        atomicAdd(&dv->ur_nf[slot], tot);      // ... and the size of the level being marked (see below)
      }
      base = (unsigned)__shfl((int)base, 0, 64);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if ((m[j] >> lane) & 1ull) ur_store(&list[base + (unsigned)__popcll(m[j] & sbx_lanemask_lt())], c[j]);
        base += (unsigned)__popcll(m[j]);
      }
    };
    for (unsigned p = begin + gwave; p < end; p += nwaves) {
      const I u = ur_load(&list[p]);
      const I s = rp[u], e = rp[u + 1];
      if ((unsigned)(e - s) > UR_HEAVY) {
        if (lane == 0) ur_store(&hq[atomicAdd(&dv->ur_nh[slot], 1u)], u);
        continue;
      }
      for (int64_t a0 = s; a0 < e; a0 += 256) {
        I c[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int64_t a = a0 + j * 64 + lane;
          c[j] = a < e ? col[a] : (I)-1;
        }
        visit4(c);
      }
    }
    if (!ur_barrier(dv, epoch)) return;
    const unsigned nh = ur_load(&dv->ur_nh[slot]);
    if (nh) {
      for (unsigned i = 0; i < nh; i++) {
        const I u = ur_load(&hq[i]);
        const I s = rp[u], e = rp[u + 1];
        for (int64_t a0 = (int64_t)s + (int64_t)gwave * 256; a0 < e; a0 += (int64_t)nwaves * 256) {
          I c[4];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int64_t a = a0 + j * 64 + lane;
            c[j] = a < e ? col[a] : (I)-1;
          }
          visit4(c);
        }
      }
      if (!ur_barrier(dv, epoch)) return;
    }
    // The level's end is its begin + what THIS level appended (a per-level slot, final behind the barrier) — not the
    // list counter itself: a workgroup that is slow to get here (a GPU shared with other processes) would read a
    // counter other workgroups are already appending the next level to, and walk a different range than they do.
    begin = end;
    end = begin + ur_load(&dv->ur_nf[slot]);
    if (end == begin) {  // a level without a path to the level below: not a symmetric pattern
      broken = true;
      break;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    dv->cone_begin = begin;
    dv->cone_end = end;
    dv->cone_k = k;
    dv->cone_status = status;
    // (a level that found nothing because some workgroup had given up at a barrier is not an unsymmetric pattern: the
    // host redoes the tie-break the safe way and the flag must not outlive this launch)
    if (broken && !ur_load(&dv->gb_abort)) dv->unsym = 1;
    dv->n_heavy = 0;
    dv->hub_overflow = 0;
  }
  __syncthreads();
  if (threadIdx.x == 0 && atomicAdd(&dv->ur_exit, 1u) == gridDim.x - 1) {
    ur_store(&dv->ur_bar, 0u);
    ur_store(&dv->ur_exit, 0u);
    for (int i = 0; i < 3; i++) ur_store(&dv->ur_nh[i], 0u), ur_store(&dv->ur_nf[i], 0u);
  }
}

// The link behind a top-down level (three kernels, none of which knows it is the last): the chain begins here.
__global__ void k_ubfs_chain_next(RcmDev *dv, ChainInit ci) {
  uc_begin(dv, ci);
  uc_advance(dv, ci.bu_ratio);
}

// The chain begun on the DEVICE, behind the small-level run at a sweep's head: when that run stopped at a frontier too
// big for it (UR_STOP, having moved) the host used to read back, decide the direction and launch the big level with the
// chain behind it.  The decision needs nothing the device does not have: this one-thread kernel takes the run's
// hand-over state as the chain's frontier and applies the host loop's rule; the frontier bitmap, the bottom-up links
// and the tail run are enqueued behind it and run if it says so — one read-back serves the head run and the chain.
__global__ void k_ubfs_chain_from_small(RcmDev *dv, long long remaining_before, double bu_ratio, unsigned level_in) {
  dv->uc_flips = 0;
  dv->uc_small_ran = 0;
  dv->uc_head = 0;
  dv->uc_mode = UC_HOST;  // (the links and the tail run leave unless told otherwise)
  if (dv->gb_abort || dv->ur_status != UR_STOP || dv->ur_level == level_in) return;
  long long rem = remaining_before - (long long)dv->ur_esum;
  if (rem < 0) rem = 0;
  dv->uc_off = dv->ur_off, dv->uc_size = dv->ur_size, dv->uc_level = dv->ur_level, dv->uc_total = dv->ur_total;
  dv->uc_remaining = rem;
  dv->uc_fe = dv->ur_fe;
  dv->uc_head = 1;
  if (dv->ur_size >= 1024 && (double)dv->ur_fe > bu_ratio * (double)rem) dv->uc_mode = UC_BU;
}

// SBX_DEBUG_RCM_CHECK=1: after an unordered sweep, every edge that leaves its visited set (there must be none) is counted
// and the first few are recorded: (visited end, unvisited end, the visited end's level).
__global__ __launch_bounds__(256) void k_check_closed(const X *__restrict__ rp, const X *__restrict__ col,
                                                      const unsigned *__restrict__ vbits, const unsigned *__restrict__ dist,
                                                      int64_t n, unsigned *__restrict__ out) {
  int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; u < n; u += stride) {
    if (!((vbits[u >> 5] >> (u & 31)) & 1u)) continue;
    for (I j = rp[u]; j < rp[u + 1]; j++) {
      const I w = col[j];
      if (!((vbits[w >> 5] >> (w & 31)) & 1u)) {
        const unsigned k = atomicAdd(&out[0], 1u);
        if (k < 8) out[1 + 3 * k] = (unsigned)u, out[2 + 3 * k] = (unsigned)w, out[3 + 3 * k] = dist[u];
      }
    }
  }
}

// One unordered sweep from fixed_root (>= 0) or dv->root: level sets only.  *too_deep is set when the sweep passed the
// depth limit (ub_max_levels) and was abandoned: the caller runs the ordered sweep instead.
static int run_ubfs(sbx_handle_t h, const BfsBuffers &b, unsigned char *claim8, unsigned *nbits_buf, unsigned *cone,
                    I fixed_root, I comp_label, BfsResult *out, bool *too_deep, bool *spec_failed = nullptr) {
  if (spec_failed) *spec_failed = false;
  const bool claim_was_clean = *b.claim_clean;
  const size_t bm_bytes = (size_t)((b.n + 31) / 32) * sizeof(unsigned);
  const int64_t words = (b.n + 63) / 64;
  const int64_t fw_blocks = (words + RCM_FW_WORDS - 1) / RCM_FW_WORDS;
  *too_deep = false;
  // the start kernel clears the sweep's bitmaps, the tie-break's cone bitmap (ubfs_pick_root) and — after a sweep that
  // did not run to its end — the claim bytes
  const StartClear sc = {b.vbits, (unsigned long long)((b.fbits - b.vbits) + bm_bytes / sizeof(unsigned)),
                         cone, (unsigned long long)(bm_bytes / sizeof(unsigned)),
                         (unsigned *)claim8, *b.claim_clean ? 0ull : (unsigned long long)((b.n + 3) / 4)};
  *b.claim_clean = false;  // (until this sweep has run to its end: every level's collection pass clears what it read)
  unsigned *dist = b.lpos;  // level positions are an ordered sweep's business: the array is free here
  static const unsigned gb_spins = sbx_env_test("SBX_DEBUG_GB_SPINS") ? (unsigned)atoll(sbx_env_test("SBX_DEBUG_GB_SPINS")) : GB_SPINS;
  SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_ubfs_start, dim3(RCM_START_GRID), dim3(256), b.rp, b.vbits, b.fbits, dist, b.q, b.dv,
              fixed_root, gb_spins, sc);
  unsigned off = 0, fsize = 1, level = 0, total = 1;
  const unsigned max_grid = (unsigned)h->num_cus * 8;
  static int heavy_per_cu = 0;
  if (heavy_per_cu == 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_bfs_expand_heavy<1>, 256, 0) != hipSuccess || nb < 1) nb = 4;
    heavy_per_cu = nb;
  }
  const unsigned heavy_grid = (unsigned)h->num_cus * (unsigned)heavy_per_cu;
  int64_t remaining = b.nnz, frontier_edges = -1;
  unsigned *cur_f = b.fbits, *cur_n = nbits_buf;  // frontier bitmap / the one a bottom-up level writes (swapped after it)
  bool fbits_valid = true;  // cur_f holds exactly the current frontier (the root, or what the level kernels left)
  unsigned rounds = 0;      // host round trips of this sweep
  static const bool dbg_check = sbx_env_test("SBX_DEBUG_RCM_CHECK") && atoi(sbx_env_test("SBX_DEBUG_RCM_CHECK")) != 0;
  std::vector<std::string> trace;
  auto note = [&](const char *what, unsigned a, unsigned b_, long long c) {
    if (!dbg_check) return;
    char buf[160];
    snprintf(buf, sizeof buf, "%s level %u off %u fsize %u total %u fe %lld rem %lld | %u %u %lld", what, level, off, fsize, total,
             (long long)frontier_edges, (long long)remaining, a, b_, c);
    trace.push_back(buf);
  };
  while (true) {
    if (++rounds > ub_max_levels()) {
      *too_deep = true;
      return SBX_OK;
    }
    if (frontier_edges < 0 || (frontier_edges <= (int64_t)UR_MAX_E && fsize <= UR_MAX_N)) {
      // small levels: as many as stay small, in one launch
      SBX_KLAUNCH(h, SBX_K_BFS_SMALL, k_ubfs_small_run, dim3(UR_GRID), dim3(256), b.rp, b.col, b.vbits, dist, b.q,
                  (I *)b.heavy, b.dv, off, fsize, level, total, (long long)frontier_edges, ub_max_levels(), 0);
      // At a sweep's start on a big graph this run usually ends at a frontier for the bottom-up kernels: the chain that
      // follows is enqueued now and begun on the device (k_ubfs_chain_from_small) — one read-back for the head run and the
      // chain instead of two.  Where the run ends otherwise (the sweep is over, a top-down level is due, too deep) the
      // half-dozen launches leave at once: only where it can pay (BfsBuffers::head_chain, a graph that is not small).
      static const bool head_chain_on = !(sbx_env_test("SBX_RCM_HEAD_CHAIN") && atoi(sbx_env_test("SBX_RCM_HEAD_CHAIN")) == 0);
      const int spec_len =
          (head_chain_on && b.head_chain && frontier_edges < 0 && b.n >= ((int64_t)1 << 17)) ? ubfs_chain() : 0;
      if (spec_len > 0) {
        const ChainInit ci0 = {0u, 0u, 0u, 0u, 0ll, ubu_ratio()};
        SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_ubfs_chain_from_small, dim3(1), dim3(1), b.dv, (long long)remaining, ubu_ratio(),
                    level);
        SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_ubfs_fbits_from_dist, dim3(sbx_grid_for(b.n, 256, 4096)), dim3(256),
                    (const unsigned *)b.vbits, (const unsigned *)dist, 0u, b.n, cur_f, (const RcmDev *)b.dv);
        unsigned *cf = cur_f, *cn = cur_n;
        for (int i = 0; i <= spec_len; i++) {
          SBX_KLAUNCH(h, SBX_K_BFS_BOTTOMUP, k_ubfs_bottom_up, dim3(max_grid), dim3(256), b.rp, b.col, b.label, comp_label,
                      b.vbits, (const unsigned *)cf, cn, dist, 0u, b.q, b.n, b.dv, 1, ci0);
          std::swap(cf, cn);
        }
        static const int tail_mode0 = sbx_env_test("SBX_DEBUG_CHAIN_TAIL_ABORT") && atoi(sbx_env_test("SBX_DEBUG_CHAIN_TAIL_ABORT")) ? 2 : 1;
        SBX_KLAUNCH(h, SBX_K_BFS_SMALL, k_ubfs_small_run, dim3(UR_GRID), dim3(256), b.rp, b.col, b.vbits, dist, b.q,
                    (I *)b.heavy, b.dv, 0u, 0u, 0u, 0u, 0ll, ub_max_levels(), tail_mode0);
      }
      SBX_LAUNCH_CHECK(h);
      // (the hook's work needs the degree counts: while this read-back is the one that delivers them it waits for the
      // next launch with a long cover)
      SBX_TRY(bfs_first_launch(b, spec_len > 0 && (!b.late_counts || *b.late_counts_ready) ? 150 : 40));
      RcmDev hs;
      SBX_TRY(sbx_readback(h, &hs, b.dv, sizeof(RcmDev)));
      if (b.tie_unverified && *b.tie_unverified) {  // (the sweep's first read-back: see BfsBuffers::tie_unverified)
        *b.tie_unverified = false;
        if (!hs.tie_done) {
          if (!spec_failed) SBX_FAIL(h, SBX_ERR_INTERNAL, "run_ubfs: an unverified tie-break in front of a sweep that cannot be redone");
          *b.claim_clean = claim_was_clean;  // (the start kernel left before it cleared anything)
          *spec_failed = true;
          return SBX_OK;
        }
      }
      if (b.late_counts && !*b.late_counts_ready) {
        *b.late_counts = hs;
        *b.late_counts_ready = true;
      }
      if (hs.gb_abort) {  // a grid barrier gave up (gb_wait): this sweep is redone by the ordered kernels
        SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_gb_reset, dim3(1), dim3(1), b.dv);
        h->rcm_gb_backoff = gb_backoff_calls();
        *too_deep = true;
        return SBX_OK;
      }
      if (spec_len > 0 && hs.uc_head) {
        // the head run stopped at a big frontier and the chain went on from there: as behind a chain the host launched
        note("small_run(head) + chain", hs.uc_flips, hs.uc_mode, (long long)hs.uc_remaining);
        if (hs.uc_flips & 1u) std::swap(cur_f, cur_n);
        rounds += hs.uc_flips;
        remaining = (int64_t)hs.uc_remaining;
        fbits_valid = hs.uc_flips > 0;  // (no link ran: the frontier is the head run's, its bitmap was not built)
        if (hs.uc_small_ran) {
          const bool moved2 = hs.ur_level != hs.uc_level;
          off = hs.ur_off, fsize = hs.ur_size, level = hs.ur_level, total = hs.ur_total;
          frontier_edges = (int64_t)hs.ur_fe;
          if (moved2) fbits_valid = false;
          if (hs.ur_status == UR_DONE) break;
          if (hs.ur_status == UR_DEEP) {
            *too_deep = true;
            return SBX_OK;
          }
          continue;
        }
        off = hs.uc_off, fsize = hs.uc_size, level = hs.uc_level, total = hs.uc_total;
        if (hs.uc_mode == UC_DONE) break;
        frontier_edges = (int64_t)hs.uc_fe;
        continue;
      }
      const bool moved = hs.ur_level != level;
      note("small_run(before)", hs.ur_status, hs.ur_level, (long long)hs.ur_esum);
      off = hs.ur_off, fsize = hs.ur_size, level = hs.ur_level, total = hs.ur_total;
      remaining -= (int64_t)hs.ur_esum;
      if (remaining < 0) remaining = 0;
      frontier_edges = (int64_t)hs.ur_fe;
      if (moved) fbits_valid = false;
      if (hs.ur_status == UR_DONE) break;
      if (hs.ur_status == UR_DEEP) {
        *too_deep = true;
        return SBX_OK;
      }
      if (moved) continue;  // (UR_STOP with a new frontier: it is looked at again, and is big)
      // UR_STOP right at the entry: the frontier handed in was too big — expand it with the kernels below
    }
    I *q_next = b.q + off + fsize;  // the next level is appended to the queue
    const UnorderedSweep us = {claim8, nullptr, dist, level + 1};
    const bool bottom_up = frontier_edges >= 0 && fsize >= 1024 && (double)frontier_edges > ubu_ratio() * (double)remaining;
    note(bottom_up ? "bottom-up" : "top-down", fbits_valid, 0, 0);
    const int chain_len = ubfs_chain();
    const ChainInit ci = {off, fsize, level, total, (long long)remaining, ubu_ratio()};
    if (bottom_up) {
      if (!fbits_valid)
        SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_ubfs_fbits_from_dist, dim3(sbx_grid_for(b.n, 256, 4096)), dim3(256),
                    (const unsigned *)b.vbits, (const unsigned *)dist, level, b.n, cur_f, (const RcmDev *)nullptr);
      // publishes the level itself (bitmaps, distances, queue): no collection pass
      SBX_KLAUNCH(h, SBX_K_BFS_BOTTOMUP, k_ubfs_bottom_up, dim3(max_grid), dim3(256), b.rp, b.col, b.label, comp_label,
                  b.vbits, (const unsigned *)cur_f, cur_n, dist, level + 1, q_next, b.n, b.dv, chain_len > 0 ? 2 : 0, ci);
      std::swap(cur_f, cur_n);
    } else {
      const unsigned waves_needed = (fsize + RCM_VPW - 1) / RCM_VPW;
      unsigned grid = (waves_needed + 3) / 4;
      if (grid > max_grid) grid = max_grid;
      if (grid < 1) grid = 1;
      SBX_KLAUNCH(h, SBX_K_BFS_EXPAND, k_bfs_expand<1>, dim3(grid), dim3(256), b.rp, b.col, (const I *)(b.q + off),
                  fsize, level + 1, (const unsigned *)b.vbits, b.ppos, b.nf_list, b.heavy, b.hub_dir, b.dv, us, 0);
      if ((b.late_counts ? b.late_counts->max_deg : b.max_deg) > (unsigned)RCM_LIGHT)
        SBX_KLAUNCH(h, SBX_K_BFS_HEAVY, k_bfs_expand_heavy<1>, dim3(heavy_grid), dim3(256), b.rp, b.col,
                    (const I *)(b.q + off), level + 1, (const unsigned *)b.vbits, b.ppos, b.nf_list,
                    (const uint64_t *)b.heavy, (const uint2 *)b.hub_dir, grid, b.dv, us);
      SBX_KLAUNCH(h, SBX_K_LEVEL_ORDER, k_ubfs_collect, dim3((unsigned)fw_blocks), dim3(256), claim8,
                  (unsigned long long *)b.vbits, (unsigned long long *)cur_f, dist, level + 1, b.rp, q_next, b.n, b.dv);
      if (chain_len > 0) SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_ubfs_chain_next, dim3(1), dim3(1), b.dv, ci);
    }
    SBX_LAUNCH_CHECK(h);
    fbits_valid = true;
    RcmDev hd;
    if (chain_len > 0) {
      // The levels behind this one are enqueued without waiting for it: up to ubfs_chain() bottom-up levels and the
      // persistent small-level kernel.  Each runs only if the state the level in front of it left says so — its last
      // workgroup out applies the rules of this loop on the device (uc_advance) — and ONE read-back serves them all.
      // A sweep of the bench matrix: a small run, three bottom-up levels, a small run — five round trips, now two.
      unsigned *cf = cur_f, *cn = cur_n;
      for (int i = 0; i < chain_len; i++) {
        SBX_KLAUNCH(h, SBX_K_BFS_BOTTOMUP, k_ubfs_bottom_up, dim3(max_grid), dim3(256), b.rp, b.col, b.label, comp_label,
                    b.vbits, (const unsigned *)cf, cn, dist, 0u, b.q, b.n, b.dv, 1, ci);
        std::swap(cf, cn);
      }
      static const int tail_mode = sbx_env_test("SBX_DEBUG_CHAIN_TAIL_ABORT") && atoi(sbx_env_test("SBX_DEBUG_CHAIN_TAIL_ABORT")) ? 2 : 1;
      SBX_KLAUNCH(h, SBX_K_BFS_SMALL, k_ubfs_small_run, dim3(UR_GRID), dim3(256), b.rp, b.col, b.vbits, dist, b.q,
                  (I *)b.heavy, b.dv, 0u, 0u, 0u, 0u, 0ll, ub_max_levels(), tail_mode);
      SBX_LAUNCH_CHECK(h);
      SBX_TRY(bfs_first_launch(b, 150));  // (a chain of big levels is in flight: the host has nothing else to do)
      SBX_TRY(sbx_readback(h, &hd, b.dv, sizeof(RcmDev)));
      if (hd.gb_abort) {
        // a grid barrier of the chain's small-level kernel gave up — before it could say that it ran at all: the sweep
        // is redone by the ordered kernels, as after the small-level kernel at the head of the loop
        SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_gb_reset, dim3(1), dim3(1), b.dv);
        h->rcm_gb_backoff = gb_backoff_calls();
        *too_deep = true;
        return SBX_OK;
      }
      if (hd.uc_flips & 1u) std::swap(cur_f, cur_n);
      rounds += hd.uc_flips;
      remaining = (int64_t)hd.uc_remaining;
      if (hd.uc_small_ran) {  // as after the small-level kernel at the head of the loop
        const bool moved = hd.ur_level != hd.uc_level;
        off = hd.ur_off, fsize = hd.ur_size, level = hd.ur_level, total = hd.ur_total;
        frontier_edges = (int64_t)hd.ur_fe;
        if (moved) fbits_valid = false;
        if (hd.ur_status == UR_DONE) break;
        if (hd.ur_status == UR_DEEP) {
          *too_deep = true;
          return SBX_OK;
        }
        continue;  // UR_STOP: a frontier for the big kernels again (the head of the loop looks at it and passes it on)
      }
      off = hd.uc_off, fsize = hd.uc_size, level = hd.uc_level, total = hd.uc_total;
      if (hd.uc_mode == UC_DONE) break;
      frontier_edges = (int64_t)hd.uc_fe;  // UC_BU (the chain was too short) / UC_HOST (a top-down level): the loop goes on
      continue;
    }
    SBX_TRY(sbx_readback(h, &hd, b.dv, sizeof(RcmDev)));
    const unsigned nf = hd.unf[(level + 1) & 1];
    note("  -> built", nf, 0, (long long)hd.ufedges[(level + 1) & 1]);
    if (nf == 0) break;
    frontier_edges = (int64_t)hd.ufedges[(level + 1) & 1];
    remaining -= frontier_edges;
    if (remaining < 0) remaining = 0;
    off += fsize;
    fsize = nf;
    total += nf;
    level++;
  }
  *b.claim_clean = true;
  if (dbg_check) {
    unsigned *chk = nullptr;
    SBX_TRY(sbx_salloc(h, 32, &chk));
    SBX_HIP(h, hipMemsetAsync(chk, 0, 32 * sizeof(unsigned), h->stream));
    SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_check_closed, dim3(1024), dim3(256), b.rp, b.col, (const unsigned *)b.vbits,
                (const unsigned *)dist, b.n, chk);
    unsigned hc[32];
    SBX_TRY(sbx_readback(h, hc, chk, sizeof(hc)));
    if (hc[0]) {
      fprintf(stderr, "[rcm check] unordered sweep left %u edges open; root %d, total %u, levels %u\n", hc[0], (int)fixed_root, total, level + 1);
      for (unsigned k = 0; k < (hc[0] < 8 ? hc[0] : 8); k++)
        fprintf(stderr, "   visited %u (level %u) -> unvisited %u\n", hc[1 + 3 * k], hc[3 + 3 * k], hc[2 + 3 * k]);
      for (auto &t : trace) fprintf(stderr, "   %s\n", t.c_str());
    }
  }
  out->count = total;
  out->levels = level + 1;
  out->last_offset = off;
  out->last_size = fsize;
  return SBX_OK;
}

// the next candidate root after an unordered sweep: first vertex in queue order among the deepest level's vertices of
// smallest degree (see the comment above k_ubfs_start); left in dv->root
__global__ void k_clear_spec_guard(RcmDev *__restrict__ dv) { dv->spec_guard = 0; }
static bool rcm_tie_spec() {  // SBX_RCM_TIE_SPEC=0: the host looks at every tie walk's outcome before it enqueues the next sweep
  static const bool on = !(sbx_env_test("SBX_RCM_TIE_SPEC") && atoi(sbx_env_test("SBX_RCM_TIE_SPEC")) == 0);
  return on;
}
static int ubfs_pick_root_slow(sbx_handle_t h, const BfsBuffers &b, unsigned *cone, const BfsResult &r, bool *aborted,
                               bool clear_guard);
// allow_unverified: the caller's next sweep can be redone (run_ubfs / run_bfs with spec_failed): a tie walk is then left
// unverified (*b.tie_unverified) and the next sweep is enqueued straight behind it
static int ubfs_pick_root(sbx_handle_t h, const BfsBuffers &b, unsigned *cone, const BfsResult &r, bool *aborted,
                          bool allow_unverified) {
  *aborted = false;
  const I *last = b.q + r.last_offset;
  I *list = b.nf_list;  // free during an unordered sweep: the marked vertices, level after level, one growing list
  // (the cone bitmap was cleared by the sweep's start kernel)
  if (r.last_size <= UB_TIES_SMALL) {
    // (the candidates' workgroup goes on to the whole tie-break where the cone is small: tie_walk; the two kernels behind
    // it leave at once unless it handed a step over to them)
    TieWalkState *tw = nullptr;
    SBX_TRY(sbx_salloc(h, 1, &tw));
    static const bool tw_dbg = sbx_env_test("SBX_DEBUG_TIE_WALK") != nullptr;
    const bool spec = allow_unverified && rcm_tie_walk() && rcm_tie_spec() && !tw_dbg && b.tie_unverified;
    SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_ubfs_ties_small, dim3(1), dim3(1024), b.rp, last, r.last_size, cone, list, b.dv, b.col,
                (const unsigned *)b.vbits, (const unsigned *)b.lpos, rcm_tie_walk() ? r.levels : 0u, tie_walk_edges(),
                tie_walk_cap(), tie_walk_single(), tw, spec ? 1u : 0u);
    if (rcm_tie_walk()) {
      SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_tie_heavy_min, dim3((unsigned)h->num_cus), dim3(256), b.rp, b.col,
                  (const unsigned *)b.vbits, (const unsigned *)b.lpos, (const I *)list, tw);
      SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_tie_heavy_adj, dim3((unsigned)h->num_cus), dim3(256), b.rp, b.col, (const I *)list, tw);
      SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_tie_walk_resume, dim3(1), dim3(1024), b.rp, b.col, (const unsigned *)b.vbits,
                  (const unsigned *)b.lpos, list, b.dv, tw, tie_walk_edges(), tie_walk_cap(), tie_walk_single());
      // Usually that was the whole tie-break — so the next sweep goes out behind it unseen: its first kernels leave if the
      // walk did (RcmDev::spec_guard), its first read-back tells, and the caller then comes back for the persistent
      // kernels (ubfs_pick_root_slow).  One round trip per tie-break saved.
      SBX_LAUNCH_CHECK(h);
      if (spec) {
        *b.tie_unverified = true;
        return SBX_OK;
      }
      // Otherwise the host looks before it enqueues the persistent kernels (a round trip more where the walk left — rare —
      // against two launches that find nothing to do in every other tie-break)
      // (no bfs_first_launch here: the three kernels are over in tens of microseconds, the hook's enqueues — the other
      // components' labelling, a dozen launches — would hold the read-back up with the caller's stream idle; the next
      // sweep's first launch takes the hook along behind a whole chain of levels)
      RcmDev hw;
      SBX_TRY(sbx_readback(h, &hw, b.dv, sizeof(RcmDev)));
      if (tw_dbg)  // (exits: see TW_LEAVE in tie_walk; 0: it named the root)
        fprintf(stderr, "[rcm tie walk] levels %u, candidates %u: exit %u at level %u (%u)\n", r.levels, hw.tie_count,
                hw.tie_walk_exit & 0xFFu, hw.tie_walk_exit >> 8, hw.tie_walk_arg);
      if (hw.tie_done) return SBX_OK;
    }
  } else {
    const unsigned g = sbx_grid_for(r.last_size, 256, 1024);
    SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_ubfs_ties_init, dim3(1), dim3(1), b.dv);
    SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_ubfs_min_degree, dim3(g), dim3(256), b.rp, last, r.last_size, b.dv);
    SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_ubfs_mark_ties, dim3(g), dim3(256), b.rp, last, r.last_size, cone, list, b.dv);
  }
  return ubfs_pick_root_slow(h, b, cone, r, aborted, false);
}
// the tie-break by the persistent cone kernels, behind the kernels that found and marked the candidates (clear_guard: ...
// and behind a next sweep whose first kernels left: RcmDev::spec_guard)
static int ubfs_pick_root_slow(sbx_handle_t h, const BfsBuffers &b, unsigned *cone, const BfsResult &r, bool *aborted,
                               bool clear_guard) {
  *aborted = false;
  I *list = b.nf_list;
  if (clear_guard) SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_clear_spec_guard, dim3(1), dim3(1), b.dv);
  SBX_LAUNCH_CHECK(h);
  RcmDev hd;
  if (r.last_size > UB_TIES_SMALL) {
    // (the one-workgroup kernel of the small case names a single candidate itself: no round trip for the count)
    SBX_TRY(bfs_first_launch(b, 20));
    SBX_TRY(sbx_readback(h, &hd, b.dv, sizeof(RcmDev)));
    if (hd.nf <= 1) {
      SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_ubfs_root_from_single_tie, dim3(1), dim3(1), b.dv);
      SBX_LAUNCH_CHECK(h);
      return SBX_OK;
    }
  }
  // T_{k-1} = the level-(k-1) neighbours of T_k: the expansion kernel in cone mode appends them to the same list (its
  // counter keeps growing); the hub kernel only runs for a level that queued hub chunks
  const unsigned max_grid = (unsigned)h->num_cus * 8;
  static int heavy_per_cu = 0;
  if (heavy_per_cu == 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_bfs_expand_heavy<2>, 256, 0) != hipSuccess || nb < 1) nb = 4;
    heavy_per_cu = nb;
  }
  const unsigned heavy_grid = (unsigned)h->num_cus * (unsigned)heavy_per_cu;
  // Small lists — most of them — are walked by the persistent kernel, several levels per launch; a level whose list is
  // too long for it goes through the expansion kernels in their cone mode (the range of the list stays on the device).
  if (r.last_size > UB_TIES_SMALL) SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_ubfs_cone_next, dim3(1), dim3(1), b.dv, 1);
  const unsigned cone_grid = max_grid < 512u ? max_grid : 512u;
  for (unsigned k = r.levels - 1; k >= 2;) {  // level 0 is the root: every T_1 member hangs under it
    SBX_KLAUNCH(h, SBX_K_BFS_SMALL, k_ubfs_cone_run, dim3(UR_GRID), dim3(256), b.rp, b.col, (const unsigned *)b.vbits,
                (const unsigned *)b.lpos, cone, list, (I *)b.heavy, b.dv, k);
    // the walk down the cone goes out behind it at once: it runs if the cone is finished (the usual end of this loop) and
    // leaves otherwise — one round trip for both instead of one each
    SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_ubfs_descend_all, dim3(UB_DESC_GRID), dim3(256), b.rp, b.col,
                (const unsigned *)b.vbits, (const unsigned *)b.lpos, (const unsigned *)cone, r.levels, b.dv, 1);
    SBX_LAUNCH_CHECK(h);
    SBX_TRY(bfs_first_launch(b, 40));
    SBX_TRY(sbx_readback(h, &hd, b.dv, sizeof(RcmDev)));  // (the walk must be known to have finished: see gb_wait)
    if (hd.gb_abort) {
      SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_gb_reset, dim3(1), dim3(1), b.dv);
      h->rcm_gb_backoff = gb_backoff_calls();
      *aborted = true;
      return SBX_OK;
    }
    if (hd.cone_status == UR_DONE || hd.unsym) return SBX_OK;
    k = hd.cone_k;  // its list is long: one level with the big kernels, then the persistent one again
    const UnorderedSweep us = {nullptr, cone, b.lpos, k - 1};
    SBX_KLAUNCH(h, SBX_K_BFS_EXPAND, k_bfs_expand<2>, dim3(cone_grid), dim3(256), b.rp, b.col, (const I *)list, 0u, k,
                (const unsigned *)b.vbits, b.ppos, list, b.heavy, b.hub_dir, b.dv, us, 0);
    if (b.max_deg > (unsigned)RCM_LIGHT)
      SBX_KLAUNCH(h, SBX_K_BFS_HEAVY, k_bfs_expand_heavy<2>, dim3(heavy_grid), dim3(256), b.rp, b.col, (const I *)list, k,
                  (const unsigned *)b.vbits, b.ppos, list, (const uint64_t *)b.heavy, (const uint2 *)b.hub_dir, cone_grid,
                  b.dv, us);
    SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_ubfs_cone_next, dim3(1), dim3(1), b.dv, 0);
    k--;
  }
  SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_ubfs_descend_all, dim3(UB_DESC_GRID), dim3(256), b.rp, b.col,
              (const unsigned *)b.vbits, (const unsigned *)b.lpos, (const unsigned *)cone, r.levels, b.dv, 0);
  SBX_LAUNCH_CHECK(h);
  SBX_TRY(bfs_first_launch(b, 40));
  SBX_TRY(sbx_readback(h, &hd, b.dv, sizeof(RcmDev)));  // (the walk must be known to have finished: see gb_wait)
  if (hd.gb_abort) {
    SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_gb_reset, dim3(1), dim3(1), b.dv);
    h->rcm_gb_backoff = gb_backoff_calls();
    *aborted = true;
  }
  return SBX_OK;
}

// parent positions back to UNSEEN after a sweep: per reached vertex, or — when the sweep reached a good part of
// the graph — by refilling the whole array (16 MB stream at HBM speed against millions of scattered stores)
static int reset_ppos(sbx_handle_t h, const I *q, unsigned count, unsigned *ppos, int64_t n) {
  if ((int64_t)count >= n / 16) {
    SBX_HIP(h, hipMemsetAsync(ppos, 0xFF, (size_t)n * sizeof(unsigned), h->stream));
  } else {
    SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_reset_visited, dim3(sbx_grid_for(count, 256, 4096)), dim3(256), q, count, ppos);
    SBX_LAUNCH_CHECK(h);
  }
  return SBX_OK;
}

}  // namespace

int sbx_rcm_reorder_x32(sbx_handle_t h, int64_t n, int64_t nnz, const void *row_ptr, const void *col_v,
                        void *inv_perm_out, sbx_rcm_stats *stats_host);
int sbx_rcm_reorder_x64(sbx_handle_t h, int64_t n, int64_t nnz, const void *row_ptr, const void *col_v,
                        void *inv_perm_out, sbx_rcm_stats *stats_host);
#ifndef SBX_RCM_I64
extern "C" int sbx_rcm_reorder(sbx_handle_t h, sbx_index_type it, int64_t n, int64_t nnz, const void *row_ptr,
                               const void *col_v, void *inv_perm_out, sbx_rcm_stats *stats_host) {
  if (!h) return SBX_ERR_BAD_ARG;
  if (it == SBX_I32_N64) return sbx_mixed_rcm_reorder(h, n, nnz, row_ptr, col_v, inv_perm_out, stats_host);
  if (n < 0 || nnz < 0 || !row_ptr || (n > 0 && !inv_perm_out) || (nnz > 0 && !col_v))
    SBX_FAIL(h, SBX_ERR_BAD_ARG, "sbx_rcm_reorder: bad argument");
  // (ids, offsets and queue positions are 32-bit inside, whatever the width of the arrays)
  if (n >= ((int64_t)1 << 31) - 1 || nnz >= ((int64_t)1 << 31))
    SBX_FAIL(h, it == SBX_I64 ? SBX_ERR_UNSUPPORTED : SBX_ERR_BAD_ARG, "sbx_rcm_reorder: dimension exceeds int32");
  return it == SBX_I64 ? sbx_rcm_reorder_x64(h, n, nnz, row_ptr, col_v, inv_perm_out, stats_host)
                       : sbx_rcm_reorder_x32(h, n, nnz, row_ptr, col_v, inv_perm_out, stats_host);
}
#endif

int SBX_RCM_ENTRY(sbx_handle_t h, int64_t n, int64_t nnz, const void *row_ptr, const void *col_v, void *inv_perm_out,
                  sbx_rcm_stats *stats_host) {
  if (stats_host) memset(stats_host, 0, sizeof(*stats_host));
  SBX_TRY(sbx_arena_begin(h));
  if (n == 0) return SBX_OK;
  const X *rp = (const X *)row_ptr, *col = (const X *)col_v;
  X *inv = (X *)inv_perm_out;

  RcmDev *dv;
  uint32_t *did_a, *drank;
  I *label, *csize, *cbase, *small_list, *mid_list, *large_list, *big_list, *q, *nf_list;
  unsigned *dist, *ppos, *vbits, *fbits, *cbits, *lpos;
  I *q_small;
  uint64_t *ka, *kb, *heavy;
  SBX_TRY(sbx_salloc(h, 1, &dv));
  SBX_TRY(sbx_salloc(h, (size_t)n, &did_a));
  SBX_TRY(sbx_salloc(h, (size_t)n, &drank));
  SBX_TRY(sbx_salloc(h, (size_t)n, &label));
  SBX_TRY(sbx_salloc(h, (size_t)n + 1, &csize));
  SBX_TRY(sbx_salloc(h, (size_t)n + 1, &cbase));
  SBX_TRY(sbx_salloc(h, (size_t)n, &small_list));
  SBX_TRY(sbx_salloc(h, (size_t)n, &large_list));
  SBX_TRY(sbx_salloc(h, (size_t)n / (RCM_SMALL + 1) + 1, &mid_list));
  SBX_TRY(sbx_salloc(h, (size_t)n, &big_list));
  SBX_TRY(sbx_salloc(h, (size_t)n, &q));
  SBX_TRY(sbx_salloc(h, (size_t)n, &nf_list));
  SBX_TRY(sbx_salloc(h, (size_t)n, &dist));
  SBX_TRY(sbx_salloc(h, (size_t)n, &ppos));
  // the three bitmaps share one allocation (64-bit aligned pieces): vbits and fbits are cleared together
  const size_t bm_words = (((size_t)(n + 31) / 32 + 1) + 1) & ~(size_t)1;
  SBX_TRY(sbx_salloc(h, 3 * bm_words, &vbits));
  fbits = vbits + bm_words;
  cbits = fbits + bm_words;
  SBX_TRY(sbx_salloc(h, (size_t)n, &q_small));
  SBX_TRY(sbx_salloc(h, (size_t)n, &lpos));
  const size_t heavy_cap = (size_t)(nnz / RCM_LIGHT + nnz / RCM_CHUNK + 1024);
  SBX_TRY(sbx_salloc(h, heavy_cap, &heavy));
  SBX_TRY(sbx_salloc(h, (size_t)n, &ka));
  SBX_TRY(sbx_salloc(h, (size_t)n, &kb));
  // (dv is cleared by k_deg_reduce; csize, dist and ppos get their initial values from k_deg_count)

  const unsigned gn = sbx_grid_for(n, 256, 8192);
  const size_t bm_bytes = (size_t)((n + 31) / 32) * sizeof(unsigned);
  // (1) global (degree,id) rank used by the Cuthill-McKee keys; first non-isolated vertex
  unsigned *ucnt = nullptr;
  SBX_TRY(sbx_salloc(h, (size_t)DEG_UNITS * 4, &ucnt));  // per unit: non-empty rows, largest degree, first non-empty vertex, rows of 255+ entries
  SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_deg_count, dim3(DEG_UNITS / 4), dim3(256), rp, n, ucnt, csize, dist, ppos);
  SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_deg_reduce, dim3(1), dim3(1024), (const unsigned *)ucnt, dv);
  SBX_LAUNCH_CHECK(h);
  // after a grid barrier gave up (a GPU shared with another process, see gb_wait) the next few calls on this handle do
  // not try the persistent kernels again
  const bool unordered_ok = rcm_unordered() && h->rcm_gb_backoff == 0;
  if (h->rcm_gb_backoff > 0) h->rcm_gb_backoff--;
  // The degree counts (non-empty rows, largest degree, first non-empty vertex) are read back with the first sweep's first
  // round trip when that sweep is an unordered one: it starts from dv->root, which k_deg_reduce set, and needs nothing
  // else the host does not know yet.  (~17 us per call: a read-back kernel and a launch gap.)
  // (only with the side stream: without one the degree ranks are enqueued before the first sweep and need the counts)
  const bool lazy_counts = unordered_ok && !h->prof_on && rcm_overlap();
  RcmDev hd0;
  bool hd0_ready = false;
  if (!lazy_counts) {
    SBX_TRY(sbx_readback(h, &hd0, dv, sizeof(RcmDev)));
    hd0_ready = true;
  }
  int64_t n_ranked = hd0_ready ? (int64_t)hd0.n_nonempty : 0;  // vertices that get a degree rank
  // Only the Cuthill-McKee sweep reads the degree ranks: they are built on a side stream while the plain sweeps —
  // launch- and latency-bound — run on the caller's stream, and joined before the first Cuthill-McKee sweep.
  const uint32_t *dorder = nullptr;
  hipStream_t main_stream = h->stream;
  const bool side = !h->prof_on && rcm_overlap();
  BfsBuffers b;
  if (side) {  // (recorded here: the side stream waits for the counts above, not for the sweep enqueued before its work)
    SBX_TRY(sbx_aux_streams(h));
    SBX_HIP(h, hipEventRecord(h->aux_event[0], main_stream));
  }
  bool ranks_enqueued = false;
  bool cc_forked = false;  // the labelling of the other components is running on a side stream of its own (below)
  std::function<int()> enqueue_ranks = [&]() -> int {
    if (ranks_enqueued) return SBX_OK;
    ranks_enqueued = true;
    if (side) {
      void *slot = nullptr;
      SBX_TRY(sbx_arena_alloc(h, SBX_RS_SLOT_BYTES, &slot));
      SBX_HIP(h, hipStreamWaitEvent(h->aux_stream[0], h->aux_event[0], 0));
      h->aux_dirty = true;
      h->stream = h->aux_stream[0];
      if (hipMemsetAsync(slot, 0, SBX_RS_SLOT_BYTES, h->stream) != hipSuccess) {
        h->stream = main_stream;
        SBX_FAIL(h, SBX_ERR_HIP, "sbx_rcm_reorder: hipMemsetAsync failed");
      }
      h->rs_override = slot;
    }
    int rc = SBX_OK;
    // one stable counting pass on min(degree, 255) over the rows in id order + the last bucket by its full degree
    // (sbx_degree.hip): 0.09 ms where a generic sort of (degree, id) pairs took 0.31
    // (hd0 — not the locals derived from it: this hook may run inside the first sweep, right behind the read-back that
    // delivered the counts)
    rc = sbx_degree_ranks(h, SBX_RCM_IT, rp, n, (int64_t)hd0.n_nonempty, (int64_t)hd0.n_top, hd0.max_deg, drank, did_a);
    dorder = did_a;
    b.dorder = dorder;
    if (side && rc == SBX_OK && hipEventRecord(h->aux_event[1], h->stream) != hipSuccess) rc = SBX_ERR_HIP;
    h->stream = main_stream;
    h->rs_override = nullptr;
    return rc;
  };
  // with a side stream the first sweep's first kernel is enqueued first (the host's ~15 enqueues for the ranks then
  // run while that kernel does); without one the ranks simply come first
  if (!side) SBX_TRY(enqueue_ranks());
  bool ranks_joined = !side;
  auto join_ranks = [&]() -> int {
    if (!ranks_enqueued) SBX_TRY(enqueue_ranks());
    if (!ranks_joined) {
      SBX_HIP(h, hipStreamWaitEvent(main_stream, h->aux_event[1], 0));
      ranks_joined = true;
      if (!cc_forked) h->aux_dirty = false;
    }
    return SBX_OK;
  };
  bool claim_clean = false;
  b.claim_clean = &claim_clean;
  b.after_first_launch = &enqueue_ranks;
  b.hook_wants_long_cover = true;  // (~15 launches)
  b.head_chain = true;  // (until the first sweep has run)
  bool tie_unverified = false;
  b.tie_unverified = &tie_unverified;
  b.side_stage = nullptr, b.side_event = nullptr, b.side_joined = nullptr, b.last_read = nullptr, b.last_read_joined = nullptr;
  b.rp = rp; b.col = col; b.vbits = vbits; b.fbits = fbits; b.lpos = lpos; b.ppos = ppos; b.label = nullptr;
  b.nnz = nnz; b.q = q; b.nf_list = nf_list; b.heavy = heavy; b.heavy_cap = heavy_cap;
  SBX_TRY(sbx_salloc(h, (size_t)std::max<int64_t>((int64_t)h->num_cus * 8, RCM_DIR_MAX), &b.hub_dir));
  b.ka = ka; b.kb = kb; b.drank = drank; b.dorder = dorder; b.n_ranked = n_ranked; b.dv = dv; b.n = n;
  SBX_TRY(sbx_salloc(h, (size_t)(n + 63) / 64 + 1, &b.fresh64));
  SBX_TRY(sbx_salloc(h, (size_t)(n + 63) / 64 + 1, &b.wcnt));
  SBX_TRY(sbx_salloc(h, (size_t)(n + 63) / 64 / RCM_FW_WORDS + 2, &b.woff));
  SBX_TRY(sbx_salloc(h, sbx_cs::scratch_words(std::min<int64_t>(n, RCM_COUNT_SORT_MAX)), &b.cs_scratch));
  b.max_deg = hd0_ready ? hd0.max_deg : 0u;
  b.late_counts = hd0_ready ? nullptr : &hd0;
  b.late_counts_ready = &hd0_ready;
  // (2) The smallest non-isolated vertex v0 is the smallest id of its component, i.e. the
  // start of that component's pseudo-peripheral search.  Its first BFS sweep is needed
  // anyway and yields the component's membership for free, so the union-find below only
  // has to label what that sweep did not reach (for power-law inputs: a sliver).
  BfsResult r0;
  r0.count = 0;
  // sweeps of the pseudo-peripheral search run unordered (level sets only, run_ubfs) unless the component turns out
  // deep and narrow
  unsigned char *claim8 = nullptr;
  unsigned *cone = nullptr, *nbits = nullptr;
  SBX_TRY(sbx_salloc(h, (size_t)n + 64, &claim8));
  SBX_TRY(sbx_salloc(h, (size_t)(bm_bytes / sizeof(unsigned)) + 2, &nbits));
  SBX_TRY(sbx_salloc(h, (size_t)(bm_bytes / sizeof(unsigned)) + 2, &cone));
  bool r0_unordered = false;
  bool deep0 = true;
  int64_t unordered_sweeps = 0;  // sweeps that kept the level sets only (statistics; the tests check the path was taken)
  if (!hd0_ready) {  // the first sweep, from the root on the device; the counts come back with its first read-back
    SBX_TRY(run_ubfs(h, b, claim8, nbits, cone, (I)-1, (I)-1, &r0, &deep0));
    if (!hd0_ready) SBX_FAIL(h, SBX_ERR_INTERNAL, "sbx_rcm_reorder: the first sweep returned without a read-back");
    n_ranked = (int64_t)hd0.n_nonempty;
    b.n_ranked = n_ranked;
    b.max_deg = hd0.max_deg;
    b.late_counts = nullptr;
  }
  if (hd0_ready && lazy_counts) b.head_chain = false;
  const I v0 = hd0.first_vertex == UNSEEN ? (I)-1 : (I)hd0.first_vertex;
  if (v0 < 0) {  // not one edge: the sweep above (if any) started from an empty row and means nothing
    r0.count = 0;
    deep0 = true;
    SBX_HIP(h, hipMemsetAsync(cbits, 0, bm_bytes, h->stream));  // (else: a copy of the first sweep's visited bitmap)
  }
  if (v0 >= 0) {
    bool deep = deep0;
    if (!lazy_counts && unordered_ok) SBX_TRY(run_ubfs(h, b, claim8, nbits, cone, v0, (I)-1, &r0, &deep));
    b.head_chain = false;
    r0_unordered = !deep;
    unordered_sweeps += r0_unordered;
    if (deep) SBX_TRY(run_bfs<false>(h, b, v0, (I)-1, &r0));
    SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_copy_words, dim3(sbx_grid_for((int64_t)(bm_bytes / sizeof(unsigned)), 256, 1024)),
                dim3(256), cbits, (const unsigned *)vbits, (int64_t)(bm_bytes / sizeof(unsigned)));
    SBX_LAUNCH_CHECK(h);
  }
  // Pseudo-peripheral search and Cuthill-McKee sweep of one host-ordered component (root = its smallest vertex): leaves
  // the component's order in q.
  struct Searched {
    BfsResult r;
    int64_t sweeps, levels, candidate;
    bool cm_done;
  };
  Searched sd0;
  bool comp0_searched = false, small_on_side = false;
  auto search_component = [&](I root, I size, bool have_first_sweep, Searched &out) -> int {
    BfsResult &r = out.r;
    int64_t prev_ecc = -1, ecc = 0;
    int64_t &sweeps = out.sweeps, &levels = out.levels, &candidate = out.candidate;
    sweeps = levels = candidate = 0;
    I fixed = root;
    bool &cm_done = out.cm_done;
    cm_done = false;
    bool deep = !unordered_ok;  // a sweep of more than UB_MAX_LEVELS levels: this component keeps ordered sweeps
    while (prev_ecc != ecc) {
      prev_ecc = ecc;
      bool unordered = false;  // the sweep just run kept no order inside its levels
      if (have_first_sweep) {
        r = r0;  // sweep (2) above was exactly this component's first sweep
        have_first_sweep = false;
        unordered = r0_unordered;
        if (!unordered) deep = true;
      } else {
        // (a sweep behind an unverified tie walk comes back at once when the walk had left: the persistent kernels then
        // finish the tie-break on the sweep r still describes — exactly what followed the tie-break before — and the
        // sweep is enqueued again)
        auto settle_tie = [&]() -> int {
          bool ab = false;
          SBX_TRY(ubfs_pick_root_slow(h, b, cone, r, &ab, true));
          if (ab) {  // a grid barrier of the tie-break gave up: the same sweep again, ordered (dv->root still is its root)
            deep = true;
            SBX_TRY(run_bfs<false>(h, b, -1, root, &r));
            SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_pick_root, dim3(sbx_grid_for(r.last_size, 256, 1024)), dim3(256), rp,
                        (const I *)(q + r.last_offset), r.last_size, dv);
            SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_set_root_from_best, dim3(1), dim3(1), (const I *)(q + r.last_offset), dv);
            SBX_TRY(reset_ppos(h, (const I *)q, r.count, ppos, n));
          }
          return SBX_OK;
        };
        if (candidate >= rcm_speculate_from()) {
          SBX_TRY(join_ranks());
          bool sf = false;
          SBX_TRY(run_bfs<true>(h, b, fixed, root, &r, &sf));
          if (sf) {
            SBX_TRY(settle_tie());
            SBX_TRY(run_bfs<true>(h, b, (I)-1, root, &r));
          }
          sweeps++;
          levels += r.levels;
          const int64_t e = (int64_t)r.levels - 1;
          const int64_t deepest = e > ecc ? e : ecc;
          if (deepest == ecc || (int64_t)r.count == deepest + 1) {
            cm_done = true;  // this candidate is the root and q already holds its Cuthill-McKee order
            break;
          }
          SBX_TRY(reset_ppos(h, (const I *)q, r.count, ppos, n));
          fixed = -1;  // k_bfs_start left the root on the device
        }
        if (!deep) {
          bool sf = false;
          SBX_TRY(run_ubfs(h, b, claim8, nbits, cone, fixed, root, &r, &deep, &sf));
          if (sf) {
            SBX_TRY(settle_tie());
            if (!deep) SBX_TRY(run_ubfs(h, b, claim8, nbits, cone, (I)-1, root, &r, &deep));
          }
          unordered = !deep;
          unordered_sweeps += unordered;
        }
        if (deep) SBX_TRY(run_bfs<false>(h, b, fixed, root, &r));
      }
      fixed = -1;  // later sweeps start from the device-resident root
      candidate++;
      sweeps++;
      levels += r.levels;
      const int64_t e = (int64_t)r.levels - 1;
      if (e > ecc) ecc = e;
      const bool path = (int64_t)r.count == ecc + 1;
      if (!path && prev_ecc != ecc) {
        bool tie_aborted = false;
        if (unordered) SBX_TRY(ubfs_pick_root(h, b, cone, r, &tie_aborted, true));
        if (unordered && tie_aborted) {
          // a grid barrier of the tie-break gave up: the same sweep again, ordered (dv->root still is its root)
          deep = true;
          unordered = false;
          SBX_TRY(run_bfs<false>(h, b, -1, root, &r));
        }
        if (!unordered) {
          SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_pick_root, dim3(sbx_grid_for(r.last_size, 256, 1024)), dim3(256), rp,
                             (const I *)(q + r.last_offset), r.last_size, dv);
          SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_set_root_from_best, dim3(1), dim3(1), (const I *)(q + r.last_offset), dv);
        }
      }
      if (!unordered) SBX_TRY(reset_ppos(h, (const I *)q, r.count, ppos, n));
      if (path) break;
    }
    // Cuthill-McKee BFS from the pseudo-peripheral vertex (:118-144)
    if (!cm_done) {
      SBX_TRY(join_ranks());
      SBX_TRY(run_bfs<true>(h, b, -1, root, &r));
      sweeps++;
      levels += r.levels;
    }
    if ((int64_t)r.count != (int64_t)size)
      SBX_FAIL(h, SBX_ERR_BAD_ARG,
               "sbx_rcm_reorder: BFS reached %u of %d vertices of a component (pattern not symmetric?)", r.count,
               size);
    return SBX_OK;
  };
  // (3) connected components of the rest; the root of each tree is the component's smallest id.  When the first sweep
  // reached every non-empty row there is no rest: one kernel places the empty rows and the stage ends (a power-law
  // input: ~0.2 ms of union-find, size scan, classification and two round trips saved)
  const bool single_component = v0 >= 0 && (int64_t)r0.count == n_ranked && r0.count > (unsigned)RCM_MID;
  RcmDev hd;
  memset(&hd, 0, sizeof(hd));
  bool mid_batched = false;
  unsigned n_host = 0;
  bool first_is_large = false;
  if (single_component) {
    SBX_KLAUNCH(h, SBX_K_CC, k_iso_positions, dim3(DEG_UNITS / 4), dim3(256), rp, n, (const unsigned *)ucnt, v0, (I)r0.count,
                inv);
    SBX_LAUNCH_CHECK(h);
    hd.n_large = 1;
    hd.n_components = (unsigned)(n - n_ranked) + 1u;
    hd.n_empty_rows = (unsigned)(n - n_ranked);
    n_host = 1;
    first_is_large = true;
  } else {
  // The first sweep's component, when it is one the host orders anyway (more than RCM_MID vertices), does not wait for
  // the labelling of the rest: union-find, size scan and classification — streaming passes over all n vertices, 0.13 ms
  // on the bench matrix — run on a side stream while the caller's stream goes on with that component's search and
  // Cuthill-McKee sweep, which are bound by launches and latency, not by bandwidth.  The search needs no labels (a sweep
  // cannot leave its component; unlabelled, the bottom-up levels merely look at the other components' vertices too).
  cc_forked = side && v0 >= 0 && r0.count > (unsigned)RCM_MID && r0_unordered && rcm_cc_overlap();
  bool counters_read = false;
  // (in two halves: on the side stream each half is enqueued behind one of the tie-break's ~45 us kernels)
  auto enqueue_cc_kernels = [&](int half) -> int {
  const unsigned gcount = gn < 1024u ? gn : 1024u;
  if (half != 1) {
  SBX_KLAUNCH(h, SBX_K_CC, k_cc_init, dim3(gn), dim3(256), rp, col, label, n, (const unsigned *)cbits);
  SBX_KLAUNCH(h, SBX_K_CC, k_cc_hook_small, dim3(gn), dim3(256), rp, col, label, n, big_list,
              (const unsigned *)cbits, dv);
  SBX_KLAUNCH(h, SBX_K_CC, k_cc_hook_big, dim3((unsigned)h->num_cus * 8), dim3(256), rp, col, label,
              (const I *)big_list, (const RcmDev *)dv);
  // (phase 0 and the classification end in one counter add per workgroup: few workgroups, or the counter word is the
  // bottleneck — 16 K adds on one word cost 0.19 ms)
  for (int phase = 0; phase < 2; phase++)
    SBX_KLAUNCH(h, SBX_K_CC, k_cc_finalize, dim3(phase == 0 ? gcount : gn), dim3(256), rp, label, csize, n,
                (const unsigned *)cbits, v0 >= 0 ? v0 : (I)0, (I)r0.count, phase, dv);
  SBX_LAUNCH_CHECK(h);
  }
  if (half != 0) {
  SBX_TRY(sbx_exclusive_scan_i32(h, csize, cbase, n + 1, nullptr));
  SBX_KLAUNCH(h, SBX_K_CC, k_classify, dim3(gcount), dim3(256), (const I *)label, (const I *)csize, (const I *)cbase, inv,
              small_list, mid_list, large_list, n, dv);
  SBX_LAUNCH_CHECK(h);
  }
  return SBX_OK;
  };
  if (cc_forked) {
    // The side stream's ten launches are enqueued behind the search's first kernels (the hook every sweep and tie-break
    // calls before its first read-back): the host needs ~50 us for them, which the caller's stream would spend idle.
    SBX_HIP(h, hipEventRecord(h->aux_event[2], main_stream));
    h->aux_dirty = true;
    int cc_stage = 0;  // halves of the labelling enqueued so far
    bool cc_joined = false, last_read_joined = false;
    RcmDev last_read;
    b.side_stage = &cc_stage, b.side_event = h->aux_event[3], b.side_joined = &cc_joined;
    b.last_read = &last_read, b.last_read_joined = &last_read_joined;
    // (a half per firing, behind any launch: both halves at one go behind a chain of big levels only — tried when the
    // tie-breaks stopped being 80 us of persistent kernels to hide behind — looked better under the profiler, whose
    // launches take 10 us of host time each, and cost 28 us on the wall clock: the side work simply started later)
    std::function<int()> enqueue_cc = [&]() -> int {
      SBX_TRY(enqueue_ranks());
      if (cc_stage >= 2) return SBX_OK;
      if (cc_stage == 0) SBX_HIP(h, hipStreamWaitEvent(h->aux_stream[1], h->aux_event[2], 0));
      h->stream = h->aux_stream[1];
      int rc = enqueue_cc_kernels(cc_stage);
      if (rc == SBX_OK && cc_stage == 1) {  // the small components too: their number stays on the device, the launch covers any
        const unsigned lanes = (unsigned)std::min<int64_t>((n / 2 + 63) / 64 * 64, (int64_t)h->num_cus * 32 * 64);
        SBX_KLAUNCH(h, SBX_K_RCM_SMALL, k_rcm_small, dim3(lanes / 64 > 0 ? lanes / 64 : 1), dim3(64), rp, col,
                    (const I *)small_list, 0u, (const I *)csize, (const I *)cbase, dist, q_small, inv, dv,
                    (const unsigned *)&dv->n_small);
        if (hipGetLastError() != hipSuccess) rc = SBX_ERR_HIP;
        if (rc == SBX_OK && hipEventRecord(h->aux_event[3], h->stream) != hipSuccess) rc = SBX_ERR_HIP;
      }
      cc_stage++;
      h->stream = main_stream;
      return rc;
    };

    b.after_first_launch = &enqueue_cc;
    b.hook_wants_long_cover = false;  // (half a dozen launches per call)
    const int src = search_component(v0, (I)r0.count, true, sd0);
    b.after_first_launch = &enqueue_ranks;
    b.hook_wants_long_cover = true;
    b.side_stage = nullptr, b.side_joined = nullptr, b.last_read = nullptr, b.last_read_joined = nullptr;
    if (src != SBX_OK) {  // (the side stream must not be left waiting for a half that never comes)
      h->stream = main_stream;
      return src;
    }
    SBX_TRY(enqueue_cc());  // (a search that never waited for anything)
    SBX_TRY(enqueue_cc());
    comp0_searched = true;
    if (!cc_joined) SBX_HIP(h, hipStreamWaitEvent(main_stream, h->aux_event[3], 0));
    cc_forked = false;
    if (ranks_joined || !ranks_enqueued) h->aux_dirty = false;
    small_on_side = true;
    // (the search's last sweep joined the side stream at its start and read *dv behind that: the labelling's counters
    // are in that read-back already)
    if (cc_joined && last_read_joined) hd = last_read, counters_read = true;
  } else {
    SBX_TRY(enqueue_cc_kernels(2));
  }
  if (!counters_read) SBX_TRY(sbx_readback(h, &hd, dv, sizeof(RcmDev)));
  if (comp0_searched) {
    // the component's order is complete in q: written now, its place read from the size scan on the device — BEHIND the
    // read-back of the labelling's counters: when this component is the only one the host orders that read-back was the
    // call's last and the kernel finishes in stream order after the call has returned (outputs are complete in stream
    // order, as for every entry point that does not end in a read-back)
    SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_write_component, dim3(sbx_grid_for(sd0.r.count, 256, 4096)), dim3(256), (const I *)q,
                sd0.r.count, (I)0, inv, (const I *)(cbase + v0));
    SBX_LAUNCH_CHECK(h);
  }
  if (hd.unsym)
    SBX_FAIL(h, SBX_ERR_BAD_ARG,
             "sbx_rcm_reorder: the pattern is not structurally symmetric (a vertex hangs under the first component "
             "without being reachable from it); RCM is defined on symmetric patterns");
  b.label = label;
  // Components of RCM_SMALL + 1 .. RCM_MID vertices: a few of them are ordered like the large ones (level-synchronous
  // sweeps driven from the host: ~60 us per level, fine for a handful); from RCM_MID_BATCH on they join the batched
  // kernel, one lane each — a collection of 10^5 meshes of a few hundred vertices is then one launch instead of
  // 10^5 host-driven searches.
  mid_batched = hd.n_mid >= (unsigned)RCM_MID_BATCH;
  n_host = hd.n_large + (mid_batched ? 0u : hd.n_mid);  // components ordered from the host
  first_is_large = r0.count > (unsigned)RCM_SMALL && !(mid_batched && r0.count <= (unsigned)RCM_MID);
  if (v0 >= 0 && !first_is_large && !r0_unordered) {
    // the pre-swept component is handled by the batched kernel: drop the sweep's marks
    SBX_TRY(reset_ppos(h, (const I *)q, r0.count, ppos, n));
  }
  // (3) small (and, batched, mid-size) components: one lane each
  if (hd.n_small && !small_on_side) {
    SBX_KLAUNCH(h, SBX_K_RCM_SMALL, k_rcm_small, dim3((hd.n_small + 63) / 64), dim3(64), rp, col,
                       (const I *)small_list, hd.n_small, (const I *)csize, (const I *)cbase, dist, q_small, inv, dv,
                       (const unsigned *)nullptr);
    SBX_LAUNCH_CHECK(h);
  }
  if (mid_batched) {
    SBX_KLAUNCH(h, SBX_K_RCM_SMALL, k_rcm_small, dim3((hd.n_mid + 63) / 64), dim3(64), rp, col, (const I *)mid_list,
                hd.n_mid, (const I *)csize, (const I *)cbase, dist, q_small, inv, dv, (const unsigned *)nullptr);
    SBX_LAUNCH_CHECK(h);
  }
  }
  // (4) large components: host-driven level-synchronous BFS
  int64_t sweeps_max = 0, levels_max = 0, largest = 0, ref_sweeps_max = 0;
  if (n_host) {
    std::vector<I> roots(n_host), sizes(n_host), bases(n_host);
    I *info = nullptr;  // roots | sizes | bases, gathered on the device: three copies whatever the component count
    SBX_TRY(sbx_salloc(h, (size_t)3 * n_host, &info));
    if (single_component) {  // root v0; every vertex in front of it has an empty row, i.e. is a component of its own
      roots[0] = v0, sizes[0] = (I)r0.count, bases[0] = v0;
    } else if (comp0_searched && n_host == 1) {  // (already written, its place taken from the device)
      roots[0] = v0, sizes[0] = (I)r0.count, bases[0] = 0;
    } else {
    if (hd.n_large)
      SBX_HIP(h, hipMemcpyAsync(info, large_list, hd.n_large * sizeof(I), hipMemcpyDeviceToDevice, h->stream));
    if (!mid_batched && hd.n_mid)
      SBX_HIP(h, hipMemcpyAsync(info + hd.n_large, mid_list, hd.n_mid * sizeof(I), hipMemcpyDeviceToDevice, h->stream));
    SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_comp_info, dim3((n_host + 255) / 256), dim3(256), (const I *)info, n_host,
                (const I *)csize, (const I *)cbase, info + n_host, info + 2 * (size_t)n_host);
    SBX_LAUNCH_CHECK(h);
    if (n_host <= 64) {  // the usual case, a few components: one polled read-back instead of three copies and a sync
      I tmp[3 * 64];
      SBX_TRY(sbx_readback(h, tmp, info, (size_t)3 * n_host * sizeof(I)));
      for (unsigned c = 0; c < n_host; c++) roots[c] = tmp[c], sizes[c] = tmp[n_host + c], bases[c] = tmp[2 * n_host + c];
    } else {
      SBX_HIP(h, hipMemcpyAsync(roots.data(), info, n_host * sizeof(I), hipMemcpyDeviceToHost, h->stream));
      SBX_HIP(h, hipMemcpyAsync(sizes.data(), info + n_host, n_host * sizeof(I), hipMemcpyDeviceToHost, h->stream));
      SBX_HIP(h, hipMemcpyAsync(bases.data(), info + 2 * (size_t)n_host, n_host * sizeof(I), hipMemcpyDeviceToHost, h->stream));
      SBX_HIP(h, hipStreamSynchronize(h->stream));
    }
    }
    // the pre-swept component goes first: its sweep state (q, ppos) is still live
    for (unsigned c = 1; c < n_host; c++)
      if (first_is_large && roots[c] == v0) {
        std::swap(roots[c], roots[0]);
        std::swap(sizes[c], sizes[0]);
        std::swap(bases[c], bases[0]);
      }
    for (unsigned c = 0; c < n_host; c++) {
      // pseudo-peripheral search from the component's smallest vertex (:22-81)
      //
      // The search ends with the first candidate root whose sweep does not deepen the level
      // structure, and the Cuthill-McKee sweep that follows starts from that same vertex: the
      // two sweeps discover identical levels and differ only in the order inside a level.  So
      // from the third candidate on (the usual place for the search to end) the Cuthill-McKee
      // sweep is run FIRST.  If the structure did not deepen it already is the final sweep —
      // one sweep saved; if it did, the plain sweep is still needed (its order inside the last
      // level breaks the ties for the next candidate) and the speculative one was wasted.
      // Either way the output is the reference's; only the number of sweeps differs.
      Searched sd;
      if (c == 0 && comp0_searched) sd = sd0;
      else SBX_TRY(search_component(roots[c], sizes[c], first_is_large && roots[c] == v0, sd));
      const BfsResult &r = sd.r;
      const int64_t sweeps = sd.sweeps, levels = sd.levels, candidate = sd.candidate;
      const bool cm_done = sd.cm_done;
      if (!(c == 0 && comp0_searched)) {
        SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_write_component, dim3(sbx_grid_for(r.count, 256, 4096)), dim3(256),
                           (const I *)q, r.count, bases[c], inv, (const I *)nullptr);
        SBX_LAUNCH_CHECK(h);
      }
      // invariant of the sweeps (k_visited_from_ppos relies on it): ppos is UNSEEN outside the running sweep
      if (c + 1 < n_host) SBX_TRY(reset_ppos(h, (const I *)q, r.count, ppos, n));
      if (sizes[c] > largest) {
        largest = sizes[c];
        sweeps_max = sweeps;
        levels_max = levels;
        // what the reference's serial algorithm runs on this component: its pseudo-peripheral sweeps (a speculative
        // Cuthill-McKee sweep that ended the search stood in for the last of them) + the Cuthill-McKee sweep
        ref_sweeps_max = candidate + (cm_done ? 1 : 0) + 1;
      }
    }
  }
  SBX_TRY(join_ranks());  // inputs without a large component never ran a Cuthill-McKee sweep
  RcmDev fin;  // (also the synchronisation point of the call)
  if (comp0_searched && n_host == 1 && !mid_batched) fin = hd;  // nothing was launched behind that read-back
  else SBX_TRY(sbx_readback(h, &fin, dv, sizeof(RcmDev)));
  if (fin.unsym)
    SBX_FAIL(h, SBX_ERR_BAD_ARG,
             "sbx_rcm_reorder: the pattern is not structurally symmetric (a BFS cannot reach its whole component); "
             "RCM is defined on symmetric patterns");
  if (stats_host) {
    stats_host->small_components = hd.n_small + (mid_batched ? hd.n_mid : 0u);
    stats_host->large_components = n_host;
    stats_host->bfs_sweeps = sweeps_max;
    stats_host->bfs_levels = levels_max;
    stats_host->edges_scanned = (int64_t)fin.edges;
    stats_host->edges_scanned_bottom_up = (int64_t)fin.edges_bu;
    stats_host->largest_component = largest;
    stats_host->components = (int64_t)hd.n_components;
    stats_host->isolated = (int64_t)hd.n_empty_rows;
    stats_host->reference_sweeps = ref_sweeps_max;
    stats_host->unordered_sweeps = unordered_sweeps;
  }
  return SBX_OK;
}
