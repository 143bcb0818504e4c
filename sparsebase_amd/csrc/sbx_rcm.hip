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
//   small comps     (<= RCM_SMALL vertices) one lane per component runs the serial
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

#include <vector>

namespace {

typedef int32_t I;
constexpr int RCM_SMALL = 64;        // components up to this size: one lane each
constexpr int RCM_LIGHT = 256;       // neighbours expanded by the discovering wave itself
constexpr int RCM_CHUNK = 1024;      // hub neighbours per workgroup chunk
constexpr int RCM_LDS_SORT = 4096;   // levels up to this size are sorted by one workgroup
constexpr unsigned UNSEEN = 0xFFFFFFFFu;

struct RcmDev {                 // device-resident scalars
  unsigned nf;                  // size of the frontier being built
  unsigned n_heavy;             // hub vertices queued for the chunked kernel
  unsigned n_small;             // small components listed
  unsigned n_large;             // large components listed
  unsigned n_cc_big;            // high-degree vertices queued by the CC hook kernel
  unsigned max_deg;
  unsigned root;                // current BFS root of the component being ordered
  unsigned pad;
  unsigned long long best;      // (degree<<32 | position) minimum over the deepest level
  unsigned long long edges;     // adjacency entries scanned (statistics)
  unsigned long long fedges;    // sum of degrees of the level being built (direction heuristic)
  unsigned long long edges_bu;  // adjacency entries scanned by the bottom-up kernel
};

// ------------------------------------------------------------------ degree rank
__global__ __launch_bounds__(256) void k_deg_keys(const I *__restrict__ rp, uint32_t *__restrict__ key,
                                                  uint32_t *__restrict__ id, int64_t n, RcmDev *__restrict__ dv) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned mx = 0;
  for (; v < n; v += stride) {
    const unsigned d = (unsigned)(rp[v + 1] - rp[v]);
    key[v] = d;
    id[v] = (uint32_t)v;
    mx = d > mx ? d : mx;
  }
  mx = sbx_wave_max(mx);
  if (sbx_lane() == 0 && mx) atomicMax(&dv->max_deg, mx);
}

__global__ __launch_bounds__(256) void k_rank_from_order(const uint32_t *__restrict__ dorder,
                                                         uint32_t *__restrict__ drank, int64_t n) {
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; k < n; k += stride) drank[dorder[k]] = (uint32_t)k;
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

__global__ __launch_bounds__(256) void k_cc_init(const I *__restrict__ rp, const I *__restrict__ col,
                                                 I *__restrict__ parent, int64_t n) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; v < n; v += stride) {
    I p = (I)v;
    if (rp[v] < rp[v + 1]) {
      const I first = col[rp[v]];  // smallest neighbour (rows are column-sorted)
      if (first < p) p = first;
    }
    parent[v] = p;
  }
}

// low-degree vertices: one lane per vertex; others queued for the wave kernel
__global__ __launch_bounds__(256) void k_cc_hook_small(const I *__restrict__ rp, const I *__restrict__ col,
                                                       I *parent, int64_t n, I *__restrict__ big_list,
                                                       RcmDev *__restrict__ dv) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; v < n; v += stride) {
    const I s = rp[v], e = rp[v + 1];
    if (e - s > 16) {
      big_list[atomicAdd(&dv->n_cc_big, 1u)] = (I)v;
      continue;
    }
    for (I j = s; j < e; j++) {
      const I u = col[j];
      if (u < (I)v) cc_hook(parent, (I)v, u);
    }
  }
}

__global__ __launch_bounds__(256) void k_cc_hook_big(const I *__restrict__ rp, const I *__restrict__ col, I *parent,
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
__global__ __launch_bounds__(256) void k_cc_finalize(I *parent, I *__restrict__ csize, int64_t n) {
  __shared__ I s_major;
  __shared__ unsigned s_major_cnt;
  int64_t v0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (threadIdx.x == 0) {
    s_major = -1;
    s_major_cnt = 0;
  }
  __syncthreads();
  for (int64_t vb = v0 - sbx_lane(); vb < n; vb += stride) {
    const int64_t v = vb + sbx_lane();
    I root = -1;
    if (v < n) {
      root = parent[v];
      while (root != parent[root]) root = parent[root];
      parent[v] = root;
    }
    uint64_t todo = __ballot(root >= 0);
    while (todo) {
      const int leader = __builtin_ctzll(todo);
      const I lr = __shfl(root, leader, 64);
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
  __syncthreads();
  if (threadIdx.x == 0 && s_major_cnt) atomicAdd(&csize[s_major], (I)s_major_cnt);
}

// classify components: singletons are final here; small / large roots are listed
__global__ __launch_bounds__(256) void k_classify(const I *__restrict__ label, const I *__restrict__ csize,
                                                  const I *__restrict__ cbase, I *__restrict__ inv,
                                                  I *__restrict__ small_list, I *__restrict__ large_list, int64_t n,
                                                  RcmDev *__restrict__ dv) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; v < n; v += stride) {
    if (label[v] != (I)v) continue;  // not a root
    const I sz = csize[v];
    if (sz == 1) inv[v] = cbase[v];
    else if (sz <= RCM_SMALL) small_list[atomicAdd(&dv->n_small, 1u)] = (I)v;
    else large_list[atomicAdd(&dv->n_large, 1u)] = (I)v;
  }
}

// ------------------------------------------------------------------ small components
// One lane runs the reference's serial algorithm for one component.  q = the
// component's own slice of the order array, dist = BFS distances (UNSEEN = unvisited).
__global__ __launch_bounds__(64) void k_rcm_small(const I *__restrict__ rp, const I *__restrict__ col,
                                                  const I *__restrict__ small_list, const I *__restrict__ csize,
                                                  const I *__restrict__ cbase, unsigned *dist, I *order,
                                                  I *__restrict__ inv, RcmDev *__restrict__ dv) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= dv->n_small) return;
  const I start = small_list[k];
  const I base = cbase[start], sz = csize[start];
  I *q = order + base;
  // --- pseudo-peripheral search (rcm_reorder.cc:22-81)
  I root = start;
  int prev_ecc = -1, ecc = 0;
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
    if (head == ecc + 1) break;  // one vertex per level
    if (prev_ecc != ecc) {
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
  for (int i = 0; i < sz; i++) dist[q[i]] = UNSEEN;  // q holds the whole component
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
        // insertion into the (degree,id)-sorted run of u's children
        const I dvv = rp[v + 1] - rp[v];
        int p = tail++;
        while (p > first) {
          const I w = q[p - 1];
          const I dw = rp[w + 1] - rp[w];
          if (dw < dvv || (dw == dvv && w < v)) break;
          q[p] = w;
          p--;
        }
        q[p] = v;
      }
    }
  }
  for (int i = 0; i < sz; i++) inv[q[i]] = base + (sz - 1 - i);  // reverse + invert (:146-160)
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
__global__ void k_bfs_start(const I *__restrict__ rp, unsigned *__restrict__ vbits, unsigned *__restrict__ fbits,
                            unsigned *__restrict__ lpos, I *__restrict__ q, RcmDev *__restrict__ dv, I fixed_root) {
  const I r = fixed_root >= 0 ? fixed_root : (I)dv->root;
  dv->root = (unsigned)r;
  vbits[r >> 5] = 1u << (r & 31);  // both bitmaps were cleared by the host for this sweep
  fbits[r >> 5] = 1u << (r & 31);
  lpos[r] = 0;
  q[0] = r;
  dv->nf = 0;
  dv->n_heavy = 0;
  dv->best = ~0ull;
  dv->fedges = (unsigned long long)(rp[r + 1] - rp[r]);  // degree sum of level 0
}

// Winners are staged per wave in LDS and appended to the frontier in batches: one
// hot counter word sustains only ~88 returning atomics per microsecond, so the
// append must not cost one atomic per ballot.
constexpr int RCM_STAGE = 512;  // staged vertices per wave

struct WaveStage {
  I *buf;                  // this wave's LDS slice
  unsigned cnt;            // wave-uniform fill level
  unsigned long long deg;  // per-lane: degrees of the vertices this lane appended
};

__device__ __forceinline__ void stage_finish(WaveStage &st, RcmDev *__restrict__ dv) {
  const unsigned long long d = sbx_wave_sum(st.deg);
  if (sbx_lane() == 0 && d) atomicAdd(&dv->fedges, d);
}

__device__ __forceinline__ void stage_flush(WaveStage &st, I *__restrict__ nf_list, RcmDev *__restrict__ dv) {
  if (st.cnt == 0) return;
  unsigned base = 0;
  if (sbx_lane() == 0) base = atomicAdd(&dv->nf, st.cnt);
  base = __shfl(base, 0, 64);
  __builtin_amdgcn_wave_barrier();
  for (unsigned i = sbx_lane(); i < st.cnt; i += 64) nf_list[base + i] = st.buf[i];
  __builtin_amdgcn_wave_barrier();
  st.cnt = 0;
}

// appends v (for lanes with `won`) to the staged frontier; all lanes of the wave call it
__device__ __forceinline__ void stage_push(I v, bool won, const I *__restrict__ rp, WaveStage &st,
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

__device__ __forceinline__ void bfs_visit(I v, unsigned p, const I *__restrict__ rp,
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

__global__ __launch_bounds__(256) void k_bfs_expand(const I *__restrict__ rp, const I *__restrict__ col,
                                                    const I *__restrict__ frontier, unsigned fsize,
                                                    unsigned next_level, const unsigned *__restrict__ vbits,
                                                    unsigned *ppos, I *__restrict__ nf_list, uint64_t *__restrict__ heavy,
                                                    RcmDev *__restrict__ dv) {
  __shared__ I s_stage[4][RCM_STAGE];
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int lane = sbx_lane();
  const int grp = lane / RCM_GROUP, gl = lane % RCM_GROUP;
  WaveStage st{s_stage[sbx_wave_in_block()], 0u, 0ull};
  unsigned long long scanned = 0;
  for (int64_t p0 = wave * RCM_VPW; p0 < fsize; p0 += nwaves * RCM_VPW) {
    const int64_t p = p0 + grp;
    I s = 0, e = 0;
    if (p < fsize) {
      const I u = frontier[p];
      s = rp[u];
      e = rp[u + 1];
    }
    if (e - s > RCM_LIGHT) {
      // hub: queue one descriptor (position, chunk) per 1024-neighbour chunk
      const unsigned nchunks = (unsigned)((e - s + RCM_CHUNK - 1) / RCM_CHUNK);
      unsigned slot = 0;
      if (gl == 0) slot = atomicAdd(&dv->n_heavy, nchunks);
      slot = __shfl(slot, grp * RCM_GROUP, 64);
      for (unsigned c = gl; c < nchunks; c += RCM_GROUP) heavy[slot + c] = ((uint64_t)p << 32) | c;
      e = s;  // nothing to do inline
    }
    if (gl == 0) scanned += (unsigned)(e - s);
    I j = s + gl;
    while (__any(j < e)) {
      const bool act = j < e;
      const I v = act ? col[j] : 0;
      bfs_visit(v, (unsigned)p, rp, vbits, ppos, st, nf_list, dv, act);
      j += RCM_GROUP;
    }
  }
  stage_flush(st, nf_list, dv);
  stage_finish(st, dv);
  scanned = sbx_wave_sum(scanned);
  if (lane == 0 && scanned) atomicAdd(&dv->edges, scanned);
}

__global__ __launch_bounds__(256) void k_bfs_expand_heavy(const I *__restrict__ rp, const I *__restrict__ col,
                                                          const I *__restrict__ frontier, unsigned next_level,
                                                          const unsigned *__restrict__ vbits, unsigned *ppos,
                                                          I *__restrict__ nf_list,
                                                          const uint64_t *__restrict__ heavy,
                                                          RcmDev *__restrict__ dv) {
  __shared__ I s_stage[4][RCM_STAGE];
  const unsigned nd = dv->n_heavy;  // chunk descriptors queued by k_bfs_expand
  WaveStage st{s_stage[sbx_wave_in_block()], 0u, 0ull};
  unsigned long long scanned = 0;
  for (unsigned d = blockIdx.x; d < nd; d += gridDim.x) {
    const uint64_t desc = heavy[d];
    const unsigned p = (unsigned)(desc >> 32), c = (unsigned)desc;
    const I u = frontier[p];
    const I s = rp[u], e = rp[u + 1];
    const I cs = s + (I)c * RCM_CHUNK;
    const I ce = (cs + RCM_CHUNK < e) ? cs + RCM_CHUNK : e;
    for (I j0 = cs; j0 < ce; j0 += 256) {
      const I j = j0 + (I)threadIdx.x;
      const bool act = j < ce;
      const I v = act ? col[j] : 0;
      bfs_visit(v, p, rp, vbits, ppos, st, nf_list, dv, act);
    }
    scanned += (unsigned long long)(ce - cs);
  }
  stage_flush(st, nf_list, dv);
  stage_finish(st, dv);
  if (threadIdx.x == 0 && scanned) atomicAdd(&dv->edges, scanned);
}

// Bottom-up expansion (direction-optimising BFS): instead of the frontier pushing along
// its edges with atomics, every still-unvisited vertex of the component pulls — it scans
// its own adjacency, keeps the smallest level position among neighbours that are in the
// current frontier (fbits / lpos) and, if it found one, joins the next level.  No atomics
// on vertex state, no hot words; chosen by the host when the frontier owns more edges
// than the unvisited remainder.
__global__ __launch_bounds__(256) void k_bfs_bottom_up(const I *__restrict__ rp, const I *__restrict__ col,
                                                       const I *__restrict__ label, I comp_label,
                                                       const unsigned *__restrict__ vbits,
                                                       const unsigned *__restrict__ fbits,
                                                       const unsigned *__restrict__ lpos, unsigned *__restrict__ ppos,
                                                       I *__restrict__ nf_list, int64_t n, RcmDev *__restrict__ dv) {
  __shared__ I s_stage[4][RCM_STAGE];
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int lane = sbx_lane();
  const int grp = lane / RCM_GROUP, gl = lane % RCM_GROUP;
  WaveStage st{s_stage[sbx_wave_in_block()], 0u, 0ull};
  unsigned long long scanned = 0;
  for (int64_t base = wave * 64; base < n; base += nwaves * 64) {
    const int64_t v = base + lane;
    bool cand = false;
    I s = 0, e = 0;
    if (v < n && !((vbits[v >> 5] >> (v & 31)) & 1u)) {
      s = rp[v];
      e = rp[v + 1];
      cand = (e > s) && label[v] == comp_label;
    }
    uint64_t todo = __ballot(cand);
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
        if (j < jend) {
          const I u = col[j];
          if ((fbits[u >> 5] >> (u & 31)) & 1u) {
            const unsigned lp = lpos[u];
            best = lp < best ? lp : best;
          }
        }
        j += RCM_GROUP;
      }
      if (gl == 0 && pick >= 0) scanned += (unsigned)(ce - cs);
#pragma unroll
      for (int d = RCM_GROUP / 2; d >= 1; d >>= 1) {
        const unsigned o = __shfl_xor(best, d, 64);
        best = o < best ? o : best;
      }
      const bool found = (gl == 0) && pick >= 0 && best != UNSEEN;
      const I nv = (I)(base + (pick < 0 ? 0 : pick));
      if (found) ppos[nv] = best;
      stage_push(nv, found, rp, st, nf_list, dv);
    }
  }
  stage_flush(st, nf_list, dv);
  stage_finish(st, dv);
  scanned = sbx_wave_sum(scanned);
  if (lane == 0 && scanned) {
    atomicAdd(&dv->edges, scanned);
    atomicAdd(&dv->edges_bu, scanned);
  }
}

template <bool CM>
__global__ __launch_bounds__(256) void k_level_keys(const I *__restrict__ nf_list, unsigned nf,
                                                    const unsigned *__restrict__ ppos,
                                                    const uint32_t *__restrict__ drank, uint64_t *__restrict__ key) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; j < nf; j += stride) {
    const I v = nf_list[j];
    key[j] = ((uint64_t)ppos[v] << 32) | (uint64_t)(CM ? drank[v] : (uint32_t)v);
  }
}

template <bool CM>
__global__ __launch_bounds__(256) void k_level_emit(const uint64_t *__restrict__ key, unsigned nf,
                                                    const uint32_t *__restrict__ dorder, I *__restrict__ q_level,
                                                    unsigned *__restrict__ vbits, unsigned *__restrict__ fbits,
                                                    unsigned *__restrict__ lpos, RcmDev *__restrict__ dv) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; j < nf; j += stride) {
    const uint32_t lo = (uint32_t)key[j];
    const I v = (I)(CM ? dorder[lo] : lo);
    q_level[j] = v;
    atomicOr(&vbits[v >> 5], 1u << (v & 31));  // this level is now ordered
    atomicOr(&fbits[v >> 5], 1u << (v & 31));  // ... and is the next frontier
    lpos[v] = (unsigned)j;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    dv->nf = 0;
    dv->n_heavy = 0;
    dv->fedges = 0;
  }
}

// small level: keys, bitonic sort in LDS and emit in one workgroup
template <bool CM>
__global__ __launch_bounds__(1024) void k_level_sort_small(const I *__restrict__ nf_list, unsigned nf,
                                                           const unsigned *__restrict__ ppos,
                                                           const uint32_t *__restrict__ drank,
                                                           const uint32_t *__restrict__ dorder,
                                                           I *__restrict__ q_level, unsigned *__restrict__ vbits,
                                                           unsigned *__restrict__ fbits, unsigned *__restrict__ lpos,
                                                           RcmDev *__restrict__ dv) {
  __shared__ uint64_t s_key[RCM_LDS_SORT];
  unsigned p2 = 1;
  while (p2 < nf) p2 <<= 1;
  for (unsigned j = threadIdx.x; j < p2; j += blockDim.x) {
    uint64_t k = ~0ull;
    if (j < nf) {
      const I v = nf_list[j];
      k = ((uint64_t)ppos[v] << 32) | (uint64_t)(CM ? drank[v] : (uint32_t)v);
    }
    s_key[j] = k;
  }
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
  for (unsigned j = threadIdx.x; j < nf; j += blockDim.x) {
    const uint32_t lo = (uint32_t)s_key[j];
    const I v = (I)(CM ? dorder[lo] : lo);
    q_level[j] = v;
    atomicOr(&vbits[v >> 5], 1u << (v & 31));
    atomicOr(&fbits[v >> 5], 1u << (v & 31));
    lpos[v] = j;
  }
  if (threadIdx.x == 0) {
    dv->nf = 0;
    dv->n_heavy = 0;
    dv->fedges = 0;
  }
}

// deepest level: vertex of strictly smallest degree, first in queue order (:64-75)
__global__ __launch_bounds__(256) void k_pick_root(const I *__restrict__ rp, const I *__restrict__ level,
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

__global__ __launch_bounds__(256) void k_write_component(const I *__restrict__ q, unsigned cnt, I base,
                                                         I *__restrict__ inv) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; j < cnt; j += stride) inv[q[j]] = base + (I)(cnt - 1 - j);
}

struct BfsBuffers {
  const I *rp, *col;
  unsigned *vbits, *fbits, *lpos, *ppos;
  const I *label;
  int64_t nnz;
  I *q;         // visiting order of the current BFS (levels concatenated)
  I *nf_list;   // unordered next frontier
  uint64_t *heavy;
  uint64_t *ka, *kb;
  const uint32_t *drank, *dorder;
  RcmDev *dv;
  int64_t n;
};

struct BfsResult {
  unsigned count;        // vertices reached
  unsigned levels;       // number of levels (eccentricity + 1)
  unsigned last_offset;  // offset of the deepest level in q
  unsigned last_size;
};

// One ordered BFS over the component containing the root (fixed_root >= 0, or the
// device-resident dv->root).  CM selects Cuthill-McKee child order.  Direction per
// level: top-down (frontier pushes) unless the frontier is large and owns more than
// half as many edges as the still-unvisited remainder — then bottom-up (pull).
template <bool CM>
int run_bfs(sbx_handle_t h, const BfsBuffers &b, I fixed_root, I comp_label, BfsResult *out) {
  const size_t bm_bytes = (size_t)((b.n + 31) / 32) * sizeof(unsigned);
  SBX_HIP(h, hipMemsetAsync(b.vbits, 0, bm_bytes, h->stream));
  SBX_HIP(h, hipMemsetAsync(b.fbits, 0, bm_bytes, h->stream));
  SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_bfs_start, dim3(1), dim3(1), b.rp, b.vbits, b.fbits, b.lpos, b.q, b.dv, fixed_root);
  unsigned off = 0, fsize = 1, level = 0, total = 1;
  const unsigned max_grid = (unsigned)h->num_cus * 8;
  int64_t remaining = b.nnz;      // adjacency entries owned by vertices not yet in any level
  int64_t frontier_edges = -1;    // degree sum of the current frontier (-1: level 0, read lazily)
  while (true) {
    bool bottom_up = false;
    if (frontier_edges >= 0 && fsize >= 8192) bottom_up = 2 * frontier_edges > remaining;
    if (bottom_up) {
      SBX_KLAUNCH(h, SBX_K_BFS_BOTTOMUP, k_bfs_bottom_up, dim3(max_grid), dim3(256), b.rp, b.col, b.label, comp_label,
                  (const unsigned *)b.vbits, (const unsigned *)b.fbits, (const unsigned *)b.lpos, b.ppos, b.nf_list,
                  b.n, b.dv);
    } else {
      const unsigned waves_needed = (fsize + RCM_VPW - 1) / RCM_VPW;
      unsigned grid = (waves_needed + 3) / 4;
      if (grid > max_grid) grid = max_grid;
      if (grid < 1) grid = 1;
      SBX_KLAUNCH(h, SBX_K_BFS_EXPAND, k_bfs_expand, dim3(grid), dim3(256), b.rp, b.col, (const I *)(b.q + off), fsize,
                  level + 1, (const unsigned *)b.vbits, b.ppos, b.nf_list, b.heavy, b.dv);
      SBX_KLAUNCH(h, SBX_K_BFS_HEAVY, k_bfs_expand_heavy, dim3(max_grid), dim3(256), b.rp, b.col,
                  (const I *)(b.q + off), level + 1, (const unsigned *)b.vbits, b.ppos, b.nf_list,
                  (const uint64_t *)b.heavy, b.dv);
    }
    SBX_LAUNCH_CHECK(h);
    RcmDev hd;
    SBX_TRY(sbx_readback(h, &hd, b.dv, sizeof(RcmDev)));
    const unsigned nf = hd.nf;
    if (nf == 0) break;
    if (frontier_edges < 0) remaining -= (int64_t)0;  // level 0's degree is part of hd.fedges history below
    frontier_edges = (int64_t)hd.fedges;              // degree sum of the level just discovered
    remaining -= frontier_edges;
    if (remaining < 0) remaining = 0;
    I *q_next = b.q + off + fsize;
    SBX_HIP(h, hipMemsetAsync(b.fbits, 0, bm_bytes, h->stream));
    if (nf <= RCM_LDS_SORT) {
      SBX_KLAUNCH(h, SBX_K_LEVEL_ORDER, (k_level_sort_small<CM>), dim3(1), dim3(1024), (const I *)b.nf_list, nf,
                  (const unsigned *)b.ppos, b.drank, b.dorder, q_next, b.vbits, b.fbits, b.lpos, b.dv);
    } else {
      const unsigned g = sbx_grid_for(nf, 256, 4096);
      SBX_KLAUNCH(h, SBX_K_LEVEL_ORDER, (k_level_keys<CM>), dim3(g), dim3(256), (const I *)b.nf_list, nf,
                  (const unsigned *)b.ppos, b.drank, b.ka);
      sbx_radix_pass passes[16];
      const int np = sbx_radix_plan(0, sbx_bits_for((uint64_t)(b.n - 1)), 32,
                                    32 + sbx_bits_for((uint64_t)(fsize - 1)), passes);
      int in_b = 0;
      SBX_TRY(sbx_radix_sort(h, 8, 0, b.ka, b.kb, nullptr, nullptr, nf, passes, np, &in_b));
      SBX_KLAUNCH(h, SBX_K_LEVEL_ORDER, (k_level_emit<CM>), dim3(g), dim3(256),
                  (const uint64_t *)(in_b ? b.kb : b.ka), nf, b.dorder, q_next, b.vbits, b.fbits, b.lpos, b.dv);
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

}  // namespace

extern "C" int sbx_rcm_reorder(sbx_handle_t h, sbx_index_type it, int64_t n, int64_t nnz, const void *row_ptr,
                               const void *col_v, void *inv_perm_out, sbx_rcm_stats *stats_host) {
  if (!h) return SBX_ERR_BAD_ARG;
  if (n < 0 || nnz < 0 || !row_ptr || (n > 0 && !inv_perm_out) || (nnz > 0 && !col_v))
    SBX_FAIL(h, SBX_ERR_BAD_ARG, "sbx_rcm_reorder: bad argument");
  if (it != SBX_I32) SBX_FAIL(h, SBX_ERR_UNSUPPORTED, "sbx_rcm_reorder: 64-bit indices not built yet");
  if (n >= ((int64_t)1 << 31) - 1 || nnz >= ((int64_t)1 << 31))
    SBX_FAIL(h, SBX_ERR_BAD_ARG, "sbx_rcm_reorder: dimension exceeds int32");
  if (stats_host) memset(stats_host, 0, sizeof(*stats_host));
  SBX_TRY(sbx_arena_begin(h));
  if (n == 0) return SBX_OK;
  const I *rp = (const I *)row_ptr, *col = (const I *)col_v;
  I *inv = (I *)inv_perm_out;

  RcmDev *dv;
  uint32_t *dkey_a, *dkey_b, *did_a, *did_b, *drank;
  I *label, *csize, *cbase, *small_list, *large_list, *big_list, *q, *nf_list;
  unsigned *dist, *ppos, *vbits, *fbits, *lpos;
  uint64_t *ka, *kb, *heavy;
  SBX_TRY(sbx_salloc(h, 1, &dv));
  SBX_TRY(sbx_salloc(h, (size_t)n, &dkey_a));
  SBX_TRY(sbx_salloc(h, (size_t)n, &dkey_b));
  SBX_TRY(sbx_salloc(h, (size_t)n, &did_a));
  SBX_TRY(sbx_salloc(h, (size_t)n, &did_b));
  SBX_TRY(sbx_salloc(h, (size_t)n, &drank));
  SBX_TRY(sbx_salloc(h, (size_t)n, &label));
  SBX_TRY(sbx_salloc(h, (size_t)n + 1, &csize));
  SBX_TRY(sbx_salloc(h, (size_t)n + 1, &cbase));
  SBX_TRY(sbx_salloc(h, (size_t)n, &small_list));
  SBX_TRY(sbx_salloc(h, (size_t)n, &large_list));
  SBX_TRY(sbx_salloc(h, (size_t)n, &big_list));
  SBX_TRY(sbx_salloc(h, (size_t)n, &q));
  SBX_TRY(sbx_salloc(h, (size_t)n, &nf_list));
  SBX_TRY(sbx_salloc(h, (size_t)n, &dist));
  SBX_TRY(sbx_salloc(h, (size_t)n, &ppos));
  SBX_TRY(sbx_salloc(h, (size_t)(n + 31) / 32 + 1, &vbits));
  SBX_TRY(sbx_salloc(h, (size_t)(n + 31) / 32 + 1, &fbits));
  SBX_TRY(sbx_salloc(h, (size_t)n, &lpos));
  SBX_TRY(sbx_salloc(h, (size_t)(nnz / RCM_LIGHT + nnz / RCM_CHUNK + 1024), &heavy));
  SBX_TRY(sbx_salloc(h, (size_t)n, &ka));
  SBX_TRY(sbx_salloc(h, (size_t)n, &kb));
  SBX_HIP(h, hipMemsetAsync(dv, 0, sizeof(RcmDev), h->stream));
  SBX_HIP(h, hipMemsetAsync(csize, 0, (size_t)(n + 1) * sizeof(I), h->stream));
  SBX_HIP(h, hipMemsetAsync(dist, 0xFF, (size_t)n * sizeof(unsigned), h->stream));
  SBX_HIP(h, hipMemsetAsync(ppos, 0xFF, (size_t)n * sizeof(unsigned), h->stream));

  const unsigned gn = sbx_grid_for(n, 256, 8192);
  // (1) connected components; the root of each tree is the component's smallest id
  SBX_KLAUNCH(h, SBX_K_CC, k_cc_init, dim3(gn), dim3(256), rp, col, label, n);
  SBX_KLAUNCH(h, SBX_K_CC, k_cc_hook_small, dim3(gn), dim3(256), rp, col, label, n, big_list, dv);
  SBX_KLAUNCH(h, SBX_K_CC, k_cc_hook_big, dim3((unsigned)h->num_cus * 8), dim3(256), rp, col, label,
                     (const I *)big_list, (const RcmDev *)dv);
  SBX_KLAUNCH(h, SBX_K_CC, k_cc_finalize, dim3(gn), dim3(256), label, csize, n);
  SBX_LAUNCH_CHECK(h);
  SBX_TRY(sbx_exclusive_scan_i32(h, csize, cbase, n + 1, nullptr));
  SBX_KLAUNCH(h, SBX_K_CC, k_classify, dim3(gn), dim3(256), (const I *)label, (const I *)csize,
                     (const I *)cbase, inv, small_list, large_list, n, dv);
  // (2) global (degree,id) rank used by the Cuthill-McKee keys
  SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_deg_keys, dim3(gn), dim3(256), rp, dkey_a, did_a, n, dv);
  SBX_LAUNCH_CHECK(h);
  RcmDev hd;
  SBX_TRY(sbx_readback(h, &hd, dv, sizeof(RcmDev)));
  const uint32_t *dorder;
  {
    sbx_radix_pass passes[16];
    const int np = sbx_radix_plan(0, sbx_bits_for(hd.max_deg), 0, 0, passes);
    int in_b = 0;
    SBX_TRY(sbx_radix_sort(h, 4, 4, dkey_a, dkey_b, did_a, did_b, n, passes, np, &in_b));
    dorder = in_b ? did_b : did_a;
    SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_rank_from_order, dim3(gn), dim3(256), dorder, drank, n);
    SBX_LAUNCH_CHECK(h);
  }
  // (3) small components: one lane each
  if (hd.n_small) {
    SBX_KLAUNCH(h, SBX_K_RCM_SMALL, k_rcm_small, dim3((hd.n_small + 63) / 64), dim3(64), rp, col,
                       (const I *)small_list, (const I *)csize, (const I *)cbase, dist, q, inv, dv);
    SBX_LAUNCH_CHECK(h);
  }
  // (4) large components: host-driven level-synchronous BFS
  int64_t sweeps_max = 0, levels_max = 0, largest = 0;
  if (hd.n_large) {
    std::vector<I> roots(hd.n_large), sizes(hd.n_large), bases(hd.n_large);
    SBX_HIP(h, hipMemcpyAsync(roots.data(), large_list, hd.n_large * sizeof(I), hipMemcpyDeviceToHost, h->stream));
    SBX_HIP(h, hipStreamSynchronize(h->stream));
    for (unsigned c = 0; c < hd.n_large; c++) {
      SBX_HIP(h, hipMemcpyAsync(&sizes[c], csize + roots[c], sizeof(I), hipMemcpyDeviceToHost, h->stream));
      SBX_HIP(h, hipMemcpyAsync(&bases[c], cbase + roots[c], sizeof(I), hipMemcpyDeviceToHost, h->stream));
    }
    SBX_HIP(h, hipStreamSynchronize(h->stream));
    BfsBuffers b;
    b.rp = rp; b.col = col; b.vbits = vbits; b.fbits = fbits; b.lpos = lpos; b.ppos = ppos; b.label = label;
    b.nnz = nnz; b.q = q; b.nf_list = nf_list; b.heavy = heavy;
    b.ka = ka; b.kb = kb; b.drank = drank; b.dorder = dorder; b.dv = dv; b.n = n;
    for (unsigned c = 0; c < hd.n_large; c++) {
      // pseudo-peripheral search from the component's smallest vertex (:22-81)
      BfsResult r;
      int64_t prev_ecc = -1, ecc = 0, sweeps = 0, levels = 0;
      I fixed = roots[c];
      while (prev_ecc != ecc) {
        prev_ecc = ecc;
        SBX_TRY(run_bfs<false>(h, b, fixed, roots[c], &r));
        fixed = -1;  // later sweeps start from the device-resident root
        sweeps++;
        levels += r.levels;
        const int64_t e = (int64_t)r.levels - 1;
        if (e > ecc) ecc = e;
        const bool path = (int64_t)r.count == ecc + 1;
        if (!path && prev_ecc != ecc) {
          SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_pick_root, dim3(sbx_grid_for(r.last_size, 256, 1024)), dim3(256), rp,
                             (const I *)(q + r.last_offset), r.last_size, dv);
          SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_set_root_from_best, dim3(1), dim3(1), (const I *)(q + r.last_offset), dv);
        }
        SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_reset_visited, dim3(sbx_grid_for(r.count, 256, 4096)), dim3(256),
                           (const I *)q, r.count, ppos);
        SBX_LAUNCH_CHECK(h);
        if (path) break;
      }
      // Cuthill-McKee BFS from the pseudo-peripheral vertex (:118-144)
      SBX_TRY(run_bfs<true>(h, b, -1, roots[c], &r));
      sweeps++;
      levels += r.levels;
      if ((int64_t)r.count != (int64_t)sizes[c])
        SBX_FAIL(h, SBX_ERR_INTERNAL,
                 "sbx_rcm_reorder: BFS reached %u of %d vertices of a component (pattern not symmetric?)", r.count,
                 sizes[c]);
      SBX_KLAUNCH(h, SBX_K_RCM_MISC, k_write_component, dim3(sbx_grid_for(r.count, 256, 4096)), dim3(256),
                         (const I *)q, r.count, bases[c], inv);
      SBX_LAUNCH_CHECK(h);
      if (sizes[c] > largest) {
        largest = sizes[c];
        sweeps_max = sweeps;
        levels_max = levels;
      }
    }
  }
  if (stats_host) {
    RcmDev fin;
    SBX_TRY(sbx_readback(h, &fin, dv, sizeof(RcmDev)));
    stats_host->small_components = hd.n_small;
    stats_host->large_components = hd.n_large;
    stats_host->bfs_sweeps = sweeps_max;
    stats_host->bfs_levels = levels_max;
    stats_host->edges_scanned = (int64_t)fin.edges;
    stats_host->edges_scanned_bottom_up = (int64_t)fin.edges_bu;
    stats_host->largest_component = largest;
    stats_host->components = -1;  // filled by callers that need it (count of roots); not tracked on device
    stats_host->isolated = -1;
  } else {
    SBX_HIP(h, hipStreamSynchronize(h->stream));
  }
  return SBX_OK;
}
