// sbx_sharded.hip — the steps of the path that shard across the GPUs of a node, one process per GPU.
//
//   sbx_comm_*                 communicator: RCCL (ncclAllGather over xGMI; librccl is loaded on first use, the
//                              library does not link against it) or a caller-supplied all-gather hook
//   sbx_permute_csr_sharded    A5 by new-row range: this rank's slab with sbx_permute_csr_rows, then the two
//                              all-gathers that give every rank the whole permuted row_ptr
//   sbx_coo_to_csr_sharded     A2 by row range of a row-sorted COO: the rank's nonzero slice by binary search, local
//                              conversion, same stitch
//   sbx_permute_csr_rows_nnz   entries of a new-row range (sizes the rank's slab before the first call)
//
// Reference counterpart of the device-to-device edge: converter/converter_order_two_cuda.cu:41-76 (peer copies between
// CUDA contexts, predicate converter/converter_cuda.cu:12-21); the reference has no multi-GPU operator — north_star
// asks for the row-range split with an all-gatherv of row_ptr.  Exchange per call: 8 bytes per rank (nnz totals) and
// Z * n bytes in total (padded equal row_ptr chunks); col/val stay row-sharded.  No collective touches the nonzeros.
#include <dlfcn.h>

#include <vector>

#include "sbx_device.h"
#include "sbx_internal.h"

struct sbx_comm_s {
  int rank, world;
  sbx_allgather_fn allgather;
  void *user;
  void *nccl_comm;  // RCCL flavour only
};

namespace {

// ---- RCCL, loaded lazily ----------------------------------------------------------------------------------
struct Id128 {  // ncclUniqueId
  char bytes[SBX_COMM_ID_BYTES];
};
struct RcclApi {
  void *lib = nullptr;
  int (*GetUniqueId)(void *) = nullptr;
  int (*CommInitRank)(void **, int, Id128 /* by value, as ncclCommInitRank takes it */, int) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;

bool rccl_load() {
  if (g_rccl.lib) return true;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void *lib = nullptr;
  for (const char *nm : names) {
    lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
    if (lib) break;
  }
  if (!lib) return false;
  RcclApi a;
  a.lib = lib;
  a.GetUniqueId = (int (*)(void *))dlsym(lib, "ncclGetUniqueId");
  a.CommInitRank = (int (*)(void **, int, Id128, int))dlsym(lib, "ncclCommInitRank");
  a.AllGather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))dlsym(lib, "ncclAllGather");
  a.CommDestroy = (int (*)(void *))dlsym(lib, "ncclCommDestroy");
  a.GetErrorString = (const char *(*)(int))dlsym(lib, "ncclGetErrorString");
  if (!a.GetUniqueId || !a.CommInitRank || !a.AllGather || !a.CommDestroy) {
    dlclose(lib);
    return false;
  }
  g_rccl = a;
  return true;
}

int rccl_allgather(void *user, const void *send, void *recv, size_t bytes, void *stream) {
  sbx_comm_t c = (sbx_comm_t)user;
  const int rc = g_rccl.AllGather(send, recv, bytes, /* ncclChar */ 0, c->nccl_comm, (hipStream_t)stream);
  return rc == 0 ? SBX_OK : SBX_ERR_HIP;
}

// ---- stitch kernels -----------------------------------------------------------------------------------------
constexpr int MAX_WORLD = 64;
struct Splits {
  int64_t lo[MAX_WORLD + 1];
};

// what a rank contributes to the first all-gather: its shard's entries and its status (0, or the status code its own
// part of the call ended with: every rank then fails the call together instead of leaving the others in a collective)
struct ShardRec {
  int64_t total, status;
};
__global__ void k_set_rec(ShardRec *dst, int64_t total, int64_t status) {
  dst->total = total;
  dst->status = status;
}
// offs[0 .. world] = positions of the shards in the global entry space; offs[world + 1] = first rank (+ 1) that
// reported a failure, offs[world + 2] = 1 if the entries do not fit the index type
__global__ void k_shard_offsets(const ShardRec *__restrict__ recs, int world, int index_bits, int64_t *__restrict__ offs) {
  int64_t run = 0, bad = 0;
  for (int r = 0; r < world; r++) {
    offs[r] = run;
    run += recs[r].total;
    if (recs[r].status != 0 && bad == 0) bad = r + 1;
  }
  offs[world] = run;
  offs[world + 1] = bad;
  offs[world + 2] = (index_bits == 32 && run >= ((int64_t)1 << 31)) ? 1 : 0;
}

// send[i] = local[i] + (the shard's offset) for the shard's rows, zero padding up to `chunk`
template <typename I>
__global__ __launch_bounds__(256) void k_seg_prepare(const I *__restrict__ local, int64_t rows,
                                                     const int64_t *__restrict__ offset, I *__restrict__ send,
                                                     int64_t chunk) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t off = *offset;
  for (; i < chunk; i += stride) send[i] = i < rows ? (I)((int64_t)local[i] + off) : (I)0;
}

// out[row] = gathered[rank(row) * chunk + row - lo(rank)], out[n] = total
template <typename I>
__global__ __launch_bounds__(256) void k_stitch(const I *__restrict__ gathered, int64_t chunk, Splits sp, int world,
                                                int64_t n, const int64_t *__restrict__ total, I *__restrict__ out) {
  int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; row <= n; row += stride) {
    if (row == n) {
      out[row] = (I)*total;
      continue;
    }
    int r = 0;  // the rank whose range holds the row (world <= 64: a short scan, the same for a whole wave mostly)
    while (r + 1 < world && sp.lo[r + 1] <= row) r++;
    out[row] = gathered[(int64_t)r * chunk + (row - sp.lo[r])];
  }
}


// first positions of rows >= lo and >= hi in a non-decreasing row array
template <typename I>
__global__ void k_row_bounds(const I *__restrict__ row, int64_t nnz, int64_t lo, int64_t hi, int64_t *__restrict__ out2) {
  const int64_t key = threadIdx.x == 0 ? lo : hi;
  int64_t a = 0, b = nnz;
  while (a < b) {
    const int64_t mid = (a + b) >> 1;
    if ((int64_t)row[mid] >= key) b = mid;
    else a = mid + 1;
  }
  out2[threadIdx.x] = a;
}

template <typename I>
__global__ __launch_bounds__(256) void k_rebase(const I *__restrict__ row, int64_t count, int64_t lo, I *__restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < count; i += stride) out[i] = (I)((int64_t)row[i] - lo);
}

// entries of the old rows that the order maps into [rb0, rb1)
template <typename I>
__global__ __launch_bounds__(256) void k_range_nnz(const I *__restrict__ rp, const I *__restrict__ row_order, int64_t n,
                                                   int64_t rb0, int64_t rb1, unsigned long long *__restrict__ out) {
  __shared__ unsigned long long s_part[4];
  unsigned long long acc = 0;
  int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; u < n; u += stride) {
    const int64_t r = row_order ? (int64_t)row_order[u] : u;
    if (r >= rb0 && r < rb1) acc += (unsigned long long)(rp[u + 1] - rp[u]);
  }
  acc = sbx_block_sum<unsigned long long, 256>(acc, s_part);
  if (threadIdx.x == 0 && acc) atomicAdd(out, acc);
}

struct NestGuard {  // the nested entry points must not rewind the caller's scratch
  sbx_handle_t h;
  explicit NestGuard(sbx_handle_t h_) : h(h_) { h->nest++; }
  ~NestGuard() { if (h->nest > 0) h->nest--; }
};

int resolve_splits(sbx_handle_t h, sbx_comm_t comm, int64_t n, const int64_t *row_splits, Splits *sp, int64_t *chunk) {
  const int world = comm->world;
  if (world < 1 || world > MAX_WORLD) SBX_FAIL(h, SBX_ERR_UNSUPPORTED, "world size %d (supported: 1..%d)", world, MAX_WORLD);
  if (row_splits) {
    for (int r = 0; r <= world; r++) sp->lo[r] = row_splits[r];
  } else {  // contiguous ranges whose sizes differ by at most one row
    const int64_t base = n / world, extra = n % world;
    int64_t lo = 0;
    for (int r = 0; r < world; r++) {
      sp->lo[r] = lo;
      lo += base + (r < extra ? 1 : 0);
    }
    sp->lo[world] = n;
  }
  if (sp->lo[0] != 0 || sp->lo[world] != n) SBX_FAIL(h, SBX_ERR_BAD_ARG, "row_splits must run from 0 to n");
  *chunk = 1;
  for (int r = 0; r < world; r++) {
    if (sp->lo[r + 1] < sp->lo[r]) SBX_FAIL(h, SBX_ERR_BAD_ARG, "row_splits must be non-decreasing");
    if (sp->lo[r + 1] - sp->lo[r] > *chunk) *chunk = sp->lo[r + 1] - sp->lo[r];
  }
  return SBX_OK;
}

// the two all-gathers + the stitch: local (rows + 1 entries, rebased to 0) -> row_ptr_out (n + 1 entries).  Everything
// between the collectives stays on the device (the shards' offsets are computed from the gathered totals by a kernel):
// the host waits ONCE, at the end, for the offsets and the ranks' status words — a rank whose own part failed
// (`status` != 0, its `local` then holds zeros) still takes part in both collectives, and every rank returns an error.
template <typename I>
int stitch(sbx_handle_t h, sbx_comm_t comm, const Splits &sp, int64_t chunk, int64_t n, const I *local, int64_t local_nnz,
           int status, I *row_ptr_out, int64_t *shard_offsets_host) {
  const int world = comm->world, rank = comm->rank;
  const int64_t rows = sp.lo[rank + 1] - sp.lo[rank];
  ShardRec *mine = nullptr, *recs = nullptr;
  int64_t *offs = nullptr;
  I *send = nullptr, *gathered = nullptr;
  SBX_TRY(sbx_salloc(h, 1, &mine));
  SBX_TRY(sbx_salloc(h, (size_t)world, &recs));
  SBX_TRY(sbx_salloc(h, (size_t)world + 3, &offs));
  SBX_TRY(sbx_salloc(h, (size_t)chunk, &send));
  SBX_TRY(sbx_salloc(h, (size_t)chunk * world, &gathered));
  // From here on a local failure (a launch that did not go out, a collective that reported an error) does not return
  // before BOTH collectives have been entered: the peers are inside them and would wait for ever.  The first failure is
  // kept and returned at the end.  (Not covered: a failure of the five small allocations above, and a device that has
  // stopped executing altogether — include/sbx.h, "Sharded calls".)
  int first_rc = SBX_OK;
  char first_err[sizeof(h->err)];
  first_err[0] = 0;
  auto note = [&](int rc, const char *what) {
    if (rc != SBX_OK && first_rc == SBX_OK) {
      first_rc = rc;
      snprintf(first_err, sizeof(first_err), "sharded call: %s", what);
    }
  };
  // (1) nnz totals + status words -> offsets of the shards in the global entry space (on the device)
  SBX_KLAUNCH(h, SBX_K_MISC, k_set_rec, dim3(1), dim3(1), mine, local_nnz, (int64_t)status);
  if (hipGetLastError() != hipSuccess) note(SBX_ERR_HIP, "launch of the shard record kernel failed");
  if (comm->allgather(comm->user, mine, recs, sizeof(ShardRec), (void *)h->stream) != SBX_OK)
    note(SBX_ERR_HIP, "all-gather of the shard totals failed");
  SBX_KLAUNCH(h, SBX_K_MISC, k_shard_offsets, dim3(1), dim3(1), (const ShardRec *)recs, world, (int)(8 * sizeof(I)), offs);
  // (2) row_ptr segments in padded equal chunks (the all-gatherv of SURVEY §8e), then every rank assembles the whole
  SBX_KLAUNCH(h, SBX_K_MISC, k_seg_prepare<I>, dim3(sbx_grid_for(chunk, 256, 4096)), dim3(256), local, rows,
              (const int64_t *)(offs + rank), send, chunk);
  if (hipGetLastError() != hipSuccess) note(SBX_ERR_HIP, "launch of the segment kernels failed");
  if (comm->allgather(comm->user, send, gathered, sizeof(I) * (size_t)chunk, (void *)h->stream) != SBX_OK)
    note(SBX_ERR_HIP, "all-gather of the row_ptr segments failed");
  if (first_rc != SBX_OK) SBX_FAIL(h, first_rc, "%s", first_err);
  SBX_KLAUNCH(h, SBX_K_MISC, k_stitch<I>, dim3(sbx_grid_for(n + 1, 256, 8192)), dim3(256), (const I *)gathered, chunk, sp,
              world, n, (const int64_t *)(offs + world), row_ptr_out);
  SBX_LAUNCH_CHECK(h);
  // (3) the one wait of the call: offsets, status, overflow
  std::vector<int64_t> off(world + 3, 0);
  SBX_TRY(sbx_readback(h, off.data(), offs, sizeof(int64_t) * (size_t)(world + 3)));
  if (shard_offsets_host)
    for (int r = 0; r <= world; r++) shard_offsets_host[r] = off[r];
  if (off[world + 1] != 0 && status == 0)
    SBX_FAIL(h, SBX_ERR_INTERNAL, "sharded call: rank %lld reported a failure (its own call returns the reason)",
             (long long)(off[world + 1] - 1));
  if (off[world + 2] != 0) SBX_FAIL(h, SBX_ERR_UNSUPPORTED, "row_ptr does not fit 32 bits");
  return SBX_OK;
}

// a rank's own part of a sharded call has failed with `rc` (message in h->err): take part in the collectives with a
// zeroed segment so that the other ranks return too, then hand the original error back
template <typename I>
int stitch_failed(sbx_handle_t h, sbx_comm_t comm, const Splits &sp, int64_t chunk, int64_t n, void *local, int64_t rows,
                  int rc, void *row_ptr_out) {
  char saved[sizeof(h->err)];
  memcpy(saved, h->err, sizeof(saved));
  if (hipMemsetAsync(local, 0, sizeof(I) * (size_t)(rows + 1), h->stream) == hipSuccess)
    (void)stitch<I>(h, comm, sp, chunk, n, (const I *)local, 0, rc, (I *)row_ptr_out, nullptr);
  memcpy(h->err, saved, sizeof(saved));
  return rc;
}

// row_ptr values at the split points of the ranks (host copy, world + 1 entries)
template <typename I>
__global__ void k_pick_splits(const I *__restrict__ rp, Splits sp, int world, int64_t *__restrict__ out) {
  const int r = threadIdx.x;
  if (r <= world) out[r] = (int64_t)rp[sp.lo[r]];
}
template <typename I>
__global__ __launch_bounds__(256) void k_rebase_ptr(const I *__restrict__ rp, int64_t count, I *__restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const I base = rp[0];
  for (; i < count; i += stride) out[i] = rp[i] - base;
}
template <typename I>
__global__ __launch_bounds__(256) void k_add_const(I *__restrict__ a, int64_t count, int64_t add) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < count; i += stride) a[i] = (I)((int64_t)a[i] + add);
}

// ---- nnz-balanced row ranges ----------------------------------------------------------------------------------
// len[new row] = entries of the old row mapped there; after an exclusive scan, split k is the first row whose prefix
// reaches k * total / world
template <typename I>
__global__ __launch_bounds__(256) void k_new_row_lengths(const I *__restrict__ rp, const I *__restrict__ row_order, int64_t n,
                                                         int64_t *__restrict__ len, int64_t *__restrict__ bad) {
  int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  bool outside = false;
  for (; u < n; u += stride) {
    const int64_t r = row_order ? (int64_t)row_order[u] : u;
    if (r < 0 || r >= n) outside = true;  // (an order vector of the caller's: never written through)
    else len[r] = (int64_t)(rp[u + 1] - rp[u]);
  }
  if (__any(outside) && sbx_lane() == 0) *bad = 1;
}
__global__ void k_balanced_splits(const int64_t *__restrict__ prefix, int64_t n, const int64_t *__restrict__ total,
                                  int world, int64_t *__restrict__ out) {
  const int k = threadIdx.x;
  if (k > world) return;
  if (k == 0 || k == world) {
    out[k] = k == 0 ? 0 : n;
    return;
  }
  const int64_t target = (*total / world) * k + (*total % world) * k / world;
  int64_t a = 0, b = n;  // first row whose exclusive prefix is >= target
  while (a < b) {
    const int64_t mid = (a + b) >> 1;
    if (prefix[mid] >= target) b = mid;
    else a = mid + 1;
  }
  out[k] = a;
}

}  // namespace

extern "C" int sbx_comm_create(int rank, int world, sbx_allgather_fn allgather, void *user, sbx_comm_t *out) {
  if (!out || !allgather || world < 1 || rank < 0 || rank >= world) return SBX_ERR_BAD_ARG;
  sbx_comm_t c = new sbx_comm_s();
  c->rank = rank;
  c->world = world;
  c->allgather = allgather;
  c->user = user;
  c->nccl_comm = nullptr;
  *out = c;
  return SBX_OK;
}

extern "C" int sbx_comm_unique_id(void *id_out) {
  if (!id_out) return SBX_ERR_BAD_ARG;
  if (!rccl_load()) return SBX_ERR_UNSUPPORTED;
  return g_rccl.GetUniqueId(id_out) == 0 ? SBX_OK : SBX_ERR_HIP;
}

extern "C" int sbx_comm_create_rccl(int device, int rank, int world, const void *unique_id, sbx_comm_t *out) {
  if (!out || !unique_id || world < 1 || rank < 0 || rank >= world) return SBX_ERR_BAD_ARG;
  if (!rccl_load()) return SBX_ERR_UNSUPPORTED;
  if (hipSetDevice(device) != hipSuccess) return SBX_ERR_NO_DEVICE;
  Id128 id;
  memcpy(id.bytes, unique_id, SBX_COMM_ID_BYTES);
  void *nc = nullptr;
  if (g_rccl.CommInitRank(&nc, world, id, rank) != 0) return SBX_ERR_HIP;
  sbx_comm_t c = new sbx_comm_s();
  c->rank = rank;
  c->world = world;
  c->allgather = rccl_allgather;
  c->user = c;
  c->nccl_comm = nc;
  *out = c;
  return SBX_OK;
}

extern "C" int sbx_comm_destroy(sbx_comm_t c) {
  if (!c) return SBX_OK;
  if (c->nccl_comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->nccl_comm);
  delete c;
  return SBX_OK;
}

extern "C" int sbx_comm_rank(sbx_comm_t c, int *rank, int *world) {
  if (!c) return SBX_ERR_BAD_ARG;
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  return SBX_OK;
}

extern "C" int sbx_permute_csr_rows_nnz(sbx_handle_t h, sbx_index_type it, int64_t n, const void *row_ptr,
                                        const void *row_order, int64_t row_begin, int64_t row_end, int64_t *nnz_host) {
  if (!h) return SBX_ERR_BAD_ARG;
  if (it == SBX_I32_N64) return sbx_mixed_permute_csr_rows_nnz(h, n, row_ptr, row_order, row_begin, row_end, nnz_host);
  if (!row_ptr || !nnz_host || n < 0 || row_begin < 0 || row_begin > row_end || row_end > n)
    SBX_FAIL(h, SBX_ERR_BAD_ARG, "sbx_permute_csr_rows_nnz: bad argument");
  SBX_TRY(sbx_arena_begin(h));
  unsigned long long *acc = nullptr;
  SBX_TRY(sbx_salloc(h, 1, &acc));
  SBX_HIP(h, hipMemsetAsync(acc, 0, sizeof(*acc), h->stream));
  if (n > 0) {
    const unsigned grid = sbx_grid_for(n, 256, 2048);
    if (it == SBX_I32)
      SBX_KLAUNCH(h, SBX_K_PERMUTE_PREP, k_range_nnz<int32_t>, dim3(grid), dim3(256), (const int32_t *)row_ptr,
                  (const int32_t *)row_order, n, row_begin, row_end, acc);
    else
      SBX_KLAUNCH(h, SBX_K_PERMUTE_PREP, k_range_nnz<int64_t>, dim3(grid), dim3(256), (const int64_t *)row_ptr,
                  (const int64_t *)row_order, n, row_begin, row_end, acc);
    SBX_LAUNCH_CHECK(h);
  }
  unsigned long long v = 0;
  SBX_TRY(sbx_readback(h, &v, acc, sizeof(v)));
  *nnz_host = (int64_t)v;
  return SBX_OK;
}

extern "C" int sbx_permute_csr_sharded(sbx_handle_t h, sbx_comm_t comm, sbx_index_type it, sbx_value_type vt, int64_t n,
                                       int64_t m, int64_t nnz, const void *row_ptr, const void *col, const void *val,
                                       const void *row_order, const void *col_order, const int64_t *row_splits,
                                       void *row_ptr_out, void *col_out, void *val_out, int64_t out_capacity,
                                       int64_t *shard_offsets_host) {
  if (!h) return SBX_ERR_BAD_ARG;
  if (it == SBX_I32_N64) SBX_FAIL(h, SBX_ERR_UNSUPPORTED, "sbx_permute_csr_sharded: SBX_I32_N64 is not taken by the sharded entry points");
  if (!comm || !row_ptr || !row_ptr_out || n < 0) SBX_FAIL(h, SBX_ERR_BAD_ARG, "sbx_permute_csr_sharded: bad argument");
  Splits sp;
  int64_t chunk = 1;
  SBX_TRY(resolve_splits(h, comm, n, row_splits, &sp, &chunk));
  const int64_t lo = sp.lo[comm->rank], hi = sp.lo[comm->rank + 1];
  SBX_TRY(sbx_arena_begin(h));
  NestGuard guard(h);
  const size_t ib = (size_t)sbx_index_bytes(it);
  void *local = nullptr;
  SBX_TRY(sbx_arena_alloc(h, ib * (size_t)(hi - lo + 1), &local));
  int64_t local_nnz = 0;
  const int rc = sbx_permute_csr_rows(h, it, vt, n, m, nnz, row_ptr, col, val, row_order, col_order, lo, hi, local,
                                      col_out, val_out, out_capacity, &local_nnz);
  if (rc != SBX_OK)  // (a slab that does not fit, a bad order vector, ...): the other ranks must not be left waiting
    return it == SBX_I32 ? stitch_failed<int32_t>(h, comm, sp, chunk, n, local, hi - lo, rc, row_ptr_out)
                         : stitch_failed<int64_t>(h, comm, sp, chunk, n, local, hi - lo, rc, row_ptr_out);
  if (it == SBX_I32)
    return stitch<int32_t>(h, comm, sp, chunk, n, (const int32_t *)local, local_nnz, 0, (int32_t *)row_ptr_out,
                           shard_offsets_host);
  return stitch<int64_t>(h, comm, sp, chunk, n, (const int64_t *)local, local_nnz, 0, (int64_t *)row_ptr_out,
                         shard_offsets_host);
}

extern "C" int sbx_coo_to_csr_sharded(sbx_handle_t h, sbx_comm_t comm, sbx_index_type it, sbx_value_type vt, int64_t n,
                                      int64_t m, int64_t nnz, const void *row, const void *col, const void *val,
                                      const int64_t *row_splits, void *row_ptr_out, void *col_out, void *val_out,
                                      int64_t out_capacity, int64_t *shard_offsets_host) {
  if (!h) return SBX_ERR_BAD_ARG;
  if (it == SBX_I32_N64) SBX_FAIL(h, SBX_ERR_UNSUPPORTED, "sbx_coo_to_csr_sharded: SBX_I32_N64 is not taken by the sharded entry points");
  if (!comm || !row_ptr_out || n < 0 || nnz < 0 || (nnz > 0 && (!row || !col)))
    SBX_FAIL(h, SBX_ERR_BAD_ARG, "sbx_coo_to_csr_sharded: bad argument");
  Splits sp;
  int64_t chunk = 1;
  SBX_TRY(resolve_splits(h, comm, n, row_splits, &sp, &chunk));
  const int64_t lo = sp.lo[comm->rank], hi = sp.lo[comm->rank + 1];
  SBX_TRY(sbx_arena_begin(h));
  NestGuard guard(h);
  const size_t ib = (size_t)sbx_index_bytes(it);
  const int vb = val ? sbx_value_bytes(vt) : 0;
  if (vb < 0) SBX_FAIL(h, SBX_ERR_BAD_ARG, "unknown value type");
  // the rank's nonzeros: the contiguous slice [a, b) of the row-sorted COO
  int64_t *bounds = nullptr;
  SBX_TRY(sbx_salloc(h, 2, &bounds));
  if (it == SBX_I32)
    SBX_KLAUNCH(h, SBX_K_MISC, k_row_bounds<int32_t>, dim3(1), dim3(2), (const int32_t *)row, nnz, lo, hi, bounds);
  else
    SBX_KLAUNCH(h, SBX_K_MISC, k_row_bounds<int64_t>, dim3(1), dim3(2), (const int64_t *)row, nnz, lo, hi, bounds);
  SBX_LAUNCH_CHECK(h);
  int64_t ab[2];
  SBX_TRY(sbx_readback(h, ab, bounds, sizeof(ab)));
  const int64_t a = ab[0], local_nnz = ab[1] - ab[0];
  void *local = nullptr, *rebased = nullptr;
  SBX_TRY(sbx_arena_alloc(h, ib * (size_t)(hi - lo + 1), &local));
  if (local_nnz > out_capacity) {
    snprintf(h->err, sizeof(h->err), "sbx_coo_to_csr_sharded: shard needs %lld entries, capacity %lld",
             (long long)local_nnz, (long long)out_capacity);
    return it == SBX_I32 ? stitch_failed<int32_t>(h, comm, sp, chunk, n, local, hi - lo, SBX_ERR_BAD_ARG, row_ptr_out)
                         : stitch_failed<int64_t>(h, comm, sp, chunk, n, local, hi - lo, SBX_ERR_BAD_ARG, row_ptr_out);
  }
  SBX_TRY(sbx_arena_alloc(h, ib * (size_t)(local_nnz > 0 ? local_nnz : 1), &rebased));
  if (local_nnz > 0) {
    const unsigned grid = sbx_grid_for(local_nnz, 256, 8192);
    if (it == SBX_I32)
      SBX_KLAUNCH(h, SBX_K_MISC, k_rebase<int32_t>, dim3(grid), dim3(256), (const int32_t *)row + a, local_nnz, lo,
                  (int32_t *)rebased);
    else
      SBX_KLAUNCH(h, SBX_K_MISC, k_rebase<int64_t>, dim3(grid), dim3(256), (const int64_t *)row + a, local_nnz, lo,
                  (int64_t *)rebased);
    SBX_LAUNCH_CHECK(h);
  }
  const int rc = sbx_coo_to_csr(h, it, vt, hi - lo, m, local_nnz, rebased, (const char *)col + ib * (size_t)a,
                                val ? (const char *)val + (size_t)vb * (size_t)a : nullptr, local, col_out,
                                val ? val_out : nullptr, SBX_FLAG_ROWS_SORTED);
  if (rc != SBX_OK)
    return it == SBX_I32 ? stitch_failed<int32_t>(h, comm, sp, chunk, n, local, hi - lo, rc, row_ptr_out)
                         : stitch_failed<int64_t>(h, comm, sp, chunk, n, local, hi - lo, rc, row_ptr_out);
  if (it == SBX_I32)
    return stitch<int32_t>(h, comm, sp, chunk, n, (const int32_t *)local, local_nnz, 0, (int32_t *)row_ptr_out,
                           shard_offsets_host);
  return stitch<int64_t>(h, comm, sp, chunk, n, (const int64_t *)local, local_nnz, 0, (int64_t *)row_ptr_out,
                         shard_offsets_host);
}

// A3 sharded (converter/converter_order_two.cc:72-160 by row range; device-to-device edge of the reference:
// converter/converter_order_two_cuda.cu:41-76): rank r expands rows [lo_r, hi_r) of the replicated CSR, i.e. the entries
// [row_ptr[lo_r], row_ptr[hi_r]), into its slab of the COO.  No collective: the shards' positions are row_ptr at the
// split points, which every rank holds.
extern "C" int sbx_csr_to_coo_sharded(sbx_handle_t h, sbx_comm_t comm, sbx_index_type it, sbx_value_type vt, int64_t n,
                                      int64_t m, int64_t nnz, const void *row_ptr, const void *col, const void *val,
                                      const int64_t *row_splits, void *row_out, void *col_out, void *val_out,
                                      int64_t out_capacity, int64_t *shard_offsets_host) {
  if (!h) return SBX_ERR_BAD_ARG;
  if (it == SBX_I32_N64) SBX_FAIL(h, SBX_ERR_UNSUPPORTED, "sbx_csr_to_coo_sharded: SBX_I32_N64 is not taken by the sharded entry points");
  if (!comm || !row_ptr || n < 0 || nnz < 0 || (nnz > 0 && (!col || !row_out || !col_out)))
    SBX_FAIL(h, SBX_ERR_BAD_ARG, "sbx_csr_to_coo_sharded: bad argument");
  Splits sp;
  int64_t chunk = 1;
  SBX_TRY(resolve_splits(h, comm, n, row_splits, &sp, &chunk));
  const int world = comm->world;
  const int64_t lo = sp.lo[comm->rank], hi = sp.lo[comm->rank + 1];
  SBX_TRY(sbx_arena_begin(h));
  NestGuard guard(h);
  const size_t ib = (size_t)sbx_index_bytes(it);
  const int vb = (val && val_out) ? sbx_value_bytes(vt) : 0;
  if (vb < 0) SBX_FAIL(h, SBX_ERR_BAD_ARG, "unknown value type");
  int64_t *cuts_dev = nullptr;
  SBX_TRY(sbx_salloc(h, (size_t)world + 1, &cuts_dev));
  if (it == SBX_I32)
    SBX_KLAUNCH(h, SBX_K_MISC, k_pick_splits<int32_t>, dim3(1), dim3(MAX_WORLD + 1), (const int32_t *)row_ptr, sp, world, cuts_dev);
  else
    SBX_KLAUNCH(h, SBX_K_MISC, k_pick_splits<int64_t>, dim3(1), dim3(MAX_WORLD + 1), (const int64_t *)row_ptr, sp, world, cuts_dev);
  SBX_LAUNCH_CHECK(h);
  std::vector<int64_t> cuts(world + 1, 0);
  SBX_TRY(sbx_readback(h, cuts.data(), cuts_dev, sizeof(int64_t) * (size_t)(world + 1)));
  if (shard_offsets_host)
    for (int r = 0; r <= world; r++) shard_offsets_host[r] = cuts[r];
  const int64_t a = cuts[comm->rank], local_nnz = cuts[comm->rank + 1] - a;
  if (local_nnz > out_capacity)
    SBX_FAIL(h, SBX_ERR_BAD_ARG, "sbx_csr_to_coo_sharded: shard needs %lld entries, capacity %lld", (long long)local_nnz,
             (long long)out_capacity);
  if (local_nnz == 0) return SBX_OK;
  // the rank's rows as a CSR of their own (row_ptr rebased to 0), expanded, then the row ids moved back to [lo, hi)
  void *local_rp = nullptr;
  SBX_TRY(sbx_arena_alloc(h, ib * (size_t)(hi - lo + 1), &local_rp));
  const unsigned g1 = sbx_grid_for(hi - lo + 1, 256, 4096), g2 = sbx_grid_for(local_nnz, 256, 8192);
  if (it == SBX_I32)
    SBX_KLAUNCH(h, SBX_K_MISC, k_rebase_ptr<int32_t>, dim3(g1), dim3(256), (const int32_t *)row_ptr + lo, hi - lo + 1, (int32_t *)local_rp);
  else
    SBX_KLAUNCH(h, SBX_K_MISC, k_rebase_ptr<int64_t>, dim3(g1), dim3(256), (const int64_t *)row_ptr + lo, hi - lo + 1, (int64_t *)local_rp);
  SBX_LAUNCH_CHECK(h);
  SBX_TRY(sbx_csr_to_coo(h, it, vt, hi - lo, m, local_nnz, local_rp, (const char *)col + ib * (size_t)a,
                         vb ? (const char *)val + (size_t)vb * (size_t)a : nullptr, row_out, col_out, vb ? val_out : nullptr,
                         0u));
  if (lo != 0) {
    if (it == SBX_I32) SBX_KLAUNCH(h, SBX_K_MISC, k_add_const<int32_t>, dim3(g2), dim3(256), (int32_t *)row_out, local_nnz, lo);
    else SBX_KLAUNCH(h, SBX_K_MISC, k_add_const<int64_t>, dim3(g2), dim3(256), (int64_t *)row_out, local_nnz, lo);
    SBX_LAUNCH_CHECK(h);
  }
  return SBX_OK;
}

// Row ranges of equal ENTRY counts for the sharded permute ("balance by nnz, not rows" — a power-law
// matrix split into equal row ranges leaves one rank with several times the work): splits_host[k] = the first new row
// whose entries start at or behind k / world of all entries.
extern "C" int sbx_balanced_row_splits(sbx_handle_t h, sbx_index_type it, int64_t n, const void *row_ptr,
                                       const void *row_order, int world, int64_t *splits_host) {
  if (!h) return SBX_ERR_BAD_ARG;
  if (it == SBX_I32_N64) SBX_FAIL(h, SBX_ERR_UNSUPPORTED, "sbx_balanced_row_splits: SBX_I32_N64 is not taken by the sharded entry points");
  if (!row_ptr || !splits_host || n < 0 || world < 1 || world > MAX_WORLD)
    SBX_FAIL(h, SBX_ERR_BAD_ARG, "sbx_balanced_row_splits: bad argument (world 1..%d)", MAX_WORLD);
  SBX_TRY(sbx_arena_begin(h));
  NestGuard guard(h);
  if (n == 0) {
    for (int k = 0; k <= world; k++) splits_host[k] = 0;
    return SBX_OK;
  }
  int64_t *len = nullptr, *total = nullptr, *out = nullptr;
  SBX_TRY(sbx_salloc(h, (size_t)n, &len));
  SBX_TRY(sbx_salloc(h, 1, &total));
  SBX_TRY(sbx_salloc(h, (size_t)world + 2, &out));  // [world + 1]: an order entry outside [0, n) was seen
  // (rows the order never names keep length 0: a vector that is not a permutation gives balanced splits of what it does name)
  SBX_HIP(h, hipMemsetAsync(len, 0, sizeof(int64_t) * (size_t)n, h->stream));
  SBX_HIP(h, hipMemsetAsync(out + world + 1, 0, sizeof(int64_t), h->stream));
  const unsigned grid = sbx_grid_for(n, 256, 8192);
  if (it == SBX_I32)
    SBX_KLAUNCH(h, SBX_K_PERMUTE_PREP, k_new_row_lengths<int32_t>, dim3(grid), dim3(256), (const int32_t *)row_ptr,
                (const int32_t *)row_order, n, len, out + world + 1);
  else
    SBX_KLAUNCH(h, SBX_K_PERMUTE_PREP, k_new_row_lengths<int64_t>, dim3(grid), dim3(256), (const int64_t *)row_ptr,
                (const int64_t *)row_order, n, len, out + world + 1);
  SBX_LAUNCH_CHECK(h);
  SBX_TRY(sbx_exclusive_scan_i64(h, len, len, n, total));
  SBX_KLAUNCH(h, SBX_K_PERMUTE_PREP, k_balanced_splits, dim3(1), dim3(MAX_WORLD + 1), (const int64_t *)len, n,
              (const int64_t *)total, world, out);
  SBX_LAUNCH_CHECK(h);
  std::vector<int64_t> res((size_t)world + 2, 0);
  SBX_TRY(sbx_readback(h, res.data(), out, sizeof(int64_t) * (size_t)(world + 2)));
  if (res[(size_t)world + 1] != 0)
    SBX_FAIL(h, SBX_ERR_BAD_ARG, "sbx_balanced_row_splits: row_order holds an entry outside [0, n)");
  for (int k = 0; k <= world; k++) splits_host[k] = res[(size_t)k];
  return SBX_OK;
}
