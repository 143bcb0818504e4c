"""Seeded synthetic sparse-matrix workloads (input synthesis only — no hot-path code).

Everything here builds *inputs* for tests and bench.py: the BASELINE configs
(C2 uniform-random COO, C3 symmetric RMAT, C5 banded) plus small structured
graphs used by the parity tests.  numpy versions run anywhere; the ``*_torch``
versions build the 100M-nnz instances directly in HBM.
"""
import numpy as np


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def csr_from_edges(n, src, dst, idx_dtype=np.int32, m=None):
    """Deduplicated CSR (column-sorted rows) from an edge list; pattern only."""
    m = n if m is None else m
    key = src.astype(np.int64) * np.int64(m) + dst.astype(np.int64)
    key = np.unique(key)
    r = (key // m).astype(np.int64)
    c = (key % m).astype(idx_dtype)
    rp = np.zeros(n + 1, np.int64)
    np.add.at(rp, r + 1, 1)
    rp = np.cumsum(rp).astype(idx_dtype)
    return rp, c


def symmetrize(src, dst):
    return np.concatenate([src, dst]), np.concatenate([dst, src])


def random_symmetric_graph(n, avg_deg=4.0, seed=0, isolated_frac=0.1, self_loop_frac=0.02,
                           n_blocks=3, idx_dtype=np.int32):
    """Multi-component symmetric graph with isolated vertices and some self loops."""
    g = _rng(seed)
    active = np.nonzero(g.random(n) >= isolated_frac)[0]
    if len(active) < 2:
        active = np.arange(min(n, 2))
    # split the active vertices into blocks so that several components exist
    blocks = np.array_split(g.permutation(active), max(1, n_blocks))
    src, dst = [], []
    for b in blocks:
        if len(b) < 2:
            continue
        e = max(1, int(len(b) * avg_deg / 2))
        src.append(b[g.integers(0, len(b), e)])
        dst.append(b[g.integers(0, len(b), e)])
    src = np.concatenate(src) if src else np.zeros(0, np.int64)
    dst = np.concatenate(dst) if dst else np.zeros(0, np.int64)
    keep = src != dst
    src, dst = src[keep], dst[keep]
    loops = active[g.random(len(active)) < self_loop_frac]
    s, d = symmetrize(src, dst)
    s = np.concatenate([s, loops])
    d = np.concatenate([d, loops])
    return csr_from_edges(n, s, d, idx_dtype)


def path_graph(n, idx_dtype=np.int32, shuffle_seed=None):
    ids = np.arange(n) if shuffle_seed is None else _rng(shuffle_seed).permutation(n)
    s, d = symmetrize(ids[:-1], ids[1:])
    return csr_from_edges(n, s, d, idx_dtype)


def star_graph(n, idx_dtype=np.int32, centre=0):
    leaves = np.array([i for i in range(n) if i != centre])
    s, d = symmetrize(np.full(len(leaves), centre), leaves)
    return csr_from_edges(n, s, d, idx_dtype)


def clique_graph(n, idx_dtype=np.int32):
    a, b = np.meshgrid(np.arange(n), np.arange(n))
    keep = a != b
    return csr_from_edges(n, a[keep], b[keep], idx_dtype)


def grid_graph(rows, cols, idx_dtype=np.int32, shuffle_seed=None):
    n = rows * cols
    ids = np.arange(n).reshape(rows, cols)
    if shuffle_seed is not None:
        ids = _rng(shuffle_seed).permutation(n).reshape(rows, cols)
    s = np.concatenate([ids[:, :-1].ravel(), ids[:-1, :].ravel()])
    d = np.concatenate([ids[:, 1:].ravel(), ids[1:, :].ravel()])
    s, d = symmetrize(s, d)
    return csr_from_edges(n, s, d, idx_dtype)


def rmat_symmetric(scale, edge_factor=16, seed=1, a=0.57, b=0.19, c=0.19, idx_dtype=np.int32):
    """Symmetric RMAT (Graph500 parameters), deduplicated, no self loops (config C3)."""
    g = _rng(seed)
    n = 1 << scale
    e = n * edge_factor
    src = np.zeros(e, np.int64)
    dst = np.zeros(e, np.int64)
    ab, abc = a + b, a + b + c
    for bit in range(scale):
        r = g.random(e)
        sbit = r >= ab                      # quadrants c,d -> row bit set
        dbit = ((r >= a) & (r < ab)) | (r >= abc)   # quadrants b,d -> col bit set
        src |= sbit.astype(np.int64) << bit
        dst |= dbit.astype(np.int64) << bit
    keep = src != dst
    s, d = symmetrize(src[keep], dst[keep])
    return csr_from_edges(n, s, d, idx_dtype)


def banded_symmetric(n, half_bandwidth, per_row=12, seed=2, idx_dtype=np.int32):
    """Symmetric banded pattern: every |i-j| <= half_bandwidth (config C5)."""
    g = _rng(seed)
    e = n * per_row
    src = g.integers(0, n, e)
    off = g.integers(-half_bandwidth, half_bandwidth + 1, e)
    dst = np.clip(src + off, 0, n - 1)
    s, d = symmetrize(src, dst)
    diag = np.arange(n)
    return csr_from_edges(n, np.concatenate([s, diag]), np.concatenate([d, diag]), idx_dtype)


def uniform_random_coo(n, m, nnz, seed=3, idx_dtype=np.int32, shuffled=False, val_dtype=np.float32):
    """nnz distinct uniform-random coordinates, row-major sorted unless shuffled (config C2)."""
    g = _rng(seed)
    key = np.unique(g.integers(0, np.int64(n) * np.int64(m), int(nnz * 1.05) + 16))
    while len(key) < nnz:
        key = np.unique(np.concatenate([key, g.integers(0, np.int64(n) * np.int64(m), nnz)]))
    key = np.sort(g.choice(key, nnz, replace=False)) if len(key) > nnz else key
    if shuffled:
        key = g.permutation(key)
    row = (key // m).astype(idx_dtype)
    col = (key % m).astype(idx_dtype)
    val = None if val_dtype is None else np.arange(nnz).astype(val_dtype)
    return row, col, val


def random_rect_csr(n, m, nnz, seed=4, idx_dtype=np.int32, sort_rows=True, dup_frac=0.0):
    """Rectangular random CSR; optionally with unsorted rows / duplicate coordinates."""
    g = _rng(seed)
    r = g.integers(0, n, nnz)
    c = g.integers(0, m, nnz)
    if dup_frac > 0 and nnz > 1:
        k = int(nnz * dup_frac)
        pick = g.integers(0, nnz, k)
        r = np.concatenate([r, r[pick]])
        c = np.concatenate([c, c[pick]])
    order = np.argsort(r, kind="stable")
    r, c = r[order], c[order]
    if sort_rows:
        order = np.lexsort((c, r))
        r, c = r[order], c[order]
    rp = np.zeros(n + 1, np.int64)
    np.add.at(rp, r + 1, 1)
    return np.cumsum(rp).astype(idx_dtype), c.astype(idx_dtype)


def random_permutation(n, seed=5, idx_dtype=np.int32):
    return _rng(seed).permutation(n).astype(idx_dtype)


# ----------------------------------------------------------------------------
# torch (device) builders for the full-size BASELINE workloads
# ----------------------------------------------------------------------------
def _csr_from_keys_torch(n, m, key, idx_dtype):
    import torch
    key = torch.unique(key)  # sorted, deduplicated
    r = torch.div(key, m, rounding_mode="floor")
    c = (key - r * m).to(idx_dtype)
    counts = torch.bincount(r, minlength=n)
    rp = torch.zeros(n + 1, dtype=torch.int64, device=key.device)
    torch.cumsum(counts, 0, out=rp[1:])
    return rp.to(idx_dtype), c


def rmat_symmetric_torch(scale, edge_factor=16, seed=1, a=0.57, b=0.19, c=0.19, device="cuda",
                         chunk=1 << 24):
    """Device-side symmetric RMAT -> (row_ptr, col) int32 tensors (config C3)."""
    import torch
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    n = 1 << scale
    e = n * edge_factor
    ab, abc = a + b, a + b + c
    keys = []
    for start in range(0, e, chunk):
        cnt = min(chunk, e - start)
        src = torch.zeros(cnt, dtype=torch.int64, device=device)
        dst = torch.zeros(cnt, dtype=torch.int64, device=device)
        for bit in range(scale):
            r = torch.rand(cnt, generator=gen, device=device)
            src |= (r >= ab).to(torch.int64) << bit
            dst |= (((r >= a) & (r < ab)) | (r >= abc)).to(torch.int64) << bit
        keep = src != dst
        src, dst = src[keep], dst[keep]
        keys.append(torch.unique(torch.cat([src * n + dst, dst * n + src])))
    return _csr_from_keys_torch(n, n, torch.cat(keys), torch.int32)


def banded_symmetric_torch(n, half_bandwidth, per_row=12, seed=2, device="cuda"):
    import torch
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    e = n * per_row
    src = torch.randint(0, n, (e,), generator=gen, device=device, dtype=torch.int64)
    off = torch.randint(-half_bandwidth, half_bandwidth + 1, (e,), generator=gen, device=device,
                        dtype=torch.int64)
    dst = torch.clamp(src + off, 0, n - 1)
    diag = torch.arange(n, device=device, dtype=torch.int64)
    key = torch.cat([src * n + dst, dst * n + src, diag * n + diag])
    return _csr_from_keys_torch(n, n, key, torch.int32)


def uniform_random_coo_torch(n, m, nnz, seed=3, device="cuda", shuffled=False):
    import torch
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    key = torch.unique(torch.randint(0, n * m, (int(nnz * 1.02) + 1024,), generator=gen, device=device,
                                     dtype=torch.int64))
    assert key.numel() >= nnz, "increase oversampling"
    key = key[:nnz] if not shuffled else key[torch.randperm(key.numel(), generator=gen, device=device)[:nnz]]
    if not shuffled:
        key, _ = torch.sort(key)
    row = torch.div(key, m, rounding_mode="floor").to(torch.int32)
    col = (key % m).to(torch.int32)
    val = torch.arange(nnz, device=device, dtype=torch.float32)
    return row, col, val
