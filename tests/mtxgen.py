"""Random Matrix Market coordinate files for the ingest tests (shared by the CPU and GPU tests)."""
import numpy as np


def random_mtx(seed, field="real", symmetry="general", n=None, nnz=None, messy=False):
    """Returns (banner + comment + size line, entry section, n, m, L).  Coordinates are distinct (and, for
    symmetric files, in the lower triangle) so that the COO constructor's unstable sort is well defined."""
    g = np.random.default_rng(seed)
    n = int(g.integers(1, 300)) if n is None else n
    m = n if symmetry != "general" else int(g.integers(1, 300))
    want = int(g.integers(0, 3000)) if nnz is None else nnz
    key = np.unique(g.integers(0, n * m, want))
    r, c = key // m, key % m
    if symmetry != "general":
        lo = r >= c if symmetry == "symmetric" else r > c     # skew-symmetric files hold no diagonal
        r, c = r[lo], c[lo]
    p = g.permutation(len(r))
    r, c = r[p], c[p]
    L = len(r)
    seps = [" ", "  ", "\t", " \t "] if messy else [" "]
    eols = ["\n", "\r\n", " \n", "\n\n"] if messy else ["\n"]
    out = []
    for i in range(L):
        sep = seps[int(g.integers(len(seps)))]
        line = f"{r[i] + 1}{sep}{c[i] + 1}"
        if field != "pattern":
            if field == "integer":
                v = str(int(g.integers(-10 ** 6, 10 ** 6)))
            else:
                kind = int(g.integers(5))
                x = float(g.standard_normal()) * 10.0 ** float(g.integers(-30, 30))
                v = {0: repr(x), 1: "%.17g" % x, 2: "%.6e" % x, 3: "%.3f" % (x % 1000.0),
                     4: "%d" % int(g.integers(-99, 99))}[kind]
                if messy and g.random() < 0.1:
                    v = v.replace("e", "E")
                if messy and g.random() < 0.05 and not v.startswith("-"):
                    v = "+" + v
            line += sep + v
        out.append(line + eols[int(g.integers(len(eols)))])
    body = "".join(out)
    if messy and body.endswith("\n") and g.random() < 0.5:
        body = body.rstrip("\n")     # no newline at the end of the file
    head = f"%%MatrixMarket matrix coordinate {field} {symmetry}\n% a comment line\n{n} {m} {L}\n"
    return head, body, n, m, L


def random_edge_list(seed, weighted=False, n=None, edges=None, dup_frac=0.2, self_frac=0.05):
    """Edge list text with duplicate edges (carrying EQUAL weights, so that which duplicate survives
    std::unique after the reference's unstable sort does not matter) and self loops."""
    g = np.random.default_rng(seed)
    n = int(g.integers(2, 400)) if n is None else n
    e = int(g.integers(0, 3000)) if edges is None else edges
    u, v = g.integers(0, n, e), g.integers(0, n, e)
    loops = g.random(e) < self_frac
    v[loops] = u[loops]
    k = int(e * dup_frac)
    if e and k:
        src = g.integers(0, e, k)
        u, v = np.concatenate([u, u[src]]), np.concatenate([v, v[src]])
    # weight = a function of the undirected pair, so every duplicate (and a reverse edge written by read_undirected) agrees
    lo, hi = np.minimum(u, v), np.maximum(u, v)
    w = ((lo * 7919 + hi * 104729) % 1000) / 8.0 - 30.0
    p = g.permutation(len(u))
    lines = []
    for i in p:
        lines.append(f"{u[i]} {v[i]}" + (f" {float(w[i])!r}" if weighted else ""))
    return "\n".join(lines) + ("\n" if lines else "")
