#!/usr/bin/env python3
"""Matrix Market ingest: device parse of a real-valued coordinate file (text resident in HBM) next to the
reference's MTXReader::ReadCOO on the host (which includes its COO-constructor sort)."""
import io, json, os, sys, tempfile, time, numpy as np, pandas as pd, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from sparsebase_amd import ops
import orc
L = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
n = 1 << 20
g = np.random.default_rng(5)
key = np.unique(g.integers(0, n * n, int(L * 1.01)))[:L]
key = key[g.permutation(len(key))]
df = pd.DataFrame({"r": key // n + 1, "c": key % n + 1, "v": g.standard_normal(len(key))})
buf = io.StringIO()
df.to_csv(buf, sep=" ", header=False, index=False, float_format=os.environ.get("MTX_FLOAT_FORMAT", "%.17g"))  # e.g. MTX_FLOAT_FORMAT=%.8g: values the one-operation fast path converts
body = buf.getvalue().encode()
L = len(key)
text = torch.frombuffer(bytearray(body), dtype=torch.uint8).cuda()
def run():
    row, col, val = ops.mtx_parse_coordinate(text, n, n, L, 3, 0, True, False, torch.int32, torch.float64)
    ops.coo_sort_(n, n, row, col, val)
    return row, col, val
run(); torch.cuda.synchronize()
t = time.perf_counter(); row, col, val = run(); torch.cuda.synchronize(); gpu_s = time.perf_counter() - t
t = time.perf_counter(); ops.mtx_parse_coordinate(text, n, n, L, 3, 0, True, False, torch.int32, torch.float64); torch.cuda.synchronize(); parse_s = time.perf_counter() - t
res = dict(entries=L, text_mb=round(len(body) / 1e6, 1), gpu_parse_s=round(parse_s, 4), gpu_parse_plus_sort_s=round(gpu_s, 4),
           parse_gb_s=round(len(body) / parse_s / 1e9, 1), pcie_upload_s_at_50GBs=round(len(body) / 50e9, 4))
if orc.ref_available():
    path = os.path.join(tempfile.mkdtemp(), "big.mtx")
    with open(path, "wb") as f:
        f.write(f"%%MatrixMarket matrix coordinate real general\n{n} {n} {L}\n".encode()); f.write(body)
    ref = orc.Ref()
    t = time.perf_counter(); rn, rm, rrow, rcol, rval = ref.mtx_read(path, True, False, np.int32, np.float64, cap=L + 8); ref_s = time.perf_counter() - t
    res.update(reference_s=round(ref_s, 2), speedup=round(ref_s / gpu_s, 1),
               identical=bool(np.array_equal(row.cpu().numpy(), rrow) and np.array_equal(col.cpu().numpy(), rcol)
                              and np.array_equal(val.cpu().numpy().view(np.uint8), rval.view(np.uint8))))
print(json.dumps(res, indent=1))
