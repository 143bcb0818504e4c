"""Runs the C++ host-layer programs (sparsebase_amd/host): the SparseBase-API mirror over the C ABI."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "sparsebase_amd", "host")
BIN = os.path.join(HOST, "bin")


@pytest.fixture(scope="module")
def built():
    from sparsebase_amd import build
    build.build()
    subprocess.check_call(["make", "-s", "-C", HOST, "all"])
    return BIN


def run(path, *args, ok=(0,), timeout=120, attempts=2):
    """Runs a host program.  A hung or crashed PROCESS is retried once (fresh GPU boxes occasionally stall
    a child's device start-up) and the first failure is reported as a warning; wrong OUTPUT is never retried —
    the callers compare the results of the run that succeeded."""
    import warnings
    last = None
    for attempt in range(attempts):
        try:
            p = subprocess.run([path, *args], capture_output=True, text=True, timeout=timeout)
        except subprocess.TimeoutExpired as e:
            last = f"TIMEOUT after {timeout}s: {path} {' '.join(map(str, args))}\n{e.stdout}\n{e.stderr}"
        else:
            if p.returncode in ok:
                if last:
                    warnings.warn("first attempt failed, retry succeeded: " + last)
                return p.stdout
            last = f"{path} {' '.join(map(str, args))} rc={p.returncode}\n{p.stdout}\n{p.stderr}"
    raise AssertionError(last)


def write_mtx(path, n, m, row, col, symmetric=False):
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate pattern " + ("symmetric" if symmetric else "general") + "\n")
        f.write(f"{n} {m} {len(row)}\n")
        for r, c in zip(row, col):
            f.write(f"{r + 1} {c + 1}\n")


def test_host_logic_cpu(built):
    out = run(os.path.join(built, "test_host_logic"))
    assert "0 failures" in out and "FAIL" not in out


@pytest.mark.gpu
def test_reference_suite_gpu(built):
    out = run(os.path.join(built, "test_reference_suite"))
    assert "0 failures" in out and "FAIL" not in out, out


@pytest.mark.gpu
def test_examples_gpu(built, tmp_path, golden_dir):
    # C1 / H1: the degree_order flow on ash958 (958 x 292, 1916 nnz, every degree 2)
    z = np.load(os.path.join(golden_dir, "ash958.npz"))
    n, m, _ = (int(x) for x in z["dims"])
    mtx = str(tmp_path / "ash958.mtx")
    write_mtx(mtx, n, m, z["file_row"], z["file_col"])
    out = run(os.path.join(built, "degree_order"), mtx)
    assert "Number of vertices: 958" in out and "Number of edges: 1916" in out
    assert "Order is correct." in out and "Transformation is correct." in out and "Inversion is correct." in out
    assert "first/last of permutation: 957 0" in out  # [957, 956, ..., 0]
    # H2: rcm_order / gray_order on a symmetric graph with a multiple-of-16 size
    from sparsebase_amd import synth
    rp, col = synth.grid_graph(16, 32, shuffle_seed=3)
    rows = np.repeat(np.arange(len(rp) - 1), np.diff(rp))
    sym = str(tmp_path / "grid.mtx")
    write_mtx(sym, len(rp) - 1, len(rp) - 1, rows, col)
    out = run(os.path.join(built, "rcm_order"), sym)
    assert "Order is correct" in out and "NOT" not in out
    bw = int(out.strip().split("bandwidth after RCM:")[1])
    assert bw <= 40  # a 16x32 grid reorders to bandwidth ~16-17; shuffled ids start at ~500
    out = run(os.path.join(built, "gray_order"), sym)
    assert "Order is correct" in out and "NOT" not in out
    # H3: expected stdout of format_conversion.cc:10-54
    out = run(os.path.join(built, "format_conversion"))
    assert out.split() == ["CSR", "10,20,30,40,50,60,", "0,1,1,2,3,3,", "0,2,4,6,6,6,6,", "COO",
                           "10,20,30,40,50,60,", "0,0,1,1,2,2,", "0,1,1,2,3,3,"]


def _run_cli(built, tmp_path, kind, rp, col, n, m, extra=()):
    (tmp_path / "rp.bin").write_bytes(np.ascontiguousarray(rp, np.int32).tobytes())
    (tmp_path / "col.bin").write_bytes(np.ascontiguousarray(col, np.int32).tobytes())
    out = tmp_path / "out.bin"
    run(os.path.join(built, "reorder_cli"), kind, str(tmp_path / "rp.bin"), str(tmp_path / "col.bin"), str(out),
        str(n), str(m), *[str(x) for x in extra])
    return np.frombuffer(out.read_bytes(), np.int32)


@pytest.mark.gpu
def test_cpp_api_reorderers_vs_reference_fixtures(built, tmp_path, golden_dir):
    """Gray (device keys + host ordering stage), RCM and Degree through the C++ API, host and device formats."""
    import json
    z = np.load(os.path.join(golden_dir, "small_cases.npz"))
    with open(os.path.join(golden_dir, "small_cases.json")) as f:
        meta = json.load(f)
    checked = 0
    for name, info in meta.items():
        rp, col = z[f"{name}/row_ptr"], z[f"{name}/col"]
        n = len(rp) - 1
        if n == 0:
            continue
        for res, thr, grp in info["gray"]:
            for dev_flag in ((), ("--device",)):
                got = _run_cli(built, tmp_path, "gray", rp, col, n, n, (res, thr, grp) + dev_flag)
                assert np.array_equal(got, z[f"{name}/gray_{res}_{thr}_{grp}"]), (name, res, thr, grp, dev_flag)
                checked += 1
        if name in ("rmat12", "sym_d", "grid_shuffled"):
            assert np.array_equal(_run_cli(built, tmp_path, "rcm", rp, col, n, n, ("--device",)), z[f"{name}/rcm"])
            assert np.array_equal(_run_cli(built, tmp_path, "degree_desc", rp, col, n, n), z[f"{name}/degree_desc"])
    assert checked >= 40


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["rmat16_ef8", "banded_64k_w16", "banded_64k_w4096"])
def test_cpp_api_gray_digests(built, tmp_path, golden_dir, name):
    """n >= 64K rows: large enough that std::sort's tie order matters."""
    import hashlib
    import json
    from sparsebase_amd import synth
    with open(os.path.join(golden_dir, "digests.json")) as f:
        d = json.load(f)[name]
    rp, col = getattr(synth, d["generator"])(**d["args"])
    n = len(rp) - 1
    for res, thr, grp, key in ((32, 10, 4, "gray_32_10_4"), (16, 20, 1, "gray_16_20_1")):
        got = _run_cli(built, tmp_path, "gray", rp, col, n, n, (res, thr, grp))
        assert hashlib.sha256(got.tobytes()).hexdigest() == d[key], (name, key)
