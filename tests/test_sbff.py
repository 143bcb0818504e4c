"""SbFF binary containers: the host layer's reader / writer against files the REAL reference
wrote (tests/golden/sbff, made by oracle/make_golden_sbff.py), and — where oracle/_ref is present — a live
exchange in both directions.  No GPU: the container level and the writers do not touch the device
(sparsebase_amd/host/examples/sbff_tool.cc); the readers' format constructors are covered by the GPU suite
(sparsebase_amd/host/tests/test_reference_suite.cc, BinaryOrderTwo.*)."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "sparsebase_amd", "host")
GOLD = os.path.join(ROOT, "tests", "golden", "sbff")


@pytest.fixture(scope="module")
def tool():
    from sparsebase_amd import build
    build.build()
    subprocess.check_call(["make", "-s", "-C", HOST, "all"])
    return os.path.join(HOST, "bin", "sbff_tool")


def dump(tool, path, ok=0):
    p = subprocess.run([tool, "dump", str(path)], capture_output=True, text=True, timeout=60)
    assert p.returncode == ok, p.stdout + p.stderr
    if ok:
        return p.stderr
    out = {}
    for line in p.stdout.splitlines():
        key, _, rest = line.partition(" ")
        if rest.startswith("["):
            meta, _, rest = rest.partition("]")
            out[key + "_meta"] = meta[1:]
        out[key] = rest.split()
    return out


def write(tool, path, kind, dims, tmp_path, **arrays):
    desc = tmp_path / "desc.txt"
    lines = [f"kind {kind}", "dims " + " ".join(str(d) for d in dims)]
    for name, a in arrays.items():
        if a is not None:
            lines.append(name + " " + " ".join(repr(float(x)) if a.dtype.kind == "f" else str(int(x)) for x in a))
    desc.write_text("\n".join(lines) + "\n")
    p = subprocess.run([tool, "write", str(path), str(desc)], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0, p.stdout + p.stderr


def blocks(path):
    """(object header text, {array name: (header text, payload bytes)}) of a container."""
    raw = open(path, "rb").read()
    head = json.loads(raw[:1024].decode())
    assert raw[:1024].rstrip(b" ").endswith(b"}") and len(raw[:1024]) == 1024
    at, arrays = 1024, {}
    for _ in range(head["array_count"]):
        h = json.loads(raw[at:at + 1024].decode())
        size = h["array_size"] * h["type_size"]
        arrays[h["name"]] = (raw[at:at + 1024], raw[at + 1024:at + 1024 + size])
        at += 1024 + size
    assert at == len(raw)
    return raw[:1024], arrays


def as_i(xs):
    return np.array([int(x) for x in xs], np.int32)


def as_f(xs):
    return np.array([float(x) for x in xs], np.float32)


# ------------------------------------------------------------------ files written by the real reference
def test_reads_reference_written_files(tool):
    d = dump(tool, os.path.join(GOLD, "ref_coo.bin"))      # binary_reader_order_two_tests.cc:7-36
    assert d["kind"] == ["coo"] and d["dims"] == ["4", "4"] and d["arrays"] == ["3"]
    assert np.array_equal(as_i(d["row"]), [1, 2, 3, 4]) and np.array_equal(as_i(d["col"]), [5, 6, 7, 8])
    assert np.array_equal(as_f(d["vals"]), np.array([0.1, 0.2, 0.3, 0.4], np.float32))
    d = dump(tool, os.path.join(GOLD, "ref_coo_pattern.bin"))
    assert d["arrays"] == ["2"] and "vals" not in d
    d = dump(tool, os.path.join(GOLD, "ref_csr.bin"))      # :38-70
    assert d["kind"] == ["csr"] and np.array_equal(as_i(d["row_ptr"]), [0, 2, 3, 3, 4])
    assert np.array_equal(as_i(d["col"]), [0, 2, 1, 3])
    assert np.array_equal(as_f(d["vals"]), np.array([0.1, 0.2, 0.3, 0.4], np.float32))
    d = dump(tool, os.path.join(GOLD, "ref_array.bin"))
    assert d["kind"] == ["array"] and d["dims"] == ["5"] and np.array_equal(as_f(d["array"]), [1, 2, 3, 4, 5])


def test_writes_the_same_blocks_as_the_reference(tool, tmp_path):
    """Same object header, same array headers, same payloads (the reference emits the arrays in
    unordered_map order, so the comparison is per block, not per file)."""
    f32 = np.float32
    cases = [
        ("ref_coo.bin", "coo", [4, 4], dict(row=as_i([1, 2, 3, 4]), col=as_i([5, 6, 7, 8]), vals=np.array([0.1, 0.2, 0.3, 0.4], f32))),
        ("ref_coo_pattern.bin", "coo", [4, 4], dict(row=as_i([1, 2, 3, 4]), col=as_i([5, 6, 7, 8]))),
        ("ref_csr.bin", "csr", [4, 4], dict(row_ptr=as_i([0, 2, 3, 3, 4]), col=as_i([0, 2, 1, 3]), vals=np.array([0.1, 0.2, 0.3, 0.4], f32))),
        ("ref_array.bin", "array", [5], dict(array=np.array([1, 2, 3, 4, 5], f32))),
    ]
    for gold, kind, dims, arrays in cases:
        mine = tmp_path / ("mine_" + gold)
        write(tool, mine, kind, dims, tmp_path, **arrays)
        assert blocks(mine) == blocks(os.path.join(GOLD, gold)), gold


# ------------------------------------------------------------------ live exchange with the real reference
@pytest.mark.parametrize("seed", range(6))
def test_exchange_with_the_reference(tool, ref, seed, tmp_path):
    g = np.random.default_rng(500 + seed)
    n = int(g.integers(1, 60))
    m = int(g.integers(1, 60))
    # the reference can only read / write objects with nnz == column count (see the reader's header comment)
    nnz = m
    row = np.sort(g.integers(0, n, nnz)).astype(np.int32)
    col = g.integers(0, m, nnz).astype(np.int32)
    vals = g.standard_normal(nnz).astype(np.float32) if seed % 3 else None
    rp = np.zeros(n + 1, np.int32)
    np.add.at(rp, row + 1, 1)
    rp = np.cumsum(rp).astype(np.int32)
    # reference -> this library
    ref.sbff_write_coo(tmp_path / "r_coo.bin", n, m, row, col, vals)
    d = dump(tool, tmp_path / "r_coo.bin")
    assert d["dims"] == [str(n), str(m)]
    assert np.array_equal(as_i(d["row"]), row) and np.array_equal(as_i(d["col"]), col)
    assert ("vals" in d) == (vals is not None)
    if vals is not None:
        assert np.array_equal(as_f(d["vals"]).view(np.uint32), vals.view(np.uint32))
        ref.sbff_write_csr(tmp_path / "r_csr.bin", n, m, rp, col, vals)
        d = dump(tool, tmp_path / "r_csr.bin")
        assert np.array_equal(as_i(d["row_ptr"]), rp) and np.array_equal(as_i(d["col"]), col)
        assert np.array_equal(as_f(d["vals"]).view(np.uint32), vals.view(np.uint32))
    # this library -> reference
    write(tool, tmp_path / "m_coo.bin", "coo", [n, m], tmp_path, row=row, col=col, vals=vals)
    rn, rm, rrow, rcol, rvals = ref.sbff_read_coo(tmp_path / "m_coo.bin")
    order = np.lexsort((col, row))              # the reference's COO constructor sorts by (row, col)
    assert (rn, rm) == (n, m) and np.array_equal(rrow, row[order]) and np.array_equal(np.sort(rcol), np.sort(col))
    if vals is not None:
        assert sorted(rvals.tolist()) == sorted(vals.tolist())
        write(tool, tmp_path / "m_csr.bin", "csr", [n, m], tmp_path, row_ptr=rp, col=col, vals=vals)
        rn, rm, rrp, rcol, rvals = ref.sbff_read_csr(tmp_path / "m_csr.bin")
        assert (rn, rm) == (n, m) and np.array_equal(rrp, rp) and np.array_equal(np.sort(rcol), np.sort(col))
    a = g.standard_normal(int(g.integers(1, 500))).astype(np.float32)
    write(tool, tmp_path / "m_arr.bin", "array", [len(a)], tmp_path, array=a)
    assert np.array_equal(ref.sbff_read_array(tmp_path / "m_arr.bin").view(np.uint32), a.view(np.uint32))
    ref.sbff_write_array(tmp_path / "r_arr.bin", a)
    assert blocks(tmp_path / "m_arr.bin") == blocks(tmp_path / "r_arr.bin")


# ------------------------------------------------------------------ what the reference cannot express
def test_csr_with_more_nonzeros_than_columns_round_trips(tool, tmp_path):
    rp, col = as_i([0, 3, 5, 9]), as_i([0, 1, 2, 0, 2, 0, 1, 2, 2])
    vals = np.arange(9, dtype=np.float32) / 4
    write(tool, tmp_path / "w.bin", "csr", [3, 3], tmp_path, row_ptr=rp, col=col, vals=vals)
    d = dump(tool, tmp_path / "w.bin")
    assert d["col_meta"] == "signed 9 x4"          # the reference would have stored 3 entries
    assert np.array_equal(as_i(d["col"]), col) and np.array_equal(as_f(d["vals"]), vals)


def _container(name, dims, arrays, endian="little"):
    def block(obj):
        s = json.dumps(obj, separators=(",", ":"), sort_keys=True).encode()
        return s + b" " * (1024 - len(s))
    out = block({"name": name, "array_count": len(arrays), "dimensions": dims, "endian": endian})
    for aname, (tname, tsize, payload, count) in arrays.items():
        out += block({"name": aname, "type": tname, "type_size": tsize, "array_size": count}) + payload
    return out


def test_big_endian_files_are_swapped(tool, tmp_path):
    row, col = [0, 1, 258], [3, 2, 65536]
    vals = [1.5, -2.25, 1e-3]
    p = tmp_path / "be.bin"
    p.write_bytes(_container("coo", [300, 70000], {
        "row": ("signed", 4, struct.pack(">3i", *row), 3),
        "col": ("signed", 4, struct.pack(">3i", *col), 3),
        "vals": ("float", 4, struct.pack(">3f", *vals), 3)}, endian="big"))
    d = dump(tool, p)
    assert d["endian"] == ["big"] and np.array_equal(as_i(d["row"]), row) and np.array_equal(as_i(d["col"]), col)
    assert np.array_equal(as_f(d["vals"]), np.array(vals, np.float32))


def test_type_and_structure_errors(tool, tmp_path):
    ints = struct.pack("<2i", 1, 2)
    p = tmp_path / "bad.bin"
    # indices stored as floats: io/sparse_file_format.h:219-222
    p.write_bytes(_container("coo", [2, 2], {"row": ("float", 4, ints, 2), "col": ("signed", 4, ints, 2)}))
    assert "Type mismatch, array type is float" in dump(tool, p, ok=2)
    # like the reference (std::is_signed_v<float> holds, :224-226), "signed" values pass the class check for a
    # floating-point ValueType and only the element size is compared; "unsigned" ones do not (:229-232)
    p.write_bytes(_container("coo", [2, 2], {"row": ("signed", 4, ints, 2), "col": ("signed", 4, ints, 2),
                                             "vals": ("unsigned", 4, ints, 2)}))
    assert "Type mismatch, array type is unsigned" in dump(tool, p, ok=2)
    # 64-bit indices read as int: :234-236
    p.write_bytes(_container("coo", [2, 2], {"row": ("signed", 8, struct.pack("<2q", 1, 2), 2), "col": ("signed", 4, ints, 2)}))
    assert "Type mismatch, array type has size 8" in dump(tool, p, ok=2)
    p.write_bytes(_container("coo", [2, 2], {"row": ("unsigned", 4, ints, 2), "col": ("signed", 4, ints, 2)}))
    assert "Type mismatch, array type is unsigned" in dump(tool, p, ok=2)
    # payload cut short
    good = _container("coo", [2, 2], {"row": ("signed", 4, ints, 2), "col": ("signed", 4, ints, 2)})
    p.write_bytes(good[:-3])
    assert "truncated" in dump(tool, p, ok=2)
    p.write_bytes(b"not a container")
    assert "truncated" in dump(tool, p, ok=2)
    p.write_bytes(b"x" * 2048)
    assert "SBFF" in dump(tool, p, ok=2)
    assert "does not exist" in dump(tool, tmp_path / "missing.bin", ok=2)
