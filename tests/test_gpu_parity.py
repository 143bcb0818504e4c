"""GPU parity tests: the HIP path (through the C ABI) against the oracle, the
reference's own known-answer vectors and the reference-generated golden fixtures.
Bit-exact everywhere (integer / index / byte-payload work).
"""
import json
import os

import numpy as np
import pytest

from sparsebase_amd import synth

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests selected but no GPU is visible (the HIP path has no CPU fallback)")
    from sparsebase_amd import ops as _ops
    return _ops


def dev(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return None if t is None else t.cpu().numpy()


def same(got, want):
    for g, w in zip(got, want):
        if w is None:
            assert g is None
        else:
            assert np.array_equal(host(g) if torch.is_tensor(g) else g, w)


@pytest.fixture(scope="module")
def kat(golden_dir):
    with open(os.path.join(golden_dir, "reference_tests.json")) as f:
        return json.load(f)


def a32(x):
    return np.array(x, np.int32)


# ----------------------------------------------------------------------------- CSC conversions
def test_kat_csc(ops, kat):
    # converter/converter_order_two_tests.cc:49-100, common.inc:14-16 (the reference pads col_ptr to n + 1 entries)
    k = kat["converter_12x9"]
    n, m = k["n"], k["m"]
    for cp, ro, vo in (ops.csr_to_csc(n, m, dev(a32(k["csr_row_ptr"])), dev(a32(k["csr_col"])), dev(a32(k["csr_vals"]))),
                       ops.coo_to_csc(n, m, dev(a32(k["coo_row"])), dev(a32(k["coo_col"])), dev(a32(k["coo_vals"])))):
        assert host(cp).tolist() == k["csc_col_ptr"][:m + 1]
        assert host(ro).tolist() == k["csc_row"] and host(vo).tolist() == k["csc_vals"]


@pytest.mark.parametrize("seed", range(6))
def test_csc_vs_oracle(ops, oracle, seed):
    g = np.random.default_rng(700 + seed)
    n, m = (int(g.integers(1, 3000)), int(g.integers(1, 3000))) if seed else (1 << 15, 1 << 15)
    nnz = int(g.integers(0, 200000)) if seed else 600000
    row = g.integers(0, n, nnz).astype(np.int32)
    col = g.integers(0, m, nnz).astype(np.int32)
    if seed % 2 == 0:
        o = np.lexsort((col, row))
        row, col = row[o], col[o]
    else:  # shuffled input: distinct coordinates so the constructor's unstable pair sort is well defined
        key = np.unique(row.astype(np.int64) * m + col)
        key = key[g.permutation(len(key))]
        row, col = (key // m).astype(np.int32), (key % m).astype(np.int32)
    nnz = len(row)
    for val in (None, g.integers(-9, 9, nnz).astype(np.int32), g.random(nnz).astype(np.float32), g.random(nnz)):
        same(ops.coo_to_csc(n, m, dev(row), dev(col), dev(val)), oracle.coo_to_csc(n, m, row, col, val))
    srow, scol, _ = oracle.coo_sort(row, col, None)
    rp, cc, _ = oracle.coo_to_csr(n, srow, scol)
    val = g.random(nnz).astype(np.float32)
    same(ops.csr_to_csc(n, m, dev(rp), dev(cc), dev(val)), oracle.csr_to_csc(m, rp, cc, val))
    # int64 tuple: native 64-bit kernels (a (column, source index) sort + a gather), every value width, both entry points
    rp64, cc64, row64, col64 = (a.astype(np.int64) for a in (rp, cc, row, col))
    for v in (val.astype(np.float64), val, None):
        same(ops.csr_to_csc(n, m, dev(rp64), dev(cc64), dev(v)), oracle.csr_to_csc(m, rp64, cc64, v))
        same(ops.coo_to_csc(n, m, dev(row64), dev(col64), dev(v)), oracle.coo_to_csc(n, m, row64, col64, v))
    # transposing twice restores the CSR
    cp, ro, vo = ops.csr_to_csc(n, m, dev(rp), dev(cc), dev(val))
    back = ops.csr_to_csc(m, n, cp, ro, vo)
    same(back, (rp, cc, val))


def test_csc_degenerate(ops, oracle):
    z = np.zeros(0, np.int32)
    same(ops.coo_to_csc(5, 7, dev(z), dev(z), None), oracle.coo_to_csc(5, 7, z, z, None))
    same(ops.csr_to_csc(3, 4, dev(np.zeros(4, np.int32)), dev(z), None), oracle.csr_to_csc(4, np.zeros(4, np.int32), z, None))
    one = np.array([2], np.int32)
    same(ops.coo_to_csc(3, 3, dev(one), dev(one), dev(np.array([1.5], np.float32))),
         oracle.coo_to_csc(3, 3, one, one, np.array([1.5], np.float32)))


# ----------------------------------------------------------------------------- Matrix Market ingest
SYMM = {"general": 0, "symmetric": 1, "skew-symmetric": 2}


@pytest.mark.parametrize("seed", range(8))
def test_mtx_parse_vs_oracle(ops, oracle, seed):
    import mtxgen
    field = ["real", "integer", "pattern", "real"][seed % 4]
    symmetry = ["general", "symmetric", "skew-symmetric", "symmetric"][(seed // 2) % 4]
    if field == "pattern" and symmetry == "skew-symmetric":
        symmetry = "symmetric"
    head, body, n, m, L = mtxgen.random_mtx(3000 + seed, field, symmetry, messy=seed >= 4,
                                            n=(5000 if seed == 7 else None), nnz=(400000 if seed == 7 else None))
    text = torch.frombuffer(bytearray(body.encode() or b" "), dtype=torch.uint8).cuda()
    fields = 2 if field == "pattern" else 3
    cases = [(None, None)] if field == "pattern" else \
        ([(np.int32, torch.int32), (np.int64, torch.int64)] if field == "integer" else
         [(np.float32, torch.float32), (np.float64, torch.float64)])
    for vnp, vt in cases:
        for upper in (False, True) if symmetry != "general" else (False,):
            for zero in (True, False):
                for inp, it in ((np.int32, torch.int32), (np.int64, torch.int64)):
                    want = oracle.mtx_parse(body.encode(), L, fields, SYMM[symmetry], zero, upper, inp, vnp)
                    got = ops.mtx_parse_coordinate(text, n, m, L, fields, SYMM[symmetry], zero, upper, it, vt)
                    assert np.array_equal(host(got[0]), want[0]) and np.array_equal(host(got[1]), want[1])
                    if vnp is not None:
                        assert np.array_equal(host(got[2]).view(np.uint8), want[2].view(np.uint8))
    if field != "pattern":   # values present but not wanted
        got = ops.mtx_parse_coordinate(text, n, m, L, 3, SYMM[symmetry], True, False, torch.int32, None)
        want = oracle.mtx_parse(body.encode(), L, 3, SYMM[symmetry], True, False, np.int32, None)
        assert got[2] is None and np.array_equal(host(got[0]), want[0]) and np.array_equal(host(got[1]), want[1])


def test_mtx_values_through_every_conversion_path(ops, oracle):
    """Decimal -> binary on the DEVICE (sbx_dec2bin.h): values for the one-operation fast path (few digits, small powers
    of ten) and around its limits, 16 - 19 digit values for Eisel-Lemire incl. subnormals, the largest finite values,
    halfway cases and overflow, and 20+ digit strings for the exact path — against the oracle's stream extraction."""
    g = np.random.default_rng(77)
    toks = ["5e-324", "4.9406564584124654e-324", "2.4703282292062327e-324", "2.4703282292062328e-324",
            "2.2250738585072014e-308", "2.2250738585072011e-308", "1.7976931348623157e308", "1.7976931348623158e308",
            "9007199254740993", "9007199254740992", "9007199254740991", "1e23", "8.5e22", "1e22", "1e-22", "1.4e-45", "7e-46",
            "3.4028235e38", "3.4028236e38", "1.17549435e-38", "16777217", "16777216", "16777215", "0.1", "0.3", "4.35",
            "1.0000000000000002", "1.0000000000000001110223024625156540424", "123456789012345678",
            "1234567890123456789", "12345678901234567890", "0.000001", "6.02214076e23", "1e-320", "9.5e-324"]
    for _ in range(30000):
        k = int(g.integers(0, 6))
        if k == 0:    # a double printed with 17 digits
            x = np.frombuffer(g.bytes(8), np.float64)[0]
            if not np.isfinite(x): continue
            toks.append("%.17g" % x)
        elif k == 1:  # a float printed with 9 digits
            x = np.frombuffer(g.bytes(4), np.float32)[0]
            if not np.isfinite(x): continue
            toks.append("%.9g" % float(x))
        elif k == 2:  # few digits, small exponents
            toks.append("%de%d" % (int(g.integers(0, 1 << 53)) >> int(g.integers(0, 53)), int(g.integers(-26, 27))))
        elif k == 3:  # digits and exponent over the whole range
            nd = int(g.integers(1, 20))
            toks.append("".join(str(int(g.integers(1 if i == 0 else 0, 10))) for i in range(nd)) + "e%d" % int(g.integers(-345, 310)))
        elif k == 4:  # halfway between adjacent doubles, printed with 19 digits
            mnt = (int(g.integers(0, 1 << 53)) | (1 << 52)) * 2 + 1
            toks.append("%.19g" % np.ldexp(float(mnt), int(g.integers(-20, 20)) - 1))
        else:         # long digit strings (exact path)
            nd = int(g.integers(20, 38))
            toks.append("".join(str(int(g.integers(1 if i == 0 else 0, 10))) for i in range(nd)) + "e%d" % int(g.integers(-330, 290)))
    toks = [t.lstrip("-") for t in toks]  # (the sign is added per line below)

    def parse(tk, vt):
        body = "".join("%d %d %s%s\n" % (i % 97 + 1, i % 89 + 1, "-" if i % 3 == 0 else "", t) for i, t in enumerate(tk))
        text = torch.frombuffer(bytearray(body.encode()), dtype=torch.uint8).cuda()
        return body, host(ops.mtx_parse_coordinate(text, 100, 100, len(tk), 3, 0, True, False, torch.int32, vt)[2])
    # doubles: Python's float() is strtod (correctly rounded; the stream extraction of the oracle refuses the values
    # that overflow or underflow, which the device converts to inf / 0 like strtod)
    _, got = parse(toks, torch.float64)
    want = np.array([(-1.0 if i % 3 == 0 else 1.0) * float(t) for i, t in enumerate(toks)], np.float64)
    bad = np.nonzero(got.view(np.uint64) != want.view(np.uint64))[0]
    assert len(bad) == 0, [toks[i] for i in bad[:5]]
    # floats: the tokens a stream reads into a float without failing, against the oracle's extraction
    ftoks = [t for t in toks if float(t) == 0.0 or 1.2e-38 < abs(float(t)) < 3.4e38]
    body, got = parse(ftoks, torch.float32)
    want = oracle.mtx_parse(body.encode(), len(ftoks), 3, 0, True, False, np.int32, np.float32)[2]
    bad = np.nonzero(got.view(np.uint32) != want.view(np.uint32))[0]
    assert len(bad) == 0, [ftoks[i] for i in bad[:5]]


def test_kat_mtx_reader(ops, kat):
    # io/mtx_reader_tests.cc:58-260 on the fixtures of io/reader_data.inc, then the COO constructor's sort
    for name, symm in (("general", 0), ("symmetric", 1), ("skew-symmetric", 2)):
        k = kat["mtx_reader"][name]
        for key, fields, vt in (("text", 2, None), ("text_values", 3, torch.float32)):
            lines = k[key].split("\n")
            i = 0
            while lines[i].startswith("%"):
                i += 1
            n, m, L = (int(x) for x in lines[i].split())
            body = "\n".join(lines[i + 1:]).encode()
            text = torch.frombuffer(bytearray(body), dtype=torch.uint8).cuda()
            row, col, val = ops.mtx_parse_coordinate(text, n, m, L, fields, symm, True, False, torch.int32, vt)
            row, col = row.clone(), col.clone()
            val = None if val is None else val.clone()
            ops.coo_sort_(n, m, row, col, val)
            assert host(row).tolist() == k["row"] and host(col).tolist() == k["col"]
            if vt is not None:
                assert np.array_equal(host(val), np.array(k["vals"], np.float32))


@pytest.mark.parametrize("seed", range(6))
def test_edge_list_vs_oracle(ops, oracle, seed):
    import mtxgen
    weighted = seed % 2 == 1
    text = mtxgen.random_edge_list(5000 + seed, weighted, n=(20000 if seed == 5 else None), edges=(300000 if seed == 5 else None))
    dtext = torch.frombuffer(bytearray(text.encode() or b" "), dtype=torch.uint8).cuda()
    for vnp, vt in ((np.float32, torch.float32), (np.float64, torch.float64)) if weighted else ((None, None),):
        for dedup in (False, True):
            for no_self in (False, True):
                for undirected in (False, True):
                    for square in (False, True):
                        for inp, it in ((np.int32, torch.int32),) if seed else ((np.int32, torch.int32), (np.int64, torch.int64)):
                            want = oracle.edge_list_parse(text.encode(), weighted, dedup, no_self, undirected, square, inp, vnp)
                            got = ops.edge_list_parse(dtext, weighted, dedup, no_self, undirected, square, it, vt)
                            assert (got[0], got[1]) == (want[0], want[1])
                            assert np.array_equal(host(got[2]), want[2]) and np.array_equal(host(got[3]), want[3])
                            if weighted:
                                assert np.array_equal(host(got[4]), want[4])


def test_mtx_parse_errors_and_edges(ops):
    from sparsebase_amd import capi
    def parse(body, L, fields=3, vt=torch.float64):
        text = torch.frombuffer(bytearray(body), dtype=torch.uint8).cuda()
        return ops.mtx_parse_coordinate(text, 9, 9, L, fields, 0, True, False, torch.int32, vt)
    r, c, v = parse(b"  1 2 0.5\n\n\t3   4\t-1e400 \r\n5 6 4.9e-324", 3)
    assert host(r).tolist() == [0, 2, 4] and host(c).tolist() == [1, 3, 5]
    assert host(v).tolist() == [0.5, -np.inf, 5e-324]
    for bad in (b"1 2 0.5\n3 4", b"1 2 0.5\n3 x 1", b"1 2 nan", b"1 2 0x10", b"1 2 1.5.2", b"0 2 1"):
        with pytest.raises(capi.SbxError):
            parse(bad, 2 if bad.count(b"\n") else 1)
    r, c, v = parse(b"1 1 1.2345678901234567891\n2 2 9007199254740993.00000000000000000001", 2)   # up to 38 significant digits are exact
    assert host(v).tolist() == [1.2345678901234567891, 9007199254740994.0]
    with pytest.raises(capi.SbxError):   # more than 38 significant digits with a non-zero tail
        parse(b"1 1 1.23456789012345678901234567890123456789012", 1)
    r, c, v = parse(b"1 1 12345678901234567890000000000000000000000000000000", 1)   # zeros beyond 38 digits are exact
    assert host(v).tolist() == [1.234567890123456789e49]
    with pytest.raises(capi.SbxError):   # a '.' in an integer field
        parse(b"1 1 2.5", 1, 3, torch.int32)


# ----------------------------------------------------------------------------- features
def test_kat_features(ops, kat):
    k = kat["feature_7x7"]  # feature/bandwidth_tests.cc:33-48, feature/profile_tests.cc:33-48
    rp, col = dev(a32(k["row_ptr"])), dev(a32(k["col"]))
    assert ops.csr_bandwidth(rp, col) == k["bandwidth"] and ops.csr_profile(rp, col) == k["profile"]


@pytest.mark.parametrize("seed", range(4))
def test_features_vs_oracle(ops, oracle, seed):
    g = np.random.default_rng(950 + seed)
    if seed == 0:
        rp, col = synth.rmat_symmetric(15, 8, seed=3)           # hubs: rows far longer than a workgroup's share
    elif seed == 1:
        rp, col = synth.banded_symmetric(50000, 37, per_row=6, seed=2)
    else:
        n = int(g.integers(1, 5000))
        key = np.unique(g.integers(0, n * n, int(g.integers(0, 60000))))
        rp, col, _ = oracle.coo_to_csr(n, (key // n).astype(np.int32), (key % n).astype(np.int32))
        if seed == 3 and len(col) > 2:                           # unsorted rows
            col = col[::-1].copy()
            rp = (len(col) - rp[::-1]).astype(np.int32)
    for dt in (np.int32, np.int64):
        r, c = rp.astype(dt), col.astype(dt)
        assert ops.csr_bandwidth(dev(r), dev(c)) == oracle.csr_bandwidth(r, c)
        assert ops.csr_profile(dev(r), dev(c)) == oracle.csr_profile(r, c)
        assert np.array_equal(host(ops.csr_degrees(dev(r))), oracle.csr_degrees(r))
        if len(c):
            for tdt, ndt in ((torch.float32, np.float32), (torch.float64, np.float64)):
                assert np.array_equal(host(ops.csr_degree_distribution(dev(r), len(c), tdt)),
                                      oracle.csr_degree_distribution(r, len(c), ndt))
    z = np.zeros(0, np.int32)
    assert ops.csr_bandwidth(dev(np.zeros(5, np.int32)), dev(z)) == 0 and ops.csr_profile(dev(np.zeros(5, np.int32)), dev(z)) == 0


# ----------------------------------------------------------------------------- reference KATs
def test_kat_conversions(ops, kat):
    for name in ("converter_12x9", "format_4x4"):
        k = kat[name]
        rp, col, val = ops.coo_to_csr(k["n"], k["m"], dev(a32(k["coo_row"])), dev(a32(k["coo_col"])),
                                      dev(a32(k["coo_vals"])))
        assert host(rp).tolist() == k["csr_row_ptr"]
        assert host(col).tolist() == k["csr_col"] and host(val).tolist() == k["csr_vals"]
        ro, co, vo = ops.csr_to_coo(k["n"], k["m"], dev(a32(k["csr_row_ptr"])), dev(a32(k["csr_col"])),
                                    dev(a32(k["csr_vals"])))
        assert host(ro).tolist() == k["coo_row"]
        assert host(co).tolist() == k["coo_col"] and host(vo).tolist() == k["coo_vals"]
        # move conversions hand col/vals over untouched (converter_order_two_tests.cc:216-240)
        c_in = dev(a32(k["coo_col"]))
        rp2, c2, _ = ops.coo_to_csr(k["n"], k["m"], dev(a32(k["coo_row"])), c_in, None, move=True)
        assert host(rp2).tolist() == k["csr_row_ptr"] and c2.data_ptr() == c_in.data_ptr()


def test_kat_ctor_sorts(ops, kat):
    k = kat["format_4x4"]
    r, c, v = dev(a32(k["coo_row_shuffled"])), dev(a32(k["coo_col_shuffled"])), dev(a32(k["coo_vals_shuffled"]))
    assert not ops.coo_is_sorted(r, c)
    ops.coo_sort_(4, 4, r, c, v)
    assert (host(r).tolist(), host(c).tolist(), host(v).tolist()) == (k["coo_row"], k["coo_col"], k["coo_vals"])
    assert ops.coo_is_sorted(r, c)
    r, c = dev(a32(k["coo_row_shuffled"])), dev(a32(k["coo_col_shuffled"]))
    ops.coo_sort_(4, 4, r, c, None)  # ValueType = void (coo_tests.cc:104-114)
    assert (host(r).tolist(), host(c).tolist()) == (k["coo_row"], k["coo_col"])

    rp = dev(a32(k["csr_row_ptr"]))
    c, v = dev(a32(k["csr_col_shuffled"])), dev(a32(k["csr_vals_shuffled"]))
    assert not ops.csr_rows_sorted(rp, c)
    ops.csr_sort_rows_(4, 4, rp, c, v)
    assert (host(c).tolist(), host(v).tolist()) == (k["csr_col"], k["csr_vals"])
    assert ops.csr_rows_sorted(rp, c)
    c = dev(a32(k["csr_col_shuffled"]))
    ops.csr_sort_rows_(4, 4, rp, c, None)
    assert host(c).tolist() == k["csr_col"]


def test_kat_permute(ops, kat):
    k = kat["functionality_3x3"]
    rp, col, val = dev(a32(k["row_ptr"])), dev(a32(k["cols"])), dev(a32(k["vals"]))
    r, c = dev(a32(k["r_reorder_vector"])), dev(a32(k["c_reorder_vector"]))
    out = ops.permute_csr(3, 3, rp, col, val, r, None)
    assert [host(x).tolist() for x in out] == [k["r_row_ptr"], k["r_cols"], k["r_vals"]]
    out = ops.permute_csr(3, 3, rp, col, val, None, c)
    assert [host(x).tolist() for x in out] == [k["c_row_ptr"], k["c_cols"], k["c_vals"]]
    out = ops.permute_csr(3, 3, rp, col, val, r, c)
    assert [host(x).tolist() for x in out] == [k["rc_row_ptr"], k["rc_cols"], k["rc_vals"]]
    back = ops.permute_csr(3, 3, *out, ops.inverse_permutation(r), ops.inverse_permutation(c))
    assert [host(x).tolist() for x in back] == [k["row_ptr"], k["cols"], k["vals"]]
    assert host(ops.inverse_permutation(dev(a32(k["perm_array"])))).tolist() == k["inverse_perm_array"]
    arr = ops.permute_array(dev(a32(k["inverse_perm_array"])), dev(np.array(k["original_array"], np.float32)))
    assert np.allclose(host(arr), np.array(k["reordered_array"], np.float32))
    assert host(ops.degree_reorder(rp, True)).tolist() == k["degree_asc"]
    assert host(ops.degree_reorder(rp, False)).tolist() == k["degree_desc"]


# ----------------------------------------------------------------------------- golden fixtures
@pytest.fixture(scope="module")
def small(golden_dir):
    z = np.load(os.path.join(golden_dir, "small_cases.npz"))
    with open(os.path.join(golden_dir, "small_cases.json")) as f:
        return z, json.load(f)


def test_golden_degree_and_permute(ops, small):
    z, meta = small
    for name in meta:
        rp, col = z[f"{name}/row_ptr"], z[f"{name}/col"]
        n = len(rp) - 1
        assert np.array_equal(host(ops.degree_reorder(dev(rp), True)), z[f"{name}/degree_asc"]), name
        assert np.array_equal(host(ops.degree_reorder(dev(rp), False)), z[f"{name}/degree_desc"]), name
        if n == 0 or f"{name}/perm_order" not in z:
            continue
        order, val = z[f"{name}/perm_order"], z[f"{name}/perm_val"]
        for tag, ro, co in (("rc", order, order), ("r", order, None), ("rcm", z[f"{name}/rcm"], z[f"{name}/rcm"])):
            got = ops.permute_csr(n, n, dev(rp), dev(col), dev(val), dev(ro), dev(co))
            same(got, [z[f"{name}/permute_{tag}/row_ptr"], z[f"{name}/permute_{tag}/col"],
                       z[f"{name}/permute_{tag}/val"]])


def test_golden_rect_convert_and_sort(ops, small):
    z, _ = small
    for k in range(4):
        name = f"rect{k}"
        n, m = (int(x) for x in z[f"{name}/dims"])
        r, c, v = dev(z[f"{name}/coo_row"]), dev(z[f"{name}/coo_col"]), dev(z[f"{name}/coo_val"])
        ops.coo_sort_(n, m, r, c, v)
        same((r, c, v), (z[f"{name}/sorted_row"], z[f"{name}/sorted_col"], z[f"{name}/sorted_val"]))
        csr = ops.coo_to_csr(n, m, r, c, v)
        same(csr, (z[f"{name}/csr_row_ptr"], z[f"{name}/csr_col"], z[f"{name}/csr_val"]))
        same(ops.csr_to_coo(n, m, *csr), (z[f"{name}/sorted_row"], z[f"{name}/sorted_col"], z[f"{name}/sorted_val"]))
        uc, uv = dev(z[f"{name}/unsorted_col"]), dev(z[f"{name}/unsorted_val"])
        ops.csr_sort_rows_(n, m, csr[0], uc, uv)
        same((uc, uv), (z[f"{name}/resorted_col"], z[f"{name}/resorted_val"]))


def test_golden_ash958(ops, golden_dir):
    z = np.load(os.path.join(golden_dir, "ash958.npz"))
    n, m, nnz = (int(x) for x in z["dims"])
    r, c = dev(z["file_row"]), dev(z["file_col"])
    ops.coo_sort_(n, m, r, c, None)  # the .mtx is column-major: exercises the COO-ctor sort
    rp, cc, _ = ops.coo_to_csr(n, m, r, c, None)
    same((rp, cc), (z["row_ptr"], z["col"]))
    dasc = ops.degree_reorder(rp, True)
    assert np.array_equal(host(dasc), z["degree_asc"])
    assert np.array_equal(host(ops.degree_reorder(rp, False)), z["degree_desc"])
    prp, pcol, _ = ops.permute_csr(n, m, rp, cc, None, dasc, None)
    same((prp, pcol), (z["rowwise_row_ptr"], z["rowwise_col"]))


# ----------------------------------------------------------------------------- oracle, seeded inputs
@pytest.mark.parametrize("seed", range(8))
def test_convert_vs_oracle(ops, oracle, seed):
    g = np.random.default_rng(seed)
    n, m = int(g.integers(1, 5000)), int(g.integers(1, 5000))
    nnz = int(g.integers(0, 200000))
    rp, col = synth.random_rect_csr(n, m, nnz, seed, sort_rows=(seed % 2 == 0), dup_frac=0.05 if seed % 3 == 0 else 0)
    for val in (None, g.integers(-99, 99, len(col)).astype(np.int32), g.random(len(col)).astype(np.float32),
                g.random(len(col))):
        want_c, want_v = oracle.csr_sort_rows(rp, col, val)
        c, v = dev(col), dev(val)
        assert ops.csr_rows_sorted(dev(rp), c) == oracle.csr_rows_sorted(rp, col)
        ops.csr_sort_rows_(n, m, dev(rp), c, v)
        same((c, v), (want_c, want_v))
        coo = ops.csr_to_coo(n, m, dev(rp), c, v)
        want_coo = oracle.csr_to_coo(rp, want_c, want_v)
        same(coo, want_coo)
        same(ops.coo_to_csr(n, m, *coo), oracle.coo_to_csr(n, *want_coo))
        same(ops.coo_to_csr(n, m, *coo, rows_sorted=True), oracle.coo_to_csr(n, *want_coo))
        rp_m, _, _ = ops.coo_to_csr(n, m, coo[0], coo[1], coo[2], move=True)
        assert np.array_equal(host(rp_m), rp)
        # shuffled COO -> ctor sort (distinct coordinates: ties are unspecified in the reference)
        key = np.unique(want_coo[0].astype(np.int64) * m + want_coo[1])
        p = g.permutation(len(key))
        rr, cc = (key // m).astype(np.int32)[p], (key % m).astype(np.int32)[p]
        vv = None if val is None else val[: len(key)].copy()
        r_d, c_d, v_d = dev(rr), dev(cc), dev(vv)
        ops.coo_sort_(n, m, r_d, c_d, v_d)
        same((r_d, c_d, v_d), oracle.coo_sort(rr, cc, vv))


@pytest.mark.parametrize("n,m,nnz", [(1 << 17, 1 << 17, 600000), (1 << 20, 1 << 12, 900000), (5000, 1 << 20, 400000),
                                     (1 << 22, 1 << 9, 700000)])
def test_coo_sort_hybrid_groups(ops, oracle, n, m, nnz):
    """The hybrid COO constructor sort (sbx_convert.hip: two or three digit passes group the records by the row's
    leading bits, the permute's LDS sort stage orders row-low-bits | column inside every group): DUPLICATE coordinates
    keep their input order (the oracle's stable rule), hub rows put single groups beyond the LDS classes (long-group
    path), empty groups in between, every value width, and dimensions that give 0 .. 6 low row bits per group."""
    g = np.random.default_rng(n + m)
    row = g.integers(0, n, nnz).astype(np.int32)
    row[: nnz // 5] = g.integers(0, 3, nnz // 5)                 # three hub rows: one group far above 8192 records
    row[nnz // 5: nnz // 4] = n - 1                                # ... and the last row of the last group
    col = g.integers(0, m, nnz).astype(np.int32)
    col[:5000] = col[5000:10000]                                   # duplicate (row, col) pairs with different values
    row[:5000] = row[5000:10000]
    for val in (None, g.integers(-99, 99, nnz).astype(np.int32), g.random(nnz)):
        r_d, c_d, v_d = dev(row.copy()), dev(col.copy()), dev(None if val is None else val.copy())
        ops.coo_sort_(n, m, r_d, c_d, v_d)
        same((r_d, c_d, v_d), oracle.coo_sort(row, col, val))


@pytest.mark.parametrize("n,m", [(1, 7), (7, 1), (3, 200), (300, 70000), (70000, 300), (70000, 70000), (2, 2)])
def test_sorts_that_read_and_write_the_callers_arrays(ops, oracle, n, m):
    # COO constructor sort and COO / CSR -> CSC let the first digit pass of their radix sort load the source arrays and
    # the last one store the destination arrays (sbx_radix_sort_io); the dimensions choose 1 ... 6 digit passes, the
    # one-pass cases keep the pack / unpack form
    g = np.random.default_rng(n * 131 + m)
    key = np.unique(g.integers(0, n * m, min(n * m, 150000)))
    key = key[g.permutation(len(key))]
    row, col = (key // m).astype(np.int32), (key % m).astype(np.int32)
    nnz = len(key)
    for val in (None, g.integers(-99, 99, nnz).astype(np.int32), g.random(nnz).astype(np.float32), g.random(nnz)):
        r_d, c_d, v_d = dev(row.copy()), dev(col.copy()), dev(None if val is None else val.copy())
        ops.coo_sort_(n, m, r_d, c_d, v_d)
        want = oracle.coo_sort(row, col, val)
        same((r_d, c_d, v_d), want)
        same(ops.coo_to_csc(n, m, dev(row), dev(col), dev(val)), oracle.coo_to_csc(n, m, row, col, val))
        rp, cc, vv = oracle.coo_to_csr(n, *want)
        same(ops.csr_to_csc(n, m, dev(rp), dev(cc), dev(vv)), oracle.csr_to_csc(m, rp, cc, vv))


def test_coo_to_csr_unsorted_rows(ops, oracle):
    # ignore_sort=true input: row_ptr is still exclusive_scan(histogram(row))
    g = np.random.default_rng(5)
    n, nnz = 777, 20000
    row = g.integers(0, n, nnz).astype(np.int32)
    col = g.integers(0, n, nnz).astype(np.int32)
    val = g.random(nnz).astype(np.float32)
    same(ops.coo_to_csr(n, n, dev(row), dev(col), dev(val)), oracle.coo_to_csr(n, row, col, val))


def test_convert_edge_cases(ops, oracle):
    # empty matrix, empty leading/trailing rows, one giant gap of empty rows, single nnz
    z = np.zeros(0, np.int32)
    same(ops.coo_to_csr(5, 5, dev(z), dev(z), None), oracle.coo_to_csr(5, z, z, None))
    row = np.array([70000, 70000, 70001, 299999], np.int32)
    col = np.array([1, 5, 2, 0], np.int32)
    val = np.array([1.5, 2.5, 3.5, 4.5], np.float64)
    n = 300007
    want = oracle.coo_to_csr(n, row, col, val)
    got = ops.coo_to_csr(n, n, dev(row), dev(col), dev(val))
    same(got, want)
    same(ops.csr_to_coo(n, n, *got), (row, col, val))
    one = np.array([3], np.int32)
    same(ops.coo_to_csr(9, 9, dev(one), dev(one), None), oracle.coo_to_csr(9, one, one, None))


@pytest.mark.parametrize("seed", range(6))
def test_permute_vs_oracle(ops, oracle, seed):
    g = np.random.default_rng(40 + seed)
    if seed < 3:
        rp, col = synth.rmat_symmetric(12 + seed, 8, seed=seed)  # power-law: short, medium and long rows
        n = m = len(rp) - 1
    else:
        n, m = int(g.integers(2, 3000)), int(g.integers(2, 3000))
        rp, col = synth.random_rect_csr(n, m, int(g.integers(1, 100000)), seed, dup_frac=0.1 if seed == 4 else 0)
    ro = synth.random_permutation(n, seed)
    co = synth.random_permutation(m, seed + 1)
    for val in (None, g.integers(-5, 5, len(col)).astype(np.int32), g.random(len(col)).astype(np.float32),
                g.random(len(col))):
        for r, c in ((ro, co), (ro, None), (None, co), (None, None)):
            got = ops.permute_csr(n, m, dev(rp), dev(col), dev(val), dev(r), dev(c))
            same(got, oracle.permute_csr(rp, col, val, r, c))


def test_permute_long_rows_and_duplicates(ops, oracle):
    # a few very long rows (radix path), medium rows (bitonic path), duplicates with values
    g = np.random.default_rng(3)
    n, m = 64, 50000
    # one row in every sort class: tile (<= 1024), block rows of 2048 / 4096 / 8192 slots, global radix
    lens = np.array([0, 1, 2, 40, 33, 1024, 1025, 2048, 2049, 3000, 4096, 4097, 5000, 8192, 8193, 9000, 16384, 16385,
                     20000] + [int(x) for x in g.integers(0, 300, n - 19)])
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    col = np.concatenate([np.sort(g.integers(0, m, l)) for l in lens]).astype(np.int32)  # sorted, with duplicates
    val = g.integers(-3, 3, len(col)).astype(np.int32)
    ro, co = synth.random_permutation(n, 1), synth.random_permutation(m, 2)
    for v in (val, val.astype(np.float32), None, val.astype(np.float64), val.astype(np.int64)):
        same(ops.permute_csr(n, m, dev(rp), dev(col), dev(v), dev(ro), dev(co)), oracle.permute_csr(rp, col, v, ro, co))
        # identity maps: nothing is unsorted, so duplicates must keep their input order
        same(ops.permute_csr(n, m, dev(rp), dev(col), dev(v), None, None), oracle.permute_csr(rp, col, v, None, None))


@pytest.mark.parametrize("idt", [np.int32, np.int64])
def test_permute_quad_class_boundaries(ops, oracle, idt):
    """k_rows_quad (sbx_rowsort.h): every class boundary and every remainder modulo the quad (a row's last, partial quad is
    cut by the row's buffer descriptor, not by a predicate), the last row of the arrays ending in a partial quad (the
    16-byte accesses of its last lane reach past the allocation's logical end), a bucket of more than eight equal-ish
    columns (the ranking loop), more than 96 (the radix list), every value width, distinct and duplicate columns, a random
    and a clustered (run-structured, RCM-like) column map.  int64: the same rows through the native 64-bit instances of
    round 3's class kernels (no narrowed copies)."""
    g = np.random.default_rng(41)
    m = 1 << 18
    lens = [129, 130, 131, 132, 255, 256, 257, 258, 259, 260, 511, 512, 513, 1023, 1024, 1025, 1026, 1027, 2047, 2048, 2049,
            4095, 4096, 4097, 4098, 8189, 8190, 8191, 8192, 700, 3001, 6002, 1, 0, 5, 300, 1301, 2302, 8003]
    n = len(lens)
    rows = [np.sort(g.choice(m, l, replace=False)) for l in lens]
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(idt)
    col = np.concatenate(rows).astype(idt)
    val = g.integers(-1000, 1000, len(col)).astype(np.int32)
    ro = synth.random_permutation(n, 3, idt)
    ro[-1], ro[int(np.argmax(ro == n - 1))] = n - 1, ro[-1]   # the last old row (8003 = 3 mod 4) stays the last new row
    rnd = synth.random_permutation(m, 4, idt)
    runs = np.arange(m, dtype=np.int64)                        # runs of 977 consecutive labels, the runs shuffled
    blocks = g.permutation((m + 976) // 977)
    clustered = np.concatenate([np.arange(b * 977, min(m, (b + 1) * 977)) for b in blocks])[:m]
    inv = np.empty(m, np.int64)
    inv[clustered] = runs
    for co in (rnd, inv.astype(idt)):
        for v in (val.astype(np.float32), None, val.astype(np.float64)):
            same(ops.permute_csr(n, m, dev(rp), dev(col), dev(v), dev(ro), dev(co)), oracle.permute_csr(rp, col, v, ro, co))
    # equal columns: 9, 40 and 200 copies of one column inside rows of every big class (ranking loop / radix list);
    # duplicates end in value order (csr.cc:143-156)
    rows2 = []
    for l, rep in ((300, 9), (900, 40), (3000, 200), (7000, 9), (200, 97), (5000, 1200)):
        rows2.append(np.sort(np.concatenate([g.choice(m, l - rep, replace=False), np.full(rep, int(g.integers(0, m)))])))
    rp2 = np.concatenate([[0], np.cumsum([len(r) for r in rows2])]).astype(idt)
    col2 = np.concatenate(rows2).astype(idt)
    ro2 = synth.random_permutation(len(rows2), 5, idt)
    for v2 in (g.integers(-9, 9, len(col2)).astype(np.int32), g.random(len(col2)).astype(np.float32), None):
        same(ops.permute_csr(len(rows2), m, dev(rp2), dev(col2), dev(v2), dev(ro2), dev(rnd)),
             oracle.permute_csr(rp2, col2, v2, ro2, rnd))


@pytest.mark.parametrize("idt", [np.int32, np.int64])
def test_permute_long_row_segments(ops, oracle, idt):
    """Rows above the one-workgroup capacity under a column map (sbx_permute.hip, k_long_seg_*): split into column-range
    segments that are sorted in LDS.  Covers: rows of 8 K .. 300 K entries (8 .. 256 segments; chunks of the compact buffer
    that straddle two rows), duplicate columns with values, a row whose relabelled columns all fall into one segment (too
    full: that row takes the global radix sort), a row the map leaves in order (the reference does not touch it: stable
    path) next to rows it scrambles, and 8-byte values (segments of at most 4096 entries).  int64: 64-bit index arrays
    through the same kernels, natively."""
    g = np.random.default_rng(11)
    n, m = 40, 1 << 20
    lens = [8193, 12000, 4097, 300000, 8200, 70000, 16385, 9000, 0, 5, 30000, 10000] + [int(x) for x in g.integers(0, 200, n - 12)]
    cols = []
    for i, l in enumerate(lens):
        if i == 7:      # tight cluster: one segment gets everything
            c = 5000 + g.choice(20000, l, replace=False)
        elif i == 11:   # columns below 2^19: the map below keeps them in order
            c = g.choice(1 << 19, l, replace=False)
        elif i == 10:   # duplicates
            c = g.integers(1 << 19, m, l)
        else:
            c = (1 << 19) + g.choice(1 << 19, l, replace=False) if l else np.zeros(0, np.int64)
        cols.append(np.sort(c))
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(idt)
    col = np.concatenate(cols).astype(idt)
    co = np.arange(m, dtype=np.int64)
    co[1 << 19:] = (1 << 19) + g.permutation(1 << 19)   # lower half stays put (monotone), upper half is scrambled
    co = co.astype(idt)
    ro = synth.random_permutation(n, 4, idt)
    val = g.integers(-3, 3, len(col)).astype(np.int32)
    for v in (val.astype(np.float32), None, val.astype(np.float64)):
        same(ops.permute_csr(n, m, dev(rp), dev(col), dev(v), dev(ro), dev(co)), oracle.permute_csr(rp, col, v, ro, co))
        same(ops.permute_csr(n, m, dev(rp), dev(col), dev(v), None, dev(co)), oracle.permute_csr(rp, col, v, None, co))
    # one column repeated 300 / 3000 times inside long rows: the segment that holds the run cannot be bucket-ranked
    # (one bucket gets them all) and goes through the radix kernel for clustered segments; equal columns end in value order
    for rep in (300, 3000):
        rows2 = [np.sort(np.concatenate([g.integers(0, m, l - rep), np.full(rep, int(g.integers(0, m)))])) for l in (20000, 9000, 5000)]
        rp2 = np.concatenate([[0], np.cumsum([len(r) for r in rows2])]).astype(idt)
        col2 = np.concatenate(rows2).astype(idt)
        ro2 = synth.random_permutation(3, rep, idt)
        for v2 in (g.integers(-9, 9, len(col2)).astype(np.int32), g.random(len(col2))):
            same(ops.permute_csr(3, m, dev(rp2), dev(col2), dev(v2), dev(ro2), dev(co)), oracle.permute_csr(rp2, col2, v2, ro2, co))
    # a monotone map over everything: every long row stays in order, duplicates keep their input order
    mono = np.arange(m, dtype=idt)
    same(ops.permute_csr(n, m, dev(rp), dev(col), dev(val), dev(ro), dev(mono)), oracle.permute_csr(rp, col, val, ro, mono))
    # shards (sbx_permute_csr_rows) go through the same stage
    whole = oracle.permute_csr(rp, col, val, ro, co)
    for r0, r1 in ((0, 17), (17, 40)):
        prp, pcol, pval = ops.permute_csr_rows(n, m, dev(rp), dev(col), dev(val), dev(ro), dev(co), r0, r1)
        a, b = int(whole[0][r0]), int(whole[0][r1])
        assert np.array_equal(host(prp), whole[0][r0:r1 + 1] - a)
        assert np.array_equal(host(pcol)[:b - a], whole[1][a:b]) and np.array_equal(host(pval)[:b - a], whole[2][a:b])


def _clustered_case(seed=5):
    """Rows whose relabelled columns sit in one tight block plus a far outlier: the bucket-rank sort of
    sbx_permute.hip meets a bucket larger than its limit there and must take the radix path."""
    g = np.random.default_rng(seed)
    m = 1 << 20
    lens = [300, 5, 700, 1024, 64, 1500, 3000, 6000, 2, 130] + [int(x) for x in g.integers(0, 260, 120)]
    cols = []
    for i, l in enumerate(lens):
        if l == 0:
            cols.append(np.zeros(0, np.int64))
            continue
        base = int(g.integers(0, m // 2))
        block = base + np.arange(max(l - 1, 1))[: max(l - 1, 0)]
        far = np.array([m - 1 - i]) if l > 1 else np.array([base])
        cols.append(np.sort(np.concatenate([block, far]))[:l])
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    col = np.concatenate(cols).astype(np.int32)
    # column map that keeps blocks tight but scrambles inside 64-column groups
    co = np.arange(m, dtype=np.int64)
    co = (co & ~63) | g.permutation(64)[co & 63]
    return len(lens), m, rp, col, co.astype(np.int32)


def test_permute_clustered_rows(ops, oracle):
    n, m, rp, col, co = _clustered_case()
    g = np.random.default_rng(9)
    ro = synth.random_permutation(n, 3)
    for val in (None, g.integers(-5, 5, len(col)).astype(np.int32), g.random(len(col))):
        same(ops.permute_csr(n, m, dev(rp), dev(col), dev(val), dev(ro), dev(co)),
             oracle.permute_csr(rp, col, val, ro, co))


def test_permute_forced_radix_path():
    """SBX_PERMUTE_FORCE_RADIX=1 sends every tile and block row through the LDS radix sort (the
    distribution-independent path behind the bucket-rank sort); read once per process, hence the child."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np, torch; sys.path[:0] = [%r, %r]\n"
        "from orc import Oracle; from sparsebase_amd import ops, synth\n"
        "orc = Oracle(); d = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()\n"
        "g = np.random.default_rng(1)\n"
        "lens = np.array([0, 1, 40, 1024, 1025, 2048, 3000, 4097, 8192, 9000] + [int(x) for x in g.integers(0, 300, 200)])\n"
        "n, m = len(lens), 70000\n"
        "rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)\n"
        "col = np.concatenate([np.sort(g.integers(0, m, l)) for l in lens]).astype(np.int32)\n"
        "ro, co = synth.random_permutation(n, 1), synth.random_permutation(m, 2)\n"
        "for v in (None, g.integers(-3, 3, len(col)).astype(np.int32), g.random(len(col))):\n"
        "    got = ops.permute_csr(n, m, d(rp), d(col), d(v), d(ro), d(co)); want = orc.permute_csr(rp, col, v, ro, co)\n"
        "    for a, b in zip(got, want):\n"
        "        assert (a is None and b is None) or np.array_equal(a.cpu().numpy(), b)\n"
        "print('forced-radix ok')\n" % (root, os.path.join(root, "tests")))
    env = dict(os.environ, SBX_PERMUTE_FORCE_RADIX="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "forced-radix ok" in r.stdout, r.stdout + r.stderr
    # SBX_PERMUTE_NO_TILE2=1: the relabelling permutes keep the equal-width tile kernel (what wider ids and arrays beyond
    # 4 GB take in production) on inputs where the key-distribution kernel would run
    env = dict(os.environ, SBX_PERMUTE_NO_TILE2="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "forced-radix ok" in r.stdout, r.stdout + r.stderr


def test_permute_rowwise_copy_path(ops, oracle):
    """Row-wise permutes run as a segmented copy; unsorted input rows must fall back to the sorting
    pipeline, and tiles that span thousands of empty rows take the per-position search."""
    g = np.random.default_rng(11)
    # (a) sorted rows, long runs of empty rows between the non-empty ones, shards
    n, m = 600000, 4096  # ~2000 empty rows between non-empty ones: tiles take the per-position search
    lens = np.zeros(n, dtype=np.int64)
    idx = g.choice(n, 300, replace=False)
    lens[idx] = g.integers(1, 200, 300)
    lens[idx[:3]] = (3000, 20000, 2048)
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    col = np.concatenate([np.sort(g.choice(max(m, l), l, replace=False) % m) for l in lens[np.sort(idx)]]).astype(np.int32)
    val = g.random(len(col)).astype(np.float32)
    ro = synth.random_permutation(n, 5)
    want = oracle.permute_csr(rp, col, val, ro, None)
    same(ops.permute_csr(n, m, dev(rp), dev(col), dev(val), dev(ro), None), want)
    same(ops.permute_csr(n, m, dev(rp), dev(col), None, dev(ro), None), oracle.permute_csr(rp, col, None, ro, None))
    co = synth.random_permutation(m, 8)  # the sorting pipeline over the same mostly-empty row ranges
    same(ops.permute_csr(n, m, dev(rp), dev(col), dev(val), dev(ro), dev(co)), oracle.permute_csr(rp, col, val, ro, co))
    same(ops.permute_csr(n, m, dev(rp), dev(col), dev(val), None, dev(co)), oracle.permute_csr(rp, col, val, None, co))
    for a, b in ((0, 7), (7, n // 2), (n // 2, n)):
        srp, scol, sval = ops.permute_csr_rows(n, m, dev(rp), dev(col), dev(val), dev(ro), None, a, b)
        lo, hi = want[0][a], want[0][b]
        assert np.array_equal(host(srp), want[0][a:b + 1] - lo)
        assert np.array_equal(host(scol), want[1][lo:hi]) and np.array_equal(host(sval), want[2][lo:hi])
    # (b) one unsorted row far into the matrix (also one exactly at a tile boundary): full sort semantics
    rp2, col2 = synth.rmat_symmetric(13, 8, seed=2)
    n2 = len(rp2) - 1
    val2 = g.integers(-4, 4, len(col2)).astype(np.int32)
    ro2 = synth.random_permutation(n2, 6)
    for victim in (int(np.argmax(np.diff(rp2))), int(np.searchsorted(rp2, 2048, side="right") - 1)):
        c = col2.copy()
        a, b = rp2[victim], rp2[victim + 1]
        if b - a >= 2:
            c[a], c[b - 1] = c[b - 1], c[a]
        same(ops.permute_csr(n2, n2, dev(rp2), dev(c), dev(val2), dev(ro2), None),
             oracle.permute_csr(rp2, c, val2, ro2, None))
        same(ops.permute_csr(n2, n2, dev(rp2), dev(c), dev(val2), None, None),
             oracle.permute_csr(rp2, c, val2, None, None))


def test_permute_row_shards(ops, oracle):
    rp, col = synth.rmat_symmetric(13, 8, seed=9)
    n = len(rp) - 1
    val = (np.arange(len(col)) % 97).astype(np.float32)
    order = synth.random_permutation(n, 4)
    want_rp, want_col, want_val = oracle.permute_csr(rp, col, val, order, order)
    cuts = [0, n // 3, n // 3, n - 5, n]
    for a, b in zip(cuts[:-1], cuts[1:]):
        srp, scol, sval = ops.permute_csr_rows(n, n, dev(rp), dev(col), dev(val), dev(order), dev(order), a, b)
        lo, hi = want_rp[a], want_rp[b]
        assert np.array_equal(host(srp), want_rp[a:b + 1] - lo)
        assert np.array_equal(host(scol), want_col[lo:hi]) and np.array_equal(host(sval), want_val[lo:hi])


def test_convert_row_shards(ops, oracle):
    """The per-rank computation of the sharded conversions (sparsebase_amd/sharded.py) on one GPU."""
    from sparsebase_amd import sharded
    rp, col = synth.rmat_symmetric(13, 8, seed=11)
    n = len(rp) - 1
    val = (np.arange(len(col)) % 53).astype(np.float32)
    row, col2, val2 = oracle.csr_to_coo(rp, col, val)
    drow, dcol, dval, drp = dev(row), dev(col2), dev(val2), dev(rp)
    for lo, hi in sharded.row_ranges(n, 3) + [(5, 5), (0, n)]:
        a, b = int(rp[lo]), int(rp[hi])
        lrp, lc, lv = sharded.coo_to_csr_shard(n, drow, dcol, dval, lo, hi, a, b)
        assert np.array_equal(host(lrp), rp[lo:hi + 1] - rp[lo])
        assert np.array_equal(host(lc), col[a:b]) and np.array_equal(host(lv), val[a:b])
        r, c, v = sharded.csr_to_coo_shard(n, drp, dev(col), dev(val), lo, hi, a, b)
        assert np.array_equal(host(r), row[a:b]) and np.array_equal(host(c), col2[a:b]) and np.array_equal(host(v), val2[a:b])


@pytest.mark.parametrize("name", ["rmat16_ef8", "banded_64k_w16", "sym_128k"])
def test_golden_digests_degree_permute(ops, golden_dir, name):
    import hashlib

    def digest(*arrays):
        h = hashlib.sha256()
        for a in arrays:
            h.update(np.ascontiguousarray(host(a) if torch.is_tensor(a) else a).tobytes())
        return h.hexdigest()

    with open(os.path.join(golden_dir, "digests.json")) as f:
        d = json.load(f)[name]
    rp, col = getattr(synth, d["generator"])(**d["args"])
    n = len(rp) - 1
    assert digest(rp, col) == d["input"]
    assert digest(ops.degree_reorder(dev(rp), True)) == d["degree_asc"]
    assert digest(ops.degree_reorder(dev(rp), False)) == d["degree_desc"]
    val = (np.arange(len(col)) % 1021).astype(np.float32)
    order = synth.random_permutation(n, seed=77)
    assert digest(*ops.permute_csr(n, n, dev(rp), dev(col), dev(val), dev(order), None)) == d["permute_random_rowwise"]


def test_gray_row_keys(ops, oracle):
    for seed, (res, thr) in enumerate([(32, 10), (16, 20), (64, 2), (16, 0)]):
        rp, col = synth.rmat_symmetric(12, 8, seed=seed) if seed % 2 == 0 else synth.banded_symmetric(4096, 40, 9, seed)
        n = len(rp) - 1
        deg, key, counts = ops.gray_row_keys(n, dev(rp), dev(col), res, thr)
        wdeg, wkey, wcounts = oracle.gray_row_keys(rp, col, n, res, thr)
        assert np.array_equal(host(deg), wdeg)
        assert np.array_equal(host(key).view(np.uint64), wkey)
        assert list(counts) == wcounts.tolist()


def _gray_stable_model(deg, key, counts, bits, thr, grp):
    """numpy statement of sbx_gray_reorder's ordering (gray_reorder.cc:181-420 with every std::sort read as a stable sort):
    returns (inv, group) — the inverse permutation and, per row, the (class, section, signed key) tuple that decides
    its place up to ties."""
    n = len(deg)
    deg = deg.astype(np.int64)
    key = key.astype(np.uint64)
    mask = np.uint64((1 << bits) - 1) if bits < 64 else np.uint64(0xFFFFFFFFFFFFFFFF)
    sparse_banded = float(np.int32(counts[1])) / float(np.int32(counts[0])) > 0.3 if counts[0] else False
    dense_banded = float(np.int32(counts[3])) / float(np.int32(counts[2])) > 0.2 if counts[2] else False
    sparse = deg <= thr
    present = np.unique(deg[sparse & (deg > 0)])
    section = np.zeros(n, np.int64)
    if not sparse_banded and len(present):
        rank = np.searchsorted(present, deg)
        section = np.where(sparse & (deg > 0), 1 + rank // grp, 0)
    n_sections = 0 if sparse_banded else (len(present) + grp - 1) // grp
    cls = np.where(sparse, section, n_sections + 1)
    skey = np.zeros(n, np.uint64)
    if not dense_banded:
        skey = np.where(~sparse, key & mask, skey)
    if not sparse_banded:
        odd = sparse & (deg > 0) & ((section - 1) % 2 == 1)
        even = sparse & (deg > 0) & ~odd
        skey = np.where(even, key & mask, skey)
        skey = np.where(odd, ~key & mask, skey)
    dkey = np.where(sparse, deg, 0)
    order = np.lexsort((np.arange(n), dkey, skey, cls))
    inv = np.empty(n, np.int64)
    inv[order] = np.arange(n)
    return inv, np.stack([cls.astype(np.uint64), skey, np.where(sparse & (sparse_banded | (deg == 0)), dkey, 0).astype(np.uint64)], 1)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["rmat_32_10_4", "rmat_16_20_2", "rmat_64_3_1", "banded_32_10_4", "banded_wide_16_0_1",
                                  "rmat_32_neg", "rmat_32_huge_thr"])
def test_gray_reorder_device_ordering(ops, oracle, case):
    """sbx_gray_reorder (exact_ties = 0): the ordering stage on the device with stable ties — equals the numpy statement
    of that order row for row, is a permutation, and equals the reference-exact oracle wherever the reference's
    comparators decide a row's place (its (class, section, key) is unique); exact_ties = 1 is refused."""
    kind, res, thr, grp = {"rmat_32_10_4": ("rmat", 32, 10, 4), "rmat_16_20_2": ("rmat", 16, 20, 2),
                           "rmat_64_3_1": ("rmat", 64, 3, 1), "banded_32_10_4": ("banded", 32, 10, 4),
                           "banded_wide_16_0_1": ("wide", 16, 0, 1), "rmat_32_neg": ("rmat", 32, -1, 3),
                           "rmat_32_huge_thr": ("rmat", 32, 1 << 20, 5)}[case]
    if kind == "rmat":
        rp, col = synth.rmat_symmetric(13, 8, seed=5)
    elif kind == "banded":
        rp, col = synth.banded_symmetric(8192, 40, 9, 3)
    else:
        rp, col = synth.banded_symmetric(8192, 2048, 9, 4)
    n = len(rp) - 1
    inv = host(ops.gray_reorder(n, dev(rp), dev(col), res, thr, grp)).astype(np.int64)
    assert np.array_equal(np.sort(inv), np.arange(n))
    deg, key, counts = oracle.gray_row_keys(rp, col, n, res, thr)
    model, group = _gray_stable_model(deg, key, counts, min(res, n), thr, grp)
    assert np.array_equal(inv, model)
    want = oracle.gray_reorder(rp, col, n, res, thr, grp).astype(np.int64)
    _, first, cnt = np.unique(group, axis=0, return_index=True, return_counts=True)
    decided = first[cnt == 1]  # rows whose place the reference's comparators decide
    assert len(decided) > 0
    assert np.array_equal(inv[decided], want[decided])
    with pytest.raises(Exception):
        ops.gray_reorder(n, dev(rp), dev(col), res, thr, grp, exact_ties=True)


def test_gray_row_keys_long_rows_and_odd_widths(ops, oracle):
    """Rows that span several 2048-nonzero tiles in both classes (threshold 0: OR across tiles; counted: per-block
    counts across tiles), column-block widths that are not powers of two, resolution 64, single-entry and empty rows."""
    g = np.random.default_rng(23)
    n, m = 700, 6400
    lens = g.integers(0, 40, n)
    lens[[3, 4, 5, 300, 699]] = (5000, 6399, 2048, 6400, 4097)
    lens[[0, 1, 2, 698]] = (0, 1, 0, 1)
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    col = np.concatenate([np.sort(g.choice(m, l, replace=False)) for l in lens]).astype(np.int32)
    for res, thr in ((64, 10), (32, 10), (64, 20000), (32, 4096), (16, 0), (50, 10)):
        deg, key, counts = ops.gray_row_keys(m, dev(rp), dev(col), res, thr)
        wdeg, wkey, wcounts = oracle.gray_row_keys(rp, col, m, res, thr)
        assert np.array_equal(host(deg), wdeg), (res, thr)
        assert np.array_equal(host(key).view(np.uint64), wkey), (res, thr)
        assert list(counts) == wcounts.tolist(), (res, thr)
    # int64 tuple goes through the same kernels
    deg, key, counts = ops.gray_row_keys(m, dev(rp.astype(np.int64)), dev(col.astype(np.int64)), 64, 10)
    wdeg, wkey, wcounts = oracle.gray_row_keys(rp, col, m, 64, 10)
    assert np.array_equal(host(deg), wdeg) and np.array_equal(host(key).view(np.uint64), wkey)


def _mixed_rows(g, n, m, lens, sort=True):
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    rows = [g.choice(m, int(l), replace=False) for l in lens]
    col = np.concatenate([np.sort(r) if sort else r for r in rows]).astype(np.int32) if len(rows) else np.zeros(0, np.int32)
    return rp, col


@pytest.mark.parametrize("res,thr", [(32, 10), (16, 20), (64, 2), (64, 40), (48, 10), (16, 0), (32, 100000)])
def test_gray_row_keys_power_law_path(ops, oracle, res, thr):
    """The kernels a power-law matrix gets (sbx_gray.hip: k_gray_rows_short notices and stops, then k_gray_rows_balanced /
    k_gray_rows_medium / k_gray_units_finish): rows of every class next to each other — empty, 1 .. 64 entries (lane
    per entry, counted rows among them: 33 .. 64 entries at resolution 32), 65 .. 1024 (one unit), 1025 .. 8192 and
    above (several units: partial slots), a hub of 40 K entries (40 units) — at power-of-two and other block widths, 32-
    and 64-bit bitmaps, thresholds on either side of the row lengths, and a last row that ends the array in the middle
    of a 16-byte load."""
    g = np.random.default_rng(1000 + res + thr)
    n = 3000
    m = res * 1024 if res != 48 else 48 * 1000   # 48: a width that is not a power of two (umulhi division)
    lens = g.integers(0, 65, n)
    lens[g.integers(0, n, 400)] = g.integers(65, 1025, 400)          # one-unit rows, two and more per 16 rows in places
    lens[g.integers(0, n, 60)] = g.integers(1025, 9000, 60)          # rows of several units, on both sides of 8192
    lens[[5, 6, 7]] = (40000, 1024, 1025)
    lens[[100, 101, 102, 103]] = (64, 65, 0, 33)
    lens[n - 1] = 1027                                                # the array ends inside a unit's last vector
    lens = np.minimum(lens, m)
    rp, col = _mixed_rows(g, n, m, lens)
    deg, key, counts = ops.gray_row_keys(m, dev(rp), dev(col), res, thr)
    wdeg, wkey, wcounts = oracle.gray_row_keys(rp, col, m, res, thr)
    assert np.array_equal(host(deg), wdeg)
    bad = np.nonzero(host(key).view(np.uint64) != wkey)[0]
    assert len(bad) == 0, (bad[:10], lens[bad[:10]])
    assert list(counts) == wcounts.tolist()


def test_gray_row_keys_power_law_path_unsorted_columns(ops, oracle):
    """The unit kernel adds one count per RUN of equal blocks, which is an optimisation for ascending columns, not an
    assumption: rows whose columns come in random order (arrays that never went through a CSR constructor) give the
    same keys."""
    g = np.random.default_rng(77)
    n, m = 2000, 32 * 512
    lens = g.integers(0, 65, n)
    lens[g.integers(0, n, 300)] = g.integers(65, 3000, 300)
    rp, col = _mixed_rows(g, n, m, lens, sort=False)
    for res, thr in ((32, 10), (64, 5)):
        deg, key, counts = ops.gray_row_keys(m, dev(rp), dev(col), res, thr)
        wdeg, wkey, wcounts = oracle.gray_row_keys(rp, col, m, res, thr)
        assert np.array_equal(host(deg), wdeg) and np.array_equal(host(key).view(np.uint64), wkey)
        assert list(counts) == wcounts.tolist()


def test_gray_row_keys_banded_with_a_few_medium_rows(ops, oracle):
    """A banded matrix with a handful of rows of 65 .. 8192 entries, never two of them among 16 consecutive rows, and two
    rows above 8192: the banded kernel keeps going, lists the long rows for k_gray_long_rows and raises the flag that
    sends the medium ones through k_gray_list_medium + the unit kernels."""
    g = np.random.default_rng(5)
    n, m = 6000, 32 * 4096
    lens = g.integers(1, 40, n)
    for r, l in ((50, 100), (900, 65), (1700, 1024), (2500, 1025), (3300, 5000), (4100, 8192), (4900, 8193), (5700, 30000)):
        lens[r] = l
    rp, col = _mixed_rows(g, n, m, lens)
    for res, thr in ((32, 10), (16, 20)):
        deg, key, counts = ops.gray_row_keys(m, dev(rp), dev(col), res, thr)
        wdeg, wkey, wcounts = oracle.gray_row_keys(rp, col, m, res, thr)
        assert np.array_equal(host(deg), wdeg) and np.array_equal(host(key).view(np.uint64), wkey)
        assert list(counts) == wcounts.tolist()


@pytest.mark.parametrize("res", [1, 2, 3, 5, 8, 12, 15, 16, 17, 31, 33, 64])
def test_gray_row_keys_every_resolution_on_short_rows(ops, oracle, res):
    """Short-row matrices at resolutions below 16: a dense row of d <= 64 entries compares its block counts with
    d / resolution, up to 64 — more counter slices than the 8-lanes-per-row kernel carries, so those calls must take the
    tile kernel (round 2 shipped the fast path without that bound for a while; tools/fuzz_ops.py found it)."""
    g = np.random.default_rng(100 + res)
    n, m = 600, res * 40
    lens = g.integers(0, min(m, 64) + 1, n)
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    col = np.concatenate([np.sort(g.choice(m, int(l), replace=False)) for l in lens]).astype(np.int32)
    for thr in (0, 1, 7, 19, 64):
        deg, key, counts = ops.gray_row_keys(m, dev(rp), dev(col), res, thr)
        wdeg, wkey, wcounts = oracle.gray_row_keys(rp, col, m, res, thr)
        assert np.array_equal(host(deg), wdeg), (res, thr)
        assert np.array_equal(host(key).view(np.uint64), wkey), (res, thr)
        assert list(counts) == wcounts.tolist(), (res, thr)


# ----------------------------------------------------------------------------- RCM
def test_rcm_kat_and_small_fixtures(ops, kat, small):
    k = kat["functionality_3x3"]
    assert host(ops.rcm_reorder(dev(a32(k["row_ptr"])), dev(a32(k["cols"])))).tolist() == k["rcm"]
    z, meta = small
    for name in meta:
        rp, col = z[f"{name}/row_ptr"], z[f"{name}/col"]
        got = host(ops.rcm_reorder(dev(rp), dev(col)))
        assert np.array_equal(got, z[f"{name}/rcm"]), name


def test_rcm_chesapeake(ops, golden_dir):
    z = np.load(os.path.join(golden_dir, "chesapeake.npz"))
    rcm = ops.rcm_reorder(dev(z["row_ptr"]), dev(z["col"]))
    assert np.array_equal(host(rcm), z["rcm"])
    prp, pcol, _ = ops.permute_csr(40, 40, dev(z["row_ptr"]), dev(z["col"]), None, rcm, rcm)
    same((prp, pcol), (z["permute_rcm_row_ptr"], z["permute_rcm_col"]))


@pytest.mark.parametrize("seed", range(10))
def test_rcm_vs_oracle_random(ops, oracle, seed):
    g = np.random.default_rng(seed)
    n = int(2 ** g.integers(6, 15))
    rp, col = synth.random_symmetric_graph(n, avg_deg=1 + seed % 7, seed=seed, n_blocks=1 + seed % 5,
                                           isolated_frac=0.05 * (seed % 4))
    got, stats = ops.rcm_reorder(dev(rp), dev(col), return_stats=True)
    assert np.array_equal(host(got), oracle.rcm_reorder(rp, col)), stats


@pytest.mark.parametrize("maker", ["path", "path_shuffled", "star", "clique", "grid", "grid_shuffled", "grid_wide",
                                   "bushy_tree", "two_hubs"])
def test_rcm_structured(ops, oracle, maker):
    if maker == "path":
        rp, col = synth.path_graph(3000)
    elif maker == "path_shuffled":
        rp, col = synth.path_graph(2049, shuffle_seed=3)
    elif maker == "star":
        rp, col = synth.star_graph(5000, centre=77)   # hub with > 1024 neighbours (chunked expansion)
    elif maker == "clique":
        rp, col = synth.clique_graph(300)
    elif maker == "grid":
        rp, col = synth.grid_graph(40, 300)
    elif maker == "grid_shuffled":
        rp, col = synth.grid_graph(128, 128, shuffle_seed=5)
    elif maker == "grid_wide":
        rp, col = synth.grid_graph(1500, 700, shuffle_seed=9)   # frontiers of ~700-1400 vertices for ~2000 levels
    elif maker == "bushy_tree":
        # every vertex gets 1..130 children: parent groups on both sides of the 64-children counting limit
        g = np.random.default_rng(17)
        src, dst, nxt, frontier = [], [], 1, [0]
        while nxt < 60000 and frontier:
            new = []
            for u in frontier:
                c = int(g.integers(1, 131)) if g.random() < 0.2 else int(g.integers(1, 4))
                for _ in range(c):
                    src.append(u); dst.append(nxt); new.append(nxt); nxt += 1
            frontier = new[:400]
        ids = g.permutation(nxt)
        s_, d_ = synth.symmetrize(ids[np.array(src)], ids[np.array(dst)])
        rp, col = synth.csr_from_edges(nxt, s_, d_)
    else:
        s1, d1 = synth.symmetrize(np.full(6000, 10), np.arange(100, 6100))
        s2, d2 = synth.symmetrize(np.full(6000, 20), np.arange(3000, 9000))
        rp, col = synth.csr_from_edges(9100, np.concatenate([s1, s2]), np.concatenate([d1, d2]))
    assert np.array_equal(host(ops.rcm_reorder(dev(rp), dev(col))), oracle.rcm_reorder(rp, col))


@pytest.mark.parametrize("name", ["rmat16_ef8", "rmat18_ef8", "banded_64k_w4096", "sym_128k"])
def test_rcm_golden_digests(ops, golden_dir, name):
    import hashlib
    with open(os.path.join(golden_dir, "digests.json")) as f:
        d = json.load(f)[name]
    rp, col = getattr(synth, d["generator"])(**d["args"])
    n = len(rp) - 1
    rcm = ops.rcm_reorder(dev(rp), dev(col))
    assert hashlib.sha256(host(rcm).tobytes()).hexdigest() == d["rcm"]
    val = (np.arange(len(col)) % 1021).astype(np.float32)
    out = ops.permute_csr(n, n, dev(rp), dev(col), dev(val), rcm, rcm)
    h = hashlib.sha256()
    for a in out:
        h.update(host(a).tobytes())
    assert h.hexdigest() == d["permute_rcm"]


# ----------------------------------------------------------------------------- 64-bit indices (<int64,int64,double>)
def test_int64_tuple_all_ops(ops, oracle):
    g = np.random.default_rng(77)
    rp32, col32 = synth.rmat_symmetric(11, 8, seed=4)
    rp, col = rp32.astype(np.int64), col32.astype(np.int64)
    n = len(rp) - 1
    val = g.random(len(col))
    order = synth.random_permutation(n, 5, np.int64)
    assert np.array_equal(host(ops.rcm_reorder(dev(rp), dev(col))), oracle.rcm_reorder(rp, col))
    assert np.array_equal(host(ops.degree_reorder(dev(rp), False)), oracle.degree_reorder(rp, False))
    same(ops.permute_csr(n, n, dev(rp), dev(col), dev(val), dev(order), dev(order)),
         oracle.permute_csr(rp, col, val, order, order))
    same(ops.permute_csr(n, n, dev(rp), dev(col), dev(val), dev(order), None), oracle.permute_csr(rp, col, val, order, None))
    assert np.array_equal(host(ops.inverse_permutation(dev(order))), oracle.inverse_permutation(order))
    coo = ops.csr_to_coo(n, n, dev(rp), dev(col), dev(val))
    same(coo, oracle.csr_to_coo(rp, col, val))
    same(ops.coo_to_csr(n, n, *coo), (rp, col, val))
    p = g.permutation(len(col))
    r, c, v = dev(host(coo[0])[p]), dev(col[p]), dev(val[p])
    assert not ops.coo_is_sorted(r, c)
    ops.coo_sort_(n, n, r, c, v)
    same((r, c, v), (host(coo[0]), col, val))
    ucol, uval = col.copy(), val.copy()
    for i in range(n):
        q = g.permutation(rp[i + 1] - rp[i])
        ucol[rp[i]:rp[i + 1]] = ucol[rp[i]:rp[i + 1]][q]
        uval[rp[i]:rp[i + 1]] = uval[rp[i]:rp[i + 1]][q]
    dc, dv_ = dev(ucol), dev(uval)
    assert not ops.csr_rows_sorted(dev(rp), dc)
    ops.csr_sort_rows_(n, n, dev(rp), dc, dv_)
    same((dc, dv_), (col, val))
    deg, key, counts = ops.gray_row_keys(n, dev(rp), dev(col), 32, 10)
    wdeg, wkey, wcounts = oracle.gray_row_keys(rp, col, n, 32, 10)
    assert np.array_equal(host(deg), wdeg) and np.array_equal(host(key).view(np.uint64), wkey)
    assert list(counts) == wcounts.tolist()
    srp, scol, sval = ops.permute_csr_rows(n, n, dev(rp), dev(col), dev(val), dev(order), dev(order), n // 4, n // 2)
    want = oracle.permute_csr(rp, col, val, order, order)
    lo, hi = want[0][n // 4], want[0][n // 2]
    assert np.array_equal(host(srp), want[0][n // 4:n // 2 + 1] - lo) and np.array_equal(host(scol), want[1][lo:hi])


def test_rcm_int64_arrays_native(ops, oracle):
    """RCM on 64-bit index arrays: the kernels of sbx_rcm.hip compiled for int64 row_ptr / col / inv_perm (sbx_rcm64.hip) read
    and write the arrays as they are — no narrowed copies — over graph families that take every branch of the search
    (hubs, many small and mid-size components, deep meshes, paths, isolated vertices); a dimension beyond what the 32-bit
    ids inside can hold is refused, not truncated."""
    from sparsebase_amd import capi
    cases = [synth.rmat_symmetric(14, 8, seed=3), synth.rmat_symmetric(16, 16, seed=1),
             synth.random_symmetric_graph(40000, avg_deg=4, seed=2, n_blocks=3, isolated_frac=0.1),
             synth.random_symmetric_graph(30000, avg_deg=1.2, seed=9, isolated_frac=0.3),
             synth.grid_graph(150, 120, shuffle_seed=4), synth.path_graph(5000, shuffle_seed=1), synth.star_graph(3000),
             synth.clique_graph(70)]
    for rp, col in cases:
        rp64, col64 = rp.astype(np.int64), col.astype(np.int64)
        want = oracle.rcm_reorder(rp, col)
        got = ops.rcm_reorder(dev(rp64), dev(col64))
        assert got.dtype == torch.int64
        assert np.array_equal(host(got), want.astype(np.int64))
    rp, col = cases[0]
    with pytest.raises(capi.SbxError) as e:  # nnz beyond 2^31 is announced through the argument: refused up front
        hd = ops.handle_for(torch.device("cuda"))
        out = torch.empty(len(rp) - 1, dtype=torch.int64, device="cuda")
        hd.check(hd.lib.sbx_rcm_reorder(hd.h, 1, len(rp) - 1, 1 << 31, dev(rp.astype(np.int64)).data_ptr(),
                                        dev(col.astype(np.int64)).data_ptr(), out.data_ptr(), None))
    assert e.value.status == 5  # SBX_ERR_UNSUPPORTED


def test_gray_row_keys_int64_arrays_native(ops, oracle):
    """The Gray key stage on 64-bit index arrays: sbx_gray.hip compiled for int64 row_ptr / col / degree_out
    (sbx_gray64.hip) reads them as they are — four columns are two 16-byte loads — through every kernel family: the
    banded kernel (4 lanes per row), the power-law kernels (tiny / listed / unit / finish), the general tile kernel
    (resolution below 16), rows beyond 8192 entries; a dimension the 32-bit values inside cannot hold is refused."""
    from sparsebase_amd import capi
    g = np.random.default_rng(41)
    cases = [(synth.banded_symmetric(8192, 40, 9, 3), 32, 10), (synth.banded_symmetric(4096, 1024, 11, 5), 16, 0),
             (synth.rmat_symmetric(14, 16, seed=2), 32, 10), (synth.rmat_symmetric(13, 8, seed=6), 64, 3),
             (synth.rmat_symmetric(12, 8, seed=7), 8, 2)]
    n_l = 1 << 15  # a few rows of ~20 000 entries among short ones
    lens = g.integers(0, 12, n_l)
    lens[[5, 700, 9000]] = (20000, 9000, 30000)
    rp_l = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    col_l = np.concatenate([np.sort(g.choice(n_l, l, replace=False)) for l in lens]).astype(np.int32)
    cases.append(((rp_l, col_l), 32, 10))
    for (rp, col), res, thr in cases:
        n = len(rp) - 1
        deg, key, counts = ops.gray_row_keys(n, dev(rp.astype(np.int64)), dev(col.astype(np.int64)), res, thr)
        wdeg, wkey, wcounts = oracle.gray_row_keys(rp, col, n, res, thr)
        assert deg.dtype == torch.int64
        assert np.array_equal(host(deg), wdeg.astype(np.int64)), (res, thr)
        assert np.array_equal(host(key).view(np.uint64), wkey), (res, thr)
        assert list(counts) == wcounts.tolist()
    (rp, col), res, thr = cases[0]
    with pytest.raises(capi.SbxError) as e:
        ops.gray_row_keys(1 << 32, dev(rp.astype(np.int64)), dev(col.astype(np.int64)), res, thr)
    assert e.value.status == 5  # SBX_ERR_UNSUPPORTED


def test_int64_values_beyond_int32(ops, oracle):
    """64-bit indices with values >= 2^31.  The conversions COO <-> CSR and the two sortedness checks run native 64-bit
    kernels: column ids of any size are accepted and the results are the oracle's, bit for bit; so do both constructor
    sorts and the permute.  What a 64-bit sort key cannot hold is refused loudly (SBX_ERR_UNSUPPORTED), never truncated."""
    from sparsebase_amd import capi
    g = np.random.default_rng(5)
    n, m = 300, 1 << 40
    lens = g.integers(0, 40, n)
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    col = np.concatenate([np.sort(g.choice(1 << 20, l, replace=False)) for l in lens]).astype(np.int64)
    col = col * (1 << 19) + (1 << 33)                       # every column id beyond 2^33, sorted inside the rows
    val = g.random(len(col))
    assert ops.csr_rows_sorted(dev(rp), dev(col)) and oracle.csr_rows_sorted(rp, col)
    coo = ops.csr_to_coo(n, m, dev(rp), dev(col), dev(val))
    same(coo, oracle.csr_to_coo(rp, col, val))
    assert ops.coo_is_sorted(coo[0], coo[1])
    same(ops.coo_to_csr(n, m, *coo), (rp, col, val))
    same(ops.coo_to_csr(n, m, *coo, rows_sorted=True), (rp, col, val))
    # an unsorted row array (ignore_sort inputs): histogram + scan, columns and values verbatim (converter_order_two.cc:180-192)
    p = g.permutation(len(col))
    r_u, c_u, v_u = host(coo[0])[p], col[p], val[p]
    same(ops.coo_to_csr(n, m, dev(r_u), dev(c_u), dev(v_u)), oracle.coo_to_csr(n, r_u, c_u, v_u))
    assert not ops.coo_is_sorted(dev(r_u), dev(c_u))
    ucol = col.copy()
    ucol[rp[5]:rp[6]] = ucol[rp[5]:rp[6]][::-1] if lens[5] > 1 else ucol[rp[5]:rp[6]]
    assert ops.csr_rows_sorted(dev(rp), dev(ucol)) == oracle.csr_rows_sorted(rp, ucol)
    # a 64-bit row id outside [0, n) must not index row_ptr
    bad_row = host(coo[0]).copy()
    bad_row[-1] = 1 << 35
    with pytest.raises(capi.SbxError) as e:
        ops.coo_to_csr(n, m, dev(bad_row), dev(col), dev(val), rows_sorted=True)
    assert e.value.status == 1  # SBX_ERR_BAD_ARG
    # the features and the degree order read 64-bit arrays natively too: column ids beyond 2^33 (bandwidth, profile —
    # what the reference computes for such a pattern, |row - col| in 64 bits), row_ptr values beyond 2^40 (nnz of any
    # size: only differences matter)
    assert ops.csr_bandwidth(dev(rp), dev(col)) == oracle.csr_bandwidth(rp, col)
    assert ops.csr_profile(dev(rp), dev(col)) == oracle.csr_profile(rp, col)
    assert ops.csr_profile(dev(rp), dev(ucol)) == oracle.csr_profile(rp, ucol)       # a row out of order: general path
    big = rp + (1 << 40)
    assert np.array_equal(host(ops.csr_degrees(dev(big))), oracle.csr_degrees(rp))
    assert np.array_equal(host(ops.csr_degree_distribution(dev(big), len(col), torch.float64)),
                          oracle.csr_degree_distribution(rp, len(col), np.float64))
    lens2 = lens.copy()
    lens2[[7, 90, 200]] = (300, 256, 70000)                                          # rows on both sides of the 255 cap
    rp2 = np.concatenate([[0], np.cumsum(lens2)]).astype(np.int64)
    for asc in (True, False):
        assert np.array_equal(host(ops.degree_reorder(dev(rp2 + (1 << 40)), asc)), oracle.degree_reorder(rp2, asc))
    # the row-wise permute (no column map) is native too: the columns travel as they are
    order = synth.random_permutation(n, 3, np.int64)
    same(ops.permute_csr(n, m, dev(rp), dev(col), dev(val), dev(order), None), oracle.permute_csr(rp, col, val, order, None, m=m))
    same(ops.permute_csr(n, m, dev(rp), dev(col), None, dev(order), None), oracle.permute_csr(rp, col, None, order, None, m=m))
    srp, scol, sval = ops.permute_csr_rows(n, m, dev(rp), dev(col), dev(val), dev(order), None, n // 3, n // 2)
    want = oracle.permute_csr(rp, col, val, order, None, m=m)
    lo, hi = want[0][n // 3], want[0][n // 2]
    assert np.array_equal(host(srp), want[0][n // 3:n // 2 + 1] - lo) and np.array_equal(host(scol), want[1][lo:hi])
    assert np.array_equal(host(sval), want[2][lo:hi])
    # RCM and the Gray keys read 64-bit arrays natively as well (sbx_rcm64.hip, sbx_gray64.hip: no narrowed copies); their
    # ids and offsets are 32-bit inside, so a dimension beyond int32 is refused instead of truncated
    grp_, gcol_ = synth.rmat_symmetric(10, 6, seed=8)
    assert np.array_equal(host(ops.rcm_reorder(dev(grp_.astype(np.int64)), dev(gcol_.astype(np.int64)))),
                          oracle.rcm_reorder(grp_, gcol_).astype(np.int64))
    gd, gk, gc = ops.gray_row_keys(len(grp_) - 1, dev(grp_.astype(np.int64)), dev(gcol_.astype(np.int64)), 16, 4)
    wd_, wk_, wc_ = oracle.gray_row_keys(grp_, gcol_, len(grp_) - 1, 16, 4)
    assert np.array_equal(host(gd), wd_.astype(np.int64)) and np.array_equal(host(gk).view(np.uint64), wk_) and list(gc) == wc_.tolist()
    with pytest.raises(capi.SbxError) as e:
        ops.gray_row_keys(m, dev(rp), dev(col), 16, 4)   # m = 2^40 columns
    assert e.value.status == 5
    # the COO constructor's sort is native: coordinates of any size inside the matrix (packed keys: 9 + 40 bits here)
    for v in (v_u, v_u.astype(np.float32), None):
        r_s, c_s, v_s = dev(r_u.copy()), dev(c_u.copy()), dev(v)
        ops.coo_sort_(n, m, r_s, c_s, v_s)
        same((r_s, c_s, v_s), oracle.coo_sort(r_u, c_u, v, n=n, m=m))
    # ... and so are the CSR constructor's row sort and the permute of rows that arrive out of order: column ids beyond
    # 2^31 do not fit the LDS sorts' 32-bit keys and take the wide path (64-bit keys: new row | column)
    scol, sval = col.copy(), val.copy()
    for i in range(n):
        q = g.permutation(lens[i])
        scol[rp[i]:rp[i + 1]], sval[rp[i]:rp[i + 1]] = col[rp[i]:rp[i + 1]][q], val[rp[i]:rp[i + 1]][q]
    for v in (sval, sval.astype(np.float32), None):
        dc, dv_ = dev(scol.copy()), dev(v)
        ops.csr_sort_rows_(n, m, dev(rp), dc, dv_)
        same((dc, dv_), oracle.csr_sort_rows(rp, scol, v, m=m))
        same(ops.permute_csr(n, m, dev(rp), dev(scol), dev(v), dev(order), None), oracle.permute_csr(rp, scol, v, order, None, m=m))
    dcol, r9 = scol.copy(), int(np.argmax(lens >= 8))
    dcol[rp[r9]:rp[r9] + 3] = dcol[rp[r9] + 3]                                        # duplicates end in value order
    same(ops.permute_csr(n, m, dev(rp), dev(dcol), dev(sval), dev(order), None), oracle.permute_csr(rp, dcol, sval, order, None, m=m))
    # a column map over ids that need 64 bits is the same wide path: see test_int64_permute_column_map_beyond_2_31
    # (tests/test_gpu_fullsize.py: the map alone is 26 GB).  What cannot be represented is refused loudly:
    with pytest.raises(capi.SbxError) as e:                                           # 40 + 40 key bits
        ops.coo_sort_(1 << 40, m, dev(r_u.copy()), dev(c_u.copy()), dev(v_u.copy()))
    assert e.value.status == 5  # SBX_ERR_UNSUPPORTED
    with pytest.raises(capi.SbxError) as e:                                           # columns outside the matrix AND beyond int32
        ops.coo_sort_(n, 1 << 20, dev(r_u.copy()), dev(c_u.copy()), dev(v_u.copy()))
    assert e.value.status == 5


def test_int64_degenerate_inputs():
    """Empty matrices, empty rows only, one entry, a shard of no rows, duplicate coordinates — through the native 64-bit
    entry points of the permute, both constructor sorts and the CSC conversions (tools/int64_degenerate.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "int64_degenerate.py")], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "int64 degenerate ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("case", ["mixed", "all_tail_one_degree", "wide_int64", "tail_of_one", "tail_of_two", "small_n"])
def test_degree_reorder_counting_passes(ops, oracle, case):
    """DegreeReorder's placement (degree_reorder.cc:22-62) through every branch of the device path: the 255-capped
    counting pass alone, a last bucket of 1 / 2 / tens of thousands / all rows, 1 - 4 digits of the tail sort, ties in
    the tail (equal degrees keep descending id order), sizes around the 64 / 256 / 1024-row units of the kernels."""
    g = np.random.default_rng(11)
    cases = []
    if case == "mixed":
        n = 200_000
        lens = g.integers(0, 300, n)
        heavy = g.random(n) < 0.3
        lens[heavy] = g.integers(255, 5000, int(heavy.sum()))          # a tail of 60 K rows with many ties
        lens[g.integers(0, n, 5)] = (1 << 20) + g.integers(0, 3, 5)      # 21-bit degrees: three digits
        cases.append((lens, np.int64))
        cases.append((np.minimum(lens, 9000), np.int32))
    elif case == "all_tail_one_degree":
        cases.append((np.full(70_001, 300), np.int32))
        cases.append((np.full(1025, 255), np.int32))
        cases.append((256 + (np.arange(1_200_000) % 3), np.int32))           # a tail above 2^20 rows: the generic sort
    elif case == "wide_int64":
        n = 50_000
        lens = g.integers(0, 1 << 30, n)                                 # 30-bit degrees: four digits, nnz ~ 2^45
        lens[g.random(n) < 0.5] = 0
        cases.append((lens, np.int64))
    elif case == "tail_of_one":
        lens = g.integers(0, 255, 3000)
        lens[1234] = 255
        cases.append((lens, np.int32))
    elif case == "tail_of_two":
        lens = g.integers(0, 255, 3000)
        lens[[17, 2999]] = (400, 400)
        cases.append((lens, np.int32))
    else:
        for n in (2, 3, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 4097):
            lens = g.integers(0, 600, n)
            cases.append((lens, np.int32))
    for lens, dt in cases:
        rp = np.concatenate([[0], np.cumsum(lens.astype(np.int64))]).astype(dt)
        for asc in (True, False):
            got = host(ops.degree_reorder(dev(rp), asc))
            assert np.array_equal(got, oracle.degree_reorder(rp, asc)), (case, len(lens), asc)


# ----------------------------------------------------------------------------- degenerate / ragged shapes
def _check_all_ops(ops, oracle, rp, col, m, seed=0):
    n, nnz = len(rp) - 1, len(col)
    g = np.random.default_rng(seed)
    val = g.integers(-4, 4, nnz).astype(np.int32)
    drp, dcol, dval = dev(rp), dev(col), dev(val)
    for asc in (True, False):
        assert np.array_equal(host(ops.degree_reorder(drp, asc)), oracle.degree_reorder(rp, asc))
    ro = synth.random_permutation(n, seed + 1) if n else np.zeros(0, np.int32)
    co = synth.random_permutation(m, seed + 2) if m else np.zeros(0, np.int32)
    for r, c in ((ro, co), (ro, None), (None, co)):
        same(ops.permute_csr(n, m, drp, dcol, dval, dev(r), dev(c)), oracle.permute_csr(rp, col, val, r, c))
    coo = ops.csr_to_coo(n, m, drp, dcol, dval)
    same(coo, oracle.csr_to_coo(rp, col, val))
    same(ops.coo_to_csr(n, m, *coo), oracle.coo_to_csr(n, *oracle.csr_to_coo(rp, col, val)))
    assert ops.csr_rows_sorted(drp, dcol) == oracle.csr_rows_sorted(rp, col)
    if n == m:
        # symmetrise the pattern for RCM
        r_idx = np.repeat(np.arange(n), np.diff(rp))
        s, d = synth.symmetrize(r_idx, col.astype(np.int64))
        srp, scol = synth.csr_from_edges(n, s, d) if n else (np.zeros(1, np.int32), np.zeros(0, np.int32))
        assert np.array_equal(host(ops.rcm_reorder(dev(srp), dev(scol))), oracle.rcm_reorder(srp, scol))
        if n >= 1 and (n < 16 or n % 16 == 0):
            deg, key, counts = ops.gray_row_keys(m, dev(srp), dev(scol), 16, 3)
            wdeg, wkey, wcounts = oracle.gray_row_keys(srp, scol, m, 16, 3)
            assert np.array_equal(host(deg), wdeg) and np.array_equal(host(key).view(np.uint64), wkey)
            assert list(counts) == wcounts.tolist()


def test_degenerate_shapes(ops, oracle):
    z = np.zeros(0, np.int32)
    _check_all_ops(ops, oracle, np.zeros(1, np.int32), z, 0)                       # 0 x 0
    _check_all_ops(ops, oracle, np.zeros(6, np.int32), z, 5)                       # all rows empty
    _check_all_ops(ops, oracle, np.array([0, 1], np.int32), np.array([0], np.int32), 1)   # 1 x 1
    _check_all_ops(ops, oracle, np.array([0, 0, 0, 3, 3], np.int32), np.array([0, 1, 3], np.int32), 4)
    # one dense row among empty ones; wide rectangular; tall rectangular
    _check_all_ops(ops, oracle, np.array([0, 0, 3000, 3000], np.int32), np.arange(3000, dtype=np.int32), 3000)
    rp, col = synth.random_rect_csr(3, 5000, 4000, 1)
    _check_all_ops(ops, oracle, rp, col, 5000)
    rp, col = synth.random_rect_csr(5000, 3, 4000, 2, dup_frac=0.5)
    _check_all_ops(ops, oracle, rp, col, 3)


@pytest.mark.parametrize("seed", range(2))
def test_rcm_several_wide_components(ops, oracle, seed):
    # three disjoint power-law graphs with shuffled ids: every component has levels wide enough for the
    # bitmap-from-parent-positions path, and that path relies on the parent positions being reset between
    # components (sbx_rcm.hip: k_visited_from_ppos, reset_ppos)
    g = np.random.default_rng(100 + seed)
    parts, offset = [], 0
    for k, scale in enumerate((17, 16, 17)):
        rp, col = synth.rmat_symmetric(scale, 8, seed=10 * seed + k)
        r_idx = np.repeat(np.arange(len(rp) - 1), np.diff(rp))
        parts.append((r_idx + offset, col.astype(np.int64) + offset))
        offset += len(rp) - 1
    relabel = g.permutation(offset)
    src = relabel[np.concatenate([p[0] for p in parts])]
    dst = relabel[np.concatenate([p[1] for p in parts])]
    rp, col = synth.csr_from_edges(offset, src, dst)
    got, stats = ops.rcm_reorder(dev(rp), dev(col), return_stats=True)
    assert np.array_equal(host(got), oracle.rcm_reorder(rp, col)), stats
    assert stats["large_components"] >= 3


@pytest.mark.parametrize("seed", range(4))
def test_rcm_many_midsize_components(ops, oracle, seed):
    # disjoint union of paths, grids, stars and cliques of 65..600 vertices with shuffled ids:
    # every component takes the host-driven large-component route
    g = np.random.default_rng(seed)
    parts, offset = [], 0
    for k in range(12):
        kind = (seed + k) % 4
        if kind == 0:
            rp, col = synth.path_graph(int(g.integers(65, 400)), shuffle_seed=k)
        elif kind == 1:
            rp, col = synth.grid_graph(int(g.integers(8, 20)), int(g.integers(9, 30)), shuffle_seed=k)
        elif kind == 2:
            rp, col = synth.star_graph(int(g.integers(70, 300)), centre=3)
        else:
            rp, col = synth.clique_graph(int(g.integers(65, 90)))
        r_idx = np.repeat(np.arange(len(rp) - 1), np.diff(rp))
        parts.append((r_idx + offset, col.astype(np.int64) + offset))
        offset += len(rp) - 1
    n = offset + 7  # a few isolated vertices at the end
    relabel = g.permutation(n)
    src = relabel[np.concatenate([p[0] for p in parts])]
    dst = relabel[np.concatenate([p[1] for p in parts])]
    rp, col = synth.csr_from_edges(n, src, dst)
    got, stats = ops.rcm_reorder(dev(rp), dev(col), return_stats=True)
    assert np.array_equal(host(got), oracle.rcm_reorder(rp, col)), stats
    assert stats["large_components"] >= 10


def _many_components(count, seed, lo=65, hi=420):
    """Disjoint union of `count` paths / 2-D grids / stars / random trees of lo..hi vertices, vertex ids shuffled."""
    g = np.random.default_rng(seed)
    srcs, dsts, offset = [], [], 0
    for k in range(count):
        kind = k % 4
        sz = int(g.integers(lo, hi))
        if kind == 0:    # path
            s, d = np.arange(sz - 1), np.arange(1, sz)
        elif kind == 1:  # grid
            r = max(2, int(np.sqrt(sz)))
            c = max(2, sz // r)
            sz = r * c
            ids = np.arange(sz).reshape(r, c)
            s = np.concatenate([ids[:, :-1].ravel(), ids[:-1, :].ravel()])
            d = np.concatenate([ids[:, 1:].ravel(), ids[1:, :].ravel()])
        elif kind == 2:  # star
            s, d = np.zeros(sz - 1, np.int64), np.arange(1, sz)
        else:            # random tree
            s, d = np.arange(1, sz), (g.random(sz - 1) * np.arange(1, sz)).astype(np.int64)
        srcs.append(s + offset)
        dsts.append(d + offset)
        offset += sz
    n = offset + 11
    relabel = g.permutation(n)
    s, d = relabel[np.concatenate(srcs)], relabel[np.concatenate(dsts)]
    s, d = synth.symmetrize(s, d)
    return synth.csr_from_edges(n, s, d)


def test_rcm_twenty_thousand_midsize_components(ops, oracle):
    """The many-component cliff: 20 000 components of 65..420 vertices.  Ordered one by one from the host this took tens
    of seconds; batched (one lane per component, sparsebase_amd/csrc/sbx_rcm.hip k_rcm_small) it must be bit-exact and
    beat the CPU restatement of the reference's serial loop (rcm_reorder.cc:104-155) on the same graph."""
    import time
    rp, col = _many_components(20000, seed=5)
    drp, dcol = dev(rp), dev(col)
    got, stats = ops.rcm_reorder(drp, dcol, return_stats=True)
    t0 = time.perf_counter()
    want = oracle.rcm_reorder(rp, col)
    cpu_s = time.perf_counter() - t0
    assert np.array_equal(host(got), want), stats
    assert stats["small_components"] >= 20000 and stats["large_components"] == 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ops.rcm_reorder(drp, dcol)
    torch.cuda.synchronize()
    gpu_s = time.perf_counter() - t0
    assert gpu_s < cpu_s, (gpu_s, cpu_s)
    # a few mid-size components (below the batching threshold) still take the host-driven route, with hubs and all
    rp2, col2 = _many_components(40, seed=6, lo=65, hi=1500)
    got2, stats2 = ops.rcm_reorder(dev(rp2), dev(col2), return_stats=True)
    assert np.array_equal(host(got2), oracle.rcm_reorder(rp2, col2)) and stats2["large_components"] == 40
    # ... and above it, sizes up to the batch limit (stars with ~2000 leaves: the in-lane heapsort of the children)
    rp3, col3 = _many_components(300, seed=7, lo=1200, hi=2040)
    got3, stats3 = ops.rcm_reorder(dev(rp3), dev(col3), return_stats=True)
    assert np.array_equal(host(got3), oracle.rcm_reorder(rp3, col3)), stats3


def _edges_to_sym_csr(n, u, v):
    uu = np.concatenate([u, v])
    vv = np.concatenate([v, u])
    key = np.unique(uu.astype(np.int64) * n + vv)
    r, c = (key // n), (key % n)
    rp = np.concatenate([[0], np.cumsum(np.bincount(r, minlength=n))]).astype(np.int32)
    return rp, c.astype(np.int32)


@pytest.mark.parametrize("shape", ["clustered", "spread", "one_parent_two_scales"])
def test_rcm_small_level_after_a_wide_one(ops, oracle, shape):
    """k_level_sort_small (sbx_rcm.hip): a frontier too wide for the single-workgroup level kernel (> 4096 vertices)
    discovers a level of <= 4096 vertices, which one workgroup orders by bucket rank.  'clustered' / 'one_parent_two_scales'
    put more than RCM_BR_MAX keys in one bucket (the bitonic fallback), 'spread' stays on the bucket path."""
    g = np.random.default_rng(5)
    wide = 6000
    n = 1 + wide + 4000 + 50000
    ids = g.permutation(n).astype(np.int64)  # shuffled labels: ids do not follow discovery order
    root, a = ids[0], ids[1:1 + wide]
    b = ids[1 + wide:1 + wide + 4000]
    u = [np.full(wide, root)]
    v = [a]
    if shape == "clustered":     # two parents far apart in the wide level; one owns 300 children
        pa = np.sort(a)
        u += [np.full(300, pa[0]), np.full(1, pa[-1])]
        v += [b[:300], b[300:301]]
    elif shape == "spread":      # 3500 children over 3500 different parents
        u += [a[:3500]]
        v += [b[:3500]]
    else:                        # one parent, children's ids in two tight far-apart clusters
        bs = np.sort(ids[1 + wide:])
        u += [np.full(401, a[7])]
        v += [np.concatenate([bs[:400], bs[-1:]])]
    # a tail below the small level so that it is not the last one
    u += [b[:100]]
    v += [ids[1 + wide + 4000:1 + wide + 4100]]
    rp, col = _edges_to_sym_csr(n, np.concatenate(u), np.concatenate(v))
    got, stats = ops.rcm_reorder(dev(rp), dev(col), return_stats=True)
    assert np.array_equal(host(got), oracle.rcm_reorder(rp, col)), stats


@pytest.mark.parametrize("mode", ["barriers_give_up", "ordered_sweeps", "unordered_everywhere", "bottom_up_early",
                                  "no_chain", "chain_of_one", "chain_tail_gives_up", "in_line", "no_head_chain", "no_tie_walk",
                                  "tie_walk_tight", "tie_walk_few_members", "tie_walk_hands_over",
                                  "no_tie_spec"])
def test_rcm_sweep_variants_in_a_child(mode):
    """The switches of the RCM's pseudo-peripheral sweeps are read once per process, hence the children:
    SBX_DEBUG_GB_SPINS=0 makes every grid barrier of the persistent kernels give up at once (what a barrier does when its
    workgroups are not all running, e.g. on a GPU shared with another process) — every sweep and tie-break is then redone
    by the one-launch-per-level kernels; SBX_RCM_UNORDERED=0 keeps the order inside every level (round 1's sweeps);
    SBX_DEBUG_UB_MAX_LEVELS lifts the depth limit so that grids and bands take the unordered sweeps too;
    SBX_DEBUG_BU_RATIO=0.3 turns the ordered sweeps bottom-up while hubs are still unvisited: their rows go through the
    chunk queue of k_bfs_bottom_up / k_bfs_bottom_up_heavy (several chunks of one row meeting in an atomicMin);
    SBX_RCM_UBFS_CHAIN=0 / 1: the unordered sweeps read back after every big level / chain one bottom-up level behind it
    (the default chains three: every way a chain can end — sweep over, small frontier, chain too short, a top-down
    level — comes up between the three settings and these graphs); SBX_DEBUG_CHAIN_TAIL_ABORT=1: the small-level kernel
    at the tail of a chain gives up at its first grid barrier, after claiming vertices and before it could report that
    it ran — the sweep must be thrown away all the same; SBX_RCM_HEAD_CHAIN=0: the call's first sweep reads back behind
    its head small-level run instead of beginning its chain on the device (the scale-19 graph is the one above the 2^17
    vertices the head chain asks for: every other mode runs WITH it there); SBX_RCM_TIE_WALK=0: every tie-break through the persistent cone kernels (by
    default the candidates' workgroup does marking and walk itself where the cone is small: k_ubfs_ties_small / tie_walk);
    SBX_DEBUG_TIE_EDGES=40 / SBX_DEBUG_TIE_CAP=3: that walk's limits lowered until these graphs leave it at every one of
    its exits — after the leading levels, in the middle of the marking, in the walk down — and the persistent kernels must
    find everything as the candidates' kernel alone leaves it; SBX_DEBUG_TIE_SINGLE=8: every smallest-member step of more
    than eight entries is handed to the grid kernel and taken up again by k_tie_walk_resume; SBX_RCM_TIE_SPEC=0: the host
    looks at every tie walk's outcome before it enqueues the next sweep (by default the sweep goes out behind the walk
    unseen, its first kernels leave if the walk did, and the persistent kernels are called in afterwards: what the modes
    with lowered limits exercise on every second tie-break); "in_line": no side streams inside the call (SBX_RCM_CC_OVERLAP=0,
    SBX_RCM_SPLIT_EXPAND=0, SBX_RCM_OVERLAP=0: the other components are labelled, a wide frontier's light rows expanded
    and the degree ranks built on the caller's stream — what a call under the handle's profiler does)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np, torch; sys.path[:0] = [%r, %r]\n"
        "from orc import Oracle; from sparsebase_amd import ops, synth\n"
        "orc = Oracle(); d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()\n"
        "cases = [synth.rmat_symmetric(16, 8, seed=3), synth.rmat_symmetric(14, 3, seed=5), synth.rmat_symmetric(19, 12, seed=2),\n"
        "         synth.random_symmetric_graph(30000, avg_deg=3, seed=2, n_blocks=3, isolated_frac=0.1),\n"
        "         synth.banded_symmetric(40000, 6, per_row=4, seed=1), synth.grid_graph(150, 150, shuffle_seed=4)]\n"
        "took = []\n"
        "for rp, col in cases:\n"
        "    got, stats = ops.rcm_reorder(d(rp), d(col), return_stats=True)\n"
        "    assert np.array_equal(got.cpu().numpy(), orc.rcm_reorder(rp, col))\n"
        "    took.append(stats['unordered_sweeps'])\n"
        "ops.profile_enable(True)  # (a call under the handle's profiler runs without side streams)\n"
        "got, stats = ops.rcm_reorder(d(cases[0][0]), d(cases[0][1]), return_stats=True)\n"
        "ops.profile_enable(False)\n"
        "assert np.array_equal(got.cpu().numpy(), orc.rcm_reorder(*cases[0]))\n"
        "print('rcm variant ok', 'unordered sweeps:', took, 'profiled:', stats['unordered_sweeps'])\n"
        % (root, os.path.join(root, "tests")))
    extra = {"barriers_give_up": {"SBX_DEBUG_GB_SPINS": "0"}, "ordered_sweeps": {"SBX_RCM_UNORDERED": "0"},
             "unordered_everywhere": {"SBX_DEBUG_UB_MAX_LEVELS": "1000000"},
             "bottom_up_early": {"SBX_DEBUG_BU_RATIO": "0.3", "SBX_RCM_UNORDERED": "0"},
             "no_chain": {"SBX_RCM_UBFS_CHAIN": "0"}, "chain_of_one": {"SBX_RCM_UBFS_CHAIN": "1"},
             "chain_tail_gives_up": {"SBX_DEBUG_CHAIN_TAIL_ABORT": "1"}, "no_head_chain": {"SBX_RCM_HEAD_CHAIN": "0"},
             "no_tie_walk": {"SBX_RCM_TIE_WALK": "0"}, "tie_walk_tight": {"SBX_DEBUG_TIE_EDGES": "40"},
             "tie_walk_few_members": {"SBX_DEBUG_TIE_CAP": "3"}, "tie_walk_hands_over": {"SBX_DEBUG_TIE_SINGLE": "8"},
             "no_tie_spec": {"SBX_RCM_TIE_SPEC": "0"},
             "in_line": {"SBX_RCM_CC_OVERLAP": "0", "SBX_RCM_SPLIT_EXPAND": "0", "SBX_RCM_OVERLAP": "0"}}[mode]
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **extra), capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0 and "rcm variant ok" in r.stdout, r.stdout + r.stderr
    # which sweeps ran (sbx_rcm_stats.unordered_sweeps): the power-law graphs' searches are unordered wherever the mode
    # allows it — with and without side streams, under the profiler — and never where it does not (a build that quietly
    # fell back to ordered sweeps without its side stream still produced the right orders)
    line = [l for l in r.stdout.splitlines() if l.startswith("rcm variant ok")][-1]
    took = eval(line.split("unordered sweeps:")[1].split("profiled:")[0])
    profiled = int(line.split("profiled:")[1])
    if mode in ("ordered_sweeps", "bottom_up_early", "barriers_give_up"):
        assert took == [0] * len(took) and profiled == 0, line
    elif mode == "chain_tail_gives_up":
        assert len(took) == 6, line  # (sweeps are thrown away at the chain's tail: how many survive is not fixed)
    else:
        assert all(t >= 1 for t in took[:3]) and profiled >= 1, line
    if mode == "unordered_everywhere":
        assert all(t >= 1 for t in took), line


@pytest.mark.parametrize("spins", [1, 2, 3, 5, 8, 13, 21, 34])
def test_rcm_barriers_that_sometimes_give_up(spins):
    """Grid barriers of the persistent kernels that give up now and then (a handful of polls allowed: what a GPU shared
    with other processes does to them, rarely): whichever kernel it hits — the small-level run at the head of a sweep,
    the one at the tail of a device-driven chain (which then never gets to say that it ran), the cone marking or the
    walk of the tie-break — the sweep is redone by the ordered kernels and the order is the reference's.  A handle backs
    off for 16 calls after a give-up, so every child starts with a different graph."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np, torch; sys.path[:0] = [%r, %r]\n"
        "from orc import Oracle; from sparsebase_amd import ops, synth\n"
        "orc = Oracle(); d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()\n"
        "cases = [synth.rmat_symmetric(16, 8, seed=3), synth.rmat_symmetric(17, 6, seed=11), synth.rmat_symmetric(15, 12, seed=5),\n"
        "         synth.random_symmetric_graph(60000, avg_deg=5, seed=2, n_blocks=2, isolated_frac=0.1)]\n"
        "k = %d %% len(cases); cases = cases[k:] + cases[:k]\n"
        "for rp, col in cases * 5:\n"
        "    assert np.array_equal(ops.rcm_reorder(d(rp), d(col)).cpu().numpy(), orc.rcm_reorder(rp, col))\n"
        "print('rcm variant ok')\n" % (root, os.path.join(root, "tests"), spins))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SBX_DEBUG_GB_SPINS=str(spins)),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "rcm variant ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("mode", ["ranked_keys", "full_keys"])
def test_rcm_cuthill_mckee_level_keys_in_a_child(mode):
    """The Cuthill-McKee sweep orders a big level either by full (parent position, degree rank) keys or — from a third
    of the ranked vertices on — by keys read off a bitmap in RANK space, already in degree-rank order, of which only the
    parent-position digits are sorted (sbx_rcm.hip: k_fresh_words_ranked).  Graphs with levels of 10^5 .. 10^6 vertices;
    SBX_DEBUG_RCM_RANKED_DIV=100000 sends every level that takes the bitmap pass that way, SBX_RCM_RANKED_KEYS=0 none.
    (Read once per process, hence the children.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np, torch; sys.path[:0] = [%r, %r]\n"
        "from orc import Oracle; from sparsebase_amd import ops, synth\n"
        "orc = Oracle(); d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()\n"
        "cases = [synth.rmat_symmetric(19, 8, seed=3), synth.rmat_symmetric(18, 16, seed=9),\n"
        "         synth.random_symmetric_graph(400000, avg_deg=6, seed=2, n_blocks=1, isolated_frac=0.05),\n"
        "         synth.random_symmetric_graph(300000, avg_deg=12, seed=7, n_blocks=2, isolated_frac=0.3)]\n"
        "for rp, col in cases:\n"
        "    assert np.array_equal(ops.rcm_reorder(d(rp), d(col)).cpu().numpy(), orc.rcm_reorder(rp, col))\n"
        "print('rcm variant ok')\n" % (root, os.path.join(root, "tests")))
    extra = {"ranked_keys": {"SBX_DEBUG_RCM_RANKED_DIV": "100000"}, "full_keys": {"SBX_RCM_RANKED_KEYS": "0"}}[mode]
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **extra), capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0 and "rcm variant ok" in r.stdout, r.stdout + r.stderr


def test_rcm_refuses_unsymmetric_patterns(ops, oracle):
    """Directed inputs (an edge list read with read_undirected=False): a BFS cannot reach its weakly connected
    component.  The reference leaks stale distances there; here every such input must end in a clean error —
    never an out-of-bounds write or a non-permutation returned as success."""
    from sparsebase_amd import capi
    g = np.random.default_rng(77)
    cases = []
    # (a) small components of directed chains 0 -> 1 -> 2 ...: from the smallest id everything is reachable,
    #     reversed chains (k -> k-1) are not: the sweep from the smallest vertex reaches nothing else
    n = 4000
    src = np.arange(1, n)
    dst = src - 1
    keep = (src % 50) != 0
    cases.append(synth.csr_from_edges(n, src[keep], dst[keep]))
    # (b) a random directed graph: big weak components, poor directed reachability
    n = 30000
    cases.append(synth.csr_from_edges(n, g.integers(0, n, 4 * n), g.integers(0, n, 4 * n)))
    # (c) a directed star into vertex 0 plus a symmetric rest: vertices hang under the first component
    n = 2000
    s1, d1 = synth.symmetrize(np.arange(0, 500), np.arange(1, 501))
    cases.append(synth.csr_from_edges(n, np.concatenate([s1, np.arange(600, 900)]), np.concatenate([d1, np.zeros(300, int)])))
    for rp, col in cases:
        with pytest.raises(capi.SbxError) as e:
            ops.rcm_reorder(dev(rp), dev(col))
        assert e.value.status == 1 and "symmetric" in str(e.value)
    # the library stays usable afterwards, and symmetric inputs are unaffected
    rp, col = synth.random_symmetric_graph(5000, seed=3)
    assert np.array_equal(host(ops.rcm_reorder(dev(rp), dev(col))), oracle.rcm_reorder(rp, col))
    _, stats = ops.rcm_reorder(dev(rp), dev(col), return_stats=True)
    deg = np.diff(rp)
    assert stats["isolated"] == int((deg == 0).sum()) and stats["components"] >= stats["isolated"] + 1
    assert stats["reference_sweeps"] >= 2 or stats["large_components"] == 0


@pytest.mark.gpu
def test_rcm_on_a_callers_nonblocking_stream_the_way_a_c_caller_reads_it(oracle):
    """include/sbx.h, Conventions: device outputs are complete in the order of the handle's stream, and sbx_rcm_reorder
    returns with the kernel that writes the big component's positions enqueued behind its last read-back.  A C caller
    with its own NON-BLOCKING stream (no implicit ordering against the null stream) reads inv_perm_out three legal
    ways — sbx_sync then a blocking hipMemcpy on the null stream; an event recorded on the handle's stream that another
    non-blocking stream waits for before its own copy; sbx_memcpy_d2h — and must get the finished order every time,
    call after call (the window is a ~30 us kernel: 30 rounds)."""
    import ctypes as C
    from sparsebase_amd import capi
    lib = capi.load()
    hip = C.CDLL("libamdhip64.so")
    hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
    hip.hipEventCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
    hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
    hip.hipStreamWaitEvent.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    hip.hipEventDestroy.argtypes = [C.c_void_p]
    hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
    NONBLOCKING, D2H = 1, 2
    rp, col = synth.rmat_symmetric(18, 12, seed=21)   # one big component: the deferred kernel writes most of the order
    n = len(rp) - 1
    want = oracle.rcm_reorder(rp, col)
    d_rp, d_col = dev(rp), dev(col)
    d_inv = torch.empty(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    s_user, s_other, ev = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert hip.hipStreamCreateWithFlags(C.byref(s_user), NONBLOCKING) == 0
    assert hip.hipStreamCreateWithFlags(C.byref(s_other), NONBLOCKING) == 0
    assert hip.hipEventCreateWithFlags(C.byref(ev), 2) == 0   # hipEventDisableTiming
    h = C.c_void_p()
    assert lib.sbx_create(0, C.byref(h)) == 0
    try:
        assert lib.sbx_set_stream(h, s_user) == 0
        got = np.empty(n, np.int32)
        p = lambda t: C.c_void_p(t.data_ptr())
        for rnd in range(30):
            mode = rnd % 3
            assert hip.hipMemsetAsync(p(d_inv), 0xFF, n * 4, s_user) == 0   # (on the handle's stream: ordered before the call)
            st = capi.RcmStats()
            rc = lib.sbx_rcm_reorder(h, 0, n, len(col), p(d_rp), p(d_col), p(d_inv), C.byref(st))
            assert rc == 0, lib.sbx_last_error(h)
            assert st.largest_component > n // 2
            got[:] = -2
            if mode == 0:
                assert lib.sbx_sync(h) == 0
                assert hip.hipMemcpy(got.ctypes.data_as(C.c_void_p), p(d_inv), n * 4, D2H) == 0
            elif mode == 1:
                assert hip.hipEventRecord(ev, s_user) == 0
                assert hip.hipStreamWaitEvent(s_other, ev, 0) == 0
                assert hip.hipMemcpyAsync(got.ctypes.data_as(C.c_void_p), p(d_inv), n * 4, D2H, s_other) == 0
                assert hip.hipStreamSynchronize(s_other) == 0
            else:
                assert lib.sbx_memcpy_d2h(h, got.ctypes.data_as(C.c_void_p), p(d_inv), n * 4) == 0
            assert np.array_equal(got, want), (rnd, mode, int((got != want).sum()))
    finally:
        lib.sbx_sync(h)
        lib.sbx_destroy(h)
        hip.hipEventDestroy(ev)
        hip.hipStreamDestroy(s_other)
        hip.hipStreamDestroy(s_user)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["rmat", "rect_dups", "empty_rows"])
def test_mixed_width_tuple_int32_ids_int64_offsets(ops, oracle, case):
    """SBX_I32_N64 — 32-bit ids with 64-bit offsets, the tuple the reference pre-instantiates for matrices with fewer than
    2^31 rows and more offsets than an int holds (CMakeLists.txt:15-17: <int | unsigned int, long long | unsigned long
    long, V>): every entry point that takes or writes row_ptr / col_ptr against the oracle's 32-bit run with the offsets
    widened.  COO <-> CSR, DegreeReorder and the degree features run the 64-bit offsets natively, the rest through the
    narrowing adapters (sbx_i64.hip)."""
    if case == "rmat":
        rp, col = synth.rmat_symmetric(12, 8, seed=31)
        n = m = len(rp) - 1
    elif case == "rect_dups":
        n, m = 3000, 5000
        rp, col = synth.random_rect_csr(n, m, 60000, 8, dup_frac=0.05)
    else:
        n = m = 4096
        rp, col = synth.random_rect_csr(n, m, 9000, 12)
        lens = np.diff(rp)
        lens[::3] = 0                                   # a third of the rows empty
        keep = np.repeat(lens > 0, np.diff(rp))
        col = col[keep]
        rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    nnz = len(col)
    val = (np.arange(nnz) % 97).astype(np.float32)
    rp64 = rp.astype(np.int64)
    d_rp, d_col, d_val = dev(rp64), dev(col), dev(val)
    assert d_rp.dtype == torch.int64 and d_col.dtype == torch.int32

    # ---- conversions (native 64-bit offsets)
    row, c2, v2 = oracle.csr_to_coo(rp, col, val)
    g_row, g_c, g_v = ops.csr_to_coo(n, m, d_rp, d_col, d_val)
    assert g_row.dtype == torch.int32
    same((g_row, g_c, g_v), (row, c2, v2))
    g_rp, g_c, g_v = ops.coo_to_csr(n, m, dev(row), dev(c2), dev(v2), offset_dtype=torch.int64)
    assert g_rp.dtype == torch.int64 and np.array_equal(host(g_rp), rp64)
    same((g_c, g_v), (col, val))
    assert np.array_equal(host(ops.coo_to_csr(n, m, dev(row), dev(c2), None, move=True, rows_sorted=True,
                                              offset_dtype=torch.int64)[0]), rp64)
    if nnz > 1:  # unsorted row[] (ignore_sort): exclusive scan of the histogram (converter_order_two.cc:180-192)
        p = synth.random_permutation(nnz, 3)
        w = oracle.coo_to_csr(n, row[p], c2[p], v2[p])
        g = ops.coo_to_csr(n, m, dev(row[p]), dev(c2[p]), dev(v2[p]), offset_dtype=torch.int64)
        assert np.array_equal(host(g[0]), w[0].astype(np.int64))
        same(g[1:], w[1:])
    # ---- CSC (col_ptr is the 64-bit array)
    w = oracle.csr_to_csc(m, rp, col, val)
    g = ops.csr_to_csc(n, m, d_rp, d_col, d_val)
    assert g[0].dtype == torch.int64 and np.array_equal(host(g[0]), w[0].astype(np.int64))
    same(g[1:], w[1:])
    g = ops.coo_to_csc(n, m, dev(row), dev(c2), dev(v2), offset_dtype=torch.int64)
    assert np.array_equal(host(g[0]), w[0].astype(np.int64))
    same(g[1:], w[1:])
    # ---- checks, constructor sort, features
    assert ops.csr_rows_sorted(d_rp, d_col) == bool(oracle.csr_rows_sorted(rp, col))
    g = np.random.default_rng(5)
    scr = col.copy()
    for r in range(n):  # shuffle inside the rows
        a, b = rp[r], rp[r + 1]
        scr[a:b] = scr[a:b][g.permutation(b - a)]
    if case != "rect_dups":  # (distinct columns: the sorted row is unique)
        sv = (np.arange(nnz) % 53).astype(np.float32)
        w = oracle.csr_sort_rows(rp, scr, sv, m=m)
        sc, svd = dev(scr), dev(sv)
        ops.csr_sort_rows_(n, m, d_rp, sc, svd)
        same((sc, svd), w)
    same((ops.csr_degrees(d_rp, id_dtype=torch.int32),), (oracle.csr_degrees(rp),))
    assert ops.csr_bandwidth(d_rp, d_col) == oracle.csr_bandwidth(rp, col)
    assert ops.csr_profile(d_rp, d_col) == oracle.csr_profile(rp, col)
    dist = host(ops.csr_degree_distribution(d_rp, nnz, torch.float64))
    assert np.array_equal(dist, np.diff(rp).astype(np.float64) / np.float64(nnz))
    # ---- reorderers: the inverse permutations are id arrays (32-bit)
    for asc in (True, False):
        inv = ops.degree_reorder(d_rp, asc, id_dtype=torch.int32)
        assert inv.dtype == torch.int32 and np.array_equal(host(inv), oracle.degree_reorder(rp, asc))
    if n == m:
        if case == "rmat":  # (RCM is defined for symmetric patterns; others are refused)
            order = ops.rcm_reorder(d_rp, d_col)
            assert order.dtype == torch.int32
            assert np.array_equal(host(order), oracle.rcm_reorder(rp, col))
        deg, key, counts = ops.gray_row_keys(m, d_rp, d_col, 32, 10)
        wd, wk, wc = oracle.gray_row_keys(rp, col, m, 32, 10)
        assert np.array_equal(host(deg), wd) and np.array_equal(host(key).view(np.uint64), wk) and list(counts) == wc.tolist()
        inv = host(ops.gray_reorder(m, d_rp, d_col, 32, 10, 4))
        assert np.array_equal(inv, host(ops.gray_reorder(m, dev(rp), d_col, 32, 10, 4)))   # = the SBX_I32 call
    # ---- permutes: row_ptr_out is 64-bit
    ro = synth.random_permutation(n, 21)
    co = synth.random_permutation(m, 22)
    for row_o, col_o in ((ro, co), (ro, None), (None, co)):
        w = oracle.permute_csr(rp, col, val, row_o, col_o, m=m)
        g = ops.permute_csr(n, m, d_rp, d_col, d_val, dev(row_o), dev(col_o))
        assert g[0].dtype == torch.int64 and np.array_equal(host(g[0]), w[0].astype(np.int64))
        same(g[1:], w[1:])
    lo, hi = n // 3, (2 * n) // 3
    w = oracle.permute_csr(rp, col, val, ro, co, m=m)
    g = ops.permute_csr_rows(n, m, d_rp, d_col, d_val, dev(ro), dev(co), lo, hi)
    a, b = w[0][lo], w[0][hi]
    assert np.array_equal(host(g[0]), (w[0][lo:hi + 1] - a).astype(np.int64))
    same(g[1:], (w[1][a:b], w[2][a:b]))
    # ---- nnz >= 2^31 is refused where the offsets inside are 32-bit (before anything is touched)
    import ctypes as C
    hd = ops.handle_for(d_rp.device)
    p = lambda t: C.c_void_p(t.data_ptr())
    out_rp = torch.empty_like(d_rp)
    rc = hd.lib.sbx_permute_csr(hd.h, 2, 3, n, m, 1 << 31, p(d_rp), p(d_col), p(d_val), p(dev(ro)), p(dev(co)), p(out_rp),
                                p(torch.empty_like(d_col)), p(torch.empty_like(d_val)))
    assert rc == 5, rc  # SBX_ERR_UNSUPPORTED


@pytest.mark.gpu
@pytest.mark.parametrize("idt", [np.int32, np.int64])
def test_permute_tile2_boundaries(ops, oracle, idt):
    """k_permute_tile2 (the tile kernel of the relabelling permutes, round 6) at its edges: ids of exactly
    CDF_MAX_COL_BITS = 25 bits (the last width it takes) and one more (the equal-width kernel takes over: same result),
    a window of thousands of empty rows between short ones (every position searches for its row), rows of 1 ... 8
    entries (one bucket), of exactly 128 (the longest a tile holds) and 129 (the first row class), duplicate columns
    with values, every value width, and a column map that sends all of a row's columns into one bin of the
    key-distribution map (consecutive new ids: the ranking loop runs through the whole row)."""
    g = np.random.default_rng(77)
    for m in ((1 << 25), (1 << 25) + 1, 5000):
        n = 9000
        lens = g.integers(0, 12, n)
        lens[100:6000] = 0                                  # a window of empty rows
        lens[[10, 11, 12, 13]] = (128, 129, 127, 1)
        lens[6100:6200] = g.integers(100, 129, 100)
        rp = np.concatenate([[0], np.cumsum(lens)]).astype(idt)
        w = min(4000, m // 2)
        cols = [np.sort(g.choice(w, l, replace=False)) for l in lens]
        col = (np.concatenate(cols) if len(cols) else np.zeros(0)).astype(idt)
        nnz = len(col)
        # column map: the ids below w go to CONSECUTIVE new ids in the middle of the range (one bin of the map)
        co = np.arange(m, dtype=np.int64)
        base = m // 2
        co[:w], co[base:base + w] = np.arange(base, base + w), np.arange(w)
        co = co.astype(idt)
        ro = synth.random_permutation(n, 5, idt)
        for val in ((np.arange(nnz) % 31).astype(np.float32), None, g.random(nnz), (np.arange(nnz) % 7).astype(np.int64)):
            want = oracle.permute_csr(rp, col, val, ro, co, m=m)
            same(ops.permute_csr(n, m, dev(rp), dev(col), dev(val), dev(ro), dev(co)), want)
    # duplicate columns inside short rows, values decide (csr.cc:143-156)
    n = m = 3000
    lens = g.integers(2, 40, n)
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(idt)
    col = np.concatenate([np.sort(g.integers(0, 50, l)) for l in lens]).astype(idt)   # 50 distinct ids: many duplicates
    val = g.integers(0, 5, len(col)).astype(np.float32)
    ro, co = synth.random_permutation(n, 6, idt), synth.random_permutation(m, 7, idt)
    same(ops.permute_csr(n, m, dev(rp), dev(col), dev(val), dev(ro), dev(co)), oracle.permute_csr(rp, col, val, ro, co, m=m))


def test_page_locked_host_blocks_and_the_oom_hook_through_the_c_abi(oracle):
    """include/sbx.h: sbx_host_alloc / sbx_host_free hand out page-locked host blocks (the download targets of the host
    layer: a pageable target is pinned and unpinned by the runtime around every copy, and the unpinning holds the
    process's next submission back for milliseconds — NOTES section 5-r6); sbx_set_oom_hook: when an allocation of the
    library's own fails, the hook is asked ONCE to give memory back, the allocation is tried again, and the entry point
    returns SBX_ERR_OOM if that fails too — with the handle still usable.  The failing request is one no GPU can serve
    (2^46 bytes): nothing is allocated by it."""
    import ctypes as C
    from sparsebase_amd import capi
    lib = capi.load()
    h = C.c_void_p()
    assert lib.sbx_create(0, C.byref(h)) == 0
    calls = []
    def hook(user, wanted):
        calls.append(int(wanted))
        return 1   # "something was freed": the library tries once more
    cb = capi.OOM_HOOK_FN(hook)
    try:
        rp, col = synth.rmat_symmetric(14, 6, seed=9)
        n = len(rp) - 1
        d_rp, d_col = dev(rp), dev(col)
        d_deg = torch.empty(n, dtype=torch.int32, device="cuda")
        p = lambda t: C.c_void_p(t.data_ptr())
        # a page-locked block as the target of the library's own download
        blk = C.c_void_p()
        assert lib.sbx_host_alloc(h, n * 4, C.byref(blk)) == 0 and blk.value
        assert lib.sbx_csr_degrees(h, 0, n, p(d_rp), p(d_deg)) == 0, lib.sbx_last_error(h)
        assert lib.sbx_memcpy_d2h(h, blk, p(d_deg), n * 4) == 0
        got = np.ctypeslib.as_array(C.cast(blk, C.POINTER(C.c_int32)), shape=(n,)).copy()
        assert np.array_equal(got, np.diff(rp).astype(np.int32))
        assert lib.sbx_host_free(h, blk) == 0
        assert lib.sbx_host_free(h, None) == 0   # (like free(NULL))
        # the hook
        assert lib.sbx_set_oom_hook(h, cb, None) == 0
        rc = lib.sbx_reserve(h, 1 << 46)
        assert rc == 4, (rc, lib.sbx_last_error(h))   # SBX_ERR_OOM
        assert len(calls) == 1 and calls[0] >= (1 << 46), calls
        # the handle goes on working (its arena was given up by the failed reservation: the next call grows a new one)
        d_inv = torch.empty(n, dtype=torch.int32, device="cuda")
        st = capi.RcmStats()
        assert lib.sbx_rcm_reorder(h, 0, n, len(col), p(d_rp), p(d_col), p(d_inv), C.byref(st)) == 0, lib.sbx_last_error(h)
        assert lib.sbx_sync(h) == 0
        assert np.array_equal(d_inv.cpu().numpy(), oracle.rcm_reorder(rp, col))
        assert len(calls) == 1
        # without a hook: the same error, nobody asked
        assert lib.sbx_set_oom_hook(h, C.cast(None, capi.OOM_HOOK_FN), None) == 0
        assert lib.sbx_reserve(h, 1 << 46) == 4 and len(calls) == 1
    finally:
        lib.sbx_sync(h)
        lib.sbx_destroy(h)


def test_gray_device_ordering_one_sort_equals_three_sorts(ops):
    """sbx_gray_reorder orders (class / section, signed key, degree, row id) with ONE radix sort over a composite key where
    the fields fit 64 bits, with three stable sorts otherwise (and under SBX_GRAY_ORDER_THREE_SORTS=1, read once per
    process: hence the child): the same inverse permutation either way, on power-law, banded and wide-band inputs."""
    import hashlib
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, hashlib, numpy as np, torch; sys.path[:0] = [%r]\n"
        "from sparsebase_amd import ops, synth\n"
        "d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()\n"
        "cases = [(synth.rmat_symmetric(15, 9, seed=4), (32, 10, 4)), (synth.rmat_symmetric(15, 9, seed=4), (16, 20, 2)),\n"
        "         (synth.rmat_symmetric(14, 5, seed=6), (8, 3, 1)), (synth.banded_symmetric(20000, 40, 9, 3), (32, 10, 4)),\n"
        "         (synth.banded_symmetric(20000, 3000, 9, 4), (16, 0, 1)), (synth.rmat_symmetric(14, 5, seed=6), (32, -1, 3))]\n"
        "for (rp, col), (res, thr, grp) in cases:\n"
        "    n = len(rp) - 1\n"
        "    for wide in (False, True):\n"
        "        a, b = (d(rp).long(), d(col).long()) if wide else (d(rp), d(col))\n"
        "        inv = ops.gray_reorder(n, a, b, res, thr, grp).cpu().numpy().astype(np.int64)\n"
        "        print('digest', hashlib.sha256(inv.tobytes()).hexdigest())\n"
        % (root,))
    outs = []
    for extra in ({}, {"SBX_GRAY_ORDER_THREE_SORTS": "1"}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **extra), capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append([l for l in r.stdout.splitlines() if l.startswith("digest")])
    assert len(outs[0]) == 12 and outs[0] == outs[1]
