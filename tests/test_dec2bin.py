"""CPU unit test of the exact decimal -> binary conversion used by the device Matrix Market parser
(sparsebase_amd/csrc/sbx_dec2bin.h compiled for the host) against strtod / strtof."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dec2bin_matches_strtod_and_strtof(tmp_path):
    exe = str(tmp_path / "dec2bin_check")
    subprocess.run(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "sparsebase_amd", "csrc"),
                    os.path.join(ROOT, "tests", "native", "dec2bin_check.cc"), "-o", exe], check=True)
    for seed in (1, 2):
        p = subprocess.run([exe, "400000", str(seed)], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0 and p.stdout.startswith("ok"), p.stdout[-2000:]
