#!/usr/bin/env python3
"""Writes tests/golden/sbff/*.bin with the REAL reference's binary writers (oracle/_ref/libsbref.so).

Run in the build container only:  python oracle/make_golden_sbff.py
The files are data: the objects of the reference's own binary IO tests
(tests/suites/sparsebase/io/binary_{reader,writer}_order_{one,two}_tests.cc) as the reference stores them.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from orc import Ref  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "sbff")
os.makedirs(OUT, exist_ok=True)
ref = Ref()
i32, f32 = np.int32, np.float32
# binary_reader_order_two_tests.cc:7-36 (COO) and :38-70 (CSR)
ref.sbff_write_coo(os.path.join(OUT, "ref_coo.bin"), 4, 4, np.array([1, 2, 3, 4], i32), np.array([5, 6, 7, 8], i32),
                   np.array([0.1, 0.2, 0.3, 0.4], f32))
ref.sbff_write_coo(os.path.join(OUT, "ref_coo_pattern.bin"), 4, 4, np.array([1, 2, 3, 4], i32), np.array([5, 6, 7, 8], i32))
ref.sbff_write_csr(os.path.join(OUT, "ref_csr.bin"), 4, 4, np.array([0, 2, 3, 3, 4], i32), np.array([0, 2, 1, 3], i32),
                   np.array([0.1, 0.2, 0.3, 0.4], f32))
# binary_reader_order_one_tests.cc
ref.sbff_write_array(os.path.join(OUT, "ref_array.bin"), np.array([1.0, 2.0, 3.0, 4.0, 5.0], f32))
for f in sorted(os.listdir(OUT)):
    print(f, os.path.getsize(os.path.join(OUT, f)))
