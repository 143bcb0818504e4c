#!/usr/bin/env python3
"""Runs pytest in-process against a variant library (tools/build_variant.py): tools/pytest_with_lib.py <variant> <pytest args>."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparsebase_amd import capi
if sys.argv[1]:
    capi.LIB_PATH = os.path.join(ROOT, "sparsebase_amd", "lib", f"libsbx_{sys.argv[1]}.so")
import pytest
sys.exit(pytest.main(sys.argv[2:]))
