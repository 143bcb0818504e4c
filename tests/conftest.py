import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from orc import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    from orc import Ref, ref_available
    if os.environ.get("SBX_ORACLE_LIB"):
        # make -C oracle asan: the sanitizer runtime is preloaded and would abort inside the REAL reference, which does
        # overflow (degree_reorder.cc:41-45, gray_reorder.cc:251, converter_order_two.cc:32): those cases stay with the plain run
        pytest.skip("sanitizer run: the real reference is exercised by the plain run only")
    if not ref_available():
        pytest.skip("oracle/_ref/libsbref.so not built (needs /root/reference)")
    return Ref()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
