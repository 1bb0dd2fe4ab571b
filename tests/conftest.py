import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle_py import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reflib():
    from oracle.oracle_py import RefLib
    if not RefLib.available():
        try:
            return RefLib()
        except Exception:
            pytest.skip("oracle/_ref/libAssemblyEnv.so not built (reference sources absent)")
    return RefLib()


@pytest.fixture(scope="session")
def shapes():
    from marl_llm_amd.shapes import synthetic_shape_set
    return synthetic_shape_set()
