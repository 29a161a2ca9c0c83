import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bwa-mem_gpu_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_py
    oracle_py.build(ref=True)
    return oracle_py.Oracle()


@pytest.fixture(scope="session")
def ref():
    import oracle_py
    oracle_py.build(ref=True)
    if not oracle_py.Ref.available():
        pytest.skip("oracle/_ref/libref.so not built (needs /root/reference)")
    return oracle_py.Ref()
