import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import _oracle

    return _oracle.load_oracle()


@pytest.fixture(scope="session")
def ref():
    """The reference's own Boost-free translation units (oracle/_ref/libbtref.so); skip when not built."""
    import _oracle

    r = _oracle.load_ref()
    if r is None:
        pytest.skip("oracle/_ref/libbtref.so not built (needs /root/reference)")
    return r


@pytest.fixture(scope="session")
def gpu_ctx():
    from bayestyper_amd import lib

    ctx = lib.Ctx(0)
    yield ctx
    ctx.close()
