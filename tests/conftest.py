import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import __graft_entry__ as entry  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    return entry.load_oracle()


@pytest.fixture(scope="session")
def oracle(orc):
    return orc.Oracle()


@pytest.fixture(scope="session")
def reference(orc):
    if not orc.Reference.available() and os.path.isdir("/root/reference"):
        # build container: compile the partial reference build from the sources where they lie
        import subprocess
        subprocess.call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL,
                        stderr=subprocess.DEVNULL)
    if not orc.Reference.available():
        pytest.skip("oracle/_ref/libdmzref.so not built (needs /root/reference, build container only)")
    return orc.Reference()


@pytest.fixture(scope="session")
def pkg():
    p = entry.load_package()
    p.build()  # hipcc cross-compiles without a GPU; a no-op when libdmz_hip.so is already there
    return p


@pytest.fixture(scope="session")
def ctx(pkg):
    c = pkg.Context(0)  # raises if there is no GPU or no built library: no CPU fallback
    yield c
    c.close()
