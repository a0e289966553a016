import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle_lib import Oracle

    return Oracle()


@pytest.fixture(scope="session")
def ref_stages():
    from oracle_lib import RefStages

    r = RefStages()
    if not r.available:
        pytest.skip("oracle/_ref not built (reference tree absent)")
    return r


@pytest.fixture(scope="session")
def ref_lib():
    from oracle_lib import Bz3, RefLib

    r = RefLib()
    if not r.available:
        pytest.skip("oracle/_ref not built (reference tree absent)")
    return Bz3(r.lib)


@pytest.fixture(scope="session")
def text():
    import datagen

    return datagen.shakespeare()


@pytest.fixture(scope="session")
def gpu_lib():
    """The product library on a real GPU.  Fails (never skips silently) if the extension is missing."""
    import bzip3_amd

    lib = bzip3_amd.load()
    assert lib.bz3_hip_device_count() > 0, "no HIP device visible: -m gpu tests must run on the MI355X box"
    return lib
