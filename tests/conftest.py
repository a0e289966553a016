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


def _prebuild_shared_artifacts():
    """Everything several tests build on first use, built once before the workers start (their builders are not made for
    concurrent callers): the product library (hipcc cross-compiles gfx950 without a GPU), the CPU checkers under oracle/, the
    emulator build of the kernel sources, the decoded fixture text."""
    from bzip3_amd.build import build as build_product

    build_product()
    from oracle_lib import build_oracle

    build_oracle()
    sys.path.insert(0, os.path.join(HERE, "emu"))
    from build_emu import build as build_emu

    build_emu()
    import datagen

    datagen.shakespeare()


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite (`-m "not gpu"`) is dominated by the kernel sources running under the fiber emulator, one test after the
    other: ten minutes on one core.  When pytest-xdist is installed the tests are spread over the machine's cores instead
    (BZ3_TEST_WORKERS=<n> picks the number, 0 or 1 keeps the run serial; an explicit -n wins).  The GPU suite is never
    distributed: its tests share one device and the opt-in machinery runs after the default-path tests."""
    if hasattr(config, "workerinput"):
        return None
    opt = config.option
    if getattr(opt, "collectonly", False) or not config.pluginmanager.hasplugin("xdist"):
        return None
    if getattr(opt, "numprocesses", None) is not None:  # an explicit -n: the workers still must not build side by side
        if opt.numprocesses:
            _prebuild_shared_artifacts()
        return None
    if (getattr(opt, "markexpr", "") or "").replace(" ", "") != "notgpu" or getattr(opt, "usepdb", False):
        return None
    env = os.environ.get("BZ3_TEST_WORKERS", "")
    workers = int(env) if env.isdigit() else min(os.cpu_count() or 1, 8)
    if workers < 2:
        return None
    _prebuild_shared_artifacts()
    opt.numprocesses = workers
    return None


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
