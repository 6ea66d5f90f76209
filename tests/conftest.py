import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_artifacts_built():
    """The .so files are git-ignored (they travel with the gpurun snapshot but not with a fresh checkout): build what is
    missing or stale before the first test.  hipcc cross-compiles gfx950 without a GPU; gcc builds the oracle."""
    from materialist_amd import build as _b

    if _b.is_stale():
        _b.build_library()
    from oracle import oracle as _o

    _o.build()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle64():
    import numpy as np
    from oracle.oracle import Oracle

    return Oracle(np.float64)


@pytest.fixture(scope="session")
def oracle32():
    import numpy as np
    from oracle.oracle import Oracle

    return Oracle(np.float32)
