import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SCENES = os.path.join(ROOT, "scenes")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    # No valid frame of this suite comes near 2^23 rounds per wave (a 4K frame takes a few hundred): a kernel that would
    # never end is cut off after seconds and the test fails with RTX_E_LIMIT instead of hanging the GPU box.  Set before
    # any test module builds a renderer of its own, not by the `renderer` fixture.
    import raytracer_amd as rt
    rt.api.DEFAULTS["round_limit"] = 1 << 23


def scene_path(name):
    return os.path.join(SCENES, name + ".test")


@pytest.fixture(scope="session")
def renderer():
    """One GPU context for the whole session (GPU tests only)."""
    import raytracer_amd as rt
    r = rt.Renderer(0)
    yield r
    r.close()
