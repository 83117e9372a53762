import pathlib
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: full-size CPU renders, not part of the default CPU suite")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def cornell():
    from raytracer_amd.scene_io import read_scene
    return read_scene(GOLDEN / "cornell.rtsc")


@pytest.fixture(scope="session")
def spheres():
    from raytracer_amd.scene_io import read_scene
    return read_scene(GOLDEN / "spheres.rtsc")
