import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

MODELS = os.path.join(ROOT, "tests", "golden", "models")
AUDIO = os.path.join(ROOT, "tests", "golden", "audio")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def model_path(name: str) -> str:
    return os.path.join(MODELS, name + ".nam")


@pytest.fixture(scope="session")
def nam_lib():
    """The built HIP library (hipcc cross-compiles on CPU; symbols can be checked without a GPU)."""
    import neuralampmodelercore_amd as nam
    if not os.path.exists(nam.lib_path()):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "neuralampmodelercore_amd", "csrc")])
    nam.load_library()
    return nam


@pytest.fixture(scope="session")
def oracle():
    import nam_oracle
    nam_oracle.build()
    return nam_oracle
