import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# oracle/ is test infrastructure: importable from tests only.
sys.path.insert(0, os.path.join(ROOT, "oracle"))
# the drop-in package (same module names as the reference: plspm.config, plspm.plspm, ...)
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd"))
sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
