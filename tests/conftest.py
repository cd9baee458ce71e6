"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path."""
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "culling: the oracle's 8-wide search steps over the back of one-sided triangles, as by default (tests/test_wide8_cpu.py turns it off otherwise)")


@pytest.fixture(scope="session")
def oracle():
    from oracle_bindings import get_oracle
    return get_oracle(False)


@pytest.fixture(scope="session")
def goldens():
    import json
    return json.loads((ROOT / "tests" / "golden" / "reference_goldens.json").read_text())
