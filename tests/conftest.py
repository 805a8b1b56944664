import pathlib
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return GOLDEN


@pytest.fixture(scope="session")
def kat():
    import json
    return json.loads((GOLDEN / "kat.json").read_text())


@pytest.fixture(scope="session")
def oracle():
    from oracle import kt_oracle
    kt_oracle.lib()
    return kt_oracle
