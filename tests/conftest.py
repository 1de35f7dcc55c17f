import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return ROOT / "tests" / "golden"


@pytest.fixture(autouse=True)
def _fresh_settings():
    """The package's MTQ_* switches are read once per process (quantization_analysis_amd/settings.py): every test starts from, and
    leaves behind, what the environment says (tests that monkeypatch a switch call settings(refresh=True) themselves)."""
    from quantization_analysis_amd.settings import settings

    settings(refresh=True)
    yield
    settings(refresh=True)
