import pathlib
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "flash-attention-ipu_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return ROOT / "tests" / "golden"


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The C-ABI library must exist (built by `__graft_entry__.build()` / make); build it if absent."""
    lib = ROOT / "flash-attention-ipu_amd" / "build" / "libflash_attention.so"
    if not lib.exists():
        import __graft_entry__

        __graft_entry__.build()
    return lib
