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


class _Knobs:
    """Forces knobs of the tile plan through fa_debug_set_option (the FA_AMD_* environment is read once at library
    load, never per launch); everything is back on "auto" after the test."""

    _NAMES = {"FA_AMD_KERNEL": "kernel", "FA_AMD_QB": "qb", "FA_AMD_WAVES": "waves", "FA_AMD_SCHED": "sched",
              "FA_AMD_SKIPMAX": "skipmax", "FA_AMD_F32": "f32_generic"}

    def __init__(self):
        from flash_attention_amd._impl import _native

        self._native = _native

    def set(self, name, value):
        name = self._NAMES.get(name, name)
        if name == "f32_generic" and isinstance(value, str):
            value = 1 if value.startswith("g") else 0
        if name != "kernel":
            value = int(value)
        self._native.set_debug_option(name, value)

    def reset(self, name=None):
        for k, v in self._native._OPTION_AUTO.items():
            if name is None or self._NAMES.get(name, name) == k:
                self._native.set_debug_option(k, v)


@pytest.fixture
def knobs():
    k = _Knobs()
    k.reset()
    yield k
    k.reset()
