"""Packaging counterpart of the reference's setup.py (reference setup.py:10-40, README.md:7 `pip install git+...`):
`pip install .` must give an importable `flash_attention_amd` whose loader finds the library pip installed next to it."""

import os
import pathlib
import shutil
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent


def test_pip_install_gives_an_importable_package(tmp_path, _built_library):
    target = tmp_path / "site"
    env = dict(os.environ)
    env["FA_AMD_OBJDIR"] = str(_built_library.parent / "obj")      # reuse the objects of the in-tree build: link only
    had_build = (ROOT / "build").exists()
    try:
        subprocess.check_call([sys.executable, "-m", "pip", "install", "--quiet", "--no-build-isolation", "--no-deps", "--no-index",
                               "--target", str(target), str(ROOT)], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    finally:
        if not had_build:
            shutil.rmtree(ROOT / "build", ignore_errors=True)
        for p in list(ROOT.glob("*.egg-info")) + list((ROOT / "flash-attention-ipu_amd").glob("*.egg-info")):
            shutil.rmtree(p, ignore_errors=True)
    assert (target / "flash_attention_amd" / "_impl" / "ops.py").exists()
    assert list(target.glob("libflash_attention*.so")), "the native library is installed next to the package"
    code = ("import flash_attention_amd, flash_attention_amd.auto, torch\n"
            "from flash_attention_amd._impl import _native\n"
            "import torch.nn.functional as F\n"
            f"assert str(_native.library_path()).startswith({str(target)!r}), _native.library_path()\n"
            "assert hasattr(F.scaled_dot_product_attention, '__wrapped__')\n"
            "out = flash_attention_amd.flash_attention_qkv_packed(torch.randn(3, 1, 128, 64), 4, 4)\n"
            "assert out.shape == (1, 128, 64)\n")
    env2 = {k: v for k, v in os.environ.items() if k != "FLASH_ATTENTION_AMD_LIB"}
    env2["PYTHONPATH"] = str(target)
    subprocess.check_call([sys.executable, "-c", code], env=env2, cwd=str(tmp_path))
