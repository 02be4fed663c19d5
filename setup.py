"""Packaging counterpart of the reference's setup.py (reference setup.py:10-40): `pip install .` builds the native library by
shelling out to the package's Makefile (hipcc, gfx950) and installs it next to the `flash_attention_amd` package, where the
loader looks for it (`<pkg-parent>/libflash_attention<EXT_SUFFIX>`, flash_attention_amd/_impl/_native.py - the reference's
second search location, flash_attention_ipu/__init__.py:13-17).

    pip install --no-build-isolation .        # (no network in this image: build with the installed setuptools)
    python -c "import flash_attention_amd, flash_attention_amd.auto"

FA_AMD_OBJDIR=<dir> reuses an existing object directory (e.g. flash-attention-ipu_amd/build/obj) instead of compiling afresh.
"""

import os
import subprocess
from pathlib import Path

import setuptools
import setuptools.command.build_ext

ROOT = Path(__file__).resolve().parent
PKG = ROOT / "flash-attention-ipu_amd"


class make_ext(setuptools.command.build_ext.build_ext):
    def build_extension(self, ext: setuptools.Extension) -> None:
        if ext.name == "libflash_attention":
            filename = (Path(self.build_lib) / self.get_ext_filename(self.get_ext_fullname(ext.name))).resolve()
            objdir = Path(os.environ.get("FA_AMD_OBJDIR") or filename.with_suffix("")).resolve()
            filename.parent.mkdir(parents=True, exist_ok=True)
            subprocess.check_call(["make", "-C", str(PKG), f"-j{os.cpu_count() or 4}", f"OUT={filename}", f"OBJDIR={objdir}"])
        else:
            super().build_extension(ext)


setuptools.setup(
    name="flash-attention-mi355",
    version="0.3.0",
    description="FlashAttention for AMD MI355X (gfx950) behind the flash-attention-ipu operator API",
    python_requires=">=3.9",
    install_requires=["torch"],
    package_dir={"": "flash-attention-ipu_amd"},
    packages=["flash_attention_amd", "flash_attention_amd._impl"],
    ext_modules=[
        setuptools.Extension(
            "libflash_attention",
            [str(p.relative_to(ROOT)) for p in sorted((PKG / "csrc").glob("*")) if p.suffix in (".hip", ".cpp", ".h")],
        )
    ],
    cmdclass=dict(build_ext=make_ext),
)
