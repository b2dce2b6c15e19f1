"""pip-installable form of the drop-in (round-5 verdict, next #9; the reference ships `toy-hgemm` as a wheel:
kernels/hgemm/setup.py:11,48, tools/install.sh).

    pip install . --no-build-isolation        (no network: setuptools / torch of this environment are used as they are)
    python setup.py bdist_wheel

builds libleetcuda_amd.so (hipcc, gfx950) and the two torch extension modules with leetcuda_amd.build, and installs
    leetcuda_amd/                      the package (capi.py, host.py, dist.py, build.py, lib/libleetcuda_amd.so, include/lc_abi.h)
    toy_hgemm.<abi>.so                 TOP-LEVEL modules, the names the reference benches import (kernels/hgemm/tools/utils.py:131,
    flash_attn_lib.<abi>.so            kernels/flash-attn/flash_attn_mma.py:222-228): no PYTHONPATH needed afterwards
The extension modules find the C-ABI library through their rpath ($ORIGIN/lib in-tree, $ORIGIN/leetcuda_amd/lib once installed)."""
import shutil
import sys
import sysconfig
from pathlib import Path

from setuptools import Distribution, setup
from setuptools.command.build_py import build_py
from setuptools.command.install import install

ROOT = Path(__file__).resolve().parent
EXT_MODULES = ("toy_hgemm", "flash_attn_lib")


class BinaryDistribution(Distribution):
    def has_ext_modules(self):   # platform wheel: it carries gfx950 code objects and CPython-ABI modules
        return True


class InstallPlat(install):
    def finalize_options(self):   # everything (the Python files too) goes where the binaries go: one tree in site-packages
        super().finalize_options()
        self.install_lib = self.install_platlib


class BuildNative(build_py):
    def run(self):
        sys.path.insert(0, str(ROOT))
        from leetcuda_amd import build as lcbuild
        lcbuild.build_abi(False)
        lcbuild.build_torch_ext(False)
        super().run()
        suffix = sysconfig.get_config_var("EXT_SUFFIX") or ".so"
        out = Path(self.build_lib)
        (out / "leetcuda_amd" / "lib").mkdir(parents=True, exist_ok=True)
        shutil.copy2(ROOT / "leetcuda_amd" / "lib" / "libleetcuda_amd.so", out / "leetcuda_amd" / "lib" / "libleetcuda_amd.so")
        (out / "leetcuda_amd" / "include").mkdir(parents=True, exist_ok=True)
        shutil.copy2(ROOT / "include" / "lc_abi.h", out / "leetcuda_amd" / "include" / "lc_abi.h")
        for name in EXT_MODULES:
            shutil.copy2(ROOT / "leetcuda_amd" / f"{name}{suffix}", out / f"{name}{suffix}")


setup(
    name="leetcuda-amd",
    version="0.6.0",
    description="MI355X-native HGEMM + FlashAttention-2 forward behind xlite-dev/LeetCUDA's toy_hgemm / flash_attn_lib entry points",
    packages=["leetcuda_amd"],
    python_requires=">=3.10",
    distclass=BinaryDistribution,
    cmdclass={"build_py": BuildNative, "install": InstallPlat},
    zip_safe=False,
)
