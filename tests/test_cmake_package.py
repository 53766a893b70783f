"""The CMake package: `find_package(altro 0.1 REQUIRED)` + `altro::altro`, as the reference's own
consumer project does it (examples/cmake/basic_cmake_project), against this repository -- installed into a scratch
prefix from the in-tree libaltro_hip.so (ALTRO_PREBUILT_LIB; the full hipcc build through CMake is the same rule set,
exercised by `cmake -S . -B build && cmake --build build`).  CPU only: the consumer programs do not touch the device."""
import os
import shutil
import subprocess

import pytest

import altro_amd
from altro_amd import build as hipbuild

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_reference_style_consumer_configures_builds_and_runs(tmp_path):
    if not shutil.which("cmake"):
        pytest.skip("cmake not on PATH")
    hipbuild.build()
    prefix, bdir, cdir = tmp_path / "prefix", tmp_path / "build", tmp_path / "consumer"

    def run(*cmd):
        p = subprocess.run(list(cmd), capture_output=True, text=True)
        assert p.returncode == 0, " ".join(cmd) + "\n" + p.stdout + p.stderr
        return p.stdout

    run("cmake", "-S", ROOT, "-B", str(bdir), "-DALTRO_PREBUILT_LIB=" + altro_amd.LIB_PATH)
    run("cmake", "--build", str(bdir))
    run("cmake", "--install", str(bdir), "--prefix", str(prefix))
    assert (prefix / "lib" / "libaltro_hip.so").exists()
    assert (prefix / "include" / "altro" / "altro.hpp").exists()
    assert (prefix / "lib" / "cmake" / "altro" / "altroConfig.cmake").exists()
    run("cmake", "-S", os.path.join(ROOT, "tests", "cmake_consumer"), "-B", str(cdir), "-DCMAKE_PREFIX_PATH=" + str(prefix))
    run("cmake", "--build", str(cdir))
    assert "Solver Initialized! (0)" in run(str(cdir / "main"))
    assert "tvlqr_TotalMemSize = " in run(str(cdir / "seam"))
