"""The CMake package (CMakeLists.txt, cmake/yak-config.cmake.in): target names of the reference's package
(reference yak/CMakeLists.txt:74-153: yak, yak_frontend, yak_marching_cubes under the namespace yak::), so that a
downstream find_package(yak) resolves unchanged.  Configures, builds (hipcc cross-compiles gfx950 without a GPU) and
installs into a scratch prefix, then builds and runs the out-of-tree consumer of examples/find_package against it."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("cmake") is None, reason="cmake not installed")
def test_find_package_yak_consumer_builds_and_runs(tmp_path):
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "check_cmake_package.sh"), str(tmp_path / "pkg")],
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "find_package(yak) consumer built and ran" in r.stdout
    # the package build runs the store-data hazard gate on its own assembly of integrate.hip, as make does
    assert "store-data hazard gate ran" in r.stdout and "0 problem(s)" in r.stdout
    assert (tmp_path / "pkg" / "build" / "integrate.hazard-checked").exists()
    cm = open(os.path.join(ROOT, "CMakeLists.txt")).read()
    assert "add_dependencies(yakhip yakhip_hazard_check)" in cm
    prefix = tmp_path / "pkg" / "prefix"
    for f in ("lib/libyak.so", "lib/libyak_frontend.so", "lib/libyak_marching_cubes.so", "lib/libyakhip.so",
              "lib/cmake/yak/yak-config.cmake", "lib/cmake/yak/yak-targets.cmake", "include/yak/yak_server.h", "include/yakhip.h"):
        assert (prefix / f).exists(), f
    targets = (prefix / "lib/cmake/yak/yak-targets.cmake").read_text()
    for t in ("yak::yak", "yak::yak_frontend", "yak::yak_marching_cubes"):
        assert t in targets
