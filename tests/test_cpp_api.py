"""The C++20 drop-in headers (include/radar/) compile with g++ -std=c++20, link against librmr.so
and behave like the reference's classes: constructors throw, hot path works on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "rm_radar_amd", "_build", "api_smoke")


def _build():
    import __graft_entry__ as g
    g.build()
    libdir = os.path.join(ROOT, "rm_radar_amd", "_build")
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "api_smoke.cpp"), "-L", libdir, "-lrmr",
                           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", EXE])


def _run():
    return subprocess.run([EXE], capture_output=True, text=True, timeout=300)


def test_cpp_api_compiles_and_fails_loudly_without_gpu():
    _build()
    import rm_radar_amd as r
    if r.device_count() > 0:
        pytest.skip("a GPU is present: covered by the gpu test")
    res = _run()
    assert res.returncode == 0, res.stdout + res.stderr
    assert "api_smoke ok" in res.stdout


@pytest.mark.gpu
def test_cpp_api_on_gpu():
    _build()
    res = _run()
    assert res.returncode == 0, res.stdout + res.stderr
    assert "located at" in res.stdout
