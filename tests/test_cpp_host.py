"""The C++ host mirror (include/fx_realtime.hpp) compiled with g++ against libfx_hip.so."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(fx, tmp_path):
    fx.load_library()
    exe = str(tmp_path / "host_mirror")
    lib_dir = os.path.dirname(fx.library_path())
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "host_mirror.cpp"), "-o", exe,
                           "-L", lib_dir, "-lfx_hip", "-Wl,-rpath," + lib_dir])
    return exe


def test_cpp_mirror_cpu(fx, tmp_path):
    exe = _build(fx, tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.gpu
def test_cpp_mirror_gpu(gpu_fx, tmp_path):
    exe = _build(gpu_fx, tmp_path)
    out = subprocess.run([exe, "--gpu"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
