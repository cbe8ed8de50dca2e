"""The header-only C++ mirror of the reference's API (include/triple_accel.hpp): compiles against the C ABI with plain g++,
fails loudly without a GPU (not-gpu), and reproduces the reference's documented answers on one (gpu)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "build", "mirror_check")


def build():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    libdir = os.path.join(ROOT, "triple_accel_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "mirror_check.cpp"), "-o", EXE,
                           "-L", libdir, "-ltriple_accel_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return EXE


def test_cpp_mirror_builds_and_refuses_without_a_gpu():
    import torch
    exe = build()
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the no-device behaviour is checked on the CPU-only box")
    out = subprocess.run([exe, "nogpu"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "nogpu: ok" in out.stdout


@pytest.mark.gpu
def test_cpp_mirror_known_answers():
    exe = build()
    out = subprocess.run([exe, "gpu"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "gpu: ok" in out.stdout
