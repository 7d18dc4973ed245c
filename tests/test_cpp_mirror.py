"""The C++ host-side mirror (bindings/cpp/hare.hpp) compiles against include/hare_hip.h, links the
C-ABI library, and behaves like the reference interface.  The single-ray Shoot runs on the host (hare_shoot_one)
and works everywhere; the batch Shoot is GPU-only and must throw without a GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(tmp_path):
    exe = str(tmp_path / "hare_example")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "bindings", "cpp"), os.path.join(ROOT, "bindings", "cpp", "example.cpp"),
                           "-L", os.path.join(ROOT, "hare_amd"), "-lhare_hip", "-Wl,-rpath," + os.path.join(ROOT, "hare_amd"),
                           "-o", exe])
    return exe


def test_cpp_mirror_compiles_and_fails_loudly_without_gpu(tmp_path, gpu_available):
    exe = build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert "Char_Step = 0.5505" in r.stdout             # host build of the grid works everywhere
    assert "hit poly" in r.stdout and "t = 1.5 " in r.stdout and "(2.000, 0.750, 1.000)" in r.stdout   # so does one ray
    if not gpu_available:
        assert r.returncode == 2 and "no HIP device visible" in r.stdout and "batch:" not in r.stdout


@pytest.mark.gpu
def test_cpp_mirror_shoots_on_gpu(tmp_path):
    exe = build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "hit poly" in r.stdout and "t = 1.5 " in r.stdout and "(2.000, 0.750, 1.000)" in r.stdout
    assert "batch: 2 hits, t = 1.5 and 1" in r.stdout
    assert "bounce: 6 hits, ray 0 t = 1.5, 2, 2; cast 2: 2 rays" in r.stdout
