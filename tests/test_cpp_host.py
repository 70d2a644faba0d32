"""The C++ host mirror (include/h2r_chips.hpp): compiles on CPU (no GPU needed to build) and, on the GPU
box, runs the reference-style RSA tests in tests/cpp/test_rsa_chip.cpp."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_rsa_chip")


def build_cpp_test():
    from halo2_rsa_amd import _build
    from oracle_lib import build as build_oracle
    _build.build_lib()
    build_oracle()
    src = os.path.join(ROOT, "tests", "cpp", "test_rsa_chip.cpp")
    deps = [src, os.path.join(ROOT, "include", "h2r_chips.hpp"), os.path.join(ROOT, "include", "h2r.h")]
    if os.path.exists(EXE) and all(os.path.getmtime(EXE) >= os.path.getmtime(d) for d in deps):
        return EXE
    cmd = ["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
           src, "-o", EXE, "-L" + os.path.join(ROOT, "halo2_rsa_amd", "lib"), "-lh2r", "-L" + os.path.join(ROOT, "oracle"), "-lh2r_oracle",
           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + os.path.join(ROOT, "halo2_rsa_amd", "lib"),
           "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return EXE


def test_cpp_host_mirror_compiles():
    assert os.path.exists(build_cpp_test())


@pytest.mark.gpu
def test_cpp_host_mirror_runs_reference_style_tests():
    exe = build_cpp_test()
    out = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "rsa_kats_limbs.txt")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "CPP_HOST_MIRROR_OK 3" in out.stdout
