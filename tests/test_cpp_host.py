"""The C++ host mirror (include/h2r_chips.hpp): compiles on CPU (no GPU needed to build) and, on the GPU
box, runs the reference-style RSA tests in tests/cpp/test_rsa_chip.cpp."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_cpp_test():
    from halo2_rsa_amd import _build
    from oracle_lib import build as build_oracle
    from cpp_build import build_cpp
    _build.build_lib()
    build_oracle()
    return build_cpp("test_rsa_chip", extra_libs=[(os.path.join(ROOT, "oracle"), "h2r_oracle")])


def test_cpp_host_mirror_compiles():
    assert os.path.exists(build_cpp_test())


@pytest.mark.gpu
def test_cpp_host_mirror_runs_reference_style_tests():
    exe = build_cpp_test()
    out = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "rsa_kats_limbs.txt")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "CPP_HOST_MIRROR_OK 3" in out.stdout


def test_host_side_under_address_sanitizer(tmp_path):
    """SURVEY section 5's sanitizer build: the C oracle and a driver of every host-only export of libh2r.so, compiled with
    -fsanitize=address,undefined and run on host-only contexts (no GPU).  Any out-of-bounds access of the oracle, or any
    memcpy / memset of the library past a caller buffer, aborts the run."""
    from halo2_rsa_amd import _build
    _build.build_lib()
    exe = str(tmp_path / "test_host_asan")
    oracle_o = str(tmp_path / "h2r_oracle_asan.o")
    san = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-g", "-O1"]
    subprocess.check_call(["gcc", "-std=gnu11"] + san + ["-c", os.path.join(ROOT, "oracle", "h2r_oracle.c"), "-o", oracle_o])
    subprocess.check_call(["g++", "-std=c++17"] + san + ["-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "test_host_asan.cpp"),
                           oracle_o, "-o", exe, "-L" + os.path.join(ROOT, "halo2_rsa_amd", "lib"), "-lh2r", "-lpthread",
                           "-Wl,-rpath," + os.path.join(ROOT, "halo2_rsa_amd", "lib"), "-Wl,-rpath,/opt/rocm/lib"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    assert "ASAN_HOST_OK 5 shapes" in out.stdout
