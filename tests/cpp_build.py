"""Builds the C++ test programs under tests/cpp/ into tests/cpp/build/ (git-ignored), named by the SHA-256 of what they are compiled
from -- source, the headers of include/, the compile line -- so that a stale binary can never run: file times play no part."""
import glob
import hashlib
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "cpp", "build")


def build_cpp(name, extra_libs=()):
    """tests/cpp/<name>.cpp -> tests/cpp/build/<name>.<hash>; links libh2r (+ extra_libs = [(dir, lib)]) and the HIP runtime."""
    src = os.path.join(ROOT, "tests", "cpp", name + ".cpp")
    cmd = ["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"), src,
           "-L" + os.path.join(ROOT, "halo2_rsa_amd", "lib"), "-lh2r", "-Wl,-rpath," + os.path.join(ROOT, "halo2_rsa_amd", "lib")]
    for d, lib in extra_libs:
        cmd += ["-L" + d, "-l" + lib, "-Wl,-rpath," + d]
    cmd += ["-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"]
    h = hashlib.sha256(" ".join(cmd).encode())
    for p in [src] + sorted(glob.glob(os.path.join(ROOT, "include", "*"))):
        with open(p, "rb") as f:
            h.update(f.read())
    exe = os.path.join(OUT, "%s.%s" % (name, h.hexdigest()[:16]))
    if not os.path.exists(exe):
        os.makedirs(OUT, exist_ok=True)
        for old in glob.glob(os.path.join(OUT, name + ".*")):
            os.remove(old)
        subprocess.check_call(cmd + ["-o", exe + ".tmp"])
        os.replace(exe + ".tmp", exe)
    return exe
