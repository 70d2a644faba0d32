"""Builds halo2_rsa_amd/lib/libh2r.so (hand-written HIP for gfx950) in-tree with hipcc."""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
SRC = os.path.join(PKG, "csrc", "h2r_api.hip")


def _deps():   # every file the library is compiled from: a newer one makes the shipped .so stale
    import glob
    return sorted(glob.glob(os.path.join(PKG, "csrc", "*")) + glob.glob(os.path.join(ROOT, "include", "*")))


DEPS = _deps()
LIB = os.path.join(PKG, "lib", "libh2r.so")


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libh2r.so cannot be built")


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build_lib(force=False, verbose=False, out=None, defines=()):
    """`out`/`defines` build a developer variant (tools/*: same-box A/B runs via H2R_LIB)."""
    if out is None and not force and not stale():
        return LIB
    out = out or LIB
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = [hipcc_path(), "-O3", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC",
           "-fvisibility=hidden", "-fvisibility-inlines-hidden", "-Wl,--version-script=" + os.path.join(PKG, "csrc", "libh2r.map"),
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "csrc"), "-o", out, SRC]
    cmd += ["-D" + d for d in defines]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    import sys
    # python -m halo2_rsa_amd._build [variant-name -DFOO=1 ...]  -> lib/variants/<name>.so
    if len(sys.argv) > 1:
        print(build_lib(out=os.path.join(PKG, "lib", "variants", sys.argv[1] + ".so"),
                        defines=[a[2:] if a.startswith("-D") else a for a in sys.argv[2:]], verbose=True))
    else:
        print(build_lib(force=True, verbose=True))
