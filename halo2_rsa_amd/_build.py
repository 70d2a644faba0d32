"""Builds halo2_rsa_amd/lib/libh2r.so (hand-written HIP for gfx950) in-tree with hipcc.

The library is five translation units (csrc/h2r_internal.hpp lists them) compiled IN PARALLEL plus a one-line unit that carries
the BUILD ID: the SHA-256 of every file the library is compiled from (csrc/*, include/*).  The shipped .so answers
`h2r_build_id()` with it, and `stale()` compares that string -- read straight out of the file, nothing is loaded -- with the hash of
the tree: mtimes play no part (an rsync or a fresh checkout reorders them).  Objects are cached under lib/obj/ by the hash of what
each one is compiled from, so touching one unit recompiles one unit."""
import concurrent.futures
import glob
import hashlib
import os
import re
import shutil
import subprocess
import sys
import time

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
UNITS = ["h2r_api.hip", "h2r_tu_trace.hip", "h2r_tu_chain.hip", "h2r_tu_step.hip", "h2r_tu_cells.hip"]
ID_UNIT = "h2r_tu_id.cpp"
LIB = os.path.join(PKG, "lib", "libh2r.so")
OBJ = os.path.join(PKG, "lib", "obj")
MARK = b"H2R_BUILD_ID="
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-fvisibility=hidden", "-fvisibility-inlines-hidden"]


def _deps():   # every file the library is compiled from
    return sorted(glob.glob(os.path.join(CSRC, "*")) + glob.glob(os.path.join(ROOT, "include", "*")))


DEPS = _deps()


def _sha(paths, extra=()):
    h = hashlib.sha256()
    for p in paths:
        h.update(os.path.relpath(p, ROOT).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    for e in extra:
        h.update(e.encode() + b"\0")
    return h.hexdigest()


def source_id(defines=()):
    """The build ID of the tree as it stands (plus a developer variant's -D flags)."""
    return _sha(_deps(), sorted(defines))


def lib_id(path=None):
    """The build ID a built library carries (None: no library, or one from before build IDs)."""
    path = path or LIB
    if not os.path.exists(path):
        return None
    with open(path, "rb") as f:
        m = re.search(MARK + rb"([0-9a-f]{64})", f.read())
    return m.group(1).decode() if m else None


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libh2r.so cannot be built")


def stale():
    return lib_id() != source_id()


def _compile(unit, obj, flags, verbose):
    cmd = [hipcc_path()] + flags + ["-c", "-o", obj, os.path.join(CSRC, unit)]
    if verbose:
        print(" ".join(cmd), flush=True)
    t0 = time.time()
    subprocess.check_call(cmd)
    return unit, time.time() - t0


def build_lib(force=False, verbose=False, out=None, defines=()):
    """`out`/`defines` build a developer variant (tools/*: same-box A/B runs via H2R_LIB).
    Returns the library's path; `build_lib.last` says what happened ("reused" | "built: ...")."""
    sid = source_id(defines)
    if out is None and not force and lib_id() == sid:
        build_lib.last = "reused (build id %s matches the sources)" % sid[:16]
        return LIB
    out = out or LIB
    os.makedirs(os.path.dirname(out), exist_ok=True)
    os.makedirs(OBJ, exist_ok=True)
    flags = FLAGS + ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + ["-D" + d for d in defines]
    headers = [p for p in _deps() if not p.endswith((".hip", ".cpp"))]
    jobs, objs, t0 = [], [], time.time()
    for unit in UNITS:
        key = _sha(headers + [os.path.join(CSRC, unit)], flags)[:20]
        obj = os.path.join(OBJ, "%s.%s.o" % (os.path.splitext(unit)[0], key))
        objs.append(obj)
        if force or not os.path.exists(obj):
            jobs.append((unit, obj))
    # the one unit that knows the build ID (host code only: a second)
    id_obj = os.path.join(OBJ, "h2r_tu_id.%s.o" % sid[:20])
    objs.append(id_obj)
    with concurrent.futures.ThreadPoolExecutor(max_workers=max(1, min(len(jobs) + 1, os.cpu_count() or 1))) as ex:
        futs = [ex.submit(_compile, u, o, flags, verbose) for (u, o) in jobs]
        futs.append(ex.submit(_compile, ID_UNIT, id_obj, flags + ['-DH2R_BUILD_ID_STR="%s"' % sid], verbose))
        times = dict(f.result() for f in futs)
    link = [hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(CSRC, "libh2r.map"), "-o", out] + objs
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.check_call(link)
    # keep the object cache bounded: per unit the objects of this build and the six most recent others (developer variants, the previous tree)
    by_unit = {}
    for old in glob.glob(os.path.join(OBJ, "*.o")):
        if old not in objs:
            by_unit.setdefault(os.path.basename(old).split(".")[0], []).append(old)
    for olds in by_unit.values():
        for old in sorted(olds, key=os.path.getmtime, reverse=True)[6:]:
            os.remove(old)
    assert lib_id(out) == sid, "the linked library does not carry the build ID"
    build_lib.last = "built in %.0f s (%s; %d unit(s) from the object cache), build id %s" % (
        time.time() - t0, ", ".join("%s %.0f s" % (u, t) for u, t in sorted(times.items())), len(UNITS) - len(jobs), sid[:16])
    return out


build_lib.last = ""


if __name__ == "__main__":
    # python -m halo2_rsa_amd._build [variant-name -DFOO=1 ...]  -> lib/variants/<name>.so
    if len(sys.argv) > 1:
        print(build_lib(out=os.path.join(PKG, "lib", "variants", sys.argv[1] + ".so"),
                        defines=[a[2:] if a.startswith("-D") else a for a in sys.argv[2:]], verbose=True))
    else:
        print(build_lib(force=True, verbose=True))
    print(build_lib.last)
