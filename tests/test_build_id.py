"""The library identifies the sources it was built from: h2r_build_id() = SHA-256 of csrc/* + include/* as halo2_rsa_amd/_build.py hashes
them; `_build.stale()` compares that string with the tree's hash -- file times play no part."""
import ctypes
import os
import shutil

from halo2_rsa_amd import _build, _lib


def test_shipped_library_carries_the_hash_of_the_tree():
    sid = _build.source_id()
    assert len(sid) == 64
    assert _build.lib_id() == sid, "libh2r.so was not built from this tree: run python -m halo2_rsa_amd._build"
    if "H2R_LIB" not in os.environ:
        L = ctypes.CDLL(_lib.lib_path())
        L.h2r_build_id.restype = ctypes.c_char_p
        assert L.h2r_build_id().decode() == sid
    assert not _build.stale()


def test_file_times_do_not_decide(tmp_path):
    """Touching every source (an rsync, a fresh checkout) leaves the library current; build_lib() reuses it and says so."""
    before = {p: os.stat(p).st_mtime for p in _build.DEPS}
    try:
        for p in _build.DEPS:
            os.utime(p, None)                      # now: newer than the library
        assert not _build.stale()
        assert _build.build_lib() == _build.LIB
        assert _build.build_lib.last.startswith("reused")
    finally:
        for p, t in before.items():
            os.utime(p, (t, t))


def test_a_changed_source_changes_the_id(tmp_path, monkeypatch):
    """One byte in one header: another ID (hashed on a copy of the tree's sources, nothing is compiled)."""
    root = tmp_path / "repo"
    shutil.copytree(os.path.join(_build.ROOT, "include"), root / "include")
    shutil.copytree(_build.CSRC, root / "halo2_rsa_amd" / "csrc")
    monkeypatch.setattr(_build, "ROOT", str(root))
    monkeypatch.setattr(_build, "CSRC", str(root / "halo2_rsa_amd" / "csrc"))
    same = _build.source_id()
    monkeypatch.undo()
    assert same == _build.source_id()              # the ID hashes relative names and contents only
    monkeypatch.setattr(_build, "ROOT", str(root))
    monkeypatch.setattr(_build, "CSRC", str(root / "halo2_rsa_amd" / "csrc"))
    with open(root / "include" / "h2r.h", "a") as f:
        f.write("\n")
    assert _build.source_id() != same
    assert _build.source_id(["H2R_DEV_KNOBS"]) != _build.source_id()   # a developer variant's flags are part of its ID
