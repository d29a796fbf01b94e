"""build.py decides staleness by content, not by mtime (a snapshot whose .so looks newer must still rebuild)."""
import os

from fragnet_amd import build


def test_digest_covers_every_source_and_flags(tmp_path, monkeypatch):
    d0 = build.source_digest()
    assert len(d0) == 64 and d0 == build.source_digest()
    assert any(p.endswith("dense_head.inc") for p in build.INCLUDED) and any(p.endswith("fn_internal.h") for p in build.INCLUDED)
    assert any(p.endswith("mol_plan.hip") for p in build.SOURCES) and any(p.endswith("tower.hip") for p in build.SOURCES)
    fake = tmp_path / "extra.inc"
    fake.write_text("// x\n")
    monkeypatch.setattr(build, "INCLUDED", build.INCLUDED + [str(fake)])
    d1 = build.source_digest()
    assert d1 != d0
    fake.write_text("// y\n")
    assert build.source_digest() != d1
    monkeypatch.setattr(build, "FLAGS", build.FLAGS + ["-DX"])
    assert build.source_digest() != d1


def test_stale_follows_the_stamp_not_the_clock(tmp_path, monkeypatch):
    out = tmp_path / "lib.so"
    monkeypatch.setattr(build, "OUT", str(out))
    monkeypatch.setattr(build, "STAMP", str(out) + ".sha256")
    assert build.stale()                               # nothing there
    out.write_bytes(b"\x7fELF")
    assert build.stale()                               # library without a stamp
    with open(build.STAMP, "w") as f:
        f.write("0" * 64)
    os.utime(out, (4102444800, 4102444800))            # year 2100: "newer" than any source
    assert build.stale()                               # wrong digest wins over the clock
    with open(build.STAMP, "w") as f:
        f.write(build.source_digest())
    assert not build.stale()


def test_loading_a_stale_library_is_refused(monkeypatch):
    """_lib.load() must not bind a library whose stamp disagrees with the sources (a bench / profile of yesterday's kernels
    looks exactly like one of today's)."""
    import pytest
    from fragnet_amd import _lib
    build.build_lib()
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(build, "FLAGS", build.FLAGS + ["-DSOMETHING_ELSE"])      # the digest no longer matches the stamp
    with pytest.raises(_lib.FragnetHipError, match="does not match the sources"):
        _lib.load()
    monkeypatch.undo()
    assert _lib.load() is not None
