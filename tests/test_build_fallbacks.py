"""Host logic of easyrag_amd/_build.py: what happens where nothing can be (re)built: a read-only install keeps
using the library that is there, a box without hipcc likewise, and without a library both raise."""
import warnings

import pytest

from easyrag_amd import _build


def _point_at(tmp_path, monkeypatch, with_lib: bool):
    lib = tmp_path / "libeasyrag_hip.so"
    if with_lib:
        lib.write_bytes(b"not a real library: build() only returns its path")
    monkeypatch.setattr(_build, "LIB_PATH", lib)
    monkeypatch.setattr(_build, "STAMP", tmp_path / ".libeasyrag_hip.stamp")      # missing: the digest check fails
    return lib


def test_unwritable_directory_uses_the_existing_library(tmp_path, monkeypatch):
    lib = _point_at(tmp_path, monkeypatch, with_lib=True)
    real_open = open

    def no_lock(path, *a, **k):
        if str(path).endswith(".lock"):
            raise PermissionError(13, "read-only install", str(path))
        return real_open(path, *a, **k)

    monkeypatch.setattr("builtins.open", no_lock)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert _build.build() == lib
    assert any("not writable" in str(x.message) for x in w)


def test_unwritable_directory_without_a_library_raises(tmp_path, monkeypatch):
    _point_at(tmp_path, monkeypatch, with_lib=False)
    real_open = open

    def no_lock(path, *a, **k):
        if str(path).endswith(".lock"):
            raise PermissionError(13, "read-only install", str(path))
        return real_open(path, *a, **k)

    monkeypatch.setattr("builtins.open", no_lock)
    with pytest.raises(OSError):
        _build.build()


def test_missing_hipcc_uses_the_existing_library_or_raises(tmp_path, monkeypatch):
    lib = _point_at(tmp_path, monkeypatch, with_lib=True)

    def no_hipcc():
        raise RuntimeError("hipcc not found")

    monkeypatch.setattr(_build, "_hipcc", no_hipcc)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert _build.build() == lib
    assert any("hipcc not found" in str(x.message) for x in w)
    lib.unlink()
    with pytest.raises(RuntimeError):
        _build.build()


def test_kernel_digest_ignores_the_text_side_and_the_public_header():
    """profiles/pmc_traffic.json is keyed by the digest of what decides the device side of a run."""
    names = [n for n in _build.SOURCES if n != "text.hip"] + list(_build.HEADERS)
    assert "text.hip" in _build.SOURCES and "text.hip" not in names
    assert all((_build.CSRC / n).exists() for n in names)
    assert len(_build._kernel_digest()) == 64
