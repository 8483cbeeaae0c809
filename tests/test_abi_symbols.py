"""The C-ABI library builds for gfx950 in this GPU-less container, loads, and exports every symbol that
include/easyrag_hip.h declares.  No compute call is made here; without a device erh_create must refuse."""
import ctypes as C
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def declared_symbols():
    text = (ROOT / "include" / "easyrag_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(erh_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_bound_and_exported():
    from easyrag_amd import _lib
    lib = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 20
    assert sorted(_lib.SIGNATURES) == syms, "ctypes table and header disagree"
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in the header but not exported"
    assert lib.erh_version() >= 100
    assert lib.erh_status_str(0) == b"ok" and lib.erh_status_str(-2).startswith(b"no usable")


def test_no_device_means_no_engine():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from easyrag_amd import _lib
    from easyrag_amd.engine import RetrievalEngine
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.erh_create(0, C.byref(h)) == _lib.ERH_ERR_NO_DEVICE and not h.value
    with pytest.raises(_lib.ErhError):
        RetrievalEngine(0)          # the product path fails loudly: there is no CPU fallback


def test_product_never_imports_the_oracle():
    for p in list((ROOT / "easyrag_amd").rglob("*.py")) + list((ROOT / "easyrag_amd" / "csrc").glob("*")):
        if p.is_file():
            txt = p.read_text(errors="ignore")
            assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f"{p} imports the oracle"
