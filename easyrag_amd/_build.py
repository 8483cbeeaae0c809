"""Build libeasyrag_hip.so in-tree with hipcc for gfx950 (no JIT cache: the built .so travels with the tree)."""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
INCLUDE = PKG_DIR.parent / "include"
MEASURE = os.environ.get("ERH_MEASURE", "0") not in ("", "0")
# the measurement build (kernel ablations, section clocks; scripts/kbench.py) is a separate library file, so the
# product library in the tree is never the one with measurement variants inside
LIB_PATH = PKG_DIR / ("libeasyrag_hip_measure.so" if MEASURE else "libeasyrag_hip.so")
STAMP = PKG_DIR / (".libeasyrag_hip_measure.stamp" if MEASURE else ".libeasyrag_hip.stamp")

SOURCES = ["api.hip", "pipeline_dense.hip", "pipeline_bm25.hip", "comm.hip", "dense_scan.hip", "dense_gemv.hip", "select.hip", "bm25.hip", "fuse.hip", "index_build.hip", "text.hip"]
HEADERS = ["common.h", "kernels.h", "handle.h"]
# kernel generations that compile only into the measurement build (scripts/kbench.py): part of ITS digest, not of the product's
MEASURE_ONLY = ["measure/dense_scan_persist.inc", "measure/dense_scan_pp12.inc", "measure/dense_scan_pp4.inc"]

HIPCC_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-ffp-contract=off",      # bit parity: no fused multiply-add unless written as one
    "-fno-fast-math",
    "-Wall",
    "-Wno-unused-function",
]


def _flags():
    """ERH_MEASURE=1 in the environment builds the measurement variants (kernel ablations, section clocks) that
    scripts/kbench.py drives; the product build carries none of them."""
    return HIPCC_FLAGS + (["-DERH_MEASURE"] if MEASURE else [])


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 to build libeasyrag_hip.so)")


def _digest() -> str:
    h = hashlib.sha256()
    for name in SOURCES + HEADERS + (MEASURE_ONLY if MEASURE else []):
        h.update((CSRC / name).read_bytes())
    h.update((INCLUDE / "easyrag_hip.h").read_bytes())
    h.update(" ".join(_flags()).encode())
    return h.hexdigest()


def _kernel_digest() -> str:
    """Digest of what decides the device side of a run (every source but the host-only text side, the internal headers,
    the flags): what profiles/pmc_traffic.json is keyed by, so that a change to the text ABI or to the public header does
    not orphan the PMC figures of unchanged kernels."""
    h = hashlib.sha256()
    for name in [n for n in SOURCES if n != "text.hip"] + HEADERS + (MEASURE_ONLY if MEASURE else []):
        h.update((CSRC / name).read_bytes())
    h.update(" ".join(_flags()).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile every HIP translation unit for gfx950 and link the shared library. Idempotent; concurrent callers (the
    ranks of a multi-GPU launch) serialise on a lock file, and a box without hipcc keeps using a library that exists."""
    dig = _digest()
    if not force and LIB_PATH.exists() and STAMP.exists() and STAMP.read_text().strip() == dig:
        return LIB_PATH
    import fcntl
    try:
        lock = open(str(LIB_PATH) + ".lock", "w")
    except OSError:
        # a read-only install: nothing can be (re)built here; a library that exists is used as it is
        if LIB_PATH.exists():
            import warnings
            warnings.warn(f"{LIB_PATH.parent} is not writable: using the existing {LIB_PATH.name} although its source "
                          f"digest stamp is missing or stale")
            return LIB_PATH
        raise
    with lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and LIB_PATH.exists() and STAMP.exists() and STAMP.read_text().strip() == dig:
                return LIB_PATH                      # another process built it while this one waited
            try:
                _hipcc()
            except RuntimeError:
                if LIB_PATH.exists():
                    import warnings
                    warnings.warn(f"hipcc not found: using the existing {LIB_PATH.name} although its source digest "
                                  f"stamp is missing or stale")
                    return LIB_PATH
                raise
            return _build_locked(dig, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(dig: str, verbose: bool) -> Path:
    hipcc = _hipcc()
    obj_dir = PKG_DIR / ("build_measure" if MEASURE else "build")
    obj_dir.mkdir(exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        obj = obj_dir / (src.replace(".hip", ".o"))
        cmd = [hipcc, *_flags(), "-I", str(INCLUDE), "-I", str(CSRC), "-c", str(CSRC / src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(str(obj))
    failed = []
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed.append(f"--- {src} ---\n{out}")
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError("hipcc failed:\n" + "\n".join(failed))
    tmp = str(LIB_PATH) + ".tmp"
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp, *objs]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout)
    os.replace(tmp, LIB_PATH)
    STAMP.write_text(dig)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in os.sys.argv, verbose=True))
