"""The host-only text side (csrc/text.hip: vocabulary, cutter, HMM, fused / threaded encode) compiled as plain C++ with
AddressSanitizer + UndefinedBehaviorSanitizer, and once more with ThreadSanitizer (the threaded encode), and driven by tests/native/text_fuzz.cc: random dictionaries, models and
byte strings (malformed UTF-8 included) through every entry point.  No GPU, no HIP."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
@pytest.mark.parametrize("sanitizers", ["address,undefined", "thread"], ids=["asan+ubsan", "tsan"])
def test_text_side_under_sanitizers(tmp_path, sanitizers):
    exe = tmp_path / "text_fuzz"
    cmd = ["g++", "-std=c++17", "-O1", "-g", f"-fsanitize={sanitizers}", "-fno-sanitize-recover=all", "-pthread",
           "-I", os.path.join(ROOT, "include"),
           "-x", "c++", os.path.join(ROOT, "easyrag_amd", "csrc", "text.hip"), os.path.join(HERE, "native", "text_fuzz.cc"),
           "-o", str(exe)]
    build = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0"))
    assert run.returncode == 0, (run.stdout[-1000:], run.stderr[-3000:])
    assert run.stdout.startswith("ok ")
