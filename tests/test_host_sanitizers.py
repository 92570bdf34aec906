"""CPU: the host-side model code (no GPU, no engine) under AddressSanitizer + UndefinedBehaviorSanitizer.

GPU sanitizers are not available on the pool, so the device code is covered by the parity tests; the C++ that parses
trees, compresses patterns and builds models is run here with instrumentation over a golden alignment."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "physher_amd", "csrc", "host")


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_code_under_asan_ubsan(tmp_path):
    exe = tmp_path / "host_sanitize"
    srcs = [os.path.join(HOST, f) for f in ("tree.cpp", "patterns.cpp", "models.cpp")] + [os.path.join(ROOT, "tests", "cpp", "host_sanitize.cpp")]
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-I" + HOST, "-I" + os.path.join(ROOT, "include"), "-o", str(exe)] + srcs)
    case = os.path.join(ROOT, "tests", "golden", "gtr_g4_t24_gaps")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([str(exe), os.path.join(case, "aln.fa"), os.path.join(case, "tree.nwk")], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host_sanitize: ok" in out.stdout
