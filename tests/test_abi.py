"""CPU-only: the C-ABI library loads and exports exactly what include/physher_amd.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    with open(os.path.join(ROOT, "include", "physher_amd.h")) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(phyamd_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from physher_amd import _lib
    lib = _lib.load()  # dlopen works without a GPU; compute calls are not made here
    declared = _declared()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/physher_amd.h but not exported"
    assert sorted(n for n, _, _ in _lib.SYMBOLS) == declared
    assert lib.phyamd_abi_version() == _lib.ABI_VERSION


def test_header_is_plain_c(tmp_path):
    """the boundary is a C ABI: the header compiles as C99 (no C++ types, no torch types) and a C caller links against the library"""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("needs gcc")
    src = tmp_path / "caller.c"
    src.write_text('#include "physher_amd.h"\n#include <stdio.h>\n'
                   'int main(void) { phyamd_config cfg = {0}; phyamd_engine *e = 0; cfg.tip_count = 1; '
                   'int rc = phyamd_create(&cfg, &e); printf("%d %d %s\\n", phyamd_abi_version(), rc, phyamd_last_error()); return rc == PHYAMD_EINVAL ? 0 : 1; }\n')
    exe = tmp_path / "caller"
    lib_dir = os.path.join(ROOT, "physher_amd")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), "-o", str(exe), str(src),
                           "-L" + lib_dir, "-lphysher_amd", "-Wl,-rpath," + lib_dir])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr  # tip_count 1 is refused before any device is touched
    assert "tip_count" in out.stdout


def test_null_and_bad_arguments_are_reported_not_crashed():
    from physher_amd import _lib
    lib = _lib.load()
    assert lib.phyamd_create(None, None) == _lib.EINVAL
    assert b"null" in lib.phyamd_last_error()
    h = ctypes.c_void_p()
    cfg = _lib.Config(1, 10, 4, 1, -1, 0, 0, None)  # a 1-tip tree is not a tree
    assert lib.phyamd_create(ctypes.byref(cfg), ctypes.byref(h)) == _lib.EINVAL
    cfg = _lib.Config(4, 0, 4, 1, -1, 0, 0, None)
    assert lib.phyamd_create(ctypes.byref(cfg), ctypes.byref(h)) == _lib.EINVAL
    assert lib.phyamd_log_likelihood(None, None) == _lib.EINVAL
    lib.phyamd_destroy(None)  # no-op


def test_no_gpu_is_a_loud_error():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from physher_amd.engine import Engine, EngineError
    with pytest.raises(EngineError):
        Engine(4, 10)


def test_product_code_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under physher_amd/ may import, load or link it."""
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, "physher_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".cpp", ".h", ".hpp", ".c", "Makefile")):
                with open(os.path.join(dp, fn), errors="ignore") as f:
                    txt = f.read()
                if re.search(r"phyoracle|from oracle|import oracle|oracle/", txt):
                    bad.append(os.path.join(dp, fn))
    assert not bad, bad
