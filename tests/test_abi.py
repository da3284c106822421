"""CPU tests of the drop-in boundary: libbtgpu.so loads and exports every symbol include/btgpu.h declares, and the
product path fails loudly (no CPU fallback) when no GPU is present."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols(header="btgpu.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from bayestyper_amd import lib

    syms = header_symbols()
    assert len(syms) >= 40
    dll = C.CDLL(lib.LIB_PATH)
    missing = [s for s in syms if not hasattr(dll, s)]
    assert not missing, missing


def test_comm_library_exports_every_declared_symbol():
    """libbtcomm.so (the RCCL exchange steps) exports what include/btcomm.h declares; libbtgpu.so itself carries no RCCL dependency"""
    syms = header_symbols("btcomm.h")
    assert len(syms) >= 7
    dll = C.CDLL(os.path.join(ROOT, "bayestyper_amd", "libbtcomm.so"))
    assert not [s for s in syms if not hasattr(dll, s)]
    import subprocess

    needed = subprocess.run(["readelf", "-d", os.path.join(ROOT, "bayestyper_amd", "libbtgpu.so")], capture_output=True, text=True).stdout
    assert "rccl" not in needed.lower() and "nccl" not in needed.lower()


def test_no_torch_types_and_c_linkage():
    for header in ("btgpu.h", "btcomm.h"):
        text = open(os.path.join(ROOT, "include", header)).read()
        assert 'extern "C"' in text
        code = re.sub(r"/\*.*?\*/", "", text, flags=re.S)   # comments may mention torch; signatures may not
        assert "torch" not in code.lower() and "at::" not in code and "#include <hip" not in code and "nccl" not in code.lower()


def test_fails_loudly_without_gpu():
    from bayestyper_amd import lib

    n = C.c_int(-1)
    assert lib.bt_device_count(C.byref(n)) == 0
    if n.value > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(lib.BtError, match="no HIP device"):
        lib.Ctx(0)
    assert lib.bt_version() >= 100


def test_product_does_not_reference_the_oracle():
    """the product (package + csrc + host + include) must not import, link or call anything under oracle/"""
    bad = []
    for base in ("bayestyper_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".hip", ".hpp", ".h", ".cpp", ".sh")):
                    t = open(os.path.join(dp, fn), errors="ignore").read()
                    if re.search(r"liboracle|libbtref|oracle/|_oracle|orc_", t):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad
