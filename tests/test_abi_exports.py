"""CPU-side checks of the drop-in boundary: the built C-ABI library exports every symbol that
include/mi355kzg.h declares (no compute calls -- there is no GPU here), and the product refuses to
construct a context without a HIP device instead of falling back to anything."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    with open(os.path.join(ROOT, "include", "mi355kzg.h")) as fh:
        txt = fh.read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mkzg_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported(product):
    lib = product.lib()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mi355kzg.h but not exported by libmi355kzg.so"
    # the Python binding's list and the header agree
    assert set(product.EXPORTED_SYMBOLS) == set(names)


def test_product_never_imports_oracle():
    """the product tree must not reference oracle/ (parity claims depend on it)"""
    pkg = os.path.join(ROOT, "go-eth-kzg_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                with open(os.path.join(dirpath, f), errors="ignore") as fh:
                    src = fh.read()
                assert "import oracle" not in src and "kzgo_" not in src and "libkzg_oracle" not in src, f


def test_no_cpu_fallback(product):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        product.Context()


def test_c_caller_compiles_links_and_fails_cleanly(product):
    """include/mi355kzg.h is plain C99 and libmi355kzg.so links into a C program (what cgo does); without a GPU the
    program's mkzg_ctx_new returns MKZG_ERR_HIP instead of aborting or falling back"""
    import subprocess
    src = os.path.join(ROOT, "tests", "c_abi", "use_all.c")
    exe = os.path.join(ROOT, "tests", "c_abi", "use_all")
    libdir = os.path.join(ROOT, "go-eth-kzg_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), src, "-o", exe,
                           "-L", libdir, "-lmi355kzg", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([exe, os.path.join(ROOT, "go-eth-kzg_amd", "trusted_setup.json")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    import torch
    if not torch.cuda.is_available():
        assert "MKZG_ERR_HIP" in r.stdout
    # every declared symbol is referenced by the C program
    with open(src) as fh:
        txt = fh.read()
    for n in declared_symbols():
        assert "(any_fn)" + n in txt, n
