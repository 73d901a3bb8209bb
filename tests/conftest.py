import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Vectors:
    """Accessor for tests/golden/consensus_vectors.json + consensus_pool.bin
    (converted from the reference's tests/**/data.yaml by make_consensus_fixtures.py)."""

    def __init__(self):
        with open(os.path.join(GOLDEN, "consensus_vectors.json")) as fh:
            self.index = json.load(fh)
        with open(os.path.join(GOLDEN, "consensus_pool.bin"), "rb") as fh:
            self.pool = fh.read()

    def val(self, x):
        """decode one fixture value; returns bytes / int / bool / None / list, or the
        marker BadHex for odd-length hex strings (unrepresentable as bytes)."""
        if isinstance(x, dict):
            if "ref" in x:
                off, ln = x["ref"]
                return self.pool[off:off + ln]
            if "hex" in x:
                return bytes.fromhex(x["hex"])
            if "badhex" in x:
                return BadHex(x["badhex"])
            if "str" in x:
                return BadHex(x["str"])
            return {k: self.val(v) for k, v in x.items()}
        if isinstance(x, list):
            return [self.val(v) for v in x]
        return x

    def cases(self, api):
        return self.index[api]


class BadHex:
    def __init__(self, s):
        self.s = s

    def __len__(self):
        return -1


def load_product():
    """import the hyphen-named product package (go-eth-kzg_amd/) as module `goethkzg_amd`"""
    import importlib.util
    if "goethkzg_amd" in sys.modules:
        return sys.modules["goethkzg_amd"]
    path = os.path.join(ROOT, "go-eth-kzg_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location("goethkzg_amd", path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["goethkzg_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def build_host_shim(name, headers, extra_flags=()):
    """g++ build of tests/host_shim/<name>_shim.cpp -- the product's HOST code (go-eth-kzg_amd/csrc/host_*.hpp and the field / group
    templates the kernels instantiate) behind a C interface for CPU unit tests.  With MKZG_SHIM_SANITIZE=1 the shim is built with
    AddressSanitizer + UndefinedBehaviorSanitizer (tests/test_host_sanitizers.py runs the host test files that way in a child
    process with the sanitizer runtime preloaded).  Returns the path of the .so."""
    import subprocess
    shim_dir = os.path.join(ROOT, "tests", "host_shim")
    san = os.environ.get("MKZG_SHIM_SANITIZE") == "1"
    so = os.path.join(shim_dir, f"lib{name}_shim{'_asan' if san else ''}.so")
    src = os.path.join(shim_dir, f"{name}_shim.cpp")
    deps = [src] + [os.path.join(ROOT, "go-eth-kzg_amd", "csrc", h) for h in headers]
    if not os.path.exists(so) or any(os.path.getmtime(f) > os.path.getmtime(so) for f in deps):
        # -O2 and no frame pointer also under the sanitizers: HostFp::mul_adx is one asm block that needs fourteen general registers
        flags = ["-O2", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"] if san else ["-O2"]
        env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}          # the compiler itself runs without the preloaded runtime
        subprocess.check_call(["g++"] + flags + ["-std=c++17", "-shared", "-fPIC", "-pthread"] + list(extra_flags) + ["-o", so, src], env=env)
    return so


@pytest.fixture(scope="session")
def product():
    return load_product()


@pytest.fixture(scope="session")
def gpu_ctx(product):
    """product Context on cuda:0 -- fails loudly (no fallback) if the HIP library or GPU is missing"""
    # torch ships its own HIP runtime: initialise it first so that both it and libmi355kzg.so share one
    # runtime instance in this process (tests use torch tensors as device buffers)
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
    ctx = product.Context()
    # The shared context of the vector / oracle tests runs the BUCKET method, explicitly: the library's default is the direct window
    # table sized from free HBM (146 GB on an empty MI355X), which tests/test_gpu_defaults.py covers on a context of its own -- holding
    # that table here for the whole session would starve the tests that build their own contexts with their own table widths.
    assert ctx.set_option("commit_direct_bits", 0) == 0
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def vectors():
    return Vectors()


@pytest.fixture(scope="session")
def oracle_ctx():
    import oracle
    ctx = oracle.OracleContext()
    yield ctx
    ctx.close()
