"""CPU sanitizer coverage of the product's HOST code and of the oracle (the reference relies on Go's race detector
and on memory safety of the language; the C++ host side gets the equivalent from ASan / UBSan / TSan here -- GPU sanitizers are not
available on the pool).

  * the host unit tests (pairing, tower arithmetic, ADX Montgomery multiplication, SHA-NI / portable SHA-256, the 29-bit-limb field and
    group templates the kernels instantiate) re-run in a child process against shims built with -fsanitize=address,undefined;
  * the host pool (parked workers, callers that help, futex countdowns) under -fsanitize=thread, driven the way the C ABI drives it;
  * the whole HOST LAYER of the product -- go-eth-kzg_amd/csrc/api*.hip compiled as plain C++: lanes and their hand-over, lazily built
    shared tables, hash rounds, option changes under load, the multi-device front -- under -fsanitize=thread on an inert HIP runtime made
    of host memory (tests/host_shim/stub_hip.cpp) with do-nothing kernel wrappers generated from the kernel headers
    (tests/host_shim/gen_stub_device.py), driven through the public C ABI by tests/host_shim/stub_device_driver.cpp;
  * the oracle's ASan/UBSan build (oracle/Makefile) over the consensus vectors once.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM_DIR = os.path.join(ROOT, "tests", "host_shim")


def _runtime(name):
    p = subprocess.check_output(["gcc", f"-print-file-name={name}"], text=True).strip()
    if not os.path.isabs(p):
        pytest.skip(f"{name} is not installed with this gcc")
    return p


def _pytest_child(files, extra_env):
    env = dict(os.environ, LD_PRELOAD=_runtime("libasan.so") + ":" + _runtime("libubsan.so"),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", **extra_env)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + [os.path.join(ROOT, "tests", f) for f in files],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
    return r.stdout


def test_host_code_under_asan_ubsan():
    out = _pytest_child(["test_host_pairing.py", "test_mont29_host.py", "test_host_sha256.py"], {"MKZG_SHIM_SANITIZE": "1"})
    assert " passed" in out and "failed" not in out, out[-500:]
    for name in ("pairing", "mont29", "sha"):
        assert os.path.exists(os.path.join(SHIM_DIR, f"lib{name}_shim_asan.so"))      # the sanitizer builds really were the ones loaded


def test_host_pool_under_tsan():
    exe = os.path.join(SHIM_DIR, "pool_stress_tsan")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-pthread", "-o", exe, os.path.join(SHIM_DIR, "pool_stress.cpp")])
    r = subprocess.run([exe], env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "0 mismatches" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]


def test_call_combiner_under_tsan():
    """go-eth-kzg_amd/csrc/host_combine.hpp alone: every item processed once by its group's leader, at most max_running() groups in flight,
    a lone caller is a group of one"""
    exe = os.path.join(SHIM_DIR, "combiner_stress_tsan")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-pthread", "-o", exe, os.path.join(SHIM_DIR, "combiner_stress.cpp")])
    r = subprocess.run([exe], env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]


def test_host_layer_on_stub_device_under_tsan():
    """ten caller threads on one context (one of them changing options and reading the profile all the while), four on a context of the
    device they are NOT on, six on a two-device context: no runtime error, tables on their own device, the callers' current device
    restored by every entry point, every "device" allocation returned -- and no data race"""
    if not os.path.exists("/opt/rocm/include/hip/hip_runtime.h"):
        pytest.skip("HIP headers not installed")
    csrc = os.path.join(ROOT, "go-eth-kzg_amd", "csrc")
    gen = os.path.join(SHIM_DIR, "stub_device_gen.cpp")
    subprocess.check_call([sys.executable, os.path.join(SHIM_DIR, "gen_stub_device.py"), gen], stdout=subprocess.DEVNULL)
    exe = os.path.join(SHIM_DIR, "stub_device_tsan")
    cmd = ["g++", "-O2", "-g", "-std=c++17", "-fsanitize=thread", "-pthread", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-Wno-unused-result", "-w", "-x", "c++"]
    cmd += [os.path.join(csrc, f) for f in ("api.hip", "api_verify.hip", "api_cells.hip", "api_multi.hip", "api_exchange.hip")]
    cmd += ["-x", "none", os.path.join(SHIM_DIR, "stub_hip.cpp"), gen, os.path.join(SHIM_DIR, "stub_device_driver.cpp"), "-ldl", "-o", exe]
    subprocess.check_call(cmd)
    # the in-process stand-in for librccl that the library's exchange step dlopens (ranks = threads): 1, 2 and 8 ranks
    stub_rccl = os.path.join(SHIM_DIR, "libstub_rccl.so")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-shared", "-fPIC", "-pthread", "-o", stub_rccl, os.path.join(SHIM_DIR, "stub_rccl.cpp")])
    for rep in range(2):
        r = subprocess.run([exe, os.path.join(ROOT, "go-eth-kzg_amd", "trusted_setup.json")], env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0:second_deadlock_stack=1", MKZG_RCCL_LIB=stub_rccl),
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "0 failed checks" in r.stdout, (r.stdout[-2000:], r.stderr[-6000:])
        assert "ThreadSanitizer" not in r.stderr, r.stderr[-6000:]


def test_oracle_vectors_under_asan_ubsan():
    out = _pytest_child(["test_oracle_vectors.py"], {"KZG_ORACLE_SANITIZE": "1", "OMP_NUM_THREADS": "4"})
    assert " passed" in out and "failed" not in out, out[-500:]
    assert os.path.exists(os.path.join(ROOT, "oracle", "libkzg_oracle_asan.so"))
