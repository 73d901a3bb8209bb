"""The lane-pair SHA-256 schedule of k_challenge_sha256, checked on the CPU through its instruction-level model."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sha_pair_model_matches_hashlib():
    spec = importlib.util.spec_from_file_location("sha_pair_model", os.path.join(ROOT, "tools", "sha_pair_model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.selftest()


def test_kernel_uses_the_modelled_sequence():
    """the macros the model mirrors are the ones the kernel is built from"""
    src = open(os.path.join(ROOT, "go-eth-kzg_amd", "csrc", "verify_kernels.hip")).read()
    for name in ("SHA_INIT_E", "SHA_BOUNDARY_A", "SHA_END_BODY", "v_xad_u32", "SHA_CORR0", "row_mirror"):
        assert name in src
