"""ctypes loader for the CPU oracle (oracle/libkzg_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (go-eth-kzg_amd/) never imports this module.

The class mirrors go-eth-kzg's Context method set (reference api.go:17-28, prove.go,
verify.go, api_eip7594.go, api_eip.go) at the byte level; every method returns
(rc, outputs...) with rc numbered as in include/mi355kzg.h.
"""
import ctypes as C
import json
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
# KZG_ORACLE_SANITIZE=1: the AddressSanitizer + UBSan build (oracle/Makefile; the process must preload the sanitizer runtime --
# tests/test_host_sanitizers.py runs the vector tests that way in a child process)
_SAN = os.environ.get("KZG_ORACLE_SANITIZE") == "1"
LIB_NAME = "libkzg_oracle_asan.so" if _SAN else "libkzg_oracle.so"
LIB_PATH = os.path.join(HERE, LIB_NAME)
DEFAULT_SETUP = os.path.join(os.path.dirname(HERE), "go-eth-kzg_amd", "trusted_setup.json")

BLOB_BYTES = 131072
CELL_BYTES = 2048
N_CELLS = 128

OK, VERIFY_FAILED = 0, 1


def build(force=False):
    """Compile the oracle with gcc (checker only; building it is not using it)."""
    srcs = ["kzg_oracle.c", "field.h", "curve.h", "pairing.h", "sha256.h", "constants.h"]
    if not force and os.path.exists(LIB_PATH):
        m = os.path.getmtime(LIB_PATH)
        if all(os.path.getmtime(os.path.join(HERE, s)) <= m for s in srcs):
            return LIB_PATH
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}
    subprocess.check_call(["make", "-C", HERE, LIB_NAME], stdout=subprocess.DEVNULL, env=env)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        vp, sz, i32, u8p = C.c_void_p, C.c_size_t, C.c_int, C.c_char_p
        L.kzgo_ctx_new.argtypes = [u8p, u8p, u8p, sz, i32, C.POINTER(vp)]
        L.kzgo_ctx_free.argtypes = [vp]
        L.kzgo_ctx_free.restype = None
        L.kzgo_blob_to_kzg_commitment.argtypes = [vp, u8p, u8p]
        L.kzgo_blob_to_kzg_commitment_batch.argtypes = [vp, u8p, sz, u8p, vp]
        L.kzgo_blob_to_kzg_commitment_batch_threads.argtypes = [vp, u8p, sz, u8p, vp, i32]
        L.kzgo_compute_kzg_proof.argtypes = [vp, u8p, u8p, u8p, u8p]
        L.kzgo_compute_blob_kzg_proof.argtypes = [vp, u8p, u8p, u8p]
        L.kzgo_compute_blob_kzg_proof_batch_threads.argtypes = [vp, u8p, u8p, sz, u8p, vp, i32]
        L.kzgo_verify_kzg_proof.argtypes = [vp, u8p, u8p, u8p, u8p]
        L.kzgo_verify_blob_kzg_proof.argtypes = [vp, u8p, u8p, u8p]
        L.kzgo_verify_blob_kzg_proof_batch.argtypes = [vp, u8p, sz, u8p, sz, u8p, sz]
        L.kzgo_compute_cells.argtypes = [vp, u8p, u8p]
        L.kzgo_compute_cells_and_kzg_proofs.argtypes = [vp, u8p, u8p, u8p]
        L.kzgo_recover_cells_and_kzg_proofs.argtypes = [vp, vp, sz, u8p, sz, u8p, u8p]
        L.kzgo_recover_cells.argtypes = [vp, vp, sz, u8p, sz, u8p]
        L.kzgo_verify_cell_kzg_proof_batch.argtypes = [vp, u8p, sz, vp, sz, u8p, sz, u8p, sz]
        L.kzgo_verify_blob_batch_partial.argtypes = [vp, u8p, u8p, u8p, sz, C.c_uint64, u8p, u8p]
        L.kzgo_verify_blob_batch_finish.argtypes = [vp, u8p, sz]
        L.kzgo_verify_cell_batch_partial.argtypes = [vp, u8p, sz, vp, u8p, u8p, C.c_uint64, u8p, u8p]
        L.kzgo_verify_cell_batch_finish.argtypes = [vp, u8p, sz]
        L.kzgo_fr_fft.argtypes = [u8p, sz, i32]
        L.kzgo_fr_coset_fft.argtypes = [u8p, sz, u8p, i32]
        L.kzgo_g1_msm.argtypes = [u8p, u8p, sz, u8p]
        L.kzgo_g1_msm_naive.argtypes = [u8p, u8p, sz, u8p]
        L.kzgo_g1_validate.argtypes = [u8p]
        L.kzgo_compute_challenge.argtypes = [u8p, u8p, u8p]
        L.kzgo_compute_challenge.restype = None
        L.kzgo_sha256.argtypes = [u8p, sz, u8p]
        L.kzgo_sha256.restype = None
        L.kzgo_evaluate_blob.argtypes = [vp, u8p, u8p, u8p]
        L.kzgo_g1_gen_mul.argtypes = [u8p, u8p]
        L.kzgo_g1_gen_mul.restype = None
        L.kzgo_rand_blob.argtypes = [C.c_int64, u8p]
        L.kzgo_rand_blob.restype = None
        _lib = L
    return _lib


def load_trusted_setup(path=DEFAULT_SETUP):
    """reference trusted_setup.go:23-27 JSON schema -> raw compressed point bytes."""
    with open(path) as fh:
        js = json.load(fh)
    unhex = lambda xs: b"".join(bytes.fromhex(x[2:]) for x in xs)
    return unhex(js["g1_monomial"]), unhex(js["g1_lagrange"]), unhex(js["g2_monomial"]), len(js["g2_monomial"])


def _buf(n):
    return C.create_string_buffer(n)


class OracleContext:
    """CPU mirror of goethkzg.Context (reference api.go:17-28)."""

    def __init__(self, setup_path=DEFAULT_SETUP, with_fk20=True):
        L = lib()
        g1m, g1l, g2m, n_g2 = load_trusted_setup(setup_path)
        h = C.c_void_p()
        rc = L.kzgo_ctx_new(g1m, g1l, g2m, n_g2, int(with_fk20), C.byref(h))
        if rc:
            raise RuntimeError(f"kzgo_ctx_new failed rc={rc}")
        self._h = h
        self._L = L

    def close(self):
        if self._h:
            self._L.kzgo_ctx_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- EIP-4844 (reference prove.go, verify.go)
    def blob_to_kzg_commitment(self, blob):
        if len(blob) != BLOB_BYTES:
            return 17, None
        out = _buf(48)
        rc = self._L.kzgo_blob_to_kzg_commitment(self._h, bytes(blob), out)
        return rc, (out.raw if rc == 0 else None)

    def blob_to_kzg_commitment_batch(self, blobs_concat, n, threads=0):
        """OpenMP over blobs; threads = 0: the OpenMP default"""
        out = _buf(48 * n)
        st = (C.c_int * n)()
        rc = self._L.kzgo_blob_to_kzg_commitment_batch_threads(self._h, blobs_concat, n, out, C.cast(st, C.c_void_p), int(threads))
        return rc, out.raw, list(st)

    def compute_kzg_proof(self, blob, z):
        if len(blob) != BLOB_BYTES or len(z) != 32:
            return 17, None, None
        p, y = _buf(48), _buf(32)
        rc = self._L.kzgo_compute_kzg_proof(self._h, bytes(blob), bytes(z), p, y)
        return (rc, p.raw, y.raw) if rc == 0 else (rc, None, None)

    def compute_blob_kzg_proof(self, blob, commitment):
        if len(blob) != BLOB_BYTES or len(commitment) != 48:
            return 17, None
        p = _buf(48)
        rc = self._L.kzgo_compute_blob_kzg_proof(self._h, bytes(blob), bytes(commitment), p)
        return rc, (p.raw if rc == 0 else None)

    def compute_blob_kzg_proof_batch(self, blobs_concat, commitments_concat, n, threads=0):
        """OpenMP over (blob, commitment) pairs; threads = 0: the OpenMP default"""
        out = _buf(48 * n)
        st = (C.c_int * n)()
        rc = self._L.kzgo_compute_blob_kzg_proof_batch_threads(self._h, blobs_concat, commitments_concat, n, out, C.cast(st, C.c_void_p), int(threads))
        return rc, out.raw, list(st)

    def verify_kzg_proof(self, commitment, z, y, proof):
        if len(commitment) != 48 or len(z) != 32 or len(y) != 32 or len(proof) != 48:
            return 17
        return self._L.kzgo_verify_kzg_proof(self._h, bytes(commitment), bytes(z), bytes(y), bytes(proof))

    def verify_blob_kzg_proof(self, blob, commitment, proof):
        if len(blob) != BLOB_BYTES or len(commitment) != 48 or len(proof) != 48:
            return 17
        return self._L.kzgo_verify_blob_kzg_proof(self._h, bytes(blob), bytes(commitment), bytes(proof))

    def verify_blob_kzg_proof_batch(self, blobs, commitments, proofs):
        if any(len(b) != BLOB_BYTES for b in blobs) or any(len(c) != 48 for c in commitments) or any(len(p) != 48 for p in proofs):
            return 17
        return self._L.kzgo_verify_blob_kzg_proof_batch(
            self._h, b"".join(blobs), len(blobs), b"".join(commitments), len(commitments), b"".join(proofs), len(proofs))

    # ---- sharded batch verification (mirror of the product's partial/finish pair)
    def verify_blob_batch_partial(self, blobs, commitments, proofs, offset, r32):
        rec = _buf(512)
        rc = self._L.kzgo_verify_blob_batch_partial(self._h, b"".join(blobs), b"".join(commitments), b"".join(proofs), len(blobs), offset, bytes(r32), rec)
        return rc, rec.raw

    def verify_blob_batch_finish(self, records):
        return self._L.kzgo_verify_blob_batch_finish(self._h, b"".join(records), len(records))

    def verify_cell_batch_partial(self, commitments, cell_indices, cells, proofs, offset, r32):
        n = len(commitments)
        assert n == len(cell_indices) == len(cells) == len(proofs)
        idx = (C.c_uint64 * max(1, n))(*cell_indices)
        rec = _buf(512)
        rc = self._L.kzgo_verify_cell_batch_partial(self._h, b"".join(commitments), n, idx, b"".join(cells), b"".join(proofs), offset, bytes(r32), rec)
        return rc, rec.raw

    def verify_cell_batch_finish(self, records):
        return self._L.kzgo_verify_cell_batch_finish(self._h, b"".join(records), len(records))

    # ---- EIP-7594 (reference api_eip7594.go, api_eip.go)
    def compute_cells(self, blob):
        if len(blob) != BLOB_BYTES:
            return 17, None
        cells = _buf(N_CELLS * CELL_BYTES)
        rc = self._L.kzgo_compute_cells(self._h, bytes(blob), cells)
        return rc, (cells.raw if rc == 0 else None)

    def compute_cells_and_kzg_proofs(self, blob):
        if len(blob) != BLOB_BYTES:
            return 17, None, None
        cells, proofs = _buf(N_CELLS * CELL_BYTES), _buf(N_CELLS * 48)
        rc = self._L.kzgo_compute_cells_and_kzg_proofs(self._h, bytes(blob), cells, proofs)
        return (rc, cells.raw, proofs.raw) if rc == 0 else (rc, None, None)

    def recover_cells_and_kzg_proofs(self, cell_ids, cells):
        if any(len(c) != CELL_BYTES for c in cells):
            return 17, None, None
        ids = (C.c_uint64 * max(1, len(cell_ids)))(*cell_ids)
        oc, op = _buf(N_CELLS * CELL_BYTES), _buf(N_CELLS * 48)
        rc = self._L.kzgo_recover_cells_and_kzg_proofs(self._h, C.cast(ids, C.c_void_p), len(cell_ids), b"".join(cells), len(cells), oc, op)
        return (rc, oc.raw, op.raw) if rc == 0 else (rc, None, None)

    def recover_cells(self, cell_ids, cells):
        if any(len(c) != CELL_BYTES for c in cells):
            return 17, None
        ids = (C.c_uint64 * max(1, len(cell_ids)))(*cell_ids)
        oc = _buf(N_CELLS * CELL_BYTES)
        rc = self._L.kzgo_recover_cells(self._h, C.cast(ids, C.c_void_p), len(cell_ids), b"".join(cells), len(cells), oc)
        return rc, (oc.raw if rc == 0 else None)

    def verify_cell_kzg_proof_batch(self, commitments, cell_indices, cells, proofs):
        if any(len(c) != 48 for c in commitments) or any(len(c) != CELL_BYTES for c in cells) or any(len(p) != 48 for p in proofs):
            return 17
        idx = (C.c_uint64 * max(1, len(cell_indices)))(*cell_indices)
        return self._L.kzgo_verify_cell_kzg_proof_batch(
            self._h, b"".join(commitments), len(commitments), C.cast(idx, C.c_void_p), len(cell_indices),
            b"".join(cells), len(cells), b"".join(proofs), len(proofs))

    def evaluate_blob(self, blob, z):
        y = _buf(32)
        rc = self._L.kzgo_evaluate_blob(self._h, bytes(blob), bytes(z), y)
        return rc, y.raw


# ---- free functions (unit-level probes)
def fr_fft(vals: bytes, inverse=False):
    n = len(vals) // 32
    b = C.create_string_buffer(vals, len(vals))
    rc = lib().kzgo_fr_fft(b, n, int(inverse))
    return rc, b.raw


def fr_coset_fft(vals: bytes, gen32: bytes, inverse=False):
    n = len(vals) // 32
    b = C.create_string_buffer(vals, len(vals))
    rc = lib().kzgo_fr_coset_fft(b, n, gen32, int(inverse))
    return rc, b.raw


def g1_msm(scalars: bytes, points: bytes, naive=False):
    n = len(scalars) // 32
    out = _buf(48)
    f = lib().kzgo_g1_msm_naive if naive else lib().kzgo_g1_msm
    rc = f(scalars, points, n, out)
    return rc, out.raw


def g1_validate(p48: bytes):
    return lib().kzgo_g1_validate(p48)


def g1_gen_mul(k: int):
    out = _buf(48)
    lib().kzgo_g1_gen_mul((k % (1 << 256)).to_bytes(32, "big"), out)
    return out.raw


def compute_challenge(blob: bytes, commitment: bytes):
    out = _buf(32)
    lib().kzgo_compute_challenge(blob, commitment, out)
    return out.raw


def sha256(data: bytes):
    out = _buf(32)
    lib().kzgo_sha256(data, len(data), out)
    return out.raw


def rand_blob(seed: int):
    """GetRandBlob (reference bench_test.go:38-46)."""
    out = _buf(BLOB_BYTES)
    lib().kzgo_rand_blob(seed, out)
    return out.raw
