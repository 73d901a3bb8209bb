"""go-eth-kzg_amd -- MI355X-native KZG engine: thin ctypes binding over libmi355kzg.so.

This module is plumbing for tests and bench.py only: the product is the C-ABI shared library
(include/mi355kzg.h) that a cgo `Context` wrapper binds.  The directory name
contains a hyphen, so load it with `importlib` (see `tests/conftest.py::load_product`) or via
`__graft_entry__.load_product()`.

There is NO CPU fallback here and nothing from oracle/ is imported: if the HIP library is
missing, or no GPU is present, construction fails loudly.

`Context` mirrors go-eth-kzg's Context method set (reference api.go:17-28, prove.go, verify.go,
api_eip7594.go); methods return (rc, outputs...) with rc numbered as include/mi355kzg.h.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI355KZG_LIB") or os.path.join(HERE, "libmi355kzg.so")      # override: A/B builds while tuning
DEFAULT_SETUP = os.path.join(HERE, "trusted_setup.json")

BYTES_PER_BLOB = 131072
BYTES_PER_CELL = 2048
CELLS_PER_EXT_BLOB = 128

OK = 0
VERIFY_FAILED = 1
ERR_BATCH_LENGTH = 2
ERR_NON_CANONICAL_SCALAR = 3
ERR_INVALID_CELL_ID = 4
ERR_NIL_INPUT = 6
ERR_TOO_MANY_GOROUTINES = 14
ERR_INVALID_POLY_SIZE = 15
ERR_BAD_ARGS = 17
ERR_HIP = 100

EXPORTED_SYMBOLS = [
    "mkzg_ctx_new", "mkzg_ctx_free", "mkzg_ctx_prepare", "mkzg_ctx_set_option", "mkzg_ctx_get_option",
    "mkzg_blob_to_kzg_commitment", "mkzg_blob_to_kzg_commitment_batch", "mkzg_blob_to_kzg_commitment_batch_device",
    "mkzg_commit_lagrange", "mkzg_profile_enable", "mkzg_profile_read", "mkzg_version",
    "mkzg_compute_kzg_proof", "mkzg_compute_blob_kzg_proof", "mkzg_compute_blob_kzg_proof_batch_device",
    "mkzg_verify_kzg_proof", "mkzg_verify_blob_kzg_proof", "mkzg_verify_blob_kzg_proof_batch", "mkzg_verify_blob_kzg_proof_batch_device",
    "mkzg_last_host_finish_us", "mkzg_valu_calibrate", "mkzg_compute_cells", "mkzg_compute_cells_batch_device",
    "mkzg_compute_cells_and_kzg_proofs", "mkzg_compute_cells_and_kzg_proofs_batch_device", "mkzg_verify_cell_kzg_proof_batch", "mkzg_verify_cell_kzg_proof_batch_partial", "mkzg_verify_cell_kzg_proof_batch_finish", "mkzg_recover_cells_and_kzg_proofs_batch", "mkzg_msm_g1",
    "mkzg_recover_cells_and_kzg_proofs", "mkzg_recover_cells",
    "mkzg_verify_blob_kzg_proof_batch_partial_device", "mkzg_verify_blob_kzg_proof_batch_finish",
    "mkzg_recover_cells_and_kzg_proofs_batch_device",
    "mkzg_evaluate_blobs_device", "mkzg_verify_cell_kzg_proof_batch_device", "mkzg_verify_cell_kzg_proof_batch_partial_device",
    "mkzg_comm_unique_id", "mkzg_comm_init", "mkzg_comm_from_nccl", "mkzg_comm_free", "mkzg_comm_last_exchange_us",
    "mkzg_verify_blob_kzg_proof_batch_exchange", "mkzg_verify_blob_kzg_proof_batch_exchange_device", "mkzg_verify_cell_kzg_proof_batch_exchange", "mkzg_verify_cell_kzg_proof_batch_exchange_device",
]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char_p), ("total_ms", C.c_double), ("launches", C.c_long)]


_lib = None


def lib():
    """dlopen libmi355kzg.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python go-eth-kzg_amd/build.py` (hipcc, gfx950). There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int
        L.mkzg_ctx_new.argtypes = [C.c_char_p, sz, C.POINTER(i32), i32, C.POINTER(vp)]
        L.mkzg_ctx_free.argtypes = [vp]
        L.mkzg_ctx_free.restype = None
        L.mkzg_ctx_set_option.argtypes = [vp, C.c_char_p, C.c_long]
        L.mkzg_ctx_prepare.argtypes = [vp, C.c_uint]
        L.mkzg_ctx_prepare.restype = C.c_int
        L.mkzg_ctx_get_option.argtypes = [vp, C.c_char_p, C.POINTER(C.c_long)]
        L.mkzg_blob_to_kzg_commitment.argtypes = [vp, C.c_char_p, i32, C.c_char_p]
        L.mkzg_blob_to_kzg_commitment_batch.argtypes = [vp, C.POINTER(C.c_char_p), sz, C.c_char_p, C.POINTER(i32)]
        L.mkzg_blob_to_kzg_commitment_batch_device.argtypes = [vp, vp, sz, vp, vp, vp]
        L.mkzg_commit_lagrange.argtypes = [vp, C.c_char_p, sz, i32, C.c_char_p]
        L.mkzg_compute_kzg_proof.argtypes = [vp, C.c_char_p, C.c_char_p, i32, C.c_char_p, C.c_char_p]
        L.mkzg_compute_blob_kzg_proof.argtypes = [vp, C.c_char_p, C.c_char_p, i32, C.c_char_p]
        L.mkzg_compute_blob_kzg_proof_batch_device.argtypes = [vp, vp, vp, sz, vp, vp, vp]
        L.mkzg_verify_kzg_proof.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p]
        L.mkzg_verify_blob_kzg_proof.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_char_p]
        L.mkzg_verify_blob_kzg_proof_batch.argtypes = [vp, C.POINTER(C.c_char_p), C.c_char_p, C.c_char_p, sz, sz, sz]
        L.mkzg_verify_blob_kzg_proof_batch_device.argtypes = [vp, vp, vp, vp, sz, vp]
        L.mkzg_compute_cells.argtypes = [vp, C.c_char_p, i32, C.c_char_p]
        L.mkzg_compute_cells_batch_device.argtypes = [vp, vp, sz, vp, vp, vp]
        L.mkzg_compute_cells_and_kzg_proofs.argtypes = [vp, C.c_char_p, i32, C.c_char_p, C.c_char_p]
        L.mkzg_compute_cells_and_kzg_proofs_batch_device.argtypes = [vp, vp, sz, vp, vp, vp, vp]
        L.mkzg_verify_cell_kzg_proof_batch.argtypes = [vp, C.c_char_p, sz, C.POINTER(C.c_uint64), sz, C.POINTER(C.c_char_p), sz, C.c_char_p, sz]
        L.mkzg_verify_cell_kzg_proof_batch_partial.argtypes = [vp, C.c_char_p, sz, C.POINTER(C.c_uint64), sz, C.POINTER(C.c_char_p), sz, C.c_char_p, sz, C.c_uint64,
                                                               C.c_char_p, C.c_char_p]
        L.mkzg_verify_cell_kzg_proof_batch_finish.argtypes = [vp, C.c_char_p, sz]
        L.mkzg_msm_g1.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), sz, C.POINTER(C.c_uint64)]
        L.mkzg_recover_cells_and_kzg_proofs_batch.argtypes = [vp, sz, C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(sz), C.POINTER(C.POINTER(C.c_char_p)), C.POINTER(sz),
                                                              C.c_char_p, C.c_char_p, C.POINTER(i32)]
        L.mkzg_recover_cells_and_kzg_proofs.argtypes = [vp, C.POINTER(C.c_uint64), sz, C.POINTER(C.c_char_p), sz, i32, C.c_char_p, C.c_char_p]
        L.mkzg_recover_cells.argtypes = [vp, C.POINTER(C.c_uint64), sz, C.POINTER(C.c_char_p), sz, i32, C.c_char_p]
        L.mkzg_comm_unique_id.argtypes = [C.c_char_p]
        L.mkzg_comm_init.argtypes = [vp, C.c_char_p, i32, i32, C.POINTER(vp)]
        L.mkzg_comm_from_nccl.argtypes = [vp, vp, i32, i32, C.POINTER(vp)]
        L.mkzg_comm_free.argtypes = [vp]
        L.mkzg_comm_free.restype = None
        L.mkzg_comm_last_exchange_us.argtypes = [vp]
        L.mkzg_comm_last_exchange_us.restype = C.c_double
        L.mkzg_verify_blob_kzg_proof_batch_exchange.argtypes = [vp, vp, C.c_char_p, C.c_char_p, sz, C.c_uint64]
        L.mkzg_verify_blob_kzg_proof_batch_exchange_device.argtypes = [vp, vp, vp, vp, sz, C.c_uint64, vp]
        L.mkzg_verify_cell_kzg_proof_batch_exchange.argtypes = [vp, C.c_char_p, vp, vp, C.c_char_p, sz, C.c_uint64]
        if hasattr(L, "mkzg_verify_cell_kzg_proof_batch_exchange_device") or "MI355KZG_LIB" not in os.environ:      # (an older build named for an A/B run may lack it)
            L.mkzg_verify_cell_kzg_proof_batch_exchange_device.argtypes = [vp, C.c_char_p, vp, vp, C.c_char_p, sz, C.c_uint64, vp]
        L.mkzg_verify_cell_kzg_proof_batch_device.argtypes = [vp, C.c_char_p, vp, vp, C.c_char_p, sz, vp]
        L.mkzg_verify_cell_kzg_proof_batch_partial_device.argtypes = [vp, C.c_char_p, vp, vp, C.c_char_p, sz, C.c_uint64, C.c_char_p, C.c_char_p, vp]
        L.mkzg_recover_cells_and_kzg_proofs_batch_device.argtypes = [vp, sz, vp, vp, vp, vp, vp, C.POINTER(C.c_int), vp]
        L.mkzg_evaluate_blobs_device.argtypes = [vp, vp, C.c_char_p, sz, i32, C.c_char_p, C.POINTER(C.c_int), vp]
        L.mkzg_verify_blob_kzg_proof_batch_partial_device.argtypes = [vp, vp, vp, vp, sz, C.c_uint64, C.c_char_p, C.c_char_p, vp]
        L.mkzg_verify_blob_kzg_proof_batch_finish.argtypes = [vp, C.c_char_p, sz]
        L.mkzg_profile_enable.argtypes = [vp, i32]
        L.mkzg_profile_read.argtypes = [vp, C.POINTER(KernelTime), i32, i32]
        L.mkzg_version.restype = C.c_char_p
        L.mkzg_last_host_finish_us.argtypes = [vp]
        L.mkzg_last_host_finish_us.restype = C.c_double
        L.mkzg_valu_calibrate.argtypes = [vp, C.POINTER(C.c_double)]
        _lib = L
    return _lib


class Context:
    """Mirror of goethkzg.Context backed by one MI355X."""

    def __init__(self, setup_path=DEFAULT_SETUP, device=None, devices=None):
        """device: one HIP device id; devices: a list of ids for a multi-device context (include/mi355kzg.h)"""
        L = lib()
        with open(setup_path, "rb") as fh:
            js = fh.read()
        h = C.c_void_p()
        if devices is not None:
            ids = (C.c_int * max(1, len(devices)))(*devices)
            rc = L.mkzg_ctx_new(js, len(js), ids, len(devices), C.byref(h))
        elif device is None:
            rc = L.mkzg_ctx_new(js, len(js), None, 0, C.byref(h))
        else:
            ids = (C.c_int * 1)(device)
            rc = L.mkzg_ctx_new(js, len(js), ids, 1, C.byref(h))
        if rc:
            raise RuntimeError(f"mkzg_ctx_new failed rc={rc} (HIP device required; no CPU fallback)")
        self._h, self._L = h, L

    def close(self):
        if getattr(self, "_h", None):
            self._L.mkzg_ctx_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def prepare(self, what=3):
        """mkzg_ctx_prepare: build the big tables now (1 commitments, 2 cells / recovery, 3 both) as NewContext4096 does at construction (api.go:90-149)"""
        return self._L.mkzg_ctx_prepare(self._h, C.c_uint(what))

    def set_option(self, key, value):
        return self._L.mkzg_ctx_set_option(self._h, key.encode(), int(value))

    def get_option(self, key):
        v = C.c_long(0)
        rc = self._L.mkzg_ctx_get_option(self._h, key.encode(), C.byref(v))
        return v.value if rc == 0 else None

    # ---- EIP-4844 -------------------------------------------------------------------------
    def BlobToKZGCommitment(self, blob, num_goroutines=0):
        """reference prove.go:13"""
        if blob is None:
            return self._L.mkzg_blob_to_kzg_commitment(self._h, None, num_goroutines, C.create_string_buffer(48)), None
        if len(blob) != BYTES_PER_BLOB:
            return ERR_BAD_ARGS, None
        out = C.create_string_buffer(48)
        rc = self._L.mkzg_blob_to_kzg_commitment(self._h, bytes(blob), num_goroutines, out)
        return rc, (out.raw if rc == 0 else None)

    def BlobToKZGCommitmentBatch(self, blobs):
        n = len(blobs)
        if any(len(b) != BYTES_PER_BLOB for b in blobs):
            return ERR_BAD_ARGS, None, None
        keep = [bytes(b) for b in blobs]
        arr = (C.c_char_p * max(1, n))(*keep)
        out = C.create_string_buffer(48 * max(1, n))
        st = (C.c_int * max(1, n))()
        rc = self._L.mkzg_blob_to_kzg_commitment_batch(self._h, arr, n, out, st)
        raw = out.raw
        return rc, [raw[48 * i:48 * i + 48] for i in range(n)], list(st)[:n]

    def blob_to_kzg_commitment_batch_device(self, d_blobs_ptr, n, d_out_ptr, d_status_ptr, stream=0):
        """raw device pointers (ints) + hipStream_t handle; asynchronous"""
        return self._L.mkzg_blob_to_kzg_commitment_batch_device(self._h, d_blobs_ptr, n, d_out_ptr, d_status_ptr, stream)

    def CommitLagrange(self, scalars32, num_goroutines=0):
        """kzg.CommitKey.Commit (reference internal/kzg/srs.go:56) on raw big-endian scalars"""
        n = len(scalars32) // 32
        out = C.create_string_buffer(48)
        rc = self._L.mkzg_commit_lagrange(self._h, bytes(scalars32), n, num_goroutines, out)
        return rc, (out.raw if rc == 0 else None)

    def ComputeKZGProof(self, blob, z, num_goroutines=0):
        """reference prove.go:85 -> (rc, proof, y)"""
        if blob is not None and len(blob) != BYTES_PER_BLOB or len(z) != 32:
            return ERR_BAD_ARGS, None, None
        proof, y = C.create_string_buffer(48), C.create_string_buffer(32)
        rc = self._L.mkzg_compute_kzg_proof(self._h, None if blob is None else bytes(blob), bytes(z), num_goroutines, proof, y)
        return (rc, proof.raw, y.raw) if rc == 0 else (rc, None, None)

    def ComputeBlobKZGProof(self, blob, commitment, num_goroutines=0):
        """reference prove.go:46"""
        if blob is not None and len(blob) != BYTES_PER_BLOB or len(commitment) != 48:
            return ERR_BAD_ARGS, None
        out = C.create_string_buffer(48)
        rc = self._L.mkzg_compute_blob_kzg_proof(self._h, None if blob is None else bytes(blob), bytes(commitment), num_goroutines, out)
        return rc, (out.raw if rc == 0 else None)

    def compute_blob_kzg_proof_batch_device(self, d_blobs_ptr, d_commitments_ptr, n, d_proofs_ptr, d_status_ptr, stream=0):
        return self._L.mkzg_compute_blob_kzg_proof_batch_device(self._h, d_blobs_ptr, d_commitments_ptr, n, d_proofs_ptr, d_status_ptr, stream)

    def VerifyKZGProof(self, commitment, z, y, proof):
        """reference verify.go:12"""
        if len(commitment) != 48 or len(z) != 32 or len(y) != 32 or len(proof) != 48:
            return ERR_BAD_ARGS
        return self._L.mkzg_verify_kzg_proof(self._h, bytes(commitment), bytes(z), bytes(y), bytes(proof))

    def VerifyBlobKZGProof(self, blob, commitment, proof):
        """reference verify.go:48"""
        if blob is not None and len(blob) != BYTES_PER_BLOB or len(commitment) != 48 or len(proof) != 48:
            return ERR_BAD_ARGS
        return self._L.mkzg_verify_blob_kzg_proof(self._h, None if blob is None else bytes(blob), bytes(commitment), bytes(proof))

    def VerifyBlobKZGProofBatch(self, blobs, commitments, proofs):
        """reference verify.go:88 (lengths of the three lists may differ -> ErrBatchLengthCheck)"""
        if any(b is not None and len(b) != BYTES_PER_BLOB for b in blobs) or any(len(x) != 48 for x in commitments) or any(len(x) != 48 for x in proofs):
            return ERR_BAD_ARGS
        keep = [None if b is None else bytes(b) for b in blobs]
        arr = (C.c_char_p * max(1, len(blobs)))(*keep)
        return self._L.mkzg_verify_blob_kzg_proof_batch(self._h, arr, b"".join(commitments), b"".join(proofs), len(blobs), len(commitments), len(proofs))

    def verify_blob_kzg_proof_batch_device(self, d_blobs_ptr, d_commitments_ptr, d_proofs_ptr, n, stream=0):
        return self._L.mkzg_verify_blob_kzg_proof_batch_device(self._h, d_blobs_ptr, d_commitments_ptr, d_proofs_ptr, n, stream)

    def evaluate_blobs_device(self, d_blobs_ptr, zs, form, stream=0):
        """Domain.EvaluateLagrangePolynomial (reference internal/domain/domain.go:193-235) for device-resident blobs at caller-chosen
        points through evaluation kernel `form` (0 coefficients + Horner, 1 product tree, 2 per-lane batch inversion)
        -> (rc, [y bytes], [status])"""
        n = len(zs)
        y = C.create_string_buffer(32 * max(1, n))
        st = (C.c_int * max(1, n))()
        rc = self._L.mkzg_evaluate_blobs_device(self._h, C.c_void_p(d_blobs_ptr), b"".join(zs), n, form, y, st, C.c_void_p(stream))
        return rc, [y.raw[32 * i:32 * i + 32] for i in range(n)], list(st)[:n]

    # ---- EIP-7594 -------------------------------------------------------------------------
    def ComputeCells(self, blob, num_goroutines=0):
        """reference api_eip7594.go:12 -> (rc, [128 cells])"""
        if blob is not None and len(blob) != BYTES_PER_BLOB:
            return ERR_BAD_ARGS, None
        out = C.create_string_buffer(CELLS_PER_EXT_BLOB * BYTES_PER_CELL)
        rc = self._L.mkzg_compute_cells(self._h, None if blob is None else bytes(blob), num_goroutines, out)
        if rc:
            return rc, None
        raw = out.raw                              # .raw copies the buffer: once, not per cell
        return 0, [raw[i * BYTES_PER_CELL:(i + 1) * BYTES_PER_CELL] for i in range(CELLS_PER_EXT_BLOB)]

    def ComputeCellsAndKZGProofs(self, blob, num_goroutines=0):
        """reference api_eip7594.go:28 -> (rc, [128 cells], [128 proofs])"""
        if blob is not None and len(blob) != BYTES_PER_BLOB:
            return ERR_BAD_ARGS, None, None
        cells = C.create_string_buffer(CELLS_PER_EXT_BLOB * BYTES_PER_CELL)
        proofs = C.create_string_buffer(CELLS_PER_EXT_BLOB * 48)
        rc = self._L.mkzg_compute_cells_and_kzg_proofs(self._h, None if blob is None else bytes(blob), num_goroutines, cells, proofs)
        if rc:
            return rc, None, None
        cr, pr = cells.raw, proofs.raw
        return (0, [cr[i * BYTES_PER_CELL:(i + 1) * BYTES_PER_CELL] for i in range(CELLS_PER_EXT_BLOB)], [pr[i * 48:(i + 1) * 48] for i in range(CELLS_PER_EXT_BLOB)])

    def compute_cells_and_kzg_proofs_batch_device(self, d_blobs_ptr, n, d_cells_ptr, d_proofs_ptr, d_status_ptr, stream=0):
        return self._L.mkzg_compute_cells_and_kzg_proofs_batch_device(self._h, d_blobs_ptr, n, d_cells_ptr, d_proofs_ptr, d_status_ptr, stream)

    def VerifyCellKZGProofBatch(self, commitments, cell_indices, cells, proofs):
        """reference api_eip7594.go:163"""
        if any(len(x) != 48 for x in commitments) or any(x is not None and len(x) != BYTES_PER_CELL for x in cells) or any(len(x) != 48 for x in proofs):
            return ERR_BAD_ARGS
        idx = (C.c_uint64 * max(1, len(cell_indices)))(*cell_indices)
        keep = [None if x is None else bytes(x) for x in cells]
        arr = (C.c_char_p * max(1, len(cells)))(*keep)
        return self._L.mkzg_verify_cell_kzg_proof_batch(self._h, b"".join(commitments), len(commitments), idx, len(cell_indices), arr, len(cells),
                                                        b"".join(proofs), len(proofs))

    def verify_cell_kzg_proof_batch_device(self, commitments, cell_indices, d_cells_ptr, proofs, stream=0):
        """VerifyCellKZGProofBatch with the cells contiguous in HBM (n x 2048 B at d_cells_ptr), the other three arrays on the host"""
        n = len(cell_indices)
        if len(commitments) != n or len(proofs) != n:
            return ERR_BATCH_LENGTH
        idx = (C.c_uint64 * max(1, n))(*cell_indices)
        return self._L.mkzg_verify_cell_kzg_proof_batch_device(self._h, b"".join(commitments), C.cast(idx, C.c_void_p), C.c_void_p(d_cells_ptr), b"".join(proofs), n, C.c_void_p(stream))

    def verify_cell_batch_partial_device(self, commitments, cell_indices, d_cells_ptr, proofs, offset, r32, stream=0):
        n = len(cell_indices)
        idx = (C.c_uint64 * max(1, n))(*cell_indices)
        rec = C.create_string_buffer(512)
        rc = self._L.mkzg_verify_cell_kzg_proof_batch_partial_device(self._h, b"".join(commitments), C.cast(idx, C.c_void_p), C.c_void_p(d_cells_ptr), b"".join(proofs), n, offset,
                                                                     bytes(r32), rec, C.c_void_p(stream))
        return rc, rec.raw

    def recover_batch_device(self, ids_per_blob, d_cells_ptr, d_cells_out_ptr, d_proofs_out_ptr, stream=0):
        """mkzg_recover_cells_and_kzg_proofs_batch_device: ids_per_blob = list of id lists; the supplied cells concatenated in HBM; results
        stay in HBM (d_proofs_out_ptr 0: cells only) -> (rc, [status per blob])"""
        nb = len(ids_per_blob)
        arrs = [(C.c_uint64 * max(1, len(x)))(*x) for x in ids_per_blob]
        pid = (C.POINTER(C.c_uint64) * max(1, nb))(*[C.cast(a, C.POINTER(C.c_uint64)) for a in arrs])
        nid = (C.c_size_t * max(1, nb))(*[len(x) for x in ids_per_blob])
        st = (C.c_int * max(1, nb))()
        rc = self._L.mkzg_recover_cells_and_kzg_proofs_batch_device(self._h, nb, C.cast(pid, C.c_void_p), C.cast(nid, C.c_void_p), C.c_void_p(d_cells_ptr), C.c_void_p(d_cells_out_ptr),
                                                                     C.c_void_p(d_proofs_out_ptr) if d_proofs_out_ptr else None, st, C.c_void_p(stream))
        return rc, list(st)[:nb]

    def verify_cell_batch_partial(self, commitments, cell_indices, cells, proofs, offset, r32):
        """this rank's share of a sharded VerifyCellKZGProofBatch -> (rc, 512-byte record); see include/mi355kzg.h"""
        idx = (C.c_uint64 * max(1, len(cell_indices)))(*cell_indices)
        keep = [None if x is None else bytes(x) for x in cells]
        arr = (C.c_char_p * max(1, len(cells)))(*keep)
        rec = C.create_string_buffer(512)
        rc = self._L.mkzg_verify_cell_kzg_proof_batch_partial(self._h, b"".join(commitments), len(commitments), idx, len(cell_indices), arr, len(cells),
                                                              b"".join(proofs), len(proofs), offset, bytes(r32), rec)
        return rc, rec.raw

    def verify_cell_batch_finish(self, records):
        return self._L.mkzg_verify_cell_kzg_proof_batch_finish(self._h, b"".join(records), len(records))

    def RecoverCellsAndComputeKZGProofs(self, cell_ids, cells, num_goroutines=0):
        """reference api_eip7594.go:144 -> (rc, [128 cells], [128 proofs])"""
        if any(x is not None and len(x) != BYTES_PER_CELL for x in cells):
            return ERR_BAD_ARGS, None, None
        ids = (C.c_uint64 * max(1, len(cell_ids)))(*cell_ids)
        keep = [None if x is None else bytes(x) for x in cells]
        arr = (C.c_char_p * max(1, len(cells)))(*keep)
        oc = C.create_string_buffer(CELLS_PER_EXT_BLOB * BYTES_PER_CELL)
        op = C.create_string_buffer(CELLS_PER_EXT_BLOB * 48)
        rc = self._L.mkzg_recover_cells_and_kzg_proofs(self._h, ids, len(cell_ids), arr, len(cells), num_goroutines, oc, op)
        if rc:
            return rc, None, None
        ocr, opr = oc.raw, op.raw
        return (0, [ocr[i * BYTES_PER_CELL:(i + 1) * BYTES_PER_CELL] for i in range(CELLS_PER_EXT_BLOB)], [opr[i * 48:(i + 1) * 48] for i in range(CELLS_PER_EXT_BLOB)])

    def RecoverCellsAndComputeKZGProofsBatch(self, jobs):
        """jobs: list of (cell_ids, cells) -> (rc, [(status, [128 cells], [128 proofs]) per job])"""
        n = len(jobs)
        id_arrs = [(C.c_uint64 * max(1, len(ids)))(*ids) for ids, _ in jobs]
        cell_arrs = [(C.c_char_p * max(1, len(cells)))(*[bytes(x) for x in cells]) for _, cells in jobs]
        pid = (C.POINTER(C.c_uint64) * max(1, n))(*[C.cast(a, C.POINTER(C.c_uint64)) for a in id_arrs])
        pcl = (C.POINTER(C.c_char_p) * max(1, n))(*[C.cast(a, C.POINTER(C.c_char_p)) for a in cell_arrs])
        nid = (C.c_size_t * max(1, n))(*[len(ids) for ids, _ in jobs])
        ncl = (C.c_size_t * max(1, n))(*[len(cells) for _, cells in jobs])
        oc = C.create_string_buffer(max(1, n) * CELLS_PER_EXT_BLOB * BYTES_PER_CELL)
        op = C.create_string_buffer(max(1, n) * CELLS_PER_EXT_BLOB * 48)
        st = (C.c_int32 * max(1, n))()
        rc = self._L.mkzg_recover_cells_and_kzg_proofs_batch(self._h, n, pid, nid, pcl, ncl, oc, op, st)
        res = []
        ocr, opr = oc.raw, op.raw              # .raw copies the whole buffer: take it once
        for b in range(n):
            if st[b]:
                res.append((st[b], None, None))
                continue
            co, po = b * CELLS_PER_EXT_BLOB * BYTES_PER_CELL, b * CELLS_PER_EXT_BLOB * 48
            res.append((0, [ocr[co + i * BYTES_PER_CELL:co + (i + 1) * BYTES_PER_CELL] for i in range(CELLS_PER_EXT_BLOB)],
                        [opr[po + i * 48:po + (i + 1) * 48] for i in range(CELLS_PER_EXT_BLOB)]))
        return rc, res

    def RecoverCells(self, cell_ids, cells, num_goroutines=0):
        """reference api_eip.go:8 -> (rc, [128 cells])"""
        if any(x is not None and len(x) != BYTES_PER_CELL for x in cells):
            return ERR_BAD_ARGS, None
        ids = (C.c_uint64 * max(1, len(cell_ids)))(*cell_ids)
        keep = [None if x is None else bytes(x) for x in cells]
        arr = (C.c_char_p * max(1, len(cells)))(*keep)
        oc = C.create_string_buffer(CELLS_PER_EXT_BLOB * BYTES_PER_CELL)
        rc = self._L.mkzg_recover_cells(self._h, ids, len(cell_ids), arr, len(cells), num_goroutines, oc)
        if rc:
            return rc, None
        ocr = oc.raw
        return 0, [ocr[i * BYTES_PER_CELL:(i + 1) * BYTES_PER_CELL] for i in range(CELLS_PER_EXT_BLOB)]

    # ---- sharded batch verification (see distributed.py)
    def verify_blob_batch_partial_device(self, d_blobs_ptr, d_commitments_ptr, d_proofs_ptr, n_local, offset, r32, stream=0):
        rec = C.create_string_buffer(512)
        rc = self._L.mkzg_verify_blob_kzg_proof_batch_partial_device(self._h, d_blobs_ptr, d_commitments_ptr, d_proofs_ptr, n_local, offset, bytes(r32), rec, stream)
        return rc, rec.raw

    def verify_blob_batch_finish(self, records):
        return self._L.mkzg_verify_blob_kzg_proof_batch_finish(self._h, b"".join(records), len(records))

    def last_host_finish_us(self):
        return self._L.mkzg_last_host_finish_us(self._h)

    # ---- profiling --------------------------------------------------------------------------
    def msm_g1(self, scalars_mont, points_mont):
        """multiexp.MultiExpG1 seam: scalars as ints in gnark's Montgomery form (s * 2^256 mod r), points as (X, Y) ints in
        Montgomery form (x * 2^384 mod p), (0, 0) = identity -> (rc, (X, Y))"""
        n = len(scalars_mont)
        sc = (C.c_uint64 * max(1, 4 * n))(*[(v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for v in scalars_mont for k in range(4)])
        pt = (C.c_uint64 * max(1, 12 * n))(*[(v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for xy in points_mont for v in xy for k in range(6)])
        out = (C.c_uint64 * 12)()
        rc = self._L.mkzg_msm_g1(self._h, sc, pt, n, out)
        x = sum(int(out[k]) << (64 * k) for k in range(6))
        y = sum(int(out[6 + k]) << (64 * k) for k in range(6))
        return rc, (x, y)

    def valu_calibrate(self):
        """sustained v_mad_u64_u32 wave-instructions per second of this GPU (diagnostic)"""
        v = C.c_double(0.0)
        rc = self._L.mkzg_valu_calibrate(self._h, C.byref(v))
        return v.value if rc == 0 else None

    def profile_enable(self, on=True):
        return self._L.mkzg_profile_enable(self._h, int(on))

    def profile_read(self, reset=True):
        recs = (KernelTime * 32)()
        n = self._L.mkzg_profile_read(self._h, recs, 32, int(reset))
        return {recs[i].name.decode(): (recs[i].total_ms, recs[i].launches) for i in range(n)}
