#!/usr/bin/env python3
"""VerifyCellKZGProofBatch over 64 x 128 = 8192 cells at the C ABI, a few calls, for a kernel + memory-copy trace."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import rand_blob, load_product  # noqa: E402


def main():
    import torch
    torch.cuda.init()
    kzg = load_product()
    ctx = kzg.Context(device=0)
    NB = 64
    blobs = [rand_blob(s % 8) for s in range(NB)]
    d_blobs = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).cuda()
    d_cells = torch.zeros(NB * 262144, dtype=torch.uint8, device="cuda")
    d_proofs = torch.zeros(NB * 128 * 48, dtype=torch.uint8, device="cuda")
    d_cm = torch.zeros(NB * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.zeros(NB, dtype=torch.int32, device="cuda")
    assert ctx.compute_cells_and_kzg_proofs_batch_device(d_blobs.data_ptr(), NB, d_cells.data_ptr(), d_proofs.data_ptr(), d_st.data_ptr(), 0) == 0
    assert ctx.blob_to_kzg_commitment_batch_device(d_blobs.data_ptr(), NB, d_cm.data_ptr(), d_st.data_ptr(), 0) == 0
    torch.cuda.synchronize()
    cms, cells, proofs = d_cm.cpu().numpy().tobytes(), d_cells.cpu().numpy().tobytes(), d_proofs.cpu().numpy().tobytes()
    n = NB * 128
    keep = [ctypes.create_string_buffer(cells[i * 2048:(i + 1) * 2048], 2048) for i in range(n)]
    arr = (ctypes.c_char_p * n)(*[ctypes.cast(k, ctypes.c_char_p) for k in keep])
    idx = (ctypes.c_uint64 * n)(*[i % 128 for i in range(n)])
    cj = b"".join(cms[48 * (i // 128):48 * (i // 128) + 48] for i in range(n))
    L = kzg.lib()
    for i in range(5):
        t0 = time.perf_counter()
        rc = L.mkzg_verify_cell_kzg_proof_batch(ctx._h, cj, n, idx, n, arr, n, proofs, n)
        print(f"call {i}: rc={rc} {1e3 * (time.perf_counter() - t0):.3f} ms", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
