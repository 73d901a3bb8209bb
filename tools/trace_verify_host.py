#!/usr/bin/env python3
"""Three host-pointer VerifyBlobKZGProofBatch calls (N = 1024) for a kernel + memory-copy trace."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import rand_blob, load_product, BLOB  # noqa: E402


def main():
    import torch
    torch.cuda.init()
    kzg = load_product()
    ctx = kzg.Context(device=0)
    N = 1024
    base = [rand_blob(s) for s in range(32)]
    d_blobs = torch.frombuffer(bytearray(b"".join(base)), dtype=torch.uint8).cuda()
    d_cm = torch.zeros(32 * 48, dtype=torch.uint8, device="cuda")
    d_pr = torch.zeros(32 * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.zeros(32, dtype=torch.int32, device="cuda")
    assert ctx.blob_to_kzg_commitment_batch_device(d_blobs.data_ptr(), 32, d_cm.data_ptr(), d_st.data_ptr(), 0) == 0
    assert ctx.compute_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), 32, d_pr.data_ptr(), d_st.data_ptr(), 0) == 0
    torch.cuda.synchronize()
    cm, pr = d_cm.cpu().numpy().tobytes(), d_pr.cpu().numpy().tobytes()
    keep = [ctypes.create_string_buffer(base[i % 32], BLOB) for i in range(N)]
    arr = (ctypes.c_char_p * N)(*[ctypes.cast(k, ctypes.c_char_p) for k in keep])
    cms = b"".join(cm[48 * (i % 32):48 * (i % 32) + 48] for i in range(N))
    prs = b"".join(pr[48 * (i % 32):48 * (i % 32) + 48] for i in range(N))
    L = kzg.lib()
    for i in range(4):
        t0 = time.perf_counter()
        rc = L.mkzg_verify_blob_kzg_proof_batch(ctx._h, arr, cms, prs, N, N, N)
        print(f"call {i}: rc={rc} {1e3 * (time.perf_counter() - t0):.3f} ms", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
