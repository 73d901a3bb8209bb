#!/usr/bin/env python3
"""Device-resident VerifyBlobKZGProofBatch, N = 1024: sustained blobs/s against the number of caller threads (one call in flight each)."""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import rand_blob, load_product  # noqa: E402


def main():
    import torch
    torch.cuda.init()
    kzg = load_product()
    ctx = kzg.Context(device=0)
    N = 1024
    base = [rand_blob(s) for s in range(32)]
    d_blobs = torch.frombuffer(bytearray(b"".join(base[i % 32] for i in range(N))), dtype=torch.uint8).cuda()
    d_cm = torch.zeros(N * 48, dtype=torch.uint8, device="cuda")
    d_pr = torch.zeros(N * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.zeros(N, dtype=torch.int32, device="cuda")
    assert ctx.blob_to_kzg_commitment_batch_device(d_blobs.data_ptr(), N, d_cm.data_ptr(), d_st.data_ptr(), 0) == 0
    assert ctx.compute_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), N, d_pr.data_ptr(), d_st.data_ptr(), 0) == 0
    torch.cuda.synchronize()
    out = {}
    for K in (1, 2, 3, 4, 6, 8, 12):
        reps = 6

        def caller(k):
            s = torch.cuda.Stream(priority=-1 if k % 2 == 0 else 0)
            for _ in range(reps):
                assert ctx.verify_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), d_pr.data_ptr(), N, s.cuda_stream) == 0

        for trial in range(2):
            ts = [threading.Thread(target=caller, args=(k,)) for k in range(K)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in ts:
                t.start()
            for t in ts:
                t.join()
            dt = time.perf_counter() - t0
        out[K] = round(K * reps * N / dt)
        print(K, out[K], flush=True)
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
