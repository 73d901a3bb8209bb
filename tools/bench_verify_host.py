#!/usr/bin/env python3
"""Host-pointer VerifyBlobKZGProofBatch (Go's []*Blob) at the C ABI, N = 1024: challenge hash on the device after the upload
(host_sha 0) against SHA-NI on the host pool while the blobs upload (host_sha 1); one caller and four callers.  GPU box."""
import ctypes
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import rand_blob, load_product, BLOB  # noqa: E402


def main():
    import torch
    torch.cuda.init()
    kzg = load_product()
    ctx = kzg.Context(device=0)
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    base = [rand_blob(s) for s in range(32)]
    d_blobs = torch.frombuffer(bytearray(b"".join(base)), dtype=torch.uint8).cuda()
    d_cm = torch.zeros(32 * 48, dtype=torch.uint8, device="cuda")
    d_pr = torch.zeros(32 * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.zeros(32, dtype=torch.int32, device="cuda")
    assert ctx.blob_to_kzg_commitment_batch_device(d_blobs.data_ptr(), 32, d_cm.data_ptr(), d_st.data_ptr(), 0) == 0
    assert ctx.compute_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), 32, d_pr.data_ptr(), d_st.data_ptr(), 0) == 0
    torch.cuda.synchronize()
    cm, pr = d_cm.cpu().numpy().tobytes(), d_pr.cpu().numpy().tobytes()
    keep = [ctypes.create_string_buffer(base[i % 32], BLOB) for i in range(N)]            # N separate allocations
    arr = (ctypes.c_char_p * N)(*[ctypes.cast(k, ctypes.c_char_p) for k in keep])
    cms = b"".join(cm[48 * (i % 32):48 * (i % 32) + 48] for i in range(N))
    prs = b"".join(pr[48 * (i % 32):48 * (i % 32) + 48] for i in range(N))
    v = min(700, N - 2)
    bad = prs[:48 * v] + prs[48 * (v + 1):48 * (v + 2)] + prs[48 * (v + 1):]
    L = kzg.lib()
    out = {}

    def caller(reps):
        for _ in range(reps):
            assert L.mkzg_verify_blob_kzg_proof_batch(ctx._h, arr, cms, prs, N, N, N) == 0

    for hs in (0, 1):
        ctx.set_option("host_sha", hs)
        assert L.mkzg_verify_blob_kzg_proof_batch(ctx._h, arr, cms, prs, N, N, N) == 0
        assert L.mkzg_verify_blob_kzg_proof_batch(ctx._h, arr, cms, bad, N, N, N) == 1
        reps = 5
        t0 = time.perf_counter()
        caller(reps)
        dt1 = (time.perf_counter() - t0) / reps
        for K in (2, 4):
            for trial in range(2):
                ts = [threading.Thread(target=caller, args=(reps,)) for _ in range(K)]
                t0 = time.perf_counter()
                for t in ts:
                    t.start()
                for t in ts:
                    t.join()
                dtk = (time.perf_counter() - t0) / (K * reps)
            out[f"host_sha{hs}_callers{K}"] = round(N / dtk)
        out[f"host_sha{hs}_one_caller"] = {"ms": round(dt1 * 1e3, 3), "blobs_per_s": round(N / dt1)}
        print(hs, out, flush=True)
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
