#!/usr/bin/env python3
"""A few device-resident VerifyBlobKZGProofBatch calls (N = 1024) for a rocprofv3 kernel + memory-copy trace:
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/trace_verify -- python3 tools/trace_verify.py"""
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
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    base = [rand_blob(s) for s in range(16)]
    d_blobs = torch.frombuffer(bytearray(b"".join(base[i % 16] for i in range(N))), dtype=torch.uint8).cuda()
    d_cm = torch.zeros(N * 48, dtype=torch.uint8, device="cuda")
    d_pr = torch.zeros(N * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.zeros(N, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert ctx.blob_to_kzg_commitment_batch_device(d_blobs.data_ptr(), N, d_cm.data_ptr(), d_st.data_ptr(), st) == 0
    assert ctx.compute_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), N, d_pr.data_ptr(), d_st.data_ptr(), st) == 0
    torch.cuda.synchronize()
    for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 5):
        t0 = time.perf_counter()
        rc = ctx.verify_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), d_pr.data_ptr(), N, st)
        print(f"verify {i}: rc={rc} {1e3 * (time.perf_counter() - t0):.3f} ms host_finish {ctx.last_host_finish_us():.0f} us", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
