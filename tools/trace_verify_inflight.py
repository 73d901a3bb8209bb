#!/usr/bin/env python3
"""K caller threads x R device-resident VerifyBlobKZGProofBatch calls (N = 1024) for a rocprofv3 kernel trace of the sustained leg:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_inflight -- python3 tools/trace_verify_inflight.py 8 6
    python tools/trace_busy.py gpurun_out/trace_inflight"""
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
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 6
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

    def caller(k, reps):
        s = torch.cuda.Stream(priority=-1 if k % 2 == 0 else 0)
        for _ in range(reps):
            assert ctx.verify_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), d_pr.data_ptr(), N, s.cuda_stream) == 0

    for trial, reps in enumerate((2, R)):
        ts = [threading.Thread(target=caller, args=(k, reps)) for k in range(K)]
        torch.cuda.synchronize()
        time.sleep(0.05)
        t0, c0 = time.perf_counter(), time.process_time()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        dt, cpu = time.perf_counter() - t0, time.process_time() - c0
        print(f"trial {trial}: {K} callers x {reps} calls: {K * reps * N / dt:.0f} blobs/s, {1e3 * dt / (K * reps):.3f} ms per call amortised; host CPUs busy {cpu / dt:.1f}", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
