#!/usr/bin/env python3
"""Device-resident BlobToKZGCommitment (direct table) and VerifyBlobKZGProofBatch, one synchronous caller, over batch sizes: ms per call
and items/s -- to look for cliffs between the shapes the library picks by size (DESIGN.md 4.1 "few blobs", 4.2)."""
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
    ctx.set_option("commit_direct_bits", 15)
    ctx.set_option("max_batch", 1024)
    NMAX = 4096
    base = [rand_blob(s) for s in range(16)]
    d_blobs = torch.frombuffer(bytearray(b"".join(base[i % 16] for i in range(NMAX))), dtype=torch.uint8).cuda()
    d_cm = torch.zeros(NMAX * 48, dtype=torch.uint8, device="cuda")
    d_pr = torch.zeros(NMAX * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.zeros(NMAX, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert ctx.blob_to_kzg_commitment_batch_device(d_blobs.data_ptr(), NMAX, d_cm.data_ptr(), d_st.data_ptr(), st) == 0
    assert ctx.compute_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), NMAX, d_pr.data_ptr(), d_st.data_ptr(), st) == 0
    torch.cuda.synchronize()

    def timeit(fn, reps=6):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    d_out = torch.zeros(NMAX * 48, dtype=torch.uint8, device="cuda")
    print("commit (direct table c=15)")
    sizes = [int(x) for x in os.environ["SWEEP_SIZES"].split(",")] if os.environ.get("SWEEP_SIZES") else (1, 2, 8, 32, 33, 64, 65, 128, 129, 160, 192, 224, 240, 255, 256, 257, 512, 1024)
    for n in sizes:
        dt = timeit(lambda: ctx.blob_to_kzg_commitment_batch_device(d_blobs.data_ptr(), n, d_out.data_ptr(), d_st.data_ptr(), st))
        print(f"  n={n:5d}  {dt * 1e3:8.3f} ms  {n / dt:10.0f} blobs/s", flush=True)
    if os.environ.get("SWEEP_COMMIT_ONLY"):
        ctx.close()
        return
    print("verify (device-resident)")
    for n in (1, 8, 32, 48, 64, 96, 128, 256, 512, 1024, 2048, 4096):
        dt = timeit(lambda: ctx.verify_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), d_pr.data_ptr(), n, st), 4)
        ctx.profile_enable(True)
        ctx.profile_read(reset=True)
        ctx.verify_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), d_pr.data_ptr(), n, st)
        prof = ctx.profile_read(reset=True)
        ctx.profile_enable(False)
        print(f"  n={n:5d}  {dt * 1e3:8.3f} ms  {n / dt:10.0f} blobs/s   " + " ".join(f"{k[2:]}={v[0]:.2f}" for k, v in prof.items() if v[1]), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
