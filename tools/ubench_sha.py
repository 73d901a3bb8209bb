#!/usr/bin/env python3
"""tools/ubench_sha.py -- duration of the Fiat-Shamir hash kernel alone (k_challenge_sha256, 1024 transcripts) for ONE library build
(MI355KZG_LIB selects it): used to A/B variants of the consumer wavefront's round, including diagnostic builds whose digests are
wrong (the verdict of the call is ignored).  GPU box."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import rand_blob, load_product  # noqa: E402


def main():
    import torch
    torch.cuda.init()
    kzg = load_product()
    ctx = kzg.Context(device=0)
    ctx.set_option("commit_direct_bits", 0)
    ctx.set_option("verify_shape", 0)
    if os.environ.get("SHA_ALONE"):
        assert ctx.set_option("debug_serial", 1) == 0      # nothing else on the chip while the hash runs
    N = 1024
    base = [rand_blob(s) for s in range(8)]
    d_blobs = torch.frombuffer(bytearray(b"".join(base[i % 8] for i in range(N))), dtype=torch.uint8).cuda()
    d_cm = torch.zeros(N * 48, dtype=torch.uint8, device="cuda")
    d_pr = torch.zeros(N * 48, dtype=torch.uint8, device="cuda")
    d_cm[0::48] = 0xC0
    d_pr[0::48] = 0xC0                      # points at infinity: valid encodings, the verdict does not matter here
    torch.cuda.synchronize()
    ctx.verify_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), d_pr.data_ptr(), N, 0)
    ctx.profile_enable(True)
    ctx.profile_read(reset=True)
    reps = 5
    for _ in range(reps):
        ctx.verify_blob_kzg_proof_batch_device(d_blobs.data_ptr(), d_cm.data_ptr(), d_pr.data_ptr(), N, 0)
    prof = ctx.profile_read(reset=True)
    ms, n = prof["k_challenge_sha256"]
    print(f"{os.path.basename(kzg.LIB_PATH)} alone={bool(os.environ.get('SHA_ALONE'))}: k_challenge_sha256 {ms / n:.4f} ms over {n} launches")
    ctx.close()


if __name__ == "__main__":
    main()
