#!/usr/bin/env python3
"""tools/ab_pairs.py -- round 6 A/B inside ONE process on ONE box: BlobToKZGCommitment of 256 device-resident blobs through the 15-bit direct
table with option commit_affine_pairs 0 (k_commit_direct_acc: 17 mixed additions per scalar) and 1 (k_commit_direct_acc_pairs: 8 affine
additions + 9 mixed additions per scalar), alternating.  Checks that both give the same 256 commitments for GetRandBlob blobs, a zero
blob, blobs of small scalars and of r - 1, then times steps over 1 and 4 streams as bench.py does and prints kernel times.
The pair kernel is a REJECTED experiment (go-eth-kzg_amd/csrc/experiments/commit_pairs_kernels.hip) and not in the product build:
    tools/build_variant.sh pairs "-DMKZG_AFFINE_PAIRS" "api.hip api_verify.hip api_cells.hip api_multi.hip api_exchange.hip msm_kernels.hip" experiments/commit_pairs_kernels.hip
    MI355KZG_LIB=build_ab/libmi355kzg_pairs.so python3 tools/ab_pairs.py        # result of round 6: profiles/r06_ab_pairs.txt"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import rand_blob, load_product, fast_blobs, R_MOD  # noqa: E402


def main():
    import torch
    torch.cuda.init()
    kzg = load_product()
    ctx = kzg.Context(device=0)
    B = 256
    assert ctx.set_option("max_batch", B) == 0
    assert ctx.set_option("commit_direct_bits", 15) == 0
    blobs = [rand_blob(s) for s in range(B - 6)]
    blobs.append(bytes(131072))                                                  # zero blob
    blobs.append(b"".join((i % 7).to_bytes(32, "big") for i in range(4096)))     # tiny scalars: mostly zero digits
    blobs.append(((R_MOD - 1).to_bytes(32, "big")) * 4096)                       # r - 1 everywhere
    blobs.append(((1 << 254).to_bytes(32, "big")) * 4096)                        # bit 254 set: the r - s recoding
    blobs.append(((1 << 15).to_bytes(32, "big")) * 4096)                         # one non-zero digit in window 1
    blobs.append((((1 << 14) | (1 << 29)).to_bytes(32, "big")) * 4096)           # digits 2^14 in windows 0 and 1 (signed recoding carries)
    ring = [torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).cuda()] + [torch.from_numpy(fast_blobs(B, 900 + i).reshape(-1)).cuda() for i in range(7)]
    NS = 4
    streams = [torch.cuda.Stream(priority=(-1 if i % 2 == 0 else 0)) for i in range(NS)]
    outs = [torch.zeros(B * 48, dtype=torch.uint8, device="cuda") for _ in range(NS)]
    sts = [torch.zeros(B, dtype=torch.int32, device="cuda") for _ in range(NS)]

    def run(batch, i=0):
        rc = ctx.blob_to_kzg_commitment_batch_device(batch.data_ptr(), B, outs[i].data_ptr(), sts[i].data_ptr(), streams[i].cuda_stream)
        assert rc == 0, rc

    ref = {}
    for mode in (0, 1):
        assert ctx.set_option("commit_affine_pairs", mode) == 0 and ctx.get_option("commit_affine_pairs") == mode
        got = []
        for q in range(len(ring)):
            run(ring[q])
            torch.cuda.synchronize()
            assert int(sts[0].abs().sum().item()) == 0
            got.append(outs[0].clone())
        ref[mode] = got
    same = [bool((a == b).all().item()) for a, b in zip(ref[0], ref[1])]
    print("same commitments, batch by batch:", same, flush=True)
    if not all(same):
        a, b = ref[0][0].cpu().numpy().reshape(B, 48), ref[1][0].cpu().numpy().reshape(B, 48)
        bad = [i for i in range(B) if (a[i] != b[i]).any()]
        print("differing blobs of batch 0:", bad[:20], "of", len(bad))
        sys.exit(1)

    def timed(ns, steps=20):
        for i in range(max(ns, 4)):
            run(ring[i % len(ring)], i % ns)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            run(ring[i % len(ring)], i % ns)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    for rep in range(3):
        for mode in (0, 1):
            assert ctx.set_option("commit_affine_pairs", mode) == 0
            t1, t4 = timed(1), timed(4)
            ctx.profile_enable(True)
            ctx.profile_read(reset=True)
            for q in range(3):
                run(ring[q])
                torch.cuda.synchronize()
            prof = ctx.profile_read(reset=True)
            ctx.profile_enable(False)
            ks = " ".join(f"{k}={v[0] / max(1, v[1]):.3f}" for k, v in prof.items() if v[1])
            print(f"pairs={mode}: 1 stream {t1:.3f} ms/step = {B / t1 * 1e3:8.0f} blobs/s   4 streams {t4:.3f} ms/step = {B / t4 * 1e3:8.0f} blobs/s   alone: {ks}", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
