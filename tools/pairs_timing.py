#!/usr/bin/env python3
"""tools/pairs_timing.py -- one 256-blob commitment through the rejected k_commit_direct_acc_pairs (experiments/commit_pairs_kernels.hip) built
with -DMKZG_PAIRS_TIMING: the kernel prints, for a few wavefronts, how its time splits into pass 1 (denominators), the inversion, pass 2
(affine sums) and pass 3 (mixed additions).
    tools/build_variant.sh pairs_timing "-DMKZG_AFFINE_PAIRS -DMKZG_PAIRS_TIMING" "api.hip api_verify.hip api_cells.hip api_multi.hip api_exchange.hip msm_kernels.hip" experiments/commit_pairs_kernels.hip
    MI355KZG_LIB=build_ab/libmi355kzg_pairs_timing.so python3 tools/pairs_timing.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import load_product, fast_blobs  # noqa: E402


def main():
    import torch
    torch.cuda.init()
    kzg = load_product()
    ctx = kzg.Context(device=0)
    B = 256
    assert ctx.set_option("max_batch", B) == 0 and ctx.set_option("commit_direct_bits", 15) == 0 and ctx.set_option("commit_affine_pairs", 1) == 0
    d = torch.from_numpy(fast_blobs(B, 901).reshape(-1)).cuda()
    out = torch.zeros(B * 48, dtype=torch.uint8, device="cuda")
    st = torch.zeros(B, dtype=torch.int32, device="cuda")
    for _ in range(2):
        assert ctx.blob_to_kzg_commitment_batch_device(d.data_ptr(), B, out.data_ptr(), st.data_ptr(), 0) == 0
        torch.cuda.synchronize()
        print("--", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
