#!/usr/bin/env python3
"""compute_cells_and_kzg_proofs_batch_device for several batch sizes, calls back to back (kernel trace: how k_fk20_toom_mul and the
addition networks scale with the number of blobs)."""
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
    NB = 128
    blobs = [rand_blob(s % 8) for s in range(NB)]
    d_blobs = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).cuda()
    d_cells = torch.zeros(NB * 262144, dtype=torch.uint8, device="cuda")
    d_proofs = torch.zeros(NB * 128 * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.zeros(NB, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for n in (64, 1, 1, 1, 2, 2, 8, 8, 16, 16, 32, 32, 64, 64, 128, 128):
        rc = ctx.compute_cells_and_kzg_proofs_batch_device(d_blobs.data_ptr(), n, d_cells.data_ptr(), d_proofs.data_ptr(), d_st.data_ptr(), st)
        assert rc == 0
    torch.cuda.synchronize()
    ctx.close()


if __name__ == "__main__":
    main()
