#!/usr/bin/env python3
"""k_fk20_msm duration in the compute path, varying what precedes it: nothing / an 8 MB host-to-device copy / a 17 MB device-to-host copy / a 1 ms idle gap."""
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
    NB = 64
    blobs = [rand_blob(s) for s in range(NB)]
    d_blobs = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).cuda()
    d_cells = torch.zeros(NB * 262144, dtype=torch.uint8, device="cuda")
    d_proofs = torch.zeros(NB * 128 * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.zeros(NB, dtype=torch.int32, device="cuda")
    h_pin = torch.empty(17 << 20, dtype=torch.uint8).pin_memory()
    d_tmp = torch.empty(17 << 20, dtype=torch.uint8, device="cuda")

    rcells = None

    def compute(pre):
        nonlocal rcells
        if pre.startswith("recover"):
            if rcells is None:
                rc, rcells = ctx.ComputeCells(blobs[0])
            ids = list(range(0, 128, 2))
            for _ in range(int(pre[7:] or 1)):
                ctx.RecoverCells(ids, [rcells[i] for i in ids])
        if pre == "h2d":
            d_tmp[:8 << 20].copy_(h_pin[:8 << 20], non_blocking=True)
        elif pre == "d2h":
            h_pin.copy_(d_tmp, non_blocking=True)
        elif pre == "gap":
            time.sleep(0.002)
        elif pre == "gap10":
            time.sleep(0.010)
        ctx.profile_enable(True)
        ctx.profile_read(reset=True)
        assert ctx.compute_cells_and_kzg_proofs_batch_device(d_blobs.data_ptr(), NB, d_cells.data_ptr(), d_proofs.data_ptr(), d_st.data_ptr(), 0) == 0
        torch.cuda.synchronize()
        p = ctx.profile_read(reset=True)
        return round(p["k_fk20_msm"][0], 3)

    compute("none")
    for pre in ("none", "recover1", "recover3", "gap10", "none"):
        print(pre, [compute(pre) for _ in range(4)], flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
