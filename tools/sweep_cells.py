#!/usr/bin/env python3
"""Device-resident ComputeCellsAndKZGProofs over batch sizes (a block carries 6-48 blobs): ms per call, blobs/s and the per-kernel
split, one synchronous caller."""
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
    if os.environ.get("FK20_BITS"):
        assert ctx.set_option("fk20_bits", int(os.environ["FK20_BITS"])) == 0
    if os.environ.get("FK20_SEGMENTS") is not None:
        assert ctx.set_option("fk20_segments", int(os.environ["FK20_SEGMENTS"])) == 0
    NB = int(os.environ.get("SWEEP_NB", "128"))
    if os.environ.get("CELLS_CHUNK"):
        assert ctx.set_option("cells_chunk", int(os.environ["CELLS_CHUNK"])) == 0
    blobs = [rand_blob(s % 8) for s in range(NB)]
    d_blobs = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).cuda()
    d_cells = torch.zeros(NB * 262144, dtype=torch.uint8, device="cuda")
    d_proofs = torch.zeros(NB * 128 * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.zeros(NB, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream

    def call(n):
        assert ctx.compute_cells_and_kzg_proofs_batch_device(d_blobs.data_ptr(), n, d_cells.data_ptr(), d_proofs.data_ptr(), d_st.data_ptr(), st) == 0

    call(64)
    torch.cuda.synchronize()
    sizes = [int(x) for x in os.environ["SWEEP_SIZES"].split(",")] if os.environ.get("SWEEP_SIZES") else (1, 3, 6, 8, 9, 12, 16, 17, 21, 24, 32, 42, 48, 64, 96, 128)
    for n in sizes:
        call(n)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            call(n)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 4
        ctx.profile_enable(True)
        ctx.profile_read(reset=True)
        call(n)
        torch.cuda.synchronize()
        prof = ctx.profile_read(reset=True)
        ctx.profile_enable(False)
        print(f"n={n:4d}  {dt * 1e3:7.3f} ms  {n / dt:8.0f} blobs/s   " + " ".join(f"{k[2:]}={v[0]:.2f}" for k, v in prof.items() if v[1]), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
