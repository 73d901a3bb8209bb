#!/usr/bin/env python3
"""ComputeCellsAndKZGProofs for a rocprofv3 kernel trace: a few single-blob calls (host pointers) and a few 64-blob device batches.
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_fk20 -- python3 tools/trace_fk20.py"""
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
    NB = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    blobs = [rand_blob(s) for s in range(NB)]
    for i in range(4):
        t0 = time.perf_counter()
        rc, cells, proofs = ctx.ComputeCellsAndKZGProofs(blobs[0])
        print(f"single {i}: rc={rc} {1e3 * (time.perf_counter() - t0):.3f} ms", flush=True)
    d_blobs = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).cuda()
    d_cells = torch.zeros(NB * 262144, dtype=torch.uint8, device="cuda")
    d_proofs = torch.zeros(NB * 128 * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.zeros(NB, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for i in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = ctx.compute_cells_and_kzg_proofs_batch_device(d_blobs.data_ptr(), NB, d_cells.data_ptr(), d_proofs.data_ptr(), d_st.data_ptr(), st)
        torch.cuda.synchronize()
        print(f"batch{NB} {i}: rc={rc} {1e3 * (time.perf_counter() - t0):.3f} ms", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
