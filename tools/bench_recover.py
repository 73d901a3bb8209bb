#!/usr/bin/env python3
"""RecoverCellsAndComputeKZGProofs for 64 blobs through mkzg_recover_cells_and_kzg_proofs_batch at the C ABI, arguments marshalled
once (what a cgo caller pays), plus the kernel breakdown of one profiled call."""
import ctypes as C
import json
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
    blobs = [rand_blob(s % 8) for s in range(NB)]
    d_blobs = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).cuda()
    d_cells = torch.zeros(NB * 262144, dtype=torch.uint8, device="cuda")
    d_proofs = torch.zeros(NB * 128 * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.zeros(NB, dtype=torch.int32, device="cuda")
    assert ctx.compute_cells_and_kzg_proofs_batch_device(d_blobs.data_ptr(), NB, d_cells.data_ptr(), d_proofs.data_ptr(), d_st.data_ptr(), 0) == 0
    torch.cuda.synchronize()
    cells, proofs = d_cells.cpu().numpy().tobytes(), d_proofs.cpu().numpy().tobytes()
    ids = list(range(0, 128, 2))
    keep = [[C.create_string_buffer(cells[(b * 128 + i) * 2048:(b * 128 + i + 1) * 2048], 2048) for i in ids] for b in range(NB)]
    id_arrs = [(C.c_uint64 * 64)(*ids) for _ in range(NB)]
    cell_arrs = [(C.c_char_p * 64)(*[C.cast(k, C.c_char_p) for k in keep[b]]) for b in range(NB)]
    pid = (C.POINTER(C.c_uint64) * NB)(*[C.cast(a, C.POINTER(C.c_uint64)) for a in id_arrs])
    pcl = (C.POINTER(C.c_char_p) * NB)(*[C.cast(a, C.POINTER(C.c_char_p)) for a in cell_arrs])
    nid = (C.c_size_t * NB)(*[64] * NB)
    oc = C.create_string_buffer(NB * 262144)
    op = C.create_string_buffer(NB * 128 * 48)
    st = (C.c_int32 * NB)()
    L = kzg.lib()
    for i in range(2):
        assert L.mkzg_recover_cells_and_kzg_proofs_batch(ctx._h, NB, pid, nid, pcl, nid, oc, op, st) == 0
    assert os.environ.get("MKZG_DBG_SKIP") or (oc.raw == cells and op.raw == proofs)
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        assert L.mkzg_recover_cells_and_kzg_proofs_batch(ctx._h, NB, pid, nid, pcl, nid, oc, op, st) == 0
    dt = (time.perf_counter() - t0) / reps
    ctx.profile_enable(True)
    ctx.profile_read(reset=True)
    L.mkzg_recover_cells_and_kzg_proofs_batch(ctx._h, NB, pid, nid, pcl, nid, oc, op, st)
    prof = ctx.profile_read(reset=True)
    print(json.dumps({"ms_per_64_blobs_c_abi": round(dt * 1e3, 3), "blobs_per_s": round(NB / dt, 1), "kernels_ms": {k: round(v[0], 3) for k, v in prof.items() if v[1]}}))
    ctx.close()


if __name__ == "__main__":
    main()
