#!/usr/bin/env python3
"""compute-path and recovery-path FK20 launches alternated in one process (kernel trace: why does k_fk20_msm take 3.5 ms in one and 2.0 in the other?)"""
import ctypes as C
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
    NB = 64
    blobs = [rand_blob(s) for s in range(NB)]
    d_blobs = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).cuda()
    d_cells = torch.zeros(NB * 262144, dtype=torch.uint8, device="cuda")
    d_proofs = torch.zeros(NB * 128 * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.zeros(NB, dtype=torch.int32, device="cuda")

    def compute():
        assert ctx.compute_cells_and_kzg_proofs_batch_device(d_blobs.data_ptr(), NB, d_cells.data_ptr(), d_proofs.data_ptr(), d_st.data_ptr(), 0) == 0
        torch.cuda.synchronize()

    compute()
    cells, proofs = d_cells.cpu().numpy().tobytes(), d_proofs.cpu().numpy().tobytes()
    which = os.environ.get("IDS", "even")
    ids = list(range(0, 128, 2)) if which == "even" else (list(range(64)) if which == "first" else list(range(128)))
    m = len(ids)
    keep = [[C.create_string_buffer(cells[(b * 128 + i) * 2048:(b * 128 + i + 1) * 2048], 2048) for i in ids] for b in range(NB)]
    id_arrs = [(C.c_uint64 * m)(*ids) for _ in range(NB)]
    cell_arrs = [(C.c_char_p * m)(*[C.cast(k, C.c_char_p) for k in keep[b]]) for b in range(NB)]
    pid = (C.POINTER(C.c_uint64) * NB)(*[C.cast(a, C.POINTER(C.c_uint64)) for a in id_arrs])
    pcl = (C.POINTER(C.c_char_p) * NB)(*[C.cast(a, C.POINTER(C.c_char_p)) for a in cell_arrs])
    nid = (C.c_size_t * NB)(*[m] * NB)
    oc = C.create_string_buffer(NB * 262144)
    op = C.create_string_buffer(NB * 128 * 48)
    st = (C.c_int32 * NB)()
    L = kzg.lib()

    def recover():
        assert L.mkzg_recover_cells_and_kzg_proofs_batch(ctx._h, NB, pid, nid, pcl, nid, oc, op, st) == 0
        assert oc.raw == cells and op.raw == proofs

    for _ in range(2):
        compute()
    for _ in range(3):
        recover()
    for _ in range(2):
        compute()
    ctx.close()


if __name__ == "__main__":
    main()
