#!/usr/bin/env python3
"""Secondary benchmark: the EIP-7594 configs of BASELINE.json on one MI355X (configs[3], configs[4]).
   config 4 : ComputeCellsAndKZGProofs, batch of 64 GetRandBlob blobs (seeds 0..63) resident in HBM
   config 5a: VerifyCellKZGProofBatch on the 64 x 128 = 8192 cells those blobs produce (blob-major)
   config 5b: RecoverCellsAndComputeKZGProofs from the 64 even-indexed cells of one blob
Prints one JSON object; not the driver's headline (bench.py is)."""
import importlib.util
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
    NB = int(os.environ.get("NB", "64"))
    blobs = [rand_blob(s) for s in range(NB)]
    d_blobs = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).cuda()
    d_cells = torch.zeros(NB * 262144, dtype=torch.uint8, device="cuda")
    d_proofs = torch.zeros(NB * 128 * 48, dtype=torch.uint8, device="cuda")
    d_cm = torch.zeros(NB * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.zeros(NB, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    t0 = time.perf_counter()
    assert ctx.compute_cells_and_kzg_proofs_batch_device(d_blobs.data_ptr(), NB, d_cells.data_ptr(), d_proofs.data_ptr(), d_st.data_ptr(), st) == 0
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t0
    assert int(d_st.abs().sum().item()) == 0
    out = {"fk20_setup_plus_first_call_s": round(setup_s, 3)}
    # config 4
    ctx.profile_enable(True)
    ctx.profile_read(reset=True)
    steps = 3
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.compute_cells_and_kzg_proofs_batch_device(d_blobs.data_ptr(), NB, d_cells.data_ptr(), d_proofs.data_ptr(), d_st.data_ptr(), st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    prof = ctx.profile_read(reset=True)
    out["compute_cells_and_kzg_proofs"] = {"batch": NB, "ms_per_batch": round(dt * 1e3, 3), "blobs_per_s": round(NB / dt, 1),
                                           "kernels_ms": {k: round(v[0] / steps, 3) for k, v in prof.items() if v[1]},
                                           "algorithmic_GBps": round(399360 * NB / dt / 1e9, 3)}
    # config 5a
    assert ctx.blob_to_kzg_commitment_batch_device(d_blobs.data_ptr(), NB, d_cm.data_ptr(), d_st.data_ptr(), st) == 0
    torch.cuda.synchronize()
    cms = bytes(d_cm.cpu().tolist())
    cells = bytes(d_cells.cpu().numpy().tobytes())
    proofs = bytes(d_proofs.cpu().numpy().tobytes())
    c_list, i_list, cell_list, p_list = [], [], [], []
    for b in range(NB):
        for k in range(128):
            c_list.append(cms[48 * b:48 * b + 48])
            i_list.append(k)
            cell_list.append(cells[b * 262144 + k * 2048:b * 262144 + (k + 1) * 2048])
            p_list.append(proofs[(b * 128 + k) * 48:(b * 128 + k + 1) * 48])
    n = min(len(c_list), 8192)
    c_list, i_list, cell_list, p_list = c_list[:n], i_list[:n], cell_list[:n], p_list[:n]
    assert ctx.VerifyCellKZGProofBatch(c_list, i_list, cell_list, p_list) == 0
    bad = list(cell_list)
    bad[7], bad[8] = bad[8], bad[7]
    assert ctx.VerifyCellKZGProofBatch(c_list, i_list, bad, p_list) == 1
    ctx.profile_read(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        assert ctx.VerifyCellKZGProofBatch(c_list, i_list, cell_list, p_list) == 0
    dt = (time.perf_counter() - t0) / steps
    prof = ctx.profile_read(reset=True)
    # the same call at the C ABI, arguments marshalled once (what a cgo caller pays: no Python list handling in the loop)
    import ctypes as C
    L = kzg.lib()
    idx = (C.c_uint64 * n)(*i_list)
    arr = (C.c_char_p * n)(*cell_list)
    cj, pj = b"".join(c_list), b"".join(p_list)
    t0 = time.perf_counter()
    for _ in range(steps):
        assert L.mkzg_verify_cell_kzg_proof_batch(ctx._h, cj, n, idx, n, arr, n, pj, n) == 0
    dt_c = (time.perf_counter() - t0) / steps
    out["verify_cell_kzg_proof_batch"] = {"cells": n, "ms_per_call_incl_host_copy_of_16MiB": round(dt * 1e3, 3), "cells_per_s": round(n / dt, 1),
                                          "ms_per_call_c_abi": round(dt_c * 1e3, 3), "cells_per_s_c_abi": round(n / dt_c, 1),
                                          "kernels_ms": {k: round(v[0] / steps, 3) for k, v in prof.items() if v[1]}}
    # config 5b
    ids = list(range(0, 128, 2))
    half = [cell_list[i] for i in ids]
    rc, oc, op = ctx.RecoverCellsAndComputeKZGProofs(ids, half)
    assert rc == 0 and oc == cell_list[:128] and op == p_list[:128]
    ctx.profile_read(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.RecoverCellsAndComputeKZGProofs(ids, half)
    dt = (time.perf_counter() - t0) / steps
    prof = ctx.profile_read(reset=True)
    out["recover_cells_and_kzg_proofs"] = {"ms_per_blob_single_call": round(dt * 1e3, 3), "kernels_ms": {k: round(v[0] / steps, 3) for k, v in prof.items() if v[1]}}
    # config 5b, throughput form: all NB blobs (their 64 even-indexed cells each) in one batch call
    NBR = min(NB, len(cell_list) // 128)        # the cell list above is capped at 8192 cells
    jobs = [(ids, [cell_list[b * 128 + i] for i in ids]) for b in range(NBR)]
    rc, res = ctx.RecoverCellsAndComputeKZGProofsBatch(jobs)
    assert rc == 0 and all(st == 0 for st, _, _ in res)
    assert res[0][1] == cell_list[:128] and res[NBR - 1][2] == p_list[(NBR - 1) * 128:NBR * 128]
    ctx.profile_read(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.RecoverCellsAndComputeKZGProofsBatch(jobs)
    dtb = (time.perf_counter() - t0) / steps
    prof = ctx.profile_read(reset=True)
    out["recover_cells_and_kzg_proofs_batch"] = {"blobs": NBR, "ms_per_batch_incl_host_marshalling": round(dtb * 1e3, 3), "blobs_per_s": round(NBR / dtb, 1),
                                                 "kernels_ms": {k: round(v[0] / steps, 3) for k, v in prof.items() if v[1]}}
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
