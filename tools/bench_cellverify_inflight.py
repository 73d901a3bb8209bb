#!/usr/bin/env python3
"""VerifyCellKZGProofBatch over 64 x 128 = 8192 cells (host pointers, arguments marshalled once as a cgo caller holds them): sustained
cells/s against the number of caller threads on one shared context, with the host CPUs the process keeps busy.
CV_DEVICE=1: the cells stay in HBM (mkzg_verify_cell_kzg_proof_batch_device) -- the same calls without gather and upload."""
import ctypes
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import rand_blob, load_product  # noqa: E402


def main():
    import torch
    torch.cuda.init()
    kzg = load_product()
    ctx = kzg.Context(device=0)
    if os.environ.get("CV_HOST_WAIT"):
        assert ctx.set_option("host_wait", int(os.environ["CV_HOST_WAIT"])) == 0
    NB = int(os.environ.get("CV_BLOBS", "64"))
    blobs = [rand_blob(s) for s in range(NB)]
    d_blobs = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).cuda()
    d_cells = torch.zeros(NB * 262144, dtype=torch.uint8, device="cuda")
    d_proofs = torch.zeros(NB * 128 * 48, dtype=torch.uint8, device="cuda")
    d_cm = torch.zeros(NB * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.zeros(NB, dtype=torch.int32, device="cuda")
    assert ctx.compute_cells_and_kzg_proofs_batch_device(d_blobs.data_ptr(), NB, d_cells.data_ptr(), d_proofs.data_ptr(), d_st.data_ptr(), 0) == 0
    assert ctx.blob_to_kzg_commitment_batch_device(d_blobs.data_ptr(), NB, d_cm.data_ptr(), d_st.data_ptr(), 0) == 0
    torch.cuda.synchronize()
    cms, cells, proofs = d_cm.cpu().numpy().tobytes(), d_cells.cpu().numpy().tobytes(), d_proofs.cpu().numpy().tobytes()
    n = NB * 128
    # every caller gets its OWN copies of the cells (CV_SHARED_CELLS=1: all callers read the same 16 MiB, which then live in the CPU's
    # caches -- what this tool did until the end of round 5, and 10-15 % more than bench.py's leg with distinct cells reports)
    shared = os.environ.get("CV_SHARED_CELLS") == "1"
    kmax = 1 if shared else max([int(x) for x in os.environ["CV_CALLERS"].split(",")] if os.environ.get("CV_CALLERS") else [16])
    keeps, arrs = [], []
    for _ in range(kmax):
        keep = [ctypes.create_string_buffer(cells[i * 2048:(i + 1) * 2048], 2048) for i in range(n)]
        keeps.append(keep)
        arrs.append((ctypes.c_char_p * n)(*[ctypes.cast(k, ctypes.c_char_p) for k in keep]))
    idx = (ctypes.c_uint64 * n)(*[i % 128 for i in range(n)])
    cj = b"".join(cms[48 * (i // 128):48 * (i // 128) + 48] for i in range(n))
    bad = bytearray(proofs)
    bad[0:48] = proofs[48:96]
    bad = bytes(bad)
    L = kzg.lib()
    dev = os.environ.get("CV_DEVICE") == "1"
    d_all = torch.frombuffer(bytearray(cells), dtype=torch.uint8).cuda() if dev else None
    ks = [int(x) for x in os.environ["CV_CALLERS"].split(",")] if os.environ.get("CV_CALLERS") else (1, 2, 3, 4, 6, 8, 12, 16)
    for K in ks:
        reps = int(os.environ.get('CV_REPS', '12'))
        errs = []

        def caller(k):
            try:
                if dev:
                    st = torch.cuda.Stream()
                    for _ in range(reps):
                        assert L.mkzg_verify_cell_kzg_proof_batch_device(ctx._h, cj, ctypes.cast(idx, ctypes.c_void_p), ctypes.c_void_p(d_all.data_ptr()), proofs, n, ctypes.c_void_p(st.cuda_stream)) == 0
                    assert L.mkzg_verify_cell_kzg_proof_batch_device(ctx._h, cj, ctypes.cast(idx, ctypes.c_void_p), ctypes.c_void_p(d_all.data_ptr()), bad, n, ctypes.c_void_p(st.cuda_stream)) == 1
                    return
                arr = arrs[k % len(arrs)]
                for _ in range(reps):
                    assert L.mkzg_verify_cell_kzg_proof_batch(ctx._h, cj, n, idx, n, arr, n, proofs, n) == 0
                assert L.mkzg_verify_cell_kzg_proof_batch(ctx._h, cj, n, idx, n, arr, n, bad, n) == 1
            except Exception as e:      # noqa: BLE001
                errs.append(repr(e))

        rates, cpus = [], []
        for trial in range(4):                  # one priming trial, then the median of three
            ts = [threading.Thread(target=caller, args=(k,)) for k in range(K)]
            t0, c0 = time.perf_counter(), time.process_time()
            for t in ts:
                t.start()
            for t in ts:
                t.join()
            dt, cpu = time.perf_counter() - t0, time.process_time() - c0
            if trial:
                rates.append(K * (reps + 1) * n / dt / 1e6)
                cpus.append(cpu / dt)
        rates.sort()
        cpus.sort()
        assert not errs, errs[0]
        print(f"{K} callers{' (cells in HBM)' if dev else ''}: {rates[1]:.2f} M cells/s (median of {[round(r, 2) for r in rates]}), {1e3 * n / rates[1] / 1e6:.3f} ms per call amortised, {cpus[1]:.1f} host CPUs busy", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
