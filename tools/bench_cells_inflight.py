#!/usr/bin/env python3
"""Device-resident ComputeCellsAndKZGProofs, 64 blobs per call: sustained blobs/s against the number of caller threads (one call in
flight each, one shared context)."""
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
    if os.environ.get("FK20_BITS"):
        assert ctx.set_option("fk20_bits", int(os.environ["FK20_BITS"])) == 0
    NB = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    blobs = [rand_blob(s) for s in range(min(NB, 16))]
    d_blobs = torch.frombuffer(bytearray(b"".join(blobs[i % len(blobs)] for i in range(NB))), dtype=torch.uint8).cuda()
    for K in ([int(x) for x in os.environ["CI_CALLERS"].split(",")] if os.environ.get("CI_CALLERS") else (1, 2, 3, 4, 6, 8)):
        bufs = [(torch.zeros(NB * 262144, dtype=torch.uint8, device="cuda"), torch.zeros(NB * 128 * 48, dtype=torch.uint8, device="cuda"),
                 torch.ones(NB, dtype=torch.int32, device="cuda")) for _ in range(K)]
        torch.cuda.synchronize()
        reps = 6

        def caller(k):
            s = torch.cuda.Stream(priority=-1 if k % 2 == 0 else 0)
            c, p, st = bufs[k]
            for _ in range(reps):
                assert ctx.compute_cells_and_kzg_proofs_batch_device(d_blobs.data_ptr(), NB, c.data_ptr(), p.data_ptr(), st.data_ptr(), s.cuda_stream) == 0
            s.synchronize()

        for trial in range(2):
            ts = [threading.Thread(target=caller, args=(k,)) for k in range(K)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in ts:
                t.start()
            for t in ts:
                t.join()
            dt = time.perf_counter() - t0
        assert all(int(b[2].abs().sum().item()) == 0 for b in bufs)
        assert all(bool((b[1] == bufs[0][1]).all().item()) for b in bufs)
        print(f"{K} callers: {K * reps * NB / dt:.0f} blobs/s, {1e3 * dt / (K * reps):.3f} ms per call amortised", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
