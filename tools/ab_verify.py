#!/usr/bin/env python3
"""tools/ab_verify.py -- A/B of library builds on ONE box for the device-resident blob verifier: lone VerifyBlobKZGProofBatch(1024) (median
of 12) and the sustained rate from 32 callers over 8 distinct input sets (median of 3 trials).  Each build runs in its own process
(MI355KZG_LIB), the builds alternate, ROUNDS times.
    python3 tools/ab_verify.py build_ab/libmi355kzg_old.so go-eth-kzg_amd/libmi355kzg.so [ROUNDS=2]
    python3 tools/ab_verify.py --one        (child: measures the library named by MI355KZG_LIB)"""
import json
import os
import statistics
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one():
    import torch
    import bench
    torch.cuda.init()
    kzg = bench.load_product()
    ctx = kzg.Context(device=0)
    N, K, reps = 1024, int(os.environ.get("AB_CALLERS", "32")), int(os.environ.get("AB_REPS", "30"))      # (12 calls per caller until round 5: inside the start-up convoy)
    st = torch.cuda.current_stream().cuda_stream
    sets = []
    for q in range(8):
        sb = torch.from_numpy(bench.fast_blobs(N, 5000 + q).reshape(-1)).cuda()
        sc = torch.zeros(N * 48, dtype=torch.uint8, device="cuda")
        sp = torch.zeros(N * 48, dtype=torch.uint8, device="cuda")
        ss = torch.zeros(N, dtype=torch.int32, device="cuda")
        assert ctx.blob_to_kzg_commitment_batch_device(sb.data_ptr(), N, sc.data_ptr(), ss.data_ptr(), st) == 0
        assert ctx.compute_blob_kzg_proof_batch_device(sb.data_ptr(), sc.data_ptr(), N, sp.data_ptr(), ss.data_ptr(), st) == 0
        torch.cuda.synchronize()
        bad = sp.clone()
        bad[0:48] = sp[48:96]
        sets.append((sb, sc, sp, bad))
    b0, c0, p0, bad0 = sets[0]
    assert ctx.verify_blob_kzg_proof_batch_device(b0.data_ptr(), c0.data_ptr(), p0.data_ptr(), N, st) == 0
    assert ctx.verify_blob_kzg_proof_batch_device(b0.data_ptr(), c0.data_ptr(), bad0.data_ptr(), N, st) == 1
    lone = []
    for _ in range(12):
        t0 = time.perf_counter()
        assert ctx.verify_blob_kzg_proof_batch_device(b0.data_ptr(), c0.data_ptr(), p0.data_ptr(), N, st) == 0
        lone.append((time.perf_counter() - t0) * 1e3)
    errs = []

    def caller(k):
        try:
            s = torch.cuda.Stream(priority=-1 if k % 2 == 0 else 0)
            b_, c_, p_, bad_ = sets[k % 8]
            for _ in range(reps):
                assert ctx.verify_blob_kzg_proof_batch_device(b_.data_ptr(), c_.data_ptr(), p_.data_ptr(), N, s.cuda_stream) == 0
            assert ctx.verify_blob_kzg_proof_batch_device(b_.data_ptr(), c_.data_ptr(), bad_.data_ptr(), N, s.cuda_stream) == 1
        except Exception as e:      # noqa: BLE001
            errs.append(repr(e))

    rates = []
    for trial in range(4):
        ts = [threading.Thread(target=caller, args=(k,)) for k in range(K)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if trial:
            rates.append(K * (reps + 1) * N / (time.perf_counter() - t0))
    assert not errs, errs[0]
    gs = {k: ctx.get_option(k) for k in ("graph_launches", "graph_instantiations", "graph_fallbacks")}
    print("AB " + json.dumps({"graphs": gs, "lib": os.path.basename(kzg.LIB_PATH), "lone_ms_median": round(statistics.median(lone), 3), "sustained_blobs_per_s_median": round(statistics.median(rates)),
                              "sustained_each": [round(r) for r in rates]}), flush=True)
    ctx.close()


def main():
    if sys.argv[1] == "--one":
        return one()
    libs = [a for a in sys.argv[1:] if not a.isdigit()]
    rounds = next((int(a) for a in sys.argv[1:] if a.isdigit()), 2)
    for r in range(rounds):
        for lib in libs:
            env = dict(os.environ, MI355KZG_LIB=os.path.abspath(lib))
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env, capture_output=True, text=True, timeout=600)
            line = next((ln for ln in out.stdout.splitlines() if ln.startswith("AB ")), None)
            print(line or ("FAILED " + lib + out.stderr[-600:]), flush=True)


if __name__ == "__main__":
    main()
