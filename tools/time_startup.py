#!/usr/bin/env python3
"""tools/time_startup.py -- cold-start costs: context construction (trusted-setup JSON -> device tables) and the first call of
each API family (lazy workspaces / tables)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import rand_blob, load_product  # noqa: E402
import torch
torch.cuda.init()
kzg = load_product()
blob = rand_blob(1)


def t(label, fn):
    t0 = time.perf_counter()
    r = fn()
    print(f"{label:48s} {1e3 * (time.perf_counter() - t0):9.1f} ms")
    return r


ctx = t("Context() (JSON, 8192 G1 + 65 G2 decodes, roots)", lambda: kzg.Context())
rc, cm = t("first BlobToKZGCommitment (bucket plan + table)", lambda: ctx.BlobToKZGCommitment(blob))
rc, pr = t("first ComputeBlobKZGProof", lambda: ctx.ComputeBlobKZGProof(blob, cm))
t("first VerifyBlobKZGProof (verify ws, G1 fixed base)", lambda: ctx.VerifyBlobKZGProof(blob, cm, pr))
rc, cells, proofs = t("first ComputeCellsAndKZGProofs (FK20 47 GB table)", lambda: ctx.ComputeCellsAndKZGProofs(blob))
t("first VerifyCellKZGProofBatch", lambda: ctx.VerifyCellKZGProofBatch([cm] * 128, list(range(128)), cells, proofs))
ids = list(range(0, 128, 2))
t("first RecoverCellsAndComputeKZGProofs", lambda: ctx.RecoverCellsAndComputeKZGProofs(ids, [cells[i] for i in ids]))
ctx.set_option("commit_direct_bits", 15)
t("first BlobToKZGCommitment with the 146 GB table", lambda: ctx.BlobToKZGCommitment(blob))
ctx.close()
