"""world_size-2 gloo test of the sharded batch-verifier orchestration (go-eth-kzg_amd/distributed.py)
on CPU.  The per-rank arithmetic is supplied by the oracle (there is no GPU here); what is under test is the
N>1 path itself: shard bounds, the broadcast of r, the all-gather of records and the agreement of the verdict
with the single-process verifier, for a valid batch, a corrupted batch and an invalid input on one rank."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import oracle
        from conftest import load_product
        import importlib.util
        spec = importlib.util.spec_from_file_location("kzg_distributed", os.path.join(ROOT, "go-eth-kzg_amd", "distributed.py"))
        D = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(D)

        octx = oracle.OracleContext(with_fk20=False)
        n_total = 5
        blobs = [oracle.rand_blob(s) for s in range(n_total)]
        cms = [octx.blob_to_kzg_commitment(b)[1] for b in blobs]
        prs = [octx.compute_blob_kzg_proof(b, c)[1] for b, c in zip(blobs, cms)]

        class OracleEngine:
            def partial(self, shard, offset, r32):
                b, c, p = shard
                return octx.verify_blob_batch_partial(b, c, p, offset, r32)

            def finish(self, records):
                return octx.verify_blob_batch_finish(records)

        lo, hi = D.shard_bounds(n_total, world, rank)
        out = {}
        out["valid"] = D.verify_blob_kzg_proof_batch_sharded(OracleEngine(), (blobs[lo:hi], cms[lo:hi], prs[lo:hi]), n_total, dist)
        bad = list(prs)
        bad[3], bad[4] = bad[4], bad[3]
        out["corrupt"] = D.verify_blob_kzg_proof_batch_sharded(OracleEngine(), (blobs[lo:hi], cms[lo:hi], bad[lo:hi]), n_total, dist)
        nc = list(blobs)
        nc[4] = b"\xff" * 131072
        out["noncanonical"] = D.verify_blob_kzg_proof_batch_sharded(OracleEngine(), (nc[lo:hi], cms[lo:hi], prs[lo:hi]), n_total, dist)
        out["bounds"] = (lo, hi)
        out["single"] = octx.verify_blob_kzg_proof_batch(blobs, cms, prs)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, out))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, {"error": traceback.format_exc()}))


def test_sharded_verify_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for r in (0, 1):
        assert "error" not in res[r], res[r].get("error")
    assert res[0]["bounds"] == (0, 3) and res[1]["bounds"] == (3, 5)
    for r in (0, 1):
        assert res[r]["valid"] == 0 and res[r]["single"] == 0
        assert res[r]["corrupt"] == 1
        assert res[r]["noncanonical"] == 3


def test_shard_bounds_cover():
    import importlib.util
    spec = importlib.util.spec_from_file_location("kzg_distributed", os.path.join(ROOT, "go-eth-kzg_amd", "distributed.py"))
    D = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(D)
    for n in (0, 1, 5, 1024, 1027):
        for w in (1, 2, 4, 8):
            b = [D.shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1
