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
        # a caller that cuts the batch differently (1 + 4 instead of 3 + 2): the offsets come from the all-gathered shard sizes, not
        # from an assumed split, so every item still meets its own power of r
        ulo, uhi = (0, 1) if rank == 0 else (1, n_total)
        out["uneven"] = D.verify_blob_kzg_proof_batch_sharded(OracleEngine(), (blobs[ulo:uhi], cms[ulo:uhi], prs[ulo:uhi]), n_total, dist)
        ubad = list(prs)
        ubad[0], ubad[1] = ubad[1], ubad[0]             # swapped across the rank boundary of the uneven cut
        out["uneven_corrupt"] = D.verify_blob_kzg_proof_batch_sharded(OracleEngine(), (blobs[ulo:uhi], cms[ulo:uhi], ubad[ulo:uhi]), n_total, dist)
        # shards that do not add up to the announced batch are refused before anything is folded
        try:
            D.verify_blob_kzg_proof_batch_sharded(OracleEngine(), (blobs[lo:hi], cms[lo:hi], prs[lo:hi]), n_total + 1, dist)
            out["mismatch"] = "accepted"
        except ValueError:
            out["mismatch"] = "refused"
        bad = list(prs)
        bad[3], bad[4] = bad[4], bad[3]
        out["corrupt"] = D.verify_blob_kzg_proof_batch_sharded(OracleEngine(), (blobs[lo:hi], cms[lo:hi], bad[lo:hi]), n_total, dist)
        nc = list(blobs)
        nc[4] = b"\xff" * 131072
        out["noncanonical"] = D.verify_blob_kzg_proof_batch_sharded(OracleEngine(), (nc[lo:hi], cms[lo:hi], prs[lo:hi]), n_total, dist)
        out["bounds"] = (lo, hi)
        ex = D.measure_exchange(dist, None, reps=3)          # the exchange step alone (bench.py reports it as multi_gpu.exchange_us)
        out["exchange"] = (ex["exchange_us"] > 0 and abs(ex["exchange_us"] - ex["broadcast_us"] - ex["all_gather_sizes_us"] - ex["all_gather_records_us"]) < 1e-6)
        out["single"] = octx.verify_blob_kzg_proof_batch(blobs, cms, prs)

        # --- sharded VerifyCellKZGProofBatch: 2 blobs x 6 cells, cut 6 / 6 across the two ranks (rows repeat inside a shard)
        fctx = oracle.OracleContext(with_fk20=True)
        tuples = []
        for bi in (0, 1):
            rc, cells, proofs = fctx.compute_cells_and_kzg_proofs(blobs[bi])
            assert rc == 0
            for ci in (0, 5, 64, 127, 33, 2):
                tuples.append((cms[bi], ci, cells[ci * 2048:(ci + 1) * 2048], proofs[ci * 48:(ci + 1) * 48]))

        class OracleCellEngine:
            def cell_partial(self, shard, offset, r32):
                return fctx.verify_cell_batch_partial(*shard, offset, r32)

            def cell_finish(self, records):
                return fctx.verify_cell_batch_finish(records)

        def run(tups):
            nt = len(tups)
            l2, h2 = D.shard_bounds(nt, world, rank)
            sh = tups[l2:h2]
            return D.verify_cell_kzg_proof_batch_sharded(OracleCellEngine(), ([t[0] for t in sh], [t[1] for t in sh], [t[2] for t in sh], [t[3] for t in sh]), nt, dist)

        out["cells_valid"] = run(tuples)
        badp = list(tuples)
        badp[7] = (badp[7][0], badp[7][1], badp[7][2], badp[8][3])                 # proof of another cell
        out["cells_corrupt"] = run(badp)
        badc = list(tuples)
        badc[10] = (badc[10][0], badc[10][1], b"\xff" * 2048, badc[10][3])          # non-canonical cell on rank 1
        out["cells_noncanonical"] = run(badc)
        out["cells_single"] = fctx.verify_cell_kzg_proof_batch([t[0] for t in tuples], [t[1] for t in tuples], [t[2] for t in tuples], [t[3] for t in tuples])
        fctx.close()
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
        assert res[r]["uneven"] == 0 and res[r]["uneven_corrupt"] == 1 and res[r]["mismatch"] == "refused"
        assert res[r]["exchange"] is True
        assert res[r]["cells_valid"] == 0 and res[r]["cells_single"] == 0
        assert res[r]["cells_corrupt"] == 1
        assert res[r]["cells_noncanonical"] == 3


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
