"""Multi-GPU orchestration of the batch verifiers (one process per GPU, torch.distributed).

Commitments / proofs / cells / recovery shard by blob with no collective at all.  The two batch verifiers have ONE exchange step
each:

  VerifyBlobKZGProofBatch   rank 0 draws the RLC scalar r and broadcasts 32 bytes; every rank folds its contiguous shard with the
                            global exponents r^(offset+i) into a 512-byte record (three partial G1 sums, one Fr sum, a status word);
  VerifyCellKZGProofBatch   the same with four partial G1 sums (proofs, weighted commitments, [I(tau)], weighted proofs -- reference
                            internal/kzg_multi/kzg_verify.go:32,41,68,80);

the records are all-gathered (RCCL over xGMI on GPUs; < 4 KB in total, purely latency-bound) and every rank adds them and runs the
single pairing check.  RCCL has no group-element reduction operator, hence gather-then-add instead of all-reduce.

SOUNDNESS of the random linear combination needs every item of the global batch folded with its OWN power of r: the ranks' index
ranges must tile [0, n_total) exactly.  The offsets are therefore never taken on trust from a caller's idea of the split: every rank
contributes the size of the shard it actually holds, the sizes are all-gathered, and offset(rank) is their prefix sum.

The `engine` argument is anything with
    partial(shard, offset, r32) -> (rc, record_bytes_512)      and      finish(list_of_records) -> rc          (blobs)
    cell_partial(shard, offset, r32) -> (rc, record)           and      cell_finish(list_of_records) -> rc     (cells)
The product engine (`ProductEngine`) calls the C ABI; tests on CPU plug an oracle-backed engine to exercise this orchestration with
the gloo backend.
"""
import os


def shard_bounds(n_total, world, rank):
    """contiguous ranges [lo, hi) -- a convenience for callers that cut a global batch (reference has no sharding)"""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _exchange_setup(n_local, n_total, dist, device):
    """-> (world, rank, offset of this rank's shard, r32).  One all_gather of the shard sizes, one broadcast of r."""
    import torch
    live = dist is not None and dist.is_initialized()          # a process group of ONE rank still runs its collectives (RCCL smoke test)
    world = dist.get_world_size() if live else 1
    rank = dist.get_rank() if live else 0
    dev = device if device is not None else torch.device("cpu")
    if live:
        mine = torch.tensor([int(n_local)], dtype=torch.int64, device=dev)
        sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(sizes, mine)
        sizes = [int(t.item()) for t in sizes]
    else:
        sizes = [int(n_local)]
    if n_total is not None and sum(sizes) != int(n_total):
        raise ValueError(f"shards hold {sum(sizes)} items in total, the batch was announced as {n_total}: the index ranges would not tile it")
    offset = sum(sizes[:rank])
    # the batch-wide scalar: chosen by rank 0, 32 bytes broadcast
    r = torch.zeros(32, dtype=torch.uint8, device=dev)
    if rank == 0:
        r = torch.frombuffer(bytearray(os.urandom(32)), dtype=torch.uint8).to(dev)
    if live:
        dist.broadcast(r, src=0)
    return world, rank, offset, bytes(r.cpu().tolist()), dev


def _gather_records(rec, world, dist, dev):
    import torch
    if dist is None or not dist.is_initialized():
        return [rec]
    mine = torch.frombuffer(bytearray(rec), dtype=torch.uint8).to(dev)
    allr = [torch.zeros(512, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(allr, mine)
    return [bytes(t.cpu().tolist()) for t in allr]


def measure_exchange(dist, device=None, reps=20):
    """What the batch verifiers' exchange step costs on its own, in microseconds per verification: the three collectives of one
    sharded call -- all_gather of the shard sizes, broadcast of the 32-byte scalar, all_gather of the 512-byte records -- timed
    between barriers, nothing else in flight.  Collective call (every rank); returns the means over `reps` repetitions."""
    import time
    import torch
    dev = device if device is not None else torch.device("cpu")
    world = dist.get_world_size()
    sync = torch.cuda.synchronize if dev.type == "cuda" else (lambda: None)
    mine = torch.tensor([1], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    r = torch.zeros(32, dtype=torch.uint8, device=dev)
    rec = torch.zeros(512, dtype=torch.uint8, device=dev)
    recs = [torch.zeros(512, dtype=torch.uint8, device=dev) for _ in range(world)]
    acc = [0.0, 0.0, 0.0]
    for it in range(reps + 3):                     # three untimed rounds: communicators, buffers
        dist.barrier()
        sync()
        t0 = time.perf_counter()
        dist.all_gather(sizes, mine)
        _ = [int(t.item()) for t in sizes]         # the call reads them on the host (prefix sum of the offsets)
        t1 = time.perf_counter()
        dist.broadcast(r, src=0)
        _ = bytes(r.cpu().tolist())
        t2 = time.perf_counter()
        dist.all_gather(recs, rec)
        _ = [bytes(t.cpu().tolist()) for t in recs]
        t3 = time.perf_counter()
        if it >= 3:
            acc[0] += t1 - t0
            acc[1] += t2 - t1
            acc[2] += t3 - t2
    us = [a / reps * 1e6 for a in acc]
    return {"all_gather_sizes_us": us[0], "broadcast_us": us[1], "all_gather_records_us": us[2], "exchange_us": sum(us)}


def _shard_len(shard, n_local):
    if n_local is not None:
        return int(n_local)
    first = shard[0]
    if hasattr(first, "__len__"):
        return len(first)
    raise ValueError("n_local is required when the shard is not a tuple of sequences")


def verify_blob_kzg_proof_batch_sharded(engine, shard, n_total=None, dist=None, device=None, n_local=None):
    """Collective call: every rank passes its own shard (any object the engine understands) of a global batch.  n_local: items in this
    rank's shard (default: len(shard[0]); the product engine's shard carries it explicitly).  Returns the same rc on every rank
    (0 ok, 1 verification failed, >1 input error)."""
    if n_local is None and len(shard) == 5 and isinstance(shard[3], int):
        n_local = shard[3]                       # ProductEngine: (d_blobs, d_commitments, d_proofs, n_local, stream)
    world, rank, offset, r32, dev = _exchange_setup(_shard_len(shard, n_local), n_total, dist, device)
    rc, rec = engine.partial(shard, offset, r32)       # rc also travels in the record's status word
    # combine + pairing (identical on every rank; cheap, and avoids a second broadcast)
    return engine.finish(_gather_records(rec, world, dist, dev))


def verify_cell_kzg_proof_batch_sharded(engine, shard, n_total=None, dist=None, device=None, n_local=None):
    """Collective call: every rank passes its own contiguous shard (commitments, cell_indices, cells, proofs) of a global batch of cells;
    `engine` needs cell_partial(shard, offset, r32) and cell_finish(records).  Same rc on every rank."""
    world, rank, offset, r32, dev = _exchange_setup(_shard_len(shard, n_local), n_total, dist, device)
    rc, rec = engine.cell_partial(shard, offset, r32)
    return engine.cell_finish(_gather_records(rec, world, dist, dev))


class ProductEngine:
    """shard = (d_blobs_ptr, d_commitments_ptr, d_proofs_ptr, n_local, stream)"""

    def __init__(self, ctx):
        self.ctx = ctx

    def partial(self, shard, offset, r32):
        d_blobs, d_cm, d_pr, n_local, stream = shard
        return self.ctx.verify_blob_batch_partial_device(d_blobs, d_cm, d_pr, n_local, offset, r32, stream)

    def finish(self, records):
        return self.ctx.verify_blob_batch_finish(records)

    def cell_partial(self, shard, offset, r32):
        """shard = (commitments, cell_indices, cells, proofs): host byte strings, as the reference API takes them"""
        cms, idx, cells, prs = shard
        return self.ctx.verify_cell_batch_partial(cms, idx, cells, prs, offset, r32)

    def cell_finish(self, records):
        return self.ctx.verify_cell_batch_finish(records)
