"""Multi-GPU orchestration of the batch verifiers (one process per GPU, torch.distributed).

VerifyCellKZGProofBatch shards the same way (verify_cell_kzg_proof_batch_sharded): cells are cut into contiguous ranges,
a rank folds its cells with r^(offset+k) into four partial G1 sums (proofs, weighted commitments, [I(tau)], weighted
proofs -- reference internal/kzg_multi/kzg_verify.go:32,41,68,80), and the 512-byte records are all-gathered.

Commitments / proofs / cells shard by blob with no collective at all.  VerifyBlobKZGProofBatch has ONE
exchange step: rank 0 draws the RLC scalar r and broadcasts 32 bytes; every rank folds its
contiguous shard with the global exponents r^(offset+i) into a 512-byte record (three partial G1 sums, one
Fr sum, a status word); the records are all-gathered (RCCL over xGMI on GPUs; < 4 KB in total, purely
latency-bound) and every rank -- or just rank 0 -- adds them and runs the single pairing check.  RCCL has
no group-element reduction operator, hence gather-then-add instead of all-reduce.

The `engine` argument is anything with
    partial(shard, offset, r32) -> (rc, record_bytes_512)      and      finish(list_of_records) -> rc
The product engine (`ProductEngine`) calls the C ABI on device-resident shards; tests on CPU plug an
oracle-backed engine to exercise this orchestration with the gloo backend.
"""
import os


def shard_bounds(n_total, world, rank):
    """contiguous ranges [lo, hi) -- blob i goes to rank i * world // n (reference has no sharding)"""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def verify_blob_kzg_proof_batch_sharded(engine, shard, n_total, dist=None, device=None):
    """Collective call: every rank passes its own shard (any object the engine understands) of a global
    batch of n_total proofs.  Returns the same rc on every rank (0 ok, 1 verification failed, >1 input error)."""
    import torch
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    lo, hi = shard_bounds(n_total, world, rank)
    dev = device if device is not None else torch.device("cpu")
    # 1. the batch-wide scalar: chosen by rank 0, 32 bytes broadcast
    r = torch.zeros(32, dtype=torch.uint8, device=dev)
    if rank == 0:
        r = torch.frombuffer(bytearray(os.urandom(32)), dtype=torch.uint8).to(dev)
    if world > 1:
        dist.broadcast(r, src=0)
    r32 = bytes(r.cpu().tolist())
    # 2. local fold
    rc, rec = engine.partial(shard, lo, r32)
    # 3. the one exchange step
    mine = torch.frombuffer(bytearray(rec), dtype=torch.uint8).to(dev)
    if world > 1:
        allr = [torch.zeros(512, dtype=torch.uint8, device=dev) for _ in range(world)]
        dist.all_gather(allr, mine)
        records = [bytes(t.cpu().tolist()) for t in allr]
    else:
        records = [rec]
    # 4. combine + pairing (identical on every rank; cheap, and avoids a second broadcast)
    return engine.finish(records)


def verify_cell_kzg_proof_batch_sharded(engine, shard, n_total, dist=None, device=None):
    """Collective call: every rank passes its own contiguous shard (commitments, cell_indices, cells, proofs) of a global
    batch of n_total cells; `engine` needs cell_partial(shard, offset, r32) and cell_finish(records).  Same rc on every rank."""
    import torch
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    lo, hi = shard_bounds(n_total, world, rank)
    dev = device if device is not None else torch.device("cpu")
    r = torch.zeros(32, dtype=torch.uint8, device=dev)
    if rank == 0:
        r = torch.frombuffer(bytearray(os.urandom(32)), dtype=torch.uint8).to(dev)
    if world > 1:
        dist.broadcast(r, src=0)
    rc, rec = engine.cell_partial(shard, lo, bytes(r.cpu().tolist()))
    mine = torch.frombuffer(bytearray(rec), dtype=torch.uint8).to(dev)
    if world > 1:
        allr = [torch.zeros(512, dtype=torch.uint8, device=dev) for _ in range(world)]
        dist.all_gather(allr, mine)
        records = [bytes(t.cpu().tolist()) for t in allr]
    else:
        records = [rec]
    return engine.cell_finish(records)


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
