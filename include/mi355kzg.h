/*
 * mi355kzg.h -- C ABI of the MI355X-native KZG engine (libmi355kzg.so).
 *
 * Drop-in boundary: the byte-level method set of go-eth-kzg's `Context`
 * (reference api.go:17-28).  Every entry point below names the reference method it replaces
 * (file:line relative to crate-crypto/go-eth-kzg); a cgo `Context` wrapper binds them 1:1
 * (INTEGRATION.md).  Plain pointers and sizes only; no C++/torch types.
 *
 * Conventions
 *   - blobs are 131072 bytes, cells 2048, commitments/proofs 48, scalars 32 (big-endian),
 *     exactly reference serialization.go:56-95.
 *   - "host" entry points take host pointers and copy; "_device" entry points take device
 *     (HBM) pointers + a hipStream_t (as void*) and enqueue asynchronously on that stream.
 *   - return value: MKZG_OK, MKZG_VERIFY_FAILED (<-> kzg.ErrVerifyOpeningProof), or one code per
 *     reference error sentinel; MKZG_ERR_HIP for a runtime failure.  Never aborts.
 *   - a context is immutable after construction and may be shared by threads, like the reference's Context
 *     (api.go:17-28, examples_test.go:11): every call works in one of the context's lanes (own streams, workspaces
 *     and pinned staging), so up to 32 calls per device run concurrently on the GPU ("lanes" option); further callers wait
 *     for a lane.  A lane keeps its workspaces (0.1-0.6 GB of HBM and up to 0.3 GB of pinned host memory, by the calls it has run)
 *     until the context is destroyed, the "lanes" limit is lowered below it, or "trim" is set.
 *   - the calling thread's current HIP device is left as it was found: every entry point switches to the context's device for the
 *     duration of the call and back.
 *   - batch entry points take any batch size (pieces are cut internally; the batch verifiers fold all pieces with ONE
 *     random scalar and finish with one pairing check).
 *   - there is NO CPU fallback: without a HIP device mkzg_ctx_new fails with MKZG_ERR_HIP.
 */
#ifndef MI355KZG_H
#define MI355KZG_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MKZG_BYTES_PER_BLOB 131072        /* reference serialization.go:33,80 */
#define MKZG_BYTES_PER_CELL 2048          /* reference serialization.go:36 */
#define MKZG_CELLS_PER_EXT_BLOB 128       /* reference serialization.go:39 */
#define MKZG_BYTES_PER_G1 48              /* reference serialization.go:14 */
#define MKZG_BYTES_PER_SCALAR 32          /* reference serialization.go:25 */

enum {
    MKZG_OK = 0,
    MKZG_VERIFY_FAILED = 1,                /* kzg.ErrVerifyOpeningProof       internal/kzg/errors.go:8 */
    MKZG_ERR_BATCH_LENGTH = 2,             /* ErrBatchLengthCheck             errors.go:6 */
    MKZG_ERR_NON_CANONICAL_SCALAR = 3,     /* ErrNonCanonicalScalar           errors.go:7 */
    MKZG_ERR_INVALID_CELL_ID = 4,          /* ErrInvalidCellID                errors.go:8 */
    MKZG_ERR_INVALID_ROW_INDEX = 5,        /* ErrInvalidRowIndex              errors.go:9 */
    MKZG_ERR_NIL_INPUT = 6,                /* ErrDeserializeNilInput          errors.go:10 */
    MKZG_ERR_NUM_CELL_IDS = 7,             /* ErrNumCellIDsNotEqualNumCells   errors.go:12 */
    MKZG_ERR_CELL_IDS_NOT_ORDERED = 8,     /* ErrCellIDsNotOrdered            errors.go:13 */
    MKZG_ERR_FOUND_INVALID_CELL_ID = 9,    /* ErrFoundInvalidCellID           errors.go:14 */
    MKZG_ERR_NOT_ENOUGH_CELLS = 10,        /* ErrNotEnoughCellsForReconstruction errors.go:15 */
    MKZG_ERR_G1_BAD_ENCODING = 11,         /* gnark G1Affine.SetBytes errors reached from serialization.go:110 */
    MKZG_ERR_G1_NOT_ON_CURVE = 12,
    MKZG_ERR_G1_NOT_IN_SUBGROUP = 13,
    MKZG_ERR_TOO_MANY_GOROUTINES = 14,     /* multiexp.ErrTooManyGoRoutines   internal/multiexp/errors.go:5 */
    MKZG_ERR_INVALID_POLY_SIZE = 15,       /* kzg.ErrInvalidPolynomialSize */
    MKZG_ERR_BAD_TRUSTED_SETUP = 16,       /* panics / kzg.ErrMinSRSSize of api.go:90-117, trusted_setup.go */
    MKZG_ERR_BAD_ARGS = 17,
    MKZG_ERR_HIP = 100                     /* HIP runtime failure or no usable device (no reference analogue) */
};

typedef struct mkzg_ctx mkzg_ctx;

/* ---- context ------------------------------------------------------------------------------ */
/* NewContext4096 (reference api.go:90-149) from the JSON of trusted_setup.go:23-27
 * (`g1_monomial`, `g1_lagrange`, `g2_monomial` hex strings).  device_ids/n_dev select the HIP
 * devices (n_dev == 0: the current device).  With n_dev > 1 the context drives all listed GPUs: host-pointer batches
 * are sharded over them (one host thread per device, partial results of the batch verifiers combined in process),
 * single-item calls go round-robin, device-pointer entry points run on the device that owns the pointers.  A device
 * id that does not exist is MKZG_ERR_BAD_ARGS.  Multi-process jobs (one rank per GPU) create one single-device context
 * per rank and use the _partial / _finish entry points below (DESIGN.md "multi-GPU"). */
int mkzg_ctx_new(const char* trusted_setup_json, size_t len, const int* device_ids, int n_dev, mkzg_ctx** out);
void mkzg_ctx_free(mkzg_ctx* ctx);
/* The reference's NewContext4096 does ALL of its precomputation at construction (api.go:90-149: commit key, FK20 tables through
 * fk20.NewFK20 at api.go:137, the recovery domain), so the first call on a Go Context costs what every later one does.  Here the big
 * tables are built by the first call that needs them (0.27 s for the 146 GB commitment table, 0.47 s for the FK20 table), so that a
 * context which only ever verifies does not pin 193 GB of HBM.  A caller that wants the reference's behaviour calls this once after
 * mkzg_ctx_new (and after any option that sizes the tables): `what` is a mask of MKZG_PREPARE_COMMIT (the commitment MSM's table as
 * "commit_direct_bits" / "msm_window" ask for it, degrading as documented there), MKZG_PREPARE_CELLS (NTT twiddles, the monomial SRS,
 * the FK20 window table, the recovery tables).  Every device of a multi-device context is prepared.  Idempotent; calls in flight on the
 * context are not disturbed. */
#define MKZG_PREPARE_COMMIT 1u
#define MKZG_PREPARE_CELLS 2u
#define MKZG_PREPARE_ALL 3u
int mkzg_ctx_prepare(mkzg_ctx* ctx, unsigned what);
/* Tuning knobs.  Changing one waits for the calls in flight on the context.
 *   "msm_window"          11..13: window width of the bucket method
 *   "max_batch"           blobs per launch of the commitment kernels (workspace size)
 *   "commit_direct_bits"  -1 (library default) = the widest direct window table that leaves room in the device's free memory for the
 *                         widest FK20 table "fk20_bits" allows (unless one is built already), a third of a gigabyte per lane and 4 GB of
 *                         slack -- 62 GB with the defaults, so that an empty 288 GB device holds both tables at their default widths
 *                         whichever is built first; 12..15 = that width (24 GB .. 146 GB of HBM; fewer group additions the wider); 0 = bucket method
 *                         (10.5 MB table + 0.5 GB workspace per lane).  The table is built by the first commitment / proof call.  If
 *                         it cannot be allocated the next narrower one is tried and finally the bucket method is used: the call
 *                         still succeeds with the same bytes; "commit_direct_bits_effective" (get) reports what is in use.
 *                         Measured on one MI355X, 256 blobs per call: 95 k blobs/s device-resident and 71 k from host pointers
 *                         with the 15-bit table, 62 k and 37 k with the bucket method -- hence the default.
 *   "fk20_bits"           widest window of the FK20 fixed-base table to try: 12 (default, 47 GB: 22 group additions per scalar), 11 (25.8 GB: 24
 *                         additions, 3 % fewer blobs/s on batches of 64 and more), 13 (86 GB: 20 instead of 24 group
 *                         additions per scalar in the 128 MSMs of ComputeCellsAndKZGProofs / recovery, but its gathers over 86 GB are
 *                         TLB-bound: measured no faster on MI355X) or 8 (4.3 GB, 32 additions per scalar).  When the HBM is short the
 *                         next narrower table is built instead (13 -> 12 -> 11 -> 8: same bytes out, "fk20_bits_effective" reports what is
 *                         in use); only when not even 4.3 GB are free does the call fail with MKZG_ERR_HIP
 *   "host_wait"           how a call waits for its device work: 0 = the HIP runtime's own wait, which POLLS (a busy CPU per waiting caller:
 *                         lowest latency), 1 = ask and sleep (an event query every 20-60 us: ~0.1 ms more latency, next to no CPU),
 *                         -1 (default) = sleep when more calls are in flight on the context than half the CPUs of its share ("cpu_share",
 *                         get only: the cgroup's CPUs divided by LOCAL_WORLD_SIZE / the context's devices), else poll.  Measured, 32
 *                         callers of VerifyBlobKZGProofBatch(1024) on 16 CPUs: 14.6 CPUs busy and 1.55-1.62 M blobs/s polling, 3.8 CPUs
 *                         busy and 1.49-1.50 M sleeping
 *   "fk20_segments"       1 (default): the 686 fixed-scalar products of ComputeCellsAndKZGProofs / recovery calls of 65..192 blobs run as
 *                         cost-equal segments dealt one wavefront per SIMD (128 blobs: 8.6 -> 7.5 ms); 0: whole products.  Same bytes.
 *   "debug_serial"        measurement aid: 1 = the blob verifier's challenge hash runs alone on the chip (the call waits for its side stream first)
 *   "lanes"               1..32 (default 32): calls that may be in flight on the context at once; lanes above the limit free their memory
 *   "trim"                (set only, any value) frees every lane's workspaces and pinned staging; they are rebuilt on demand
 *   "host_chunk"          blobs per chunk of the host-pointer commitment pipeline (gather / upload / kernels overlap per chunk; 128)
 *   "host_gather_tasks"   host-pool tasks per chunk of that pipeline's gather (1..64; 16)
 *   "cells_chunk"         1..512 (default 256): blobs per piece of a ComputeCellsAndKZGProofs / recovery batch (a lane's FK20 workspace grows
 *                         to the pieces it has seen, 1.6 MB per blob)
 *   "host_zero_copy"      1 (default): the commitment kernel reads the gathered blobs straight from pinned host memory over PCIe
 *                         (2 KB per wavefront per 17 group additions); 0: an explicit copy to HBM precedes each chunk's kernel
 *   "host_validate"       1 (default): the single-proof entry points (VerifyKZGProof, VerifyBlobKZGProof, ComputeBlobKZGProof) and host-pointer
 *                         VerifyBlobKZGProofBatch calls of up to 32 items (a block carries at most nine blobs: EIP-7691) and VerifyCellKZGProofBatch calls of up to
 *                         96 points (distinct commitments + proofs: a column sidecar is one cell of every blob of the block) decode and subgroup-test their points on
 *                         the calling thread and the host pool (~45 us per point; same checks, same codes) instead of waiting for the
 *                         device's chain (0.64 ms on a quad of lanes however few the points); 0: the device validates.
 *                         Larger batches and the device-pointer forms always validate on the device
 *   "host_fin"            1 (default): host-pointer commitment calls of up to four blobs (the single-blob entry point, small groups of
 *                         the call combiner) get the sum's projective coordinates back and convert + compress on the calling thread (~8 us
 *                         per blob) instead of waiting for the one-wavefront inversion on the device (~90 us); 0: the device finishes
 *   "host_sha"            host-pointer verifiers: Fiat-Shamir challenges hashed by the host pool while the blobs upload (1),
 *                         on the device after the upload (0), library's choice (-1, default: a call that is alone on the context hashes on
 *                         the host -- pairs of transcripts interleaved per core, 35 us each, under the chunked upload, the device taking
 *                         the first transcripts of very large pieces -- up to 160 blobs per pool thread; with other calls in flight 20,
 *                         then 8 per thread, and the device's shared hash rounds above that);
 *                         ComputeBlobKZGProof follows the same switch
 *   "verify_shape"        VerifyBlobKZGProofBatch: 0 = latency shape (the call's own hash launch with the points' chains and the two
 *                         challenge-free MSMs beside it on side streams; a quad of lanes per point in the subgroup tests), 1 =
 *                         throughput shape (one stream per call, hash launches shared between the calls in flight, one lane per point:
 *                         half the instructions, longer dependent chains), -1 (default) = throughput shape when at least two other
 *                         calls are in flight on the context.  Either way y = f(z) is one pass over the blob once its challenge is
 *                         known (the numerator tree of the barycentric formula).  Same results either way. */
int mkzg_ctx_set_option(mkzg_ctx* ctx, const char* key, long value);
/* reads an option back; also "commit_direct_bits_effective", "fk20_bits_effective", "devices", "host_threads" (threads of one device's
 * host pool including the caller: the CPUs the cgroup grants divided by LOCAL_WORLD_SIZE or the context's devices; the environment
 * variable MKZG_HOST_THREADS overrides), "pool_node_cpus" (CPUs the pool's workers are bound to when MKZG_POOL_AFFINITY=1 is set in the
 * environment: those of the NUMA node the GPU hangs off, within the process's own affinity mask -- for hosts that keep their blobs on
 * that node; 0 = the workers float, the default) */
int mkzg_ctx_get_option(mkzg_ctx* ctx, const char* key, long* value);

/* ---- EIP-4844 ------------------------------------------------------------------------------ */
/* Context.BlobToKZGCommitment (reference prove.go:13-33).  num_threads mirrors numGoRoutines:
 * ignored by the GPU path but >= 1024 still returns MKZG_ERR_TOO_MANY_GOROUTINES
 * (reference internal/multiexp/multiexp.go:45-50). */
int mkzg_blob_to_kzg_commitment(mkzg_ctx* ctx, const uint8_t* blob, int num_threads, uint8_t out48[48]);
/* batch of independent blobs given as an array of host pointers (Go's []*Blob); per_item_status
 * (may be NULL) receives one code per blob; the return value is the first non-zero item code. */
int mkzg_blob_to_kzg_commitment_batch(mkzg_ctx* ctx, const uint8_t* const* blobs, size_t n, uint8_t* out48n, int* per_item_status);
/* same, blobs contiguous in HBM (n x 131072 B), outputs to HBM; asynchronous on `stream`. */
int mkzg_blob_to_kzg_commitment_batch_device(mkzg_ctx* ctx, const void* d_blobs, size_t n, void* d_out48n, int* d_status, void* stream);

/* fine seam: multiexp.MultiExpG1 (reference internal/multiexp/multiexp.go:20-26) against the
 * context's commit key: n <= 4096 scalars (32-byte big-endian, canonical) -> compressed point.
 * Mirrors kzg.CommitKey.Commit's length check (reference internal/kzg/srs.go:56-62). */
int mkzg_commit_lagrange(mkzg_ctx* ctx, const uint8_t* scalars32, size_t n, int num_threads, uint8_t out48[48]);

/* Context.ComputeKZGProof (reference prove.go:85-111): proof and claimed value y = f(z). */
int mkzg_compute_kzg_proof(mkzg_ctx* ctx, const uint8_t* blob, const uint8_t z32[32], int num_threads, uint8_t proof48[48], uint8_t y32[32]);
/* Context.ComputeBlobKZGProof (reference prove.go:46-77). */
int mkzg_compute_blob_kzg_proof(mkzg_ctx* ctx, const uint8_t* blob, const uint8_t commitment48[48], int num_threads, uint8_t out48[48]);
/* same for a device-resident batch (blobs n x 131072 B, commitments n x 48 B -> proofs n x 48 B, per-item status) */
int mkzg_compute_blob_kzg_proof_batch_device(mkzg_ctx* ctx, const void* d_blobs, const void* d_commitments48, size_t n, void* d_proofs48, int* d_status, void* stream);
/* Context.VerifyKZGProof (reference verify.go:12-43). */
int mkzg_verify_kzg_proof(mkzg_ctx* ctx, const uint8_t commitment48[48], const uint8_t z32[32], const uint8_t y32[32], const uint8_t proof48[48]);
/* Context.VerifyBlobKZGProof (reference verify.go:48-83). */
int mkzg_verify_blob_kzg_proof(mkzg_ctx* ctx, const uint8_t* blob, const uint8_t commitment48[48], const uint8_t proof48[48]);
/* Context.VerifyBlobKZGProofBatch (reference verify.go:88-143; also covers ...BatchPar, verify.go:152-169,
 * which must return the same verdict).  blobs: array of host pointers; the three counts mirror the
 * three Go slices (mismatch -> MKZG_ERR_BATCH_LENGTH).  Any number of proofs. */
int mkzg_verify_blob_kzg_proof_batch(mkzg_ctx* ctx, const uint8_t* const* blobs, const uint8_t* commitments48n, const uint8_t* proofs48n, size_t n_blobs, size_t n_comm,
                                     size_t n_proofs);
/* same, inputs contiguous in HBM; synchronises `stream` internally (the verdict needs the host pairing). */
int mkzg_verify_blob_kzg_proof_batch_device(mkzg_ctx* ctx, const void* d_blobs, const void* d_commitments48, const void* d_proofs48, size_t n, void* stream);

/* Sharded VerifyBlobKZGProofBatch for multi-GPU jobs (one context/process per GPU): every rank folds its
 * shard (global indices offset .. offset+n_local) with the batch-wide scalar r32 (32 bytes big-endian, chosen by
 * rank 0 and broadcast) into a 512-byte record; after an all-gather any rank combines the records and runs the
 * single pairing check.  Equivalent to reference internal/kzg/kzg_verify.go:111-202 on the concatenated batch.
 * SOUNDNESS: the index ranges [offset, offset + n_local) of the ranks must tile the batch exactly -- two items folded with
 * the same power of r can cancel each other's errors, and the verdict is still an ordinary accept / reject. */
int mkzg_verify_blob_kzg_proof_batch_partial_device(mkzg_ctx* ctx, const void* d_blobs, const void* d_commitments48, const void* d_proofs48, size_t n_local,
                                                    uint64_t offset, const uint8_t r32[32], uint8_t record512[512], void* stream);
int mkzg_verify_blob_kzg_proof_batch_finish(mkzg_ctx* ctx, const uint8_t* records512, size_t n_records);

/* ---- multi-process jobs: the batch verifiers' exchange step inside the library ------------------------------------------------
 * One process per GPU.  Commitments, proofs, cells and recovery shard by blob and need nothing below.  The two batch
 * verifiers fold a random linear combination over the WHOLE batch (reference internal/kzg/kzg_verify.go:111-202,
 * internal/kzg_multi/kzg_verify.go:16-105), so the ranks meet once: ONE all-gather of 48 bytes per rank -- the shard sizes (offset =
 * prefix sum: the shards must tile the announced batch) and, from rank 0, the random scalar r -- put in flight while the rank already
 * hashes, decodes and uploads its shard (none of that needs r), then the all-gather of one 512-byte record per rank; every rank adds
 * the partial sums and runs the one pairing check.  The collectives are RCCL's, issued by the library: the host program (Go over cgo)
 * needs no RCCL binding of its own, and librccl is dlopen'ed by the first call below -- single-GPU users need no RCCL installed
 * (MKZG_RCCL_LIB overrides the library name).  The `_partial` / `_finish` pairs above remain for hosts that move the records
 * themselves (go-eth-kzg_amd/distributed.py does, through torch.distributed).
 *   mkzg_comm_unique_id   rank 0: 128 opaque bytes for every rank of the job, by whatever channel the host program has
 *   mkzg_comm_init        collective: same id, own rank, own single-device context -> a communicator (ncclCommInitRank inside)
 *   mkzg_comm_from_nccl   instead: wrap an ncclComm_t the host program already has (not destroyed by mkzg_comm_free)
 * The `_exchange` calls are COLLECTIVE: every rank of the communicator makes the same call with its own shard (the ranks' shards in
 * rank order are the batch; a rank may bring none) and the same n_global, and every rank gets the verdict of the whole batch: 0,
 * MKZG_VERIFY_FAILED, the input error of the lowest failing rank (the reference's precedence over the concatenated batch,
 * verify.go:102-123), or MKZG_ERR_BATCH_LENGTH on every rank when the shards do not add up to n_global.  One exchange at a time per
 * communicator (serialised inside). */
typedef struct mkzg_comm mkzg_comm;
int mkzg_comm_unique_id(uint8_t id128[128]);
int mkzg_comm_init(mkzg_ctx* ctx, const uint8_t id128[128], int rank, int world, mkzg_comm** out);
int mkzg_comm_from_nccl(mkzg_ctx* ctx, void* nccl_comm, int rank, int world, mkzg_comm** out);
void mkzg_comm_free(mkzg_comm* comm);         /* no exchange may be in flight on it; before mkzg_ctx_free of its context */
/* microseconds this rank WAITED in the two collectives of its last exchange (slower ranks' lateness included; the part of the first
 * collective that ran beside the rank's own hashing / decoding / uploads is not) */
double mkzg_comm_last_exchange_us(mkzg_comm* comm);
/* Context.VerifyBlobKZGProofBatch (reference verify.go:88-143) over the ranks: host pointers as in the reference ... */
int mkzg_verify_blob_kzg_proof_batch_exchange(mkzg_comm* comm, const uint8_t* const* blobs, const uint8_t* commitments48n, const uint8_t* proofs48n, size_t n_local,
                                              uint64_t n_global);
/* ... or this rank's shard contiguous in HBM */
int mkzg_verify_blob_kzg_proof_batch_exchange_device(mkzg_comm* comm, const void* d_blobs, const void* d_commitments48, const void* d_proofs48, size_t n_local, uint64_t n_global,
                                                     void* stream);
/* Context.VerifyCellKZGProofBatch (reference api_eip7594.go:163-214) over the ranks: this rank's (commitment, index, cell, proof) tuples */
int mkzg_verify_cell_kzg_proof_batch_exchange(mkzg_comm* comm, const uint8_t* commitments48n, const uint64_t* cell_indices, const uint8_t* const* cells,
                                              const uint8_t* proofs48n, size_t n_local, uint64_t n_global);
/* ... or this rank's cells contiguous in HBM (n_local x 2048 B, in the order of the other three arrays, which stay host arrays as in
 * the reference): no gather and no upload of 2 KiB per cell */
int mkzg_verify_cell_kzg_proof_batch_exchange_device(mkzg_comm* comm, const uint8_t* commitments48n, const uint64_t* cell_indices, const void* d_cells,
                                                     const uint8_t* proofs48n, size_t n_local, uint64_t n_global, void* stream);

/* Domain.EvaluateLagrangePolynomial (reference internal/domain/domain.go:193-235, with DeserializeBlob's canonicity check,
 * serialization.go:134-146): y_i = f_i(z_i) for n device-resident blobs and n caller-chosen evaluation points (host memory, n x 32 B
 * big-endian, canonical), points of the domain included (domain.go:163-171, :199-203).  The verifiers derive z from a hash and the
 * proof path returns y beside its proof; this entry point evaluates alone, through each of the three kernels the library has for it:
 *   form 0  inverse NTT + Horner on the coefficients (the EIP-7594 transform; no verifier uses it for this any more: a cross-check),
 *   form 1  the numerator tree of the barycentric formula over the product tree of the domain, no inversion (the batch verifier's),
 *   form 2  barycentric formula with one batch inversion per lane (the proof path's kernel, whose quotient is discarded here).
 * All forms must return the same bytes.  y32n / status: host memory (n x 32 B, n ints); status[i] is the blob's code, else z's.
 * Returns the first non-zero status.  n <= 4096.  Synchronises `stream`. */
int mkzg_evaluate_blobs_device(mkzg_ctx* ctx, const void* d_blobs, const uint8_t* z32n, size_t n, int form, uint8_t* y32n, int* status, void* stream);

/* The finer seam: multiexp.MultiExpG1 (reference internal/multiexp/multiexp.go:20-26) with gnark-crypto's in-memory layouts.
 * scalars: n x 4 u64 (fr.Element, little-endian limbs, Montgomery R = 2^256); points: n x 12 u64 (G1Affine = X, Y as fp.Element,
 * Montgomery R = 2^384; the identity is (0, 0)); out: one G1Affine in the same layout.  No point validation (as the reference). */
int mkzg_msm_g1(mkzg_ctx* ctx, const uint64_t* scalars_mont4, const uint64_t* points_mont12, size_t n, uint64_t out_mont12[12]);

/* ---- EIP-7594 ------------------------------------------------------------------------------ */
/* Context.ComputeCells (reference api_eip7594.go:12-26): 128 cells of 2048 bytes, contiguous. */
int mkzg_compute_cells(mkzg_ctx* ctx, const uint8_t* blob, int num_threads, uint8_t* cells_out /* 128 x 2048 */);
int mkzg_compute_cells_batch_device(mkzg_ctx* ctx, const void* d_blobs, size_t n, void* d_cells, int* d_status, void* stream);

/* Context.ComputeCellsAndKZGProofs (reference api_eip7594.go:28-52): cells 128 x 2048 B, proofs 128 x 48 B.
 * The first call builds the FK20 tables (fk20.NewFK20, reference fk20.go:23-52; 47 GB of HBM, 0.5 s; see "fk20_bits"). */
int mkzg_compute_cells_and_kzg_proofs(mkzg_ctx* ctx, const uint8_t* blob, int num_threads, uint8_t* cells_out, uint8_t* proofs_out);
int mkzg_compute_cells_and_kzg_proofs_batch_device(mkzg_ctx* ctx, const void* d_blobs, size_t n, void* d_cells, void* d_proofs, int* d_status, void* stream);

/* Context.VerifyCellKZGProofBatch (reference api_eip7594.go:163-214): one commitment, cell index, cell and
 * proof per entry; the four counts mirror the four Go slices.  Any number of cells. */
int mkzg_verify_cell_kzg_proof_batch(mkzg_ctx* ctx, const uint8_t* commitments48n, size_t n_commitments, const uint64_t* cell_indices, size_t n_indices,
                                     const uint8_t* const* cells, size_t n_cells, const uint8_t* proofs48n, size_t n_proofs);
/* same with the cells -- 2048 of a tuple's 2152 bytes -- contiguous in HBM (n x 2048 B, in the order of the other three arrays, which
 * stay host arrays as in the reference); skips the gather and the 16 MiB upload of an 8192-cell call.  Synchronises `stream`. */
int mkzg_verify_cell_kzg_proof_batch_device(mkzg_ctx* ctx, const uint8_t* commitments48n, const uint64_t* cell_indices, const void* d_cells, const uint8_t* proofs48n, size_t n,
                                            void* stream);
/* Sharded form for multi-GPU jobs (the algebra of reference internal/kzg_multi/kzg_verify.go:16-105 split at its
 * four group sums): a rank folds ITS cells -- global positions offset .. offset + n - 1 of the batch -- with the batch-wide
 * scalar r32 (32 bytes, big-endian, reduced mod r; the same on every rank) into a 512-byte record: int32 status at 0, then
 * compressed sum r^k proof_k at 16, sum w_row C_row at 64, [I(tau)]G1 at 112, sum r^k h_k^64 proof_k at 160.  Records of all
 * ranks (all-gathered, 512 B each) go to _finish on any rank: sums are added, one pairing check decides.  As above, the
 * ranges [offset, offset + n) of the ranks must tile the batch exactly. */
int mkzg_verify_cell_kzg_proof_batch_partial(mkzg_ctx* ctx, const uint8_t* commitments48n, size_t n_commitments, const uint64_t* cell_indices, size_t n_indices,
                                             const uint8_t* const* cells, size_t n_cells, const uint8_t* proofs48n, size_t n_proofs, uint64_t offset,
                                             const uint8_t r32[32], uint8_t record512[512]);
int mkzg_verify_cell_kzg_proof_batch_partial_device(mkzg_ctx* ctx, const uint8_t* commitments48n, const uint64_t* cell_indices, const void* d_cells, const uint8_t* proofs48n,
                                                    size_t n, uint64_t offset, const uint8_t r32[32], uint8_t record512[512], void* stream);
int mkzg_verify_cell_kzg_proof_batch_finish(mkzg_ctx* ctx, const uint8_t* records512, size_t n_records);

/* Context.RecoverCellsAndComputeKZGProofs (reference api_eip7594.go:144-161): cell_ids strictly ascending, < 128,
 * at least 64 of them; cells are host pointers in the same order. */
int mkzg_recover_cells_and_kzg_proofs(mkzg_ctx* ctx, const uint64_t* cell_ids, size_t n_ids, const uint8_t* const* cells, size_t n_cells, int num_threads,
                                      uint8_t* cells_out, uint8_t* proofs_out);
/* n_blobs independent recoveries in one call (no reference analogue: the throughput form for sharded recovery jobs).
 * cell_ids[b] / cells[b]: the n_ids[b] == n_cells[b] supplied cells of blob b, each list validated like the single-blob call;
 * cells_out: n_blobs x 128 x 2048 B, proofs_out: n_blobs x 128 x 48 B; status[b]: the reference's error for blob b (its outputs are
 * then left untouched).  Returns the first non-zero status. */
int mkzg_recover_cells_and_kzg_proofs_batch(mkzg_ctx* ctx, size_t n_blobs, const uint64_t* const* cell_ids, const size_t* n_ids, const uint8_t* const* const* cells,
                                            const size_t* n_cells, uint8_t* cells_out, uint8_t* proofs_out, int* status);
/* Context.RecoverCells (reference api_eip.go:8-15). */
int mkzg_recover_cells(mkzg_ctx* ctx, const uint64_t* cell_ids, size_t n_ids, const uint8_t* const* cells, size_t n_cells, int num_threads, uint8_t* cells_out);
/* RecoverCellsAndComputeKZGProofs / RecoverCells (reference api_eip7594.go:144-161, api_eip.go:8-15) for a batch whose supplied cells
 * are already in HBM and whose results stay there: d_cells = blob b's n_ids[b] cells in the order of its ids, the blobs back to back
 * (every blob's cells, rejected ones included: the layout is the caller's); d_cells_out n_blobs x 128 x 2048 B; d_proofs_out n_blobs x
 * 128 x 48 B, or NULL for the cells alone; ids and per-blob status on the host.  Outputs of blobs with a non-zero status are undefined.
 * Returns the first non-zero status.  Synchronises `stream`. */
int mkzg_recover_cells_and_kzg_proofs_batch_device(mkzg_ctx* ctx, size_t n_blobs, const uint64_t* const* cell_ids, const size_t* n_ids, const void* d_cells, void* d_cells_out,
                                                   void* d_proofs_out, int* status, void* stream);

/* ---- profiling hooks (bench.py's roofline section) ------------------------------------------ */
/* When enabled, every kernel launch of the hot path is bracketed by hipEvents on its stream. */
int mkzg_profile_enable(mkzg_ctx* ctx, int on);
/* Synchronises, then writes up to `cap` records; returns the number of kernel kinds. */
typedef struct { const char* name; double total_ms; long launches; } mkzg_kernel_time;
int mkzg_profile_read(mkzg_ctx* ctx, mkzg_kernel_time* out, int cap, int reset);

/* host-side share (Horner + [e]G + pairing) of the last verification call, in microseconds */
double mkzg_last_host_finish_us(mkzg_ctx* ctx);
/* diagnostic: sustained v_mad_u64_u32 issue rate of this GPU (wave-instructions per second, two wavefronts per SIMD);
 * bench.py quotes the kernels' VALU efficiency against it (boxes of one pool differ by up to 1.5x) */
int mkzg_valu_calibrate(mkzg_ctx* ctx, double* wave_instr_per_s);

const char* mkzg_version(void);

#ifdef __cplusplus
}
#endif
#endif
