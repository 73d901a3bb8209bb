// api_exchange.hip -- the batch verifiers' ONE exchange step as a product feature (include/mi355kzg.h, "multi-process jobs").
//
// A multi-GPU job runs one process per GPU.  Commitments, proofs, cells and recovery shard by blob with no
// collective; the two batch verifiers (reference verify.go:88-143 + internal/kzg/kzg_verify.go:111-202; api_eip7594.go:163-214 +
// internal/kzg_multi/kzg_verify.go:16-105) fold a random linear combination over the WHOLE batch, so the ranks meet once:
//     all-gather of 48 B per rank: the shard size, and r from rank 0  -> offset(rank) = prefix sum; the shards must tile the announced
//                                                                        batch; every item is folded with its own power of r
//     all-gather of one 512-byte record per rank                      -> any rank adds the partial group sums and runs the one pairing check
// RCCL has no group-element reduction, hence gather-then-add.  The collectives are RCCL's (ncclAllGather over xGMI),
// issued from HERE -- the host above the C ABI stays in the reference's language (Go over cgo) and needs no binding of its own to
// RCCL.  librccl is dlopen'ed by the first call that needs it: a single-GPU user of libmi355kzg.so needs no RCCL at all.
#include <dlfcn.h>
#include <sys/random.h>
#include <chrono>
#include <mutex>
#include <vector>
#include "ctx.h"
#include "api_multi.h"

namespace {

// the few RCCL entry points used, by their documented C ABI (rccl.h: ncclUniqueId is 128 opaque bytes passed by value)
struct NcclUniqueId { char internal[128]; };
typedef void* NcclComm;
enum { NCCL_SUCCESS = 0, NCCL_UINT8 = 1 };
struct RcclApi {
    void* handle = nullptr;
    int (*GetUniqueId)(NcclUniqueId*) = nullptr;
    int (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, NcclComm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
RcclApi* rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {getenv("MKZG_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            if ((api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        }
        if (!api.handle) { fprintf(stderr, "[mi355kzg] librccl not found (set MKZG_RCCL_LIB): %s\n", dlerror()); return; }
        auto sym = [&](const char* s) { return dlsym(api.handle, s); };
        api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
        if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather) {
            fprintf(stderr, "[mi355kzg] librccl lacks an expected symbol\n");
            api.handle = nullptr;
        }
    });
    return api.handle ? &api : nullptr;
}

}  // namespace

// one communicator of the job as this library sees it: the caller's ncclComm_t (or one made by mkzg_comm_init), this rank's context,
// and the few hundred bytes of device / pinned scratch the three collectives need.  Collectives on one communicator must be issued in
// the same order by every rank: an exchange holds `mu` from its first collective to its last.
struct mkzg_comm {
    mkzg_ctx* ctx = nullptr;
    NcclComm comm = nullptr;
    bool owned = false;
    int rank = 0, world = 1;
    hipStream_t stream = nullptr;
    uint8_t* d_buf = nullptr;       // [0, 512) send | [512, 512 + world * 512) receive
    uint8_t* h_buf = nullptr;       // pinned mirror
    std::mutex mu;
    double last_exchange_us = 0;
};

#define NCCLCHK(expr)                                                                                      \
    do {                                                                                                   \
        const int e__ = (expr);                                                                            \
        if (e__ != NCCL_SUCCESS) {                                                                         \
            fprintf(stderr, "[mi355kzg] RCCL error %d (%s) at %s:%d\n", e__, (rccl() && rccl()->GetErrorString) ? rccl()->GetErrorString(e__) : "?", __FILE__, __LINE__); \
            return MKZG_ERR_HIP;                                                                           \
        }                                                                                                  \
    } while (0)

static int comm_finish_setup(mkzg_comm* m) {
    DeviceScope dev(m->ctx->device);
    HIPCHK(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
    HIPCHK(hipMalloc(&m->d_buf, 512 + (size_t)m->world * 512));
    HIPCHK(hipHostMalloc(&m->h_buf, 512 + (size_t)m->world * 512, hipHostMallocDefault));
    return MKZG_OK;
}
// all-gather of `bytes` (<= 512) per rank in two halves: _begin puts the upload, the collective and the download on the communicator's
// stream and returns; _end waits for them (for the slowest rank, that is) and hands out h_recv[world][bytes]
static int comm_all_gather_begin(mkzg_comm* m, const void* h_send, size_t bytes) {
    RcclApi* R = rccl();
    memcpy(m->h_buf, h_send, bytes);
    HIPCHK(hipMemcpyAsync(m->d_buf, m->h_buf, bytes, hipMemcpyHostToDevice, m->stream));
    NCCLCHK(R->AllGather(m->d_buf, m->d_buf + 512, bytes, NCCL_UINT8, m->comm, m->stream));
    HIPCHK(hipMemcpyAsync(m->h_buf + 512, m->d_buf + 512, bytes * (size_t)m->world, hipMemcpyDeviceToHost, m->stream));
    return MKZG_OK;
}
static int comm_all_gather_end(mkzg_comm* m, size_t bytes, void* h_recv) {
    HIPCHK(hipStreamSynchronize(m->stream));
    memcpy(h_recv, m->h_buf + 512, bytes * (size_t)m->world);
    return MKZG_OK;
}
static int comm_all_gather(mkzg_comm* m, const void* h_send, size_t bytes, void* h_recv) {
    const int rc = comm_all_gather_begin(m, h_send, bytes);
    return rc ? rc : comm_all_gather_end(m, bytes, h_recv);
}

// One exchange, the same for both verifiers and both argument shapes:
//   1. ONE all-gather of 48 bytes per rank -- the shard size, a rank-local argument error if there is one (such a rank announces an
//      empty shard and still takes part in every collective: returning early would leave the other ranks waiting in them), and, from
//      rank 0, the random scalar r (254 bits: r < 2^254 < the group order, what the partial folds take as reduced).  Until round 5
//      this was an all-gather of sizes followed by a broadcast of r, both finished BEFORE any local work started; now the collective is
//      only put in flight here and the fold is started with a RlcLate (ctx.h): hashing, decoding, subgroup tests, gathers and uploads
//      of the shard need neither r nor the offset and run while the slowest rank is still arriving;
//   2. the local fold; where it first needs r it waits for the collective: offset = prefix sum of the sizes, and the shards must
//      tile the announced batch -- two items folded with the same power of r could cancel each other's errors, so the offsets are
//      never taken on trust: else MKZG_ERR_BATCH_LENGTH on every rank (every rank sees the same sizes);
//   3. all-gather of the 512-byte records, the finish on every rank.  A rank's error travels in its record: all ranks return the
//      same code, the lowest failing rank's first (verify.go:102-123 order over the concatenated batch).
struct SetupRecord { uint64_t n_local; int32_t arg_rc; uint32_t pad; uint8_t r_le[32]; };
static_assert(sizeof(SetupRecord) == 48, "setup record");

template <class Fold, class Finish>
static int exchange_run(mkzg_comm* m, int arg_rc, uint64_t n_local, uint64_t n_global, Fold&& fold /* int(RlcLate*, uint8_t rec[512]) */, Finish&& finish /* int(const uint8_t*, size_t) */) {
    std::lock_guard<std::mutex> g(m->mu);
    DeviceScope dev(m->ctx->device);
    SetupRecord mine;
    memset(&mine, 0, sizeof mine);
    mine.n_local = arg_rc ? 0 : n_local;
    mine.arg_rc = arg_rc;
    if (m->rank == 0) {
        if (getrandom(mine.r_le, 32, 0) == 32) mine.r_le[31] &= 0x3f;
        else if (!mine.arg_rc) mine.arg_rc = MKZG_ERR_HIP;               // no randomness: said to every rank THROUGH the collective
    }
    double wait_us = 0;
    int setup_rc = comm_all_gather_begin(m, &mine, sizeof mine);
    int setup_verdict = MKZG_OK;                                         // what every rank concludes from the setup records alone: the same everywhere
    RlcLate late;
    late.wait = [&](uint32_t r_words[8], uint64_t* offset) -> int {
        if (setup_rc) return setup_rc;
        std::vector<SetupRecord> all((size_t)m->world);
        const auto t0 = std::chrono::steady_clock::now();
        setup_rc = comm_all_gather_end(m, sizeof(SetupRecord), all.data());
        wait_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (setup_rc) return setup_rc;
        uint64_t sum = 0, off = 0;
        for (int k = 0; k < m->world; k++) {
            if (all[(size_t)k].arg_rc && !setup_verdict) setup_verdict = all[(size_t)k].arg_rc;
            if (k < m->rank) off += all[(size_t)k].n_local;
            sum += all[(size_t)k].n_local;
        }
        if (!setup_verdict && sum != n_global) {
            fprintf(stderr, "[mi355kzg] rank %d: the shards hold %llu items in all, the batch was announced as %llu: the index ranges would not tile it\n", m->rank,
                    (unsigned long long)sum, (unsigned long long)n_global);
            setup_verdict = MKZG_ERR_BATCH_LENGTH;
        }
        memcpy(r_words, all[0].r_le, 32);
        *offset = off;
        return setup_verdict;                                            // non-zero: the fold stops where it asked, the records' collective still runs
    };
    uint8_t rec[512];
    memset(rec, 0, sizeof rec);
    int frc = MKZG_OK;
    if (!arg_rc) frc = fold(&late, rec);                                 // (an empty shard folds to the identity sums: a well-formed record)
    (void)late.get();                                                    // a fold that never asked for r (or did not run) has not waited yet
    if (setup_rc) return setup_rc;                                       // the transport itself failed: nothing more can be agreed on
    {   // a rank that did not fold, or whose fold stopped before it wrote its record, still publishes its status
        const int32_t st = arg_rc ? arg_rc : frc;
        int32_t have;
        memcpy(&have, rec, 4);
        if (st && !have) { memset(rec, 0, sizeof rec); memcpy(rec, &st, 4); }
    }
    std::vector<uint8_t> all((size_t)m->world * 512);
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = comm_all_gather(m, rec, 512, all.data());
    m->last_exchange_us = wait_us + std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (rc) return rc;
    if (setup_verdict) return setup_verdict;
    return finish(all.data(), (size_t)m->world);
}

extern "C" {

// rank 0: 128 bytes to hand to every rank of the job by whatever channel the host program has (a file, a socket, its own RPC)
int mkzg_comm_unique_id(uint8_t id128[128]) {
    if (!id128) return MKZG_ERR_BAD_ARGS;
    RcclApi* R = rccl();
    if (!R) return MKZG_ERR_HIP;
    NcclUniqueId id;
    NCCLCHK(R->GetUniqueId(&id));
    memcpy(id128, id.internal, 128);
    return MKZG_OK;
}
// collective over the `world` ranks of the job: every rank calls it with the same id, its own rank and its own (single-device) context
int mkzg_comm_init(mkzg_ctx* ctx, const uint8_t id128[128], int rank, int world, mkzg_comm** out) {
    if (!ctx || !id128 || !out || world < 1 || rank < 0 || rank >= world || !ctx->peers.empty()) return MKZG_ERR_BAD_ARGS;
    *out = nullptr;
    RcclApi* R = rccl();
    if (!R) return MKZG_ERR_HIP;
    mkzg_comm* m = new mkzg_comm();
    m->ctx = ctx;
    m->rank = rank;
    m->world = world;
    NcclUniqueId id;
    memcpy(id.internal, id128, 128);
    {
        DeviceScope dev(ctx->device);
        const int e = R->CommInitRank(&m->comm, world, id, rank);
        if (e != NCCL_SUCCESS) {
            fprintf(stderr, "[mi355kzg] ncclCommInitRank failed: %d\n", e);
            delete m;
            return MKZG_ERR_HIP;
        }
    }
    m->owned = true;
    const int rc = comm_finish_setup(m);
    if (rc) { mkzg_comm_free(m); return rc; }
    *out = m;
    return MKZG_OK;
}
// around a communicator the host program already has (ncclComm_t from its own ncclCommInitRank, or the one a framework made): not owned
int mkzg_comm_from_nccl(mkzg_ctx* ctx, void* nccl_comm, int rank, int world, mkzg_comm** out) {
    if (!ctx || !nccl_comm || !out || world < 1 || rank < 0 || rank >= world || !ctx->peers.empty()) return MKZG_ERR_BAD_ARGS;
    *out = nullptr;
    if (!rccl()) return MKZG_ERR_HIP;
    mkzg_comm* m = new mkzg_comm();
    m->ctx = ctx;
    m->comm = nccl_comm;
    m->rank = rank;
    m->world = world;
    const int rc = comm_finish_setup(m);
    if (rc) { mkzg_comm_free(m); return rc; }
    *out = m;
    return MKZG_OK;
}
void mkzg_comm_free(mkzg_comm* m) {
    if (!m) return;
    {
        DeviceScope dev(m->ctx->device);
        if (m->stream) { (void)hipStreamSynchronize(m->stream); (void)hipStreamDestroy(m->stream); }
        if (m->d_buf) (void)hipFree(m->d_buf);
        if (m->h_buf) (void)hipHostFree(m->h_buf);
        if (m->owned && m->comm && rccl()) (void)rccl()->CommDestroy(m->comm);
    }
    delete m;
}
// microseconds this rank WAITED in the two collectives of its last exchange (slower ranks' lateness included; what overlapped with its own fold is not)
double mkzg_comm_last_exchange_us(mkzg_comm* m) { return m ? m->last_exchange_us : 0.0; }

// Context.VerifyBlobKZGProofBatch (reference verify.go:88-143) over a batch that is spread over the ranks of `comm`: this rank brings
// n_local (blob, commitment, proof) triples -- the ranks' shards in rank order are the batch -- and every rank gets the verdict of the
// WHOLE batch.  Collective: every rank of the communicator calls it, with the same n_global (the size of the whole batch as the caller
// announces it; shards that do not add up to it are refused on every rank with MKZG_ERR_BATCH_LENGTH).  An input error in any shard --
// a null argument of one rank included -- is returned on every rank, the lowest rank's first (verify.go:102-123 order over the
// concatenated batch).
int mkzg_verify_blob_kzg_proof_batch_exchange_device(mkzg_comm* m, const void* d_blobs, const void* d_commitments48, const void* d_proofs48, size_t n_local, uint64_t n_global,
                                                     void* stream) {
    if (!m) return MKZG_ERR_BAD_ARGS;
    const int arg_rc = (n_local && (!d_blobs || !d_commitments48 || !d_proofs48)) ? MKZG_ERR_BAD_ARGS : MKZG_OK;
    return exchange_run(m, arg_rc, n_local, n_global,
        [&](RlcLate* late, uint8_t rec[512]) {
            VerifyFold total;
            const int rc = sd_verify_blob_fold_device(m->ctx, d_blobs, d_commitments48, d_proofs48, n_local, 0, nullptr, (hipStream_t)stream, &total, late);
            verify_fold_record(rec, rc, total);
            return rc;
        },
        [&](const uint8_t* all, size_t world) { return mkzg_verify_blob_kzg_proof_batch_finish(m->ctx, all, world); });
}
// the same with the reference's argument shape: host pointers (Go's []*Blob), this rank's shard
int mkzg_verify_blob_kzg_proof_batch_exchange(mkzg_comm* m, const uint8_t* const* blobs, const uint8_t* commitments48n, const uint8_t* proofs48n, size_t n_local, uint64_t n_global) {
    if (!m) return MKZG_ERR_BAD_ARGS;
    const int arg_rc = (n_local && (!blobs || !commitments48n || !proofs48n)) ? MKZG_ERR_BAD_ARGS : MKZG_OK;
    return exchange_run(m, arg_rc, n_local, n_global,
        [&](RlcLate* late, uint8_t rec[512]) {
            VerifyFold total;
            const int rc = sd_verify_blob_fold_host(m->ctx, blobs, commitments48n, proofs48n, n_local, 0, nullptr, &total, late);
            verify_fold_record(rec, rc, total);
            return rc;
        },
        [&](const uint8_t* all, size_t world) { return mkzg_verify_blob_kzg_proof_batch_finish(m->ctx, all, world); });
}
// Context.VerifyCellKZGProofBatch (reference api_eip7594.go:163-214) over a batch spread over the ranks: this rank's n_local
// (commitment, cell index, cell, proof) tuples, host pointers as in the reference
int mkzg_verify_cell_kzg_proof_batch_exchange(mkzg_comm* m, const uint8_t* commitments48n, const uint64_t* cell_indices, const uint8_t* const* cells, const uint8_t* proofs48n,
                                              size_t n_local, uint64_t n_global) {
    if (!m) return MKZG_ERR_BAD_ARGS;
    const int arg_rc = (n_local && (!commitments48n || !cell_indices || !cells || !proofs48n)) ? MKZG_ERR_BAD_ARGS : MKZG_OK;
    return exchange_run(m, arg_rc, n_local, n_global,
        [&](RlcLate* late, uint8_t rec[512]) {
            CellFold f;
            const int rc = sd_cell_verify_fold(m->ctx, commitments48n, n_local, cell_indices, n_local, cells, n_local, proofs48n, n_local, 0, nullptr, &f, nullptr, nullptr, late);
            cell_fold_record(rec, rc, f);
            return rc;
        },
        [&](const uint8_t* all, size_t world) { return mkzg_verify_cell_kzg_proof_batch_finish(m->ctx, all, world); });
}
// ... with this rank's cells -- 2048 of a tuple's 2152 bytes -- contiguous in HBM (n_local x 2048 B, in the order of the other three
// arrays, which stay host arrays as in the reference): no gather, no 16 MiB upload per 8192 cells (round 5; the blob verifier has had
// its _exchange_device since round 4)
int mkzg_verify_cell_kzg_proof_batch_exchange_device(mkzg_comm* m, const uint8_t* commitments48n, const uint64_t* cell_indices, const void* d_cells, const uint8_t* proofs48n,
                                                     size_t n_local, uint64_t n_global, void* stream) {
    if (!m) return MKZG_ERR_BAD_ARGS;
    const int arg_rc = (n_local && (!commitments48n || !cell_indices || !d_cells || !proofs48n)) ? MKZG_ERR_BAD_ARGS : MKZG_OK;
    return exchange_run(m, arg_rc, n_local, n_global,
        [&](RlcLate* late, uint8_t rec[512]) {
            CellFold f;
            const int rc = sd_cell_verify_fold(m->ctx, commitments48n, n_local, cell_indices, n_local, nullptr, n_local, proofs48n, n_local, 0, nullptr, &f, (const uint8_t*)d_cells,
                                               (hipStream_t)stream, late);
            cell_fold_record(rec, rc, f);
            return rc;
        },
        [&](const uint8_t* all, size_t world) { return mkzg_verify_cell_kzg_proof_batch_finish(m->ctx, all, world); });
}

}  // extern "C"
