// api_exchange.hip -- the batch verifiers' ONE exchange step as a product feature (include/mi355kzg.h, "multi-process jobs").
//
// A multi-GPU job runs one process per GPU.  Commitments, proofs, cells and recovery shard by blob with no
// collective; the two batch verifiers (reference verify.go:88-143 + internal/kzg/kzg_verify.go:111-202; api_eip7594.go:163-214 +
// internal/kzg_multi/kzg_verify.go:16-105) fold a random linear combination over the WHOLE batch, so the ranks meet once:
//     all-gather of the shard sizes (8 B per rank)   -> offset(rank) = prefix sum; the shards must tile the announced batch
//     broadcast of r (32 B, drawn by rank 0)          -> every item of the global batch is folded with its own power of r
//     all-gather of one 512-byte record per rank      -> any rank adds the partial group sums and runs the one pairing check
// RCCL has no group-element reduction, hence gather-then-add.  The collectives are RCCL's (ncclAllGather / ncclBroadcast over xGMI),
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
    int (*Broadcast)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
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
        api.Broadcast = (decltype(api.Broadcast))sym("ncclBroadcast");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
        if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather || !api.Broadcast) {
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
// all-gather of `bytes` (<= 512) per rank: h_send -> h_recv[world][bytes]
static int comm_all_gather(mkzg_comm* m, const void* h_send, size_t bytes, void* h_recv) {
    RcclApi* R = rccl();
    memcpy(m->h_buf, h_send, bytes);
    HIPCHK(hipMemcpyAsync(m->d_buf, m->h_buf, bytes, hipMemcpyHostToDevice, m->stream));
    NCCLCHK(R->AllGather(m->d_buf, m->d_buf + 512, bytes, NCCL_UINT8, m->comm, m->stream));
    HIPCHK(hipMemcpyAsync(m->h_buf + 512, m->d_buf + 512, bytes * (size_t)m->world, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    memcpy(h_recv, m->h_buf + 512, bytes * (size_t)m->world);
    return MKZG_OK;
}
static int comm_broadcast32(mkzg_comm* m, uint8_t r32[32]) {
    RcclApi* R = rccl();
    memcpy(m->h_buf, r32, 32);
    HIPCHK(hipMemcpyAsync(m->d_buf, m->h_buf, 32, hipMemcpyHostToDevice, m->stream));
    NCCLCHK(R->Broadcast(m->d_buf, m->d_buf, 32, NCCL_UINT8, 0, m->comm, m->stream));
    HIPCHK(hipMemcpyAsync(m->h_buf, m->d_buf, 32, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    memcpy(r32, m->h_buf, 32);
    return MKZG_OK;
}
// sizes -> (offset of this rank, whether the shards tile the announced batch); then r from rank 0.  r < 2^254 < the group order: the two
// partial-fold entry points take it reduced, and 254 random bits are what the soundness of the combination needs
static int exchange_setup(mkzg_comm* m, uint64_t n_local, uint64_t n_global, uint64_t* offset, uint8_t r32[32]) {
    std::vector<uint64_t> sizes((size_t)m->world);
    int rc = comm_all_gather(m, &n_local, 8, sizes.data());
    if (rc) return rc;
    uint64_t sum = 0, off = 0;
    for (int k = 0; k < m->world; k++) { if (k < m->rank) off += sizes[(size_t)k]; sum += sizes[(size_t)k]; }
    *offset = off;
    memset(r32, 0, 32);
    if (m->rank == 0) {
        if (getrandom(r32, 32, 0) == 32) r32[0] &= 0x3f;
        else memset(r32, 0xff, 32);                       // no randomness: say so to every rank THROUGH the broadcast (a rank that returned here would leave the others waiting in it)
    }
    rc = comm_broadcast32(m, r32);
    if (rc) return rc;
    if (r32[0] == 0xff) return MKZG_ERR_HIP;
    // every rank sees the same sizes, so every rank refuses the same way -- after the broadcast, so that no rank is left waiting in it
    if (sum != n_global) {
        fprintf(stderr, "[mi355kzg] rank %d: the shards hold %llu items in all, the batch was announced as %llu: the index ranges would not tile it\n", m->rank,
                (unsigned long long)sum, (unsigned long long)n_global);
        return MKZG_ERR_BATCH_LENGTH;
    }
    return MKZG_OK;
}
// the verifiers' 512-byte record of a host-side fold (api_verify.hip: status | A | Cc | D | e)
static void blob_record(uint8_t record512[512], int rc, const VerifyFold& total) {
    memset(record512, 0, 512);
    int32_t st = rc;
    memcpy(record512, &st, 4);
    if (rc != MKZG_OK) return;
    G1AffH a;
    g1h_to_affine(a, total.A);
    g1h_compress(record512 + 16, a);
    g1h_to_affine(a, total.Cc);
    g1h_compress(record512 + 64, a);
    g1h_to_affine(a, total.D);
    g1h_compress(record512 + 112, a);
    for (int i = 0; i < 8; i++) {
        const uint32_t w = total.e[7 - i];
        record512[160 + 4 * i] = (uint8_t)(w >> 24); record512[161 + 4 * i] = (uint8_t)(w >> 16); record512[162 + 4 * i] = (uint8_t)(w >> 8); record512[163 + 4 * i] = (uint8_t)w;
    }
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
// microseconds the three collectives of the last exchange took on this rank (waits for slower ranks included)
double mkzg_comm_last_exchange_us(mkzg_comm* m) { return m ? m->last_exchange_us : 0.0; }

// Context.VerifyBlobKZGProofBatch (reference verify.go:88-143) over a batch that is spread over the ranks of `comm`: this rank brings
// n_local (blob, commitment, proof) triples -- the ranks' shards in rank order are the batch -- and every rank gets the verdict of the
// WHOLE batch.  Collective: every rank of the communicator calls it, with the same n_global (the size of the whole batch as the caller
// announces it; shards that do not add up to it are refused on every rank with MKZG_ERR_BATCH_LENGTH).  An input error in any shard is
// returned on every rank, the lowest rank's first (verify.go:102-123 order over the concatenated batch).
int mkzg_verify_blob_kzg_proof_batch_exchange_device(mkzg_comm* m, const void* d_blobs, const void* d_commitments48, const void* d_proofs48, size_t n_local, uint64_t n_global,
                                                     void* stream) {
    if (!m || (n_local && (!d_blobs || !d_commitments48 || !d_proofs48))) return MKZG_ERR_BAD_ARGS;
    std::lock_guard<std::mutex> g(m->mu);
    DeviceScope dev(m->ctx->device);
    uint64_t offset = 0;
    uint8_t r32[32], rec[512];
    auto t0 = std::chrono::steady_clock::now();
    int rc = exchange_setup(m, n_local, n_global, &offset, r32);
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (rc) return rc;
    (void)mkzg_verify_blob_kzg_proof_batch_partial_device(m->ctx, d_blobs, d_commitments48, d_proofs48, n_local, offset, r32, rec, stream);     // the status travels in the record
    std::vector<uint8_t> all((size_t)m->world * 512);
    t0 = std::chrono::steady_clock::now();
    rc = comm_all_gather(m, rec, 512, all.data());
    m->last_exchange_us = us + std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (rc) return rc;
    return mkzg_verify_blob_kzg_proof_batch_finish(m->ctx, all.data(), (size_t)m->world);
}
// the same with the reference's argument shape: host pointers (Go's []*Blob), this rank's shard
int mkzg_verify_blob_kzg_proof_batch_exchange(mkzg_comm* m, const uint8_t* const* blobs, const uint8_t* commitments48n, const uint8_t* proofs48n, size_t n_local, uint64_t n_global) {
    if (!m || (n_local && (!blobs || !commitments48n || !proofs48n))) return MKZG_ERR_BAD_ARGS;
    std::lock_guard<std::mutex> g(m->mu);
    DeviceScope dev(m->ctx->device);
    uint64_t offset = 0;
    uint8_t r32[32], rec[512];
    auto t0 = std::chrono::steady_clock::now();
    int rc = exchange_setup(m, n_local, n_global, &offset, r32);
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (rc) return rc;
    uint8_t r_le[32];
    for (int i = 0; i < 32; i++) r_le[i] = r32[31 - i];
    VerifyFold total;
    verify_fold_clear(total);
    int prc = MKZG_OK;
    if (n_local) prc = sd_verify_blob_fold_host(m->ctx, blobs, commitments48n, proofs48n, n_local, offset, r_le, &total);
    blob_record(rec, prc, total);
    std::vector<uint8_t> all((size_t)m->world * 512);
    t0 = std::chrono::steady_clock::now();
    rc = comm_all_gather(m, rec, 512, all.data());
    m->last_exchange_us = us + std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (rc) return rc;
    return mkzg_verify_blob_kzg_proof_batch_finish(m->ctx, all.data(), (size_t)m->world);
}
// Context.VerifyCellKZGProofBatch (reference api_eip7594.go:163-214) over a batch spread over the ranks: this rank's n_local
// (commitment, cell index, cell, proof) tuples, host pointers as in the reference
int mkzg_verify_cell_kzg_proof_batch_exchange(mkzg_comm* m, const uint8_t* commitments48n, const uint64_t* cell_indices, const uint8_t* const* cells, const uint8_t* proofs48n,
                                              size_t n_local, uint64_t n_global) {
    if (!m || (n_local && (!commitments48n || !cell_indices || !cells || !proofs48n))) return MKZG_ERR_BAD_ARGS;
    std::lock_guard<std::mutex> g(m->mu);
    DeviceScope dev(m->ctx->device);
    uint64_t offset = 0;
    uint8_t r32[32], rec[512];
    auto t0 = std::chrono::steady_clock::now();
    int rc = exchange_setup(m, n_local, n_global, &offset, r32);
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (rc) return rc;
    (void)mkzg_verify_cell_kzg_proof_batch_partial(m->ctx, commitments48n, n_local, cell_indices, n_local, cells, n_local, proofs48n, n_local, offset, r32, rec);
    std::vector<uint8_t> all((size_t)m->world * 512);
    t0 = std::chrono::steady_clock::now();
    rc = comm_all_gather(m, rec, 512, all.data());
    m->last_exchange_us = us + std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (rc) return rc;
    return mkzg_verify_cell_kzg_proof_batch_finish(m->ctx, all.data(), (size_t)m->world);
}

}  // extern "C"
