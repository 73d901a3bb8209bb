// tests/host_shim/stub_device_driver.cpp -- drives the product's HOST layer (go-eth-kzg_amd/csrc/api*.hip compiled as plain C++) through
// its public C ABI on the inert HIP runtime of stub_hip.cpp, under ThreadSanitizer: many caller threads on one shared context and on a
// two-device context, first calls racing for the lazily built tables, option changes (lanes, trim, verify_shape, table widths) while
// calls are in flight, profile read-outs, hash rounds (the blob verifier's throughput shape needs three calls in flight), the host
// pool's gather pipelines.  Outputs are meaningless (the kernels are stubs): what is checked is that no call reports a runtime error,
// that tables land on their context's device, that the caller's current device survives every call -- and that TSan stays silent.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "../../include/mi355kzg.h"

extern "C" int stub_hip_live_allocations();
static std::atomic<int> g_bad{0};
#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) { fprintf(stderr, "CHECK failed at line %d: %s\n", __LINE__, #cond); g_bad++; } \
    } while (0)
static bool soft(int rc) { return rc != MKZG_ERR_HIP && rc != MKZG_ERR_BAD_ARGS && rc != MKZG_ERR_BAD_TRUSTED_SETUP; }

static void caller(mkzg_ctx* ctx, int id, int rounds, int my_device) {
    CHECK(hipSetDevice(my_device) == hipSuccess);
    const int NB = 6, NC = 64;
    std::vector<std::vector<uint8_t>> blobs(NB, std::vector<uint8_t>(MKZG_BYTES_PER_BLOB, 0));
    for (int b = 0; b < NB; b++) blobs[b][31] = (uint8_t)(1 + id + b);
    std::vector<const uint8_t*> bp(NB);
    for (int b = 0; b < NB; b++) bp[b] = blobs[b].data();
    std::vector<uint8_t> cms(NB * 48, 0), prs(NB * 48, 0), out(NB * 48), cells(128 * 2048), proofs(128 * 48), y(32), z(32, 0);
    for (int b = 0; b < NB; b++) { cms[48 * b] = 0xc0; prs[48 * b] = 0xc0; }
    std::vector<int> st(NB);
    std::vector<std::vector<uint8_t>> cellv(128, std::vector<uint8_t>(2048, 0));
    std::vector<const uint8_t*> cp(NC);
    std::vector<uint64_t> idx(NC), ids(NC);
    std::vector<uint8_t> ccm(NC * 48, 0), cpr(NC * 48, 0);
    for (int i = 0; i < NC; i++) { cp[i] = cellv[i].data(); idx[i] = (uint64_t)(2 * i); ids[i] = (uint64_t)(2 * i); ccm[48 * i] = 0xc0; cpr[48 * i] = 0xc0; }
    void* d_blobs = nullptr;                       // "device" inputs for the device-pointer entry points, on the CONTEXT's first device
    void* d_cm = nullptr;
    void* d_pr = nullptr;
    void* d_out = nullptr;
    int* d_st = nullptr;
    CHECK(hipSetDevice(0) == hipSuccess);
    CHECK(hipMalloc(&d_blobs, (size_t)NB * MKZG_BYTES_PER_BLOB) == hipSuccess && hipMalloc(&d_cm, NB * 48) == hipSuccess && hipMalloc(&d_pr, NB * 48) == hipSuccess);
    CHECK(hipMalloc(&d_out, NB * 48) == hipSuccess && hipMalloc((void**)&d_st, NB * sizeof(int)) == hipSuccess);
    CHECK(hipSetDevice(my_device) == hipSuccess);
    hipStream_t s;
    CHECK(hipStreamCreateWithPriority(&s, 0, id & 1 ? 0 : -1) == hipSuccess);
    for (int r = 0; r < rounds; r++) {
        switch ((r + id) % 11) {
            case 0: CHECK(soft(mkzg_blob_to_kzg_commitment_batch(ctx, bp.data(), NB, out.data(), st.data()))); break;
            case 1: CHECK(soft(mkzg_verify_blob_kzg_proof_batch(ctx, bp.data(), cms.data(), prs.data(), NB, NB, NB))); break;
            case 2: CHECK(soft(mkzg_verify_blob_kzg_proof_batch_device(ctx, d_blobs, d_cm, d_pr, NB, s))); break;
            case 3: CHECK(soft(mkzg_compute_cells_and_kzg_proofs(ctx, bp[0], 0, cells.data(), proofs.data()))); break;
            case 4: {
                // (every third time with a nil cell: the gather that starts at the top of the call zero-fills its slot, the call reports it after the device work)
                const uint8_t* keep = cp[5];
                if (r % 3 == 0) cp[5] = nullptr;
                const int vrc = mkzg_verify_cell_kzg_proof_batch(ctx, ccm.data(), NC, idx.data(), NC, cp.data(), NC, cpr.data(), NC);
                cp[5] = keep;
                CHECK(vrc != MKZG_ERR_HIP && vrc != MKZG_ERR_BAD_TRUSTED_SETUP);
                break;
            }
            case 5: CHECK(soft(mkzg_recover_cells_and_kzg_proofs(ctx, ids.data(), NC, cp.data(), NC, 0, cells.data(), proofs.data()))); break;
            case 6: CHECK(soft(mkzg_compute_blob_kzg_proof(ctx, bp[1], cms.data(), 0, out.data()))); break;
            case 7: CHECK(soft(mkzg_blob_to_kzg_commitment_batch_device(ctx, d_blobs, NB, d_out, d_st, s))); break;
            // single-blob calls that meet on the context leave as one batch call (host_combine.hpp): leaders, members, the verifier's fallback
            case 8: CHECK(soft(mkzg_blob_to_kzg_commitment(ctx, bp[r % NB], 0, out.data()))); break;
            case 9: CHECK(soft(mkzg_verify_blob_kzg_proof(ctx, bp[r % NB], cms.data(), prs.data()))); break;
            default: CHECK(soft(mkzg_verify_kzg_proof(ctx, cms.data(), z.data(), y.data(), prs.data()))); break;
        }
        int dev = -1;
        CHECK(hipGetDevice(&dev) == hipSuccess && dev == my_device);          // every entry point restores the caller's device
    }
    (void)hipStreamDestroy(s);
    (void)hipFree(d_blobs); (void)hipFree(d_cm); (void)hipFree(d_pr); (void)hipFree(d_out); (void)hipFree(d_st);
}

static void tuner(mkzg_ctx* ctx, std::atomic<bool>* stop) {
    int i = 0;
    while (!stop->load()) {
        switch (i++ % 8) {
            case 7: CHECK(mkzg_ctx_prepare(ctx, 1u + (unsigned)(i % 3)) == MKZG_OK); break;      // tables up front, beside the calls that build them lazily and the options that drop them
            case 0: CHECK(mkzg_ctx_set_option(ctx, "lanes", 3 + i % 13) == MKZG_OK); break;
            case 1: CHECK(mkzg_ctx_set_option(ctx, "trim", 1) == MKZG_OK); break;
            case 2: CHECK(mkzg_ctx_set_option(ctx, "verify_shape", (i % 3) - 1) == MKZG_OK); break;
            case 3: CHECK(mkzg_ctx_set_option(ctx, "commit_direct_bits", (i & 1) ? 0 : -1) == MKZG_OK); break;
            case 4: { mkzg_kernel_time kt[32]; CHECK(mkzg_profile_read(ctx, kt, 32, 1) > 0); break; }
            case 5: CHECK(mkzg_ctx_set_option(ctx, "fk20_bits", (i % 2) ? 8 : 11) == MKZG_OK); break;
            default: { long v = 0; CHECK(mkzg_ctx_get_option(ctx, "tables_on_own_device", &v) == MKZG_OK && v == 1); break; }
        }
        std::this_thread::sleep_for(std::chrono::microseconds(300));
    }
}

static void run(const std::string& js, const std::vector<int>& devices, int threads, int rounds) {
    mkzg_ctx* ctx = nullptr;
    int rc = mkzg_ctx_new(js.data(), js.size(), devices.data(), (int)devices.size(), &ctx);
    CHECK(rc == MKZG_OK && ctx);
    if (!ctx) return;
    CHECK(mkzg_profile_enable(ctx, 1) == MKZG_OK);
    std::atomic<bool> stop{false};
    std::thread tn(tuner, ctx, &stop);
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++) th.emplace_back(caller, ctx, t, rounds, t % 2);      // half of the callers sit on the other device
    for (auto& t : th) t.join();
    stop = true;
    tn.join();
    long v = 0;
    CHECK(mkzg_ctx_get_option(ctx, "tables_on_own_device", &v) == MKZG_OK && v == 1);
    CHECK(mkzg_ctx_get_option(ctx, "devices", &v) == MKZG_OK && v == (long)devices.size());
    mkzg_ctx_free(ctx);
}

// The exchange step of the batch verifiers inside the library (api_exchange.hip) with `world` ranks = threads, each with a context on
// its own stub device, over the in-process stand-in for librccl (stub_rccl.cpp, named through MKZG_RCCL_LIB).  The kernels are stubs, so
// what is checked is the protocol: every rank returns the SAME code; shards that do not add up to the announced batch are refused on
// every rank; an input error found on the host in one rank's shard (a nil blob, a nil cell) is returned by every rank; a rank may
// bring nothing; nobody hangs; TSan stays silent.
static void exchange_scenario(const std::string& js, int world) {
    uint8_t id[128];
    CHECK(mkzg_comm_unique_id(id) == MKZG_OK);
    std::vector<int> got(8 * (size_t)world, -1);
    std::vector<std::thread> th;
    for (int rank = 0; rank < world; rank++) th.emplace_back([&, rank] {
        CHECK(hipSetDevice((rank + 1) % world) == hipSuccess);                  // the caller sits on another device than its context's
        mkzg_ctx* ctx = nullptr;
        const int dev = rank;
        CHECK(mkzg_ctx_new(js.data(), js.size(), &dev, 1, &ctx) == MKZG_OK && ctx);
        mkzg_comm* cm = nullptr;
        CHECK(mkzg_comm_init(ctx, id, rank, world, &cm) == MKZG_OK && cm);
        const size_t NB = 3 + (size_t)rank;                                     // uneven shards
        std::vector<std::vector<uint8_t>> blobs(NB, std::vector<uint8_t>(MKZG_BYTES_PER_BLOB, 0));
        std::vector<const uint8_t*> bp(NB);
        for (size_t b = 0; b < NB; b++) bp[b] = blobs[b].data();
        std::vector<uint8_t> cms(NB * 48, 0), prs(NB * 48, 0);
        for (size_t b = 0; b < NB; b++) { cms[48 * b] = 0xc0; prs[48 * b] = 0xc0; }
        uint64_t total = 0;
        for (int k = 0; k < world; k++) total += 3 + (uint64_t)k;
        int* g = &got[8 * (size_t)rank];
        g[0] = mkzg_verify_blob_kzg_proof_batch_exchange(cm, bp.data(), cms.data(), prs.data(), NB, total);
        g[1] = mkzg_verify_blob_kzg_proof_batch_exchange(cm, bp.data(), cms.data(), prs.data(), NB, total + 1);        // announced size is wrong
        if (rank == world - 1) bp[1] = nullptr;                                                                          // a nil blob in the LAST shard
        g[2] = mkzg_verify_blob_kzg_proof_batch_exchange(cm, bp.data(), cms.data(), prs.data(), NB, total);
        bp[1] = blobs[1].data();
        // cells: rank 0 brings nothing; rank 1 (if any) a nil cell
        const size_t NC = rank == 0 ? 0 : 8;
        std::vector<std::vector<uint8_t>> cellv(8, std::vector<uint8_t>(2048, 0));
        std::vector<const uint8_t*> cp(8);
        std::vector<uint64_t> idx(8);
        std::vector<uint8_t> ccm(8 * 48, 0), cpr(8 * 48, 0);
        for (int i = 0; i < 8; i++) { cp[i] = cellv[i].data(); idx[i] = (uint64_t)i; ccm[48 * i] = 0xc0; cpr[48 * i] = 0xc0; }
        g[3] = mkzg_verify_cell_kzg_proof_batch_exchange(cm, ccm.data(), idx.data(), cp.data(), cpr.data(), NC, 8 * (uint64_t)(world - 1));
        if (rank == 1) cp[5] = nullptr;
        g[4] = mkzg_verify_cell_kzg_proof_batch_exchange(cm, ccm.data(), idx.data(), cp.data(), cpr.data(), NC, 8 * (uint64_t)(world - 1));
        CHECK(mkzg_comm_last_exchange_us(cm) >= 0.0);
        cp[5] = cellv[5].data();
        // a rank-local ARGUMENT error (the last rank passes no proofs): it travels through the collectives instead of leaving the other
        // ranks waiting in them, and every rank returns it
        g[5] = mkzg_verify_cell_kzg_proof_batch_exchange(cm, ccm.data(), idx.data(), cp.data(), (rank == world - 1 && NC) ? nullptr : cpr.data(), NC, 8 * (uint64_t)(world - 1));
        g[6] = mkzg_verify_blob_kzg_proof_batch_exchange(cm, rank == world - 1 ? nullptr : bp.data(), cms.data(), prs.data(), NB, total);
        // the cells of this rank contiguous in "HBM" (stub device memory)
        void* d_cells = nullptr;
        CHECK(hipSetDevice(dev) == hipSuccess);
        CHECK(hipMalloc(&d_cells, 8 * 2048) == hipSuccess);
        CHECK(hipSetDevice((rank + 1) % world) == hipSuccess);
        g[7] = mkzg_verify_cell_kzg_proof_batch_exchange_device(cm, ccm.data(), idx.data(), d_cells, cpr.data(), NC, 8 * (uint64_t)(world - 1), nullptr);
        CHECK(hipFree(d_cells) == hipSuccess);
        int d = -1;
        CHECK(hipGetDevice(&d) == hipSuccess && d == (rank + 1) % world);
        mkzg_comm_free(cm);
        mkzg_ctx_free(ctx);
    });
    for (auto& t : th) t.join();
    for (int rank = 1; rank < world; rank++)
        for (int q = 0; q < 8; q++) CHECK(got[8 * (size_t)rank + q] == got[q]);           // every rank the same verdict
    CHECK(soft(got[0]));
    CHECK(got[1] == MKZG_ERR_BATCH_LENGTH);
    CHECK(got[2] == MKZG_ERR_NIL_INPUT);
    CHECK(soft(got[3]));
    CHECK(world < 2 || got[4] == MKZG_ERR_NIL_INPUT);
    CHECK(world < 2 ? soft(got[5]) : got[5] == MKZG_ERR_BAD_ARGS);
    CHECK(got[6] == MKZG_ERR_BAD_ARGS);
    CHECK(soft(got[7]));
}

// An eight-device context (api_multi.hip): one full context per device behind the front, host-pointer batches cut into contiguous
// shards.  Checked: the devices option, every shared table on its own device after calls that build them, the pools' sizes (eight
// contexts share the CPUs), per-item results at the right positions across the shards (a blob with too few cells in the recovery batch,
// nil inputs in the last shard of both verifiers).
static void eight_device_scenario(const std::string& js) {
    const int devs[8] = {0, 1, 2, 3, 4, 5, 6, 7};
    mkzg_ctx* ctx = nullptr;
    CHECK(mkzg_ctx_new(js.data(), js.size(), devs, 8, &ctx) == MKZG_OK && ctx);
    if (!ctx) return;
    long v = 0;
    CHECK(mkzg_ctx_get_option(ctx, "devices", &v) == MKZG_OK && v == 8);
    CHECK(mkzg_ctx_get_option(ctx, "host_threads", &v) == MKZG_OK && v >= 1 && v <= 16);
    const long share_threads = v;
    CHECK(mkzg_ctx_get_option(ctx, "cpu_share", &v) == MKZG_OK && v >= 2 && share_threads <= v);    // a device's pool never exceeds its share of the CPUs
    const size_t NB = 19;                                                       // 8 shards of 2 or 3
    std::vector<std::vector<uint8_t>> blobs(NB, std::vector<uint8_t>(MKZG_BYTES_PER_BLOB, 0));
    std::vector<const uint8_t*> bp(NB);
    for (size_t b = 0; b < NB; b++) { blobs[b][31] = (uint8_t)(b + 1); bp[b] = blobs[b].data(); }
    std::vector<uint8_t> cms(NB * 48, 0), prs(NB * 48, 0), out(NB * 48);
    for (size_t b = 0; b < NB; b++) { cms[48 * b] = 0xc0; prs[48 * b] = 0xc0; }
    std::vector<int> st(NB, -1);
    CHECK(soft(mkzg_blob_to_kzg_commitment_batch(ctx, bp.data(), NB, out.data(), st.data())));
    CHECK(soft(mkzg_verify_blob_kzg_proof_batch(ctx, bp.data(), cms.data(), prs.data(), NB, NB, NB)));
    bp[NB - 1] = nullptr;                                                       // last item of the last shard
    CHECK(mkzg_verify_blob_kzg_proof_batch(ctx, bp.data(), cms.data(), prs.data(), NB, NB, NB) == MKZG_ERR_NIL_INPUT);
    CHECK(mkzg_blob_to_kzg_commitment_batch(ctx, bp.data(), NB, out.data(), st.data()) == MKZG_ERR_NIL_INPUT);
    bp[NB - 1] = blobs[NB - 1].data();
    // cells
    const size_t NC = 8 * 16;
    std::vector<std::vector<uint8_t>> cellv(NC, std::vector<uint8_t>(2048, 0));
    std::vector<const uint8_t*> cp(NC);
    std::vector<uint64_t> idx(NC);
    std::vector<uint8_t> ccm(NC * 48, 0), cpr(NC * 48, 0);
    for (size_t i = 0; i < NC; i++) { cp[i] = cellv[i].data(); idx[i] = (uint64_t)(i % 128); ccm[48 * i] = 0xc0; cpr[48 * i] = 0xc0; }
    CHECK(soft(mkzg_verify_cell_kzg_proof_batch(ctx, ccm.data(), NC, idx.data(), NC, cp.data(), NC, cpr.data(), NC)));
    cp[NC - 3] = nullptr;
    CHECK(mkzg_verify_cell_kzg_proof_batch(ctx, ccm.data(), NC, idx.data(), NC, cp.data(), NC, cpr.data(), NC) == MKZG_ERR_NIL_INPUT);
    cp[NC - 3] = cellv[NC - 3].data();
    idx[NC - 2] = 128;                                                          // checked for the whole batch before anything else (api_eip7594.go:184-188)
    cp[1] = nullptr;
    CHECK(mkzg_verify_cell_kzg_proof_batch(ctx, ccm.data(), NC, idx.data(), NC, cp.data(), NC, cpr.data(), NC) == MKZG_ERR_INVALID_CELL_ID);
    idx[NC - 2] = (NC - 2) % 128;
    cp[1] = cellv[1].data();
    // recovery batch: 11 blobs over 8 devices, blob 9 with 63 cells only -> its status alone carries the error
    const size_t RB = 11;
    std::vector<uint64_t> ids(64);
    for (int i = 0; i < 64; i++) ids[(size_t)i] = (uint64_t)(2 * i);
    std::vector<const uint64_t*> pid(RB, ids.data());
    std::vector<const uint8_t* const*> pcl(RB, cp.data());
    std::vector<size_t> nid(RB, 64);
    nid[9] = 63;
    std::vector<uint8_t> oc(RB * 262144), op(RB * 6144);
    std::vector<int> rst(RB, -1);
    const int rrc = mkzg_recover_cells_and_kzg_proofs_batch(ctx, RB, pid.data(), nid.data(), pcl.data(), nid.data(), oc.data(), op.data(), rst.data());
    CHECK(rrc == MKZG_ERR_NOT_ENOUGH_CELLS);
    for (size_t b = 0; b < RB; b++) CHECK(b == 9 ? rst[b] == MKZG_ERR_NOT_ENOUGH_CELLS : soft(rst[b]) && rst[b] != MKZG_ERR_NOT_ENOUGH_CELLS);
    CHECK(mkzg_ctx_get_option(ctx, "tables_on_own_device", &v) == MKZG_OK && v == 1);
    mkzg_ctx_free(ctx);
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s trusted_setup.json\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    std::string js;
    char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) js.append(buf, n);
    fclose(f);
    const int before = stub_hip_live_allocations();
    run(js, {0}, 10, 45);              // one device, ten callers: lanes, hash rounds, lazy tables, options under load
    run(js, {1}, 4, 18);               // a context on the device the callers are NOT on
    run(js, {0, 1}, 6, 27);            // the multi-device front
    eight_device_scenario(js);         // eight devices behind one front: shard placement, pool sizes, results at the right positions
    if (getenv("MKZG_RCCL_LIB")) {     // the exchange step over the in-process stand-in for librccl
        exchange_scenario(js, 1);
        exchange_scenario(js, 2);
        exchange_scenario(js, 8);
    }
    CHECK(stub_hip_live_allocations() == before);       // everything a context allocated on the "device" came back
    printf("stub_device_driver: %d failed checks\n", g_bad.load());
    return g_bad.load() ? 1 : 0;
}
