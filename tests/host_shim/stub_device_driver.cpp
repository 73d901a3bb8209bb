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
        switch ((r + id) % 9) {
            case 0: CHECK(soft(mkzg_blob_to_kzg_commitment_batch(ctx, bp.data(), NB, out.data(), st.data()))); break;
            case 1: CHECK(soft(mkzg_verify_blob_kzg_proof_batch(ctx, bp.data(), cms.data(), prs.data(), NB, NB, NB))); break;
            case 2: CHECK(soft(mkzg_verify_blob_kzg_proof_batch_device(ctx, d_blobs, d_cm, d_pr, NB, s))); break;
            case 3: CHECK(soft(mkzg_compute_cells_and_kzg_proofs(ctx, bp[0], 0, cells.data(), proofs.data()))); break;
            case 4: CHECK(soft(mkzg_verify_cell_kzg_proof_batch(ctx, ccm.data(), NC, idx.data(), NC, cp.data(), NC, cpr.data(), NC))); break;
            case 5: CHECK(soft(mkzg_recover_cells_and_kzg_proofs(ctx, ids.data(), NC, cp.data(), NC, 0, cells.data(), proofs.data()))); break;
            case 6: CHECK(soft(mkzg_compute_blob_kzg_proof(ctx, bp[1], cms.data(), 0, out.data()))); break;
            case 7: CHECK(soft(mkzg_blob_to_kzg_commitment_batch_device(ctx, d_blobs, NB, d_out, d_st, s))); break;
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
        switch (i++ % 7) {
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
    CHECK(stub_hip_live_allocations() == before);       // everything a context allocated on the "device" came back
    printf("stub_device_driver: %d failed checks\n", g_bad.load());
    return g_bad.load() ? 1 : 0;
}
