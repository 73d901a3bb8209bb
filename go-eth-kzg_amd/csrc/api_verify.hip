// api_verify.hip -- C ABI entry points of the EIP-4844 proof paths (include/mi355kzg.h):
//   Context.VerifyBlobKZGProofBatch / VerifyBlobKZGProof / VerifyKZGProof   reference verify.go:12-143
//   Context.ComputeBlobKZGProof / ComputeKZGProof                           reference prove.go:46-111
// What runs where: every per-blob / per-point operation (SHA-256 challenge, canonical checks,
// barycentric evaluation, quotient, point decoding, subgroup checks, all MSM bucket work) is a
// HIP kernel.  The host does, per BATCH: draw r, combine 3 x 32 window sums by Horner, one
// fixed-base scalar multiplication, and the 2-pairing product check with prepared G2 lines
// (reference internal/kzg/kzg_verify.go:160-199).
// Every call works in one Lane (ctx.h): calls on a shared context overlap on the GPU, and the host finish of one call runs while
// the next call's challenge hash does (reference verify.go:152-169 gets that from goroutines).  A call that finds others in flight takes
// the throughput shape and shares hash launches with them (hash_round_join below).
#include <sys/random.h>
#include <chrono>
#include <functional>
#include <thread>
#include "ctx.h"
#include "host_validate_dev.hpp"
#include "api_multi.h"
#include "host_sha256.hpp"

// ---------------------------------------------------------------------------------------------
// workspaces
// ---------------------------------------------------------------------------------------------
static const size_t V_MISC_WORDS = 16 + 4 * VMSM_WINDOWS * 48;
// bytes of the verdict block up to and including `aux_items` of its aux region's 32-byte items (ctx.h Lane::d_v_block)
size_t verdict_block_bytes(size_t cap, size_t aux_items) { return 3 * cap * sizeof(int) + (V_MISC_WORDS + 16) * 4 + aux_items * 32; }
static const size_t HOST_PIECE = 2048;      // blobs staged at once by the host-pointer verifier (256 MiB pinned + device)

int ensure_verify_ws(Lane* L, size_t n) {
    if (n <= L->v_cap) return MKZG_OK;
    const size_t cap = std::max<size_t>(n, 64);
    L->v_cap = 0;
    int rc = dev_realloc(L->d_v_raw, 2 * cap * 48);
    if (!rc) rc = dev_realloc(L->d_v_aff, 2 * cap * 128);
    if (!rc) rc = dev_realloc(L->d_v_z, cap * 48);
    if (!rc) rc = dev_realloc(L->d_v_y, cap * 48);
    if (!rc) rc = dev_realloc(L->d_v_zrows, cap * blob_eval_scratch_words() * 4);
    if (!rc) rc = dev_realloc(L->d_v_s1, cap * 32);
    if (!rc) rc = dev_realloc(L->d_v_s3, cap * 32);
    if (!rc) {
        // the verdict block (ctx.h): what comes back in one copy first, device-only scratch behind; zero-filled -- k_vmsm_window keeps a
        // ticket per (job, window) in the spare words of its slice records and resets it itself
        const size_t head = verdict_block_bytes(cap, cap);
        rc = dev_realloc_zero(L->d_v_block, head + cap * sizeof(int) + VMSM_PART_WORDS * 4);
        if (!rc) {
            L->d_v_st = (int*)L->d_v_block;
            L->d_v_misc = (uint32_t*)(L->d_v_block + 3 * cap * sizeof(int));
            L->d_v_ry = L->d_v_misc + V_MISC_WORDS + 16;
            L->d_v_stx = (int*)(L->d_v_block + head);
            L->d_v_parts = (uint32_t*)(L->d_v_block + head + cap * sizeof(int));
        }
    }
    L->h_v_bytes = 0;
    if (!rc) rc = pinned_realloc(L->h_v, verdict_block_bytes(cap, cap));      // the same layout: statuses | misc | 64 B | aux
    if (rc) return rc;
    L->h_v_bytes = verdict_block_bytes(cap, cap);
    L->v_cap = cap;
    return MKZG_OK;
}
// n: scalars / statuses per call (z, y, status arrays); n_quot: polynomials whose 4096 quotient rows are held at once
int ensure_quot_ws(Lane* L, size_t n, size_t n_quot) {
    if (n_quot > L->quot_cap) {
        L->quot_cap = 0;
        int rc = dev_realloc(L->d_quot, n_quot * 4096 * 48);
        if (rc) return rc;
        L->quot_cap = n_quot;
    }
    if (n <= L->q_cap) return MKZG_OK;
    L->q_cap = 0;
    int rc = dev_realloc(L->d_q_z, n * 48);
    if (!rc) rc = dev_realloc(L->d_q_y, n * 48);
    if (!rc) rc = dev_realloc(L->d_q_st, 2 * n * sizeof(int));
    if (rc) return rc;
    L->q_cap = n;
    return MKZG_OK;
}

void verify_fold_clear(VerifyFold& f) {
    G1H::set_inf(f.A);
    G1H::set_inf(f.Cc);
    G1H::set_inf(f.D);
    memset(f.e, 0, 32);
}
void verify_fold_add(VerifyFold& acc, const VerifyFold& f) {
    G1H::add(acc.A, acc.A, f.A);
    G1H::add(acc.Cc, acc.Cc, f.Cc);
    G1H::add(acc.D, acc.D, f.D);
    fr_words_add(acc.e, acc.e, f.e);
}
// the random linear combination scalar (kzg_verify.go:136-141), 32 little-endian bytes straight from the OS CSPRNG
int draw_rlc_scalar(uint8_t r_le32[32]) { return getrandom(r_le32, 32, 0) == 32 ? MKZG_OK : MKZG_ERR_HIP; }

// ---------------------------------------------------------------------------------------------
// verification core.  Inputs already on the device:
//   d_commitments, d_proofs : n x 48 B compressed points
//   d_blobs (n x 131072 B): y is derived, and z too unless d_z32 brings the challenges (hashed on the host while the blobs uploaded);
//   without blobs d_z32 / d_y32 are the caller's scalars (n x 32 B big-endian).
// Error precedence mirrors the reference loops (verify.go:102-140, verify.go:15-33).
// Device work + Horner: A = sum r^i Q_i, Cc = sum r^i C_i, D = sum r^i z_i Q_i, e = sum r^i y_i with i = exp_offset + local index.
// n <= VMSM_MAX_N; larger batches are cut by the callers and folded piece by piece with one r.
// ---------------------------------------------------------------------------------------------
// Two shapes of the blob verifier, same results.  LATENCY (a caller waits for this one call): everything that does not need the
// challenge runs beside the 2.7 ms SHA-256 chain -- among it an inverse NTT of every blob, so that y = f(z) is a short Horner pass
// once z is known -- and the subgroup tests take a quad of lanes per point.  THROUGHPUT (other calls share the chip, the chain
// lengths hide behind them): y by the barycentric formula straight from the blob (domain.go:193-235 -- 2.1x fewer instructions than
// inverse NTT + Horner, and no 201 MB of coefficients written and read back), one lane per point in the subgroup tests (2.1x fewer
// again), and all of it on ONE stream.  By default a call takes the throughput shape when at least two other calls are in flight on the
// context.
// Hash rounds (throughput shape).  The challenge kernel is a 2.7 ms dependent chain per blob whatever the batch, and HIP runs all the
// streams of a process through a handful of in-order hardware queues: eight calls in flight, each with its own hash launch, held every
// queue for 3 ms at a time and a context verified ~0.65 M blobs/s however many callers it had.  So calls that find other calls in
// flight do not launch the hash themselves: they join the ROUND that is forming.  The first caller to find no round forming opens one
// and leads it: it waits until the round before the previous one has finished (on the host; whoever arrives meanwhile joins), closes
// the round and launches ONE kernel over all its jobs on one of the context's hash streams (rounds take them in turn: hash_in_flight() are in flight)
// -- streams of the lowest priority, so that no other stream of the process shares their queues.  Everyone then waits for the round on the host (the call is synchronous anyway; a device-side wait
// would hold the call's own queue, and with it whatever else HIP mapped there) and carries on in its own stream.
// (A round that also validated its calls' points in one launch on a stream of its own was measured in round 4 and dropped -- DESIGN.md
// Appendix A: the 1.3 ms chain runs in every call's own stream while the call waits for its round anyway.)
// rounds in flight: three since round 5's first session (two until then), four with deep rounds (below).  *Measured* (tools/ab_r05_11.sh, 32 callers, one box, alternating): 2 in flight
// 2.25-2.35 M blobs/s, 3: 2.40-2.45 M, 4: 2.35 M -- the rounds are what bounds the leg (rounds in flight x calls per round / 4 ms per
// round), and a third one fills the gap between a round's end and its stream's next launch.  MKZG_HASH_IN_FLIGHT overrides (A/B aid).
// FOUR with deep rounds (below: a round then holds its stream for the evaluation and the MSMs of its calls too; tools/ab_r05_27.sh, _28.sh:
// 2.19 M with three, 2.21-2.25 M with four where plain rounds give 1.95-2.08 M with either) -- one per hardware queue of the streams' priority level.
static int hash_in_flight() {
    static const int v = [] { const char* e = getenv("MKZG_HASH_IN_FLIGHT"); const int n = e ? atoi(e) : 0; return n >= 1 && n <= mkzg_ctx::HASH_STREAMS_MAX ? n : 4; }();
    return v;
}
// DEEP ROUNDS (round 5).  A context runs ONE kernel at a time per hardware queue and has four queues per priority level for all its
// streams (tools/ubench_streams.hip): behind its round a call used to put three more launches on its own stream --
// evaluation, window sums, the copy back -- each of which waited its turn behind other calls' kernels and then had the queue, and a
// fraction of the chip, to itself (192 wavefronts of window sums on 1024 SIMDs).  A call that can (device hash, r known at the call:
// everything but the exchange forms) hands the round a DeepJob instead, and the round's leader enqueues behind the hash, on the round's
// stream: ONE evaluation launch for all its calls (k_blob_eval_tree_jobs), a wait for every call's validation, ONE launch of all their
// MSMs (k_vmsm_window_calls), their verdict copies.  The call waits for the round's end on the host and goes straight to its host finish.
static bool deep_rounds_on(const mkzg_ctx* c) {
    static const int env = getenv("MKZG_DEEP_ROUNDS") ? (atoi(getenv("MKZG_DEEP_ROUNDS")) != 0 ? 1 : 0) : -1;      // A/B aid
    const bool on = env >= 0 ? env == 1 : c->opt_deep_rounds.load(std::memory_order_relaxed) != 0;
    return on && !c->profiling.load(std::memory_order_relaxed);          // (the profile's per-kernel events bracket a call's own launches)
}
static int deep_round_launch(mkzg_ctx* c, HashRound& r, hipStream_t hs) {
    int idx[SHA_MAX_JOBS], nd = 0;
    for (int j = 0; j < r.jobs.count; j++) if (r.is_deep[j]) idx[nd++] = j;
    for (int lo = 0; lo < nd; lo += DEEP_MAX_JOBS) {
        const int cnt = std::min(DEEP_MAX_JOBS, nd - lo);
        EvalJobs ev;
        VmsmCalls vm;
        ev.count = vm.count = cnt;
        for (int q = 0; q < cnt; q++) {
            const DeepJob& d = r.deep[idx[lo + q]];
            ev.blobs[q] = d.blobs; ev.zrows[q] = d.zrows; ev.y_out[q] = d.y; ev.status[q] = d.st_b;
            ev.s1[q] = d.s1; ev.s3[q] = d.s3; ev.ry[q] = d.ry;
            memcpy(ev.r[q], d.r, 32);
            ev.exp_offset[q] = d.exp_offset;
            ev.n[q] = d.n;
            vm.s1[q] = d.s1; vm.s3[q] = d.s3; vm.aff_q[q] = d.aff_q; vm.aff_c[q] = d.aff_c; vm.parts[q] = d.parts; vm.out[q] = d.wsum;
            vm.n[q] = d.n;
        }
        blob_eval_jobs_device(ev, c->d_roots_brp, hs);
        for (int q = 0; q < cnt; q++) HIPCHK(hipStreamWaitEvent(hs, r.deep[idx[lo + q]].validated, 0));
        if (vmsm_calls_device(vm, hs)) return MKZG_ERR_BAD_ARGS;
        HIPCHK(hipGetLastError());
    }
    for (int q = 0; q < nd; q++) {
        const DeepJob& d = r.deep[idx[q]];
        HIPCHK(hipMemcpyAsync(d.h_dst, d.d_src, d.copy_bytes, hipMemcpyDeviceToHost, hs));
    }
    return MKZG_OK;
}
static int hash_round_join(mkzg_ctx* c, const uint8_t* d_blobs, const uint8_t* d_commitments, int n, uint32_t* d_z, uint32_t* d_zrows, hipEvent_t ready, std::shared_ptr<HashRound>* out,
                           const DeepJob* deep = nullptr) {
    std::shared_ptr<HashRound> r, prev;
    bool leader = false;
    {
        std::unique_lock<std::mutex> lk(c->hash_mu);
        for (int q = 0; q < hash_in_flight(); q++) {
            hipStream_t& hs = c->hash_stream[q];
            if (hs) continue;
            int least = 0, greatest = 0;
            (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
#ifndef MKZG_HASH_PRIO
#define MKZG_HASH_PRIO 0
#endif
            const int prio = MKZG_HASH_PRIO == 2 ? greatest : MKZG_HASH_PRIO == 1 ? (least + greatest) / 2 : least;
            if (hipStreamCreateWithPriority(&hs, hipStreamNonBlocking, prio) != hipSuccess) { hs = nullptr; (void)hipGetLastError(); return MKZG_ERR_HIP; }
        }
        if (c->hash_forming && c->hash_forming->jobs.count >= SHA_MAX_JOBS) return MKZG_ERR_BAD_ARGS;     // cannot happen: one job per lane
        if (!c->hash_forming) {
            c->hash_forming = std::make_shared<HashRound>();
            prev = c->hash_hist[c->hash_rounds % (uint64_t)hash_in_flight()];      // the round that used the stream this one will take: hash_in_flight() rounds back
            leader = true;
        }
        r = c->hash_forming;
        const int j = r->jobs.count++;
        r->jobs.blobs[j] = d_blobs;
        r->jobs.commitments[j] = d_commitments;
        r->jobs.z_out[j] = d_z;
        r->jobs.zrows_out[j] = d_zrows;
        r->jobs.n[j] = n;
        r->ready[j] = ready;
        r->is_deep[j] = deep != nullptr;
        if (deep) r->deep[j] = *deep;
        if (leader && prev) c->hash_cv.wait(lk, [&] { return prev->launched; });
    }
    if (leader) {
        if (prev && prev->done) {                                            // everything the round that used this round's stream launched has finished: the stream is free
            // (a crowded context sleeps here too: hipEventSynchronize polls on this runtime, and one leader per round would keep a CPU spinning)
            if (wait_sleeping(c)) (void)wait_event_sleeping(prev->done);
            else (void)hipEventSynchronize(prev->done);
        }
        hipStream_t hs;
        {
            std::lock_guard<std::mutex> g(c->hash_mu);
            c->hash_forming.reset();                                           // closed: later arrivals open the next round
            const uint64_t slot = c->hash_rounds++ % (uint64_t)hash_in_flight();
            c->hash_hist[slot] = r;
            hs = c->hash_stream[slot];
        }
        int rc = MKZG_OK;
        bool any_deep = false, all_deep = true;
        for (int j = 0; j < r->jobs.count; j++) { any_deep = any_deep || r->is_deep[j]; all_deep = all_deep && r->is_deep[j]; }
        if (hipEventCreateWithFlags(&r->done, hipEventDisableTiming | hipEventBlockingSync) != hipSuccess) { r->done = nullptr; rc = MKZG_ERR_HIP; }
        // `hashed`: for the calls of a mixed round that carry on in their own streams (the exchange forms, whose r is still on its way)
        if (!rc && any_deep && !all_deep && hipEventCreateWithFlags(&r->hashed, hipEventDisableTiming | hipEventBlockingSync) != hipSuccess) { r->hashed = nullptr; rc = MKZG_ERR_HIP; }
        for (int j = 0; j < r->jobs.count && !rc; j++)
            if (r->ready[j] && hipStreamWaitEvent(hs, r->ready[j], 0) != hipSuccess) rc = MKZG_ERR_HIP;
        if (!rc) {
            challenge_jobs_device(r->jobs, hs);
            if (hipGetLastError() != hipSuccess) rc = MKZG_ERR_HIP;
            if (!rc && r->hashed && hipEventRecord(r->hashed, hs) != hipSuccess) rc = MKZG_ERR_HIP;
            if (!rc && any_deep) rc = deep_round_launch(c, *r, hs);
            if (hipEventRecord(r->done, hs) != hipSuccess && !rc) rc = MKZG_ERR_HIP;      // (recorded whatever happened: the stream's next round waits for it)
        }
        if (rc) (void)hipGetLastError();
        {
            std::lock_guard<std::mutex> g(c->hash_mu);
            r->rc = rc;
            r->launched = true;
        }
        c->hash_cv.notify_all();
    } else {
        std::unique_lock<std::mutex> lk(c->hash_mu);
        c->hash_cv.wait(lk, [&] { return r->launched; });
    }
    *out = r;
    return r->rc;
}
#ifndef MKZG_TP_VALIDATE_MODE
#define MKZG_TP_VALIDATE_MODE 3             // one lane per point (half the instructions of the quad form, a longer chain)
#endif
#ifndef MKZG_LAT_VALIDATE_MODE
#define MKZG_LAT_VALIDATE_MODE(n) 0         // the latency shape: a quad of lanes per point
#endif
static bool throughput_shape(const mkzg_ctx* c) {
    if (c->opt_verify_shape >= 0) return c->opt_verify_shape == 1;
    return c->lanes_busy.load(std::memory_order_relaxed) >= 3;
}
#ifdef MKZG_CPU_PROF
// diagnostic build (tools: build_variant.sh cpuprof "-DMKZG_CPU_PROF" api_verify.hip): thread CPU time of the caller by section of a verifier call
#include <time.h>
static std::atomic<long long> g_cpu_ns[8];
static std::atomic<long long> g_wall_ns[8];
static std::atomic<long long> g_cpu_calls{0};
struct CpuMark {
    long long c, w;
    static long long now(clockid_t id) { timespec t; clock_gettime(id, &t); return t.tv_sec * 1000000000ll + t.tv_nsec; }
    CpuMark() : c(now(CLOCK_THREAD_CPUTIME_ID)), w(now(CLOCK_MONOTONIC)) {}
    void lap(int k) { long long c2 = now(CLOCK_THREAD_CPUTIME_ID), w2 = now(CLOCK_MONOTONIC); g_cpu_ns[k] += c2 - c; g_wall_ns[k] += w2 - w; c = c2; w = w2; }
};
struct CpuProfPrinter { ~CpuProfPrinter() {
    long long n = g_cpu_calls.load(); if (!n) return;
    static const char* nm[8] = {"launch1", "hash_join", "round_wait", "launch2", "final_wait", "status+horner", "finish", "-"};
    for (int k = 0; k < 7; k++) fprintf(stderr, "cpuprof %-14s cpu %8.1f us  wall %8.1f us per call\n", nm[k], g_cpu_ns[k] / 1e3 / n, g_wall_ns[k] / 1e3 / n);
} } g_cpu_prof_printer;
#define CPU_LAP(k) cpu_mark.lap(k)
#else
#define CPU_LAP(k) ((void)0)
#endif
// the end of every fold: one round trip (statuses, e, window sums), the reference's error precedence, Horner over the window sums
// copy_enqueued: the caller has put the verdict block's copy on `s` already (inside its launch graph)
static int verify_fold_tail(mkzg_ctx* c, Lane* L, size_t n, bool with_blobs, int y_bad, hipStream_t s, VerifyFold* out, bool sum_ry_on_host = false, bool copy_enqueued = false) {
    const size_t cap = L->v_cap;
    uint32_t* h_misc = (uint32_t*)(L->h_v + 3 * cap * sizeof(int));
    HIPCHK(hipGetLastError());
    // 5. one round trip: statuses, e, window sums
    int* h_st = (int*)L->h_v;
    uint32_t* h_ry = h_misc + V_MISC_WORDS + 16;
    if (!copy_enqueued) HIPCHK(hipMemcpyAsync(h_st, L->d_v_block, verdict_block_bytes(cap, sum_ry_on_host ? n : 0), hipMemcpyDeviceToHost, s));      // statuses, misc (and the r^i y_i words): ONE copy
    LANE_SYNC(c, L, s);                     // polls for a lone caller, asks and sleeps when the context is crowded (ctx.h)
    L->drained = true;                      // the side streams were joined into s by the caller; nothing of this call is left on the device
    if (with_blobs) {
        for (size_t i = 0; i < n; i++) {                           // verify.go:102-123 order: commitment, proof, blob
            if (h_st[i]) return h_st[i];
            if (h_st[cap + i]) return h_st[cap + i];
            if (h_st[2 * cap + i]) return h_st[2 * cap + i];
        }
    } else {
        for (size_t i = 0; i < n; i++) {                           // verify.go:15-33 order: y, z, commitment, proof
            if (y_bad) return y_bad;
            if (h_st[2 * cap + i]) return h_st[2 * cap + i];
            if (h_st[i]) return h_st[i];
            if (h_st[cap + i]) return h_st[cap + i];
        }
    }
    // 6. Horner over the window sums (the 255 serial doublings stay on the host)
    auto t_h0 = std::chrono::steady_clock::now();
    const uint32_t* wsum = h_misc + 16;
    auto horner = [&](int m, G1HX& o) { g1h_horner_words(o, wsum + (size_t)m * VMSM_WINDOWS * 48, VMSM_WINDOWS, VMSM_WINDOW_BITS); };
    {   // three independent 255-doubling chains: one per host thread
        G1HX* dst[3] = {&out->D, &out->A, &out->Cc};
        const int job[3] = {2, 0, 1};
        c->pool->parallel(3, [&](int i) { horner(job[i], *dst[i]); });
    }
    if (sum_ry_on_host) {
        memset(out->e, 0, 32);
        for (size_t i = 0; i < n; i++) fr_words_add(out->e, out->e, h_ry + 8 * i);
    } else {
        memcpy(out->e, h_misc + 8, 32);
    }
    L->last_host_finish_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_h0).count();
    return MKZG_OK;
}
// The points of the single-proof entry points (VerifyKZGProof, VerifyBlobKZGProof: verify.go:12-83) and of batches of up to G1_ROWS_MAX
// items -- a block's blobs, what a consensus client hands VerifyBlobKZGProofBatch (verify.go:88-143) -- validated on the HOST
// (host_validate.hpp: ~45 us per point on one core where a quad of GPU lanes takes 0.64 ms for the same dependent chain, however few the
// points), one task per point on the pool and this thread; one small launch puts the records and status words where the validation
// kernel leaves them.
static void host_validate_points(mkzg_ctx* c, const uint8_t* c48n, const uint8_t* q48n, size_t n, uint32_t* aff_c, int* st_c, uint32_t* aff_q, int* st_q, hipStream_t s2) {
    host_validate_arrays(c->pool, c48n, n, q48n, n, aff_c, st_c, aff_q, st_q, s2);
}
static bool host_validate_on(const mkzg_ctx* c, size_t n) {
    return c->opt_host_validate.load(std::memory_order_relaxed) != 0 && host_validate_pays(2 * n, c->pool ? c->pool->size() : 0, HOST_VALIDATE_MAX_POINTS);
}
static int verify_fold(mkzg_ctx* c, Lane* L, const uint8_t* d_blobs, const uint8_t* d_z32, const uint8_t* d_y32, const uint8_t* d_commitments, const uint8_t* d_proofs, size_t n,
                       uint32_t exp_offset, const uint8_t* r_le32, hipStream_t s, VerifyFold* out, RlcLate* late = nullptr,
                       const uint8_t* h_c48 = nullptr, const uint8_t* h_q48 = nullptr /* the points' bytes on the host (n x 48 each), for host_validate_points */) {
    if (n == 0 || n > (size_t)VMSM_MAX_N) return MKZG_ERR_BAD_ARGS;
    int rc = ensure_verify_ws(L, n);
    if (rc) return rc;
    const bool tp = throughput_shape(c);
    // Was the caller's stream idle when the call arrived?  Then nothing that produces this call's inputs can still be pending on it, and
    // the hash round need not wait for this call's marker -- which sits in a hardware queue the stream SHARES with other calls' streams,
    // behind their kernels: *measured* with 32 callers the round's launch-to-finish time was 8.5 ms for a 2.6-4.7 ms kernel.
    // (decided below, where the side stream is known: lane_fork)
#ifdef MKZG_CPU_PROF
    CpuMark cpu_mark;
    g_cpu_calls++;
#endif
    const size_t cap = L->v_cap;
    uint32_t* aff_c = L->d_v_aff;
    uint32_t* aff_q = L->d_v_aff + cap * 32;
    int* st_c = L->d_v_st;
    int* st_q = L->d_v_st + cap;
    int* st_b = L->d_v_st + 2 * cap;
    int* st_x = L->d_v_stx;                 // scratch: status of a second scalar array
    // 0. everything that needs r but not the challenges runs beside the challenge hash
    uint32_t r_words[8];                        // r travels in the kernel arguments (rwords.h), not through a 32-byte upload
    if (!late) memcpy(r_words, r_le32, 32);
    // r and the exponents' offset out of a collective still in flight (RlcLate, ctx.h): asked for where they are first used
    auto need_r = [&]() -> int {
        if (!late) return MKZG_OK;
        const int lrc = late->get();
        if (lrc) return lrc;
        if (late->offset + exp_offset + n > 0xffffffffull) return MKZG_ERR_BAD_ARGS;
        memcpy(r_words, late->r_words, 32);
        exp_offset += (uint32_t)late->offset;
        late = nullptr;
        return MKZG_OK;
    };
    uint32_t* d_parts = L->d_v_parts;
    uint32_t* d_wsum = L->d_v_misc + 16;
    const size_t parts_per_job = (size_t)VMSM_WINDOWS * VMSM_KMAX * 64, wsum_per_job = (size_t)VMSM_WINDOWS * 48;
    int vrc = 0;
    bool deep = false;                          // the call's round launches what follows the hash (deep rounds)
    bool tp_ry = false;                         // the call's sum r^i y_i is added up on the host from L->d_v_ry (throughput shape)
    bool tp_copy_enqueued = false;              // ... and the verdict block's copy back is on the stream already (throughput shape: part of its launch graph)
    // 1. decode + validate the points (serialization.go:108-115)
    //    on the side stream, beside the (longer, latency-bound) SHA-256 chain
    //    (throughput shape: on the call's own stream, ahead of the hash -- HIP multiplexes all streams of the process onto a few hardware
    //    queues, in order within each, so a side stream's short kernels would wait behind some other call's 3 ms hash kernel and a
    //    context with eight calls in flight had two of them progressing; one stream per call keeps one queue per call)
    hipStream_t s2 = tp ? s : lane_side_stream(L, s);
    const bool s_idle = lane_fork(L, s, s2);
    if (h_c48 && h_q48 && !tp && !late && host_validate_on(c, n)) host_validate_points(c, h_c48, h_q48, n, aff_c, st_c, aff_q, st_q, s2);
    else KT(KT_SUBGROUP, s2, g1_validate2_device(d_commitments, (int)n, aff_c, st_c, d_proofs, (int)n, aff_q, st_q, s2, tp ? MKZG_TP_VALIDATE_MODE : MKZG_LAT_VALIDATE_MODE(n)));
    if (d_blobs) {
        //    ... and, for blobs, the powers of r and the two MSMs that use nothing else (sum r^i Q_i, sum r^i C_i: kzg_verify.go:150,225)
        //    (throughput shape: none of that -- the three MSMs go out as ONE launch after the challenge, see below)
        //    (challenges hashed on the host: there is no hash to run beside -- the points' chains on the side stream, the evaluation on the
        //    main one, and the three MSMs as one launch after both)
        //    The blobs themselves wait for their challenge: y = f(z) is ONE pass over the blob once z is known (k_blob_eval_tree, with the
        //    canonicity check of DeserializeBlob, serialization.go:134-146) -- up to round 4 the latency shape ran an inverse NTT of every blob
        //    beside the hash (5x the instructions, 201 MB of coefficient rows per lane) to make that pass a Horner evaluation of the same length.
    }
    const bool early_msms = d_blobs && !tp && !d_z32;
    auto step0 = [&]() -> int {
        KT(KT_RLC, s2, rlc_scalars_device(r_words, nullptr, nullptr, (int)n, exp_offset, L->d_v_s1, nullptr, nullptr, 1, s2));
        const uint32_t* scal[2] = {L->d_v_s1, L->d_v_s1};
        const uint32_t* pts[2] = {aff_q, aff_c};
        const int njob[2] = {(int)n, (int)n};
        KT(KT_VMSM, s2, vrc = vmsm_windows_device(scal, pts, njob, 2, d_parts, d_wsum, s2));
        return vrc ? MKZG_ERR_BAD_ARGS : MKZG_OK;
    };
    const bool step0_late = early_msms && late != nullptr;    // r still on its way: the hash goes out first, these behind it (beside it on the device all the same)
    if (early_msms && !step0_late) { rc = step0(); if (rc) return rc; }
    if (!step0_late) HIPCHK(hipEventRecord(L->ev_join, s2));
    // 2. challenge and claimed value per blob, or caller-supplied z / y
    int y_bad = 0;
    if (d_blobs) {
        if (d_z32) {
            scalars_to_fr_device(d_z32, (int)n, L->d_v_z, st_x, s);          // challenges hashed on the host (canonical by construction)
        } else if (tp) {
            std::shared_ptr<HashRound> round;                                // the hash of this call and of whoever else is waiting for one
            CPU_LAP(0);
            DeepJob dj;
            deep = !late && deep_rounds_on(c);                               // (r is known: not an exchange form still waiting for its collective)
            if (deep) {
                dj.validated = L->ev_join;                                   // recorded above, behind the validation on this call's stream
                dj.blobs = d_blobs; dj.zrows = L->d_v_zrows; dj.y = L->d_v_y; dj.st_b = st_b;
                dj.s1 = L->d_v_s1; dj.s3 = L->d_v_s3; dj.ry = L->d_v_ry;
                memcpy(dj.r, r_words, 32);
                dj.exp_offset = exp_offset;
                dj.n = (int)n;
                dj.aff_q = aff_q; dj.aff_c = aff_c; dj.parts = d_parts; dj.wsum = d_wsum;
                dj.h_dst = L->h_v; dj.d_src = L->d_v_block; dj.copy_bytes = verdict_block_bytes(cap, n);
            }
            // (the round's launch also leaves what the evaluation needs of each z alone: L->d_v_zrows)
            rc = hash_round_join(c, d_blobs, d_commitments, (int)n, L->d_v_z, L->d_v_zrows, s_idle ? nullptr : L->ev_fork, &round, deep ? &dj : nullptr);   // ev_fork: recorded on s before any of this call's kernels
            CPU_LAP(1);
            hipEvent_t wait_for = round ? ((!deep && round->hashed) ? round->hashed : round->done) : nullptr;
            if (wait_for) {                                                  // on the host: a device-side wait would hold the call's queue
                int wr = MKZG_OK;
                if (wait_sleeping(c)) wr = wait_event_sleeping(wait_for);
                else if (hipEventSynchronize(wait_for) != hipSuccess) { (void)hipGetLastError(); wr = MKZG_ERR_HIP; }
                if (wr) {
                    // the round's kernels may still be writing this lane's challenges and rows: nobody gets the lane before the rounds' streams are idle
                    tl_hip_error = true;
                    for (auto& hs : c->hash_stream) if (hs) (void)hipStreamSynchronize(hs);
                    (void)hipGetLastError();
                    return wr;
                }
            }
            CPU_LAP(2);
            if (rc) return rc;
        } else {
            if (c->opt_debug_serial.load(std::memory_order_relaxed)) HIPCHK(hipStreamSynchronize(s2));
            KT(KT_SHA256, s, challenge_device(d_blobs, d_commitments, (int)n, L->d_v_z, s, L->d_v_zrows));     // (leaves the evaluation's z-only rows too)
        }
        if ((rc = need_r())) return rc;                                      // (a late r: the hash, the decoding and the subgroup tests are running by now)
        if (step0_late) {
            if ((rc = step0())) return rc;
            HIPCHK(hipEventRecord(L->ev_join, s2));
        }
        HIPCHK(hipStreamWaitEvent(s, L->ev_join, 0));
        if (tp && deep) {
            // the round has run this call's evaluation and MSMs and copied its verdict block back (deep rounds, above)
            tp_ry = true;
            tp_copy_enqueued = true;
        } else if (tp) {
            // throughput shape: every launch is time on the call's hardware queue, which it shares with other calls' streams -- the
            // evaluation's z-only rows came with the round's hash launch, the evaluation kernel's last lane leaves the blob's r^i, r^i z_i and
            // r^i y_i (the host adds the latter up), and ONE launch takes the three MSMs of kzg_verify.go:150,225,179
            const bool rows_ready = !d_z32;
            // (everything from here to the copy back is one fixed sequence on the call's stream: a launch graph, ctx.h lane_graph_run)
            const uint32_t* scal[3] = {L->d_v_s1, L->d_v_s1, L->d_v_s3};
            const uint32_t* pts[3] = {aff_q, aff_c, aff_q};
            const int njob[3] = {(int)n, (int)n, (int)n};
            rc = lane_graph_run(c, L, 0, s, [&]() -> int {
                KT(KT_EVAL_TREE, s, blob_eval_device(d_blobs, L->d_v_z, c->d_roots_brp, c->d_inv_roots_brp, (int)n, L->d_v_y, st_b, L->d_v_zrows, s, rows_ready, r_words, exp_offset, L->d_v_s1,
                                                     L->d_v_s3, L->d_v_ry));
                int v = 0;
                KT(KT_VMSM, s, v = vmsm_windows_device(scal, pts, njob, 3, d_parts, d_wsum, s));
                if (v) return MKZG_ERR_BAD_ARGS;
                HIPCHK(hipMemcpyAsync(L->h_v, L->d_v_block, verdict_block_bytes(cap, n), hipMemcpyDeviceToHost, s));
                return MKZG_OK;
            });
            if (rc) return rc;
            tp_ry = true;
            tp_copy_enqueued = true;
        } else if (d_z32) {
            // challenges from the host: the rows of z by a small kernel ahead, then as above
            KT(KT_EVAL_TREE, s, blob_eval_device(d_blobs, L->d_v_z, c->d_roots_brp, c->d_inv_roots_brp, (int)n, L->d_v_y, st_b, L->d_v_zrows, s, false, r_words, exp_offset, L->d_v_s1, L->d_v_s3,
                                                 L->d_v_ry));
            tp_ry = true;
            const uint32_t* scal[3] = {L->d_v_s1, L->d_v_s1, L->d_v_s3};
            const uint32_t* pts[3] = {aff_q, aff_c, aff_q};
            const int njob[3] = {(int)n, (int)n, (int)n};
            KT(KT_VMSM, s, vrc = vmsm_windows_device(scal, pts, njob, 3, d_parts, d_wsum, s, 512, 1));       // 3 jobs x 32 windows x 2 slices at 1024 proofs: still a quad per bucket
        } else {
            // 3. what depends on the challenges: y = f(z) with r^i z_i and r^i y_i from the lane that finishes each blob (the host adds the
            //    latter up), and the MSM sum r^i z_i Q_i (kzg_verify.go:179)
            KT(KT_EVAL_TREE, s, blob_eval_device(d_blobs, L->d_v_z, c->d_roots_brp, c->d_inv_roots_brp, (int)n, L->d_v_y, st_b, L->d_v_zrows, s, true, r_words, exp_offset, L->d_v_s1, L->d_v_s3,
                                                 L->d_v_ry));
            tp_ry = true;
            const uint32_t* scal[1] = {L->d_v_s3};
            const uint32_t* pts[1] = {aff_q};
            const int njob[1] = {(int)n};
            // the one MSM the caller waits for after the hash: four slices per window, a quad of lanes per bucket (0.52 -> 0.32 ms at 1024 proofs)
            KT(KT_VMSM, s, vrc = vmsm_windows_device(scal, pts, njob, 1, d_parts + 2 * parts_per_job, d_wsum + 2 * wsum_per_job, s, 256, 1));
        }
    } else {
        // VerifyKZGProof checks y before z (verify.go:15-23); st_b ends up non-zero if either fails
        scalars_to_fr_device(d_z32, (int)n, L->d_v_z, st_b, s);
        scalars_to_fr_device(d_y32, (int)n, L->d_v_y, st_x, s);
        HIPCHK(hipStreamWaitEvent(s, L->ev_join, 0));
        {
            std::vector<int> sty(n);
            HIPCHK(hipMemcpyAsync(sty.data(), st_x, n * sizeof(int), hipMemcpyDeviceToHost, s));
            LANE_SYNC(c, L, s);
            for (int v : sty) if (v) y_bad = v;
        }
        KT(KT_RLC, s, rlc_scalars_device(r_words, L->d_v_z, L->d_v_y, (int)n, exp_offset, L->d_v_s1, L->d_v_s3, L->d_v_misc + 8, 0, s));
        // the three MSMs of kzg_verify.go:150,225,179: sum r^i Q_i, sum r^i C_i, sum r^i z_i Q_i
        const uint32_t* scal[3] = {L->d_v_s1, L->d_v_s1, L->d_v_s3};
        const uint32_t* pts[3] = {aff_q, aff_c, aff_q};
        const int njob[3] = {(int)n, (int)n, (int)n};
        KT(KT_VMSM, s, vrc = vmsm_windows_device(scal, pts, njob, 3, d_parts, d_wsum, s));
    }
    if (vrc) return MKZG_ERR_BAD_ARGS;
    CPU_LAP(3);
    return verify_fold_tail(c, L, n, d_blobs != nullptr, y_bad, s, out, tp_ry, tp_copy_enqueued);
}
// kzg_verify.go:160-199: F = Cc - [e]G + D ; e(F, G2) e(-A, [alpha]G2) == 1
int sd_verify_finish(mkzg_ctx* c, const VerifyFold& f) {
    G1HX eG, Fsum, negA;
    uint64_t e64[4];
    for (int i = 0; i < 4; i++) e64[i] = (uint64_t)f.e[2 * i] | ((uint64_t)f.e[2 * i + 1] << 32);
    g1h_fixed_mul(eG, c->gen_fixed, e64);
    G1H::neg(eG, eG);
    G1H::add(Fsum, f.Cc, eG);
    G1H::add(Fsum, Fsum, f.D);
    G1H::neg(negA, f.A);
    G1AffH ps[2];
    g1h_to_affine(ps[0], Fsum);
    g1h_to_affine(ps[1], negA);
    const G2Prepared* qs[2] = {&c->g2_gen, &c->g2_alpha};
    return pairing_check_prepared_2mt(ps, qs, c->pool, miller_segments(c)) ? MKZG_OK : MKZG_VERIFY_FAILED;
}
static int timed_finish(mkzg_ctx* c, Lane* L, const VerifyFold& f) {
    auto t0 = std::chrono::steady_clock::now();
    const int rc = sd_verify_finish(c, f);
    L->last_host_finish_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    c->last_host_finish_us.store(L->last_host_finish_us);
    return rc;
}
// all of a device-resident batch: pieces of VMSM_MAX_N items folded with ONE r and the global exponents, one pairing check
static int verify_device_all(mkzg_ctx* c, Lane* L, const uint8_t* d_blobs, const uint8_t* d_commitments, const uint8_t* d_proofs, size_t n, hipStream_t s) {
    if (n == 0) return MKZG_OK;                                   // kzg_verify.go:121-123
    uint8_t r[32];
    if (draw_rlc_scalar(r)) return MKZG_ERR_HIP;
    L->last_host_finish_us = 0;
    VerifyFold total, f;
    verify_fold_clear(total);
    for (size_t off = 0; off < n; off += VMSM_MAX_N) {
        const size_t m = std::min<size_t>(VMSM_MAX_N, n - off);
        int rc = verify_fold(c, L, d_blobs + off * MKZG_BYTES_PER_BLOB, nullptr, nullptr, d_commitments + off * 48, d_proofs + off * 48, m, (uint32_t)off, r, s, &f);
        if (rc) return rc;
        verify_fold_add(total, f);
    }
    return timed_finish(c, L, total);
}

// ---------------------------------------------------------------------------------------------
// host-pointer batches.  Go hands over []*Blob: scattered, pageable 128 KiB arrays.  They are gathered into the lane's pinned staging
// area by the host pool in chunks, every chunk's copy to the device in flight while the next one is gathered; with host_sha the same
// threads hash each transcript right after copying it (the blob is then still in their cache) and the device skips its own hash.
// A nil blob stands in as the zero blob and is reported with the reference's precedence afterwards.
// ---------------------------------------------------------------------------------------------
// Where the challenges of a host-pointer call are hashed.  The device hash is a latency chain of 2050 compressions per blob -- 2.7 ms
// whether the call brings 1 blob or 1024, and only after the upload; a host core with the SHA extensions hashes a PAIR of transcripts in
// 71 us (two chains interleaved in one instruction stream, host_sha256.hpp: 35 us each against 53 us one at a time) and SIXTEEN on the
// lanes of AVX-512 registers in 380 us (24 us each), beside the upload.
// *Measured* on the 16-CPU boxes (tools/ubench_h2d.cpp): 1024 transcripts on the pool's 15 threads in 2.7 ms; gathered with streaming
// stores and uploaded chunk by chunk at the same time, 134 MB are on the device after 2.8-2.9 ms (3.8-4.0 with memcpy: its
// write-allocate reads compete with the DMA engine for the memory controllers).  So a LONE call hashes on the host (pipelined: below);
// calls that share the context and its pool keep the host for the gather and hash on the device (the throughput shape's shared
// rounds), as do calls beyond a few dozen blobs per pool thread.
static bool use_host_sha(const mkzg_ctx* c, size_t m) {
    if (c->opt_host_sha >= 0) return c->opt_host_sha != 0;
    const int busy = c->lanes_busy.load(std::memory_order_relaxed);
    const size_t per_thread = busy <= 1 ? 160 : (busy <= 4 ? 20 : 8);
    return m <= per_thread * (size_t)std::max(1, c->pool->size());
}
// Of a lone call's m transcripts, how many the DEVICE hashes (the first ones: they are uploaded first).  The host finishes its share at
// about m x t_copy + (m - d) x t_hash, the device at t_launch + d x t_upload + 2.75 ms; d balances the two (nothing up to ~1100 blobs on 15
// pool threads: *measured* at 1024 blobs 4.41 ms with none, 4.41 with 12 %, 4.74 with 25 %; 600 of a 2048-blob piece).  MKZG_DEV_HASH_PERCENT overrides (A/B).
// the host's share takes the sixteen-lane AVX-512 form when it gives every pool thread and the caller at least one task of sixteen
static bool host_sha_wide(const mkzg_ctx* c, size_t host_share) {
    static const bool no_wide = getenv("MKZG_NO_WIDE_SHA") != nullptr;          // A/B aid
    return !no_wide && sha256_have_avx512() && host_share >= 16 * (size_t)(c->pool->size() + 1);
}
static size_t device_hash_share(const mkzg_ctx* c, size_t m) {
    static const int forced = [] { const char* e = getenv("MKZG_DEV_HASH_PERCENT"); return e ? atoi(e) : -1; }();
    if (forced >= 0) return std::min(m, m * (size_t)std::min(forced, 100) / 100);
    const double T = (double)std::max(1, c->pool->size());
    // per transcript over the whole pool (us): 35.3 per core for a SHA-NI pair's half, 23.7 on sixteen AVX-512 lanes -- priced by the form the
    // host will really take (host_sha_wide; the device's share only shrinks the host's, and a share that large is far above the
    // threshold); the copies mostly hide under the upload; a chunk goes up at ~47 GB/s
    const double hash_us = (host_sha_wide(c, m) ? 23.7 : 35.3) / T, copy_us = std::max(0.6, 6.0 / T), up_us = 2.8, chain_us = 2750.0 + 200.0;
    const double d = ((copy_us + hash_us) * (double)m - chain_us) / (hash_us + up_us);
    if (d < 64) return 0;
    return std::min(m, (size_t)d / 64 * 64);
}

// Host-pointer blobs of one piece, PIPELINED (a lone call: nobody else's kernels fill the gaps, so every stage runs under the upload).
//   host pool : copy tasks of all chunks first (streaming stores into the pinned staging area, chunk order), then hash tasks -- pairs of
//               transcripts, read from the caller's blobs again (they come from the last-level cache or DRAM at 3.7 GB/s per thread) --
//               for the blobs the device does not hash;
//   copy stream : chunk k goes up the moment its copy tasks have signed off (a first small chunk starts the link early);
//   side stream : commitments + proofs (uploaded first), decoding + r-torsion tests, the powers of r, the two MSMs that need nothing
//               else (sum r^i Q_i, sum r^i C_i), and the inverse NTT of each chunk behind its upload when the main stream is taken;
//   main stream : the device's share of the challenge hash (the first blobs: launched as soon as they are up), else the chunks' inverse
//               NTTs; after the last challenge: evaluation (Horner from the coefficients), r^i z_i, the one MSM that needs the
//               challenges, one round trip.
// *Measured* (bench.py -> drop_in.verify_blob_kzg_proof_batch, N = 1024): 7.6-7.8 ms per call before (gather, upload, THEN the device
// pipeline) -> see DESIGN.md section 4.2 for the current figure.
static int fold_host_pipelined(mkzg_ctx* c, Lane* L, const uint8_t* const* blobs, const uint8_t* commitments48n, const uint8_t* proofs48n, size_t m, uint32_t exp_offset,
                               const uint8_t* r_le32, VerifyFold* out, RlcLate* late = nullptr) {
    hipStream_t s = L->stream, s2 = L->side, cp = L->side2;
    int rc = MKZG_OK;
    const size_t cap = L->v_cap;
    uint32_t* aff_c = L->d_v_aff;
    uint32_t* aff_q = L->d_v_aff + cap * 32;
    int* st_c = L->d_v_st;
    int* st_q = L->d_v_st + cap;
    int* st_b = L->d_v_st + 2 * cap;
    int* st_x = L->d_v_stx;
    uint32_t r_words[8];
    if (!late) memcpy(r_words, r_le32, 32);
    uint32_t* d_parts = L->d_v_parts;
    uint32_t* d_wsum = L->d_v_misc + 16;
    const size_t parts_per_job = (size_t)VMSM_WINDOWS * VMSM_KMAX * 64, wsum_per_job = (size_t)VMSM_WINDOWS * 48;
    uint8_t* h_z = L->h_stage + m * MKZG_BYTES_PER_BLOB;              // m x 32 B
    uint8_t* h_pts = h_z + m * 32;                                     // m x 48 B commitments | m x 48 B proofs
    uint8_t* d_z = L->d_blobs + m * MKZG_BYTES_PER_BLOB;
    const size_t n_dev = device_hash_share(c, m);                      // transcripts [0, n_dev) are hashed by the device
    // below 256 blobs the three MSMs leave as one launch at the end -- and so they do when r is still on its way (RlcLate: nothing here
    // needs it before the evaluation, and this thread must not stop feeding the link to wait for another rank)
    const bool early_msm = m >= 256 && !late;
    // chunk plan: a small first chunk starts the link early, then chunks of 128
    struct Chunk { size_t lo, m; };
    std::vector<Chunk> plan;
    const size_t CH = 128;
    for (size_t lo = 0; lo < m;) {
        size_t k = std::min(m - lo, plan.empty() && m > 64 ? (size_t)32 : CH);
        if (m - lo - k < 16 && m - lo <= CH + 16) k = m - lo;
        plan.push_back({lo, k});
        lo += k;
    }
    const size_t nch = plan.size();
    // copy tasks: 8 blobs each; hash tasks: 4 transcripts (two SHA-NI pairs) each -- or 16, one per AVX-512 lane, when the host's share
    // gives every pool thread (and this one) at least one such task
    const bool wide_sha = host_sha_wide(c, m - n_dev);
    const size_t CPT = 8, HPT = wide_sha ? 16 : 4;
    std::vector<size_t> ctask_chunk, ctask_lo;
    std::vector<Countdown> left(nch);
    for (size_t k = 0; k < nch; k++) {
        int cnt = 0;
        for (size_t lo = plan[k].lo; lo < plan[k].lo + plan[k].m; lo += CPT, cnt++) { ctask_chunk.push_back(k); ctask_lo.push_back(lo); }
        left[k].set(cnt);
    }
    const int n_copy = (int)ctask_lo.size();
    const int n_hash = (int)((m - n_dev + HPT - 1) / HPT);
    Countdown copy_left(n_copy + n_hash) /* every task of the call */, hash_left(n_hash);
    const std::function<void(int)> copy_task = [&](int t) {
        const size_t k = ctask_chunk[(size_t)t], lo = ctask_lo[(size_t)t], hi = std::min(plan[k].lo + plan[k].m, lo + CPT);
        for (size_t i = lo; i < hi; i++) {
            uint8_t* dst = L->h_stage + i * MKZG_BYTES_PER_BLOB;
            if (blobs[i]) copy_to_staging(dst, blobs[i], MKZG_BYTES_PER_BLOB);
            else memset(dst, 0, MKZG_BYTES_PER_BLOB);
        }
        left[k].arrive();
    };
    const std::function<void(int)> hash_task = [&](int t) {
        const size_t lo = n_dev + (size_t)t * HPT, hi = std::min(m, lo + HPT);
        // a nil blob stands in as the zero blob: the call fails with the reference's precedence afterwards (not its staged copy: a copy
        // task may still be writing that one)
        static const uint8_t zero_blob[MKZG_BYTES_PER_BLOB] = {0};
        auto src = [&](size_t i) { return blobs[i] ? blobs[i] : zero_blob; };
        size_t i = lo;
#if defined(__x86_64__)
        if (wide_sha && hi - lo == 16) {
            const uint8_t* bp[16];
            const uint8_t* cp16[16];
            for (int q = 0; q < 16; q++) { bp[q] = src(lo + (size_t)q); cp16[q] = commitments48n + (lo + (size_t)q) * 48; }
            host_compute_challenge_x16(h_z + lo * 32, bp, cp16);
            return;
        }
#endif
        for (; i + 1 < hi; i += 2) host_compute_challenge_x2(h_z + i * 32, src(i), commitments48n + i * 48, h_z + (i + 1) * 32, src(i + 1), commitments48n + (i + 1) * 48);
        if (i < hi) host_compute_challenge(h_z + i * 32, src(i), commitments48n + i * 48);
    };
    // points first: their chains are the longest thing on the side stream
    memcpy(h_pts, commitments48n, m * 48);
    memcpy(h_pts + m * 48, proofs48n, m * 48);
    // ONE queue, interleaved: the copies of chunk k + 1 ahead of the hashes of chunk k's blobs.  The link takes a chunk every 0.3 ms and the
    // pool gathers one in 0.1: with all copies first (the first version of this pipeline) the uploads were fed 2.5x faster than they could
    // leave while the hashes -- the longer half of the host's work, and what the call's tail waits for -- started 1.2 ms late; interleaved,
    // copies and hashes both end with the uploads (trace: challenges complete 3.4-3.6 -> see DESIGN.md section 4.2).
    std::vector<int> order;                                     // >= 0: copy task; < 0: ~(hash task)
    order.reserve((size_t)n_copy + (size_t)n_hash);
    {
        size_t hnext = 0, cnext = 0;
        auto push_hash_upto = [&](size_t blob_end) { while (hnext < (size_t)n_hash && n_dev + hnext * HPT < blob_end) order.push_back(~(int)hnext++); };
        for (size_t k = 0; k < nch; k++) {
            while (cnext < ctask_chunk.size() && ctask_chunk[cnext] == k) order.push_back((int)cnext++);
            if (k >= 1) push_hash_upto(plan[k - 1].lo + plan[k - 1].m);
        }
        push_hash_upto(m);
    }
    const std::function<void(int)> mixed = [&](int t) {
        const int code = order[(size_t)t];
        if (code >= 0) copy_task(code);
        else { hash_task(~code); hash_left.arrive(); }
    };
    c->pool->run_async((int)order.size(), mixed, &copy_left);
    // every exit path below must wait for the tasks: they reference this frame
    rc = [&]() -> int {
        const bool idle = lane_fork(L, s, s2, cp);
        (void)idle;
        int vrc = 0;
        // the points' launches (decoding + subgroup tests 0.7 ms, the two MSMs that need nothing else 0.5 ms: done long before the last
        // challenge) are issued right BEHIND the first chunk's upload: seven launches are 0.15 ms of this thread's time, and the link
        // should not wait for them -- every later chunk follows the first back to back
        auto launch_points = [&]() -> int {
            HIPCHK(hipMemcpyAsync(L->d_v_raw, h_pts, m * 48, hipMemcpyHostToDevice, s2));
            HIPCHK(hipMemcpyAsync(L->d_v_raw + cap * 48, h_pts + m * 48, m * 48, hipMemcpyHostToDevice, s2));
            if (n_dev) HIPCHK(hipEventRecord(L->ev_decoded, s2));                  // the commitments are up: the device's share of the hash may read them
            if (!late && host_validate_on(c, m)) host_validate_points(c, commitments48n, proofs48n, m, aff_c, st_c, aff_q, st_q, s2);
            else KT(KT_SUBGROUP, s2, g1_validate2_device(L->d_v_raw, (int)m, aff_c, st_c, L->d_v_raw + cap * 48, (int)m, aff_q, st_q, s2, MKZG_LAT_VALIDATE_MODE(m)));
            if (early_msm) {
                KT(KT_RLC, s2, rlc_scalars_device(r_words, nullptr, nullptr, (int)m, exp_offset, L->d_v_s1, nullptr, nullptr, 1, s2));
                const uint32_t* scal[2] = {L->d_v_s1, L->d_v_s1};
                const uint32_t* pts[2] = {aff_q, aff_c};
                const int njob[2] = {(int)m, (int)m};
                KT(KT_VMSM, s2, vrc = vmsm_windows_device(scal, pts, njob, 2, d_parts, d_wsum, s2));
                if (vrc) return MKZG_ERR_BAD_ARGS;
            }
            return MKZG_OK;
        };
        if (nch == 1) { const int prc = launch_points(); if (prc) return prc; }     // a call of one chunk waits for the points' 0.7 ms chain, not for the link: points first
        bool hash_launched = n_dev == 0;
        for (size_t k = 0; k < nch; k++) {
            const size_t lo = plan[k].lo, n = plan[k].m;
            c->pool->wait(left[k]);
            HIPCHK(hipMemcpyAsync(L->d_blobs + lo * MKZG_BYTES_PER_BLOB, L->h_stage + lo * MKZG_BYTES_PER_BLOB, n * MKZG_BYTES_PER_BLOB, hipMemcpyHostToDevice, cp));
            HIPCHK(hipEventRecord(L->ev_chunk[k & 7], cp));
            if (k == 0 && nch > 1) { const int prc = launch_points(); if (prc) return prc; }
            if (!hash_launched && lo + n >= n_dev) {
                // the device's share is up with this chunk: its 2.7 ms chain starts now, on the main stream
                HIPCHK(hipStreamWaitEvent(s, L->ev_chunk[k & 7], 0));
                HIPCHK(hipStreamWaitEvent(s, L->ev_decoded, 0));
                KT(KT_SHA256, s, challenge_device(L->d_blobs, L->d_v_raw, (int)n_dev, L->d_v_z, s));
                hash_launched = true;
            }
            if (k + 1 == nch) HIPCHK(hipStreamWaitEvent(s, L->ev_chunk[k & 7], 0));          // the copy stream is in order: the last chunk's event covers them all
        }
        HIPCHK(hipEventRecord(L->ev_join, s2));
        // the host's share of the challenges: what the call waits for from here on, so this thread hashes too
        c->pool->wait(hash_left, true);
        if (n_dev < m) {
            HIPCHK(hipMemcpyAsync(d_z + n_dev * 32, h_z + n_dev * 32, (m - n_dev) * 32, hipMemcpyHostToDevice, s));
            scalars_to_fr_device(d_z + n_dev * 32, (int)(m - n_dev), L->d_v_z + n_dev * 12, st_x + n_dev, s);       // canonical by construction
        }
        HIPCHK(hipStreamWaitEvent(s, L->ev_join, 0));
        if (late) {
            const int lrc = late->get();
            if (lrc) return lrc;
            if (late->offset + exp_offset + m > 0xffffffffull) return MKZG_ERR_BAD_ARGS;
            memcpy(r_words, late->r_words, 32);
            exp_offset += (uint32_t)late->offset;
        }
        // y = f(z) in one pass over the blobs (canonicity check included), r^i z_i and r^i y_i from the lane that finishes each blob
        KT(KT_EVAL_TREE, s, blob_eval_device(L->d_blobs, L->d_v_z, c->d_roots_brp, c->d_inv_roots_brp, (int)m, L->d_v_y, st_b, L->d_v_zrows, s, false, r_words, exp_offset, L->d_v_s1, L->d_v_s3,
                                             L->d_v_ry));
        if (early_msm) {
            const uint32_t* scal[1] = {L->d_v_s3};
            const uint32_t* pts[1] = {aff_q};
            const int njob[1] = {(int)m};
            KT(KT_VMSM, s, vrc = vmsm_windows_device(scal, pts, njob, 1, d_parts + 2 * parts_per_job, d_wsum + 2 * wsum_per_job, s, 256, 1));
        } else {
            const uint32_t* scal[3] = {L->d_v_s1, L->d_v_s1, L->d_v_s3};
            const uint32_t* pts[3] = {aff_q, aff_c, aff_q};
            const int njob[3] = {(int)m, (int)m, (int)m};
            KT(KT_VMSM, s, vrc = vmsm_windows_device(scal, pts, njob, 3, d_parts, d_wsum, s, 512, 1));
        }
        if (vrc) return MKZG_ERR_BAD_ARGS;
        return MKZG_OK;
    }();
    c->pool->wait(copy_left);
    c->pool->wait(hash_left);
    if (rc) {
        // an early exit may have left the copy stream or the side stream with work the main stream never joined
        L->failed = true;
        return rc;
    }
    // (the main stream waited for the copy stream's last chunk and joined the side stream)
    return verify_fold_tail(c, L, m, true, 0, s, out, true);
}

static int fold_host_piece(mkzg_ctx* c, Lane* L, const uint8_t* const* blobs, const uint8_t* commitments48n, const uint8_t* proofs48n, size_t m, uint32_t exp_offset,
                           const uint8_t* r_le32, VerifyFold* out, int* rc_nil /* out: MKZG_ERR_NIL_INPUT precedence result, or 0 */, RlcLate* late = nullptr) {
    *rc_nil = 0;
    hipStream_t s = L->stream;
    const bool hsha = use_host_sha(c, m);
    int rc = ensure_staging(L, m, MKZG_BYTES_PER_BLOB + 128);
    if (!rc) rc = ensure_verify_ws(L, m);
    if (rc) return rc;
    size_t first_nil = m;
    for (size_t i = 0; i < m; i++) if (!blobs[i]) { first_nil = i; break; }
    if (hsha && !throughput_shape(c)) {
        rc = fold_host_pipelined(c, L, blobs, commitments48n, proofs48n, m, exp_offset, r_le32, out, late);
    } else {
        uint8_t* h_z = L->h_stage + m * MKZG_BYTES_PER_BLOB;              // m x 32 B
        uint8_t* h_pts = h_z + m * 32;                                     // m x 48 B commitments | m x 48 B proofs
        uint8_t* d_z = L->d_blobs + m * MKZG_BYTES_PER_BLOB;
        memcpy(h_pts, commitments48n, m * 48);
        memcpy(h_pts + m * 48, proofs48n, m * 48);
        HIPCHK(hipMemcpyAsync(L->d_v_raw, h_pts, m * 48, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(L->d_v_raw + L->v_cap * 48, h_pts + m * 48, m * 48, hipMemcpyHostToDevice, s));
        // the gather runs ahead on the host pool (all chunks' tasks queued at once); this thread waits for chunk k only and puts its copy in flight
        const size_t CH = 128;
        const size_t nch = (m + CH - 1) / CH;
        const int TPC = hsha ? (int)std::min<size_t>(std::min(m, CH), (size_t)std::max(1, c->pool->size())) : (m >= 16 ? 8 : 1);
        std::vector<Countdown> left(nch);
        for (auto& a : left) a.set(TPC);
        Countdown all_left((int)nch * TPC);
        const std::function<void(int)> gather = [&](int t) {
            const size_t k = (size_t)(t / TPC), lo = k * CH, hi = std::min(m, lo + CH), per = (hi - lo + TPC - 1) / TPC;
            const size_t q = (size_t)(t % TPC);
            for (size_t i = lo + q * per; i < std::min(hi, lo + (q + 1) * per); i++) {
                uint8_t* dst = L->h_stage + i * MKZG_BYTES_PER_BLOB;
                if (blobs[i]) copy_to_staging(dst, blobs[i], MKZG_BYTES_PER_BLOB);
                else memset(dst, 0, MKZG_BYTES_PER_BLOB);
                if (hsha) host_compute_challenge(h_z + i * 32, blobs[i] ? blobs[i] : dst, commitments48n + i * 48);
            }
            left[k].arrive();
        };
        c->pool->run_async((int)nch * TPC, gather, &all_left);
        rc = [&]() -> int {
            for (size_t k = 0; k < nch; k++) {
                const size_t lo = k * CH, hi = std::min(m, lo + CH);
                c->pool->wait(left[k]);
                HIPCHK(hipMemcpyAsync(L->d_blobs + lo * MKZG_BYTES_PER_BLOB, L->h_stage + lo * MKZG_BYTES_PER_BLOB, (hi - lo) * MKZG_BYTES_PER_BLOB, hipMemcpyHostToDevice, s));
            }
            return MKZG_OK;
        }();
        c->pool->wait(all_left);                      // the tasks reference this frame
        if (rc) return rc;
        if (hsha) HIPCHK(hipMemcpyAsync(d_z, h_z, m * 32, hipMemcpyHostToDevice, s));
        rc = verify_fold(c, L, L->d_blobs, hsha ? d_z : nullptr, nullptr, L->d_v_raw, L->d_v_raw + L->v_cap * 48, m, exp_offset, r_le32, s, out, late,
                         commitments48n, proofs48n);
    }
    if (first_nil < m) {
        // the reference deserialises commitment i and proof i before blob i, so a nil blob only surfaces after their checks: an error
        // at an index below the nil blob (or at its commitment / proof) wins; otherwise nil
        if (rc == MKZG_ERR_HIP || rc == MKZG_ERR_BAD_ARGS) return rc;
        const int* h_st = (const int*)L->h_v;
        const size_t cap = L->v_cap;
        *rc_nil = MKZG_ERR_NIL_INPUT;
        for (size_t i = 0; i <= first_nil; i++) {
            if (h_st[i]) { *rc_nil = h_st[i]; break; }
            if (h_st[cap + i]) { *rc_nil = h_st[cap + i]; break; }
            if (i < first_nil && h_st[2 * cap + i]) { *rc_nil = h_st[2 * cap + i]; break; }
        }
        return MKZG_OK;
    }
    return rc;
}
// n items with host pointers in one lane: pieces of HOST_PIECE blobs share r and carry global exponents
static int fold_host_all(mkzg_ctx* c, Lane* L, const uint8_t* const* blobs, const uint8_t* commitments48n, const uint8_t* proofs48n, size_t n, uint64_t exp_offset,
                         const uint8_t* r_le32, VerifyFold* total, RlcLate* late = nullptr) {
    verify_fold_clear(*total);
    VerifyFold f;
    for (size_t off = 0; off < n; off += HOST_PIECE) {
        const size_t m = std::min<size_t>(HOST_PIECE, n - off);
        int rc_nil = 0;
        int rc = fold_host_piece(c, L, blobs + off, commitments48n + off * 48, proofs48n + off * 48, m, (uint32_t)(exp_offset + off), r_le32, &f, &rc_nil, late);
        if (rc_nil) return rc_nil;
        if (rc) return rc;
        verify_fold_add(*total, f);
    }
    return MKZG_OK;
}
int sd_verify_blob_fold_host(mkzg_ctx* c, const uint8_t* const* blobs, const uint8_t* commitments48n, const uint8_t* proofs48n, size_t n, uint64_t exp_offset,
                             const uint8_t r_le32[32], VerifyFold* out, RlcLate* late) {
    verify_fold_clear(*out);
    if (n == 0) return MKZG_OK;
    LaneGuard G(c);
    G.L->last_host_finish_us = 0;
    return G.fail(fold_host_all(c, G.L, blobs, commitments48n, proofs48n, n, exp_offset, r_le32, out, late));
}
// a shard contiguous in HBM: pieces of VMSM_MAX_N items with the global exponents r^(exp_offset + i) (with `late`: r^(late->offset + i),
// r_le32 unused); the caller's stream is synchronised (the sums come back to the host)
int sd_verify_blob_fold_device(mkzg_ctx* c, const void* d_blobs, const void* d_commitments48, const void* d_proofs48, size_t n, uint64_t exp_offset, const uint8_t* r_le32,
                               hipStream_t stream, VerifyFold* out, RlcLate* late) {
    verify_fold_clear(*out);
    if (n == 0) return MKZG_OK;
    if (exp_offset + n > 0xffffffffull) return MKZG_ERR_BAD_ARGS;
    LaneGuard G(c, stream);
    G.L->last_host_finish_us = 0;
    VerifyFold f;
    int rc = MKZG_OK;
    for (size_t off = 0; off < n && !rc; off += VMSM_MAX_N) {
        const size_t m = std::min<size_t>(VMSM_MAX_N, n - off);
        rc = verify_fold(c, G.L, (const uint8_t*)d_blobs + off * MKZG_BYTES_PER_BLOB, nullptr, nullptr, (const uint8_t*)d_commitments48 + off * 48, (const uint8_t*)d_proofs48 + off * 48, m,
                         (uint32_t)(exp_offset + off), r_le32, G.s, &f, late);
        if (!rc) verify_fold_add(*out, f);
    }
    c->last_host_finish_us.store(G.L->last_host_finish_us);
    return G.fail(rc);
}
// the verifiers' 512-byte record of a fold: status | A | Cc | D | e (include/mi355kzg.h)
void verify_fold_record(uint8_t record512[512], int rc, const VerifyFold& total) {
    memset(record512, 0, 512);
    const int32_t st = rc;
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

// Context.VerifyBlobKZGProofBatch (reference verify.go:88-143), device-resident inputs.  Any n.
int mkzg_verify_blob_kzg_proof_batch_device(mkzg_ctx* c, const void* d_blobs, const void* d_commitments48, const void* d_proofs48, size_t n, void* stream) {
    if (!c || (n && (!d_blobs || !d_commitments48 || !d_proofs48))) return MKZG_ERR_BAD_ARGS;
    if (!c->peers.empty() && !(c = ctx_for_pointer(c, d_blobs))) return MKZG_ERR_BAD_ARGS;
    if (n == 0) return MKZG_OK;
    LaneGuard G(c, (hipStream_t)stream);
    return G.fail(verify_device_all(c, G.L, (const uint8_t*)d_blobs, (const uint8_t*)d_commitments48, (const uint8_t*)d_proofs48, n, G.s));
}

int mkzg_verify_blob_kzg_proof_batch(mkzg_ctx* c, const uint8_t* const* blobs, const uint8_t* commitments48n, const uint8_t* proofs48n, size_t n_blobs, size_t n_comm,
                                     size_t n_proofs) {
    if (!c) return MKZG_ERR_BAD_ARGS;
    if (n_blobs != n_comm || n_blobs != n_proofs) return MKZG_ERR_BATCH_LENGTH;      // verify.go:91-95
    const size_t n = n_blobs;
    if (n == 0) return MKZG_OK;
    if (!blobs || !commitments48n || !proofs48n) return MKZG_ERR_BAD_ARGS;
    if (!c->peers.empty()) return md_verify_blob_kzg_proof_batch(c, blobs, commitments48n, proofs48n, n);
    uint8_t r[32];
    if (draw_rlc_scalar(r)) return MKZG_ERR_HIP;
    LaneGuard G(c);
    G.L->last_host_finish_us = 0;
    VerifyFold total;
    int rc = fold_host_all(c, G.L, blobs, commitments48n, proofs48n, n, 0, r, &total);
    if (rc) return G.fail(rc);
    return timed_finish(c, G.L, total);
}

}   // extern "C"
// one (blob, commitment, proof) on its own: what mkzg_verify_blob_kzg_proof is for a caller that is alone
static int verify_blob_one(mkzg_ctx* c, const uint8_t* blob, const uint8_t commitment48[48], const uint8_t proof48[48]) {
    uint8_t r[32];
    if (draw_rlc_scalar(r)) return MKZG_ERR_HIP;
    LaneGuard G(c);
    G.L->last_host_finish_us = 0;
    VerifyFold f;
    int rc_nil = 0;
    const uint8_t* one[1] = {blob};
    int rc = fold_host_piece(c, G.L, one, commitment48, proof48, 1, 0, r, &f, &rc_nil);
    if (rc > MKZG_VERIFY_FAILED && rc != MKZG_ERR_HIP && rc != MKZG_ERR_BAD_ARGS) {
        // the single-proof variant deserialises the blob first (verify.go:51-54)
        const int* h_st = (const int*)G.L->h_v;
        if (h_st[2 * G.L->v_cap]) rc = h_st[2 * G.L->v_cap];
    }
    if (rc) return G.fail(rc);
    return timed_finish(c, G.L, f);
}
extern "C" {
// Context.VerifyBlobKZGProof (reference verify.go:48-83): order blob, commitment, proof.
// Calls that find other such calls in flight are verified TOGETHER (host_combine.hpp): one VerifyBlobKZGProofBatch over the group --
// one random linear combination, one set of launches, ONE pairing on the host where every call alone runs its own (0.4 ms of several
// cores each: sixteen cores verify ~30 k proofs/s that way, whatever the GPU does).  The batch verdict "every item fine" is every
// member's verdict; anything else -- a wrong proof or an input error ANYWHERE in the group -- says nothing about the others, so every
// member then runs its own call as before (the reference's per-call error precedence, verify.go:51-67, is that path's).
int mkzg_verify_blob_kzg_proof(mkzg_ctx* c, const uint8_t* blob, const uint8_t commitment48[48], const uint8_t proof48[48]) {
    if (!c || !commitment48 || !proof48) return MKZG_ERR_BAD_ARGS;
    if (!blob) return MKZG_ERR_NIL_INPUT;
    c = md_pick(c);
    if (!c->opt_combine.load(std::memory_order_relaxed)) return verify_blob_one(c, blob, commitment48, proof48);
    mkzg_ctx::VerifyCall call{blob, commitment48, proof48, MKZG_OK, false};
    c->comb_verify.run(&call, [c](std::vector<mkzg_ctx::VerifyCall*>& items) {
        const size_t n = items.size();
        if (n == 1) {
            items[0]->rc = verify_blob_one(c, items[0]->blob, items[0]->commitment48, items[0]->proof48);
            return;
        }
        c->comb_verify.count_combined(n);
        std::vector<const uint8_t*> ptrs(n);
        std::vector<uint8_t> cms(48 * n), prs(48 * n);
        for (size_t i = 0; i < n; i++) {
            ptrs[i] = items[i]->blob;
            memcpy(cms.data() + 48 * i, items[i]->commitment48, 48);
            memcpy(prs.data() + 48 * i, items[i]->proof48, 48);
        }
        const int rc = mkzg_verify_blob_kzg_proof_batch(c, ptrs.data(), cms.data(), prs.data(), n, n, n);
        for (size_t i = 0; i < n; i++) {
            items[i]->rc = rc;
            items[i]->again = rc != MKZG_OK;           // each member finds out for itself, on its own thread
        }
    });
    if (call.again) return verify_blob_one(c, blob, commitment48, proof48);
    return call.rc;
}

// Context.VerifyKZGProof (reference verify.go:12-43)
int mkzg_verify_kzg_proof(mkzg_ctx* c, const uint8_t commitment48[48], const uint8_t z32[32], const uint8_t y32[32], const uint8_t proof48[48]) {
    if (!c || !commitment48 || !z32 || !y32 || !proof48) return MKZG_ERR_BAD_ARGS;
    c = md_pick(c);
    uint8_t r[32];
    if (draw_rlc_scalar(r)) return MKZG_ERR_HIP;
    LaneGuard G(c);
    Lane* L = G.L;
    L->last_host_finish_us = 0;
    int rc = ensure_verify_ws(L, 1);
    if (!rc) rc = ensure_staging(L, 1, 256);
    if (rc) return G.fail(rc);
    memcpy(L->h_stage, z32, 32);
    memcpy(L->h_stage + 32, y32, 32);
    memcpy(L->h_stage + 64, commitment48, 48);
    memcpy(L->h_stage + 112, proof48, 48);
    HIPCHK(hipMemcpyAsync(L->d_blobs, L->h_stage, 64, hipMemcpyHostToDevice, G.s));
    HIPCHK(hipMemcpyAsync(L->d_v_raw, L->h_stage + 64, 48, hipMemcpyHostToDevice, G.s));
    HIPCHK(hipMemcpyAsync(L->d_v_raw + L->v_cap * 48, L->h_stage + 112, 48, hipMemcpyHostToDevice, G.s));
    VerifyFold f;
    rc = verify_fold(c, L, nullptr, L->d_blobs, L->d_blobs + 32, L->d_v_raw, L->d_v_raw + L->v_cap * 48, 1, 0, r, G.s, &f, nullptr, commitment48, proof48);
    if (rc) return G.fail(rc);
    return timed_finish(c, L, f);
}

// ---- proofs -------------------------------------------------------------------------------
// Context.ComputeBlobKZGProof for a device-resident batch (reference prove.go:46-77 per item):
// status[i] receives the per-item code; proofs of failed items are undefined.
// d_z32: the challenges when the caller hashed the transcripts on the host (n x 32 B big-endian, canonical), else nullptr.
static int compute_blob_proofs_device(mkzg_ctx* c, Lane* L, const uint8_t* d_blobs, const uint8_t* d_commitments48, size_t n, uint8_t* d_proofs48, int* d_status, hipStream_t s,
                                      const uint8_t* d_z32 = nullptr, uint32_t* d_raw = nullptr /* n == 1: ask for the proof as XYZZ words for commit_host_finish */,
                                      bool* raw_used = nullptr, const uint8_t* h_cm48 = nullptr /* n == 1: the commitment's bytes on the host */,
                                      int* h_st_cm_out = nullptr /* n == 1 and the commitment validated on the host: its status goes HERE and the caller, who syncs
                                                                    anyway for the proof, merges it with the blob's (L->d_q_st[0]): no sync of this function's own */) {
    // Slices of up to 4096 proofs.  Per slice the latency-bound stages run ONCE over the whole slice -- Fiat-Shamir challenges
    // (a 2050-compression chain per blob whatever the count) on the caller's stream, commitment decoding + subgroup checks on the
    // side stream -- and only the quotient + MSM stage is cut into max_batch pieces (its workspace is 196 KB per blob).
    size_t done = 0;
    while (done < n) {
        const size_t slice = std::min<size_t>(n - done, 4096);
        int rc = ensure_quot_ws(L, std::max<size_t>((size_t)c->opt_max_batch, slice), (size_t)c->opt_max_batch);
        if (rc) return rc;
        rc = ensure_verify_ws(L, slice);
        if (rc) return rc;
        const uint8_t* blobs = d_blobs + done * MKZG_BYTES_PER_BLOB;
        const uint8_t* cms = d_commitments48 + done * 48;
        int* st_blob = L->d_q_st;
        int* st_cm = L->d_q_st + L->q_cap;
        // commitment must be a valid subgroup point (prove.go:56-62)
        hipStream_t s2 = lane_side_stream(L, s);
        lane_fork(L, s, s2);
        int h_st_cm = -1;                                  // n == 1 with the commitment's bytes on the host: its status from host_validate.hpp (below)
        if (!(n == 1 && h_cm48 && host_validate_on(c, 1))) g1_validate2_device(cms, (int)slice, L->d_v_aff, st_cm, nullptr, 0, nullptr, nullptr, s2);
        HIPCHK(hipEventRecord(L->ev_join, s2));
        if (d_z32) scalars_to_fr_device(d_z32 + done * 32, (int)slice, L->d_q_z, L->d_v_stx, s);
        else KT(KT_SHA256, s, challenge_device(blobs, cms, (int)slice, L->d_q_z, s));
        for (size_t off = 0; off < slice; off += (size_t)c->opt_max_batch) {
            const size_t chunk = std::min<size_t>(slice - off, (size_t)c->opt_max_batch);
            KT(KT_QUOTIENT, s, blob_quotient_device(blobs + off * MKZG_BYTES_PER_BLOB, L->d_q_z + off * 12, c->d_roots_brp, c->d_inv_roots_brp, (int)chunk, L->d_q_y + off * 12,
                                                    L->d_quot, st_blob + off, s));
            rc = commit_scalars_device(c, L, (const uint8_t*)L->d_quot, (size_t)4096 * 48, 1, chunk, d_proofs48 + (done + off) * 48, d_status + done + off, s, n == 1 ? d_raw : nullptr, n == 1 ? raw_used : nullptr);
            if (rc) return rc;
        }
        HIPCHK(hipStreamWaitEvent(s, L->ev_join, 0));
        // merge statuses: blob error first, then commitment error (prove.go:49-62)
        HIPCHK(hipGetLastError());
        // (into the lane's pinned area: a copy into pageable memory blocks and polls inside the runtime until the stream gets there)
        int* stb = (int*)L->h_v;
        int* stc = stb + L->v_cap;
        if (n == 1 && h_cm48 && h_st_cm_out && host_validate_on(c, 1)) {      // the single-proof entry point: everything comes back with ITS one copy + sync
            uint32_t row[32];
            *h_st_cm_out = g1h_validate_row(h_cm48, row);                     // on this thread while the quotient and its MSM run
            return MKZG_OK;
        }
        HIPCHK(hipMemcpyAsync(stb, st_blob, slice * sizeof(int), hipMemcpyDeviceToHost, s));
        if (n == 1 && h_cm48 && host_validate_on(c, 1)) {  // the one commitment's checks on this thread while the quotient and its MSM run
            uint32_t row[32];
            h_st_cm = g1h_validate_row(h_cm48, row);
        } else {
            HIPCHK(hipMemcpyAsync(stc, st_cm, slice * sizeof(int), hipMemcpyDeviceToHost, s));
        }
        LANE_SYNC(c, L, s);
        if (h_st_cm >= 0) stc[0] = h_st_cm;
        bool any = false;
        for (size_t i = 0; i < slice; i++) { stb[i] = stb[i] ? stb[i] : stc[i]; any = any || stb[i]; }
        if (any) HIPCHK(hipMemcpyAsync(d_status + done, stb, slice * sizeof(int), hipMemcpyHostToDevice, s));
        if (any) LANE_SYNC(c, L, s);
        done += slice;
    }
    return MKZG_OK;
}
int mkzg_compute_blob_kzg_proof_batch_device(mkzg_ctx* c, const void* d_blobs, const void* d_commitments48, size_t n, void* d_proofs48, int* d_status, void* stream) {
    if (!c || (n && (!d_blobs || !d_commitments48 || !d_proofs48 || !d_status))) return MKZG_ERR_BAD_ARGS;
    if (!c->peers.empty() && !(c = ctx_for_pointer(c, d_blobs))) return MKZG_ERR_BAD_ARGS;
    if (n == 0) return MKZG_OK;
    LaneGuard G(c, (hipStream_t)stream);
    return G.fail(compute_blob_proofs_device(c, G.L, (const uint8_t*)d_blobs, (const uint8_t*)d_commitments48, n, (uint8_t*)d_proofs48, d_status, G.s));
}

int mkzg_compute_blob_kzg_proof(mkzg_ctx* c, const uint8_t* blob, const uint8_t commitment48[48], int num_threads, uint8_t out48[48]) {
    if (!c || !commitment48 || !out48) return MKZG_ERR_BAD_ARGS;
    if (!blob) return MKZG_ERR_NIL_INPUT;
    c = md_pick(c);
    LaneGuard G(c);
    Lane* L = G.L;
    int rc = ensure_staging(L, 1, MKZG_BYTES_PER_BLOB + 128);
    if (!rc) rc = ensure_verify_ws(L, 1);
    if (rc) return G.fail(rc);
    memcpy(L->h_stage, blob, MKZG_BYTES_PER_BLOB);
    memcpy(L->h_stage + MKZG_BYTES_PER_BLOB, commitment48, 48);
    const bool hsha = use_host_sha(c, 1);                      // one transcript: ~70 us on this thread instead of the 2.7 ms device chain
    if (hsha) host_compute_challenge(L->h_stage + MKZG_BYTES_PER_BLOB + 64, L->h_stage, commitment48);
    HIPCHK(hipMemcpyAsync(L->d_blobs, L->h_stage, MKZG_BYTES_PER_BLOB + 96, hipMemcpyHostToDevice, G.s));
    // the proof's inversion + compression on this thread instead of an ~90 us one-wavefront chain (api.hip commit_host_finish) when the direct
    // table serves the call (commit_scalars_device says so)
    bool hfin = false;
    int st_cm_host = -1;                                           // >= 0: the commitment was validated on this thread (compute_blob_proofs_device did not sync)
    rc = compute_blob_proofs_device(c, L, L->d_blobs, L->d_blobs + MKZG_BYTES_PER_BLOB, 1, L->d_out, L->d_status, G.s, hsha ? L->d_blobs + MKZG_BYTES_PER_BLOB + 64 : nullptr,
                                    reinterpret_cast<uint32_t*>(L->d_out), &hfin, commitment48, &st_cm_host);
    if (rc) return G.fail(rc);
    HIPCHK(hipMemcpyAsync(L->h_out, L->d_out, hfin ? 256 : 48, hipMemcpyDeviceToHost, G.s));
    HIPCHK(hipMemcpyAsync(L->h_out + 256, st_cm_host >= 0 ? L->d_q_st : L->d_status, sizeof(int), hipMemcpyDeviceToHost, G.s));
    LANE_SYNC(c, G.L, G.s);
    L->drained = true;
    int st = 0;
    memcpy(&st, L->h_out + 256, sizeof(int));                      // the merged status, or (host-validated commitment) the blob's: blob error first, then the commitment's (prove.go:49-62)
    if (!st && st_cm_host > 0) st = st_cm_host;
    if (st) return st;
    if (hfin) commit_host_finish(reinterpret_cast<const uint32_t*>(L->h_out), out48);
    else memcpy(out48, L->h_out, 48);
    if (num_threads >= 1024) return MKZG_ERR_TOO_MANY_GOROUTINES;
    return MKZG_OK;
}

// Context.ComputeKZGProof (reference prove.go:85-111): proof and claimed value at a caller-chosen z
int mkzg_compute_kzg_proof(mkzg_ctx* c, const uint8_t* blob, const uint8_t z32[32], int num_threads, uint8_t proof48[48], uint8_t y32[32]) {
    if (!c || !z32 || !proof48 || !y32) return MKZG_ERR_BAD_ARGS;
    if (!blob) return MKZG_ERR_NIL_INPUT;
    c = md_pick(c);
    LaneGuard G(c);
    Lane* L = G.L;
    hipStream_t s = G.s;
    int rc = ensure_staging(L, 1, MKZG_BYTES_PER_BLOB + 128);
    if (!rc) rc = ensure_quot_ws(L, (size_t)c->opt_max_batch, (size_t)c->opt_max_batch);
    if (rc) return G.fail(rc);
    memcpy(L->h_stage, blob, MKZG_BYTES_PER_BLOB);
    memcpy(L->h_stage + MKZG_BYTES_PER_BLOB, z32, 32);
    HIPCHK(hipMemcpyAsync(L->d_blobs, L->h_stage, MKZG_BYTES_PER_BLOB + 32, hipMemcpyHostToDevice, s));
    int* st_z = L->d_q_st + L->q_cap;
    scalars_to_fr_device(L->d_blobs + MKZG_BYTES_PER_BLOB, 1, L->d_q_z, st_z, s);
    blob_quotient_device(L->d_blobs, L->d_q_z, c->d_roots_brp, c->d_inv_roots_brp, 1, L->d_q_y, L->d_quot, L->d_q_st, s);
    bool hfin = false;                                             // as in mkzg_compute_blob_kzg_proof
    rc = commit_scalars_device(c, L, (const uint8_t*)L->d_quot, (size_t)4096 * 48, 1, 1, L->d_out, L->d_status, s, reinterpret_cast<uint32_t*>(L->d_out), &hfin);
    if (rc) return G.fail(rc);
    fr_to_scalars_device(L->d_q_y, 1, L->d_blobs + MKZG_BYTES_PER_BLOB + 64, s);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(L->h_out, L->d_out, hfin ? 256 : 48, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(L->h_out + 256, L->d_blobs + MKZG_BYTES_PER_BLOB + 64, 32, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(L->h_out + 288, L->d_q_st, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(L->h_out + 292, st_z, sizeof(int), hipMemcpyDeviceToHost, s));
    LANE_SYNC(c, L, s);
    L->drained = true;
    int stb = 0, stz = 0;
    memcpy(&stb, L->h_out + 288, sizeof(int));
    memcpy(&stz, L->h_out + 292, sizeof(int));
    if (stb) return stb;                                           // blob first (prove.go:88-91)
    if (stz) return stz;
    if (hfin) commit_host_finish(reinterpret_cast<const uint32_t*>(L->h_out), proof48);
    else memcpy(proof48, L->h_out, 48);
    memcpy(y32, L->h_out + 256, 32);
    if (num_threads >= 1024) return MKZG_ERR_TOO_MANY_GOROUTINES;
    return MKZG_OK;
}

// Domain.EvaluateLagrangePolynomial for caller-chosen points through each evaluation kernel (include/mi355kzg.h): the parity pin of the
// three forms against each other and the oracle, on-domain points included
int mkzg_evaluate_blobs_device(mkzg_ctx* c, const void* d_blobs, const uint8_t* z32n, size_t n, int form, uint8_t* y32n, int* status, void* stream) {
    if (!c || form < 0 || form > 2 || n > 4096 || (n && (!d_blobs || !z32n || !y32n || !status))) return MKZG_ERR_BAD_ARGS;
    if (!c->peers.empty() && !(c = ctx_for_pointer(c, d_blobs))) return MKZG_ERR_BAD_ARGS;
    if (n == 0) return MKZG_OK;
    LaneGuard G(c, (hipStream_t)stream);
    Lane* L = G.L;
    hipStream_t s = G.s;
    int rc = ensure_verify_ws(L, n);
    if (!rc && form == 0) rc = ensure_twiddles(c);
    if (!rc && form == 0 && n > L->v_coeff_cap) {
        L->v_coeff_cap = 0;
        rc = dev_realloc(L->d_v_coeffs, n * 4096 * 48);
        if (!rc) L->v_coeff_cap = n;
    }
    if (!rc && form == 2) rc = ensure_quot_ws(L, n, n);
    if (rc) return G.fail(rc);
    const size_t cap = L->v_cap;
    int* st_b = L->d_v_st + 2 * cap;
    int* st_z = L->d_v_stx;
    const uint8_t* blobs = (const uint8_t*)d_blobs;
    HIPCHK(hipMemcpyAsync(L->d_v_s3, z32n, n * 32, hipMemcpyHostToDevice, s));
    scalars_to_fr_device((const uint8_t*)L->d_v_s3, (int)n, L->d_v_z, st_z, s);
    if (form == 0) {
        cells_from_blobs_device(blobs, (int)n, c->d_tw_fwd, c->d_tw_inv, L->d_v_coeffs, nullptr, st_b, s);
        poly_eval_device(L->d_v_coeffs, L->d_v_z, (int)n, L->d_v_y, st_b, s);
    } else if (form == 1) {
        blob_eval_device(blobs, L->d_v_z, c->d_roots_brp, c->d_inv_roots_brp, (int)n, L->d_v_y, st_b, L->d_v_zrows, s);
    } else {
        blob_quotient_device(blobs, L->d_v_z, c->d_roots_brp, c->d_inv_roots_brp, (int)n, L->d_v_y, L->d_quot, st_b, s);
    }
    fr_to_scalars_device(L->d_v_y, (int)n, (uint8_t*)L->d_v_s1, s);
    HIPCHK(hipGetLastError());
    std::vector<int> stb(n), stz(n);
    HIPCHK(hipMemcpyAsync(y32n, L->d_v_s1, n * 32, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(stb.data(), st_b, n * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(stz.data(), st_z, n * sizeof(int), hipMemcpyDeviceToHost, s));
    LANE_SYNC(c, L, s);
    L->drained = true;
    int first = MKZG_OK;
    for (size_t i = 0; i < n; i++) {
        status[i] = stb[i] ? stb[i] : stz[i];
        if (status[i] && !first) first = status[i];
    }
    return first;
}

// The finer seam of the reference, multiexp.MultiExpG1 (internal/multiexp/multiexp.go:20-26): sum s_i P_i with gnark-crypto's in-memory
// layouts on both sides -- scalars n x 4 u64 (fr.Element, Montgomery R = 2^256), points n x 12 u64 (G1Affine: X, Y as fp.Element, Montgomery
// R = 2^384, identity (0,0)), result one G1Affine.  Like the reference it does not validate the points.  Any n (pieces of 8192).
int mkzg_msm_g1(mkzg_ctx* c, const uint64_t* scalars_mont4, const uint64_t* points_mont12, size_t n, uint64_t out_mont12[12]) {
    if (!c || !out_mont12 || (n && (!scalars_mont4 || !points_mont12))) return MKZG_ERR_BAD_ARGS;
    c = md_pick(c);
    LaneGuard G(c);
    Lane* L = G.L;
    hipStream_t s = G.s;
    G1HX total;
    G1H::set_inf(total);
    for (size_t done = 0; done < n; done += VMSM_MAX_N) {
        const size_t m = std::min<size_t>(VMSM_MAX_N, n - done);
        int rc = ensure_verify_ws(L, m);
        if (rc) return G.fail(rc);
        // d_v_raw holds 2 x cap x 48 B = cap x 96 B: exactly the points; d_v_s3 takes the raw scalars, d_v_s1 the converted ones
        HIPCHK(hipMemcpyAsync(L->d_v_raw, points_mont12 + done * 12, m * 96, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(L->d_v_s3, scalars_mont4 + done * 4, m * 32, hipMemcpyHostToDevice, s));
        gnark_to_vmsm_device(L->d_v_s3, (const uint32_t*)L->d_v_raw, (int)m, L->d_v_s1, L->d_v_aff, s);
        const uint32_t* scal[1] = {L->d_v_s1};
        const uint32_t* pts[1] = {L->d_v_aff};
        const int nj[1] = {(int)m};
        if (vmsm_windows_device(scal, pts, nj, 1, L->d_v_parts, L->d_v_misc + 16, s, 2048, 1)) return G.fail(MKZG_ERR_BAD_ARGS);
        HIPCHK(hipGetLastError());
        uint32_t* h_misc = (uint32_t*)(L->h_v + 3 * L->v_cap * sizeof(int));
        HIPCHK(hipMemcpyAsync(h_misc, L->d_v_misc, (16 + VMSM_WINDOWS * 48) * 4, hipMemcpyDeviceToHost, s));
        LANE_SYNC(c, L, s);
        G1HX part;
        g1h_horner_words(part, h_misc + 16, VMSM_WINDOWS, VMSM_WINDOW_BITS);
        G1H::add(total, total, part);
    }
    G1AffH a;
    g1h_to_affine(a, total);
    if (a.inf) {
        memset(out_mont12, 0, 96);
    } else {
        memcpy(out_mont12, a.x.l, 48);             // HostFp is 6 x u64 Montgomery with R = 2^384: gnark's fp.Element as is
        memcpy(out_mont12 + 6, a.y.l, 48);
    }
    return MKZG_OK;
}

// ---- sharded batch verification (multi-process jobs, one context per GPU): each rank folds its shard with the GLOBAL exponents
// r^(offset + i) and publishes one 512-byte record; any rank adds the records and runs the pairing.
// Record: [0..4) status (LE int) | [16..64) A | [64..112) Cc | [112..160) D (compressed G1) | [160..192) e (32 B big-endian) | zero padding
int mkzg_verify_blob_kzg_proof_batch_partial_device(mkzg_ctx* c, const void* d_blobs, const void* d_commitments48, const void* d_proofs48, size_t n_local, uint64_t offset,
                                                    const uint8_t r32[32], uint8_t record512[512], void* stream) {
    if (!c || !r32 || !record512) return MKZG_ERR_BAD_ARGS;
    memset(record512, 0, 512);
    int rc = MKZG_OK;
    VerifyFold total;
    verify_fold_clear(total);
    if (n_local && (!d_blobs || !d_commitments48 || !d_proofs48)) rc = MKZG_ERR_BAD_ARGS;
    else if (!c->peers.empty() && n_local && !(c = ctx_for_pointer(c, d_blobs))) rc = MKZG_ERR_BAD_ARGS;
    else {
        uint8_t r_le[32];
        for (int i = 0; i < 32; i++) r_le[i] = r32[31 - i];              // big-endian bytes -> little-endian words
        rc = sd_verify_blob_fold_device(c, d_blobs, d_commitments48, d_proofs48, n_local, offset, r_le, (hipStream_t)stream, &total, nullptr);
    }
    verify_fold_record(record512, rc, total);                           // whatever the outcome, the record says so: it is what travels
    return rc;
}
int mkzg_verify_blob_kzg_proof_batch_finish(mkzg_ctx* c, const uint8_t* records512, size_t n_records) {
    if (!c || (n_records && !records512)) return MKZG_ERR_BAD_ARGS;
    if (!c->peers.empty()) c = c->peers[0];
    VerifyFold f;
    verify_fold_clear(f);
    for (size_t k = 0; k < n_records; k++) {                       // first failing rank (lowest indices) decides the error
        int32_t st;
        memcpy(&st, records512 + 512 * k, 4);
        if (st) return st;
    }
    for (size_t k = 0; k < n_records; k++) {
        const uint8_t* rec = records512 + 512 * k;
        G1AffH a;
        G1HX t;
        G1HX* dst[3] = {&f.A, &f.Cc, &f.D};
        for (int q = 0; q < 3; q++) {
            if (!g1h_decompress(a, rec + 16 + 48 * q)) return MKZG_ERR_BAD_ARGS;
            if (a.inf) continue;
            G1H::from_affine(t, a.x, a.y);
            G1H::add(*dst[q], *dst[q], t);
        }
        uint32_t ew[8];
        for (int i = 0; i < 8; i++) ew[7 - i] = ((uint32_t)rec[160 + 4 * i] << 24) | ((uint32_t)rec[161 + 4 * i] << 16) | ((uint32_t)rec[162 + 4 * i] << 8) | rec[163 + 4 * i];
        fr_words_add(f.e, f.e, ew);
    }
    auto t0 = std::chrono::steady_clock::now();
    const int rc = sd_verify_finish(c, f);
    c->last_host_finish_us.store(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    return rc;
}

// host-side share of the last verification (microseconds): Horner + [e]G + pairing
double mkzg_last_host_finish_us(mkzg_ctx* c) {
    if (!c) return 0.0;
    if (!c->peers.empty()) c = c->peers[0];
    return c->last_host_finish_us.load();
}

}  // extern "C"
