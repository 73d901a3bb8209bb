// ctx.h -- the context object behind the C ABI (internal to libmi355kzg.so).
//
// A context is what go-eth-kzg's `Context` is (reference api.go:17-28): immutable after construction and shared by any
// number of caller threads (reference verify.go:152-169, examples_test.go:11).  Everything a call MUTATES lives in a Lane
// (streams, events, workspaces, pinned staging); a context owns NLANE of them and a call holds exactly one from entry to
// return, so NLANE calls run concurrently on one GPU and only the lane hand-over takes the context mutex.  What calls
// SHARE -- the SRS-derived tables in HBM and the host-side pairing precomputation -- is built once (lazily, under init_mu)
// and is read-only afterwards.  One context per device; a multi-device context is a primary plus one peer per further
// device (api_multi.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <condition_variable>
#include <memory>
#include <functional>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/mi355kzg.h"
#include "msm_kernels.h"
#include "verify_kernels.h"
#include "cells_kernels.h"
#include "fk20_kernels.h"
#include "recover_kernels.h"
#include "host_g1.hpp"
#include "host_pool.hpp"
#include "host_copy.hpp"
#include "host_combine.hpp"

using namespace kzg;

// a HIP call failed on this thread since its lane was acquired: the lane's release then synchronises all of its streams (work may be
// in flight on any of them when HIPCHK returns from the middle of a call)
extern thread_local bool tl_hip_error;
#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e__ = (expr);                                                                       \
        if (e__ != hipSuccess) {                                                                       \
            fprintf(stderr, "[mi355kzg] HIP error %s at %s:%d: %s\n", hipGetErrorName(e__), __FILE__, __LINE__, #expr); \
            tl_hip_error = true;                                                                       \
            return MKZG_ERR_HIP;                                                                       \
        }                                                                                              \
    } while (0)

// device / pinned buffers that grow on demand: the old block is released and the pointer cleared BEFORE the new allocation, so a
// failed hipMalloc leaves a null pointer and a zero capacity behind, never a dangling one
template <class T>
static inline int dev_realloc(T*& p, size_t bytes) {
    if (p) { (void)hipFree(p); p = nullptr; }
    HIPCHK(hipMalloc(&p, bytes ? bytes : 16));
    return MKZG_OK;
}
// The batch-wide scalar r and a shard's first global index arriving LATE (api_exchange.hip: both come out of a collective that waits
// for the slowest rank of the job).  A fold handed one of these launches what needs neither -- hashing, decoding, subgroup tests,
// gathers and uploads -- and calls get() where r is first used; exponents are then (offset + the fold's own index).
struct RlcLate {
    std::function<int(uint32_t r_words[8], uint64_t* offset)> wait;   // blocks; fills r (eight little-endian words, reduced) and the offset
    bool done = false;
    int rc = 0;
    uint32_t r_words[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t offset = 0;
    int get() {
        if (!done) { rc = wait(r_words, &offset); done = true; }
        return rc;
    }
};
// ... zero-filled and complete before it returns (ticket words of the kernels that fold their own last step: k_vmsm_window, k_cell_interp)
template <class T>
static inline int dev_realloc_zero(T*& p, size_t bytes) {
    int rc = dev_realloc(p, bytes);
    if (rc) return rc;
    // (on the calling thread's own stream, not the legacy null stream: that one synchronises implicitly with every blocking stream of the
    // process, which HIP refuses -- hipErrorStreamCaptureImplicit -- while another call's stream is being captured into its launch graph)
    HIPCHK(hipMemsetAsync(p, 0, bytes ? bytes : 16, hipStreamPerThread));
    HIPCHK(hipStreamSynchronize(hipStreamPerThread));
    return MKZG_OK;
}
template <class T>
static inline int pinned_realloc(T*& p, size_t bytes) {
    if (p) { (void)hipHostFree(p); p = nullptr; }
    HIPCHK(hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault));
    return MKZG_OK;
}
template <class T>
static inline void dev_free(T*& p) {
    if (p) { (void)hipFree(p); p = nullptr; }
}

// The calling thread's current HIP device is the caller's business (a PyTorch or HIP program keeps allocating and launching on it
// after our call returns): every entry point switches to the context's device for the duration of the call and switches BACK.
// Also taken by everything that allocates or launches outside a lane (the lazily built shared tables, construction, destruction):
// a pool worker or a caller thread may sit on any device.
struct DeviceScope {
    int prev = -1, dev;
    explicit DeviceScope(int device) : dev(device) {
        if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
        if (prev != dev) (void)hipSetDevice(dev);
    }
    ~DeviceScope() { if (prev >= 0 && prev != dev) (void)hipSetDevice(prev); }
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
};

struct mkzg_ctx;

// Per-call mutable state.  Owned by one call at a time (lane_acquire / lane_release in api.hip).
struct Lane {
    int id = 0;
    bool busy = false;
    uint64_t released_at = 0;              // hand-over order: the lane that has been free longest goes first
    hipStream_t stream = nullptr;          // main stream of host-pointer entry points (device-pointer entry points bring their own)
    hipStream_t side = nullptr;            // point validation runs beside SHA-256 / evaluation
    hipStream_t side2 = nullptr;           // the cell verifier's subgroup test beside decoding + MSMs
    hipStream_t aux[2] = {nullptr, nullptr};   // with stream and side2: the four compute streams of the host-pointer commit pipeline
    hipEvent_t ev_cs[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_decoded = nullptr;
    hipEvent_t ev_block = nullptr;         // hipEventBlockingSync: a caller that shares the host with many others sleeps on it instead of polling the stream
    hipEvent_t done = nullptr;             // recorded on the call's main stream at release: the next owner's stream waits for it (no host sync)
    hipStream_t last_stream = nullptr;     // the stream of the lane's last call: a caller that pipelines calls on ONE stream keeps one lane (lane_acquire)
    bool used = false;
    int others_active = 0;                 // at hand-over: lanes held by other calls or whose last call is still running on the GPU (latency vs throughput choices)
    bool warm = false;                     // has run a call before (workspaces allocated)
    // ---- commitment MSM workspaces
    CommitPlan plan;                       // bucket method: per-lane arrays; plan.d_table points at the context's shared table
    int plan_c = 0, plan_batch = 0;
    DirectSlot dslot;                      // direct-table method: lane accumulators
    int dslot_batch = 0;
    size_t dslot_records = 0;              // capacity of dslot.d_parts in 256-byte XYZZ records (512 per blob of max_batch at first; 4096 once a one-scalar-per-lane shape asked for it)
    // ---- staging of host-pointer entry points
    uint8_t* h_stage = nullptr;            // pinned
    size_t h_stage_bytes = 0;
    uint8_t* d_blobs = nullptr;
    size_t d_blobs_bytes = 0;
    uint8_t* d_out = nullptr;
    int* d_status = nullptr;
    size_t d_out_items = 0;
    uint8_t* h_out = nullptr;              // pinned: results of the chunked commit path (48 B + status per blob)
    size_t h_out_items = 0;
    hipEvent_t ev_chunk[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    // ---- EIP-4844 verification workspace, sized for v_cap proofs
    size_t v_cap = 0;
    uint8_t* d_v_raw = nullptr;            // 2 v_cap x 48 B   (commitments | proofs)
    uint32_t* d_v_aff = nullptr;           // 2 v_cap affine rows
    // ONE device block for everything a verifier call brings back (round 5: three copies per call until then, each a packet in a hardware
    // queue the call shares): [statuses 3 v_cap ints (commitments | proofs | blobs)] [misc: r, e, window sums] [16 words] [aux v_cap x 8
    // words: the blob verifier's r^i y_i, or the cell verifier's three verdict arrays of v_cap ints each] -- the pinned mirror h_v has the
    // same layout, so ONE copy of a prefix of the block serves both verifiers; behind it, device only: a scratch status array and the
    // slice sums of the MSMs (with the tickets of k_vmsm_window: zero when allocated)
    uint8_t* d_v_block = nullptr;
    int* d_v_st = nullptr;                 // -> block
    int* d_v_stx = nullptr;                // -> block: v_cap ints of scratch (status of a second scalar array)
    uint32_t* d_v_parts = nullptr;         // -> block: VMSM_PART_WORDS
    uint32_t* d_v_z = nullptr;             // v_cap Fr rows
    uint32_t* d_v_y = nullptr;
    uint32_t* d_v_zrows = nullptr;          // v_cap x 12 rows of 12 words: z^(2^L), L = 0..11 -- what the evaluation needs of z alone (the hash launch or k_blob_eval_prep writes them)
    uint32_t* d_v_s1 = nullptr;            // v_cap x 8 words
    uint32_t* d_v_s3 = nullptr;
    uint32_t* d_v_ry = nullptr;            // -> block (aux): v_cap x 8 words: r^i y_i per blob (throughput shape: the evaluation kernel writes them, the host adds them up)
    uint32_t* d_v_misc = nullptr;          // -> block: [0..8) r, [8..16) e, [16 .. 16+4*32*48) window sums
    uint8_t* h_v = nullptr;                // pinned: statuses + misc
    size_t h_v_bytes = 0;
    uint32_t* d_v_coeffs = nullptr;        // coefficient rows of the blobs (v_coeff_cap x 4096 x 12 words)
    size_t v_coeff_cap = 0;
    // quotient workspace (proof computation)
    size_t q_cap = 0, quot_cap = 0;        // capacities of the scalar/status arrays and of the quotient rows
    uint32_t* d_quot = nullptr;            // quot_cap x 4096 Fr rows
    uint32_t* d_q_z = nullptr;
    uint32_t* d_q_y = nullptr;
    int* d_q_st = nullptr;
    // ---- EIP-7594
    size_t cell_cap = 0;                   // blobs per launch for the cell paths
    uint32_t* d_coeffs = nullptr;          // cell_cap x 4096 Fr rows
    uint8_t* d_cells = nullptr;            // cell_cap x 262144 B
    int* d_cell_st = nullptr;
    size_t cv_cap = 0;                     // VerifyCellKZGProofBatch workspace, cv_cap cells
    size_t cv_cells_cap = 0;               // cells the lane can stage itself (host-pointer calls; device-pointer calls bring theirs)
    uint8_t* d_cv_cells = nullptr;         // cv_cells_cap x 2048 B
    uint8_t* d_cv_in = nullptr;            // one upload per call: commitment rows | proofs | cell indices (u64) | row indices (u32), packed by the call's counts
    uint32_t* d_cv_rp = nullptr;           // cv_cap Fr rows
    uint32_t* d_cv_weights = nullptr;      // cv_cap x 8 words
    uint32_t* d_cv_partials = nullptr;     // CELL_INTERP_BLOCKS x 64 Fr rows
    uint32_t* d_cv_interp = nullptr;       // 64 x 8 words
    size_t rec_cap = 0;                    // blobs the recovery scratch / position arrays hold
    uint32_t* d_rec_scratch = nullptr;
    int* d_rec_pos = nullptr;              // per blob 128 positions + 64 missing ids, then one count per blob
    size_t fk_cap = 0;
    uint32_t* d_fk_scalars = nullptr;      // fk_cap x 128 x 64 x 8 words
    uint32_t* d_fk_results = nullptr;      // 2 x fk_cap x 128 XYZZ rows (MSM results + second buffer of the transform stage)
    void* d_fk_seg = nullptr;              // accumulators and tables of the segmented products (fk20_kernels.hip k_fk20_toom_seg): calls of 65..192 blobs, 159 KB per product and 64 blobs
    size_t fk_seg_bytes = 0;
    uint32_t* d_fk_parts = nullptr;        // fk_cap x 128 x 24 XYZZ rows: per-lane partial sums of the FK20 MSMs, then the transform stage's terms
    uint8_t* d_fk_proofs = nullptr;        // fk_cap x 128 x 48 B
    // ---- profiling (hipEvent brackets on the streams this lane's kernels are launched on)
    KernelTimer timer;
    double last_host_finish_us = 0;
    bool failed = false;                   // set by an error exit: the release then synchronises every stream of the lane
    // launch graphs (round 5, lane_graph_run below): a call's fixed sequence of launches behind its last host-side decision is captured
    // and replayed through ONE executable graph per lane and slot; hipGraphExecUpdate carries the call's scalars and sizes over
    hipGraphExec_t graph_exec[3] = {nullptr, nullptr, nullptr};     // slot 0: the blob verifier's throughput shape behind its hash round; 1 / 2: a device-resident commitment batch over blobs / over quotient rows
    bool drained = false;                  // set by a call that has waited for all its device work: the next owner need not wait for, or query, `done`
};

// One launch of the Fiat-Shamir hash kernel for all the calls that were waiting for one (api_verify.hip hash_round_join)
// What a call of the blob verifier's throughput shape hands its round so that the ROUND carries on behind the hash (deep rounds,
// api_verify.hip): the evaluation of its blobs with its share of the random linear combination, its three MSMs, its verdict block's
// copy back -- on the round's stream, the first two as ONE launch each for all the round's calls.
struct DeepJob {
    hipEvent_t validated = nullptr;             // recorded on the call's stream behind the validation of its points (the MSMs read their affine rows)
    const uint8_t* blobs = nullptr;
    const uint32_t* zrows = nullptr;
    uint32_t* y = nullptr;
    int* st_b = nullptr;
    uint32_t* s1 = nullptr;
    uint32_t* s3 = nullptr;
    uint32_t* ry = nullptr;
    uint32_t r[8] = {};
    uint32_t exp_offset = 0;
    int n = 0;
    const uint32_t* aff_q = nullptr;
    const uint32_t* aff_c = nullptr;
    uint32_t* parts = nullptr;
    uint32_t* wsum = nullptr;
    void* h_dst = nullptr;                      // the verdict block: device -> the lane's pinned mirror
    const void* d_src = nullptr;
    size_t copy_bytes = 0;
};
struct HashRound {
    kzg::ShaJobs jobs{};
    hipEvent_t ready[kzg::SHA_MAX_JOBS] = {};   // per job: its inputs are ready on the call's stream
    DeepJob deep[kzg::SHA_MAX_JOBS];
    bool is_deep[kzg::SHA_MAX_JOBS] = {};
    hipEvent_t hashed = nullptr;                // the hash launch has finished (what a call that carries on in its own stream waits for)
    hipEvent_t done = nullptr;                  // everything the round launched has finished
    bool launched = false;
    int rc = 0;
    ~HashRound() { if (done) (void)hipEventDestroy(done); if (hashed) (void)hipEventDestroy(hashed); }
};

struct mkzg_ctx {
    int device = 0;
    std::vector<mkzg_ctx*> peers;          // further devices of a multi-device context (owned by this, the primary)
    std::atomic<unsigned> rr{0};           // round-robin cursor of single-item calls over the devices
    // ---- lanes
#ifndef MKZG_NLANE
#define MKZG_NLANE 32            // measured (tools/trace.py verify_inflight): VerifyBlobKZGProofBatch(1024) 1.34 M blobs/s from 16 callers, 1.47 M from 24, 1.55-1.60 M from 32
#endif
    static constexpr int NLANE = MKZG_NLANE;
    Lane lanes[NLANE];
    std::atomic<int> lanes_busy{0};        // calls in flight on this context right now
    // ---- hash rounds (throughput shape of the blob verifier)
    std::mutex hash_mu;
    std::condition_variable hash_cv;
    static constexpr int HASH_STREAMS_MAX = 4;
    std::shared_ptr<HashRound> hash_forming, hash_hist[HASH_STREAMS_MAX];   // the round being filled; the last launched ones (ring by round number)
    hipStream_t hash_stream[HASH_STREAMS_MAX] = {nullptr, nullptr, nullptr, nullptr};   // rounds take them in turn: hash_in_flight() of them are used
    uint64_t hash_rounds = 0;
    std::mutex mu;                         // lane hand-over only
    std::condition_variable cv;
    uint64_t release_seq = 0;
    std::mutex init_mu;                    // lazy construction of the shared tables below
    hipStream_t init_stream = nullptr;
    kzg::HostPool* pool = nullptr;         // persistent host threads for the host steps (host_pool.hpp), shared by the lanes
    // ---- options (mkzg_ctx_set_option).  Plain ints change only while no call is in flight (the setter holds every lane); the atomic
    // ones are read by calls in flight and may change under them (each call reads them once per decision)
    int opt_c = 13;
    int opt_direct_bits = -1;              // -1: the widest direct table that leaves 40 GB of the free HBM alone (default); 0: bucket method; 12..15: that width
    int opt_max_batch = 256;
    int opt_fk20_bits = 12;                // widest FK20 direct table to try: 13 (86 GB), 12 (47 GB), 11 (25.8 GB), 8 (4.3 GB); narrower ones are the fallback
    std::atomic<int> opt_host_chunk{128};              // blobs per chunk of the host-pointer commitment pipeline (measured: tools/bench_host_path.py)
    std::atomic<int> opt_host_gather_tasks{16};        // gather tasks per chunk of the host-pointer commitment pipeline (measured, tools/sweep_host_commit.py: 3.37 ms per 256 blobs with 16, 3.55 with 4)
#ifdef MKZG_AFFINE_PAIRS
    std::atomic<int> opt_affine_pairs{0};              // A/B builds only (experiments/commit_pairs_kernels.hip): adjacent windows of a scalar added in affine coordinates first
#endif
    std::atomic<int> opt_host_fin{1};                  // host-pointer commitments of up to four blobs: the kernel leaves XYZZ sums, the calling thread inverts and compresses
    std::atomic<int> opt_host_validate{1};             // single-proof entry points: their one or two points are decoded and subgroup-tested on the calling thread (host_validate.hpp)
    std::atomic<int> opt_host_zero_copy{1};            // host-pointer commitments: the accumulate kernel reads the pinned staging area itself (1) or a device copy (0)
    // single-item calls of the reference API served in groups through the batch paths when several are in flight (host_combine.hpp)
    struct CommitCall { const uint8_t* blob; uint8_t* out48; int rc; };
    struct CellsCall { const uint8_t* blob; uint8_t* cells_out; uint8_t* proofs_out; int rc; };
    struct VerifyCall { const uint8_t* blob; const uint8_t* commitment48; const uint8_t* proof48; int rc; bool again; };
    kzg::Combiner<CommitCall> comb_commit;
    kzg::Combiner<CellsCall> comb_cells;
    kzg::Combiner<VerifyCall> comb_verify;
    std::atomic<int> opt_deep_rounds{1};               // "deep_rounds": the blob verifier's hash rounds also launch their calls' evaluations and MSMs (one launch each per round) and copy the verdicts back; 0: every call carries on in its own stream
    std::atomic<int> opt_launch_graphs{0};             // "launch_graphs": 1 replays fixed launch sequences through executable graphs (lane_graph_run), 0 (default, by measurement) launches one by one
    std::atomic<int> opt_combine{1};                   // "combine_single_calls": 1 (default) groups concurrent single-blob calls, 0 every call on its own
    std::atomic<int> opt_verify_shape{-1};             // VerifyBlobKZGProofBatch: latency shape (0), throughput shape (1), by the calls in flight (-1); api_verify.hip
    int opt_cells_chunk = 256;             // most blobs per launch set of a ComputeCellsAndKZGProofs batch (a lane's FK20 workspace grows to the pieces it has seen: 1.6 MB per blob)
    int opt_lanes = NLANE;                 // lanes calls may occupy at once (each keeps its workspaces and pinned staging until trimmed or destroyed)
    std::atomic<int> opt_fk20_segments{1}; // the FK20 products of calls of 65..192 blobs in segments (fk20_kernels.hip k_fk20_toom_seg); 0: whole products -- same bytes, for A/B and tests
    std::atomic<int> opt_host_wait{-1};    // how a call waits for the GPU: 0 the runtime's own wait (polls), 1 query + sleep, -1 by the calls in flight and the CPUs of the context's share; lane_sync (api.hip)
    int pool_node_cpus = 0;                // CPUs of the GPU's NUMA node the pool's workers are bound to (0: not bound)
    int cpu_share = 1;                     // CPUs this context may count on: the cgroup's, divided by the ranks on the node / the devices of the context
    std::atomic<int> opt_debug_serial{0};  // measurement aid: the blob verifier waits for its side stream before it launches the challenge hash (the hash kernel alone on the chip)
    std::atomic<int> opt_host_sha{-1};                 // host-pointer verifiers: challenges by SHA-NI on the host pool (1), on the device (0), auto (-1)
    std::atomic<bool> profiling{false};
    // ---- SRS
    std::vector<uint8_t> g1_monomial, g1_lagrange, g2_monomial;   // raw compressed bytes
    uint32_t* d_lagrange_aff = nullptr;    // 4096 affine rows, natural order of the JSON
    // ---- shared tables (read-only once built)
    uint32_t* d_commit_table = nullptr;    // bucket method: [nwin][4096] affine rows for commit_table_c
    std::atomic<int> commit_table_c{0};    // published (release) after the table is complete; readers acquire before touching d_commit_table
    uint32_t* d_direct = nullptr;          // direct window table
    std::atomic<int> direct_bits_built{0}; // 0 with opt_direct_bits != 0 and direct_failed: degraded to the bucket method (published after the build)
    std::atomic<bool> direct_failed{false}; // the table could not be allocated at any width: stop retrying
    uint32_t* d_roots_brp = nullptr;       // 4096 Fr rows (12 words), bit-reversed domain (api.go:132)
    uint32_t* d_inv_roots_brp = nullptr;
    G1HFixedBase gen_fixed;                // host byte-window table of the G1 generator ([e]G of the batch verifiers)
    G2Prepared g2_gen, g2_alpha;           // prepared [1]G2, [alpha]G2 (kzg.OpeningKey, internal/kzg/srs.go:10-20)
    G2Prepared g2_neg_gen, g2_alpha64;     // prepared -[1]G2, [alpha^64]G2 (kzgmulti.OpeningKey, internal/kzg_multi/kzg_verify.go:89-92)
    std::atomic<bool> twiddles_ready{false};
    uint32_t* d_tw_fwd = nullptr;          // 4096 Fr rows: w_8192^j
    uint32_t* d_tw_inv = nullptr;          // 4096 Fr rows: w_8192^-j
    uint32_t* d_g1tw_fwd = nullptr;        // 64 x 8 words: w_128^j as plain scalars
    uint32_t* d_g1tw_inv = nullptr;
    std::atomic<bool> monomial_ready{false};
    uint32_t* d_monomial_aff = nullptr;    // 4096 affine rows
    std::atomic<bool> rec_ready{false};
    uint32_t* d_pow7 = nullptr;
    uint32_t* d_ipow7 = nullptr;
    std::atomic<bool> fk_ready{false};
    uint32_t* d_fk_t_aff = nullptr;        // 8192 affine rows T[k][i]
    uint32_t* d_fk_table = nullptr;        // direct window table of the FK20 bases (fk20_table_bytes(fk20_bits_built))
    int fk20_bits_built = 0;
    // host finish timing (last verification call), microseconds
    std::atomic<double> last_host_finish_us{0.0};
};

// ---- lanes (api.hip)
// Blocks until a lane is free; the call's main stream -- the caller's (device-pointer entry points; may be the null stream) or the
// lane's own -- is made to wait for the lane's previous user.
Lane* lane_acquire(mkzg_ctx* c, bool user_stream, hipStream_t s);
// Hands the lane back; s = the call's main stream.  After an error exit every stream of the lane is synchronised first.
void lane_release(mkzg_ctx* c, Lane* L, hipStream_t s);
// all lanes at once (option changes, profile read-out, destruction): returns with every lane owned by the caller and the device idle
void lanes_quiesce(mkzg_ctx* c);
void lanes_resume(mkzg_ctx* c);
// A side stream for a call whose main stream is `main` (the lane's own, or the caller's in the device-pointer entry points) that can
// never share a hardware queue with it: HIP keeps one pool of in-order hardware queues per stream PRIORITY and binds a stream to a queue
// of its pool when it is created.  With the caller's stream and the lane's side stream in the same pool it was a matter of luck whether the
// side stream's kernels ran beside the main stream's 2.7 ms hash kernel or queued up behind it (VerifyBlobKZGProofBatch of 1024: 4.35 or
// 6.05 ms per call, fixed for the life of the process).
inline hipStream_t lane_side_stream(Lane* L, hipStream_t main) {
    if (main == L->stream) return L->side;
    int p = 0, least = 0, greatest = 0;
    if (hipStreamGetPriority(main, &p) != hipSuccess) { (void)hipGetLastError(); p = 0; }
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    return p == greatest ? L->side : L->stream;          // L->stream has the greatest priority, L->side the middle one
}

struct LaneGuard {                          // one call = one guard
    DeviceScope dev;                        // first member: the device is current before the lane is taken and until after it is handed back
    mkzg_ctx* c;
    Lane* L;
    hipStream_t s;                          // the call's main stream
    explicit LaneGuard(mkzg_ctx* c_) : dev(c_->device), c(c_), L(lane_acquire(c_, false, nullptr)), s(L->stream) {}                     // host-pointer entry points
    LaneGuard(mkzg_ctx* c_, hipStream_t user_stream) : dev(c_->device), c(c_), L(lane_acquire(c_, true, user_stream)), s(user_stream) {}   // device-pointer entry points
    ~LaneGuard() { if (L) lane_release(c, L, s); }
    LaneGuard(const LaneGuard&) = delete;
    LaneGuard& operator=(const LaneGuard&) = delete;
    int fail(int rc) { if (L && rc != MKZG_OK) L->failed = true; return rc; }    // error exit after device work was enqueued
};
// Waiting for the GPU (api.hip).  hipStreamSynchronize polls, and on this runtime hipEventSynchronize does too, even for an event made with
// hipEventBlockingSync: *measured* (thread CPU clock around the calls, 32 callers of the blob verifier on one context) 7.7 ms of CPU per
// call, 6.1 of them in the last wait, 13 of the 16 granted CPUs busy.  A call that shares the context with more calls than half the CPUs
// of the context's share records an event and asks + sleeps instead (20 us steps, then 60; the thread's timer slack lowered meanwhile):
// 3.8 CPUs busy, 1.49-1.50 M blobs/s against 1.55-1.62 M.  A lone caller keeps the polling wait: it is the lowest latency.
// Side streams of a call inherit what is pending on its main stream (the entry points are stream-ordered) through an event recorded on
// the main stream -- a marker that waits its turn in a hardware queue the stream shares with other calls' streams.  A main stream with
// nothing pending (hipStreamQuery) has nothing to hand on: no marker, the side streams start at once.  Returns whether s was idle.
bool lane_fork(Lane* L, hipStream_t s, hipStream_t side_a = nullptr, hipStream_t side_b = nullptr);
bool wait_sleeping(const mkzg_ctx* c);
int wait_event_sleeping(hipEvent_t ev);
int lane_sync(mkzg_ctx* c, Lane* L, hipStream_t s);
#define LANE_SYNC(c, L, s) do { const int ls__ = lane_sync(c, L, s); if (ls__) return ls__; } while (0)

// LAUNCH GRAPHS.  `enqueue` puts a fixed sequence of kernels and copies on `s` (no host wait, no allocation, no stream other than s)
// and returns 0 or an error.  With graphs on, the sequence is CAPTURED instead of launched, the lane's executable graph of that slot is
// updated with the capture (same topology every call; the kernels' by-value arguments -- r, the exponents' offset, the sizes -- are
// what changes) and launched: the packets of the whole sequence reach the hardware queue together instead of one API call at a time
// between other threads' launches.  OFF by default (option "launch_graphs"): *measured* (round 5, 32 callers of the blob verifier, graphs
// off / on alternating on one box each): 2.34 -> 2.52 M blobs/s on the first box (tools/profile_r05.sh), 2.26-2.31 -> 2.08 M on the
// second (tools/ab_r05_14.sh), 2.12-2.27 against 2.04-2.23 M on the third (tools/ab_r05_15.sh: 1650 launches, 32 instantiations, no
// fall-back); the headline's two launches per step 97.4 -> 95.7 k blobs/s.  No consistent gain: what binds the legs is queue time, which
// a graph launch does not shorten.  Any failure of the capture machinery falls back to launching plainly (the
// stub runtime of the sanitizer tests has no graphs at all); profiling runs plainly too (its events would be captured).
bool launch_graphs_on(const mkzg_ctx* c);
extern std::atomic<long long> g_graph_launches, g_graph_instantiations, g_graph_fallbacks;     // diagnostics ("graph_stats" options)
template <class F>
static inline int lane_graph_run(mkzg_ctx* c, Lane* L, int slot, hipStream_t s, F&& enqueue) {
    if (!launch_graphs_on(c)) return enqueue();
    if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); return enqueue(); }
    const bool hip_err_before = tl_hip_error;
    const int rc = enqueue();                                   // recorded, not run
    hipGraph_t g = nullptr;
    const hipError_t ee = hipStreamEndCapture(s, &g);
    if (rc) {                                                   // the sequence itself reported an error: nothing was put on the stream
        if (g) (void)hipGraphDestroy(g);
        (void)hipGetLastError();
        return rc;
    }
    bool ok = ee == hipSuccess && g != nullptr;
    if (ok && L->graph_exec[slot]) {
        hipGraphNode_t bad = nullptr;
        hipGraphExecUpdateResult res;
        if (hipGraphExecUpdate(L->graph_exec[slot], g, &bad, &res) != hipSuccess) {     // another topology (a size that takes other kernels): a new executable
            (void)hipGetLastError();
            (void)hipGraphExecDestroy(L->graph_exec[slot]);
            L->graph_exec[slot] = nullptr;
        }
    }
    if (ok && !L->graph_exec[slot]) { ok = hipGraphInstantiate(&L->graph_exec[slot], g, nullptr, nullptr, 0) == hipSuccess; g_graph_instantiations++; }
    if (g) (void)hipGraphDestroy(g);
    if (ok) ok = hipGraphLaunch(L->graph_exec[slot], s) == hipSuccess;
    if (ok) { g_graph_launches++; return MKZG_OK; }
    g_graph_fallbacks++;
    (void)hipGetLastError();
    if (L->graph_exec[slot]) { (void)hipGraphExecDestroy(L->graph_exec[slot]); L->graph_exec[slot] = nullptr; }
    tl_hip_error = hip_err_before;
    return enqueue();                                           // plainly
}
#define KT(kind, s, call)                                         \
    do {                                                          \
        const bool kt_on__ = c->profiling.load(std::memory_order_relaxed); \
        if (kt_on__) L->timer.begin(kind, s);                     \
        call;                                                     \
        if (kt_on__) L->timer.end(kind, s);                       \
    } while (0)

// The verifiers' last step is two Miller loops + one final exponentiation on the host (host_pairing.hpp).  A call that is alone on the
// context waits for exactly that, and the pool's other threads are idle: it cuts every loop into three (two) segments on a thread each
// (0.51 -> ~0.40 ms for the whole check on an EPYC 9575F); with more calls in flight the threads are needed elsewhere: one loop per thread.
static inline int miller_segments(const mkzg_ctx* c) {
    static const int forced = [] { const char* e = getenv("MKZG_MILLER_SEGMENTS"); return e ? atoi(e) : 0; }();      // A/B aid
    if (forced) return forced;
    const int busy = c->lanes_busy.load(std::memory_order_relaxed), threads = c->pool ? c->pool->size() + 1 : 1;
    if (busy <= 1 && threads >= 6) return 3;
    if (busy <= 2 && threads >= 4) return 2;
    return 1;
}

// ---- shared tables, built once (api.hip, api_cells.hip)
int ensure_commit_table(mkzg_ctx* c);
int ensure_direct(mkzg_ctx* c);             // MKZG_OK also when the table cannot be had: direct_bits_built then says what is in use
int ensure_twiddles(mkzg_ctx* c);
int ensure_monomial(mkzg_ctx* c);
int ensure_fk20_tables(mkzg_ctx* c);
int ensure_rec_tables(mkzg_ctx* c);
// ---- per-lane workspaces
int ensure_staging(Lane* L, size_t n_items, size_t in_bytes_per_item);
int ensure_verify_ws(Lane* L, size_t n);
size_t verdict_block_bytes(size_t cap, size_t aux_items);      // api_verify.hip: bytes of Lane::d_v_block that come back in one copy
int ensure_quot_ws(Lane* L, size_t n, size_t n_quot);
int ensure_cells_ws(Lane* L, size_t n);
int ensure_fk20_ws(Lane* L, size_t n);
void free_lane(Lane* L);
// commitments of n polynomials in Lagrange form (api.hip); the caller holds lane L and runs on stream s
int commit_scalars_device(mkzg_ctx* c, Lane* L, const uint8_t* d_scalars, size_t stride_bytes, int input_kind, size_t n, uint8_t* d_out48n, int* d_status, hipStream_t s,
                          uint32_t* d_raw = nullptr, bool* raw_used = nullptr);
void commit_host_finish(const uint32_t* w56, uint8_t out48[48]);      // api.hip: XYZZ limb words of a sum -> 48 compressed bytes on the calling thread
// device a pointer lives on (-1: not a device pointer)
int pointer_device(const void* p);
mkzg_ctx* ctx_for_pointer(mkzg_ctx* c, const void* p);
