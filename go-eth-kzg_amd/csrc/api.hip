// api.hip -- the C ABI of libmi355kzg.so (include/mi355kzg.h): context construction, the lanes that let calls on a
// shared context overlap, and the host-side mirror of go-eth-kzg's Context methods for commitments.  Host logic only:
// every field/group operation happens in HIP kernels (msm_kernels.hip); there is no CPU fallback.
//
// Reference being mirrored (paths relative to crate-crypto/go-eth-kzg):
//   NewContext4096            api.go:90-149   (+ trusted_setup.go:90-192 JSON / point decoding)
//   Context.BlobToKZGCommitment  prove.go:13-33
//   kzg.CommitKey.Commit      internal/kzg/srs.go:56-62
//   multiexp.MultiExpG1       internal/multiexp/multiexp.go:20-50
#include <functional>
#include <thread>
#include <sys/prctl.h>
#include <ctype.h>
#include <sched.h>
#include <time.h>
#include "ctx.h"
#include "api_multi.h"

// ---------------------------------------------------------------------------------------------
// trusted-setup JSON (reference trusted_setup.go:23-27): {"g1_monomial":[...], "g1_lagrange":[...],
// "g2_monomial":[...]} with 0x-prefixed hex strings.  Minimal scanner, no allocation surprises.
// ---------------------------------------------------------------------------------------------
static int hexval(char c) {
    if (c >= '0' && c <= '9') return c - '0';
    if (c >= 'a' && c <= 'f') return c - 'a' + 10;
    if (c >= 'A' && c <= 'F') return c - 'A' + 10;
    return -1;
}
static bool parse_hex_array(const char* js, size_t len, const char* key, size_t item_bytes, std::vector<uint8_t>& out) {
    std::string k = std::string("\"") + key + "\"";
    const char* end = js + len;
    const char* p = std::search(js, end, k.begin(), k.end());
    if (p == end) return false;
    p += k.size();
    while (p < end && *p != '[') p++;
    if (p == end) return false;
    p++;
    out.clear();
    while (p < end) {
        while (p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t' || *p == ',')) p++;
        if (p == end) return false;
        if (*p == ']') return true;
        if (*p != '"') return false;
        p++;
        if (end - p < 2 || p[0] != '0' || (p[1] != 'x' && p[1] != 'X')) return false;   // trim0xPrefix panics otherwise
        p += 2;
        size_t n = 0;
        while (p + 1 < end && *p != '"') {
            int h = hexval(p[0]), l = hexval(p[1]);
            if (h < 0 || l < 0) return false;
            out.push_back((uint8_t)(h * 16 + l));
            p += 2;
            n++;
        }
        if (p >= end || *p != '"' || n != item_bytes) return false;
        p++;
    }
    return false;
}

// ---------------------------------------------------------------------------------------------
// lanes
// ---------------------------------------------------------------------------------------------
thread_local bool tl_hip_error = false;
static int lane_create(Lane* L, int id) {
    L->id = id;
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    // three priorities -> three pools of hardware queues: main streams of the lanes do not share queues with side streams
    HIPCHK(hipStreamCreateWithPriority(&L->stream, hipStreamNonBlocking, greatest));
    HIPCHK(hipStreamCreateWithPriority(&L->side, hipStreamNonBlocking, (least + greatest) / 2));
    HIPCHK(hipStreamCreateWithPriority(&L->side2, hipStreamNonBlocking, least));
    for (auto& a : L->aux) HIPCHK(hipStreamCreateWithPriority(&a, hipStreamNonBlocking, (least + greatest) / 2));
    for (auto& e : L->ev_cs) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&L->ev_fork, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&L->ev_join, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&L->ev_decoded, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&L->done, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&L->ev_block, hipEventDisableTiming | hipEventBlockingSync));
    for (auto& e : L->ev_chunk) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return MKZG_OK;
}
static void free_lane_plan(Lane* L) {
    dev_free(L->plan.d_entries); dev_free(L->plan.d_task_off); dev_free(L->plan.d_tasks); dev_free(L->plan.d_ntasks);
    dev_free(L->plan.d_parts); dev_free(L->plan.d_partials);
    L->plan = CommitPlan();
    L->plan_c = 0; L->plan_batch = 0;
}
static void free_lane_direct(Lane* L) {
    dev_free(L->dslot.d_parts); dev_free(L->dslot.d_bad);
#ifdef MKZG_AFFINE_PAIRS
    dev_free(L->dslot.d_prefix);
#endif
    L->dslot_batch = 0;
    L->dslot_records = 0;
}
void free_lane(Lane* L) {
    free_lane_plan(L);
    free_lane_direct(L);
    if (L->h_stage) { (void)hipHostFree(L->h_stage); L->h_stage = nullptr; }
    if (L->h_out) { (void)hipHostFree(L->h_out); L->h_out = nullptr; }
    if (L->h_v) { (void)hipHostFree(L->h_v); L->h_v = nullptr; }
    L->h_stage_bytes = 0; L->h_out_items = 0; L->h_v_bytes = 0;
    dev_free(L->d_blobs); dev_free(L->d_out); dev_free(L->d_status);
    L->d_blobs_bytes = 0; L->d_out_items = 0;
    dev_free(L->d_v_raw); dev_free(L->d_v_aff); dev_free(L->d_v_block); dev_free(L->d_v_z); dev_free(L->d_v_y); dev_free(L->d_v_zrows); dev_free(L->d_v_s1); dev_free(L->d_v_s3);
    L->d_v_st = nullptr; L->d_v_stx = nullptr; L->d_v_ry = nullptr; L->d_v_misc = nullptr; L->d_v_parts = nullptr;
    dev_free(L->d_v_coeffs);
    L->v_cap = 0; L->v_coeff_cap = 0;
    dev_free(L->d_quot); dev_free(L->d_q_z); dev_free(L->d_q_y); dev_free(L->d_q_st);
    L->q_cap = 0; L->quot_cap = 0;
    dev_free(L->d_coeffs); dev_free(L->d_cells); dev_free(L->d_cell_st);
    L->cell_cap = 0;
    dev_free(L->d_cv_cells); dev_free(L->d_cv_in); dev_free(L->d_cv_rp); dev_free(L->d_cv_weights); dev_free(L->d_cv_partials);
    dev_free(L->d_cv_interp);
    L->cv_cap = 0;
    L->cv_cells_cap = 0;
    dev_free(L->d_rec_scratch); dev_free(L->d_rec_pos);
    L->rec_cap = 0;
    dev_free(L->d_fk_scalars); dev_free(L->d_fk_results); dev_free(L->d_fk_parts); dev_free(L->d_fk_proofs); dev_free(L->d_fk_seg);
    L->fk_seg_bytes = 0;
    L->fk_cap = 0;
}
static void lane_destroy(Lane* L) {
    free_lane(L);
    if (L->stream) (void)hipStreamDestroy(L->stream);
    if (L->side) (void)hipStreamDestroy(L->side);
    if (L->side2) (void)hipStreamDestroy(L->side2);
    for (auto& a : L->aux) if (a) (void)hipStreamDestroy(a);
    for (auto& e : L->ev_cs) if (e) (void)hipEventDestroy(e);
    hipEvent_t evs[] = {L->ev_fork, L->ev_join, L->ev_decoded, L->done, L->ev_block};
    for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
    for (auto& e : L->ev_chunk) if (e) (void)hipEventDestroy(e);
}

// Which free lane: (0) for a device-pointer call, the lane whose last call ran on the SAME stream (stream order serialises the two
// calls anyway; *measured* with three ComputeCellsAndKZGProofs calls of 128-256 blobs issued back to back on one stream: the second
// and third used to take fresh lanes, grow their workspaces -- hipFree waits for the device -- and return after 16-28 ms, and
// sometimes cost 5 ms per call; now they return after 0.1 ms each and hold one set of workspaces); (1) the most recently used one whose previous call has finished on the GPU -- its workspaces are allocated and
// warm, so a lone synchronous caller keeps to one lane; (2) else a lane that has been used before, the one released longest ago (its
// work ends soonest: asynchronous callers issuing faster than the GPU retires alternate between lanes and overlap, as the
// two-workspace scheme of the commitment path always did); (3) a lane never used -- cold: its first call allocates -- only when
// every warm lane is taken.
Lane* lane_acquire(mkzg_ctx* c, bool user_stream, hipStream_t s) {
    Lane* L = nullptr;
    {
        std::unique_lock<std::mutex> lk(c->mu);
        for (;;) {
            Lane* idle = nullptr;
            Lane* warm = nullptr;
            Lane* cold = nullptr;
            Lane* same = nullptr;              // free, and its last call ran on this caller's stream
            int active = 0;
            for (int i = 0; i < c->opt_lanes; i++) {
                Lane* q = &c->lanes[i];
                if (q->busy) { active++; continue; }
                if (!q->warm) { if (!cold) cold = q; continue; }
                if (!warm || q->released_at < warm->released_at) warm = q;
                if (user_stream && q->last_stream == s && (!same || q->released_at > same->released_at)) same = q;
                const bool finished = !q->used || hipEventQuery(q->done) == hipSuccess;
                if (!finished) active++;
                if (finished && (!idle || q->released_at > idle->released_at)) idle = q;
            }
            // (0) asynchronous calls issued back to back on ONE stream run one after the other whatever lane they take: the lane of the
            // previous call on that stream has the workspaces already, and stream order makes its reuse safe -- a fresh lane per call in
            // flight would allocate its own (and hipFree of an outgrown block waits for the whole device)
            L = same ? same : idle ? idle : (cold && user_stream ? cold : (warm ? warm : cold));
            if (L) { L->others_active = active; break; }
            c->cv.wait(lk);
        }
        (void)hipGetLastError();           // hipEventQuery reports hipErrorNotReady through the sticky error too
        L->busy = true;
        L->warm = true;
        c->lanes_busy.fetch_add(1, std::memory_order_relaxed);
    }
    L->failed = false;
    L->drained = false;
    tl_hip_error = false;
    if (L->used) (void)hipStreamWaitEvent(user_stream ? s : L->stream, L->done, 0);      // the previous owner may have returned before the GPU was done
    return L;
}
void lane_release(mkzg_ctx* c, Lane* L, hipStream_t s) {
    if (L->failed || tl_hip_error) {       // error exit with work possibly in flight on any of the lane's streams
        tl_hip_error = false;
        (void)hipStreamSynchronize(s);
        (void)hipStreamSynchronize(L->stream);
        (void)hipStreamSynchronize(L->side);
        (void)hipStreamSynchronize(L->side2);
        for (auto& a : L->aux) (void)hipStreamSynchronize(a);
        (void)hipGetLastError();
    }
    // a call that has waited for its last device work on the host hands on an idle lane: no event, and the next acquire does not have to
    // guess from hipEventQuery (an event recorded microseconds ago reads "not ready", and the caller would be sent to a cold lane --
    // a synchronous caller looping over one context walked through all the lanes that way, allocating workspaces in each)
    L->last_stream = s;
    if (L->drained) {
        L->used = false;
    } else {
        (void)hipEventRecord(L->done, s);
        L->used = true;
    }
    {
        std::lock_guard<std::mutex> g(c->mu);
        L->busy = false;
        L->released_at = ++c->release_seq;
        c->lanes_busy.fetch_sub(1, std::memory_order_relaxed);
    }
    c->cv.notify_all();
}
bool lane_fork(Lane* L, hipStream_t s, hipStream_t side_a, hipStream_t side_b) {
    if (hipStreamQuery(s) == hipSuccess) return true;
    (void)hipGetLastError();                   // hipErrorNotReady is reported through the sticky error too
    if (hipEventRecord(L->ev_fork, s) != hipSuccess) { tl_hip_error = true; return false; }
    if (side_a && side_a != s && hipStreamWaitEvent(side_a, L->ev_fork, 0) != hipSuccess) tl_hip_error = true;
    if (side_b && side_b != s && hipStreamWaitEvent(side_b, L->ev_fork, 0) != hipSuccess) tl_hip_error = true;
    return false;
}
std::atomic<long long> g_graph_launches{0}, g_graph_instantiations{0}, g_graph_fallbacks{0};
// MKZG_TP_GRAPH=0 / 1 in the environment overrides the option (A/B aid)
bool launch_graphs_on(const mkzg_ctx* c) {
    static const int env = getenv("MKZG_TP_GRAPH") ? (atoi(getenv("MKZG_TP_GRAPH")) != 0 ? 1 : 0) : -1;
    const bool on = env >= 0 ? env == 1 : c->opt_launch_graphs.load(std::memory_order_relaxed) != 0;
    return on && !c->profiling.load(std::memory_order_relaxed);
}
bool wait_sleeping(const mkzg_ctx* c) {
    const int o = c->opt_host_wait.load(std::memory_order_relaxed);
    if (o >= 0) return o == 1;
    return c->lanes_busy.load(std::memory_order_relaxed) > std::max(1, c->cpu_share / 2);
}
int wait_event_sleeping(hipEvent_t ev) {
    // the default timer slack (50 us) would more than double every step; the caller's value is put back before returning
    const int slack = prctl(PR_GET_TIMERSLACK, 0, 0, 0, 0);
    if (slack > 1000) (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0);
    int rc = MKZG_OK;
    for (int i = 0;; i++) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) break;
        (void)hipGetLastError();               // hipErrorNotReady is reported through the sticky error too
        if (e != hipErrorNotReady) { rc = MKZG_ERR_HIP; tl_hip_error = true; break; }
        timespec ts{0, i < 8 ? 20000 : 60000};
        nanosleep(&ts, nullptr);
    }
    if (slack > 1000) (void)prctl(PR_SET_TIMERSLACK, (unsigned long)slack, 0, 0, 0);
    return rc;
}
int lane_sync(mkzg_ctx* c, Lane* L, hipStream_t s) {
    if (wait_sleeping(c)) {
        HIPCHK(hipEventRecord(L->ev_block, s));
        return wait_event_sleeping(L->ev_block);
    }
    HIPCHK(hipStreamSynchronize(s));
    return MKZG_OK;
}
void lanes_quiesce(mkzg_ctx* c) {
    {
        std::unique_lock<std::mutex> lk(c->mu);
        for (int i = 0; i < mkzg_ctx::NLANE; i++) {
            c->cv.wait(lk, [&] { return !c->lanes[i].busy; });
            c->lanes[i].busy = true;
        }
    }
    DeviceScope dev(c->device);
    (void)hipDeviceSynchronize();
}
void lanes_resume(mkzg_ctx* c) {
    {
        std::lock_guard<std::mutex> g(c->mu);
        for (int i = 0; i < mkzg_ctx::NLANE; i++) { c->lanes[i].busy = false; c->lanes[i].used = false; }
    }
    c->cv.notify_all();
}

int pointer_device(const void* p) {
    hipPointerAttribute_t a;
    if (!p || hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return -1; }
    if (a.type != hipMemoryTypeDevice) return -1;
    return a.device;
}

// ---------------------------------------------------------------------------------------------
// shared tables
// ---------------------------------------------------------------------------------------------
// bucket method: T[j][i] = 2^(c j) L_brp[i]   (api.go:131: the commit key is the bit-reversed Lagrange SRS)
int ensure_commit_table(mkzg_ctx* c) {
    std::lock_guard<std::mutex> g(c->init_mu);
    if (c->d_commit_table && c->commit_table_c == c->opt_c) return MKZG_OK;
    DeviceScope dev(c->device);
    CommitPlanSizes sz;
    commit_plan_sizes(c->opt_c, 1, &sz);
    c->commit_table_c = 0;
    int rc = dev_realloc(c->d_commit_table, sz.table_bytes);
    if (rc) return rc;
    build_table_device(c->opt_c, c->d_lagrange_aff, c->d_commit_table, 4096, 12, c->init_stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->init_stream));
    c->commit_table_c = c->opt_c;
    return MKZG_OK;
}
// Direct window table.  The table is sized for an otherwise empty 288 GB part; a drop-in must degrade, not fail: if the width asked
// for cannot be allocated the next narrower one is tried, and when none fits the context falls back to the bucket method
// (direct_bits_built == 0, direct_failed) -- same bytes out either way.
// What the auto-sized table (commit_direct_bits -1) leaves free: the widest FK20 table this context will try, unless one is built
// already (whichever table comes first must not push the other one down a width: with both at their defaults -- 146 GB and the 47 GB
// 12-bit FK20 table -- an empty 288 GB device holds both), the lanes' workspaces and pinned-side device buffers (0.1-0.6 GB per lane that
// has run a call; a third of a gigabyte each on average) and 4 GB of slack for the caller's own allocations.
static size_t direct_headroom(const mkzg_ctx* c) {
    size_t h = (size_t)4 << 30;
    h += (size_t)c->opt_lanes * ((size_t)340 << 20);
    if (!c->fk_ready.load(std::memory_order_acquire)) h += fk20_table_bytes(std::min(13, std::max(8, c->opt_fk20_bits)));
    return h;
}
int ensure_direct(mkzg_ctx* c) {
    std::lock_guard<std::mutex> g(c->init_mu);
    if (c->opt_direct_bits == 0 || c->direct_failed) return MKZG_OK;
    if (c->d_direct && c->direct_bits_built) return MKZG_OK;
    DeviceScope dev(c->device);
    int want = c->opt_direct_bits;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
    if (want < 0) want = 15;                                  // auto: the widest table that leaves the headroom free
    for (int cb = want; cb >= 12; cb--) {
        const size_t bytes = direct_table_bytes(cb);
        if (c->opt_direct_bits < 0 && free_b && bytes + direct_headroom(c) > free_b) continue;
        if (free_b && bytes + ((size_t)1 << 30) > free_b) continue;               // cannot fit: do not even ask
        const int nwin = (255 + cb - 1) / cb;
        uint32_t* tmp = nullptr;
        if (hipMalloc(&c->d_direct, bytes) != hipSuccess) { (void)hipGetLastError(); c->d_direct = nullptr; continue; }
        if (hipMalloc(&tmp, (size_t)nwin * 4096 * 128) != hipSuccess) { (void)hipGetLastError(); dev_free(c->d_direct); continue; }
        build_table_any_device(cb, c->d_lagrange_aff, tmp, 4096, 12, c->init_stream);
        build_direct_device(cb, tmp, c->d_direct, c->init_stream);
        const bool ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(c->init_stream) == hipSuccess;
        (void)hipFree(tmp);
        if (!ok) { dev_free(c->d_direct); return MKZG_ERR_HIP; }
        c->direct_bits_built = cb;
        if (cb != c->opt_direct_bits && c->opt_direct_bits > 0)
            fprintf(stderr, "[mi355kzg] direct window table: %d bits requested, %d bits fit in free HBM\n", c->opt_direct_bits, cb);
        return MKZG_OK;
    }
    fprintf(stderr, "[mi355kzg] direct window table does not fit in free HBM (%.1f GB free): using the bucket method\n", (double)free_b / 1e9);
    c->direct_failed = true;
    c->direct_bits_built = 0;
    return MKZG_OK;
}
static int ensure_lane_plan(mkzg_ctx* c, Lane* L) {
    if (L->plan_c == c->opt_c && L->plan_batch >= c->opt_max_batch && L->plan.d_entries) { L->plan.d_table = c->d_commit_table; return MKZG_OK; }
    free_lane_plan(L);
    CommitPlanSizes sz;
    commit_plan_sizes(c->opt_c, c->opt_max_batch, &sz);
    CommitPlan& p = L->plan;
    p.c = c->opt_c;
    p.max_batch = c->opt_max_batch;
    p.d_table = c->d_commit_table;
    HIPCHK(hipMalloc(&p.d_entries, sz.entries_bytes));
    HIPCHK(hipMalloc(&p.d_task_off, sz.task_off_bytes));
    HIPCHK(hipMalloc(&p.d_tasks, sz.tasks_bytes));
    HIPCHK(hipMalloc(&p.d_ntasks, sz.ntasks_bytes));
    HIPCHK(hipMalloc(&p.d_parts, sz.parts_bytes));
    HIPCHK(hipMalloc(&p.d_partials, sz.partials_bytes));
    L->plan_c = c->opt_c;
    L->plan_batch = c->opt_max_batch;
    return MKZG_OK;
}
// The lane accumulators of the direct mode: 512 records per blob of max_batch to begin with (eight scalars per lane at a full batch, and
// every few-blob shape); a call whose shape needs more -- one scalar per lane from 129 blobs on: 4096 records per blob -- grows the
// buffer once (hipFree waits for the lane's earlier, possibly still running, calls).
static int ensure_lane_direct(mkzg_ctx* c, Lane* L, size_t need_records = 0) {
    const size_t base = (size_t)c->opt_max_batch * 512;
#ifdef MKZG_AFFINE_PAIRS
    const bool pairs = c->opt_affine_pairs.load(std::memory_order_relaxed) != 0 && c->direct_bits_built == 15;
    if (L->dslot_batch >= c->opt_max_batch && L->dslot.d_parts && L->dslot_records >= std::max(base, need_records) && pairs == (L->dslot.d_prefix != nullptr)) return MKZG_OK;
#else
    if (L->dslot_batch >= c->opt_max_batch && L->dslot.d_parts && L->dslot_records >= std::max(base, need_records)) return MKZG_OK;
#endif
    const size_t records = std::max(std::max(base, need_records), L->dslot_batch >= c->opt_max_batch ? L->dslot_records : (size_t)0);
    free_lane_direct(L);
    HIPCHK(hipMalloc(&L->dslot.d_parts, records * 256));
    HIPCHK(hipMalloc(&L->dslot.d_bad, direct_bad_bytes(c->opt_max_batch)));
#ifdef MKZG_AFFINE_PAIRS
    if (pairs && hipMalloc(&L->dslot.d_prefix, direct_prefix_bytes(c->opt_max_batch)) != hipSuccess) {
        (void)hipGetLastError();                 // no room for the pair kernel's scratch (6 MB per blob of max_batch): the plain accumulate kernel runs
        L->dslot.d_prefix = nullptr;
    }
#endif
    L->dslot_batch = c->opt_max_batch;
    L->dslot_records = records;
    return MKZG_OK;
}

// Commitments of n polynomials in Lagrange form against the bit-reversed Lagrange SRS, with the method the context is
// configured for (direct window table or buckets).  input_kind 0: 32-byte big-endian scalars (blobs, canonicity checked);
// 1: 12-word Montgomery Fr rows (quotient polynomials of the proof path).  The caller holds lane L; pieces of max_batch items
// run back to back on s through the lane's one workspace.  ws_off: first workspace item to use (chunks of one call that run on
// different streams take disjoint item ranges; n + ws_off <= max_batch then).
static int commit_piece(mkzg_ctx* c, Lane* L, const uint8_t* d_scalars, size_t stride_bytes, int input_kind, int n, int ws_off, uint8_t* d_out48n, int* d_status, hipStream_t s,
                        uint32_t* d_raw = nullptr) {
    KernelTimer* kt = c->profiling.load(std::memory_order_relaxed) ? &L->timer : nullptr;
    if (c->direct_bits_built) {
        DirectSlot slot = L->dslot;
        if (!ws_off) direct_pick_shape(n, L->dslot_batch, &slot);        // a call of few blobs spreads each over more lanes
        if ((size_t)(n + ws_off) * (4096 / slot.k) * slot.wsplit > L->dslot_records) {
            int rc = ensure_lane_direct(c, L, (size_t)L->dslot_batch * 4096);
            if (rc) return rc;
            slot.d_parts = L->dslot.d_parts;
            slot.d_bad = L->dslot.d_bad;
        }
        slot.fin_quad = L->others_active == 0;                          // nobody else on the GPU: the short finish; else the cheap one (it hides behind the next call)
        const DirectSlot ws = direct_slot_offset(slot, ws_off);
        commit_direct_device(c->direct_bits_built, c->d_direct, ws, d_scalars, stride_bytes, input_kind, n, d_out48n, d_status, s, kt, d_raw);
    } else {
        const CommitPlan pl = commit_plan_offset(L->plan, ws_off);
        commit_batch_device(pl, d_scalars, stride_bytes, input_kind, n, d_out48n, d_status, s, kt);
    }
    return MKZG_OK;
}
static int commit_prepare(mkzg_ctx* c, Lane* L) {
    if (c->opt_direct_bits && !c->direct_bits_built && !c->direct_failed) {
        int rc = ensure_direct(c);
        if (rc) return rc;
    }
    if (c->direct_bits_built) return ensure_lane_direct(c, L);
    if (c->commit_table_c.load(std::memory_order_acquire) != c->opt_c) {      // published after the table is complete; 0 until then
        int rc = ensure_commit_table(c);
        if (rc) return rc;
    }
    return ensure_lane_plan(c, L);
}
// d_raw (a request; *raw_used says whether it was granted): the sums as XYZZ limb words, 64 words per item, for commit_host_finish instead of compressed bytes in d_out48n
int commit_scalars_device(mkzg_ctx* c, Lane* L, const uint8_t* d_scalars, size_t stride_bytes, int input_kind, size_t n, uint8_t* d_out48n, int* d_status, hipStream_t s,
                          uint32_t* d_raw, bool* raw_used) {
    int rc = commit_prepare(c, L);
    if (rc) return rc;
    // decided HERE, behind commit_prepare (the table's state is read under the ordering its builder's mutex gives): XYZZ words instead of
    // compressed bytes only through the direct table, for one piece, with option host_fin on
    if (d_raw && !(c->direct_bits_built && n <= (size_t)c->opt_max_batch && c->opt_host_fin.load(std::memory_order_relaxed) != 0)) d_raw = nullptr;
    if (raw_used) *raw_used = d_raw != nullptr;
    if (c->direct_bits_built) {
        // whatever the pieces' shapes will need of the lane's workspace is allocated HERE: the launches below go out as one launch graph
        // (ctx.h lane_graph_run), and a capture must not meet an allocation
        for (size_t done = 0; done < n; done += (size_t)c->opt_max_batch) {
            const int chunk = (int)std::min<size_t>(n - done, (size_t)c->opt_max_batch);
            DirectSlot slot = L->dslot;
            direct_pick_shape(chunk, L->dslot_batch, &slot);
            if ((size_t)chunk * (4096 / slot.k) * slot.wsplit > L->dslot_records) {
                rc = ensure_lane_direct(c, L, (size_t)L->dslot_batch * 4096);
                if (rc) return rc;
            }
        }
    }
    rc = lane_graph_run(c, L, input_kind ? 2 : 1, s, [&]() -> int {
        size_t done = 0;
        while (done < n) {
            const int chunk = (int)std::min<size_t>(n - done, (size_t)c->opt_max_batch);
            const int prc = commit_piece(c, L, d_scalars + done * stride_bytes, stride_bytes, input_kind, chunk, 0, d_out48n + done * 48, d_status + done, s, d_raw);
            if (prc) return prc;
            done += chunk;
        }
        return MKZG_OK;
    });
    if (rc) return rc;
    HIPCHK(hipGetLastError());
    return MKZG_OK;
}

int ensure_staging(Lane* L, size_t n_items, size_t in_bytes_per_item) {
    const size_t need = n_items * in_bytes_per_item;
    if (L->h_stage_bytes < need) {
        L->h_stage_bytes = 0;
        int rc = pinned_realloc(L->h_stage, need);
        if (rc) return rc;
        L->h_stage_bytes = need;
    }
    if (L->d_blobs_bytes < need) {
        L->d_blobs_bytes = 0;
        int rc = dev_realloc(L->d_blobs, need);
        if (rc) return rc;
        L->d_blobs_bytes = need;
    }
    n_items = std::max<size_t>(n_items, 16);                  // the few-blob entry points park XYZZ sums (256 B per item) where the compressed points go
    if (L->d_out_items < n_items) {
        L->d_out_items = 0;
        int rc = dev_realloc(L->d_out, n_items * 48);
        if (!rc) rc = dev_realloc(L->d_status, n_items * sizeof(int));
        if (rc) return rc;
        L->d_out_items = n_items;
    }
    if (L->h_out_items < n_items) {
        L->h_out_items = 0;
        int rc = pinned_realloc(L->h_out, n_items * 52 + 256);          // 48 B + status per item; the single-proof calls park a few more words
        if (rc) return rc;
        L->h_out_items = n_items;
    }
    return MKZG_OK;
}

// XYZZ limb words of a sum (as k_commit_direct_fin leaves them with `raw`) -> the 48 compressed bytes: g1.hpp's to_affine (one divsteps
// inversion) and the zcash compression (serialization.go:98-100), the device's own templates compiled for the host
constexpr size_t HOST_FIN_MAX = 4;
void commit_host_finish(const uint32_t* w56, uint8_t out48[48]) {
    G1X acc;
    for (int i = 0; i < 14; i++) { acc.x.l[i] = w56[i]; acc.y.l[i] = w56[14 + i]; acc.zz.l[i] = w56[28 + i]; acc.zzz.l[i] = w56[42 + i]; }
    G1Aff a;
    g1x_to_affine(a, acc);
    uint32_t w12[12];
    g1_compress_words(w12, a);
    for (int i = 0; i < 12; i++) {
        const uint32_t v = __builtin_bswap32(w12[i]);
        memcpy(out48 + 4 * i, &v, 4);
    }
}
// Host-pointer commitments, one piece of at most max_batch blobs.  The blobs (Go's []*Blob: scattered, pageable) are gathered into the
// lane's pinned staging area by the host pool, which runs AHEAD of the calling thread: the gather tasks of all chunks are queued at
// once, the caller waits for chunk k's tasks only and issues chunk k's device work while chunks k + 1.. are still being gathered.
//   direct table : every chunk gets its own accumulate launch on one of four compute streams -- the accumulate kernels are grids of
//                  single-wave workgroups, so the chunks run side by side as they arrive (four 64-blob chunks = the 2048 wavefronts
//                  of one 256-blob launch), each in its own item range of the lane's workspace; the first chunk is a small one so
//                  that the GPU starts early, and ONE finish launch follows the last chunk.  With host_zero_copy the kernel reads its
//                  scalars straight from the pinned staging area (every wavefront fetches 2 KB per 17 table additions: the PCIe link is
//                  asked for ~25 GB/s spread over the kernel's duration), which takes the explicit copy off the critical path;
//                  otherwise each chunk is copied on the lane's copy stream and its kernel waits for that copy.
//   bucket method: its four kernels per launch are latency-bound on small batches, so ONE launch over the piece follows the last copy.
// One synchronisation at the end.
static int commit_host_piece(mkzg_ctx* c, Lane* L, const uint8_t* const* blobs, size_t n, uint8_t* out48n, int* st_out) {
    int rc = ensure_staging(L, (size_t)c->opt_max_batch, MKZG_BYTES_PER_BLOB);
    if (!rc) rc = commit_prepare(c, L);
    if (rc) return rc;
    const bool direct = c->direct_bits_built != 0, zc = direct && c->opt_host_zero_copy != 0;
    DirectSlot dslot = L->dslot;
    if (direct) {
        direct_pick_shape((int)n, L->dslot_batch, &dslot);
        if (dslot.serial) dslot.k = 8;                                  // the chunks' launches fill the chip together: eight scalars per lane
        dslot.fin_quad = L->others_active == 0;
    }
    // A FEW blobs (the single-blob entry point, small groups of the call combiner): the caller waits for a one-wavefront finish kernel
    // whose inversion + compression are an ~90 us dependent chain; the same conversion takes the calling thread ~8 us per blob (the limb
    // code of g1.hpp / mont29.hpp compiles for the host), so the kernel leaves the XYZZ sums and the host finishes them.
    const bool host_fin = direct && n <= HOST_FIN_MAX && c->opt_host_fin.load(std::memory_order_relaxed) != 0 && n * 256 <= L->d_out_items * 48 && n * 256 <= L->h_out_items * 48;
    KernelTimer* kt = c->profiling.load(std::memory_order_relaxed) ? &L->timer : nullptr;
    const size_t CH = (size_t)std::max(1, c->opt_host_chunk.load(std::memory_order_relaxed));
    hipStream_t copy = L->side, cs[4] = {L->stream, L->side2, L->aux[0], L->aux[1]};
    bool cs_used[4] = {true, false, false, false};
    // chunk plan
    struct Chunk { size_t lo, m; };
    std::vector<Chunk> plan;
    for (size_t lo = 0; lo < n;) {
        size_t m = std::min(n - lo, (plan.empty() && direct && CH >= 32) ? CH / 2 : CH);
        if (n - lo - m < CH / 2 + 1 && n - lo <= CH + CH / 2) m = n - lo;          // no tiny tail chunk: the last one takes up to 1.5 chunks
        plan.push_back({lo, m});
        lo += m;
    }
    // gather tasks: TPC per chunk, queued in chunk order
    const int TPC = n >= 8 ? std::max(1, c->opt_host_gather_tasks.load(std::memory_order_relaxed)) : 1;
    const int ntask = (int)plan.size() * TPC;
    std::vector<Countdown> left(plan.size());
    for (auto& a : left) a.set(TPC);
    Countdown all_left(ntask);
    const std::function<void(int)> gather = [&](int t) {
        const Chunk& ch = plan[(size_t)(t / TPC)];
        const int q = t % TPC;
        const size_t per = (ch.m + TPC - 1) / TPC;
        for (size_t i = ch.lo + (size_t)q * per; i < std::min(ch.lo + ch.m, ch.lo + (size_t)(q + 1) * per); i++) copy_to_staging(L->h_stage + i * MKZG_BYTES_PER_BLOB, blobs[i], MKZG_BYTES_PER_BLOB);
        left[(size_t)(t / TPC)].arrive();
    };
    c->pool->run_async(ntask, gather, &all_left);
    // every exit path below must wait for the gather tasks: they reference this frame
    auto drain = [&] { c->pool->wait(all_left); };
    rc = [&]() -> int {
        const bool idle = lane_fork(L, L->stream, zc ? nullptr : copy);      // every stream of the call inherits the lane hand-over
        for (size_t k = 0; k < plan.size(); k++) {
            const size_t lo = plan[k].lo, m = plan[k].m;
            c->pool->wait(left[k]);              // runs queued gather tasks itself while there are any
            if (!zc) HIPCHK(hipMemcpyAsync(L->d_blobs + lo * MKZG_BYTES_PER_BLOB, L->h_stage + lo * MKZG_BYTES_PER_BLOB, m * MKZG_BYTES_PER_BLOB, hipMemcpyHostToDevice, copy));
            if (direct) {
                const int q = (int)(k & 3);
                hipStream_t s = cs[q];
                if (!cs_used[q]) { if (!idle) HIPCHK(hipStreamWaitEvent(s, L->ev_fork, 0)); cs_used[q] = true; }
                if (!zc) {
                    HIPCHK(hipEventRecord(L->ev_chunk[k & 7], copy));
                    HIPCHK(hipStreamWaitEvent(s, L->ev_chunk[k & 7], 0));
                }
                const uint8_t* src = (zc ? L->h_stage : L->d_blobs) + lo * MKZG_BYTES_PER_BLOB;
                commit_direct_acc_device(c->direct_bits_built, c->d_direct, direct_slot_offset(dslot, (int)lo), src, MKZG_BYTES_PER_BLOB, 0, (int)m, s, kt);
            }
        }
        // join the compute streams (direct) / the copy stream into the lane's main stream
        for (int q = 1; q < 4; q++) {
            if (!cs_used[q]) continue;
            HIPCHK(hipEventRecord(L->ev_cs[q], cs[q]));
            HIPCHK(hipStreamWaitEvent(L->stream, L->ev_cs[q], 0));
        }
        if (!zc) {
            HIPCHK(hipEventRecord(L->ev_cs[0], copy));
            HIPCHK(hipStreamWaitEvent(L->stream, L->ev_cs[0], 0));
        }
        if (direct) commit_direct_fin_device(dslot, (int)n, L->d_out, L->d_status, L->stream, kt, host_fin ? reinterpret_cast<uint32_t*>(L->d_out) : nullptr);
        else (void)commit_piece(c, L, L->d_blobs, MKZG_BYTES_PER_BLOB, 0, (int)n, 0, L->d_out, L->d_status, L->stream);      // bucket method: no buffer to grow
        HIPCHK(hipMemcpyAsync(L->h_out, L->d_out, host_fin ? n * 256 : n * 48, hipMemcpyDeviceToHost, L->stream));
        HIPCHK(hipMemcpyAsync(L->h_out + L->h_out_items * 48, L->d_status, n * sizeof(int), hipMemcpyDeviceToHost, L->stream));
        HIPCHK(hipGetLastError());
        LANE_SYNC(c, L, L->stream);
        L->drained = true;                  // every stream of the call was joined into L->stream above
        return MKZG_OK;
    }();
    drain();
    if (rc) return rc;
    memcpy(st_out, L->h_out + L->h_out_items * 48, n * sizeof(int));
    if (host_fin) {
        for (size_t i = 0; i < n; i++) {
            if (st_out[i]) { memset(out48n + 48 * i, 0, 48); continue; }
            commit_host_finish(reinterpret_cast<const uint32_t*>(L->h_out + 256 * i), out48n + 48 * i);
        }
    } else {
        memcpy(out48n, L->h_out, n * 48);
    }
    return MKZG_OK;
}

extern "C" {

const char* mkzg_version(void) { return "mi355kzg 0.2 (gfx950)"; }
// NewContext4096's precomputation (api.go:90-149) on request: see the header
int mkzg_ctx_prepare(mkzg_ctx* c, unsigned what) {
    if (!c || (what & ~MKZG_PREPARE_ALL)) return MKZG_ERR_BAD_ARGS;
    const std::vector<mkzg_ctx*> all = c->peers.empty() ? std::vector<mkzg_ctx*>{c} : c->peers;
    for (mkzg_ctx* p : all) {
        LaneGuard G(p);                                  // a call like any other: option changes that rebuild a table wait for it (lanes_quiesce)
        if (what & MKZG_PREPARE_COMMIT) {
            int rc = MKZG_OK;
            if (p->opt_direct_bits && !p->direct_bits_built && !p->direct_failed) rc = ensure_direct(p);
            if (!rc && !p->direct_bits_built && p->commit_table_c.load(std::memory_order_acquire) != p->opt_c) rc = ensure_commit_table(p);
            if (rc) return rc;
        }
        if (what & MKZG_PREPARE_CELLS) {
            int rc = ensure_fk20_tables(p);             // twiddles and the monomial SRS on its way
            if (!rc) rc = ensure_rec_tables(p);
            if (rc) return rc;
        }
    }
    return MKZG_OK;
}
int mkzg_valu_calibrate(mkzg_ctx* c, double* wave_instr_per_s) {
    if (!c || !wave_instr_per_s) return MKZG_ERR_BAD_ARGS;
    if (!c->peers.empty()) c = c->peers[0];
    LaneGuard G(c);
    *wave_instr_per_s = valu_calibrate_device(G.s);
    return *wave_instr_per_s > 0.0 ? MKZG_OK : G.fail(MKZG_ERR_HIP);
}

static void ctx_destroy_one(mkzg_ctx* c) {
    if (!c) return;
    DeviceScope dev(c->device);
    if (c->init_stream) lanes_quiesce(c);           // calls still in flight on other threads finish first
    (void)hipDeviceSynchronize();
    for (int i = 0; i < mkzg_ctx::NLANE; i++) lane_destroy(&c->lanes[i]);
    dev_free(c->d_lagrange_aff); dev_free(c->d_commit_table); dev_free(c->d_direct);
    dev_free(c->d_roots_brp); dev_free(c->d_inv_roots_brp);
    dev_free(c->d_tw_fwd); dev_free(c->d_tw_inv); dev_free(c->d_g1tw_fwd); dev_free(c->d_g1tw_inv);
    dev_free(c->d_monomial_aff); dev_free(c->d_pow7); dev_free(c->d_ipow7); dev_free(c->d_fk_t_aff); dev_free(c->d_fk_table);
    c->hash_forming.reset();
    for (auto& h : c->hash_hist) h.reset();
    for (int i = 0; i < mkzg_ctx::NLANE; i++) for (auto& ge : c->lanes[i].graph_exec) if (ge) { (void)hipGraphExecDestroy(ge); ge = nullptr; }
    for (auto& hs : c->hash_stream) if (hs) (void)hipStreamDestroy(hs);
    if (c->init_stream) (void)hipStreamDestroy(c->init_stream);
    delete c->pool;
    delete c;
}
void mkzg_ctx_free(mkzg_ctx* c) {
    if (!c) return;
    if (!c->peers.empty()) {
        for (mkzg_ctx* p : c->peers) ctx_destroy_one(p);
        delete c->pool;
        delete c;
        return;
    }
    ctx_destroy_one(c);
}

// CPUs this process may really burn: the hardware threads capped by the cgroup CPU quota (a container can show 256 and grant 16)
static unsigned usable_cpus() {
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0};
        unsigned long period = 0;
        if (fscanf(f, "%31s %lu", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
            const unsigned long quota = strtoul(q, nullptr, 10);
            if (quota >= period) n = std::min<unsigned>(n, (unsigned)(quota / period));
        }
        fclose(f);
    }
    return n;
}
// Worker threads of one context's host pool: its SHARE of the CPUs the cgroup grants, minus the calling thread.  A node runs one rank
// per GPU (LOCAL_WORLD_SIZE, set by torch.distributed.run and by bench.py's own launcher) or one context over several devices
// (`sharers`): eight ranks on a 16-CPU quota get one worker each -- 16 runnable threads in all, where min(15, cpus - 1) per context
// was 120 workers plus eight spinning callers.  MKZG_HOST_THREADS (callers included) overrides.
static unsigned cpu_share_for(int sharers) {
    int ranks = 1;
    if (const char* e = getenv("LOCAL_WORLD_SIZE")) ranks = std::max(1, atoi(e));
    return std::max(2u, usable_cpus() / (unsigned)std::max(ranks, std::max(1, sharers)));
}
static int pool_workers_for(int sharers) {
    if (const char* e = getenv("MKZG_HOST_THREADS")) {
        const int v = atoi(e);
        if (v >= 1) return std::min(v, 64) - 1;
    }
    return (int)std::min(15u, cpu_share_for(sharers) - 1);
}
// CPUs of the NUMA node the device hangs off (sysfs: the PCI function's numa_node, the node's cpulist), intersected with the CPUs this
// process may run on; empty when anything is missing, when the machine has one node, when fewer CPUs than the pool needs remain -- and
// unless MKZG_POOL_AFFINITY=1 asks for it.  HIP places pinned host allocations on that node, but where the CALLER's blobs live decides more:
// *measured* on a two-socket box (host-pointer VerifyBlobKZGProofBatch(1024), ms per lone call): the whole process on the GPU's node
// (taskset) 3.92-3.96, on the other node 4.0-4.9; only the pool bound to the GPU's node 3.9 / 4.8 / 5.8 over three processes (the
// caller's blobs sat on the other socket in two of them: every gather and hash a remote read), pool floating 4.0 / 4.2 / 4.0.  So the
// workers float by default -- the scheduler keeps them near the data they stream -- and a host that allocates its blobs on the GPU's node
// can opt in.
static cpu_set_t device_node_cpus(int device, int need) {
    cpu_set_t out;
    CPU_ZERO(&out);
    const char* e = getenv("MKZG_POOL_AFFINITY");
    if (!e || atoi(e) == 0) return out;
    char bdf[32] = {0};
    if (hipDeviceGetPCIBusId(bdf, sizeof bdf, device) != hipSuccess) { (void)hipGetLastError(); return out; }
    for (char* q = bdf; *q; q++) *q = (char)tolower(*q);
    char path[160];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bdf);
    int node = -1;
    if (FILE* f = fopen(path, "r")) { if (fscanf(f, "%d", &node) != 1) node = -1; fclose(f); }
    if (node < 0) return out;
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE* f = fopen(path, "r");
    if (!f) return out;
    char list[4096] = {0};
    const bool got = fgets(list, sizeof list, f) != nullptr;
    fclose(f);
    if (!got) return out;
    cpu_set_t mine;
    CPU_ZERO(&mine);
    if (sched_getaffinity(0, sizeof mine, &mine) != 0) return out;
    int n_all = CPU_COUNT(&mine);
    char* save = nullptr;                                 // strtok_r: contexts are created concurrently, one thread per device
    for (char* tok = strtok_r(list, ",\n", &save); tok; tok = strtok_r(nullptr, ",\n", &save)) {
        int a = 0, b = 0;
        const int k = sscanf(tok, "%d-%d", &a, &b);
        if (k == 1) b = a;
        if (k < 1) continue;
        for (int cpu = a; cpu <= b && cpu < CPU_SETSIZE; cpu++) if (CPU_ISSET(cpu, &mine)) CPU_SET(cpu, &out);
    }
    if (CPU_COUNT(&out) == n_all || CPU_COUNT(&out) < std::max(2, need)) CPU_ZERO(&out);      // one node, or too few CPUs left: leave the scheduler alone
    return out;
}
static int ctx_create_one(const char* js, size_t len, int device, int sharers, mkzg_ctx** out) {
    mkzg_ctx* c = new mkzg_ctx();
    {
        const int workers = pool_workers_for(sharers);
        const cpu_set_t cpus = device_node_cpus(device, workers);
        c->pool = new kzg::HostPool(workers, &cpus);
        c->pool_node_cpus = CPU_COUNT(&cpus);
    }
    c->cpu_share = (int)cpu_share_for(sharers);
    c->device = device;
    // reference api.go:93-95 (ErrMinSRSSize) and the fixed-size arrays of trusted_setup.go:23-27
    if (!parse_hex_array(js, len, "g1_monomial", 48, c->g1_monomial) || !parse_hex_array(js, len, "g1_lagrange", 48, c->g1_lagrange) ||
        !parse_hex_array(js, len, "g2_monomial", 96, c->g2_monomial) || c->g1_monomial.size() != 4096 * 48 || c->g1_lagrange.size() != 4096 * 48 ||
        c->g2_monomial.size() < 2 * 96) {
        delete c->pool;
        delete c;
        return MKZG_ERR_BAD_TRUSTED_SETUP;
    }
    DeviceScope dev(c->device);                  // the caller's current device is restored on return
    int rc = [&]() -> int {
        HIPCHK(hipStreamCreateWithFlags(&c->init_stream, hipStreamNonBlocking));
        for (int i = 0; i < mkzg_ctx::NLANE; i++) {
            int r = lane_create(&c->lanes[i], i);
            if (r) return r;
        }
        // decode the Lagrange SRS on the device (no subgroup check: trusted_setup.go:104-117)
        uint8_t* d_raw = nullptr;
        int* d_st = nullptr;
        HIPCHK(hipMalloc(&d_raw, 4096 * 48));
        HIPCHK(hipMalloc(&d_st, 4096 * sizeof(int)));
        HIPCHK(hipMalloc(&c->d_lagrange_aff, 4096 * 128));
        HIPCHK(hipMemcpyAsync(d_raw, c->g1_lagrange.data(), 4096 * 48, hipMemcpyHostToDevice, c->init_stream));
        g1_decompress_device(d_raw, 4096, c->d_lagrange_aff, d_st, c->init_stream);
        HIPCHK(hipGetLastError());
        std::vector<int> st(4096);
        HIPCHK(hipMemcpyAsync(st.data(), d_st, 4096 * sizeof(int), hipMemcpyDeviceToHost, c->init_stream));
        HIPCHK(hipStreamSynchronize(c->init_stream));
        (void)hipFree(d_raw);
        (void)hipFree(d_st);
        for (int v : st) if (v) return MKZG_ERR_BAD_TRUSTED_SETUP;   // reference panics (trusted_setup.go:152-156)
        // domain of the blob polynomial, bit-reversed (api.go:127-132)
        HIPCHK(hipMalloc(&c->d_roots_brp, 4096 * 48));
        HIPCHK(hipMalloc(&c->d_inv_roots_brp, 4096 * 48));
        build_roots_device(c->d_roots_brp, c->d_inv_roots_brp, 12, c->init_stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(c->init_stream));
        // opening key (api.go:104-123): G2 generator and [alpha]G2, Miller-loop lines prepared once
        G2Aff q0, q1;
        if (!g2_decompress(q0, c->g2_monomial.data()) || !g2_decompress(q1, c->g2_monomial.data() + 96)) return MKZG_ERR_BAD_TRUSTED_SETUP;
        g2_prepare(c->g2_gen, q0);
        g2_prepare(c->g2_alpha, q1);
        // EIP-7594 opening key needs [alpha^64]G2 (api.go:113-115 panics below 64 points; we degrade: cell verify unavailable)
        if (c->g2_monomial.size() >= 65 * 96) {
            G2Aff q64, nq0;
            if (!g2_decompress(q64, c->g2_monomial.data() + 64 * 96)) return MKZG_ERR_BAD_TRUSTED_SETUP;
            g2_neg(nq0, q0);
            g2_prepare(c->g2_alpha64, q64);
            g2_prepare(c->g2_neg_gen, nq0);
        }
        // byte-window table of the G1 generator for the [e]G of the batch verifiers (kzg_verify.go:169)
        G1HX gen;
        g1h_generator(gen);
        g1h_fixed_build(c->gen_fixed, gen);
        return MKZG_OK;
    }();
    if (rc) { ctx_destroy_one(c); return rc; }
    *out = c;
    return MKZG_OK;
}

int mkzg_ctx_new(const char* js, size_t len, const int* device_ids, int n_dev, mkzg_ctx** out) {
    if (!js || !out || n_dev < 0 || (n_dev > 0 && !device_ids)) return MKZG_ERR_BAD_ARGS;
    *out = nullptr;
    int ndev_present = 0;
    if (hipGetDeviceCount(&ndev_present) != hipSuccess || ndev_present <= 0) {
        fprintf(stderr, "[mi355kzg] no HIP device: this library has no CPU fallback\n");
        return MKZG_ERR_HIP;
    }
    for (int i = 0; i < n_dev; i++) if (device_ids[i] < 0 || device_ids[i] >= ndev_present) return MKZG_ERR_BAD_ARGS;
    if (n_dev <= 1) {
        int dev = 0;
        if (n_dev == 1) dev = device_ids[0];
        else (void)hipGetDevice(&dev);
        return ctx_create_one(js, len, dev, 1, out);
    }
    // several devices: a front object that owns one full context per listed device and shards batches over them (api_multi.hip)
    mkzg_ctx* front = new mkzg_ctx();
    front->device = device_ids[0];
    front->pool = new kzg::HostPool(n_dev - 1);
    for (int i = 0; i < n_dev; i++) {
        mkzg_ctx* p = nullptr;
        int rc = ctx_create_one(js, len, device_ids[i], n_dev, &p);
        if (rc) { mkzg_ctx_free(front); return rc; }
        front->peers.push_back(p);
    }
    *out = front;
    return MKZG_OK;
}

static int set_option_one(mkzg_ctx* c, const char* key, long value) {
    DeviceScope dev(c->device);
    if (!strcmp(key, "msm_window")) {
        if (value < 11 || value > 13) return MKZG_ERR_BAD_ARGS;
        lanes_quiesce(c);
        c->opt_c = (int)value;
        lanes_resume(c);
        return MKZG_OK;
    }
    if (!strcmp(key, "commit_direct_bits")) {
        if (value != 0 && value != -1 && (value < 12 || value > 15)) return MKZG_ERR_BAD_ARGS;
        lanes_quiesce(c);
        if (c->opt_direct_bits != (int)value) {
            c->opt_direct_bits = (int)value;
            dev_free(c->d_direct);
            c->direct_bits_built = 0;
            c->direct_failed = false;
        }
        lanes_resume(c);
        return MKZG_OK;
    }
    if (!strcmp(key, "max_batch")) {
        if (value < 1 || value > 4096) return MKZG_ERR_BAD_ARGS;
        lanes_quiesce(c);
        c->opt_max_batch = (int)value;
        lanes_resume(c);
        return MKZG_OK;
    }
    if (!strcmp(key, "fk20_bits")) {
        if (value != 8 && value != 11 && value != 12 && value != 13) return MKZG_ERR_BAD_ARGS;
        lanes_quiesce(c);
        if (c->opt_fk20_bits != (int)value) {
            c->opt_fk20_bits = (int)value;
            if (c->fk_ready.load() && c->fk20_bits_built != (int)value) {      // rebuilt by the next call that needs it
                dev_free(c->d_fk_table);
                dev_free(c->d_fk_t_aff);
                c->fk20_bits_built = 0;
                c->fk_ready.store(false);
            }
        }
        lanes_resume(c);
        return MKZG_OK;
    }
    if (!strcmp(key, "lanes")) {                 // calls in flight at once; lanes beyond the limit give their memory back
        if (value < 1 || value > mkzg_ctx::NLANE) return MKZG_ERR_BAD_ARGS;
        lanes_quiesce(c);
        {
            std::lock_guard<std::mutex> g(c->mu);          // lane_acquire reads it under the hand-over mutex
            c->opt_lanes = (int)value;
        }
        for (int i = c->opt_lanes; i < mkzg_ctx::NLANE; i++) { free_lane(&c->lanes[i]); c->lanes[i].warm = false; }
        lanes_resume(c);
        return MKZG_OK;
    }
    if (!strcmp(key, "trim")) {                  // every lane's workspaces and pinned staging back to the system (they are rebuilt on demand)
        lanes_quiesce(c);
        for (int i = 0; i < mkzg_ctx::NLANE; i++) { free_lane(&c->lanes[i]); c->lanes[i].warm = false; }
        lanes_resume(c);
        return MKZG_OK;
    }
    if (!strcmp(key, "cells_chunk")) {
        if (value < 1 || value > 512) return MKZG_ERR_BAD_ARGS;
        lanes_quiesce(c);
        c->opt_cells_chunk = (int)value;
        lanes_resume(c);
        return MKZG_OK;
    }
    if (!strcmp(key, "host_chunk")) {
        if (value < 1 || value > 4096) return MKZG_ERR_BAD_ARGS;
        c->opt_host_chunk = (int)value;
        return MKZG_OK;
    }
    if (!strcmp(key, "host_gather_tasks")) {
        if (value < 1 || value > 64) return MKZG_ERR_BAD_ARGS;
        c->opt_host_gather_tasks = (int)value;
        return MKZG_OK;
    }
    if (!strcmp(key, "host_zero_copy")) {
        if (value < 0 || value > 1) return MKZG_ERR_BAD_ARGS;
        c->opt_host_zero_copy = (int)value;
        return MKZG_OK;
    }
#ifdef MKZG_AFFINE_PAIRS
    if (!strcmp(key, "commit_affine_pairs")) {   // A/B builds only
        if (value < 0 || value > 1) return MKZG_ERR_BAD_ARGS;
        lanes_quiesce(c);                        // lanes pick the scratch up (or drop it) at their next commitment call
        c->opt_affine_pairs = (int)value;
        lanes_resume(c);
        return MKZG_OK;
    }
#endif
    if (!strcmp(key, "host_validate")) {
        if (value < 0 || value > 1) return MKZG_ERR_BAD_ARGS;
        c->opt_host_validate = (int)value;
        return MKZG_OK;
    }
    if (!strcmp(key, "host_fin")) {
        if (value < 0 || value > 1) return MKZG_ERR_BAD_ARGS;
        c->opt_host_fin = (int)value;
        return MKZG_OK;
    }
    if (!strcmp(key, "host_sha")) {
        if (value < -1 || value > 1) return MKZG_ERR_BAD_ARGS;
        c->opt_host_sha = (int)value;
        return MKZG_OK;
    }
    if (!strcmp(key, "debug_serial")) {
        c->opt_debug_serial = value != 0;
        return MKZG_OK;
    }
    if (!strcmp(key, "fk20_segments")) {
        c->opt_fk20_segments = value != 0;
        return MKZG_OK;
    }
    if (!strcmp(key, "host_wait")) {
        if (value < -1 || value > 1) return MKZG_ERR_BAD_ARGS;
        c->opt_host_wait = (int)value;
        return MKZG_OK;
    }
    if (!strcmp(key, "verify_shape")) {
        if (value < -1 || value > 1) return MKZG_ERR_BAD_ARGS;
        c->opt_verify_shape = (int)value;
        return MKZG_OK;
    }
    if (!strcmp(key, "deep_rounds")) {
        if (value != 0 && value != 1) return MKZG_ERR_BAD_ARGS;
        c->opt_deep_rounds = (int)value;
        return MKZG_OK;
    }
    if (!strcmp(key, "launch_graphs")) {
        if (value < 0 || value > 1) return MKZG_ERR_BAD_ARGS;
        c->opt_launch_graphs = (int)value;
        return MKZG_OK;
    }
    if (!strcmp(key, "combine_single_calls")) {
        if (value < 0 || value > 1) return MKZG_ERR_BAD_ARGS;
        c->opt_combine = (int)value;
        return MKZG_OK;
    }
    return MKZG_ERR_BAD_ARGS;
}
int mkzg_ctx_set_option(mkzg_ctx* c, const char* key, long value) {
    if (!c || !key) return MKZG_ERR_BAD_ARGS;
    if (c->peers.empty()) return set_option_one(c, key, value);
    for (mkzg_ctx* p : c->peers) {
        int rc = set_option_one(p, key, value);
        if (rc) return rc;
    }
    return MKZG_OK;
}
int mkzg_ctx_get_option(mkzg_ctx* c, const char* key, long* value) {
    if (!c || !key || !value) return MKZG_ERR_BAD_ARGS;
    mkzg_ctx* q = c->peers.empty() ? c : c->peers[0];
    if (!strcmp(key, "msm_window")) *value = q->opt_c;
    else if (!strcmp(key, "commit_direct_bits")) *value = q->opt_direct_bits;
    else if (!strcmp(key, "commit_direct_bits_effective")) *value = q->direct_bits_built;      // 0 until the first commitment builds the table, or after degrading to buckets
    else if (!strcmp(key, "max_batch")) *value = q->opt_max_batch;
    else if (!strcmp(key, "fk20_bits")) *value = q->opt_fk20_bits;
    else if (!strcmp(key, "fk20_bits_effective")) *value = q->fk20_bits_built;                   // 0 until the first call that needs the table
    else if (!strcmp(key, "host_chunk")) *value = q->opt_host_chunk;
    else if (!strcmp(key, "cells_chunk")) *value = q->opt_cells_chunk;
    else if (!strcmp(key, "host_gather_tasks")) *value = q->opt_host_gather_tasks;
    else if (!strcmp(key, "host_sha")) *value = q->opt_host_sha;
    else if (!strcmp(key, "verify_shape")) *value = q->opt_verify_shape;
    else if (!strcmp(key, "combine_single_calls")) *value = q->opt_combine;
    else if (!strcmp(key, "launch_graphs")) *value = q->opt_launch_graphs;
    else if (!strcmp(key, "deep_rounds")) *value = q->opt_deep_rounds;
    else if (!strcmp(key, "graph_launches")) *value = (long)g_graph_launches.load();                 // process-wide diagnostics of lane_graph_run
    else if (!strcmp(key, "graph_instantiations")) *value = (long)g_graph_instantiations.load();
    else if (!strcmp(key, "graph_fallbacks")) *value = (long)g_graph_fallbacks.load();
    else if (!strcmp(key, "combined_calls")) {                                                  // single-item calls served as members of a group of two or more so far
        *value = 0;
        const std::vector<mkzg_ctx*> all = c->peers.empty() ? std::vector<mkzg_ctx*>{c} : c->peers;
        for (mkzg_ctx* p : all) *value += (long)(p->comb_commit.combined_items() + p->comb_cells.combined_items() + p->comb_verify.combined_items());
    }
    else if (!strcmp(key, "host_wait")) *value = q->opt_host_wait;
    else if (!strcmp(key, "fk20_segments")) *value = q->opt_fk20_segments;
    else if (!strcmp(key, "debug_serial")) *value = q->opt_debug_serial;
    else if (!strcmp(key, "cpu_share")) *value = q->cpu_share;
    else if (!strcmp(key, "host_zero_copy")) *value = q->opt_host_zero_copy;
    else if (!strcmp(key, "host_fin")) *value = q->opt_host_fin;
    else if (!strcmp(key, "host_validate")) *value = q->opt_host_validate;
#ifdef MKZG_AFFINE_PAIRS
    else if (!strcmp(key, "commit_affine_pairs")) *value = q->opt_affine_pairs;
#endif
    else if (!strcmp(key, "devices")) *value = c->peers.empty() ? 1 : (long)c->peers.size();
    else if (!strcmp(key, "tables_on_own_device")) {
        // diagnostic: 1 when every shared table built so far lives on the device of the (per-device) context that owns it
        *value = 1;
        const std::vector<mkzg_ctx*> all = c->peers.empty() ? std::vector<mkzg_ctx*>{c} : c->peers;
        for (mkzg_ctx* p : all) {
            std::lock_guard<std::mutex> g(p->init_mu);     // tables are built under it
            const void* tabs[] = {p->d_lagrange_aff, p->d_commit_table, p->d_direct, p->d_roots_brp, p->d_inv_roots_brp, p->d_tw_fwd, p->d_tw_inv, p->d_g1tw_fwd,
                                  p->d_g1tw_inv, p->d_monomial_aff, p->d_pow7, p->d_ipow7, p->d_fk_t_aff, p->d_fk_table};
            for (const void* t : tabs) if (t && pointer_device(t) != p->device) *value = 0;
        }
    }
    else if (!strcmp(key, "lanes")) *value = q->opt_lanes;
    else if (!strcmp(key, "pool_node_cpus")) *value = q->pool_node_cpus;                        // CPUs of the GPU's NUMA node the pool's workers are bound to (0: not bound)
    else if (!strcmp(key, "host_threads")) *value = q->pool->size() + 1;                         // workers of one device's pool + the calling thread
    else return MKZG_ERR_BAD_ARGS;
    return MKZG_OK;
}

// device-resident batch: the product's hot path.  Asynchronous on `stream`.
int mkzg_blob_to_kzg_commitment_batch_device(mkzg_ctx* c, const void* d_blobs, size_t n, void* d_out48n, int* d_status, void* stream) {
    if (!c || (!d_blobs && n) || (!d_out48n && n) || (!d_status && n)) return MKZG_ERR_BAD_ARGS;
    if (!c->peers.empty() && !(c = ctx_for_pointer(c, d_blobs))) return MKZG_ERR_BAD_ARGS;
    if (n == 0) return MKZG_OK;
    LaneGuard G(c, (hipStream_t)stream);
    return G.fail(commit_scalars_device(c, G.L, (const uint8_t*)d_blobs, MKZG_BYTES_PER_BLOB, 0, n, (uint8_t*)d_out48n, d_status, G.s));
}

int mkzg_blob_to_kzg_commitment_batch(mkzg_ctx* c, const uint8_t* const* blobs, size_t n, uint8_t* out48n, int* per_item_status) {
    if (!c || (n && (!blobs || !out48n))) return MKZG_ERR_BAD_ARGS;
    if (n == 0) return MKZG_OK;
    for (size_t i = 0; i < n; i++) if (!blobs[i]) return MKZG_ERR_NIL_INPUT;      // serialization.go:135-137
    if (!c->peers.empty()) return md_blob_to_kzg_commitment_batch(c, blobs, n, out48n, per_item_status);
    int first = MKZG_OK;
    std::vector<int> st;
    LaneGuard G(c);
    for (size_t done = 0; done < n;) {
        const size_t m = std::min<size_t>(n - done, (size_t)c->opt_max_batch);
        st.resize(m);
        int rc = commit_host_piece(c, G.L, blobs + done, m, out48n + done * 48, st.data());
        if (rc) return G.fail(rc);
        for (size_t i = 0; i < m; i++) {
            if (per_item_status) per_item_status[done + i] = st[i];
            if (st[i] && !first) first = st[i];
        }
        done += m;
    }
    return first;
}

int mkzg_blob_to_kzg_commitment(mkzg_ctx* c, const uint8_t* blob, int num_threads, uint8_t out48[48]) {
    if (!c || !out48) return MKZG_ERR_BAD_ARGS;
    if (!blob) return MKZG_ERR_NIL_INPUT;
    if (!c->peers.empty()) c = c->peers[c->rr++ % c->peers.size()];
    // calls that find others in flight leave as one batch (host_combine.hpp); a caller that is alone is a group of one: the path it always took
    mkzg_ctx::CommitCall call{blob, out48, MKZG_OK};
    auto process = [c](std::vector<mkzg_ctx::CommitCall*>& items) {
        const size_t n = items.size();
        if (n == 1) {
            const uint8_t* one[1] = {items[0]->blob};
            int st = 0;
            items[0]->rc = mkzg_blob_to_kzg_commitment_batch(c, one, 1, items[0]->out48, &st);
            return;
        }
        c->comb_commit.count_combined(n);
        std::vector<const uint8_t*> ptrs(n);
        std::vector<uint8_t> out(48 * n);
        std::vector<int> st(n, 0);
        for (size_t i = 0; i < n; i++) ptrs[i] = items[i]->blob;
        const int rc = mkzg_blob_to_kzg_commitment_batch(c, ptrs.data(), n, out.data(), st.data());
        const bool broken = rc == MKZG_ERR_HIP || rc == MKZG_ERR_BAD_ARGS;           // not an item's status: nobody got a result
        for (size_t i = 0; i < n; i++) {
            items[i]->rc = broken ? rc : st[i];
            if (!items[i]->rc) memcpy(items[i]->out48, out.data() + 48 * i, 48);
        }
    };
    if (c->opt_combine.load(std::memory_order_relaxed)) {
        c->comb_commit.run(&call, process);
    } else {
        std::vector<mkzg_ctx::CommitCall*> one{&call};
        process(one);
    }
    if (call.rc) return call.rc;
    if (num_threads >= 1024) return MKZG_ERR_TOO_MANY_GOROUTINES;                  // multiexp.go:45-50
    return MKZG_OK;
}

int mkzg_commit_lagrange(mkzg_ctx* c, const uint8_t* scalars32, size_t n, int num_threads, uint8_t out48[48]) {
    if (!c || !out48 || (!scalars32 && n)) return MKZG_ERR_BAD_ARGS;
    if (n == 0 || n > 4096) return MKZG_ERR_INVALID_POLY_SIZE;                     // srs.go:57-59
    if (num_threads >= 1024) return MKZG_ERR_TOO_MANY_GOROUTINES;
    if (!c->peers.empty()) c = c->peers[c->rr++ % c->peers.size()];
    std::vector<uint8_t> padded(MKZG_BYTES_PER_BLOB, 0);                             // zero scalars contribute nothing
    memcpy(padded.data(), scalars32, n * 32);
    const uint8_t* one[1] = {padded.data()};
    int st = 0;
    return mkzg_blob_to_kzg_commitment_batch(c, one, 1, out48, &st);
}

int mkzg_profile_enable(mkzg_ctx* c, int on) {
    if (!c) return MKZG_ERR_BAD_ARGS;
    if (!c->peers.empty()) { for (mkzg_ctx* p : c->peers) p->profiling = on != 0; return MKZG_OK; }
    c->profiling = on != 0;
    return MKZG_OK;
}
int mkzg_profile_read(mkzg_ctx* c, mkzg_kernel_time* out, int cap, int reset) {
    if (!c || (cap > 0 && !out)) return 0;
    for (int i = 0; i < KT_COUNT && i < cap; i++) { out[i].name = kKernelNames[i]; out[i].total_ms = 0; out[i].launches = 0; }
    std::vector<mkzg_ctx*> all = c->peers.empty() ? std::vector<mkzg_ctx*>{c} : c->peers;
    for (mkzg_ctx* q : all) {
        DeviceScope dev(q->device);
        lanes_quiesce(q);
        for (int l = 0; l < mkzg_ctx::NLANE; l++) {
            KernelTimer& t = q->lanes[l].timer;
            t.collect();
            for (int i = 0; i < KT_COUNT && i < cap; i++) { out[i].total_ms += t.total_ms[i]; out[i].launches += t.launches[i]; }
            if (reset) t.reset();
        }
        lanes_resume(q);
    }
    return KT_COUNT;
}

}  // extern "C"

mkzg_ctx* ctx_for_pointer(mkzg_ctx* c, const void* p) {
    if (c->peers.empty()) return c;
    const int dev = pointer_device(p);
    for (mkzg_ctx* q : c->peers) if (q->device == dev) return q;
    return nullptr;
}
