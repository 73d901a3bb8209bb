// api.hip -- the C ABI of libmi355kzg.so (include/mi355kzg.h): context construction and the
// host-side mirror of go-eth-kzg's Context methods for the hot path.  Host logic only: every
// field/group operation happens in HIP kernels (msm_kernels.hip); there is no CPU fallback.
//
// Reference being mirrored (paths relative to crate-crypto/go-eth-kzg):
//   NewContext4096            api.go:90-149   (+ trusted_setup.go:90-192 JSON / point decoding)
//   Context.BlobToKZGCommitment  prove.go:13-33
//   kzg.CommitKey.Commit      internal/kzg/srs.go:56-62
//   multiexp.MultiExpG1       internal/multiexp/multiexp.go:20-50
#include <thread>
#include "ctx.h"

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

int ensure_plan(mkzg_ctx* c) {
    if (c->plan_ready) return MKZG_OK;
    CommitPlanSizes sz;
    commit_plan_sizes(c->opt_c, c->opt_max_batch, &sz);
    uint32_t* table = nullptr;
    HIPCHK(hipMalloc(&table, sz.table_bytes));
    for (int i = 0; i < mkzg_ctx::NSLOT; i++) {
        CommitPlan& p = c->slots[i];
        p.c = c->opt_c;
        p.max_batch = c->opt_max_batch;
        p.d_table = table;
        HIPCHK(hipMalloc(&p.d_entries, sz.entries_bytes));
        HIPCHK(hipMalloc(&p.d_task_off, sz.task_off_bytes));
        HIPCHK(hipMalloc(&p.d_tasks, sz.tasks_bytes));
        HIPCHK(hipMalloc(&p.d_ntasks, sz.ntasks_bytes));
        HIPCHK(hipMalloc(&p.d_parts, sz.parts_bytes));
        HIPCHK(hipMalloc(&p.d_partials, sz.partials_bytes));
        if (!c->slot_done[i]) HIPCHK(hipEventCreateWithFlags(&c->slot_done[i], hipEventDisableTiming));
        if (!c->acc_done[i]) HIPCHK(hipEventCreateWithFlags(&c->acc_done[i], hipEventDisableTiming));
        c->slot_used[i] = false;
    }
    c->plan = c->slots[0];
    // api.go:131: the commit key is the bit-reversed Lagrange SRS
    build_table_device(c->plan.c, c->d_lagrange_aff, table, 4096, 12, c->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    c->plan_ready = true;
    return MKZG_OK;
}
static void free_plan(mkzg_ctx* c) {
    if (c->slots[0].d_table) (void)hipFree(c->slots[0].d_table);
    for (int i = 0; i < mkzg_ctx::NSLOT; i++) {
        CommitPlan& p = c->slots[i];
        void* ptrs[] = {p.d_entries, p.d_task_off, p.d_tasks, p.d_ntasks, p.d_parts, p.d_partials};
        for (void* q : ptrs) if (q) (void)hipFree(q);
        p = CommitPlan();
        c->slot_used[i] = false;
    }
    c->plan = CommitPlan();
    c->plan_ready = false;
}
// workspace hand-over between streams: the new user waits (on the device) for the previous one
int acquire_slot(mkzg_ctx* c, hipStream_t s) {
    const int slot = (int)(c->slot_counter++ % mkzg_ctx::NSLOT);
    if (c->slot_used[slot] && hipStreamWaitEvent(s, c->slot_done[slot], 0) != hipSuccess) return -1;
    c->slots[slot].wait_before_acc = (c->opt_chain_acc && c->last_acc_slot >= 0) ? c->acc_done[c->last_acc_slot] : nullptr;
    c->slots[slot].record_after_acc = c->acc_done[slot];
    return slot;
}
int release_slot(mkzg_ctx* c, int slot, hipStream_t s) {
    HIPCHK(hipEventRecord(c->slot_done[slot], s));
    c->slot_used[slot] = true;
    c->last_acc_slot = slot;
    return MKZG_OK;
}

extern "C" {

const char* mkzg_version(void) { return "mi355kzg 0.1 (gfx950)"; }
int mkzg_valu_calibrate(mkzg_ctx* c, double* wave_instr_per_s) {
    if (!c || !wave_instr_per_s) return MKZG_ERR_BAD_ARGS;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    *wave_instr_per_s = valu_calibrate_device(c->stream);
    return *wave_instr_per_s > 0.0 ? MKZG_OK : MKZG_ERR_HIP;
}

void mkzg_ctx_free(mkzg_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    free_plan(c);
    if (c->d_lagrange_aff) (void)hipFree(c->d_lagrange_aff);
    if (c->d_direct) (void)hipFree(c->d_direct);
    for (int i = 0; i < mkzg_ctx::NSLOT; i++) {
        if (c->dslots[i].d_parts) (void)hipFree(c->dslots[i].d_parts);
        if (c->dslots[i].d_bad) (void)hipFree(c->dslots[i].d_bad);
    }
    if (c->d_roots_brp) (void)hipFree(c->d_roots_brp);
    if (c->d_inv_roots_brp) (void)hipFree(c->d_inv_roots_brp);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->d_blobs) (void)hipFree(c->d_blobs);
    if (c->d_out) (void)hipFree(c->d_out);
    if (c->d_status) (void)hipFree(c->d_status);
    free_verify_ws(c);
    free_cells(c);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->side2) (void)hipStreamDestroy(c->side2);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_decoded) (void)hipEventDestroy(c->ev_decoded);
    if (c->fk_ws_done) (void)hipEventDestroy(c->fk_ws_done);
    delete c->pool;
    delete c;
}

int mkzg_ctx_new(const char* js, size_t len, const int* device_ids, int n_dev, mkzg_ctx** out) {
    if (!js || !out) return MKZG_ERR_BAD_ARGS;
    *out = nullptr;
    int ndev_present = 0;
    if (hipGetDeviceCount(&ndev_present) != hipSuccess || ndev_present <= 0) {
        fprintf(stderr, "[mi355kzg] no HIP device: this library has no CPU fallback\n");
        return MKZG_ERR_HIP;
    }
    mkzg_ctx* c = new mkzg_ctx();
    c->pool = new kzg::HostPool(7);
    if (n_dev > 0 && device_ids) c->device = device_ids[0];
    else (void)hipGetDevice(&c->device);
    // reference api.go:93-95 (ErrMinSRSSize) and the fixed-size arrays of trusted_setup.go:23-27
    if (!parse_hex_array(js, len, "g1_monomial", 48, c->g1_monomial) || !parse_hex_array(js, len, "g1_lagrange", 48, c->g1_lagrange) ||
        !parse_hex_array(js, len, "g2_monomial", 96, c->g2_monomial) || c->g1_monomial.size() != 4096 * 48 || c->g1_lagrange.size() != 4096 * 48 ||
        c->g2_monomial.size() < 2 * 96) {
        delete c;
        return MKZG_ERR_BAD_TRUSTED_SETUP;
    }
    int rc = [&]() -> int {
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&c->side2, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->ev_decoded, hipEventDisableTiming));
        // decode the Lagrange SRS on the device (no subgroup check: trusted_setup.go:104-117)
        uint8_t* d_raw = nullptr;
        int* d_st = nullptr;
        HIPCHK(hipMalloc(&d_raw, 4096 * 48));
        HIPCHK(hipMalloc(&d_st, 4096 * sizeof(int)));
        HIPCHK(hipMalloc(&c->d_lagrange_aff, 4096 * 128));
        HIPCHK(hipMemcpyAsync(d_raw, c->g1_lagrange.data(), 4096 * 48, hipMemcpyHostToDevice, c->stream));
        g1_decompress_device(d_raw, 4096, c->d_lagrange_aff, d_st, c->stream);
        HIPCHK(hipGetLastError());
        std::vector<int> st(4096);
        HIPCHK(hipMemcpyAsync(st.data(), d_st, 4096 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        (void)hipFree(d_raw);
        (void)hipFree(d_st);
        for (int v : st) if (v) return MKZG_ERR_BAD_TRUSTED_SETUP;   // reference panics (trusted_setup.go:152-156)
        // domain of the blob polynomial, bit-reversed (api.go:127-132)
        HIPCHK(hipMalloc(&c->d_roots_brp, 4096 * 48));
        HIPCHK(hipMalloc(&c->d_inv_roots_brp, 4096 * 48));
        build_roots_device(c->d_roots_brp, c->d_inv_roots_brp, 12, c->stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(c->stream));
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
        return MKZG_OK;
    }();
    if (rc) { mkzg_ctx_free(c); return rc; }
    *out = c;
    return MKZG_OK;
}

int mkzg_ctx_set_option(mkzg_ctx* c, const char* key, long value) {
    if (!c || !key) return MKZG_ERR_BAD_ARGS;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    (void)hipSetDevice(c->device);
    if (!strcmp(key, "msm_window")) {
        if (value < 11 || value > 13) return MKZG_ERR_BAD_ARGS;
        if (c->opt_c != (int)value) { c->opt_c = (int)value; free_plan(c); }
        return MKZG_OK;
    }
    if (!strcmp(key, "commit_direct_bits")) {
        if (value != 0 && (value < 12 || value > 15)) return MKZG_ERR_BAD_ARGS;
        if (c->opt_direct_bits != (int)value) {
            c->opt_direct_bits = (int)value;
            if (c->d_direct) { (void)hipFree(c->d_direct); c->d_direct = nullptr; c->direct_bits_built = 0; }
        }
        return MKZG_OK;
    }
    if (!strcmp(key, "chain_accumulate")) {
        if (value != 0 && value != 1) return MKZG_ERR_BAD_ARGS;
        c->opt_chain_acc = (int)value;
        return MKZG_OK;
    }
    if (!strcmp(key, "max_batch")) {
        if (value < 1 || value > 4096) return MKZG_ERR_BAD_ARGS;
        if (c->opt_max_batch != (int)value) { c->opt_max_batch = (int)value; free_plan(c); }
        return MKZG_OK;
    }
    return MKZG_ERR_BAD_ARGS;
}

// Commitments of n polynomials in Lagrange form against the bit-reversed Lagrange SRS, with the method the context is
// configured for (direct window table or buckets).  input_kind 0: 32-byte big-endian scalars (blobs, canonicity checked);
// 1: 12-word Montgomery Fr rows (quotient polynomials of the proof path).  Caller holds c->mu and has set the device.
int commit_scalars_device(mkzg_ctx* c, const uint8_t* d_scalars, size_t stride_bytes, int input_kind, size_t n, uint8_t* d_out48n, int* d_status, hipStream_t s) {
    if (c->opt_direct_bits) {
        if (c->direct_bits_built != c->opt_direct_bits) {
            const int cb = c->opt_direct_bits, nwin = (255 + cb - 1) / cb;
            uint32_t* tmp = nullptr;
            if (c->d_direct) (void)hipFree(c->d_direct);
            c->d_direct = nullptr;
            c->direct_bits_built = 0;
            if (hipMalloc(&c->d_direct, direct_table_bytes(cb)) != hipSuccess) {       // e.g. 146 GB not available on this device
                (void)hipGetLastError();
                c->d_direct = nullptr;
                return MKZG_ERR_HIP;
            }
            if (hipMalloc(&tmp, (size_t)nwin * 4096 * 128) != hipSuccess) {
                (void)hipGetLastError();
                (void)hipFree(c->d_direct);
                c->d_direct = nullptr;
                return MKZG_ERR_HIP;
            }
            build_table_any_device(cb, c->d_lagrange_aff, tmp, 4096, 12, c->stream);
            build_direct_device(cb, tmp, c->d_direct, c->stream);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(c->stream));
            (void)hipFree(tmp);
            c->direct_bits_built = cb;
        }
        if (c->dslot_batch != c->opt_max_batch) {
            HIPCHK(hipDeviceSynchronize());
            for (int i = 0; i < mkzg_ctx::NSLOT; i++) {
                if (c->dslots[i].d_parts) (void)hipFree(c->dslots[i].d_parts);
                if (c->dslots[i].d_bad) (void)hipFree(c->dslots[i].d_bad);
                c->dslots[i].d_parts = nullptr; c->dslots[i].d_bad = nullptr;
                HIPCHK(hipMalloc(&c->dslots[i].d_parts, direct_parts_bytes(c->opt_max_batch)));
                HIPCHK(hipMalloc(&c->dslots[i].d_bad, (size_t)c->opt_max_batch * 64 * sizeof(int)));
                if (!c->slot_done[i]) HIPCHK(hipEventCreateWithFlags(&c->slot_done[i], hipEventDisableTiming));
                if (!c->acc_done[i]) HIPCHK(hipEventCreateWithFlags(&c->acc_done[i], hipEventDisableTiming));
                c->slot_used[i] = false;
            }
            c->last_acc_slot = -1;
            c->dslot_batch = c->opt_max_batch;
        }
        size_t done = 0;
        while (done < n) {
            const int chunk = (int)std::min<size_t>(n - done, (size_t)c->opt_max_batch);
            const int slot = (int)(c->slot_counter++ % mkzg_ctx::NSLOT);
            if (c->slot_used[slot]) HIPCHK(hipStreamWaitEvent(s, c->slot_done[slot], 0));
            c->dslots[slot].wait_before_acc = (c->opt_chain_acc && c->last_acc_slot >= 0) ? c->acc_done[c->last_acc_slot] : nullptr;
            c->dslots[slot].record_after_acc = c->acc_done[slot];
            commit_direct_device(c->opt_direct_bits, c->d_direct, c->dslots[slot], d_scalars + done * stride_bytes, stride_bytes, input_kind, chunk, d_out48n + done * 48,
                                 d_status + done, s, c->profiling ? &c->timer : nullptr);
            int rr = release_slot(c, slot, s);
            if (rr) return rr;
            done += chunk;
        }
        HIPCHK(hipGetLastError());
        return MKZG_OK;
    }
    int rc = ensure_plan(c);
    if (rc) return rc;
    size_t done = 0;
    while (done < n) {
        int chunk = (int)std::min<size_t>(n - done, (size_t)c->plan.max_batch);
        const int slot = acquire_slot(c, s);
        if (slot < 0) return MKZG_ERR_HIP;
        commit_batch_device(c->slots[slot], d_scalars + done * stride_bytes, stride_bytes, input_kind, chunk, d_out48n + done * 48, d_status + done, s,
                            c->profiling ? &c->timer : nullptr);
        int rr = release_slot(c, slot, s);
        if (rr) return rr;
        done += chunk;
    }
    HIPCHK(hipGetLastError());
    return MKZG_OK;
}
// device-resident batch: the product's hot path
int mkzg_blob_to_kzg_commitment_batch_device(mkzg_ctx* c, const void* d_blobs, size_t n, void* d_out48n, int* d_status, void* stream) {
    if (!c || (!d_blobs && n) || (!d_out48n && n) || (!d_status && n)) return MKZG_ERR_BAD_ARGS;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    return commit_scalars_device(c, (const uint8_t*)d_blobs, MKZG_BYTES_PER_BLOB, 0, n, (uint8_t*)d_out48n, d_status, (hipStream_t)stream);
}

int ensure_staging(mkzg_ctx* c, size_t n_items, size_t in_bytes_per_item) {
    size_t need = n_items * in_bytes_per_item;
    if (c->h_stage_bytes < need) {
        if (c->h_stage) (void)hipHostFree(c->h_stage);
        c->h_stage = nullptr; c->h_stage_bytes = 0;
        HIPCHK(hipHostMalloc(&c->h_stage, need, hipHostMallocDefault));
        c->h_stage_bytes = need;
    }
    if (c->d_blobs_bytes < need) {
        if (c->d_blobs) (void)hipFree(c->d_blobs);
        c->d_blobs = nullptr; c->d_blobs_bytes = 0;
        HIPCHK(hipMalloc(&c->d_blobs, need));
        c->d_blobs_bytes = need;
    }
    if (c->d_out_items < n_items) {
        if (c->d_out) (void)hipFree(c->d_out);
        if (c->d_status) (void)hipFree(c->d_status);
        c->d_out = nullptr; c->d_status = nullptr; c->d_out_items = 0;
        HIPCHK(hipMalloc(&c->d_out, n_items * 48));
        HIPCHK(hipMalloc(&c->d_status, n_items * sizeof(int)));
        c->d_out_items = n_items;
    }
    return MKZG_OK;
}

int mkzg_blob_to_kzg_commitment_batch(mkzg_ctx* c, const uint8_t* const* blobs, size_t n, uint8_t* out48n, int* per_item_status) {
    if (!c || (n && (!blobs || !out48n))) return MKZG_ERR_BAD_ARGS;
    if (n == 0) return MKZG_OK;
    for (size_t i = 0; i < n; i++) if (!blobs[i]) return MKZG_ERR_NIL_INPUT;      // serialization.go:135-137
    int first = MKZG_OK;
    size_t done = 0;
    while (done < n) {
        std::lock_guard<std::recursive_mutex> g(c->mu);
        HIPCHK(hipSetDevice(c->device));
        size_t chunk = std::min<size_t>(n - done, (size_t)c->opt_max_batch);
        int rc = ensure_staging(c, chunk, MKZG_BYTES_PER_BLOB);
        if (rc) return rc;
        // gather the blobs into the pinned staging area with a few host threads (33 MB for 256 blobs: a single memcpy
        // stream takes about as long as the GPU's whole commitment pass)
        {
            const size_t nthr = chunk >= 32 ? 4 : 1, per = (chunk + nthr - 1) / nthr;
            auto gather = [&](size_t q) {
                for (size_t i = q * per; i < std::min(chunk, (q + 1) * per); i++) memcpy(c->h_stage + i * MKZG_BYTES_PER_BLOB, blobs[done + i], MKZG_BYTES_PER_BLOB);
            };
            c->pool->parallel((int)nthr, [&](int q) { gather((size_t)q); });
        }
        HIPCHK(hipMemcpyAsync(c->d_blobs, c->h_stage, chunk * MKZG_BYTES_PER_BLOB, hipMemcpyHostToDevice, c->stream));
        rc = mkzg_blob_to_kzg_commitment_batch_device(c, c->d_blobs, chunk, c->d_out, c->d_status, c->stream);
        if (rc) return rc;
        std::vector<int> st(chunk);
        HIPCHK(hipMemcpyAsync(out48n + done * 48, c->d_out, chunk * 48, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(st.data(), c->d_status, chunk * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        for (size_t i = 0; i < chunk; i++) {
            if (per_item_status) per_item_status[done + i] = st[i];
            if (st[i] && !first) first = st[i];
        }
        done += chunk;
    }
    return first;
}

int mkzg_blob_to_kzg_commitment(mkzg_ctx* c, const uint8_t* blob, int num_threads, uint8_t out48[48]) {
    if (!c || !out48) return MKZG_ERR_BAD_ARGS;
    if (!blob) return MKZG_ERR_NIL_INPUT;
    const uint8_t* one[1] = {blob};
    int st = 0;
    int rc = mkzg_blob_to_kzg_commitment_batch(c, one, 1, out48, &st);
    if (rc) return rc;
    if (num_threads >= 1024) return MKZG_ERR_TOO_MANY_GOROUTINES;                  // multiexp.go:45-50
    return MKZG_OK;
}

int mkzg_commit_lagrange(mkzg_ctx* c, const uint8_t* scalars32, size_t n, int num_threads, uint8_t out48[48]) {
    if (!c || !out48 || (!scalars32 && n)) return MKZG_ERR_BAD_ARGS;
    if (n == 0 || n > 4096) return MKZG_ERR_INVALID_POLY_SIZE;                     // srs.go:57-59
    if (num_threads >= 1024) return MKZG_ERR_TOO_MANY_GOROUTINES;
    std::vector<uint8_t> padded(MKZG_BYTES_PER_BLOB, 0);                             // zero scalars contribute nothing
    memcpy(padded.data(), scalars32, n * 32);
    const uint8_t* one[1] = {padded.data()};
    int st = 0;
    return mkzg_blob_to_kzg_commitment_batch(c, one, 1, out48, &st);
}

int mkzg_profile_enable(mkzg_ctx* c, int on) {
    if (!c) return MKZG_ERR_BAD_ARGS;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    c->profiling = on != 0;
    return MKZG_OK;
}
int mkzg_profile_read(mkzg_ctx* c, mkzg_kernel_time* out, int cap, int reset) {
    if (!c) return 0;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    c->timer.collect();
    for (int i = 0; i < KT_COUNT && i < cap; i++) {
        out[i].name = kKernelNames[i];
        out[i].total_ms = c->timer.total_ms[i];
        out[i].launches = c->timer.launches[i];
    }
    if (reset) c->timer.reset();
    return KT_COUNT;
}

}  // extern "C"
