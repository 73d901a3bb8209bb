// api_cells.hip -- C ABI entry points of the EIP-7594 cell paths (include/mi355kzg.h):
//   Context.ComputeCells                  reference api_eip7594.go:12-26
//   Context.ComputeCellsAndKZGProofs      reference api_eip7594.go:28-52
// All field work happens in cells_kernels.hip / fk20_kernels.hip; the host only stages bytes.
#include <sys/random.h>
#include <string>
#include <thread>
#include <unordered_map>
#include "ctx.h"

static const size_t V_MISC_WORDS_C = 16 + 4 * VMSM_WINDOWS * 48;
#define KT(kind, s, call)                                  \
    do {                                                   \
        if (c->profiling) c->timer.begin(kind, s);         \
        call;                                              \
        if (c->profiling) c->timer.end(kind, s);           \
    } while (0)

// The cells / FK20 / recovery workspaces are shared by every call; entry points that return before the GPU is done record
// fk_ws_done, and every user of the workspaces first makes its stream wait for it (no host synchronisation).
static int fk_ws_acquire(mkzg_ctx* c, hipStream_t s) {
    if (!c->fk_ws_done) HIPCHK(hipEventCreateWithFlags(&c->fk_ws_done, hipEventDisableTiming));
    if (c->fk_ws_used) HIPCHK(hipStreamWaitEvent(s, c->fk_ws_done, 0));
    return MKZG_OK;
}
static int fk_ws_release(mkzg_ctx* c, hipStream_t s) {
    HIPCHK(hipEventRecord(c->fk_ws_done, s));
    c->fk_ws_used = true;
    return MKZG_OK;
}

void free_cells(mkzg_ctx* c) {
    void* ptrs[] = {c->d_tw_fwd, c->d_tw_inv, c->d_g1tw_fwd, c->d_g1tw_inv, c->d_coeffs, c->d_cells, c->d_cell_st,
                    c->d_monomial_aff, c->d_fk_t_aff, c->d_fk_table, c->d_fk_scalars, c->d_fk_results, c->d_fk_proofs, c->d_fk_plan, c->d_fk_parts,
                    c->d_cv_cells, c->d_cv_cell_idx, c->d_cv_row_idx, c->d_cv_rp, c->d_cv_weights, c->d_cv_partials, c->d_cv_interp, c->d_cv_st,
                    c->d_pow7, c->d_ipow7, c->d_rec_scratch, c->d_rec_pos};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    c->d_tw_fwd = c->d_tw_inv = c->d_g1tw_fwd = c->d_g1tw_inv = c->d_coeffs = nullptr;
    c->d_cells = nullptr;
    c->d_cell_st = nullptr;
    c->d_monomial_aff = c->d_fk_t_aff = c->d_fk_table = c->d_fk_scalars = c->d_fk_results = c->d_fk_plan = c->d_fk_parts = nullptr;
    c->d_fk_proofs = nullptr;
    c->d_cv_cells = nullptr; c->d_cv_cell_idx = nullptr; c->d_cv_row_idx = nullptr; c->d_cv_rp = nullptr; c->d_cv_weights = nullptr;
    c->d_cv_partials = nullptr; c->d_cv_interp = nullptr; c->d_cv_st = nullptr; c->cv_cap = 0;
    c->d_pow7 = c->d_ipow7 = c->d_rec_scratch = nullptr; c->d_rec_pos = nullptr; c->rec_ready = false; c->rec_cap = 0;
    c->cells_ready = false;
    c->cell_cap = 0;
    c->fk_ready = false;
    c->fk_cap = 0;
}
// monomial G1 SRS on the device (decoded without subgroup checks, trusted_setup.go:104-117)
int ensure_monomial(mkzg_ctx* c) {
    if (c->d_monomial_aff) return MKZG_OK;
    uint8_t* d_raw = nullptr;
    int* d_st = nullptr;
    uint32_t* d_aff = nullptr;
    HIPCHK(hipMalloc(&d_raw, 4096 * 48));
    HIPCHK(hipMalloc(&d_st, 4096 * sizeof(int)));
    HIPCHK(hipMalloc(&d_aff, 4096 * 128));
    HIPCHK(hipMemcpyAsync(d_raw, c->g1_monomial.data(), 4096 * 48, hipMemcpyHostToDevice, c->stream));
    g1_decompress_device(d_raw, 4096, d_aff, d_st, c->stream);
    std::vector<int> st(4096);
    HIPCHK(hipMemcpyAsync(st.data(), d_st, 4096 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    (void)hipFree(d_raw);
    (void)hipFree(d_st);
    for (int v : st) if (v) { (void)hipFree(d_aff); return MKZG_ERR_BAD_TRUSTED_SETUP; }
    c->d_monomial_aff = d_aff;
    return MKZG_OK;
}
static int ensure_cv_ws(mkzg_ctx* c, size_t n) {
    if (n <= c->cv_cap) return MKZG_OK;
    void* ptrs[] = {c->d_cv_cells, c->d_cv_cell_idx, c->d_cv_row_idx, c->d_cv_rp, c->d_cv_weights, c->d_cv_partials, c->d_cv_interp, c->d_cv_st};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    c->cv_cap = 0;
    size_t cap = std::max<size_t>(n, 128);
    HIPCHK(hipMalloc(&c->d_cv_cells, cap * 2048));
    HIPCHK(hipMalloc(&c->d_cv_cell_idx, cap * 8));
    HIPCHK(hipMalloc(&c->d_cv_row_idx, cap * 4));
    HIPCHK(hipMalloc(&c->d_cv_rp, cap * 48));
    HIPCHK(hipMalloc(&c->d_cv_weights, cap * 32));
    HIPCHK(hipMalloc(&c->d_cv_partials, (size_t)CELL_INTERP_BLOCKS * 64 * 48));
    HIPCHK(hipMalloc(&c->d_cv_interp, 64 * 32));
    HIPCHK(hipMalloc(&c->d_cv_st, 3 * cap * sizeof(int)));
    c->cv_cap = cap;
    return MKZG_OK;
}
// fk20.NewFK20 (reference fk20.go:23-52, called from api.go:137): done lazily on first use
int ensure_fk20(mkzg_ctx* c, size_t n) {
    int rc = ensure_cells(c, n);
    if (rc) return rc;
    if (!c->fk_ready) {
        rc = ensure_monomial(c);
        if (rc) return rc;
        HIPCHK(hipMalloc(&c->d_fk_t_aff, 8192 * 128));
        HIPCHK(hipMalloc(&c->d_fk_table, fk20_table_bytes()));
        fk20_setup_device(c->d_monomial_aff, c->d_g1tw_fwd, c->d_fk_t_aff, c->d_fk_table, c->stream);
        {
            std::vector<uint32_t> plan(FK_PLAN_WORDS);
            fk20_build_plan(plan.data());
            HIPCHK(hipMalloc(&c->d_fk_plan, FK_PLAN_WORDS * sizeof(uint32_t)));
            HIPCHK(hipMemcpy(c->d_fk_plan, plan.data(), FK_PLAN_WORDS * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(c->stream));
        c->fk_ready = true;
    }
    if (n > c->fk_cap) {
        if (c->d_fk_scalars) (void)hipFree(c->d_fk_scalars);
        if (c->d_fk_results) (void)hipFree(c->d_fk_results);
        if (c->d_fk_parts) (void)hipFree(c->d_fk_parts);
        if (c->d_fk_proofs) (void)hipFree(c->d_fk_proofs);
        c->fk_cap = 0;
        HIPCHK(hipMalloc(&c->d_fk_scalars, n * 128 * 64 * 32));
        HIPCHK(hipMalloc(&c->d_fk_results, 2 * n * 128 * 256));
        HIPCHK(hipMalloc(&c->d_fk_parts, n * 128 * 24 * 256));   // up to 24 partial sums per MSM (fk20_msm_device); the DFT levels reuse it (1024 rows per blob)
        HIPCHK(hipMalloc(&c->d_fk_proofs, n * 128 * 48));
        c->fk_cap = n;
    }
    return MKZG_OK;
}
int ensure_cells(mkzg_ctx* c, size_t n) {
    if (!c->cells_ready) {
        HIPCHK(hipMalloc(&c->d_tw_fwd, 4096 * 48));
        HIPCHK(hipMalloc(&c->d_tw_inv, 4096 * 48));
        HIPCHK(hipMalloc(&c->d_g1tw_fwd, 64 * 32));
        HIPCHK(hipMalloc(&c->d_g1tw_inv, 64 * 32));
        build_twiddles_device(c->d_tw_fwd, c->d_tw_inv, c->d_g1tw_fwd, c->d_g1tw_inv, c->stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(c->stream));
        c->cells_ready = true;
    }
    if (n > c->cell_cap) {
        if (c->d_coeffs) (void)hipFree(c->d_coeffs);
        if (c->d_cells) (void)hipFree(c->d_cells);
        if (c->d_cell_st) (void)hipFree(c->d_cell_st);
        c->cell_cap = 0;
        HIPCHK(hipMalloc(&c->d_coeffs, n * 4096 * 48));
        HIPCHK(hipMalloc(&c->d_cells, n * 262144));
        HIPCHK(hipMalloc(&c->d_cell_st, n * sizeof(int)));
        c->cell_cap = n;
    }
    return MKZG_OK;
}

extern "C" {

// Context.ComputeCells for a device-resident batch: blobs n x 131072 B -> cells n x 128 x 2048 B
int mkzg_compute_cells_batch_device(mkzg_ctx* c, const void* d_blobs, size_t n, void* d_cells, int* d_status, void* stream) {
    if (!c || (n && (!d_blobs || !d_cells || !d_status))) return MKZG_ERR_BAD_ARGS;
    if (n == 0) return MKZG_OK;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    int rc = ensure_cells(c, 1);
    if (rc) return rc;
    cells_from_blobs_device((const uint8_t*)d_blobs, (int)n, c->d_tw_fwd, c->d_tw_inv, nullptr, (uint8_t*)d_cells, d_status, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return MKZG_OK;
}

int mkzg_compute_cells(mkzg_ctx* c, const uint8_t* blob, int num_threads, uint8_t* cells_out) {
    (void)num_threads;                                            // ignored by the reference too (api_eip7594.go:54)
    if (!c || !cells_out) return MKZG_ERR_BAD_ARGS;
    if (!blob) return MKZG_ERR_NIL_INPUT;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    int rc = ensure_cells(c, 1);
    if (rc) return rc;
    rc = ensure_staging(c, 1, MKZG_BYTES_PER_BLOB);
    if (rc) return rc;
    memcpy(c->h_stage, blob, MKZG_BYTES_PER_BLOB);
    HIPCHK(hipMemcpyAsync(c->d_blobs, c->h_stage, MKZG_BYTES_PER_BLOB, hipMemcpyHostToDevice, c->stream));
    rc = mkzg_compute_cells_batch_device(c, c->d_blobs, 1, c->d_cells, c->d_cell_st, c->stream);
    if (rc) return rc;
    int st = 0;
    HIPCHK(hipMemcpyAsync(&st, c->d_cell_st, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(cells_out, c->d_cells, 262144, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return st;
}

// Context.ComputeCellsAndKZGProofs for a device-resident batch (reference api_eip7594.go:28-52):
// blobs n x 131072 B -> cells n x 128 x 2048 B, proofs n x 128 x 48 B, per-item status
int mkzg_compute_cells_and_kzg_proofs_batch_device(mkzg_ctx* c, const void* d_blobs, size_t n, void* d_cells, void* d_proofs, int* d_status, void* stream) {
    if (!c || (n && (!d_blobs || !d_cells || !d_proofs || !d_status))) return MKZG_ERR_BAD_ARGS;
    if (n == 0) return MKZG_OK;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    const size_t chunk_max = 64;
    int rc = ensure_fk20(c, std::min(n, chunk_max));
    if (rc) return rc;
    rc = fk_ws_acquire(c, s);
    if (rc) return rc;
    size_t done = 0;
    while (done < n) {
        size_t chunk = std::min(n - done, c->fk_cap);
        const uint8_t* blobs = (const uint8_t*)d_blobs + done * MKZG_BYTES_PER_BLOB;
        KT(KT_CELLS_NTT, s, cells_from_blobs_device(blobs, (int)chunk, c->d_tw_fwd, c->d_tw_inv, c->d_coeffs, (uint8_t*)d_cells + done * 262144, d_status + done, s));
        KT(KT_FK20_SCALARS, s, fk20_scalars_device(c->d_coeffs, (int)chunk, c->d_tw_fwd, d_status + done, c->d_fk_scalars, s));
        KT(KT_FK20_MSM, s, fk20_msm_device(c->d_fk_scalars, c->d_fk_table, (int)chunk, d_status + done, c->d_fk_parts, c->d_fk_results, s));
        KT(KT_FK20_FFTS, s, fk20_ffts_device(c->d_fk_results, c->d_fk_results + c->fk_cap * 128 * 64, c->d_fk_plan, (int)chunk, d_status + done, (uint8_t*)d_proofs + done * 128 * 48, c->d_fk_parts, s));
        done += chunk;
    }
    HIPCHK(hipGetLastError());
    return fk_ws_release(c, s);
}

int mkzg_compute_cells_and_kzg_proofs(mkzg_ctx* c, const uint8_t* blob, int num_threads, uint8_t* cells_out, uint8_t* proofs_out) {
    (void)num_threads;
    if (!c || !cells_out || !proofs_out) return MKZG_ERR_BAD_ARGS;
    if (!blob) return MKZG_ERR_NIL_INPUT;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    int rc = ensure_fk20(c, 1);
    if (rc) return rc;
    rc = ensure_staging(c, 1, MKZG_BYTES_PER_BLOB);
    if (rc) return rc;
    memcpy(c->h_stage, blob, MKZG_BYTES_PER_BLOB);
    HIPCHK(hipMemcpyAsync(c->d_blobs, c->h_stage, MKZG_BYTES_PER_BLOB, hipMemcpyHostToDevice, c->stream));
    rc = mkzg_compute_cells_and_kzg_proofs_batch_device(c, c->d_blobs, 1, c->d_cells, c->d_fk_proofs, c->d_cell_st, c->stream);
    if (rc) return rc;
    int st = 0;
    HIPCHK(hipMemcpyAsync(&st, c->d_cell_st, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(cells_out, c->d_cells, 262144, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(proofs_out, c->d_fk_proofs, 128 * 48, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return st;
}

// Context.VerifyCellKZGProofBatch (reference api_eip7594.go:163-214 + internal/kzg_multi/kzg_verify.go:16-105).
// commitments: n_c x 48 B (one per cell, duplicates allowed); cell_indices: n_i; cells: n_cells host pointers;
// proofs: n_p x 48 B.  At most 8192 cells per call.
// The fold (everything up to the four group sums) and the finish (one pairing check) are separate so that multi-GPU jobs can
// shard the cells: a rank folds its cells with the batch-wide r and the GLOBAL exponents r^(exp_offset + k), ranks exchange
// one 512-byte record, and the sums add up.  sums: [0] sum r^k pi_k, [1] sum w_row C_row, [2] [I(tau)]G1,
// [3] sum r^k h_k^64 pi_k   (kzg_verify.go:32,41,68,80)
struct CellFold {
    G1HX sums[4];
};
static int cell_verify_fold(mkzg_ctx* c, const uint8_t* commitments48n, size_t n_c, const uint64_t* cell_indices, size_t n_i, const uint8_t* const* cells,
                            size_t n_cells, const uint8_t* proofs48n, size_t n_p, uint32_t exp_offset, const uint8_t* r32 /* big-endian, or null: draw it */, CellFold* out) {
    for (int m = 0; m < 4; m++) G1H::set_inf(out->sums[m]);
    if (!c) return MKZG_ERR_BAD_ARGS;
    // deduplicateKZGCommitments (api_eip7594.go:238-265)
    std::unordered_map<std::string, uint32_t> seen;
    std::vector<uint32_t> row_idx(n_c);
    std::vector<uint8_t> rows;
    for (size_t i = 0; i < n_c; i++) {
        // cells usually arrive blob-major: a commitment equal to its predecessor needs no hashing
        if (i > 0 && !memcmp(commitments48n + 48 * i, commitments48n + 48 * (i - 1), 48)) { row_idx[i] = row_idx[i - 1]; continue; }
        std::string key((const char*)commitments48n + 48 * i, 48);
        auto it = seen.find(key);
        if (it == seen.end()) {
            it = seen.emplace(key, (uint32_t)seen.size()).first;
            rows.insert(rows.end(), commitments48n + 48 * i, commitments48n + 48 * i + 48);
        }
        row_idx[i] = it->second;
    }
    const size_t n = n_c, n_rows = seen.size();
    if (n != n_i || n != n_cells || n != n_p) return MKZG_ERR_BATCH_LENGTH;          // api_eip7594.go:166-171
    if (n == 0) return MKZG_OK;                                                        // :173-175
    if (!cell_indices || !cells || !proofs48n) return MKZG_ERR_BAD_ARGS;
    for (size_t i = 0; i < n; i++) if (cell_indices[i] >= MKZG_CELLS_PER_EXT_BLOB) return MKZG_ERR_INVALID_CELL_ID;   // :184-188
    if (n > (size_t)VMSM_MAX_N) return MKZG_ERR_BAD_ARGS;
    if (c->g2_alpha64.lines.empty()) return MKZG_ERR_BAD_TRUSTED_SETUP;
    size_t first_nil = n;
    for (size_t i = 0; i < n; i++) if (!cells[i]) { first_nil = i; break; }
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    int rc = ensure_cells(c, 1);
    if (!rc) rc = ensure_monomial(c);
    if (!rc) rc = ensure_verify_ws(c, n);
    if (!rc) rc = ensure_cv_ws(c, n);
    if (!rc) rc = ensure_staging(c, 1, n * 2048);
    if (rc) return rc;
    const size_t cap = c->v_cap;
    uint32_t* h_misc = (uint32_t*)(c->h_v + 3 * cap * sizeof(int));
    // uploads, part 1: the points (small) -- their validation starts on the side stream before the 2 KiB cells are staged
    HIPCHK(hipMemcpyAsync(c->d_v_raw, rows.data(), n_rows * 48, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c->d_v_raw + cap * 48, proofs48n, n * 48, hipMemcpyHostToDevice, s));
    // points
    uint32_t* aff_rows = c->d_v_aff;
    uint32_t* aff_q = c->d_v_aff + cap * 32;
    int* st_rows = c->d_v_st;
    int* st_q = c->d_v_st + cap;
    // point decoding + subgroup checks on the side stream, beside the interpolation scalars (independent until the MSMs)
    hipStream_t s2 = c->side;
    HIPCHK(hipEventRecord(c->ev_fork, s));
    HIPCHK(hipStreamWaitEvent(s2, c->ev_fork, 0));
    // Decoding (0.43 ms) and the y-free subgroup test (0.97 ms) as two launches on two streams: the MSMs below need the decoded rows
    // only, so they start as soon as the scalars are ready instead of after the longer chain.  A point outside the subgroup is still a
    // point of the curve (the group law is total there), and its verdict makes the call fail before any sum is looked at.
    hipStream_t s3 = c->side2;
    int* sub_rows = c->d_cv_st + c->cv_cap;
    int* sub_q = c->d_cv_st + 2 * c->cv_cap;
    HIPCHK(hipStreamWaitEvent(s3, c->ev_fork, 0));
    KT(KT_DECOMPRESS, s2, g1_validate2_device(c->d_v_raw, (int)n_rows, aff_rows, st_rows, c->d_v_raw + cap * 48, (int)n, aff_q, st_q, s2, 1));
    HIPCHK(hipEventRecord(c->ev_decoded, s2));
    KT(KT_SUBGROUP, s3, g1_validate2_device(c->d_v_raw, (int)n_rows, nullptr, sub_rows, c->d_v_raw + cap * 48, (int)n, nullptr, sub_q, s3, 2));
    HIPCHK(hipEventRecord(c->ev_join, s3));
    // uploads, part 2 (host memcpy of n x 2 KiB into the pinned staging area runs while the GPU validates the points)
    {
        const std::vector<uint8_t> zero_cell(2048, 0);           // stands in for nil entries (reported after the device work)
        auto gather = [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; i++) memcpy(c->h_stage + i * 2048, cells[i] ? cells[i] : zero_cell.data(), 2048);
        };
        if (n >= 1024) {
            // 16 MiB for 8192 cells: gathered by eight host threads in chunks of 2048 cells, each chunk's copy to the device in flight
            // while the next one is gathered
            const size_t chunk = 2048;
            for (size_t lo = 0; lo < n; lo += chunk) {
                const size_t hi = std::min(n, lo + chunk), q = (hi - lo + 7) / 8;
                c->pool->parallel(8, [&](int i) { gather(std::min(hi, lo + (size_t)i * q), std::min(hi, lo + (size_t)(i + 1) * q)); });
                HIPCHK(hipMemcpyAsync(c->d_cv_cells + lo * 2048, c->h_stage + lo * 2048, (hi - lo) * 2048, hipMemcpyHostToDevice, s));
            }
        } else {
            gather(0, n);
            HIPCHK(hipMemcpyAsync(c->d_cv_cells, c->h_stage, n * 2048, hipMemcpyHostToDevice, s));
        }
        HIPCHK(hipMemcpyAsync(c->d_cv_cell_idx, cell_indices, n * 8, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(c->d_cv_row_idx, row_idx.data(), n * 4, hipMemcpyHostToDevice, s));
        if (r32) {
            for (int i = 0; i < 8; i++) h_misc[i] = ((uint32_t)r32[28 - 4 * i] << 24) | ((uint32_t)r32[29 - 4 * i] << 16) | ((uint32_t)r32[30 - 4 * i] << 8) | r32[31 - 4 * i];
        } else if (getrandom(h_misc, 32, 0) != 32) {
            return MKZG_ERR_HIP;
        }
        HIPCHK(hipMemcpyAsync(c->d_v_misc, h_misc, 32, hipMemcpyHostToDevice, s));
    }
    // scalars
    KT(KT_CELL_SCALARS, s, cell_verify_scalars_device(c->d_v_misc, exp_offset, c->d_cv_cell_idx, c->d_cv_row_idx, c->d_cv_cells, (int)n, (int)n_rows, c->d_tw_fwd, c->d_tw_inv, c->d_cv_rp,
                                                       c->d_v_s1, c->d_v_s3, c->d_cv_weights, c->d_cv_partials, c->d_cv_interp, c->d_cv_st, s));
    HIPCHK(hipStreamWaitEvent(s, c->ev_decoded, 0));
    // MSMs (kzg_verify.go:32,41,68,80): sum r^k pi_k ; sum w_row C_row ; sum interp_i [tau^i]G1 ; sum r^k h^64 pi_k
    const uint32_t* scal[4] = {c->d_v_s1, c->d_cv_weights, c->d_cv_interp, c->d_v_s3};
    const uint32_t* pts[4] = {aff_q, aff_rows, c->d_monomial_aff, aff_q};
    const int njob[4] = {(int)n, (int)n_rows, 64, (int)n};
    int vrc = 0;
    KT(KT_VMSM, s, vrc = vmsm_windows_device(scal, pts, njob, 4, c->d_v_misc + V_MISC_WORDS_C, c->d_v_misc + 16, s));
    if (vrc) return MKZG_ERR_BAD_ARGS;
    HIPCHK(hipGetLastError());
    int* h_st = (int*)c->h_v;
    std::vector<int> st_cells(n), h_sub_rows(n_rows), h_sub_q(n);
    HIPCHK(hipStreamWaitEvent(s, c->ev_join, 0));               // the subgroup verdicts
    HIPCHK(hipMemcpyAsync(h_st, c->d_v_st, 2 * cap * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(st_cells.data(), c->d_cv_st, n * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(h_sub_rows.data(), sub_rows, n_rows * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(h_sub_q.data(), sub_q, n * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(h_misc, c->d_v_misc, V_MISC_WORDS_C * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    // error precedence of api_eip7594.go:190-213: all commitments, then all proofs, then all cells
    for (size_t i = 0; i < n_rows; i++) {
        if (h_st[i]) return h_st[i];
        if (h_sub_rows[i]) return h_sub_rows[i];
    }
    for (size_t i = 0; i < n; i++) {
        if (h_st[cap + i]) return h_st[cap + i];
        if (h_sub_q[i]) return h_sub_q[i];
    }
    for (size_t i = 0; i < n; i++) {
        if (i == first_nil) return MKZG_ERR_NIL_INPUT;
        if (st_cells[i]) return st_cells[i];
    }
    // Horner over the window sums: four independent 255-doubling chains, one host thread each
    const uint32_t* wsum = h_misc + 16;
    auto horner = [&](int m) { g1h_horner_words(out->sums[m], wsum + (size_t)m * VMSM_WINDOWS * 48, VMSM_WINDOWS, VMSM_WINDOW_BITS); };
    {
        c->pool->parallel(4, [&](int m) { horner(m); });
    }
    return MKZG_OK;
}
// kzg_verify.go:85-104: e(sum r^k pi_k, [tau^64]G2) == e(sum w C - [I(tau)] + sum r^k h^64 pi_k, G2)
static int cell_verify_finish(mkzg_ctx* c, const CellFold& f) {
    G1HX rl, t;
    G1H::neg(t, f.sums[2]);
    G1H::add(rl, f.sums[1], t);
    G1H::add(rl, rl, f.sums[3]);
    G1AffH ps[2];
    g1h_to_affine(ps[0], f.sums[0]);
    g1h_to_affine(ps[1], rl);
    const G2Prepared* qs[2] = {&c->g2_alpha64, &c->g2_neg_gen};
    return pairing_check_prepared_2mt(ps, qs, c->pool) ? MKZG_OK : MKZG_VERIFY_FAILED;
}
int mkzg_verify_cell_kzg_proof_batch(mkzg_ctx* c, const uint8_t* commitments48n, size_t n_c, const uint64_t* cell_indices, size_t n_i, const uint8_t* const* cells,
                                     size_t n_cells, const uint8_t* proofs48n, size_t n_p) {
    if (!c) return MKZG_ERR_BAD_ARGS;
    if (n_c == n_i && n_c == n_cells && n_c == n_p && n_c == 0) return MKZG_OK;          // api_eip7594.go:173-175
    CellFold f;
    int rc = cell_verify_fold(c, commitments48n, n_c, cell_indices, n_i, cells, n_cells, proofs48n, n_p, 0, nullptr, &f);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    return cell_verify_finish(c, f);
}
// Sharded form (see CellFold): record = status@0 (int32), compressed sums[0..3] at 16, 64, 112, 160.
int mkzg_verify_cell_kzg_proof_batch_partial(mkzg_ctx* c, const uint8_t* commitments48n, size_t n_c, const uint64_t* cell_indices, size_t n_i, const uint8_t* const* cells,
                                             size_t n_cells, const uint8_t* proofs48n, size_t n_p, uint64_t offset, const uint8_t r32[32], uint8_t record512[512]) {
    if (!c || !r32 || !record512) return MKZG_ERR_BAD_ARGS;
    memset(record512, 0, 512);
    CellFold f;
    int rc = cell_verify_fold(c, commitments48n, n_c, cell_indices, n_i, cells, n_cells, proofs48n, n_p, (uint32_t)offset, r32, &f);
    int32_t st = rc;
    memcpy(record512, &st, 4);
    if (rc == MKZG_OK) {
        for (int m = 0; m < 4; m++) {
            G1AffH a;
            g1h_to_affine(a, f.sums[m]);
            g1h_compress(record512 + 16 + 48 * m, a);
        }
    }
    return rc;
}
int mkzg_verify_cell_kzg_proof_batch_finish(mkzg_ctx* c, const uint8_t* records512, size_t n_records) {
    if (!c || (n_records && !records512)) return MKZG_ERR_BAD_ARGS;
    if (c->g2_alpha64.lines.empty()) return MKZG_ERR_BAD_TRUSTED_SETUP;
    for (size_t k = 0; k < n_records; k++) {                       // first failing rank (lowest indices) decides the error
        int32_t st;
        memcpy(&st, records512 + 512 * k, 4);
        if (st) return st;
    }
    CellFold f;
    for (int m = 0; m < 4; m++) G1H::set_inf(f.sums[m]);
    for (size_t k = 0; k < n_records; k++) {
        for (int m = 0; m < 4; m++) {
            G1AffH a;
            G1HX t;
            if (!g1h_decompress(a, records512 + 512 * k + 16 + 48 * m)) return MKZG_ERR_BAD_ARGS;
            if (a.inf) continue;
            G1H::from_affine(t, a.x, a.y);
            G1H::add(f.sums[m], f.sums[m], t);
        }
    }
    std::lock_guard<std::recursive_mutex> g(c->mu);
    return cell_verify_finish(c, f);
}

// shared front end of RecoverCells / RecoverCellsAndComputeKZGProofs (reference api_eip7594.go:93-142) for n_blobs independent
// recoveries: per-blob validation in the reference's order, then ONE set of launches over the blobs that passed (blockIdx.y = blob),
// so that a batch runs the NTT stages and FK20 at batch efficiency (a single recovery is a chain of latency-bound launches).
// status[b] receives the reference's error for blob b; device outputs: c->d_cells (n x 128 x 2048 B), c->d_fk_proofs (n x 128 x 48 B).
static int recover_validate(const uint64_t* cell_ids, size_t n_ids, const uint8_t* const* cells, size_t n_cells) {
    if (n_ids != n_cells) return MKZG_ERR_NUM_CELL_IDS;                                  // :94-96
    if ((n_ids && !cell_ids) || (n_cells && !cells)) return MKZG_ERR_BAD_ARGS;
    for (size_t i = 1; i < n_ids; i++) if (cell_ids[i] <= cell_ids[i - 1]) return MKZG_ERR_CELL_IDS_NOT_ORDERED;   // :99-101
    for (size_t i = 0; i < n_ids; i++) if (cell_ids[i] >= MKZG_CELLS_PER_EXT_BLOB) return MKZG_ERR_FOUND_INVALID_CELL_ID;   // :104-108
    if (n_ids < 64) return MKZG_ERR_NOT_ENOUGH_CELLS;                                    // :111-113
    for (size_t i = 0; i < n_cells; i++) if (!cells[i]) return MKZG_ERR_NIL_INPUT;       // serialization.go:197-199
    return MKZG_OK;
}
static int recover_front(mkzg_ctx* c, size_t n_blobs, const uint64_t* const* cell_ids, const size_t* n_ids, const uint8_t* const* const* cells, const size_t* n_cells,
                         bool with_proofs, int* status) {
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    size_t total_cells = 0;
    for (size_t b = 0; b < n_blobs; b++) {
        status[b] = recover_validate(cell_ids[b], n_ids[b], cells[b], n_cells[b]);
        if (!status[b]) total_cells += n_cells[b];
    }
    int rc = with_proofs ? ensure_fk20(c, n_blobs) : ensure_cells(c, n_blobs);
    if (!rc) rc = ensure_staging(c, 1, std::max<size_t>(total_cells, 1) * 2048);
    if (rc) return rc;
    rc = fk_ws_acquire(c, s);
    if (rc) return rc;
    if (!c->rec_ready) {
        HIPCHK(hipMalloc(&c->d_pow7, 8192 * 48));
        HIPCHK(hipMalloc(&c->d_ipow7, 4096 * 48));
        recover_tables_device(c->d_pow7, c->d_ipow7, s);
        c->rec_ready = true;
    }
    if (n_blobs > c->rec_cap) {
        if (c->d_rec_scratch) (void)hipFree(c->d_rec_scratch);
        if (c->d_rec_pos) (void)hipFree(c->d_rec_pos);
        c->d_rec_scratch = nullptr; c->d_rec_pos = nullptr; c->rec_cap = 0;
        HIPCHK(hipMalloc(&c->d_rec_scratch, n_blobs * recover_scratch_words() * 4));
        HIPCHK(hipMalloc(&c->d_rec_pos, n_blobs * (192 + 1) * sizeof(int)));
        c->rec_cap = n_blobs;
    }
    // positions: per blob 128 cell positions (global index into the staged cells, -1 = missing) + up to 64 missing ids; counts after them
    std::vector<int> pos(n_blobs * 193, -1);
    int* nmiss = pos.data() + n_blobs * 192;
    size_t at = 0;
    for (size_t b = 0; b < n_blobs; b++) {
        int* pb = pos.data() + b * 192;
        nmiss[b] = 0;
        if (status[b]) {                      // rejected blob: give the kernels a well-formed dummy (all cells "missing" is not valid, so
            for (int i = 0; i < 128; i++) pb[i] = -1;      // mark everything absent and no missing ids: the stages then run on zeros, output ignored)
            continue;
        }
        for (size_t i = 0; i < n_ids[b]; i++) pb[cell_ids[b][i]] = (int)(at + i);
        for (uint32_t cid = 0; cid < 128; cid++) {
            if (pb[cid] < 0) {
                uint32_t r = 0;
                for (int q = 0; q < 7; q++) if (cid & (1u << q)) r |= 1u << (6 - q);       // BitReverseInt(cellID, 128)  (:117-122)
                pb[128 + nmiss[b]++] = (int)r;
            }
        }
        for (size_t i = 0; i < n_cells[b]; i++) memcpy(c->h_stage + (at + i) * 2048, cells[b][i], 2048);
        at += n_cells[b];
    }
    if (total_cells) HIPCHK(hipMemcpyAsync(c->d_blobs, c->h_stage, total_cells * 2048, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c->d_rec_pos, pos.data(), pos.size() * sizeof(int), hipMemcpyHostToDevice, s));
    KT(KT_RECOVER, s, recover_coeffs_device(c->d_blobs, c->d_rec_pos, c->d_rec_pos + n_blobs * 192, (int)n_blobs, c->d_tw_fwd, c->d_tw_inv, c->d_pow7, c->d_ipow7,
                                            c->d_rec_scratch, c->d_coeffs, c->d_cell_st, s));
    KT(KT_CELLS_NTT, s, cells_from_coeffs_device(c->d_coeffs, (int)n_blobs, c->d_tw_fwd, c->d_tw_inv, c->d_cells, c->d_cell_st, s));
    if (with_proofs) {
        KT(KT_FK20_SCALARS, s, fk20_scalars_device(c->d_coeffs, (int)n_blobs, c->d_tw_fwd, c->d_cell_st, c->d_fk_scalars, s));
        KT(KT_FK20_MSM, s, fk20_msm_device(c->d_fk_scalars, c->d_fk_table, (int)n_blobs, c->d_cell_st, c->d_fk_parts, c->d_fk_results, s));
        KT(KT_FK20_FFTS, s, fk20_ffts_device(c->d_fk_results, c->d_fk_results + c->fk_cap * 128 * 64, c->d_fk_plan, (int)n_blobs, c->d_cell_st, c->d_fk_proofs, c->d_fk_parts, s));
    }
    HIPCHK(hipGetLastError());
    // device-side statuses (non-canonical field elements in the cells) join the host-side ones
    std::vector<int> dst(n_blobs);
    HIPCHK(hipMemcpyAsync(dst.data(), c->d_cell_st, n_blobs * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (size_t b = 0; b < n_blobs; b++) if (!status[b]) status[b] = dst[b];
    return MKZG_OK;
}

// Context.RecoverCellsAndComputeKZGProofs (reference api_eip7594.go:144-161)
int mkzg_recover_cells_and_kzg_proofs(mkzg_ctx* c, const uint64_t* cell_ids, size_t n_ids, const uint8_t* const* cells, size_t n_cells, int num_threads, uint8_t* cells_out,
                                      uint8_t* proofs_out) {
    (void)num_threads;
    if (!c || !cells_out || !proofs_out || (n_ids && !cell_ids) || (n_cells && !cells)) return MKZG_ERR_BAD_ARGS;
    int st = recover_validate(cell_ids, n_ids, cells, n_cells);
    if (st) return st;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    int rc = recover_front(c, 1, &cell_ids, &n_ids, &cells, &n_cells, true, &st);
    if (rc) return rc;
    if (st) return st;
    HIPCHK(hipMemcpyAsync(cells_out, c->d_cells, 262144, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(proofs_out, c->d_fk_proofs, 128 * 48, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return MKZG_OK;
}
// n_blobs independent recoveries in one call (no reference analogue; throughput form of the above, used by sharded recovery jobs).
// cell_ids[b] / cells[b]: the n_ids[b] = n_cells[b] supplied cells of blob b.  cells_out: n_blobs x 128 x 2048 B, proofs_out: n_blobs x 128 x 48 B,
// status[b]: the reference's error for blob b (outputs of a failed blob are left untouched).  Returns the first non-zero status.
int mkzg_recover_cells_and_kzg_proofs_batch(mkzg_ctx* c, size_t n_blobs, const uint64_t* const* cell_ids, const size_t* n_ids, const uint8_t* const* const* cells,
                                            const size_t* n_cells, uint8_t* cells_out, uint8_t* proofs_out, int* status) {
    if (!c || (n_blobs && (!cell_ids || !n_ids || !cells || !n_cells || !cells_out || !proofs_out || !status))) return MKZG_ERR_BAD_ARGS;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    int first = MKZG_OK;
    for (size_t done = 0; done < n_blobs; done += 64) {                 // pieces of 64 blobs bound the workspaces
        const size_t m = std::min<size_t>(64, n_blobs - done);
        int rc = recover_front(c, m, cell_ids + done, n_ids + done, cells + done, n_cells + done, true, status + done);
        if (rc) return rc;
        for (size_t b = 0; b < m; b++) {
            if (status[done + b]) { if (!first) first = status[done + b]; continue; }
            HIPCHK(hipMemcpyAsync(cells_out + (done + b) * 262144, c->d_cells + b * 262144, 262144, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipMemcpyAsync(proofs_out + (done + b) * 128 * 48, c->d_fk_proofs + b * 128 * 48, 128 * 48, hipMemcpyDeviceToHost, c->stream));
        }
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return first;
}
// Context.RecoverCells (reference api_eip.go:8-15)
int mkzg_recover_cells(mkzg_ctx* c, const uint64_t* cell_ids, size_t n_ids, const uint8_t* const* cells, size_t n_cells, int num_threads, uint8_t* cells_out) {
    (void)num_threads;
    if (!c || !cells_out || (n_ids && !cell_ids) || (n_cells && !cells)) return MKZG_ERR_BAD_ARGS;
    int st = recover_validate(cell_ids, n_ids, cells, n_cells);
    if (st) return st;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    int rc = recover_front(c, 1, &cell_ids, &n_ids, &cells, &n_cells, false, &st);
    if (rc) return rc;
    if (st) return st;
    HIPCHK(hipMemcpyAsync(cells_out, c->d_cells, 262144, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return MKZG_OK;
}

}  // extern "C"
