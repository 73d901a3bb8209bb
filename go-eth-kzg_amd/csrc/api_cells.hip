// api_cells.hip -- C ABI entry points of the EIP-7594 cell paths (include/mi355kzg.h):
//   Context.ComputeCells                  reference api_eip7594.go:12-26
//   Context.ComputeCellsAndKZGProofs      reference api_eip7594.go:28-52
//   Context.VerifyCellKZGProofBatch       reference api_eip7594.go:163-214
//   Context.RecoverCellsAndComputeKZGProofs / RecoverCells   reference api_eip7594.go:144-161, api_eip.go:8-15
// All field work happens in cells_kernels.hip / fk20_kernels.hip / recover_kernels.hip; the host stages bytes, orders the
// reference's checks and finishes the verifier with one pairing.  Every call works in one Lane (ctx.h).
#include <sys/random.h>
#include <functional>
#include <string>
#include <thread>
#include <unordered_map>
#include "ctx.h"
#include "host_validate_dev.hpp"
#include "api_multi.h"

static const size_t V_MISC_WORDS_C = 16 + 4 * VMSM_WINDOWS * 48;

// ---------------------------------------------------------------------------------------------
// shared tables
// ---------------------------------------------------------------------------------------------
// NTT twiddles (4096 Fr rows each way) and the G1 transforms' twiddles as plain scalars
int ensure_twiddles(mkzg_ctx* c) {
    if (c->twiddles_ready.load(std::memory_order_acquire)) return MKZG_OK;
    std::lock_guard<std::mutex> g(c->init_mu);
    if (c->twiddles_ready.load(std::memory_order_relaxed)) return MKZG_OK;
    DeviceScope dev(c->device);                  // shared tables live on the context's device whatever device the calling thread is on
    int rc = dev_realloc(c->d_tw_fwd, 4096 * 48);
    if (!rc) rc = dev_realloc(c->d_tw_inv, 4096 * 48);
    if (!rc) rc = dev_realloc(c->d_g1tw_fwd, 64 * 32);
    if (!rc) rc = dev_realloc(c->d_g1tw_inv, 64 * 32);
    if (rc) return rc;
    build_twiddles_device(c->d_tw_fwd, c->d_tw_inv, c->d_g1tw_fwd, c->d_g1tw_inv, c->init_stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->init_stream));
    c->twiddles_ready.store(true, std::memory_order_release);
    return MKZG_OK;
}
// monomial G1 SRS on the device (decoded without subgroup checks, trusted_setup.go:104-117)
int ensure_monomial(mkzg_ctx* c) {
    if (c->monomial_ready.load(std::memory_order_acquire)) return MKZG_OK;
    std::lock_guard<std::mutex> g(c->init_mu);
    if (c->monomial_ready.load(std::memory_order_relaxed)) return MKZG_OK;
    DeviceScope dev(c->device);
    uint8_t* d_raw = nullptr;
    int* d_st = nullptr;
    int rc = dev_realloc(d_raw, 4096 * 48);
    if (!rc) rc = dev_realloc(d_st, 4096 * sizeof(int));
    if (!rc) rc = dev_realloc(c->d_monomial_aff, 4096 * 128);
    std::vector<int> st(4096, 0);
    if (!rc) {
        rc = [&]() -> int {
            HIPCHK(hipMemcpyAsync(d_raw, c->g1_monomial.data(), 4096 * 48, hipMemcpyHostToDevice, c->init_stream));
            g1_decompress_device(d_raw, 4096, c->d_monomial_aff, d_st, c->init_stream);
            HIPCHK(hipMemcpyAsync(st.data(), d_st, 4096 * sizeof(int), hipMemcpyDeviceToHost, c->init_stream));
            HIPCHK(hipStreamSynchronize(c->init_stream));
            return MKZG_OK;
        }();
    }
    dev_free(d_raw);
    dev_free(d_st);
    if (rc) return rc;
    for (int v : st) if (v) { dev_free(c->d_monomial_aff); return MKZG_ERR_BAD_TRUSTED_SETUP; }
    c->monomial_ready.store(true, std::memory_order_release);
    return MKZG_OK;
}
// fk20.NewFK20 (reference fk20.go:23-52, called from api.go:137): done lazily on first use.
// The direct window table is sized for an otherwise empty 288 GB part; like the commitment table it DEGRADES instead of failing when
// another tenant holds the HBM (the reference's ComputeCellsAndKZGProofs / recovery never fail for resource reasons,
// api_eip7594.go:28-52,144-161): 13-bit windows (86 GB) if asked for, else 11-bit (25.8 GB), else 8-bit (4.3 GB: 32 instead of 24
// additions per scalar) -- same bytes out at every width; "fk20_bits_effective" reports what is in use.  Only when not even the
// 4.3 GB table fits does the call fail (MKZG_ERR_HIP).
int ensure_fk20_tables(mkzg_ctx* c) {
    if (c->fk_ready.load(std::memory_order_acquire)) return MKZG_OK;
    int rc = ensure_twiddles(c);
    if (!rc) rc = ensure_monomial(c);
    if (rc) return rc;
    std::lock_guard<std::mutex> g(c->init_mu);
    if (c->fk_ready.load(std::memory_order_relaxed)) return MKZG_OK;
    DeviceScope dev(c->device);
    rc = dev_realloc(c->d_fk_t_aff, 8192 * 128);
    if (rc) return rc;
    dev_free(c->d_fk_table);                                   // a block left behind by an attempt that failed after its allocation
    c->fk20_bits_built = 0;
    static const int widths[4] = {13, 12, 11, 8};
    for (int bits : widths) {
        if (bits > c->opt_fk20_bits) continue;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
        // headroom: the lanes' workspaces (0.1-0.6 GB per busy lane) must still fit beside the table; the 86 GB table is a luxury and keeps 8 GB free
        const size_t headroom = bits == 13 ? (size_t)8 << 30 : (size_t)1 << 30;
        if (free_b && fk20_table_bytes(bits) + headroom > free_b) continue;
        if (hipMalloc(&c->d_fk_table, fk20_table_bytes(bits)) != hipSuccess) { (void)hipGetLastError(); c->d_fk_table = nullptr; continue; }
        fk20_setup_device(c->d_monomial_aff, c->d_g1tw_fwd, c->d_fk_t_aff, c->d_fk_table, bits, c->init_stream);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->init_stream) != hipSuccess) {
            (void)hipGetLastError();
            dev_free(c->d_fk_table);
            tl_hip_error = true;
            return MKZG_ERR_HIP;
        }
        if (bits != c->opt_fk20_bits)
            fprintf(stderr, "[mi355kzg] FK20 window table: %d bits requested, %d bits (%.1f GB) fit in free HBM\n", c->opt_fk20_bits, bits, (double)fk20_table_bytes(bits) / 1e9);
        c->fk20_bits_built = bits;
        c->fk_ready.store(true, std::memory_order_release);
        return MKZG_OK;
    }
    fprintf(stderr, "[mi355kzg] no FK20 window table fits in free HBM (the smallest is %.1f GB)\n", (double)fk20_table_bytes(8) / 1e9);
    return MKZG_ERR_HIP;
}
int ensure_rec_tables(mkzg_ctx* c) {
    if (c->rec_ready.load(std::memory_order_acquire)) return MKZG_OK;
    int rc = ensure_twiddles(c);
    if (rc) return rc;
    std::lock_guard<std::mutex> g(c->init_mu);
    if (c->rec_ready.load(std::memory_order_relaxed)) return MKZG_OK;
    DeviceScope dev(c->device);
    rc = dev_realloc(c->d_pow7, 8192 * 48);
    if (!rc) rc = dev_realloc(c->d_ipow7, 4096 * 48);
    if (rc) return rc;
    recover_tables_device(c->d_pow7, c->d_ipow7, c->init_stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->init_stream));
    c->rec_ready.store(true, std::memory_order_release);
    return MKZG_OK;
}

// ---------------------------------------------------------------------------------------------
// per-lane workspaces
// ---------------------------------------------------------------------------------------------
int ensure_cells_ws(Lane* L, size_t n) {
    if (n <= L->cell_cap) return MKZG_OK;
    L->cell_cap = 0;
    int rc = dev_realloc(L->d_coeffs, n * 4096 * 48);
    if (!rc) rc = dev_realloc(L->d_cells, n * 262144);
    if (!rc) rc = dev_realloc(L->d_cell_st, n * sizeof(int));
    if (rc) return rc;
    L->cell_cap = n;
    return MKZG_OK;
}
int ensure_fk20_ws(Lane* L, size_t n) {
    int rc = ensure_cells_ws(L, n);
    if (rc) return rc;
    {   // the segmented products' state (calls of 65..192 blobs: 109 / 218 / 327 MB); without it the products run whole, which is slower but the same bytes
        const size_t want = fk20_seg_count((int)n) ? fk20_seg_bytes((int)n) : 0;
        if (want > L->fk_seg_bytes) {
            L->fk_seg_bytes = 0;
            if (L->d_fk_seg) { (void)hipFree(L->d_fk_seg); L->d_fk_seg = nullptr; }
            if (hipMalloc(&L->d_fk_seg, want) == hipSuccess) L->fk_seg_bytes = want;
            else { (void)hipGetLastError(); L->d_fk_seg = nullptr; }
        }
    }
    if (n <= L->fk_cap) return MKZG_OK;
    L->fk_cap = 0;
    rc = dev_realloc(L->d_fk_scalars, n * 128 * 64 * 32);
    if (!rc) rc = dev_realloc(L->d_fk_results, 2 * n * 128 * 256);
    if (!rc) rc = dev_realloc(L->d_fk_parts, std::max<size_t>(n, 2 * FK_SPLIT_MAX_BATCH) * fk20_parts_rows() * 256);   // calls of up to FK_SPLIT_MAX_BATCH blobs use twice the rows per blob
    if (!rc) rc = dev_realloc(L->d_fk_proofs, n * 128 * 48);
    if (rc) return rc;
    L->fk_cap = n;
    return MKZG_OK;
}
#ifndef MKZG_CV_SUBGROUP_WIDE_FROM
// points (rows + proofs) from which the cell verifier's r-torsion test takes one lane per point; 0: never.  *Measured* (tools/
// bench_cellverify_inflight.py with CV_BLOBS, quad / one lane, ms per call from one caller and M cells/s from six): 1024 cells 1.71 / 2.00
// and 2.06 / 1.44; 2048 cells 1.92 / 1.86 and 2.70 / 3.55; 4096: 2.18 / 1.93 and 3.60 / 4.04; 6144: 2.48 / 2.42 and 4.91 / 6.51;
// 8192: 2.88-2.91 / 2.26-2.50 and 5.7 / 6.7-7.0
#define MKZG_CV_SUBGROUP_WIDE_FROM 2048
#endif
// n_cells: cells the lane stages itself (host-pointer calls; 0 when the caller's cells are already in HBM); n: entries of the call
static int ensure_cv_ws(Lane* L, size_t n_cells, size_t n) {
    if (n_cells > L->cv_cells_cap) {
        L->cv_cells_cap = 0;
        const size_t cc = std::max<size_t>(n_cells, 128);
        int rc = dev_realloc(L->d_cv_cells, cc * 2048);
        if (rc) return rc;
        L->cv_cells_cap = cc;
    }
    if (n <= L->cv_cap) return MKZG_OK;
    L->cv_cap = 0;
    const size_t cap = std::max<size_t>(n, 128);
    int rc = dev_realloc(L->d_cv_in, cap * (48 + 48 + 8 + 4));
    if (!rc) rc = dev_realloc(L->d_cv_rp, cap * 48);
    if (!rc) rc = dev_realloc(L->d_cv_weights, cap * 32);
    if (!rc) rc = dev_realloc_zero(L->d_cv_partials, CELL_INTERP_SCRATCH_BYTES);      // the tickets at its end
    if (!rc) rc = dev_realloc(L->d_cv_interp, 64 * 32);
    if (rc) return rc;
    L->cv_cap = cap;
    return MKZG_OK;
}
static int ensure_rec_ws(Lane* L, size_t n_blobs) {
    if (n_blobs <= L->rec_cap) return MKZG_OK;
    L->rec_cap = 0;
    int rc = dev_realloc(L->d_rec_scratch, n_blobs * recover_scratch_words() * 4);
    if (!rc) rc = dev_realloc(L->d_rec_pos, n_blobs * (192 + 1) * sizeof(int));
    if (rc) return rc;
    L->rec_cap = n_blobs;
    return MKZG_OK;
}

// cells + proofs of n device-resident blobs in pieces of the lane's FK20 workspace; asynchronous on s
static int cells_and_proofs_device(mkzg_ctx* c, Lane* L, const uint8_t* d_blobs, size_t n, uint8_t* d_cells, uint8_t* d_proofs, int* d_status, hipStream_t s) {
    // Pieces of at most cells_chunk blobs, cut evenly.  The MSM is VALU-bound at any size from 64 blobs on, but the transform stage is a
    // latency-bound round of 686 wavefronts whatever the piece holds up to 64 blobs per wavefront row: 2.3 ms for 64 blobs, 5.9 ms for
    // 256 (measured, tools/sweep_cells.py: 256 blobs in 14.6 ms as one piece against 18.5 ms as four).
    const size_t chunk_max = (size_t)c->opt_cells_chunk, pieces = (n + chunk_max - 1) / chunk_max, even = (n + pieces - 1) / pieces;
    int rc = ensure_fk20_tables(c);
    if (!rc) rc = ensure_fk20_ws(L, even);
    if (rc) return rc;
    size_t done = 0;
    while (done < n) {
        const size_t chunk = std::min(n - done, even);
        const uint8_t* blobs = d_blobs + done * MKZG_BYTES_PER_BLOB;
        KT(KT_CELLS_NTT, s, cells_from_blobs_device(blobs, (int)chunk, c->d_tw_fwd, c->d_tw_inv, L->d_coeffs, d_cells + done * 262144, d_status + done, s));
        KT(KT_FK20_SCALARS, s, fk20_scalars_device(L->d_coeffs, (int)chunk, c->d_tw_fwd, d_status + done, L->d_fk_scalars, s));
        KT(KT_FK20_MSM, s, fk20_msm_device(L->d_fk_scalars, c->d_fk_table, c->fk20_bits_built, (int)chunk, d_status + done, L->d_fk_parts, L->d_fk_results, s));
        KT(KT_FK20_FFTS, s, fk20_ffts_device(L->d_fk_results, L->d_fk_results + L->fk_cap * 128 * 64, (int)chunk, d_status + done, d_proofs + done * 128 * 48, L->d_fk_parts, c->opt_fk20_segments.load(std::memory_order_relaxed) && fk20_seg_bytes((int)chunk) <= L->fk_seg_bytes ? L->d_fk_seg : nullptr, s));
        done += chunk;
    }
    HIPCHK(hipGetLastError());
    return MKZG_OK;
}

extern "C" {

// Context.ComputeCells for a device-resident batch: blobs n x 131072 B -> cells n x 128 x 2048 B
int mkzg_compute_cells_batch_device(mkzg_ctx* c, const void* d_blobs, size_t n, void* d_cells, int* d_status, void* stream) {
    if (!c || (n && (!d_blobs || !d_cells || !d_status))) return MKZG_ERR_BAD_ARGS;
    if (!c->peers.empty() && !(c = ctx_for_pointer(c, d_blobs))) return MKZG_ERR_BAD_ARGS;
    if (n == 0) return MKZG_OK;
    int rc = ensure_twiddles(c);
    if (rc) return rc;
    LaneGuard G(c, (hipStream_t)stream);
    Lane* L = G.L;
    KT(KT_CELLS_NTT, G.s, cells_from_blobs_device((const uint8_t*)d_blobs, (int)n, c->d_tw_fwd, c->d_tw_inv, nullptr, (uint8_t*)d_cells, d_status, G.s));
    HIPCHK(hipGetLastError());
    return MKZG_OK;
}

int mkzg_compute_cells(mkzg_ctx* c, const uint8_t* blob, int num_threads, uint8_t* cells_out) {
    (void)num_threads;                                            // ignored by the reference too (api_eip7594.go:54)
    if (!c || !cells_out) return MKZG_ERR_BAD_ARGS;
    if (!blob) return MKZG_ERR_NIL_INPUT;
    c = md_pick(c);
    int rc = ensure_twiddles(c);
    if (rc) return rc;
    LaneGuard G(c);
    Lane* L = G.L;
    rc = ensure_cells_ws(L, 1);
    if (!rc) rc = ensure_staging(L, 1, MKZG_BYTES_PER_BLOB);
    if (rc) return G.fail(rc);
    memcpy(L->h_stage, blob, MKZG_BYTES_PER_BLOB);
    HIPCHK(hipMemcpyAsync(L->d_blobs, L->h_stage, MKZG_BYTES_PER_BLOB, hipMemcpyHostToDevice, G.s));
    KT(KT_CELLS_NTT, G.s, cells_from_blobs_device(L->d_blobs, 1, c->d_tw_fwd, c->d_tw_inv, nullptr, L->d_cells, L->d_cell_st, G.s));
    HIPCHK(hipGetLastError());
    int st = 0;
    HIPCHK(hipMemcpyAsync(&st, L->d_cell_st, sizeof(int), hipMemcpyDeviceToHost, G.s));
    HIPCHK(hipMemcpyAsync(cells_out, L->d_cells, 262144, hipMemcpyDeviceToHost, G.s));
    LANE_SYNC(c, G.L, G.s);
    G.L->drained = true;
    return st;
}

// Context.ComputeCellsAndKZGProofs for a device-resident batch (reference api_eip7594.go:28-52):
// blobs n x 131072 B -> cells n x 128 x 2048 B, proofs n x 128 x 48 B, per-item status
int mkzg_compute_cells_and_kzg_proofs_batch_device(mkzg_ctx* c, const void* d_blobs, size_t n, void* d_cells, void* d_proofs, int* d_status, void* stream) {
    if (!c || (n && (!d_blobs || !d_cells || !d_proofs || !d_status))) return MKZG_ERR_BAD_ARGS;
    if (!c->peers.empty() && !(c = ctx_for_pointer(c, d_blobs))) return MKZG_ERR_BAD_ARGS;
    if (n == 0) return MKZG_OK;
    LaneGuard G(c, (hipStream_t)stream);
    return G.fail(cells_and_proofs_device(c, G.L, (const uint8_t*)d_blobs, n, (uint8_t*)d_cells, (uint8_t*)d_proofs, d_status, G.s));
}

}   // extern "C"
// the cells and proofs of n blobs given as host pointers, in one lane: the blobs staged side by side, ONE pass of the batch kernels,
// every item's 262,144 + 6,144 bytes straight into its caller's buffers.  rcs[i]: the item's own status.
static int cells_and_proofs_host(mkzg_ctx* c, size_t n, const uint8_t* const* blobs, uint8_t* const* cells_out, uint8_t* const* proofs_out, int* rcs) {
    LaneGuard G(c);
    Lane* L = G.L;
    const size_t per_item = MKZG_BYTES_PER_BLOB + 262144 + 128 * 48 + 64;    // staged: the blob in, its cells and proofs out (pinned both ways)
    int rc = ensure_fk20_tables(c);
    if (!rc) rc = ensure_fk20_ws(L, n);
    if (!rc) rc = ensure_staging(L, n, per_item);
    if (rc) return G.fail(rc);
    uint8_t* h_cells = L->h_stage + n * MKZG_BYTES_PER_BLOB;
    uint8_t* h_proofs = h_cells + n * 262144;
    int* h_st = (int*)(h_proofs + n * 128 * 48);
    if (n == 1) memcpy(L->h_stage, blobs[0], MKZG_BYTES_PER_BLOB);
    else c->pool->parallel((int)n, [&](int i) { copy_to_staging(L->h_stage + (size_t)i * MKZG_BYTES_PER_BLOB, blobs[i], MKZG_BYTES_PER_BLOB); });
    HIPCHK(hipMemcpyAsync(L->d_blobs, L->h_stage, n * MKZG_BYTES_PER_BLOB, hipMemcpyHostToDevice, G.s));
    rc = cells_and_proofs_device(c, L, L->d_blobs, n, L->d_cells, L->d_fk_proofs, L->d_cell_st, G.s);
    if (rc) return G.fail(rc);
    HIPCHK(hipMemcpyAsync(h_st, L->d_cell_st, n * sizeof(int), hipMemcpyDeviceToHost, G.s));
    if (n == 1) {                                                            // a lone call: straight into the caller's buffers, as before
        HIPCHK(hipMemcpyAsync(cells_out[0], L->d_cells, 262144, hipMemcpyDeviceToHost, G.s));
        HIPCHK(hipMemcpyAsync(proofs_out[0], L->d_fk_proofs, 128 * 48, hipMemcpyDeviceToHost, G.s));
    } else {                                                                 // a group: two copies into pinned memory, then the pool hands them out
        HIPCHK(hipMemcpyAsync(h_cells, L->d_cells, n * 262144, hipMemcpyDeviceToHost, G.s));
        HIPCHK(hipMemcpyAsync(h_proofs, L->d_fk_proofs, n * 128 * 48, hipMemcpyDeviceToHost, G.s));
    }
    LANE_SYNC(c, G.L, G.s);
    G.L->drained = true;
    if (n > 1)
        c->pool->parallel((int)n, [&](int i) {
            if (h_st[i]) return;
            memcpy(cells_out[i], h_cells + (size_t)i * 262144, 262144);
            memcpy(proofs_out[i], h_proofs + (size_t)i * 128 * 48, 128 * 48);
        });
    for (size_t i = 0; i < n; i++) rcs[i] = h_st[i];
    return MKZG_OK;
}
extern "C" {
// Context.ComputeCellsAndKZGProofs (reference api_eip7594.go:28-52).  Calls that find other such calls in flight share one pass of the
// batch kernels (host_combine.hpp): the FK20 stage takes 2.0 ms for one blob and 4.8 ms for sixty-four.
int mkzg_compute_cells_and_kzg_proofs(mkzg_ctx* c, const uint8_t* blob, int num_threads, uint8_t* cells_out, uint8_t* proofs_out) {
    (void)num_threads;
    if (!c || !cells_out || !proofs_out) return MKZG_ERR_BAD_ARGS;
    if (!blob) return MKZG_ERR_NIL_INPUT;
    c = md_pick(c);
    mkzg_ctx::CellsCall call{blob, cells_out, proofs_out, MKZG_OK};
    auto process = [c](std::vector<mkzg_ctx::CellsCall*>& items) {
        const size_t n = items.size();
        if (n > 1) c->comb_cells.count_combined(n);
        std::vector<const uint8_t*> in(n);
        std::vector<uint8_t*> oc(n), op(n);
        std::vector<int> rcs(n, MKZG_OK);
        for (size_t i = 0; i < n; i++) { in[i] = items[i]->blob; oc[i] = items[i]->cells_out; op[i] = items[i]->proofs_out; }
        const int rc = cells_and_proofs_host(c, n, in.data(), oc.data(), op.data(), rcs.data());
        for (size_t i = 0; i < n; i++) items[i]->rc = rc ? rc : rcs[i];
    };
    if (c->opt_combine.load(std::memory_order_relaxed)) {
        c->comb_cells.run(&call, process);
    } else {
        std::vector<mkzg_ctx::CellsCall*> one{&call};
        process(one);
    }
    return call.rc;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// Context.VerifyCellKZGProofBatch (reference api_eip7594.go:163-214 + internal/kzg_multi/kzg_verify.go:16-105).
// The fold (everything up to the four group sums) and the finish (one pairing check) are separate so that the cells can be sharded --
// over pieces of one call (any batch size), over the devices of a multi-device context, over the ranks of a multi-process job: a
// shard folds its cells with the batch-wide r and the GLOBAL exponents r^(exp_offset + k), and the sums add up.
// sums: [0] sum r^k pi_k, [1] sum w_row C_row, [2] [I(tau)]G1, [3] sum r^k h_k^64 pi_k   (kzg_verify.go:32,41,68,80)
// ---------------------------------------------------------------------------------------------
static const size_t CELL_PIECE = VMSM_MAX_N;       // cells per fold (the variable-base MSM kernels take up to 8192 points per job)

// one piece (n <= CELL_PIECE) in lane L; r_words: the batch-wide scalar as 8 little-endian words
// cells: host pointers (the reference's []*Cell), or d_cells: the piece's n x 2048 B contiguous in HBM (no gather, no upload; s is then the
// caller's stream)
static int cell_fold_piece(mkzg_ctx* c, Lane* L, const uint8_t* commitments48n, const uint64_t* cell_indices, const uint8_t* const* cells, const uint8_t* d_cells,
                           const uint8_t* proofs48n, size_t n, uint32_t exp_offset, const uint32_t r_words_in[8], hipStream_t s, CellFold* out, RlcLate* late = nullptr) {
    size_t first_nil = n;
    if (!d_cells) for (size_t i = 0; i < n; i++) if (!cells[i]) { first_nil = i; break; }
    int rc = ensure_verify_ws(L, n);
    if (!rc) rc = ensure_cv_ws(L, d_cells ? 0 : n, n);
    if (!rc) rc = ensure_staging(L, 1, (d_cells ? 0 : n * 2048) + n * 48 + n * 48 + n * 12 + 64);      // (rows: at most n of them)
    if (rc) return rc;
    // The cells (n x 2 KiB behind scattered host pointers) are gathered into the pinned staging area by the host pool -- all chunks' tasks
    // queued at once, FIRST THING in the call (round 5: the timeline of a lone call showed its first upload 0.22 ms after its start, behind
    // the commitments' deduplication and seven API calls of this thread; `tools/ab_r05_38.sh`): the pool runs ahead while this thread
    // deduplicates, uploads the small arrays and launches the points' kernels, and waits for chunk k only when it gets there.
    const size_t CH = 2048;
    const size_t nch = d_cells ? 0 : (n + CH - 1) / CH;
    const int TPC = n >= 256 ? 4 : 1;
    std::vector<Countdown> left(nch);
    for (auto& a : left) a.set(TPC);
    Countdown all_left((int)nch * TPC);
    const std::function<void(int)> gather = [&](int t) {
        const size_t k = (size_t)(t / TPC), lo = k * CH, hi = std::min(n, lo + CH), per = (hi - lo + TPC - 1) / TPC, q = (size_t)(t % TPC);
        for (size_t i = lo + q * per; i < std::min(hi, lo + (q + 1) * per); i++) {
            if (cells[i]) copy_to_staging(L->h_stage + i * 2048, cells[i], 2048);
            else memset(L->h_stage + i * 2048, 0, 2048);           // stands in for nil entries (reported after the device work)
        }
        left[k].arrive();
    };
    struct GatherJoin {                        // the tasks reference this frame: every way out waits for them
        mkzg_ctx* c; Countdown* all; bool started;
        ~GatherJoin() { if (started) c->pool->wait(*all); }
    } gather_join{c, &all_left, false};
    if (nch) { c->pool->run_async((int)nch * TPC, gather, &all_left); gather_join.started = true; }
    // deduplicateKZGCommitments (api_eip7594.go:238-265); per piece -- the sum over rows is linear, so pieces need not agree on the rows
    std::unordered_map<std::string, uint32_t> seen;
    std::vector<uint32_t> row_idx(n);
    std::vector<uint8_t> rows;
    for (size_t i = 0; i < n; i++) {
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
    const size_t n_rows = seen.size();
    const size_t cap = L->v_cap;
    uint32_t* h_misc = (uint32_t*)(L->h_v + 3 * cap * sizeof(int));
    // uploads, part 1: the points (small) -- their validation starts on the side stream before the 2 KiB cells are staged.  Everything
    // the device reads asynchronously sits in the lane's pinned staging area: [cells | rows | proofs | cell indices | row indices]
    uint8_t* h_rows = L->h_stage + (d_cells ? 0 : n * 2048);
    uint8_t* h_proofs = h_rows + n_rows * 48;
    uint8_t* h_cidx = h_proofs + n * 48;
    uint8_t* h_ridx = h_cidx + n * 8;
    memcpy(h_rows, rows.data(), n_rows * 48);
    memcpy(h_proofs, proofs48n, n * 48);
    memcpy(h_cidx, cell_indices, n * 8);
    memcpy(h_ridx, row_idx.data(), n * 4);
    // ONE upload for the four small arrays (every copy is a launch in the call's hardware queue, which it shares with other calls: ~0.09 ms each under load)
    uint8_t* d_rows = L->d_cv_in;
    uint8_t* d_proofs = d_rows + n_rows * 48;
    const uint64_t* d_cidx = (const uint64_t*)(d_proofs + n * 48);
    const uint32_t* d_ridx = (const uint32_t*)(d_proofs + n * 48 + n * 8);
    HIPCHK(hipMemcpyAsync(d_rows, h_rows, n_rows * 48 + n * (48 + 8 + 4), hipMemcpyHostToDevice, s));
    // points
    uint32_t* aff_rows = L->d_v_aff;
    uint32_t* aff_q = L->d_v_aff + cap * 32;
    int* st_rows = L->d_v_st;
    int* st_q = L->d_v_st + cap;
    // point decoding + subgroup checks on the side streams, beside the interpolation scalars (independent until the MSMs).
    // Decoding (0.43 ms) and the y-free subgroup test (0.97 ms) as two launches on two streams: the MSMs below need the decoded rows
    // only, so they start as soon as the scalars are ready instead of after the longer chain.  A point outside the subgroup is still a
    // point of the curve (the group law is total there), and its verdict makes the call fail before any sum is looked at.
    hipStream_t s2 = lane_side_stream(L, s), s3 = L->side2;
    // the call's three verdict arrays (cell statuses, subgroup verdicts of the rows, of the proofs: cap ints each) sit in the aux region of
    // the lane's verdict block (ctx.h: 8 cap words, where the blob verifier keeps its r^i y_i), behind the decoding statuses and the window
    // sums: everything the call brings back is ONE copy of a prefix of that block
    int* d_cv_st = (int*)L->d_v_ry;
    int* sub_rows = d_cv_st + cap;
    int* sub_q = d_cv_st + 2 * cap;
    lane_fork(L, s, s2, s3);
    // Calls of a few cells -- a column sidecar is ONE cell of every blob of the block, a custody set a few columns (EIP-7594) -- wait for
    // nothing but the points' chain: 0.7 ms on the device however few the points, ~45 us per point on a host core.  Up to
    // HOST_VALIDATE_MAX_POINTS points are decoded and subgroup-tested on the host pool and this thread (host_validate.hpp, option
    // host_validate) and go to the device as records; the verdict arrays are the kernels'.
    if (!late && c->opt_host_validate.load(std::memory_order_relaxed) != 0 && host_validate_pays(n + n_rows, c->pool ? c->pool->size() : 0, HOST_VALIDATE_MAX_CELL_POINTS)) {
        host_validate_arrays(c->pool, rows.data(), n_rows, proofs48n, n, aff_rows, st_rows, aff_q, st_q, s2, sub_rows, sub_q);
        HIPCHK(hipEventRecord(L->ev_decoded, s2));
        HIPCHK(hipEventRecord(L->ev_join, s2));
    } else {
    KT(KT_DECOMPRESS, s2, g1_validate2_device(d_rows, (int)n_rows, aff_rows, st_rows, d_proofs, (int)n, aff_q, st_q, s2, 1));
    HIPCHK(hipEventRecord(L->ev_decoded, s2));
    // the r-torsion test: a quad of lanes per point for calls of few cells (its 0.75 ms chain is what such a call waits for); one lane per
    // point when the uploads, the interpolation and the MSMs take longer than the 1.3 ms chain anyway (half the instructions, and the
    // other kernels' workgroups find free SIMDs: see k_g1_validate)
    const int sub_mode = MKZG_CV_SUBGROUP_WIDE_FROM > 0 && n + n_rows >= (size_t)MKZG_CV_SUBGROUP_WIDE_FROM ? 4 : 2;
    KT(KT_SUBGROUP, s3, g1_validate2_device(d_rows, (int)n_rows, nullptr, sub_rows, d_proofs, (int)n, nullptr, sub_q, s3, sub_mode));
    HIPCHK(hipEventRecord(L->ev_join, s3));
    }
    // scalars, first part: r^k, r^k h_k^64 and the rows' weights need the indices only -- in the call's queue AHEAD of the cells' upload, where
    // they run while the host still gathers (A/B with twelve callers: 9.4 against 8.6-9.3 M cells/s)
    // (r still on its way from a collective, RlcLate: the points' chains are running by now; the cells' gather below does not overlap the
    // wait -- a rank of a multi-process job that brings host pointers pays the slowest rank's lateness there)
    uint32_t r_words[8];
    if (late) {
        const int lrc = late->get();
        if (lrc) return lrc;
        if (late->offset + exp_offset + n > 0xffffffffull) return MKZG_ERR_BAD_ARGS;
        memcpy(r_words, late->r_words, 32);
        exp_offset += (uint32_t)late->offset;
    } else {
        memcpy(r_words, r_words_in, 32);
    }
    KT(KT_CELL_SCALARS, s, cell_verify_scalars_device(r_words, exp_offset, d_cidx, d_ridx, nullptr, (int)n, (int)n_rows, c->d_tw_fwd, c->d_tw_inv, L->d_cv_rp, L->d_v_s1, L->d_v_s3,
                                                       L->d_cv_weights, L->d_cv_partials, L->d_cv_interp, d_cv_st, s, 1));
    // uploads, part 2: the cells, chunk by chunk as the pool (started at the top of the call) signs them off
    for (size_t k = 0; k < nch; k++) {
        const size_t lo = k * CH, hi = std::min(n, lo + CH);
        c->pool->wait(left[k]);
        HIPCHK(hipMemcpyAsync(L->d_cv_cells + lo * 2048, L->h_stage + lo * 2048, (hi - lo) * 2048, hipMemcpyHostToDevice, s));
    }
    // scalars, second part: the interpolation reads the cells
    KT(KT_CELL_SCALARS, s, cell_verify_scalars_device(r_words, exp_offset, d_cidx, d_ridx, d_cells ? d_cells : L->d_cv_cells, (int)n, (int)n_rows, c->d_tw_fwd, c->d_tw_inv, L->d_cv_rp,
                                                       L->d_v_s1, L->d_v_s3, L->d_cv_weights, L->d_cv_partials, L->d_cv_interp, d_cv_st, s, 2));
    HIPCHK(hipStreamWaitEvent(s, L->ev_decoded, 0));
    // MSMs (kzg_verify.go:32,41,68,80): sum r^k pi_k ; sum w_row C_row ; sum interp_i [tau^i]G1 ; sum r^k h^64 pi_k
    const uint32_t* scal[4] = {L->d_v_s1, L->d_cv_weights, L->d_cv_interp, L->d_v_s3};
    const uint32_t* pts[4] = {aff_q, aff_rows, c->d_monomial_aff, aff_q};
    const int njob[4] = {(int)n, (int)n_rows, 64, (int)n};
    int vrc = 0;
    // (a quad of lanes per bucket when the launch leaves the chip mostly empty: one blob's 128 cells, not a block's 8192)
    KT(KT_VMSM, s, vrc = vmsm_windows_device(scal, pts, njob, 4, L->d_v_parts, L->d_v_misc + 16, s, 2048, 1));
    if (vrc) return MKZG_ERR_BAD_ARGS;
    HIPCHK(hipGetLastError());
    int* h_st = (int*)L->h_v;
    // every verdict array lands in the lane's PINNED mirror of the verdict block: a device-to-host copy into pageable memory is a
    // blocking copy that polls inside the runtime until the stream gets there -- with twelve callers that was twelve CPUs spinning beside
    // the pool (tools/bench_cellverify_inflight.py), whatever `host_wait` said.  One copy since round 5 (three until then).
    int* st_cells = (int*)(h_misc + V_MISC_WORDS_C + 16);
    int* h_sub_rows = st_cells + cap;
    int* h_sub_q = st_cells + 2 * cap;
    HIPCHK(hipStreamWaitEvent(s, L->ev_join, 0));               // the subgroup verdicts
    HIPCHK(hipMemcpyAsync(h_st, L->d_v_block, verdict_block_bytes(cap, (3 * cap * sizeof(int) + 31) / 32), hipMemcpyDeviceToHost, s));
    LANE_SYNC(c, L, s);
    L->drained = true;
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
    c->pool->parallel(4, [&](int m) { g1h_horner_words(out->sums[m], wsum + (size_t)m * VMSM_WINDOWS * 48, VMSM_WINDOWS, VMSM_WINDOW_BITS); });
    return MKZG_OK;
}
// The reference validates in three passes over the WHOLE batch (all commitments, then all proofs, then all cells, api_eip7594.go:190-213);
// pieces are folded in order, so among several invalid inputs the error reported is the one of the earliest piece -- the vectors and
// the reference's tests pin error-vs-failure only.
int sd_cell_verify_fold(mkzg_ctx* c, const uint8_t* commitments48n, size_t n_c, const uint64_t* cell_indices, size_t n_i, const uint8_t* const* cells, size_t n_cells,
                        const uint8_t* proofs48n, size_t n_p, uint64_t exp_offset, const uint8_t* r_be32, CellFold* out, const uint8_t* d_cells, hipStream_t user_stream, RlcLate* late) {
    for (int m = 0; m < 4; m++) G1H::set_inf(out->sums[m]);
    if (!c) return MKZG_ERR_BAD_ARGS;
    const size_t n = n_c;
    if (n != n_i || n != n_cells || n != n_p) return MKZG_ERR_BATCH_LENGTH;          // api_eip7594.go:166-171
    if (n == 0) return MKZG_OK;                                                        // :173-175
    if (!commitments48n || !cell_indices || (!cells && !d_cells) || !proofs48n) return MKZG_ERR_BAD_ARGS;
    for (size_t i = 0; i < n; i++) if (cell_indices[i] >= MKZG_CELLS_PER_EXT_BLOB) return MKZG_ERR_INVALID_CELL_ID;   // :184-188
    if (exp_offset + n > 0xffffffffull) return MKZG_ERR_BAD_ARGS;
    if (c->g2_alpha64.lines.empty()) return MKZG_ERR_BAD_TRUSTED_SETUP;
    int rc = ensure_twiddles(c);
    if (!rc) rc = ensure_monomial(c);
    if (rc) return rc;
    uint32_t r_words[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (late) {
        // (asked for inside the first piece, behind the points' launches)
    } else if (r_be32) {
        for (int i = 0; i < 8; i++) r_words[i] = ((uint32_t)r_be32[28 - 4 * i] << 24) | ((uint32_t)r_be32[29 - 4 * i] << 16) | ((uint32_t)r_be32[30 - 4 * i] << 8) | r_be32[31 - 4 * i];
    } else if (getrandom(r_words, 32, 0) != 32) {
        return MKZG_ERR_HIP;
    }
    std::unique_ptr<LaneGuard> guard(d_cells ? new LaneGuard(c, user_stream) : new LaneGuard(c));
    LaneGuard& G = *guard;
    CellFold f;
    for (size_t off = 0; off < n; off += CELL_PIECE) {
        const size_t m = std::min(CELL_PIECE, n - off);
        rc = cell_fold_piece(c, G.L, commitments48n + off * 48, cell_indices + off, d_cells ? nullptr : cells + off, d_cells ? d_cells + off * 2048 : nullptr, proofs48n + off * 48, m,
                             (uint32_t)(exp_offset + off), r_words, G.s, &f, late);
        if (rc) return G.fail(rc);
        for (int q = 0; q < 4; q++) G1H::add(out->sums[q], out->sums[q], f.sums[q]);
    }
    return MKZG_OK;
}
// kzg_verify.go:85-104: e(sum r^k pi_k, [tau^64]G2) == e(sum w C - [I(tau)] + sum r^k h^64 pi_k, G2)
int sd_cell_verify_finish(mkzg_ctx* c, const CellFold& f) {
    G1HX rl, t;
    G1H::neg(t, f.sums[2]);
    G1H::add(rl, f.sums[1], t);
    G1H::add(rl, rl, f.sums[3]);
    G1AffH ps[2];
    g1h_to_affine(ps[0], f.sums[0]);
    g1h_to_affine(ps[1], rl);
    const G2Prepared* qs[2] = {&c->g2_alpha64, &c->g2_neg_gen};
    return pairing_check_prepared_2mt(ps, qs, c->pool, miller_segments(c)) ? MKZG_OK : MKZG_VERIFY_FAILED;
}

// shared front end of RecoverCells / RecoverCellsAndComputeKZGProofs (reference api_eip7594.go:93-142) for n_blobs independent
// recoveries: per-blob validation in the reference's order, then ONE set of launches over the blobs that passed (blockIdx.y = blob),
// so that a batch runs the NTT stages and FK20 at batch efficiency (a single recovery is a chain of latency-bound launches).
// status[b] receives the reference's error for blob b; device outputs: L->d_cells (n x 128 x 2048 B), L->d_fk_proofs (n x 128 x 48 B).
static int recover_validate(const uint64_t* cell_ids, size_t n_ids, const uint8_t* const* cells, size_t n_cells, bool cells_on_device = false) {
    if (n_ids != n_cells) return MKZG_ERR_NUM_CELL_IDS;                                  // :94-96
    if ((n_ids && !cell_ids) || (n_cells && !cells && !cells_on_device)) return MKZG_ERR_BAD_ARGS;
    for (size_t i = 1; i < n_ids; i++) if (cell_ids[i] <= cell_ids[i - 1]) return MKZG_ERR_CELL_IDS_NOT_ORDERED;   // :99-101
    for (size_t i = 0; i < n_ids; i++) if (cell_ids[i] >= MKZG_CELLS_PER_EXT_BLOB) return MKZG_ERR_FOUND_INVALID_CELL_ID;   // :104-108
    if (n_ids < 64) return MKZG_ERR_NOT_ENOUGH_CELLS;                                    // :111-113
    if (!cells_on_device) for (size_t i = 0; i < n_cells; i++) if (!cells[i]) return MKZG_ERR_NIL_INPUT;       // serialization.go:197-199
    return MKZG_OK;
}
// d_cells_in: the supplied cells already in HBM -- blob b's n_ids[b] cells in id order, the blobs back to back (rejected blobs' cells
// included: the layout is the caller's) -- instead of host pointers; d_cells_out / d_proofs_out: where the kernels leave the results
// (null: the lane's own buffers).  s: the call's stream.
static int recover_front(mkzg_ctx* c, Lane* L, size_t n_blobs, const uint64_t* const* cell_ids, const size_t* n_ids, const uint8_t* const* const* cells, const size_t* n_cells,
                         bool with_proofs, int* status, hipStream_t s, const uint8_t* d_cells_in = nullptr, uint8_t* d_cells_out = nullptr, uint8_t* d_proofs_out = nullptr) {
    size_t total_cells = 0;
    for (size_t b = 0; b < n_blobs; b++) {
        status[b] = recover_validate(cell_ids[b], n_ids[b], d_cells_in ? nullptr : cells[b], n_cells[b], d_cells_in != nullptr);
        if (!status[b] && !d_cells_in) total_cells += n_cells[b];
    }
    int rc = ensure_rec_tables(c);
    if (!rc && with_proofs) rc = ensure_fk20_tables(c);
    if (!rc) rc = with_proofs ? ensure_fk20_ws(L, n_blobs) : ensure_cells_ws(L, n_blobs);
    if (!rc) rc = ensure_rec_ws(L, n_blobs);
    const size_t pos_bytes = n_blobs * 193 * sizeof(int);
    if (!rc) rc = ensure_staging(L, 1, std::max<size_t>(total_cells, 1) * 2048 + pos_bytes);
    if (rc) return rc;
    // positions: per blob 128 cell positions (global index into the staged cells, -1 = missing) + up to 64 missing ids; counts after them
    int* pos = (int*)(L->h_stage + std::max<size_t>(total_cells, 1) * 2048);
    for (size_t i = 0; i < n_blobs * 193; i++) pos[i] = -1;
    int* nmiss = pos + n_blobs * 192;
    size_t at = 0;
    for (size_t b = 0; b < n_blobs; b++) {
        int* pb = pos + b * 192;
        nmiss[b] = 0;
        if (status[b]) { if (d_cells_in) at += n_cells[b]; continue; }    // rejected blob: everything absent and no missing ids -- the stages then run on zeros, output ignored
        for (size_t i = 0; i < n_ids[b]; i++) pb[cell_ids[b][i]] = (int)(at + i);
        for (uint32_t cid = 0; cid < 128; cid++) {
            if (pb[cid] < 0) {
                uint32_t r = 0;
                for (int q = 0; q < 7; q++) if (cid & (1u << q)) r |= 1u << (6 - q);       // BitReverseInt(cellID, 128)  (:117-122)
                pb[128 + nmiss[b]++] = (int)r;
            }
        }
        at += n_cells[b];
    }
    // gather the supplied cells into the pinned staging area (the host pool shares the blobs)
    if (!d_cells_in) {
        std::vector<size_t> base(n_blobs + 1, 0);
        for (size_t b = 0; b < n_blobs; b++) base[b + 1] = base[b] + (status[b] ? 0 : n_cells[b]);
        const int nthr = total_cells >= 1024 ? 8 : 1;
        const size_t per = (n_blobs + nthr - 1) / nthr;
        c->pool->parallel(nthr, [&](int q) {
            for (size_t b = (size_t)q * per; b < std::min(n_blobs, (size_t)(q + 1) * per); b++) {
                if (status[b]) continue;
                for (size_t i = 0; i < n_cells[b]; i++) copy_to_staging(L->h_stage + (base[b] + i) * 2048, cells[b][i], 2048);
            }
        });
    }
    if (total_cells) HIPCHK(hipMemcpyAsync(L->d_blobs, L->h_stage, total_cells * 2048, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(L->d_rec_pos, pos, pos_bytes, hipMemcpyHostToDevice, s));
    KT(KT_RECOVER, s, recover_coeffs_device(d_cells_in ? d_cells_in : L->d_blobs, L->d_rec_pos, L->d_rec_pos + n_blobs * 192, (int)n_blobs, c->d_tw_fwd, c->d_tw_inv, c->d_pow7, c->d_ipow7,
                                            L->d_rec_scratch, L->d_coeffs, L->d_cell_st, s));
    KT(KT_CELLS_NTT, s, cells_from_coeffs_device(L->d_coeffs, (int)n_blobs, c->d_tw_fwd, c->d_tw_inv, d_cells_out ? d_cells_out : L->d_cells, L->d_cell_st, s));
    if (with_proofs) {
        KT(KT_FK20_SCALARS, s, fk20_scalars_device(L->d_coeffs, (int)n_blobs, c->d_tw_fwd, L->d_cell_st, L->d_fk_scalars, s));
        KT(KT_FK20_MSM, s, fk20_msm_device(L->d_fk_scalars, c->d_fk_table, c->fk20_bits_built, (int)n_blobs, L->d_cell_st, L->d_fk_parts, L->d_fk_results, s));
        KT(KT_FK20_FFTS, s, fk20_ffts_device(L->d_fk_results, L->d_fk_results + L->fk_cap * 128 * 64, (int)n_blobs, L->d_cell_st, d_proofs_out ? d_proofs_out : L->d_fk_proofs, L->d_fk_parts, c->opt_fk20_segments.load(std::memory_order_relaxed) && fk20_seg_bytes((int)n_blobs) <= L->fk_seg_bytes ? L->d_fk_seg : nullptr, s));
    }
    HIPCHK(hipGetLastError());
    // device-side statuses (non-canonical field elements in the cells) join the host-side ones
    // (into pinned memory -- the positions' staging words, whose upload is ahead of this copy in stream order: a copy into pageable memory
    // blocks and polls inside the runtime until the stream gets there)
    int* dst = pos;
    HIPCHK(hipMemcpyAsync(dst, L->d_cell_st, n_blobs * sizeof(int), hipMemcpyDeviceToHost, s));
    LANE_SYNC(c, L, s);
    for (size_t b = 0; b < n_blobs; b++) if (!status[b]) status[b] = dst[b];
    return MKZG_OK;
}

extern "C" {

int mkzg_verify_cell_kzg_proof_batch(mkzg_ctx* c, const uint8_t* commitments48n, size_t n_c, const uint64_t* cell_indices, size_t n_i, const uint8_t* const* cells,
                                     size_t n_cells, const uint8_t* proofs48n, size_t n_p) {
    if (!c) return MKZG_ERR_BAD_ARGS;
    if (n_c != n_i || n_c != n_cells || n_c != n_p) return MKZG_ERR_BATCH_LENGTH;          // api_eip7594.go:166-171
    if (n_c == 0) return MKZG_OK;                                                          // :173-175
    if (!c->peers.empty()) {
        if (!commitments48n || !cell_indices || !cells || !proofs48n) return MKZG_ERR_BAD_ARGS;
        for (size_t i = 0; i < n_c; i++) if (cell_indices[i] >= MKZG_CELLS_PER_EXT_BLOB) return MKZG_ERR_INVALID_CELL_ID;
        return md_verify_cell_kzg_proof_batch(c, commitments48n, cell_indices, cells, proofs48n, n_c);
    }
    CellFold f;
    int rc = sd_cell_verify_fold(c, commitments48n, n_c, cell_indices, n_i, cells, n_cells, proofs48n, n_p, 0, nullptr, &f);
    if (rc) return rc;
    return sd_cell_verify_finish(c, f);
}
// the same with the cells -- the bulk of the call's bytes -- already in HBM (n x 2048 B, contiguous, in the order of the other three
// arrays, which stay host arrays as in the reference: 104 B per cell); synchronises `stream` (the verdict needs the host pairing)
int mkzg_verify_cell_kzg_proof_batch_device(mkzg_ctx* c, const uint8_t* commitments48n, const uint64_t* cell_indices, const void* d_cells, const uint8_t* proofs48n, size_t n, void* stream) {
    if (!c) return MKZG_ERR_BAD_ARGS;
    if (n == 0) return MKZG_OK;                                                            // api_eip7594.go:173-175
    if (!commitments48n || !cell_indices || !d_cells || !proofs48n) return MKZG_ERR_BAD_ARGS;
    if (!c->peers.empty() && !(c = ctx_for_pointer(c, d_cells))) return MKZG_ERR_BAD_ARGS;
    CellFold f;
    int rc = sd_cell_verify_fold(c, commitments48n, n, cell_indices, n, nullptr, n, proofs48n, n, 0, nullptr, &f, (const uint8_t*)d_cells, (hipStream_t)stream);
    if (rc) return rc;
    return sd_cell_verify_finish(c, f);
}
// Sharded form (see CellFold): record = status@0 (int32), compressed sums[0..3] at 16, 64, 112, 160.
}   // extern "C"
void cell_fold_record(uint8_t record512[512], int rc, const CellFold& f) {
    memset(record512, 0, 512);
    const int32_t st = rc;
    memcpy(record512, &st, 4);
    if (rc != MKZG_OK) return;
    for (int m = 0; m < 4; m++) {
        G1AffH a;
        g1h_to_affine(a, f.sums[m]);
        g1h_compress(record512 + 16 + 48 * m, a);
    }
}
extern "C" {
int mkzg_verify_cell_kzg_proof_batch_partial(mkzg_ctx* c, const uint8_t* commitments48n, size_t n_c, const uint64_t* cell_indices, size_t n_i, const uint8_t* const* cells,
                                             size_t n_cells, const uint8_t* proofs48n, size_t n_p, uint64_t offset, const uint8_t r32[32], uint8_t record512[512]) {
    if (!c || !r32 || !record512) return MKZG_ERR_BAD_ARGS;
    c = md_pick(c);
    memset(record512, 0, 512);
    CellFold f;
    int rc = sd_cell_verify_fold(c, commitments48n, n_c, cell_indices, n_i, cells, n_cells, proofs48n, n_p, offset, r32, &f);
    cell_fold_record(record512, rc, f);
    return rc;
}
// sharded form over device-resident cells (this rank's shard)
int mkzg_verify_cell_kzg_proof_batch_partial_device(mkzg_ctx* c, const uint8_t* commitments48n, const uint64_t* cell_indices, const void* d_cells, const uint8_t* proofs48n, size_t n,
                                                    uint64_t offset, const uint8_t r32[32], uint8_t record512[512], void* stream) {
    if (!c || !r32 || !record512) return MKZG_ERR_BAD_ARGS;
    memset(record512, 0, 512);
    CellFold f;
    int rc;
    if (n && !d_cells) rc = MKZG_ERR_BAD_ARGS;
    else if (!c->peers.empty() && n && !(c = ctx_for_pointer(c, d_cells))) rc = MKZG_ERR_BAD_ARGS;
    else rc = sd_cell_verify_fold(c, commitments48n, n, cell_indices, n, nullptr, n, proofs48n, n, offset, r32, &f, (const uint8_t*)d_cells, (hipStream_t)stream);
    cell_fold_record(record512, rc, f);                       // whatever the outcome, the record says so: it is what travels
    return rc;
}
int mkzg_verify_cell_kzg_proof_batch_finish(mkzg_ctx* c, const uint8_t* records512, size_t n_records) {
    if (!c || (n_records && !records512)) return MKZG_ERR_BAD_ARGS;
    if (!c->peers.empty()) c = c->peers[0];
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
    return sd_cell_verify_finish(c, f);
}

// Context.RecoverCellsAndComputeKZGProofs (reference api_eip7594.go:144-161)
int mkzg_recover_cells_and_kzg_proofs(mkzg_ctx* c, const uint64_t* cell_ids, size_t n_ids, const uint8_t* const* cells, size_t n_cells, int num_threads, uint8_t* cells_out,
                                      uint8_t* proofs_out) {
    (void)num_threads;
    if (!c || !cells_out || !proofs_out || (n_ids && !cell_ids) || (n_cells && !cells)) return MKZG_ERR_BAD_ARGS;
    int st = recover_validate(cell_ids, n_ids, cells, n_cells);
    if (st) return st;
    c = md_pick(c);
    LaneGuard G(c);
    Lane* L = G.L;
    int rc = recover_front(c, L, 1, &cell_ids, &n_ids, &cells, &n_cells, true, &st, G.s);
    if (rc) return G.fail(rc);
    if (st) return st;
    HIPCHK(hipMemcpyAsync(cells_out, L->d_cells, 262144, hipMemcpyDeviceToHost, G.s));
    HIPCHK(hipMemcpyAsync(proofs_out, L->d_fk_proofs, 128 * 48, hipMemcpyDeviceToHost, G.s));
    LANE_SYNC(c, G.L, G.s);
    G.L->drained = true;
    return MKZG_OK;
}
// n_blobs independent recoveries in one call (no reference analogue; throughput form of the above, used by sharded recovery jobs).
// cell_ids[b] / cells[b]: the n_ids[b] = n_cells[b] supplied cells of blob b.  cells_out: n_blobs x 128 x 2048 B, proofs_out: n_blobs x 128 x 48 B,
// status[b]: the reference's error for blob b (outputs of a failed blob are left untouched).  Returns the first non-zero status.
int mkzg_recover_cells_and_kzg_proofs_batch(mkzg_ctx* c, size_t n_blobs, const uint64_t* const* cell_ids, const size_t* n_ids, const uint8_t* const* const* cells,
                                            const size_t* n_cells, uint8_t* cells_out, uint8_t* proofs_out, int* status) {
    if (!c || (n_blobs && (!cell_ids || !n_ids || !cells || !n_cells || !cells_out || !proofs_out || !status))) return MKZG_ERR_BAD_ARGS;
    if (n_blobs == 0) return MKZG_OK;
    if (!c->peers.empty()) return md_recover_cells_and_kzg_proofs_batch(c, n_blobs, cell_ids, n_ids, cells, n_cells, cells_out, proofs_out, status);
    LaneGuard G(c);
    Lane* L = G.L;
    int first = MKZG_OK;
    const size_t chunk_max = (size_t)c->opt_cells_chunk, pieces = (n_blobs + chunk_max - 1) / chunk_max, even = (n_blobs + pieces - 1) / pieces;
    for (size_t done = 0; done < n_blobs; done += even) {               // even pieces of at most cells_chunk blobs bound the workspaces
        const size_t m = std::min<size_t>(even, n_blobs - done);
        int rc = recover_front(c, L, m, cell_ids + done, n_ids + done, cells + done, n_cells + done, true, status + done, G.s);
        if (rc) return G.fail(rc);
        // good blobs leave through the lane's pinned staging area (a device-to-pageable copy is staged by the runtime at a fraction of
        // the PCIe rate): one device copy per run of good blobs, then the host pool spreads them to the caller's buffers
        rc = ensure_staging(L, 1, m * (262144 + 128 * 48));
        if (rc) return G.fail(rc);
        uint8_t* h_cells = L->h_stage;
        uint8_t* h_proofs = L->h_stage + m * 262144;
        for (size_t b = 0; b < m;) {
            if (status[done + b]) { if (!first) first = status[done + b]; b++; continue; }
            size_t e = b;
            while (e < m && !status[done + e]) e++;
            HIPCHK(hipMemcpyAsync(h_cells + b * 262144, L->d_cells + b * 262144, (e - b) * 262144, hipMemcpyDeviceToHost, G.s));
            HIPCHK(hipMemcpyAsync(h_proofs + b * 128 * 48, L->d_fk_proofs + b * 128 * 48, (e - b) * 128 * 48, hipMemcpyDeviceToHost, G.s));
            b = e;
        }
        LANE_SYNC(c, G.L, G.s);
        G.L->drained = true;
        const int nthr = m >= 8 ? 8 : 1;
        const size_t per = (m + nthr - 1) / nthr;
        c->pool->parallel(nthr, [&](int q) {
            for (size_t b = (size_t)q * per; b < std::min(m, (size_t)(q + 1) * per); b++) {
                if (status[done + b]) continue;
                memcpy(cells_out + (done + b) * 262144, h_cells + b * 262144, 262144);
                memcpy(proofs_out + (done + b) * 128 * 48, h_proofs + b * 128 * 48, 128 * 48);
            }
        });
    }
    return first;
}
// RecoverCellsAndComputeKZGProofs / RecoverCells for a batch whose supplied cells are already in HBM and whose results stay there
// (include/mi355kzg.h).  d_cells: blob b's n_ids[b] cells in the order of its ids, the blobs back to back; d_cells_out: n_blobs x 128 x
// 2048 B; d_proofs_out: n_blobs x 128 x 48 B, or null for the cells alone.  Synchronises `stream` (the per-blob statuses come back).
int mkzg_recover_cells_and_kzg_proofs_batch_device(mkzg_ctx* c, size_t n_blobs, const uint64_t* const* cell_ids, const size_t* n_ids, const void* d_cells, void* d_cells_out,
                                                   void* d_proofs_out, int* status, void* stream) {
    if (!c || (n_blobs && (!cell_ids || !n_ids || !d_cells || !d_cells_out || !status))) return MKZG_ERR_BAD_ARGS;
    if (n_blobs == 0) return MKZG_OK;
    if (!c->peers.empty() && !(c = ctx_for_pointer(c, d_cells))) return MKZG_ERR_BAD_ARGS;
    LaneGuard G(c, (hipStream_t)stream);
    Lane* L = G.L;
    int first = MKZG_OK;
    const bool with_proofs = d_proofs_out != nullptr;
    const size_t chunk_max = (size_t)c->opt_cells_chunk, pieces = (n_blobs + chunk_max - 1) / chunk_max, even = (n_blobs + pieces - 1) / pieces;
    size_t cell_off = 0;
    for (size_t done = 0; done < n_blobs; done += even) {
        const size_t m = std::min<size_t>(even, n_blobs - done);
        int rc = recover_front(c, L, m, cell_ids + done, n_ids + done, nullptr, n_ids + done, with_proofs, status + done, G.s, (const uint8_t*)d_cells + cell_off * 2048,
                               (uint8_t*)d_cells_out + done * 262144, with_proofs ? (uint8_t*)d_proofs_out + done * 128 * 48 : nullptr);
        if (rc) return G.fail(rc);
        for (size_t b = 0; b < m; b++) {
            cell_off += n_ids[done + b];
            if (status[done + b] && !first) first = status[done + b];
        }
    }
    G.L->drained = true;                       // recover_front waited for the statuses on the call's stream
    return first;
}
// Context.RecoverCells (reference api_eip.go:8-15)
int mkzg_recover_cells(mkzg_ctx* c, const uint64_t* cell_ids, size_t n_ids, const uint8_t* const* cells, size_t n_cells, int num_threads, uint8_t* cells_out) {
    (void)num_threads;
    if (!c || !cells_out || (n_ids && !cell_ids) || (n_cells && !cells)) return MKZG_ERR_BAD_ARGS;
    int st = recover_validate(cell_ids, n_ids, cells, n_cells);
    if (st) return st;
    c = md_pick(c);
    LaneGuard G(c);
    Lane* L = G.L;
    int rc = recover_front(c, L, 1, &cell_ids, &n_ids, &cells, &n_cells, false, &st, G.s);
    if (rc) return G.fail(rc);
    if (st) return st;
    HIPCHK(hipMemcpyAsync(cells_out, L->d_cells, 262144, hipMemcpyDeviceToHost, G.s));
    LANE_SYNC(c, G.L, G.s);
    G.L->drained = true;
    return MKZG_OK;
}

}  // extern "C"
