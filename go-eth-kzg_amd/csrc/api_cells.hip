// api_cells.hip -- C ABI entry points of the EIP-7594 cell paths (include/mi355kzg.h):
//   Context.ComputeCells                  reference api_eip7594.go:12-26
//   Context.ComputeCellsAndKZGProofs      reference api_eip7594.go:28-52
// All field work happens in cells_kernels.hip / fk20_kernels.hip; the host only stages bytes.
#include "ctx.h"

void free_cells(mkzg_ctx* c) {
    void* ptrs[] = {c->d_tw_fwd, c->d_tw_inv, c->d_g1tw_fwd, c->d_g1tw_inv, c->d_coeffs, c->d_cells, c->d_cell_st,
                    c->d_monomial_aff, c->d_fk_t_aff, c->d_fk_table, c->d_fk_scalars, c->d_fk_results, c->d_fk_proofs};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    c->d_tw_fwd = c->d_tw_inv = c->d_g1tw_fwd = c->d_g1tw_inv = c->d_coeffs = nullptr;
    c->d_cells = nullptr;
    c->d_cell_st = nullptr;
    c->d_monomial_aff = c->d_fk_t_aff = c->d_fk_table = c->d_fk_scalars = c->d_fk_results = nullptr;
    c->d_fk_proofs = nullptr;
    c->cells_ready = false;
    c->cell_cap = 0;
    c->fk_ready = false;
    c->fk_cap = 0;
}
// fk20.NewFK20 (reference fk20.go:23-52, called from api.go:137): done lazily on first use
int ensure_fk20(mkzg_ctx* c, size_t n) {
    int rc = ensure_cells(c, n);
    if (rc) return rc;
    if (!c->fk_ready) {
        uint8_t* d_raw = nullptr;
        int* d_st = nullptr;
        HIPCHK(hipMalloc(&d_raw, 4096 * 48));
        HIPCHK(hipMalloc(&d_st, 4096 * sizeof(int)));
        HIPCHK(hipMalloc(&c->d_monomial_aff, 4096 * 128));
        HIPCHK(hipMalloc(&c->d_fk_t_aff, 8192 * 128));
        HIPCHK(hipMalloc(&c->d_fk_table, fk20_table_bytes()));
        HIPCHK(hipMemcpyAsync(d_raw, c->g1_monomial.data(), 4096 * 48, hipMemcpyHostToDevice, c->stream));
        g1_decompress_device(d_raw, 4096, c->d_monomial_aff, d_st, c->stream);
        std::vector<int> st(4096);
        HIPCHK(hipMemcpyAsync(st.data(), d_st, 4096 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        (void)hipFree(d_raw);
        (void)hipFree(d_st);
        for (int v : st) if (v) return MKZG_ERR_BAD_TRUSTED_SETUP;
        fk20_setup_device(c->d_monomial_aff, c->d_g1tw_fwd, c->d_fk_t_aff, c->d_fk_table, c->stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(c->stream));
        c->fk_ready = true;
    }
    if (n > c->fk_cap) {
        if (c->d_fk_scalars) (void)hipFree(c->d_fk_scalars);
        if (c->d_fk_results) (void)hipFree(c->d_fk_results);
        if (c->d_fk_proofs) (void)hipFree(c->d_fk_proofs);
        c->fk_cap = 0;
        HIPCHK(hipMalloc(&c->d_fk_scalars, n * 128 * 64 * 32));
        HIPCHK(hipMalloc(&c->d_fk_results, n * 128 * 256));
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
    size_t done = 0;
    while (done < n) {
        size_t chunk = std::min(n - done, c->fk_cap);
        const uint8_t* blobs = (const uint8_t*)d_blobs + done * MKZG_BYTES_PER_BLOB;
        cells_from_blobs_device(blobs, (int)chunk, c->d_tw_fwd, c->d_tw_inv, c->d_coeffs, (uint8_t*)d_cells + done * 262144, d_status + done, s);
        fk20_scalars_device(c->d_coeffs, (int)chunk, c->d_tw_fwd, d_status + done, c->d_fk_scalars, s);
        fk20_proofs_device(c->d_fk_scalars, c->d_fk_table, c->d_g1tw_fwd, c->d_g1tw_inv, (int)chunk, d_status + done, c->d_fk_results,
                           (uint8_t*)d_proofs + done * 128 * 48, s);
        done += chunk;
    }
    HIPCHK(hipGetLastError());
    return MKZG_OK;
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

}  // extern "C"
