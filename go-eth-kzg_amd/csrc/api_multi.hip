// api_multi.hip -- one context over several devices (mkzg_ctx_new with n_dev > 1; see api_multi.h).
// The reference has one Context per process and spreads work over goroutines (verify.go:152-169, multiexp.go:20-26); here the
// batch entry points spread contiguous shards over the listed GPUs, one host thread per device, with no exchange at all for
// commitments / recovery (independent blobs) and an in-process sum of the per-device partial results for the two batch
// verifiers -- the same fold / finish split the multi-process form ships as 512-byte records over RCCL.
#include <sys/random.h>
#include "ctx.h"
#include "api_multi.h"

// contiguous shard of device d among nd for n items
static inline void shard(size_t n, size_t nd, size_t d, size_t* lo, size_t* hi) {
    const size_t base = n / nd, rem = n % nd;
    *lo = d * base + std::min(d, rem);
    *hi = *lo + base + (d < rem ? 1 : 0);
}

int md_blob_to_kzg_commitment_batch(mkzg_ctx* c, const uint8_t* const* blobs, size_t n, uint8_t* out48n, int* per_item_status) {
    const size_t nd = c->peers.size();
    std::vector<int> rcs(nd, 0), st(n, 0);
    c->pool->parallel((int)nd, [&](int d) {
        size_t lo, hi;
        shard(n, nd, (size_t)d, &lo, &hi);
        if (hi > lo) rcs[d] = mkzg_blob_to_kzg_commitment_batch(c->peers[d], blobs + lo, hi - lo, out48n + lo * 48, st.data() + lo);
    });
    int first = MKZG_OK;
    for (size_t d = 0; d < nd; d++) if (rcs[d] == MKZG_ERR_HIP || rcs[d] == MKZG_ERR_BAD_ARGS) return rcs[d];
    for (size_t i = 0; i < n; i++) {
        if (per_item_status) per_item_status[i] = st[i];
        if (st[i] && !first) first = st[i];
    }
    return first;
}

int md_verify_blob_kzg_proof_batch(mkzg_ctx* c, const uint8_t* const* blobs, const uint8_t* commitments48n, const uint8_t* proofs48n, size_t n) {
    const size_t nd = c->peers.size();
    uint8_t r[32];
    if (draw_rlc_scalar(r)) return MKZG_ERR_HIP;
    std::vector<int> rcs(nd, 0);
    std::vector<VerifyFold> folds(nd);
    c->pool->parallel((int)nd, [&](int d) {
        size_t lo, hi;
        shard(n, nd, (size_t)d, &lo, &hi);
        verify_fold_clear(folds[d]);
        if (hi > lo) rcs[d] = sd_verify_blob_fold_host(c->peers[d], blobs + lo, commitments48n + lo * 48, proofs48n + lo * 48, hi - lo, lo, r, &folds[d]);
    });
    for (size_t d = 0; d < nd; d++) if (rcs[d]) return rcs[d];          // the shard with the lowest indices decides the error (verify.go:102-123)
    VerifyFold total;
    verify_fold_clear(total);
    for (size_t d = 0; d < nd; d++) verify_fold_add(total, folds[d]);
    return sd_verify_finish(c->peers[0], total);
}

int md_verify_cell_kzg_proof_batch(mkzg_ctx* c, const uint8_t* commitments48n, const uint64_t* cell_indices, const uint8_t* const* cells, const uint8_t* proofs48n, size_t n) {
    const size_t nd = c->peers.size();
    uint8_t r_be[32];
    if (getrandom(r_be, 32, 0) != 32) return MKZG_ERR_HIP;
    r_be[0] &= 0x3f;                                                     // < 2^254 < r: every device reads the same canonical scalar
    std::vector<int> rcs(nd, 0);
    std::vector<CellFold> folds(nd);
    c->pool->parallel((int)nd, [&](int d) {
        size_t lo, hi;
        shard(n, nd, (size_t)d, &lo, &hi);
        for (int m = 0; m < 4; m++) G1H::set_inf(folds[d].sums[m]);
        if (hi > lo) rcs[d] = sd_cell_verify_fold(c->peers[d], commitments48n + lo * 48, hi - lo, cell_indices + lo, hi - lo, cells + lo, hi - lo, proofs48n + lo * 48, hi - lo, lo, r_be, &folds[d]);
    });
    for (size_t d = 0; d < nd; d++) if (rcs[d]) return rcs[d];
    CellFold total;
    for (int m = 0; m < 4; m++) G1H::set_inf(total.sums[m]);
    for (size_t d = 0; d < nd; d++) for (int m = 0; m < 4; m++) G1H::add(total.sums[m], total.sums[m], folds[d].sums[m]);
    return sd_cell_verify_finish(c->peers[0], total);
}

int md_recover_cells_and_kzg_proofs_batch(mkzg_ctx* c, size_t n_blobs, const uint64_t* const* cell_ids, const size_t* n_ids, const uint8_t* const* const* cells,
                                          const size_t* n_cells, uint8_t* cells_out, uint8_t* proofs_out, int* status) {
    const size_t nd = c->peers.size();
    std::vector<int> rcs(nd, 0);
    c->pool->parallel((int)nd, [&](int d) {
        size_t lo, hi;
        shard(n_blobs, nd, (size_t)d, &lo, &hi);
        if (hi > lo) rcs[d] = mkzg_recover_cells_and_kzg_proofs_batch(c->peers[d], hi - lo, cell_ids + lo, n_ids + lo, cells + lo, n_cells + lo, cells_out + lo * 262144,
                                                                      proofs_out + lo * 128 * 48, status + lo);
    });
    for (size_t d = 0; d < nd; d++) if (rcs[d] == MKZG_ERR_HIP || rcs[d] == MKZG_ERR_BAD_ARGS) return rcs[d];
    for (size_t b = 0; b < n_blobs; b++) if (status[b]) return status[b];
    return MKZG_OK;
}
