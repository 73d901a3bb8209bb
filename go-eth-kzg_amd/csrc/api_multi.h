// api_multi.h -- internal seams between the single-device entry points and the multi-device front (api_multi.hip).
//
// A context created over several devices (mkzg_ctx_new with n_dev > 1) is a front object that owns one full context per device
// (mkzg_ctx::peers).  Batch entry points with host pointers are sharded over the devices by contiguous ranges -- one host thread per
// device -- and the two batch verifiers combine their per-device partial sums in process (the same fold / finish split that the
// multi-process form exchanges as 512-byte records, include/mi355kzg.h); single-item calls go round-robin; device-pointer entry
// points run on the device that owns the pointer.
#pragma once
#include "ctx.h"

// ---- EIP-4844 batch verifier, split at its four sums (reference internal/kzg/kzg_verify.go:141-199)
struct VerifyFold { G1HX A, Cc, D; uint32_t e[8]; };
void verify_fold_clear(VerifyFold& f);
void verify_fold_add(VerifyFold& acc, const VerifyFold& f);
// host-pointer fold of n items with global exponents r^(exp_offset + i); r_le32: the batch-wide scalar as 32 little-endian bytes.
// first_nil (may be null) receives the index of the first nil blob or n; statuses are resolved inside (reference verify.go:102-123 order).
// late (may be null): r and the offset still on their way (RlcLate, ctx.h) -- exponents are then r^(late->offset + exp_offset + i)
int sd_verify_blob_fold_host(mkzg_ctx* c, const uint8_t* const* blobs, const uint8_t* commitments48n, const uint8_t* proofs48n, size_t n, uint64_t exp_offset,
                             const uint8_t r_le32[32], VerifyFold* out, RlcLate* late = nullptr);
int sd_verify_blob_fold_device(mkzg_ctx* c, const void* d_blobs, const void* d_commitments48, const void* d_proofs48, size_t n, uint64_t exp_offset, const uint8_t* r_le32,
                               hipStream_t stream, VerifyFold* out, RlcLate* late = nullptr);
void verify_fold_record(uint8_t record512[512], int rc, const VerifyFold& total);
int sd_verify_finish(mkzg_ctx* c, const VerifyFold& f);
int draw_rlc_scalar(uint8_t r_le32[32]);

// ---- EIP-7594 cell batch verifier, split at its four group sums (reference internal/kzg_multi/kzg_verify.go:32,41,68,80)
struct CellFold { G1HX sums[4]; };
int sd_cell_verify_fold(mkzg_ctx* c, const uint8_t* commitments48n, size_t n_c, const uint64_t* cell_indices, size_t n_i, const uint8_t* const* cells, size_t n_cells,
                        const uint8_t* proofs48n, size_t n_p, uint64_t exp_offset, const uint8_t* r_be32 /* null: draw */, CellFold* out,
                        const uint8_t* d_cells = nullptr /* the cells contiguous in HBM instead of host pointers */, hipStream_t user_stream = nullptr,
                        RlcLate* late = nullptr /* r and the offset still on their way: r_be32 unused, exponents r^(late->offset + exp_offset + k) */);
void cell_fold_record(uint8_t record512[512], int rc, const CellFold& f);
int sd_cell_verify_finish(mkzg_ctx* c, const CellFold& f);

// ---- multi-device fronts (c->peers non-empty)
int md_blob_to_kzg_commitment_batch(mkzg_ctx* c, const uint8_t* const* blobs, size_t n, uint8_t* out48n, int* per_item_status);
int md_verify_blob_kzg_proof_batch(mkzg_ctx* c, const uint8_t* const* blobs, const uint8_t* commitments48n, const uint8_t* proofs48n, size_t n);
int md_verify_cell_kzg_proof_batch(mkzg_ctx* c, const uint8_t* commitments48n, const uint64_t* cell_indices, const uint8_t* const* cells, const uint8_t* proofs48n, size_t n);
int md_recover_cells_and_kzg_proofs_batch(mkzg_ctx* c, size_t n_blobs, const uint64_t* const* cell_ids, const size_t* n_ids, const uint8_t* const* const* cells,
                                          const size_t* n_cells, uint8_t* cells_out, uint8_t* proofs_out, int* status);
static inline mkzg_ctx* md_pick(mkzg_ctx* c) { return c->peers.empty() ? c : c->peers[c->rr++ % c->peers.size()]; }
