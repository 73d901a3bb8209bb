// ctx.h -- the context object behind the C ABI (internal to libmi355kzg.so).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
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

using namespace kzg;

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e__ = (expr);                                                                       \
        if (e__ != hipSuccess) {                                                                       \
            fprintf(stderr, "[mi355kzg] HIP error %s at %s:%d: %s\n", hipGetErrorName(e__), __FILE__, __LINE__, #expr); \
            return MKZG_ERR_HIP;                                                                       \
        }                                                                                              \
    } while (0)

struct mkzg_ctx {
    int device = 0;
    hipStream_t stream = nullptr;          // internal stream for host-pointer entry points
    hipStream_t side = nullptr;            // side stream: point validation runs beside SHA-256 / evaluation
    hipStream_t side2 = nullptr;           // second side stream: the cell verifier's subgroup test beside decoding + MSMs
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_decoded = nullptr;
    std::recursive_mutex mu;
    kzg::HostPool* pool = nullptr;         // persistent host threads for the verifiers' host steps (host_pool.hpp)
    // SRS
    std::vector<uint8_t> g1_monomial, g1_lagrange, g2_monomial;   // raw compressed bytes
    uint32_t* d_lagrange_aff = nullptr;    // 4096 affine rows, natural order of the JSON
    // commit MSM: one shared table, NSLOT independent workspaces so that consecutive calls on different
    // streams overlap (the latency-bound bucket reduction of one call hides under the accumulation of the
    // next); slot_done[i] orders the reuse of workspace i across streams without host synchronisation
    static constexpr int NSLOT = 2;
    CommitPlan plan;                       // == slots[0] (kept for the single-slot users)
    CommitPlan slots[NSLOT];
    hipEvent_t slot_done[NSLOT] = {nullptr, nullptr};
    bool slot_used[NSLOT] = {false, false};
    unsigned slot_counter = 0;
    hipEvent_t acc_done[NSLOT] = {nullptr, nullptr};   // recorded after each call's accumulate kernel
    int last_acc_slot = -1;
    int opt_c = 13;
    int opt_chain_acc = 0;                 // 1: the accumulate kernel of a call waits for the previous call's accumulate kernel
    int opt_direct_bits = 0;               // 0: bucket method; 12..15: direct-table commitment (HBM-hungry, see msm_kernels.hip)
    uint32_t* d_direct = nullptr;
    int direct_bits_built = 0;
    DirectSlot dslots[NSLOT];              // direct mode workspaces (share slot_done / acc_done with the bucket mode)
    int dslot_batch = 0;
    int opt_max_batch = 256;
    bool plan_ready = false;
    // staging
    uint8_t* h_stage = nullptr;            // pinned
    size_t h_stage_bytes = 0;
    uint8_t* d_blobs = nullptr;
    size_t d_blobs_bytes = 0;
    uint8_t* d_out = nullptr;
    int* d_status = nullptr;
    size_t d_out_items = 0;
    // profiling
    bool profiling = false;
    KernelTimer timer;
    // ---- EIP-4844 proofs / verification
    uint32_t* d_roots_brp = nullptr;       // 4096 Fr rows (12 words), bit-reversed domain (api.go:132)
    uint32_t* d_inv_roots_brp = nullptr;
    G1HFixedBase gen_fixed;                // host byte-window table of the G1 generator ([e]G of the batch verifiers)
    G2Prepared g2_gen, g2_alpha;           // prepared [1]G2, [alpha]G2 (kzg.OpeningKey, internal/kzg/srs.go:10-20)
    G2Prepared g2_neg_gen, g2_alpha64;     // prepared -[1]G2, [alpha^64]G2 (kzgmulti.OpeningKey, internal/kzg_multi/kzg_verify.go:89-92)
    // verify workspace, sized for v_cap proofs
    size_t v_cap = 0;
    uint8_t* d_v_raw = nullptr;            // 2 v_cap x 48 B   (commitments | proofs)
    uint32_t* d_v_aff = nullptr;           // 2 v_cap affine rows
    int* d_v_st = nullptr;                 // 3 v_cap ints (commitments | proofs | blobs)
    uint32_t* d_v_z = nullptr;             // v_cap Fr rows
    uint32_t* d_v_y = nullptr;
    uint32_t* d_v_s1 = nullptr;            // v_cap x 8 words
    uint32_t* d_v_s3 = nullptr;
    uint32_t* d_v_misc = nullptr;          // [0..8) r, [8..16) e, [16 .. 16+3*32*48) window sums
    uint8_t* h_v = nullptr;                // pinned: statuses + misc
    size_t h_v_bytes = 0;
    // quotient workspace (proof computation), q_cap blobs
    size_t q_cap = 0, quot_cap = 0;        // capacities of the scalar/status arrays and of the quotient rows
    uint32_t* d_quot = nullptr;            // q_cap x 4096 Fr rows
    uint32_t* d_q_z = nullptr;
    uint32_t* d_q_y = nullptr;
    int* d_q_st = nullptr;
    // ---- EIP-7594
    uint32_t* d_v_coeffs = nullptr;        // batch verifier: coefficient rows of the blobs (v_coeff_cap x 4096 x 12 words)
    size_t v_coeff_cap = 0;
    uint32_t* d_tw_fwd = nullptr;          // 4096 Fr rows: w_8192^j
    uint32_t* d_tw_inv = nullptr;          // 4096 Fr rows: w_8192^-j
    uint32_t* d_g1tw_fwd = nullptr;        // 64 x 8 words: w_128^j as plain scalars
    uint32_t* d_g1tw_inv = nullptr;
    bool cells_ready = false;
    size_t cell_cap = 0;                   // blobs per launch for the cell paths
    uint32_t* d_coeffs = nullptr;          // cell_cap x 4096 Fr rows
    uint8_t* d_cells = nullptr;            // cell_cap x 262144 B
    int* d_cell_st = nullptr;
    // VerifyCellKZGProofBatch workspace, cv_cap cells
    size_t cv_cap = 0;
    uint8_t* d_cv_cells = nullptr;         // cv_cap x 2048 B
    uint64_t* d_cv_cell_idx = nullptr;
    uint32_t* d_cv_row_idx = nullptr;
    uint32_t* d_cv_rp = nullptr;           // cv_cap Fr rows
    uint32_t* d_cv_weights = nullptr;      // cv_cap x 8 words
    uint32_t* d_cv_partials = nullptr;     // CELL_INTERP_BLOCKS x 64 Fr rows
    uint32_t* d_cv_interp = nullptr;       // 64 x 8 words
    int* d_cv_st = nullptr;                // 3 x cv_cap: cell statuses, subgroup verdicts of the commitment rows, of the proofs
    // recovery
    bool rec_ready = false;
    size_t rec_cap = 0;                    // blobs the recovery scratch / position arrays hold
    uint32_t* d_pow7 = nullptr;
    uint32_t* d_ipow7 = nullptr;
    uint32_t* d_rec_scratch = nullptr;
    int* d_rec_pos = nullptr;              // per blob 128 positions + 64 missing ids, then one count per blob
    // FK20
    bool fk_ready = false;
    hipEvent_t fk_ws_done = nullptr;       // orders reuse of the cells / FK20 / recovery workspaces across streams (asynchronous entry points)
    bool fk_ws_used = false;
    uint32_t* d_monomial_aff = nullptr;    // 4096 affine rows
    uint32_t* d_fk_t_aff = nullptr;        // 8192 affine rows T[k][i]
    uint32_t* d_fk_table = nullptr;        // direct window table (25.8 GB)
    size_t fk_cap = 0;
    uint32_t* d_fk_scalars = nullptr;      // fk_cap x 128 x 64 x 8 words
    uint32_t* d_fk_results = nullptr;      // 2 x fk_cap x 128 XYZZ rows (MSM results + ping-pong buffer of the G1 DFT levels)
    uint32_t* d_fk_parts = nullptr;        // fk_cap x 128 x 16 XYZZ rows: per-lane partial sums of the FK20 MSMs
    uint32_t* d_fk_plan = nullptr;         // FK_PLAN_WORDS lane descriptors of the G1 DFT levels
    uint8_t* d_fk_proofs = nullptr;        // fk_cap x 128 x 48 B
    // host finish timing (last call), microseconds
    double last_host_finish_us = 0, last_verify_total_us = 0;
};

int ensure_plan(mkzg_ctx* c);
int acquire_slot(mkzg_ctx* c, hipStream_t s);          // returns slot index (>= 0) or -1
int release_slot(mkzg_ctx* c, int slot, hipStream_t s);
extern "C" int commit_scalars_device(mkzg_ctx* c, const uint8_t* d_scalars, size_t stride_bytes, int input_kind, size_t n, uint8_t* d_out48n, int* d_status, hipStream_t s);
extern "C" int ensure_staging(mkzg_ctx* c, size_t n_items, size_t in_bytes_per_item);   // defined inside api.hip's extern "C" block
int ensure_verify_ws(mkzg_ctx* c, size_t n);
int ensure_quot_ws(mkzg_ctx* c, size_t n, size_t n_quot);
void free_verify_ws(mkzg_ctx* c);
int ensure_cells(mkzg_ctx* c, size_t n);
void free_cells(mkzg_ctx* c);
int ensure_fk20(mkzg_ctx* c, size_t n);
int ensure_monomial(mkzg_ctx* c);

