// msm_kernels.h -- host-visible launch interface of msm_kernels.hip (internal to libmi355kzg.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <vector>
#include "../../include/mi355kzg.h"

namespace kzg {

enum KernelKind { KT_DIGITS_SORT = 0, KT_BUCKET_ACC, KT_REDUCE1, KT_REDUCE2, KT_DECOMPRESS, KT_SUBGROUP, KT_SHA256, KT_EVAL, KT_QUOTIENT, KT_RLC, KT_VMSM, KT_CELLS_NTT, KT_FK20_SCALARS, KT_FK20_MSM, KT_FK20_FFTS, KT_CELL_SCALARS, KT_RECOVER, KT_COMMIT_DIRECT, KT_COMMIT_DIRECT_FIN, KT_POLY_EVAL, KT_EVAL_TREE, KT_COUNT };
static const char* const kKernelNames[KT_COUNT] = {"k_digits_sort", "k_bucket_accumulate", "k_bucket_reduce1", "k_bucket_reduce2", "k_g1_decompress",
                                                   "k_g1_validate", "k_challenge_sha256", "k_blob_eval", "k_blob_quotient", "k_rlc_scalars", "k_vmsm_window",
                                                   "k_cells_ntt", "k_fk20_scalars", "k_fk20_msm", "k_fk20_ffts", "k_cell_verify_scalars", "k_recover_*", "k_commit_direct_acc", "k_commit_direct_fin",
                                                   "k_poly_eval", "k_blob_eval_tree"};      // k_blob_eval: the per-lane-inversion form (proof path: k_blob_quotient = k_blob_eval<1>)

// hipEvent bracket around each kernel launch, on the stream the kernel is launched on.
struct KernelTimer {
    struct Rec { hipEvent_t e0, e1; int kind; };
    std::vector<Rec> pending;
    double total_ms[KT_COUNT] = {0};
    long launches[KT_COUNT] = {0};
    hipEvent_t cur0 = nullptr;
    void begin(int kind, hipStream_t s) {
        (void)kind;
        (void)hipEventCreate(&cur0);
        (void)hipEventRecord(cur0, s);
    }
    void end(int kind, hipStream_t s) {
        hipEvent_t e1;
        (void)hipEventCreate(&e1);
        (void)hipEventRecord(e1, s);
        pending.push_back({cur0, e1, kind});
        cur0 = nullptr;
    }
    void collect() {   // caller has synchronised the stream(s)
        for (auto& r : pending) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) { total_ms[r.kind] += ms; launches[r.kind]++; }
            (void)hipEventDestroy(r.e0);
            (void)hipEventDestroy(r.e1);
        }
        pending.clear();
    }
    void reset() {
        for (int i = 0; i < KT_COUNT; i++) { total_ms[i] = 0; launches[i] = 0; }
    }
};

struct CommitPlanSizes { int nwin, nbucket, max_tasks; size_t table_bytes, entries_bytes, task_off_bytes, tasks_bytes, ntasks_bytes, parts_bytes, partials_bytes; };
struct CommitPlan {
    int c = 13;                 // window bits
    int max_batch = 0;
    uint32_t* d_table = nullptr;    // [nwin][4096] affine rows
    uint32_t* d_entries = nullptr;   // [batch][4096*nwin] sorted (point, window, sign) entries
    uint32_t* d_task_off = nullptr;  // [batch][nbucket+1] first task id of each bucket
    uint32_t* d_tasks = nullptr;     // [batch][max_tasks] uint2 task descriptors, longest first
    uint32_t* d_ntasks = nullptr;    // [batch]
    uint32_t* d_parts = nullptr;     // [batch][max_tasks] XYZZ partial bucket sums, bucket-major
    uint32_t* d_partials = nullptr;  // [batch][nbucket/16][2] XYZZ chunk sums
};
// the same workspace seen from batch item `off` on (chunks of one call run on different streams over disjoint item ranges)
CommitPlan commit_plan_offset(const CommitPlan& pl, int off);

void commit_plan_sizes(int c, int max_batch, CommitPlanSizes* sz);
// scalars: input_kind 0 = 32-byte big-endian scalars, 4096 per item, items `stride` bytes apart
void commit_batch_device(const CommitPlan& pl, const uint8_t* d_scalars, size_t stride, int input_kind, int n, uint8_t* d_out48, int* d_status, hipStream_t s, KernelTimer* kt);
void g1_decompress_device(const uint8_t* d_in48, int n, uint32_t* d_out_aff, int* d_status, hipStream_t s);
void g1_decompress2_device(const uint8_t* d_in48_a, int n_a, uint32_t* d_out_aff_a, int* d_st_a, const uint8_t* d_in48_b, int n_b, uint32_t* d_out_aff_b, int* d_st_b,
                           hipStream_t s);   // two arrays, one launch
void build_table_device(int c, const uint32_t* d_src_aff, uint32_t* d_table, int n, int bitrev_bits, hipStream_t s);
// direct-table commitment (window bits c in 12..15)
size_t direct_table_bytes(int c);
void build_table_any_device(int c, const uint32_t* d_src_aff, uint32_t* d_table, int n, int bitrev_bits, hipStream_t s);
void build_direct_device(int c, const uint32_t* d_table, uint32_t* d_direct, hipStream_t s);
// per-call workspace of the direct mode: up to 4096 lane accumulators per blob + one rejection flag per accumulate workgroup
struct DirectSlot {
    uint32_t* d_parts = nullptr;
    int* d_bad = nullptr;
    int k = 8;                 // scalars per lane of the accumulate kernel (8, 4, 2, 1): 4096 / k partial sums per blob; see direct_pick_shape
    int wsplit = 1;            // 2 (k == 1, up to eight blobs): a pair of lanes per scalar, lower / upper windows; 8192 partial sums per blob
    int serial = 1;            // 1: the finish kernel sums the 4096 / (64 k) records of a lane serially; 0: two butterfly folds first (calls of few blobs)
    int fin_quad = 1;          // serial form: the finish on four wavefronts per blob with a quad of lanes per record column (shorter chain, 2.1x the instructions)
#ifdef MKZG_AFFINE_PAIRS
    uint4* d_prefix = nullptr; // A/B builds with experiments/commit_pairs_kernels.hip: scratch of the pair kernel (direct_prefix_bytes per blob of max_batch), else null
#endif
};
void direct_pick_shape(int n, int max_batch, DirectSlot* slot);
size_t direct_parts_bytes(int max_batch);
size_t direct_bad_bytes(int max_batch);
#ifdef MKZG_AFFINE_PAIRS
size_t direct_prefix_bytes(int max_batch);                 // experiments/commit_pairs_kernels.hip
void commit_direct_acc_pairs_launch(int input_kind, dim3 grid, hipStream_t s, const uint8_t* d_scalars, size_t stride, const uint32_t* d_direct, const DirectSlot& ws);
#endif
DirectSlot direct_slot_offset(const DirectSlot& ws, int off);
double valu_calibrate_device(hipStream_t s);   // sustained v_mad_u64_u32 wave-instructions per second on this GPU
void commit_direct_acc_device(int c, const uint32_t* d_direct, const DirectSlot& ws, const uint8_t* d_scalars, size_t stride, int input_kind, int n, hipStream_t s, KernelTimer* kt);
// d_raw (may be null): n x 64 words that receive the sums as XYZZ limb words instead of d_out48 receiving compressed bytes (the caller converts on the host)
void commit_direct_fin_device(const DirectSlot& ws, int n, uint8_t* d_out48, int* d_status, hipStream_t s, KernelTimer* kt, uint32_t* d_raw = nullptr);
void commit_direct_device(int c, const uint32_t* d_direct, const DirectSlot& ws, const uint8_t* d_scalars, size_t stride, int input_kind, int n, uint8_t* d_out48,
                          int* d_status, hipStream_t s, KernelTimer* kt, uint32_t* d_raw = nullptr);

}  // namespace kzg
