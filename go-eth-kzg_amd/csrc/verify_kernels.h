// verify_kernels.h -- host-visible launch interface of verify_kernels.hip (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rwords.h"
#include "../../include/mi355kzg.h"

namespace kzg {

// Fr rows are 12 words (9 limbs + padding), affine G1 rows 32 words, as in msm_kernels.hip.
void challenge_device(const uint8_t* d_blobs, const uint8_t* d_commitments, int n, uint32_t* d_z, hipStream_t s,
                      uint32_t* d_zrows = nullptr /* blob_eval_scratch_words() per blob: also leave the rows blob_eval_device needs of every z (rows_ready) */);
// the same for up to SHA_MAX_JOBS (blobs, commitments, n, z) jobs -- one per call in flight -- in ONE launch; every n > 0
#ifndef MKZG_NLANE
#define MKZG_NLANE 32
#endif
constexpr int SHA_MAX_JOBS = MKZG_NLANE;        // one job per lane at most
struct ShaJobs {
    const uint8_t* blobs[SHA_MAX_JOBS];
    const uint8_t* commitments[SHA_MAX_JOBS];
    uint32_t* z_out[SHA_MAX_JOBS];
    uint32_t* zrows_out[SHA_MAX_JOBS];      // per job, may be null: blob_eval_scratch_words() words per blob, what the tree evaluation needs of z alone
    int n[SHA_MAX_JOBS];
    int first_block[SHA_MAX_JOBS];
    int count;
};
void challenge_jobs_device(ShaJobs& jobs, hipStream_t s);
// DEEP ROUNDS (api_verify.hip): what follows the challenges of a round's calls -- y = f(z) with the calls' shares of the random linear
// combination, then their three MSMs each -- as ONE launch per step over up to DEEP_MAX_JOBS calls (kernel arguments: 1.7 KB / 1.0 KB)
constexpr int DEEP_MAX_JOBS = 16;
struct EvalJobs {
    const uint8_t* blobs[DEEP_MAX_JOBS];
    const uint32_t* zrows[DEEP_MAX_JOBS];       // from the round's hash launch (ShaJobs::zrows_out)
    uint32_t* y_out[DEEP_MAX_JOBS];
    int* status[DEEP_MAX_JOBS];
    uint32_t* s1[DEEP_MAX_JOBS];                // r^i, r^i z_i (the MSMs' scalars), r^i y_i (for the host), 8 plain words per blob
    uint32_t* s3[DEEP_MAX_JOBS];
    uint32_t* ry[DEEP_MAX_JOBS];
    uint32_t r[DEEP_MAX_JOBS][8];
    uint32_t exp_offset[DEEP_MAX_JOBS];
    int n[DEEP_MAX_JOBS];
    int first_block[DEEP_MAX_JOBS];
    int count;
};
struct VmsmCalls {
    const uint32_t* s1[DEEP_MAX_JOBS];
    const uint32_t* s3[DEEP_MAX_JOBS];
    const uint32_t* aff_q[DEEP_MAX_JOBS];       // the call's proofs / commitments as affine rows
    const uint32_t* aff_c[DEEP_MAX_JOBS];
    uint32_t* parts[DEEP_MAX_JOBS];             // 3 x [32][VMSM_KMAX][64] words of slice sums (tickets zero)
    uint32_t* out[DEEP_MAX_JOBS];               // 3 x [32][48] words
    int n[DEEP_MAX_JOBS];
    int k[DEEP_MAX_JOBS];
    int count;
};
void blob_eval_jobs_device(EvalJobs& jobs, const uint32_t* d_roots, hipStream_t s);
int vmsm_calls_device(VmsmCalls& calls, hipStream_t s);
// d_zrows: n x blob_eval_scratch_words() words of scratch (what depends on z alone, from a one-lane-per-blob kernel ahead of the evaluation)
size_t blob_eval_scratch_words();
// rows_ready: d_zrows were written by the challenge launch already (ShaJobs::zrows_out, challenge_device's d_zrows).  h_r8 (HOST pointer to the eight words of r, may be
// null): the evaluation kernel's last lane also leaves the blob's share of the random linear combination -- d_s1[b] = r^(exp_offset + b),
// d_s3[b] = r^i z_b (8 plain words each, the MSM scalars), d_ry[b] = r^i y_b (8 plain words: the host adds them up) -- instead of a
// k_rlc_scalars launch behind it
void blob_eval_device(const uint8_t* d_blobs, const uint32_t* d_z, const uint32_t* d_roots, const uint32_t* d_inv_roots, int n, uint32_t* d_y, int* d_status, uint32_t* d_zrows,
                      hipStream_t s, bool rows_ready = false, const uint32_t* h_r8 = nullptr, uint32_t exp_offset = 0, uint32_t* d_s1 = nullptr, uint32_t* d_s3 = nullptr,
                      uint32_t* d_ry = nullptr);
void blob_quotient_device(const uint8_t* d_blobs, const uint32_t* d_z, const uint32_t* d_roots, const uint32_t* d_inv_roots, int n, uint32_t* d_y, uint32_t* d_quot,
                          int* d_status, hipStream_t s);
// y_b = f_b(z_b) from coefficient rows (n x 4096 x 12 words, as cells_from_blobs_device publishes them); blobs with a non-zero status are skipped
void poly_eval_device(const uint32_t* d_coeffs, const uint32_t* d_z, int n, uint32_t* d_y, const int* d_status, hipStream_t s);
// decode (gnark SetBytes checks) + subgroup test of two arrays of compressed points in one launch: affine rows and status words out
void g1_validate2_device(const uint8_t* d_in48_a, int n_a, uint32_t* d_aff_a, int* d_st_a, const uint8_t* d_in48_b, int n_b, uint32_t* d_aff_b, int* d_st_b, hipStream_t s,
                         int mode = 0 /* 0: decode + subgroup test (a lane quad per point); 1: decode only; 2: subgroup test only, verdicts to d_st_*; 3: as 0, one lane per point */);
// The affine records and status words of up to G1_ROWS_MAX points validated on the HOST (host_validate.hpp: the single-proof entry points
// and the verifiers' calls of a few items), into the places where g1_validate2_device leaves them -- by value in the kernel arguments
// (3.3 KB): no staging buffer, no copy kernel that reads host memory.  A chunk holds na records of array a (the blob verifier's
// commitments, the cell verifier's distinct commitments) from index first_a on, then nb records of array b (the proofs) from first_b on;
// larger sets go out as several chunks.  st: the decoding status (0, MKZG_ERR_G1_BAD_ENCODING, MKZG_ERR_G1_NOT_ON_CURVE); sub: 0 or
// MKZG_ERR_G1_NOT_IN_SUBGROUP.  With d_sub_* null the two are one status word as k_g1_validate<0> writes it (decoding error first).
constexpr int G1_ROWS_MAX = 24;      // records per chunk
struct G1Rows {
    uint32_t row[G1_ROWS_MAX][32];
    int st[G1_ROWS_MAX];
    int sub[G1_ROWS_MAX];
    int na, nb, first_a, first_b;
};
void g1_put_rows_device(const G1Rows& rp, uint32_t* d_aff_a, int* d_st_a, uint32_t* d_aff_b, int* d_st_b, hipStream_t s, int* d_sub_a = nullptr, int* d_sub_b = nullptr);
// r^(exp_offset + i) for i < n (exp_offset: global index of the shard's first proof)
// h_r8: HOST pointer to the eight words of r
void rlc_scalars_device(const uint32_t* h_r8, const uint32_t* d_z, const uint32_t* d_y, int n, uint32_t exp_offset, uint32_t* d_s1, uint32_t* d_s3, uint32_t* d_e,
                        int mode /* 0 all, 1 s1 only, 2 s3 and e only */, hipStream_t s);
// n_msm <= 4 jobs of n_per_job[m] <= 8192 (scalar, point) pairs; d_out[msm][32 windows][48 words]
// gnark-crypto Montgomery layouts (fr.Element 4 x u64, G1Affine 12 x u64) -> plain 8-word scalars + radix-2^29 affine rows
void gnark_to_vmsm_device(const uint32_t* d_scalars_mont, const uint32_t* d_points_mont, int n, uint32_t* d_scalars8, uint32_t* d_aff, hipStream_t s);
constexpr int VMSM_KMAX = 8;                                        // slices of the point range per (job, window)
constexpr size_t VMSM_PART_WORDS = (size_t)4 * 32 * VMSM_KMAX * 64;   // scratch of vmsm_windows_device (XYZZ rows)
int vmsm_windows_device(const uint32_t* const* d_scalars, const uint32_t* const* d_points, const int* n_per_job, int n_msm, uint32_t* d_parts, uint32_t* d_out, hipStream_t s,
                        int points_per_slice = 2048 /* 2048: 8192-point jobs fill the GPU in one round of workgroups */,
                        int latency = 0 /* 1: a caller waits for this launch alone -- a quad of lanes per bucket when the grid leaves the chip mostly empty */);
void build_roots_device(uint32_t* d_roots_brp, uint32_t* d_inv_roots_brp, int log_n, hipStream_t s);
void scalars_to_fr_device(const uint8_t* d_in32, int n, uint32_t* d_out12, int* d_status, hipStream_t s);
void fr_to_scalars_device(const uint32_t* d_in12, int n, uint8_t* d_out32, hipStream_t s);
constexpr int VMSM_WINDOWS = 32, VMSM_WINDOW_BITS = 8, VMSM_MAX_N = 8192;

}  // namespace kzg
