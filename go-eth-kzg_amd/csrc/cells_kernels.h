// cells_kernels.h -- host-visible launch interface of cells_kernels.hip (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rwords.h"
#include "../../include/mi355kzg.h"

namespace kzg {

// tw_fwd / tw_inv: 4096 Fr rows (12 words) of w_8192^(+-j); g1tw_*: 64 x 8 plain words of w_128^(+-j)
void build_twiddles_device(uint32_t* d_tw_fwd, uint32_t* d_tw_inv, uint32_t* d_g1tw_fwd, uint32_t* d_g1tw_inv, hipStream_t s);
// blobs (n x 131072 B) -> coefficient rows (n x 4096 x 12 words, may be null) + extended blob bytes (n x 262144 B) + status
void cells_from_blobs_device(const uint8_t* d_blobs, int n, const uint32_t* d_tw_fwd, const uint32_t* d_tw_inv, uint32_t* d_coeffs, uint8_t* d_cells, int* d_status,
                             hipStream_t s);
// coefficient rows -> extended blob bytes (items with status != 0 are skipped)
void cells_from_coeffs_device(const uint32_t* d_coeffs, int n, const uint32_t* d_tw_fwd, const uint32_t* d_tw_inv, uint8_t* d_cells, int* d_status, hipStream_t s);
// coefficient rows -> FK20 scalars [n][128][64][8 words]
void fk20_scalars_device(const uint32_t* d_coeffs, int n, const uint32_t* d_tw_fwd, const int* d_status, uint32_t* d_scalars, hipStream_t s);

// VerifyCellKZGProofBatch scalar work: r -> r^k rows, s1 (r^k), s4 (r^k h^64), per-row weights, interpolation
// polynomial (64 plain scalars); d_partials: CELL_INTERP_BLOCKS x 64 Fr rows of scratch
constexpr int CELL_INTERP_BLOCKS = 2048;
// ... followed by one row set per group of 64 workgroups and the tickets of k_cell_interp's own final sum (ZERO when allocated; the kernel resets them)
constexpr size_t CELL_INTERP_SCRATCH_BYTES = ((size_t)CELL_INTERP_BLOCKS * 64 + (size_t)(CELL_INTERP_BLOCKS / 64) * 64) * 48 + 64 * 4;
// h_r8: HOST pointer to the eight words of r (it travels in the kernel arguments)
void cell_verify_scalars_device(const uint32_t* h_r8, uint32_t exp_offset, const uint64_t* d_cell_idx, const uint32_t* d_row_idx, const uint8_t* d_cells, int n, int n_rows,
                                const uint32_t* d_tw_fwd, const uint32_t* d_tw_inv, uint32_t* d_rp_rows, uint32_t* d_s1, uint32_t* d_s4, uint32_t* d_weights8,
                                uint32_t* d_partials, uint32_t* d_interp8, int* d_cell_status, hipStream_t s,
                                int phase = 0 /* 1: the launches that need the indices and r only; 2: those that read the cells; 0: all */);

}  // namespace kzg
