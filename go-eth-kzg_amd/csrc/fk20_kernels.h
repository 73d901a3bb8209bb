// fk20_kernels.h -- host-visible launch interface of fk20_kernels.hip (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "../../include/mi355kzg.h"

namespace kzg {

constexpr int FK_C = 11;         // window bits of the direct table
constexpr int FK_WPL = 3;        // windows per lane of k_fk20_msm
constexpr int FK_LANES = 8;      // lanes per MSM-64 (FK_LANES * FK_WPL = FK_NWIN windows; 64 / FK_LANES MSMs per wavefront)
constexpr int FK_NWIN = 24;      // 11-bit windows of a 255-bit scalar (264 bits)
constexpr int FK_NMULT = 1024;   // multiples 1..1024 (signed Booth digits)

size_t fk20_table_bytes();
// monomial SRS (4096 affine rows) -> T (8192 affine rows, [k][i]) -> direct table (fk20_table_bytes())
void fk20_setup_device(const uint32_t* d_monomial_aff, const uint32_t* d_g1tw_fwd, uint32_t* d_t_aff, uint32_t* d_table, hipStream_t s);
constexpr int FK_DIRECT_MAX_BATCH = 8;   // up to this many blobs per call the MSM runs one window per lane (latency over occupancy)
size_t fk20_parts_rows();        // XYZZ rows of scratch per blob shared by the MSM and the transform stage

// scalars [n][128][64][8 words] -> parts [n][128][16] XYZZ rows (scratch) -> results [n][128] XYZZ rows
void fk20_msm_device(const uint32_t* d_scalars, const uint32_t* d_table, int n, const int* d_status, uint32_t* d_parts, uint32_t* d_results, hipStream_t s);
// results [n][128] XYZZ rows + tmp of the same size -> proofs [n][128][48 B]; d_parts: fk20_parts_rows() x n XYZZ rows of scratch
void fk20_ffts_device(uint32_t* d_results, uint32_t* d_tmp, int n, const int* d_status, uint8_t* d_proofs48, uint32_t* d_parts, hipStream_t s);

}  // namespace kzg
