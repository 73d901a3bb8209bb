// fk20_kernels.h -- host-visible launch interface of fk20_kernels.hip (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "../../include/mi355kzg.h"

namespace kzg {

// window width of the FK20 direct table: 11 bits (24 windows, 25.8 GB), 13 bits (20 windows, 86 GB: 17 % fewer additions per MSM)
// or 8 bits (32 windows, 4.3 GB: the width a context degrades to when the HBM is short)
int fk20_nwin(int cbits);
size_t fk20_table_bytes(int cbits);
// monomial SRS (4096 affine rows) -> T (8192 affine rows, [k][i]) -> direct table (fk20_table_bytes(cbits))
void fk20_setup_device(const uint32_t* d_monomial_aff, const uint32_t* d_g1tw_fwd, uint32_t* d_t_aff, uint32_t* d_table, int cbits, hipStream_t s);
constexpr int FK_SPLIT_MAX_BATCH = 8;    // up to this many blobs 48 lanes per MSM (24 windows x 2 halves of the 64 points): 128 x 48 partial sums per blob
constexpr int FK_DIRECT_MAX_BATCH = 16;  // up to this many blobs per call the MSM runs one window per lane (latency over occupancy)
constexpr int FK_MID_MAX_BATCH = 39;     // ... and up to this many two windows per lane (26 wavefronts per blob: 1014 at 39 blobs)
constexpr int FK_PAIR_MAX_BATCH = 32;    // ... and up to this many on a pair of lanes each
constexpr int FK_QUAD_MAX_BATCH = 16;    // up to this many blobs per call the 686 products of the transform network run on a quad of lanes each
size_t fk20_parts_rows();        // XYZZ rows of scratch per blob shared by the MSM and the transform stage

// scalars [n][128][64][8 words] -> parts [n][128][16] XYZZ rows (scratch) -> results [n][128] XYZZ rows
void fk20_msm_device(const uint32_t* d_scalars, const uint32_t* d_table, int cbits, int n, const int* d_status, uint32_t* d_parts, uint32_t* d_results, hipStream_t s);
// results [n][128] XYZZ rows + tmp of the same size -> proofs [n][128][48 B]; d_parts: fk20_parts_rows() x n XYZZ rows of scratch
// d_seg: fk20_seg_bytes(n) bytes when fk20_seg_count(n) != 0 (the products of calls of 65..192 blobs run in segments), else unused / nullptr
void fk20_ffts_device(uint32_t* d_results, uint32_t* d_tmp, int n, const int* d_status, uint8_t* d_proofs48, uint32_t* d_parts, void* d_seg, hipStream_t s);
int fk20_seg_count(int n);
size_t fk20_seg_bytes(int n);

}  // namespace kzg
