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
constexpr int FK_PLAN_WORDS = 128 * 8 + 64 * 16 + 128 * 8 + 128 * 8 + 128 * 64 + 128 * 64;
constexpr int FK_DIRECT_MAX_BATCH = 8;   // up to this many blobs per call the G1 transforms run as single-level DFTs (latency over work)   // lanes of the four DFT levels (see fk20_kernels.hip)
void fk20_build_plan(uint32_t* plan /* FK_PLAN_WORDS, host */);

// scalars [n][128][64][8 words] -> parts [n][128][16] XYZZ rows (scratch) -> results [n][128] XYZZ rows
void fk20_msm_device(const uint32_t* d_scalars, const uint32_t* d_table, int n, const int* d_status, uint32_t* d_parts, uint32_t* d_results, hipStream_t s);
// results [n][128] XYZZ rows (clobbered) + tmp of the same size -> proofs [n][128][48 B]
void fk20_ffts_device(uint32_t* d_results, uint32_t* d_tmp, const uint32_t* d_plan, int n, const int* d_status, uint8_t* d_proofs48, uint32_t* d_parts /* >= 1024 x n XYZZ rows */, hipStream_t s);

}  // namespace kzg
