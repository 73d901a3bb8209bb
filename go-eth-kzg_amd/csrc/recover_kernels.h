// recover_kernels.h -- host-visible launch interface of recover_kernels.hip (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "../../include/mi355kzg.h"

namespace kzg {

// pow7: 8192 Fr rows of 7^i; ipow7: 4096 Fr rows of 7^-i
void recover_tables_device(uint32_t* d_pow7, uint32_t* d_ipow7, hipStream_t s);
size_t recover_scratch_words();
// n_blobs recoveries in one set of launches.  cells: every supplied cell of the call, contiguous (2048 B each); pos: per blob 192 ints --
// [0,128): index of cell id c in `cells` (global index) or -1, [128,192): bit-reversed ids of the absent cells; n_missing: per blob.
// scratch: n_blobs x recover_scratch_words(); coeffs: n_blobs x 4096 Fr rows; status: n_blobs.
void recover_coeffs_device(const uint8_t* d_cells, const int* d_pos, const int* d_n_missing, int n_blobs, const uint32_t* d_tw_fwd, const uint32_t* d_tw_inv,
                           const uint32_t* d_pow7, const uint32_t* d_ipow7, uint32_t* d_scratch, uint32_t* d_coeffs, int* d_status, hipStream_t s);

}  // namespace kzg
