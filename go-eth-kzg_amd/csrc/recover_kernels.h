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
// cells: the supplied cells, contiguous (n x 2048 B); cell_pos[128]: index of cell id among them or -1;
// missing: bit-reversed ids of the absent cells.  Output: 4096 coefficient rows; status[0].
void recover_coeffs_device(const uint8_t* d_cells, const int* d_cell_pos, const uint32_t* d_missing, int n_missing, const uint32_t* d_tw_fwd, const uint32_t* d_tw_inv,
                           const uint32_t* d_pow7, const uint32_t* d_ipow7, uint32_t* d_scratch, uint32_t* d_coeffs, int* d_status, hipStream_t s);

}  // namespace kzg
