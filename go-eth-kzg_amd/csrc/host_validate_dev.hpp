// host_validate_dev.hpp -- the host-side point validation of host_validate.hpp handed to the device: points validated on the host pool and the
// calling thread go to the places where the validation kernels leave their records (verify_kernels.h G1Rows / g1_put_rows_device).
// Included by the host layer (api_verify.hip, api_cells.hip); host_validate.hpp itself stays free of HIP types (CPU unit tests build it alone).
#pragma once
#include <algorithm>
#include <functional>
#include "host_validate.hpp"
#include "verify_kernels.h"

namespace kzg {

// na + nb points (two arrays of 48-byte encodings) validated on the pool and the calling thread, one task per point, and handed to the device
// in chunks of G1_ROWS_MAX records (verify_kernels.h G1Rows / g1_put_rows_device).  d_sub_*: see g1_put_rows_device.
constexpr size_t HOST_VALIDATE_MAX_POINTS = 64;          // blob verifier: four rounds of sixteen pool threads, ~0.2 ms against the device's 0.64-0.7 ms chain
constexpr size_t HOST_VALIDATE_MAX_CELL_POINTS = 96;     // cell verifier (A/B, tools/lat_cellverify.py: 81 points 1.28 -> 1.17 ms, 129 points 1.30 -> 1.36-1.55): eight columns of nine blobs
// Does the host pay?  ~45 us per point spread over the pool's workers and the caller must stay under the device's ~0.6 ms chain with room
// to spare: a rank of an 8-GPU job on a 16-CPU quota has ONE worker (16 points), a lone context on the same box fifteen.
inline bool host_validate_pays(size_t points, int pool_workers, size_t cap) {
    return points <= cap && points <= (size_t)8 * (size_t)(pool_workers + 1);      // measured with MKZG_HOST_THREADS=2: 12 points 1.36 against 1.58 ms, 18 points 1.64 against 1.58
}
template <class Pool>
inline void host_validate_arrays(Pool* pool, const uint8_t* a48n, size_t na, const uint8_t* b48n, size_t nb, uint32_t* d_aff_a, int* d_st_a, uint32_t* d_aff_b, int* d_st_b,
                                 hipStream_t s, int* d_sub_a = nullptr, int* d_sub_b = nullptr) {
    constexpr size_t CH = (size_t)G1_ROWS_MAX;
    const size_t total = na + nb;
    G1Rows rp[(HOST_VALIDATE_MAX_CELL_POINTS + CH - 1) / CH];
    const std::function<void(int)> task = [&](int t) {           // point t of the concatenation a | b sits in chunk t / CH, slot t % CH
        const size_t k = (size_t)t / CH, slot = (size_t)t % CH;
        const uint8_t* src = (size_t)t < na ? a48n + 48 * (size_t)t : b48n + 48 * ((size_t)t - na);
        const int st = g1h_validate_row(src, rp[k].row[slot]);
        rp[k].st[slot] = st == MKZG_ERR_G1_NOT_IN_SUBGROUP ? 0 : st;
        rp[k].sub[slot] = st == MKZG_ERR_G1_NOT_IN_SUBGROUP ? st : 0;
    };
    pool->parallel((int)total, task);
    for (size_t k = 0; k * CH < total; k++) {
        const size_t lo = k * CH, hi = std::min(total, lo + CH);
        const size_t a_lo = std::min(lo, na), a_hi = std::min(hi, na);
        rp[k].na = (int)(a_hi - a_lo);
        rp[k].nb = (int)((hi - lo) - (a_hi - a_lo));
        rp[k].first_a = (int)a_lo;
        rp[k].first_b = (int)(lo > na ? lo - na : 0);
        g1_put_rows_device(rp[k], d_aff_a, d_st_a, d_aff_b, d_st_b, s, d_sub_a, d_sub_b);
    }
}

}  // namespace kzg
