// pub_ticket.hpp -- device-only helpers (included by the kernel sources)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
namespace kzg {
// Publishing a few words to ANOTHER workgroup of the same launch (the tickets of the kernels that fold their own last step).
// __threadfence() on gfx950 is `buffer_wbl2 sc1` + `buffer_inv sc1`: the whole L2 of the XCD written back and invalidated -- once per
// workgroup that was hundreds of L2 flushes per launch, and every kernel running beside it lost its cached rows (*measured*: the cell
// verifier's window sums 0.62 -> 1.5 ms).  Agent-scope RELAXED atomics are plain stores / loads with the sc1 bit (coherent across the
// XCDs, nothing flushed); in-order issue plus s_waitcnt 0 before the ticket's atomic makes the data visible before the ticket is.
__device__ __forceinline__ void pub_store(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t pub_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t pub_ticket(uint32_t* t) {
    __builtin_amdgcn_s_waitcnt(0);
    return __hip_atomic_fetch_add(t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
}  // namespace kzg
