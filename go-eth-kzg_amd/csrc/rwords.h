// rwords.h -- the batch verifiers' random scalar r as eight little-endian words, passed to kernels BY VALUE (kernel arguments): a
// 32-byte hipMemcpyAsync from pinned memory leaves as a copy kernel that reads host memory, and in the cell verifier's trace that kernel
// did not finish before the point validation running on the side streams did (0.2 ms per call between the last upload and the first
// scalar kernel).
#pragma once
#include <stdint.h>
namespace kzg {
struct RWords { uint32_t w[8]; };
}  // namespace kzg
