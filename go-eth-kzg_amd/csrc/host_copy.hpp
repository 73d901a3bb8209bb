// host_copy.hpp -- the gather of host-pointer inputs (Go's []*Blob, []*Cell: scattered pageable arrays, reference serialization.go:56-95)
// into a lane's pinned staging area.
//
// The staging area is written once by a host core and read once by the GPU's DMA engine; nothing on the host reads it back.  A plain
// memcpy of a 128 KiB blob stays below glibc's non-temporal threshold, so every destination line is first READ into the cache
// (write-allocate) and later evicted: three units of DRAM traffic per byte gathered, and the staged bytes push the callers' blobs out
// of the last-level cache.  Streaming stores write the lines without reading them: two units.  Measured in tools/ubench_h2d.cpp.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace kzg {

#if defined(__x86_64__)
__attribute__((target("avx2"))) static inline void copy_stream_avx2(uint8_t* dst, const uint8_t* src, size_t n) {
    // dst 32-byte aligned, n a multiple of 128
    for (size_t i = 0; i < n; i += 128) {
        const __m256i a = _mm256_loadu_si256((const __m256i*)(src + i)), b = _mm256_loadu_si256((const __m256i*)(src + i + 32));
        const __m256i c = _mm256_loadu_si256((const __m256i*)(src + i + 64)), d = _mm256_loadu_si256((const __m256i*)(src + i + 96));
        _mm256_stream_si256((__m256i*)(dst + i), a);
        _mm256_stream_si256((__m256i*)(dst + i + 32), b);
        _mm256_stream_si256((__m256i*)(dst + i + 64), c);
        _mm256_stream_si256((__m256i*)(dst + i + 96), d);
    }
    _mm_sfence();               // the DMA engine is told about these bytes by another thread: make them globally visible first
}
static inline bool copy_have_avx2() {
    static const bool have = __builtin_cpu_supports("avx2");
    return have;
}
#endif

// dst: staging (written, never read by the host); falls back to memcpy for odd sizes / alignments / CPUs
static inline void copy_to_staging(uint8_t* dst, const uint8_t* src, size_t n) {
#if defined(__x86_64__)
    if (copy_have_avx2() && (((uintptr_t)dst) & 31) == 0 && (n & 127) == 0) { copy_stream_avx2(dst, src, n); return; }
#endif
    memcpy(dst, src, n);
}

}  // namespace kzg
