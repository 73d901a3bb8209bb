// api_verify.hip -- C ABI entry points of the EIP-4844 proof paths (include/mi355kzg.h):
//   Context.VerifyBlobKZGProofBatch / VerifyBlobKZGProof / VerifyKZGProof   reference verify.go:12-143
//   Context.ComputeBlobKZGProof / ComputeKZGProof                           reference prove.go:46-111
// What runs where: every per-blob / per-point operation (SHA-256 challenge, canonical checks,
// barycentric evaluation, quotient, point decoding, subgroup checks, all MSM bucket work) is a
// HIP kernel.  The host does, per BATCH: draw r, combine 3 x 32 window sums by Horner, one
// fixed-base scalar multiplication, and the 2-pairing product check with prepared G2 lines
// (reference internal/kzg/kzg_verify.go:160-199).
#include <sys/random.h>
#include <chrono>
#include <thread>
#include "ctx.h"

// ---------------------------------------------------------------------------------------------
// workspaces
// ---------------------------------------------------------------------------------------------
static const size_t V_MISC_WORDS = 16 + 4 * VMSM_WINDOWS * 48;
#define KT(kind, s, call)                                  \
    do {                                                   \
        if (c->profiling) c->timer.begin(kind, s);         \
        call;                                              \
        if (c->profiling) c->timer.end(kind, s);           \
    } while (0)

void free_verify_ws(mkzg_ctx* c) {
    void* ptrs[] = {c->d_v_raw, c->d_v_aff, c->d_v_st, c->d_v_z, c->d_v_y, c->d_v_s1, c->d_v_s3, c->d_v_misc, c->d_quot, c->d_q_z, c->d_q_y, c->d_q_st, c->d_v_coeffs};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (c->h_v) (void)hipHostFree(c->h_v);
    c->d_v_raw = nullptr; c->d_v_aff = nullptr; c->d_v_st = nullptr; c->d_v_z = nullptr; c->d_v_y = nullptr; c->d_v_s1 = nullptr; c->d_v_s3 = nullptr;
    c->d_v_misc = nullptr; c->d_quot = nullptr; c->d_q_z = nullptr; c->d_q_y = nullptr; c->d_q_st = nullptr; c->h_v = nullptr;
    c->d_v_coeffs = nullptr; c->v_coeff_cap = 0;
    c->v_cap = 0; c->q_cap = 0; c->quot_cap = 0; c->h_v_bytes = 0;
}
int ensure_verify_ws(mkzg_ctx* c, size_t n) {
    if (n <= c->v_cap) return MKZG_OK;
    size_t cap = std::max<size_t>(n, 64);
    void* ptrs[] = {c->d_v_raw, c->d_v_aff, c->d_v_st, c->d_v_z, c->d_v_y, c->d_v_s1, c->d_v_s3, c->d_v_misc};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (c->h_v) (void)hipHostFree(c->h_v);
    c->v_cap = 0;
    HIPCHK(hipMalloc(&c->d_v_raw, 2 * cap * 48));
    HIPCHK(hipMalloc(&c->d_v_aff, 2 * cap * 128));
    HIPCHK(hipMalloc(&c->d_v_st, 3 * cap * sizeof(int)));
    HIPCHK(hipMalloc(&c->d_v_z, cap * 48));
    HIPCHK(hipMalloc(&c->d_v_y, cap * 48));
    HIPCHK(hipMalloc(&c->d_v_s1, cap * 32));
    HIPCHK(hipMalloc(&c->d_v_s3, cap * 32));
    HIPCHK(hipMalloc(&c->d_v_misc, (V_MISC_WORDS + VMSM_PART_WORDS) * 4));      // window sums for the host + slice sums (device only)
    c->h_v_bytes = 3 * cap * sizeof(int) + V_MISC_WORDS * 4 + 64;
    HIPCHK(hipHostMalloc(&c->h_v, c->h_v_bytes, hipHostMallocDefault));
    c->v_cap = cap;
    return MKZG_OK;
}
// n: scalars / statuses per call (z, y, status arrays); n_quot: polynomials whose 4096 quotient rows are held at once
int ensure_quot_ws(mkzg_ctx* c, size_t n, size_t n_quot) {
    if (n_quot > c->quot_cap) {
        if (c->d_quot) (void)hipFree(c->d_quot);
        c->d_quot = nullptr;
        c->quot_cap = 0;
        HIPCHK(hipMalloc(&c->d_quot, n_quot * 4096 * 48));
        c->quot_cap = n_quot;
    }
    if (n <= c->q_cap) return MKZG_OK;
    void* ptrs[] = {c->d_q_z, c->d_q_y, c->d_q_st};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    c->d_q_z = c->d_q_y = nullptr;
    c->d_q_st = nullptr;
    c->q_cap = 0;
    HIPCHK(hipMalloc(&c->d_q_z, n * 48));
    HIPCHK(hipMalloc(&c->d_q_y, n * 48));
    HIPCHK(hipMalloc(&c->d_q_st, 2 * n * sizeof(int)));
    c->q_cap = n;
    return MKZG_OK;
}

// ---------------------------------------------------------------------------------------------
// verification core.  Inputs already on the device:
//   d_commitments, d_proofs : n x 48 B compressed points
//   either d_blobs (n x 131072 B; z and y are derived) or d_z32/d_y32 (n x 32 B big-endian scalars)
// Error precedence mirrors the reference loops (verify.go:102-140, verify.go:15-33).
// ---------------------------------------------------------------------------------------------
struct VerifyFold { G1HX A, Cc, D; uint32_t e[8]; };
// device work + Horner: A = sum r^i Q_i, Cc = sum r^i C_i, D = sum r^i z_i Q_i, e = sum r^i y_i with i = exp_offset + local index;
// r32 == nullptr: draw r here.
static int verify_fold(mkzg_ctx* c, const uint8_t* d_blobs, const uint8_t* d_z32, const uint8_t* d_y32, const uint8_t* d_commitments, const uint8_t* d_proofs, size_t n,
                       uint32_t exp_offset, const uint8_t* r32, hipStream_t s, VerifyFold* out) {
    if (n > (size_t)VMSM_MAX_N) return MKZG_ERR_BAD_ARGS;
    int rc = ensure_verify_ws(c, n);
    if (rc) return rc;
    if (d_blobs) {
        // coefficient rows of the blobs (inverse NTT beside the challenge hash) and the NTT twiddles they need
        rc = ensure_cells(c, 0);
        if (rc) return rc;
        if (n > c->v_coeff_cap) {
            if (c->d_v_coeffs) (void)hipFree(c->d_v_coeffs);
            c->d_v_coeffs = nullptr;
            c->v_coeff_cap = 0;
            HIPCHK(hipMalloc(&c->d_v_coeffs, n * 4096 * 48));
            c->v_coeff_cap = n;
        }
    }
    const size_t cap = c->v_cap;
    uint32_t* aff_c = c->d_v_aff;
    uint32_t* aff_q = c->d_v_aff + cap * 32;
    int* st_c = c->d_v_st;
    int* st_q = c->d_v_st + cap;
    int* st_b = c->d_v_st + 2 * cap;
    // 0. the random linear combination scalar r (kzg_verify.go:136-141) is drawn first: everything that needs r but not the
    //    challenges runs beside the challenge hash
    uint32_t* h_misc = (uint32_t*)(c->h_v + 3 * cap * sizeof(int));
    uint8_t rbytes[32];
    if (r32) {
        for (int i = 0; i < 32; i++) rbytes[i] = r32[31 - i];        // big-endian bytes -> little-endian words
    } else if (getrandom(rbytes, 32, 0) != 32) {
        return MKZG_ERR_HIP;
    }
    memcpy(h_misc, rbytes, 32);
    uint32_t* d_parts = c->d_v_misc + V_MISC_WORDS;
    uint32_t* d_wsum = c->d_v_misc + 16;
    const size_t parts_per_job = (size_t)VMSM_WINDOWS * VMSM_KMAX * 64, wsum_per_job = (size_t)VMSM_WINDOWS * 48;
    int vrc = 0;
    // 1. decode + validate the points (serialization.go:108-115)
    //    on the side stream, beside the (longer, latency-bound) SHA-256 chain
    hipStream_t s2 = c->side;
    HIPCHK(hipEventRecord(c->ev_fork, s));
    HIPCHK(hipStreamWaitEvent(s2, c->ev_fork, 0));
    KT(KT_SUBGROUP, s2, g1_validate2_device(d_commitments, (int)n, aff_c, st_c, d_proofs, (int)n, aff_q, st_q, s2));
    if (d_blobs) {
        //    ... and, for blobs, their canonicity check and inverse NTT (DeserializeBlob, serialization.go:134-146): the claimed value
        //    y = f(z) is then a 4096-term polynomial evaluation once z is known, instead of the barycentric formula with its inversions;
        //    the powers of r and the two MSMs that use nothing else (sum r^i Q_i, sum r^i C_i: kzg_verify.go:150,225)
        KT(KT_CELLS_NTT, s2, cells_from_blobs_device(d_blobs, (int)n, c->d_tw_fwd, c->d_tw_inv, c->d_v_coeffs, nullptr, st_b, s2));
        HIPCHK(hipMemcpyAsync(c->d_v_misc, h_misc, 32, hipMemcpyHostToDevice, s2));
        KT(KT_RLC, s2, rlc_scalars_device(c->d_v_misc, nullptr, nullptr, (int)n, exp_offset, c->d_v_s1, nullptr, nullptr, 1, s2));
        const uint32_t* scal[2] = {c->d_v_s1, c->d_v_s1};
        const uint32_t* pts[2] = {aff_q, aff_c};
        const int njob[2] = {(int)n, (int)n};
        KT(KT_VMSM, s2, vrc = vmsm_windows_device(scal, pts, njob, 2, d_parts, d_wsum, s2));
        if (vrc) return MKZG_ERR_BAD_ARGS;
    }
    HIPCHK(hipEventRecord(c->ev_join, s2));
    // 2. challenge and claimed value per blob, or caller-supplied z / y
    int y_bad = 0;
    if (d_blobs) {
        KT(KT_SHA256, s, challenge_device(d_blobs, d_commitments, (int)n, c->d_v_z, s));
        HIPCHK(hipStreamWaitEvent(s, c->ev_join, 0));
        KT(KT_EVAL, s, poly_eval_device(c->d_v_coeffs, c->d_v_z, (int)n, c->d_v_y, st_b, s));
        // 3. what depends on the challenges: r^i z_i, sum r^i y_i, and the MSM sum r^i z_i Q_i (kzg_verify.go:179)
        KT(KT_RLC, s, rlc_scalars_device(c->d_v_misc, c->d_v_z, c->d_v_y, (int)n, exp_offset, c->d_v_s1, c->d_v_s3, c->d_v_misc + 8, 2, s));
        const uint32_t* scal[1] = {c->d_v_s3};
        const uint32_t* pts[1] = {aff_q};
        const int njob[1] = {(int)n};
        KT(KT_VMSM, s, vrc = vmsm_windows_device(scal, pts, njob, 1, d_parts + 2 * parts_per_job, d_wsum + 2 * wsum_per_job, s, 256));
    } else {
        // VerifyKZGProof checks y before z (verify.go:15-23); st_b ends up non-zero if either fails
        scalars_to_fr_device(d_z32, (int)n, c->d_v_z, st_b, s);
        int* st_y = (int*)(c->d_v_misc + 16);   // scratch inside the window-sum area (overwritten later)
        scalars_to_fr_device(d_y32, (int)n, c->d_v_y, st_y, s);
        HIPCHK(hipStreamWaitEvent(s, c->ev_join, 0));
        {
            // need the y-status before it is overwritten by the window sums
            std::vector<int> sty(n);
            HIPCHK(hipMemcpyAsync(sty.data(), c->d_v_misc + 16, n * sizeof(int), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            for (int v : sty) if (v) y_bad = v;
        }
        HIPCHK(hipMemcpyAsync(c->d_v_misc, h_misc, 32, hipMemcpyHostToDevice, s));
        KT(KT_RLC, s, rlc_scalars_device(c->d_v_misc, c->d_v_z, c->d_v_y, (int)n, exp_offset, c->d_v_s1, c->d_v_s3, c->d_v_misc + 8, 0, s));
        // the three MSMs of kzg_verify.go:150,225,179: sum r^i Q_i, sum r^i C_i, sum r^i z_i Q_i
        const uint32_t* scal[3] = {c->d_v_s1, c->d_v_s1, c->d_v_s3};
        const uint32_t* pts[3] = {aff_q, aff_c, aff_q};
        const int njob[3] = {(int)n, (int)n, (int)n};
        KT(KT_VMSM, s, vrc = vmsm_windows_device(scal, pts, njob, 3, d_parts, d_wsum, s));
    }
    if (vrc) return MKZG_ERR_BAD_ARGS;
    HIPCHK(hipGetLastError());
    // 5. one round trip: statuses, e, window sums
    int* h_st = (int*)c->h_v;
    HIPCHK(hipMemcpyAsync(h_st, c->d_v_st, 3 * cap * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(h_misc, c->d_v_misc, V_MISC_WORDS * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (d_blobs) {
        for (size_t i = 0; i < n; i++) {                           // verify.go:102-123 order: commitment, proof, blob
            if (h_st[i]) return h_st[i];
            if (h_st[cap + i]) return h_st[cap + i];
            if (h_st[2 * cap + i]) return h_st[2 * cap + i];
        }
    } else {
        for (size_t i = 0; i < n; i++) {                           // verify.go:15-33 order: y, z, commitment, proof
            if (y_bad) return y_bad;
            if (h_st[2 * cap + i]) return h_st[2 * cap + i];
            if (h_st[i]) return h_st[i];
            if (h_st[cap + i]) return h_st[cap + i];
        }
    }
    // 6. Horner over the window sums (the 255 serial doublings stay on the host)
    auto t_h0 = std::chrono::steady_clock::now();
    const uint32_t* wsum = h_misc + 16;
    auto horner = [&](int m, G1HX& out) { g1h_horner_words(out, wsum + (size_t)m * VMSM_WINDOWS * 48, VMSM_WINDOWS, VMSM_WINDOW_BITS); };
    {   // three independent 255-doubling chains: one per host thread
        G1HX* dst[3] = {&out->D, &out->A, &out->Cc};
        const int job[3] = {2, 0, 1};
        c->pool->parallel(3, [&](int i) { horner(job[i], *dst[i]); });
    }
    memcpy(out->e, h_misc + 8, 32);
    c->last_host_finish_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_h0).count();
    return MKZG_OK;
}
// kzg_verify.go:160-199: F = Cc - [e]G + D ; e(F, G2) e(-A, [alpha]G2) == 1
static int verify_finish(mkzg_ctx* c, const VerifyFold& f) {
    auto t_h0 = std::chrono::steady_clock::now();
    G1HX eG, gen, Fsum, negA;
    uint64_t e64[4];
    for (int i = 0; i < 4; i++) e64[i] = (uint64_t)f.e[2 * i] | ((uint64_t)f.e[2 * i + 1] << 32);
    if (!c->gen_fixed.ready) {
        g1h_generator(gen);
        g1h_fixed_build(c->gen_fixed, gen);
    }
    g1h_fixed_mul(eG, c->gen_fixed, e64);
    G1H::neg(eG, eG);
    G1H::add(Fsum, f.Cc, eG);
    G1H::add(Fsum, Fsum, f.D);
    G1H::neg(negA, f.A);
    G1AffH ps[2];
    g1h_to_affine(ps[0], Fsum);
    g1h_to_affine(ps[1], negA);
    const G2Prepared* qs[2] = {&c->g2_gen, &c->g2_alpha};
    bool ok = pairing_check_prepared_2mt(ps, qs, c->pool);
    c->last_host_finish_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_h0).count();
    return ok ? MKZG_OK : MKZG_VERIFY_FAILED;
}
static int verify_core(mkzg_ctx* c, const uint8_t* d_blobs, const uint8_t* d_z32, const uint8_t* d_y32, const uint8_t* d_commitments, const uint8_t* d_proofs, size_t n,
                       hipStream_t s) {
    if (n == 0) return MKZG_OK;                                   // kzg_verify.go:121-123
    VerifyFold f;
    int rc = verify_fold(c, d_blobs, d_z32, d_y32, d_commitments, d_proofs, n, 0, nullptr, s, &f);
    if (rc) return rc;
    return verify_finish(c, f);
}


// upload helper: gather n host items of `bytes` each into the context staging area and copy to `dst`
static int upload_items(mkzg_ctx* c, const uint8_t* const* items, size_t n, size_t bytes, uint8_t* d_dst, hipStream_t s) {
    for (size_t i = 0; i < n; i++) memcpy(c->h_stage + i * bytes, items[i], bytes);
    HIPCHK(hipMemcpyAsync(d_dst, c->h_stage, n * bytes, hipMemcpyHostToDevice, s));
    return MKZG_OK;
}

extern "C" {

// Context.VerifyBlobKZGProofBatch (reference verify.go:88-143), device-resident inputs.
int mkzg_verify_blob_kzg_proof_batch_device(mkzg_ctx* c, const void* d_blobs, const void* d_commitments48, const void* d_proofs48, size_t n, void* stream) {
    if (!c || (n && (!d_blobs || !d_commitments48 || !d_proofs48))) return MKZG_ERR_BAD_ARGS;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    return verify_core(c, (const uint8_t*)d_blobs, nullptr, nullptr, (const uint8_t*)d_commitments48, (const uint8_t*)d_proofs48, n, (hipStream_t)stream);
}

int mkzg_verify_blob_kzg_proof_batch(mkzg_ctx* c, const uint8_t* const* blobs, const uint8_t* commitments48n, const uint8_t* proofs48n, size_t n_blobs, size_t n_comm,
                                     size_t n_proofs) {
    if (!c) return MKZG_ERR_BAD_ARGS;
    if (n_blobs != n_comm || n_blobs != n_proofs) return MKZG_ERR_BATCH_LENGTH;      // verify.go:91-95
    const size_t n = n_blobs;
    if (n == 0) return MKZG_OK;
    if (!blobs || !commitments48n || !proofs48n) return MKZG_ERR_BAD_ARGS;
    if (n > (size_t)VMSM_MAX_N) return MKZG_ERR_BAD_ARGS;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    // the reference deserialises commitment i and proof i before blob i, so a nil blob only
    // surfaces after their checks; keep that precedence by deferring the nil error
    size_t first_nil = n;
    for (size_t i = 0; i < n; i++) if (!blobs[i]) { first_nil = i; break; }
    int rc = ensure_staging(c, n, MKZG_BYTES_PER_BLOB);
    if (rc) return rc;
    rc = ensure_verify_ws(c, n);
    if (rc) return rc;
    std::vector<uint8_t> zero_blob;
    std::vector<const uint8_t*> ptrs(blobs, blobs + n);
    if (first_nil < n) { zero_blob.assign(MKZG_BYTES_PER_BLOB, 0); for (size_t i = 0; i < n; i++) if (!ptrs[i]) ptrs[i] = zero_blob.data(); }
    rc = upload_items(c, ptrs.data(), n, MKZG_BYTES_PER_BLOB, c->d_blobs, c->stream);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(c->d_v_raw, commitments48n, n * 48, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_v_raw + c->v_cap * 48, proofs48n, n * 48, hipMemcpyHostToDevice, c->stream));
    rc = verify_core(c, c->d_blobs, nullptr, nullptr, c->d_v_raw, c->d_v_raw + c->v_cap * 48, n, c->stream);
    if (first_nil < n) {
        // an error at an index below the nil blob (or at its commitment/proof) wins; otherwise nil
        int* h_st = (int*)c->h_v;
        const size_t cap = c->v_cap;
        for (size_t i = 0; i <= first_nil; i++) {
            if (h_st[i]) return h_st[i];
            if (h_st[cap + i]) return h_st[cap + i];
            if (i < first_nil && h_st[2 * cap + i]) return h_st[2 * cap + i];
        }
        return MKZG_ERR_NIL_INPUT;
    }
    return rc;
}

// Context.VerifyBlobKZGProof (reference verify.go:48-83): order blob, commitment, proof
int mkzg_verify_blob_kzg_proof(mkzg_ctx* c, const uint8_t* blob, const uint8_t commitment48[48], const uint8_t proof48[48]) {
    if (!c || !commitment48 || !proof48) return MKZG_ERR_BAD_ARGS;
    if (!blob) return MKZG_ERR_NIL_INPUT;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    const uint8_t* one[1] = {blob};
    int rc = mkzg_verify_blob_kzg_proof_batch(c, one, commitment48, proof48, 1, 1, 1);
    if (rc > MKZG_VERIFY_FAILED) {
        // single-proof variant deserialises the blob first (verify.go:51-54)
        int* h_st = (int*)c->h_v;
        if (h_st[2 * c->v_cap]) return h_st[2 * c->v_cap];
    }
    return rc;
}

// Context.VerifyKZGProof (reference verify.go:12-43)
int mkzg_verify_kzg_proof(mkzg_ctx* c, const uint8_t commitment48[48], const uint8_t z32[32], const uint8_t y32[32], const uint8_t proof48[48]) {
    if (!c || !commitment48 || !z32 || !y32 || !proof48) return MKZG_ERR_BAD_ARGS;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    int rc = ensure_verify_ws(c, 1);
    if (rc) return rc;
    rc = ensure_staging(c, 1, 256);
    if (rc) return rc;
    memcpy(c->h_stage, z32, 32);
    memcpy(c->h_stage + 32, y32, 32);
    HIPCHK(hipMemcpyAsync(c->d_blobs, c->h_stage, 64, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_v_raw, commitment48, 48, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_v_raw + c->v_cap * 48, proof48, 48, hipMemcpyHostToDevice, c->stream));
    return verify_core(c, nullptr, c->d_blobs, c->d_blobs + 32, c->d_v_raw, c->d_v_raw + c->v_cap * 48, 1, c->stream);
}

// ---- proofs -------------------------------------------------------------------------------
// Context.ComputeBlobKZGProof for a device-resident batch (reference prove.go:46-77 per item):
// status[i] receives the per-item code; proofs of failed items are undefined.
int mkzg_compute_blob_kzg_proof_batch_device(mkzg_ctx* c, const void* d_blobs, const void* d_commitments48, size_t n, void* d_proofs48, int* d_status, void* stream) {
    if (!c || (n && (!d_blobs || !d_commitments48 || !d_proofs48 || !d_status))) return MKZG_ERR_BAD_ARGS;
    if (n == 0) return MKZG_OK;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    // Slices of up to 4096 proofs.  Per slice the latency-bound stages run ONCE over the whole slice -- Fiat-Shamir challenges
    // (a 2050-compression chain per blob whatever the count) on the caller's stream, commitment decoding + subgroup checks on the
    // side stream -- and only the quotient + MSM stage is cut into max_batch pieces (its workspace is 196 KB per blob).
    size_t done = 0;
    while (done < n) {
        const size_t slice = std::min<size_t>(n - done, 4096);
        int rc = ensure_quot_ws(c, std::max<size_t>((size_t)c->opt_max_batch, slice), (size_t)c->opt_max_batch);
        if (rc) return rc;
        rc = ensure_verify_ws(c, slice);
        if (rc) return rc;
        const uint8_t* blobs = (const uint8_t*)d_blobs + done * MKZG_BYTES_PER_BLOB;
        const uint8_t* cms = (const uint8_t*)d_commitments48 + done * 48;
        int* st_blob = c->d_q_st;
        int* st_cm = c->d_q_st + c->q_cap;
        // commitment must be a valid subgroup point (prove.go:56-62)
        HIPCHK(hipEventRecord(c->ev_fork, s));
        HIPCHK(hipStreamWaitEvent(c->side, c->ev_fork, 0));
        g1_validate2_device(cms, (int)slice, c->d_v_aff, st_cm, nullptr, 0, nullptr, nullptr, c->side);
        HIPCHK(hipEventRecord(c->ev_join, c->side));
        KT(KT_SHA256, s, challenge_device(blobs, cms, (int)slice, c->d_q_z, s));
        for (size_t off = 0; off < slice; off += (size_t)c->opt_max_batch) {
            const size_t chunk = std::min<size_t>(slice - off, (size_t)c->opt_max_batch);
            KT(KT_QUOTIENT, s, blob_quotient_device(blobs + off * MKZG_BYTES_PER_BLOB, c->d_q_z + off * 12, c->d_roots_brp, c->d_inv_roots_brp, (int)chunk, c->d_q_y + off * 12,
                                                    c->d_quot, st_blob + off, s));
            rc = commit_scalars_device(c, (const uint8_t*)c->d_quot, (size_t)4096 * 48, 1, chunk, (uint8_t*)d_proofs48 + (done + off) * 48, d_status + done + off, s);
            if (rc) return rc;
        }
        HIPCHK(hipStreamWaitEvent(s, c->ev_join, 0));
        // merge statuses: blob error first, then commitment error (prove.go:49-62)
        HIPCHK(hipGetLastError());
        std::vector<int> stb(slice), stc(slice);
        HIPCHK(hipMemcpyAsync(stb.data(), st_blob, slice * sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(stc.data(), st_cm, slice * sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        bool any = false;
        for (size_t i = 0; i < slice; i++) { stb[i] = stb[i] ? stb[i] : stc[i]; any = any || stb[i]; }
        if (any) HIPCHK(hipMemcpyAsync(d_status + done, stb.data(), slice * sizeof(int), hipMemcpyHostToDevice, s));
        if (any) HIPCHK(hipStreamSynchronize(s));
        done += slice;
    }
    return MKZG_OK;
}

int mkzg_compute_blob_kzg_proof(mkzg_ctx* c, const uint8_t* blob, const uint8_t commitment48[48], int num_threads, uint8_t out48[48]) {
    if (!c || !commitment48 || !out48) return MKZG_ERR_BAD_ARGS;
    if (!blob) return MKZG_ERR_NIL_INPUT;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    int rc = ensure_staging(c, 1, MKZG_BYTES_PER_BLOB);
    if (rc) return rc;
    rc = ensure_verify_ws(c, 1);
    if (rc) return rc;
    memcpy(c->h_stage, blob, MKZG_BYTES_PER_BLOB);
    HIPCHK(hipMemcpyAsync(c->d_blobs, c->h_stage, MKZG_BYTES_PER_BLOB, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_v_raw, commitment48, 48, hipMemcpyHostToDevice, c->stream));
    rc = mkzg_compute_blob_kzg_proof_batch_device(c, c->d_blobs, c->d_v_raw, 1, c->d_out, c->d_status, c->stream);
    if (rc) return rc;
    int st = 0;
    HIPCHK(hipMemcpyAsync(out48, c->d_out, 48, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(&st, c->d_status, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (st) return st;
    if (num_threads >= 1024) return MKZG_ERR_TOO_MANY_GOROUTINES;
    return MKZG_OK;
}

// Context.ComputeKZGProof (reference prove.go:85-111): proof and claimed value at a caller-chosen z
int mkzg_compute_kzg_proof(mkzg_ctx* c, const uint8_t* blob, const uint8_t z32[32], int num_threads, uint8_t proof48[48], uint8_t y32[32]) {
    if (!c || !z32 || !proof48 || !y32) return MKZG_ERR_BAD_ARGS;
    if (!blob) return MKZG_ERR_NIL_INPUT;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    int rc = ensure_staging(c, 1, MKZG_BYTES_PER_BLOB + 128);
    if (rc) return rc;
    rc = ensure_quot_ws(c, (size_t)c->opt_max_batch, (size_t)c->opt_max_batch);
    if (rc) return rc;
    hipStream_t s = c->stream;
    memcpy(c->h_stage, blob, MKZG_BYTES_PER_BLOB);
    memcpy(c->h_stage + MKZG_BYTES_PER_BLOB, z32, 32);
    HIPCHK(hipMemcpyAsync(c->d_blobs, c->h_stage, MKZG_BYTES_PER_BLOB + 32, hipMemcpyHostToDevice, s));
    int* st_z = c->d_q_st + c->q_cap;
    scalars_to_fr_device(c->d_blobs + MKZG_BYTES_PER_BLOB, 1, c->d_q_z, st_z, s);
    blob_quotient_device(c->d_blobs, c->d_q_z, c->d_roots_brp, c->d_inv_roots_brp, 1, c->d_q_y, c->d_quot, c->d_q_st, s);
    rc = commit_scalars_device(c, (const uint8_t*)c->d_quot, (size_t)4096 * 48, 1, 1, c->d_out, c->d_status, s);
    if (rc) return rc;
    fr_to_scalars_device(c->d_q_y, 1, c->d_blobs + MKZG_BYTES_PER_BLOB + 64, s);
    HIPCHK(hipGetLastError());
    int stb = 0, stz = 0;
    HIPCHK(hipMemcpyAsync(&stb, c->d_q_st, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&stz, st_z, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(proof48, c->d_out, 48, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(y32, c->d_blobs + MKZG_BYTES_PER_BLOB + 64, 32, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (stb) return stb;                                           // blob first (prove.go:88-91)
    if (stz) return stz;
    if (num_threads >= 1024) return MKZG_ERR_TOO_MANY_GOROUTINES;
    return MKZG_OK;
}

// The finer seam of the reference, multiexp.MultiExpG1 (internal/multiexp/multiexp.go:20-26): sum s_i P_i with gnark-crypto's in-memory
// layouts on both sides -- scalars n x 4 u64 (fr.Element, Montgomery R = 2^256), points n x 12 u64 (G1Affine: X, Y as fp.Element, Montgomery
// R = 2^384, identity (0,0)), result one G1Affine.  Like the reference it does not validate the points.  Any n (pieces of 8192).
int mkzg_msm_g1(mkzg_ctx* c, const uint64_t* scalars_mont4, const uint64_t* points_mont12, size_t n, uint64_t out_mont12[12]) {
    if (!c || !out_mont12 || (n && (!scalars_mont4 || !points_mont12))) return MKZG_ERR_BAD_ARGS;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    G1HX total;
    G1H::set_inf(total);
    for (size_t done = 0; done < n; done += VMSM_MAX_N) {
        const size_t m = std::min<size_t>(VMSM_MAX_N, n - done);
        int rc = ensure_verify_ws(c, m);
        if (rc) return rc;
        // d_v_raw holds 2 x cap x 48 B = cap x 96 B: exactly the points; d_v_s3 takes the raw scalars, d_v_s1 the converted ones
        HIPCHK(hipMemcpyAsync(c->d_v_raw, points_mont12 + done * 12, m * 96, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(c->d_v_s3, scalars_mont4 + done * 4, m * 32, hipMemcpyHostToDevice, s));
        gnark_to_vmsm_device(c->d_v_s3, (const uint32_t*)c->d_v_raw, (int)m, c->d_v_s1, c->d_v_aff, s);
        const uint32_t* scal[1] = {c->d_v_s1};
        const uint32_t* pts[1] = {c->d_v_aff};
        const int nj[1] = {(int)m};
        if (vmsm_windows_device(scal, pts, nj, 1, c->d_v_misc + V_MISC_WORDS, c->d_v_misc + 16, s)) return MKZG_ERR_BAD_ARGS;
        HIPCHK(hipGetLastError());
        uint32_t* h_misc = (uint32_t*)(c->h_v + 3 * c->v_cap * sizeof(int));
        HIPCHK(hipMemcpyAsync(h_misc, c->d_v_misc, (16 + VMSM_WINDOWS * 48) * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        G1HX part;
        g1h_horner_words(part, h_misc + 16, VMSM_WINDOWS, VMSM_WINDOW_BITS);
        G1H::add(total, total, part);
    }
    G1AffH a;
    g1h_to_affine(a, total);
    if (a.inf) {
        memset(out_mont12, 0, 96);
    } else {
        memcpy(out_mont12, a.x.l, 48);             // HostFp is 6 x u64 Montgomery with R = 2^384: gnark's fp.Element as is
        memcpy(out_mont12 + 6, a.y.l, 48);
    }
    return MKZG_OK;
}

// ---- sharded batch verification (multi-GPU): each rank folds its shard with the GLOBAL exponents
// r^(offset + i) and publishes one 512-byte record; any rank adds the records and runs the pairing.
// Record: [0..4) status (LE int) | [16..64) A | [64..112) Cc | [112..160) D (compressed G1) | [160..192) e (32 B big-endian) | zero padding
int mkzg_verify_blob_kzg_proof_batch_partial_device(mkzg_ctx* c, const void* d_blobs, const void* d_commitments48, const void* d_proofs48, size_t n_local, uint64_t offset,
                                                    const uint8_t r32[32], uint8_t record512[512], void* stream) {
    if (!c || !r32 || !record512 || (n_local && (!d_blobs || !d_commitments48 || !d_proofs48))) return MKZG_ERR_BAD_ARGS;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    memset(record512, 0, 512);
    VerifyFold f;
    G1H::set_inf(f.A);
    G1H::set_inf(f.Cc);
    G1H::set_inf(f.D);
    memset(f.e, 0, 32);
    int rc = MKZG_OK;
    if (n_local) rc = verify_fold(c, (const uint8_t*)d_blobs, nullptr, nullptr, (const uint8_t*)d_commitments48, (const uint8_t*)d_proofs48, n_local, (uint32_t)offset, r32,
                                  (hipStream_t)stream, &f);
    int32_t st = rc;
    memcpy(record512, &st, 4);
    if (rc == MKZG_OK) {
        G1AffH a;
        g1h_to_affine(a, f.A);
        g1h_compress(record512 + 16, a);
        g1h_to_affine(a, f.Cc);
        g1h_compress(record512 + 64, a);
        g1h_to_affine(a, f.D);
        g1h_compress(record512 + 112, a);
        for (int i = 0; i < 8; i++) {
            uint32_t w = f.e[7 - i];
            record512[160 + 4 * i] = (uint8_t)(w >> 24); record512[161 + 4 * i] = (uint8_t)(w >> 16); record512[162 + 4 * i] = (uint8_t)(w >> 8); record512[163 + 4 * i] = (uint8_t)w;
        }
    }
    return rc;
}
int mkzg_verify_blob_kzg_proof_batch_finish(mkzg_ctx* c, const uint8_t* records512, size_t n_records) {
    if (!c || (n_records && !records512)) return MKZG_ERR_BAD_ARGS;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    VerifyFold f;
    G1H::set_inf(f.A);
    G1H::set_inf(f.Cc);
    G1H::set_inf(f.D);
    memset(f.e, 0, 32);
    for (size_t k = 0; k < n_records; k++) {                       // first failing rank (lowest indices) decides the error
        int32_t st;
        memcpy(&st, records512 + 512 * k, 4);
        if (st) return st;
    }
    for (size_t k = 0; k < n_records; k++) {
        const uint8_t* rec = records512 + 512 * k;
        G1AffH a;
        G1HX t;
        G1HX* dst[3] = {&f.A, &f.Cc, &f.D};
        for (int q = 0; q < 3; q++) {
            if (!g1h_decompress(a, rec + 16 + 48 * q)) return MKZG_ERR_BAD_ARGS;
            if (a.inf) continue;
            G1H::from_affine(t, a.x, a.y);
            G1H::add(*dst[q], *dst[q], t);
        }
        uint32_t ew[8];
        for (int i = 0; i < 8; i++) ew[7 - i] = ((uint32_t)rec[160 + 4 * i] << 24) | ((uint32_t)rec[161 + 4 * i] << 16) | ((uint32_t)rec[162 + 4 * i] << 8) | rec[163 + 4 * i];
        fr_words_add(f.e, f.e, ew);
    }
    c->last_host_finish_us = 0;
    return verify_finish(c, f);
}

// host-side share of the last verification (microseconds): Horner + [e]G + pairing
double mkzg_last_host_finish_us(mkzg_ctx* c) { return c ? c->last_host_finish_us : 0.0; }

}  // extern "C"
