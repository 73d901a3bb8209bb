/*
 * oracle/kzg_oracle.c -- CPU restatement of go-eth-kzg's Context API (EIP-4844 + EIP-7594).
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the MI355X product in
 * go-eth-kzg_amd/: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  The product never links, imports or executes anything from oracle/.
 *
 * Parity status: PINNED.  Every function below is checked bit-for-bit against the
 * reference's own consensus-spec vectors (tests/golden/consensus_vectors.json, converted
 * from /root/reference/tests/ by tests/golden/make_consensus_fixtures.py) and the
 * reference's unit-test KATs -- see tests/test_oracle_vectors.py, tests/test_oracle_kats.py.
 * The reference itself cannot be built here (no Go toolchain; gnark-crypto v0.16.0 is not
 * vendored), so there is no oracle/_ref.
 *
 * Each function cites the reference file:line it follows (paths relative to /root/reference).
 */
#include <stdint.h>
#include <stdlib.h>
#include <omp.h>
#include <string.h>
#include <stdio.h>
#include "field.h"
#include "curve.h"
#include "pairing.h"
#include "sha256.h"

#define KZGO_EXPORT __attribute__((visibility("default")))

/* return codes: shared numbering with include/mi355kzg.h */
enum {
    KZG_OK = 0,
    KZG_VERIFY_FAILED = 1,             /* kzg.ErrVerifyOpeningProof  internal/kzg/errors.go:8 */
    KZG_ERR_BATCH_LENGTH = 2,          /* ErrBatchLengthCheck        errors.go:6 */
    KZG_ERR_NON_CANONICAL_SCALAR = 3,  /* ErrNonCanonicalScalar      errors.go:7 */
    KZG_ERR_INVALID_CELL_ID = 4,       /* ErrInvalidCellID           errors.go:8 */
    KZG_ERR_INVALID_ROW_INDEX = 5,     /* ErrInvalidRowIndex         errors.go:9 */
    KZG_ERR_NIL_INPUT = 6,             /* ErrDeserializeNilInput     errors.go:10 */
    KZG_ERR_NUM_CELL_IDS = 7,          /* ErrNumCellIDsNotEqualNumCells errors.go:12 */
    KZG_ERR_CELL_IDS_NOT_ORDERED = 8,  /* ErrCellIDsNotOrdered       errors.go:13 */
    KZG_ERR_FOUND_INVALID_CELL_ID = 9, /* ErrFoundInvalidCellID      errors.go:14 */
    KZG_ERR_NOT_ENOUGH_CELLS = 10,     /* ErrNotEnoughCellsForReconstruction errors.go:15 */
    KZG_ERR_G1_BAD_ENCODING = 11,      /* gnark SetBytes decode errors */
    KZG_ERR_G1_NOT_ON_CURVE = 12,
    KZG_ERR_G1_NOT_IN_SUBGROUP = 13,
    KZG_ERR_TOO_MANY_GOROUTINES = 14,  /* multiexp.ErrTooManyGoRoutines internal/multiexp/errors.go:5 */
    KZG_ERR_INVALID_POLY_SIZE = 15,
    KZG_ERR_BAD_TRUSTED_SETUP = 16,
    KZG_ERR_BAD_ARGS = 17,
};

#define N_BLOB 4096        /* ScalarsPerBlob      serialization.go:33 */
#define N_EXT 8192         /* scalarsPerExtBlob   serialization.go:54 */
#define N_CELLS 128        /* CellsPerExtBlob     serialization.go:39 */
#define CELL_SZ 64         /* scalarsPerCell      serialization.go:42 */
#define BLOB_BYTES 131072
#define CELL_BYTES 2048

/* ------------------------------------------------------------------ domain */
/* reference internal/domain/domain.go:127-149 */
static uint64_t bitrev(uint64_t k, uint64_t n) {
    int lg = __builtin_ctzll(n); uint64_t r = 0;
    for (int i = 0; i < lg; i++) if (k & (1ull << i)) r |= 1ull << (lg - 1 - i);
    return r;
}
static void bitrev_fr(fr_t *v, uint64_t n) { for (uint64_t i = 0; i < n; i++) { uint64_t j = bitrev(i, n); if (j > i) { fr_t t = v[i]; v[i] = v[j]; v[j] = t; } } }
static void bitrev_g1(g1a_t *v, uint64_t n) { for (uint64_t i = 0; i < n; i++) { uint64_t j = bitrev(i, n); if (j > i) { g1a_t t = v[i]; v[i] = v[j]; v[j] = t; } } }

typedef struct {
    uint64_t n; fr_t n_inv, gen, gen_inv;
    fr_t *roots, *inv_roots; /* natural order unless reversed */
} domain_t;

/* reference internal/domain/domain.go:51-98 */
static void domain_init(domain_t *d, uint64_t n) {
    fr_t root; fr_from_plain(&root, FR_ROOT_2_32);
    int lg = __builtin_ctzll(n);
    d->n = n;
    fr_pow_u64(&d->gen, &root, 1ull << (32 - lg));
    fr_inv(&d->gen_inv, &d->gen);
    fr_set_u64(&d->n_inv, n); fr_inv(&d->n_inv, &d->n_inv);
    d->roots = malloc(n * sizeof(fr_t)); d->inv_roots = malloc(n * sizeof(fr_t));
    fr_t cur; fr_one(&cur);
    for (uint64_t i = 0; i < n; i++) { d->roots[i] = cur; fr_mul(&cur, &cur, &d->gen); }
    fr_batch_inv(d->inv_roots, d->roots, n);
}
static void domain_free(domain_t *d) { free(d->roots); free(d->inv_roots); }
/* reference internal/domain/domain.go:154-157 */
static void domain_reverse_roots(domain_t *d) { bitrev_fr(d->roots, d->n); bitrev_fr(d->inv_roots, d->n); }

/* reference internal/domain/fft.go:109-144: in-place radix-2 DIF (Gentleman-Sande) + bit-reverse */
static void fft_fr_inplace(fr_t *v, uint64_t n, const fr_t *root) {
    if (n <= 1) return;
    for (uint64_t size = n; size >= 2; size >>= 1) {
        uint64_t half = size >> 1;
        fr_t wstep = *root;
        for (uint64_t i = n / size; i > 1; i--) fr_mul(&wstep, &wstep, root);
        for (uint64_t start = 0; start < n; start += size) {
            fr_t w; fr_one(&w);
            for (uint64_t k = 0; k < half; k++) {
                fr_t *a = &v[start + k], *b = &v[start + k + half], t;
                fr_sub(&t, a, b); fr_add(a, a, b); fr_mul(b, &t, &w);
                fr_mul(&w, &w, &wstep);
            }
        }
    }
    bitrev_fr(v, n);
}
/* reference internal/domain/fft.go:95-107 */
static void domain_fft(const domain_t *d, fr_t *v) { fft_fr_inplace(v, d->n, &d->gen); }
static void domain_ifft(const domain_t *d, fr_t *v) {
    fft_fr_inplace(v, d->n, &d->gen_inv);
    for (uint64_t i = 0; i < d->n; i++) fr_mul(&v[i], &v[i], &d->n_inv);
}
/* reference internal/domain/coset_fft.go:41-70 */
static void coset_fft(const domain_t *d, const fr_t *coset_gen, fr_t *v) {
    fr_t s; fr_one(&s);
    for (uint64_t i = 0; i < d->n; i++) { fr_mul(&v[i], &v[i], &s); fr_mul(&s, &s, coset_gen); }
    fft_fr_inplace(v, d->n, &d->gen);
}
static void coset_ifft(const domain_t *d, const fr_t *inv_coset_gen, fr_t *v) {
    fft_fr_inplace(v, d->n, &d->gen_inv);
    for (uint64_t i = 0; i < d->n; i++) fr_mul(&v[i], &v[i], &d->n_inv);
    fr_t s; fr_one(&s);
    for (uint64_t i = 0; i < d->n; i++) { fr_mul(&v[i], &v[i], &s); fr_mul(&s, &s, inv_coset_gen); }
}

/* reference internal/domain/fft.go:49-92: recursive radix-2 FFT over G1 (natural in, natural out) */
static void fft_g1_rec(g1j_t *out, const g1j_t *in, uint64_t n, uint64_t stride, const fr_t *root) {
    if (n == 1) { out[0] = in[0]; return; }
    fr_t root2; fr_sqr(&root2, root);
    uint64_t h = n / 2;
    g1j_t *even = malloc(h * sizeof(g1j_t)), *odd = malloc(h * sizeof(g1j_t));
    fft_g1_rec(even, in, h, stride * 2, &root2);
    fft_g1_rec(odd, in + stride, h, stride * 2, &root2);
    fr_t w, one; fr_one(&w); fr_one(&one);
    for (uint64_t k = 0; k < h; k++) {
        g1j_t t;
        if (fr_eq(&w, &one)) t = odd[k]; else g1j_mul_fr(&t, &odd[k], &w);
        g1j_add(&out[k], &even[k], &t);
        g1j_sub(&out[k + h], &even[k], &t);
        fr_mul(&w, &w, root);
    }
    free(even); free(odd);
}
/* reference internal/domain/fft.go:23-42 */
static void domain_fft_g1(const domain_t *d, g1j_t *v) {
    g1j_t *o = malloc(d->n * sizeof(g1j_t)); fft_g1_rec(o, v, d->n, 1, &d->gen); memcpy(v, o, d->n * sizeof(g1j_t)); free(o);
}
static void domain_ifft_g1(const domain_t *d, g1j_t *v) {
    g1j_t *o = malloc(d->n * sizeof(g1j_t)); fft_g1_rec(o, v, d->n, 1, &d->gen_inv);
    for (uint64_t i = 0; i < d->n; i++) g1j_mul_fr(&v[i], &o[i], &d->n_inv);
    free(o);
}

/* ------------------------------------------------------------------ MSM */
/* Stands in for gnark's G1Affine.MultiExp reached through reference
 * internal/multiexp/multiexp.go:20-26: sum_i s_i * P_i.  Bucket (Pippenger) method with
 * signed window digits; the result is independent of the window size. */
/* per-thread scratch that grows on demand and is reused by the next call (the CPU baseline commits thousands of blobs per thread) */
static void *tls_scratch(int slot, size_t bytes) {
    static __thread void *buf[2];
    static __thread size_t cap[2];
    if (cap[slot] < bytes) { free(buf[slot]); buf[slot] = malloc(bytes); cap[slot] = bytes; }
    return buf[slot];
}
static void msm_g1(g1j_t *out, const fr_t *scalars, const g1a_t *points, size_t n) {
    g1j_set_inf(out);
    if (n == 0) return;
    int c = n < 4 ? 1 : n < 32 ? 3 : n < 256 ? 5 : n < 2048 ? 8 : n < 16384 ? 10 : 12;
    int nwin = (256 + c - 1) / c + 1;
    size_t nb = (size_t)1 << (c - 1);
    int32_t *digits = tls_scratch(0, n * nwin * sizeof(int32_t));
    for (size_t i = 0; i < n; i++) {
        u64 v[5]; fr_to_plain(v, &scalars[i]); v[4] = 0;
        int carry = 0;
        for (int w = 0; w < nwin; w++) {
            int bit = w * c; int32_t d = 0;
            if (bit < 256) {
                u64 lo = v[bit >> 6] >> (bit & 63);
                if ((bit & 63) + c > 64) lo |= v[(bit >> 6) + 1] << (64 - (bit & 63));
                d = (int32_t)(lo & (((u64)1 << c) - 1));
            }
            d += carry; carry = 0;
            if (d > (int32_t)nb) { d -= (int32_t)(1 << c); carry = 1; }
            digits[i * nwin + w] = d;
        }
    }
    g1j_t *buckets = tls_scratch(1, nb * sizeof(g1j_t));
    for (int w = nwin - 1; w >= 0; w--) {
        for (int k = 0; k < c; k++) g1j_dbl(out, out);
        for (size_t b = 0; b < nb; b++) g1j_set_inf(&buckets[b]);
        for (size_t i = 0; i < n; i++) {
            int32_t d = digits[i * nwin + w];
            if (d > 0) g1j_add_affine(&buckets[d - 1], &buckets[d - 1], &points[i]);
            else if (d < 0) { g1a_t np; g1a_neg(&np, &points[i]); if (!g1a_is_inf(&points[i])) g1j_add_affine(&buckets[-d - 1], &buckets[-d - 1], &np); }
        }
        g1j_t run, sum; g1j_set_inf(&run); g1j_set_inf(&sum);
        for (size_t b = nb; b-- > 0;) { g1j_add(&run, &run, &buckets[b]); g1j_add(&sum, &sum, &run); }
        g1j_add(out, out, &sum);
    }
}

/* ------------------------------------------------------------------ context */
typedef struct kzgo_ctx {
    g1a_t *g1_monomial;       /* 4096 */
    g1a_t *g1_lagrange_brp;   /* 4096, bit-reversed (api.go:131) */
    g2a_t *g2; size_t n_g2;
    domain_t dom;             /* 4096, roots reversed (api.go:132) */
    domain_t dom_ext;         /* 8192, natural roots */
    domain_t dom128, dom64;
    g1a_t *fk20_table;        /* [128][64] transposed FFT'd SRS slices (toeplitz.go:50-93) */
    fr_t coset_shift[N_CELLS], coset_shift_inv[N_CELLS], coset_shift_pow[N_CELLS]; /* kzg_multi/srs.go:60-103 */
    fr_t rec_coset_gen, rec_coset_gen_inv; /* erasure_code.go:57-60 */
} kzgo_ctx;

/* reference internal/kzg_multi/fk20/fk20.go:23-52 + toeplitz.go:50-93 */
static void fk20_precompute(kzgo_ctx *c) {
    c->fk20_table = malloc(128 * 64 * sizeof(g1a_t));
    /* srs reversed, first 64 dropped, stride 64 -> 64 vectors of 63 points, padded to 128 */
    #pragma omp parallel for schedule(dynamic)
    for (int i = 0; i < 64; i++) {
        g1j_t vec[128];
        for (int j = 0; j < 128; j++) g1j_set_inf(&vec[j]);
        for (int j = 0; j < 63; j++) {
            int idx_trunc = i + 64 * j;           /* index into srsTruncated */
            int idx_rev = idx_trunc + 64;         /* index into reversed srs */
            g1j_from_affine(&vec[j], &c->g1_monomial[4095 - idx_rev]);
        }
        domain_fft_g1(&c->dom128, vec);
        for (int k = 0; k < 128; k++) g1j_to_affine(&c->fk20_table[k * 64 + i], &vec[k]);
    }
}

KZGO_EXPORT void kzgo_ctx_free(kzgo_ctx *c) {
    if (!c) return;
    free(c->g1_monomial); free(c->g1_lagrange_brp); free(c->g2); free(c->fk20_table);
    domain_free(&c->dom); domain_free(&c->dom_ext); domain_free(&c->dom128); domain_free(&c->dom64);
    free(c);
}

/* reference api.go:90-149 + trusted_setup.go:90-100 (points are decoded WITHOUT subgroup checks).
 * Inputs: raw compressed points as they appear (hex-decoded) in trusted_setup.json. */
KZGO_EXPORT int kzgo_ctx_new(const uint8_t *g1_monomial48, const uint8_t *g1_lagrange48, const uint8_t *g2_monomial96,
                             size_t n_g2, int with_fk20, kzgo_ctx **out) {
    if (n_g2 < 65) return KZG_ERR_BAD_TRUSTED_SETUP;
    kzgo_ctx *c = calloc(1, sizeof *c);
    c->g1_monomial = malloc(N_BLOB * sizeof(g1a_t)); c->g1_lagrange_brp = malloc(N_BLOB * sizeof(g1a_t));
    c->g2 = malloc(n_g2 * sizeof(g2a_t)); c->n_g2 = n_g2;
    int bad = 0;
    #pragma omp parallel for reduction(|:bad)
    for (int i = 0; i < N_BLOB; i++) {
        bad |= g1a_decompress(&c->g1_monomial[i], g1_monomial48 + 48 * i, 0) != 0;
        bad |= g1a_decompress(&c->g1_lagrange_brp[i], g1_lagrange48 + 48 * i, 0) != 0;
    }
    for (size_t i = 0; i < n_g2; i++) bad |= g2a_decompress(&c->g2[i], g2_monomial96 + 96 * i) != 0;
    domain_init(&c->dom, N_BLOB); domain_init(&c->dom_ext, N_EXT); domain_init(&c->dom128, 128); domain_init(&c->dom64, 64);
    if (bad) { kzgo_ctx_free(c); return KZG_ERR_BAD_TRUSTED_SETUP; }
    bitrev_g1(c->g1_lagrange_brp, N_BLOB);      /* api.go:131 */
    domain_reverse_roots(&c->dom);              /* api.go:132 */
    /* kzg_multi/srs.go:60-92: coset shifts h_k = bitrev(roots_8192)[64 k] */
    for (int k = 0; k < N_CELLS; k++) {
        c->coset_shift[k] = c->dom_ext.roots[bitrev(k * CELL_SZ, N_EXT)];
        fr_inv(&c->coset_shift_inv[k], &c->coset_shift[k]);
        fr_pow_u64(&c->coset_shift_pow[k], &c->coset_shift[k], CELL_SZ);
    }
    fr_set_u64(&c->rec_coset_gen, 7); fr_inv(&c->rec_coset_gen_inv, &c->rec_coset_gen);
    if (with_fk20) fk20_precompute(c);
    *out = c;
    return KZG_OK;
}

/* ------------------------------------------------------------------ (de)serialisation */
/* reference serialization.go:134-146 */
static int deserialize_blob(fr_t *poly, const uint8_t *blob) {
    if (!blob) return KZG_ERR_NIL_INPUT;
    for (int i = 0; i < N_BLOB; i++) if (!fr_from_be32_canonical(&poly[i], blob + 32 * i)) return KZG_ERR_NON_CANONICAL_SCALAR;
    return KZG_OK;
}
/* reference serialization.go:196-216 */
static int deserialize_cell(fr_t *ev, const uint8_t *cell) {
    if (!cell) return KZG_ERR_NIL_INPUT;
    for (int i = 0; i < CELL_SZ; i++) if (!fr_from_be32_canonical(&ev[i], cell + 32 * i)) return KZG_ERR_NON_CANONICAL_SCALAR;
    return KZG_OK;
}
/* reference serialization.go:108-115 (gnark SetBytes: flags, x < p, on curve, subgroup) */
static int deserialize_g1(g1a_t *p, const uint8_t *b) {
    switch (g1a_decompress(p, b, 1)) {
        case G1DEC_OK: return KZG_OK;
        case G1DEC_BAD_ENCODING: return KZG_ERR_G1_BAD_ENCODING;
        case G1DEC_NOT_ON_CURVE: return KZG_ERR_G1_NOT_ON_CURVE;
        default: return KZG_ERR_G1_NOT_IN_SUBGROUP;
    }
}
static void serialize_g1j(uint8_t out[48], const g1j_t *p) { g1a_t a; g1j_to_affine(&a, p); g1a_compress(out, &a); }

/* reference fiatshamir.go:22-33 */
static void compute_challenge(fr_t *z, const uint8_t *blob, const uint8_t *commitment) {
    sha256_ctx h; uint8_t d[32], deg[16] = {0};
    deg[14] = (N_BLOB >> 8) & 0xff; deg[15] = N_BLOB & 0xff;
    sha256_init(&h);
    sha256_update(&h, (const uint8_t *)"FSBLOBVERIFY_V1_", 16);
    sha256_update(&h, deg, 16);
    sha256_update(&h, blob, BLOB_BYTES);
    sha256_update(&h, commitment, 48);
    sha256_final(&h, d);
    fr_from_be32_reduce(z, d);
}

/* reference internal/domain/domain.go:163-235 */
static int64_t find_root_index(const domain_t *d, const fr_t *z) {
    for (uint64_t i = 0; i < d->n; i++) if (fr_eq(z, &d->roots[i])) return (int64_t)i;
    return -1;
}
static void eval_lagrange(fr_t *y, int64_t *index, const domain_t *d, const fr_t *poly, const fr_t *z) {
    *index = find_root_index(d, z);
    if (*index >= 0) { *y = poly[*index]; return; }
    fr_t *den = malloc(d->n * sizeof(fr_t)), *inv = malloc(d->n * sizeof(fr_t));
    for (uint64_t i = 0; i < d->n; i++) fr_sub(&den[i], z, &d->roots[i]);
    fr_batch_inv(inv, den, d->n);
    fr_t res, t; fr_zero(&res);
    for (uint64_t i = 0; i < d->n; i++) { fr_mul(&t, &poly[i], &d->roots[i]); fr_mul(&t, &t, &inv[i]); fr_add(&res, &res, &t); }
    fr_t one; fr_one(&one);
    fr_pow_u64(&t, z, d->n); fr_sub(&t, &t, &one); fr_mul(&t, &t, &d->n_inv); fr_mul(y, &t, &res);
    free(den); free(inv);
}

/* ------------------------------------------------------------------ EIP-4844 */
/* reference prove.go:13-33 -> internal/kzg/srs.go:56-62 -> internal/multiexp/multiexp.go:20 */
KZGO_EXPORT int kzgo_blob_to_kzg_commitment(const kzgo_ctx *c, const uint8_t *blob, uint8_t out[48]) {
    fr_t *poly = malloc(N_BLOB * sizeof(fr_t));
    int rc = deserialize_blob(poly, blob);
    if (rc) { free(poly); return rc; }
    g1j_t r; msm_g1(&r, poly, c->g1_lagrange_brp, N_BLOB);
    serialize_g1j(out, &r);
    free(poly);
    return KZG_OK;
}
/* batch helper for the CPU baseline: independent blobs across `threads` OpenMP threads (0: the OpenMP default).  The caller picks
 * the count from the CPUs it may really use -- a cgroup quota of 16 CPUs under 256 visible ones makes the OpenMP default thrash. */
KZGO_EXPORT int kzgo_blob_to_kzg_commitment_batch_threads(const kzgo_ctx *c, const uint8_t *blobs, size_t n, uint8_t *out48, int *status, int threads) {
    int any = 0;
    if (threads <= 0) threads = omp_get_max_threads();
    #pragma omp parallel for schedule(dynamic) reduction(|:any) num_threads(threads)
    for (size_t i = 0; i < n; i++) {
        int rc = kzgo_blob_to_kzg_commitment(c, blobs + i * BLOB_BYTES, out48 + 48 * i);
        if (status) status[i] = rc;
        any |= rc;
    }
    return any ? KZG_ERR_BAD_ARGS : KZG_OK;
}
KZGO_EXPORT int kzgo_blob_to_kzg_commitment_batch(const kzgo_ctx *c, const uint8_t *blobs, size_t n, uint8_t *out48, int *status) {
    return kzgo_blob_to_kzg_commitment_batch_threads(c, blobs, n, out48, status, 0);
}
KZGO_EXPORT int kzgo_omp_max_threads(void) { return omp_get_max_threads(); }

/* reference internal/kzg/kzg_prove.go:14-180 */
static void compute_quotient(fr_t *q, const domain_t *d, const fr_t *f, int64_t index, const fr_t *fz, const fr_t *z) {
    uint64_t n = d->n;
    fr_t *den = malloc(n * sizeof(fr_t)), *inv = malloc(n * sizeof(fr_t));
    if (index < 0) {                                 /* kzg_prove.go:81-112 */
        for (uint64_t i = 0; i < n; i++) fr_sub(&den[i], &d->roots[i], z);
        fr_batch_inv(inv, den, n);
        for (uint64_t i = 0; i < n; i++) { fr_t t; fr_sub(&t, &f[i], fz); fr_mul(&q[i], &inv[i], &t); }
    } else {                                         /* kzg_prove.go:118-180 */
        fr_t fzz = f[index], zz = d->roots[index], invz = d->inv_roots[index];
        for (uint64_t i = 0; i < n; i++) fr_sub(&den[i], &d->roots[i], &zz);
        fr_one(&den[index]);
        fr_batch_inv(inv, den, n);
        fr_zero(&q[index]);
        for (uint64_t j = 0; j < n; j++) {
            if ((int64_t)j == index) continue;
            fr_t qj, qmj;
            fr_sub(&qj, &f[j], &fzz); fr_mul(&qj, &qj, &inv[j]); q[j] = qj;
            fr_neg(&qmj, &qj); fr_mul(&qmj, &qmj, &d->roots[j]); fr_mul(&qmj, &qmj, &invz);
            fr_add(&q[index], &q[index], &qmj);
        }
    }
    free(den); free(inv);
}
static void open_at(const kzgo_ctx *c, const fr_t *poly, const fr_t *z, g1j_t *proof, fr_t *y) {
    int64_t idx; eval_lagrange(y, &idx, &c->dom, poly, z);
    fr_t *q = malloc(N_BLOB * sizeof(fr_t));
    compute_quotient(q, &c->dom, poly, idx, y, z);
    msm_g1(proof, q, c->g1_lagrange_brp, N_BLOB);
    free(q);
}
/* reference prove.go:85-111 */
KZGO_EXPORT int kzgo_compute_kzg_proof(const kzgo_ctx *c, const uint8_t *blob, const uint8_t z32[32], uint8_t proof48[48], uint8_t y32[32]) {
    fr_t *poly = malloc(N_BLOB * sizeof(fr_t)); fr_t z, y; g1j_t pr;
    int rc = deserialize_blob(poly, blob);
    if (!rc && !fr_from_be32_canonical(&z, z32)) rc = KZG_ERR_NON_CANONICAL_SCALAR;
    if (!rc) { open_at(c, poly, &z, &pr, &y); serialize_g1j(proof48, &pr); fr_to_be32(y32, &y); }
    free(poly); return rc;
}
/* reference prove.go:46-77 */
KZGO_EXPORT int kzgo_compute_blob_kzg_proof(const kzgo_ctx *c, const uint8_t *blob, const uint8_t commitment48[48], uint8_t proof48[48]) {
    fr_t *poly = malloc(N_BLOB * sizeof(fr_t)); fr_t z, y; g1j_t pr; g1a_t cm;
    int rc = deserialize_blob(poly, blob);
    if (!rc) rc = deserialize_g1(&cm, commitment48);
    if (!rc) { compute_challenge(&z, blob, commitment48); open_at(c, poly, &z, &pr, &y); serialize_g1j(proof48, &pr); }
    free(poly); return rc;
}
/* test helper: the same over independent (blob, commitment) pairs on `threads` OpenMP threads (0: the OpenMP default) -- so that the
 * full-size GPU tests can hold many proofs of a 1024-blob batch against this restatement in seconds */
KZGO_EXPORT int kzgo_compute_blob_kzg_proof_batch_threads(const kzgo_ctx *c, const uint8_t *blobs, const uint8_t *commitments48, size_t n, uint8_t *proofs48, int *status, int threads) {
    int any = 0;
    if (threads <= 0) threads = omp_get_max_threads();
    #pragma omp parallel for schedule(dynamic) reduction(|:any) num_threads(threads)
    for (size_t i = 0; i < n; i++) {
        int rc = kzgo_compute_blob_kzg_proof(c, blobs + i * BLOB_BYTES, commitments48 + 48 * i, proofs48 + 48 * i);
        if (status) status[i] = rc;
        any |= rc;
    }
    return any ? KZG_ERR_BAD_ARGS : KZG_OK;
}

/* reference internal/kzg/kzg_verify.go:35-100 */
static int verify_single(const kzgo_ctx *c, const g1a_t *commitment, const g1a_t *proof, const fr_t *z, const fr_t *y) {
    g2a_t neg_g2, zg2, amz; g1a_t ps[2]; g2a_t qs[2];
    g2a_neg(&neg_g2, &c->g2[0]);
    g2a_mul_fr(&zg2, &c->g2[0], z); g2a_neg(&zg2, &zg2); g2a_add(&amz, &c->g2[1], &zg2);     /* [alpha - z]G2 */
    g1j_t gen, yg, cj, f;
    fp_t gx, gy; fp_from_plain(&gx, G1_GEN_X); fp_from_plain(&gy, G1_GEN_Y);
    gen.x = gx; gen.y = gy; fp_one(&gen.z);
    g1j_mul_fr(&yg, &gen, y);
    g1j_from_affine(&cj, commitment); g1j_sub(&f, &cj, &yg);                                   /* [f(alpha) - f(z)]G1 */
    g1j_to_affine(&ps[0], &f); ps[1] = *proof; qs[0] = neg_g2; qs[1] = amz;
    return pairing_check(ps, qs, 2) ? KZG_OK : KZG_VERIFY_FAILED;
}
/* reference verify.go:12-43 */
KZGO_EXPORT int kzgo_verify_kzg_proof(const kzgo_ctx *c, const uint8_t commitment48[48], const uint8_t z32[32], const uint8_t y32[32], const uint8_t proof48[48]) {
    fr_t z, y; g1a_t cm, pr; int rc;
    if (!fr_from_be32_canonical(&y, y32)) return KZG_ERR_NON_CANONICAL_SCALAR;
    if (!fr_from_be32_canonical(&z, z32)) return KZG_ERR_NON_CANONICAL_SCALAR;
    if ((rc = deserialize_g1(&cm, commitment48))) return rc;
    if ((rc = deserialize_g1(&pr, proof48))) return rc;
    return verify_single(c, &cm, &pr, &z, &y);
}
/* reference verify.go:48-83 */
KZGO_EXPORT int kzgo_verify_blob_kzg_proof(const kzgo_ctx *c, const uint8_t *blob, const uint8_t commitment48[48], const uint8_t proof48[48]) {
    fr_t *poly = malloc(N_BLOB * sizeof(fr_t)); fr_t z, y; g1a_t cm, pr; int64_t idx;
    int rc = deserialize_blob(poly, blob);
    if (!rc) rc = deserialize_g1(&cm, commitment48);
    if (!rc) rc = deserialize_g1(&pr, proof48);
    if (!rc) { compute_challenge(&z, blob, commitment48); eval_lagrange(&y, &idx, &c->dom, poly, &z); rc = verify_single(c, &cm, &pr, &z, &y); }
    free(poly); return rc;
}
static void random_fr(fr_t *r) {
    uint8_t b[32]; FILE *f = fopen("/dev/urandom", "rb");
    if (!f || fread(b, 1, 32, f) != 32) { for (int i = 0; i < 32; i++) b[i] = (uint8_t)rand(); }
    if (f) fclose(f);
    fr_from_be32_reduce(r, b);
}
/* reference verify.go:88-143 + internal/kzg/kzg_verify.go:111-231.  blobs are contiguous. */
KZGO_EXPORT int kzgo_verify_blob_kzg_proof_batch(const kzgo_ctx *c, const uint8_t *blobs, size_t n_blobs, const uint8_t *commitments48, size_t n_comm,
                                                 const uint8_t *proofs48, size_t n_proofs) {
    if (n_blobs != n_comm || n_blobs != n_proofs) return KZG_ERR_BATCH_LENGTH;
    size_t n = n_blobs;
    g1a_t *cms = malloc((n + 1) * sizeof(g1a_t)), *prs = malloc((n + 1) * sizeof(g1a_t));
    fr_t *zs = malloc((n + 1) * sizeof(fr_t)), *ys = malloc((n + 1) * sizeof(fr_t)), *poly = malloc(N_BLOB * sizeof(fr_t));
    int rc = KZG_OK;
    for (size_t i = 0; i < n && !rc; i++) {
        int64_t idx;
        if ((rc = deserialize_g1(&cms[i], commitments48 + 48 * i))) break;
        if ((rc = deserialize_g1(&prs[i], proofs48 + 48 * i))) break;
        if ((rc = deserialize_blob(poly, blobs + i * BLOB_BYTES))) break;
        compute_challenge(&zs[i], blobs + i * BLOB_BYTES, commitments48 + 48 * i);
        eval_lagrange(&ys[i], &idx, &c->dom, poly, &zs[i]);
    }
    if (!rc) {
        if (n == 0) rc = KZG_OK;
        else if (n == 1) rc = verify_single(c, &cms[0], &prs[0], &zs[0], &ys[0]);
        else {
            fr_t r, *rp = malloc(n * sizeof(fr_t)), *rz = malloc(n * sizeof(fr_t)), fe, t;
            random_fr(&r);
            fr_one(&rp[0]); for (size_t i = 1; i < n; i++) fr_mul(&rp[i], &rp[i - 1], &r);   /* utils.ComputePowers */
            g1j_t fq, fc, fzq, gen, ye;
            msm_g1(&fq, rp, prs, n);                                                           /* kzg_verify.go:150 */
            fr_zero(&fe); for (size_t i = 0; i < n; i++) { fr_mul(&t, &ys[i], &rp[i]); fr_add(&fe, &fe, &t); }
            msm_g1(&fc, rp, cms, n);                                                           /* fold :212-231 */
            fp_from_plain(&gen.x, G1_GEN_X); fp_from_plain(&gen.y, G1_GEN_Y); fp_one(&gen.z);
            g1j_mul_fr(&ye, &gen, &fe); g1j_sub(&fc, &fc, &ye);                                /* :169-172 */
            for (size_t i = 0; i < n; i++) fr_mul(&rz[i], &rp[i], &zs[i]);
            msm_g1(&fzq, rz, prs, n); g1j_add(&fc, &fc, &fzq);                                 /* :176-185 */
            g1j_neg(&fq, &fq);                                                                 /* :188 */
            g1a_t ps[2]; g2a_t qs[2];
            g1j_to_affine(&ps[0], &fc); g1j_to_affine(&ps[1], &fq); qs[0] = c->g2[0]; qs[1] = c->g2[1];
            rc = pairing_check(ps, qs, 2) ? KZG_OK : KZG_VERIFY_FAILED;
            free(rp); free(rz);
        }
    }
    free(cms); free(prs); free(zs); free(ys); free(poly);
    return rc;
}

/* Sharded form of BatchVerifyMultiPoints for multi-GPU jobs: a rank folds the proofs with global indices
 * offset..offset+n with the batch-wide r into (A, C, D, e); records are added and checked by one pairing.
 * Same algebra as internal/kzg/kzg_verify.go:141-199 split at the sums.  Record layout as
 * include/mi355kzg.h (mkzg_verify_blob_kzg_proof_batch_partial_device). */
KZGO_EXPORT int kzgo_verify_blob_batch_partial(const kzgo_ctx *c, const uint8_t *blobs, const uint8_t *commitments48, const uint8_t *proofs48, size_t n, uint64_t offset,
                                               const uint8_t r32[32], uint8_t record[512]) {
    memset(record, 0, 512);
    g1a_t *cms = malloc((n + 1) * sizeof(g1a_t)), *prs = malloc((n + 1) * sizeof(g1a_t));
    fr_t *zs = malloc((n + 1) * sizeof(fr_t)), *ys = malloc((n + 1) * sizeof(fr_t)), *poly = malloc(N_BLOB * sizeof(fr_t));
    fr_t *rp = malloc((n + 1) * sizeof(fr_t)), *rz = malloc((n + 1) * sizeof(fr_t));
    int rc = KZG_OK;
    for (size_t i = 0; i < n && !rc; i++) {
        int64_t idx;
        if ((rc = deserialize_g1(&cms[i], commitments48 + 48 * i))) break;
        if ((rc = deserialize_g1(&prs[i], proofs48 + 48 * i))) break;
        if ((rc = deserialize_blob(poly, blobs + i * BLOB_BYTES))) break;
        compute_challenge(&zs[i], blobs + i * BLOB_BYTES, commitments48 + 48 * i);
        eval_lagrange(&ys[i], &idx, &c->dom, poly, &zs[i]);
    }
    int32_t st = rc;
    memcpy(record, &st, 4);
    if (!rc) {
        fr_t r, fe, t;
        fr_from_be32_reduce(&r, r32);
        fr_pow_u64(&rp[0], &r, offset);
        for (size_t i = 1; i < n; i++) fr_mul(&rp[i], &rp[i - 1], &r);
        fr_zero(&fe);
        for (size_t i = 0; i < n; i++) { fr_mul(&t, &ys[i], &rp[i]); fr_add(&fe, &fe, &t); fr_mul(&rz[i], &rp[i], &zs[i]); }
        g1j_t A, Cc, D;
        msm_g1(&A, rp, prs, n);
        msm_g1(&Cc, rp, cms, n);
        msm_g1(&D, rz, prs, n);
        serialize_g1j(record + 16, &A);
        serialize_g1j(record + 64, &Cc);
        serialize_g1j(record + 112, &D);
        fr_to_be32(record + 160, &fe);
    }
    free(cms); free(prs); free(zs); free(ys); free(poly); free(rp); free(rz);
    return rc;
}
KZGO_EXPORT int kzgo_verify_blob_batch_finish(const kzgo_ctx *c, const uint8_t *records, size_t n_records) {
    g1j_t A, Cc, D, gen, ye; fr_t e;
    g1j_set_inf(&A); g1j_set_inf(&Cc); g1j_set_inf(&D); fr_zero(&e);
    for (size_t k = 0; k < n_records; k++) { int32_t st; memcpy(&st, records + 512 * k, 4); if (st) return st; }
    for (size_t k = 0; k < n_records; k++) {
        const uint8_t *rec = records + 512 * k;
        g1a_t p; fr_t t;
        if (g1a_decompress(&p, rec + 16, 0)) return KZG_ERR_BAD_ARGS;
        g1j_add_affine(&A, &A, &p);
        if (g1a_decompress(&p, rec + 64, 0)) return KZG_ERR_BAD_ARGS;
        g1j_add_affine(&Cc, &Cc, &p);
        if (g1a_decompress(&p, rec + 112, 0)) return KZG_ERR_BAD_ARGS;
        g1j_add_affine(&D, &D, &p);
        if (!fr_from_be32_canonical(&t, rec + 160)) return KZG_ERR_BAD_ARGS;
        fr_add(&e, &e, &t);
    }
    fp_from_plain(&gen.x, G1_GEN_X); fp_from_plain(&gen.y, G1_GEN_Y); fp_one(&gen.z);
    g1j_mul_fr(&ye, &gen, &e); g1j_sub(&Cc, &Cc, &ye); g1j_add(&Cc, &Cc, &D); g1j_neg(&A, &A);
    g1a_t ps[2]; g2a_t qs[2];
    g1j_to_affine(&ps[0], &Cc); g1j_to_affine(&ps[1], &A); qs[0] = c->g2[0]; qs[1] = c->g2[1];
    return pairing_check(ps, qs, 2) ? KZG_OK : KZG_VERIFY_FAILED;
}

/* ------------------------------------------------------------------ EIP-7594 */
/* reference internal/kzg_multi/fk20/fk20.go:58-74: pad, FFT-8192, bit-reverse */
static void compute_extended(const kzgo_ctx *c, const fr_t *coeff, fr_t *ext /*8192*/) {
    memcpy(ext, coeff, N_BLOB * sizeof(fr_t));
    for (int i = N_BLOB; i < N_EXT; i++) fr_zero(&ext[i]);
    domain_fft(&c->dom_ext, ext);
    bitrev_fr(ext, N_EXT);
}
/* reference api_eip7594.go:79-91 + serialization.go:184-194 */
static void serialize_cells(uint8_t *cells_out, const fr_t *ext) { for (int i = 0; i < N_EXT; i++) fr_to_be32(cells_out + 32 * i, &ext[i]); }

/* reference fk20.go:76-124 + toeplitz.go:17-29,95-125 */
static void fk20_proofs(const kzgo_ctx *c, const fr_t *coeff, uint8_t *proofs_out /*128*48*/) {
    static const int L = 64;
    fr_t (*circ)[128] = malloc(64 * sizeof *circ);
    for (int i = 0; i < 64; i++) {
        /* reversed coefficients, stride 64 from offset i: row[j] = rev[i + 64 j] = coeff[4095 - i - 64 j] */
        fr_t row[64];
        for (int j = 0; j < 64; j++) row[j] = coeff[4095 - i - 64 * j];
        for (int j = 0; j < 128; j++) fr_zero(&circ[i][j]);
        circ[i][0] = row[0];                                     /* column = [row0, 0, ...] */
        for (int j = 1; j < L; j++) circ[i][64 + j] = row[64 - j];
        domain_fft(&c->dom128, circ[i]);
    }
    g1j_t res[128];
    #pragma omp parallel for schedule(dynamic)
    for (int k = 0; k < 128; k++) {
        fr_t sc[64];
        for (int i = 0; i < 64; i++) sc[i] = circ[i][k];        /* transpose */
        msm_g1(&res[k], sc, &c->fk20_table[k * 64], 64);
    }
    domain_ifft_g1(&c->dom128, res);
    for (int k = 64; k < 128; k++) g1j_set_inf(&res[k]);        /* keep first half, pad with identity */
    domain_fft_g1(&c->dom128, res);
    for (int k = 0; k < 128; k++) serialize_g1j(proofs_out + 48 * bitrev(k, 128), &res[k]);  /* BitReverse */
    free(circ);
}
/* reference api_eip7594.go:12-26 */
KZGO_EXPORT int kzgo_compute_cells(const kzgo_ctx *c, const uint8_t *blob, uint8_t *cells_out) {
    fr_t *poly = malloc(N_BLOB * sizeof(fr_t)), *ext = malloc(N_EXT * sizeof(fr_t));
    int rc = deserialize_blob(poly, blob);
    if (!rc) { bitrev_fr(poly, N_BLOB); domain_ifft(&c->dom, poly); compute_extended(c, poly, ext); serialize_cells(cells_out, ext); }
    free(poly); free(ext); return rc;
}
/* reference api_eip7594.go:28-52 */
KZGO_EXPORT int kzgo_compute_cells_and_kzg_proofs(const kzgo_ctx *c, const uint8_t *blob, uint8_t *cells_out, uint8_t *proofs_out) {
    if (!c->fk20_table) return KZG_ERR_BAD_ARGS;
    fr_t *poly = malloc(N_BLOB * sizeof(fr_t)), *ext = malloc(N_EXT * sizeof(fr_t));
    int rc = deserialize_blob(poly, blob);
    if (!rc) {
        bitrev_fr(poly, N_BLOB); domain_ifft(&c->dom, poly);
        compute_extended(c, poly, ext); serialize_cells(cells_out, ext);
        fk20_proofs(c, poly, proofs_out);
    }
    free(poly); free(ext); return rc;
}

/* reference internal/erasure_code/erasure_code.go:75-90,110-164 */
static void recover_coeffs(const kzgo_ctx *c, fr_t *data /*8192, natural order, zeros at missing*/, const uint64_t *missing, size_t n_missing, fr_t *coeff_out /*4096*/) {
    /* vanishing polynomial over the missing block roots (natural-order 128-domain), naive product */
    fr_t shortz[N_CELLS + 1]; size_t deg = 0; fr_one(&shortz[0]);
    for (size_t i = 0; i < n_missing; i++) {
        fr_t negx; fr_neg(&negx, &c->dom128.roots[missing[i]]);
        fr_t nxt[N_CELLS + 1];
        for (size_t j = 0; j <= deg + 1; j++) fr_zero(&nxt[j]);
        for (size_t j = 0; j <= deg; j++) { fr_t t; fr_mul(&t, &shortz[j], &negx); fr_add(&nxt[j], &nxt[j], &t); fr_add(&nxt[j + 1], &nxt[j + 1], &shortz[j]); }
        deg++; memcpy(shortz, nxt, (deg + 1) * sizeof(fr_t));
    }
    fr_t *zx = calloc(N_EXT, sizeof(fr_t)), *zev = malloc(N_EXT * sizeof(fr_t)), *ez = malloc(N_EXT * sizeof(fr_t)), *inv = malloc(N_EXT * sizeof(fr_t));
    for (size_t i = 0; i <= deg; i++) zx[i * CELL_SZ] = shortz[i];
    memcpy(zev, zx, N_EXT * sizeof(fr_t)); domain_fft(&c->dom_ext, zev);
    for (int i = 0; i < N_EXT; i++) fr_mul(&ez[i], &data[i], &zev[i]);
    domain_ifft(&c->dom_ext, ez);
    coset_fft(&c->dom_ext, &c->rec_coset_gen, zx);
    coset_fft(&c->dom_ext, &c->rec_coset_gen, ez);
    fr_batch_inv(inv, zx, N_EXT);
    for (int i = 0; i < N_EXT; i++) fr_mul(&ez[i], &ez[i], &inv[i]);
    coset_ifft(&c->dom_ext, &c->rec_coset_gen_inv, ez);
    memcpy(coeff_out, ez, N_BLOB * sizeof(fr_t));
    free(zx); free(zev); free(ez); free(inv);
}
/* reference api_eip7594.go:93-142 */
static int recover_poly(const kzgo_ctx *c, const uint64_t *ids, size_t n_ids, const uint8_t *cells, size_t n_cells, fr_t *coeff) {
    if (n_ids != n_cells) return KZG_ERR_NUM_CELL_IDS;
    for (size_t i = 1; i < n_ids; i++) if (ids[i] <= ids[i - 1]) return KZG_ERR_CELL_IDS_NOT_ORDERED;
    for (size_t i = 0; i < n_ids; i++) if (ids[i] >= N_CELLS) return KZG_ERR_FOUND_INVALID_CELL_ID;
    if (n_ids < N_BLOB / CELL_SZ) return KZG_ERR_NOT_ENOUGH_CELLS;
    uint64_t missing[N_CELLS]; size_t nm = 0; uint8_t have[N_CELLS] = {0};
    for (size_t i = 0; i < n_ids; i++) have[ids[i]] = 1;
    for (uint64_t k = 0; k < N_CELLS; k++) if (!have[k]) missing[nm++] = bitrev(k, N_CELLS);
    fr_t *ext = calloc(N_EXT, sizeof(fr_t));
    for (size_t i = 0; i < n_ids; i++) {
        int rc = deserialize_cell(&ext[ids[i] * CELL_SZ], cells + i * CELL_BYTES);
        if (rc) { free(ext); return rc; }
    }
    bitrev_fr(ext, N_EXT);
    recover_coeffs(c, ext, missing, nm, coeff);
    free(ext);
    return KZG_OK;
}
/* reference api_eip7594.go:144-161 */
KZGO_EXPORT int kzgo_recover_cells_and_kzg_proofs(const kzgo_ctx *c, const uint64_t *ids, size_t n_ids, const uint8_t *cells, size_t n_cells,
                                                  uint8_t *cells_out, uint8_t *proofs_out) {
    if (!c->fk20_table) return KZG_ERR_BAD_ARGS;
    fr_t *coeff = malloc(N_BLOB * sizeof(fr_t)), *ext = malloc(N_EXT * sizeof(fr_t));
    int rc = recover_poly(c, ids, n_ids, cells, n_cells, coeff);
    if (!rc) { compute_extended(c, coeff, ext); serialize_cells(cells_out, ext); fk20_proofs(c, coeff, proofs_out); }
    free(coeff); free(ext); return rc;
}
/* reference api_eip.go:8-15 */
KZGO_EXPORT int kzgo_recover_cells(const kzgo_ctx *c, const uint64_t *ids, size_t n_ids, const uint8_t *cells, size_t n_cells, uint8_t *cells_out) {
    fr_t *coeff = malloc(N_BLOB * sizeof(fr_t)), *ext = malloc(N_EXT * sizeof(fr_t));
    int rc = recover_poly(c, ids, n_ids, cells, n_cells, coeff);
    if (!rc) { compute_extended(c, coeff, ext); serialize_cells(cells_out, ext); }
    free(coeff); free(ext); return rc;
}

/* reference api_eip7594.go:163-265 + internal/kzg_multi/kzg_verify.go:16-105.  cells contiguous. */
KZGO_EXPORT int kzgo_verify_cell_kzg_proof_batch(const kzgo_ctx *c, const uint8_t *commitments48, size_t n_comm, const uint64_t *cell_indices, size_t n_idx,
                                                 const uint8_t *cells, size_t n_cells, const uint8_t *proofs48, size_t n_proofs) {
    /* deduplicateKZGCommitments api_eip7594.go:238-265 */
    size_t n = n_comm, n_rows = 0;
    uint64_t *row_idx = malloc((n + 1) * sizeof(uint64_t)); size_t *row_first = malloc((n + 1) * sizeof(size_t));
    for (size_t i = 0; i < n; i++) {
        size_t j;
        for (j = 0; j < n_rows; j++) if (!memcmp(commitments48 + 48 * row_first[j], commitments48 + 48 * i, 48)) break;
        if (j == n_rows) row_first[n_rows++] = i;
        row_idx[i] = j;
    }
    int rc = KZG_OK;
    if (n != n_idx || n != n_cells || n != n_proofs) rc = KZG_ERR_BATCH_LENGTH;
    else if (n == 0) { free(row_idx); free(row_first); return KZG_OK; }
    if (!rc) for (size_t i = 0; i < n; i++) if (cell_indices[i] >= N_CELLS) { rc = KZG_ERR_INVALID_CELL_ID; break; }
    g1a_t *rows = malloc((n_rows + 1) * sizeof(g1a_t)), *prs = malloc((n + 1) * sizeof(g1a_t));
    fr_t *ev = malloc((n + 1) * CELL_SZ * sizeof(fr_t));
    for (size_t i = 0; i < n_rows && !rc; i++) rc = deserialize_g1(&rows[i], commitments48 + 48 * row_first[i]);
    for (size_t i = 0; i < n && !rc; i++) rc = deserialize_g1(&prs[i], proofs48 + 48 * i);
    for (size_t i = 0; i < n && !rc; i++) rc = deserialize_cell(&ev[i * CELL_SZ], cells + i * CELL_BYTES);
    if (!rc) {
        fr_t r, *rp = malloc(n * sizeof(fr_t)), *wts = calloc(n_rows, sizeof(fr_t)), *wrp = malloc(n * sizeof(fr_t)), interp[CELL_SZ];
        random_fr(&r);
        fr_one(&rp[0]); for (size_t i = 1; i < n; i++) fr_mul(&rp[i], &rp[i - 1], &r);
        g1j_t sum_proofs, sum_comms, sum_interp, sum_wproofs, rl;
        msm_g1(&sum_proofs, rp, prs, n);                                                  /* kzg_verify.go:32 */
        for (size_t k = 0; k < n; k++) fr_add(&wts[row_idx[k]], &wts[row_idx[k]], &rp[k]);
        msm_g1(&sum_comms, wts, rows, n_rows);                                            /* :41 */
        for (int i = 0; i < CELL_SZ; i++) fr_zero(&interp[i]);
        for (size_t k = 0; k < n; k++) {                                                  /* :50-66 */
            fr_t *e = &ev[k * CELL_SZ];
            bitrev_fr(e, CELL_SZ);
            coset_ifft(&c->dom64, &c->coset_shift_inv[cell_indices[k]], e);
            for (int i = 0; i < CELL_SZ; i++) { fr_t t; fr_mul(&t, &e[i], &rp[k]); fr_add(&interp[i], &interp[i], &t); }
        }
        msm_g1(&sum_interp, interp, c->g1_monomial, CELL_SZ);                            /* :68 */
        for (size_t k = 0; k < n; k++) fr_mul(&wrp[k], &c->coset_shift_pow[cell_indices[k]], &rp[k]);
        msm_g1(&sum_wproofs, wrp, prs, n);                                                /* :80 */
        g1j_sub(&rl, &sum_comms, &sum_interp); g1j_add(&rl, &rl, &sum_wproofs);           /* :85-87 */
        g1a_t ps[2]; g2a_t qs[2];
        g1j_to_affine(&ps[0], &sum_proofs); g1j_to_affine(&ps[1], &rl);
        qs[0] = c->g2[CELL_SZ]; g2a_neg(&qs[1], &c->g2[0]);                               /* :89-97 */
        rc = pairing_check(ps, qs, 2) ? KZG_OK : KZG_VERIFY_FAILED;
        free(rp); free(wts); free(wrp);
    }
    free(row_idx); free(row_first); free(rows); free(prs); free(ev);
    return rc;
}

/* Sharded form of VerifyMultiPointKZGProofBatch for multi-GPU jobs: the same algebra (internal/kzg_multi/kzg_verify.go:16-105)
 * split at its four group sums.  A rank folds the cells with global positions offset..offset+n-1 using the batch-wide r;
 * record layout as include/mi355kzg.h (mkzg_verify_cell_kzg_proof_batch_partial). */
KZGO_EXPORT int kzgo_verify_cell_batch_partial(const kzgo_ctx *c, const uint8_t *commitments48, size_t n, const uint64_t *cell_indices, const uint8_t *cells,
                                               const uint8_t *proofs48, uint64_t offset, const uint8_t r32[32], uint8_t record[512]) {
    memset(record, 0, 512);
    size_t n_rows = 0;
    uint64_t *row_idx = malloc((n + 1) * sizeof(uint64_t)); size_t *row_first = malloc((n + 1) * sizeof(size_t));
    for (size_t i = 0; i < n; i++) {
        size_t j;
        for (j = 0; j < n_rows; j++) if (!memcmp(commitments48 + 48 * row_first[j], commitments48 + 48 * i, 48)) break;
        if (j == n_rows) row_first[n_rows++] = i;
        row_idx[i] = j;
    }
    int rc = KZG_OK;
    for (size_t i = 0; i < n; i++) if (cell_indices[i] >= N_CELLS) { rc = KZG_ERR_INVALID_CELL_ID; break; }
    g1a_t *rows = malloc((n_rows + 1) * sizeof(g1a_t)), *prs = malloc((n + 1) * sizeof(g1a_t));
    fr_t *ev = malloc((n + 1) * CELL_SZ * sizeof(fr_t));
    for (size_t i = 0; i < n_rows && !rc; i++) rc = deserialize_g1(&rows[i], commitments48 + 48 * row_first[i]);
    for (size_t i = 0; i < n && !rc; i++) rc = deserialize_g1(&prs[i], proofs48 + 48 * i);
    for (size_t i = 0; i < n && !rc; i++) rc = deserialize_cell(&ev[i * CELL_SZ], cells + i * CELL_BYTES);
    int32_t st = rc;
    memcpy(record, &st, 4);
    if (!rc) {
        fr_t r, *rp = malloc((n + 1) * sizeof(fr_t)), *wts = calloc(n_rows + 1, sizeof(fr_t)), *wrp = malloc((n + 1) * sizeof(fr_t)), interp[CELL_SZ];
        fr_from_be32_reduce(&r, r32);
        if (n) fr_pow_u64(&rp[0], &r, offset);
        for (size_t i = 1; i < n; i++) fr_mul(&rp[i], &rp[i - 1], &r);
        g1j_t sums[4];
        msm_g1(&sums[0], rp, prs, n);
        for (size_t k = 0; k < n; k++) fr_add(&wts[row_idx[k]], &wts[row_idx[k]], &rp[k]);
        msm_g1(&sums[1], wts, rows, n_rows);
        for (int i = 0; i < CELL_SZ; i++) fr_zero(&interp[i]);
        for (size_t k = 0; k < n; k++) {
            fr_t *e = &ev[k * CELL_SZ];
            bitrev_fr(e, CELL_SZ);
            coset_ifft(&c->dom64, &c->coset_shift_inv[cell_indices[k]], e);
            for (int i = 0; i < CELL_SZ; i++) { fr_t t; fr_mul(&t, &e[i], &rp[k]); fr_add(&interp[i], &interp[i], &t); }
        }
        msm_g1(&sums[2], interp, c->g1_monomial, CELL_SZ);
        for (size_t k = 0; k < n; k++) fr_mul(&wrp[k], &c->coset_shift_pow[cell_indices[k]], &rp[k]);
        msm_g1(&sums[3], wrp, prs, n);
        for (int m = 0; m < 4; m++) serialize_g1j(record + 16 + 48 * m, &sums[m]);
        free(rp); free(wts); free(wrp);
    }
    free(row_idx); free(row_first); free(rows); free(prs); free(ev);
    return rc;
}
KZGO_EXPORT int kzgo_verify_cell_batch_finish(const kzgo_ctx *c, const uint8_t *records, size_t n_records) {
    for (size_t k = 0; k < n_records; k++) { int32_t st; memcpy(&st, records + 512 * k, 4); if (st) return st; }
    g1j_t sums[4], rl;
    for (int m = 0; m < 4; m++) g1j_set_inf(&sums[m]);
    for (size_t k = 0; k < n_records; k++)
        for (int m = 0; m < 4; m++) {
            g1a_t p;
            if (g1a_decompress(&p, records + 512 * k + 16 + 48 * m, 0)) return KZG_ERR_BAD_ARGS;
            g1j_add_affine(&sums[m], &sums[m], &p);
        }
    g1j_sub(&rl, &sums[1], &sums[2]); g1j_add(&rl, &rl, &sums[3]);
    g1a_t ps[2]; g2a_t qs[2];
    g1j_to_affine(&ps[0], &sums[0]); g1j_to_affine(&ps[1], &rl);
    qs[0] = c->g2[CELL_SZ]; g2a_neg(&qs[1], &c->g2[0]);
    return pairing_check(ps, qs, 2) ? KZG_OK : KZG_VERIFY_FAILED;
}

/* ------------------------------------------------------------------ unit-level probes (tests only) */
/* n scalars (32 B big-endian each), in place; inverse!=0 -> IfftFr. n power of two <= 2^20 */
KZGO_EXPORT int kzgo_fr_fft(uint8_t *vals32, size_t n, int inverse) {
    domain_t d; domain_init(&d, n);
    fr_t *v = malloc(n * sizeof(fr_t));
    for (size_t i = 0; i < n; i++) if (!fr_from_be32_canonical(&v[i], vals32 + 32 * i)) { free(v); domain_free(&d); return KZG_ERR_NON_CANONICAL_SCALAR; }
    if (inverse) domain_ifft(&d, v); else domain_fft(&d, v);
    for (size_t i = 0; i < n; i++) fr_to_be32(vals32 + 32 * i, &v[i]);
    free(v); domain_free(&d); return KZG_OK;
}
KZGO_EXPORT int kzgo_fr_coset_fft(uint8_t *vals32, size_t n, const uint8_t gen32[32], int inverse) {
    domain_t d; domain_init(&d, n); fr_t g, gi;
    if (!fr_from_be32_canonical(&g, gen32)) { domain_free(&d); return KZG_ERR_NON_CANONICAL_SCALAR; }
    fr_inv(&gi, &g);
    fr_t *v = malloc(n * sizeof(fr_t));
    for (size_t i = 0; i < n; i++) fr_from_be32_reduce(&v[i], vals32 + 32 * i);
    if (inverse) coset_ifft(&d, &gi, v); else coset_fft(&d, &g, v);
    for (size_t i = 0; i < n; i++) fr_to_be32(vals32 + 32 * i, &v[i]);
    free(v); domain_free(&d); return KZG_OK;
}
/* sum s_i P_i for compressed points (subgroup-unchecked) and 32-byte BE scalars (reduced) */
KZGO_EXPORT int kzgo_g1_msm(const uint8_t *scalars32, const uint8_t *points48, size_t n, uint8_t out48[48]) {
    fr_t *s = malloc((n + 1) * sizeof(fr_t)); g1a_t *p = malloc((n + 1) * sizeof(g1a_t)); int rc = KZG_OK;
    for (size_t i = 0; i < n; i++) { fr_from_be32_reduce(&s[i], scalars32 + 32 * i); if (g1a_decompress(&p[i], points48 + 48 * i, 0)) rc = KZG_ERR_G1_BAD_ENCODING; }
    if (!rc) { g1j_t r; msm_g1(&r, s, p, n); serialize_g1j(out48, &r); }
    free(s); free(p); return rc;
}
/* naive double-and-add sum, independent of msm_g1 (reference internal/multiexp/multiexp_test.go:98-133) */
KZGO_EXPORT int kzgo_g1_msm_naive(const uint8_t *scalars32, const uint8_t *points48, size_t n, uint8_t out48[48]) {
    g1j_t acc; g1j_set_inf(&acc);
    for (size_t i = 0; i < n; i++) {
        fr_t s; g1a_t p; g1j_t pj, t;
        fr_from_be32_reduce(&s, scalars32 + 32 * i);
        if (g1a_decompress(&p, points48 + 48 * i, 0)) return KZG_ERR_G1_BAD_ENCODING;
        g1j_from_affine(&pj, &p); g1j_mul_fr(&t, &pj, &s); g1j_add(&acc, &acc, &t);
    }
    serialize_g1j(out48, &acc); return KZG_OK;
}
/* full validation of a compressed point, as gnark SetBytes */
KZGO_EXPORT int kzgo_g1_validate(const uint8_t in48[48]) { g1a_t p; return deserialize_g1(&p, in48); }
KZGO_EXPORT void kzgo_compute_challenge(const uint8_t *blob, const uint8_t commitment48[48], uint8_t out32[32]) { fr_t z; compute_challenge(&z, blob, commitment48); fr_to_be32(out32, &z); }
KZGO_EXPORT void kzgo_sha256(const uint8_t *d, size_t n, uint8_t out[32]) { sha256_ctx h; sha256_init(&h); sha256_update(&h, d, n); sha256_final(&h, out); }
/* barycentric evaluation of a blob at z (reference internal/domain/domain.go:179-235 with the Context's reversed roots) */
KZGO_EXPORT int kzgo_evaluate_blob(const kzgo_ctx *c, const uint8_t *blob, const uint8_t z32[32], uint8_t y32[32]) {
    fr_t *poly = malloc(N_BLOB * sizeof(fr_t)); fr_t z, y; int64_t idx;
    int rc = deserialize_blob(poly, blob);
    if (!rc && !fr_from_be32_canonical(&z, z32)) rc = KZG_ERR_NON_CANONICAL_SCALAR;
    if (!rc) { eval_lagrange(&y, &idx, &c->dom, poly, &z); fr_to_be32(y32, &y); }
    free(poly); return rc;
}
/* commitment with an arbitrary caller-supplied SRS (compressed points): Lagrange-form KZG commit
 * over a small domain, for the reference's TestCommitRegression KAT (internal/kzg/srs_test.go:61-71) */
KZGO_EXPORT int kzgo_commit_with_srs(const uint8_t *scalars32, const uint8_t *srs48, size_t n, uint8_t out48[48]) { return kzgo_g1_msm(scalars32, srs48, n, out48); }
/* [k]G for the standard generator, k as 32-byte BE (reduced) */
KZGO_EXPORT void kzgo_g1_gen_mul(const uint8_t k32[32], uint8_t out48[48]) {
    fr_t k; g1j_t gen, r; fr_from_be32_reduce(&k, k32);
    fp_from_plain(&gen.x, G1_GEN_X); fp_from_plain(&gen.y, G1_GEN_Y); fp_one(&gen.z);
    g1j_mul_fr(&r, &gen, &k); serialize_g1j(out48, &r);
}
/* GetRandBlob (reference bench_test.go:17-46): scalar at byte offset i = SHA256(int64_be(seed+i)) mod r */
KZGO_EXPORT void kzgo_rand_blob(int64_t seed, uint8_t *blob) {
    for (int i = 0; i < BLOB_BYTES; i += 32) {
        uint8_t in[8], d[32]; uint64_t s = (uint64_t)(seed + i); fr_t t;
        for (int j = 0; j < 8; j++) in[j] = (uint8_t)(s >> (56 - 8 * j));
        kzgo_sha256(in, 8, d); fr_from_be32_reduce(&t, d); fr_to_be32(blob + i, &t);
    }
}
