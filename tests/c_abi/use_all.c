/* tests/c_abi/use_all.c -- the drop-in boundary used the way a cgo / C caller uses it: plain C99, include/mi355kzg.h only,
 * linked against libmi355kzg.so.  Compiled and linked by tests/test_abi_exports.py on the CPU (no GPU needed: with no HIP
 * device mkzg_ctx_new must fail cleanly, which main() checks); on a GPU box the same binary checks results: the zero blob, one
 * consensus-spec vector byte for byte (tests/golden/c_caller_vector.bin = blob | commitment | blob proof), and the verifiers'
 * accept / reject on it. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mi355kzg.h"

/* every entry point is referenced once, so that a signature drift between header and library fails the link or the compile */
typedef void (*any_fn)(void);
static any_fn all_symbols[] = {
    (any_fn)mkzg_ctx_new, (any_fn)mkzg_ctx_free, (any_fn)mkzg_ctx_set_option, (any_fn)mkzg_ctx_get_option, (any_fn)mkzg_blob_to_kzg_commitment,
    (any_fn)mkzg_blob_to_kzg_commitment_batch, (any_fn)mkzg_blob_to_kzg_commitment_batch_device, (any_fn)mkzg_commit_lagrange,
    (any_fn)mkzg_compute_kzg_proof, (any_fn)mkzg_compute_blob_kzg_proof, (any_fn)mkzg_compute_blob_kzg_proof_batch_device,
    (any_fn)mkzg_verify_kzg_proof, (any_fn)mkzg_verify_blob_kzg_proof, (any_fn)mkzg_verify_blob_kzg_proof_batch,
    (any_fn)mkzg_verify_blob_kzg_proof_batch_device, (any_fn)mkzg_verify_blob_kzg_proof_batch_partial_device,
    (any_fn)mkzg_verify_blob_kzg_proof_batch_finish, (any_fn)mkzg_compute_cells, (any_fn)mkzg_compute_cells_batch_device,
    (any_fn)mkzg_compute_cells_and_kzg_proofs, (any_fn)mkzg_compute_cells_and_kzg_proofs_batch_device,
    (any_fn)mkzg_verify_cell_kzg_proof_batch, (any_fn)mkzg_verify_cell_kzg_proof_batch_partial, (any_fn)mkzg_verify_cell_kzg_proof_batch_finish,
    (any_fn)mkzg_recover_cells_and_kzg_proofs, (any_fn)mkzg_recover_cells_and_kzg_proofs_batch, (any_fn)mkzg_recover_cells, (any_fn)mkzg_profile_enable, (any_fn)mkzg_profile_read,
    (any_fn)mkzg_last_host_finish_us, (any_fn)mkzg_valu_calibrate, (any_fn)mkzg_msm_g1, (any_fn)mkzg_version, (any_fn)mkzg_evaluate_blobs_device, (any_fn)mkzg_recover_cells_and_kzg_proofs_batch_device, (any_fn)mkzg_verify_cell_kzg_proof_batch_device, (any_fn)mkzg_verify_cell_kzg_proof_batch_partial_device, (any_fn)mkzg_comm_unique_id, (any_fn)mkzg_comm_init, (any_fn)mkzg_comm_from_nccl, (any_fn)mkzg_comm_free,
    (any_fn)mkzg_comm_last_exchange_us, (any_fn)mkzg_verify_blob_kzg_proof_batch_exchange, (any_fn)mkzg_verify_blob_kzg_proof_batch_exchange_device,
    (any_fn)mkzg_verify_cell_kzg_proof_batch_exchange, (any_fn)mkzg_verify_cell_kzg_proof_batch_exchange_device, (any_fn)mkzg_ctx_prepare,
};

static char* slurp(const char* path, long* len) {
    FILE* f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    *len = ftell(f);
    fseek(f, 0, SEEK_SET);
    char* p = (char*)malloc((size_t)*len + 1);
    if (p && fread(p, 1, (size_t)*len, f) != (size_t)*len) { free(p); p = NULL; }
    fclose(f);
    return p;
}
static int report(const char* what, int rc, int want_rc, int ok) {
    ok = ok && rc == want_rc;
    printf("%s: rc=%d %s\n", what, rc, ok ? "ok" : "WRONG");
    return ok;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s trusted_setup.json [vector.bin]\n", argv[0]); return 2; }
    long len = 0;
    char* js = slurp(argv[1], &len);
    if (!js) return 2;
    printf("%s, %d entry points\n", mkzg_version(), (int)(sizeof(all_symbols) / sizeof(all_symbols[0])));
    mkzg_ctx* ctx = NULL;
    int rc = mkzg_ctx_new(js, (size_t)len, NULL, 0, &ctx);
    free(js);
    if (rc == MKZG_ERR_HIP) { printf("no HIP device: mkzg_ctx_new -> MKZG_ERR_HIP (no fallback)\n"); return 0; }
    if (rc != MKZG_OK) { printf("mkzg_ctx_new -> %d\n", rc); return 1; }
    int all = 1;
    all &= report("mkzg_ctx_prepare(bad mask)", mkzg_ctx_prepare(ctx, 8u), MKZG_ERR_BAD_ARGS, 1);
    all &= report("mkzg_ctx_prepare(commit)", mkzg_ctx_prepare(ctx, MKZG_PREPARE_COMMIT), MKZG_OK, 1);
    static uint8_t zero[MKZG_BYTES_PER_BLOB];           /* the zero polynomial commits to the identity: c0 00 .. 00 */
    uint8_t out[48], proof[48];
    rc = mkzg_blob_to_kzg_commitment(ctx, zero, 0, out);
    int ok = out[0] == 0xc0;
    for (int i = 1; i < 48; i++) ok = ok && out[i] == 0;
    all &= report("commitment of the zero blob", rc, MKZG_OK, ok);
    if (argc >= 3) {
        long vlen = 0;
        uint8_t* v = (uint8_t*)slurp(argv[2], &vlen);
        if (!v || vlen != MKZG_BYTES_PER_BLOB + 96) { printf("bad vector file\n"); return 2; }
        const uint8_t *blob = v, *want_cm = v + MKZG_BYTES_PER_BLOB, *want_pr = want_cm + 48;
        rc = mkzg_blob_to_kzg_commitment(ctx, blob, 0, out);
        all &= report("commitment vector", rc, MKZG_OK, !memcmp(out, want_cm, 48));
        rc = mkzg_compute_blob_kzg_proof(ctx, blob, want_cm, 0, proof);
        all &= report("proof vector", rc, MKZG_OK, !memcmp(proof, want_pr, 48));
        all &= report("verify valid", mkzg_verify_blob_kzg_proof(ctx, blob, want_cm, want_pr), MKZG_OK, 1);
        all &= report("verify corrupted", mkzg_verify_blob_kzg_proof(ctx, blob, want_cm, want_cm), MKZG_VERIFY_FAILED, 1);
        /* a batch of three through Go's []*Blob shape: array of pointers, contiguous commitments and proofs */
        const uint8_t* blobs[3] = {blob, zero, blob};
        uint8_t cms[3 * 48], prs[3 * 48];
        memset(cms, 0, sizeof(cms)); memset(prs, 0, sizeof(prs));
        memcpy(cms, want_cm, 48); memcpy(cms + 96, want_cm, 48);
        memcpy(prs, want_pr, 48); memcpy(prs + 96, want_pr, 48);
        cms[48] = 0xc0; prs[48] = 0xc0;                 /* zero blob: identity commitment, identity proof */
        all &= report("batch verify valid", mkzg_verify_blob_kzg_proof_batch(ctx, blobs, cms, prs, 3, 3, 3), MKZG_OK, 1);
        memcpy(prs + 96, want_cm, 48);
        all &= report("batch verify corrupted", mkzg_verify_blob_kzg_proof_batch(ctx, blobs, cms, prs, 3, 3, 3), MKZG_VERIFY_FAILED, 1);
        free(v);
    }
    mkzg_ctx_free(ctx);
    return all ? 0 : 1;
}
