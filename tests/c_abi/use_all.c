/* tests/c_abi/use_all.c -- the drop-in boundary used the way a cgo / C caller uses it: plain C99, include/mi355kzg.h only,
 * linked against libmi355kzg.so.  Compiled and linked by tests/test_abi_exports.py on the CPU (no GPU needed: with no HIP
 * device mkzg_ctx_new must fail cleanly, which main() checks); on a GPU box the same binary runs one commitment. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mi355kzg.h"

/* every entry point is referenced once, so that a signature drift between header and library fails the link or the compile */
typedef void (*any_fn)(void);
static any_fn all_symbols[] = {
    (any_fn)mkzg_ctx_new, (any_fn)mkzg_ctx_free, (any_fn)mkzg_ctx_set_option, (any_fn)mkzg_blob_to_kzg_commitment,
    (any_fn)mkzg_blob_to_kzg_commitment_batch, (any_fn)mkzg_blob_to_kzg_commitment_batch_device, (any_fn)mkzg_commit_lagrange,
    (any_fn)mkzg_compute_kzg_proof, (any_fn)mkzg_compute_blob_kzg_proof, (any_fn)mkzg_compute_blob_kzg_proof_batch_device,
    (any_fn)mkzg_verify_kzg_proof, (any_fn)mkzg_verify_blob_kzg_proof, (any_fn)mkzg_verify_blob_kzg_proof_batch,
    (any_fn)mkzg_verify_blob_kzg_proof_batch_device, (any_fn)mkzg_verify_blob_kzg_proof_batch_partial_device,
    (any_fn)mkzg_verify_blob_kzg_proof_batch_finish, (any_fn)mkzg_compute_cells, (any_fn)mkzg_compute_cells_batch_device,
    (any_fn)mkzg_compute_cells_and_kzg_proofs, (any_fn)mkzg_compute_cells_and_kzg_proofs_batch_device,
    (any_fn)mkzg_verify_cell_kzg_proof_batch, (any_fn)mkzg_verify_cell_kzg_proof_batch_partial, (any_fn)mkzg_verify_cell_kzg_proof_batch_finish,
    (any_fn)mkzg_recover_cells_and_kzg_proofs, (any_fn)mkzg_recover_cells_and_kzg_proofs_batch, (any_fn)mkzg_recover_cells, (any_fn)mkzg_profile_enable, (any_fn)mkzg_profile_read,
    (any_fn)mkzg_last_host_finish_us, (any_fn)mkzg_valu_calibrate, (any_fn)mkzg_msm_g1, (any_fn)mkzg_version,
};

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s trusted_setup.json\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    fseek(f, 0, SEEK_END);
    long len = ftell(f);
    fseek(f, 0, SEEK_SET);
    char* js = (char*)malloc((size_t)len + 1);
    if (fread(js, 1, (size_t)len, f) != (size_t)len) return 2;
    fclose(f);
    printf("%s, %d entry points\n", mkzg_version(), (int)(sizeof(all_symbols) / sizeof(all_symbols[0])));
    mkzg_ctx* ctx = NULL;
    int rc = mkzg_ctx_new(js, (size_t)len, NULL, 0, &ctx);
    free(js);
    if (rc == MKZG_ERR_HIP) { printf("no HIP device: mkzg_ctx_new -> MKZG_ERR_HIP (no fallback)\n"); return 0; }
    if (rc != MKZG_OK) { printf("mkzg_ctx_new -> %d\n", rc); return 1; }
    static uint8_t blob[MKZG_BYTES_PER_BLOB];           /* the zero polynomial commits to the identity: c0 00 .. 00 */
    uint8_t out[48];
    rc = mkzg_blob_to_kzg_commitment(ctx, blob, 0, out);
    int ok = rc == MKZG_OK && out[0] == 0xc0;
    for (int i = 1; i < 48; i++) ok = ok && out[i] == 0;
    printf("commitment of the zero blob: rc=%d %s\n", rc, ok ? "ok" : "WRONG");
    mkzg_ctx_free(ctx);
    return ok ? 0 : 1;
}
