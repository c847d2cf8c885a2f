/* Minimal C caller of the engine's C ABI (include/mktfhe.h): two-party KMS NAND on a small batch.
 * Counterpart of the reference's test/KMS.jl flow: CRS -> party_keygen -> setup -> lwe_ith_encrypt -> NAND ->
 * lwe_decrypt.  Build (from the repo root):
 *   gcc -O2 -Iinclude examples/kms_nand.c -o examples/kms_nand -Lmktfhe_amd/lib -lmktfhe_hip -Wl,-rpath,$PWD/mktfhe_amd/lib
 */
#include <stdio.h>
#include <stdlib.h>

#include "mktfhe.h"

#define CK(call) do { int _r = (call); if (_r < 0) { fprintf(stderr, "%s failed: %d (%s)\n", #call, _r, mkt_last_error(ctx)); return 1; } } while (0)

int main(int argc, char **argv) {
    /* KMS2party (src/tfhe/params.jl:47-53), optionally with a reduced n / N for a quick run */
    mkt_params p = { MKT_KMS, 560, 2048, 2, 64, 3, 12, 2, 7, 3, 10, 8, 2, 0, 0 };
    if (argc > 2) { p.n = atoi(argv[1]); p.N = atoi(argv[2]); }
    const double alpha = 131072.0, beta = 85.4084;
    const int B = 8, len = p.k * p.n + 1;
    mkt_ctx *ctx = NULL;

    uint64_t *crs = malloc(sizeof(uint64_t) * (size_t)p.l_uni * p.N);
    /* pinned seeds so the example is reproducible -- a real client passes NULL (fresh OS entropy per call) */
    uint8_t seed[32];
    CK(mkt_client_test_seed(1, seed));
    CK(mkt_client_crs(&p, seed, crs));
    mkt_client_party *party[2];
    for (int i = 0; i < 2; i++) CK(mkt_client_party_keygen(&p, seed, i, crs, alpha, beta, &party[i]));

    CK(mkt_ctx_create(&p, MKT_ARITH_F64REF, 0, &ctx));
    CK(mkt_load_crs(ctx, crs, MKT_FMT_INT_COEFF));
    for (int i = 0; i < 2; i++) {
        size_t nb;
        CK(mkt_load_brk(ctx, i, mkt_client_brk(party[i], &nb), MKT_FMT_INT_COEFF));
        CK(mkt_load_ksk(ctx, i, mkt_client_ksk(party[i], &nb)));
        CK(mkt_load_rlk(ctx, i, mkt_client_rlk_d(party[i], &nb), mkt_client_rlk_f(party[i], &nb), MKT_FMT_INT_COEFF));
        CK(mkt_load_pubkey(ctx, i, mkt_client_pubkey(party[i], &nb), MKT_FMT_INT_COEFF));
    }

    uint32_t *x = malloc(sizeof(uint32_t) * (size_t)B * len), *y = malloc(sizeof(uint32_t) * (size_t)B * len), *z = malloc(sizeof(uint32_t) * (size_t)B * len);
    int bx[8], by[8], bad = 0;
    for (int j = 0; j < B; j++) {
        bx[j] = j & 1; by[j] = (j >> 1) & 1;
        CK(mkt_client_lwe_encrypt(&p, party[0], 0, bx[j], alpha, NULL, x + (size_t)j * len));   /* party 0's bit, fresh randomness */
        CK(mkt_client_lwe_encrypt(&p, party[1], 1, by[j], alpha, NULL, y + (size_t)j * len));   /* party 1's bit */
    }
    CK(mkt_gate_batch(ctx, MKT_NAND, x, y, z, B, MKT_MEM_HOST));
    for (int j = 0; j < B; j++) {
        int m = mkt_client_lwe_decrypt(&p, (const mkt_client_party *const *)party, 2, z + (size_t)j * len);
        printf("NAND(%d, %d) = %d\n", bx[j], by[j], m);
        bad += m != !(bx[j] && by[j]);
    }
    mkt_ctx_destroy(ctx);
    for (int i = 0; i < 2; i++) mkt_client_party_destroy(party[i]);
    free(crs); free(x); free(y); free(z);
    printf(bad ? "FAILED\n" : "ok\n");
    return bad != 0;
}
