/* One evaluator over several GPUs from plain C (include/mktfhe.h, mkt_multi_*): the caller is ONE process, as a Julia caller of
 * the ccall shim is; keys are uploaded once on the first device and replicated device to device; the batch is cut into
 * contiguous shards and every shard writes its slice of the one output array.  Also shows a different gate per ciphertext
 * pair (mkt_multi_gate_batch_ops, the shape of the reference's test/KMS.jl:29-34) and the native MUX.
 *   gcc -O2 -Iinclude examples/multi_nand.c -o examples/multi_nand -Lmktfhe_amd/lib -lmktfhe_hip -Wl,-rpath,$PWD/mktfhe_amd/lib
 *   examples/multi_nand <n> <N> <shards>      shards on device 0 .. ndev-1, wrapping around (one GPU: logical shards of device 0)
 */
#include <stdio.h>
#include <stdlib.h>

#include "mktfhe.h"

#define CK(call) do { int _r = (call); if (_r < 0) { fprintf(stderr, "%s failed: %d (%s)\n", #call, _r, mkt_multi_last_error(m)); return 1; } } while (0)

int main(int argc, char **argv) {
    mkt_params p = { MKT_KMS, 560, 2048, 2, 64, 3, 12, 2, 7, 3, 10, 8, 2, 0, 0 };        /* KMS2party, src/tfhe/params.jl:47-53 */
    int shards = 3, ndev = 1;
    if (argc > 2) { p.n = atoi(argv[1]); p.N = atoi(argv[2]); }
    if (argc > 3) shards = atoi(argv[3]);
    if (argc > 4) ndev = atoi(argv[4]);
    const double alpha = 131072.0, beta = 85.4084;
    const int B = 13, len = p.k * p.n + 1;                                             /* 13 over 3 shards: 5 + 4 + 4 */
    mkt_multi *m = NULL;
    int devices[64];
    if (shards < 1 || shards > 64 || ndev < 1) return 2;
    for (int i = 0; i < shards; i++) devices[i] = i % ndev;

    uint8_t seed[32];
    uint64_t *crs = malloc(sizeof(uint64_t) * (size_t)p.l_uni * p.N);
    CK(mkt_client_test_seed(1, seed));                                                 /* pinned: reproducible example, NOT for real keys */
    CK(mkt_client_crs(&p, seed, crs));
    mkt_client_party *party[2];
    for (int i = 0; i < 2; i++) CK(mkt_client_party_keygen(&p, seed, i, crs, alpha, beta, &party[i]));

    CK(mkt_multi_create(&p, MKT_ARITH_F64REF, devices, shards, 0, &m));
    CK(mkt_multi_load_crs(m, crs, MKT_FMT_INT_COEFF));
    for (int i = 0; i < 2; i++) {
        size_t nb;
        CK(mkt_multi_load_brk(m, i, mkt_client_brk(party[i], &nb), MKT_FMT_INT_COEFF));
        CK(mkt_multi_load_ksk(m, i, mkt_client_ksk(party[i], &nb)));
        CK(mkt_multi_load_rlk(m, i, mkt_client_rlk_d(party[i], &nb), mkt_client_rlk_f(party[i], &nb), MKT_FMT_INT_COEFF));
        CK(mkt_multi_load_pubkey(m, i, mkt_client_pubkey(party[i], &nb), MKT_FMT_INT_COEFF));
    }
    CK(mkt_multi_replicate(m));                                                        /* keys to every device; immutable from here */

    uint32_t *x = malloc(4 * (size_t)B * len), *y = malloc(4 * (size_t)B * len), *z = malloc(4 * (size_t)B * len), *w = malloc(4 * (size_t)B * len);
    uint8_t ops[13];
    int bx[13], by[13], bad = 0;
    for (int j = 0; j < B; j++) {
        bx[j] = j & 1; by[j] = (j >> 1) & 1; ops[j] = (uint8_t)(j % 6);
        CK(mkt_client_lwe_encrypt(&p, party[0], 0, bx[j], alpha, NULL, x + (size_t)j * len));
        CK(mkt_client_lwe_encrypt(&p, party[1], 1, by[j], alpha, NULL, y + (size_t)j * len));
    }
    CK(mkt_multi_gate_batch_ops(m, ops, x, y, z, B, MKT_MEM_HOST));                    /* gate j % 6 on pair j */
    CK(mkt_multi_mux_batch(m, x, y, z, w, B, MKT_MEM_HOST));                           /* w = x ? y : z */
    for (int j = 0; j < B; j++) {
        const int g = mkt_client_lwe_decrypt(&p, (const mkt_client_party *const *)party, 2, z + (size_t)j * len);
        const int u = mkt_client_lwe_decrypt(&p, (const mkt_client_party *const *)party, 2, w + (size_t)j * len);
        const int a = bx[j], b = by[j];
        const int want[6] = { !(a && b), a && b, a || b, a ^ b, !(a ^ b), !(a || b) };
        bad += g != want[ops[j]];
        bad += u != (a ? b : want[ops[j]]);
        size_t lo, hi;
        printf("pair %2d gate %d(%d, %d) = %d   mux = %d\n", j, ops[j], a, b, g, u);
        if (j < shards) { CK(mkt_multi_shard_range(m, B, j, &lo, &hi)); printf("   shard %d on device %d: gates [%zu, %zu)\n", j, mkt_multi_device(m, j), lo, hi); }
    }
    mkt_multi_destroy(m);
    for (int i = 0; i < 2; i++) mkt_client_party_destroy(party[i]);
    free(crs); free(x); free(y); free(z); free(w);
    printf(bad ? "FAILED\n" : "ok\n");
    return bad != 0;
}
