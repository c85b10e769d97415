/* A host WITHOUT Python or torch: plain C11 against include/mfar_hip.h + libmfar_hip.so -- what a maintainer binding the scorer from
 * another language gets.  Builds an index from host memory, runs the synchronous scorer (mfar_search_two_stage) and the batch pipeline
 * (mfar_pipeline_*) on host buffers, checks that both return the same bits, and writes inputs + outputs to a binary file that
 * tests/test_gpu_integration.py compares with the C oracle.
 *     abi_client <out.bin> D F E Q n_batches
 * File: int32 D, F, E, Q, NB | slab f32 [F][D][E] | W f32 [E][F] | mask f32 [F] | q f32 [NB][Q][E] | ids i64 [NB][Q][100] | scores f32 [NB][Q][100] */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mfar_hip.h"

#define K 100
#define CHECK(call)                                                                         \
    do {                                                                                    \
        int rc_ = (call);                                                                   \
        if (rc_ != MFAR_OK) {                                                               \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, mfar_last_error());               \
            return 1;                                                                       \
        }                                                                                   \
    } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static float rnd(void) { /* xorshift64*: uniform in [-1, 1) */
    rng_state ^= rng_state >> 12;
    rng_state ^= rng_state << 25;
    rng_state ^= rng_state >> 27;
    return (float)((double)((rng_state * 0x2545F4914F6CDD1Dull) >> 11) / 9007199254740992.0 * 2.0 - 1.0);
}

int main(int argc, char** argv) {
    if (argc != 7) {
        fprintf(stderr, "usage: %s out.bin D F E Q n_batches\n", argv[0]);
        return 2;
    }
    const int D = atoi(argv[2]), F = atoi(argv[3]), E = atoi(argv[4]), Q = atoi(argv[5]), NB = atoi(argv[6]);
    if (mfar_version() != MFAR_ABI_VERSION) {
        fprintf(stderr, "library ABI %d, header %d\n", mfar_version(), MFAR_ABI_VERSION);
        return 1;
    }
    int ndev = 0;
    CHECK(mfar_device_count(&ndev));
    if (ndev < 1) {
        fprintf(stderr, "no device\n");
        return 1;
    }
    float* slab = malloc((size_t)F * D * E * 4);
    float* W = malloc((size_t)E * F * 4);
    float* mask = malloc((size_t)F * 4);
    float* q = malloc((size_t)NB * Q * E * 4);
    int64_t *ids = malloc((size_t)NB * Q * K * 8), *ids2 = malloc((size_t)NB * Q * K * 8);
    float *sc = malloc((size_t)NB * Q * K * 4), *sc2 = malloc((size_t)NB * Q * K * 4);
    int32_t* nv = malloc((size_t)NB * Q * 4);
    if (!slab || !W || !mask || !q || !ids || !ids2 || !sc || !sc2 || !nv) return 1;
    for (size_t i = 0; i < (size_t)F * D * E; ++i) slab[i] = 0.5f * rnd() + 0.05f;
    for (size_t i = 0; i < (size_t)E * F; ++i) W[i] = 0.05f * rnd();
    for (int f = 0; f < F; ++f) mask[f] = f == 1 ? 0.0f : 1.0f;
    for (size_t i = 0; i < (size_t)NB * Q * E; ++i) q[i] = 0.5f * rnd() + 0.1f;

    mfar_index* ix = NULL;
    CHECK(mfar_index_create(&ix, 0, D, 0, F, E, MFAR_DTYPE_F32));
    for (int f = 0; f < F; ++f) CHECK(mfar_index_write_rows(ix, f, 0, D, slab + (size_t)f * D * E, 0, NULL));

    /* 1. one synchronous call per batch, host pointers */
    for (int b = 0; b < NB; ++b)
        CHECK(mfar_search_two_stage(ix, q + (size_t)b * Q * E, Q, W, 1, mask, K, K, 1, ids + (size_t)b * Q * K, sc + (size_t)b * Q * K,
                                    nv + (size_t)b * Q, NULL, NULL, NULL, 0, NULL));
    for (int i = 0; i < NB * Q; ++i)
        if (nv[i] != K) {
            fprintf(stderr, "n_valid[%d] = %d\n", i, nv[i]);
            return 1;
        }

    /* 2. the batch pipeline, host pointers, results `lag` batches late */
    mfar_pipeline* pl = NULL;
    CHECK(mfar_pipeline_create(&pl, ix, W, 1, mask, K, K, 1, Q, 0, 0, 0));
    int depth = 0, coalesce = 0, qpl = 0, lag = 0;
    CHECK(mfar_pipeline_info(pl, &depth, &coalesce, &qpl, &lag, NULL));
    int64_t* tickets = malloc((size_t)NB * 8);
    int taken = 0;
    for (int b = 0; b < NB; ++b) {
        CHECK(mfar_pipeline_submit(pl, q + (size_t)b * Q * E, Q, 0, NULL, &tickets[b]));
        if (b >= lag) {
            CHECK(mfar_pipeline_result(pl, tickets[taken], ids2 + (size_t)taken * Q * K, sc2 + (size_t)taken * Q * K, NULL, 0, NULL));
            ++taken;
        }
    }
    for (; taken < NB; ++taken)
        CHECK(mfar_pipeline_result(pl, tickets[taken], ids2 + (size_t)taken * Q * K, sc2 + (size_t)taken * Q * K, NULL, 0, NULL));
    int64_t n_redone = -1;
    CHECK(mfar_pipeline_info(pl, NULL, NULL, NULL, NULL, &n_redone));
    mfar_pipeline_destroy(pl);
    if (memcmp(ids, ids2, (size_t)NB * Q * K * 8) || memcmp(sc, sc2, (size_t)NB * Q * K * 4)) {
        fprintf(stderr, "pipeline and synchronous results differ\n");
        return 1;
    }
    mfar_index_destroy(ix);

    FILE* fo = fopen(argv[1], "wb");
    if (!fo) return 1;
    const int32_t hdr[5] = {D, F, E, Q, NB};
    fwrite(hdr, 4, 5, fo);
    fwrite(slab, 4, (size_t)F * D * E, fo);
    fwrite(W, 4, (size_t)E * F, fo);
    fwrite(mask, 4, (size_t)F, fo);
    fwrite(q, 4, (size_t)NB * Q * E, fo);
    fwrite(ids, 8, (size_t)NB * Q * K, fo);
    fwrite(sc, 4, (size_t)NB * Q * K, fo);
    fclose(fo);
    printf("OK depth=%d coalesce=%d queries_per_launch=%d lag=%d redone=%lld\n", depth, coalesce, qpl, lag, (long long)n_redone);
    return 0;
}
