/* Pure-C host of the drop-in boundary (include/apsu_he.h): no Python, no torch.
 *   gcc -O2 -Iinclude examples/c_abi_demo.c -Lapsu_amd -lapsu_he_gpu -Wl,-rpath,$PWD/apsu_amd -o c_abi_demo
 *   ./c_abi_demo tests/params/1M-1024-com.json
 * Runs the tier-1 NTT round trip, the tier-2 path (ComputePowers + eval_bundles on a synthetic BinBundle with a
 * mask drawn by apsu_he_mask_generate_blake2xb, the reference's generator), the multi-device handle (apsu_he_multi_* / apsu_he_eval_all on devices {0, 0}:
 * two engines on one GPU) and the N3 framing of a ResultPackage, and prints FNV-1a checksums of every result.  tests/test_gpu_c_host.py builds
 * it, runs it, and compares the checksums with the same calls made through the Python binding. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <hip/hip_runtime_api.h>
#include "apsu_he.h"

#define CHECK(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, apsu_he_last_error()); return 1; } } while (0)

static uint64_t fnv(const void *p, size_t bytes)
{
    const unsigned char *b = (const unsigned char *)p;
    uint64_t h = 0xcbf29ce484222325ULL;
    for (size_t i = 0; i < bytes; i++) { h ^= b[i]; h *= 0x100000001b3ULL; }
    return h;
}

static uint64_t mix(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s PSUParams.json\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    char *json = (char *)calloc(1, 1 << 16);
    if (fread(json, 1, (1 << 16) - 1, f) == 0) return 2;
    fclose(f);

    apsu_he_ctx *ctx = NULL;
    CHECK(apsu_he_create(json, 0, &ctx));
    apsu_he_info info;
    CHECK(apsu_he_get_info(ctx, &info));
    const size_t n = info.poly_modulus_degree;
    const int K = info.coeff_modulus_size, first = info.first_chain_idx, Lf = first + 1;
    printf("abi %d n %zu K %d first %d sources %u targets %u\n", apsu_he_abi_version(), n, K, first, info.source_power_count,
           info.target_power_count);

    /* tier 1: transform_to_ntt / from_ntt round trip on a pseudo-random size-2 ciphertext */
    uint64_t *ct = (uint64_t *)malloc(2 * Lf * n * 8), *ct0 = (uint64_t *)malloc(2 * Lf * n * 8);
    for (int p = 0; p < 2; p++)
        for (int j = 0; j < Lf; j++)
            for (size_t k = 0; k < n; k++) ct[((size_t)p * Lf + j) * n + k] = mix(1 + ((size_t)p * Lf + j) * n + k) % info.coeff_modulus[j];
    memcpy(ct0, ct, 2 * Lf * n * 8);
    CHECK(apsu_he_transform_to_ntt(ctx, ct, 2, first));
    printf("ntt %016llx\n", (unsigned long long)fnv(ct, 2 * Lf * n * 8));
    CHECK(apsu_he_transform_from_ntt(ctx, ct, 2, first));
    printf("roundtrip %s\n", memcmp(ct, ct0, 2 * Lf * n * 8) == 0 ? "ok" : "MISMATCH");

    /* tier 2: query powers for bundle index 0, one synthetic BinBundle, mask generated on the device */
    const uint32_t ns = info.source_power_count;
    uint64_t *src = (uint64_t *)malloc((size_t)ns * 2 * Lf * n * 8);
    const uint64_t **srcp = (const uint64_t **)malloc(ns * sizeof(*srcp));
    for (uint32_t s = 0; s < ns; s++) {
        for (int p = 0; p < 2; p++)
            for (int j = 0; j < Lf; j++)
                for (size_t k = 0; k < n; k++)
                    src[(((size_t)s * 2 + p) * Lf + j) * n + k] = mix(77 + (((size_t)s * 2 + p) * Lf + j) * n + k) % info.coeff_modulus[j];
        srcp[s] = src + (size_t)s * 2 * Lf * n;
    }
    apsu_he_relin *rk = NULL;
    if (info.using_keyswitching) {
        size_t words = (size_t)(K - 1) * 2 * K * n;
        uint64_t *ksk = (uint64_t *)malloc(words * 8);
        for (int d = 0; d < K - 1; d++)
            for (int c = 0; c < 2; c++)
                for (int j = 0; j < K; j++)
                    for (size_t k = 0; k < n; k++) {
                        size_t i = (((size_t)d * 2 + c) * K + j) * n + k;
                        ksk[i] = mix(1000003 + i) % info.coeff_modulus[j];
                    }
        CHECK(apsu_he_relin_upload(ctx, ksk, &rk));
        free(ksk);
    }
    uint32_t idx = 0;
    apsu_he_powers *pw = NULL;
    CHECK(apsu_he_compute_powers(ctx, &idx, 1, srcp, 0, rk, &pw));
    apsu_he_bundle *bundle = NULL;
    CHECK(apsu_he_db_random_bundle(ctx, 0, 0, info.max_items_per_bin - 1, 4242, &bundle));
    uint64_t *mask_dev = NULL;
    if (hipMalloc((void **)&mask_dev, n * 8) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    uint64_t *blocks = (uint64_t *)malloc((size_t)info.items_per_bundle * 2 * 8);
    uint64_t prng_seed[8];                                      /* seal::prng_seed_type; the reference fills it with random_bytes */
    for (int i = 0; i < 8; i++) prng_seed[i] = mix(99 + i);
    CHECK(apsu_he_mask_generate_blake2xb(ctx, prng_seed, 0, 1, mask_dev, NULL, blocks));
    printf("blocks %016llx\n", (unsigned long long)fnv(blocks, (size_t)info.items_per_bundle * 16));
    uint64_t *out = (uint64_t *)malloc(2 * n * 8);
    const apsu_he_bundle *bl[1] = { bundle };
    const uint64_t *ml[1] = { mask_dev };
    CHECK(apsu_he_eval_bundles(ctx, bl, 1, pw, rk, ml, 1, out, 0));
    printf("result %016llx\n", (unsigned long long)fnv(out, 2 * n * 8));

    /* several devices behind one handle: the same BinBundle on slot 1 of {0, 0}, every bundle index's sources on the host */
    {
        int devices[2] = { 0, 0 }, id = -1, slot[1];
        uint32_t bi[1] = { 0 }, ci[1] = { 0 }, dg[1] = { info.max_items_per_bin - 1 };
        CHECK(apsu_he_partition_bundles(info.bundle_idx_count, 2, bi, ci, dg, 1, slot));
        apsu_he_multi *m = NULL;
        CHECK(apsu_he_multi_create(json, devices, 2, &m));
        if (info.using_keyswitching) {
            size_t words = (size_t)(K - 1) * 2 * K * n;
            uint64_t *ksk = (uint64_t *)malloc(words * 8);
            for (size_t i = 0; i < words; i++) ksk[i] = mix(1000003 + i) % info.coeff_modulus[(i / n) % K];
            CHECK(apsu_he_multi_relin_upload(m, ksk));
            free(ksk);
        }
        CHECK(apsu_he_multi_db_random_bundle(m, 1 - slot[0], 0, 0, info.max_items_per_bin - 1, 4242, &id));
        const uint64_t **all_src = (const uint64_t **)malloc((size_t)info.bundle_idx_count * ns * sizeof(*all_src));
        for (uint32_t b = 0; b < info.bundle_idx_count; b++)
            for (uint32_t s2 = 0; s2 < ns; s2++) all_src[(size_t)b * ns + s2] = srcp[s2];
        uint64_t *mask_host = (uint64_t *)malloc(n * 8), *out2 = (uint64_t *)malloc(2 * n * 8);
        if (hipMemcpy(mask_host, mask_dev, n * 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        const uint64_t *mh[1] = { mask_host };
        CHECK(apsu_he_eval_all(m, all_src, mh, out2, -1));
        printf("multi %s (bundle id %d)\n", memcmp(out, out2, 2 * n * 8) == 0 ? "ok" : "MISMATCH", id);
        /* the same with page-locked query buffers (read and written in place by the kernels) and the reference's phase timers */
        {
            const size_t ctw = (size_t)2 * (first + 1) * n;
            uint64_t *pin_src = NULL, *pin_mask = NULL, *pin_out = NULL;
            CHECK(apsu_he_host_alloc((size_t)ns * ctw * 8, (void **)&pin_src));
            CHECK(apsu_he_host_alloc(n * 8, (void **)&pin_mask));
            CHECK(apsu_he_host_alloc(2 * n * 8, (void **)&pin_out));
            for (uint32_t s2 = 0; s2 < ns; s2++) memcpy(pin_src + (size_t)s2 * ctw, srcp[s2], ctw * 8);
            memcpy(pin_mask, mask_host, n * 8);
            for (uint32_t b = 0; b < info.bundle_idx_count; b++)
                for (uint32_t s2 = 0; s2 < ns; s2++) all_src[(size_t)b * ns + s2] = pin_src + (size_t)s2 * ctw;
            const uint64_t *pm[1] = { pin_mask };
            CHECK(apsu_he_multi_phase_enable(m, 1));
            CHECK(apsu_he_eval_all_ex(m, all_src, pm, pin_out, -1, APSU_HE_IO_SRC_PINNED | APSU_HE_IO_MASKS_PINNED | APSU_HE_IO_OUT_PINNED, 0));
            uint64_t cnt[APSU_HE_PHASES]; double avg[APSU_HE_PHASES];
            CHECK(apsu_he_multi_phase_read(m, cnt, avg, NULL, NULL, 1));
            printf("pinned %s (%s %.3f ms, %s %.3f ms, %s %.3f ms)\n", memcmp(out, pin_out, 2 * n * 8) == 0 && cnt[0] == 1 ? "ok" : "MISMATCH",
                   apsu_he_phase_name(0), avg[0], apsu_he_phase_name(1), avg[1], apsu_he_phase_name(2), avg[2]);
            CHECK(apsu_he_host_free(pin_src)); CHECK(apsu_he_host_free(pin_mask)); CHECK(apsu_he_host_free(pin_out));
        }
        CHECK(apsu_he_multi_destroy(m));
        free(all_src); free(mask_host); free(out2);
    }

    /* N3: the result as a ResultPackage message (result_package.fbs), parsed back */
    {
        uint8_t *msg = NULL; size_t msg_n = 0;
        CHECK(apsu_he_wire_build_result_package(0, 3, (const uint8_t *)out, 2 * n * 8, 0, 0, 0, NULL, NULL, &msg, &msg_n));
        uint32_t b2 = 9, c2 = 9, nl = 9; const uint8_t *blob = NULL; size_t blob_n = 0;
        CHECK(apsu_he_wire_parse_result_package(msg, msg_n, &b2, &c2, &blob, &blob_n, NULL, NULL, &nl));
        printf("wire %s (%zu bytes)\n", (b2 == 0 && c2 == 3 && nl == 0 && blob_n == 2 * n * 8 && memcmp(blob, out, blob_n) == 0) ? "ok" : "MISMATCH", msg_n);
        CHECK(apsu_he_wire_buffer_free(msg));
    }

    /* N3: the result as SEAL would serialise it at the last level (zlib body), loaded back; parms_id of that level */
    {
        apsu_he_seal_ctx *sc = NULL;
        CHECK(apsu_he_seal_ctx_create(json, &sc));
        uint8_t *blob = NULL; size_t blob_n = 0, used = 0;
        CHECK(apsu_he_seal_ct_save(sc, 0, 0, 2, out, NULL, APSU_HE_SEAL_COMPR_ZLIB, 4, 0, &blob, &blob_n));
        uint64_t pid[4], want[4], *back = (uint64_t *)malloc(2 * n * 8);
        int ci = -9, seeded = -9;
        CHECK(apsu_he_seal_ct_load(sc, blob, blob_n, pid, &ci, NULL, NULL, NULL, NULL, &seeded, back, 2 * n, &used));
        CHECK(apsu_he_seal_parms_id(sc, 0, want));
        printf("seal %s (%zu -> %zu bytes)\n", (ci == 0 && !seeded && used == blob_n && !memcmp(pid, want, 32) && !memcmp(back, out, 2 * n * 8)) ? "ok" : "MISMATCH",
               2 * n * 8, blob_n);
        free(back);
        CHECK(apsu_he_wire_buffer_free(blob));
        CHECK(apsu_he_seal_ctx_free(sc));
    }

    /* N2: the database (here: the one BinBundle) into a file, mapped and loaded back -- the same query result; N1: an item's field elements */
    {
        char path[64];
        snprintf(path, sizeof(path), "/tmp/apsu_he_demo_%d.db", (int)getpid());
        CHECK(apsu_he_db_file_save(ctx, path, bl, 1));
        apsu_he_db_file *f = NULL;
        CHECK(apsu_he_db_file_open(path, &f));
        int cnt = 0; uint64_t fbytes = 0; uint32_t fb = 9, fc = 9, fd = 0;
        CHECK(apsu_he_db_file_count(f, &cnt, &fbytes));
        CHECK(apsu_he_db_file_entry(f, 0, &fb, &fc, &fd, NULL));
        apsu_he_bundle *again = NULL;
        CHECK(apsu_he_db_file_load(ctx, f, 0, &again));
        CHECK(apsu_he_db_file_close(f));
        remove(path);
        uint64_t *out3 = (uint64_t *)malloc(2 * n * 8);
        const apsu_he_bundle *bl3[1] = { again };
        CHECK(apsu_he_eval_bundles(ctx, bl3, 1, pw, rk, ml, 1, out3, 0));
        printf("dbfile %s (%d BinBundle, %llu bytes)\n", (cnt == 1 && fb == 0 && fc == 0 && fd == info.max_items_per_bin - 1 && !memcmp(out, out3, 2 * n * 8)) ? "ok" : "MISMATCH",
               cnt, (unsigned long long)fbytes);
        free(out3);
        CHECK(apsu_he_bundle_free(again));
        uint8_t item[16];
        for (int i = 0; i < 16; i++) item[i] = (uint8_t)(0x10 * (i % 8) + 0x0f - i);
        uint64_t felts[32];
        CHECK(apsu_he_algebraize_items(ctx, item, 1, 0, felts, 0));
        printf("felts %016llx\n", (unsigned long long)fnv(felts, 2 * 8));
    }

    /* error behaviour: too few powers for a bundle index that was not computed */
    apsu_he_bundle *other = NULL;
    CHECK(apsu_he_db_random_bundle(ctx, 1, 0, 3, 1, &other));
    const apsu_he_bundle *bl2[1] = { other };
    int rc = apsu_he_eval_bundles(ctx, bl2, 1, pw, rk, ml, 1, out, 0);
    printf("missing-powers status %d (%s)\n", rc, rc == APSU_HE_INVALID_ARGUMENT ? "invalid_argument" : "unexpected");

    (void)hipFree(mask_dev);
    CHECK(apsu_he_bundle_free(other));
    CHECK(apsu_he_bundle_free(bundle));
    CHECK(apsu_he_powers_free(pw));
    if (rk) CHECK(apsu_he_relin_free(rk));
    CHECK(apsu_he_destroy(ctx));
    printf("done\n");
    return 0;
}
