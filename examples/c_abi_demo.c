/* The drop-in boundary driven from plain C -- no Python, no torch: what a maintainer binding the library from another host
 * language would write.  Build (any C compiler; the HIP runtime is used only for device memory):
 *
 *   gcc -std=c99 -O2 -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ examples/c_abi_demo.c \
 *       -Latlaspatch_amd -latlaspatch_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/atlaspatch_amd -Wl,-rpath,/opt/rocm/lib \
 *       -o examples/c_abi_demo
 *
 * 1. coordinates (reference utils/contours.py:41-131 + services/extraction.py:67-128): a synthetic 64 x 64 tissue mask with a
 *    hole -> contours -> grid rows of a 16 000-px slide, 256-px patches.
 * 2. encoder (reference models/patch/base.py:76-107 on vit_b_16, here 2 blocks deep): seeded parameters and seeded uint8 tiles
 *    -> float32 features.  tests/test_c_abi_demo.py feeds the same seeds through the Python host side and expects the same bits.
 * Prints "rows <n> first <x> <y>" and "feat <n> <dim> <sum> <f[0][0]> <f[n-1][dim-1]>" (hex floats).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "atlaspatch_hip.h"

#define CHECK(call)                                                                       \
    do {                                                                                  \
        int rc_ = (call);                                                                 \
        if (rc_ != 0) {                                                                   \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, ap_last_error());         \
            return 1;                                                                     \
        }                                                                                 \
    } while (0)
#define HIPCHECK(call)                                                                    \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e_));             \
            return 1;                                                                     \
        }                                                                                 \
    } while (0)

/* 32-bit LCG (Numerical Recipes constants): the test regenerates the same stream in numpy */
static uint32_t lcg_state;
static uint32_t lcg(void) { lcg_state = lcg_state * 1664525u + 1013904223u; return lcg_state; }
static float lcg_unit(void) { return (float)(lcg() >> 8) * (1.0f / 16777216.0f) - 0.5f; }       /* [-0.5, 0.5) */

static int set_param(ap_vit* m, const char* name, size_t count, float scale, float offset) {
    float* buf = (float*)malloc(count * sizeof(float));
    size_t i;
    int rc;
    if (!buf) return 1;
    for (i = 0; i < count; ++i) buf[i] = offset + scale * lcg_unit();
    rc = ap_vit_set_param(m, name, buf, count);
    free(buf);
    if (rc != 0) fprintf(stderr, "ap_vit_set_param(%s): %s\n", name, ap_last_error());
    return rc;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 5, depth = 2, dim = 768, mlp = 3072, tokens = 197;
    char name[128];
    int cus = 0;
    size_t hbm = 0;
    if (ap_abi_version() != AP_ABI_VERSION) { fprintf(stderr, "ABI %d, header is %d\n", ap_abi_version(), AP_ABI_VERSION); return 1; }
    CHECK(ap_device_info(0, name, (int)sizeof(name), &cus, &hbm));
    fprintf(stderr, "device 0: %s, %d CUs, %.0f GB\n", name, cus, (double)hbm / 1e9);

    /* ---- 1. coordinates */
    {
        enum { S = 64 };
        static float mask[S * S];
        ap_contours* c = NULL;
        int32_t* rows;
        size_t n_rows = 0, cap = 8192;
        int y, x;
        for (y = 0; y < S; ++y)
            for (x = 0; x < S; ++x) {
                const int in = (x - 30) * (x - 30) + (y - 34) * (y - 34) <= 24 * 24;
                const int hole = (x - 36) * (x - 36) + (y - 30) * (y - 30) <= 7 * 7;
                mask[y * S + x] = (in && !hole) ? 1.0f : 0.0f;
            }
        CHECK(ap_contours_from_mask(mask, S, S, 0.0, 16, 10, 16000.0 / S, 16000.0 / S, &c, NULL));
        rows = (int32_t*)malloc(cap * 5 * sizeof(int32_t));
        CHECK(ap_grid_coords(c, 256, 256, 256, 256, 0, rows, cap, &n_rows, NULL));
        printf("rows %zu first %d %d contours %d holes %d\n", n_rows, n_rows ? rows[0] : -1, n_rows ? rows[1] : -1,
               ap_contours_count(c), ap_contours_count(c) ? ap_contours_num_holes(c, 0) : 0);
        free(rows);
        ap_contours_destroy(c);
    }

    /* ---- 2. encoder */
    {
        ap_vit_config cfg;
        ap_vit* m = NULL;
        const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
        uint8_t *tiles_h, *tiles_d = NULL;
        float *out_d = NULL, *out_h;
        void* ws = NULL;
        size_t ws_bytes, tile_bytes = (size_t)n * 256 * 256 * 3, i;
        double sum = 0.0;
        int b;
        char key[96];
        CHECK(ap_vit_config_init(&cfg, sizeof(cfg)));
        cfg.image_size = 224; cfg.patch_size = 16; cfg.dim = dim; cfg.depth = depth; cfg.heads = 12; cfg.mlp_dim = mlp;
        cfg.ln_eps = 1e-6f; cfg.compute_dtype = AP_F16; cfg.pool = AP_POOL_CLS; cfg.pool_ln_eps = 1e-5f;
        CHECK(ap_vit_create(&cfg, &m));
        lcg_state = 12345u;
        if (set_param(m, "patch_embed.weight", (size_t)dim * 3 * 16 * 16, 0.08f, 0.f) || set_param(m, "patch_embed.bias", dim, 0.04f, 0.f) ||
            set_param(m, "cls_token", dim, 0.04f, 0.f) || set_param(m, "pos_embed", (size_t)tokens * dim, 0.04f, 0.f) ||
            set_param(m, "norm.weight", dim, 0.2f, 1.f) || set_param(m, "norm.bias", dim, 0.04f, 0.f)) return 1;
        for (b = 0; b < depth; ++b) {
#define P(suffix, count, scale, offset)                                   \
    do {                                                                  \
        snprintf(key, sizeof(key), "blocks.%d." suffix, b);               \
        if (set_param(m, key, (count), (scale), (offset))) return 1;      \
    } while (0)
            P("ln1.weight", dim, 0.2f, 1.f); P("ln1.bias", dim, 0.04f, 0.f);
            P("qkv.weight", (size_t)3 * dim * dim, 0.08f, 0.f); P("qkv.bias", 3 * dim, 0.04f, 0.f);
            P("proj.weight", (size_t)dim * dim, 0.08f, 0.f); P("proj.bias", dim, 0.04f, 0.f);
            P("ln2.weight", dim, 0.2f, 1.f); P("ln2.bias", dim, 0.04f, 0.f);
            P("fc1.weight", (size_t)mlp * dim, 0.08f, 0.f); P("fc1.bias", mlp, 0.04f, 0.f);
            P("fc2.weight", (size_t)dim * mlp, 0.08f, 0.f); P("fc2.bias", dim, 0.04f, 0.f);
#undef P
        }
        CHECK(ap_vit_finalize(m));
        tiles_h = (uint8_t*)malloc(tile_bytes);
        lcg_state = 777u;
        for (i = 0; i < tile_bytes; ++i) tiles_h[i] = (uint8_t)(lcg() >> 24);
        HIPCHECK(hipMalloc((void**)&tiles_d, tile_bytes));
        HIPCHECK(hipMemcpy(tiles_d, tiles_h, tile_bytes, hipMemcpyHostToDevice));
        ws_bytes = ap_vit_workspace_bytes(m, n);
        HIPCHECK(hipMalloc(&ws, ws_bytes));
        HIPCHECK(hipMalloc((void**)&out_d, (size_t)n * dim * sizeof(float)));
        CHECK(ap_vit_forward_u8(m, tiles_d, n, 256, 256, mean, stdv, out_d, ws, ws_bytes, NULL));
        HIPCHECK(hipDeviceSynchronize());
        out_h = (float*)malloc((size_t)n * dim * sizeof(float));
        HIPCHECK(hipMemcpy(out_h, out_d, (size_t)n * dim * sizeof(float), hipMemcpyDeviceToHost));
        for (i = 0; i < (size_t)n * dim; ++i) sum += out_h[i];
        printf("feat %d %d %a %a %a\n", n, ap_vit_embed_dim(m), sum, out_h[0], out_h[(size_t)n * dim - 1]);
        if (argc > 2) {                                   /* raw float32 dump for the test */
            FILE* f = fopen(argv[2], "wb");
            if (!f || fwrite(out_h, sizeof(float), (size_t)n * dim, f) != (size_t)n * dim) { fprintf(stderr, "cannot write %s\n", argv[2]); return 1; }
            fclose(f);
        }
        free(out_h); free(tiles_h);
        (void)hipFree(out_d); (void)hipFree(ws); (void)hipFree(tiles_d);
        ap_vit_destroy(m);
    }
    return 0;
}
