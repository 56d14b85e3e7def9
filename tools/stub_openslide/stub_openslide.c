/* A stand-in libopenslide for tests and the bench (libopenslide itself is not in this image and cannot be fetched).
 *
 * Exports the handful of openslide_* entry points the product's native hook (csrc/openslide_host.cpp) and
 * openslide-python's lowlevel layer bind, with libopenslide's conventions:
 *   - openslide_read_region fills PREMULTIPLIED ARGB, one native-endian uint32 0xAARRGGBB per pixel;
 *   - (x, y) are level-0 coordinates, (w, h) are in pixels of `level`;
 *   - everything outside the level's extent is transparent (all-zero words);
 *   - read_region is thread-safe on one handle; errors are sticky and reported by openslide_get_error.
 * A "slide" is a text file `STUBSLIDE <width> <height> <seed> <alpha_period>`: three levels (downsamples 1, 4, 16), pixels =
 * a counter-based hash of (seed, level, x, y); every alpha_period-th pixel (0 = never) is PARTIALLY transparent
 * (alpha 1..254, colour premultiplied with integer truncation), which is what exercises the un-premultiply arithmetic.
 * Not a decoder and not derived from OpenSlide's sources: interface only.
 *
 *   gcc -O2 -shared -fPIC -o libopenslide.so.1 stub_openslide.c
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct stub_slide {
    int64_t w, h;
    uint32_t seed;
    uint32_t alpha_period;
    const char* error;
} openslide_t;

static const int64_t kDs[3] = {1, 4, 16};

static inline uint32_t mix(uint32_t a) {
    a ^= a >> 16; a *= 0x7feb352dU; a ^= a >> 15; a *= 0x846ca68bU; a ^= a >> 16;
    return a;
}

openslide_t* openslide_open(const char* path) {
    FILE* f = fopen(path, "r");
    if (!f) return NULL;
    char tag[16] = "";
    long long w = 0, h = 0;
    unsigned seed = 0, period = 0;
    int got = fscanf(f, "%15s %lld %lld %u %u", tag, &w, &h, &seed, &period);
    fclose(f);
    if (got != 5 || strcmp(tag, "STUBSLIDE") != 0 || w <= 0 || h <= 0) return NULL;
    openslide_t* s = (openslide_t*)calloc(1, sizeof(openslide_t));
    s->w = w; s->h = h; s->seed = seed; s->alpha_period = period;
    return s;
}

void openslide_close(openslide_t* s) { free(s); }
const char* openslide_get_error(openslide_t* s) { return s ? s->error : "null handle"; }
int32_t openslide_get_level_count(openslide_t* s) { (void)s; return 3; }
double openslide_get_level_downsample(openslide_t* s, int32_t level) { (void)s; return level >= 0 && level < 3 ? (double)kDs[level] : -1.0; }

void openslide_get_level_dimensions(openslide_t* s, int32_t level, int64_t* w, int64_t* h) {
    if (level < 0 || level >= 3) { *w = *h = -1; return; }
    *w = s->w / kDs[level];
    *h = s->h / kDs[level];
}

void openslide_get_level0_dimensions(openslide_t* s, int64_t* w, int64_t* h) { *w = s->w; *h = s->h; }

void openslide_read_region(openslide_t* s, uint32_t* dest, int64_t x, int64_t y, int32_t level, int64_t w, int64_t h) {
    if (!dest || w <= 0 || h <= 0) return;
    if (level < 0 || level >= 3) {                 /* libopenslide: an invalid level gives a cleared buffer, no error */
        memset(dest, 0, (size_t)w * h * 4);
        return;
    }
    const int64_t ds = kDs[level], lw = s->w / ds, lh = s->h / ds;
    const int64_t x0 = x / ds, y0 = y / ds;          /* floor for the non-negative coordinates the path produces */
    const uint32_t base = mix(s->seed * 0x9E3779B9U + (uint32_t)level * 0x85EBCA6BU);
    for (int64_t j = 0; j < h; ++j) {
        const int64_t ly = y0 + j;
        uint32_t* row = dest + j * w;
        if (ly < 0 || ly >= lh) { memset(row, 0, (size_t)w * 4); continue; }
        const uint32_t rowkey = mix(base ^ (uint32_t)ly * 0xC2B2AE35U);
        for (int64_t i = 0; i < w; ++i) {
            const int64_t lx = x0 + i;
            if (lx < 0 || lx >= lw) { row[i] = 0; continue; }
            const uint32_t v = mix(rowkey + (uint32_t)lx * 0x27D4EB2FU);
            uint32_t r = v & 0xff, g = (v >> 8) & 0xff, b = (v >> 16) & 0xff, a = 255;
            if (s->alpha_period && (v >> 24) % s->alpha_period == 0 && ((lx + ly) & 3) == 0) {
                a = 1 + (v >> 9) % 254;               /* 1 .. 254 */
                r = r * a / 255; g = g * a / 255; b = b * a / 255;      /* premultiplied: every channel <= alpha */
            }
            row[i] = a << 24 | r << 16 | g << 8 | b;
        }
    }
}
