// Native batched OpenSlide tile read for the tile ring's host threads (no device work).
//
// The reference reads one tile per H5 row in the interpreter (services/feature_embedding.py:86-95 ->
// core/wsi/openslide_wsi.py:184-205): openslide-python's read_region (ctypes call + ARGB -> RGBA conversion + PIL image)
// followed by .convert("RGB") and np.array -- per-tile Python work that caps a CPython thread pool near 10 k tiles/s.
// The ring's decode threads call ap_host_openslide_read_tiles once per CHUNK of rows instead: n regions are read by
// libopenslide itself (openslide_read_region, thread-safe on one handle) and converted straight into consecutive RGB
// slots of the pinned staging buffer, all outside the interpreter lock.
//
// Pixels are exactly what openslide-python + PIL give:
//   libopenslide fills premultiplied ARGB, one native-endian uint32 0xAARRGGBB per pixel;
//   openslide-python (_convert.c, argb2rgba) un-premultiplies:  a == 0 -> pixel left as it is (0 for a premultiplied
//     buffer), a == 255 -> channels as stored, else c' = (uint8)(255 * c / a) (integer division);
//   PIL's RGBA -> RGB conversion drops the alpha byte (no compositing), so transparent padding outside the slide is black
//     (SURVEY.md 9.4).
//
// libopenslide is not part of this image: it is resolved at first use with dlopen (ATLASPATCH_LIBOPENSLIDE = explicit path,
// else libopenslide.so.1 / .so.0 / .so); without it every entry point returns AP_ERR_UNSUPPORTED and the backend keeps
// reading tile by tile through openslide-python.
#include <dlfcn.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>
#include "ap_common.h"

namespace {

struct Api {
    void* (*open)(const char*);
    void (*close)(void*);
    const char* (*get_error)(void*);
    void (*read_region)(void*, uint32_t*, int64_t, int64_t, int32_t, int64_t, int64_t);
    int32_t (*level_count)(void*);
    void (*level_dimensions)(void*, int32_t, int64_t*, int64_t*);
    bool ok = false;
    char why[256] = "";
    char name[256] = "";
};

Api g_api;
std::once_flag g_once;

void load_api() {
    Api& a = g_api;
    const char* env = getenv("ATLASPATCH_LIBOPENSLIDE");
    const char* names[] = {env && *env ? env : nullptr, "libopenslide.so.1", "libopenslide.so.0", "libopenslide.so"};
    void* h = nullptr;
    for (const char* n : names) {
        if (!n) continue;
        h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (h) { snprintf(a.name, sizeof(a.name), "%s", n); break; }
        if (n == env) {             // an explicit path that does not load is an error, not a reason to try the system's
            snprintf(a.why, sizeof(a.why), "ATLASPATCH_LIBOPENSLIDE=%s does not load (%s)", n, dlerror());
            return;
        }
    }
    if (!h) { snprintf(a.why, sizeof(a.why), "libopenslide not found (tried .so.1, .so.0, .so)"); return; }
    a.open = (decltype(a.open))dlsym(h, "openslide_open");
    a.close = (decltype(a.close))dlsym(h, "openslide_close");
    a.get_error = (decltype(a.get_error))dlsym(h, "openslide_get_error");
    a.read_region = (decltype(a.read_region))dlsym(h, "openslide_read_region");
    a.level_count = (decltype(a.level_count))dlsym(h, "openslide_get_level_count");
    a.level_dimensions = (decltype(a.level_dimensions))dlsym(h, "openslide_get_level_dimensions");
    if (!a.open || !a.close || !a.get_error || !a.read_region || !a.level_count || !a.level_dimensions) {
        snprintf(a.why, sizeof(a.why), "%s lacks a required openslide_* symbol", a.name);
        return;
    }
    a.ok = true;
}

// openslide-python's argb2rgba followed by PIL's RGBA -> RGB: premultiplied ARGB words -> packed RGB bytes
inline void argb_to_rgb(const uint32_t* src, unsigned char* dst, size_t pixels) {
    for (size_t i = 0; i < pixels; ++i) {
        const uint32_t v = src[i];
        const uint32_t a = v >> 24;
        uint32_t r = (v >> 16) & 0xff, g = (v >> 8) & 0xff, b = v & 0xff;
        if (a != 255 && a != 0) {
            r = (255u * r / a) & 0xff;          // (u8) truncation of the quotient, as the C extension stores it
            g = (255u * g / a) & 0xff;
            b = (255u * b / a) & 0xff;
        }
        // a == 0: openslide-python leaves the word untouched; PIL then reads its little-endian bytes (B, G, R, 0) as R, G, B, A.
        // For a genuinely premultiplied buffer R = G = B = 0, i.e. black padding; an inconsistent word keeps that byte order.
        if (a == 0) { const uint32_t t = r; r = b; b = t; }
        dst[3 * i + 0] = (unsigned char)r;
        dst[3 * i + 1] = (unsigned char)g;
        dst[3 * i + 2] = (unsigned char)b;
    }
}

}  // namespace

struct ap_openslide {
    void* osr = nullptr;
    int32_t levels = 0;
};

extern "C" int ap_host_openslide_available(void) {
    std::call_once(g_once, load_api);
    if (!g_api.ok) ap::set_error("ap_host_openslide: %s", g_api.why);
    return g_api.ok ? 1 : 0;
}

extern "C" int ap_host_openslide_open(const char* path, ap_openslide** out) {
    AP_REQUIRE(path && out, "ap_host_openslide_open: null argument");
    *out = nullptr;
    std::call_once(g_once, load_api);
    if (!g_api.ok) {
        ap::set_error("ap_host_openslide_open: %s", g_api.why);
        return AP_ERR_UNSUPPORTED;
    }
    void* osr = g_api.open(path);
    AP_REQUIRE(osr, "ap_host_openslide_open: %s is not a slide libopenslide can open", path);
    if (const char* err = g_api.get_error(osr)) {
        ap::set_error("ap_host_openslide_open: %s: %s", path, err);
        g_api.close(osr);
        return AP_ERR_INVALID;
    }
    ap_openslide* h = new ap_openslide;
    h->osr = osr;
    h->levels = g_api.level_count(osr);
    *out = h;
    return AP_OK;
}

extern "C" int ap_host_openslide_read_tiles(ap_openslide* h, const int64_t* xy, int n, int level, int w, int hgt, void* dst) {
    AP_REQUIRE(h && h->osr && dst && (xy || n == 0) && n >= 0 && w > 0 && hgt > 0, "ap_host_openslide_read_tiles: bad arguments");
    AP_REQUIRE(level >= 0 && level < h->levels, "ap_host_openslide_read_tiles: level %d of %d", level, h->levels);
    static thread_local std::vector<uint32_t> argb;       // one scratch region per decode thread
    const size_t pixels = (size_t)w * hgt;
    argb.resize(pixels);
    for (int i = 0; i < n; ++i) {
        g_api.read_region(h->osr, argb.data(), xy[2 * i], xy[2 * i + 1], level, w, hgt);
        if (const char* err = g_api.get_error(h->osr)) {            // sticky: the handle is unusable from here on
            ap::set_error("ap_host_openslide_read_tiles: region (%lld, %lld) level %d: %s", (long long)xy[2 * i],
                          (long long)xy[2 * i + 1], level, err);
            return AP_ERR_INVALID;
        }
        argb_to_rgb(argb.data(), (unsigned char*)dst + (size_t)i * pixels * 3, pixels);
    }
    return AP_OK;
}

extern "C" void ap_host_openslide_close(ap_openslide* h) {
    if (!h) return;
    if (h->osr && g_api.ok) g_api.close(h->osr);
    delete h;
}
