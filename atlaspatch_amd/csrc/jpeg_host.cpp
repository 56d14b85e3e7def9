// Native batched JPEG tile decode for the tile ring's host threads (no device work).
//
// The reference reads one tile per H5 row in the interpreter (services/feature_embedding.py:86-95 ->
// core/wsi/openslide_wsi.py:184-205: read_region(...).convert("RGB")); a real slide's tiles are JPEG streams.  The
// ring's decode threads call ap_host_decode_jpeg_tiles once per chunk: read + decode n tile files straight into
// consecutive slots of the pinned staging buffer, outside the interpreter lock.
//
// Decoder: the system's libjpeg-turbo (libjpeg.so.8, the jpeg8 ABI build Ubuntu ships; Pillow's own decoder is the
// same code base, so the pixels are the ones PIL.Image.open(...).convert("RGB") yields -- tested bit for bit).  The
// image has the shared object but no development headers, so the handful of declarations needed from <jpeglib.h>
// (JPEG_LIB_VERSION 80 layout) are restated below; the library itself validates them: jpeg_CreateDecompress
// rejects a caller whose sizeof(jpeg_decompress_struct) differs from its own, and the first use runs a self-check.
// If the library is missing or disagrees the entry point returns AP_ERR_UNSUPPORTED and the backend keeps reading
// tile by tile through Pillow.
#include <dlfcn.h>
#include <csetjmp>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>
#include "ap_common.h"

namespace {

// ---- <jpeglib.h> subset, JPEG_LIB_VERSION 80 (libjpeg-turbo 2.x built --with-jpeg8), LP64 ----------------------------
typedef int boolean_t;
typedef unsigned int JDIMENSION;
typedef unsigned char JSAMPLE;
typedef JSAMPLE* JSAMPROW;
typedef JSAMPROW* JSAMPARRAY;

struct jpeg_error_mgr {
    void (*error_exit)(void* cinfo);
    void (*emit_message)(void* cinfo, int msg_level);
    void (*output_message)(void* cinfo);
    void (*format_message)(void* cinfo, char* buffer);
    void (*reset_error_mgr)(void* cinfo);
    int msg_code;
    union { int i[8]; char s[80]; } msg_parm;
    int trace_level;
    long num_warnings;
    const char* const* jpeg_message_table;
    int last_jpeg_message;
    const char* const* addon_message_table;
    int first_addon_message;
    int last_addon_message;
};

struct jpeg_decompress_struct {
    jpeg_error_mgr* err;                 // jpeg_common_fields
    void* mem;
    void* progress;
    void* client_data;
    boolean_t is_decompressor;
    int global_state;
    void* src;
    JDIMENSION image_width, image_height;
    int num_components;
    int jpeg_color_space;
    int out_color_space;
    unsigned int scale_num, scale_denom;
    double output_gamma;
    boolean_t buffered_image, raw_data_out;
    int dct_method;
    boolean_t do_fancy_upsampling, do_block_smoothing, quantize_colors;
    int dither_mode;
    boolean_t two_pass_quantize;
    int desired_number_of_colors;
    boolean_t enable_1pass_quant, enable_external_quant, enable_2pass_quant;
    JDIMENSION output_width, output_height;
    int out_color_components, output_components, rec_outbuf_height, actual_number_of_colors;
    JSAMPARRAY colormap;
    JDIMENSION output_scanline;
    int input_scan_number;
    JDIMENSION input_iMCU_row;
    int output_scan_number;
    JDIMENSION output_iMCU_row;
    int (*coef_bits)[64];
    void* quant_tbl_ptrs[4];
    void* dc_huff_tbl_ptrs[4];
    void* ac_huff_tbl_ptrs[4];
    int data_precision;
    void* comp_info;
    boolean_t is_baseline;               // >= v8
    boolean_t progressive_mode, arith_code;
    unsigned char arith_dc_L[16], arith_dc_U[16], arith_ac_K[16];
    unsigned int restart_interval;
    boolean_t saw_JFIF_marker;
    unsigned char JFIF_major_version, JFIF_minor_version, density_unit;
    unsigned short X_density, Y_density;
    boolean_t saw_Adobe_marker;
    unsigned char Adobe_transform;
    boolean_t CCIR601_sampling;
    void* marker_list;
    int max_h_samp_factor, max_v_samp_factor;
    int min_DCT_h_scaled_size, min_DCT_v_scaled_size;      // >= v7
    JDIMENSION total_iMCU_rows;
    JSAMPLE* sample_range_limit;
    int comps_in_scan;
    void* cur_comp_info[4];
    JDIMENSION MCUs_per_row, MCU_rows_in_scan;
    int blocks_in_MCU;
    int MCU_membership[10];
    int Ss, Se, Ah, Al;
    int block_size;                      // >= v8
    const int* natural_order;
    int lim_Se;
    int unread_marker;
    void *master, *main_, *coef, *post, *inputctl, *marker, *entropy, *idct, *upsample, *cconvert, *cquantize;
};

constexpr int kJpegLibVersion = 80;
constexpr int JCS_RGB = 2;
constexpr int JPEG_HEADER_OK = 1;

struct Api {
    jpeg_error_mgr* (*std_error)(jpeg_error_mgr*);
    void (*create)(jpeg_decompress_struct*, int, size_t);
    void (*mem_src)(jpeg_decompress_struct*, const unsigned char*, unsigned long);
    int (*read_header)(jpeg_decompress_struct*, boolean_t);
    boolean_t (*start)(jpeg_decompress_struct*);
    JDIMENSION (*read_scanlines)(jpeg_decompress_struct*, JSAMPARRAY, JDIMENSION);
    boolean_t (*finish)(jpeg_decompress_struct*);
    void (*destroy)(jpeg_decompress_struct*);
    bool ok = false;
    char why[200] = "";
};

struct ErrJmp {
    jpeg_error_mgr pub;
    jmp_buf jb;
    char text[200];
};

void on_error(void* cinfo) {
    jpeg_decompress_struct* c = (jpeg_decompress_struct*)cinfo;
    ErrJmp* e = (ErrJmp*)c->err;
    char buf[200] = "";
    if (e->pub.format_message) e->pub.format_message(cinfo, buf);
    snprintf(e->text, sizeof(e->text), "%s", buf);
    longjmp(e->jb, 1);
}
// libjpeg reports recoverable stream damage (premature end of data, bad Huffman code, ...) as WARNINGS (msg_level < 0):
// it injects a fake EOI / grey blocks and carries on.  Pillow raises on such a tile ("image file is truncated"), so a
// damaged tile must fail here too instead of being embedded as grey pixels: count the warnings, check after decoding.
void on_message(void* cinfo, int msg_level) {
    if (msg_level < 0) ((jpeg_decompress_struct*)cinfo)->err->num_warnings++;
}

Api g_api;
std::once_flag g_once;

// decode one in-memory JPEG into dst (rows of `stride` bytes); returns 0 or fills `why`
int decode_one(const Api& api, const unsigned char* data, size_t size, int want_w, int want_h, unsigned char* dst,
               size_t stride, char* why, size_t why_cap) {
    jpeg_decompress_struct c;
    ErrJmp err;
    memset(&c, 0, sizeof(c));
    c.err = api.std_error(&err.pub);
    err.pub.error_exit = on_error;
    err.pub.emit_message = on_message;
    err.text[0] = 0;
    volatile bool created = false;          // written between setjmp and a possible longjmp
    if (setjmp(err.jb)) {
        snprintf(why, why_cap, "libjpeg: %s", err.text);
        if (created) api.destroy(&c);
        return -1;
    }
    api.create(&c, kJpegLibVersion, sizeof(jpeg_decompress_struct));
    created = true;
    api.mem_src(&c, data, (unsigned long)size);
    if (api.read_header(&c, 1) != JPEG_HEADER_OK) { snprintf(why, why_cap, "not a JPEG stream"); api.destroy(&c); return -1; }
    c.out_color_space = JCS_RGB;          // what PIL's convert("RGB") yields for YCbCr and greyscale streams alike
    api.start(&c);
    if ((int)c.output_width != want_w || (int)c.output_height != want_h || c.output_components != 3) {
        snprintf(why, why_cap, "tile is %u x %u x %d, expected %d x %d x 3", c.output_width, c.output_height,
                 c.output_components, want_w, want_h);
        api.destroy(&c);
        return -1;
    }
    while (c.output_scanline < c.output_height) {
        JSAMPROW rows[4];
        const JDIMENSION left = c.output_height - c.output_scanline, nrows = left < 4 ? left : 4;
        for (JDIMENSION r = 0; r < nrows; ++r) rows[r] = dst + (size_t)(c.output_scanline + r) * stride;
        api.read_scanlines(&c, rows, nrows);
    }
    api.finish(&c);
    const long warnings = err.pub.num_warnings;
    api.destroy(&c);
    if (warnings > 0) {
        snprintf(why, why_cap, "libjpeg reported %ld warning(s): truncated or corrupt JPEG stream", warnings);
        return -1;
    }
    return 0;
}

// 8 x 8 grey JFIF (all pixels 128): the load-time self-check of the restated declarations
const unsigned char kProbe[] = {
    0xFF,0xD8,0xFF,0xE0,0x00,0x10,0x4A,0x46,0x49,0x46,0x00,0x01,0x01,0x00,0x00,0x01,0x00,0x01,0x00,0x00,
    0xFF,0xDB,0x00,0x43,0x00,
    1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,
    0xFF,0xC0,0x00,0x0B,0x08,0x00,0x08,0x00,0x08,0x01,0x01,0x11,0x00,
    0xFF,0xC4,0x00,0x14,0x00, 1,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0, 0x00,                 // DC table: one code (length 1) -> category 0
    0xFF,0xC4,0x00,0x14,0x10, 1,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0, 0x00,                 // AC table: one code (length 1) -> EOB
    0xFF,0xDA,0x00,0x08,0x01,0x01,0x00,0x00,0x3F,0x00,
    0x3F,                                                                             // bits 0 (DC diff 0), 0 (EOB), padding 1s
    0xFF,0xD9};

void load_api() {
    Api& a = g_api;
    void* h = dlopen("libjpeg.so.8", RTLD_NOW | RTLD_LOCAL);
    if (!h) { snprintf(a.why, sizeof(a.why), "libjpeg.so.8 not found (%s)", dlerror()); return; }
    auto sym = [&](const char* name) { return dlsym(h, name); };
    a.std_error = (decltype(a.std_error))sym("jpeg_std_error");
    a.create = (decltype(a.create))sym("jpeg_CreateDecompress");
    a.mem_src = (decltype(a.mem_src))sym("jpeg_mem_src");
    a.read_header = (decltype(a.read_header))sym("jpeg_read_header");
    a.start = (decltype(a.start))sym("jpeg_start_decompress");
    a.read_scanlines = (decltype(a.read_scanlines))sym("jpeg_read_scanlines");
    a.finish = (decltype(a.finish))sym("jpeg_finish_decompress");
    a.destroy = (decltype(a.destroy))sym("jpeg_destroy_decompress");
    if (!a.std_error || !a.create || !a.mem_src || !a.read_header || !a.start || !a.read_scanlines || !a.finish || !a.destroy) {
        snprintf(a.why, sizeof(a.why), "libjpeg.so.8 lacks a required symbol");
        return;
    }
    unsigned char px[8 * 8 * 3];
    memset(px, 0, sizeof(px));
    char why[160] = "";
    if (decode_one(a, kProbe, sizeof(kProbe), 8, 8, px, 24, why, sizeof(why)) != 0) {
        snprintf(a.why, sizeof(a.why), "libjpeg.so.8 self-check failed: %s", why);
        return;
    }
    for (unsigned char v : px)
        if (v != 128) { snprintf(a.why, sizeof(a.why), "libjpeg.so.8 self-check decoded %d, expected 128", (int)v); return; }
    a.ok = true;
}

}  // namespace

extern "C" int ap_host_decode_jpeg_tiles(void* dst, const char* const* paths, int n, int side) {
    AP_REQUIRE(dst && (paths || n == 0) && n >= 0 && side > 0, "ap_host_decode_jpeg_tiles: bad arguments");
    std::call_once(g_once, load_api);
    if (!g_api.ok) {
        ap::set_error("ap_host_decode_jpeg_tiles: %s", g_api.why);
        return AP_ERR_UNSUPPORTED;
    }
    std::vector<unsigned char> buf;
    const size_t tile_bytes = (size_t)side * side * 3;
    for (int i = 0; i < n; ++i) {
        AP_REQUIRE(paths[i], "ap_host_decode_jpeg_tiles: null path %d", i);
        FILE* f = fopen(paths[i], "rb");
        AP_REQUIRE(f, "ap_host_decode_jpeg_tiles: cannot open %s", paths[i]);
        fseek(f, 0, SEEK_END);
        const long size = ftell(f);
        fseek(f, 0, SEEK_SET);
        buf.resize(size > 0 ? (size_t)size : 1);
        const size_t got = size > 0 ? fread(buf.data(), 1, (size_t)size, f) : 0;
        fclose(f);
        AP_REQUIRE(size > 0 && got == (size_t)size, "ap_host_decode_jpeg_tiles: short read of %s", paths[i]);
        char why[200] = "";
        if (decode_one(g_api, buf.data(), (size_t)size, side, side, (unsigned char*)dst + (size_t)i * tile_bytes,
                       (size_t)side * 3, why, sizeof(why)) != 0) {
            ap::set_error("ap_host_decode_jpeg_tiles: %s: %s", paths[i], why);
            return AP_ERR_INVALID;
        }
    }
    return AP_OK;
}
