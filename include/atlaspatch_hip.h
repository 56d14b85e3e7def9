/*
 * atlaspatch_hip.h -- C ABI of the MI355X (gfx950) hot path of the AtlasPatch drop-in.
 *
 * One shared library (libatlaspatch_hip.so), plain pointers and sizes, no torch or
 * C++ types in any signature.  Every entry point returns 0 on success or a
 * negative AP_ERR_* code; nothing throws, nothing synchronises the device unless
 * its comment says so, and the caller owns every buffer.  Device pointers are
 * HIP device pointers (what torch.Tensor.data_ptr() returns on ROCm); a stream is
 * a hipStream_t passed as void* (NULL = the default stream).
 *
 * Each block cites the reference interface it replaces (paths under
 * /root/reference/atlas_patch).  INTEGRATION.md shows the ctypes stubs a
 * maintainer of the reference would add.
 */
#ifndef ATLASPATCH_HIP_H
#define ATLASPATCH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AP_OK 0
#define AP_ERR_INVALID (-1)      /* bad argument (null pointer, size, unsupported shape) */
#define AP_ERR_HIP (-2)          /* a HIP runtime call failed; see ap_last_error() */
#define AP_ERR_WORKSPACE (-3)    /* workspace too small */
#define AP_ERR_UNSUPPORTED (-4)  /* valid request this build has no kernel for */
#define AP_ERR_STATE (-5)        /* object not finalised / already finalised */
#define AP_ERR_CAPACITY (-6)     /* output buffer too small; needed count is reported */

/* element types of device tensors */
#define AP_F32 0
#define AP_F16 1
#define AP_BF16 2

typedef void* ap_stream_t;

/* ---- library ---------------------------------------------------------------------- */
#define AP_ABI_VERSION 20                 /* what THIS header declares; compare with ap_abi_version() before any other call */
int ap_abi_version(void);                 /* bumps when a signature or a struct layout changes */
const char* ap_last_error(void);          /* thread-local text of the last failure */
int ap_device_info(int device, char* name, int name_cap, int* cu_count, size_t* hbm_bytes);

/* ---- K1: patch preprocess ----------------------------------------------------------
 * Replaces the per-item CPU transform + collate + H2D + cast of
 * models/patch/base.py:42-45,84-99 (PatchDataset.__getitem__ + DataLoader + .to()).
 * y = ((float(x) / 255) - mean[c]) / std[c] with two true divisions and no FMA
 * contraction, i.e. bit-identical in f32 to torchvision's ToTensor + Normalize;
 * the cast to f16/bf16 happens after normalisation (base.py:95-99).
 *
 * src: device uint8 [n, h, w, 3] (HWC RGB).  Output window: rows crop_top..crop_top+oh,
 * cols crop_left..crop_left+ow (centre crop of torchvision ImageClassification; no resample). */
int ap_preproc_u8hwc_to_chw(const uint8_t* src, int n, int h, int w,
                            int crop_top, int crop_left, int oh, int ow,
                            const float mean[3], const float stdv[3],
                            void* dst /* [n,3,oh,ow] */, int dst_dtype, ap_stream_t stream);

/* Same arithmetic, output laid out for the patch-embed GEMM: dst[(i*gh+py)*gw+px][c*ps*ps+ky*ps+kx]
 * with gh = oh/ps, gw = ow/ps, row length ld (>= 3*ps*ps, zero padded). */
int ap_preproc_u8hwc_to_patchrows(const uint8_t* src, int n, int h, int w,
                                  int crop_top, int crop_left, int oh, int ow, int ps,
                                  const float mean[3], const float stdv[3],
                                  void* dst, int ld, int dst_dtype, ap_stream_t stream);

/* ---- host side of the tile ring ----------------------------------------------------------
 * Copies n decoded tiles (host pointers, bytes_each bytes each) into consecutive slots of a pinned staging buffer.
 * Plain memcpy outside the interpreter lock: the decode threads of the ring that replaces the serial tile loop of
 * services/feature_embedding.py:81-96 call it once per chunk, so an in-memory tile source is not serialised by
 * per-tile NumPy slice assignments.  Host only, no device work. */
int ap_host_gather_tiles(void* dst, const void* const* src, int n, size_t bytes_each);
/* The same for tiles stored as zlib-deflated raw RGB files (the synthetic slide's compressed tile store): reads and
 * inflates n files straight into consecutive slots, one call per chunk -- the shape a native reader for a real slide
 * format takes (the reference's per-tile Python read, services/feature_embedding.py:86-95, cannot leave the interpreter). */
int ap_host_inflate_tiles(void* dst, const char* const* paths, int n, size_t bytes_each);

/* Passport strings of services/storage.py:387-392 for n coords rows, without the per-row Python f-string (58 938 rows of a
 * 100 000 x 100 000 slide: 75 ms in the interpreter): out[i] = prefix + "__x{X}_y{Y}_rw{RW}_rh{RH}_lv{LV}" + suffix, where the
 * caller passes prefix = slide stem and suffix = "_mag{MAG}_tmag{TMAG}_total{TOTAL}"; each entry is `width` bytes (NumPy
 * S<width>: NUL padded, truncated when longer).  coords: HOST int32 [n, 5].  Host only. */
int ap_host_format_passports(const int32_t* coords, int n, const char* prefix, const char* suffix, char* out, int width);

/* The same for tiles stored as baseline JPEG files (what a real slide's tiles are; core/wsi/openslide_wsi.py:184-205 decodes
 * them one by one in the interpreter): reads and decodes n files into consecutive side x side x 3 RGB slots with the
 * system's libjpeg-turbo (libjpeg.so.8, loaded on first use; the pixels are the ones PIL's Image.open(...).convert("RGB")
 * gives).  AP_ERR_UNSUPPORTED when that library is not usable on this host (the caller then decodes tile by tile),
 * AP_ERR_INVALID for a tile that is not side x side or not decodable. */
int ap_host_decode_jpeg_tiles(void* dst, const char* const* paths, int n, int side);

/* The same for a slide OpenSlide can open (the reference's production tile source): core/wsi/openslide_wsi.py:184-205 reads
 * one region per call in the interpreter -- openslide-python's read_region (premultiplied ARGB -> RGBA in its C helper, a
 * PIL image) + .convert("RGB") + np.array.  ap_host_openslide_read_tiles reads n regions of w x h pixels at `level` with
 * libopenslide's own openslide_read_region (thread-safe on one handle) and writes packed RGB into consecutive w*h*3-byte
 * slots: un-premultiplied exactly as openslide-python does (alpha 255: as stored; 0 < alpha < 255: (uint8)(255 * c / alpha);
 * alpha 0: the word's bytes as they lie, i.e. black for a premultiplied buffer), alpha dropped as PIL's RGBA -> RGB does.
 * xy: HOST int64 [n, 2] level-0 coordinates of the regions' top-left corners (openslide's convention).
 * libopenslide is resolved at first use by dlopen ($ATLASPATCH_LIBOPENSLIDE = explicit path, else libopenslide.so.1 /
 * .so.0 / .so): ap_host_openslide_available() = 1 / 0; without it open returns AP_ERR_UNSUPPORTED and the caller keeps the
 * per-tile openslide-python path.  The handle is independent of any openslide-python object on the same file. */
typedef struct ap_openslide ap_openslide;
int ap_host_openslide_available(void);
int ap_host_openslide_open(const char* path, ap_openslide** out);
int ap_host_openslide_read_tiles(ap_openslide* slide, const int64_t* xy, int n, int level, int w, int h, void* dst_rgb);
void ap_host_openslide_close(ap_openslide* slide);

/* Host twin of ap_synth_tiles (below): renders n square tiles of a synthetic slide into consecutive slots of a pinned
 * staging buffer, outside the interpreter lock -- the synthetic slide's "native decoder" behind the ring's batched read
 * hook, so the host -> ring -> HBM path can be driven at full rate on the 100 000 x 100 000 slide.  xy: HOST int32 [n, 2]
 * level-0 corners; ellipses: HOST int64 [k, 4].  Bit-identical to core/wsi/synth_pixels.py::render_region. */
int ap_host_synth_tiles(void* dst, const int32_t* xy, int n, int side, int level_ds, int level,
                        int64_t width, int64_t height, uint32_t seed, const int64_t* ellipses, int k);

/* ---- Pillow-exact tile resampling ------------------------------------------------------
 * Replaces the PIL resize inside the per-item transform of encoders whose transform starts with Resize:
 * timm's Resize(224, bicubic) for uni_v1 (models/patch/uni.py:48-49) and open_clip's Resize(448, bicubic)
 * for conch_v1 (models/patch/conch.py:35-38), applied by PatchDataset.__getitem__ (models/patch/base.py:42-45).
 * Bit-identical to PIL.Image.resize for uint8 RGB: horizontal then vertical pass, 22-bit fixed-point weights.
 * bounds_* : device int32 [out, 2] (first input index, tap count); coeffs_* : device int32 [out, ksize_*]
 * (the tables of Pillow's precompute_coeffs + normalize_coeffs_8bpc, built by the host);
 * src uint8 [n, h, w, 3] -> dst uint8 [n, oh, ow, 3]; tmp: device scratch of n*h*ow*3 bytes. */
int ap_resample_u8(const uint8_t* src, int n, int h, int w, uint8_t* dst, int oh, int ow,
                   const int32_t* bounds_x, const int32_t* coeffs_x, int ksize_x,
                   const int32_t* bounds_y, const int32_t* coeffs_y, int ksize_y,
                   uint8_t* tmp, ap_stream_t stream);

/* Pillow's Image.reduce((fx, fy), box) for one uint8 RGB image: the integer box reduction that Image.thumbnail(...,
 * reducing_gap=2.0) -- the reference's thumbnail call, services/segmentation.py:202-206, Pillow defaults -- runs before its
 * bicubic resize.  out[oy][ox] = ((sum of the block + n / 2) * (uint32)(2^32 / (256 n))) >> 24 (uint32 arithmetic, the
 * multiplier evaluated in float32), partial blocks at the right / bottom edge averaged over the pixels they have.
 * src: device uint8 [h, w, 3]; box = (box_x, box_y, box_w, box_h) inside it; dst: device uint8
 * [ceil(box_h / fy), ceil(box_w / fx), 3].  Bit-identical to Pillow (tested against it).  The resize that follows is
 * ap_resample_u8 with tables built for Pillow's `box` argument (utils/resample.py::pillow_resample_tables). */
int ap_pillow_reduce_u8(const uint8_t* src, int h, int w, int box_x, int box_y, int box_w, int box_h, int fx, int fy,
                        uint8_t* dst, ap_stream_t stream);

/* ---- cv2.resize for uint8 RGB ----------------------------------------------------------
 * Replaces the host cv2.resize calls of the path: services/feature_embedding.py:94-95 and
 * services/extraction.py:112-113 (cv2.resize(patch, (ps, ps)), INTER_LINEAR, on every tile whose level read is
 * not patch_size) and core/wsi/iwsi.py:305-321 (the 1.25x thumbnail: INTER_AREA when shrinking, INTER_CUBIC when
 * enlarging, INTER_LINEAR on request).  OpenCV's 8-bit arithmetic restated: integer-ratio area averages
 * (2 x 2 -> (sum + 2) >> 2, which INTER_LINEAR at exactly 2 x 2 is re-routed to), float32 area cells for other
 * shrink ratios, 11-bit fixed-point bilinear / bicubic (A = -0.75) with OpenCV's two-stage rounding; dsize equal
 * to the source size is a copy.  Per-axis tables are built by the library on first use of a shape (one
 * synchronous upload), then cached.
 * src: device uint8 [n, h, w, 3] -> dst: device uint8 [n, oh, ow, 3].  interpolation: cv2's constants.
 * flags: AP_CV_CUBIC_SCALAR = bicubic vertical pass in int32 for every element (OpenCV's scalar code); default =
 * what an x86-64 build executes (float32 vector loop for all but the last (ow * 3) % 8 elements of a row). */
#define AP_CV_INTER_LINEAR 1
#define AP_CV_INTER_CUBIC 2
#define AP_CV_INTER_AREA 3
#define AP_CV_CUBIC_SCALAR 1
int ap_cv2_resize_u8(const uint8_t* src, int n, int h, int w, uint8_t* dst, int oh, int ow,
                     int interpolation, int flags, ap_stream_t stream);

/* ---- tile content statistics (--no-fast-mode filters) -------------------------------
 * Replaces utils/image.py:7-41 (is_black_patch / is_white_patch), which services/extraction.py:112-116
 * applies to every candidate tile: counts[i] = { #pixels with cv2 RGB2GRAY < black_thresh,
 * #pixels with cv2 RGB2HSV S < white_sat_thresh and V >= white_value_thresh } (OpenCV's 8-bit fixed-point
 * conversions, bit-exact).  The caller compares count / (h*w) with 0.7 like the reference.
 * tiles: device uint8 [n, h, w, 3] (16-byte aligned, h*w % 16 == 0); counts: device uint32 [n, 2]. */
int ap_tile_content_counts(const uint8_t* tiles, int n, int h, int w, int black_thresh,
                           int white_sat_thresh, int white_value_thresh, uint32_t* counts,
                           ap_stream_t stream);

/* ---- ViT encoder -------------------------------------------------------------------
 * Replaces `forward_fn(batch) or model(batch)` of models/patch/base.py:100 for the
 * ViT family the reference registers as vit_b_16 / vit_l_16 (models/patch/vit.py:9-38)
 * and uni_v1 (models/patch/uni.py:13-60): conv patch-embed, CLS + pos-embed, pre-LN
 * blocks (packed-QKV MHA, erf-GELU MLP, optional LayerScale), final LN, CLS token. */
typedef struct ap_vit ap_vit;

/* ABI v20: the structure is growth-safe.  Its first member is its own size as the CALLER compiled it; new fields are only
 * ever appended and all-zero always means "the behaviour before the field existed".  ap_vit_create reads exactly
 * struct_size bytes and accepts only sizes this structure has had in some ABI version:
 *   struct_size == ap_sizeof_vit_config()            the caller and the library agree;
 *   an earlier ABI's size (>= AP_VIT_CONFIG_SIZE_V20)  an older caller: the fields it does not know are taken as zero;
 *   struct_size > ap_sizeof_vit_config()               a newer caller on an older library: AP_ERR_UNSUPPORTED, unread;
 *   anything else                                      AP_ERR_INVALID, unread -- in particular a binding written for ABI <= 19,
 *                                                      whose structure had no size member and started with image_size
 *                                                      (224, 448, 518: never a size of this structure).
 * Fill the structure through ap_vit_config_init (C: `ap_vit_config_init(&cfg, sizeof cfg)`; ctypes:
 * `lib.ap_vit_config_init(byref(cfg), sizeof(cfg))`) and a binding can never hand over an uninitialised tail. */
typedef struct ap_vit_config {
    uint32_t struct_size; /* sizeof(ap_vit_config) as the caller sees it; written by ap_vit_config_init */
    int image_size;      /* 224 */
    int patch_size;      /* 16 */
    int dim;             /* 768 / 1024 */
    int depth;           /* 12 / 24 */
    int heads;           /* 12 / 16 */
    int mlp_dim;         /* 3072 / 4096 */
    float ln_eps;        /* 1e-6 (torchvision, timm) or 1e-12 (HF default) */
    int layer_scale;     /* 1: per-branch gamma (timm init_values, uni.py:35) */
    int compute_dtype;   /* AP_F16 / AP_BF16 / AP_F32: MFMA operand type; accumulation,
                            residual stream, LayerNorm and softmax are always f32 */
    int pool;            /* AP_POOL_CLS: final LN, class token (vit.py, uni.py).
                            AP_POOL_ATTN: final LN on all tokens, then the attentional pooler of
                            CONCH's visual tower with ONE query (models/patch/conch.py:52,
                            encode_image(proj_contrast=False, normalize=False)): LN_k, k/v
                            projection to pool_dim, softmax pooling per head, out_proj, LN.
                            AP_POOL_CLS_MEAN: final LN on all tokens, out = [class token | mean of the PATCH tokens]
                            (2 * dim floats; register tokens excluded): torch.cat([cls, patch_tokens.mean(1)], -1) of
                            models/patch/midnight.py:58-61, virchow.py:58-61,111-114, hoptimus.py:158-161 */
    int pool_dim;        /* 512 (conch_v1); heads of 64 */
    int pool_heads;      /* 8 */
    float pool_ln_eps;   /* 1e-5 (open_clip LayerNorm) */
    /* ---- ABI v17: the rest of the reference's ViT zoo (models/patch/vit.py:9-15 vit_b_32 / vit_l_32 / vit_h_14,
     *      models/patch/uni.py:62-125 uni_v2).  All zero = the v16 behaviour. */
    int reg_tokens;      /* register tokens between the class token and the patches (timm reg_tokens; uni_v2: 8) */
    int no_embed_class;  /* 1: pos_embed has image_size^2 / patch_size^2 rows and is added to the PATCH tokens only (timm
                            no_embed_class, uni_v2); 0: pos_embed covers class (+ register) tokens too */
    int mlp_type;        /* AP_MLP_GELU: fc1 [mlp_dim, dim] -> GELU(erf) -> fc2 [dim, mlp_dim];
                            AP_MLP_SWIGLU: timm SwiGLUPacked -- fc1 [2 * mlp_dim, dim], silu(x[:, :mlp_dim]) * x[:, mlp_dim:],
                            fc2 [dim, mlp_dim] (uni_v2: mlp_dim 4096) */
    int head_dim;        /* 0 = dim / heads (must be 64); else the width of one head of q / k / v as STORED: qkv.weight has
                            3 * heads * head_dim rows, proj.weight heads * head_dim columns (64, 128, or -- float16 / bfloat16 --
                            96).  A model whose true head width is none of these (vit_h_14, Virchow: 80) is uploaded zero-padded
                            to the next one (96) with attn_scale set */
    float attn_scale;    /* 0 = 1 / sqrt(head_dim); else the softmax scale (vit_h_14: 1 / sqrt(80)) */
    /* ---- ABI v18: CLIP vision towers (models/patch/clip.py, plip.py, quilt.py: open_clip VisionTransformer / transformers
     *      CLIPVisionModel).  All zero = the v17 behaviour. */
    int pre_norm;        /* 1: LayerNorm on the embedded tokens (class + position added) before the first block (CLIP ln_pre /
                            pre_layrnorm); parameters pre_norm.weight | bias [dim] */
    int act;             /* AP_ACT_GELU (erf) or AP_ACT_QUICK_GELU (x * sigmoid(1.702 x), the OpenAI CLIP weights); AP_MLP_GELU only */
    int proj_dim;        /* 0, or P: the pooled vector (AP_POOL_CLS: final LN of the class token) is multiplied by
                            head_proj.weight [P, dim] without bias (CLIP visual projection: encode_image / get_image_features);
                            multiple of 128; ap_vit_embed_dim = P */
    int rope;            /* 1: no absolute position embedding use beyond pos_embed (upload zeros) and a rotary embedding on q / k of the
                            PATCH tokens in every block (transformers DINOv3ViTModel, models/patch/dinov3.py): parameters
                            rope.cos | rope.sin f32 [patches, head_dim] (the host builds them as the HF module does);
                            head_dim must be the true head width */
} ap_vit_config;
#define AP_VIT_CONFIG_SIZE_V20 92u   /* struct_size + the 22 fields of ABI v19: the smallest size ap_vit_create accepts */
size_t ap_sizeof_vit_config(void);   /* sizeof(ap_vit_config) inside the library */
/* Zero-fills sizeof_caller bytes at cfg and records sizeof_caller in cfg->struct_size.  AP_ERR_INVALID when cfg is NULL,
 * sizeof_caller < AP_VIT_CONFIG_SIZE_V20 or not a multiple of 4. */
int ap_vit_config_init(ap_vit_config* cfg, size_t sizeof_caller);
#define AP_ACT_GELU 0
#define AP_ACT_QUICK_GELU 1
#define AP_MLP_GELU 0
#define AP_MLP_SWIGLU 1
#define AP_POOL_CLS 0
#define AP_POOL_ATTN 1
#define AP_POOL_CLS_MEAN 2

/* Validates cfg->struct_size as described above, then every field; *out is written only on success. */
int ap_vit_create(const ap_vit_config* cfg, ap_vit** out);
void ap_vit_destroy(ap_vit* m);

/* Upload one parameter (host float32, `count` elements).  Names:
 *   patch_embed.weight [dim,3,ps,ps]  patch_embed.bias [dim]  cls_token [dim]
 *   pos_embed [1+gh*gw, dim]          norm.weight / norm.bias [dim]
 *   blocks.<i>.ln1.weight|bias  blocks.<i>.qkv.weight [3dim,dim] (rows q;k;v)  blocks.<i>.qkv.bias
 *   blocks.<i>.proj.weight [dim,dim] | .bias   blocks.<i>.ls1 [dim]
 *   blocks.<i>.ln2.weight|bias  blocks.<i>.fc1.weight [mlp,dim] | .bias
 *   blocks.<i>.fc2.weight [dim,mlp] | .bias    blocks.<i>.ls2 [dim]
 * AP_POOL_ATTN adds (P = pool_dim):
 *   attn_pool.ln_k.weight|bias [dim]   attn_pool.kv.weight [2P, dim] (rows k_proj; v_proj)   attn_pool.kv.bias [2P]
 *   attn_pool.q [P]  (the projected query q_proj(ln_q(query)) + bias: input independent, computed by the host)
 *   attn_pool.out.weight [P, P] | .bias [P]   attn_pool.ln_out.weight|bias [P]
 * Synchronous (copies before returning).
 * Every upload that a derived buffer depends on UN-FINALISES the object -- forwards return AP_ERR_STATE until
 * ap_vit_finalize succeeds again:
 *   cls_token, reg_tokens, pos_embed   (since ABI v17; before it a class-token upload took effect at once) the prefix rows
 *                                      "class / register token + its position row" are built by ap_vit_finalize;
 *   pos_embed, blocks.*                f16 / bf16: the folded weights / the T copy of pos_embed are stale (see ap_vit_finalize).
 * patch_embed.*, norm.*, pre_norm.*, rope.*, head_proj.weight and attn_pool.* take effect immediately. */
int ap_vit_set_param(ap_vit* m, const char* name, const float* host, size_t count);
/* n parameters in one call (same semantics as n ap_vit_set_param calls, in order; stops at the first error) */
int ap_vit_set_params(ap_vit* m, const char* const* names, const float* const* host, const size_t* counts, int n);
/* Checks every parameter was set and, for f16 / bf16, builds the derived weights of the fused-LayerNorm dataflow from the
 * float32 uploads (LayerNorm gain / LayerScale folded into the block matrices, column sums, folded biases, a T copy of
 * pos_embed); the float32 copies are released afterwards.  Setting a blocks.* parameter or pos_embed later un-finalises the
 * object: forwards return AP_ERR_STATE until ap_vit_finalize succeeds again, which needs the qkv / proj / fc1 / fc2
 * matrices of every block uploaded again (AP_ERR_STATE otherwise) -- stale folded weights can never run. */
int ap_vit_finalize(ap_vit* m);

/* Options (defaults: off; the environment variables AP_VIT_FULL_LAST_BLOCK / AP_VIT_OVERLAP set the defaults once, when
 * the object is created -- nothing on the launch path reads the environment):
 *   AP_VIT_OPT_FULL_LAST_BLOCK    compute the last block for every token instead of the CLS row only (same features)
 *   AP_VIT_OPT_TWO_HALF_OVERLAP   run a batch >= 512 as two halves on two streams (same features)
 *   AP_VIT_OPT_F32_STREAM         f16 / bf16 only: keep the residual stream in float32 and normalise it with standalone
 *                                 add+LayerNorm launches (round 1's dataflow; environment default AP_VIT_F32_STREAM).
 *                                 The default for f16 / bf16 keeps the stream in the compute type -- what the reference's
 *                                 own model.half() does -- with LayerNorm fused into the neighbouring GEMMs
 *   AP_VIT_OPT_EXACT_CLS          (ABI v19; default ON, environment AP_VIT_NO_EXACT_CLS turns the default off) f16 / bf16 with
 *                                 the stream in the compute type and a class-token pooling (AP_POOL_CLS, AP_POOL_CLS_MEAN):
 *                                 the class rows' residual stream is additionally carried in float32 -- after every proj /
 *                                 fc2 launch an n-row GEMM adds the unrounded branch of the class rows to it, and the
 *                                 stream's class row becomes its rounding.  The features are the class row of the stream and
 *                                 nearly all of the 16-bit stream's error in them is that row's own 2 x depth roundings:
 *                                 ViT-B/16 float16 1.26e-3 -> 7.7e-4 against the CPU fp32 path, for < 1 % of the step
 *   AP_VIT_OPT_SPLIT_F16          (additive to ABI v20; float32 compute type only; default ON, environment AP_VIT_EXACT_F32 turns the default
 *                                 off) the float32-ACCURATE
 *                                 fast mode behind `--feature-precision float32` (cli.py:175-181, models/patch/base.py:95-106):
 *                                 every buffer, LayerNorm, softmax and the residual stream stay float32; only the GEMMs' inner
 *                                 products change from the exact f32 MFMA to three f16 MFMA passes on hi / lo halves
 *                                 (x = hi + 2^-11 lo, 22 of 24 mantissa bits; w a ~= w_hi a_hi + 2^-11 (w_hi a_lo + w_lo a_hi), f32
 *                                 accumulation).  The weights are split once, when the option is switched on (a second copy of the
 *                                 matrices, kept up to date by ap_vit_set_param).  Measured on ViT-B/16, depth 12, against the CPU fp32
 *                                 path: 1.1e-6 norm-wise / 1.4e-5 element-wise maximum (the exact chain: 2.0e-6 / 3.4e-5 -- its 768- to
 *                                 3072-long sequential f32 sums round more often than the MFMA's 16-wide ones) at 1.8x the rate.
 *                                 GEMM operands must stay inside f16's range (|x| < 65504; LayerNorm / attention / GELU outputs and
 *                                 weights do).  Off = the exact f32 MFMA chain (v_mfma_f32_32x32x2_f32) */
#define AP_VIT_OPT_FULL_LAST_BLOCK 0
#define AP_VIT_OPT_TWO_HALF_OVERLAP 1
#define AP_VIT_OPT_F32_STREAM 2
#define AP_VIT_OPT_EXACT_CLS 3
#define AP_VIT_OPT_SPLIT_F16 4
int ap_vit_set_option(ap_vit* m, int option, int value);

size_t ap_vit_workspace_bytes(const ap_vit* m, int n);
int ap_vit_embed_dim(const ap_vit* m);   /* dim (AP_POOL_CLS), pool_dim (AP_POOL_ATTN) or 2 * dim (AP_POOL_CLS_MEAN) */

/* Optional per-launch timing with HIP events recorded on the forward's own stream (what
 * bench.py's roofline block reads).  Off by default; when on, every kernel launch of a forward
 * is bracketed by two events.  ap_vit_profile_read synchronises on them, returns summed
 * milliseconds and launch counts per kind since the last read, and resets. */
#define AP_PROF_PREPROC 0
#define AP_PROF_GEMM_PATCH_EMBED 1
#define AP_PROF_GEMM_QKV 2
#define AP_PROF_GEMM_PROJ 3
#define AP_PROF_GEMM_FC1 4
#define AP_PROF_GEMM_FC2 5
#define AP_PROF_ATTENTION 6
#define AP_PROF_LAYERNORM 7
#define AP_PROF_CLS_TAIL 8      /* last block after its K/V projection, CLS rows only (see ap_vit_forward_u8) */
#define AP_PROF_KINDS 9
int ap_vit_profile_enable(ap_vit* m, int on);
int ap_vit_profile_read(ap_vit* m, double* ms_by_kind, long long* launches_by_kind, int kinds);

/* patches: device uint8 [n, h, w, 3]; centre-cropped to image_size, normalised with
 * mean/std, embedded.  out: device float32 [n, dim].  Asynchronous on `stream`.
 * For the CLS readout the last block computes K / V for every token and everything after that for the CLS row
 * only (nothing reads the other rows; identical features; AP_VIT_OPT_FULL_LAST_BLOCK disables it). */
int ap_vit_forward_u8(ap_vit* m, const uint8_t* patches, int n, int h, int w,
                      const float mean[3], const float stdv[3],
                      float* out, void* workspace, size_t workspace_bytes, ap_stream_t stream);

/* x: device [n, 3, image_size, image_size] already normalised, dtype x_dtype (AP_F32 or the
 * compute dtype): the boundary a plugin `preprocess` feeds (models/patch/custom.py:31-43). */
int ap_vit_forward_chw(ap_vit* m, const void* x, int x_dtype, int n,
                       float* out, void* workspace, size_t workspace_bytes, ap_stream_t stream);

/* ---- single operators of the encoder (the same kernels ap_vit_forward_* chains) ------
 * The operator-level seam a plugin `forward_fn` (models/patch/custom.py:31-43) can bind when it
 * builds its own block structure, and what the per-kernel parity tests call.  All pointers are
 * device pointers; dtype is the MFMA operand type (AP_F16 / AP_BF16 / AP_F32).
 *
 * ap_gemm:  C[m][n] = sum_k A[m][k] * W[n][k] + bias[n], then the epilogue:
 *   AP_EPI_BIAS        out T   [M, ldo] = C                       (nn.Linear)
 *   AP_EPI_BIAS_GELU   out T   [M, ldo] = gelu_erf(C)             (Linear + nn.GELU())
 *   AP_EPI_BIAS_RESID  out f32 [M, ldo] += C * (gamma ? gamma[n] : 1)   (residual add, LayerScale)
 *   AP_EPI_BIAS_QUICK_GELU  out T [M, ldo] = C * sigmoid(1.702 C)   (CLIP's QuickGELU: models/patch/clip.py, plip.py)
 * A: T [M, lda], W: T [N, ldw] (both K-contiguous, the checkpoint's [out, in] layout), bias /
 * gamma: f32 [N].  N % 128 == 0 and K % (128 / sizeof(T)) == 0.  impl: 0 = pick, 128 = the
 * 128x128-tile kernel, 256 = the persistent 256x256-tile kernel (f16 / bf16, N % 256 == 0,
 * K % 128 == 0); variant selects a schedule variant of the 256 kernel (0 = default). */
#define AP_EPI_BIAS 0
#define AP_EPI_BIAS_GELU 1
#define AP_EPI_BIAS_RESID 2
#define AP_EPI_BIAS_QUICK_GELU 10
int ap_gemm(int dtype, int epilogue, const void* A, int lda, const void* W, int ldw,
            int M, int N, int K, const float* bias, const float* gamma, void* out, int ldo,
            int impl, int variant, ap_stream_t stream);
/* The float32-accurate fast product (what AP_VIT_OPT_SPLIT_F16 runs; added to ABI v20 without a layout change).
 * ap_split_f16_weights: w32 f32 [count] (count % 32 == 0: whole rows of a K that is a multiple of 32) -> out, the same
 * number of BYTES: per 32 consecutive values 32 f16 `hi` followed by 32 f16 `lo`, hi = f16(w), lo = f16((w - hi) * 2^11).
 * ap_gemm(AP_F32, ..., impl = 129) then takes such rows as W (ldw still in f32 elements), float32 A / bias / out, and
 * computes  sum_k  w_hi a_hi + 2^-11 (w_hi a_lo + w_lo a_hi)  with a split in registers, f32 accumulation: ~2^-22 per
 * product against the exact f32 chain of impl 128, at 2-3x its rate.  K % 32 == 0, N % 128 == 0. */
int ap_split_f16_weights(const float* w32, void* out, size_t count, ap_stream_t stream);
/* The same product as a row-wise float32 layer with an optional SEPARATE residual (the SAM2 trunk's Linear / MLP layers,
 * services/segmentation.py:120-180 runs them in float32):
 *   out[m][n] = act(sum_k A[m][k] W[n][k] + bias[n]) + resid[m][n]        act: 0 none, 1 GELU (erf)
 * A f32 [M, lda], w_split = ap_split_f16_weights of the f32 [N, K] matrix, bias f32 [N] or NULL, resid f32 [M, ldr] or NULL,
 * out f32 [M, ldo].  N % 32 == 0 (any such N: the 128-wide tile's tail is masked), K % 32 == 0, strides multiples of 4,
 * pointers 16-byte aligned.  A row's result does not depend on M (no split-K): batches stay bit-identical to single calls. */
int ap_gemm_split_f16(const float* A, int lda, const void* w_split, int M, int N, int K, const float* bias, int act,
                      const float* resid, int ldr, float* out, int ldo, ap_stream_t stream);
/* ... with the window (un)partition of the SAM2 trunk's windowed blocks folded in (sam2 hieradet.py window_partition /
 * window_unpartition as the reference's SAM2ImagePredictor runs them; replaces ap_window_partition / ap_window_unpartition_add
 * around the qkv / proj layers).  The token grid is b images of h x w tokens, cut into ws x ws windows, zero-padded at the
 * right / bottom edge; M = b * ceil(h / ws) * ceil(w / ws) * ws * ws window-order rows.
 *   win_mode 1: A is [b * h * w, lda] in IMAGE order; product row m takes the image row of window-order row m (zeros for a
 *               padding row); out [M, ldo] in window order (the qkv layer after norm1).
 *   win_mode 2: A is [M, lda] in window order; out / resid are [b * h * w, ld] in IMAGE order, padding rows are dropped
 *               (the proj layer + residual add). */
int ap_gemm_split_f16_windows(const float* A, int lda, const void* w_split, int M, int N, int K, const float* bias, int act,
                              const float* resid, int ldr, float* out, int ldo, int win_mode, int b, int h, int w, int ws,
                              ap_stream_t stream);

/* ---- fused-LayerNorm operators (what ap_vit_forward_* chains for f16 / bf16 unless AP_VIT_OPT_F32_STREAM is set) ----
 * The pre-LN block of the reference's encoders (nn.LayerNorm -> nn.Linear, models/patch/vit.py / uni.py / conch.py via
 * timm / torchvision Block.forward: x = x + ls1 * attn(norm1(x)); x = x + ls2 * mlp(norm2(x))) with the residual
 * stream x kept in T and NO standalone LayerNorm pass:
 *   LN(x) W^T + b  =  rstd[m] * (sum_k x[m][k] W'[n][k]  -  mean[m] * colsum[n]) + bias'[n]
 * with W' = T(W * gamma) (the caller folds the LayerNorm gain into the weights), colsum[n] = sum_k W'[n][k],
 * bias' = b + W beta and rowstats[m] = (rstd, -mean * rstd) of row m of x.
 *   AP_EPI_NORM         out T [M, ldo] = rstd * acc + (-mean rstd) * colsum[n] + bias[n]
 *   AP_EPI_NORM_GELU    out = gelu(that)
 *   AP_EPI_NORM_QUICK_GELU  out = that * sigmoid(1.702 * that)
 *   AP_EPI_NORM_SWIGLU  timm SwiGLUPacked (models/patch/uni.py:91-93, uni_v2) with the gate in the epilogue: W / colsum / bias rows
 *                       INTERLEAVED in groups of 64 -- rows 64q .. 64q+31 = fc1 rows 32q .. (x1), rows 64q+32 .. 64q+63 = fc1 rows
 *                       N/2 + 32q .. (x2) -- and out T [M, N / 2]: out[m][32q + j] = silu(norm x1) * norm x2, one rounding
 *   AP_EPI_RESID_STATS  out T [M, ldo] (in place) = T(out + T(acc + bias[n]))  -- the residual add -- and
 *                       partial f32 [M, N / 64, 2] = per row and 64-column group (sum, sum of squares) of the NEW row;
 *                       ap_rowstats_finalize turns them into the next rowstats (deterministic, fixed order, double).
 * impl: 0 = pick, 256 = the persistent 256 x 256 kernel (N % 256 == 0, K % 128 == 0; any M >= 1), 128 = the 128 x 128
 * kernel (N % 128 == 0, K % 64 == 0); the two give the same bits (the statistics are summed in one fixed order).  f16 / bf16.
 * ap_stream_init: tok f32 [rows, dim] -> x T [rows, dim] + rowstats of the rounded rows (two-pass). */
#define AP_EPI_NORM 4
#define AP_EPI_NORM_GELU 5
#define AP_EPI_RESID_STATS 6
#define AP_EPI_NORM_SWIGLU 8
#define AP_EPI_NORM_QUICK_GELU 9
int ap_gemm_fused(int dtype, int epilogue, const void* A, int lda, const void* W, int ldw, int M, int N, int K,
                  const float* bias, const float* colsum, const float* rowstats, float* partial,
                  void* out, int ldo, int impl, ap_stream_t stream);
int ap_stream_init(int dtype, const float* tok, int rows, int dim, float eps, void* x, float* rowstats,
                   ap_stream_t stream);
int ap_rowstats_finalize(const float* partial, int rows, int groups, int dim, float eps, float* rowstats,
                         ap_stream_t stream);

/* Diagnostics for the persistent kernel's instrumented twin (impl 257; the product kernel, impl 256, carries no
 * diagnostic code): when device_buf is non-null every later impl-257 launch records, per (workgroup, tile) with tile < tiles_per_workgroup, eight int64 stamps of the
 * 100-MHz wall clock: [0] tile start, [1] K loop done, [2] staged stream drained, [3] bias loaded,
 * [4] epilogue done.  device_buf: int64 [grid, tiles_per_workgroup, 8].  Pass NULL to switch off. */
int ap_gemm_trace(long long* device_buf, int tiles_per_workgroup);

/* LayerNorm over the last dimension of f32 rows (row stride `stride` elements) -> dense
 * out [rows, dim] of out_dtype (nn.LayerNorm, eps inside the sqrt, biased variance). */
int ap_layernorm(int out_dtype, const float* x, long stride, int rows, int dim,
                 const float* gamma, const float* beta, float eps, void* out, ap_stream_t stream);

/* Multi-head self-attention on packed projections: qkv T [n * tokens, 3 * heads * head_dim]
 * (q | k | v), out T [n * tokens, heads * head_dim]; softmax(q k^T / sqrt(head_dim)) v in f32
 * (F.scaled_dot_product_attention without mask / dropout).  head_dim: 64 (any dtype), 96 or 128 (float16 / bfloat16). */
int ap_attention(int dtype, const void* qkv, void* out, int n, int tokens, int heads, int head_dim,
                 ap_stream_t stream);

/* ---- float32 operator set of the SAM2 (Hiera-T) tissue segmenter ------------------------
 * Replaces the torch modules behind SAM2ImagePredictor.set_image / predict as the reference drives them
 * (services/segmentation.py:120-140: one 1024 x 1024 thumbnail per slide, box prompt = whole image,
 * multimask_output=False, mask_threshold 0.0).  The host chains these (atlaspatch_amd/services/sam2_hip.py);
 * all tensors are float32, channels-last ([tokens, C]), device pointers.
 *
 * ap_sgemm: out[b][m][n] = act(alpha * sum_k A[b][m][k] * W[b][n][k] + bias[n]) + resid[b][m][n]
 *   (W[b][k][n] when w_is_kn).  act: 0 none, 1 GELU (erf), 2 ReLU.  ld* = row strides, stride* = batch strides
 *   in elements; bias / resid may be NULL.  Any M, N, K. */
int ap_sgemm(const float* A, long lda, long strideA, const float* W, long ldw, long strideW, int w_is_kn,
             int batch, int M, int N, int K, float alpha, const float* bias, int act,
             const float* resid, long ldr, long strideR, float* out, long ldo, long strideO, ap_stream_t stream);
/* The same for `stack` equally shaped problems stacked along M (a batch of images through a row-wise layer), NT weights,
 * no batch strides: every output row goes through exactly the arithmetic it would go through alone (the split-K plan is
 * the single problem's), so the result does not depend on how many images are stacked; M % stack == 0. */
int ap_sgemm_stacked(const float* A, long lda, const float* W, long ldw, int stack, int M, int N, int K, const float* bias,
                     int act, const float* resid, long ldr, float* out, long ldo, ap_stream_t stream);
int ap_softmax_rows(float* x, long ld, int rows, int cols, ap_stream_t stream);     /* in place */
/* Fused attention of the trunk (hieradet.py MultiScaleAttention -> F.scaled_dot_product_attention), image-wide blocks
 * (batch = 1) and windowed blocks (batch = number of windows, window b owns rows b * tq .. of q / out and b * tk .. of k / v):
 * out[b*tq + t][h*d + c] = sum_j softmax_j(scale * q[b*tq + t][h*d + :] . k[b*tk + j][h*d + :]) v[b*tk + j][h*d + c];
 * float32, head h at column h * d; exact-f32 MFMA, online softmax, the scores never reach memory.  d in {32, 64, 96};
 * any tq, tk >= 1; row strides multiples of 4 floats, q / k / out 16-byte aligned. */
int ap_sattention_f32(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv, int batch, int heads,
                      int tq, int tk, int d, float scale, float* out, long ldo, ap_stream_t stream);
/* The same operator (same arguments, float32 in and out) with both products as split-f16 MFMA passes:
 * a b ~= a_hi b_hi + a_hi b_lo + a_lo b_hi on v_mfma_f32_32x32x16_f16 with f32 accumulation, hi = f16(x), lo = f16(x - hi)
 * (float32-accurate: ~2^-22 per product; the softmax stays float32).  2-3x the rate of the exact chain on the image-wide blocks. */
int ap_sattention_split_f16(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv, int batch, int heads,
                            int tq, int tk, int d, float scale, float* out, long ldo, ap_stream_t stream);
/* uint8 [h, w, 3] -> rows [(h/4)*(w/4), 147] of ((x/255) - mean) / std in (c, ky, kx) order: the im2col of
 * Hiera's PatchEmbed conv (7x7, stride 4, pad 3). */
int ap_sam2_patchify(const uint8_t* image, int h, int w, const float mean[3], const float stdv[3], float* out,
                     ap_stream_t stream);
/* hieradet.py window_partition / window_unpartition on [b, h, w, c]: windows [b * ceil(h/ws) * ceil(w/ws), ws*ws, c],
 * zero padded / cropped. */
int ap_window_partition(const float* x, int b, int h, int w, int c, int ws, float* win, ap_stream_t stream);
int ap_window_unpartition(const float* win, int b, int h, int w, int c, int ws, float* x, ap_stream_t stream);
/* x = resid + window_unpartition(win): the block's `x = shortcut + drop_path(attn(...))` in the same pass (hieradet.py MultiScaleBlock) */
int ap_window_unpartition_add(const float* win, const float* resid, int b, int h, int w, int c, int ws, float* x,
                              ap_stream_t stream);
/* 2x2 stride-2 max pool on [b, h, w, c] whose pixels are ld_in elements apart -> dense [b, h/2, w/2, c] */
int ap_maxpool2x2(const float* in, long ld_in, int b, int h, int w, int c, float* out, ap_stream_t stream);
int ap_add(float* out, const float* a, const float* b, size_t n, ap_stream_t stream);
int ap_add_rowvec(float* out, const float* a, const float* vec, size_t rows, int cols, ap_stream_t stream);
int ap_gelu(float* x, size_t n, ap_stream_t stream);
/* FpnNeck top-down: out [2h, 2w, c] = lateral + nearest-x2(prev [h, w, c]) */
int ap_upsample2x_add(float* out, const float* lateral, const float* prev, int h, int w, int c, ap_stream_t stream);
/* ConvTranspose2d(kernel 2, stride 2) epilogue: g [h*w, cout*4] (column co*4 + dy*2 + dx) -> out [2h, 2w, cout]
 * = g + bias[co] (+ skip) (GELU when act = 1) */
int ap_convt2x2_shuffle(const float* g, const float* bias, const float* skip, float* out, int h, int w, int cout,
                        int act, ap_stream_t stream);
/* postprocess_masks: bilinear x4 (align_corners False) of logits [size, size], then > threshold -> float {0,1} */
int ap_bilinear_up4_threshold(const float* logits, int size, float threshold, float* mask, ap_stream_t stream);
/* _resize_mask (services/segmentation.py:112-118: PIL NEAREST back to the thumbnail's shape) as a gather:
 * dst[y][x] = src[yidx[y]][xidx[x]]; yidx int32 [oh], xidx int32 [ow] (device; built on the host with Pillow's
 * accumulated-double index arithmetic, utils/resample.py::pillow_nearest_index; entries must lie inside src). */
int ap_gather2d_f32(const float* src, int src_h, int src_w, const int32_t* yidx, const int32_t* xidx, int oh, int ow,
                    float* dst, ap_stream_t stream);

/* ---- tissue mask -> patch coordinates ----------------------------------------------
 * Replaces utils/contours.py:41-131 (mask_to_contours, scale_contours) and the grid scan of
 * services/extraction.py:67-128 (_in_tissue, _iter_patch_entries, FourPointContainment). */
typedef struct ap_contours ap_contours;

/* mask: HOST float32 [h, w].  Thresholds (> 0.5) on the device, follows borders
 * (Suzuki-Abe, RETR_CCOMP / CHAIN_APPROX_NONE semantics), applies the area / hole filters
 * of mask_to_contours and scales to level 0 with scale_contours' float32 truncation.
 * Synchronous. */
int ap_contours_from_mask(const float* mask, int h, int w, double tissue_area_thresh,
                          int min_hole_area, int max_n_holes, double sx, double sy,
                          ap_contours** out, ap_stream_t stream);
void ap_contours_destroy(ap_contours* c);
int ap_contours_count(const ap_contours* c);                 /* tissue contours */
int ap_contours_num_holes(const ap_contours* c, int i);
/* points of tissue contour i (hole < 0) or of its hole `hole`; scaled=0 -> mask space.
 * Returns the point count; copies min(count, cap) int32 (x, y) pairs into xy. */
int ap_contours_points(const ap_contours* c, int i, int hole, int scaled, int32_t* xy, int cap);

/* Grid scan on the device.  coords: HOST int32 [cap, 5] rows (x, y, read_w, read_h, level) in
 * the reference's order.  *n_rows receives the total (also when it exceeds cap ->
 * AP_ERR_CAPACITY).  Synchronous. */
int ap_grid_coords(const ap_contours* c, int patch_size_src, int step_src,
                   int read_w, int read_h, int level,
                   int32_t* coords, size_t cap, size_t* n_rows, ap_stream_t stream);

/* ---- synthetic slide tiles ------------------------------------------
 * Renders tiles of a synthetic slide straight into HBM: dst uint8 [n, ps, ps, 3] for the n
 * level-0 top-left corners xy (device int32 [n, 2]).  Bit-identical to
 * atlaspatch_amd/core/wsi/synth_pixels.py.  ellipses: device int64 [k, 4]. */
int ap_synth_tiles(const int32_t* xy, int n, int ps, int level_ds, int level,
                   int64_t width, int64_t height, uint32_t seed,
                   const int64_t* ellipses, int k, uint8_t* dst, ap_stream_t stream);
/* One w x h region of pyramid level `level` whose level-0 corner is (x, y): dst uint8 [h, w, 3].  The whole-level read of
 * the thumbnail path (core/wsi/iwsi.py:246-323 reads the level nearest 1.25x in full) for synthetic slides, in HBM. */
int ap_synth_region(int64_t x, int64_t y, int w, int h, int level_ds, int level,
                    int64_t width, int64_t height, uint32_t seed,
                    const int64_t* ellipses, int k, uint8_t* dst, ap_stream_t stream);

/* Measurement aid (bench.py): one launch of 1024 single-wave workgroups; each writes, into the slot of the compute unit it ran on
 * -- slot = XCC_ID (3 bits) << 8 | SE_ID (3) << 5 | SH_ID (1) << 4 | CU_ID (4), from HW_REG_XCC_ID / HW_REG_HW_ID -- the pair
 * out[2 slot + 0] = s_memtime (shader-clock ticks), out[2 slot + 1] = s_memrealtime (100 MHz ticks) as one 16-byte store.
 * s_memtime counters of different compute units are NOT aligned with each other (offsets of 1e7 ticks and more inside one XCD),
 * so only stamps of the SAME compute unit may be differenced.  out: AP_CLOCK_PROBE_SLOTS * 2 int64 in device memory, zeroed by
 * the caller.  Two probes on one stream around a region give the average shader clock there per compute unit that both reached:
 * GHz = 0.1 * d(memtime) / d(memrealtime).  No reference counterpart.  (ABI v18: v17 keyed the slots by XCD only.) */
#define AP_CLOCK_PROBE_SLOTS 2048
int ap_clock_probe(long long* out, ap_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ATLASPATCH_HIP_H */
