// ViT encoder object behind the C ABI: parameter storage in HBM, workspace carving and the
// launch sequence of one forward pass.  Host C++ only; every kernel
// it launches lives in gemm.hip / attention.hip / elementwise.hip / preproc.hip.
//
// HBM layout for a batch of n images (T = compute dtype, Tk = 1 + (S/ps)^2 tokens, M = n*Tk):
//   tok   f32 [M, D]      residual stream (always f32)
//   xn    T   [M, D]      LayerNorm output (GEMM A operand)
//   qkv   T   [M, 3D]     packed q | k | v
//   att   T   [M, D]      attention output
//   delta, delta2 T [M, D] branch outputs (fc2 / proj) waiting to be folded into tok by a LayerNorm launch
//   hid   T   [M, mlp]    GELU(fc1) ; the patch-row matrix [n*P, Kpe] aliases it
// Weights are converted once to T, K-contiguous ([out, in], exactly the checkpoint layout).
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include "ap_common.h"
#include <cstdlib>

namespace ap {

struct Param {
    size_t count = 0;      // f32 elements expected
    int rows = 0, cols = 0, ld = 0;   // matrices converted to T: [rows, ld] (cols <= ld, zero padded)
    bool matrix = false;
    void* dev = nullptr;   // f32 vector or T matrix
    float* dev32 = nullptr; // matrices, 16-bit compute types: the f32 upload [rows, ld], kept until ap_vit_finalize has folded it
    void* split = nullptr; // float32 compute type, matrices: the [hi 32 | lo 32] f16 rows of the split-f16 GEMM (AP_VIT_OPT_SPLIT_F16)
    bool set = false;
};

}  // namespace ap

namespace ap {
// parameter pointers of one block, resolved once by ap_vit_finalize (no name lookups on the launch path)
struct BlockParams {
    const float *ln1_w, *ln1_b, *qkv_b, *proj_b, *ln2_w, *ln2_b, *fc1_b, *fc2_b, *ls1, *ls2;
    const Param *qkv, *proj, *fc1, *fc2;
};
// Fused-LayerNorm path (16-bit compute types): the weights that consume a LayerNorm output carry its gain
// (W' = T(W * gamma), colsum[n] = sum_k W'[n][k], bias' = b + W beta); the branch projections carry LayerScale.
struct FusedBlock {
    void *qkv_w = nullptr, *fc1_w = nullptr, *proj_w = nullptr, *fc2_w = nullptr;      // T [rows, ld] (ld of the plain Param)
    float *qkv_cs = nullptr, *qkv_b = nullptr, *fc1_cs = nullptr, *fc1_b = nullptr, *proj_b = nullptr, *fc2_b = nullptr;
};
struct PoolParams {
    const float *ln_k_w, *ln_k_b, *kv_b, *q, *out_b, *ln_out_w, *ln_out_b;
    const Param *kv, *out;
};
}  // namespace ap

struct ap_vit {
    ap_vit_config cfg;
    int grid = 0, patches = 0, tokens = 0, kpe = 0;
    int prefix = 1;                         // class token + register tokens
    int pos_rows = 0, pos_row0 = 1;         // rows of pos_embed; the row that belongs to patch 0
    int hd = 64, dattn = 0;                 // head width as stored (64 / 128), heads * hd = width of q, k, v and of proj's input
    int fc1_rows = 0;                       // mlp_dim (GELU) or 2 * mlp_dim (SwiGLU packed)
    float attn_scale = 0.125f;
    float* prefix_dev = nullptr;            // f32 [prefix, dim]: class / register tokens (+ their position rows), built at finalize
    std::map<std::string, ap::Param> params;
    bool finalized = false;
    // resolved at finalize
    std::vector<ap::BlockParams> blocks;
    ap::PoolParams pool{};
    const ap::Param* pe_w = nullptr;
    const float *pe_b = nullptr, *cls = nullptr, *pos = nullptr, *norm_w = nullptr, *norm_b = nullptr;
    const float *pre_w = nullptr, *pre_b = nullptr;        // CLIP ln_pre
    const float *rope_cos = nullptr, *rope_sin = nullptr;  // DINOv3 rotary tables f32 [patches, head_dim]
    const ap::Param* head_proj = nullptr;                       // CLIP visual projection [P, dim]
    float* zero_bias = nullptr;                             // f32 [proj_dim] zeros (the projection has no bias)
    // options (ap_vit_set_option; the defaults come from the environment once, at creation)
    bool full_last_block = false, two_half_overlap = false, f32_stream = false;
    bool exact_cls = true;      // fused dataflow: the class rows' residual stream is also kept in f32 (blocks_fused)
    bool split_f16 = false;     // float32 compute type: GEMMs as three f16 MFMA passes on hi / lo halves (gemm.hip) instead of f32 MFMA
    std::vector<ap::FusedBlock> fused;      // filled by ap_vit_finalize for f16 / bf16
    std::vector<void*> fused_allocs;
    void* pos16 = nullptr;                  // position embedding in the compute type (fused patch embedding), in fused_allocs
    bool fold_dirty = false;                // a parameter the folded weights depend on was set after the last finalize
    std::vector<float*> pending_free;       // ap_vit_set_params: f32 uploads to release once its stream has drained
    int device = 0;
    // optional per-launch HIP-event timing (ap_vit_profile_*): kind -> events of the last forwards
    bool profile = false;
    std::vector<hipEvent_t> ev_pool;
    std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> ev_used;
    size_t ev_next = 0;
    // experimental two-half overlap (AP_VIT_OVERLAP=1): second stream + hand-off events
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
};

namespace {
hipEvent_t next_event(ap_vit* m) {
    if (m->ev_next == m->ev_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        m->ev_pool.push_back(e);
    }
    return m->ev_pool[m->ev_next++];
}
struct ScopedTimer {      // records start/stop events around one launch when profiling is on
    ap_vit* m; int kind; hipStream_t s; hipEvent_t a = nullptr, b = nullptr;
    ScopedTimer(ap_vit* m_, int kind_, hipStream_t s_) : m(m_), kind(kind_), s(s_) {
        if (m->profile) { a = next_event(m); b = next_event(m); if (a) (void)hipEventRecord(a, s); }
    }
    ~ScopedTimer() {
        if (m->profile && a && b) { (void)hipEventRecord(b, s); m->ev_used.push_back({kind, {a, b}}); }
    }
};
}  // namespace

namespace {

using ap::Param;

int alloc_param(ap_vit* m, const std::string& name, int rows, int cols, bool matrix) {
    Param p;
    p.rows = rows; p.cols = cols; p.matrix = matrix;
    p.count = (size_t)rows * cols;
    const int kq = 64;    // K padded to 64 elements: whole 128-byte K tiles for every dtype
    p.ld = matrix ? (int)ap::align_up(cols, kq) : cols;
    const size_t bytes = matrix ? (size_t)rows * p.ld * ap::dtype_size(m->cfg.compute_dtype)
                                : p.count * sizeof(float);
    AP_HIP_CHECK(hipMalloc(&p.dev, bytes));
    AP_HIP_CHECK(hipMemset(p.dev, 0, bytes));
    m->params[name] = p;
    return AP_OK;
}

const Param* find(const ap_vit* m, const std::string& name) {
    auto it = m->params.find(name);
    return it == m->params.end() ? nullptr : &it->second;
}

// weight operand of a GEMM on the float32 path: the f32 matrix, or its split rows when AP_VIT_OPT_SPLIT_F16 is on
inline bool use_split(const ap_vit* m) { return m->split_f16 && m->cfg.compute_dtype == AP_F32; }
inline const void* wsel(const ap_vit* m, const Param* p) { return use_split(m) ? p->split : p->dev; }

struct Workspace {
    float* tok; void* xn; void* qkv; void* att; void* hid; void* hid2; void* delta; void* delta2;
    void* x16; float* rowstats; float* partial;      // fused-LayerNorm path: T stream [M, D], f32 [M, 2], f32 [M, D / 64, 2]
    float* cls32;                                    //   and the class rows' exact residual stream, f32 [n, D]
    float* cls_branch;                               //   + their unrounded branch of the last proj / fc2 launch, f32 [n, D]
    size_t total;
};

Workspace carve(const ap_vit* m, int n, char* base) {
    const size_t es = ap::dtype_size(m->cfg.compute_dtype);
    const size_t M = (size_t)n * m->tokens, D = m->cfg.dim;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += ap::align_up(bytes, 256); return o; };
    Workspace w;
    const size_t o_tok = take(M * D * 4);
    const size_t o_xn = take(M * D * es);
    const size_t o_qkv = take(M * 3 * (size_t)m->dattn * es);
    const size_t o_att = take(M * (size_t)m->dattn * es);
    const size_t o_delta = take(M * D * es);
    const size_t o_delta2 = take(M * D * es);
    // SwiGLU: fc1 output [M, 2 mlp] followed by the gated product [M, mlp] (ws.hid2)
    size_t hid_bytes = M * (size_t)(m->cfg.mlp_type == AP_MLP_SWIGLU ? 3 * m->cfg.mlp_dim : m->cfg.mlp_dim) * es;
    const size_t pe_bytes = (size_t)n * m->patches * m->kpe * es;
    if (pe_bytes > hid_bytes) hid_bytes = pe_bytes;
    const size_t o_hid = take(hid_bytes);
    const bool has_fused = m->cfg.compute_dtype != AP_F32;
    const size_t o_x16 = has_fused ? take(M * D * es) : 0;
    const size_t o_rs = has_fused ? take(M * 2 * sizeof(float)) : 0;
    const size_t o_part = has_fused ? take(M * (D / 64) * 2 * sizeof(float)) : 0;
    const size_t o_cls32 = has_fused ? take((size_t)n * D * sizeof(float)) : 0;
    const size_t o_clsbr = has_fused ? take((size_t)n * D * sizeof(float)) : 0;
    w.x16 = has_fused ? base + o_x16 : nullptr;
    w.rowstats = has_fused ? (float*)(base + o_rs) : nullptr;
    w.partial = has_fused ? (float*)(base + o_part) : nullptr;
    w.cls32 = has_fused ? (float*)(base + o_cls32) : nullptr;
    w.cls_branch = has_fused ? (float*)(base + o_clsbr) : nullptr;
    w.tok = (float*)(base + o_tok); w.xn = base + o_xn; w.qkv = base + o_qkv;
    w.att = base + o_att; w.hid = base + o_hid; w.delta = base + o_delta; w.delta2 = base + o_delta2; w.total = off;
    w.hid2 = m->cfg.mlp_type == AP_MLP_SWIGLU ? (char*)w.hid + M * (size_t)(2 * m->cfg.mlp_dim) * es : w.hid;
    return w;
}

// What the block loop leaves for the final LayerNorm: tok (f32, row stride tok_stride as seen from the CLS rows) plus at
// most one branch output still to be added.
struct StreamTail {
    const void* pending = nullptr; const float* pending_ls = nullptr; long pending_stride = 0; long tok_stride = 0;
};

int patch_embed(ap_vit* m, int n, const Workspace& w, hipStream_t stream) {
    const ap_vit_config& c = m->cfg;
    const int dt = c.compute_dtype, D = c.dim;
    int rc;
    // patch embedding: tok[img][1 + p] = pe_row @ W^T + b + pos[1 + p]
    {
        const Param* wpe = m->pe_w;
        ap::GemmArgs g{};
        g.A = w.hid; g.lda = m->kpe; g.W = wsel(m, wpe); g.split = use_split(m); g.ldw = wpe->ld;
        g.M = n * m->patches; g.N = D; g.K = m->kpe;
        g.bias = m->pe_b;
        g.pos = m->pos;
        g.out = w.tok; g.ldo = D; g.P = m->patches; g.R = m->prefix; g.pos_row0 = m->pos_row0;
        { ScopedTimer t(m, AP_PROF_GEMM_PATCH_EMBED, stream);
          if ((rc = ap::launch_gemm(dt, ap::EPI_PATCH_EMBED, g, stream)) != AP_OK) return rc; }
        if ((rc = ap::launch_cls_init(w.tok, m->prefix_dev, m->prefix, n, m->tokens, D, stream)) != AP_OK) return rc;
    }
    return AP_OK;
}

// fused path: the patch embedding writes the T stream and its partial sums directly (EPI_PATCH_STREAM), the class-token
// rows come from a tiny kernel, one finalisation turns the partial sums into the first row statistics
int patch_embed_stream(ap_vit* m, int n, const Workspace& w, hipStream_t stream) {
    const ap_vit_config& c = m->cfg;
    const int dt = c.compute_dtype, D = c.dim;
    int rc;
    ap::GemmArgs g{};
    g.A = w.hid; g.lda = m->kpe; g.W = m->pe_w->dev; g.ldw = m->pe_w->ld;
    g.M = n * m->patches; g.N = D; g.K = m->kpe;
    g.bias = m->pe_b; g.pos16 = m->pos16; g.P = m->patches; g.R = m->prefix; g.pos_row0 = m->pos_row0;
    g.out = w.x16; g.ldo = D; g.partial = w.partial;
    { ScopedTimer t(m, AP_PROF_GEMM_PATCH_EMBED, stream);
      if ((rc = ap::launch_gemm(dt, ap::EPI_PATCH_STREAM, g, stream)) != AP_OK) return rc; }
    ScopedTimer t(m, AP_PROF_LAYERNORM, stream);
    if ((rc = ap::launch_cls_stream(dt, m->prefix_dev, m->prefix, 0, n, m->tokens, D, w.x16, w.partial, stream)) != AP_OK) return rc;
    return ap::launch_rowstats_finalize(w.partial, n * m->tokens, D / 64, D, c.ln_eps, w.rowstats, stream);
}

// ---- block loop, f32 residual stream (float32 mode; AP_VIT_OPT_F32_STREAM for f16 / bf16)
int blocks_f32_stream(ap_vit* m, int n, const Workspace& w, StreamTail& st, hipStream_t stream) {
    const ap_vit_config& c = m->cfg;
    const int dt = c.compute_dtype, D = c.dim, M = n * m->tokens;
    const int DA = m->dattn, H = c.mlp_dim, F1 = m->fc1_rows;
    const bool swiglu = c.mlp_type == AP_MLP_SWIGLU;
    int rc;
    // Residual stream: tok (f32) is only ever touched by the add+LayerNorm kernel.  A branch GEMM
    // (proj, fc2) stores its output delta = acc + bias in T; LayerNorm launches fold it into the
    // stream in f32 (tok += delta * layer_scale) before normalising.  The stream is written back once
    // per block: ln1 normalises tok + fc2_prev WITHOUT storing it, ln2 folds fc2_prev and proj together
    // (same f32 operation order) and stores -> 620 MB less HBM traffic per block at n = 1024.
    const void* pending = nullptr;       // branch output not yet added to tok
    long pending_stride = D;             // its row stride as seen from the final CLS LayerNorm
    // AP_VIT_OPT_FULL_LAST_BLOCK computes the last block for every token (A/B of the CLS-only tail; same features)
    const bool cls_tail = c.pool == AP_POOL_CLS && !m->full_last_block;
    const float* pending_ls = nullptr;   // ... and its LayerScale vector (applied in f32 by the add)
    for (int i = 0; i < c.depth; ++i) {
        const ap::BlockParams& bp = m->blocks[i];
        { ScopedTimer t(m, AP_PROF_LAYERNORM, stream);
          if ((rc = ap::launch_add2_layernorm(dt, dt, w.tok, D, pending, D, pending_ls, nullptr, 0, nullptr, /*store=*/0,
                                              M, D, bp.ln1_w, bp.ln1_b, c.ln_eps, w.xn,
                                              stream)) != AP_OK) return rc; }
        if (i == c.depth - 1 && cls_tail) {
            // ---- last block, CLS readout: nothing reads this block's output for the patch tokens, so only what the
            // CLS row depends on is computed: K and V of every token, then the CLS row alone through q-projection,
            // attention, proj, ln2, fc1, fc2.  Same operators, same operation order per row -> same features.
            const size_t es = ap::dtype_size(dt);
            const Param* wq = bp.qkv;
            const long cls_stride = (long)m->tokens * D;
            char* q_cls = (char*)w.att;                                   // T [n, DA]
            char* a_cls = (char*)w.att + (size_t)n * DA * es;             // T [n, DA]
            {
                ap::GemmArgs g{};                                          // k | v for all rows
                g.A = w.xn; g.lda = D; g.W = (const char*)wsel(m, wq) + (size_t)DA * wq->ld * es; g.split = use_split(m); g.ldw = wq->ld;
                g.M = M; g.N = 2 * DA; g.K = D; g.bias = bp.qkv_b + DA;
                g.out = (char*)w.qkv + (size_t)DA * es; g.ldo = 3 * DA;
                ScopedTimer t(m, AP_PROF_GEMM_QKV, stream);
                if ((rc = ap::launch_gemm(dt, ap::EPI_BIAS_STORE, g, stream)) != AP_OK) return rc;
                if (c.rope && (rc = ap::launch_rope(dt, w.qkv, n, m->tokens, m->prefix, c.heads, m->hd, m->rope_cos, m->rope_sin, 2,
                                                    stream)) != AP_OK) return rc;     // the class rows' q is not rotated
            }
            // n-row GEMMs on the 128x128 kernel: more workgroups than 256x256 tiles would give, bit-identical results
            ScopedTimer t(m, AP_PROF_CLS_TAIL, stream);
            {
                ap::GemmArgs g{};                                          // q for the CLS rows
                g.A = w.xn; g.lda = (int)cls_stride; g.W = wsel(m, wq); g.split = use_split(m); g.ldw = wq->ld;
                g.M = n; g.N = DA; g.K = D; g.bias = bp.qkv_b; g.out = q_cls; g.ldo = DA;
                if ((rc = ap::launch_gemm_impl(dt, ap::EPI_BIAS_STORE, g, 128, 0, stream)) != AP_OK) return rc;
            }
            if ((rc = ap::launch_attention_cls(dt, q_cls, w.qkv, 3 * DA, DA, 2 * DA, a_cls, n, m->tokens, c.heads,
                                               m->hd, m->attn_scale, stream)) != AP_OK) return rc;
            {
                ap::GemmArgs g{};
                g.A = a_cls; g.lda = DA; g.W = wsel(m, bp.proj); g.split = use_split(m); g.ldw = bp.proj->ld;
                g.M = n; g.N = D; g.K = DA; g.bias = bp.proj_b; g.out = w.delta2; g.ldo = D;
                if ((rc = ap::launch_gemm_impl(dt, ap::EPI_BIAS_STORE, g, 128, 0, stream)) != AP_OK) return rc;
            }
            if ((rc = ap::launch_add2_layernorm(dt, dt, w.tok, cls_stride, pending, cls_stride, pending_ls, w.delta2, D,
                                                bp.ls1, /*store=*/1, n, D,
                                                bp.ln2_w, bp.ln2_b, c.ln_eps, w.xn, stream)) != AP_OK)
                return rc;
            {
                ap::GemmArgs g{};
                g.A = w.xn; g.lda = D; g.W = wsel(m, bp.fc1); g.split = use_split(m); g.ldw = bp.fc1->ld;
                g.M = n; g.N = F1; g.K = D; g.bias = bp.fc1_b; g.out = w.hid; g.ldo = F1;
                if ((rc = ap::launch_gemm_impl(dt, swiglu ? ap::EPI_BIAS_STORE : (c.act == AP_ACT_QUICK_GELU ? ap::EPI_BIAS_QGELU : ap::EPI_BIAS_GELU), g, 128, 0, stream)) != AP_OK) return rc;
                if (swiglu && (rc = ap::launch_swiglu(dt, w.hid, n, H, w.hid2, stream)) != AP_OK) return rc;
            }
            {
                ap::GemmArgs g{};
                g.A = w.hid2; g.lda = H; g.W = wsel(m, bp.fc2); g.split = use_split(m); g.ldw = bp.fc2->ld;
                g.M = n; g.N = D; g.K = H; g.bias = bp.fc2_b; g.out = w.delta; g.ldo = D;
                if ((rc = ap::launch_gemm_impl(dt, ap::EPI_BIAS_STORE, g, 128, 0, stream)) != AP_OK) return rc;
            }
            pending = w.delta;
            pending_ls = bp.ls2;
            pending_stride = D;
            break;
        }
        {
            ap::GemmArgs g{};
            g.A = w.xn; g.lda = D; g.W = wsel(m, bp.qkv); g.split = use_split(m); g.ldw = bp.qkv->ld;
            g.M = M; g.N = 3 * DA; g.K = D; g.bias = bp.qkv_b; g.out = w.qkv; g.ldo = 3 * DA;
            ScopedTimer t(m, AP_PROF_GEMM_QKV, stream);
            if ((rc = ap::launch_gemm(dt, ap::EPI_BIAS_STORE, g, stream)) != AP_OK) return rc;
            if (c.rope && (rc = ap::launch_rope(dt, w.qkv, n, m->tokens, m->prefix, c.heads, m->hd, m->rope_cos, m->rope_sin, 3,
                                                stream)) != AP_OK) return rc;
        }
        { ScopedTimer t(m, AP_PROF_ATTENTION, stream);
          if (use_split(m) && n <= 65535) {
              // float32, split-f16 products: the fused float32 attention of the SAM2 operator set in its split form (image = window,
              // q | k | v at column offsets of the packed rows) -- same arithmetic class as the GEMMs around it; 46 -> 27.5 ms of a
              // 2048-tile ViT-B/16 step against the exact-f32 strip kernel (profiles/r06e_split_f16_attention.txt)
              const float* qp = (const float*)w.qkv;
              if ((rc = ap::launch_sattention(qp, 3 * DA, qp + DA, 3 * DA, qp + 2 * DA, 3 * DA, n, c.heads, m->tokens, m->tokens, m->hd,
                                              m->attn_scale, (float*)w.att, DA, stream, /*exact=*/false)) != AP_OK) return rc;
          } else if ((rc = ap::launch_attention(dt, w.qkv, w.att, n, m->tokens, c.heads, m->hd, m->attn_scale,
                                                stream)) != AP_OK) return rc; }
        {
            ap::GemmArgs g{};
            g.A = w.att; g.lda = DA; g.W = wsel(m, bp.proj); g.split = use_split(m); g.ldw = bp.proj->ld;
            g.M = M; g.N = D; g.K = DA; g.bias = bp.proj_b;
            g.out = w.delta2; g.ldo = D;
            ScopedTimer t(m, AP_PROF_GEMM_PROJ, stream);
            if ((rc = ap::launch_gemm(dt, ap::EPI_BIAS_STORE, g, stream)) != AP_OK) return rc;
        }
        { ScopedTimer t(m, AP_PROF_LAYERNORM, stream);
          if ((rc = ap::launch_add2_layernorm(dt, dt, w.tok, D, pending, D, pending_ls, w.delta2, D,
                                              bp.ls1, /*store=*/1, M, D, bp.ln2_w,
                                              bp.ln2_b, c.ln_eps, w.xn, stream)) != AP_OK) return rc; }
        {
            ap::GemmArgs g{};
            g.A = w.xn; g.lda = D; g.W = wsel(m, bp.fc1); g.split = use_split(m); g.ldw = bp.fc1->ld;
            g.M = M; g.N = F1; g.K = D; g.bias = bp.fc1_b; g.out = w.hid; g.ldo = F1;
            ScopedTimer t(m, AP_PROF_GEMM_FC1, stream);
            if ((rc = ap::launch_gemm(dt, swiglu ? ap::EPI_BIAS_STORE : (c.act == AP_ACT_QUICK_GELU ? ap::EPI_BIAS_QGELU : ap::EPI_BIAS_GELU), g, stream)) != AP_OK) return rc;
            if (swiglu && (rc = ap::launch_swiglu(dt, w.hid, M, H, w.hid2, stream)) != AP_OK) return rc;
        }
        {
            ap::GemmArgs g{};
            g.A = w.hid2; g.lda = H; g.W = wsel(m, bp.fc2); g.split = use_split(m); g.ldw = bp.fc2->ld;
            g.M = M; g.N = D; g.K = H; g.bias = bp.fc2_b;
            g.out = w.delta; g.ldo = D;
            ScopedTimer t(m, AP_PROF_GEMM_FC2, stream);
            if ((rc = ap::launch_gemm(dt, ap::EPI_BIAS_STORE, g, stream)) != AP_OK) return rc;
        }
        pending = w.delta;
        pending_ls = bp.ls2;
    }
    st.pending = pending; st.pending_ls = pending_ls;
    st.tok_stride = (long)m->tokens * D;
    st.pending_stride = cls_tail ? pending_stride : (long)m->tokens * D;
    return AP_OK;
}

// ---- block loop, fused LayerNorm (f16 / bf16 default).  The residual stream x lives in HBM in T and is the A operand
// of the qkv / fc1 GEMMs directly: their weights carry the LayerNorm gain and the epilogue applies the row statistics
// (EPI_NORM_*); the proj / fc2 GEMMs add their result to x in place and emit per-row partial sums of the new row
// (EPI_RESID_STATS) from which a tiny kernel builds the next statistics.  No standalone add+LayerNorm pass: per element
// and block the stream costs 2 x (2 B read + 2 B written) inside GEMM epilogues instead of 22 B in two streaming passes.
int blocks_fused(ap_vit* m, int n, const Workspace& w, StreamTail& st, hipStream_t stream) {
    const ap_vit_config& c = m->cfg;
    const int dt = c.compute_dtype, D = c.dim, M = n * m->tokens, G = D / 64;
    const int DA = m->dattn, H = c.mlp_dim, F1 = m->fc1_rows;
    const bool swiglu = c.mlp_type == AP_MLP_SWIGLU;
    const size_t es = ap::dtype_size(dt);
    int rc;
    const bool cls_tail = c.pool == AP_POOL_CLS && !m->full_last_block;
    // Exact class rows (default for the class-token poolings; AP_VIT_OPT_EXACT_CLS): the features ARE the class row of the
    // residual stream, and of the 16-bit stream's error in them almost all is the class row's OWN 2 x depth roundings (the
    // patch rows' reach it only through attention, averaged over the tokens: 1.26e-3 -> 7.7e-4 against the CPU fp32 path on
    // ViT-B/16, float16).  So the class rows are ALSO carried in f32 (cls32 [n, D]): every proj / fc2 launch leaves the class
    // rows' unrounded branch (accumulator + bias, f32) in cls_branch beside its normal output (GemmArgs::cls_branch), a tiny
    // kernel adds it to cls32, and the stream's class row -- the A operand of the next GEMM -- becomes T(cls32) with its
    // partial sums.  One 64-thread-per-group launch per GEMM; M >= 2^24 rows (the epilogue's row / tokens estimate) turns it off.
    const bool exact_cls = m->exact_cls && (c.pool == AP_POOL_CLS || c.pool == AP_POOL_CLS_MEAN) && M < (1 << 24);
    const long cls_stride = (long)m->tokens;         // rows between two images' class rows
    if (exact_cls) {
        ScopedTimer t(m, AP_PROF_CLS_TAIL, stream);
        // the class rows as the stream starts: the row the patch embedding (or CLIP's ln_pre) left in f32, or the shared prefix row
        if (c.pre_norm) {
            AP_HIP_CHECK(hipMemcpy2DAsync(w.cls32, (size_t)D * 4, w.tok, (size_t)cls_stride * D * 4, (size_t)D * 4, n,
                                          hipMemcpyDeviceToDevice, stream));
        } else if ((rc = ap::launch_cls_init(w.cls32, m->prefix_dev, 1, n, 1, D, stream)) != AP_OK) return rc;
    }
    // cls32 += the branch the launch before left in cls_branch, then stream row <- T(cls32)
    auto exact_cls_update = [&]() -> int {
        ScopedTimer t(m, AP_PROF_LAYERNORM, stream);
        return ap::launch_cls_exact_update(dt, w.cls32, w.cls_branch, n, m->tokens, D, w.x16, w.partial, stream);
    };
    // statistics of the new rows; with the exact class rows the same launch folds the branch into cls32 and re-rounds the
    // stream's class rows (launch_rowstats_finalize_cls)
    auto finalize_stats = [&]() -> int {
        ScopedTimer t(m, AP_PROF_LAYERNORM, stream);
        return ap::launch_rowstats_finalize_cls(w.partial, M, G, D, c.ln_eps, w.rowstats, dt, exact_cls ? w.cls32 : nullptr,
                                                w.cls_branch, w.x16, n, m->tokens, stream);
    };
    for (int i = 0; i < c.depth; ++i) {
        const ap::BlockParams& bp = m->blocks[i];
        const ap::FusedBlock& fb = m->fused[i];
        if (i == c.depth - 1 && cls_tail) {
            // ---- last block, CLS readout (see blocks_f32_stream): K and V of every token from the stream, then the CLS
            // rows alone continue on the f32 path (their stream rows widened to f32, plain LayerNorm launches, unfolded
            // weights on the 128x128 kernel).
            {
                ap::GemmArgs g{};
                g.A = w.x16; g.lda = D; g.W = (const char*)fb.qkv_w + (size_t)DA * bp.qkv->ld * es; g.ldw = bp.qkv->ld;
                g.M = M; g.N = 2 * DA; g.K = D; g.bias = fb.qkv_b + DA; g.colsum = fb.qkv_cs + DA; g.rowstats = w.rowstats;
                g.out = (char*)w.qkv + (size_t)DA * es; g.ldo = 3 * DA;
                ScopedTimer t(m, AP_PROF_GEMM_QKV, stream);
                if ((rc = ap::launch_gemm(dt, ap::EPI_NORM_STORE, g, stream)) != AP_OK) return rc;
                if (c.rope && (rc = ap::launch_rope(dt, w.qkv, n, m->tokens, m->prefix, c.heads, m->hd, m->rope_cos, m->rope_sin, 2,
                                                    stream)) != AP_OK) return rc;
            }
            ScopedTimer t(m, AP_PROF_CLS_TAIL, stream);
            if (exact_cls) AP_HIP_CHECK(hipMemcpyAsync(w.tok, w.cls32, (size_t)n * D * 4, hipMemcpyDeviceToDevice, stream));
            else if ((rc = ap::launch_stream_to_f32(dt, w.x16, (long)m->tokens * D, n, D, w.tok, stream)) != AP_OK) return rc;
            if ((rc = ap::launch_add2_layernorm(dt, dt, w.tok, D, nullptr, 0, nullptr, nullptr, 0, nullptr, /*store=*/0,
                                                n, D, bp.ln1_w, bp.ln1_b, c.ln_eps, w.xn, stream)) != AP_OK) return rc;
            char* q_cls = (char*)w.att;                                   // T [n, DA]
            char* a_cls = (char*)w.att + (size_t)n * DA * es;             // T [n, DA]
            {
                ap::GemmArgs g{};                                          // q for the CLS rows
                g.A = w.xn; g.lda = D; g.W = bp.qkv->dev; g.ldw = bp.qkv->ld;
                g.M = n; g.N = DA; g.K = D; g.bias = bp.qkv_b; g.out = q_cls; g.ldo = DA;
                if ((rc = ap::launch_gemm_impl(dt, ap::EPI_BIAS_STORE, g, 128, 0, stream)) != AP_OK) return rc;
            }
            if ((rc = ap::launch_attention_cls(dt, q_cls, w.qkv, 3 * DA, DA, 2 * DA, a_cls, n, m->tokens, c.heads,
                                               m->hd, m->attn_scale, stream)) != AP_OK) return rc;
            {
                ap::GemmArgs g{};
                g.A = a_cls; g.lda = DA; g.W = bp.proj->dev; g.ldw = bp.proj->ld;
                g.M = n; g.N = D; g.K = DA; g.bias = bp.proj_b; g.out = w.delta2; g.ldo = D;
                if ((rc = ap::launch_gemm_impl(dt, ap::EPI_BIAS_STORE, g, 128, 0, stream)) != AP_OK) return rc;
            }
            if ((rc = ap::launch_add2_layernorm(dt, dt, w.tok, D, nullptr, 0, nullptr, w.delta2, D, bp.ls1, /*store=*/1, n, D,
                                                bp.ln2_w, bp.ln2_b, c.ln_eps, w.xn, stream)) != AP_OK) return rc;
            {
                ap::GemmArgs g{};
                g.A = w.xn; g.lda = D; g.W = bp.fc1->dev; g.ldw = bp.fc1->ld;
                g.M = n; g.N = F1; g.K = D; g.bias = bp.fc1_b; g.out = w.hid; g.ldo = F1;
                if ((rc = ap::launch_gemm_impl(dt, swiglu ? ap::EPI_BIAS_STORE : (c.act == AP_ACT_QUICK_GELU ? ap::EPI_BIAS_QGELU : ap::EPI_BIAS_GELU), g, 128, 0, stream)) != AP_OK) return rc;
                if (swiglu && (rc = ap::launch_swiglu(dt, w.hid, n, H, w.hid2, stream)) != AP_OK) return rc;
            }
            {
                ap::GemmArgs g{};
                g.A = w.hid2; g.lda = H; g.W = bp.fc2->dev; g.ldw = bp.fc2->ld;
                g.M = n; g.N = D; g.K = H; g.bias = bp.fc2_b; g.out = w.delta; g.ldo = D;
                if ((rc = ap::launch_gemm_impl(dt, ap::EPI_BIAS_STORE, g, 128, 0, stream)) != AP_OK) return rc;
            }
            st.pending = w.delta; st.pending_ls = bp.ls2; st.pending_stride = D; st.tok_stride = D;
            return AP_OK;
        }
        {
            ap::GemmArgs g{};
            g.A = w.x16; g.lda = D; g.W = fb.qkv_w; g.ldw = bp.qkv->ld;
            g.M = M; g.N = 3 * DA; g.K = D; g.bias = fb.qkv_b; g.colsum = fb.qkv_cs; g.rowstats = w.rowstats;
            g.out = w.qkv; g.ldo = 3 * DA;
            ScopedTimer t(m, AP_PROF_GEMM_QKV, stream);
            if ((rc = ap::launch_gemm(dt, ap::EPI_NORM_STORE, g, stream)) != AP_OK) return rc;
            if (c.rope && (rc = ap::launch_rope(dt, w.qkv, n, m->tokens, m->prefix, c.heads, m->hd, m->rope_cos, m->rope_sin, 3,
                                                stream)) != AP_OK) return rc;
        }
        { ScopedTimer t(m, AP_PROF_ATTENTION, stream);
          if ((rc = ap::launch_attention(dt, w.qkv, w.att, n, m->tokens, c.heads, m->hd, m->attn_scale, stream)) != AP_OK) return rc; }
        {
            ap::GemmArgs g{};
            g.A = w.att; g.lda = DA; g.W = fb.proj_w; g.ldw = bp.proj->ld;
            g.M = M; g.N = D; g.K = DA; g.bias = fb.proj_b; g.out = w.x16; g.ldo = D; g.partial = w.partial;
            if (exact_cls) { g.cls_branch = w.cls_branch; g.cls_tokens = m->tokens; }
            ScopedTimer t(m, AP_PROF_GEMM_PROJ, stream);
            if ((rc = ap::launch_gemm(dt, ap::EPI_RESID_STATS, g, stream)) != AP_OK) return rc;
        }
        if ((rc = finalize_stats()) != AP_OK) return rc;
        {
            ap::GemmArgs g{};
            g.A = w.x16; g.lda = D; g.W = fb.fc1_w; g.ldw = bp.fc1->ld;
            g.M = M; g.N = F1; g.K = D; g.bias = fb.fc1_b; g.colsum = fb.fc1_cs; g.rowstats = w.rowstats;
            // SwiGLU: the folded fc1 weights are row-interleaved (x1 | x2 of the same 32 output columns in one wave's tile), so
            // the gate runs in the epilogue on the f32 values: out = hid2 [M, H] directly, one rounding
            g.out = swiglu ? w.hid2 : w.hid; g.ldo = swiglu ? H : F1;
            ScopedTimer t(m, AP_PROF_GEMM_FC1, stream);
            if ((rc = ap::launch_gemm(dt, swiglu ? ap::EPI_NORM_SWIGLU : (c.act == AP_ACT_QUICK_GELU ? ap::EPI_NORM_QGELU : ap::EPI_NORM_GELU), g, stream)) != AP_OK) return rc;
        }
        {
            ap::GemmArgs g{};
            g.A = w.hid2; g.lda = H; g.W = fb.fc2_w; g.ldw = bp.fc2->ld;
            g.M = M; g.N = D; g.K = H; g.bias = fb.fc2_b; g.out = w.x16; g.ldo = D; g.partial = w.partial;
            if (exact_cls) { g.cls_branch = w.cls_branch; g.cls_tokens = m->tokens; }
            ScopedTimer t(m, AP_PROF_GEMM_FC2, stream);
            if ((rc = ap::launch_gemm(dt, ap::EPI_RESID_STATS, g, stream)) != AP_OK) return rc;
        }
        if (i + 1 < c.depth) { if ((rc = finalize_stats()) != AP_OK) return rc; }
        else if (exact_cls && (rc = exact_cls_update()) != AP_OK) return rc;      // last block, every token: no statistics follow
    }
    // every block ran on the stream (attentional pooling, or AP_VIT_OPT_FULL_LAST_BLOCK): widen it for the final LayerNorm
    if ((rc = ap::launch_stream_to_f32(dt, w.x16, D, M, D, w.tok, stream)) != AP_OK) return rc;
    if (exact_cls)
        AP_HIP_CHECK(hipMemcpy2DAsync(w.tok, (size_t)cls_stride * D * 4, w.cls32, (size_t)D * 4, (size_t)D * 4, n,
                                      hipMemcpyDeviceToDevice, stream));
    st.pending = nullptr; st.pending_ls = nullptr; st.pending_stride = 0; st.tok_stride = (long)m->tokens * D;
    return AP_OK;
}

int run_blocks(ap_vit* m, int n, const Workspace& w, float* out, hipStream_t stream) {
    const ap_vit_config& c = m->cfg;
    const int dt = c.compute_dtype, D = c.dim, M = n * m->tokens;
    int rc;
    const bool fused = dt != AP_F32 && !m->f32_stream;
    // K of the patch-embed GEMM is kpe (3 ps^2 padded to 64): the persistent kernel wants K % 128 == 0, the 128 x 128 one K % 64
    if (c.pre_norm) {
        // CLIP ln_pre: the embedded tokens are normalised before the first block.  The f32 token matrix is built as in the
        // f32-stream dataflow, normalised in place (row-wise kernel: a row is read whole before it is written), and -- fused
        // dataflow -- rounded into the T stream together with its first row statistics (launch_stream_init)
        if ((rc = patch_embed(m, n, w, stream)) != AP_OK) return rc;
        { ScopedTimer t(m, AP_PROF_LAYERNORM, stream);
          if ((rc = ap::launch_layernorm(AP_F32, w.tok, D, M, D, m->pre_w, m->pre_b, c.ln_eps, w.tok, stream)) != AP_OK) return rc;
          if (fused && (rc = ap::launch_stream_init(dt, w.tok, M, D, c.ln_eps, w.x16, w.rowstats, stream)) != AP_OK) return rc; }
    } else if ((rc = fused ? patch_embed_stream(m, n, w, stream) : patch_embed(m, n, w, stream)) != AP_OK) return rc;
    StreamTail st;
    if ((rc = fused ? blocks_fused(m, n, w, st, stream) : blocks_f32_stream(m, n, w, st, stream)) != AP_OK) return rc;
    const void* pending = st.pending;
    const float* pending_ls = st.pending_ls;
    if (c.pool == AP_POOL_CLS && c.proj_dim > 0) {
        // CLIP: final LayerNorm on the class rows in the compute type (xn, T [n, D]), the bias-free visual projection on the
        // 128 x 128 kernel (att, T [n, P]), widened to f32.  (The reference's half-precision model does the same roundings.)
        const int P = c.proj_dim;
        if ((rc = ap::launch_add_layernorm(dt, dt, w.tok, st.tok_stride, pending, st.pending_stride, pending_ls, n, D, m->norm_w,
                                           m->norm_b, c.ln_eps, w.xn, stream)) != AP_OK) return rc;
        ap::GemmArgs g{};
        g.A = w.xn; g.lda = D; g.W = wsel(m, m->head_proj); g.split = use_split(m); g.ldw = m->head_proj->ld; g.M = n; g.N = P; g.K = D;
        g.bias = m->zero_bias; g.out = dt == AP_F32 ? (void*)out : w.att; g.ldo = P;
        if ((rc = ap::launch_gemm_impl(dt, ap::EPI_BIAS_STORE, g, 128, 0, stream)) != AP_OK) return rc;
        return dt == AP_F32 ? AP_OK : ap::launch_stream_to_f32(dt, w.att, P, n, P, out, stream);
    }
    if (c.pool == AP_POOL_CLS)
        // final LayerNorm on the CLS row of every image (last fc2 output folded in first) -> out f32 [n, D]
        return ap::launch_add_layernorm(dt, AP_F32, w.tok, st.tok_stride, pending, st.pending_stride, pending_ls,
                                        n, D, m->norm_w,
                                        m->norm_b, c.ln_eps, out, stream);

    if (c.pool == AP_POOL_CLS_MEAN) {
        // final LN on ALL tokens (f32 [M, D], reuses qkv), then [class token | mean of the patch tokens] -> out f32 [n, 2 D]
        float* y = (float*)w.qkv;
        if ((rc = ap::launch_add_layernorm(dt, AP_F32, w.tok, D, pending, D, pending_ls, M, D, m->norm_w, m->norm_b, c.ln_eps, y,
                                           stream)) != AP_OK) return rc;
        return ap::launch_cls_mean_pool(y, n, m->tokens, m->prefix, D, out, stream);
    }

    // ---- AP_POOL_ATTN (CONCH visual tower): final LN on ALL tokens, then the one-query attentional pooler.
    // Buffers: y f32 [M, D] reuses qkv, xk T [M, D] = xn, kv T [M, 2P] reuses hid, pooled T [n, P] = att,
    // o32 f32 [n, P] reuses delta.
    const int P = c.pool_dim;
    const ap::PoolParams& pp = m->pool;
    float* y = (float*)w.qkv;
    if ((rc = ap::launch_add_layernorm(dt, AP_F32, w.tok, D, pending, D, pending_ls, M, D,
                                       m->norm_w,
                                       m->norm_b, c.ln_eps, y, stream)) != AP_OK) return rc;
    if ((rc = ap::launch_layernorm(dt, y, D, M, D, pp.ln_k_w, pp.ln_k_b, c.pool_ln_eps, w.xn,
                                   stream)) != AP_OK) return rc;
    {
        const Param* wkv = pp.kv;
        ap::GemmArgs g{};
        g.A = w.xn; g.lda = D; g.W = wkv->dev; g.ldw = wkv->ld; g.M = M; g.N = 2 * P; g.K = D;
        g.bias = pp.kv_b; g.out = w.hid; g.ldo = 2 * P;
        if ((rc = ap::launch_gemm(dt, ap::EPI_BIAS_STORE, g, stream)) != AP_OK) return rc;
    }
    if ((rc = ap::launch_attn_pool(dt, w.hid, pp.q, w.att, n, m->tokens, c.pool_heads, stream)) != AP_OK) return rc;
    float* o32 = (float*)w.delta;
    AP_HIP_CHECK(hipMemsetAsync(o32, 0, (size_t)n * P * sizeof(float), stream));
    {
        const Param* wo = pp.out;
        ap::GemmArgs g{};
        g.A = w.att; g.lda = P; g.W = wo->dev; g.ldw = wo->ld; g.M = n; g.N = P; g.K = P;
        g.bias = pp.out_b; g.out = o32; g.ldo = P;
        if ((rc = ap::launch_gemm(dt, ap::EPI_BIAS_RESID, g, stream)) != AP_OK) return rc;    // 0 + (acc + bias), f32
    }
    return ap::launch_layernorm(AP_F32, o32, P, n, P, pp.ln_out_w, pp.ln_out_b, c.pool_ln_eps, out, stream);
}

// AP_VIT_OPT_SPLIT_F16: every matrix of a float32 encoder also as [hi 32 | lo 32] f16 rows (same bytes again: ViT-B 0.34 GB)
int refresh_split(Param& p, hipStream_t stream) {
    return ap::launch_split_f16_weights((const float*)p.dev, p.split, (size_t)p.rows * p.ld, stream);
}
int build_split(ap_vit* m) {
    for (auto& kv : m->params) {
        Param& p = kv.second;
        if (!p.matrix || p.split) continue;
        AP_HIP_CHECK(hipMalloc(&p.split, (size_t)p.rows * p.ld * sizeof(float)));
        int rc = refresh_split(p, nullptr);
        if (rc != AP_OK) return rc;
    }
    AP_HIP_CHECK(hipDeviceSynchronize());
    return AP_OK;
}

int check_forward_args(const ap_vit* m, int n, const void* in, const float* out, const void* ws,
                       size_t ws_bytes) {
    AP_REQUIRE(m != nullptr, "vit: null handle");
    if (!m->finalized) { ap::set_error("vit: ap_vit_finalize has not been called"); return AP_ERR_STATE; }
    AP_REQUIRE(n >= 0, "vit: negative batch");
    if (n == 0) return AP_OK;
    AP_REQUIRE(in && out && ws, "vit: null buffer");
    AP_REQUIRE(((uintptr_t)ws & 255) == 0, "vit: workspace must be 256-byte aligned");
    if (ws_bytes < ap_vit_workspace_bytes(m, n)) {
        ap::set_error("vit: workspace %zu bytes < required %zu", ws_bytes, ap_vit_workspace_bytes(m, n));
        return AP_ERR_WORKSPACE;
    }
    return AP_OK;
}

}  // namespace

extern "C" {

size_t ap_sizeof_vit_config(void) { return sizeof(ap_vit_config); }

int ap_vit_config_init(ap_vit_config* cfg, size_t sizeof_caller) {
    AP_REQUIRE(cfg, "vit_config_init: null argument");
    AP_REQUIRE(sizeof_caller >= AP_VIT_CONFIG_SIZE_V20 && sizeof_caller % 4 == 0 && sizeof_caller <= 4096,
               "vit_config_init: %zu is not the size of an ap_vit_config (ABI v20: %u bytes, this library: %zu)", sizeof_caller,
               AP_VIT_CONFIG_SIZE_V20, sizeof(ap_vit_config));
    memset(cfg, 0, sizeof_caller);
    cfg->struct_size = (uint32_t)sizeof_caller;
    return AP_OK;
}

int ap_vit_create(const ap_vit_config* cfg, ap_vit** out) {
    AP_REQUIRE(cfg && out, "vit_create: null argument");
    static_assert(sizeof(ap_vit_config) == AP_VIT_CONFIG_SIZE_V20, "ap_vit_config grew: append only, keep AP_VIT_CONFIG_SIZE_V20, and drop this assert");
    // ---- growth-safe hand-over (ABI v20): never read a byte the caller did not declare, and only sizes this structure has had.
    // A binding written for ABI <= 19 (no size member, image_size first) presents 224 / 448 / 518 here: refused unread.
    static const size_t known_sizes[] = {AP_VIT_CONFIG_SIZE_V20};          // append sizeof(ap_vit_config) of every later ABI
    const size_t given = cfg->struct_size;
    bool known = false;
    for (size_t k : known_sizes) known = known || given == k;
    if (!known && given > sizeof(ap_vit_config) && given % 4 == 0 && given <= 4096 && (given - sizeof(ap_vit_config)) <= 64) {
        ap::set_error("vit_create: cfg->struct_size = %zu is larger than this library's ap_vit_config (%zu bytes, ABI %d): the binding was "
                      "generated from a newer include/atlaspatch_hip.h than the library was built from", given, sizeof(ap_vit_config),
                      AP_ABI_VERSION);
        return AP_ERR_UNSUPPORTED;
    }
    AP_REQUIRE(known,
               "vit_create: cfg->struct_size = %zu is not a size ap_vit_config has had (ABI v20: %u bytes; this library: %zu): fill the "
               "structure with ap_vit_config_init(&cfg, sizeof cfg); a binding written for ABI <= 19 (no struct_size member, "
               "image_size first) must be regenerated from include/atlaspatch_hip.h", given, AP_VIT_CONFIG_SIZE_V20, sizeof(ap_vit_config));
    ap_vit_config c;
    memset(&c, 0, sizeof(c));
    memcpy(&c, cfg, given < sizeof(c) ? given : sizeof(c));        // an older caller's missing tail stays zero = the old behaviour
    c.struct_size = (uint32_t)sizeof(c);
    AP_REQUIRE(c.image_size > 0 && c.patch_size > 0 && c.image_size % c.patch_size == 0,
               "vit_create: image %d / patch %d", c.image_size, c.patch_size);
    AP_REQUIRE(c.patch_size >= 4 && c.patch_size <= 32 && c.patch_size % 2 == 0,
               "vit_create: patch size %d unsupported by this build (even, 4 .. 32)", c.patch_size);
    AP_REQUIRE(c.dim > 0 && c.heads > 0 && c.dim % c.heads == 0, "vit_create: dim %d / heads %d", c.dim, c.heads);
    const int hd = c.head_dim > 0 ? c.head_dim : c.dim / c.heads;
    AP_REQUIRE(hd == 64 || hd == 128 || (hd == 96 && c.compute_dtype != AP_F32),
               "vit_create: head_dim %d: q / k / v heads must be stored 64, 96 (f16 / bf16) or 128 wide (zero-pad other widths and set attn_scale)", hd);
    AP_REQUIRE(c.dim % 128 == 0 && c.mlp_dim % 128 == 0, "vit_create: dim and mlp_dim must be multiples of 128");
    AP_REQUIRE(c.reg_tokens >= 0 && c.reg_tokens <= 64, "vit_create: reg_tokens %d", c.reg_tokens);
    AP_REQUIRE(c.mlp_type == AP_MLP_GELU || c.mlp_type == AP_MLP_SWIGLU, "vit_create: mlp_type %d", c.mlp_type);
    AP_REQUIRE(c.attn_scale >= 0.f, "vit_create: attn_scale %g", (double)c.attn_scale);
    AP_REQUIRE(c.depth > 0, "vit_create: depth %d", c.depth);
    AP_REQUIRE(c.compute_dtype == AP_F16 || c.compute_dtype == AP_BF16 || c.compute_dtype == AP_F32,
               "vit_create: compute dtype %d", c.compute_dtype);
    AP_REQUIRE(c.pool == AP_POOL_CLS || c.pool == AP_POOL_ATTN || c.pool == AP_POOL_CLS_MEAN, "vit_create: pool %d", c.pool);
    if (c.pool == AP_POOL_ATTN) {
        AP_REQUIRE(c.compute_dtype != AP_F32, "vit_create: the attentional pooler runs in float16 / bfloat16 only");
        AP_REQUIRE(c.pool_heads > 0 && c.pool_dim == c.pool_heads * 64 && c.pool_dim % 128 == 0 &&
                   2 * c.pool_dim <= c.mlp_dim && 2 * c.pool_dim <= 3 * c.dim,
                   "vit_create: pool_dim %d / pool_heads %d (heads of 64, multiple of 128)", c.pool_dim, c.pool_heads);
    }
    AP_REQUIRE(c.rope == 0 || (c.head_dim == 0 || c.head_dim * c.heads == c.dim), "vit_create: rope needs the true head width");
    AP_REQUIRE(c.act == AP_ACT_GELU || (c.act == AP_ACT_QUICK_GELU && c.mlp_type == AP_MLP_GELU), "vit_create: act %d", c.act);
    AP_REQUIRE(c.proj_dim == 0 || (c.pool == AP_POOL_CLS && c.proj_dim % 128 == 0 && c.proj_dim <= c.dim),
               "vit_create: proj_dim %d (class-token pooling, a multiple of 128, at most dim)", c.proj_dim);
    const int g = c.image_size / c.patch_size;
    const int prefix = 1 + c.reg_tokens;
    AP_REQUIRE(c.compute_dtype != AP_F32 || (prefix + g * g <= 288 && hd == 64),
               "vit_create: %d tokens / head width %d exceed the float32 attention kernel's limits (288 tokens, 64 wide); use "
               "float16 / bfloat16", prefix + g * g, hd);
    ap_vit* m = new ap_vit();
    m->cfg = c;
    m->grid = g; m->patches = g * g; m->prefix = prefix; m->tokens = prefix + g * g;
    m->pos_rows = c.no_embed_class ? g * g : m->tokens;
    m->pos_row0 = c.no_embed_class ? 0 : prefix;
    m->hd = hd; m->dattn = c.heads * hd;
    m->attn_scale = c.attn_scale > 0.f ? c.attn_scale : 1.0f / sqrtf((float)hd);
    m->fc1_rows = c.mlp_type == AP_MLP_SWIGLU ? 2 * c.mlp_dim : c.mlp_dim;
    m->kpe = (int)ap::align_up(3 * c.patch_size * c.patch_size, 64);
    AP_HIP_CHECK(hipGetDevice(&m->device));
    m->full_last_block = getenv("AP_VIT_FULL_LAST_BLOCK") != nullptr;     // defaults only; ap_vit_set_option changes them
    m->two_half_overlap = getenv("AP_VIT_OVERLAP") != nullptr;
    m->f32_stream = getenv("AP_VIT_F32_STREAM") != nullptr;
    m->exact_cls = getenv("AP_VIT_NO_EXACT_CLS") == nullptr;
    m->split_f16 = c.compute_dtype == AP_F32 && getenv("AP_VIT_EXACT_F32") == nullptr;      // float32: split-f16 products unless asked otherwise
    int rc = AP_OK;
    auto add = [&](const std::string& name, int rows, int cols, bool matrix) {
        if (rc == AP_OK) rc = alloc_param(m, name, rows, cols, matrix);
    };
    const int D = c.dim;
    add("patch_embed.weight", D, 3 * c.patch_size * c.patch_size, true);
    add("patch_embed.bias", 1, D, false);
    add("cls_token", 1, D, false);
    if (c.reg_tokens > 0) add("reg_tokens", c.reg_tokens, D, false);
    add("pos_embed", m->pos_rows, D, false);
    add("norm.weight", 1, D, false);
    add("norm.bias", 1, D, false);
    if (c.pre_norm) { add("pre_norm.weight", 1, D, false); add("pre_norm.bias", 1, D, false); }
    if (c.rope) { add("rope.cos", m->patches, hd, false); add("rope.sin", m->patches, hd, false); }
    if (c.proj_dim > 0) {
        add("head_proj.weight", c.proj_dim, D, true);
        if (rc == AP_OK && (hipMalloc((void**)&m->zero_bias, (size_t)c.proj_dim * sizeof(float)) != hipSuccess ||
                            hipMemset(m->zero_bias, 0, (size_t)c.proj_dim * sizeof(float)) != hipSuccess)) {
            ap::set_error("vit_create: hipMalloc of the projection's zero bias failed");
            rc = AP_ERR_HIP;
        }
    }
    for (int i = 0; i < c.depth; ++i) {
        const std::string b = "blocks." + std::to_string(i) + ".";
        add(b + "ln1.weight", 1, D, false); add(b + "ln1.bias", 1, D, false);
        add(b + "qkv.weight", 3 * m->dattn, D, true); add(b + "qkv.bias", 1, 3 * m->dattn, false);
        add(b + "proj.weight", D, m->dattn, true); add(b + "proj.bias", 1, D, false);
        add(b + "ln2.weight", 1, D, false); add(b + "ln2.bias", 1, D, false);
        add(b + "fc1.weight", m->fc1_rows, D, true); add(b + "fc1.bias", 1, m->fc1_rows, false);
        add(b + "fc2.weight", D, c.mlp_dim, true); add(b + "fc2.bias", 1, D, false);
        if (c.layer_scale) { add(b + "ls1", 1, D, false); add(b + "ls2", 1, D, false); }
    }
    if (c.pool == AP_POOL_ATTN) {
        const int P = c.pool_dim;
        add("attn_pool.ln_k.weight", 1, D, false); add("attn_pool.ln_k.bias", 1, D, false);
        add("attn_pool.kv.weight", 2 * P, D, true); add("attn_pool.kv.bias", 1, 2 * P, false);
        add("attn_pool.q", 1, P, false);
        add("attn_pool.out.weight", P, P, true); add("attn_pool.out.bias", 1, P, false);
        add("attn_pool.ln_out.weight", 1, P, false); add("attn_pool.ln_out.bias", 1, P, false);
    }
    if (rc == AP_OK && m->split_f16) rc = build_split(m);
    if (rc != AP_OK) { ap_vit_destroy(m); return rc; }
    *out = m;
    return AP_OK;
}

void ap_vit_destroy(ap_vit* m) {
    if (!m) return;
    for (auto& kv : m->params) {
        if (kv.second.dev) (void)hipFree(kv.second.dev);
        if (kv.second.dev32) (void)hipFree(kv.second.dev32);
        if (kv.second.split) (void)hipFree(kv.second.split);
    }
    for (void* p : m->fused_allocs) (void)hipFree(p);
    if (m->prefix_dev) (void)hipFree(m->prefix_dev);
    if (m->zero_bias) (void)hipFree(m->zero_bias);
    for (hipEvent_t e : m->ev_pool) (void)hipEventDestroy(e);
    if (m->ev_fork) (void)hipEventDestroy(m->ev_fork);
    if (m->ev_join) (void)hipEventDestroy(m->ev_join);
    if (m->side) (void)hipStreamDestroy(m->side);
    delete m;
}

int ap_vit_set_param(ap_vit* m, const char* name, const float* host, size_t count) {
    AP_REQUIRE(m && name && host, "vit_set_param: null argument");
    auto it = m->params.find(name);
    AP_REQUIRE(it != m->params.end(), "vit_set_param: unknown parameter '%s'", name);
    Param& p = it->second;
    AP_REQUIRE(count == p.count, "vit_set_param: '%s' expects %zu values, got %zu", name, p.count, count);
    if (!p.matrix) {
        AP_HIP_CHECK(hipMemcpy(p.dev, host, count * sizeof(float), hipMemcpyHostToDevice));
    } else {
        float* tmp = nullptr;
        AP_HIP_CHECK(hipMalloc((void**)&tmp, (size_t)p.rows * p.ld * sizeof(float)));
        AP_HIP_CHECK(hipMemset(tmp, 0, (size_t)p.rows * p.ld * sizeof(float)));
        AP_HIP_CHECK(hipMemcpy2D(tmp, (size_t)p.ld * sizeof(float), host, (size_t)p.cols * sizeof(float),
                                 (size_t)p.cols * sizeof(float), p.rows, hipMemcpyHostToDevice));
        int rc = ap::launch_convert(m->cfg.compute_dtype, tmp, p.dev, (size_t)p.rows * p.ld, nullptr);
        if (rc == AP_OK && p.split) rc = refresh_split(p, nullptr);
        if (rc != AP_OK) { (void)hipFree(tmp); return rc; }
        AP_HIP_CHECK(hipDeviceSynchronize());
        if (p.dev32) { (void)hipFree(p.dev32); p.dev32 = nullptr; }
        // 16-bit compute types: ap_vit_finalize folds the LayerNorm gains / LayerScale into the block weights from the
        // f32 values (one rounding), so the upload is kept until then
        const bool is_block = strncmp(name, "blocks.", 7) == 0;
        if (m->cfg.compute_dtype != AP_F32 && is_block) p.dev32 = tmp;
        else AP_HIP_CHECK(hipFree(tmp));
    }
    p.set = true;
    const bool was_finalized = m->finalized;
    if (strcmp(name, "cls_token") == 0 || strcmp(name, "reg_tokens") == 0 || strcmp(name, "pos_embed") == 0)
        m->finalized = false;              // the prefix rows (class / register tokens + position rows) are rebuilt by finalize
    if (was_finalized && m->cfg.compute_dtype != AP_F32 &&
        (strncmp(name, "blocks.", 7) == 0 || strcmp(name, "pos_embed") == 0)) {
        // the folded weights / the T copy of the position embedding are stale now: a forward needs a new ap_vit_finalize
        m->fold_dirty = true;
        m->finalized = false;
    }
    return AP_OK;
}

// Every parameter of a checkpoint in ONE call: the per-tensor work then runs outside the interpreter lock of a Python
// caller (the encoder is built on a side thread while the CLI's phase 1 runs, services/feature_embedding.py), and it is
// pipelined instead of synchronous per tensor: the host pages (typically an mmap of the checkpoint file) are copied into one
// of two pinned staging buffers by this thread while the previous buffer's H2D copy and f32 -> T conversion run on a private
// stream; one synchronisation at the end.  Same result as n ap_vit_set_param calls in order.
int ap_vit_set_params(ap_vit* m, const char* const* names, const float* const* hosts, const size_t* counts, int n) {
    AP_REQUIRE(m && (n == 0 || (names && hosts && counts)), "vit_set_params: null argument");
    size_t largest = 0;
    for (int i = 0; i < n; ++i) {
        AP_REQUIRE(names[i] && hosts[i], "vit_set_params: null entry %d", i);
        auto it = m->params.find(names[i]);
        AP_REQUIRE(it != m->params.end(), "vit_set_param: unknown parameter '%s'", names[i]);
        AP_REQUIRE(counts[i] == it->second.count, "vit_set_param: '%s' expects %zu values, got %zu", names[i], it->second.count,
                   counts[i]);
        largest = counts[i] > largest ? counts[i] : largest;
    }
    if (n == 0) return AP_OK;
    struct Stage { float* pin = nullptr; hipEvent_t done = nullptr; bool busy = false; } st[2];
    hipStream_t stream = nullptr;
    int rc = AP_OK;
    auto fail = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && rc == AP_OK) { ap::set_error("vit_set_params: %s failed: %s", what, hipGetErrorString(e)); rc = AP_ERR_HIP; }
        return e != hipSuccess;
    };
    if (fail(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), "hipStreamCreate")) return rc;
    for (auto& b : st)
        if (fail(hipHostMalloc((void**)&b.pin, largest * sizeof(float), hipHostMallocDefault), "hipHostMalloc") ||
            fail(hipEventCreateWithFlags(&b.done, hipEventDisableTiming), "hipEventCreate"))
            break;
    for (int i = 0; i < n && rc == AP_OK; ++i) {
        Param& p = m->params.find(names[i])->second;
        Stage& b = st[i & 1];
        if (b.busy && fail(hipEventSynchronize(b.done), "hipEventSynchronize")) break;
        memcpy(b.pin, hosts[i], counts[i] * sizeof(float));
        if (!p.matrix) {
            if (fail(hipMemcpyAsync(p.dev, b.pin, counts[i] * sizeof(float), hipMemcpyHostToDevice, stream), "hipMemcpyAsync")) break;
        } else {
            float* tmp = nullptr;
            const size_t padded = (size_t)p.rows * p.ld;
            if (fail(hipMalloc((void**)&tmp, padded * sizeof(float)), "hipMalloc")) break;
            if (p.ld != p.cols && fail(hipMemsetAsync(tmp, 0, padded * sizeof(float), stream), "hipMemsetAsync")) { (void)hipFree(tmp); break; }
            if (fail(hipMemcpy2DAsync(tmp, (size_t)p.ld * sizeof(float), b.pin, (size_t)p.cols * sizeof(float),
                                      (size_t)p.cols * sizeof(float), p.rows, hipMemcpyHostToDevice, stream), "hipMemcpy2DAsync")) {
                (void)hipFree(tmp); break;
            }
            int crc = ap::launch_convert(m->cfg.compute_dtype, tmp, p.dev, padded, stream);
            if (crc == AP_OK && p.split) crc = refresh_split(p, stream);
            if (crc != AP_OK) { rc = crc; (void)hipFree(tmp); break; }
            if (p.dev32) { (void)hipFree(p.dev32); p.dev32 = nullptr; }     // (hipFree waits for the device: a re-upload is rare)
            const bool is_block = strncmp(names[i], "blocks.", 7) == 0;
            if (m->cfg.compute_dtype != AP_F32 && is_block) p.dev32 = tmp;   // kept until ap_vit_finalize has folded it
            else {
                // freed once the stream is done with it: collect and release after the final synchronisation
                p.dev32 = nullptr;
                m->pending_free.push_back(tmp);
            }
        }
        if (fail(hipEventRecord(b.done, stream), "hipEventRecord")) break;
        b.busy = true;
        p.set = true;
        const bool was_finalized = m->finalized;
        if (strcmp(names[i], "cls_token") == 0 || strcmp(names[i], "reg_tokens") == 0 || strcmp(names[i], "pos_embed") == 0)
            m->finalized = false;
        if (was_finalized && m->cfg.compute_dtype != AP_F32 &&
            (strncmp(names[i], "blocks.", 7) == 0 || strcmp(names[i], "pos_embed") == 0)) {
            m->fold_dirty = true;
            m->finalized = false;
        }
    }
    if (stream) fail(hipStreamSynchronize(stream), "hipStreamSynchronize");
    for (float* t : m->pending_free) (void)hipFree(t);
    m->pending_free.clear();
    for (auto& b : st) {
        if (b.pin) (void)hipHostFree(b.pin);
        if (b.done) (void)hipEventDestroy(b.done);
    }
    if (stream) (void)hipStreamDestroy(stream);
    return rc;
}

int ap_vit_finalize(ap_vit* m) {
    AP_REQUIRE(m, "vit_finalize: null handle");
    for (auto& kv : m->params)
        if (!kv.second.set) {
            ap::set_error("vit_finalize: parameter '%s' was never set", kv.first.c_str());
            return AP_ERR_STATE;
        }
    auto vec = [&](const std::string& name) -> const float* {
        const Param* p = find(m, name);
        return p ? (const float*)p->dev : nullptr;
    };
    m->pe_w = find(m, "patch_embed.weight");
    m->pe_b = vec("patch_embed.bias"); m->cls = vec("cls_token"); m->pos = vec("pos_embed");
    m->norm_w = vec("norm.weight"); m->norm_b = vec("norm.bias");
    m->pre_w = vec("pre_norm.weight"); m->pre_b = vec("pre_norm.bias");
    m->rope_cos = vec("rope.cos"); m->rope_sin = vec("rope.sin");
    m->head_proj = m->cfg.proj_dim > 0 ? find(m, "head_proj.weight") : nullptr;
    {   // class / register token rows with their position rows folded in (f32; rebuilt on every finalize: it is tiny)
        if (!m->prefix_dev) AP_HIP_CHECK(hipMalloc((void**)&m->prefix_dev, (size_t)m->prefix * m->cfg.dim * sizeof(float)));
        int prc = ap::launch_prefix_build(m->cls, vec("reg_tokens"), m->cfg.reg_tokens, m->cfg.no_embed_class ? nullptr : m->pos,
                                          m->cfg.dim, m->prefix_dev, nullptr);
        if (prc != AP_OK) return prc;
        // the build ran on the legacy stream: a first forward on a non-blocking stream is not ordered behind it, and two of
        // the three ways out of this function (float32; nothing to fold again) used to return without waiting
        AP_HIP_CHECK(hipDeviceSynchronize());
    }
    m->blocks.resize(m->cfg.depth);
    for (int i = 0; i < m->cfg.depth; ++i) {
        const std::string b = "blocks." + std::to_string(i) + ".";
        ap::BlockParams& bp = m->blocks[i];
        bp.ln1_w = vec(b + "ln1.weight"); bp.ln1_b = vec(b + "ln1.bias");
        bp.ln2_w = vec(b + "ln2.weight"); bp.ln2_b = vec(b + "ln2.bias");
        bp.qkv_b = vec(b + "qkv.bias"); bp.proj_b = vec(b + "proj.bias");
        bp.fc1_b = vec(b + "fc1.bias"); bp.fc2_b = vec(b + "fc2.bias");
        bp.ls1 = m->cfg.layer_scale ? vec(b + "ls1") : nullptr;
        bp.ls2 = m->cfg.layer_scale ? vec(b + "ls2") : nullptr;
        bp.qkv = find(m, b + "qkv.weight"); bp.proj = find(m, b + "proj.weight");
        bp.fc1 = find(m, b + "fc1.weight"); bp.fc2 = find(m, b + "fc2.weight");
    }
    if (m->cfg.pool == AP_POOL_ATTN) {
        ap::PoolParams& pp = m->pool;
        pp.ln_k_w = vec("attn_pool.ln_k.weight"); pp.ln_k_b = vec("attn_pool.ln_k.bias");
        pp.kv = find(m, "attn_pool.kv.weight"); pp.kv_b = vec("attn_pool.kv.bias"); pp.q = vec("attn_pool.q");
        pp.out = find(m, "attn_pool.out.weight"); pp.out_b = vec("attn_pool.out.bias");
        pp.ln_out_w = vec("attn_pool.ln_out.weight"); pp.ln_out_b = vec("attn_pool.ln_out.bias");
    }
    // ---- fused-LayerNorm weights (f16 / bf16): folded once from the f32 uploads, which are released afterwards
    bool any32 = false;
    for (auto& kv : m->params) any32 = any32 || kv.second.dev32 != nullptr;
    if (m->cfg.compute_dtype != AP_F32 && !any32 && !m->fused.empty() && !m->fold_dirty) {
        m->finalized = true;           // finalize called again without new uploads: the folded weights stand
        return AP_OK;
    }
    for (void* p : m->fused_allocs) (void)hipFree(p);
    m->fused_allocs.clear();
    m->fused.clear();
    if (m->cfg.compute_dtype != AP_F32) {
        const int dt = m->cfg.compute_dtype, D = m->cfg.dim, H = m->cfg.mlp_dim, DA = m->dattn, F1 = m->fc1_rows;
        const size_t es = ap::dtype_size(dt);
        m->fused.resize(m->cfg.depth);
        int rc = AP_OK;
        auto dalloc = [&](size_t bytes) -> void* {
            void* p = nullptr;
            if (rc == AP_OK && hipMalloc(&p, bytes) != hipSuccess) { ap::set_error("vit_finalize: hipMalloc of %zu bytes failed", bytes); rc = AP_ERR_HIP; }
            if (p) m->fused_allocs.push_back(p);
            return p;
        };
        for (int i = 0; i < m->cfg.depth && rc == AP_OK; ++i) {
            const ap::BlockParams& bp = m->blocks[i];
            ap::FusedBlock& fb = m->fused[i];
            for (const Param* p : {bp.qkv, bp.proj, bp.fc1, bp.fc2})
                if (!p->dev32) { ap::set_error("vit_finalize: block %d: a block parameter (or pos_embed) changed after a finalize -- the LayerNorm / LayerScale folding needs the float32 values of all four matrices of every block again: upload qkv / proj / fc1 / fc2 weights before finalizing", i); return AP_ERR_STATE; }
            fb.qkv_w = dalloc((size_t)3 * DA * bp.qkv->ld * es); fb.qkv_cs = (float*)dalloc(3 * DA * 4); fb.qkv_b = (float*)dalloc(3 * DA * 4);
            fb.fc1_w = dalloc((size_t)F1 * bp.fc1->ld * es); fb.fc1_cs = (float*)dalloc(F1 * 4); fb.fc1_b = (float*)dalloc(F1 * 4);
            fb.proj_w = dalloc((size_t)D * bp.proj->ld * es); fb.proj_b = (float*)dalloc(D * 4);
            fb.fc2_w = dalloc((size_t)D * bp.fc2->ld * es); fb.fc2_b = (float*)dalloc(D * 4);
            if (rc != AP_OK) break;
            if ((rc = ap::launch_fold_ln(dt, bp.qkv->dev32, 3 * DA, D, bp.qkv->ld, bp.ln1_w, bp.ln1_b, bp.qkv_b, fb.qkv_w, fb.qkv_cs, fb.qkv_b, nullptr)) != AP_OK) break;
            if ((rc = ap::launch_fold_ln(dt, bp.fc1->dev32, F1, D, bp.fc1->ld, bp.ln2_w, bp.ln2_b, bp.fc1_b, fb.fc1_w, fb.fc1_cs, fb.fc1_b, nullptr,
                                         m->cfg.mlp_type == AP_MLP_SWIGLU ? H : 0)) != AP_OK) break;
            if ((rc = ap::launch_fold_ls(dt, bp.proj->dev32, D, DA, bp.proj->ld, bp.ls1, bp.proj_b, fb.proj_w, fb.proj_b, nullptr)) != AP_OK) break;
            if ((rc = ap::launch_fold_ls(dt, bp.fc2->dev32, D, H, bp.fc2->ld, bp.ls2, bp.fc2_b, fb.fc2_w, fb.fc2_b, nullptr)) != AP_OK) break;
        }
        if (rc == AP_OK) {
            m->pos16 = dalloc((size_t)m->pos_rows * D * es);
            if (rc == AP_OK) rc = ap::launch_convert(dt, m->pos, m->pos16, (size_t)m->pos_rows * D, nullptr);
        }
        if (rc != AP_OK) return rc;
        AP_HIP_CHECK(hipDeviceSynchronize());
        for (auto& kv : m->params)
            if (kv.second.dev32) { (void)hipFree(kv.second.dev32); kv.second.dev32 = nullptr; }
    }
    m->fold_dirty = false;
    m->finalized = true;
    return AP_OK;
}

int ap_vit_set_option(ap_vit* m, int option, int value) {
    AP_REQUIRE(m, "vit_set_option: null handle");
    if (option == AP_VIT_OPT_FULL_LAST_BLOCK) m->full_last_block = value != 0;
    else if (option == AP_VIT_OPT_TWO_HALF_OVERLAP) m->two_half_overlap = value != 0;
    else if (option == AP_VIT_OPT_F32_STREAM) m->f32_stream = value != 0;
    else if (option == AP_VIT_OPT_EXACT_CLS) m->exact_cls = value != 0;
    else if (option == AP_VIT_OPT_SPLIT_F16) {
        AP_REQUIRE(value == 0 || m->cfg.compute_dtype == AP_F32, "vit_set_option: AP_VIT_OPT_SPLIT_F16 is a mode of the float32 compute type");
        if (value != 0) { const int rc = build_split(m); if (rc != AP_OK) return rc; }
        m->split_f16 = value != 0;
    }
    else { ap::set_error("vit_set_option: unknown option %d", option); return AP_ERR_INVALID; }
    return AP_OK;
}

size_t ap_vit_workspace_bytes(const ap_vit* m, int n) {
    if (!m || n <= 0) return 0;
    const size_t whole = carve(m, n, nullptr).total;
    const int na = (n + 1) / 2;                       // the two-half mode carves two independent workspaces
    const size_t halves = carve(m, na, nullptr).total + (n - na > 0 ? carve(m, n - na, nullptr).total : 0);
    return whole > halves ? whole : halves;
}

int ap_vit_embed_dim(const ap_vit* m) {
    if (!m) return 0;
    if (m->cfg.proj_dim > 0) return m->cfg.proj_dim;
    return m->cfg.pool == AP_POOL_ATTN ? m->cfg.pool_dim : (m->cfg.pool == AP_POOL_CLS_MEAN ? 2 * m->cfg.dim : m->cfg.dim);
}

int ap_vit_profile_enable(ap_vit* m, int on) {
    AP_REQUIRE(m, "vit_profile_enable: null handle");
    m->profile = on != 0;
    m->ev_used.clear();
    m->ev_next = 0;
    return AP_OK;
}

int ap_vit_profile_read(ap_vit* m, double* ms_by_kind, long long* launches_by_kind, int kinds) {
    AP_REQUIRE(m && ms_by_kind && launches_by_kind && kinds >= AP_PROF_KINDS, "vit_profile_read: bad arguments");
    for (int k = 0; k < kinds; ++k) { ms_by_kind[k] = 0.0; launches_by_kind[k] = 0; }
    for (auto& u : m->ev_used) {
        AP_HIP_CHECK(hipEventSynchronize(u.second.second));
        float ms = 0.f;
        AP_HIP_CHECK(hipEventElapsedTime(&ms, u.second.first, u.second.second));
        ms_by_kind[u.first] += ms;
        launches_by_kind[u.first] += 1;
    }
    m->ev_used.clear();
    m->ev_next = 0;
    return AP_OK;
}

int ap_vit_forward_u8(ap_vit* m, const uint8_t* patches, int n, int h, int w, const float mean[3],
                      const float stdv[3], float* out, void* workspace, size_t workspace_bytes,
                      ap_stream_t stream) {
    int rc = check_forward_args(m, n, patches, out, workspace, workspace_bytes);
    if (rc != AP_OK || n == 0) return rc;
    const int S = m->cfg.image_size;
    AP_REQUIRE(h >= S && w >= S, "vit_forward_u8: %dx%d tiles smaller than the %d model input "
               "(resampling preprocess not in this build)", h, w, S);
    // torchvision CenterCrop: top = int(round((h - S) / 2.0)) (banker's rounding)
    auto crop_off = [](int full, int size) { int d = full - size; return (d / 2) + ((d & 1) && ((d / 2) & 1) ? 1 : 0); };
    hipStream_t s = (hipStream_t)stream;
    auto forward_part = [&](const uint8_t* tiles, int cnt, float* dst, char* base, hipStream_t st) -> int {
        const Workspace ws = carve(m, cnt, base);
        if (m->kpe != 3 * m->cfg.patch_size * m->cfg.patch_size)
            AP_HIP_CHECK(hipMemsetAsync(ws.hid, 0, (size_t)cnt * m->patches * m->kpe * ap::dtype_size(m->cfg.compute_dtype), st));
        int r;
        { ScopedTimer t(m, AP_PROF_PREPROC, st);
          r = ap::preproc_patchrows(tiles, cnt, h, w, crop_off(h, S), crop_off(w, S), S, S, m->cfg.patch_size,
                                    mean, stdv, ws.hid, m->kpe, m->cfg.compute_dtype, st); }
        if (r != AP_OK) return r;
        return run_blocks(m, cnt, ws, dst, st);
    };
    // Experimental (AP_VIT_OPT_TWO_HALF_OVERLAP): the batch as two independent halves on two streams, so that the HBM-bound
    // add+LayerNorm launches of one half can run beside the VALU-bound attention of the other.  Same features (every
    // image is independent of its batch neighbours, tested bit for bit).
    const bool overlap = m->two_half_overlap && n >= 512;
    if (!overlap) return forward_part(patches, n, out, (char*)workspace, s);
    if (!m->side) {
        AP_HIP_CHECK(hipStreamCreateWithFlags(&m->side, hipStreamNonBlocking));
        AP_HIP_CHECK(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
        AP_HIP_CHECK(hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming));
    }
    const int na = (n + 1) / 2, nb = n - na;
    const size_t bytes_a = carve(m, na, nullptr).total;
    const int D = ap_vit_embed_dim(m);
    AP_HIP_CHECK(hipEventRecord(m->ev_fork, s));
    AP_HIP_CHECK(hipStreamWaitEvent(m->side, m->ev_fork, 0));
    if ((rc = forward_part(patches, na, out, (char*)workspace, s)) != AP_OK) return rc;
    if ((rc = forward_part(patches + (size_t)na * h * w * 3, nb, out + (size_t)na * D, (char*)workspace + bytes_a, m->side)) != AP_OK)
        return rc;
    AP_HIP_CHECK(hipEventRecord(m->ev_join, m->side));
    AP_HIP_CHECK(hipStreamWaitEvent(s, m->ev_join, 0));
    return AP_OK;
}

int ap_vit_forward_chw(ap_vit* m, const void* x, int x_dtype, int n, float* out, void* workspace,
                       size_t workspace_bytes, ap_stream_t stream) {
    int rc = check_forward_args(m, n, x, out, workspace, workspace_bytes);
    if (rc != AP_OK || n == 0) return rc;
    AP_REQUIRE(x_dtype == AP_F32 || x_dtype == m->cfg.compute_dtype,
               "vit_forward_chw: input dtype %d must be f32 or the compute dtype", x_dtype);
    const Workspace ws = carve(m, n, (char*)workspace);
    hipStream_t s = (hipStream_t)stream;
    if (m->kpe != 3 * m->cfg.patch_size * m->cfg.patch_size)
        AP_HIP_CHECK(hipMemsetAsync(ws.hid, 0, (size_t)n * m->patches * m->kpe * ap::dtype_size(m->cfg.compute_dtype), s));
    rc = ap::launch_chw_to_patchrows(x_dtype, m->cfg.compute_dtype, x, n, m->cfg.image_size,
                                     m->cfg.patch_size, ws.hid, m->kpe, s);
    if (rc != AP_OK) return rc;
    return run_blocks(m, n, ws, out, s);
}

}  // extern "C"
