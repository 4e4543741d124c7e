// Context, weight repacking and the forward schedule behind the C ABI declared
// in include/slsmi.h.  All device work goes through the launchers in kernels.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "ctx.h"

using namespace slsmi;

// K-split scratch of slsmi_gemm_bf16_static: one array per GPU that runs this library, held until the
// process ends.  In host code the array's NAME is a host-side handle, not a device address: the address
// of each device's copy comes from hipGetSymbolAddress, once per device, and only that pointer is used.
constexpr size_t STATIC_SCRATCH_BYTES = 16u << 20;
static __device__ __align__(256) float scratch[STATIC_SCRATCH_BYTES / sizeof(float)];

static std::mutex g_static_mu;
static float* g_static_ptr[64] = {nullptr};

// Device address of `scratch` on the current device (null on failure).  The first call on a device asks
// the runtime; later calls only read the table.
static float* static_scratch_ptr() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(g_static_mu);
    if (!g_static_ptr[dev]) {
        void* q = nullptr;
        if (hipGetSymbolAddress(&q, HIP_SYMBOL(scratch)) != hipSuccess || !q) return nullptr;
        g_static_ptr[dev] = reinterpret_cast<float*>(q);
    }
    return g_static_ptr[dev];
}

namespace {

constexpr int SLS_FC1_PARTS = 17;  // K slices of the SLS head's fc1 (22848 = 17 x 21 K-tiles at T = 201)

struct Plan {
    int B = 0, L = 0;
    int n = 0;
    int T[SLSMI_MAX_CONV];
    int P[SLSMI_MAX_CONV];
    int frames = 0;
    long M = 0;
    int slab_rows = 0;
    long sls_in = 0, sls_ld = 0;
    size_t convA = 0, convB = 0, feat = 0, x = 0, h = 0, ao = 0, qkv = 0, f = 0, xg = 0;
    size_t conv32 = 0;   // fp32 output of the conv layers that reach their LayerNorm unrounded
    size_t feat3 = 0;    // feature LayerNorm output as [hi | lo | hi] rows of 1536
    size_t xs = 0, pooled = 0, fc1out = 0, rowdots = 0, slsw = 0;
    long x_stride = 0, dots_stride = 0;  // elements between the per-layer buffers of the SLS head
    size_t enc = 0, encb = 0, recon = 0, saepool = 0;
    size_t saehid = 0;   // classifier hidden layer [B][256]
    size_t wkeep = 0;    // window TopK: per-window keep flags
    size_t nframes = 0;  // int32 [B]: frames per utterance of a variable-length batch
    // folded LayerNorms: two sets (attention / feed-forward sub-layer) of per-row slice sums [16][M][2] and of
    // per-row (mean, 1 / deviation) [M][2]
    size_t lnslots = 0, lnstat = 0;
    size_t lnslots_stride = 0, lnstat_stride = 0;  // bytes between the two sets
    size_t total = 0;
};

}  // namespace

namespace {

long numel(const int64_t* shape, int ndim) {
    long n = 1;
    for (int i = 0; i < ndim; ++i) n *= shape[i];
    return n;
}

bool shape_is(const int64_t* shape, int ndim, std::initializer_list<long> want) {
    if ((size_t)ndim != want.size()) return false;
    int i = 0;
    for (long w : want)
        if (shape[i++] != w) return false;
    return true;
}

std::string shape_str(const int64_t* shape, int ndim) {
    std::string s = "[";
    for (int i = 0; i < ndim; ++i) s += (i ? "," : "") + std::to_string((long)shape[i]);
    return s + "]";
}

int copy_f32(slsmi_ctx* c, float** dst, const void* src, long n, float scale, hipStream_t s) {
    int r = dev_alloc(c, dst, (size_t)n);
    if (r) return r;
    if (c->reloading && n <= 8192) {  // the sources outlive a reload: the small ones go in a few launches at its end
        c->pending_copies.push_back(ScaleCopy{reinterpret_cast<const float*>(src), *dst, (int)n, scale});
        return SLSMI_OK;
    }
    HIP_TRY(c, launch_scale_copy_f32(reinterpret_cast<const float*>(src), *dst, n, scale, s));
    return SLSMI_OK;
}

int copy_bf16(slsmi_ctx* c, bf16_t** dst, const void* src, long n, float scale, hipStream_t s) {
    int r = dev_alloc(c, dst, (size_t)n);
    if (r) return r;
    HIP_TRY(c, launch_cvt_f32_bf16(reinterpret_cast<const float*>(src), *dst, n, scale, c->cfg.operand_format, s));
    return SLSMI_OK;
}

bool make_plan(const slsmi_ctx* c, int B, int L, Plan* p) {
    const slsmi_config& cfg = c->cfg;
    if (B <= 0 || L <= 0) return false;
    p->B = B;
    p->L = L;
    p->n = cfg.n_conv;
    if (!conv_frames(cfg, L, p->T)) return false;
    const int n = cfg.n_conv;
    p->frames = p->T[n - 1];
    if (p->frames > 512 || p->frames < 3) return false;
    // Rows stored per utterance after every conv layer: P[i-1] = stride_i * P[i], so
    // that output row m of the whole batch reads input rows stride*m .. stride*m + k - 1
    // and each conv layer is ONE flat GEMM over all utterances.
    int last = (p->frames + 7) / 8 * 8;
    for (;; last += 8) {
        p->P[n - 1] = last;
        bool ok = true;
        for (int i = n - 1; i >= 1; --i) {
            p->P[i - 1] = p->P[i] * cfg.conv_stride[i];
            if (p->P[i - 1] < p->T[i - 1]) ok = false;
        }
        if (ok) break;
    }
    p->M = (long)B * p->frames;
    const long Mpad = (p->M + 127) / 128 * 128 + 128;
    p->slab_rows = (p->frames + 128 + 7) / 8 * 8;
    p->sls_in = (long)(p->frames / 3) * (1024 / 3);
    p->sls_ld = (p->sls_in + 63) / 64 * 64;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += align_up(bytes);
        return o;
    };
    const size_t rowsA = (size_t)B * p->P[0] + 64;
    const size_t rowsB = (n > 1 ? (size_t)B * p->P[1] : 0) + 64;
    p->convA = take(rowsA * 512 * 2);
    p->convB = take(rowsB * 512 * 2);
    p->feat = take((size_t)Mpad * 512 * 2);
    p->feat3 = take((size_t)Mpad * 1536 * 2);
    {
        size_t rows32 = 0;
        for (int i = 1; i < n; ++i)
            if (c->conv_fp32_mask >> i & 1u) rows32 = std::max(rows32, (size_t)B * p->P[i] + 64);
        p->conv32 = take(rows32 * 512 * 4);
    }
    p->x = take((size_t)Mpad * 1024 * 4);
    p->h = take((size_t)Mpad * 1024 * 2);
    p->ao = take((size_t)Mpad * 1024 * 2);
    p->qkv = take((size_t)Mpad * 3072 * 2);
    p->f = take((size_t)Mpad * cfg.ffn_dim * 2);
    p->xg = take((size_t)B * 16 * p->slab_rows * 64 * 2);
    p->nframes = take((size_t)B * 4);
    p->lnslots_stride = align_up((size_t)Mpad * 32 * 4);
    p->lnstat_stride = align_up((size_t)Mpad * 2 * 4);
    p->lnslots = take(2 * p->lnslots_stride);
    p->lnstat = take(2 * p->lnstat_stride);
    if (cfg.head == SLSMI_HEAD_SLS) {
        // every layer output stays in its own buffer until the head has summed them
        p->x_stride = Mpad * 1024;
        p->dots_stride = 16 * Mpad;
        p->xs = take((size_t)cfg.layers * p->x_stride * 4);
        p->rowdots = take((size_t)cfg.layers * p->dots_stride * 4);
        p->slsw = take((size_t)cfg.layers * B * 4);
        p->pooled = take((size_t)(B + 128) * 3 * p->sls_ld * 2);
        p->fc1out = take((size_t)SLS_FC1_PARTS * (B + 128) * 1024 * 4);
    } else if (cfg.head == SLSMI_HEAD_SAE) {
        if (cfg.sae_features != SLSMI_SAE_OFF) {
            p->enc = take((size_t)Mpad * cfg.sae_dict * 4);
            p->encb = take((size_t)Mpad * cfg.sae_dict * 2);
            if (cfg.sae_window > 1) {
                if (p->frames < cfg.sae_window) return false;
                p->wkeep = take((size_t)window_topk_scratch_bytes(B, p->frames, cfg.sae_dict, cfg.sae_window));
            }
        }
        p->recon = take((size_t)Mpad * 1024 * 4);
        p->saepool = take((size_t)B * (cfg.sae_features == SLSMI_SAE_SPARSE ? cfg.sae_dict : 1024) * 4);
        p->saehid = take((size_t)B * 256 * 4);
    }
    p->total = off;
    return true;
}

bool parse_layer(const std::string& name, const char* prefix, int* idx, std::string* rest) {
    const size_t n = strlen(prefix);
    if (name.compare(0, n, prefix) != 0) return false;
    size_t i = n;
    int v = 0;
    bool any = false;
    while (i < name.size() && name[i] >= '0' && name[i] <= '9') {
        v = v * 10 + (name[i] - '0');
        ++i;
        any = true;
    }
    if (!any || i >= name.size() || name[i] != '.') return false;
    *idx = v;
    *rest = name.substr(i + 1);
    return true;
}

}  // namespace

extern "C" {

int slsmi_version(void) { return 100; }

void slsmi_default_config(slsmi_config* cfg, int head) {
    if (!cfg) return;
    memset(cfg, 0, sizeof(*cfg));
    cfg->n_conv = 7;
    const int k[7] = {10, 3, 3, 3, 3, 2, 2};
    const int s[7] = {5, 2, 2, 2, 2, 2, 2};
    for (int i = 0; i < 7; ++i) {
        cfg->conv_dim[i] = 512;
        cfg->conv_kernel[i] = k[i];
        cfg->conv_stride[i] = s[i];
    }
    cfg->embed_dim = 1024;
    cfg->ffn_dim = 4096;
    cfg->heads = 16;
    cfg->layers = 24;
    cfg->conv_pos = 128;
    cfg->conv_pos_groups = 16;
    cfg->ln_eps = 1e-5f;
    cfg->head = head;
    cfg->sae_dict = 4096;
    cfg->sae_k = 128;
    cfg->bn_eps = 1e-5f;
    cfg->operand_format = SLSMI_FMT_BF16;
    cfg->sae_features = SLSMI_SAE_SPARSE;
    cfg->sae_window = 1;
}

int slsmi_create(int device, const slsmi_config* cfg, slsmi_ctx** out) {
    if (!cfg || !out) return SLSMI_ERR_INVALID;
    *out = nullptr;
    if (cfg->n_conv < 2 || cfg->n_conv > SLSMI_MAX_CONV) return SLSMI_ERR_INVALID;
    for (int i = 0; i < cfg->n_conv; ++i)
        if (cfg->conv_dim[i] != 512 || cfg->conv_kernel[i] < 1 || cfg->conv_stride[i] < 1) return SLSMI_ERR_INVALID;
    if (cfg->conv_kernel[0] != 10 || cfg->conv_stride[0] != 5) return SLSMI_ERR_INVALID;
    if (cfg->embed_dim != 1024 || cfg->heads != 16 || cfg->conv_pos != 128 || cfg->conv_pos_groups != 16)
        return SLSMI_ERR_INVALID;
    if (cfg->ffn_dim <= 0 || cfg->ffn_dim % 128 != 0 || cfg->layers < 1 || cfg->layers > 64) return SLSMI_ERR_INVALID;
    if (cfg->operand_format != SLSMI_FMT_BF16 && cfg->operand_format != SLSMI_FMT_F16) return SLSMI_ERR_INVALID;
    if (cfg->head != SLSMI_HEAD_SLS && cfg->head != SLSMI_HEAD_SAE && cfg->head != SLSMI_HEAD_NONE)
        return SLSMI_ERR_INVALID;
    if (cfg->head == SLSMI_HEAD_SAE) {
        if (cfg->sae_features < SLSMI_SAE_SPARSE || cfg->sae_features > SLSMI_SAE_OFF) return SLSMI_ERR_INVALID;
        const int d = cfg->sae_dict;
        if (cfg->sae_features != SLSMI_SAE_OFF &&
            (!(d == 1024 || d == 2048 || d == 4096 || d == 8192) || cfg->sae_k < 1 || cfg->sae_k > d))
            return SLSMI_ERR_INVALID;
        if (cfg->sae_window < 0 || cfg->sae_window > 512) return SLSMI_ERR_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return SLSMI_ERR_HIP;
    slsmi_ctx* c = new (std::nothrow) slsmi_ctx();
    if (!c) return SLSMI_ERR_HIP;
    c->device = device;
    c->cfg = *cfg;
    c->enc.resize(cfg->layers);
    *out = c;
    return SLSMI_OK;
}

int slsmi_load_weight(slsmi_ctx* c, const char* name_c, const void* data, int dtype, const int64_t* shape,
                      int ndim, void* stream) {
    if (!c || !name_c || !data || !shape || ndim < 0 || ndim > 4) return fail(c, SLSMI_ERR_INVALID, "bad argument");
    if (dtype != SLSMI_F32) return fail(c, SLSMI_ERR_INVALID, "only fp32 source tensors are supported");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    std::string name(name_c);
    const slsmi_config& cfg = c->cfg;
    const long n = numel(shape, ndim);
    auto bad_shape = [&]() {
        return fail(c, SLSMI_ERR_INVALID, "unexpected shape " + shape_str(shape, ndim) + " for " + name);
    };
    auto done = [&]() {
        c->loaded[name] = true;
        c->finalized = false;
        if (!c->reloading) {
            train_forget_fresh_copies(c, &name);   // loaded anew: whatever a backward left of it is void
            slsmi_ctx::WeightSource& w = c->sources[name];
            w.data = data;
            w.ndim = ndim;
            for (int i = 0; i < ndim; ++i) w.shape[i] = shape[i];
        }
        return (int)SLSMI_OK;
    };
    int r;
    const std::string bb = "ssl_model.model.";
    if (name.compare(0, bb.size(), bb) == 0) {
        const std::string k = name.substr(bb.size());
        int idx;
        std::string rest;
        if (k == "mask_emb" || k.compare(0, 10, "quantizer.") == 0 || k.compare(0, 10, "project_q.") == 0 ||
            k.compare(0, 11, "final_proj.") == 0)
            return SLSMI_OK;  // pre-training leftovers the forward never reads
        if (parse_layer(k, "feature_extractor.conv_layers.", &idx, &rest)) {
            if (idx < 0 || idx >= cfg.n_conv) return fail(c, SLSMI_ERR_NAME, "conv layer out of range: " + name);
            ConvLayer& L = c->conv[idx];
            const int kk = cfg.conv_kernel[idx];
            if (rest == "0.weight") {
                if (idx == 0) {
                    if (!shape_is(shape, ndim, {512, 1, kk})) return bad_shape();
                    if ((r = copy_f32(c, &L.w0, data, n, 1.f, s))) return r;
                } else {
                    if (!shape_is(shape, ndim, {512, 512, kk})) return bad_shape();
                    if ((r = dev_alloc(c, &L.w, (size_t)n)) || (r = dev_alloc(c, &L.w2, (size_t)2 * n))) return r;
                    HIP_TRY(c, launch_repack_conv_w(reinterpret_cast<const float*>(data), L.w, 512, 512, kk, cfg.operand_format, s));
                    HIP_TRY(c, launch_repack_conv_w_split(reinterpret_cast<const float*>(data), L.w2, 512, 512, kk, cfg.operand_format, s));
                }
                return done();
            }
            float** dst = rest == "0.bias" ? &L.bias : rest == "2.1.weight" ? &L.ln_g : rest == "2.1.bias" ? &L.ln_b : nullptr;
            if (!dst) return fail(c, SLSMI_ERR_NAME, "unknown weight " + name);
            if (!shape_is(shape, ndim, {512})) return bad_shape();
            if ((r = copy_f32(c, dst, data, n, 1.f, s))) return r;
            return done();
        }
        if (parse_layer(k, "encoder.layers.", &idx, &rest)) {
            if (idx < 0 || idx >= cfg.layers) return fail(c, SLSMI_ERR_NAME, "encoder layer out of range: " + name);
            EncLayer& E = c->enc[idx];
            const long C = 1024, Fd = cfg.ffn_dim;
            static const char* proj[3] = {"q_proj", "k_proj", "v_proj"};
            for (int q = 0; q < 3; ++q) {
                const float scale = q == 0 ? 0.125f : 1.0f;  // head_dim ** -0.5, folded into q
                if (rest == std::string("self_attn.") + proj[q] + ".weight") {
                    if (!shape_is(shape, ndim, {C, C})) return bad_shape();
                    if ((r = dev_alloc(c, &E.wqkv, (size_t)3 * C * C))) return r;
                    long ldo = 0;
                    if (bf16_t* wT = train_layer_wT(c, idx, q, &ldo)) {
                        if (c->reloading && train_copy_is_fresh(c, name)) {
                            // (the backward that stepped this tensor wrote both copies)
                        } else if (c->reloading)
                            c->pending_w.push_back(CvtTransposeW{reinterpret_cast<const float*>(data), E.wqkv + q * C * C, wT,
                                                                 ldo, (int)C, (int)C, scale});
                        else
                            HIP_TRY(c, launch_cvt_transpose_w(reinterpret_cast<const float*>(data), C, (int)C, scale,
                                                              cfg.operand_format, E.wqkv + q * C * C, wT, ldo, s));
                    } else
                        HIP_TRY(c, launch_cvt_f32_bf16(reinterpret_cast<const float*>(data), E.wqkv + q * C * C, n, scale, cfg.operand_format, s));
                    return done();
                }
                if (rest == std::string("self_attn.") + proj[q] + ".bias") {
                    if (!shape_is(shape, ndim, {C})) return bad_shape();
                    if ((r = dev_alloc(c, &E.bqkv, (size_t)3 * C))) return r;
                    if (c->reloading)
                        c->pending_copies.push_back(ScaleCopy{reinterpret_cast<const float*>(data), E.bqkv + q * C, (int)n, scale});
                    else
                        HIP_TRY(c, launch_scale_copy_f32(reinterpret_cast<const float*>(data), E.bqkv + q * C, n, scale, s));
                    return done();
                }
            }
            struct W { const char* key; bf16_t** dst; long a, b; int which; } ws[] = {
                {"self_attn.out_proj.weight", &E.wo, C, C, 3}, {"fc1.weight", &E.w1, Fd, C, 4}, {"fc2.weight", &E.w2, C, Fd, 5}};
            for (auto& w : ws)
                if (rest == w.key) {
                    if (!shape_is(shape, ndim, {w.a, w.b})) return bad_shape();
                    long ldo = 0;
                    if (bf16_t* wT = train_layer_wT(c, idx, w.which, &ldo)) {
                        if ((r = dev_alloc(c, w.dst, (size_t)n))) return r;
                        if (c->reloading && train_copy_is_fresh(c, name)) {
                            // (the backward that stepped this tensor wrote both copies)
                        } else if (c->reloading)
                            c->pending_w.push_back(CvtTransposeW{reinterpret_cast<const float*>(data), *w.dst, wT, ldo,
                                                                 (int)w.a, (int)w.b, 1.f});
                        else
                            HIP_TRY(c, launch_cvt_transpose_w(reinterpret_cast<const float*>(data), w.a, (int)w.b, 1.f,
                                                              cfg.operand_format, *w.dst, wT, ldo, s));
                    } else if ((r = copy_bf16(c, w.dst, data, n, 1.f, s))) {
                        return r;
                    }
                    return done();
                }
            struct V { const char* key; float** dst; long a; } vs[] = {
                {"self_attn.out_proj.bias", &E.bo, C}, {"fc1.bias", &E.b1, Fd}, {"fc2.bias", &E.b2, C},
                {"self_attn_layer_norm.weight", &E.ln1_g, C}, {"self_attn_layer_norm.bias", &E.ln1_b, C},
                {"final_layer_norm.weight", &E.ln2_g, C}, {"final_layer_norm.bias", &E.ln2_b, C}};
            for (auto& v : vs)
                if (rest == v.key) {
                    if (!shape_is(shape, ndim, {v.a})) return bad_shape();
                    if ((r = copy_f32(c, v.dst, data, n, 1.f, s))) return r;
                    return done();
                }
            return fail(c, SLSMI_ERR_NAME, "unknown weight " + name);
        }
        if (k == "post_extract_proj.weight") {
            if (!shape_is(shape, ndim, {1024, 512})) return bad_shape();
            if ((r = copy_bf16(c, &c->proj_w, data, n, 1.f, s))) return r;
            if ((r = dev_alloc(c, &c->proj_w3, (size_t)3 * n))) return r;
            HIP_TRY(c, launch_cvt_split3_rows(reinterpret_cast<const float*>(data), c->proj_w3, 1024, 512, 512, cfg.operand_format, s));
            return done();
        }
        if (k == "encoder.pos_conv.0.weight_v") {
            if (!shape_is(shape, ndim, {1024, 64, 128})) return bad_shape();
            if ((r = copy_f32(c, &c->pos_v, data, n, 1.f, s))) return r;
            return done();
        }
        if (k == "encoder.pos_conv.0.weight_g") {
            if (!shape_is(shape, ndim, {1, 1, 128})) return bad_shape();
            if ((r = copy_f32(c, &c->pos_g, data, n, 1.f, s))) return r;
            return done();
        }
        struct V { const char* key; float** dst; long a; } vs[] = {
            {"post_extract_proj.bias", &c->proj_b, 1024}, {"layer_norm.weight", &c->feat_ln_g, 512},
            {"layer_norm.bias", &c->feat_ln_b, 512}, {"encoder.pos_conv.0.bias", &c->pos_b, 1024},
            {"encoder.layer_norm.weight", &c->final_ln_g, 1024}, {"encoder.layer_norm.bias", &c->final_ln_b, 1024}};
        for (auto& v : vs)
            if (k == v.key) {
                if (!shape_is(shape, ndim, {v.a})) return bad_shape();
                if ((r = copy_f32(c, v.dst, data, n, 1.f, s))) return r;
                return done();
            }
        return fail(c, SLSMI_ERR_NAME, "unknown weight " + name);
    }
    // ---- heads
    if (name == "first_bn.num_batches_tracked" || name == "sae.k") return SLSMI_OK;
    if (cfg.head == SLSMI_HEAD_SLS) {
        static const char* bnk[4] = {"first_bn.weight", "first_bn.bias", "first_bn.running_mean", "first_bn.running_var"};
        for (int i = 0; i < 4; ++i)
            if (name == bnk[i]) {
                if (n != 1) return bad_shape();
                if ((r = dev_alloc(c, &c->bn, 4))) return r;
                HIP_TRY(c, launch_scale_copy_f32(reinterpret_cast<const float*>(data), c->bn + i, 1, 1.f, s));
                return done();
            }
        if (name == "fc1.weight") {
            if (ndim != 2 || shape[0] != 1024) return bad_shape();
            if (c->sls_fc1_w && c->sls_in != shape[1])  // the buffer is sized by the first load
                return fail(c, SLSMI_ERR_INVALID, "fc1.weight loaded again with other in-features (needs a new context)");
            c->sls_in = shape[1];
            c->sls_ld = (c->sls_in + 63) / 64 * 64;
            if ((r = dev_alloc(c, &c->sls_fc1_w, (size_t)1024 * 3 * c->sls_ld))) return r;
            HIP_TRY(c, launch_cvt_split3_rows(reinterpret_cast<const float*>(data), c->sls_fc1_w, 1024, c->sls_in, c->sls_ld, cfg.operand_format, s));
            return done();
        }
        struct V { const char* key; float** dst; long cnt; } vs[] = {
            {"fc0.weight", &c->fc0_w, 1024}, {"fc0.bias", &c->fc0_b, 1}, {"fc1.bias", &c->sls_fc1_b, 1024},
            {"fc3.weight", &c->fc3_w, 2048}, {"fc3.bias", &c->fc3_b, 2}};
        for (auto& v : vs)
            if (name == v.key) {
                if (n != v.cnt) return bad_shape();
                if ((r = copy_f32(c, v.dst, data, n, 1.f, s))) return r;
                return done();
            }
    } else if (cfg.head == SLSMI_HEAD_SAE) {
        const long D = cfg.sae_dict;
        const long Dc = cfg.sae_features == SLSMI_SAE_SPARSE ? D : 1024;  // classifier input (model.py:177-179)
        const bool has_sae = cfg.sae_features != SLSMI_SAE_OFF;
        if (!has_sae && name.compare(0, 4, "sae.") == 0) return fail(c, SLSMI_ERR_NAME, "unknown weight " + name);
        if (name == "sae.encoder.weight") {
            if (!shape_is(shape, ndim, {D, 1024})) return bad_shape();
            if ((r = copy_bf16(c, &c->sae_enc_w, data, n, 1.f, s))) return r;
            return done();
        }
        if (name == "sae.decoder.weight") {
            if (!shape_is(shape, ndim, {1024, D})) return bad_shape();
            if ((r = copy_bf16(c, &c->sae_dec_w, data, n, 1.f, s))) return r;
            return done();
        }
        struct V { const char* key; float** dst; long cnt; } vs[] = {
            {"sae.encoder.bias", &c->sae_enc_b, D}, {"sae.b_dec", &c->sae_b_dec, 1024},
            {"classifier.0.weight", &c->cls_ln_g, Dc}, {"classifier.0.bias", &c->cls_ln_b, Dc},
            {"classifier.1.weight", &c->cls_w1, 256 * Dc}, {"classifier.1.bias", &c->cls_b1, 256},
            {"classifier.4.weight", &c->cls_w2, 512}, {"classifier.4.bias", &c->cls_b2, 2}};
        for (auto& v : vs)
            if (name == v.key) {
                if (n != v.cnt) return bad_shape();
                if ((r = copy_f32(c, v.dst, data, n, 1.f, s))) return r;
                return done();
            }
    }
    return fail(c, SLSMI_ERR_NAME, "unknown weight " + name);
}

static int finalize_impl(slsmi_ctx* c, void* stream, bool synchronise);

int slsmi_reload_weights(slsmi_ctx* c, void* stream) {
    if (!c) return SLSMI_ERR_INVALID;
    if (c->sources.empty()) return fail(c, SLSMI_ERR_MISSING, "no weights have been loaded");
    c->reloading = true;
    c->pending_copies.clear();
    c->pending_w.clear();
    int rc = SLSMI_OK;
    for (const auto& kv : c->sources) {
        rc = slsmi_load_weight(c, kv.first.c_str(), kv.second.data, SLSMI_F32, kv.second.shape, kv.second.ndim, stream);
        if (rc != SLSMI_OK) break;
    }
    c->reloading = false;
    if (rc == SLSMI_OK) {
        const hipError_t e = launch_scale_copy_batch(c->pending_copies.data(), (int)c->pending_copies.size(),
                                                     reinterpret_cast<hipStream_t>(stream));
        if (e != hipSuccess) rc = hip_fail(c, e, "launch_scale_copy_batch");
    }
    if (rc == SLSMI_OK) {
        const hipError_t e = launch_cvt_transpose_w_batch(c->pending_w.data(), (int)c->pending_w.size(), c->cfg.operand_format,
                                                          reinterpret_cast<hipStream_t>(stream));
        if (e != hipSuccess) rc = hip_fail(c, e, "launch_cvt_transpose_w_batch");
    }
    c->pending_copies.clear();
    c->pending_w.clear();
    train_forget_fresh_copies(c);
    // the sources are the caller's live parameters: nothing here has to wait for the stream
    return rc == SLSMI_OK ? finalize_impl(c, stream, false) : rc;
}

int slsmi_finalize(slsmi_ctx* c, void* stream) { return finalize_impl(c, stream, true); }

// synchronise: wait for the stream before returning, so that the caller may release the buffers it passed to
// slsmi_load_weight (set-up); a reload leaves everything enqueued.
static int finalize_impl(slsmi_ctx* c, void* stream, bool synchronise) {
    if (!c) return SLSMI_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const slsmi_config& cfg = c->cfg;
    (void)static_scratch_ptr();  // set-up time: the first slsmi_gemm_bf16_static on this device then only reads the table
    {   // scratch for the K split of GEMMs with few rows (small batches, the training step)
        int rc = dev_alloc(c, &c->split_ws, GEMM_SPLIT_WS_BYTES / 4);
        if (rc != SLSMI_OK) return rc;
    }
    std::vector<std::string> need;
    const std::string bb = "ssl_model.model.";
    for (int i = 0; i < cfg.n_conv; ++i) {
        const std::string p = bb + "feature_extractor.conv_layers." + std::to_string(i) + ".";
        for (const char* k : {"0.weight", "0.bias", "2.1.weight", "2.1.bias"}) need.push_back(p + k);
    }
    for (const char* k : {"layer_norm.weight", "layer_norm.bias", "post_extract_proj.weight", "post_extract_proj.bias",
                          "encoder.pos_conv.0.weight_g", "encoder.pos_conv.0.weight_v", "encoder.pos_conv.0.bias",
                          "encoder.layer_norm.weight", "encoder.layer_norm.bias"})
        need.push_back(bb + k);
    for (int l = 0; l < cfg.layers; ++l) {
        const std::string p = bb + "encoder.layers." + std::to_string(l) + ".";
        for (const char* k : {"self_attn.q_proj.weight", "self_attn.q_proj.bias", "self_attn.k_proj.weight",
                              "self_attn.k_proj.bias", "self_attn.v_proj.weight", "self_attn.v_proj.bias",
                              "self_attn.out_proj.weight", "self_attn.out_proj.bias", "self_attn_layer_norm.weight",
                              "self_attn_layer_norm.bias", "fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias",
                              "final_layer_norm.weight", "final_layer_norm.bias"})
            need.push_back(p + k);
    }
    if (cfg.head == SLSMI_HEAD_SLS) {
        for (const char* k : {"first_bn.weight", "first_bn.bias", "first_bn.running_mean", "first_bn.running_var",
                              "fc0.weight", "fc0.bias", "fc1.weight", "fc1.bias", "fc3.weight", "fc3.bias"})
            need.push_back(k);
    } else if (cfg.head == SLSMI_HEAD_SAE) {
        if (cfg.sae_features != SLSMI_SAE_OFF)
            for (const char* k : {"sae.encoder.weight", "sae.encoder.bias", "sae.b_dec", "sae.decoder.weight"})
                need.push_back(k);
        for (const char* k : {"classifier.0.weight", "classifier.0.bias", "classifier.1.weight", "classifier.1.bias",
                              "classifier.4.weight", "classifier.4.bias"})
            need.push_back(k);
    }
    for (const auto& k : need)
        if (!c->loaded.count(k)) return fail(c, SLSMI_ERR_MISSING, "missing weight " + k);

    int r;
    if ((r = dev_alloc(c, &c->pos_w, (size_t)1024 * 64 * 128))) return r;
    if ((r = dev_alloc(c, &c->pos_norm, 128 + 256 * 128))) return r;   // the norms and the scratch of their sums
    HIP_TRY(c, launch_repack_posconv(c->pos_v, c->pos_g, c->pos_w, c->pos_norm, cfg.operand_format, s));
    // (the SLS head's BatchNorm affine is computed on the device from c->bn by the kernel that applies it)
    for (int l = 0; l < cfg.layers; ++l) {  // what the GEMMs that apply a folded LayerNorm need beside their weights
        EncLayer& E = c->enc[l];
        if ((r = dev_alloc(c, &E.qkv_g, 3072)) || (r = dev_alloc(c, &E.qkv_c, 3072)) ||
            (r = dev_alloc(c, &E.fc1_g, (size_t)cfg.ffn_dim)) || (r = dev_alloc(c, &E.fc1_c, (size_t)cfg.ffn_dim)))
            return r;
    }
    c->fold_consts_ready = false;  // made by the first scoring forward that folds (a training loop reloads every step and never scores)
    if (c->train && (r = train_prepare(c, s))) return r;
    if (synchronise) HIP_TRY(c, hipStreamSynchronize(s));
    c->finalized = true;
    return SLSMI_OK;
}

int slsmi_num_frames(const slsmi_ctx* c, int length) {
    if (!c) return SLSMI_ERR_INVALID;
    int T[SLSMI_MAX_CONV];
    if (!conv_frames(c->cfg, length, T)) return SLSMI_ERR_INVALID;
    return T[c->cfg.n_conv - 1];
}

size_t slsmi_workspace_bytes(const slsmi_ctx* c, int batch, int length) {
    if (!c) return 0;
    Plan p;
    if (!make_plan(c, batch, length, &p)) return 0;
    return p.total;
}

}  // extern "C"

namespace {

// The forward schedule.  lengths: null, or device int32 [B] sample counts of a variable-length batch.
int forward_impl(slsmi_ctx* c, const float* wav, int B, int L, const int32_t* lengths, void* workspace,
                 size_t workspace_bytes, float* logp_out, float* x_out, float* layers_out, float* sae_loss_out,
                 void* stream) {
    if (!c) return SLSMI_ERR_INVALID;
    if (!wav || !workspace || (!logp_out && c->cfg.head != SLSMI_HEAD_NONE))
        return fail(c, SLSMI_ERR_INVALID, "null argument");
    if (!c->finalized) return fail(c, SLSMI_ERR_MISSING, "slsmi_finalize has not been called after the last weight load");
    const slsmi_config& cfg = c->cfg;
    Plan p;
    if (!make_plan(c, B, L, &p))
        return fail(c, SLSMI_ERR_INVALID, "unsupported batch/length (frames must be in [3, 512] and at least sae_window)");
    if (workspace_bytes < p.total || (reinterpret_cast<uintptr_t>(workspace) & (ALIGN - 1)))
        return fail(c, SLSMI_ERR_WORKSPACE, "workspace too small or not 256-byte aligned");
    if (cfg.head == SLSMI_HEAD_SLS && p.sls_in != c->sls_in)
        return fail(c, SLSMI_ERR_INVALID, "fc1.weight in-features " + std::to_string(c->sls_in) +
                                              " do not match this input length (" + std::to_string(p.sls_in) + ")");
    if (sae_loss_out && cfg.head != SLSMI_HEAD_SAE) return fail(c, SLSMI_ERR_INVALID, "sae_loss_out needs the SAE head");
    if (lengths && cfg.head == SLSMI_HEAD_SLS)
        return fail(c, SLSMI_ERR_INVALID, "the SLS head's fc1 fixes the frame count: no variable-length batches");
    if (lengths && cfg.head == SLSMI_HEAD_SAE && cfg.sae_window > 1)
        return fail(c, SLSMI_ERR_INVALID, "window TopK is defined on fixed-length batches only");
    if (lengths && sae_loss_out)
        return fail(c, SLSMI_ERR_INVALID, "the SAE loss is not defined for a variable-length batch");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    char* ws = reinterpret_cast<char*>(workspace);
    auto B16 = [&](size_t off) { return reinterpret_cast<bf16_t*>(ws + off); };
    auto F32 = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    const int n = cfg.n_conv;
    const float eps = cfg.ln_eps;
    const int fmt = cfg.operand_format;
    const int T = p.frames;
    const long M = p.M;
    // Range report (slsmi_set_range_report): after each 16-bit store, its largest |value| and its count of saturated
    // values go to slot `at` of the caller's array (SLSMI_RANGE_* in slsmi.h)
    if (c->range_out) HIP_TRY(c, hipMemsetAsync(c->range_out, 0, (size_t)c->range_capacity * 8, s));
    auto RANGE = [&](int at, const bf16_t* buf, long rows, int cols, long ld, int grp_in = 0, int grp_valid = 0) -> hipError_t {
        if (!c->range_out || at < 0 || at >= c->range_capacity) return hipSuccess;
        return launch_range_stat(buf, rows, cols, ld, fmt, c->range_out + 2 * at, s, grp_in, grp_valid);
    };
    // Variable-length batch (wav2vec2.py:567-586): utterance b owns its first nframes[b] frames.  The
    // conv stack and every row-wise stage run on all T rows (a valid frame reads only samples of its own
    // utterance's real part); the padded frames are zeroed where the positional conv reads them, masked
    // as attention keys and left out of the pooled mean.
    int* nframes = nullptr;
    if (lengths) {
        nframes = reinterpret_cast<int*>(ws + p.nframes);
        ConvGeometry geo;
        geo.n = n;
        for (int i = 0; i < n; ++i) {
            geo.kernel[i] = cfg.conv_kernel[i];
            geo.stride[i] = cfg.conv_stride[i];
        }
        HIP_TRY(c, launch_frames_from_lengths(lengths, nframes, B, geo, T, s));
    }

    // ---- feature extractor (wav2vec2.py:843-851)
    HIP_TRY(c, launch_conv0_ln_gelu(wav, B, L, p.T[0], c->conv[0].w0, c->conv[0].bias, c->conv[0].ln_g,
                                    c->conv[0].ln_b, eps, B16(p.convA), p.P[0], fmt, s));
    HIP_TRY(c, RANGE(SLSMI_RANGE_CONV + 0, B16(p.convA), (long)B * p.P[0], 512, 512, p.P[0], p.T[0]));
    bf16_t* cur = B16(p.convA);
    bf16_t* nxt = B16(p.convB);
    for (int i = 1; i < n; ++i) {
        const bool last = i == n - 1;
        bf16_t* dst = last ? B16(p.feat) : nxt;
        // split: multiply by the weights' two 16-bit halves (K doubles, A repeats); wide: hand the LayerNorm fp32
        // values, as the reference's Fp32LayerNorm gets them (wav2vec2.py:807-812)
        const bool split = (c->conv_split_mask >> i & 1u) != 0, wide = (c->conv_fp32_mask >> i & 1u) != 0;
        const int Kc = cfg.conv_kernel[i] * 512;
        GemmArgs g;
        g.fmt = fmt;
        g.split_ws = c->split_scoring ? c->split_ws : nullptr; g.split_ws_bytes = GEMM_SPLIT_WS_BYTES;
        g.A = cur;
        g.lda = (long)cfg.conv_stride[i] * 512;
        g.W = split ? c->conv[i].w2 : c->conv[i].w;
        g.ldw = split ? 2L * Kc : Kc;
        g.bias = c->conv[i].bias;
        g.C = wide ? static_cast<void*>(F32(p.conv32)) : static_cast<void*>(dst);
        g.ldc = 512;
        g.M = B * p.P[i];
        g.N = 512;
        g.K = split ? 2 * Kc : Kc;
        g.ka = split ? Kc : 0;
        g.mode = wide ? GEMM_F32 : GEMM_BF16;
        if (last) {
            g.grp_in = p.P[i];
            g.grp_valid = T;
            g.grp_out = T;
            // (the 256 x 128 kernel's fp32 epilogue is slow on this small, row-compacting launch: 81 us against
            // 46 us for 128-row tiles at 64 clips)
            if (wide) g.bn = g.N % 128 == 0 ? 128 : 64;
        }
        HIP_TRY(c, launch_gemm(g, s));
        const long rows = last ? M : (long)B * p.P[i];
        if (wide)
            HIP_TRY(c, launch_ln512_f32(F32(p.conv32), dst, rows, c->conv[i].ln_g, c->conv[i].ln_b, eps, 1, fmt, s));
        else
            HIP_TRY(c, launch_ln512_bf16(dst, dst, rows, c->conv[i].ln_g, c->conv[i].ln_b, eps, 1, fmt, s));
        HIP_TRY(c, RANGE(SLSMI_RANGE_CONV + i, dst, rows, 512, 512, last ? 0 : p.P[i], last ? 0 : p.T[i]));
        if (!last) {
            nxt = cur;
            cur = dst;
        }
    }
    // ---- LayerNorm(512) + post_extract_proj (wav2vec2.py:563-564, :595-596): the normalised features as
    // [hi | lo | hi] rows against [W_hi | W_hi | W_lo] -- three small passes instead of one rounding of each operand
    HIP_TRY(c, launch_ln512_split3(B16(p.feat), B16(p.feat3), M, c->feat_ln_g, c->feat_ln_b, eps, fmt, s));
    HIP_TRY(c, RANGE(SLSMI_RANGE_FEATURES, B16(p.feat3), M, 512, 1536));
    {
        GemmArgs g;
        g.fmt = fmt;
        g.split_ws = c->split_scoring ? c->split_ws : nullptr; g.split_ws_bytes = GEMM_SPLIT_WS_BYTES;
        g.A = B16(p.feat3); g.lda = 1536; g.W = c->proj_w3; g.ldw = 1536; g.bias = c->proj_b;
        g.C = F32(p.x); g.ldc = 1024; g.M = (int)M; g.N = 1024; g.K = 1536; g.k_algo = 512; g.mode = GEMM_F32;
        HIP_TRY(c, launch_gemm(g, s));
    }
    // ---- x += GELU(SamePad(pos_conv(x))) (wav2vec2.py:915-917)
    HIP_TRY(c, launch_posconv_pack(F32(p.x), B16(p.xg), B, T, p.slab_rows, fmt, s, nframes));
    if (c->posconv_slab && !c->split_scoring && posconv_slab_supported(T, p.slab_rows)) {
        // a (clip, group)'s slab stays in LDS for all 128 taps; same bits as the batched GEMM below
        HIP_TRY(c, launch_posconv_slab(B16(p.xg), c->pos_w, c->pos_b, F32(p.x), B, T, p.slab_rows, fmt, s));
    } else {
        GemmArgs g;
        g.fmt = fmt;
        g.split_ws = c->split_scoring ? c->split_ws : nullptr; g.split_ws_bytes = GEMM_SPLIT_WS_BYTES;
        g.A = B16(p.xg); g.lda = 64; g.W = c->pos_w; g.ldw = 8192; g.bias = c->pos_b;
        g.C = F32(p.x); g.ldc = 1024; g.M = T; g.N = 64; g.K = 8192; g.mode = GEMM_F32_RESID_GELU; g.bn = 64;
        g.nz = B * 16; g.zA = (long)p.slab_rows * 64; g.zWmod = 16; g.zW = 64L * 8192; g.zBias = 64;
        g.zCdiv = 16; g.zC1 = (long)T * 1024; g.zC2 = 64;
        HIP_TRY(c, launch_gemm(g, s));
    }
    // ---- encoder layers (wav2vec2.py:1042-1062)
    // SLS head: layer l's output x_l is weighted by sigmoid(fc0 . mean_t x_l), a weight known only
    // once the whole layer is.  Every layer therefore writes its output to a buffer of its own (the
    // attention residual GEMM reads x_{l-1} and writes buffer l, the FFN one updates it in place and
    // leaves the per-row dots fc0 . x_l[row] beside it); after the last layer one kernel turns all the
    // dots into the 24 B weights and one pass sums and pools the 24 buffers.
    const bool sls = cfg.head == SLSMI_HEAD_SLS;
    // The two LayerNorms of a layer are folded into the GEMMs around them (GemmArgs::ln_mode): the residual GEMM in
    // front of a LayerNorm also writes the next GEMM's 16-bit operand and the row sums its statistics follow from,
    // the GEMM behind it finishes the normalisation in its epilogue.  Set 0 of the sums / statistics belongs to the
    // attention sub-layer's LayerNorm, set 1 to the feed-forward one.  That is the form for batches the 208 x 256
    // kernel serves best (`fold`).  Other batches take the row-wise form (`fold_rows`): one kernel per LayerNorm
    // writes the same 16-bit operand and the rows' factors, the consumer GEMM applies them on whichever kernel its
    // shape selects -- the same arithmetic, so a score does not depend on which form its batch took.  Low-latency
    // mode (K-split kernels, which do not carry the fold) keeps separate LayerNorm kernels.
    const bool fold_any = c->ln_fold && !c->split_scoring;
    const bool fold = fold_any && gemm_ln_fold_fits((int)M, cfg.ffn_dim, c->ln_fold_force_wide);
    const bool fold_rows = fold_any && !fold;
    float* const aff = F32(p.lnslots);  // row-wise form: (alpha, -ubar) per row
    float* const slots0 = F32(p.lnslots);
    float* const slots1 = reinterpret_cast<float*>(ws + p.lnslots + p.lnslots_stride);
    float* const stat0 = F32(p.lnstat);
    float* const stat1 = reinterpret_cast<float*>(ws + p.lnstat + p.lnstat_stride);
    float* xin = F32(p.x);  // x_{l-1}
    if (fold_any && !c->fold_consts_ready) {
        for (int l = 0; l < cfg.layers; ++l) {
            const EncLayer& E = c->enc[l];
            HIP_TRY(c, launch_ln_fold_consts(E.wqkv, E.bqkv, E.ln1_g, E.ln1_b, E.qkv_g, E.qkv_c, 3072, fmt, s));
            HIP_TRY(c, launch_ln_fold_consts(E.w1, E.b1, E.ln2_g, E.ln2_b, E.fc1_g, E.fc1_c, cfg.ffn_dim, fmt, s));
        }
        c->fold_consts_ready = true;
    }
    if (fold && cfg.layers > 0)
        HIP_TRY(c, launch_ln1024_fold_prep(xin, B16(p.h), stat0, slots0, M, c->enc[0].ln1_g, eps, fmt, s));
    for (int l = 0; l < cfg.layers; ++l) {
        const EncLayer& E = c->enc[l];
        float* xl = sls ? F32(p.xs) + (size_t)l * p.x_stride : F32(p.x);
        if (fold_rows)
            HIP_TRY(c, launch_ln1024_fold_rows(xin, l == 0 ? nullptr : stat0, B16(p.h), aff, stat1, M, E.ln1_g, eps, fmt, s));
        else if (!fold)
            HIP_TRY(c, launch_ln1024_f32_bf16(xin, B16(p.h), M, E.ln1_g, E.ln1_b, eps, fmt, s));
        GemmArgs g;
        g.fmt = fmt;
        g.split_ws = c->split_scoring ? c->split_ws : nullptr; g.split_ws_bytes = GEMM_SPLIT_WS_BYTES;
        g.A = B16(p.h); g.lda = 1024; g.W = E.wqkv; g.ldw = 1024; g.bias = E.bqkv;
        g.C = B16(p.qkv); g.ldc = 3072; g.M = (int)M; g.N = 3072; g.K = 1024; g.mode = GEMM_BF16;
        if (fold) {  // statistics of x_{l-1} -> stat1
            g.ln_mode = GemmArgs::LN_APPLY; g.bias = E.qkv_c; g.ln_vec = E.qkv_g; g.ln_proxy = stat0; g.ln_slots = slots0;
            g.ln_stat = stat1; g.ln_eps = eps;
        } else if (fold_rows) {
            g.ln_mode = GemmArgs::LN_APPLY; g.bias = E.qkv_c; g.ln_vec = E.qkv_g; g.ln_aff = aff;
        }
        HIP_TRY(c, RANGE(SLSMI_RANGE_LAYER0 + SLSMI_RANGE_PER_LAYER * l + 0, B16(p.h), M, 1024, 1024));   // LayerNorm 1 operand
        HIP_TRY(c, launch_gemm(g, s));
        HIP_TRY(c, RANGE(SLSMI_RANGE_LAYER0 + SLSMI_RANGE_PER_LAYER * l + 1, B16(p.qkv), M, 3072, 3072));
        HIP_TRY(c, launch_attention(B16(p.qkv), B16(p.ao), B, T, fmt, s, nframes));
        HIP_TRY(c, RANGE(SLSMI_RANGE_LAYER0 + SLSMI_RANGE_PER_LAYER * l + 2, B16(p.ao), M, 1024, 1024));
        g = GemmArgs();
        g.fmt = fmt;
        g.split_ws = c->split_scoring ? c->split_ws : nullptr; g.split_ws_bytes = GEMM_SPLIT_WS_BYTES;
        g.A = B16(p.ao); g.lda = 1024; g.W = E.wo; g.ldw = 1024; g.bias = E.bo;
        g.C = xl; g.resid = xin; g.ldc = 1024; g.M = (int)M; g.N = 1024; g.K = 1024; g.mode = GEMM_F32_RESID;
        if (fold) {
            g.ln_mode = GemmArgs::LN_EMIT; g.ln_proxy = stat1; g.ln_vec = E.ln2_g; g.ln_a = B16(p.h); g.ln_slots = slots1;
        }
        HIP_TRY(c, launch_gemm(g, s));
        if (fold_rows)
            HIP_TRY(c, launch_ln1024_fold_rows(xl, stat1, B16(p.h), aff, stat0, M, E.ln2_g, eps, fmt, s));
        else if (!fold)
            HIP_TRY(c, launch_ln1024_f32_bf16(xl, B16(p.h), M, E.ln2_g, E.ln2_b, eps, fmt, s));
        g = GemmArgs();
        g.fmt = fmt;
        g.split_ws = c->split_scoring ? c->split_ws : nullptr; g.split_ws_bytes = GEMM_SPLIT_WS_BYTES;
        g.A = B16(p.h); g.lda = 1024; g.W = E.w1; g.ldw = 1024; g.bias = E.b1;
        g.C = B16(p.f); g.ldc = cfg.ffn_dim; g.M = (int)M; g.N = cfg.ffn_dim; g.K = 1024; g.mode = GEMM_BF16_GELU;
        if (fold) {  // statistics of the attention sub-layer's output -> stat0
            g.ln_mode = GemmArgs::LN_APPLY; g.bias = E.fc1_c; g.ln_vec = E.fc1_g; g.ln_proxy = stat1; g.ln_slots = slots1;
            g.ln_stat = stat0; g.ln_eps = eps;
        } else if (fold_rows) {
            g.ln_mode = GemmArgs::LN_APPLY; g.bias = E.fc1_c; g.ln_vec = E.fc1_g; g.ln_aff = aff;
        }
        HIP_TRY(c, RANGE(SLSMI_RANGE_LAYER0 + SLSMI_RANGE_PER_LAYER * l + 3, B16(p.h), M, 1024, 1024));   // LayerNorm 2 operand
        HIP_TRY(c, launch_gemm(g, s));
        HIP_TRY(c, RANGE(SLSMI_RANGE_LAYER0 + SLSMI_RANGE_PER_LAYER * l + 4, B16(p.f), M, cfg.ffn_dim, cfg.ffn_dim));
        g = GemmArgs();
        g.fmt = fmt;
        g.split_ws = c->split_scoring ? c->split_ws : nullptr; g.split_ws_bytes = GEMM_SPLIT_WS_BYTES;
        g.A = B16(p.f); g.lda = cfg.ffn_dim; g.W = E.w2; g.ldw = cfg.ffn_dim; g.bias = E.b2;
        g.C = xl; g.ldc = 1024; g.M = (int)M; g.N = 1024; g.K = cfg.ffn_dim; g.mode = GEMM_F32_RESID;
        if (sls) {
            g.dot_w = c->fc0_w;
            g.dot_out = F32(p.rowdots) + (size_t)l * p.dots_stride;
        }
        if (fold && l + 1 < cfg.layers) {  // the next layer's first LayerNorm
            g.ln_mode = GemmArgs::LN_EMIT; g.ln_proxy = stat0; g.ln_vec = c->enc[l + 1].ln1_g; g.ln_a = B16(p.h);
            g.ln_slots = slots0;
        }
        HIP_TRY(c, launch_gemm(g, s));
        xin = xl;
        if (layers_out)
            HIP_TRY(c, hipMemcpyAsync(layers_out + (size_t)l * M * 1024, xl, (size_t)M * 1024 * 4,
                                      hipMemcpyDeviceToDevice, s));
    }
    // ---- final LayerNorm (wav2vec2.py:905-906) and heads
    const bool sae = cfg.head == SLSMI_HEAD_SAE;
    const int feat = sae ? cfg.sae_features : SLSMI_SAE_SPARSE;
    const bool has_sae = sae && feat != SLSMI_SAE_OFF;
    if (sae_loss_out && !has_sae) return fail(c, SLSMI_ERR_INVALID, "sae_loss_out needs a head with the auto-encoder");
    // fp32 copy of the final-LN output: wanted by the caller, by the SAE loss, or pooled directly
    float* xfin = x_out ? x_out : (sae && (sae_loss_out || !has_sae) ? F32(p.recon) : nullptr);
    if (xfin || sae)
        HIP_TRY(c, launch_ln1024_f32_f32(xin, xfin, has_sae ? B16(p.h) : nullptr, has_sae ? c->sae_b_dec : nullptr,
                                         M, c->final_ln_g, c->final_ln_b, eps, fmt, s));
    if (cfg.head == SLSMI_HEAD_SLS) {
        HIP_TRY(c, launch_sls_weights(F32(p.rowdots), p.dots_stride, cfg.layers, c->fc0_b, F32(p.slsw), B, T, 16, s));
        HIP_TRY(c, launch_sls_sum_pool(F32(p.xs), p.x_stride, cfg.layers, F32(p.slsw), B16(p.pooled), 3 * p.sls_ld, B, T,
                                       0.f, 0.f, fmt, s, c->bn, cfg.bn_eps, p.sls_ld));
        HIP_TRY(c, RANGE(SLSMI_RANGE_POOLED, B16(p.pooled), B, (int)p.sls_ld, 3 * p.sls_ld));
        // fc1 has M = B rows only: split K over SLS_FC1_PARTS workgroup slices, each writing its partial
        // sums to its own slice; the tail kernel adds the slices in a fixed order (reproducible bits)
        GemmArgs g;
        g.fmt = fmt;
        g.split_ws = c->split_scoring ? c->split_ws : nullptr; g.split_ws_bytes = GEMM_SPLIT_WS_BYTES;
        // pooled features [hi | lo | hi] against fc1's [W_hi | W_hi | W_lo]: neither operand's rounding is felt
        g.A = B16(p.pooled); g.lda = 3 * p.sls_ld; g.W = c->sls_fc1_w; g.ldw = 3 * p.sls_ld; g.bias = nullptr;
        g.C = F32(p.fc1out); g.ldc = 1024; g.M = B; g.N = 1024; g.K = (int)(3 * p.sls_ld); g.k_algo = (int)p.sls_in; g.mode = GEMM_F32;
        g.splitk = SLS_FC1_PARTS;
        g.zC1 = (long)(B + 128) * 1024;
        HIP_TRY(c, launch_gemm(g, s));
        HIP_TRY(c, launch_sls_tail(F32(p.fc1out), SLS_FC1_PARTS, g.zC1, c->sls_fc1_b, c->fc3_w, c->fc3_b, logp_out, B, s));
    } else if (sae && !has_sae) {
        // use_sae=False: mean over time of the encoder output, then the classifier (model.py:242-247)
        HIP_TRY(c, launch_col_mean(xfin, nullptr, F32(p.saepool), B, T, 1024, s, nframes));
        HIP_TRY(c, launch_sae_tail(F32(p.saepool), c->cls_ln_g, c->cls_ln_b, c->cls_w1, c->cls_b1, c->cls_w2,
                                   c->cls_b2, F32(p.saehid), logp_out, B, 1024, s));
    } else if (sae) {
        const int D = cfg.sae_dict;
        const bool use_recon = feat == SLSMI_SAE_RECON;
        const bool decode = sae_loss_out || use_recon;
        GemmArgs g;
        g.fmt = fmt;
        g.split_ws = c->split_scoring ? c->split_ws : nullptr; g.split_ws_bytes = GEMM_SPLIT_WS_BYTES;
        g.A = B16(p.h); g.lda = 1024; g.W = c->sae_enc_w; g.ldw = 1024; g.bias = c->sae_enc_b;
        g.C = F32(p.enc); g.ldc = D; g.M = (int)M; g.N = D; g.K = 1024; g.mode = GEMM_F32_RELU;
        HIP_TRY(c, launch_gemm(g, s));
        if (cfg.sae_window > 1)  // model_window_topk.py:106-110
            HIP_TRY(c, launch_window_topk(F32(p.enc), decode ? B16(p.encb) : nullptr,
                                          reinterpret_cast<uint8_t*>(ws + p.wkeep), B, T, D, cfg.sae_k, cfg.sae_window,
                                          fmt, s));
        else
            HIP_TRY(c, launch_topk_rows(F32(p.enc), decode ? B16(p.encb) : nullptr, M, D, cfg.sae_k, fmt, s));
        float* recon = F32(p.x);  // the residual stream is dead after the final LayerNorm
        if (decode) {
            g = GemmArgs();
            g.fmt = fmt;
        g.split_ws = c->split_scoring ? c->split_ws : nullptr; g.split_ws_bytes = GEMM_SPLIT_WS_BYTES;
            g.A = B16(p.encb); g.lda = D; g.W = c->sae_dec_w; g.ldw = D; g.bias = nullptr;
            g.C = recon; g.ldc = 1024; g.M = (int)M; g.N = 1024; g.K = D; g.mode = GEMM_F32;
            HIP_TRY(c, launch_gemm(g, s));
        }
        if (sae_loss_out) {
            HIP_TRY(c, launch_mse(recon, c->sae_b_dec, xfin, M, F32(p.saepool), sae_loss_out, s));  // saepool: scratch until the pooling below
        }
        if (use_recon) {
            // classify mean_t (decoder(code) + b_dec) (model.py:231-233)
            HIP_TRY(c, launch_col_mean(recon, c->sae_b_dec, F32(p.saepool), B, T, 1024, s, nframes));
            HIP_TRY(c, launch_sae_tail(F32(p.saepool), c->cls_ln_g, c->cls_ln_b, c->cls_w1, c->cls_b1, c->cls_w2,
                                       c->cls_b2, F32(p.saehid), logp_out, B, 1024, s));
        } else {
            HIP_TRY(c, launch_col_mean(F32(p.enc), nullptr, F32(p.saepool), B, T, D, s, nframes));
            HIP_TRY(c, launch_sae_tail(F32(p.saepool), c->cls_ln_g, c->cls_ln_b, c->cls_w1, c->cls_b1, c->cls_w2,
                                       c->cls_b2, F32(p.saehid), logp_out, B, D, s));
        }
    }
    return SLSMI_OK;
}

}  // namespace

extern "C" {

int slsmi_forward(slsmi_ctx* c, const float* wav, int B, int L, void* workspace, size_t workspace_bytes,
                  float* logp_out, float* x_out, float* layers_out, float* sae_loss_out, void* stream) {
    return forward_impl(c, wav, B, L, nullptr, workspace, workspace_bytes, logp_out, x_out, layers_out, sae_loss_out,
                        stream);
}

int slsmi_forward_varlen(slsmi_ctx* c, const float* wav, int B, int L, const int32_t* lengths, void* workspace,
                         size_t workspace_bytes, float* logp_out, float* x_out, float* layers_out, void* stream) {
    if (!c) return SLSMI_ERR_INVALID;
    if (!lengths) return fail(c, SLSMI_ERR_INVALID, "null lengths");
    return forward_impl(c, wav, B, L, lengths, workspace, workspace_bytes, logp_out, x_out, layers_out, nullptr, stream);
}

int slsmi_workspace_view(const slsmi_ctx* c, int batch, int length, int which, size_t* offset, size_t* bytes) {
    if (!c || !offset || !bytes) return SLSMI_ERR_INVALID;
    Plan p;
    if (!make_plan(c, batch, length, &p)) return SLSMI_ERR_INVALID;
    if (which == SLSMI_VIEW_SAE_CODE && c->cfg.head == SLSMI_HEAD_SAE && c->cfg.sae_features != SLSMI_SAE_OFF) {
        *offset = p.enc;
        *bytes = (size_t)p.M * c->cfg.sae_dict * 4;
        return SLSMI_OK;
    }
    return SLSMI_ERR_INVALID;
}

const char* slsmi_last_error(const slsmi_ctx* c) { return c ? c->err.c_str() : "null context"; }

int slsmi_set_low_latency(slsmi_ctx* c, int enable) {
    if (!c) return SLSMI_ERR_INVALID;
    c->split_scoring = enable != 0;
    return SLSMI_OK;
}

int slsmi_set_ln_fold(slsmi_ctx* c, int enable) {
    if (!c) return SLSMI_ERR_INVALID;
    c->ln_fold = enable != 0;
    c->ln_fold_force_wide = enable == 2;
    return SLSMI_OK;
}

int slsmi_set_conv_precision(slsmi_ctx* c, unsigned split_mask, unsigned fp32_mask) {
    if (!c) return SLSMI_ERR_INVALID;
    const unsigned layers = ((1u << c->cfg.n_conv) - 1u) & ~1u;  // conv layers 1 .. n - 1
    if ((split_mask | fp32_mask) & ~layers) return fail(c, SLSMI_ERR_INVALID, "mask names a conv layer that is not a GEMM layer of this model");
    c->conv_split_mask = split_mask;
    c->conv_fp32_mask = fp32_mask;
    return SLSMI_OK;
}

int slsmi_set_range_report(slsmi_ctx* c, uint32_t* out, int capacity) {
    if (!c || capacity < 0 || (out && capacity < 1)) return SLSMI_ERR_INVALID;
    c->range_out = out;
    c->range_capacity = out ? capacity : 0;
    return SLSMI_OK;
}

int slsmi_set_posconv_slab(slsmi_ctx* c, int enable) {
    if (!c) return SLSMI_ERR_INVALID;
    c->posconv_slab = enable != 0;
    return SLSMI_OK;
}

int slsmi_destroy(slsmi_ctx* c) {
    if (!c) return SLSMI_ERR_INVALID;
    (void)hipSetDevice(c->device);
    for (void* p : c->allocs) (void)hipFree(p);
    if (c->train) train_destroy(c);
    delete c;
    return SLSMI_OK;
}

// ---------------------------------------------------------------- single-kernel entry points
static int hip_rc(hipError_t e) { return e == hipSuccess ? SLSMI_OK : (e == hipErrorInvalidValue ? SLSMI_ERR_INVALID : SLSMI_ERR_HIP); }

// The single-kernel GEMM entry points have no context: they share one K-split scratch per device, made
// on first use.  Calls through them must not overlap in time on one device.
static float* standalone_split_ws() {
    static std::mutex mu;
    static float* scratch[64] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!scratch[dev]) {
        void* q = nullptr;
        if (hipMalloc(&q, GEMM_SPLIT_WS_BYTES) != hipSuccess) return nullptr;
        scratch[dev] = reinterpret_cast<float*>(q);
    }
    return scratch[dev];
}

int slsmi_gemm_bf16(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, const float* bias, void* C,
                    int64_t ldc, int M, int N, int K, int mode, int bn, int splitk, int fmt, void* stream) {
    GemmArgs g;
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.C = C; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K; g.mode = mode; g.bn = bn; g.splitk = splitk < 1 ? 1 : splitk; g.fmt = fmt;
    g.split_ws = standalone_split_ws(); g.split_ws_bytes = g.split_ws ? GEMM_SPLIT_WS_BYTES : 0;
    return hip_rc(launch_gemm(g, reinterpret_cast<hipStream_t>(stream)));
}

int slsmi_gemm_bf16_period(const uint16_t* A, int64_t lda, int ka, const uint16_t* W, int64_t ldw, const float* bias, void* C,
                           int64_t ldc, int M, int N, int K, int mode, int bn, int fmt, void* stream) {
    GemmArgs g;
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.C = C; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K; g.ka = ka; g.mode = mode; g.bn = bn; g.fmt = fmt;
    g.split_ws = standalone_split_ws(); g.split_ws_bytes = g.split_ws ? GEMM_SPLIT_WS_BYTES : 0;
    return hip_rc(launch_gemm(g, reinterpret_cast<hipStream_t>(stream)));
}

// ---- the GEMM entry point that allocates nothing (its scratch: `scratch` at the top of this file)
int slsmi_gemm_bf16_static(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, const float* bias, void* C,
                           int64_t ldc, int M, int N, int K, int mode, int bn, int splitk, int fmt, void* stream) {
    GemmArgs g;
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.C = C; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K; g.mode = mode; g.bn = bn; g.splitk = splitk < 1 ? 1 : splitk; g.fmt = fmt;
    // the same K split slsmi_gemm_bf16 makes with its 64 MiB scratch, or an error: never a different one
    const size_t need = gemm_split_scratch_bytes(g, GEMM_SPLIT_WS_BYTES);
    if (need > STATIC_SCRATCH_BYTES) {
        fprintf(stderr, "slsmi_gemm_bf16_static: this GEMM (M=%d N=%d K=%d) is cut along K and needs %zu bytes of "
                        "scratch; the static array holds %zu\n", M, N, K, need, STATIC_SCRATCH_BYTES);
        return SLSMI_ERR_WORKSPACE;
    }
    if (need > 0) {
        g.split_ws = static_scratch_ptr();
        if (!g.split_ws) return SLSMI_ERR_HIP;
        g.split_ws_bytes = GEMM_SPLIT_WS_BYTES;  // the slices chosen fit (checked above)
    }
    return hip_rc(launch_gemm(g, reinterpret_cast<hipStream_t>(stream)));
}

int slsmi_time_gemm_bf16(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, const float* bias, void* C,
                         int64_t ldc, int M, int N, int K, int mode, int bn, int splitk, int fmt, int reps,
                         float* ms_out, void* stream) {
    if (!ms_out || reps < 1) return SLSMI_ERR_INVALID;
    GemmArgs g;
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.C = C; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K; g.mode = mode; g.bn = bn; g.splitk = splitk < 1 ? 1 : splitk; g.fmt = fmt;
    g.split_ws = standalone_split_ws(); g.split_ws_bytes = g.split_ws ? GEMM_SPLIT_WS_BYTES : 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess) return SLSMI_ERR_HIP;
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return SLSMI_ERR_HIP; }
    int rc = hip_rc(launch_gemm(g, s));  // warm-up launch, also validates the arguments
    if (rc == SLSMI_OK) {
        (void)hipEventRecord(e0, s);
        for (int i = 0; i < reps && rc == SLSMI_OK; ++i) rc = hip_rc(launch_gemm(g, s));
        (void)hipEventRecord(e1, s);
        if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(ms_out, e0, e1) != hipSuccess)
            rc = SLSMI_ERR_HIP;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

int slsmi_debug_mfma_rate(float* sink, int64_t sink_floats, int fmt, int iters, double* tflops_out, void* stream) {
    if (!sink || !tflops_out || iters < 1 || (fmt != 0 && fmt != 1)) return SLSMI_ERR_INVALID;
    int dev = 0, n_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return SLSMI_ERR_HIP;
    const int blocks = 2 * n_cu;  // two waves per SIMD
    if (sink_floats < (int64_t)blocks * 256) return SLSMI_ERR_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess) return SLSMI_ERR_HIP;
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return SLSMI_ERR_HIP; }
    int rc = hip_rc(launch_mfma_rate(sink, blocks, iters, fmt, s));  // warm-up: the clock settles under the load
    float ms = 0.f;
    if (rc == SLSMI_OK) {
        (void)hipEventRecord(e0, s);
        rc = hip_rc(launch_mfma_rate(sink, blocks, iters, fmt, s));
        (void)hipEventRecord(e1, s);
        if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) rc = SLSMI_ERR_HIP;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc == SLSMI_OK) *tflops_out = (double)blocks * 4 * iters * 16 * 16384.0 / (ms * 1e-3) / 1e12;
    return rc;
}

int slsmi_adam_step(const int64_t* table, const int32_t* items, int64_t n_items, int chunk_elems, double lr, double beta1,
                    double beta2, double eps, double weight_decay, void* stream) {
    if (!table || !items || n_items < 1 || chunk_elems < 4 || (chunk_elems & 3)) return SLSMI_ERR_INVALID;
    if (!(beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1. && eps >= 0.)) return SLSMI_ERR_INVALID;
    static_assert(sizeof(long long) == sizeof(int64_t) && sizeof(int2) == 2 * sizeof(int32_t), "table layouts");
    return hip_rc(launch_adam(reinterpret_cast<const long long*>(table), reinterpret_cast<const int2*>(items), (long)n_items,
                              chunk_elems, lr, beta1, beta2, eps, weight_decay, reinterpret_cast<hipStream_t>(stream)));
}

int slsmi_debug_gemm_timing_begin(int max_launches) {
    if (max_launches < 1 || max_launches > 65536) return SLSMI_ERR_INVALID;
    return hip_rc(gemm_timing_begin(max_launches));
}

int slsmi_debug_gemm_timing_end(int* launches, int* dropped, double* seconds, double* flops) {
    if (!launches || !dropped || !seconds || !flops) return SLSMI_ERR_INVALID;
    return hip_rc(gemm_timing_end(launches, dropped, seconds, flops));
}

int slsmi_conv0_ln_gelu(const float* wav, int B, int L, int T, const float* w, const float* bias, const float* gamma,
                        const float* beta, float eps, uint16_t* out, int64_t rows_per_utt, int fmt, void* stream) {
    return hip_rc(launch_conv0_ln_gelu(wav, B, L, T, w, bias, gamma, beta, eps, out, rows_per_utt, fmt,
                                       reinterpret_cast<hipStream_t>(stream)));
}

int slsmi_posconv(const uint16_t* xg, const uint16_t* w, const float* bias, float* x, int B, int T, int64_t slab_rows,
                  int fmt, int form, void* stream) {
    if (!xg || !w || !bias || !x || B < 1 || T < 1 || slab_rows < T + 127 || slab_rows > (1 << 20) || (fmt != 0 && fmt != 1))
        return SLSMI_ERR_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (form == 1) {
        if (!posconv_slab_supported(T, (int)slab_rows)) return SLSMI_ERR_INVALID;
        return hip_rc(launch_posconv_slab(xg, w, bias, x, B, T, (int)slab_rows, fmt, s));
    }
    if (form != 0) return SLSMI_ERR_INVALID;
    GemmArgs g;
    g.fmt = fmt;
    g.A = xg; g.lda = 64; g.W = w; g.ldw = 8192; g.bias = bias;
    g.C = x; g.ldc = 1024; g.M = T; g.N = 64; g.K = 8192; g.mode = GEMM_F32_RESID_GELU; g.bn = 64;
    g.nz = B * 16; g.zA = slab_rows * 64; g.zWmod = 16; g.zW = 64L * 8192; g.zBias = 64;
    g.zCdiv = 16; g.zC1 = (long)T * 1024; g.zC2 = 64;
    return hip_rc(launch_gemm(g, s));
}

int slsmi_gemm_tn(const uint16_t* a, int64_t lda, const uint16_t* b, int64_t ldb, int b_fmt, float* c, int64_t ldc,
                  int64_t M, int N, int K, int nz, int64_t za, int64_t zb, int64_t zc, float alpha, void* stream) {
    if (b_fmt != 0 && b_fmt != 1) return SLSMI_ERR_INVALID;
    return hip_rc(launch_gemm_tn(a, lda, za, b, ldb, zb, b_fmt, c, ldc, zc, M, N, K, nz, alpha,
                                 reinterpret_cast<hipStream_t>(stream)));
}

int slsmi_gemm_tn_grouped(const uint16_t* a, int64_t lda, const uint16_t* b, int64_t ldb, int b_fmt, float* c, int64_t ldc,
                          int64_t M, int N, int K, int nz, int64_t za, int64_t zb, int64_t zc, float alpha, int group_rows,
                          int64_t ga, int64_t gb, void* stream) {
    if ((b_fmt != 0 && b_fmt != 1) || group_rows < 1) return SLSMI_ERR_INVALID;
    return hip_rc(launch_gemm_tn(a, lda, za, b, ldb, zb, b_fmt, c, ldc, zc, M, N, K, nz, alpha,
                                 reinterpret_cast<hipStream_t>(stream), group_rows, ga, gb));
}

int slsmi_ln512_bf16(const uint16_t* in, uint16_t* out, int64_t rows, const float* gamma, const float* beta, float eps,
                     int gelu, int fmt, void* stream) {
    return hip_rc(launch_ln512_bf16(in, out, rows, gamma, beta, eps, gelu, fmt, reinterpret_cast<hipStream_t>(stream)));
}

int slsmi_ln512_f32(const float* in, uint16_t* out, int64_t rows, const float* gamma, const float* beta, float eps, int gelu,
                    int fmt, void* stream) {
    return hip_rc(launch_ln512_f32(in, out, rows, gamma, beta, eps, gelu, fmt, reinterpret_cast<hipStream_t>(stream)));
}

int slsmi_ln512_split3(const uint16_t* in, uint16_t* out, int64_t rows, const float* gamma, const float* beta, float eps,
                       int fmt, void* stream) {
    return hip_rc(launch_ln512_split3(in, out, rows, gamma, beta, eps, fmt, reinterpret_cast<hipStream_t>(stream)));
}

int slsmi_ln1024_f32_bf16(const float* in, uint16_t* out, int64_t rows, const float* gamma, const float* beta,
                          float eps, int fmt, void* stream) {
    return hip_rc(launch_ln1024_f32_bf16(in, out, rows, gamma, beta, eps, fmt, reinterpret_cast<hipStream_t>(stream)));
}

int slsmi_attention(const uint16_t* qkv, uint16_t* out, int B, int T, int fmt, void* stream) {
    return hip_rc(launch_attention(qkv, out, B, T, fmt, reinterpret_cast<hipStream_t>(stream)));
}

int slsmi_attention_varlen(const uint16_t* qkv, uint16_t* out, int B, int T, const int32_t* nframes, int fmt,
                           void* stream) {
    if (!nframes) return SLSMI_ERR_INVALID;
    return hip_rc(launch_attention(qkv, out, B, T, fmt, reinterpret_cast<hipStream_t>(stream), nframes));
}

int slsmi_gemm_resid_rowdot(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, const float* bias,
                            const float* resid, float* C, int64_t ldc, int M, int N, int K, int bn, int fmt,
                            const float* dot_w, float* dot_out, void* stream) {
    GemmArgs g;
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.C = C; g.ldc = ldc;
    g.resid = resid == C ? nullptr : resid;
    g.M = M; g.N = N; g.K = K; g.mode = GEMM_F32_RESID; g.bn = bn; g.fmt = fmt;
    g.dot_w = dot_w; g.dot_out = dot_out;
    g.split_ws = standalone_split_ws(); g.split_ws_bytes = g.split_ws ? GEMM_SPLIT_WS_BYTES : 0;
    if (!dot_w || !dot_out) return SLSMI_ERR_INVALID;
    return hip_rc(launch_gemm(g, reinterpret_cast<hipStream_t>(stream)));
}

int slsmi_gemm_bf16_ln(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, const float* bias,
                       const float* resid, void* C, int64_t ldc, int M, int N, int K, int mode, int fmt, int ln_mode,
                       const float* ln_proxy, const float* ln_vec, uint16_t* ln_a, float* ln_slots, float* ln_stat,
                       float ln_eps, const float* dot_w, float* dot_out, void* stream) {
    if (ln_mode != GemmArgs::LN_EMIT && ln_mode != GemmArgs::LN_APPLY) return SLSMI_ERR_INVALID;
    GemmArgs g;
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.C = C; g.ldc = ldc;
    g.resid = resid == C ? nullptr : resid;
    g.M = M; g.N = N; g.K = K; g.mode = mode; g.fmt = fmt;
    g.dot_w = dot_w; g.dot_out = dot_out;
    g.ln_mode = ln_mode; g.ln_proxy = ln_proxy; g.ln_vec = ln_vec; g.ln_a = ln_a; g.ln_slots = ln_slots;
    g.ln_stat = ln_stat; g.ln_eps = ln_eps;
    return hip_rc(launch_gemm(g, reinterpret_cast<hipStream_t>(stream)));
}

int slsmi_ln_fold_prep(const float* x, uint16_t* ln_a, float* ln_proxy, float* ln_slots, int64_t rows, const float* gamma,
                       float eps, int fmt, void* stream) {
    return hip_rc(launch_ln1024_fold_prep(x, ln_a, ln_proxy, ln_slots, rows, gamma, eps, fmt,
                                          reinterpret_cast<hipStream_t>(stream)));
}

int slsmi_ln_fold_rows(const float* x, const float* proxy_in, uint16_t* ln_a, float* aff, float* stat, int64_t rows,
                       const float* gamma, float eps, int fmt, void* stream) {
    return hip_rc(launch_ln1024_fold_rows(x, proxy_in, ln_a, aff, stat, rows, gamma, eps, fmt,
                                          reinterpret_cast<hipStream_t>(stream)));
}

int slsmi_gemm_bf16_ln_rows(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, const float* c, const float* g,
                            const float* aff, void* C, int64_t ldc, int M, int N, int K, int mode, int bn, int fmt,
                            void* stream) {
    GemmArgs a;
    a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.bias = c; a.C = C; a.ldc = ldc;
    a.M = M; a.N = N; a.K = K; a.mode = mode; a.bn = bn; a.fmt = fmt;
    a.ln_mode = GemmArgs::LN_APPLY; a.ln_vec = g; a.ln_aff = aff;
    if (!aff) return SLSMI_ERR_INVALID;
    return hip_rc(launch_gemm(a, reinterpret_cast<hipStream_t>(stream)));
}

int slsmi_ln_fold_consts(const uint16_t* W, const float* bias, const float* gamma, const float* beta, float* g, float* c,
                         int N, int fmt, void* stream) {
    return hip_rc(launch_ln_fold_consts(W, bias, gamma, beta, g, c, N, fmt, reinterpret_cast<hipStream_t>(stream)));
}

int slsmi_sls_weights(const float* dots, int64_t layer_stride, int layers, int slices, const float* fc0_b, float* w,
                      int B, int T, void* stream) {
    return hip_rc(launch_sls_weights(dots, layer_stride, layers, fc0_b, w, B, T, slices,
                                     reinterpret_cast<hipStream_t>(stream)));
}

int slsmi_sls_sum_pool(const float* xs, int64_t layer_stride, int layers, const float* w, uint16_t* pooled,
                       int64_t ldp, int B, int T, float bn_scale, float bn_shift, int fmt, void* stream) {
    return hip_rc(launch_sls_sum_pool(xs, layer_stride, layers, w, pooled, ldp, B, T, bn_scale, bn_shift, fmt,
                                      reinterpret_cast<hipStream_t>(stream)));
}

int slsmi_window_topk(float* enc, uint16_t* enc_bf16, void* scratch, size_t scratch_bytes, int B, int T, int dict,
                      int k, int window, int fmt, void* stream) {
    const long need = window_topk_scratch_bytes(B, T, dict, window);
    if (need <= 0 || !scratch || scratch_bytes < (size_t)need) return SLSMI_ERR_WORKSPACE;
    return hip_rc(launch_window_topk(enc, enc_bf16, reinterpret_cast<uint8_t*>(scratch), B, T, dict, k, window, fmt,
                                     reinterpret_cast<hipStream_t>(stream)));
}

size_t slsmi_window_topk_scratch_bytes(int B, int T, int dict, int window) {
    const long n = window_topk_scratch_bytes(B, T, dict, window);
    return n > 0 ? (size_t)n : 0;
}

int slsmi_topk_rows(float* enc, uint16_t* enc_bf16, int64_t rows, int dict, int k, int fmt, void* stream) {
    return hip_rc(launch_topk_rows(enc, enc_bf16, rows, dict, k, fmt, reinterpret_cast<hipStream_t>(stream)));
}

}  // extern "C"
