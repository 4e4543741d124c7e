// Training step behind the C ABI: a forward that keeps what the backward needs, and the backward
// (reference main.py:201-258: model(batch_x, return_sae_loss=True) -> loss.backward()).
// The in-tree model main.py trains is covered in all its forms: XLS-R backbone + TopK SAE (per frame or by window
// votes) + classifier on the code, on its reconstruction or on the encoder output (model.py:171-260,
// model_window_topk.py:271-393); the upstream SLS head sits behind the same backbone step (sls_train.hip).
// Gradients of every parameter from the positional conv upwards in this file's first part and of the conv
// feature extractor in its second.
#include <hip/hip_runtime.h>

#include <cstring>
#include <algorithm>
#include <set>
#include <string>

#include "ctx.h"

using namespace slsmi;

namespace slsmi {

struct LayerT {
    bf16_t* wqkvT = nullptr;  // [1024][3072]
    bf16_t* woT = nullptr;    // [1024][1024]
    bf16_t* w1T = nullptr;    // [1024][ffn]
    bf16_t* w2T = nullptr;    // [ffn][1024]
    bool have[6] = {false, false, false, false, false, false};  // q, k, v, out_proj, fc1, fc2 written by the weight load
};

struct TrainState {
    std::vector<LayerT> lt;
    bf16_t* sae_encT = nullptr;  // [1024][dict]
    bf16_t* sae_decT = nullptr;  // [dict][1024]
    bf16_t* projT = nullptr;     // [512][1024]
    bf16_t* pos_wf = nullptr;    // flipped positional-conv weight [16][64 ci][128][64 co]
    bf16_t* convT[SLSMI_MAX_CONV] = {nullptr};  // [k*512][512]
    std::map<std::string, float*> grads;
    // encoder-layer weight matrices whose Adam step the next backward takes in the weight-gradient kernel's epilogue
    // (slsmi_train_set_fused_adam; emptied by that backward)
    struct Fused {
        float* p;
        float* m;
        float* v;
        long long step;
    };
    std::map<std::string, Fused> fused;
    std::set<std::string> fresh;   // names whose 16-bit copies that backward wrote as well (train_copy_is_fresh)
    double adam_lr = 0, adam_b1 = 0, adam_b2 = 0, adam_eps = 0, adam_wd = 0;
    bool adam_hyper_set = false;
    bool ready = false;
    // what the last slsmi_train_forward ran on: slsmi_train_backward must be given the same
    const void* fwd_ws = nullptr;
    const float* fwd_wav = nullptr;
    int fwd_B = 0, fwd_L = 0;
    bool fwd_mask = false;
    bool fwd_backbone_only = false;
    bool fwd_sls = false;   // the upstream SLS head ran behind the backbone (slsmi_train_sls_forward)
    // the caller's dropout / layerdrop draws for the next forward and its backward (slsmi_train_set_masks)
    slsmi_train_masks masks = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    std::vector<uint8_t> layer_keep;  // copy of masks.layer_keep (empty: every layer runs)
    // attention-probability dropout of the next forward and its backward (slsmi_train_set_attention_masks)
    const uint8_t* attn_keep_qk = nullptr;
    const uint8_t* attn_keep_kq = nullptr;
    int64_t attn_layer_stride = 0;
    int attn_pitch = 0;
    float attn_scale = 1.f;
    // recorded by the backward on its stream once every encoder-layer parameter gradient has been enqueued (null: none)
    hipEvent_t layer_grads_event = nullptr;
};

}  // namespace slsmi

namespace {

constexpr int FB = SLSMI_FMT_BF16;

struct TPlan {
    int B = 0, L = 0, n = 0, frames = 0, slab = 0;
    int T[SLSMI_MAX_CONV], P[SLSMI_MAX_CONV];
    long M = 0, Mpad = 0, Mp = 0;
    int wide = 0;  // widest activation row: max(ffn, 3072, dict)
    size_t a[SLSMI_MAX_CONV], y[SLSMI_MAX_CONV];
    size_t featln = 0, x0 = 0, xg = 0, u = 0, xs = 0;
    size_t h1 = 0, qkv = 0, ao = 0, xmid = 0, h2 = 0, fpre = 0, f = 0;  // per layer, strides below
    size_t s_row16 = 0, s_row32 = 0, s_qkv = 0, s_ffn = 0;
    size_t xf = 0, hsae = 0, code = 0, code16 = 0, recon = 0, pooled = 0, hidden = 0, logp = 0, mask = 0, msepart = 0, wkeep = 0;
    // backward scratch
    size_t G0 = 0, G1 = 0, T1 = 0, g16 = 0, big16 = 0, colpart = 0, lnpart = 0, dcode = 0;
    // the output gradients of the encoder layers' fc2, fc1, out_proj and q/k/v, kept per layer ([Mpad][cols] bfloat16
    // each) for the weight and bias gradients that run behind the layer loop
    size_t rD2 = 0, rD1 = 0, rDo = 0, rDq = 0;
    size_t lnparts = 0;
    size_t stats = 0, dh1 = 0, dhb = 0, dpool = 0, du16 = 0, dug = 0, dwfold = 0;
    size_t cda = 0, cda2 = 0, cdy = 0, cdcol = 0, ctA = 0, ctB = 0, c0part = 0, c0sum = 0;  // conv stack
    long Rp = 0;
    // the upstream SLS head in training form (contexts with that head only): per-layer time means, layer weights and
    // their gradients, the weighted sum, BatchNorm partials and statistics, pooled features and the window positions
    // they came from, fc1 / fc3 values, and the backward's buffers up to the gradient at every layer output
    long sls_P = 0;
    size_t sy0 = 0, swl = 0, sdwl = 0, sdz = 0, sfull = 0, spart = 0, sstat = 0, spool = 0, samax = 0, sh1 = 0, sa3 = 0, slogp = 0;
    size_t sdh1 = 0, sdpart = 0, sdbn = 0, spart2 = 0, sdlay = 0;
    size_t total = 0;
};

bool make_tplan(const slsmi_ctx* c, int B, int L, TPlan* p) {
    const slsmi_config& cfg = c->cfg;
    if (B <= 0 || B > 64 || L <= 0) return false;
    p->B = B;
    p->L = L;
    p->n = cfg.n_conv;
    if (!conv_frames(cfg, L, p->T)) return false;
    const int n = cfg.n_conv;
    p->frames = p->T[n - 1];
    if (p->frames > 256 || p->frames < 3) return false;  // the attention backward holds T <= 256 frames
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
    p->Mpad = (p->M + 127) / 128 * 128 + 128;
    p->Mp = (p->M + 127) / 128 * 128;
    p->slab = (p->frames + 128 + 7) / 8 * 8;
    const int D = cfg.sae_dict;
    p->wide = cfg.ffn_dim > 3072 ? cfg.ffn_dim : 3072;
    if (D > p->wide) p->wide = D;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += align_up(bytes);
        return o;
    };
    for (int i = 0; i < n; ++i) {
        const size_t rows = (size_t)B * p->P[i] + 64;
        p->a[i] = take(rows * 512 * 2);
        p->y[i] = i ? take(rows * 512 * 2) : 0;
    }
    p->featln = take(((size_t)B * p->P[n - 1] + 64) * 512 * 2);
    const size_t Mpad = (size_t)p->Mpad;
    p->x0 = take(Mpad * 1024 * 4);
    p->xg = take((size_t)B * 16 * p->slab * 64 * 2);
    p->u = take(Mpad * 1024 * 4);
    p->s_row32 = align_up(Mpad * 1024 * 4);
    p->s_row16 = align_up(Mpad * 1024 * 2);
    p->s_qkv = align_up(Mpad * 3072 * 2);
    p->s_ffn = align_up(Mpad * cfg.ffn_dim * 2);
    const size_t NL = (size_t)cfg.layers;
    p->xs = take((NL + 1) * p->s_row32);
    p->h1 = take(NL * p->s_row16);  // h1, ao, h2 at equal distances: one batched transpose per layer in the backward
    p->ao = take(NL * p->s_row16);
    p->h2 = take(NL * p->s_row16);
    p->qkv = take(NL * p->s_qkv);
    p->xmid = take(NL * p->s_row32);
    p->fpre = take(NL * p->s_ffn);
    p->f = take(NL * p->s_ffn);
    p->xf = take(Mpad * 1024 * 4);
    p->hsae = take(Mpad * 1024 * 2);
    p->code = take(Mpad * D * 4);
    p->code16 = take(Mpad * D * 2);
    p->recon = take(Mpad * 1024 * 4);
    p->pooled = take((size_t)B * D * 4);
    p->hidden = take((size_t)B * 256 * 4);
    p->logp = take((size_t)B * 2 * 4);
    p->mask = take((size_t)B * 256 * 4);
    p->msepart = take((size_t)MSE_PARTS * 4);
    if (cfg.head == SLSMI_HEAD_SAE && cfg.sae_features != SLSMI_SAE_OFF && cfg.sae_window > 1) {
        if (p->frames < cfg.sae_window) return false;
        p->wkeep = take((size_t)window_topk_scratch_bytes(B, p->frames, D, cfg.sae_window));
    }
    p->G0 = take(Mpad * 1024 * 4);
    p->G1 = take(Mpad * 1024 * 4);
    p->T1 = take(Mpad * 1024 * 4);
    p->g16 = take(Mpad * 1024 * 2);
    p->big16 = take(Mpad * (size_t)p->wide * 2);
    p->rD2 = take(NL * p->s_row16);
    p->rD1 = take(NL * p->s_ffn);
    p->rDo = take(NL * p->s_row16);
    p->rDq = take(NL * p->s_qkv);
    p->colpart = take((size_t)COLSUM_MAX_PARTS * 8192 * 4);
    p->lnpart = take((size_t)LN_BWD_MAX_GROUPS * 2 * 1024 * 4);
    p->lnparts = take(2 * NL * (size_t)ln_bwd_groups(p->M) * 2 * 1024 * 4);   // the same for each of the encoder layers' LayerNorms
    p->dcode = take(Mpad * D * 4);
    p->stats = take((size_t)B * 2 * 4);
    p->dh1 = take((size_t)B * 256 * 4);
    p->dhb = take((size_t)B * D * 4);
    p->dpool = take((size_t)B * D * 4);
    p->du16 = take(Mpad * 1024 * 2);
    p->dug = take((size_t)B * 16 * p->slab * 64 * 2);
    p->dwfold = take((size_t)1024 * 8192 * 4);
    // conv feature extractor: gradients at a_i / a_{i-1} (fp32), at y_i (bfloat16), the GEMM's per-slice
    // gradient, transposed operands with the row count rounded up for the K dimension
    const size_t R0 = (size_t)B * p->P[0], R1 = n > 1 ? (size_t)B * p->P[1] : 0;
    p->Rp = (long)((R1 + 127) / 128 * 128);
    p->cda = take(R0 * 512 * 4);
    p->cda2 = take((R1 + 64) * 512 * 4);
    p->cdy = take((R1 + 64) * 512 * 2);
    p->cdcol = take((R1 + 64) * (size_t)3 * 512 * 2);
    p->ctA = take((size_t)512 * p->Rp * 2);
    p->ctB = take((size_t)3 * 512 * p->Rp * 2);
    p->c0part = take((size_t)COLSUM_MAX_PARTS * 6656 * 4);
    p->c0sum = take((size_t)6656 * 4);
    if (cfg.head == SLSMI_HEAD_SLS) {
        const size_t Bz = (size_t)B, pairs = Bz * NL;
        p->sls_P = (long)(p->frames / 3) * (1024 / 3);
        const size_t P = (size_t)p->sls_P;
        p->sy0 = take(pairs * 1024 * 4);
        p->swl = take(pairs * 4);
        p->sdwl = take((size_t)sls_dweight_chunks(p->frames) * pairs * 4);
        p->sdz = take(pairs * 4);
        p->sfull = take(Mpad * 1024 * 4);
        p->spart = take((size_t)p->M * 2 * 4);
        p->sstat = take(64);
        p->spool = take(Bz * P * 4);
        p->samax = take(Bz * P);
        p->sh1 = take(Bz * 1024 * 4);
        p->sa3 = take(Bz * 2 * 4);
        p->slogp = take(Bz * 2 * 4);
        p->sdh1 = take(Bz * 1024 * 4);
        p->sdpart = take((size_t)sls_fc1_chunks() * Bz * P * 4);
        p->sdbn = take(Mpad * 1024 * 4);
        p->spart2 = take(Bz * (size_t)(p->frames / 3) * 2 * 4);
        p->sdlay = take(NL * (size_t)p->M * 1024 * 4);
    }
    p->total = off;
    return true;
}

// backbone_only: the step ends at the encoder layers' outputs (any head)
int check_train(slsmi_ctx* c, bool backbone_only = false) {
    if (!c) return SLSMI_ERR_INVALID;
    if (!c->train) return fail(c, SLSMI_ERR_INVALID, "slsmi_enable_training has not been called on this context");
    if (!c->finalized || !c->train->ready)
        return fail(c, SLSMI_ERR_MISSING, "slsmi_finalize has not been called after the last weight load");
    const slsmi_config& cfg = c->cfg;
    if (!backbone_only && cfg.head != SLSMI_HEAD_SAE)
        return fail(c, SLSMI_ERR_INVALID, "slsmi_train_forward / _backward run the in-tree model (any of its forms); the upstream "
                                          "SLS model trains on slsmi_train_sls_forward, other heads on slsmi_train_backbone_forward");
    return SLSMI_OK;
}

// C[M][N] = A[M][K] W[N][K]^T in bfloat16 operands
hipError_t gemm_bf(const slsmi_ctx* c, const bf16_t* A, long lda, const bf16_t* W, long ldw, void* C, long ldc, long M, int N, long K,
                   int mode, const float* resid, hipStream_t s, int* deferred_slices = nullptr) {
    GemmArgs g;
    g.fmt = FB;
    g.deferred_slices = deferred_slices;
    g.split_ws = c->split_ws; g.split_ws_bytes = c->split_ws ? GEMM_SPLIT_WS_BYTES : 0;
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = nullptr; g.C = C; g.ldc = ldc;
    g.M = (int)M; g.N = N; g.K = (int)K; g.mode = mode; g.resid = resid;
    return launch_gemm(g, s);
}

}  // namespace

namespace slsmi {

// Called by slsmi_finalize when the context is in training mode: transposed bfloat16 copies of the
// GEMM weights (input gradients are dy (W^T)^T on the same C = A W^T kernel).
int train_prepare(slsmi_ctx* c, hipStream_t s) {
    TrainState* t = c->train;
    const slsmi_config& cfg = c->cfg;
    const int fmt = cfg.operand_format;
    const long C = 1024, F = cfg.ffn_dim, D = cfg.sae_dict;
    t->lt.resize(cfg.layers);
    int r;
    for (int l = 0; l < cfg.layers; ++l) {
        const EncLayer& E = c->enc[l];
        LayerT& T = t->lt[l];
        if ((r = dev_alloc(c, &T.wqkvT, (size_t)C * 3 * C))) return r;
        if ((r = dev_alloc(c, &T.woT, (size_t)C * C))) return r;
        if ((r = dev_alloc(c, &T.w1T, (size_t)C * F))) return r;
        if ((r = dev_alloc(c, &T.w2T, (size_t)F * C))) return r;
        // (matrices loaded after training was switched on came with their transpose: train_layer_wT)
        if (!(T.have[0] && T.have[1] && T.have[2])) HIP_TRY(c, launch_transpose16(E.wqkv, C, 3 * C, (int)C, T.wqkvT, 3 * C, fmt, s));
        if (!T.have[3]) HIP_TRY(c, launch_transpose16(E.wo, C, C, (int)C, T.woT, C, fmt, s));
        if (!T.have[4]) HIP_TRY(c, launch_transpose16(E.w1, C, F, (int)C, T.w1T, F, fmt, s));
        if (!T.have[5]) HIP_TRY(c, launch_transpose16(E.w2, F, C, (int)F, T.w2T, C, fmt, s));
        for (bool& h : T.have) h = true;  // in step with the forward's copies from here on
    }
    if (cfg.head == SLSMI_HEAD_SAE && cfg.sae_features != SLSMI_SAE_OFF) {
        if ((r = dev_alloc(c, &t->sae_encT, (size_t)C * D))) return r;
        if ((r = dev_alloc(c, &t->sae_decT, (size_t)D * C))) return r;
        HIP_TRY(c, launch_transpose16(c->sae_enc_w, C, D, (int)C, t->sae_encT, D, fmt, s));
        HIP_TRY(c, launch_transpose16(c->sae_dec_w, D, C, (int)D, t->sae_decT, C, fmt, s));
    }
    if ((r = dev_alloc(c, &t->projT, (size_t)512 * C))) return r;
    HIP_TRY(c, launch_transpose16(c->proj_w, 512, C, 512, t->projT, C, fmt, s));
    if ((r = dev_alloc(c, &t->pos_wf, (size_t)16 * 64 * 128 * 64))) return r;
    HIP_TRY(c, launch_posconv_flip(c->pos_w, fmt, t->pos_wf, s));
    for (int i = 1; i < cfg.n_conv; ++i) {
        const long K = (long)cfg.conv_kernel[i] * 512;
        if ((r = dev_alloc(c, &t->convT[i], (size_t)K * 512))) return r;
        HIP_TRY(c, launch_transpose16(c->conv[i].w, K, 512, (int)K, t->convT[i], 512, fmt, s));
    }
    t->ready = true;
    return SLSMI_OK;
}

bf16_t* train_layer_wT(slsmi_ctx* c, int layer, int which, long* ldo) {
    TrainState* t = c->train;
    if (!t || layer < 0 || layer >= c->cfg.layers || which < 0 || which > 5) return nullptr;
    if ((int)t->lt.size() != c->cfg.layers) t->lt.resize(c->cfg.layers);
    LayerT& T = t->lt[layer];
    const long C = 1024, F = c->cfg.ffn_dim;
    bf16_t** slot = which <= 2 ? &T.wqkvT : which == 3 ? &T.woT : which == 4 ? &T.w1T : &T.w2T;
    const size_t count = which <= 2 ? (size_t)C * 3 * C : which == 3 ? (size_t)C * C : (size_t)C * F;
    if (dev_alloc(c, slot, count) != SLSMI_OK) return nullptr;
    T.have[which] = true;
    *ldo = which <= 2 ? 3 * C : which == 3 ? C : which == 4 ? F : C;
    return which <= 2 ? *slot + which * C : *slot;
}

bool train_copy_is_fresh(const slsmi_ctx* c, const std::string& name) {
    return c->train && c->train->fresh.count(name) != 0;
}

void train_forget_fresh_copies(slsmi_ctx* c, const std::string* name) {
    if (!c->train) return;
    if (name)
        c->train->fresh.erase(*name);
    else
        c->train->fresh.clear();
}

void train_destroy(slsmi_ctx* c) {
    delete c->train;
    c->train = nullptr;
}

}  // namespace slsmi

extern "C" {

int slsmi_enable_training(slsmi_ctx* c) {
    if (!c) return SLSMI_ERR_INVALID;
    if (!c->train) {
        c->train = new (std::nothrow) TrainState();
        if (!c->train) return fail(c, SLSMI_ERR_HIP, "out of host memory");
        c->finalized = false;  // the transposed weight copies are made by the next slsmi_finalize
    }
    return SLSMI_OK;
}

int slsmi_train_set_layer_grads_event(slsmi_ctx* c, void* event) {
    if (!c) return SLSMI_ERR_INVALID;
    if (!c->train) return fail(c, SLSMI_ERR_INVALID, "slsmi_enable_training has not been called");
    c->train->layer_grads_event = reinterpret_cast<hipEvent_t>(event);
    return SLSMI_OK;
}

int slsmi_train_set_masks(slsmi_ctx* c, const slsmi_train_masks* masks) {
    if (!c) return SLSMI_ERR_INVALID;
    if (!c->train) return fail(c, SLSMI_ERR_INVALID, "slsmi_enable_training has not been called on this context");
    TrainState* t = c->train;
    t->masks = slsmi_train_masks{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    t->layer_keep.clear();
    if (!masks) return SLSMI_OK;
    for (const float* q : {masks->input, masks->encoder, masks->attn_out, masks->ffn_act, masks->ffn_out})
        if (q && (reinterpret_cast<uintptr_t>(q) & 3)) return fail(c, SLSMI_ERR_INVALID, "misaligned mask");
    t->masks = *masks;
    if (masks->layer_keep) t->layer_keep.assign(masks->layer_keep, masks->layer_keep + c->cfg.layers);
    t->masks.layer_keep = nullptr;  // the copy is what counts
    return SLSMI_OK;
}

int slsmi_train_set_attention_masks(slsmi_ctx* c, const uint8_t* keep_qk, const uint8_t* keep_kq, int64_t layer_stride,
                                    int pitch, float scale) {
    if (!c) return SLSMI_ERR_INVALID;
    if (!c->train) return fail(c, SLSMI_ERR_INVALID, "slsmi_enable_training has not been called on this context");
    TrainState* t = c->train;
    t->attn_keep_qk = t->attn_keep_kq = nullptr;
    if (!keep_qk && !keep_kq) return SLSMI_OK;
    if (!keep_qk || !keep_kq || pitch < 32 || pitch % 32 != 0 || layer_stride < pitch || !(scale >= 1.f) ||
        ((reinterpret_cast<uintptr_t>(keep_qk) | reinterpret_cast<uintptr_t>(keep_kq) | (uintptr_t)layer_stride) & 3))
        return fail(c, SLSMI_ERR_INVALID, "attention masks: both orientations, a row pitch that is a multiple of 32, 4-byte alignment");
    t->attn_keep_qk = keep_qk;
    t->attn_keep_kq = keep_kq;
    t->attn_layer_stride = layer_stride;
    t->attn_pitch = pitch;
    t->attn_scale = scale;
    return SLSMI_OK;
}

int slsmi_set_grad(slsmi_ctx* c, const char* name, float* dst) {
    if (!c || !name) return SLSMI_ERR_INVALID;
    if (!c->train) return fail(c, SLSMI_ERR_INVALID, "slsmi_enable_training has not been called on this context");
    if (dst && (reinterpret_cast<uintptr_t>(dst) & 15)) return fail(c, SLSMI_ERR_INVALID, "gradient buffers must be 16-byte aligned");
    if (dst)
        c->train->grads[name] = dst;
    else
        c->train->grads.erase(name);
    return SLSMI_OK;
}

int slsmi_train_set_fused_adam_hyper(slsmi_ctx* c, double lr, double beta1, double beta2, double eps, double weight_decay) {
    if (!c) return SLSMI_ERR_INVALID;
    if (!c->train) return fail(c, SLSMI_ERR_INVALID, "slsmi_enable_training has not been called on this context");
    if (!(lr >= 0) || !(eps >= 0) || !(weight_decay >= 0) || !(beta1 >= 0 && beta1 < 1) || !(beta2 >= 0 && beta2 < 1))
        return fail(c, SLSMI_ERR_INVALID, "invalid Adam hyper-parameters");
    TrainState* t = c->train;
    t->adam_lr = lr;
    t->adam_b1 = beta1;
    t->adam_b2 = beta2;
    t->adam_eps = eps;
    t->adam_wd = weight_decay;
    t->adam_hyper_set = true;
    return SLSMI_OK;
}

int slsmi_train_set_fused_adam(slsmi_ctx* c, const char* name, float* param, float* exp_avg, float* exp_avg_sq, int64_t step) {
    if (!c || !name) return SLSMI_ERR_INVALID;
    if (!c->train) return fail(c, SLSMI_ERR_INVALID, "slsmi_enable_training has not been called on this context");
    TrainState* t = c->train;
    if (!param) {
        t->fused.erase(name);
        return SLSMI_OK;
    }
    // only the tensors whose gradient is a product of the layer-batched weight-gradient launches
    static const char* const kinds[] = {"self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight",
                                        "self_attn.out_proj.weight", "fc1.weight", "fc2.weight"};
    const std::string nm(name), pre = "ssl_model.model.encoder.layers.";
    bool ok = false;
    if (nm.compare(0, pre.size(), pre) == 0) {
        size_t dot = nm.find('.', pre.size());
        if (dot != std::string::npos && dot > pre.size()) {
            bool digits = true;
            for (size_t i = pre.size(); i < dot; ++i) digits = digits && nm[i] >= '0' && nm[i] <= '9';
            const int layer = digits && dot - pre.size() < 6 ? atoi(nm.substr(pre.size(), dot - pre.size()).c_str()) : -1;
            const std::string rest = nm.substr(dot + 1);
            for (const char* k : kinds) ok = ok || rest == k;
            ok = ok && layer >= 0 && layer < c->cfg.layers;
        }
    }
    if (!ok) return fail(c, SLSMI_ERR_INVALID, std::string("the fused step exists for the encoder layers' projection and feed-forward weights, not for ") + name);
    if (!exp_avg || !exp_avg_sq || step < 1 ||
        ((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(exp_avg) | reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15))
        return fail(c, SLSMI_ERR_INVALID, "fused Adam: parameter and both moments (16-byte aligned), step count >= 1");
    t->fused[nm] = TrainState::Fused{param, exp_avg, exp_avg_sq, (long long)step};
    return SLSMI_OK;
}

int slsmi_train_drop_fused_copies(slsmi_ctx* c) {
    if (!c) return SLSMI_ERR_INVALID;
    if (c->train) train_forget_fresh_copies(c);
    return SLSMI_OK;
}

size_t slsmi_train_workspace_bytes(const slsmi_ctx* c, int batch, int length) {
    if (!c) return 0;
    TPlan p;
    if (!make_tplan(c, batch, length, &p)) return 0;
    return p.total;
}

}  // extern "C"

// What the SLS entry points add to a backbone step.
struct SlsStep {
    const float* fc1_w = nullptr;   // the caller's fp32 fc1.weight [1024][P] (the context holds only its 16-bit split form)
    float* logp_out = nullptr;      // forward
    float* bn_batch_out = nullptr;  // forward, may be null
    const float* dlogp = nullptr;   // backward
};

static int sls_kept_layers(slsmi_ctx* c, const std::vector<uint8_t>& keep, SlsKept* k) {
    const int NL = c->cfg.layers;
    if (NL > 64) return fail(c, SLSMI_ERR_INVALID, "the SLS head in training form reads at most 64 layers");
    k->n = 0;
    for (int l = 0; l < NL; ++l)
        if (keep.empty() || keep[l]) k->idx[k->n++] = l;
    if (k->n == 0) return fail(c, SLSMI_ERR_INVALID, "layerdrop skipped every layer: the SLS head has nothing to read");
    return SLSMI_OK;
}

constexpr float SLS_BN_EPS = 1e-5f;  // nn.BatchNorm2d's default

// layers_out != null: backbone only -- the step ends at the encoder layers, whose outputs are copied to
// layers_out [layers][B * T][1024]
static int train_forward_impl(slsmi_ctx* c, const float* wav, int B, int L, void* workspace, size_t workspace_bytes,
                              const float* dropout_mask, float* logp_out, float* sae_loss_out, float* layers_out,
                              void* stream, const SlsStep* sls = nullptr) {
    const bool backbone_only = layers_out != nullptr || sls != nullptr;
    int rc = check_train(c, backbone_only);
    if (rc) return rc;
    if (!wav || !workspace || (!logp_out && !backbone_only)) return fail(c, SLSMI_ERR_INVALID, "null argument");
    const slsmi_config& cfg = c->cfg;
    TPlan p;
    if (!make_tplan(c, B, L, &p))
        return fail(c, SLSMI_ERR_INVALID, "unsupported batch/length for training (batch <= 64, frames in [3, 256])");
    if (workspace_bytes < p.total || (reinterpret_cast<uintptr_t>(workspace) & (ALIGN - 1)))
        return fail(c, SLSMI_ERR_WORKSPACE, "workspace too small or not 256-byte aligned");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    char* ws = reinterpret_cast<char*>(workspace);
    auto B16 = [&](size_t off) { return reinterpret_cast<bf16_t*>(ws + off); };
    auto F32 = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    const int n = cfg.n_conv, fmt = cfg.operand_format, T = p.frames, D = cfg.sae_dict, Fd = cfg.ffn_dim;
    const float eps = cfg.ln_eps;
    const long M = p.M;

    // ---- feature extractor: every block's conv output y_i and activation a_i kept (padded row layout)
    HIP_TRY(c, launch_conv0_ln_gelu(wav, B, L, p.T[0], c->conv[0].w0, c->conv[0].bias, c->conv[0].ln_g, c->conv[0].ln_b,
                                    eps, B16(p.a[0]), p.P[0], fmt, s));
    for (int i = 1; i < n; ++i) {
        GemmArgs g;
        g.fmt = fmt;
        g.split_ws = c->split_ws; g.split_ws_bytes = GEMM_SPLIT_WS_BYTES;
        g.A = B16(p.a[i - 1]); g.lda = (long)cfg.conv_stride[i] * 512; g.W = c->conv[i].w;
        g.ldw = (long)cfg.conv_kernel[i] * 512; g.bias = c->conv[i].bias; g.C = B16(p.y[i]); g.ldc = 512;
        g.M = B * p.P[i]; g.N = 512; g.K = cfg.conv_kernel[i] * 512; g.mode = GEMM_BF16;
        HIP_TRY(c, launch_gemm(g, s));
        HIP_TRY(c, launch_ln512_bf16(B16(p.y[i]), B16(p.a[i]), (long)B * p.P[i], c->conv[i].ln_g, c->conv[i].ln_b, eps, 1, fmt, s));
    }
    const long rows_last = (long)B * p.P[n - 1];
    HIP_TRY(c, launch_ln512_bf16(B16(p.a[n - 1]), B16(p.featln), rows_last, c->feat_ln_g, c->feat_ln_b, eps, 0, fmt, s));
    {   // post_extract_proj over the padded rows; the epilogue keeps the T frames of every utterance
        GemmArgs g;
        g.fmt = fmt;
        g.split_ws = c->split_ws; g.split_ws_bytes = GEMM_SPLIT_WS_BYTES;
        g.A = B16(p.featln); g.lda = 512; g.W = c->proj_w; g.ldw = 512; g.bias = c->proj_b;
        g.C = F32(p.x0); g.ldc = 1024; g.M = (int)rows_last; g.N = 1024; g.K = 512; g.mode = GEMM_F32;
        g.grp_in = p.P[n - 1]; g.grp_valid = T; g.grp_out = T;
        HIP_TRY(c, launch_gemm(g, s));
    }
    const slsmi_train_masks& mk = c->train->masks;
    const std::vector<uint8_t>& keep = c->train->layer_keep;
    if (mk.input) HIP_TRY(c, launch_mul_f32(F32(p.x0), F32(p.x0), mk.input, nullptr, M * 1024, s));  // dropout_input
    // ---- u = pos_conv(x0); x1 = x0 + gelu(u)
    HIP_TRY(c, launch_posconv_pack(F32(p.x0), B16(p.xg), B, T, p.slab, fmt, s));
    {
        GemmArgs g;
        g.fmt = fmt;
        g.split_ws = c->split_ws; g.split_ws_bytes = GEMM_SPLIT_WS_BYTES;
        g.A = B16(p.xg); g.lda = 64; g.W = c->pos_w; g.ldw = 8192; g.bias = c->pos_b;
        g.C = F32(p.u); g.ldc = 1024; g.M = T; g.N = 64; g.K = 8192; g.mode = GEMM_F32; g.bn = 64;
        g.nz = B * 16; g.zA = (long)p.slab * 64; g.zWmod = 16; g.zW = 64L * 8192; g.zBias = 64;
        g.zCdiv = 16; g.zC1 = (long)T * 1024; g.zC2 = 64;
        HIP_TRY(c, launch_gemm(g, s));
    }
    HIP_TRY(c, launch_add_gelu(F32(p.x0), F32(p.u), F32(p.xs), M * 1024, s));
    if (mk.encoder) HIP_TRY(c, launch_mul_f32(F32(p.xs), F32(p.xs), mk.encoder, nullptr, M * 1024, s));  // wav2vec2.py:933
    // ---- encoder layers
    float* const tmp32 = F32(p.T1);  // a backward scratch, free during the forward: a GEMM's output on its way through a mask
    bool h1_ready = false;
    for (int l = 0; l < cfg.layers; ++l) {
        const EncLayer& E = c->enc[l];
        float* xin = F32(p.xs + (size_t)l * p.s_row32);
        float* xout = F32(p.xs + (size_t)(l + 1) * p.s_row32);
        if (!keep.empty() && !keep[l]) {  // layerdrop: the layer is not run
            HIP_TRY(c, hipMemcpyAsync(xout, xin, (size_t)M * 1024 * 4, hipMemcpyDeviceToDevice, s));
            continue;
        }
        const float* m1 = mk.attn_out ? mk.attn_out + (size_t)l * M * 1024 : nullptr;
        const float* m2 = mk.ffn_act ? mk.ffn_act + (size_t)l * M * Fd : nullptr;
        const float* m3 = mk.ffn_out ? mk.ffn_out + (size_t)l * M * 1024 : nullptr;
        float* xmid = F32(p.xmid + (size_t)l * p.s_row32);
        bf16_t* h1 = B16(p.h1 + (size_t)l * p.s_row16);
        bf16_t* qkv = B16(p.qkv + (size_t)l * p.s_qkv);
        bf16_t* ao = B16(p.ao + (size_t)l * p.s_row16);
        bf16_t* h2 = B16(p.h2 + (size_t)l * p.s_row16);
        bf16_t* fpre = B16(p.fpre + (size_t)l * p.s_ffn);
        bf16_t* f = B16(p.f + (size_t)l * p.s_ffn);
        // (h1 may be there already: the previous layer's last GEMM left its K slices to a kernel that summed them and
        // normalised the sum in one pass)
        if (!h1_ready) HIP_TRY(c, launch_ln1024_f32_bf16(xin, h1, M, E.ln1_g, E.ln1_b, eps, fmt, s));
        h1_ready = false;
        GemmArgs g;
        g.fmt = fmt;
        g.split_ws = c->split_ws; g.split_ws_bytes = GEMM_SPLIT_WS_BYTES;
        g.A = h1; g.lda = 1024; g.W = E.wqkv; g.ldw = 1024; g.bias = E.bqkv;
        g.C = qkv; g.ldc = 3072; g.M = (int)M; g.N = 3072; g.K = 1024; g.mode = GEMM_BF16;
        HIP_TRY(c, launch_gemm(g, s));
        if (c->train->attn_keep_qk) {
            if (c->train->attn_pitch < T || c->train->attn_layer_stride < (int64_t)B * 16 * T * c->train->attn_pitch)
                return fail(c, SLSMI_ERR_INVALID, "attention masks do not cover this batch");
            HIP_TRY(c, launch_attention_dropout(qkv, ao, B, T, fmt, c->train->attn_keep_qk + l * c->train->attn_layer_stride,
                                                c->train->attn_pitch, c->train->attn_scale, s));
        } else {
            HIP_TRY(c, launch_attention(qkv, ao, B, T, fmt, s));
        }
        g = GemmArgs();
        g.fmt = fmt;
        g.split_ws = c->split_ws; g.split_ws_bytes = GEMM_SPLIT_WS_BYTES;
        g.A = ao; g.lda = 1024; g.W = E.wo; g.ldw = 1024; g.bias = E.bo;
        g.C = xmid; g.resid = xin; g.ldc = 1024; g.M = (int)M; g.N = 1024; g.K = 1024; g.mode = GEMM_F32_RESID;
        if (m1) { g.C = tmp32; g.resid = nullptr; g.mode = GEMM_F32; }
        // Where the launcher cuts K (few rows), the slices are summed by the LayerNorm kernel that follows: one pass
        // instead of a summing kernel and a LayerNorm
        int cut = 1;
        if (!m1) g.deferred_slices = &cut;
        HIP_TRY(c, launch_gemm(g, s));
        if (m1) HIP_TRY(c, launch_mul_f32(xmid, tmp32, m1, xin, M * 1024, s));  // x_in + dropout1(...)
        if (cut > 1)
            HIP_TRY(c, launch_ln1024_sum_f32_bf16(c->split_ws, cut, M * 1024, xin, xmid, h2, M, E.ln2_g, E.ln2_b, eps, fmt, s));
        else
            HIP_TRY(c, launch_ln1024_f32_bf16(xmid, h2, M, E.ln2_g, E.ln2_b, eps, fmt, s));
        g = GemmArgs();
        g.fmt = fmt;
        g.split_ws = c->split_ws; g.split_ws_bytes = GEMM_SPLIT_WS_BYTES;
        g.A = h2; g.lda = 1024; g.W = E.w1; g.ldw = 1024; g.bias = E.b1;
        if (M < 1024) {  // the 128-row kernel's shapes: f = gelu(.) and the value before it from one epilogue
            g.C = f; g.pre_out = fpre; g.ldc = Fd; g.M = (int)M; g.N = Fd; g.K = 1024; g.mode = GEMM_BF16_GELU;
            HIP_TRY(c, launch_gemm(g, s));
        } else {
            g.C = fpre; g.ldc = Fd; g.M = (int)M; g.N = Fd; g.K = 1024; g.mode = GEMM_BF16;
            HIP_TRY(c, launch_gemm(g, s));
            HIP_TRY(c, launch_gelu16(fpre, f, M * Fd, fmt, s));
        }
        if (m2) HIP_TRY(c, launch_mul16_f32(f, m2, M * Fd, fmt, s));  // dropout2
        g = GemmArgs();
        g.fmt = fmt;
        g.split_ws = c->split_ws; g.split_ws_bytes = GEMM_SPLIT_WS_BYTES;
        g.A = f; g.lda = Fd; g.W = E.w2; g.ldw = Fd; g.bias = E.b2;
        g.C = xout; g.resid = xmid; g.ldc = 1024; g.M = (int)M; g.N = 1024; g.K = Fd; g.mode = GEMM_F32_RESID;
        if (m3) { g.C = tmp32; g.resid = nullptr; g.mode = GEMM_F32; }
        // the same with the next layer's first LayerNorm, when that layer runs
        const bool next_runs = l + 1 < cfg.layers && (keep.empty() || keep[l + 1]);
        cut = 1;
        if (!m3 && next_runs) g.deferred_slices = &cut;
        HIP_TRY(c, launch_gemm(g, s));
        if (m3) HIP_TRY(c, launch_mul_f32(xout, tmp32, m3, xmid, M * 1024, s));  // x_mid + dropout3(...)
        if (cut > 1) {
            const EncLayer& En = c->enc[l + 1];
            HIP_TRY(c, launch_ln1024_sum_f32_bf16(c->split_ws, cut, M * 1024, xmid, xout, B16(p.h1 + (size_t)(l + 1) * p.s_row16), M,
                                                  En.ln1_g, En.ln1_b, eps, fmt, s));
            h1_ready = true;
        }
    }
    TrainState* t = c->train;
    t->fwd_ws = workspace;
    t->fwd_wav = wav;
    t->fwd_B = B;
    t->fwd_L = L;
    t->fwd_mask = dropout_mask != nullptr;
    t->fwd_backbone_only = backbone_only;
    t->fwd_sls = sls != nullptr;
    if (sls) {   // ---- the upstream SLS head on the layer outputs where they lie
        SlsKept kept;
        if ((rc = sls_kept_layers(c, keep, &kept))) return rc;
        const long P = p.sls_P;
        const float* xs1 = F32(p.xs + p.s_row32);
        const long ls = (long)(p.s_row32 / 4);
        HIP_TRY(c, launch_sls_layer_means(xs1, ls, kept, B, T, F32(p.sy0), s));
        HIP_TRY(c, launch_sls_layer_weights(F32(p.sy0), c->fc0_w, c->fc0_b, B * kept.n, F32(p.swl), s));
        HIP_TRY(c, launch_sls_weighted_sum(xs1, ls, kept, F32(p.swl), B, T, F32(p.sfull), F32(p.spart), s));
        HIP_TRY(c, launch_sls_bn_stats(F32(p.spart), M, SLS_BN_EPS, F32(p.sstat), sls->bn_batch_out, s));
        HIP_TRY(c, launch_sls_bn_pool(F32(p.sfull), F32(p.sstat), c->bn, B, T, F32(p.spool), reinterpret_cast<uint8_t*>(ws + p.samax), s));
        HIP_TRY(c, launch_sls_fc1_fwd(F32(p.spool), sls->fc1_w, c->sls_fc1_b, B, P, F32(p.sh1), s));
        HIP_TRY(c, launch_sls_tail_fwd(F32(p.sh1), c->fc3_w, c->fc3_b, B, F32(p.sa3), F32(p.slogp), s));
        HIP_TRY(c, hipMemcpyAsync(sls->logp_out, F32(p.slogp), (size_t)B * 2 * 4, hipMemcpyDeviceToDevice, s));
        return SLSMI_OK;
    }
    if (backbone_only) {
        for (int l = 0; l < cfg.layers; ++l)
            HIP_TRY(c, hipMemcpyAsync(layers_out + (size_t)l * M * 1024, F32(p.xs + (size_t)(l + 1) * p.s_row32),
                                      (size_t)M * 1024 * 4, hipMemcpyDeviceToDevice, s));
        return SLSMI_OK;
    }
    // ---- final LayerNorm, the auto-encoder in the form the model was built with, classifier (model.py:212-247;
    // window TopK: model_window_topk.py:106-110)
    const int feat = cfg.sae_features;
    const bool has_sae = feat != SLSMI_SAE_OFF, use_recon = feat == SLSMI_SAE_RECON;
    const int Dc = feat == SLSMI_SAE_SPARSE ? D : 1024;   // what the classifier pools: the code, or 1024 features
    if (sae_loss_out && !has_sae) return fail(c, SLSMI_ERR_INVALID, "sae_loss_out needs a model with the auto-encoder");
    float* xlast = F32(p.xs + (size_t)cfg.layers * p.s_row32);
    HIP_TRY(c, launch_ln1024_f32_f32(xlast, F32(p.xf), has_sae ? B16(p.hsae) : nullptr, has_sae ? c->sae_b_dec : nullptr, M,
                                     c->final_ln_g, c->final_ln_b, eps, fmt, s));
    if (has_sae) {
        GemmArgs g;
        g.fmt = fmt;
        g.split_ws = c->split_ws; g.split_ws_bytes = GEMM_SPLIT_WS_BYTES;
        g.A = B16(p.hsae); g.lda = 1024; g.W = c->sae_enc_w; g.ldw = 1024; g.bias = c->sae_enc_b;
        g.C = F32(p.code); g.ldc = D; g.M = (int)M; g.N = D; g.K = 1024; g.mode = GEMM_F32_RELU;
        HIP_TRY(c, launch_gemm(g, s));
        if (cfg.sae_window > 1)
            HIP_TRY(c, launch_window_topk(F32(p.code), B16(p.code16), reinterpret_cast<uint8_t*>(ws + p.wkeep), B, T, D, cfg.sae_k,
                                          cfg.sae_window, fmt, s));
        else
            HIP_TRY(c, launch_topk_rows(F32(p.code), B16(p.code16), M, D, cfg.sae_k, fmt, s));
        g = GemmArgs();
        g.fmt = fmt;
        g.split_ws = c->split_ws; g.split_ws_bytes = GEMM_SPLIT_WS_BYTES;
        g.A = B16(p.code16); g.lda = D; g.W = c->sae_dec_w; g.ldw = D; g.bias = nullptr;
        g.C = F32(p.recon); g.ldc = 1024; g.M = (int)M; g.N = 1024; g.K = D; g.mode = GEMM_F32;
        HIP_TRY(c, launch_gemm(g, s));
        if (sae_loss_out) HIP_TRY(c, launch_mse(F32(p.recon), c->sae_b_dec, F32(p.xf), M, F32(p.msepart), sae_loss_out, s));
    }
    if (!has_sae)
        HIP_TRY(c, launch_col_mean(F32(p.xf), nullptr, F32(p.pooled), B, T, 1024, s));
    else if (use_recon)
        HIP_TRY(c, launch_col_mean(F32(p.recon), c->sae_b_dec, F32(p.pooled), B, T, 1024, s));
    else
        HIP_TRY(c, launch_col_mean(F32(p.code), nullptr, F32(p.pooled), B, T, D, s));
    if (dropout_mask)
        HIP_TRY(c, hipMemcpyAsync(F32(p.mask), dropout_mask, (size_t)B * 256 * 4, hipMemcpyDeviceToDevice, s));
    HIP_TRY(c, launch_sae_tail(F32(p.pooled), c->cls_ln_g, c->cls_ln_b, c->cls_w1, c->cls_b1, c->cls_w2, c->cls_b2,
                               F32(p.hidden), F32(p.logp), B, Dc, s, dropout_mask ? F32(p.mask) : nullptr));
    HIP_TRY(c, hipMemcpyAsync(logp_out, F32(p.logp), (size_t)B * 2 * 4, hipMemcpyDeviceToDevice, s));
    return SLSMI_OK;
}

// dlayers != null: backbone only -- the loss gradient arrives at the encoder layers' outputs
// (dlayers [layers][B * T][1024]) instead of at the model's two outputs
static int train_backward_impl(slsmi_ctx* c, const float* wav, int B, int L, void* workspace, size_t workspace_bytes,
                               const float* dlogp, const float* dsae_loss, int used_dropout_mask, const float* dlayers,
                               void* stream, const SlsStep* sls = nullptr) {
    const bool backbone_only = dlayers != nullptr || sls != nullptr;
    int rc = check_train(c, backbone_only);
    if (rc) return rc;
    if (!wav || !workspace || (!dlogp && !backbone_only)) return fail(c, SLSMI_ERR_INVALID, "null argument");
    const slsmi_config& cfg = c->cfg;
    TPlan p;
    if (!make_tplan(c, B, L, &p)) return fail(c, SLSMI_ERR_INVALID, "unsupported batch/length for training");
    if (workspace_bytes < p.total || (reinterpret_cast<uintptr_t>(workspace) & (ALIGN - 1)))
        return fail(c, SLSMI_ERR_WORKSPACE, "workspace too small or not 256-byte aligned");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    char* ws = reinterpret_cast<char*>(workspace);
    auto B16 = [&](size_t off) { return reinterpret_cast<bf16_t*>(ws + off); };
    auto F32 = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    TrainState* t = c->train;
    if (t->fwd_ws != workspace || t->fwd_B != B || t->fwd_L != L || t->fwd_wav != wav ||
        t->fwd_mask != (used_dropout_mask != 0) || t->fwd_backbone_only != backbone_only || t->fwd_sls != (sls != nullptr))
        return fail(c, SLSMI_ERR_INVALID, "slsmi_train_backward must follow slsmi_train_forward on the same input, "
                                          "batch, length, workspace and mask use");
    if (!t->fused.empty() && !t->adam_hyper_set)
        return fail(c, SLSMI_ERR_INVALID, "slsmi_train_set_fused_adam without slsmi_train_set_fused_adam_hyper");
    t->fwd_ws = nullptr;  // one backward per forward: the scratch of the backward overwrites what it read
    auto G = [&](const std::string& name) -> float* {
        auto it = t->grads.find(name);
        return it == t->grads.end() ? nullptr : it->second;
    };
    const int n = cfg.n_conv, fmt = cfg.operand_format, T = p.frames, D = cfg.sae_dict, Fd = cfg.ffn_dim;
    const float eps = cfg.ln_eps;
    const long M = p.M;
    const std::string bb = "ssl_model.model.";
    float* colpart = F32(p.colpart);
    float* lnpart = F32(p.lnpart);
    // weight gradient dW[N][K] = dY^T X for dY bfloat16 [M][N] and X 16-bit [M][K] of format xfmt, both row-major as
    // they lie (no transposed copies)
    auto weight_grad = [&](const bf16_t* dY, long ldy, int N, const bf16_t* X, long ldx, int K, int xfmt, float* dst) -> int {
        if (!dst) return SLSMI_OK;
        HIP_TRY(c, launch_gemm_tn(dY, ldy, 0, X, ldx, 0, xfmt, dst, K, 0, M, N, K, 1, 1.0f, s));
        return SLSMI_OK;
    };
    // The encoder layers' weight and bias gradients run after the layer loop, one launch per kind of matrix over all
    // layers (24 x 64 tiles fill the device; a layer's own 64 tiles would not, and would have to cut K instead).
    // The loop leaves each layer's output gradients behind, row-major as the input-gradient GEMMs read them; the
    // weight-gradient kernel contracts them with the saved activations over the rows, no transposed copies.
    const int NLy = cfg.layers;
    float* const lnparts = F32(p.lnparts);
    const size_t lnpart_one = (size_t)ln_bwd_groups(M) * 2 * 1024;
#define TRY(expr)            \
    do {                     \
        int _r = (expr);     \
        if (_r) return _r;   \
    } while (0)

    const slsmi_train_masks mk = t->masks;
    const std::vector<uint8_t> keep = t->layer_keep;
    t->masks = slsmi_train_masks{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // one forward and its backward
    // A LayerNorm backward also leaves the bfloat16 copy of its result where the next GEMM reads it (that layer's kept
    // output gradient), when nothing stands between the two: no dropout mask on that edge, no layer skipped.
    const bool fuse16 = keep.empty();
    bool d2_ready = false;   // the fc2 output gradient of the layer about to run is already in its buffer
    float* gx = F32(p.G1);
    float* galt = F32(p.G0);
    bf16_t* big16 = B16(p.big16);
    bf16_t* g16 = B16(p.g16);
    if (sls) {   // ---- the SLS head's backward: its parameters' gradients and the loss gradient at every layer output
        SlsKept kept;
        if ((rc = sls_kept_layers(c, keep, &kept))) return rc;
        const long P = p.sls_P;
        const float* xs1 = F32(p.xs + p.s_row32);
        const long ls = (long)(p.s_row32 / 4);
        float* dbn = F32(p.sdbn);
        float* stat = F32(p.sstat);
        HIP_TRY(c, launch_sls_tail_bwd(sls->dlogp, F32(p.slogp), F32(p.sa3), F32(p.sh1), c->fc3_w, B, F32(p.sdh1), G("fc3.weight"),
                                       G("fc3.bias"), G("fc1.bias"), s));
        HIP_TRY(c, launch_sls_fc1_bwd(F32(p.sdh1), F32(p.spool), sls->fc1_w, B, P, G("fc1.weight"), F32(p.sdpart), s));
        HIP_TRY(c, hipMemsetAsync(dbn, 0, (size_t)M * 1024 * 4, s));
        HIP_TRY(c, launch_sls_pool_bwd(F32(p.sdpart), reinterpret_cast<const uint8_t*>(ws + p.samax), F32(p.sfull), stat, c->bn, B, T,
                                       dbn, F32(p.spart2), s));
        HIP_TRY(c, launch_sls_bn_bwd_stats(F32(p.spart2), B * (T / 3), M * 1024, stat + 4, G("first_bn.weight"), G("first_bn.bias"), s));
        HIP_TRY(c, launch_sls_dfull(dbn, F32(p.sfull), stat, stat + 4, c->bn, M, s));
        HIP_TRY(c, launch_sls_dweights(xs1, ls, kept, dbn, B, T, F32(p.sdwl), s));
        HIP_TRY(c, launch_sls_fc0_bwd(F32(p.sdwl), sls_dweight_chunks(T), F32(p.swl), F32(p.sy0), B * kept.n, F32(p.sdz),
                                      G("fc0.weight"), G("fc0.bias"), s));
        if (kept.n < cfg.layers) HIP_TRY(c, hipMemsetAsync(ws + p.sdlay, 0, (size_t)cfg.layers * M * 1024 * 4, s));
        HIP_TRY(c, launch_sls_dlayers(dbn, F32(p.swl), F32(p.sdz), c->fc0_w, kept, B, T, F32(p.sdlay), M * 1024, s));
        dlayers = F32(p.sdlay);
    }
    if (backbone_only) {
        // the final LayerNorm is not on the path from the layers' outputs to the loss
        if (float* d = G(bb + "encoder.layer_norm.weight")) HIP_TRY(c, hipMemsetAsync(d, 0, 1024 * 4, s));
        if (float* d = G(bb + "encoder.layer_norm.bias")) HIP_TRY(c, hipMemsetAsync(d, 0, 1024 * 4, s));
        HIP_TRY(c, hipMemcpyAsync(gx, dlayers + (size_t)(cfg.layers - 1) * M * 1024, (size_t)M * 1024 * 4,
                                  hipMemcpyDeviceToDevice, s));
    } else {
    // ---- classifier
    const int feat = cfg.sae_features;
    const bool has_sae = feat != SLSMI_SAE_OFF, use_recon = feat == SLSMI_SAE_RECON;
    const int Dc = feat == SLSMI_SAE_SPARSE ? D : 1024;
    if (dsae_loss && !has_sae) return fail(c, SLSMI_ERR_INVALID, "dsae_loss needs a model with the auto-encoder");
    HIP_TRY(c, launch_cls_bwd(dlogp, F32(p.logp), F32(p.pooled), F32(p.hidden), used_dropout_mask ? F32(p.mask) : nullptr,
                              c->cls_ln_g, c->cls_ln_b, c->cls_w1, c->cls_w2, B, Dc, F32(p.stats), F32(p.dh1), F32(p.dhb),
                              F32(p.dpool), G("classifier.0.weight"), G("classifier.0.bias"), G("classifier.1.weight"),
                              G("classifier.1.bias"), G("classifier.4.weight"), G("classifier.4.bias"), s));
    // ---- auto-encoder.  dxf collects the gradient at the final LayerNorm's output in G0.
    float* dxf = F32(p.G0);
    if (!has_sae) {   // use_sae=False: the classifier pools the encoder output itself (model.py:242-247)
        HIP_TRY(c, launch_spread_rows(F32(p.dpool), M, T, dxf, s));
    } else {
    const float* dcode = nullptr;
    if (use_recon) {
        // the classifier pools decoder(code) + b_dec (model.py:228-233): its gradient and the loss's meet at the reconstruction
        HIP_TRY(c, launch_recon_grad(F32(p.recon), c->sae_b_dec, F32(p.xf), dsae_loss, F32(p.dpool), M, T, g16, dxf, F32(p.T1), s));
        if (float* d = G("sae.b_dec")) HIP_TRY(c, launch_colsum(F32(p.T1), -1, 1024, M, 1024, 1.0f, 0, colpart, d, s));
    } else if (dsae_loss) {
        HIP_TRY(c, launch_mse_grad(F32(p.recon), c->sae_b_dec, F32(p.xf), dsae_loss, M, g16, dxf, s));
        if (float* d = G("sae.b_dec")) HIP_TRY(c, launch_colsum(dxf, -1, 1024, M, 1024, -1.0f, 0, colpart, d, s));
    }
    if (use_recon || dsae_loss) {
        // decoder: recon = code W_dec^T
        TRY(weight_grad(g16, 1024, 1024, B16(p.code16), D, D, fmt, G("sae.decoder.weight")));
        HIP_TRY(c, gemm_bf(c, g16, 1024, t->sae_decT, 1024, F32(p.dcode), D, M, D, 1024, GEMM_F32, nullptr, s));
        dcode = F32(p.dcode);
    } else {
        HIP_TRY(c, hipMemsetAsync(dxf, 0, (size_t)M * 1024 * 4, s));
        if (float* d = G("sae.b_dec")) HIP_TRY(c, hipMemsetAsync(d, 0, 1024 * 4, s));
        if (float* d = G("sae.decoder.weight")) HIP_TRY(c, hipMemsetAsync(d, 0, (size_t)1024 * D * 4, s));
    }
    HIP_TRY(c, launch_sae_code_grad(F32(p.code), use_recon ? nullptr : F32(p.dpool), dcode, big16, M, T, D, s));
    if (float* d = G("sae.encoder.bias")) HIP_TRY(c, launch_colsum(big16, FB, D, M, D, 1.0f, 0, colpart, d, s));
    TRY(weight_grad(big16, D, D, B16(p.hsae), 1024, 1024, fmt, G("sae.encoder.weight")));
    HIP_TRY(c, gemm_bf(c, big16, D, t->sae_encT, D, F32(p.T1), 1024, M, 1024, D, GEMM_F32, nullptr, s));
    if (float* d = G("sae.b_dec")) HIP_TRY(c, launch_colsum(F32(p.T1), -1, 1024, M, 1024, -1.0f, 1, colpart, d, s));
    HIP_TRY(c, launch_add_f32(dxf, F32(p.T1), M * 1024, s));
    }
    // ---- final LayerNorm: G1 = gradient at the last layer's output
    d2_ready = fuse16 && !mk.ffn_out;
    HIP_TRY(c, launch_ln_bwd(1024, F32(p.xs + (size_t)cfg.layers * p.s_row32), -1, dxf, c->final_ln_g, nullptr, eps, 0,
                             nullptr, gx, 0, lnpart, G(bb + "encoder.layer_norm.weight"), G(bb + "encoder.layer_norm.bias"),
                             M, 0, 0, s, nullptr, d2_ready ? B16(p.rD2 + (size_t)(cfg.layers - 1) * p.s_row16) : nullptr));
    }
    // ---- encoder layers, last to first.  gx: gradient at the layer's output; galt: the other buffer.
    const uint8_t* const attn_qk = t->attn_keep_qk;
    const uint8_t* const attn_kq = t->attn_keep_kq;
    const int64_t attn_stride = t->attn_layer_stride;
    const int attn_pitch = t->attn_pitch;
    const float attn_scale = t->attn_scale;
    t->attn_keep_qk = t->attn_keep_kq = nullptr;
    t->layer_keep.clear();
    for (int l = cfg.layers - 1; l >= 0; --l) {
        const EncLayer& E = c->enc[l];
        const LayerT& W = t->lt[l];
        const std::string lp = bb + "encoder.layers." + std::to_string(l) + ".";
        if (!keep.empty() && !keep[l]) {  // layerdrop skipped this layer: the gradient passes, its parameters get none
            static const struct { const char* name; long n; } ps[] = {
                {"self_attn.q_proj.weight", 1024L * 1024}, {"self_attn.q_proj.bias", 1024}, {"self_attn.k_proj.weight", 1024L * 1024},
                {"self_attn.k_proj.bias", 1024}, {"self_attn.v_proj.weight", 1024L * 1024}, {"self_attn.v_proj.bias", 1024},
                {"self_attn.out_proj.weight", 1024L * 1024}, {"self_attn.out_proj.bias", 1024},
                {"self_attn_layer_norm.weight", 1024}, {"self_attn_layer_norm.bias", 1024}, {"fc1.weight", -1024}, {"fc1.bias", -1},
                {"fc2.weight", -1024}, {"fc2.bias", 1024}, {"final_layer_norm.weight", 1024}, {"final_layer_norm.bias", 1024}};
            for (const auto& q : ps)
                if (float* d = G(lp + q.name))
                    HIP_TRY(c, hipMemsetAsync(d, 0, (size_t)(q.n < 0 ? -q.n * Fd : q.n) * 4, s));
            if (backbone_only && l > 0) HIP_TRY(c, launch_add_f32(gx, dlayers + (size_t)(l - 1) * M * 1024, M * 1024, s));
            continue;
        }
        const float* m1 = mk.attn_out ? mk.attn_out + (size_t)l * M * 1024 : nullptr;
        const float* m2 = mk.ffn_act ? mk.ffn_act + (size_t)l * M * Fd : nullptr;
        const float* m3 = mk.ffn_out ? mk.ffn_out + (size_t)l * M * 1024 : nullptr;
        const float* xin = F32(p.xs + (size_t)l * p.s_row32);
        const float* xmid = F32(p.xmid + (size_t)l * p.s_row32);
        const bf16_t* qkv = B16(p.qkv + (size_t)l * p.s_qkv);
        const bf16_t* ao = B16(p.ao + (size_t)l * p.s_row16);
        const bf16_t* fpre = B16(p.fpre + (size_t)l * p.s_ffn);
        // fc2: x_out = x_mid + dropout3(f W2^T + b2)
        const float* dy2 = gx;  // gradient at the GEMM's output
        if (m3) {
            HIP_TRY(c, launch_mul_f32(F32(p.T1), gx, m3, nullptr, M * 1024, s));
            dy2 = F32(p.T1);
        }
        bf16_t* const d2 = B16(p.rD2 + (size_t)l * p.s_row16);   // this layer's output gradients, kept
        bf16_t* const d1 = B16(p.rD1 + (size_t)l * p.s_ffn);
        bf16_t* const dO = B16(p.rDo + (size_t)l * p.s_row16);
        bf16_t* const dq = B16(p.rDq + (size_t)l * p.s_qkv);
        if (!d2_ready) HIP_TRY(c, launch_cvt_bf16(dy2, d2, M * 1024, s));
        HIP_TRY(c, gemm_bf(c, d2, 1024, W.w2T, 1024, d1, Fd, M, Fd, 1024, GEMM_BF16, nullptr, s));
        if (m2) HIP_TRY(c, launch_mul16_f32(d1, m2, M * Fd, FB, s));  // back through dropout2
        HIP_TRY(c, launch_gelu_bwd16(fpre, fmt, d1, M * Fd, s));
        // fc1: f_pre = h2 W1^T + b1
        // (where the launcher cuts K, the LayerNorm backward behind the GEMM adds the slices as it reads them)
        int cut = 1;
        HIP_TRY(c, gemm_bf(c, d1, Fd, W.w1T, Fd, F32(p.T1), 1024, M, 1024, Fd, GEMM_F32, nullptr, s, &cut));
        const float* dyn = cut > 1 ? c->split_ws : F32(p.T1);
        // final_layer_norm: galt = gx + LN'(x_mid; T1)   (gradient at x_mid)
        // (its parameter gradients: from the partial sums, for all layers together behind the loop)
        const bool dO_ready = fuse16 && !m1;
        HIP_TRY(c, launch_ln_bwd(1024, xmid, -1, dyn, E.ln2_g, nullptr, eps, 0, gx, galt, 0,
                                 lnparts + (2 * l + 1) * lnpart_one, nullptr, nullptr, M, 0, 0, s, nullptr,
                                 dO_ready ? dO : nullptr, cut, M * 1024));
        // out_proj: x_mid = x_in + dropout1(ao Wo^T + bo)
        const float* dy1 = galt;
        if (m1) {
            HIP_TRY(c, launch_mul_f32(F32(p.T1), galt, m1, nullptr, M * 1024, s));
            dy1 = F32(p.T1);
        }
        if (!dO_ready) HIP_TRY(c, launch_cvt_bf16(dy1, dO, M * 1024, s));
        bf16_t* dao = B16(p.du16);  // [M][1024] bfloat16 scratch
        HIP_TRY(c, gemm_bf(c, dO, 1024, W.woT, 1024, dao, 1024, M, 1024, 1024, GEMM_BF16, nullptr, s));
        // attention core -> dqkv [M][3072]
        HIP_TRY(c, launch_attention_bwd(qkv, ao, dao, dq, B, T, fmt, s,
                                        attn_qk ? attn_qk + l * attn_stride : nullptr, attn_kq ? attn_kq + l * attn_stride : nullptr,
                                        attn_pitch, attn_scale));
        // q/k/v projections (q weights and bias are held pre-scaled by 1/8: the gradient at the
        // reference's parameters carries the same factor)
        cut = 1;
        HIP_TRY(c, gemm_bf(c, dq, 3072, W.wqkvT, 3072, F32(p.T1), 1024, M, 1024, 3072, GEMM_F32, nullptr, s, &cut));
        dyn = cut > 1 ? c->split_ws : F32(p.T1);
        // self_attn_layer_norm: gx = galt + LN'(x_in; T1)   (gradient at x_in = output of layer l - 1)
        // (+ the head's own gradient at that output when the head sits on every layer: layer l - 1's output also feeds it)
        d2_ready = fuse16 && l > 0 && !mk.ffn_out;
        HIP_TRY(c, launch_ln_bwd(1024, xin, -1, dyn, E.ln1_g, nullptr, eps, 0, galt, gx, 0,
                                 lnparts + (2 * l) * lnpart_one, nullptr, nullptr, M, 0, 0, s,
                                 backbone_only && l > 0 ? dlayers + (size_t)(l - 1) * M * 1024 : nullptr,
                                 d2_ready ? B16(p.rD2 + (size_t)(l - 1) * p.s_row16) : nullptr, cut, M * 1024));
    }
    {   // ---- the layers' LayerNorm parameter gradients from the partial sums of their backward launches
        std::vector<float*> dgm(2 * NLy, nullptr), dbt(2 * NLy, nullptr);
        for (int l = 0; l < NLy; ++l) {
            if (!keep.empty() && !keep[l]) continue;
            const std::string lp = bb + "encoder.layers." + std::to_string(l) + ".";
            dgm[2 * l] = G(lp + "self_attn_layer_norm.weight");
            dbt[2 * l] = G(lp + "self_attn_layer_norm.bias");
            dgm[2 * l + 1] = G(lp + "final_layer_norm.weight");
            dbt[2 * l + 1] = G(lp + "final_layer_norm.bias");
        }
        HIP_TRY(c, launch_ln_part_sums(lnparts, (long)lnpart_one, ln_bwd_groups(M), 1024, dgm.data(), dbt.data(), 2 * NLy, s));
    }
    // ---- the layers' weight gradients dW_l[N][K] = alpha dY_l^T X_l and bias gradients (column sums of dY_l).  One
    // launch over all layers when the caller's buffers of the kind lie at equal distances and no layer was skipped;
    // one launch per layer with the same tiles otherwise (same bits either way).
    auto layer_grads = [&](int which, const char* wname, const char* bname, const bf16_t* dy, long ldy, long z_dy, int N,
                           const bf16_t* x, long z_x, int K, float alpha) -> int {
        std::vector<float*> d(NLy), db(NLy);
        std::vector<TnAdamEntry> fz;   // layers whose tensor of this kind takes its Adam step in the product's epilogue
        std::vector<std::string> fresh_names;   // ... and gets its 16-bit copies rewritten there
        bool all = true;
        for (int l = 0; l < NLy; ++l) {
            const bool kept = keep.empty() || keep[l];
            const std::string lp = bb + "encoder.layers." + std::to_string(l) + ".";
            d[l] = kept ? G(lp + wname) : nullptr;
            db[l] = kept ? G(lp + bname) : nullptr;
            if (kept && !t->fused.empty()) {
                auto it = t->fused.find(lp + wname);
                if (it != t->fused.end()) {
                    // The step's epilogue also writes the library's 16-bit copies of the tensor (the forward's operand
                    // and its bfloat16 transpose) when the tensor stepped IS the buffer the weight was loaded from: the
                    // next reload then skips it.
                    bf16_t *copy = nullptr, *outT = nullptr;
                    auto src = c->sources.find(lp + wname);
                    if (src != c->sources.end() && src->second.data == it->second.p && t->lt[l].have[which]) {
                        const EncLayer& E = c->enc[l];
                        const LayerT& W = t->lt[l];
                        copy = which <= 2 ? E.wqkv + (size_t)which * 1024 * 1024 : which == 3 ? E.wo : which == 4 ? E.w1 : E.w2;
                        outT = which <= 2 ? W.wqkvT + which * 1024 : which == 3 ? W.woT : which == 4 ? W.w1T : W.w2T;
                        fresh_names.push_back(lp + wname);
                    }
                    fz.push_back(TnAdamEntry{it->second.p, it->second.m, it->second.v, it->second.step, l, copy, outT, nullptr});
                    d[l] = nullptr;   // no gradient is stored for it
                }
            }
            all = all && d[l];
        }
        // The bias gradients (column sums of dY) ride in the weight-gradient launches, which stage those rows anyway;
        // a layer whose weight gets neither a gradient nor a fused step keeps the separate column-sum launch.
        std::vector<float*> db_left = db;
        if (!fz.empty()) {
            for (TnAdamEntry& e : fz) {
                e.db = db[e.zsrc];
                db_left[e.zsrc] = nullptr;
            }
            // (entries with and without copies go in separate launches: the table carries one pitch and scale)
            const long ldo = which <= 2 ? 3072 : which == 4 ? Fd : 1024;
            std::stable_partition(fz.begin(), fz.end(), [](const TnAdamEntry& e) { return e.copy != nullptr; });
            size_t with = 0;
            while (with < fz.size() && fz[with].copy) ++with;
            if (with)
                HIP_TRY(c, launch_gemm_tn_adam(dy, ldy, z_dy, x, K, z_x, fmt, M, N, K, alpha, fz.data(), (int)with, t->adam_lr,
                                               t->adam_b1, t->adam_b2, t->adam_eps, t->adam_wd, ldo, which == 0 ? 0.125f : 1.0f,
                                               fmt, s));
            if (with < fz.size())
                HIP_TRY(c, launch_gemm_tn_adam(dy, ldy, z_dy, x, K, z_x, fmt, M, N, K, alpha, fz.data() + with,
                                               (int)(fz.size() - with), t->adam_lr, t->adam_b1, t->adam_b2, t->adam_eps,
                                               t->adam_wd, ldo, 1.0f, fmt, s));
            for (const std::string& nm : fresh_names) t->fresh.insert(nm);   // enqueued: the next reload skips them
        }
        const long step = NLy > 1 && all ? (long)(d[1] - d[0]) : 0;
        bool even = all && NLy > 1 && NLy <= TN_ADAM_BATCH && step >= (long)N * K && step % 4 == 0;
        for (int l = 2; even && l < NLy; ++l) even = d[l] - d[l - 1] == step;
        if (even) {
            HIP_TRY(c, launch_gemm_tn(dy, ldy, z_dy, x, K, z_x, fmt, d[0], K, step, M, N, K, NLy, alpha, s, 0, 0, 0, db.data()));
            return SLSMI_OK;
        }
        for (int l = 0; l < NLy; ++l)
            if (d[l]) {
                float* one = db_left[l];
                HIP_TRY(c, launch_gemm_tn(dy + l * z_dy, ldy, 0, x + l * z_x, K, 0, fmt, d[l], K, 0, M, N, K, 1, alpha, s, 0, 0, 0, &one));
                db_left[l] = nullptr;
            }
        bool any_left = false;
        for (int l = 0; l < NLy; ++l) any_left = any_left || db_left[l];
        if (any_left) HIP_TRY(c, launch_colsum_layers(dy, z_dy, ldy, M, N, alpha, db_left.data(), NLy, s));
        return SLSMI_OK;
    };
    {
        const long z16 = (long)(p.s_row16 / 2), zf = (long)(p.s_ffn / 2), zq = (long)(p.s_qkv / 2);
        const bf16_t* d2 = B16(p.rD2);
        const bf16_t* d1 = B16(p.rD1);
        const bf16_t* dO = B16(p.rDo);
        const bf16_t* dq = B16(p.rDq);
        TRY(layer_grads(5, "fc2.weight", "fc2.bias", d2, 1024, z16, 1024, B16(p.f), zf, Fd, 1.0f));
        TRY(layer_grads(4, "fc1.weight", "fc1.bias", d1, Fd, zf, Fd, B16(p.h2), z16, 1024, 1.0f));
        TRY(layer_grads(3, "self_attn.out_proj.weight", "self_attn.out_proj.bias", dO, 1024, z16, 1024, B16(p.ao), z16, 1024, 1.0f));
        // (the forward holds q's weight and bias scaled by 1/8: the gradient at the reference's parameters carries it too)
        TRY(layer_grads(0, "self_attn.q_proj.weight", "self_attn.q_proj.bias", dq, 3072, zq, 1024, B16(p.h1), z16, 1024, 0.125f));
        TRY(layer_grads(1, "self_attn.k_proj.weight", "self_attn.k_proj.bias", dq + 1024, 3072, zq, 1024, B16(p.h1), z16, 1024, 1.0f));
        TRY(layer_grads(2, "self_attn.v_proj.weight", "self_attn.v_proj.bias", dq + 2048, 3072, zq, 1024, B16(p.h1), z16, 1024, 1.0f));
    }
    t->fused.clear();   // one backward, one step
    // every gradient of the encoder layers' parameters (95 % of the model's) is enqueued: a gradient exchange may
    // start behind this point while the positional conv's and the feature extractor's backward still run
    if (t->layer_grads_event) HIP_TRY(c, hipEventRecord(t->layer_grads_event, s));
    if (mk.encoder) HIP_TRY(c, launch_mul_f32(gx, gx, mk.encoder, nullptr, M * 1024, s));  // back through wav2vec2.py:933
    // ---- positional conv: x1 = x0 + gelu(u)
    HIP_TRY(c, launch_posconv_grad_pack(gx, F32(p.u), B16(p.du16), B16(p.dug), B, T, p.slab, s));
    if (float* d = G(bb + "encoder.pos_conv.0.bias"))
        HIP_TRY(c, launch_colsum(B16(p.du16), FB, 1024, M, 1024, 1.0f, 0, colpart, d, s));
    {   // dx0 = dx1 + transposed conv of du: the forward's batched GEMM on the flipped weight
        GemmArgs g;
        g.fmt = FB;
        g.A = B16(p.dug) + 64; g.lda = 64; g.W = t->pos_wf; g.ldw = 8192; g.bias = nullptr;
        g.C = galt; g.resid = gx; g.ldc = 1024; g.M = T; g.N = 64; g.K = 8192; g.mode = GEMM_F32_RESID; g.bn = 64;
        g.nz = B * 16; g.zA = (long)p.slab * 64; g.zWmod = 16; g.zW = 64L * 8192; g.zBias = 0;
        g.zCdiv = 16; g.zC1 = (long)T * 1024; g.zC2 = 64;
        HIP_TRY(c, launch_gemm(g, s));
    }
    float* dv = G(bb + "encoder.pos_conv.0.weight_v");
    float* dg = G(bb + "encoder.pos_conv.0.weight_g");
    if (dv || dg) {
        // dW[g][co][(j, ci)] = sum over (b, t) of du[b t][g 64 + co] * xg[b][g][t + j][ci]: per group one batch entry of
        // the weight-gradient kernel over all B T rows -- the group's 64 columns of du as they lie, the overlapping-row
        // (Toeplitz) view of the group's padded input slab, a group of T rows per clip -- no transposed operand copies
        HIP_TRY(c, launch_gemm_tn(B16(p.du16), 1024, 64, B16(p.xg), 64, (long)p.slab * 64, fmt, F32(p.dwfold), 8192, 64L * 8192, M,
                                  64, 8192, 16, 1.0f, s, T, (long)T * 1024, 16L * p.slab * 64));
        HIP_TRY(c, launch_weightnorm_bwd(F32(p.dwfold), c->pos_v, c->pos_g, dv, dg, colpart, s));
    }
    if (mk.input) HIP_TRY(c, launch_mul_f32(galt, galt, mk.input, nullptr, M * 1024, s));  // back through dropout_input
    // galt: gradient at x0 (post_extract_proj output, compact rows)
    // ---- post_extract_proj: x0 = featln Wp^T + bp (featln in the padded row layout)
    if (float* d = G(bb + "post_extract_proj.bias")) HIP_TRY(c, launch_colsum(galt, -1, 1024, M, 1024, 1.0f, 0, colpart, d, s));
    HIP_TRY(c, launch_cvt_bf16(galt, g16, M * 1024, s));
    if (float* d = G(bb + "post_extract_proj.weight")) {
        // per clip one batch entry of the weight-gradient kernel over the clip's T valid rows of featln (padded row
        // layout) and of the compact gradient; the clips' partial sums are added in index order
        float* parts = F32(p.ctB);  // [B][1024][512]
        HIP_TRY(c, launch_gemm_tn(g16, 1024, (long)T * 1024, B16(p.featln), 512, (long)p.P[n - 1] * 512, fmt, parts, 512,
                                  1024L * 512, T, 1024, 512, B, 1.0f, s));
        HIP_TRY(c, launch_conv_grad_unpack(parts, d, 1024, 512, 1, B, s));
    }
    // ---- gradient at featln (padded rows), feature LayerNorm
    HIP_TRY(c, gemm_bf(c, g16, 1024, t->projT, 1024, F32(p.T1), 512, M, 512, 1024, GEMM_F32, nullptr, s));
    float* da = F32(p.cda2);      // gradient at a_i, padded rows
    float* da_prev = F32(p.cda);  // gradient at a_{i-1}
    HIP_TRY(c, launch_expand_rows(F32(p.T1), da_prev, B, T, p.P[n - 1], 512, s));
    HIP_TRY(c, launch_ln_bwd(512, B16(p.a[n - 1]), fmt, da_prev, c->feat_ln_g, nullptr, eps, 0, nullptr, da, 0, lnpart,
                             G(bb + "layer_norm.weight"), G(bb + "layer_norm.bias"), (long)B * p.P[n - 1], p.P[n - 1], T, s));
    // ---- conv blocks n-1 .. 1: a_i = gelu(LN(y_i)), y_i = flat GEMM over rows of a_{i-1}
    for (int i = n - 1; i >= 1; --i) {
        const std::string cp = bb + "feature_extractor.conv_layers." + std::to_string(i) + ".";
        const int k = cfg.conv_kernel[i], st = cfg.conv_stride[i];
        if (k > 3) return fail(c, SLSMI_ERR_INVALID, "conv kernels wider than 3 are not supported past the first layer");
        const long R = (long)B * p.P[i];
        bf16_t* dy = B16(p.cdy);
        HIP_TRY(c, launch_ln_bwd(512, B16(p.y[i]), fmt, da, c->conv[i].ln_g, c->conv[i].ln_b, eps, 1, nullptr, dy, 1, lnpart,
                                 G(cp + "2.1.weight"), G(cp + "2.1.bias"), R, p.P[i], p.T[i], s));
        if (float* d = G(cp + "0.bias")) HIP_TRY(c, launch_colsum(dy, FB, 512, R, 512, 1.0f, 0, colpart, d, s));
        if (float* d = G(cp + "0.weight")) {
            // dW[o][(tap, ci)] = sum over clips b and their valid frames t of dy[b][t][o] a_{i-1}[b][t st + tap][ci]:
            // the weight-gradient kernel contracts the row-major dy with the overlapping-row view of a_{i-1} as both
            // lie (no transposed copies), one batch entry per clip over its T_i valid frames -- rows past them are
            // never read -- and the unpack adds the clips' partial sums in index order
            float* parts = F32(p.ctB);  // [B][512][k 512]
            HIP_TRY(c, launch_gemm_tn(dy, 512, (long)p.P[i] * 512, B16(p.a[i - 1]), (long)st * 512, (long)p.P[i - 1] * 512, fmt,
                                      parts, (long)k * 512, (long)512 * k * 512, p.T[i], 512, k * 512, B, 1.0f, s));
            HIP_TRY(c, launch_conv_grad_unpack(parts, d, 512, 512, k, B, s));
        }
        HIP_TRY(c, gemm_bf(c, dy, 512, t->convT[i], 512, B16(p.cdcol), (long)k * 512, R, k * 512, 512, GEMM_BF16, nullptr, s));
        // da is dead once dy exists, so the output may even take its place: only cda holds the rows of a_0
        float* out = i == 1 ? F32(p.cda) : (da == F32(p.cda2) ? F32(p.cda) : F32(p.cda2));
        HIP_TRY(c, launch_col2im(B16(p.cdcol), out, B, p.P[i - 1], p.P[i], k, st, s));
        da = out;
    }
    // ---- first conv block
    {
        const std::string cp = bb + "feature_extractor.conv_layers.0.";
        float* dW = G(cp + "0.weight");
        float* dB = G(cp + "0.bias");
        float* dGm = G(cp + "2.1.weight");
        float* dBt = G(cp + "2.1.bias");
        if (dW || dB || dGm || dBt) {
            HIP_TRY(c, launch_conv0_bwd(wav, B, L, p.T[0], p.P[0], c->conv[0].w0, c->conv[0].bias, c->conv[0].ln_g,
                                        c->conv[0].ln_b, eps, da, F32(p.c0part), F32(p.c0sum), s));
            const float* sum = F32(p.c0sum);
            if (dW) HIP_TRY(c, hipMemcpyAsync(dW, sum, 5120 * 4, hipMemcpyDeviceToDevice, s));
            if (dB) HIP_TRY(c, hipMemcpyAsync(dB, sum + 5120, 512 * 4, hipMemcpyDeviceToDevice, s));
            if (dGm) HIP_TRY(c, hipMemcpyAsync(dGm, sum + 5632, 512 * 4, hipMemcpyDeviceToDevice, s));
            if (dBt) HIP_TRY(c, hipMemcpyAsync(dBt, sum + 6144, 512 * 4, hipMemcpyDeviceToDevice, s));
        }
    }
#undef TRY
    return SLSMI_OK;
}

extern "C" {

int slsmi_train_forward(slsmi_ctx* c, const float* wav, int B, int L, void* workspace, size_t workspace_bytes,
                        const float* dropout_mask, float* logp_out, float* sae_loss_out, void* stream) {
    return train_forward_impl(c, wav, B, L, workspace, workspace_bytes, dropout_mask, logp_out, sae_loss_out, nullptr, stream);
}

int slsmi_train_backward(slsmi_ctx* c, const float* wav, int B, int L, void* workspace, size_t workspace_bytes,
                         const float* dlogp, const float* dsae_loss, int used_dropout_mask, void* stream) {
    return train_backward_impl(c, wav, B, L, workspace, workspace_bytes, dlogp, dsae_loss, used_dropout_mask, nullptr, stream);
}

int slsmi_train_backbone_forward(slsmi_ctx* c, const float* wav, int B, int L, void* workspace, size_t workspace_bytes,
                                 float* layers_out, void* stream) {
    if (!layers_out) return fail(c, SLSMI_ERR_INVALID, "null argument");
    return train_forward_impl(c, wav, B, L, workspace, workspace_bytes, nullptr, nullptr, nullptr, layers_out, stream);
}

int slsmi_train_backbone_backward(slsmi_ctx* c, const float* wav, int B, int L, void* workspace, size_t workspace_bytes,
                                  const float* dlayers, void* stream) {
    if (!dlayers) return fail(c, SLSMI_ERR_INVALID, "null argument");
    return train_backward_impl(c, wav, B, L, workspace, workspace_bytes, nullptr, nullptr, 0, dlayers, stream);
}

static int check_sls_step(slsmi_ctx* c, const float* fc1_weight, int L) {
    if (!c) return SLSMI_ERR_INVALID;
    if (c->cfg.head != SLSMI_HEAD_SLS) return fail(c, SLSMI_ERR_INVALID, "this context was not created with the SLS head");
    if (!fc1_weight || (reinterpret_cast<uintptr_t>(fc1_weight) & 3)) return fail(c, SLSMI_ERR_INVALID, "fc1_weight: null or misaligned");
    int T[SLSMI_MAX_CONV];
    if (!conv_frames(c->cfg, L, T)) return fail(c, SLSMI_ERR_INVALID, "input too short for the conv stack");
    const long P = (long)(T[c->cfg.n_conv - 1] / 3) * (1024 / 3);
    if (P != c->sls_in)
        return fail(c, SLSMI_ERR_INVALID, "fc1.weight has " + std::to_string(c->sls_in) + " in-features, inputs of this length pool to " +
                                              std::to_string(P));
    return SLSMI_OK;
}

int slsmi_train_sls_forward(slsmi_ctx* c, const float* wav, int B, int L, void* workspace, size_t workspace_bytes,
                            const float* fc1_weight, float* logp_out, float* bn_batch_out, void* stream) {
    int rc = check_sls_step(c, fc1_weight, L);
    if (rc) return rc;
    if (!logp_out) return fail(c, SLSMI_ERR_INVALID, "null argument");
    SlsStep st;
    st.fc1_w = fc1_weight;
    st.logp_out = logp_out;
    st.bn_batch_out = bn_batch_out;
    return train_forward_impl(c, wav, B, L, workspace, workspace_bytes, nullptr, nullptr, nullptr, nullptr, stream, &st);
}

int slsmi_train_sls_backward(slsmi_ctx* c, const float* wav, int B, int L, void* workspace, size_t workspace_bytes,
                             const float* fc1_weight, const float* dlogp, void* stream) {
    int rc = check_sls_step(c, fc1_weight, L);
    if (rc) return rc;
    if (!dlogp) return fail(c, SLSMI_ERR_INVALID, "null argument");
    SlsStep st;
    st.fc1_w = fc1_weight;
    st.dlogp = dlogp;
    return train_backward_impl(c, wav, B, L, workspace, workspace_bytes, nullptr, nullptr, 0, nullptr, stream, &st);
}

}  // extern "C"
