// Context of one model on one device and the helpers every translation unit behind the C ABI
// shares (engine.hip: set-up and the scoring forward; train.hip: the training step).
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/slsmi.h"
#include "kernels.h"

namespace slsmi {

constexpr size_t ALIGN = 256;
inline size_t align_up(size_t v, size_t a = ALIGN) { return (v + a - 1) / a * a; }

struct ConvLayer {
    float* w0 = nullptr;    // layer 0 only: fp32 [512][10]
    bf16_t* w = nullptr;    // layers >= 1: bf16 [512][k*512]
    bf16_t* w2 = nullptr;   // layers >= 1: [512][2][k*512]: w and the remainder of its rounding (scoring forward)
    float* bias = nullptr;
    float* ln_g = nullptr;
    float* ln_b = nullptr;
};

struct EncLayer {
    bf16_t* wqkv = nullptr;  // [3072][1024], q rows pre-scaled
    float* bqkv = nullptr;   // [3072]
    bf16_t* wo = nullptr;    // [1024][1024]
    float* bo = nullptr;
    float* ln1_g = nullptr;
    float* ln1_b = nullptr;
    bf16_t* w1 = nullptr;    // [ffn][1024]
    float* b1 = nullptr;
    bf16_t* w2 = nullptr;    // [1024][ffn]
    float* b2 = nullptr;
    float* ln2_g = nullptr;
    float* ln2_b = nullptr;
    // LayerNorm folded into the q/k/v and fc1 GEMMs (GemmArgs::LN_APPLY): g = W gamma, c = bias + W beta
    float* qkv_g = nullptr;  // [3072]
    float* qkv_c = nullptr;
    float* fc1_g = nullptr;  // [ffn]
    float* fc1_c = nullptr;
};


// Transposed bfloat16 copies of the GEMM weights ([K][N] for a weight [N][K]) and the gradient
// destinations of a context in training mode (train.hip).
struct TrainState;

}  // namespace slsmi

struct slsmi_ctx;
namespace slsmi {
int train_prepare(slsmi_ctx* c, hipStream_t s);  // transposed weight copies, at slsmi_finalize
void train_destroy(slsmi_ctx* c);
}  // namespace slsmi

struct slsmi_ctx {
    int device = 0;
    slsmi_config cfg;
    std::string err;
    std::vector<void*> allocs;
    std::map<std::string, bool> loaded;
    // where each weight was last loaded from (slsmi_reload_weights reads the same buffers again)
    struct WeightSource {
        const void* data = nullptr;
        int64_t shape[4] = {0, 0, 0, 0};
        int ndim = 0;
    };
    std::map<std::string, WeightSource> sources;
    bool reloading = false;
    std::vector<slsmi::ScaleCopy> pending_copies;  // small vectors of a reload, copied together at its end
    std::vector<slsmi::CvtTransposeW> pending_w;   // and its weight matrices, converted together by shape
    bool finalized = false;

    slsmi::ConvLayer conv[SLSMI_MAX_CONV];
    float* feat_ln_g = nullptr;
    float* feat_ln_b = nullptr;
    slsmi::bf16_t* proj_w = nullptr;
    slsmi::bf16_t* proj_w3 = nullptr;    // [1024][3 * 512]: [hi | hi | lo] (scoring forward)
    float* proj_b = nullptr;
    float* pos_v = nullptr;   // fp32 staging [1024][64][128]
    float* pos_g = nullptr;   // fp32 staging [128]
    float* pos_norm = nullptr;
    slsmi::bf16_t* pos_w = nullptr;  // [16][64][8192]
    float* pos_b = nullptr;
    std::vector<slsmi::EncLayer> enc;
    float* final_ln_g = nullptr;
    float* final_ln_b = nullptr;
    // SLS head
    float* bn = nullptr;      // [4]: weight, bias, running_mean, running_var
    float* fc0_w = nullptr;
    float* fc0_b = nullptr;
    slsmi::bf16_t* sls_fc1_w = nullptr;  // [1024][3 * sls_ld]: [hi | hi | lo]
    long sls_in = 0, sls_ld = 0;
    float* sls_fc1_b = nullptr;
    float* fc3_w = nullptr;
    float* fc3_b = nullptr;
    // SAE head
    slsmi::bf16_t* sae_enc_w = nullptr;  // [dict][1024]
    float* sae_enc_b = nullptr;
    float* sae_b_dec = nullptr;
    slsmi::bf16_t* sae_dec_w = nullptr;  // [1024][dict]
    float* cls_ln_g = nullptr;
    float* cls_ln_b = nullptr;
    float* cls_w1 = nullptr;
    float* cls_b1 = nullptr;
    float* cls_w2 = nullptr;
    float* cls_b2 = nullptr;
    slsmi::TrainState* train = nullptr;  // set by slsmi_enable_training
    float* split_ws = nullptr;           // K-split scratch of this context's GEMMs (made by slsmi_finalize)
    bool split_scoring = false;          // slsmi_set_low_latency: the scoring forward may split K too
    bool ln_fold = true;                 // the scoring forward folds the encoder's LayerNorms into its GEMMs
    bool ln_fold_force_wide = false;     // (tests) the 208 x 256 form for every batch it can run, not only where it wins
    bool posconv_slab = true;            // the scoring forward runs the positional conv on the slab-resident kernel (clips of <= 208 frames)
    bool fold_consts_ready = false;      // EncLayer::qkv_g .. fc1_c match the loaded weights
    // Scoring forward, conv layers 1 .. 6 (bit i = layer i): which multiply by the split weights (w2), and which
    // hand their output to the LayerNorm as fp32 (slsmi_set_conv_precision)
    // slsmi_set_range_report: device array [range_capacity][2] the scoring forward fills with the range of its 16-bit
    // stores (null: off)
    unsigned* range_out = nullptr;
    int range_capacity = 0;
    unsigned conv_split_mask = SLSMI_CONV_SPLIT_DEFAULT;
    unsigned conv_fp32_mask = SLSMI_CONV_FP32_DEFAULT;
    // (the projection after the feature LayerNorm and the SLS head's fc1 always multiply split operands by split weights)
};


namespace slsmi {

inline int fail(slsmi_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}

inline int hip_fail(slsmi_ctx* c, hipError_t e, const char* what) {
    return fail(c, SLSMI_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

#define HIP_TRY(ctx, expr)                                   \
    do {                                                     \
        hipError_t _e = (expr);                              \
        if (_e != hipSuccess) return hip_fail(ctx, _e, #expr); \
    } while (0)

template <typename T>
int dev_alloc(slsmi_ctx* c, T** p, size_t count) {
    if (*p) return SLSMI_OK;
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, align_up(count * sizeof(T)));
    if (e != hipSuccess) return hip_fail(c, e, "hipMalloc");
    c->allocs.push_back(q);
    *p = reinterpret_cast<T*>(q);
    return SLSMI_OK;
}

// Training mode only (else nullptr): the transposed bfloat16 copy of an encoder layer's matrix, allocated on first
// use, for the weight load to fill together with the forward's copy.  which: 0 q, 1 k, 2 v (thirds of the
// [1024][3072] matrix: the returned pointer is the third's first column), 3 out_proj, 4 fc1, 5 fc2; *ldo: its row stride.
bf16_t* train_layer_wT(slsmi_ctx* c, int layer, int which, long* ldo);
// Weights whose 16-bit copies the last training backward wrote itself (the Adam step taken in the weight-gradient
// kernel's epilogue also writes them): a reload has nothing to do for them.  The set lasts until the next reload;
// loading a weight by name, or train_forget_fresh_copies, withdraws entries.
bool train_copy_is_fresh(const slsmi_ctx* c, const std::string& name);
void train_forget_fresh_copies(slsmi_ctx* c, const std::string* name = nullptr);   // null: all

inline bool conv_frames(const slsmi_config& cfg, int length, int* T) {
    long cur = length;
    for (int i = 0; i < cfg.n_conv; ++i) {
        if (cur < cfg.conv_kernel[i]) return false;
        cur = (cur - cfg.conv_kernel[i]) / cfg.conv_stride[i] + 1;
        T[i] = (int)cur;
    }
    return cur >= 1;
}


}  // namespace slsmi
