/* libslsmi -- C ABI of the MI355X (gfx950) scoring path:
 * XLS-R-300M wav2vec2 forward + SLS / TopK-SAE head -> log-probabilities.
 *
 * The reference has no native layer: its boundary for this path is the Python
 * call  model.Model(args, device).forward(wav)  (reference model.py:150-260,
 * upstream SLS form Model(args, device)).  Each entry point below names the
 * reference code it replaces.  Conventions:
 *   - every pointer argument that is not `const char*`, `slsmi_ctx*` or an
 *     out-parameter of a create call is a DEVICE pointer owned by the caller;
 *   - work is enqueued on the given HIP stream (a `hipStream_t` passed as
 *     void*, NULL = default stream); compute calls never synchronise, allocate
 *     or free;  slsmi_create / slsmi_load_weight / slsmi_finalize /
 *     slsmi_destroy may allocate and synchronise (set-up time);
 *   - return value 0 = success, negative = slsmi_status; nothing throws or
 *     exits.  slsmi_last_error(ctx) gives a readable message for the last
 *     failure on that context;
 *   - a context is bound to one device; calls on one context must be
 *     serialised by the caller, distinct contexts are independent.
 */
#ifndef SLSMI_H_
#define SLSMI_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct slsmi_ctx slsmi_ctx;

typedef enum slsmi_status {
    SLSMI_OK = 0,
    SLSMI_ERR_INVALID = -1,     /* bad argument / unsupported geometry */
    SLSMI_ERR_HIP = -2,         /* a HIP runtime call failed */
    SLSMI_ERR_MISSING = -3,     /* a required weight was never loaded */
    SLSMI_ERR_WORKSPACE = -4,   /* workspace too small or misaligned */
    SLSMI_ERR_NAME = -5,        /* unknown weight name */
    SLSMI_ERR_AUDIO_OPEN = -6,  /* audio file cannot be read */
    SLSMI_ERR_AUDIO_FORMAT = -7,/* not a WAV / FLAC file this library decodes */
    SLSMI_ERR_AUDIO_DATA = -8,  /* corrupt or empty audio stream (bad sync, CRC mismatch, ...) */
    SLSMI_ERR_AUDIO_RATE = -9   /* sample rate differs from the expected one (no resampling here) */
} slsmi_status;

enum { SLSMI_HEAD_SLS = 0, SLSMI_HEAD_SAE = 1, SLSMI_HEAD_NONE = 2 /* backbone only */ };
enum { SLSMI_F32 = 0 };
/* 16-bit format of GEMM operands and stored activations (fp32 accumulation,
 * fp32 residual stream and statistics in both): same MFMA rate, 8 vs 11
 * mantissa bits. */
enum { SLSMI_FMT_BF16 = 0, SLSMI_FMT_F16 = 1 };
enum { SLSMI_SAE_SPARSE = 0, SLSMI_SAE_RECON = 1, SLSMI_SAE_OFF = 2 };

#define SLSMI_MAX_CONV 8

/* Architecture description; replaces the fairseq Wav2Vec2Config the reference
 * reads from the checkpoint (wav2vec/wav2vec2.py:39-252) plus the head
 * arguments of Model.__init__ (model.py:150-160). */
typedef struct slsmi_config {
    int32_t n_conv;                      /* 7 */
    int32_t conv_dim[SLSMI_MAX_CONV];    /* 512 each (only 512 supported) */
    int32_t conv_kernel[SLSMI_MAX_CONV]; /* 10,3,3,3,3,2,2 */
    int32_t conv_stride[SLSMI_MAX_CONV]; /* 5,2,2,2,2,2,2 */
    int32_t embed_dim;                   /* 1024 (only 1024 supported) */
    int32_t ffn_dim;                     /* 4096 (multiple of 128) */
    int32_t heads;                       /* 16 (head width 64 only) */
    int32_t layers;                      /* 24 */
    int32_t conv_pos;                    /* 128 */
    int32_t conv_pos_groups;             /* 16 */
    float ln_eps;                        /* 1e-5 */
    int32_t head;                        /* SLSMI_HEAD_SLS | SLSMI_HEAD_SAE | SLSMI_HEAD_NONE */
    int32_t sae_dict;                    /* 4096 */
    int32_t sae_k;                       /* 128 */
    float bn_eps;                        /* 1e-5 (SLS head BatchNorm2d) */
    int32_t operand_format;              /* SLSMI_FMT_BF16 | SLSMI_FMT_F16 */
    int32_t sae_features;                /* SLSMI_HEAD_SAE only: what the classifier pools (model.py:171-179,
                                          * :227-242): SLSMI_SAE_SPARSE (use_sae, use_sparse_features),
                                          * SLSMI_SAE_RECON (use_sae, reconstruction), SLSMI_SAE_OFF (use_sae=False) */
    int32_t sae_window;                  /* SLSMI_HEAD_SAE only: 0 or 1 = TopK per frame (model.py:73-77); w >= 2 = the
                                          * window TopK of model_window_topk.py:118-203 (sae_window_size, default 8
                                          * there), windows of w frames every max(1, w / 2) frames */
} slsmi_config;

int slsmi_version(void);

/* Fills *cfg with the XLS-R-300M geometry and the given head. */
void slsmi_default_config(slsmi_config* cfg, int head);

/* Replaces SSLModel.__init__ + Model.__init__ (model.py:109-126, :150-193):
 * creates an empty context on `device`. */
int slsmi_create(int device, const slsmi_config* cfg, slsmi_ctx** out);

/* Small batches (no counterpart in the reference): with enable != 0 the scoring forward cuts the K dimension
 * of GEMMs that have too few output tiles to fill the chip (a single clip has 201 rows) into slices summed in
 * a fixed order by a second kernel -- a single clip scores in 2.3 ms instead of 3.7 ms.  A score then depends
 * on the batch SIZE in its last bits (fp32 summation order; about 2e-4 after 24 layers), which the default
 * (off) rules out: there an utterance scores the same to 2e-5 whatever batch it is in.  Results stay
 * reproducible either way.  The training step always splits. */
int slsmi_set_low_latency(slsmi_ctx* ctx, int enable);

/* The scoring forward folds the encoder layers' LayerNorms into the GEMMs around them (slsmi_gemm_bf16_ln below;
 * on by default).  enable = 0: separate LayerNorm kernels, as the training forward and low-latency mode use.
 * Both forms compute the reference's Linear(LayerNorm(x)); they round at different points, so scores differ in
 * their last bits (within the tolerance of the path). */
int slsmi_set_ln_fold(slsmi_ctx* ctx, int enable);   /* 2 (tests): the 208 x 256 form for every batch it can run */
/* 1 (default): the scoring forward runs the positional conv on the slab-resident kernel where it applies
 * (clips in segments of 208 frames; same bits as the batched-GEMM form, which 0 selects everywhere). */
int slsmi_set_posconv_slab(slsmi_ctx* ctx, int enable);
/* Where the scoring forward's feature extractor spends extra precision (wav2vec2.py:803-813; bit i of a mask = conv
 * layer i, layers 1 .. n_conv - 1).  split_mask: the layer multiplies by both 16-bit halves of its weights (the
 * rounded weight and the remainder of that rounding: the K loop doubles, the weights' rounding no longer shows).
 * fp32_mask: the layer's output reaches its LayerNorm as fp32, as the reference's Fp32LayerNorm sees it
 * (wav2vec2.py:807-812), instead of rounded to 16 bits in between.  Defaults below: what holds every score of the
 * 2048-clip fixture within 1e-3 of the fp32 oracle; 0, 0 is the cheapest form.  Takes effect at the next forward;
 * slsmi_workspace_bytes follows the fp32 mask. */
/* Range report of the scoring forward's 16-bit stores (no counterpart in the reference, which computes in fp32):
 * whether a checkpoint's activations fit the operand format.  While `out` (device memory, capacity x 2 uint32) is
 * set, every slsmi_forward / slsmi_forward_varlen first clears it and then writes, for store number i, out[2 i] =
 * the bits of the largest |value| as an IEEE float and out[2 i + 1] = how many values sit at or beyond the
 * format's largest finite value (65504 for f16: the stores saturate there).  Store numbers: SLSMI_RANGE_CONV + i =
 * output of conv block i; SLSMI_RANGE_FEATURES = the normalised features in front of post_extract_proj;
 * SLSMI_RANGE_POOLED = the SLS head's pooled features; SLSMI_RANGE_LAYER0 + SLSMI_RANGE_PER_LAYER * l + k for
 * encoder layer l, k = 0 the operand its first LayerNorm yields, 1 q/k/v, 2 the attention output, 3 the operand of
 * the second LayerNorm, 4 the feed-forward's hidden activations.  Adds one small kernel per store (for inspection
 * runs, not for throughput).  NULL switches it off. */
#define SLSMI_RANGE_CONV 0
#define SLSMI_RANGE_FEATURES 8
#define SLSMI_RANGE_POOLED 9
#define SLSMI_RANGE_LAYER0 10
#define SLSMI_RANGE_PER_LAYER 5
int slsmi_set_range_report(slsmi_ctx* ctx, uint32_t* out, int capacity);
#define SLSMI_CONV_SPLIT_DEFAULT 0x78u /* layers 3-6 */
#define SLSMI_CONV_FP32_DEFAULT 0x78u  /* layers 3-6 */
int slsmi_set_conv_precision(slsmi_ctx* ctx, unsigned split_mask, unsigned fp32_mask);

/* Replaces nn.Module.load_state_dict for this path (main.py:562-592).  `name`
 * is the reference state-dict key without any 'module.' prefix, e.g.
 * "ssl_model.model.encoder.layers.3.fc1.weight", "fc0.weight",
 * "sae.encoder.weight".  `data` is a device pointer to a contiguous tensor of
 * `dtype` (SLSMI_F32) with the given shape; it is read during the call's
 * stream work and not retained: the library repacks into its own bf16/fp32
 * copies (folding the q scaling and the positional-conv weight norm).  Keys
 * the forward never reads (mask_emb, quantizer.*, project_q.*, final_proj.*,
 * *.num_batches_tracked, sae.k) are accepted and ignored. */
int slsmi_load_weight(slsmi_ctx* ctx, const char* name, const void* data, int dtype,
                      const int64_t* shape, int ndim, void* stream);

/* Checks that every weight the configured path needs is present and finishes
 * derived tensors.  Must be called after the last slsmi_load_weight and before
 * slsmi_forward; may be called again after weights were reloaded. */
int slsmi_finalize(slsmi_ctx* ctx, void* stream);

/* Loads every weight again from the device buffer it was last loaded from, then finalizes: the
 * library's copies follow an optimizer that updated the caller's tensors in place (the reference
 * trains with Adam on the module's own parameters, main.py:526, :236-238, and its forward reads
 * them directly; here one call per step replaces the per-tensor loads).  The caller guarantees that
 * each buffer passed to slsmi_load_weight is still alive at the same address with the same shape.  Unlike
 * slsmi_finalize this call only enqueues work on `stream` and never waits for the device. */
int slsmi_reload_weights(slsmi_ctx* ctx, void* stream);

/* Frames produced for an input of `length` samples
 * (wav2vec/wav2vec2.py:523-538); negative on error. */
int slsmi_num_frames(const slsmi_ctx* ctx, int length);

/* Bytes of scratch slsmi_forward needs for a batch of `batch` utterances of
 * `length` samples; 0 on error. */
size_t slsmi_workspace_bytes(const slsmi_ctx* ctx, int batch, int length);

/* Replaces Model.forward in eval mode (model.py:195-260; upstream SLS
 * Model.forward; backbone wav2vec/wav2vec2.py:540-647 with mask=False,
 * features_only=True):
 *   wav        [batch][length] fp32
 *   workspace  >= slsmi_workspace_bytes(), 256-byte aligned
 *   logp_out   [batch][2] fp32 log-probabilities (NULL only with SLSMI_HEAD_NONE)
 *   x_out      NULL or [batch][T][1024] fp32: encoder output after the final
 *              LayerNorm (what SSLModel.extract_feat returns, model.py:140)
 *   layers_out NULL or [layers][batch][T][1024] fp32: per-layer outputs before
 *              the final LayerNorm (layer_results, wav2vec2.py:958; batch-major)
 *   sae_loss_out NULL or [1] fp32: SAE reconstruction MSE (model.py:224-225);
 *              SAE head only
 */
int slsmi_forward(slsmi_ctx* ctx, const float* wav, int batch, int length, void* workspace,
                  size_t workspace_bytes, float* logp_out, float* x_out, float* layers_out,
                  float* sae_loss_out, void* stream);

/* The same forward for a batch of utterances of different lengths, each zero- or
 * anything-padded to `length` samples: lengths [batch] int32 (DEVICE pointer) holds
 * the sample count of every utterance.  Replaces the padding_mask path of the
 * reference backbone (wav2vec/wav2vec2.py:567-586 frame mask from sample counts,
 * :912-913 padded frames zeroed before the positional conv, :1050 key-padding mask
 * of every attention; no caller in the reference passes a mask -- its datasets cut
 * or tile every clip to 64600 samples -- so this is the batched form of running
 * Model.forward on each utterance alone at its own length, which it reproduces):
 * utterance b's result depends on its first lengths[b] samples only; the time mean
 * of the SAE-type heads (model.py:245) runs over its own frames.  A count whose
 * frame number falls outside [1, T] is clamped.  Rows of x_out / layers_out past
 * an utterance's own frames are unspecified.  Not for the SLS head (its fc1 fixes
 * T); no SAE loss. */
int slsmi_forward_varlen(slsmi_ctx* ctx, const float* wav, int batch, int length, const int32_t* lengths,
                         void* workspace, size_t workspace_bytes, float* logp_out, float* x_out,
                         float* layers_out, void* stream);

/* Where a by-product of the last slsmi_forward on this (batch, length) sits in the
 * workspace: byte offset and size.  SLSMI_VIEW_SAE_CODE: the TopK sparse code,
 * fp32 [batch * T][sae_dict] (`encoded`, model.py:221; what
 * Model.get_interpretability_info reads, model.py:262-293).  Valid until the next
 * forward on the same workspace. */
enum { SLSMI_VIEW_SAE_CODE = 0 };
int slsmi_workspace_view(const slsmi_ctx* ctx, int batch, int length, int which, size_t* offset,
                         size_t* bytes);

/* ---- training step (reference main.py:201-258: model(batch_x, return_sae_loss=True), the loss,
 * loss.backward()).  Covers the in-tree model in every form main.py can build: XLS-R backbone + TopK SAE + classifier
 * (model.py:195-260) with the classifier on the sparse code, on its reconstruction (use_sparse_features=False,
 * model.py:228-233) or on the encoder output without an auto-encoder (use_sae=False, model.py:171-179), and TopK per
 * frame or by window votes (sae_window > 1: model_window_topk.py:271-393, main.py:493-505).  A TopK selection is a
 * constant mask for the backward, as in the reference.  The loss itself (NLLLoss with class weights,
 * main.py:524-525, and compute_total_loss, model.py:295-299) stays with the caller: the library
 * takes the gradient at the two outputs and writes the gradient of every registered parameter.
 * Gradient contractions use bfloat16 operands with fp32 accumulation.  batch <= 64, frames <= 256. */

/* Switches a context to training mode: the next slsmi_finalize also builds the transposed weight
 * copies the backward GEMMs read.  (nn.Module.train(), main.py:204.) */
int slsmi_enable_training(slsmi_ctx* ctx);

/* Where slsmi_train_backward writes (overwrites) the gradient of parameter `name` (reference
 * state-dict key, e.g. "ssl_model.model.encoder.layers.3.fc1.weight"): a device fp32 buffer of the
 * parameter's shape, 16-byte aligned; NULL removes the entry.  Parameters without an entry get no
 * gradient.  Replaces autograd's accumulation into param.grad (main.py:236-238).
 * The encoder layers' weight matrices of one kind (fc1.weight of layers 0, 1, ...) are written by one GEMM launch
 * over all layers when their buffers lie at equal distances in layer order (slices of one [layers][rows][cols]
 * tensor), by one launch per layer otherwise; the values are the same. */
int slsmi_set_grad(slsmi_ctx* ctx, const char* name, float* dst);

size_t slsmi_train_workspace_bytes(const slsmi_ctx* ctx, int batch, int length);

/* Model.forward in train mode (model.py:195-260) keeping every intermediate the backward needs in
 * the workspace.  dropout_mask: NULL, or device [batch][256] multipliers of the classifier's
 * Dropout(0.3) (model.py:187; 0 or 1 / 0.7), drawn by the caller.  logp_out [batch][2];
 * sae_loss_out NULL or [1]. */
int slsmi_train_forward(slsmi_ctx* ctx, const float* wav, int batch, int length, void* workspace,
                        size_t workspace_bytes, const float* dropout_mask, float* logp_out,
                        float* sae_loss_out, void* stream);

/* loss.backward() (main.py:237) for the last slsmi_train_forward on this workspace: dlogp
 * [batch][2] = dLoss/dlogp, dsae_loss NULL or [1] = dLoss/dsae_loss (device pointers); wav: the same
 * input again (the first conv block is re-computed from it).  used_dropout_mask: whether that
 * forward was given a mask. */
int slsmi_train_backward(slsmi_ctx* ctx, const float* wav, int batch, int length, void* workspace,
                         size_t workspace_bytes, const float* dlogp, const float* dsae_loss,
                         int used_dropout_mask, void* stream);

/* The random draws of the backbone's train mode, made by the caller (the reference's extract_feat leaves the
 * backbone in train mode, model.py:130-132, so a checkpoint whose configuration has non-zero dropout / layerdrop
 * applies them while fine-tuning).  Every pointer is device memory and may be NULL (= not applied); the float
 * arrays hold multipliers, 0 or 1 / (1 - p), in [batch * frames] row order:
 *   input     [rows][1024]           dropout_input on the projected features (wav2vec2.py:598)
 *   encoder   [rows][1024]           F.dropout after the positional conv (:933)
 *   attn_out  [layers][rows][1024]   dropout1 of every layer (:1053)
 *   ffn_act   [layers][rows][ffn]    dropout2 (:1059)
 *   ffn_out   [layers][rows][1024]   dropout3 (:1061)
 *   layer_keep [layers] bytes, HOST memory: 0 = layerdrop skips the layer (:941-942): its output equals its
 *             input (the reference appends no layer result for it) and its parameters get zero gradients.
 * attention_dropout inside MultiheadAttention: slsmi_train_set_attention_masks.
 * dropout_features (:599) acts on a copy only the pre-training loss reads.
 * slsmi_train_set_masks applies to the NEXT slsmi_train_forward / slsmi_train_backbone_forward and its backward,
 * which must find the arrays unchanged; after that backward (or the next call with NULL) nothing is applied. */
typedef struct slsmi_train_masks {
    const float* input;
    const float* encoder;
    const float* attn_out;
    const float* ffn_act;
    const float* ffn_out;
    const uint8_t* layer_keep;
} slsmi_train_masks;
int slsmi_train_set_masks(slsmi_ctx* ctx, const slsmi_train_masks* masks);

/* MultiheadAttention's dropout on the attention probabilities in the training step (attention_dropout,
 * wav2vec2.py:1012; the class's default is 0.1 and the backbone runs in train mode while fine-tuning): the caller's
 * draws as bytes, 1 = kept, in both orientations -- keep_qk [layers][batch * 16][query][pitch] and keep_kq
 * [layers][batch * 16][key][pitch] with keep_kq[..][k][q] == keep_qk[..][q][k] (device memory; pitch a multiple of 32
 * and >= frames; layer_stride bytes between layers) -- and scale = 1 / (1 - p) for the kept ones.  Applies to the
 * NEXT training forward and its backward, like slsmi_train_set_masks; NULL pointers: no attention dropout. */
int slsmi_train_set_attention_masks(slsmi_ctx* ctx, const uint8_t* keep_qk, const uint8_t* keep_kq,
                                    int64_t layer_stride, int pitch, float scale);

/* Data-parallel training (no counterpart in the reference, which trains under single-process nn.DataParallel,
 * main.py:518): `event` is a hipEvent_t of the caller (NULL: none) that every later slsmi_train_backward /
 * slsmi_train_backbone_backward records on its stream once the gradients of ALL encoder-layer parameters
 * (ssl_model.model.encoder.layers.*: 95 % of the model's bytes) have been enqueued -- about 1.5 ms of kernels before
 * the backward's last one at 5 clips.  A stream that waits for the event may read those gradients and start their
 * exchange while the positional conv's and the feature extractor's backward still run.  The event must outlive
 * the context or be withdrawn (NULL) first. */
int slsmi_train_set_layer_grads_event(slsmi_ctx* ctx, void* event);

/* Optimizer step inside the backward (no counterpart in the reference, whose loop calls loss.backward() and then
 * torch.optim.Adam.step(), main.py:236-238; this is that step for the tensors named here, taken where their gradient
 * is produced).  slsmi_train_set_fused_adam names an encoder-layer weight matrix
 * (ssl_model.model.encoder.layers.<l>.{self_attn.{q,k,v,out}_proj,fc1,fc2}.weight: 95 % of the model's bytes), its
 * fp32 parameter tensor and its two Adam moments (dense, the parameter's shape, 16-byte aligned, device memory) and
 * the tensor's step count of the coming step (>= 1).  The NEXT training backward (any of the three forms) then does
 * not store that tensor's gradient: the weight-gradient kernel's epilogue applies torch.optim.Adam's update
 * (single-tensor form, L2 weight decay, no amsgrad; hyper-parameters from slsmi_train_set_fused_adam_hyper) to
 * param / exp_avg / exp_avg_sq in place -- the same bits as storing the gradient and running slsmi_adam_step on it --
 * and the registrations are used up (one backward, one step; register again for the next).  A gradient buffer
 * registered for the same name with slsmi_set_grad is left unwritten.  A layer skipped by layerdrop takes no step.
 * param == NULL withdraws a name.  The caller's weight reload (slsmi_reload_weights) follows as after any step.
 * When `param` is the very buffer the weight was loaded from (slsmi_load_weight keeps the address for reloads), the
 * same epilogue also writes the library's own 16-bit copies of the new values (the forward's operand and its
 * transposed bfloat16 form, bit for bit what a reload would make of them), and the NEXT slsmi_reload_weights skips
 * that tensor.  A caller who changes such a parameter by other means between that backward and the reload calls
 * slsmi_train_drop_fused_copies first: the reload then reads every tensor again.
 * Single-process training only: a gradient that is never stored cannot be averaged over ranks. */
int slsmi_train_set_fused_adam_hyper(slsmi_ctx* ctx, double lr, double beta1, double beta2, double eps, double weight_decay);
int slsmi_train_set_fused_adam(slsmi_ctx* ctx, const char* name, float* param, float* exp_avg, float* exp_avg_sq,
                               int64_t step);
int slsmi_train_drop_fused_copies(slsmi_ctx* ctx);

/* The training step of the backbone alone, for heads whose own arithmetic the caller differentiates (the
 * upstream SLS head: main.py:201-258 with Model(args, device)).  slsmi_train_backbone_forward runs the feature
 * extractor and the encoder in training form and copies every layer's output (what extract_feat returns as
 * layer_results, model_backup.py:167-184) to layers_out [layers][batch * frames][1024] fp32;
 * slsmi_train_backbone_backward takes the loss gradient at those outputs, dlayers in the same layout, and writes
 * the gradient of every backbone parameter registered with slsmi_set_grad (the final encoder LayerNorm, which
 * those outputs do not pass, gets zeros).  Same workspace, pairing and limits as slsmi_train_forward /
 * slsmi_train_backward; any head configuration. */
int slsmi_train_backbone_forward(slsmi_ctx* ctx, const float* wav, int batch, int length, void* workspace,
                                 size_t workspace_bytes, float* layers_out, void* stream);
int slsmi_train_backbone_backward(slsmi_ctx* ctx, const float* wav, int batch, int length, void* workspace,
                                  size_t workspace_bytes, const float* dlayers, void* stream);

/* The whole training step of the upstream SLS model, Model(args, device) (README.md:46; main.py:201-258): the backbone
 * as above and, behind it, the SLS head in training form -- getAttenF's per-layer time means and stacked layer
 * outputs (model_backup.py:186-202), sigmoid(fc0) layer weights, the weighted layer sum, BatchNorm2d(1) on BATCH
 * statistics, SELU, max_pool2d(3, 3), fc1, SELU, fc3, SELU, log-softmax -- in fp32 kernels, forward and backward.
 * Contexts created with SLSMI_HEAD_SLS only.  fc1_weight: the caller's fp32 fc1.weight [1024][(frames / 3) * 341]
 * on the device (the context itself keeps that matrix only in the scoring forward's 16-bit form); the other head
 * parameters are the context's loaded copies.  logp_out [batch][2].  bn_batch_out: NULL or [2], the batch mean and
 * the unbiased batch variance nn.BatchNorm2d folds into running_mean / running_var with its momentum (the caller
 * owns those buffers).  A layer skipped by layerdrop (slsmi_train_set_masks) contributes no layer result.
 * slsmi_train_sls_backward: dlogp [batch][2] = dLoss/dlogp; writes the gradient of every registered parameter
 * (slsmi_set_grad): the backbone's, and fc0.weight, fc0.bias, first_bn.weight, first_bn.bias, fc1.weight, fc1.bias,
 * fc3.weight, fc3.bias.  Same workspace (slsmi_train_workspace_bytes), pairing and limits as
 * slsmi_train_forward / slsmi_train_backward. */
int slsmi_train_sls_forward(slsmi_ctx* ctx, const float* wav, int batch, int length, void* workspace,
                            size_t workspace_bytes, const float* fc1_weight, float* logp_out, float* bn_batch_out,
                            void* stream);
int slsmi_train_sls_backward(slsmi_ctx* ctx, const float* wav, int batch, int length, void* workspace,
                             size_t workspace_bytes, const float* fc1_weight, const float* dlogp, void* stream);

const char* slsmi_last_error(const slsmi_ctx* ctx);
int slsmi_destroy(slsmi_ctx* ctx);

/* ---- audio ingest (host threads; no GPU work) -------------------------------------
 * Replaces librosa.load(path, sr=16000) + pad(X, 64600) of the reference's Dataset classes
 * (data_utils_SSL.py:58-65, :109-115, :130-135) for WAV (8/16/24/32-bit PCM, 32/64-bit
 * float) and FLAC files: samples are scaled like libsndfile's float read (integer PCM of
 * b bits divided by 2^(b-1)), channels are averaged.  A file whose rate differs from
 * `expect_rate` is refused with SLSMI_ERR_AUDIO_RATE (expect_rate > 0), taken as it is (0)
 * or converted to -expect_rate by slsmi_audio_resample (< 0): see slsmi_audio_load_batch. */

/* Header facts of one file; any output pointer may be NULL. */
int slsmi_audio_info(const char* path, int* sample_rate, int* channels, int64_t* frames);

/* Decodes one file to mono float32: writes min(frames, capacity) samples to `out`, the full
 * frame count to *frames_out. */
int slsmi_audio_read(const char* path, float* out, int64_t capacity, int64_t* frames_out,
                     int* sample_rate);

/* Sample-rate conversion of a mono clip, the step librosa.load(path, sr=16000) performs on a file at another
 * rate (data_utils_SSL.py:112, :132): band-limited (Kaiser-windowed sinc, 64 zero crossings) interpolation to
 * ceil(n * dst_rate / src_rate) samples, of which min(that, capacity) are written; *n_out gets the full count.
 * librosa's resampler is a third-party library that is not part of the reference tree: the output agrees with
 * an ideal band-limited conversion (tested on tones), not bit for bit with librosa's. */
int slsmi_audio_resample(const float* in, int64_t n, int src_rate, int dst_rate, float* out, int64_t capacity,
                         int64_t* n_out);

/* Decodes `n` files on `threads` host threads and writes clip i, cropped to `cut` samples or
 * repeated until it fills them (pad(), data_utils_SSL.py:58-65), to out[i * cut ..]; `out` is
 * host memory (pinned, for the copy to the device that follows).  status (NULL or [n]) receives
 * a code per file; a failed file leaves zeros.  Returns SLSMI_OK only if every file decoded.
 * expect_rate > 0: a file at another rate fails with SLSMI_ERR_AUDIO_RATE; 0: any rate, as is;
 * < 0: files at another rate are converted to -expect_rate (slsmi_audio_resample) before the crop. */
int slsmi_audio_load_batch(const char* const* paths, int n, int64_t cut, int expect_rate, float* out,
                           int threads, int* status);

/* ---- single-kernel entry points (used by the parity tests; the forward is
 * built from exactly these kernels) ------------------------------------------ */

/* Every entry point below takes `fmt` = SLSMI_FMT_BF16 | SLSMI_FMT_F16 for its
 * 16-bit inputs/outputs (raw words). */

/* C[m][n] (op)= sum_k A[m][k] * W[n][k] + bias[n]; A, W raw 16-bit words.
 * mode: 0 bf16 out, 1 bf16 gelu, 2 fp32 out, 3 fp32 += , 4 fp32 += gelu,
 * 5 fp32 relu, 6 fp32 atomic split-K (bias ignored; C must be zeroed).
 * bn: 0 lets the library pick its kernel; 128 or 64: the two-stage 128-row kernel with that tile width;
 * 1: the 256 x 128 three-stage kernel; 256: the 208 x 256 kernel (N % 256 == 0, K % 128 == 0, K >= 256).
 * All of them add the K-tiles in the same order onto the bias: the result does not depend on the choice.
 * Replaces the nn.Linear / Conv1d contractions of wav2vec2.py:595-596,
 * :803-813, :1045-1062. */
int slsmi_gemm_bf16(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw,
                    const float* bias, void* C, int64_t ldc, int M, int N, int K, int mode,
                    int bn, int splitk, int fmt, void* stream);

/* slsmi_gemm_bf16 for callers that cannot allocate (no counterpart in the reference: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  slsmi_gemm_bf16 cuts a GEMM of few output
 * tiles along K and keeps the partial sums in a 64 MiB scratch the library allocates on first use; this
 * entry point computes the same result (same split, same bits) with the partial sums in a 16 MiB array
 * that is part of the library's code object (every GPU the library runs on holds its own copy until the
 * process ends), and calls no hipMalloc / hipFree.  A GEMM whose split would need more than 16 MiB
 * (slices x M x N x 4 bytes) is refused with SLSMI_ERR_WORKSPACE and a message on stderr; GEMMs that are
 * not split use no scratch and have no size limit.  Every call on a GPU shares that GPU's array: calls
 * that overlap in time on one GPU (two streams, two threads) race.  Nothing in the array has to be
 * cleared between calls (every slice is written whole before it is read). */
int slsmi_gemm_bf16_static(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw,
                           const float* bias, void* C, int64_t ldc, int M, int N, int K, int mode,
                           int bn, int splitk, int fmt, void* stream);

/* slsmi_gemm_bf16 with an A operand that repeats along K with period `ka` (a multiple of 64 that divides K):
 * C[m][n] = sum over k < K of A[m][k mod ka] * W[n][k] + bias[n].  With W = [W_hi | W_lo], K = 2 ka, that is
 * A (W_hi + W_lo)^T in one launch -- the product with a weight kept in two 16-bit halves, which is how the scoring
 * forward runs the conv layers named by slsmi_set_conv_precision (Conv1d of wav2vec2.py:803-813).  Same modes,
 * tile choices and K-tile order as slsmi_gemm_bf16. */
int slsmi_gemm_bf16_period(const uint16_t* A, int64_t lda, int ka, const uint16_t* W, int64_t ldw,
                           const float* bias, void* C, int64_t ldc, int M, int N, int K, int mode,
                           int bn, int fmt, void* stream);

/* Measurement helper (no counterpart in the reference): launches the same
 * GEMM `reps` times on `stream` between two HIP events recorded on that
 * stream, waits for the second event and stores the elapsed milliseconds for
 * all `reps` launches in *ms_out (host pointer).  Synchronises; not for use
 * inside a serving loop. */
int slsmi_time_gemm_bf16(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw,
                         const float* bias, void* C, int64_t ldc, int M, int N, int K, int mode,
                         int bn, int splitk, int fmt, int reps, float* ms_out, void* stream);

/* Diagnostic with no counterpart in the reference: the 16x16x32 MFMA rate (TFLOP/s, dense) the current device
 * sustains from registers alone -- two waves per SIMD issuing back-to-back MFMAs, `iters` x 16 per wave, timed by
 * HIP events on `stream` after a warm-up launch.  sink: device scratch of at least 512 x CUs floats.
 * Synchronises.  bench.py reports it beside the roofline fraction: it is the bound of a GEMM K loop on that box. */
int slsmi_debug_mfma_rate(float* sink, int64_t sink_floats, int fmt, int iters, double* tflops_out, void* stream);

/* The optimizer step of the training recipe (main.py:526 builds torch.optim.Adam(lr, weight_decay); main.py:238
 * calls optimizer.step()) over ALL parameter tensors in one launch.  Arithmetic of torch.optim.Adam (no amsgrad,
 * L2 weight decay added to the gradient) in fp32:
 *   g' = g + weight_decay p;  m += (1 - beta1)(g' - m);  v = beta2 v + (1 - beta2) g'^2;
 *   p -= lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps)
 * table: device array of n_tensors x 6 64-bit words {p, g, m, v (device addresses of fp32 arrays), n (elements),
 * step (the tensor's own step count, from 1)}: torch keeps one step count per parameter, and a parameter without a
 * gradient in some step (a layer skipped by layerdrop, wav2vec2.py:941-942) is left out of that step's table and
 * falls behind the others.  The hyper-parameters are doubles and every constant derived from them (1 - beta,
 * lr / (1 - beta1^step), 1 / sqrt(1 - beta2^step)) is formed in double and rounded once, as torch forms them.
 * items: device array of n_items x {tensor index, chunk index} (32-bit each): one workgroup per item updates
 * elements chunk * chunk_elems .. of that tensor (chunk_elems a multiple of 4; the caller lists every chunk of
 * every tensor once).  Asynchronous on `stream`; p, m, v are updated in place, g is read. */
int slsmi_adam_step(const int64_t* table, const int32_t* items, int64_t n_items, int chunk_elems, double lr, double beta1,
                    double beta2, double eps, double weight_decay, void* stream);

/* Diagnostic with no counterpart in the reference: times the MFMA GEMM launches of the library calls made between
 * begin and end where they run -- inside a forward, operands coming from the kernels before them -- instead of
 * shape by shape in isolation.  While the window is open every GEMM launch (of any context of the process, on the
 * device that was current at begin) is bracketed by two HIP events recorded on the stream the kernel is launched
 * on; at most max_launches are timed, further ones run untimed and are counted in *dropped.  end closes the
 * window, waits for the recorded events (synchronises) and stores the number of launches timed, the sum of their
 * durations in seconds and the sum of their 2 M N K.  One window at a time; the events cost a few microseconds of
 * stream time each, so nothing timed as a whole should run inside a window. */
int slsmi_debug_gemm_timing_begin(int max_launches);
int slsmi_debug_gemm_timing_end(int* launches, int* dropped, double* seconds, double* flops);

/* First conv block, wav2vec2.py:803-813 (in_d = 1): wav [B][L] fp32 ->
 * out bf16 [B][rows_per_utt][512], frames t < T written. */
int slsmi_conv0_ln_gelu(const float* wav, int B, int L, int T, const float* w, const float* bias,
                        const float* gamma, const float* beta, float eps, uint16_t* out,
                        int64_t rows_per_utt, int fmt, void* stream);

/* Positional conv of the encoder on packed input, wav2vec2.py:915-917 (Conv1d k = 128, groups = 16, SamePad,
 * GELU, residual): x fp32 [B * T][1024] += gelu(conv(xg) + bias).  xg 16-bit [B][16][slab_rows][64] with frame t
 * of a clip at slab row 64 + t and zero rows around the frames (slab_rows >= T + 127); w 16-bit
 * [1024][128][64] (output channel, tap, input channel of the group; weight norm applied); bias [1024].
 * form 0: batched MFMA GEMM over the Toeplitz view (any T); form 1: the group's slab resident in LDS, 208 frames of a clip at a time (slab_rows a multiple of 8).
 * Both forms give the same bits. */
int slsmi_posconv(const uint16_t* xg, const uint16_t* w, const float* bias, float* x, int B, int T,
                  int64_t slab_rows, int fmt, int form, void* stream);

/* The weight-gradient contraction of the training step, for parity tests: c_z[n][k] = alpha * sum over m < M of
 * a_z[m][n] * b_z[m][k] for z < nz (matrices za / zb / zc elements apart) -- dW = dY^T X of a Linear layer
 * (what autograd computes for main.py:236, loss.backward()) straight from the row-major output gradient a
 * (bfloat16 [M][lda]) and the row-major saved input b (format b_fmt: 0 bfloat16, 1 half; [M][ldb]); c fp32 [N][ldc].
 * N and K multiples of 128; lda, ldb multiples of 8 (ldb may be smaller than K: rows of b that overlap, the view a
 * strided Conv1d takes of its input); pointers 16-byte aligned. */
int slsmi_gemm_tn(const uint16_t* a, int64_t lda, const uint16_t* b, int64_t ldb, int b_fmt, float* c, int64_t ldc,
                  int64_t M, int N, int K, int nz, int64_t za, int64_t zb, int64_t zc, float alpha, void* stream);

/* The same with the M rows in groups of group_rows whose first rows lie ga / gb elements apart in a / b (row m is row
 * m % group_rows of group m / group_rows), and with N a multiple of 64 when K is a multiple of 256: the weight
 * gradient of the encoder's grouped positional conv (wav2vec2.py:866-874 under loss.backward(), main.py:236) --
 * per group, a = the 64 columns of the output gradient, b = the overlapping-row (Toeplitz) view of the group's
 * padded input slab, one group of rows per clip -- without transposed operand copies. */
int slsmi_gemm_tn_grouped(const uint16_t* a, int64_t lda, const uint16_t* b, int64_t ldb, int b_fmt, float* c,
                          int64_t ldc, int64_t M, int N, int K, int nz, int64_t za, int64_t zb, int64_t zc, float alpha,
                          int group_rows, int64_t ga, int64_t gb, void* stream);

/* LayerNorm over 512 bf16 channels (+ optional erf-GELU), wav2vec2.py:807-812, :563-564. */
int slsmi_ln512_bf16(const uint16_t* in, uint16_t* out, int64_t rows, const float* gamma,
                     const float* beta, float eps, int gelu, int fmt, void* stream);

/* The same from fp32 rows: a conv block whose output reaches its LayerNorm unrounded, as the reference's
 * Fp32LayerNorm sees it (wav2vec2.py:807-812; slsmi_set_conv_precision). */
int slsmi_ln512_f32(const float* in, uint16_t* out, int64_t rows, const float* gamma, const float* beta, float eps,
                    int gelu, int fmt, void* stream);
/* LayerNorm over 512 16-bit channels without GELU (wav2vec2.py:563-564), every row written as [hi | lo | hi]
 * (1536 values: the rounded result, the remainder of that rounding, the rounded result again): the operand of
 * post_extract_proj (wav2vec2.py:595-596) against weights laid out [W_hi | W_hi | W_lo].  out != in. */
int slsmi_ln512_split3(const uint16_t* in, uint16_t* out, int64_t rows, const float* gamma, const float* beta,
                       float eps, int fmt, void* stream);

/* LayerNorm over 1024 fp32 channels -> bf16, wav2vec2.py:1045, :1057. */
int slsmi_ln1024_f32_bf16(const float* in, uint16_t* out, int64_t rows, const float* gamma,
                          const float* beta, float eps, int fmt, void* stream);

/* LayerNorm(1024) folded into the two GEMMs around it (the pre-LN encoder layer, wav2vec2.py:1045-1046 and
 * :1057-1059: LayerNorm -> Linear).  Three pieces; the scoring forward is built from them.
 * (1) slsmi_gemm_bf16_ln, ln_mode 1 ("emit"): the residual GEMM C = resid + A W^T + bias (N = 1024) that
 * produces the row x a LayerNorm will read.  Beside C it writes, with (m, s) = ln_proxy[row] (a mean and a
 * reciprocal deviation chosen before x is known: those of the row one sub-layer earlier),
 *   ln_a[row][n] = (x[n] - m) s ln_vec[n]  (16 bits; ln_vec = the LayerNorm's weight), the next GEMM's operand;
 *   ln_slots[j][row] = (sum u, sum u^2) over columns 64 j .. 64 j + 63 of u = (x - m) s     ([16][M][2] floats).
 * (2) slsmi_gemm_bf16_ln, ln_mode 2 ("apply"): the GEMM behind the LayerNorm, A = that ln_a, K = 1024,
 * mode 0 or 1.  From ln_slots and the same ln_proxy it takes ubar = mean u and alpha = 1 / sqrt(var u + eps s^2)
 * and computes C = act(alpha (A W^T - ubar g) + c) with g = ln_vec and c = bias as slsmi_ln_fold_consts makes
 * them: exactly Linear(LayerNorm(x)), because alpha (u - ubar) = (x - mean x) / sqrt(var x + eps).  ln_stat
 * (may be NULL) receives x's exact (mean, 1 / sqrt(var + eps)) per row: the ln_proxy of the next emit.
 * dot_w / dot_out as in slsmi_gemm_resid_rowdot (emit only; may be NULL).  Both run on the 208 x 256 kernel
 * only; apply needs a workgroup per 208 rows (M <= 208 x the CU count).
 * (3) slsmi_ln_fold_prep starts a chain where no such GEMM precedes the LayerNorm: ln_a, ln_slots and
 * ln_proxy = the row's own (mean, 1 / sqrt(var + eps)) from x fp32 [rows][1024]. */
int slsmi_gemm_bf16_ln(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, const float* bias,
                       const float* resid, void* C, int64_t ldc, int M, int N, int K, int mode, int fmt, int ln_mode,
                       const float* ln_proxy, const float* ln_vec, uint16_t* ln_a, float* ln_slots, float* ln_stat,
                       float ln_eps, const float* dot_w, float* dot_out, void* stream);
int slsmi_ln_fold_prep(const float* x, uint16_t* ln_a, float* ln_proxy, float* ln_slots, int64_t rows,
                       const float* gamma, float eps, int fmt, void* stream);
/* The row-wise form of the same fold, for shapes the 208 x 256 kernel is a poor fit for (small batches):
 * slsmi_ln_fold_rows does per row what an emit GEMM and an apply GEMM's prologue do between them -- x fp32
 * [rows][1024], (m, s) = proxy_in[row] (NULL: the row's own statistics) -> ln_a, aff[row] = (alpha, -ubar) and
 * stat[row] = x's exact (mean, 1 / sqrt(var + eps)) -- and slsmi_gemm_bf16_ln_rows is the apply GEMM given aff
 * (mode 0 or 1, K = 1024; bn 0, 128, 64 or 1 as in slsmi_gemm_bf16; never cut along K).  Same arithmetic as the
 * other form up to the order of two fp32 additions: a score does not depend on which form its batch took. */
int slsmi_ln_fold_rows(const float* x, const float* proxy_in, uint16_t* ln_a, float* aff, float* stat, int64_t rows,
                       const float* gamma, float eps, int fmt, void* stream);
int slsmi_gemm_bf16_ln_rows(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, const float* c,
                            const float* g, const float* aff, void* C, int64_t ldc, int M, int N, int K, int mode,
                            int bn, int fmt, void* stream);
/* g[n] = sum_k gamma[k] W[n][k], c[n] = bias[n] + sum_k beta[k] W[n][k] for a 16-bit W [N][1024]. */
int slsmi_ln_fold_consts(const uint16_t* W, const float* bias, const float* gamma, const float* beta, float* g,
                         float* c, int N, int fmt, void* stream);

/* Attention core of fairseq MultiheadAttention (call site wav2vec2.py:1046-1052):
 * qkv bf16 [B*T][3072] (q pre-scaled by 1/8) -> out bf16 [B*T][1024]. */
int slsmi_attention(const uint16_t* qkv, uint16_t* out, int B, int T, int fmt, void* stream);
/* The same with a key-padding mask (wav2vec2.py:1050): utterance b attends to its first
 * nframes[b] (device int32 [B], clamped to [1, T]) keys; T is the row pitch. */
int slsmi_attention_varlen(const uint16_t* qkv, uint16_t* out, int B, int T, const int32_t* nframes, int fmt,
                           void* stream);

/* The SLS weighted layer sum (getAttenF, model_backup.py:186-202, + upstream
 * fc0 / sigmoid / sum over layers / BatchNorm2d(1) / SELU / max_pool2d(3,3)), in
 * three pieces.
 * (1) The residual GEMM C = resid + A W^T + bias (mode 3 of slsmi_gemm_bf16; the
 * FFN's second linear, wav2vec2.py:1059-1062) that also writes the row dots of
 * the new C with dot_w (= fc0.weight): dot_out[s][m] = sum over columns
 * 64 s .. 64 s + 63 of C[m][n] * dot_w[n]; dot_out is [N / 64][M] floats.
 * resid: the old values, same layout as C; NULL or == C: in place. */
int slsmi_gemm_resid_rowdot(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw,
                            const float* bias, const float* resid, float* C, int64_t ldc, int M, int N,
                            int K, int bn, int fmt, const float* dot_w, float* dot_out, void* stream);
/* (2) The layer weights w[l][b] = sigmoid(fc0_b + mean_t sum_s dots_l[s][b T + t])
 * for l < layers; dots_l = dots + l * layer_stride ([slices][B T] floats each);
 * w is [layers][B]. */
int slsmi_sls_weights(const float* dots, int64_t layer_stride, int layers, int slices, const float* fc0_b,
                      float* w, int B, int T, void* stream);
/* (3) f = sum_l w[l][b] * x_l (x_l = xs + l * layer_stride, fp32 [B][T][1024]),
 * then BatchNorm2d(1) eval affine + SELU + max_pool2d(3,3) -> 16-bit [B][ldp],
 * columns past (T / 3) * 341 zeroed. */
int slsmi_sls_sum_pool(const float* xs, int64_t layer_stride, int layers, const float* w, uint16_t* pooled,
                       int64_t ldp, int B, int T, float bn_scale, float bn_shift, int fmt, void* stream);

/* Per-row top-k of the SAE activations, in place (model.py:73-77). */
int slsmi_topk_rows(float* enc, uint16_t* enc_bf16, int64_t rows, int dict, int k, int fmt, void* stream);


/* Window TopK of the SAE activations, in place (AutoEncoderTopK._window_topk,
 * model_window_topk.py:118-203): enc fp32 [B][T][dict], non-negative; windows of `window`
 * frames every max(1, window / 2) frames keep the k features with the largest activation
 * sums; a frame's vote for a feature is its activation times the number of covering windows
 * that kept it; every frame keeps the activations of its k largest votes.  Where a frame has
 * fewer than k positive votes the reference's topk also returns zero-vote features and leaves
 * which ones to its implementation; this library takes the lowest-numbered ones.
 * Needs T >= window >= 2 (the reference's unfold fails below that), dict in {256 .. 8192,
 * powers of two}; scratch: device memory of slsmi_window_topk_scratch_bytes() bytes. */
int slsmi_window_topk(float* enc, uint16_t* enc_bf16, void* scratch, size_t scratch_bytes, int B, int T,
                      int dict, int k, int window, int fmt, void* stream);
size_t slsmi_window_topk_scratch_bytes(int B, int T, int dict, int window);

#ifdef __cplusplus
}
#endif
#endif /* SLSMI_H_ */
