// Host-side launcher declarations for the gfx950 kernels of libslsmi.
// Every launcher only enqueues work on the given stream: no allocation, no
// synchronisation.  All pointers are device pointers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace slsmi {

typedef uint16_t bf16_t;  // one raw 16-bit operand word (bf16 or f16 by `fmt`)

// Epilogue of the bf16 MFMA GEMM  C[m][n] (op)= sum_k A[m][k] * W[n][k] (+ bias[n])
enum GemmMode {
    GEMM_BF16 = 0,            // C bf16  = acc + bias
    GEMM_BF16_GELU = 1,       // C bf16  = gelu(acc + bias)
    GEMM_F32 = 2,             // C fp32  = acc + bias
    GEMM_F32_RESID = 3,       // C fp32  = resid + acc + bias  (residual stream; in place by default)
    GEMM_F32_RESID_GELU = 4,  // C fp32 += gelu(acc + bias)    (positional conv)
    GEMM_F32_RELU = 5,        // C fp32  = relu(acc + bias)    (SAE encoder)
    GEMM_F32_ATOMIC = 6,      // C fp32  atomically += acc     (split-K partial sums, no bias)
};

struct GemmArgs {
    const bf16_t* A = nullptr;  // [M][K] row-major, row stride lda elements (rows may overlap: lda < K is fine)
    long lda = 0;
    const bf16_t* W = nullptr;  // [N][K] row-major, row stride ldw
    long ldw = 0;
    const float* bias = nullptr;  // [N] or null
    void* C = nullptr;            // [M][N] row stride ldc (bf16 or fp32 by mode)
    long ldc = 0;
    int M = 0, N = 0, K = 0;      // N % bn == 0, K % 64 == 0
    // A repeats along K with this period (0: it does not; else a multiple of 64 that divides K): column k of the
    // product reads A[m][k mod ka].  With W = [W_hi | W_lo] (the two 16-bit halves of a weight, ldw = K = 2 ka) one
    // launch computes A W_hi^T + A W_lo^T, K-tiles in ascending order on every kernel.
    int ka = 0;
    // The contraction length of the layer this launch computes, for FLOP accounting (0: K).  A launch that multiplies
    // split operands (K = 2 ka or 3 x the layer's width) does more matrix work than the layer's 2 M N k_algo.
    int k_algo = 0;
    int mode = GEMM_BF16;
    int bn = 0;                   // block tile width: 128 or 64 (two-stage kernels); 0 = library picks
                                  // (256 x 128 three-stage kernel for large plain GEMMs)
    int fmt = 0;                  // 16-bit operand/storage format: 0 bf16, 1 f16
    // batch over blockIdx.z: A += z*zA; W += (z % zWmod)*zW; bias += (z % zWmod)*zBias;
    // C += (z / zCdiv)*zC1 + (z % zCdiv)*zC2   (element offsets)
    int nz = 1;
    long zA = 0;
    int zWmod = 1;
    long zW = 0, zBias = 0;
    int zCdiv = 1;
    long zC1 = 0, zC2 = 0;
    // split-K over blockIdx.z (nz must be 1): GEMM_F32_ATOMIC adds every part into one C; GEMM_F32
    // writes part z to C + z * zC1 (bias in part 0), for a consumer that sums the parts in a fixed order
    int splitk = 1;
    // row compaction: when grp_in > 0, input row m belongs to group m / grp_in at
    // position m % grp_in; it is stored at output row group*grp_out + position,
    // and only if position < grp_valid.
    int grp_in = 0, grp_valid = 0, grp_out = 0;
    // row dots (GEMM_F32_RESID only): when dot_w != null, also writes
    // dot_out[s][m] = sum_{n in 64 s .. 64 s + 63} C_new[m][n] * dot_w[n]   (dot_out is [N / 64][M])
    const float* dot_w = nullptr;
    float* dot_out = nullptr;
    // GEMM_BF16_GELU: also store the 16-bit value before the GELU here (row stride ldc), for a backward pass that needs
    // both.  128-row kernel only: the launcher keeps such a GEMM off the persistent kernels and off the K split.
    bf16_t* pre_out = nullptr;
    // GEMM_F32 / GEMM_F32_RESID with split_ws: where the launcher cuts K into S > 1 slices on its own, it leaves the
    // slices' partial sums in split_ws ([S][M][N] fp32, the bias in slice 0) and does NOT run the summing kernel --
    // the caller's next kernel adds them (in slice order, then the residual) as it reads them.  *deferred_slices = S;
    // 1: K was not cut and C has been written as usual.
    int* deferred_slices = nullptr;
    // residual modes: C = resid + (...) with the old values read from `resid` (same strides and batch
    // offsets as C) instead of from C itself; null = in place
    const float* resid = nullptr;
    // LayerNorm folded into the two GEMMs around it (N = 1024 on the producer, K = 1024 on the consumer; 208 x 256
    // kernel only, any other shape is refused).  The residual GEMM that produces a row x of the stream
    // (GEMM_F32_RESID, ln_mode = LN_EMIT) also writes a = (x - m) s gamma in 16 bits to ln_a, the consumer's A
    // operand, where (m, s) = ln_proxy[row] is a mean and reciprocal deviation known BEFORE x is (the exact ones of
    // the stream's previous LayerNorm input, one sub-layer back: close enough to keep a in the 16-bit range), and
    // per 64-column slice the sums of u = (x - m) s and of u^2 to ln_slots[slice][row].  The consumer (GEMM_BF16 or
    // GEMM_BF16_GELU, ln_mode = LN_APPLY, A = that a, same ln_proxy and ln_slots) derives from the sums, with
    // ubar = mean(u), alpha = 1 / sqrt(var(u) + eps s^2), and computes
    //     C = act(alpha (a W^T - ubar g) + c),   g[n] = sum_k gamma[k] W[n][k],  c[n] = bias[n] + sum_k beta[k] W[n][k]
    // (g = ln_vec, c = bias; both in fp32 from the 16-bit W), which is LayerNorm(x) W^T + bias with x's exact
    // statistics: alpha (u - ubar) = (x - mean x) / sqrt(var x + eps).  It also writes x's exact
    // (mean, 1 / sqrt(var + eps)) to ln_stat[row], the next producer's ln_proxy.
    enum { LN_NONE = 0, LN_EMIT = 1, LN_APPLY = 2 };
    int ln_mode = LN_NONE;
    const float* ln_proxy = nullptr;  // [M][2]
    const float* ln_vec = nullptr;    // LN_EMIT: gamma [N]; LN_APPLY: g [N]
    bf16_t* ln_a = nullptr;           // LN_EMIT: [M][N] out, row stride N
    float* ln_slots = nullptr;        // [16][M][2]: LN_EMIT out, LN_APPLY in
    float* ln_stat = nullptr;         // LN_APPLY: [M][2] out, or null
    float ln_eps = 0.f;
    // LN_APPLY with the rows' factors already known (launch_ln1024_fold_rows): ln_aff [M][2] = (alpha, -ubar) instead
    // of ln_proxy / ln_slots.  This form runs on the 128-row and 256 x 128 kernels (never cut along K), so it serves
    // the shapes the 208 x 256 kernel is a poor fit for; the arithmetic is the same up to the order of two fp32
    // additions.
    const float* ln_aff = nullptr;
    // Scratch for the library's own split of K (null: never split).  A GEMM with few rows has few output
    // tiles -- M = 201 (one clip) gives 16 for N = 1024 -- and would leave most CUs idle while a few walk the
    // whole K.  With a scratch the launcher cuts K into S slices (S * tiles ~ the CU count), every slice
    // writes its partial sums to scratch[z][M][N], and a second kernel adds the slices in index order and
    // applies the mode's epilogue (bias, activation, residual, 16-bit output, row dots): same bits every run.
    // Launches that share a scratch must not overlap in time.
    float* split_ws = nullptr;
    size_t split_ws_bytes = 0;
};
hipError_t launch_gemm(const GemmArgs& a, hipStream_t s);
// Whether an encoder layer's GEMMs (N = 1024 / 3072 / ffn over M rows) are best served by the 208 x 256 kernel with
// the LayerNorms folded into them (LN_EMIT / LN_APPLY): the library itself would pick that kernel for each of them,
// and every tile row of 208 gets a workgroup of its own.  Otherwise the row-wise form (launch_ln1024_fold_rows +
// GemmArgs::ln_aff) computes the same thing on the other kernels.
// any_batch: only the second condition.
bool gemm_ln_fold_fits(int M, int ffn, bool any_batch = false);
constexpr size_t GEMM_SPLIT_WS_BYTES = 64u << 20;  // what the contexts and the single-kernel entry points provide
// Bytes of K-split scratch launch_gemm would use for `a` if it were given a scratch of `limit` bytes
// (0: that GEMM is not cut along K).  a.split_ws is ignored.
size_t gemm_split_scratch_bytes(const GemmArgs& a, size_t limit);

// One Adam step (torch.optim.Adam's single-tensor arithmetic, L2 weight decay) over a table of fp32 tensors in one
// launch.  tab: n_tensors x {p, g, m, v, n, step} (64-bit words, device memory; step = the tensor's own step count,
// from 1); items: n_items x {tensor, chunk}, one workgroup each, covering elements chunk * chunk_elems ... of the
// tensor (chunk_elems a multiple of 4).
hipError_t launch_adam(const long long* tab, const int2* items, long n_items, int chunk_elems, double lr, double beta1,
                       double beta2, double eps, double weight_decay, hipStream_t s);

// Timing window over launch_gemm (diagnostic): between begin and end every launch_gemm call of the process is
// bracketed by two HIP events on its stream (at most max_launches of them; further ones run untimed and are
// counted in *dropped).  end waits for the recorded events and returns the launches timed, the sum of their
// durations and of their 2 M N K nz.  One window at a time.
hipError_t gemm_timing_begin(int max_launches);
hipError_t gemm_timing_end(int* launches, int* dropped, double* seconds, double* flops);

// conv0: Conv1d(1,512,k=10,s=5,bias) + LayerNorm(512 over channels) + erf-GELU.
// wav fp32 [B][L] -> out bf16 [B][out_rows_per_utt][512] (rows t < T written).
hipError_t launch_conv0_ln_gelu(const float* wav, int B, int L, int T, const float* w /*[512][10]*/,
                                const float* bias, const float* gamma, const float* beta, float eps,
                                bf16_t* out, long out_rows_per_utt, int fmt, hipStream_t s);

// LayerNorm over 512 channels of bf16 rows, optional erf-GELU, in place or to out.
hipError_t launch_ln512_bf16(const bf16_t* in, bf16_t* out, long rows, const float* gamma,
                             const float* beta, float eps, int gelu, int fmt, hipStream_t s);
// LayerNorm over 512 channels without GELU, rows written as [hi | lo | hi] (1536 values); out != in.
hipError_t launch_ln512_split3(const bf16_t* in, bf16_t* out, long rows, const float* gamma, const float* beta, float eps,
                               int fmt, hipStream_t s);
// The same from fp32 rows (a conv output kept unrounded up to its LayerNorm).
hipError_t launch_ln512_f32(const float* in, bf16_t* out, long rows, const float* gamma, const float* beta, float eps,
                            int gelu, int fmt, hipStream_t s);
// LayerNorm over 1024 channels of fp32 rows -> bf16 rows (GEMM A operand).
hipError_t launch_ln1024_f32_bf16(const float* in, bf16_t* out, long rows, const float* gamma,
                                  const float* beta, float eps, int fmt, hipStream_t s);
// The same on rows that are still a GEMM's K slices (GemmArgs::deferred_slices): x[r] = part[0][r] + ... + part[S-1][r]
// (index order) + resid[r] (nullable) is written to x (fp32) and normalised into out; same bits as the summing kernel
// followed by launch_ln1024_f32_bf16.
hipError_t launch_ln1024_sum_f32_bf16(const float* part, int slices, long slice, const float* resid, float* x, bf16_t* out,
                                      long rows, const float* gamma, const float* beta, float eps, int fmt, hipStream_t s);
// Final encoder LayerNorm: fp32 rows -> fp32 out (may be null) and, when
// sub != null, bf16 (y - sub[c]) rows for the SAE encoder.
hipError_t launch_ln1024_f32_f32(const float* in, float* out, bf16_t* out_sub, const float* sub,
                                 long rows, const float* gamma, const float* beta, float eps, int fmt,
                                 hipStream_t s);

// Entry of a chain of folded LayerNorms (GemmArgs::ln_mode): stands in for an LN_EMIT producer with the row's own
// statistics as (m, s).  in fp32 [rows][1024] -> a_out 16-bit [rows][1024], proxy [rows][2], slots [16][rows][2].
hipError_t launch_ln1024_fold_prep(const float* in, bf16_t* a_out, float* proxy, float* slots, long rows,
                                   const float* gamma, float eps, int fmt, hipStream_t s);
// The row-wise form of the fold, for GEMMs that run on the other kernels: x fp32 [rows][1024], (m, s) = proxy_in[row]
// (null: the row's own statistics) -> a_out 16-bit, aff [rows][2] = (alpha, -ubar) for an LN_APPLY GEMM given
// ln_aff, stat [rows][2] = x's exact (mean, 1 / sqrt(var + eps)): the next call's proxy_in.
hipError_t launch_ln1024_fold_rows(const float* in, const float* proxy_in, bf16_t* a_out, float* aff, float* stat, long rows,
                                   const float* gamma, float eps, int fmt, hipStream_t s);
// g[n] = sum_k gamma[k] W[n][k], c[n] = bias[n] + sum_k beta[k] W[n][k] for a 16-bit W [N][1024] (bias may be null):
// the vectors an LN_APPLY GEMM takes as ln_vec and bias.
hipError_t launch_ln_fold_consts(const bf16_t* W, const float* bias, const float* gamma, const float* beta, float* g_out,
                                 float* c_out, int N, int fmt, hipStream_t s);

// Positional-conv input: x fp32 [B*T][1024] -> xg bf16 [B][16][rows_per_slab][64] with 64
// zero rows before and zero rows after the T real rows.
// nframes (null or [B]): frames t >= nframes[b] are packed as zeros.
hipError_t launch_posconv_pack(const float* x, bf16_t* xg, int B, int T, int rows_per_slab, int fmt,
                               hipStream_t s, const int* nframes = nullptr);

// Diagnostic (diag.hip): range of a 16-bit buffer [rows][cols] (row stride ld): out2[0] = max |value| as float bits
// (merged by unsigned maximum), out2[1] += values at or beyond the format's largest finite value.  cols % 8 == 0.
// grp_in > 0: only rows at position < grp_valid inside each group of grp_in rows are looked at.
hipError_t launch_range_stat(const bf16_t* x, long rows, int cols, long ld, int fmt, unsigned* out2, hipStream_t s,
                             int grp_in = 0, int grp_valid = 0);
// Diagnostic (diag.hip): back-to-back 16x16x32 MFMAs from registers on every SIMD; sink: blocks x 256 floats
hipError_t launch_mfma_rate(float* sink, int blocks, int iters, int fmt, hipStream_t s);

// The positional conv with a (clip, group)'s packed input resident in LDS (posconv.hip): x += gelu(conv(xg) + bias)
// for clips of at most 208 frames; w = the repacked weights [16 * 64][128 taps][64], the same bits as the batched
// GEMM form (N = 64, K = 8192, lda = 64, GEMM_F32_RESID_GELU) on the same operands.
bool posconv_slab_supported(int T, int slab_rows);
hipError_t launch_posconv_slab(const bf16_t* xg, const bf16_t* w, const float* bias, float* x, int B, int T, int slab_rows,
                               int fmt, hipStream_t s);

// Frames per utterance from sample counts: nframes[b] = conv-stack output length of lengths[b]
// samples, clamped to [1, T].
struct ConvGeometry {
    int n = 0;
    int kernel[8] = {0}, stride[8] = {0};
};
hipError_t launch_frames_from_lengths(const int* lengths, int* nframes, int B, const ConvGeometry& geo, int T,
                                      hipStream_t s);

// Self-attention core for 16 heads of width 64 (q pre-scaled):
// qkv bf16 [B*T][3072] -> out bf16 [B*T][1024].  T <= 512.  nframes (null or [B]): utterance b
// attends to its first nframes[b] keys only.
hipError_t launch_attention(const bf16_t* qkv, bf16_t* out, int B, int T, int fmt, hipStream_t s,
                            const int* nframes = nullptr);
// The same with MultiheadAttention's dropout on the probabilities (training forward, T <= 256): keep
// [B * 16][T][pitch] bytes (1 = kept), pitch a multiple of 32 and >= T; kept probabilities are scaled by drop_scale.
hipError_t launch_attention_dropout(const bf16_t* qkv, bf16_t* out, int B, int T, int fmt, const uint8_t* keep, int pitch,
                                    float drop_scale, hipStream_t s);

// SLS head -----------------------------------------------------------------
// w[l][b] = sigmoid(fc0_b + mean_t sum_s dots_l[s][b T + t]), l < layers, from the row dots of launch_gemm
// (dot_out); dots_l = dots + l * layer_stride is [slices][B T]; w is [layers][B]
hipError_t launch_sls_weights(const float* dots, long layer_stride, int layers, const float* fc0_b, float* w, int B,
                              int T, int slices, hipStream_t s);
// weighted layer sum + BN(eval) affine + SELU + max_pool2d(3,3) in one pass:
// x_l = xs + l * layer_stride fp32 [B][T][1024], w [layers][B] -> pooled bf16 [.][ldp] (columns past
// (T / 3) * 341 zeroed)
// bn4 (device, nullable): the BatchNorm's weight, bias, running mean and running variance; when given, the
// affine is computed from them on the device (eps = bn_eps) and bn_scale / bn_shift are ignored
hipError_t launch_sls_sum_pool(const float* xs, long layer_stride, int layers, const float* w, bf16_t* pooled, long ldp,
                               int B, int T, float bn_scale, float bn_shift, int fmt, hipStream_t s,
                               const float* bn4 = nullptr, float bn_eps = 0.f, long seg = 0);
// (seg > 0: a pooled row is [hi | lo | hi], three segments of seg values with ldp >= 3 seg -- the rounded feature, the
// remainder of that rounding, the rounded feature again: the A operand for fc1 weights laid out [W_hi | W_hi | W_lo])
// selu(sum of the `parts` partial fc1 outputs (part_stride elements apart) + b1) -> fc3 -> selu ->
// log_softmax : -> logp [B][2]
hipError_t launch_sls_tail(const float* fc1out, int parts, long part_stride, const float* b1, const float* w3,
                           const float* b3, float* logp, int B, hipStream_t s);

// SAE head -----------------------------------------------------------------
// keep the k largest entries of every fp32 row (others -> 0), in place; also
// writes a bf16 copy when enc_bf16 != null.
hipError_t launch_topk_rows(float* enc, bf16_t* enc_bf16, long rows, int dict, int k, int fmt, hipStream_t s);
// Window TopK (model_window_topk.py:118-203), in place on enc fp32 [B][T][dict] (non-negative): windows of
// `window` frames every max(1, window / 2) frames keep their k largest activation sums; a frame keeps
// the activations of its k largest votes (activation x number of covering windows that kept the
// feature), zero-vote features filling up from the lowest feature number.  keep: scratch of
// window_topk_scratch_bytes().  T >= window >= 2.
hipError_t launch_window_topk(float* enc, bf16_t* enc_bf16, uint8_t* keep, int B, int T, int dict, int k, int window,
                              int fmt, hipStream_t s);
long window_topk_scratch_bytes(int B, int T, int dict, int window);
// pooled[b][j] = mean_t enc[b*T+t][j] (+ add[j] when add != null); with nframes the mean runs over
// the first nframes[b] frames of utterance b
hipError_t launch_col_mean(const float* enc, const float* add, float* pooled, int B, int T, int dict, hipStream_t s,
                           const int* nframes = nullptr);
// LayerNorm(dict) -> Linear(dict,256)+ReLU -> Linear(256,2) -> log_softmax; hidden: scratch [B][256] floats;
// dict a multiple of 256
// mask (training): null or [B][256] multipliers applied to the hidden layer (the Dropout of model.py:187)
hipError_t launch_sae_tail(const float* pooled, const float* ln_g, const float* ln_b, const float* w1,
                           const float* b1, const float* w2, const float* b2, float* hidden, float* logp, int B,
                           int dict, hipStream_t s, const float* mask = nullptr);
// out[0] = mean((recon + b_dec - x)^2) over rows*1024 elements; `partial` is scratch of MSE_PARTS floats
constexpr int MSE_PARTS = 1024;
hipError_t launch_mse(const float* recon, const float* b_dec, const float* x, long rows, float* partial, float* out,
                      hipStream_t s);

// Training path: gradient kernels ------------------------------------------------------------
// out[c][r] = in[r * ld + c] (r < R, c < C) as bfloat16, input of format src_fmt; out row stride ldo >= R,
// columns R .. ldo - 1 zeroed (a K dimension padded for the GEMM).
// tb: rows at position >= grp_valid of every grp_in rows read as zeros; nz > 1: entry z = z1 * zdiv + z2 reads
// at in + z1 * in1 + z2 * in2 and writes its R rows at out + z1 * out1 + z2 * out2 (no zero padding then).
struct TransposeBatch {
    int grp_in = 0, grp_valid = 0;
    int nz = 1, zdiv = 1;
    long in1 = 0, in2 = 0, out1 = 0, out2 = 0;
};
hipError_t launch_transpose16(const bf16_t* in, long ld, long R, int C, bf16_t* out, long ldo, int src_fmt,
                              hipStream_t s, const TransposeBatch& tb = TransposeBatch());
// fp32 in[R][C] -> bfloat16 copy[R][C] and outT[c][r] (row stride ldo >= R, zeros for R <= r < ldo)
hipError_t launch_cvt_transpose(const float* in, long R, int C, bf16_t* copy, bf16_t* outT, long ldo, hipStream_t s);
// a weight matrix for the training step: fp32 in[R][C] * scale -> copy[R][C] (format copy_fmt) and its bfloat16
// transpose outT[c * ldo + r]
hipError_t launch_cvt_transpose_w(const float* in, long R, int C, float scale, int copy_fmt, bf16_t* copy, bf16_t* outT,
                                  long ldo, hipStream_t s);
// ... and many of them: matrices of equal shape share launches of up to CVT_W_BATCH
constexpr int CVT_W_BATCH = 64;
struct CvtTransposeW {
    const float* src;
    bf16_t* copy;
    bf16_t* outT;
    long ldo;
    int R, C;
    float scale;
};
hipError_t launch_cvt_transpose_w_batch(const CvtTransposeW* items, int count, int copy_fmt, hipStream_t s);
// out[n] (+)= scale * sum_m in[m][n]; in fp32 (in_fmt < 0) or 16-bit of format in_fmt; partial: scratch of
// colsum_parts(M) * N floats
constexpr int COLSUM_MAX_PARTS = 256;
int colsum_parts(long M);
hipError_t launch_colsum(const void* in, int in_fmt, long ld, long M, int N, float scale, int accumulate,
                         float* partial, float* out, hipStream_t s);
// Column sums of bfloat16 matrices that lie z_in elements apart, one per layer: dst[l][n] = scale * sum over m < M of
// in[l z_in + m ld + n], n < N (N % 64 == 0), added in a fixed order; layers with a null dst are skipped.
constexpr int COLSUM_LAYERS_PER_LAUNCH = 48;
hipError_t launch_colsum_layers(const bf16_t* in, long z_in, long ld, long M, int N, float scale, float* const* dst,
                                int layers, hipStream_t s);
// Weight gradients without transposed operand copies: C_z[n][k] = alpha * sum over m < M of A_z[m][n] B_z[m][k],
// z < nz (operands and results zA / zB / zC elements apart); A bfloat16 [M][lda], B of format b_fmt [M][ldb], C fp32
// [N][ldc]; N and K multiples of 128, or N a multiple of 64 with K a multiple of 256 (64 columns of A per workgroup).
// grp > 0: the M rows come in groups of grp whose first rows lie gA / gB elements apart (row m = row m % grp of group
// m / grp): clips of a batch whose rows are not one arithmetic sequence.
// The weight gradient of `count` tensors of one shape [N][K] (batch entries: operand slices zsrc apart by zA / zB) with
// the Adam step taken in the epilogue instead of a stored gradient (backward.hip, TnAdamTable): p, m, v dense fp32
// [N][K], 16-byte aligned; step = the tensor's step count of THIS step (>= 1).
constexpr int TN_ADAM_BATCH = 32;
// copy / outT (both or neither): 16-bit copies of the new values that the same epilogue writes -- copy[N][K] =
// pack(p * copy_scale) in format copy_fmt, outT[K][N] with row pitch ldo = that value as bfloat16 (what
// launch_cvt_transpose_w makes of the tensor: a weight reload then has nothing to do for it).
struct TnAdamEntry {
    float* p;
    float* m;
    float* v;
    long long step;
    int zsrc;
    bf16_t* copy;
    bf16_t* outT;
    float* db;   // null, or where alpha * the column sums of the entry's A go (the bias gradient beside the weight gradient)
};
hipError_t launch_gemm_tn_adam(const bf16_t* A, long lda, long zA, const bf16_t* B, long ldb, long zB, int b_fmt, long M, int N,
                               int K, float alpha, const TnAdamEntry* entries, int count, double lr, double beta1, double beta2,
                               double eps, double weight_decay, long ldo, float copy_scale, int copy_fmt, hipStream_t s);
hipError_t launch_gemm_tn(const bf16_t* A, long lda, long zA, const bf16_t* B, long ldb, long zB, int b_fmt, float* C,
                          long ldc, long zC, long M, int N, int K, int nz, float alpha, hipStream_t s, int grp = 0,
                          long gA = 0, long gB = 0, float* const* db = nullptr);
// (db: nz pointers, null entries allowed -- alpha * the column sums of A_z go to db[z][0 .. N): the bias gradient beside
// the weight gradient, from the rows the kernel stages anyway; 128-column form, no row groups, nz <= TN_ADAM_BATCH)
// g[n] (bfloat16, in place) *= gelu'(pre[n]), pre of format pre_fmt; n % 8 == 0
hipError_t launch_gelu_bwd16(const bf16_t* pre, int pre_fmt, bf16_t* g, long n, hipStream_t s);
// three column segments of width seg (seg % 64 == 0) of one matrix at once: out[i][n] = scale[i] * sum_m in[m][i seg + n]
// (a null out[i] is skipped)
hipError_t launch_colsum3(const void* in, int in_fmt, long ld, long M, int seg, float* const out[3], const float scale[3],
                          float* partial, hipStream_t s);
// Backward of y = LayerNorm_C(x) (and of a = gelu(y) when gelu): dy fp32 [rows][C] is the gradient at y (at a);
// out = resid + resid2 + dL/dx as fp32 (both nullable; copy16, nullable: its bfloat16 rounding as well) or dL/dx as
// bfloat16 (out16); dgamma / dbeta [C] (nullable);
// part: scratch of ln_bwd_groups(rows) * 2 * C floats.  x: fp32 (x_fmt < 0) or 16-bit of format x_fmt.
// Supported: C = 1024 fp32 x, fp32 out; C = 512 16-bit x with (gelu, out16) or (no gelu, fp32 out).
// grp_in > 0: rows at position >= grp_valid inside each group of grp_in rows give zero and count for nothing.
constexpr int LN_BWD_MAX_GROUPS = 1024;
int ln_bwd_groups(long rows);
// dy_slices > 1 (C = 1024, fp32 x): dy is the sum of that many arrays dy_slice floats apart, added in index order (a
// GEMM's K slices, GemmArgs::deferred_slices)
hipError_t launch_ln_bwd(int C, const void* x, int x_fmt, const float* dy, const float* gamma, const float* beta,
                         float eps, int gelu, const float* resid, void* out, int out16, float* part, float* dgamma,
                         float* dbeta, long rows, int grp_in, int grp_valid, hipStream_t s, const float* resid2 = nullptr,
                         bf16_t* copy16 = nullptr, int dy_slices = 1, long dy_slice = 0);
// The parameter gradients of `count` LayerNorms from the partial sums their launch_ln_bwd calls left (dgamma = dbeta =
// nullptr there): part + z * z_part is [groups][2 C]; its column sums go to dgamma[z] (first C) and dbeta[z] (null
// entries: not wanted), added in group order.  One launch per 64 LayerNorms.
hipError_t launch_ln_part_sums(const float* part, long z_part, int groups, int C, float* const* dgamma, float* const* dbeta,
                               int count, hipStream_t s);
// out = gelu(in), 16-bit rows of format fmt; n % 8 == 0
hipError_t launch_gelu16(const bf16_t* in, bf16_t* out, long n, int fmt, hipStream_t s);
// g[R][C] (bfloat16, in place) *= gelu'(pre), pre of format pre_fmt; the result is also written transposed to
// outT[c][r] (row stride ldo >= R, zeros for R <= r < ldo)
hipError_t launch_gelu_bwd16_transpose(const bf16_t* pre, int pre_fmt, bf16_t* g, long R, int C, bf16_t* outT, long ldo,
                                       hipStream_t s);
hipError_t launch_scale_f32(float* x, long n, float scale, hipStream_t s);
// Backward of launch_attention: qkv, o (forward output) of format fmt, dout bfloat16 [B*T][1024] ->
// dqkv bfloat16 [B*T][3072] (gradient at the pre-scaled q, at k and at v).  T <= 256.
// keep_qk / keep_kq (both or neither): the dropout of launch_attention_dropout, [B * 16][query][key] and
// [B * 16][key][query] bytes with row pitch `pitch`.
hipError_t launch_attention_bwd(const bf16_t* qkv, const bf16_t* o, const bf16_t* dout, bf16_t* dqkv, int B, int T,
                                int fmt, hipStream_t s, const uint8_t* keep_qk = nullptr, const uint8_t* keep_kq = nullptr,
                                int pitch = 0, float drop_scale = 1.f);

// Backward of the in-tree classifier (LayerNorm(dict) -> Linear(dict, 256) -> ReLU -> dropout mask -> Linear(256, 2)
// -> log_softmax): dlogp, logp [B][2]; pooled [B][dict]; hidden [B][256] (ReLU output); mask null or [B][256].
// Scratch: stats [B][2], dh1 [B][256], dhb [B][dict].  Outputs (each nullable except dpool): dpool [B][dict] and
// the parameter gradients.  B <= 64, dict % 256 == 0.
hipError_t launch_cls_bwd(const float* dlogp, const float* logp, const float* pooled, const float* hidden,
                          const float* mask, const float* ln_g, const float* ln_b, const float* w1, const float* w2,
                          int B, int dict, float* stats, float* dh1, float* dhb, float* dpool, float* dln_g,
                          float* dln_b, float* dw1, float* db1, float* dw2, float* db2, hipStream_t s);
// denc (bfloat16 [M][dict]) = code != 0 ? dpool[m / T] / T + dcode : 0   (dpool, dcode nullable)
hipError_t launch_sae_code_grad(const float* code, const float* dpool, const float* dcode, bf16_t* denc, long M, int T,
                                int dict, hipStream_t s);
// drecon (bfloat16) = dloss[0] * 2 (recon + b_dec - x) / (rows * 1024);  dx = -drecon (fp32)
hipError_t launch_mse_grad(const float* recon, const float* b_dec, const float* x, const float* dloss, long rows,
                           bf16_t* drecon, float* dx, hipStream_t s);
// The same with the reconstruction as the classifier's input: drecon / total = dpool[m / T] / T (+ the loss's share,
// dloss nullable) at recon + b_dec, as bfloat16 and as fp32; dx = minus the loss's share
hipError_t launch_recon_grad(const float* recon, const float* b_dec, const float* x, const float* dloss, const float* dpool,
                             long rows, int T, bf16_t* drecon, float* dx, float* total, hipStream_t s);
// out[m][c] = dpool[m / T][c] / T over rows of 1024 (back through a time mean)
hipError_t launch_spread_rows(const float* dpool, long rows, int T, float* out, hipStream_t s);
hipError_t launch_add_f32(float* a, const float* b, long n, hipStream_t s);
// out = (resid ? resid : 0) + a * m (out may be a or resid); x *= m on 16-bit values of format fmt
hipError_t launch_mul_f32(float* out, const float* a, const float* m, const float* resid, long n, hipStream_t s);
hipError_t launch_mul16_f32(bf16_t* x, const float* m, long n, int fmt, hipStream_t s);
hipError_t launch_cvt_bf16(const float* in, bf16_t* out, long n, hipStream_t s);
// out = x + gelu(u)
hipError_t launch_add_gelu(const float* x, const float* u, float* out, long n, hipStream_t s);
// du = dx o gelu'(u): compact bfloat16 [B T][1024] and packed [B][16][rows_per_slab][64] (frame t at slab row 64 + t)
hipError_t launch_posconv_grad_pack(const float* dx, const float* u, bf16_t* du, bf16_t* dug, int B, int T,
                                    int rows_per_slab, hipStream_t s);
// wf[g][ci][127 - j][co] = w[g][co][j][ci] as bfloat16
hipError_t launch_posconv_flip(const bf16_t* w, int fmt, bf16_t* wf, hipStream_t s);
// weight_norm(dim=2) backward from the gradient at the folded weight [1024 co][128 j][64 ci];
// scratch: 1024 * 256 + 256 floats
hipError_t launch_weightnorm_bwd(const float* dwf, const float* v, const float* g, float* dv, float* dg, float* scratch,
                                 hipStream_t s);

// compact fp32 rows [B T][C] -> padded rows [B P][C], zeros past T
hipError_t launch_expand_rows(const float* in, float* out, int B, int T, int P, int C, hipStream_t s);
// da (fp32 [B Pin][512]) from dcol (bfloat16 [B Pout][k 512]) of a conv layer run as a flat GEMM; Pin = stride Pout
hipError_t launch_col2im(const bf16_t* dcol, float* da, int B, int Pin, int Pout, int k, int stride, hipStream_t s);
// [co][k][ci] -> [co][ci][k]
hipError_t launch_conv_grad_unpack(const float* packed, float* dst, int co, int ci, int k, int slices, hipStream_t s);
// Backward of launch_conv0_ln_gelu: da fp32 [B][P][512] -> sums[6656] = dW [512][10], dbias, dgamma, dbeta [512 each];
// part: scratch of conv0_bwd_groups(B T) * 6656 floats
int conv0_bwd_groups(long frames);
hipError_t launch_conv0_bwd(const float* wav, int B, int L, int T, long P, const float* w, const float* bias,
                            const float* gamma, const float* beta, float eps, const float* da, float* part,
                            float* sums, hipStream_t s);

// Weight repacking -----------------------------------------------------------
hipError_t launch_cvt_f32_bf16(const float* src, bf16_t* dst, long n, float scale, int fmt, hipStream_t s);
// [rows][cols] fp32 -> [rows][ld] bf16, zero padded columns cols..ld-1
hipError_t launch_cvt_pad_rows(const float* src, bf16_t* dst, long rows, long cols, long ld, int fmt,
                               hipStream_t s);
// [rows][cols] fp32 -> [rows][3 ld]: [hi | hi | lo], every segment zero padded to ld columns
hipError_t launch_cvt_split3_rows(const float* src, bf16_t* dst, long rows, long cols, long ld, int fmt, hipStream_t s);
// Conv1d weight [co][ci][k] fp32 -> bf16 [co][k][ci]
hipError_t launch_repack_conv_w(const float* src, bf16_t* dst, int co, int ci, int k, int fmt, hipStream_t s);
// ... -> bf16 [co][2][k][ci]: the rounded weight and, k * ci further on, the remainder of that rounding
hipError_t launch_repack_conv_w_split(const float* src, bf16_t* dst, int co, int ci, int k, int fmt, hipStream_t s);
// weight-normed grouped conv: v [1024][64][128], g [128] -> bf16 [16 groups][64 co][128 j][64 ci]
// (norm_scratch: 128 + 256 * 128 floats; the first 128 are the norms afterwards)
hipError_t launch_repack_posconv(const float* v, const float* g, bf16_t* dst, float* norm_scratch, int fmt,
                                 hipStream_t s);
hipError_t launch_scale_copy_f32(const float* src, float* dst, long n, float scale, hipStream_t s);
// up to SCALE_COPY_BATCH such copies in one launch (one workgroup per copy)
constexpr int SCALE_COPY_BATCH = 128;
struct ScaleCopy {
    const float* src;
    float* dst;
    int n;
    float scale;
};
hipError_t launch_scale_copy_batch(const ScaleCopy* items, int count, hipStream_t s);

// ---- the upstream SLS head in training form (sls_train.hip).  Layer outputs x_l are [rows][1024] fp32, layer l at
// xs + l * layer_stride; SlsKept lists the layers the head reads (layerdrop appends no result for a skipped layer).
struct SlsKept {
    int n;
    int idx[64];
};
int sls_fc1_chunks();  // row chunks of fc1's weight in launch_sls_fc1_bwd: dpart is [chunks][B][P]
hipError_t launch_sls_layer_means(const float* xs, long layer_stride, const SlsKept& kept, int B, int T, float* y0, hipStream_t s);
hipError_t launch_sls_layer_weights(const float* y0, const float* fc0_w, const float* fc0_b, int pairs, float* wl, hipStream_t s);
hipError_t launch_sls_weighted_sum(const float* xs, long layer_stride, const SlsKept& kept, const float* wl, int B, int T,
                                   float* full, float* part, hipStream_t s);
hipError_t launch_sls_bn_stats(const float* part, long rows, float eps, float* stat, float* batch_out, hipStream_t s);
hipError_t launch_sls_bn_pool(const float* full, const float* stat, const float* bn, int B, int T, float* pooled, uint8_t* amax,
                              hipStream_t s);
hipError_t launch_sls_fc1_fwd(const float* pooled, const float* W1, const float* b1, int B, long P, float* h1, hipStream_t s);
hipError_t launch_sls_tail_fwd(const float* h1, const float* w3, const float* b3, int B, float* a3, float* logp, hipStream_t s);
hipError_t launch_sls_tail_bwd(const float* dlogp, const float* logp, const float* a3, const float* h1, const float* w3, int B,
                               float* dh1, float* dW3, float* db3, float* db1, hipStream_t s);
hipError_t launch_sls_fc1_bwd(const float* dh1, const float* pooled, const float* W1, int B, long P, float* dW1, float* dpart,
                              hipStream_t s);
hipError_t launch_sls_pool_bwd(const float* dpart, const uint8_t* amax, const float* full, const float* stat, const float* bn,
                               int B, int T, float* dbn, float* part2, hipStream_t s);
hipError_t launch_sls_bn_bwd_stats(const float* part2, int n, long count, float* stat2, float* dgamma, float* dbeta, hipStream_t s);
hipError_t launch_sls_dfull(float* d, const float* full, const float* stat, const float* stat2, const float* bn, long rows,
                            hipStream_t s);
int sls_dweight_chunks(int T);  // frame chunks of launch_sls_dweights: dwl_part is [chunks][B][kept]
hipError_t launch_sls_dweights(const float* xs, long layer_stride, const SlsKept& kept, const float* dfull, int B, int T,
                               float* dwl_part, hipStream_t s);
hipError_t launch_sls_fc0_bwd(const float* dwl_part, int chunks, const float* wl, const float* y0, int pairs, float* dz,
                              float* dfc0_w, float* dfc0_b, hipStream_t s);
hipError_t launch_sls_dlayers(const float* dfull, const float* wl, const float* dz, const float* fc0_w, const SlsKept& kept,
                              int B, int T, float* dlayers, long dl_stride, hipStream_t s);

}  // namespace slsmi
