/*
 * libasrhip -- C ABI of the MI355X (gfx950) kernels behind the DFCNN(+SE)+CTC /
 * Transformer hot path of 786440445/ASR_DFCNN_Transformer.
 *
 * The reference has no FFI of its own: its hot path is the set of TensorFlow-1.x /
 * Keras / python_speech_features calls listed in SURVEY.md section 2.2.  Each entry
 * point below replaces one of those calls; the reference call site is cited as
 * file:line under the reference checkout.  INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add.
 *
 * Conventions (all entry points):
 *   - every pointer is CALLER-OWNED DEVICE memory unless marked "host"; the library
 *     never allocates, frees or synchronises;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*), asynchronously;
 *   - return value: 0 = enqueued, negative asr_status on a bad descriptor;
 *   - reductions run in a fixed order: repeated calls give bit-identical results;
 *   - activations are fp32, NHWC.  A "padded plane" is the layout
 *         float [B][H+1][W+1][C]   (interior at [h+1][w+1]; row 0 and column 0 are zero)
 *     i.e. ONE shared zero border: the right border of a row is the left border of the next
 *     row, the bottom border of an image is the top border of the next image.  It is preceded
 *     and followed by at least (W+3) zero guard pixels (the last image's bottom border lives
 *     in the tail guard); kernels only ever write the interior, so borders/guards stay zero
 *     once the buffer was zeroed.  `x` arguments of padded planes point at pixel 0 (after
 *     the leading guard).
 */
#ifndef ASR_HIP_H
#define ASR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* Every declaration below is an exported entry point; the library itself is built with -fvisibility=hidden, so its dynamic
 * symbol table holds these names and nothing else (tests/test_abi_cpu.py). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

typedef enum {
    ASR_OK = 0,
    ASR_ERR_BAD_ARG = -1,      /* shape/stride/alignment the kernels do not support */
    ASR_ERR_LAUNCH = -2,       /* hipLaunch reported an error                        */
    ASR_ERR_UNSUPPORTED = -3
} asr_status;

int asr_version(void);
/* Name of the contraction kernel instantiation the calling thread's last asr_tap_gemm / asr_tap_gemm_pw /
 * asr_tap_gemm_wino / asr_tap_wgrad call enqueued, spelled as rocprofv3 --kernel-trace prints it
 * without the namespace (e.g. "tap_gemm_kernel_v5<128, 64, 2, 2, 9, 16, 3, 3, 0>"); "" before the first such call.
 * bench.py keys its HIP-event timings by it, so that `roofline.kernel` always names the row of the rocprof summary
 * the launcher's dispatch rules really picked.  The pointer stays valid for the life of the library. */
const char* asr_last_kernel(void);
/* Last HIP error string seen by the library on this thread (host pointer, static). */
const char* asr_last_error(void);

/* ------------------------------------------------------------------ K1 fbank
 * Replaces python_speech_features.logfbank + sklearn.preprocessing.scale as called by
 * compute_fbank_from_api (util/wav_util.py:22-31).  float64 arithmetic on device.
 *   signal   [B][max_samples] f32, utterance b uses nsamples[b] samples
 *   nsamples [B] i32 (device)
 *   fb_start/fb_count [nfilt] i32, fb_weight [nfilt][fb_width] f64: the banded mel
 *            filterbank over the nfft/2+1 power-spectrum bins: filter j covers bins
 *            fb_start[j] .. fb_start[j]+fb_count[j]-1 (host code builds them)
 *   twiddle  [nfft/2][2] f64: cos/sin(-2*pi*k/nfft)
 *   logfb    [B][max_frames][nfilt] f64 workspace (log filterbank energies)
 *   out      [B][t_pad][nfilt] f32: standardised features, rows >= frames(b) are zero
 *   frames   [B] i32 out: number of frames of each utterance
 */
int asr_fbank(const float* signal, const int32_t* nsamples, int B, int max_samples,
              int frame_len, int frame_step, int nfft, double preemph, int nfilt,
              const int32_t* fb_start, const int32_t* fb_count, const double* fb_weight, int fb_width,
              const double* twiddle, double* logfb, int max_frames,
              float* out, int t_pad, int32_t* frames, void* stream);

/* ------------------------------------------------------------------ K1b low-frame-rate stacking
 * build_LFR_features(inputs, m, n) (util/utils.py:7-31, called per utterance by get_transformer_batch,
 * end2end/data_loader.py:284) on the padded feature batch asr_fbank wrote, for the whole batch at once:
 *   feat   [B][t_pad][D] f32, frames [B] i32 (device; valid rows of each utterance)
 *   out    [B][t_out][m*D] f32: out[b][i] = rows i*n .. i*n+m-1 of utterance b side by side, rows past its end
 *          replaced by its LAST row (:25-29), for i < ceil(frames[b] / n); all-zero rows after that
 *          (wav_padding, end2end/data_loader.py:84-99).  D % 4 == 0.
 */
int asr_lfr(const float* feat, const int32_t* frames, int B, int t_pad, int D, int m, int n, int t_out,
            float* out, void* stream);

/* ------------------------------------------------------------------ K2/K6 tap-GEMM
 * One kernel family serves conv3x3/conv1x1 forward, their data-gradient, and dense
 * layers (tf.layers.conv2d / tf.layers.dense: lm_and_am/model/acoustic_model2.py:102-121):
 *
 *   acc[m][n] = sum_tap sum_k  A[m + off(tap)][k] * Wt(tap,k,n)
 *   v = acc + bias[n];  if relu: v = max(v,0);          out_a[row_a(m)][n] = v
 *   y = scale[n]*v + shift[n];  out_y[row_y(m)][n] (+)= y
 *
 * ntaps = 9: A is a padded plane [B][H+1][W+1][K] (m = padded pixel index,
 *            off = dh*(W+1)+dw), rows at border positions are not written;
 * ntaps = 1: A is a plain [M][K] matrix when H == 0 (dense), or a padded plane
 *            (1x1 conv) when H > 0.
 * wmode 0: Wt(tap,k,n) = W[(tap*K + k)*ldw + n]          (HWIO as stored by the model)
 * wmode 1: Wt(tap,k,n) = W[((ntaps-1-tap)*N + n)*ldw + k] (data-gradient: transposed,
 *          taps mirrored) -- no separate packed copy of the weights is needed.
 * y_unpadded: out_y rows are [B][H][W] (feeds the dense head) instead of padded.
 * Requirements: K % 4 == 0, N % 4 == 0, lda/ldw/ldo % 4 == 0, 16-byte aligned bases.
 * Arithmetic: fp32 MFMA (v_mfma_f32_32x32x2_f32), i.e. exact fp32 FMA chains.
 */
typedef struct {
    int M;            /* rows: padded pixel count B*(H+1)*(W+1), or matrix rows     */
    int K, N;
    int lda, ldw;     /* row pitches of A and W in floats                          */
    int ldo_a, ldo_y; /* row pitches of out_a / out_y                              */
    int ntaps;        /* 1, 9, or 4 (2x2 forward window: phase-split stride-2 conv)  */
    int B, H, W;      /* geometry of the padded plane (H == 0: plain matrix)       */
    int wmode;        /* 0 | 1                                                     */
    int relu;         /* activation after the bias: 0 none, 1 max(.,0), 2 tanh      */
    int accumulate;   /* out_y += instead of =                                     */
    int y_unpadded;   /* out_y is [B*H*W][ldo_y]                                   */
} asr_gemm_desc;

int asr_tap_gemm(const asr_gemm_desc* d, const float* A, const float* W,
                 const float* bias, const float* scale, const float* shift,
                 float* out_a, float* out_y, void* stream);

/* Weight gradient of the same family:
 *   dW[tap][k][n] = sum_m A[m + off(tap)][k] * dZ[m][n]
 * dZ must be zero at border pixels (it is, when produced by asr_cell_bwd_pre).
 * `partials` is a workspace of asr_tap_wgrad_workspace() bytes; the reduction over
 * pixel chunks is a second, fixed-order pass (deterministic).  */
size_t asr_tap_wgrad_workspace(const asr_gemm_desc* d);
int asr_tap_wgrad(const asr_gemm_desc* d, const float* A, const float* dZ, int ldz,
                  float* dW, float* partials, void* stream);
/* asr_tap_wgrad routes 3x3 layers with 32 or 64 k input channels, 64 n output channels, an even height and enough pixels to the
 * Winograd F(3x3, 2x2) kernel (csrc/wino_wgrad.hip: 16 instead of 36 multiplies per 2x2 tile and channel pair, fp32, fixed-order
 * reduction).  asr_tap_wgrad_direct never does: the direct kernels, kept as the independent implementation the tests compare with. */
int asr_tap_wgrad_direct(const asr_gemm_desc* d, const float* A, const float* dZ, int ldz,
                         float* dW, float* partials, void* stream);

/* ------------------------------------------------------------------ first cell (Cin = 1)
 * cnn_cell(32, wav_input, pool=True): conv3x3(1->C) + bias + ReLU + frozen BN + 2x2 pool
 * fused into one HBM pass (acoustic_model2.py:39,126-133; acoustic_model.py:39).
 *   x [B][T][F] f32 (unpadded, single channel);  w [9][C]; y padded plane [B][T/2][F/2][C]
 *   pool: 1 = average (acoustic_model2/3), 2 = max (acoustic_model, cnn_ctc)
 * Backward recomputes the pre-pool activations from x instead of storing 1.3 GB:
 *   dy padded plane (grad wrt y) -> grads[4][..] = dw[9][C], db[C], dgamma[C], dbeta[C]
 * (bn_scale = gamma/sqrt(moving_var+eps), bn_shift = beta - moving_mean*bn_scale;
 *  dgamma is returned w.r.t. bn_scale, the host divides by sqrt(var+eps).)
 */
int asr_cell1_fwd(const float* x, int B, int T, int F, int C, const float* w, const float* bias,
                  const float* bn_scale, const float* bn_shift, int pool, float* y, void* stream);
size_t asr_cell1_bwd_workspace(int B, int T, int F, int C);
int asr_cell1_bwd(const float* x, int B, int T, int F, int C, const float* w, const float* bias,
                  const float* bn_scale, const float* bn_shift, int pool, const float* dy,
                  float* dw, float* db, float* dscale, float* dshift, float* partials, void* stream);

/* ------------------------------------------------------------------ K3/K4 cell epilogue / prologue
 * pool + BN affine of a cell whose conv ran with asr_tap_gemm (out_a = post-ReLU):
 *   y[b][h/2][w/2][c] = pool2x2( bn_scale[c]*a + bn_shift[c] )      (padded planes)
 */
int asr_pool_fwd(const float* a, int B, int H, int W, int C, const float* bn_scale,
                 const float* bn_shift, int pool, float* y, void* stream);

/* Backward prologue of a cell: from dL/d(cell output) to dZ (grad at the conv output,
 * before bias/ReLU), plus the per-channel sums
 *   dshift = sum dy, dscale = sum dy*a, dbias = sum dz       (fixed-order reduction)
 * dy_layout: 0 = padded plane at the cell's own resolution, 1 = padded plane at the
 * pooled resolution (pool != 0), 2 = unpadded [B][H][W][C] (from the dense head).
 */
size_t asr_cell_bwd_pre_workspace(int B, int H, int W, int C);
int asr_cell_bwd_pre(const float* dy, int dy_layout, const float* a, int B, int H, int W, int C,
                     const float* bn_scale, const float* bn_shift, int pool,
                     float* dz, float* dscale, float* dshift, float* dbias,
                     float* partials, void* stream);

/* ------------------------------------------------------------------ K5 squeeze-excitation
 * squeeze_excitation_layer + residual add (acoustic_model2.py:41,141-148):
 *   xt = bn_scale*x + bn_shift; s = mean_hw(xt); e = sigmoid(W2 relu(W1 s + b1) + b2)
 *   out = main + xt * e
 * state (f32, caller-owned): s[B][C], r[B][hid], e[B][C]  (asr_se_state_floats)
 */
size_t asr_se_state_floats(int B, int C, int hid);
size_t asr_se_fwd_workspace(int B, int H, int W, int C);          /* bytes of `partials` */
int asr_se_fwd(const float* main_in, const float* x, int B, int H, int W, int C, int hid,
               const float* bn_scale, const float* bn_shift, const float* w1, const float* b1,
               const float* w2, const float* b2, float* state, float* partials, float* out,
               void* stream);
/* asr_se_fwd with the squeeze sums made by the launch that wrote x (asr_tap_gemm_wino_sums): sums [B][nsplit][C] partial rows. */
int asr_se_fwd_sums(const float* main_in, const float* x, int B, int H, int W, int C, int hid,
                    const float* bn_scale, const float* bn_shift, const float* w1, const float* b1,
                    const float* w2, const float* b2, float* state, const float* sums, int nsplit, float* out, void* stream);
/* dout = dL/d(out).  dL/d(main) IS dout (identity branch), so it is not produced here: the
 * caller lets later contributions accumulate into the dout buffer.  dx = dL/dx (=; plus dout
 * when add_dout, for the acoustic_model3 wiring where main == x), parameter grads (=). */
size_t asr_se_bwd_workspace(int B, int H, int W, int C, int hid);
int asr_se_bwd(const float* dout, const float* x, int B, int H, int W, int C, int hid,
               const float* bn_scale, const float* bn_shift, const float* w1, const float* w2,
               const float* state, int add_dout, float* dx, float* dscale, float* dshift,
               float* dw1, float* db1, float* dw2, float* db2, float* partials, void* stream);
/* The same with the backward prologue of the conv cell that produced x fused in (round 4): when x is the BN output of a cell that
 * nothing else reads (acoustic_model2.py:41,107-148: cell -> squeeze_excitation_layer), dx is that cell's complete output gradient, and
 * instead of writing it and running asr_cell_bwd_pre(pool 0) over it, the same pass applies the cell's BN and ReLU backward:
 * cell_dz (padded plane like x) = dL/d(conv + bias), cell_dscale / cell_dshift / cell_dbias [C] = its three channel sums; cell_a = the
 * cell's post-ReLU pre-BN activations, cell_scale its BN scale.  dx is not produced.  Blocks, slots and summation order are those of
 * the two calls it replaces: the same bits.  partials: asr_se_bwd_cell_workspace bytes. */
size_t asr_se_bwd_cell_workspace(int B, int H, int W, int C, int hid);
int asr_se_bwd_cell(const float* dout, const float* x, int B, int H, int W, int C, int hid,
                    const float* bn_scale, const float* bn_shift, const float* w1, const float* w2,
                    const float* state, int add_dout, float* dscale, float* dshift,
                    float* dw1, float* db1, float* dw2, float* db2,
                    const float* cell_a, const float* cell_scale, float* cell_dz, float* cell_dscale, float* cell_dshift,
                    float* cell_dbias, float* partials, void* stream);
/* asr_se_bwd_cell with that reduction handed in (xsums [B][nsplit][C] partial rows from asr_tap_gemm_wino_sesum). */
int asr_se_bwd_cell_sums(const float* dout, const float* x, int B, int H, int W, int C, int hid,
                         const float* bn_scale, const float* bn_shift, const float* w1, const float* w2,
                         const float* state, int add_dout, float* dscale, float* dshift,
                         float* dw1, float* db1, float* dw2, float* db2,
                         const float* cell_a, const float* cell_scale, float* cell_dz, float* cell_dscale, float* cell_dshift,
                         float* cell_dbias, const float* xsums, int nsplit, float* partials, void* stream);
/* dst (+)= src over the interior of a padded plane (residual gradient fan-in). */
int asr_axpy(float* dst, const float* src, size_t n, float alpha, int accumulate, void* stream);

/* ------------------------------------------------------------------ K7 softmax / log head
 * logits = log(transpose(softmax(d), [1,0,2]) + 1e-7)  (acoustic_model2.py:67-68)
 *   d [B][T][V] -> logits_tm [T][B][V];  backward: g_tm [T][B][V] -> dd [B][T][V] * gscale
 */
int asr_softmax_log_fwd(const float* d, int B, int T, int V, float eps, float* logits_tm, void* stream);
int asr_softmax_log_bwd(const float* logits_tm, const float* g_tm, int B, int T, int V, float eps,
                        float gscale, float* dd, void* stream);
/* dense + ReLU backward helper: dz = dy * (h > 0) where h is the post-ReLU output. */
int asr_relu_bwd(const float* dy, const float* h, size_t n, float* dz, void* stream);
/* dz = (h > 0) ? dy * scale : 0.  Backward of Dense(relu) followed by tf.layers.dropout (transformer.py:139-154) when the
 * forward dropped h out in place: a dropped element is 0 like an inactive one, so the mask needs no second draw and the
 * dropout backward is its 1 / (1 - rate) factor.  n % 4 == 0, 16-byte aligned pointers. */
int asr_relu_bwd_scaled(const float* dy, const float* h, size_t n, float scale, float* dz, void* stream);
/* column sums of a [rows][cols] matrix (bias gradients), fixed order. */
size_t asr_colsum_workspace(int rows, int cols);
int asr_colsum(const float* x, int rows, int cols, int ldx, float* out, float* partials, void* stream);

/* ------------------------------------------------------------------ K8 CTC loss + gradient
 * tf.nn.ctc_loss_v2(sparse_labels, logits, ..., blank_index=V-1) (acoustic_model2.py:79-80):
 * logits_tm are unnormalised; the op applies its own log-softmax.
 *   labels [B][max_label] i32 (already without zeros: see dense_to_sparse), label_len [B],
 *   seq_len [B];  loss [B] f32;  grad [T][B][V] f32 (zero for t >= seq_len[b]);
 *   status [B] i32: 0 ok, 1 = "not enough time for target transition sequence"
 *   (loss = +inf, grad = 0 for that utterance -- TF raises InvalidArgumentError),
 *   2 = padding row: seq_len[b] == 0 and label_len[b] == 0 mark a row that is not part of
 *   the batch (the reference feeds B' <= B surviving rows, lm_and_am/data_loader.py:149-156;
 *   a fixed-size device batch carries the missing rows as padding): loss = 0, grad = 0.
 * workspace: asr_ctc_workspace() bytes, 16-byte aligned.  alpha/beta recursion runs in float64.
 * Two launches since round 5 (was three): the row pass (log-sum-exp of every frame, the dense softmax term of the gradient, the
 * emission probabilities of the utterance's lattice states, and the status -- each row's wave checks its own utterance), then one
 * workgroup per utterance: probabilities into LDS when Tb x (S rounded up to 4, + 4) doubles fit the launch's 144 KB budget, one
 * wave each for alpha / beta (two states per lane up to 128 states), occupancies without a division (the beta wave stores
 * beta / p); utterances whose linear-domain lattice loses states are redone in the log domain inside the same launch.
 */
size_t asr_ctc_workspace(int T, int B, int max_label);
int asr_ctc_loss(const float* logits_tm, int T, int B, int V, const int32_t* labels, int max_label,
                 const int32_t* label_len, const int32_t* seq_len, int blank,
                 float* loss, float* grad, int32_t* status, void* workspace, void* stream);

/* ------------------------------------------------------------------ K9 greedy decode, K10 edit distance
 * tf.nn.ctc_greedy_decoder(logits, seq_len) (acoustic_model2.py:69): argmax (lowest index
 * on ties), merge repeats, drop blank = V-1.
 *   out_ids [B][T] i32 (-1 padded), out_len [B], neg_sum_logits [B] f32
 * tf.edit_distance(decoded, sparse_labels, normalize=True) (acoustic_model2.py:72):
 *   dist [B] f32 = levenshtein / len(truth)  (inf if truth empty and hyp not).
 */
size_t asr_ctc_greedy_workspace(int T, int B);
int asr_ctc_greedy(const float* logits_tm, int T, int B, int V, const int32_t* seq_len, int blank,
                   int32_t* out_ids, int32_t* out_len, float* neg_sum_logits, void* workspace, void* stream);
int asr_edit_distance(const int32_t* hyp, int hyp_pitch, const int32_t* hyp_len,
                      const int32_t* truth, int truth_pitch, const int32_t* truth_len,
                      int B, float* dist, void* stream);

/* ------------------------------------------------------------------ K11 Adam (TF form)
 * tf.train.AdamOptimizer (acoustic_model2.py:90): lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
 * theta -= lr_t * m / (sqrt(v) + eps).  One launch over the flat parameter buffer.
 * `gscale` multiplies the gradient first (1/world after a sum all-reduce).
 */
int asr_adam_tf(float* theta, const float* grad, float* m, float* v, size_t n,
                float lr_t, float beta1, float beta2, float eps, float gscale, void* stream);

/* ====================================================================== Transformer path
 * end2end/transformer.py (used by lm_and_am/model/language_model.py and end2end/model.py).
 * Projections / FFN / vocabulary GEMMs are asr_tap_gemm / asr_tap_wgrad with ntaps = 1.
 */

/* K13 scaled_dot_product_attention + head split/merge (transformer.py:89-115,144-151).
 *   Q [N][Tq][C], K, V [N][Tk][C]: the relu'd projections; C = H*64; head h = columns h*64..h*64+63
 *   scores = Q_h K_h^T / 8; key mask: keys whose per-head K row sums to 0 get -2^32+1;
 *   causal: keys > query get -2^32+1; softmax; rows whose per-head |Q| sum is 0 are zeroed
 *   (query mask, applied AFTER the softmax); O = P V_h merged back to [N][Tq][C].
 *   lse [2][N][H][Tq] (row max and log of the row sum of the masked scores, kept apart because
 *   the max can be the -2^32+1 fill value; both in base-2 units of the scaled scores, i.e. times
 *   log2(e): the kernels exponentiate with v_exp_f32) is saved for the backward -- an opaque carry.
 * Backward: dQ, dK, dV (=).  Masked scores receive no gradient (tf.where), V still does.
 *   relu_grad != 0: Q, K, V are outputs of Dense(activation=relu) (transformer.py:139-141) and the
 *   returned gradients are those of the pre-activations, i.e. dQ *= (Q > 0) etc., fused in the store.
 *   delta_ws: N*H*Tq floats.  Deterministic (no atomics: dK/dV and dQ are separate passes). */
/*   dropout_rate > 0: tf.layers.dropout on the attention weights (transformer.py:111) with the counter-based mask
 *   keep(element index ((n*H + h)*Tq + q)*Tk + k, seed); kept weights are scaled by 1/(1-rate).  The backward must be
 *   given the same rate and seed (the mask is regenerated, not stored).  N*H*Tq*Tk must stay below 2^32. */
int asr_attention_fwd(const float* Q, const float* K, const float* V, int N, int Tq, int Tk, int C, int H,
                      int causal, float dropout_rate, unsigned int seed, float* O, float* lse, void* stream);
int asr_attention_bwd(const float* Q, const float* K, const float* V, const float* O, const float* dO,
                      const float* lse, int N, int Tq, int Tk, int C, int H, int causal, int relu_grad,
                      float dropout_rate, unsigned int seed,
                      float* dQ, float* dK, float* dV, float* delta_ws, void* stream);
/* The same with row pitches: Q (and dQ) rows are ldq floats apart, K, V (and dK, dV) rows ldk floats apart (>= C, multiples
 * of 4); O, dO stay dense [.][C].  This is how the engines run the three projections of a block as ONE GEMM: Q, K and V are
 * column blocks of one [rows][3C] buffer (self-attention) or K, V of one [rows][2C] buffer (encoder-decoder attention). */
int asr_attention_fwd_p(const float* Q, const float* K, const float* V, int N, int Tq, int Tk, int C, int H,
                        int ldq, int ldk, int causal, float dropout_rate, unsigned int seed, float* O, float* lse,
                        void* stream);
int asr_attention_bwd_p(const float* Q, const float* K, const float* V, const float* O, const float* dO,
                        const float* lse, int N, int Tq, int Tk, int C, int H, int ldq, int ldk, int causal,
                        int relu_grad, float dropout_rate, unsigned int seed,
                        float* dQ, float* dK, float* dV, float* delta_ws, void* stream);
/* Round 5: the per-row statistics of an attention call taken ONCE.  asr_attention_stats writes, for every (sample, head), the query
 * mask of its Tq queries (1 / 0: the head slice of the Q row is not / is all zeros) and the key bias of its Tk keys (+inf / the fill value
 * in min() form: the head slice of the K row sums to something / to zero) -- stats = asr_attention_stats_floats(N, Tq, Tk, H) floats:
 * [N][H][Tq] masks, then [N][H][Tk] biases.  The _s entry points take them (stats may be NULL: then, as in the _p forms, every workgroup
 * recomputes them from Q / K -- 4 to 16 times per (sample, head), 6-12 % of the kernels' time); the values are the same, so are all results. */
size_t asr_attention_stats_floats(int N, int Tq, int Tk, int H);
int asr_attention_stats(const float* Q, const float* K, int N, int Tq, int Tk, int C, int H, int ldq, int ldk, float* stats, void* stream);
int asr_attention_fwd_s(const float* Q, const float* K, const float* V, int N, int Tq, int Tk, int C, int H,
                        int ldq, int ldk, int causal, float dropout_rate, unsigned int seed, float* O, float* lse,
                        const float* stats, void* stream);
int asr_attention_bwd_s(const float* Q, const float* K, const float* V, const float* O, const float* dO,
                        const float* lse, int N, int Tq, int Tk, int C, int H, int ldq, int ldk, int causal,
                        int relu_grad, float dropout_rate, unsigned int seed,
                        float* dQ, float* dK, float* dV, float* delta_ws, const float* stats, void* stream);
/* dst[r][0..cols) (+)= src[r][0..cols) for r < rows, with row pitches ldd / lds (floats; everything a multiple of 4):
 * packs separate weight matrices into the column blocks of a fused one and scatters its gradient back. */
int asr_copy2d(float* dst, int ldd, const float* src, int lds, int rows, int cols, int accumulate, void* stream);
/* The same for a table of copies in one launch (the [wq | wk | wv] packs of every attention block of a step, the three
 * gradient blocks of a packed weight gradient).  items_dev: n_items asr_copy2d_item in DEVICE memory, written once by the
 * caller; every item obeys asr_copy2d's rules (cols % 4 == 0, pitches % 4 == 0, 16-byte aligned pointers); max_elems =
 * the largest rows * cols among them (sizes the grid). */
typedef struct asr_copy2d_item { float* dst; const float* src; int ldd, lds, rows, cols; } asr_copy2d_item;
int asr_copy2d_batch(const asr_copy2d_item* items_dev, int n_items, int max_elems, int accumulate, void* stream);
/* Several transposes in one launch: dst[c][r] = src[r][c] for every item (rows / cols / lds describe src, ldd the pitch of dst,
 * no alignment rule).  The Transformer engines transpose every dense kernel once per step with it, so that the forward GEMMs
 * read both operands K-contiguous (asr_tap_gemm_nt).  max_elems = the largest rows * cols of the table. */
int asr_transpose_batch(const asr_copy2d_item* items_dev, int n_items, int max_elems, void* stream);
/* tf.layers.dense forward (end2end/transformer.py:130-136,211-222; language_model.py:44-52) as asr_tap_gemm with ntaps 1 /
 * wmode 0, given BOTH the kernel W [K][N] (pitch d->ldw) and its transposed copy Wt [N][K] (pitch ldwt): large problems run on
 * the LDS-DMA kernel of gemm1.hip, which wants both operands K-contiguous, the others fall through to asr_tap_gemm on W.
 * Same arithmetic per output element up to the order of the K sum inside 8-wide groups. */
int asr_tap_gemm_nt(const asr_gemm_desc* d, const float* A, const float* W, const float* Wt, int ldwt,
                    const float* bias, const float* scale, const float* shift, float* out_a, float* out_y, void* stream);
/* tf.layers.dropout(x, rate, training=True) (transformer.py:154,226; model.py:290; language_model.py:34):
 *   y[i] = keep(i, seed) ? x[i] / (1 - rate) : 0 with keep(i, seed) = (murmur3_fmix(i * 0x9E3779B1 + seed) >> 8) >= rate * 2^24.
 *   In place is allowed; the same call on the gradient is the backward.  TensorFlow's random stream is not reproduced. */
int asr_dropout(const float* x, size_t n, float rate, unsigned int seed, float* y, void* stream);

/* K15 layer_norm (transformer.py:4-27) fused with the residual add in front of it:
 *   x = a (+ b);  y = gamma*(x-mean)/sqrt(var+eps) + beta  (biased variance, eps 1e-8)
 *   xhat [rows][C] and rstd [rows] are saved for the backward.
 * Backward: dx (= or +=) -- the same dx is the gradient of both a and b. */
int asr_add_layernorm_fwd(const float* a, const float* b, const float* gamma, const float* beta, int rows, int C,
                          float eps, float* y, float* xhat, float* rstd, void* stream);
size_t asr_layernorm_bwd_workspace(int rows, int C);
int asr_layernorm_bwd(const float* dy, const float* xhat, const float* rstd, const float* gamma, int rows, int C,
                      float* dx, int accumulate, float* dgamma, float* dbeta, float* partials, void* stream);
/* The same pass with the consumers of dx fused in (the residual sub-layers of transformer.py:139-158, 211-232 in reverse):
 *   dx  (may be NULL)  = the LayerNorm input gradient itself,
 *   dx2 (may be NULL) (+)= dx            -- the residual branch's fan-in (accumulate2: add to what dx2 holds),
 *   dz  (may be NULL)  = m ? dx * zscale : 0 -- the gradient of the sum's OTHER operand in front of what produced it: m =
 *       (z > 0 when z is given: Dense(relu), asr_relu_bwd) AND (asr_dropout's keep(row * C + col, drop_seed) when drop_rate > 0:
 *       the operand was dropped by asr_add_layernorm_fwd_dropout); zscale is the caller's 1 / (1 - rate) (or 1).
 * Every value is the one the separate calls (asr_layernorm_bwd, asr_axpy alpha 1, asr_relu_bwd / asr_dropout) produce, bit for
 * bit.  At least one output; C % 4 == 0, C <= 2048 and 16-byte aligned pointers, else ASR_ERR_UNSUPPORTED. */
int asr_layernorm_bwd_fused(const float* dy, const float* xhat, const float* rstd, const float* gamma, int rows, int C,
                            float* dx, float* dx2, int accumulate2, const float* z, float drop_rate, unsigned drop_seed,
                            float zscale, float* dz, float* dgamma, float* dbeta, float* partials, void* stream);
/* Deferred form: with dgamma == dbeta == NULL the block partials of the two sums stay in `partials`
 * ([asr_layernorm_bwd_blocks(rows)][2 C] floats: sum dy * xhat | sum dy) and the caller reduces them later -- the gradients of
 * affine parameters are needed by the optimiser only, so a backward pass can leave all of them to ONE asr_colsum_multi_batch. */
int asr_layernorm_bwd_blocks(int rows);
/* Several fixed-order column reductions in two launches.  items_dev: n_items entries in DEVICE memory; entry = a [rows][ld] matrix of
 * partial rows whose first sum(width) columns form nseg <= 4 consecutive segments that go to out[0 .. nseg-1]; tmp = 64 * sum(width)
 * floats of scratch per entry.  Per entry the arithmetic and summation order of the reductions the producers run themselves
 * (two levels: at most 64 splits of consecutive rows, four chains each): the same bits.  max_cols = the largest sum(width). */
typedef struct asr_reduce_item { const float* partials; float* tmp; int rows, ld, nseg; int width[4]; float* out[4]; } asr_reduce_item;
int asr_colsum_multi_batch(const asr_reduce_item* items_dev, int n_items, int max_cols, void* stream);
/* asr_add_layernorm_fwd on (asr_dropout(a, rate, seed), b) without the pass over a: a is read once and stays as it is (the
 * backward regenerates the mask: asr_layernorm_bwd_fused).  Same bits as the two calls.  rows * C < 2^32; C % 4 == 0, C <= 2048,
 * 16-byte aligned pointers, else ASR_ERR_UNSUPPORTED. */
int asr_add_layernorm_fwd_dropout(const float* a, const float* b, const float* gamma, const float* beta, int rows, int C,
                                  float eps, float rate, unsigned seed, float* y, float* xhat, float* rstd, void* stream);

/* K12 embedding (transformer.py:30-55): out[n][t] = scale * table[ids[n][t]] (row 0 reads as zeros
 * when zero_pad) + pos[t] (either table or pos may be NULL).  Backward is a deterministic
 * segmented sum: perm = positions sorted by id, uniq[u] = the u-th distinct id, seg[u]..seg[u+1]
 * its slice of perm (built on the host from the fed ids); rows of ids that do not occur are
 * not written (zero the gradient buffer first). */
int asr_embed_fwd(const float* table, const int32_t* ids, const float* pos, int N, int T, int C,
                  int zero_pad, float scale, float* out, void* stream);
int asr_embed_bwd(const float* dout, const int32_t* perm, const int32_t* uniq, const int32_t* seg, int n_uniq,
                  int C, int zero_pad, float scale, float* dtable, void* stream);
/* The same gradient straight from the fed ids [rows] (no host-side sort, nothing to upload): a wave per table row v < V adds the
 * rows of dout whose id is v in ascending position -- the order of the stable sort above, hence the same bits.  Ids outside
 * [0, V) contribute nothing; rows of the table whose id does not occur are not written.  C % 4 == 0, C <= 1024, 16-byte
 * aligned dout / dtable, else ASR_ERR_UNSUPPORTED.  (tf.gradients of tf.nn.embedding_lookup, transformer.py:30-55.) */
int asr_embed_bwd_ids(const float* dout, const int32_t* ids, int rows, int V, int C, int zero_pad, float scale,
                      float* dtable, void* stream);

/* K16 label_smoothing(one_hot(y)) + softmax_cross_entropy_with_logits_v2 + istarget masking
 * (end2end/model.py:342-356, language_model.py:55-67).  logits [rows][ld] (ld >= V, pad columns ignored),
 * target [rows] (may be -1: one_hot = 0 but the row still counts), pad_id = 0.
 *   loss_rows [rows]; preds [rows] = argmax; stats [rows][2] = {loss*ist, (pred==target)*ist};
 *   dlogits [rows][ld] = d(mean_loss)/dlogits with inv_count = 1/sum(ist) (NULL: forward only). */
int asr_smoothed_ce(const float* logits, int ld, const int32_t* target, int rows, int V, float eps, int pad_id,
                    float inv_count, float* loss_rows, int32_t* preds, float* stats, float* dlogits, void* stream);

/* ====================================================================== end2end pre-net
 * end2end/model.py:214-264 (Transformer_Model.pre_net) + dot_product_attention :134-172.
 * 3x3 stride-1 convs are asr_tap_gemm / asr_tap_wgrad (ntaps = 9; desc.relu = 2 selects tanh).  The 64 -> 64
 * stride-2 conv runs as a 2x2-tap conv (ntaps = 4) over the "phase split" plane of its input: a padded plane of
 * (H/2, W/2) pixels x 4*C channels whose channel block (h&1)*2+(w&1) holds input pixel (h, w); its weights are
 * expanded with asr_conv_s2_expand and its weight gradient folded back with asr_conv_s2_gather.
 */

/* Pixel addressing of a logical [B][H][W][C] tensor (base pointer = pixel (0,0,0) of the buffer / plane row 0):
 *   kind 0  padded plane   [B][H+1][W+1][ld]   (interior pixel (h,w) at row h+1, column w+1)
 *   kind 1  plain NHWC     [B][H][W][ld]
 *   kind 2  phase-split padded plane [B][H/2+1][W/2+1][ld], ld >= 4*C, channel block ((h&1)*2+(w&1))*C
 * C must be a multiple of 4 with C/4 a divisor of 256; ld a multiple of 4. */
typedef struct { int kind, B, H, W, C, ld; } asr_pixmap;

/* conv2d(1 -> 64, 3x3, stride 2, 'same', tanh) on x [B][T][F] (T, F even): a1 [B][T/2][F/2][64]  (model.py:219).
 * Backward: dz = dL/d(pre-activation) [B][T/2][F/2][64] -> dw [3][3][1][64], db [64]. */
int asr_prenet_conv1_fwd(const float* x, const float* w, const float* bias, int B, int T, int F, float* a1, void* stream);
size_t asr_prenet_conv1_bwd_workspace(int B, int T, int F);
int asr_prenet_conv1_bwd(const float* x, const float* dz, int B, int T, int F, float* dw, float* db,
                         float* workspace, void* stream);

/* tf.layers.batch_normalization(training=True) (model.py:220,222,228-232,264,266): batch moments over B*H*W
 * (biased variance, accumulated in float64), y = gamma*(a-mean)*rstd + beta, optionally y = relu(y + res) (:267).
 * Backward: dz = gamma*rstd*(dy - mean(dy) - xhat*mean(dy*xhat)) * act'(.) where the normalised tensor a is the
 * OUTPUT of the activation of the conv in front (act 0 none, 1 relu: a > 0, 2 tanh: 1 - a^2); dgamma, dbeta (=). */
size_t asr_bn_workspace(const asr_pixmap* m);
int asr_bn_stats(const float* src, const asr_pixmap* m, float eps, float* mean, float* rstd, void* workspace, void* stream);
/* Moving statistics of keras.layers.BatchNormalization (cnn_ctc.py:106-107 `norm`; Keras 2.3.1 normalization.py call()):
 * with mean / rstd of the current batch (asr_bn_stats; count = B*H*W samples per channel) the moving mean / variance take one
 * momentum step -- the batch variance made unbiased by count / (count - (1 + eps)) first, as Keras does --; inf_rstd, if given,
 * receives 1 / sqrt(moving_var + eps), the factor of the inference-mode normalisation (Model.predict, cnn_ctc.py:82).  With
 * mean = rstd = NULL only inf_rstd is computed. */
int asr_bn_moving(const float* mean, const float* rstd, int C, float eps, float count, float momentum,
                  float* mov_mean, float* mov_var, float* inf_rstd, void* stream);
int asr_bn_apply(const float* src, const asr_pixmap* sm, const float* mean, const float* rstd, const float* gamma,
                 const float* beta, const float* res, const asr_pixmap* rm, int relu, float* dst, const asr_pixmap* dm,
                 void* stream);
int asr_bn_bwd(const float* dy, const asr_pixmap* ym, const float* a, const asr_pixmap* am, const float* mean,
               const float* rstd, const float* gamma, int act, float* dz, const asr_pixmap* zm, float* dgamma,
               float* dbeta, void* workspace, void* stream);
/* dst = dy * (y > 0): gradient of the closing relu(f2 + out) (model.py:267) */
int asr_relu_mask(const float* dy, const asr_pixmap* ym, const float* y, const asr_pixmap* om, float* dst,
                  const asr_pixmap* dm, void* stream);

/* Gradient of MaxPooling2D(2,2) on padded planes (Keras variant, lm_and_am/model/cnn_ctc.py:108-109,124-131):
 * dy [B][H/2+1][W/2+1][C], y = the pooled tensor's input [B][H+1][W+1][C] -> dx (=, every interior pixel written);
 * the first maximum of a window in row-major order takes the gradient. */
int asr_maxpool_bwd(const float* dy, const float* y, int B, int H, int W, int C, float* dx, void* stream);

/* HWIO [3][3][Cin][Cout] <-> the 2x2-tap weights [4][4*Cin][Cout] of the phase-split stride-2 conv */
int asr_conv_s2_expand(const float* w, int Cin, int Cout, float* W4, void* stream);
int asr_conv_s2_gather(const float* dW4, int Cin, int Cout, float* dw, void* stream);
/* Round 6: the DATA-gradient of that layer (tf.gradients through the stride-2 tanh conv, end2end/model.py:225-229), one launch per
 * phase of the input gradient instead of one 4-tap GEMM with a 64-deep contraction per tap: phase (rp, cp) receives
 * (rp ? 1 : 2) x (cp ? 1 : 2) of the four taps (the other (tap, phase) blocks of W4 are structurally zero), so each phase is a C -> C
 * convolution with its own tap list on weights in MFMA fragment order.
 *   asr_conv_s2_arrange: W4 [4][4 C][C] (asr_conv_s2_expand with Cin = Cout = C, C % 64 == 0) -> its nine non-zero (phase, tap) blocks in
 *     data-gradient view, out_bytes >= asr_conv_s2_arrange_bytes(C) (checked before the launch); once per optimiser step.
 *   asr_conv_s2_dgrad: d = the descriptor asr_tap_gemm takes for this data-gradient (ntaps 4, wmode 1, K = C, N = 4 C, pixel-indexed);
 *     dZ padded plane [.][lda], dx the phase-split plane [.][ldo_y >= 4 C]: every interior pixel of all 4 C columns is written.
 *     Equal to asr_tap_gemm(d, dZ, W4, ..., dx) up to the order of the K sums. */
size_t asr_conv_s2_arrange_bytes(int C);
int asr_conv_s2_arrange(const float* W4, int C, float* out, size_t out_bytes, void* stream);
int asr_conv_s2_dgrad(const asr_gemm_desc* d, const float* dZ, const float* Wf9, float* dx, void* stream);

/* tf.transpose(q, [0,3,1,2]) and back (model.py:234-256): 64 channels [choff, choff+64) of a padded plane
 * [B][H+1][W+1][ld] (W = 80) <-> T-layout [B][H][64][W]; the reverse direction can add two sources. */
int asr_plane_to_T(const float* plane, int B, int H, int W, int ld, int choff, float* dst, void* stream);
int asr_T_to_plane(const float* srcA, const float* srcB, int B, int H, int W, int ld, int choff, float* plane, void* stream);

/* dot_product_attention(q_time, k_time, v_time, mask=False) (model.py:252): per (batch, channel) attention over
 * the time axis, head width 80, no masks; tensors in T-layout = [N][T][H*80] with H = 64 "heads".
 * lse [N][H][Tq]; delta_ws N*H*Tq floats. */
int asr_attention_nomask_fwd(const float* Q, const float* K, const float* V, int N, int Tq, int Tk, int C, int H,
                             float* O, float* lse, void* stream);
int asr_attention_nomask_bwd(const float* Q, const float* K, const float* V, const float* O, const float* dO,
                             const float* lse, int N, int Tq, int Tk, int C, int H,
                             float* dQ, float* dK, float* dV, float* delta_ws, void* stream);
/* dot_product_attention(q_fre, k_fre, v_fre, mask=False) (model.py:253): per (batch, channel) attention over the
 * 80 frequency bins with the time axis as depth (scale 1/sqrt(T)); T-layout tensors [B][T][64][80].
 * P [B][64][80][80] (attention weights) is saved for the backward; dS_ws: same size. */
int asr_freq_attention_fwd(const float* Q, const float* K, const float* V, int B, int T, float* P, float* O, void* stream);
int asr_freq_attention_bwd(const float* Q, const float* K, const float* V, const float* P, const float* dO, int B, int T,
                           float* dQ, float* dK, float* dV, float* dS_ws, void* stream);

/* layer_norm(a + r) over the 64 channels of a pixel map (model.py:261; eps 1e-8, biased variance).
 * xhat (same map) and rstd [B*H*W] are saved; backward: dx (=), dgamma, dbeta (=). */
int asr_pix_add_ln_fwd(const float* a, const float* r, const asr_pixmap* m, const float* gamma, const float* beta,
                       float eps, float* y, float* xhat, float* rstd, void* stream);
size_t asr_pix_ln_bwd_workspace(const asr_pixmap* m);
int asr_pix_ln_bwd(const float* dy, const float* xhat, const float* rstd, const asr_pixmap* m, const float* gamma,
                   float* dx, float* dgamma, float* dbeta, float* workspace, void* stream);

/* Dense forward GEMM of asr_tap_gemm (ntaps 1, wmode 0, H = 0) with the contraction split `splits` ways over the grid and a
 * second pass that adds the partial planes in a fixed order and applies bias / ReLU / affine: for a deep K with few output
 * tiles (tf.layers.dense(128) on the 6400-wide reshape, acoustic_model.py:53).  K % (32 * splits) == 0, 2 <= splits <= 16;
 * workspace = asr_tap_gemm_splitk_workspace(d, splits) bytes, 16-byte aligned.  Equal to asr_tap_gemm up to the order of the K
 * sum; reproducible run to run. */
size_t asr_tap_gemm_splitk_workspace(const asr_gemm_desc* d, int splits);
int asr_tap_gemm_splitk(const asr_gemm_desc* d, const float* A, const float* W, const float* bias, const float* scale,
                        const float* shift, float* out_a, float* out_y, int splits, void* workspace, void* stream);

/* The same split on the LDS-DMA kernel of asr_tap_gemm_nt (gemm1.hip): both operands K-contiguous -- A [M][K] (pitch d->lda) and
 * Bt [N][K] (pitch ldb): a forward layer's transposed kernel (asr_transpose_batch), or the kernel W [K_out][N_out] itself for a
 * data-gradient dX = dY . W^T (then d->K = the layer's output width, d->N its input width).  d->wmode is not read.  N >= 64,
 * K % (32 * splits) == 0, 2 <= splits <= 16; workspace = asr_tap_gemm_nt_splitk_workspace(d, splits) bytes, 16-byte aligned;
 * workspace_bytes = what the caller really holds there: a smaller buffer is refused (ASR_ERR_BAD_ARG) before anything is launched.
 * Round 5: the 6400 -> 128 hidden dense of acoustic_model.py:53 (ten splits: 500 workgroups for the chip's 512 slots) and the
 * 1536 -> 128 data-gradient behind it, which asr_tap_gemm_splitk ran on 64 x 64 register-staged tiles. */
size_t asr_tap_gemm_nt_splitk_workspace(const asr_gemm_desc* d, int splits);
int asr_tap_gemm_nt_splitk(const asr_gemm_desc* d, const float* A, const float* Bt, int ldb, const float* bias, const float* scale,
                           const float* shift, float* out_a, float* out_y, int splits, void* workspace, size_t workspace_bytes, void* stream);

/* Data-gradient of a dense layer fed by a Dense(relu) layer, with that layer's ReLU backward in the epilogue (round 5):
 *   dX[rows][d->N] = (dY[rows][d->K] . W[d->N][d->K]^T) where H[rows][d->N] > 0, else 0        (d: ntaps 1, wmode 1, accumulate 0; H and dX share
 * the pitch d->ldo_y) -- asr_tap_gemm + asr_relu_bwd in one launch on the LDS-DMA kernel (gemm1_relumask_kernel), as the two calls where that
 * kernel does not take the shape; the same bits.  `end2end/transformer.py:204-222` feedforward backward: no pass over the 32768 x 2048 result. */
int asr_tap_gemm_relu_bwd(const asr_gemm_desc* d, const float* dY, const float* W, const float* H, float* dX, void* stream);

/* fp32 contraction on PRE-ARRANGED weights: same arguments, arithmetic (v_mfma_f32_32x32x2_f32, fp32 accumulate) and
 * epilogue as asr_tap_gemm; the weight tensor is first copied into MFMA fragment order, once per optimiser step:
 *   asr_arrange_weights(W, ntaps, K, N, ldw, wmode, out): out = fp32 [ntaps][ceil(K/8)][ceil(N/32)][64 lanes][4], lane
 *   32h+i = column 32*block+i, contraction indices 8*group+4h..+3, zero padded; wmode 1 takes the data-gradient view of a
 *   forward tensor (K, N = the GEMM's, i.e. swapped; taps mirrored), after which desc.wmode is ignored.
 * The B operands then arrive as one coalesced float4 per lane straight from L2 (no weight tile in LDS, two barriers per
 * K chunk instead of two per tap). */
size_t asr_arrange_weights_bytes(int ntaps, int K, int N);
int asr_arrange_weights(const float* W, int ntaps, int K, int N, int ldw, int wmode, float* out, void* stream);
int asr_tap_gemm_pw(const asr_gemm_desc* d, const float* A, const float* Warranged,
                    const float* bias, const float* scale, const float* shift,
                    float* out_a, float* out_y, void* stream);

/* ------------------------------------------------------------------ Winograd F(2x2, 3x3) (round 2; the engines' default
 * for the 3x3 convolutions it supports, forward and data-gradient; the WEIGHT gradient's Winograd form needs no entry of its own:
 * asr_tap_wgrad above)
 * The 3x3 convolution of asr_tap_gemm (ntaps 9; forward, or data-gradient view with wmode 1) with 16 instead of 36
 * multiplies per 2x2 output tile, input and output channel -- still fp32, results agree with asr_tap_gemm to rounding
 * (tests/test_wino_gpu.py).  Weights are transformed once per optimiser step:
 *   asr_winograd_weights2(W HWIO [3][3][Cin][Cout] with pitch ldw, K, N, ldw, wmode, out, out_bytes)
 *     (round 5: replaces asr_winograd_weights, whose output doubled in round 4 under an unchanged name; out_bytes below
 *     asr_winograd_weights_bytes(K, N) = 2 x 16 K N floats is ASR_ERR_BAD_ARG, nothing is launched)
 *     wmode 0: K = Cin, N = Cout;  wmode 1: the data-gradient view, K = Cout, N = Cin, taps mirrored (as asr_arrange_weights)
 *     out (asr_winograd_weights_bytes(K, N) bytes, round 4: two layouts side by side): [16][K][N], and 16 K N floats further
 *     the chunk-major form [K / 8][16][N / 32][2][2][32][2] (written when K % 8 == 0 and N % 32 == 0) whose 1 KB blocks wino11_kernel
 *     copies into LDS as they lie.
 * asr_winograd_supported(d): ntaps 9, K % 8 == 0, input plane below 2 GiB, and N % 64 == 0 -- or N % 32 == 0 on planes whose
 *   (W + 1) / 2 tile columns split into blocks of 11..15 (the 25-, 50-, 100-wide planes of the models: yes) -- otherwise use
 *   asr_tap_gemm[_pw].  Odd plane heights are supported since round 4 (T_pad 1000: 125 x 25 planes).
 * asr_tap_gemm_wino: same contract and epilogue options as asr_tap_gemm_pw.  Two kernels behind it (csrc/wino.hip), chosen from
 * the descriptor's widths and plane geometry alone (never the batch or the device): wino11_kernel (round 4: 64 tiles x 32
 * channels per item, eight waves of 128 registers -- one transform row each --, two workgroups per CU, the epilogue fixed per
 * kernel instantiation) for every shape whose tile columns split into column blocks; wino8_kernel (64 x 64 items, plain tile
 * order) for the rest.  Identical arithmetic per output element up to the order of the inverse transform's additions. */
size_t asr_winograd_weights_bytes(int K, int N);
int asr_winograd_weights2(const float* W, int K, int N, int ldw, int wmode, float* out, size_t out_bytes, void* stream);
int asr_winograd_supported(const asr_gemm_desc* d);
/* Round 5: the forward conv of a cell that is the BRANCH of a squeeze-excitation block (acoustic_model2.py:40-41 `h1_1 = cnn_cell(..)`,
 * `squeeze_excitation_layer(h1_1, ..)`): asr_tap_gemm_wino plus the per-image channel sums of the cell's BN output y, written as partial
 * rows y_sums[rows][N], rows = asr_winograd_sum_rows(d), the rows of image b = [b * rows / B, (b + 1) * rows / B).  asr_se_fwd_sums takes
 * them in place of its own pass over the plane (Global_Average_Pooling, acoustic_model2.py:135-137).  asr_winograd_sum_rows(d) == 0:
 * not a shape of wino11_kernel's forward epilogue -- use asr_tap_gemm_wino + asr_se_fwd. */
int asr_winograd_sum_rows(const asr_gemm_desc* d);
int asr_tap_gemm_wino_sums(const asr_gemm_desc* d, const float* A, const float* Ut, const float* bias, const float* scale,
                           const float* shift, float* out_a, float* out_y, float* y_sums, void* stream);
/* The mirror in the backward pass: the data-gradient (wmode 1, no accumulate) that completes dL/d(output of an SE block) also leaves the
 * reduction the block's backward starts with -- xsums[rows][N] partial rows (rows = asr_winograd_sum_rows(d), image-contiguous) of
 * sum over pixels of dy * (se_scale * x + se_shift), x = the block's branch plane [B][H + 1][W + 1][N] -- for asr_se_bwd_cell_sums. */
int asr_tap_gemm_wino_sesum(const asr_gemm_desc* d, const float* dZ, const float* Ut, const float* x, const float* se_scale,
                            const float* se_shift, float* dy, float* xsums, void* stream);
int asr_tap_gemm_wino(const asr_gemm_desc* d, const float* A, const float* Wt,
                      const float* bias, const float* scale, const float* shift,
                      float* out_a, float* out_y, void* stream);
/* Forward conv of a POOLED cell in one launch (replaces asr_tap_gemm_wino + asr_pool_fwd, acoustic_model.py:120-130 /
 * acoustic_model2.py:126-133: conv + bias + ReLU -> BN -> 2x2 pool): out_a = ReLU(conv + bias) as asr_tap_gemm_wino writes
 * it (ldo_a == N), y_pooled [B][H/2+1][W/2+1][N] = pool (1 average, 2 maximum) of bn_scale * out_a + bn_shift, bit-identical
 * to asr_pool_fwd on the stored activation -- the four pixels of a Winograd tile are one pooling window, so the activation
 * is not read back. */
int asr_tap_gemm_wino_pool(const asr_gemm_desc* d, const float* A, const float* Wt, const float* bias, const float* bn_scale,
                           const float* bn_shift, float* out_a, int pool, float* y_pooled, void* stream);

/* ------------------------------------------------------------------ fused backward prologue (round 2)
 * The data-gradient of cell k, with the backward of cell k-1's [pool ->] BN -> ReLU (what asr_cell_bwd_pre computes in a
 * pass of its own: tf.gradients through average_pooling2d / max_pooling2d, batch_normalization and relu,
 * acoustic_model2.py:107-133, acoustic_model.py:103-130) applied in the GEMM's epilogue: the tile of dL/dy(k-1) never
 * goes to memory, dZ(k-1) is written directly, and the per-channel sums come out as tile partials folded in a fixed order.
 *   d          the data-gradient descriptor (wmode 1, ntaps 9 or 1, pixel-indexed: H, W = the plane of cell k-1's OUTPUT)
 *   dZ, W      as for asr_tap_gemm (W HWIO), asr_tap_gemm_pw (prearranged == 1: data-gradient view from asr_arrange_weights) or
 *              asr_tap_gemm_wino (prearranged == 2: asr_winograd_weights2 with wmode 1)
 *   pool       0 none, 1 average 2x2, 2 maximum 2x2 (first maximum of bn_scale * a + bn_shift in row-major window order)
 *   gate_H/W   cell k-1's pre-pool plane: H x W for pool 0, 2H x 2W otherwise (odd sizes are not supported: use asr_cell_bwd_pre)
 *   gate_a     cell k-1's post-ReLU pre-BN activations, padded plane [B][gate_H+1][gate_W+1][N]
 *   dy_prev    with d->accumulate: the plane holding the gradient contributions already made to y(k-1) (read only)
 *   dz_out     padded plane like gate_a: dL/d(conv + bias) of cell k-1; every interior pixel is written
 *   dscale, dshift, dbias [N];  partials: asr_tap_gemm_gated_workspace(d) bytes */
size_t asr_tap_gemm_gated_workspace(const asr_gemm_desc* d);
int asr_tap_gemm_gated(const asr_gemm_desc* d, const float* dZ, const float* W, int prearranged,
                       int pool, int gate_H, int gate_W, const float* gate_a,
                       const float* bn_scale, const float* bn_shift, const float* dy_prev,
                       float* dz_out, float* dscale, float* dshift, float* dbias, float* partials, void* stream);

/* Round 5: the same fusion where cell k-1 hands its output to a DENSE layer (the reshape -> tf.layers.dense heads,
 * acoustic_model.py:48-50, acoustic_model2.py:62-66): the dense layer's data-gradient dL/d(flat) = dZ . W^T is [B * gate_H] x
 * [gate_W * gate_C] -- row = (image, pixel row), column = pixel column * gate_C + channel -- and its epilogue applies the BN / ReLU
 * backward of the (un-pooled) cell: no dL/d(flat) tensor and no asr_cell_bwd_pre pass.  dZ(k-1) bit for bit what asr_tap_gemm +
 * asr_cell_bwd_pre (layout 2) give, the three channel sums folded in another fixed order.
 *   d          the dense layer's data-gradient descriptor: ntaps 1, wmode 1, H = 0, M = B * gate_H, K = its output width,
 *              N = gate_W * gate_C, no accumulate
 *   W          the dense kernel [gate_W * gate_C][K] (row-major, pitch d->ldw)
 *   a_plane, dz_out   cell k-1's activation / dZ planes [B][gate_H + 1][gate_W + 1][gate_C];  dscale, dshift, dbias [gate_C]
 *   partials   asr_tap_gemm_gated_dense_workspace(d, gate_W, gate_C) bytes
 * asr_tap_gemm_gated_dense_supported: gate_C % 32 == 0 and a GEMM large enough for the LDS-DMA kernel (gemm1_kernel); otherwise
 * ASR_ERR_UNSUPPORTED and the caller keeps the two-pass form. */
int asr_tap_gemm_gated_dense_supported(const asr_gemm_desc* d, int gate_H, int gate_W, int gate_C);
size_t asr_tap_gemm_gated_dense_workspace(const asr_gemm_desc* d, int gate_W, int gate_C);
int asr_tap_gemm_gated_dense(const asr_gemm_desc* d, const float* dZ, const float* W, int gate_H, int gate_W, int gate_C,
                             const float* a_plane, const float* bn_scale, const float* bn_shift, float* dz_out,
                             float* dscale, float* dshift, float* dbias, float* partials, void* stream);

/* ------------------------------------------------------------------ maximum pool, compact form (round 4)
 * A max-pooled cell (acoustic_model.py:120-130: conv + bias + ReLU -> BN -> max_pooling2d 2x2) whose forward conv AND the
 * data-gradient that completes its output gradient both run on the Winograd kernel needs no pre-pool activation plane: the forward
 * launch writes the pooled output, the activation AT each window's maximum (a_max, pooled geometry [B][H/2+1][W/2+1][N]) and the
 * maximum's window position (index: two bit planes of 32 channels per pooled pixel, asr_poolmax_index_bytes); the gated data-gradient
 * reads those two instead of four activation pixels per window.  A quarter of the activation traffic in both directions, and bit for bit
 * the values of asr_tap_gemm_wino_pool + asr_tap_gemm_gated(pool = 2) (first maximum of bn_scale * a + bn_shift in row-major window
 * order; the positions that are not the maximum contribute exact zeros to the channel sums in both forms).
 *   asr_winograd_poolmax_supported(fwd, bwd): fwd = the cell's forward descriptor (pre-pool plane H x W, both even), bwd = the
 *   data-gradient descriptor whose output is the cell's pooled plane; a function of widths and geometry only. */
size_t asr_poolmax_index_bytes(int B, int H2, int W2, int N);
int asr_winograd_poolmax_supported(const asr_gemm_desc* fwd, const asr_gemm_desc* bwd);
int asr_tap_gemm_wino_poolmax(const asr_gemm_desc* d, const float* A, const float* Wt, const float* bias, const float* bn_scale,
                              const float* bn_shift, float* y_pooled, float* a_max, unsigned* index, void* stream);
/* as asr_tap_gemm_gated(prearranged = 2, pool = 2) with (a_max, index) in place of gate_a; gate_H / gate_W = the pre-pool plane */
int asr_tap_gemm_gated_poolmax(const asr_gemm_desc* d, const float* dZ, const float* Wt, int gate_H, int gate_W, const float* a_max,
                               const unsigned* index, const float* bn_scale, const float* bn_shift, const float* dy_prev,
                               float* dz_out, float* dscale, float* dshift, float* dbias, float* partials, void* stream);

/* Round 5: the same for AVERAGE-pooled cells (the "maxpool" of acoustic_model2.py:116-124 is an average pool): a_sum = the sum of each
 * window's four post-ReLU activations (pooled geometry), index = four bit planes per pooled pixel and 32-channel block with the ReLU sign
 * of each window position (asr_poolavg_index_bytes).  dZ, dshift, dbias: the bits of asr_tap_gemm_wino_pool (pool 1) + asr_tap_gemm_gated
 * (pool 1); dscale: the same value with the window's four products folded into one multiply-add.  Same support rule as the maximum form
 * (asr_winograd_poolmax_supported). */
size_t asr_poolavg_index_bytes(int B, int H2, int W2, int N);
int asr_tap_gemm_wino_poolavg(const asr_gemm_desc* d, const float* A, const float* Ut, const float* bias, const float* scale,
                              const float* shift, float* y_pooled, float* a_sum, unsigned* index, void* stream);
int asr_tap_gemm_gated_poolavg(const asr_gemm_desc* d, const float* dZ, const float* Ut, int gate_H, int gate_W, const float* a_sum,
                               const unsigned* index, const float* bn_scale, const float* bn_shift, const float* dy_prev,
                               float* dz_out, float* dscale, float* dshift, float* dbias, float* partials, void* stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* ASR_HIP_H */
