/*
 * visinger_hip.h -- C ABI of the MI355X-native (gfx950) VISinger variational-inference hot path.
 *
 * The reference (jisang93/VISinger) has no FFI: its boundary for this path is the Python nn.Module API of
 * modules/visinger/{encoder,flow,decoder,predictor}.py and modules/rel_transformer.py.  This header
 * is the C ABI that sits directly under that API: every entry point below is what one reference
 * ``nn.Module.forward`` (cited file:line, paths under the reference tree) binds to.  The Python mirror of the
 * module API that calls it through ctypes lives in visinger_amd/modules/ (see INTEGRATION.md).
 *
 * Conventions
 *   - all tensors are fp32, contiguous, device (HBM) pointers unless stated; layout [B, C, T] (T fastest);
 *   - frame masks are fp32 [B, T] (the reference's [B, 1, T] nonpadding mask);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); nothing here synchronises the device;
 *   - handles own their packed weights and workspaces in HBM (hipMalloc at create / first use of a shape, never
 *     on the steady-state path);  weights are handed over in the reference's own state_dict layout
 *     (``weight_v`` / ``weight_g`` / ``bias``) and are folded (g * v / ||v||) and re-packed into MFMA fragment
 *     order on the device by the *_set_weights call;
 *   - every function returns VS_OK (0) or an error code; vs_last_error() returns a thread-local message.
 *     Nothing aborts, nothing falls back to a CPU path.
 */
#ifndef VISINGER_HIP_H
#define VISINGER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VS_API __attribute__((visibility("default")))

enum vs_status { VS_OK = 0, VS_EINVAL = 1, VS_EHIP = 2, VS_EUNSUPPORTED = 3, VS_ENOMEM = 4 };

VS_API const char *vs_last_error(void);
/* Name of the kernel instance the calling thread's last vs_conv_forward / vs_respair_forward / vs_relattn_fwd launched, spelled as
 * rocprofv3 prints it without the namespace and argument list ("conv_split_kernel<1, 8, 4, 1, 6>"): the dispatch is decided in
 * this library, so a caller that attributes per-launch timings to kernel instances (bench.py's roofline line) reads it back
 * here instead of restating the selection.  Thread-local; "" before the first launch.  (No reference counterpart.)          */
VS_API const char *vs_last_kernel_name(void);
VS_API int vs_abi_version(void);          /* 7: vs_conv_set_weights_batch, vs_weight_norm_multi_fwd / _bwd, vs_conv_wgrad_bias, vs_wn_step_fwd / _bwd, vs_l1_mean_fwd / _bwd; 6: vs_source_hash, vs_bias_grad, vs_conv_set_weights_pair; 5: vs_relattn_fwd_work / vs_relattn_kv_work_bytes; 4: vs_set_option / vs_get_option / vs_reset_option; 3: vs_dtype in vs_conv_io_t; 2: vs_relattn_fwd(math) */
/* sha256 (hex) over the sources this library was compiled from (kernels, headers, textual includes, the build recipe), embedded by
 * visinger_amd/csrc/build.py.  The loader recomputes it over the tree it sits in and refuses a library built from other sources (a
 * stale object that an mtime check would pass after a checkout).  (No reference counterpart: the reference has no native code.)  */
VS_API const char *vs_source_hash(void);
/* Dispatch switches (A/B comparisons and debugging; never needed for correct results).  The library reads the environment
 * variables of the same names ONCE, when it is loaded; afterwards only these calls change a switch, and no launch path touches
 * the environment.  Names and meanings: INTEGRATION.md "Switches".  Unknown name -> VS_EINVAL.  (No reference counterpart: the
 * reference steers nothing on this path but `hparams`.)                                                                       */
VS_API int vs_set_option(const char *name, long long value);
VS_API int vs_get_option(const char *name, long long *value);
VS_API int vs_reset_option(const char *name);
/* number of HIP devices visible / name of device 0 written into buf (diagnostics for the loader) */
VS_API int vs_device_info(char *buf, size_t buf_bytes);

/* ------------------------------------------------------------------------------------------------------------
 * a12  weight-norm fold: w[r, :] = g[r] * v[r, :] / ||v[r, :]||_2
 *      (torch.nn.utils.weight_norm at modules/visinger/encoder.py:147,154,164; decoder.py:24,72-87)          */
VS_API int vs_weightnorm_fold(const float *v, const float *g, float *w, int64_t rows, int64_t cols, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * Conv1d / ConvTranspose1d operator handle -- every nn.Conv1d / nn.ConvTranspose1d site of the path
 * (encoder.py:88,90,152,163; flow.py:60,62; decoder.py:19,24-26,34,72-87; rel_transformer.py:120-128,332-333).
 * Implicit GEMM on the exact-fp32 matrix instruction (v_mfma_f32_32x32x2_f32).                                */
typedef struct vs_conv vs_conv_t;

enum vs_conv_kind {
    VS_CONV1D = 0,           /* y[b,co,t] = bias[co] + sum w[co,ci,k] x[b,ci,t + k*dil - pad]                  */
    VS_CONV_TRANSPOSE1D = 1, /* nn.ConvTranspose1d, w is [c_in, c_out, k]; `dilation` carries the stride        */
    VS_CONV1D_PAIRED = 2     /* Conv1d whose c_out = 2*H rows are consumed as (row c, row H+c) pairs by a fused
                                epilogue: WaveNet gate (encoder.py:206-213) or affine coupling (flow.py:71-85) */
};
enum vs_in_act {
    VS_IN_NONE = 0,
    VS_IN_LRELU = 1,      /* leaky_relu(x, 0.1)                 */
    VS_IN_MASK = 2,       /* x * mask[b,t]                      */
    VS_IN_LRELU_MASK = 3  /* leaky_relu(x, 0.1) * mask[b,t]     (decoder.py:93-95) */
};
enum vs_out_act { VS_OUT_NONE = 0, VS_OUT_TANH = 1, VS_OUT_RELU = 2 };
enum vs_pair_mode { VS_PAIR_GATE = 0, VS_PAIR_COUPLING_FWD = 1, VS_PAIR_COUPLING_INV = 2 };
enum vs_out_mode {
    VS_OUT_LINEAR = 0,            /* y = act((v + res + acc) * scale) [* mask]                                  */
    VS_OUT_COUPLING_MEAN_FWD = 1, /* y = v*mask + res*mask                 (flow.py:74,78 with mean_only)        */
    VS_OUT_COUPLING_MEAN_INV = 2  /* y = (res - v*mask) * mask             (flow.py:74,83 with mean_only)        */
};

/* flags for vs_conv_create */
#define VS_CONV_FLIP_IN 1u  /* read input channels in reversed order  (folds flow.py:88-95 Flip into the weights) */
#define VS_CONV_FLIP_OUT 2u /* write output rows in reversed order (within each half for PAIRED)                 */
#define VS_CONV_ADJOINT 4u  /* VS_CONV1D only: vs_conv_set_weights is handed the weight of the FORWARD conv this handle is the grad-input
                               of, [c_in, c_out, k] of THIS handle, and packs its adjoint w'[co, ci, t] = w[ci, co, k - 1 - t] (channel
                               transpose + tap reversal by index arithmetic: no flipped copy per training step); g must be NULL      */

VS_API int vs_conv_create(vs_conv_t **out, int kind, int c_in, int c_out, int k, int dilation_or_stride, int padding,
                          unsigned flags);
VS_API void vs_conv_destroy(vs_conv_t *h);
/* w: [c_out, c_in, k] ([c_in, c_out, k] for transpose).  g == NULL: w is the effective weight;
 * g != NULL: w is weight_v and g is weight_g ([dim0,1,1]) -> folded on the device.  bias may be NULL.          */
VS_API int vs_conv_set_weights(vs_conv_t *h, const float *w, const float *g, const float *bias, void *stream);
/* The same for TWO handles fed from one plain (already folded) weight `w` -- a conv and the VS_CONV_ADJOINT handle of its grad-input -- in
 * one pair of launches instead of two: the training step packs every weight version for both (autograd.HipConvFn).  bias0: h0's bias or
 * NULL; h1 gets none.  Falls back to two vs_conv_set_weights calls outside the split-f16 arithmetic.  (No reference counterpart.)       */
VS_API int vs_conv_set_weights_pair(vs_conv_t *h0, vs_conv_t *h1, const float *w, const float *bias0, void *stream);
/* The same for n handles, each from its own plain (already folded) weight w[i] and bias[i] (bias, or bias[i], may be NULL) -- every conv of a
 * network at the top of a training pass -- in TWO launches for the whole batch (fp32 fragments + largest |w| of every handle; then the f16
 * planes): the reference re-derives each module's weight inside its forward (torch.nn.utils.weight_norm's hook, modules/commons/... conv
 * modules); a training step packed ~340 handles with four launches a pair.  No handle may appear twice.  Handles outside VS_MATH_SPLIT3 (and
 * the <= 4-row VALU convs) take vs_conv_set_weights inside the call.  Stages a table through pinned host memory: not capturable into a graph.  */
VS_API int vs_conv_set_weights_batch(vs_conv_t *const *handles, const float *const *w, const float *const *bias, int n, void *stream);

/* Arithmetic of the matrix contraction of one conv handle (inputs, outputs and accumulation are fp32 in every mode):
 *   VS_MATH_F32    exact-fp32 MFMA (v_mfma_f32_32x32x2_f32), Winograd F(2,3) instances where they measured faster
 *   VS_MATH_SPLIT6 operands split exactly into three bf16 planes, the six leading cross products on v_mfma_f32_32x32x16_bf16:
 *                  fp32-class result (dropped terms <= 2^-23 relative per product) at 16/6 of the fp32 matrix rate (default)
 *   VS_MATH_SPLIT3 operands split into two f16 planes, x * s = xh + xl (22 significant bits; s = a power of two per staged tile /
 *                  per conv weight that puts the largest magnitude below 2^15, found on the device, taken out again in fp32), the
 *                  three leading cross products on v_mfma_f32_32x32x16_f16: representation + dropped term <= 2^-22 relative per
 *                  product, below the rounding of the fp32 accumulation itself; 16/3 of the fp32 matrix rate
 *   VS_MATH_BF16   operands rounded to bf16 (RNE), fp32 accumulate: BASELINE.json's long-form bf16 configuration
 * May be called before or after vs_conv_set_weights (the bf16 planes are re-packed from the fp32 fragments).
 * No reference counterpart: its arithmetic is whatever torch picks (config/models/base_config.yaml:5 `amp: false` = fp32).   */
enum vs_conv_math { VS_MATH_F32 = 0, VS_MATH_BF16 = 1, VS_MATH_SPLIT3 = 3, VS_MATH_SPLIT6 = 6 };
VS_API int vs_conv_set_math(vs_conv_t *h, int math, void *stream);
VS_API int vs_conv_get_math(const vs_conv_t *h);

/* Element type of a conv launch's activation tensors.  VS_DTYPE_BF16 = bf16-RESIDENT activations (BASELINE.json config 5: "bf16
 * activations"): with VS_MATH_BF16 every product rounds its operands to bf16 anyway, and on the bf16 matrix pipe the 128- / 256-channel
 * convs are HBM-bound with fp32 tensors (x + residual + y at 4.3-4.6 TB/s).  Plain VS_CONV1D / VS_CONV_TRANSPOSE1D launches in
 * VS_MATH_BF16 with VS_OUT_LINEAR outputs and no split_row; all strides stay in ELEMENTS; mask and bias_b stay fp32; y is rounded to
 * nearest even once, after residual / accumulate / scale / activation in fp32.                                                  */
enum vs_dtype { VS_DTYPE_F32 = 0, VS_DTYPE_BF16 = 1 };

/* One output destination of a conv launch.  Strides are in floats; row stride is always the time length. */
typedef struct vs_conv_out {
    float *y;            /* [B, rows, T_out] with batch stride y_bs                                              */
    const float *res;    /* optional residual input, same indexing as y (batch stride res_bs); may alias y       */
    const float *acc;    /* optional accumulate input, same indexing as y (batch stride acc_bs); may alias y     */
    int64_t y_bs, res_bs, acc_bs;
    float scale;         /* applied after the adds; 0.0f means unset (= 1.0f)                                    */
    int out_act;         /* enum vs_out_act                                                                      */
    int out_mask;        /* != 0: multiply by mask[b, t]                                                         */
    int mode;            /* enum vs_out_mode                                                                     */
} vs_conv_out_t;

typedef struct vs_conv_io {
    const float *x;      /* [B, c_in, T] with batch stride x_bs (0 -> c_in*T)                                    */
    int64_t x_bs;
    int64_t B, T;
    int in_act;          /* enum vs_in_act, applied while staging x into LDS                                     */
    int x_dtype;         /* enum vs_dtype of x (ABI 3; occupies what was padding: 0 = fp32 as before)            */
    const float *mask;   /* [B, T] frame mask for VS_IN_MASK / out_mask / coupling modes (NULL if unused)        */
    const float *bias_b; /* optional per-item bias [B, c_out] (conditioning), row stride bias_b_bs               */
    int64_t bias_b_bs;
    int split_row;       /* rows [0, split_row) go to out[0], rows [split_row, c_out) to out[1] (row - split_row);
                            0 or >= c_out: everything goes to out[0]                                             */
    int y_dtype;         /* enum vs_dtype of out[0].y / res / acc (ABI 3; was padding)                           */
    vs_conv_out_t out[2];
    int pair_mode;       /* PAIRED only: enum vs_pair_mode; result rows = c_out/2, written through out[0]
                            (res = x1 for the coupling modes)                                                    */
    float *logdet;       /* PAIRED + COUPLING_FWD: [B] accumulated (+=) with sum(logs * mask); may be NULL       */
} vs_conv_io_t;

VS_API int vs_conv_forward(vs_conv_t *h, const vs_conv_io_t *io, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * a11  A whole MRF residual block (or any even-length prefix / suffix of its conv chain) in ONE launch on the split-f16 arithmetic:
 *      for each pair (conv1, conv2):  x = conv2(lrelu(conv1(lrelu(x)))) + x;   y = (x [+ io->out[0].acc]) * io->out[0].scale
 *      (modules/visinger/decoder.py:91-104; the accumulate input and the scale carry the MRF sum of decoder.py:52-56).
 *      convs = {convs1[0], convs2[0], convs1[1], convs2[1], ...}: 2, 4 or 6 handles of 32, 64 or 128 channels, one odd kernel size <= 11,
 *      "same" padding, all in VS_MATH_SPLIT3 (x / acc / y fp32 tensors) or all in VS_MATH_BF16 (x / acc / y bf16-RESIDENT, x_dtype =
 *      y_dtype = VS_DTYPE_BF16: BASELINE.json configs[4]); the other combinations are refused.  The residual stream stays in registers (fp32) between the pairs; each tile recomputes
 *      its receptive halo instead of exchanging it.  io: x, B, T, in_act = VS_IN_LRELU, out[0].y / acc / scale; nothing else.        */
VS_API int vs_resblock_supported(vs_conv_t *const *convs, int nconv);
VS_API int vs_resblock_forward(vs_conv_t *const *convs, int nconv, const vs_conv_io_t *io, void *stream);

/* One launch for a residual pair of the MRF blocks on the 32- / 64-channel stages (decoder.py:92-101):
 *     y = conv2(lrelu(conv1(lrelu(x)) + b1)) + b2 + res [+ acc] [* scale]
 * conv1 / conv2 are existing handles (weights set): same channel count C in {32, 64}, same odd k <= 13, "same" padding, conv2
 * undilated.  io: x, B, T, in_act = VS_IN_LRELU, out[0] = {y, res, acc, strides, scale}; no mask / bias_b / split / activation.
 * vs_respair_supported() tells whether a pair qualifies (callers fall back to two vs_conv_forward launches otherwise).       */
VS_API int vs_respair_supported(const vs_conv_t *conv1, const vs_conv_t *conv2);
VS_API int vs_respair_forward(vs_conv_t *conv1, vs_conv_t *conv2, const vs_conv_io_t *io, void *stream);

/* Weight gradient of a stride-1 (dilated) conv, the backward of nn.Conv1d under the training step (trainer.py:306-384 ->
 * autograd of encoder.py / flow.py / decoder.py convs):  gw[co, ci, k] = sum_{b,t} gy[b, co, t] * x[b, ci, t + k*dil - pad],
 * x read as 0 outside [0, T_in).  gy: [B, c_out, T_out], x: [B, c_in, T_in].  The reduction over (b, t) is cut into slices
 * that each write one partial plane: gw_planes is [vs_conv_wgrad_planes(...)][c_out, c_in, k], every plane fully written, and
 * gw = sum of the planes (deterministic; the caller reduces).  k <= 16, (k-1)*dil <= 64.
 * (The grad-input is vs_conv_forward with the reversed / transposed weight.)                                            */
VS_API int vs_conv_wgrad_planes(int64_t B, int64_t c_out, int64_t c_in, int64_t T_out, int k);
VS_API int vs_conv_wgrad(const float *gy, const float *x, float *gw_planes, int64_t B, int64_t c_out, int64_t c_in,
                         int64_t T_out, int64_t T_in, int k, int dil, int pad, void *stream);
/* The same with the reduction of the planes done here and, with_bias != 0, the conv's bias gradient gb[co] = sum_{b,t} gy[b, co, t] from the
 * same pass over gy (the workgroups of the first c_in tile sum their gy rows): what autograd computes for the `weight` and `bias` arguments of
 * torch.nn.functional.conv1d, in two launches.  work: vs_conv_wgrad_planes(...) * (c_out * c_in * k + (with_bias ? c_out : 0)) floats of
 * scratch; out: c_out * c_in * k floats of gw followed, with_bias, by c_out floats of gb.  Deterministic (fixed plane order).              */
VS_API int vs_conv_wgrad_bias(const float *gy, const float *x, float *work, float *out, int with_bias, int64_t B, int64_t c_out,
                              int64_t c_in, int64_t T_out, int64_t T_in, int k, int dil, int pad, void *stream);
/* length of the time axis produced for an input of length T */
VS_API int64_t vs_conv_out_len(const vs_conv_t *h, int64_t T);

/* ------------------------------------------------------------------------------------------------------------
 * a6  windowed relative-position self-attention core: MultiHeadAttention.attention
 *     (modules/rel_transformer.py:148-179 with the relative helpers 181-243).
 *     q, k, v: [B, n_heads*k_channels, T] (the outputs of conv_q/k/v; they may live in one [B, 3C, T] buffer, hence
 *     the shared batch stride).  rel_k / rel_v: [n_heads_rel, 2*window+1, k_channels] (emb_rel_k / emb_rel_v);
 *     window_size < 0 disables the relative terms.  mask: [B, T] frame mask m, attention mask = m[i]*m[j]; masked
 *     scores are SET to -1e4 (fully masked rows come out uniform, never NaN).  out: [B, n_heads*k_channels, T].
 *     Streaming softmax: no [T, T] tensor is materialised (the reference's `self.attn` is not produced).
 *     math (enum vs_conv_math, declared below): VS_MATH_BF16 rounds q / sqrt(dk), k, v and the probabilities to bf16 and runs both
 *     GEMMs on the bf16 matrix instruction with fp32 accumulation (softmax statistics and relative terms stay fp32) -- BASELINE.json's
 *     long-form bf16 configuration; VS_MATH_SPLIT3 (the default arithmetic of the path) / VS_MATH_SPLIT6: the same operands as two f16
 *     planes under power-of-two scales (per query, per K tile, running per V tile) with three cross products / as three bf16 planes with
 *     six -- fp32-class results on the f16 / bf16 matrix instruction, heads of up to 128 channels; VS_MATH_F32, and every shape those
 *     kernels do not take (T % 4 != 0, unaligned rows, wider heads in the split arithmetics): the exact-fp32 matrix instruction.        */
VS_API int vs_relattn_fwd(const float *q, const float *k, const float *v, int64_t qkv_batch_stride, const float *rel_k,
                          const float *rel_v, const float *mask, float *out, int64_t out_batch_stride, int64_t B,
                          int n_heads, int k_channels, int64_t T, int window_size, int n_heads_rel, int math, void *stream);

/* The same with the KEYS of every (batch, head) cut into `ksplit` ranges (1..16) that run as separate workgroups and are merged by a
 * second kernel: for launches too small to fill the chip (a single utterance is 16 workgroups of 32 serial key tiles).  `work`:
 * B * n_heads * ksplit * (k_channels + 2 + 2 * window + 1) * T floats of scratch.  Only the f16- / bf16-pipe kernels (VS_MATH_SPLIT3 /
 * VS_MATH_SPLIT6 / VS_MATH_BF16 on shapes they take) split; any other case runs exactly as vs_relattn_fwd.                           */
VS_API int vs_relattn_fwd_ksplit(const float *q, const float *k, const float *v, int64_t qkv_batch_stride, const float *rel_k,
                                 const float *rel_v, const float *mask, float *out, int64_t out_batch_stride, int64_t B, int n_heads,
                                 int k_channels, int64_t T, int window_size, int n_heads_rel, int math, float *work, int ksplit,
                                 void *stream);

/* The same with caller-provided scratch for PRE-PACKED keys / values (VS_MATH_BF16, long sequences): every 32- / 64-key tile of K and V
 * is converted to bf16 and laid out as the attention kernel's LDS image ONCE per launch (attn_pack_kv_kernel) instead of once per
 * 128-query block inside it (T / 128 times).  vs_relattn_kv_work_bytes() tells how many bytes `kv_work` (16-byte aligned) must hold for
 * a launch, 0 where pre-packing does not apply (other arithmetics, T < 1024, VS_NO_ATTN_KVPACK); with kv_work NULL or too small the
 * call behaves exactly as vs_relattn_fwd_ksplit.  Outputs are bit-identical either way (the same bf16 operands in the same order).   */
VS_API size_t vs_relattn_kv_work_bytes(int64_t B, int n_heads, int k_channels, int64_t T, int math);
VS_API int vs_relattn_fwd_work(const float *q, const float *k, const float *v, int64_t qkv_batch_stride, const float *rel_k,
                               const float *rel_v, const float *mask, float *out, int64_t out_batch_stride, int64_t B, int n_heads,
                               int k_channels, int64_t T, int window_size, int n_heads_rel, int math, float *work, int ksplit,
                               void *kv_work, size_t kv_work_bytes, void *stream);

/* a7  channel LayerNorm with its neighbours fused (rel_transformer.py:24-42; call sites 297-299, 305-307, 314-316):
 *     y = ((LayerNorm_C(a + r) * gamma + beta) + g) * mask      r, g, mask optional.
 *     g is the conditioning added at the top of the NEXT encoder layer: [B, C, T] (g_time_stride = 1,
 *     g_batch_stride = C*T) or [B, C, 1] broadcast over time (g_time_stride = 0, g_batch_stride = C).            */
VS_API int vs_layernorm_c_fwd(const float *a, const float *r, const float *gamma, const float *beta, const float *g,
                              int64_t g_batch_stride, int g_time_stride, const float *mask, float *y, int64_t B,
                              int64_t C, int64_t T, float eps, void *stream);

/* Training path: the elementwise neighbours of the convs as one forward and one backward launch each, where autograd
 * through PyTorch-ROCm ops took 10-25 small kernels.
 *   vs_gate_fwd / vs_gate_bwd: the WaveNet gate (modules/visinger/encoder.py:206-213 fused_add_tanh_sigmoid_multiply + the conditioning
 *     add of :177-180): acts[b, c, t] = tanh(x_in[b, c, t] + g[b, c]) * sigmoid(x_in[b, H + c, t] + g[b, H + c]).  x_in: [B, 2H, T];
 *     g: optional per-item bias, row b at g + b * g_bs (the layer's 2H-slice of the cond_layer output); acts / dacts: [B, H, T];
 *     dx_in: [B, 2H, T] = d loss / d x_in (= d / d g before the sum over t); dg (optional): += sum_t dx_in[b, :, t], row stride dg_bs.
 *   vs_layernorm_c_bwd: backward of y = LayerNorm_C(a + r) * gamma + beta (rel_transformer.py:33-42; r optional): dx [B, C, T] is the
 *     gradient w.r.t. a (and r); dgamma / dbeta are PER-ITEM partial sums: row b of a zero-initialised [B, 2, C] buffer holds
 *     (dgamma_b, dbeta_b) -- pass dgamma = buf, dbeta = buf + C -- accumulated with float atomics over the item's 64-frame blocks
 *     only (one [C] row for the whole batch serialised 128 workgroups on every address); the caller sums over b.                */
VS_API int vs_gate_fwd(const float *x_in, const float *g, int64_t g_bs, float *acts, int64_t B, int64_t H, int64_t T, void *stream);
VS_API int vs_gate_bwd(const float *x_in, const float *g, int64_t g_bs, const float *dacts, float *dx_in, float *dg, int64_t dg_bs,
                       int64_t B, int64_t H, int64_t T, void *stream);
/*   vs_wn_step_fwd / vs_wn_step_bwd: the residual / skip update behind a WaveNet layer's res_skip conv (modules/visinger/encoder.py:186-193):
 *     x_new = (x + rs[:, :H]) * mask, out_new = out_acc + rs[:, H:] with rs [B, 2H, T], x / out_acc / x_new / out_new [B, H, T], mask [B, T];
 *     out_acc NULL: the first layer (out_new = rs[:, H:]).  Backward: d_rs[:, :H] = dx = dx_new * mask, d_rs[:, H:] = dout_new (NULL inputs: zeros);
 *     the gradient of out_acc is dout_new itself.                                                                                              */
VS_API int vs_wn_step_fwd(const float *x, const float *rs, const float *out_acc, const float *mask, float *x_new, float *out_new, int64_t B,
                          int64_t H, int64_t T, void *stream);
/*   vs_l1_mean_fwd / vs_l1_mean_bwd: out[0] = mean |a - b| over n elements and d/da = sign(a - b) * gout[0] / n (the feature-matching loss terms,
 *     tasks/visinger.py:162-169), one launch each, deterministic.  work: 257 floats the caller keeps (work[256] starts as 0 and returns to 0);
 *     calls that share `work` must be ordered on one stream.                                                                                       */
VS_API int vs_l1_mean_fwd(const float *a, const float *b, float *work, float *out, int64_t n, void *stream);
VS_API int vs_l1_mean_bwd(const float *a, const float *b, const float *gout, float *da, int64_t n, void *stream);
VS_API int vs_wn_step_bwd(const float *dx_new, const float *dout_new, const float *mask, float *d_rs, float *dx, int64_t B, int64_t H,
                          int64_t T, void *stream);
/* gb[c] = sum over (b, t) of gy[b, c, t]: the bias gradient of every conv of the training path (what autograd computes for the `bias`
 * argument of torch.nn.functional.conv1d for the convs of modules/visinger under tasks/visinger.py:53-89), deterministic, one launch.     */
VS_API int vs_bias_grad(const float *gy, float *gb, int64_t B, int64_t C, int64_t T, void *stream);
/* torch.nn.utils.weight_norm (dim 0) of n tensors in one launch each way: w_i[r, :] = g_i[r] * v_i[r, :] / ||v_i[r, :]||, the row norms kept
 * for the backward (what torch._weight_norm / its backward compute per tensor; the reference applies weight_norm to every conv of
 * modules/visinger and the discriminators).  table: DEVICE memory, n rows of 8 x int64 {v, g, w, norm, rows, cols, first row, 0} with
 * first row = the running sum of `rows`; total_rows = the sum.  grads: DEVICE memory, n rows of 4 x int64 {gw, gv, gg, 0}: gw = dL/dw_i
 * (0: the tensor takes no gradient and its outputs are left untouched), gv / gg receive dL/dv_i and dL/dg_i.                                  */
VS_API int vs_weight_norm_multi_fwd(const void *table, int64_t n, int64_t total_rows, void *stream);
VS_API int vs_weight_norm_multi_bwd(const void *table, const void *grads, int64_t n, int64_t total_rows, void *stream);
VS_API int vs_layernorm_c_bwd(const float *a, const float *r, const float *gamma, const float *dy, float *dx, float *dgamma,
                              float *dbeta, int64_t B, int64_t C, int64_t T, float eps, void *stream);

/* Layout kernels of a strided dense conv run as the stride-1 conv of its input phases (the period / scale discriminators' stride-3 / stride-1
 * convs on vs_conv_forward, modules/discriminator.py:20-24, 55-60):
 *   vs_phase_stack:   x [N, C, T] (element strides sn, sc, st: any view) -> xf [stride * C][cols], cols >= N * Hq:
 *                     xf[r * C + c][n * Hq + j] = x[n][c][j * stride + r - pad], zero outside [0, T) and in the columns past the last
 *                     item -- N items end to end as ONE sequence;
 *   vs_phase_items:   to_sequence = 1: items src [N, C, Tv] -> dst [C][ld] at column n * Hq + j (zeros for Tv <= j < Hq);
 *                     to_sequence = 0: the inverse, sequence src [C][ld] -> items dst [N, C, Tv];
 *   vs_phase_unstack: the adjoint of vs_phase_stack, gxf [stride * C][ld] -> gx [N, C, T].                                              */
VS_API int vs_phase_stack(const float *x, int64_t sn, int64_t sc, int64_t st, float *xf, int64_t N, int64_t C, int64_t T, int stride, int pad,
                          int64_t Hq, int64_t cols, void *stream);
VS_API int vs_phase_items(const float *src, float *dst, int64_t N, int64_t C, int64_t Tv, int64_t Hq, int64_t ld, int to_sequence, void *stream);
VS_API int vs_phase_unstack(const float *gxf, int64_t ld, float *gx, int64_t N, int64_t C, int64_t T, int stride, int pad, int64_t Hq,
                            void *stream);

/* Training-mode attention core (rel_transformer.py:148-179 incl. the relative terms of :181-243 and the dropout of :173), streaming:
 * no [T, T] tensor in either direction.  q / k / v / out / dout / dq / dk / dv: [B, n_heads * k_channels, T] (batch strides given, 0 =
 * dense); rel_k / rel_v: [n_heads_rel, 2 * window + 1, k_channels] (window_size < 0: none); mask: [B, T] or NULL (scores of masked
 * (query, key) pairs are filled with -1e4 as the reference does; their gradient is zero).  Exact-fp32 MFMA; heads of <= 128 channels.
 *   forward:  out, and lse [2][B, n_heads, T] = per query the maximum score m and log sum_k exp(score - m) (saved for the backward; kept
 *             apart because a fully masked row has m = -1e4, where fp32 m + log(sum) would round the log away);
 *   dropout:  p_drop on the probabilities, mask = a counter-based hash of (seed, batch * head, query, key): pass the same seed to the
 *             backward.  Not torch's random stream (the reference's own masks are not reproducible across devices either);
 *   backward: dq, dk, dv; d rel_k / d rel_v as PARTIAL sums drel_*_part [B * n_heads * ceil(T / 32)][2 * window + 1][k_channels] that the
 *             caller reduces over the leading axis (per head, or over all heads when the tables are shared) -- deterministic, no atomics;
 *             work: B * n_heads * T * (3 + 4 * window) floats of scratch (row sums D, per-query relative dots).                       */
VS_API int vs_relattn_train_fwd(const float *q, const float *k, const float *v, int64_t qkv_batch_stride, const float *rel_k,
                                const float *rel_v, const float *mask, float *out, int64_t out_batch_stride, float *lse, int64_t B,
                                int n_heads, int k_channels, int64_t T, int window_size, int n_heads_rel, float p_drop, uint64_t seed,
                                void *stream);
VS_API int vs_relattn_train_bwd(const float *q, const float *k, const float *v, int64_t qkv_batch_stride, const float *rel_k,
                                const float *rel_v, const float *mask, const float *out, const float *dout, int64_t out_batch_stride,
                                const float *lse, float *dq, float *dk, float *dv, int64_t grad_batch_stride, float *work,
                                float *drel_k_part, float *drel_v_part, int64_t B, int n_heads, int k_channels, int64_t T, int window_size,
                                int n_heads_rel, float p_drop, uint64_t seed, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * f2  on-device linear / mel spectrograms (utils/audio/mel_processing.py:15-38: torchaudio Spectrogram / MelSpectrogram, power 2).
 *     The framed windowed DFT is a strided conv of the reflect-padded waveform with the (cos | -sin) * hann basis and the mel
 *     projection a 1x1 conv with the HTK filterbank: both run on vs_conv_forward (visinger_amd/audio.py builds the two bases).
 *     Between them:  p[b, f, t] = y[b, f, t]^2 + y[b, F + f, t]^2   on the transform output y [B, 2F, T] (rows f real, F + f imaginary),
 *     and for the mel loss on generated audio (tasks/base.py:232-238) its backward  dy[b, f] = 2 y[b, f] dp[b, f],
 *     dy[b, F + f] = 2 y[b, F + f] dp[b, f].                                                                                      */
VS_API int vs_spec_power_fwd(const float *y, float *p, int64_t B, int64_t F, int64_t T, void *stream);
VS_API int vs_spec_power_bwd(const float *y, const float *dp, float *dy, int64_t B, int64_t F, int64_t T, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * a13 grouped / strided Conv1d of the scale discriminator (modules/discriminator.py:55-60) and its gradients.
 *     x: [B, c_in, T], w: [c_out, c_in/groups, k], y / gy: [B, c_out, T_out], T_out = (T + 2*pad - k)/stride + 1.
 *     vs_gconv1d_bwd_weight writes one plane per batch item, gw_planes [B][c_out, c_in/groups, k]; gw = their sum.
 *     (The dense convs of both discriminators run on vs_conv_forward; strided ones through their de-interleaved phases.)  */
VS_API int vs_gconv1d_fwd(const float *x, const float *w, const float *bias, float *y, int64_t B, int64_t c_in, int64_t c_out,
                          int64_t T, int k, int stride, int pad, int groups, void *stream);
VS_API int vs_gconv1d_bwd_data(const float *gy, const float *w, float *gx, int64_t B, int64_t c_in, int64_t c_out, int64_t T,
                               int k, int stride, int pad, int groups, void *stream);
VS_API int vs_gconv1d_bwd_weight(const float *gy, const float *x, float *gw_planes, int64_t B, int64_t c_in, int64_t c_out,
                                 int64_t T, int k, int stride, int pad, int groups, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * a9  integer frame bookkeeping -- values are moved, never recomputed: bit-exact.
 *     vs_expand_states:  models/commons/align_ops.py:22-26.  out[b, t, :] = mel2token[b, t] ? h[b, mel2token[b,t]-1, :] : 0.
 *                        h: [B, T_tokens, C] (h_channels_first = 0) or [B, C, T_tokens] (1); out likewise with T_frames.
 *     vs_make_positions: modules/rel_transformer.py:78-88.  positions = cumsum(x != pad) * (x != pad) + pad  (int64),
 *                        x: [B, T] fp32 (the channel-0 slice the callers pass), pad = padding_idx.
 *     vs_slice_segments: modules/commons/utils.py:86-92.  out[b, :, s] = x[b, :, ids_str[b] + s],  x: [B, C, T].
 *     vs_mel2token_to_dur: utils/audio/align.py:105-129.  dur[b, i-1] = #{t : mel2token[b, t] == i}, i in 1..T_tokens
 *                        (index 0 = padding, dropped); clamped to max_dur when max_dur >= 0.  int64 in, int64 out.        */
VS_API int vs_expand_states(const float *h, const int64_t *mel2token, float *out, int64_t B, int64_t T_tokens,
                            int64_t T_frames, int64_t C, int h_channels_first, int out_channels_first, void *stream);
VS_API int vs_make_positions(const float *x, int64_t *positions, int64_t B, int64_t T, int64_t padding_idx, void *stream);
VS_API int vs_slice_segments(const float *x, const int64_t *ids_str, float *out, int64_t B, int64_t C, int64_t T,
                             int64_t segment_size, void *stream);
VS_API int vs_mel2token_to_dur(const int64_t *mel2token, int64_t *dur, int64_t B, int64_t T_frames, int64_t T_tokens,
                               int64_t max_dur, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* VISINGER_HIP_H */
