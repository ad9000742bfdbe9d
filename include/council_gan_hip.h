/*
 * council_gan_hip.h -- C-ABI of the MI355X (gfx950) Council-GAN hot-path library.
 *
 * This is the drop-in boundary.  The reference (Onr/Council-GAN) has no
 * FFI of its own: all arithmetic of its hot path is dispatched from `networks.py` /
 * `trainer_council.py` into PyTorch/ATen/cuDNN.  Each entry point below names the reference
 * call site (file:line in /root/reference) whose arithmetic it replaces.
 *
 * Conventions
 *   - every tensor is fp32, physically NHWC (PyTorch `channels_last` strides over a logical
 *     NCHW shape), dense; weights are physically [Cout][KH][KW][Cin] (channels_last OIHW);
 *   - raw device pointers + explicit sizes, no torch types; the caller (PyTorch's caching
 *     allocator) owns every buffer including workspaces;
 *   - kernels are enqueued on `stream` and never synchronise.  The data path has no global mutable state: what a
 *     call computes depends on its arguments only, and calls are reentrant per stream (per-launch side channels
 *     are thread-local and live for one call).  The ONE piece of process-wide state is the kernel-selection table
 *     `cg_tuning` below -- which of several result-equivalent kernels / tile shapes a launch gets -- read from the
 *     CG_* environment once, at first use, and changed only through cg_tuning_set(); plus the opt-in per-launch
 *     timing table of cg_prof_enable (measurement only) and the RCCL communicator handles the caller owns;
 *   - return value 0 = ok, negative = error (see cg_last_error()); nothing throws across the ABI.
 */
#ifndef COUNCIL_GAN_HIP_H
#define COUNCIL_GAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* cg_stream_t; /* hipStream_t */

#define CG_OK 0
#define CG_ERR_ARG (-1)
#define CG_ERR_LAUNCH (-2)
#define CG_ERR_WORKSPACE (-3)

#define CG_ACT_NONE 0
#define CG_ACT_RELU 1
#define CG_ACT_LRELU 2 /* LeakyReLU(0.2), networks.py:497 */
#define CG_ACT_TANH 3

#define CG_MAX_TAPS 64

/* Geometry of one implicit-GEMM convolution pass (forward, data-gradient or weight-gradient).
 * The input is gathered from up to two NHWC sources concatenated along C (networks.py:152
 * torch.cat((x, x_input), 1)), optionally through a nearest 2x upsample (networks.py:385),
 * at `T` taps; output position (oy, ox) of the enumerated Ho x Wo grid reads logical input
 * (oy*stride + dy[t], ox*stride + dx[t]) -- zero outside (ZeroPad2d, networks.py:474) -- and
 * is written to (oy*osy + ooy, ox*osx + oox) of an HoF x WoF output tensor.  Forward conv:
 * dy[t] = kh - pad, osy = 1.  Data-gradient of a stride-2 conv: four parity classes, each a
 * stride-1 pass with osy = osx = 2. */
typedef struct cg_conv_geom {
    int32_t N, H, W;    /* stored spatial dims of the source tensor(s) */
    int32_t C1, C2;     /* channels of source 1 / source 2 (0 = no second source) */
    int32_t up;         /* 1: sources are read through a nearest 2x upsample */
    int32_t Ho, Wo;     /* enumerated output grid */
    int32_t HoF, WoF;   /* output tensor spatial dims */
    int32_t osy, osx, ooy, oox;
    int32_t stride;
    int32_t T;          /* number of taps (<= CG_MAX_TAPS) */
    int32_t Cout;
    int32_t act;        /* CG_ACT_* applied in the epilogue (forward only) */
    int8_t dy[CG_MAX_TAPS];
    int8_t dx[CG_MAX_TAPS];
} cg_conv_geom;

/* Member-batched ("grouped") launches: the SAME layer of `n` council members as one launch.  The members' activations
 * are consecutive sample blocks of one batched NHWC tensor (member z owns samples [z*N/n, (z+1)*N/n) of every activation /
 * gradient tensor, geometry N = all members' samples); their parameters -- weights, biases and the gradients of both --
 * sit `stride` fp32 ELEMENTS apart (one pool per optimizer kind; for split {hi, lo} weights the same stride in elements,
 * a multiple of 32).  Pointers name member 0's tensors.  NULL or n == 1: an ordinary call.  Replaces the reference's
 * sequential member loops (trainer_council.py:328,558,747,826,858): members are independent models. */
typedef struct cg_group {
    int32_t n;
    int32_t reserved;
    int64_t stride;
} cg_group;

const char* cg_last_error(void);
int cg_version(void);

/* ---- convolution (nn.Conv2d after ZeroPad2d, networks.py:513,515-516; bare 1x1 at :44,142-143,348;
 *      nn.Linear of the MLP as a 1x1 conv on an N x 1 x 1 x C tensor, networks.py:531) ---------- */

/* y = act(conv(x) + bias).  w packed [Cout][T][C1+C2]; bias may be NULL. */
int cg_conv2d_fwd(const cg_conv_geom* g, const float* x1, const float* x2, const float* w,
                  const float* bias, float* y, cg_stream_t stream);

/* Forward conv that ALSO emits instance-norm partial statistics of y from its epilogue when the fused path applies
 * (pipelined kernel, act == none, H*W a multiple of the block's row tile): stats[((n*S + s)*Cout + c)*2 + {0,1}] =
 * {sum y, sum y^2} over rows [s*R, (s+1)*R) of sample n, R returned in *rows_per_partial (0 = not fused: run
 * cg_instnorm_stats).  stats_bytes >= ceil(M/64)*Cout*16 always suffices.  Conv2dBlock conv -> norm, networks.py:515-518. */
int cg_conv2d_fwd_stats(const cg_conv_geom* g, const float* x1, const float* x2, const float* w, const float* bias,
                        float* y, double* stats, size_t stats_bytes, int* rows_per_partial, cg_stream_t stream);

/* ---- split-precision ("fp16 x 3") convolutions -------------------------------------------
 * cg_split_f16: every x[i] -> hi = f16(scale*x[i]), lo = f16(scale*x[i] - hi), round-to-nearest-even, stored in the
 * {hi, lo} layout described below.
 * cg_conv2d_fwd_x3: y = act(conv(x) / w_scale + bias) with x and w given in that form (x: NHWC, w: [Cout][T][C]
 * pre-multiplied by the power of two w_scale), a*b evaluated as ah*bh + ah*bl + al*bh on the fp16 MFMA with fp32
 * accumulation (22 significand bits: error below the fp32 kernel's accumulation round-off); same geometry, epilogue
 * and optional instance-norm partials as cg_conv2d_fwd_stats.  tile_cfg -1 = heuristic. */
#define CG_X3_WSCALE 1024.0f
/* Layout of a {hi, lo} tensor.  The library is built with the halves interleaved per 32 elements: element i (flat
 * physical index; every split tensor has a channel count that is a multiple of 32) has its hi half at
 * 64*(i/32) + i%32 and its lo half CG_X3_LO_ELEMS halves further on, so one 32-channel K-slice of a pixel / weight row
 * is one 128-byte line.  Every `*_lo_elems` argument below must then be CG_X3_LO_ELEMS, and a sub-tensor that starts
 * at element `off` (a multiple of 32) starts 4*off bytes into the buffer.  cg_x3_interleaved() returns 0 for an A/B
 * build with two separate planes (-DCG_X3_INTERLEAVE=0), where `*_lo_elems` is the distance of the lo plane in halves. */
#define CG_X3_LO_ELEMS 32
int cg_x3_interleaved(void);
int cg_split_f16(const float* x, void* out, size_t n, size_t lo_elems, float scale, cg_stream_t stream);
int cg_conv2d_fwd_x3(const cg_conv_geom* g, const void* x_hi, size_t x_lo_elems, const void* w_hi, size_t w_lo_elems,
                     float w_scale, const float* x_scale_dev, const float* bias, float* y, void* y_split,
                     size_t y_lo_elems, double* stats, size_t stats_bytes, int* rows_per_partial, int tile_cfg,
                     float* amax_state, int* amax_nslots, cg_stream_t stream);
/* Grouped / general form.  w_scale_dev: device-side power-of-two scale the weights were split with
 * (cg_split_f16_dynamic_capped; multiplies the static w_scale; NULL = static only). */
int cg_conv2d_fwd_x3_g(const cg_conv_geom* g, const cg_group* group, const void* x_hi, size_t x_lo_elems, const void* w_hi,
                       size_t w_lo_elems, float w_scale, const float* w_scale_dev, const float* x_scale_dev,
                       const float* bias, float* y, void* y_split, size_t y_lo_elems, double* stats, size_t stats_bytes,
                       int* rows_per_partial, int tile_cfg, float* amax_state, int* amax_nslots, cg_stream_t stream);
/* fp32 forward, grouped / general form: instance-norm partials (stats, rows_per_partial) and per-block output maxima
 * (amax_state, amax_nslots) are optional services, NULL = off (cg_conv2d_fwd_stats / cg_conv2d_fwd_amax semantics). */
int cg_conv2d_fwd_g(const cg_conv_geom* g, const cg_group* group, const float* x1, const float* x2, const float* w,
                    const float* bias, float* y, double* stats, size_t stats_bytes, int* rows_per_partial,
                    float* amax_state, int* amax_nslots, cg_stream_t stream);
/* The thin-input first layers (3 / 6 / 12 -> 64 channels: the generators' 7x7, the discriminators' 4x4 stride-2 and two-source 3x3,
 * the decoder's 12 -> 64 1x1; networks.py:44,152,385-386) under the split-precision datapath: fp32 tensors in and out exactly like
 * cg_conv2d_fwd_g (no instance-norm partials), the products evaluated as wh*xh + wh*xl + wl*xh on the fp16 MFMA from {hi, lo}
 * halves built inside the kernel.  cg_conv2d_fwd_thin_x3_ok(g) != 0 iff the geometry is one of those layers. */
int cg_conv2d_fwd_thin_x3_ok(const cg_conv_geom* g);
int cg_conv2d_fwd_thin_x3_g(const cg_conv_geom* g, const cg_group* group, const float* x1, const float* x2, const float* w,
                            const float* bias, float* y, float* amax_state, int* amax_nslots, cg_stream_t stream);
/* amax_state / amax_nslots (both or neither): the epilogue leaves max|y| per block in amax_state[2 .. 2 + *amax_nslots)
 * (a CG_SPLIT_STATE_FLOATS buffer) for cg_split_f16_dynamic(y, ..., state, nslots), which then skips its own reduction
 * pass over y (launches with more than 1024 blocks share 1024 slots through an atomic max); *amax_nslots = 0 when the
 * launch cannot provide them (strided output classes).  cg_conv2d_fwd_amax is
 * cg_conv2d_fwd with the same service (the 3/6-channel first layers, whose outputs feed split-precision layers). */
int cg_conv2d_fwd_amax(const cg_conv_geom* g, const float* x1, const float* x2, const float* w, const float* bias, float* y,
                       float* amax_state, int* amax_nslots, cg_stream_t stream);
/* Tensors of arbitrary magnitude (gradients, un-normalised activations) are split with a per-tensor power-of-two scale
 * chosen ON THE DEVICE: state[0] <- max|x|, state[1] <- scale = 2^(12 - floor(log2 max|x|)) (scaled peak in [4096, 8192): three binades below fp16's maximum);
 * state[2 .. CG_SPLIT_STATE_FLOATS) is scratch (per-block maxima).  The halves hold scale*x;
 * consumers take `state + 1` as x_scale_dev / dz_scale_dev and undo the scale in their epilogue.  No host sync. */
#define CG_SPLIT_STATE_FLOATS 1026
int cg_split_f16_dynamic(const float* x, void* out, size_t n, size_t lo_elems, float* state, int nslots,
                         cg_stream_t stream);   /* nslots > 0: a producer kernel already left that many per-block maxima
                                                   in state[2..] (cg_instnorm_bwd, cg_conv2d_fwd_amax, cg_conv2d_fwd_x3);
                                                   0: measure here */
/* The same with the scale capped: scale = min(max_scale, 2^(12 - floor(log2 max|x|))), max_scale a power of two (0 = no
 * cap).  Used for the WEIGHTS of an optimizer pool with max_scale = CG_X3_WSCALE: ordinary weights (|w| < 8) keep
 * the static 2^10, larger ones (a loaded checkpoint) get the smaller scale that keeps their hi halves finite. */
int cg_split_f16_dynamic_capped(const float* x, void* out, size_t n, size_t lo_elems, float* state, int nslots,
                                float max_scale, cg_stream_t stream);
/* dz = dy * act'(y) straight into split form (cg_act_bwd + cg_split_f16_dynamic without the fp32 round trip); dz
 * (optional) additionally receives the fp32 values.  dy_nslots > 0: state[2 .. 2 + dy_nslots) already holds per-block maxima
 * of dy left by its producer (cg_conv2d_dgrad_x3_run) -- |dz| <= |dy| for relu / lrelu / tanh, so the scale is taken from
 * them and the measuring pass is skipped (unless dz is wanted in fp32 as well). */
int cg_act_bwd_split(const float* dy, const float* y, size_t n, int act, void* out, size_t lo_elems, float* state,
                     int dy_nslots, float* dz, cg_stream_t stream);
/* split-precision data gradient (cg_conv2d_dgrad with dz pre-split by cg_split_f16_dynamic; needs Cout % 32 == 0);
 * ws: cg_conv2d_dgrad_workspace(g, nci) bytes (holds the re-laid-out, split weights) */
int cg_conv2d_dgrad_x3(const cg_conv_geom* g, const void* dz_split, size_t dz_lo_elems, const float* dz_scale_dev,
                       const float* w, int ci0, int nci, float* dx, void* ws, size_t ws_bytes, cg_stream_t stream);
/* The two halves of it, so that the re-laid-out weights are prepared ONCE per weight version instead of once per launch
 * (grouped: all members in one call):
 *   _prep: w [Cout][T][Cin] fp32 of every member -> {hi, lo} planes of scale*w in the per-class [ci][tc][co] layout the
 *          data-gradient kernel reads; member m at wt + 4*m*cg_conv2d_dgrad_x3_wt_elems(g, nci) bytes.
 *          scale = w_scale, times the device-side *w_scale_dev when given.
 *   _run:  dx from dz and the prepared weights (same w_scale / w_scale_dev). */
size_t cg_conv2d_dgrad_x3_wt_elems(const cg_conv_geom* g, int nci);
int cg_conv2d_dgrad_x3_prep(const cg_conv_geom* g, const cg_group* group, const float* w, int ci0, int nci, float w_scale,
                            const float* w_scale_dev, void* wt, size_t wt_bytes, cg_stream_t stream);
int cg_conv2d_dgrad_x3_run(const cg_conv_geom* g, const cg_group* group, const void* dz_split, size_t dz_lo_elems,
                           const float* dz_scale_dev, const void* wt, float w_scale, const float* w_scale_dev, int ci0,
                           int nci, float* dx, float* amax_state, int* amax_nslots, cg_stream_t stream);
/* ---- nearest-2x upsample + 3x3 convolution (nn.Upsample(scale_factor=2) in front of a Conv2dBlock, networks.py:385-386,
 *      513-516) evaluated as the 4x4 stride-2 TRANSPOSED convolution it is: the 3x3 taps that read the same source pixel are
 *      added up first (fp32), 16 effective taps per 4 output pixels instead of 36 -- 2.25x fewer multiply-adds, results equal
 *      up to fp32 rounding of the tap sums.  W_F[u][v] = sum_{kh in S(u), kw in S(v)} W[kh][kw], S(0) = {2}, S(1) = {1,2},
 *      S(2) = {0,1}, S(3) = {0}.
 *   cg_upconv_prep_x3:   w [Cout][3][3][Cin] fp32 of every member -> {hi, lo} planes of scale * W_F, scale = w_scale (x the
 *                        device value *w_scale_dev when given), in two layouts of cg_upconv_wt_elems(Cout, Cin) elements per
 *                        member (4 bytes each): wt_fwd = the four output-parity classes [cls][Cout][2x2 taps][Cin] the forward
 *                        reads, wt_bwd = [Cin][4][4][Cout], the weight of the 4x4 stride-2 pad-1 convolution over dz that IS the
 *                        data gradient (run it with cg_conv2d_fwd_x3_g; its weight gradient with cg_conv2d_wgrad_x3_g, input dz,
 *                        output gradient x, gives dW_F).  Either output may be NULL.
 *   cg_upconv2d_fwd_x3:  y = conv(upsample2x(x)) + bias from x in {hi, lo} form and wt_fwd; g = the geometry of the 3x3 layer
 *                        on the upsampled source (up = 1).  stats / rows_per_partial as cg_conv2d_fwd_stats.
 *   cg_upconv_fold_dw:   dw[co][kh][kw][ci] (+)= sum_{u: kh in S(u), v: kw in S(v)} dwf[ci][u][v][co]   (members: dwf packed,
 *                        dw at group->stride).
 *   cg_colsum_split:     db[c] (+)= sum over rows of a {hi, lo} tensor [rows_total][C] / *scale_dev -- the bias gradient of a layer
 *                        whose dz exists in split form only (rows_total = all members' rows, member-major). */
/* Two 1x1 convolutions with nothing between them, composed: the council discriminator ends in Conv2d(dim, dim, 1) -> Conv2d(dim, 1, 1)
 * with no activation in between (networks.py:142-143), i.e. W2 (W1 y + b1) + b2 = (W2 W1) y + (W2 b1 + b2): one dim -> 1 convolution.
 *   cg_compose1x1_fwd: out (per member, out_stride floats) = {w_eff[0 .. C), b_eff}; W1 [C][C] (row = output channel), W2 [C].
 *   cg_compose1x1_bwd: d = {d w_eff, d b_eff} (the gradients the dim -> 1 convolution produced) -> ACCUMULATES dW1[j][k] += W2[j] d
 *                      w_eff[k], db1[j] += W2[j] d b_eff, dW2[j] += sum_k d w_eff[k] W1[j][k] + d b_eff b1[j], db2 += d b_eff.
 * Members: the four parameters and their gradient buffers at group->stride (one optimizer pool). */
int cg_compose1x1_fwd(const cg_group* group, const float* W1, const float* b1, const float* W2, const float* b2, int C, float* out,
                      int out_stride, cg_stream_t stream);
int cg_compose1x1_bwd(const cg_group* group, const float* d, int d_stride, const float* W1, const float* b1, const float* W2, int C,
                      float* dW1, float* db1, float* dW2, float* db2, cg_stream_t stream);
/* {hi, lo} planes -> fp32: out[i] = (hi[i] + lo[i]) / *scale_dev (scale_dev NULL = 1) in the tensor's physical element order */
int cg_unsplit_f16(const void* z_split, size_t lo_elems, const float* scale_dev, float* out, size_t n, cg_stream_t stream);
size_t cg_upconv_wt_elems(int Cout, int Cin);
int cg_upconv_prep_x3(const cg_group* group, const float* w, int Cout, int Cin, float w_scale, const float* w_scale_dev,
                      void* wt_fwd, void* wt_bwd, cg_stream_t stream);
int cg_upconv2d_fwd_x3(const cg_conv_geom* g, const cg_group* group, const void* x_split, size_t x_lo_elems,
                       const float* x_scale_dev, const void* wt_fwd, float w_scale, const float* w_scale_dev, const float* bias,
                       float* y, double* stats, size_t stats_bytes, int* rows_per_partial, cg_stream_t stream);
int cg_upconv_fold_dw(const cg_group* group, const float* dwf, float* dw, int Cout, int Cin, int accumulate, cg_stream_t stream);
size_t cg_colsum_split_workspace(int C, int nmember);
int cg_colsum_split(const cg_group* group, const void* z_split, size_t lo_elems, const float* scale_dev, long rows_total, int C,
                    float* db, int accumulate, void* ws, size_t ws_bytes, cg_stream_t stream);
/* ---- bounded split: un-normalised outputs leave the split-precision kernels as {hi, lo} planes -------------------------------
 * A discriminator activation (networks.py:44-52, 142-152: Conv2dBlock + LeakyReLU, no norm) or a gradient has no natural scale,
 * so its consumer used to measure it and split it in a pass of its own (cg_split_f16_dynamic).  A bound is known BEFORE the
 * producing kernel runs: |y| <= L1 * max|x| + max|bias| with L1 = the largest row (forward) / column (data gradient) 1-norm of
 * the weights (cg_weight_l1_bound, once per weight version) and max|x| = the measured maximum the input's producer left behind.
 * Every block derives the same power-of-two scale 2^(14 - floor(log2 bound)) from it and the epilogue writes the planes --
 * overflow-free by construction, the scaled peak in practice 2^3 ... 2^6 below 2^15 (signed sums do not reach their 1-norm
 * bound), i.e. every element within 2^-12 of the peak still carries 22 bits.  No reference counterpart (the reference keeps fp32).
 *   l1_ctl     device float[2] from cg_weight_l1_bound; NULL: no bounded split (act_src may still be set)
 *   in_state   split state of the INPUT operand: [0] = its measured max |x| -- or, in_nslots > 0: [2 .. 2 + in_nslots) holds
 *              per-block maxima a producer left (cg_conv2d_fwd_x3_e itself, cg_conv2d_fwd_amax, cg_instnorm_bwd ...)
 *   act_src    hi plane ({hi, lo} form, same geometry as the output) of the tensor the output is the gradient OF; the output is
 *              multiplied by act'(act_src), act_type = CG_ACT_RELU / CG_ACT_LRELU (sign-only derivatives): the activation
 *              backward of the layer below, fused into the data-gradient epilogue (trainer_council.py:779,882 loss.backward())
 *   out_state  CG_SPLIT_STATE_FLOATS floats: [0] = the bound, [1] = the scale of the planes, [2 ..] = block maxima of the outputs
 *   addend     fp32 tensor of the output's own geometry (NHWC) added to every output value after bias / activation factor, or
 *              NULL: the gradient that reaches a ResBlock's input through its skip connection (`out += residual`,
 *              networks.py:459-460) joins the data gradient of the block's first convolution in that kernel's epilogue instead
 *              of in a pass of its own; bit for bit the separate fp32 addition (the value is rounded to fp32 first).  Block
 *              maxima / planes / statistics are those of the SUM. */
typedef struct cg_x3_epilogue {
    const float* l1_ctl;
    const float* in_state;
    int32_t in_nslots;
    int32_t act_type;
    const void* act_src;
    float* out_state;
    const float* addend;
} cg_x3_epilogue;
size_t cg_weight_l1_workspace(int Cin, int nmember);      /* for by_input_channel = 1 (deterministic two-stage column sums) */
int cg_weight_l1_bound(const cg_group* group, const float* w, int Cout, int T, int Cin, const float* bias, int by_input_channel,
                       float* out2, void* ws, size_t ws_bytes, cg_stream_t stream);
int cg_conv2d_fwd_x3_e(const cg_conv_geom* g, const cg_group* group, const void* x_split, size_t x_lo_elems, const void* w_split,
                       size_t w_lo_elems, float w_scale, const float* w_scale_dev, const float* x_scale_dev, const float* bias,
                       float* y /* may be NULL */, void* y_split, size_t y_lo_elems, const cg_x3_epilogue* epi, int tile_cfg,
                       int* amax_nslots, cg_stream_t stream);
int cg_conv2d_dgrad_x3_run_e(const cg_conv_geom* g, const cg_group* group, const void* dz_split, size_t dz_lo_elems,
                             const float* dz_scale_dev, const void* wt, float w_scale, const float* w_scale_dev, int ci0, int nci,
                             float* dx /* may be NULL */, void* dx_split, size_t dx_lo_elems, const cg_x3_epilogue* epi,
                             float* amax_state, int* amax_nslots, cg_stream_t stream);
/* split-precision weight gradient: cg_conv2d_wgrad with x and dz given in {hi, lo} form (+ device-side scales,
 * NULL = 1).  cg_conv2d_wgrad_x3_ok(g) != 0 iff the layer qualifies (one source, every k-tile inside one tap,
 * channel counts multiples of 32, power-of-two output plane); workspace as cg_conv2d_wgrad_workspace(g). */
int cg_conv2d_wgrad_x3_ok(const cg_conv_geom* g);
int cg_conv2d_wgrad_x3(const cg_conv_geom* g, const void* x_split, size_t x_lo_elems, const float* x_scale_dev,
                       const void* dz_split, size_t dz_lo_elems, const float* dz_scale_dev, float* dw, float* dbias,
                       int accumulate, void* ws, size_t ws_bytes, cg_stream_t stream);
int cg_conv2d_wgrad_x3_ok_g(const cg_conv_geom* g, const cg_group* group);
int cg_conv2d_wgrad_x3_g(const cg_conv_geom* g, const cg_group* group, const void* x_split, size_t x_lo_elems,
                         const float* x_scale_dev, const void* dz_split, size_t dz_lo_elems, const float* dz_scale_dev,
                         float* dw, float* dbias, int accumulate, void* ws, size_t ws_bytes, cg_stream_t stream);
/* y_split of cg_conv2d_fwd_x3 (optional): the output ALSO in {hi, lo} form for a convolution that consumes it next */
/* instance norm / AdaIN apply that writes the split form of its output (and / or fp32): the producer side of
 * cg_conv2d_fwd_x3.  y may be NULL (split only); y_split holds N*HW*C {hi, lo} pairs in the layout above. */
int cg_instnorm_apply_split(const float* x, const float* mean, const float* rstd, const float* gamma,
                            const float* beta, int gstride, const float* residual, float* y, void* y_split,
                            size_t y_lo_elems, int N, int HW, int C, int act, cg_stream_t stream);

/* Same, with the block-tile configuration forced (tuning / A-B benchmarking hook; -1 = heuristic). */
int cg_conv2d_fwd_tile(const cg_conv_geom* g, const float* x1, const float* x2, const float* w,
                       const float* bias, float* y, int tile_cfg, cg_stream_t stream);

/* dW[Cout][T][C1+C2] (+)= sum over output positions of dz (x) gathered input; geometry as the
 * forward pass.  `ws` holds split-K partials: cg_conv2d_wgrad_workspace() bytes.  accumulate != 0
 * adds into dw (and dbias).  dbias (optional) = column sums of dz, accumulated inside the same kernel. */
size_t cg_conv2d_wgrad_workspace(const cg_conv_geom* g);
int cg_conv2d_wgrad(const cg_conv_geom* g, const float* x1, const float* x2, const float* dz, float* dw,
                    float* dbias, int accumulate, void* ws, size_t ws_bytes, cg_stream_t stream);
/* grouped: member z's gradient goes to dw + z*stride (dbias likewise); a member's split plan and summation order do not
 * depend on the number of members in the launch */
size_t cg_conv2d_wgrad_workspace_g(const cg_conv_geom* g, const cg_group* group);
int cg_conv2d_wgrad_g(const cg_conv_geom* g, const cg_group* group, const float* x1, const float* x2, const float* dz,
                      float* dw, float* dbias, int accumulate, void* ws, size_t ws_bytes, cg_stream_t stream);
/* the same for a layer whose data gradient nobody needs (the discriminators' first layers in dis_update / dis_council_update,
 * trainer_council.py:779,882: the input is an image): dy is the gradient of the ACTIVATED output y = act(conv(x)) and the
 * activation backward dz = dy * act'(y) (networks.py:44-47, the LeakyReLU behind the first Conv2dBlock) happens while the
 * kernel loads dz -- no pass that writes dz.  Only the thin-input layers (<= 12 input channels, 64 outputs) take it:
 * cg_conv2d_wgrad_act_ok says whether g does; other geometries are refused.  Bit-identical to cg_act_bwd + cg_conv2d_wgrad_g. */
int cg_conv2d_wgrad_act_ok(const cg_conv_geom* g);
int cg_conv2d_wgrad_act_g(const cg_conv_geom* g, const cg_group* group, const float* x1, const float* x2, const float* dy,
                          const float* y, int act, float* dw, float* dbias, int accumulate, void* ws, size_t ws_bytes,
                          cg_stream_t stream);

/* dx (w.r.t. input channels [ci0, ci0+nci) of the convolution described by the FORWARD geometry g) from
 * dz [N, Ho, Wo, Cout]; dx is the dense [N, H<<up, W<<up, nci] tensor at the resolution the taps see (a nearest-2x
 * input is folded back by cg_upsample2x_bwd).  Weights are re-laid-out into `ws` (cg_conv2d_dgrad_workspace
 * bytes) and the stride^2 output-parity passes run as one launch.  networks.py:513 (autograd of nn.Conv2d). */
size_t cg_conv2d_dgrad_workspace(const cg_conv_geom* g, int nci);
int cg_conv2d_dgrad(const cg_conv_geom* g, const float* dz, const float* w, int ci0, int nci, float* dx, void* ws,
                    size_t ws_bytes, cg_stream_t stream);
size_t cg_conv2d_dgrad_workspace_g(const cg_conv_geom* g, const cg_group* group, int nci);
int cg_conv2d_dgrad_g(const cg_conv_geom* g, const cg_group* group, const float* dz, const float* w, int ci0, int nci,
                      float* dx, void* ws, size_t ws_bytes, cg_stream_t stream);

/* Kernel-selection table: the library's only process-wide state (see "Conventions").  Every field selects among
 * kernels / tile shapes whose results agree to the last bits (fp32 datapath: bit for bit); none changes what a call
 * means.  Defaults come from the environment variable named with each field, read once at the first use of the library.
 * A multi-threaded host sets the table before its first launch.  There is no reference counterpart (the reference
 * leaves algorithm choice to cuDNN, train.py:61 `cudnn.deterministic`). */
typedef struct cg_tuning {
    int32_t fwd_thin;        /* CG_FWD_THIN (1): thin-input first layers on the spatial-tile kernel */
    int32_t wgrad_thin;      /* CG_WGRAD_THIN (1): their weight gradient on conv_wgrad_thin_kernel */
    int32_t wgrad_x3_bm256;  /* CG_WGRAD_X3_BM256 (2): 256 x 128 weight-gradient tile: 0 never, 1 always, 2 where measured to win */
    int32_t wgrad_x3_wide;   /* CG_WGRAD_X3_WIDE (2): 256 x 256 LDS-DMA weight-gradient tile: 0 never, 1 always, 2 where measured to win */
    int32_t wgrad_x3_perm;   /* CG_WGRAD_X3_PERM (0): v_perm loader instead of the transposing LDS read */
    int32_t wgrad_legacy;    /* (0): non-pipelined fp32 weight-gradient kernel */
    int32_t x3_wide;         /* CG_X3_WIDE (16): wide LDS-DMA forward tile: 0 never, 16 256x256 where it wins, 17 256x128, 1 both */
    int32_t x3_thin_out;     /* CG_X3_THIN_OUT (20): tile for <= 32 output channels: 0 off, 20 = 128x32, 21 = 256x32 */
    int32_t x3_korder;       /* CG_X3_KORDER (0): channel-slice-major K order: 1 = the wide LDS-DMA tile, 2 = the register-staged tiles too.
                              * Memory-side traffic of the wide tile 5.0x -> 1.13x algorithmic, time +1.3 ... 3 %: measured, off */
    int32_t tile_rows_scale; /* CG_TILE_ROWS_SCALE (1): TEST HOOK -- choose tiles as if a launch had k x its rows, so that a
                              * batch-1 parity run exercises the tiles the batch-k benchmark selects */
    int32_t no_amax_atomic;  /* CG_NO_AMAX_ATOMIC (0): launches with > 1024 blocks do not report output maxima */
    int32_t wgrad_x3_multitap; /* CG_WGRAD_X3_MULTITAP (1): 128-wide K-tiles spanning several taps for 32 / 64 input channels */
    int32_t x3_cls_minor;    /* CG_X3_CLS_MINOR (1): multi-class launches (strided data gradients, upsample-convolutions) order their blocks
                              * class-minor, so that the output-parity classes of one row tile share an XCD's L2 */
    int32_t x3_generic_epilogue; /* CG_X3_GENERIC_EPILOGUE (0): A/B switch -- 1 = the split-precision forward / data-gradient kernels always take
                              * the generic copy of their epilogue value loop (round 4's code path) instead of the copies compiled for
                              * {no activation, LeakyReLU} x {with, without statistics}; results are bit-identical either way */
    int32_t wgrad_xcd_group; /* CG_WGRAD_XCD_GROUP (1): split-precision weight-gradient grids order their blocks so that the tap tiles of one
                              * (member, position range) -- which stream the same dz rows and overlapping x rows -- run on ONE XCD and meet in
                              * its L2 instead of being dealt round-robin over all eight (bit-identical results) */
    int32_t fp32_chunked_sum; /* CG_FP32_CHUNKED_SUM (1): the exact-fp32 forward / data-gradient kernel sums its K range in chunks of four
                              * K-slices (blocked summation: ~3.7x less accumulation round-off than one chain of K/2 MFMAs); 0 = one chain
                              * (round 1-5 arithmetic).  Changes the last bits of fp32-datapath results, nothing else */
} cg_tuning;
int cg_tuning_get(cg_tuning* out);
int cg_tuning_set(const cg_tuning* in);

/* Single-field wrappers over cg_tuning kept for the A/B tools; each returns the previous setting. */
/* A/B switch: force the non-pipelined weight-gradient kernel (tuning / regression checks only). */
int cg_conv2d_wgrad_legacy(int on);
/* Split-precision weight gradients of layers with Cout % 256 == 0 on a 256 x 128 tile / 16 waves (also CG_WGRAD_X3_BM256):
 * 0 = never, 1 = wherever the layer qualifies, 2 (default) = where it was measured to win (>= 64 such tiles over all
 * members).  Returns the previous mode.  Workspace queries follow it. */
int cg_conv2d_wgrad_x3_bm256(int mode);
/* 256 x 256 LDS-DMA tile of the split-precision weight gradient for layers with Cout % 256 == 0 and C1 % 256 == 0 (also
 * CG_WGRAD_X3_WIDE): 0 = never, 1 = wherever the layer qualifies, 2 (default) = from 16 such tiles over all members.
 * Returns the previous mode.  Workspace queries follow it. */
int cg_conv2d_wgrad_x3_wide(int mode);
/* A/B switch (also CG_FWD_THIN=0; on by default since round 3): the thin-input layers (3 / 6 / 12 -> 64 channels: the generators' 7x7
 * and the discriminators' 4x4 stride-2 / two-source 3x3 first convolutions, networks.py:44,152,385-386) on the
 * spatial-tile kernel (tile configuration 40 of cg_conv2d_fwd_tile).  Returns the previous setting. */
int cg_conv2d_fwd_thin(int on);
/* The same layers' weight (+ bias) gradient on conv_wgrad_thin_kernel (on by default; CG_WGRAD_THIN=0 turns it off).
 * Returns the previous setting.  Workspace queries follow it. */
int cg_conv2d_wgrad_thin(int on);

/* Weight re-layout for the data-gradient pass: out[ci - ci0][tc][co] = w[co][tapmap[tc]][ci],
 * ci in [ci0, ci0+nci).  w is [Cout][T][Cin]; out is [nci][Tc][Cout]. */
int cg_weight_transpose(const float* w, float* out, int Cout, int T, int Cin, int ci0, int nci,
                        const int32_t* tapmap_host, int Tc, cg_stream_t stream);

/* y = act(x) (standalone activation; convs and norms normally fuse it) */
int cg_act_fwd(const float* x, float* y, size_t n, int act, cg_stream_t stream);
/* dz = dy * act'(y), from the activation OUTPUT y (relu / lrelu / tanh), networks.py:494-503 */
int cg_act_bwd(const float* dy, const float* y, float* dz, size_t n, int act, cg_stream_t stream);

/* ---- instance norm / AdaIN (nn.InstanceNorm2d networks.py:483; F.batch_norm on the
 *      (1, B*C, H, W) view, networks.py:640-653), fused with activation and residual add
 *      (ResBlock `out += residual`, networks.py:457-461) ------------------------------------ */

/* mean[n*C+c], rstd[n*C+c] over HW (biased variance, eps inside the sqrt). ws: N*C*splits*2 doubles */
size_t cg_instnorm_workspace(int N, int HW, int C);
int cg_instnorm_stats(const float* x, int N, int HW, int C, float eps, float* mean, float* rstd, void* ws,
                      size_t ws_bytes, cg_stream_t stream);
/* mean / rstd from the partials of cg_conv2d_fwd_stats */
int cg_instnorm_stats_from_partials(const double* part, int N, int HW, int C, int rows_per_partial, float eps,
                                    float* mean, float* rstd, cg_stream_t stream);
/* y = act((x-mean)*rstd*gamma + beta) + residual; gamma/beta NULL = plain IN, else sample n /
 * channel c reads gamma[n*gstride + c] (gstride = C for a dense [N*C] vector; = row length when
 * gamma/beta point into the MLP output [N][P], networks.py:303-312); residual or NULL */
int cg_instnorm_apply(const float* x, const float* mean, const float* rstd, const float* gamma,
                      const float* beta, int gstride, const float* residual, float* y, int N, int HW, int C,
                      int act, cg_stream_t stream);
/* backward: dz = dy*act'(z); dx = rstd*gamma*(dz - mean(dz) - xhat*mean(dz*xhat));
 * dgamma = sum dz*xhat, dbeta = sum dz (written at [n*gstride + c]; NULL for plain IN). */
int cg_instnorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                    const float* beta, int gstride, float* dx, float* dgamma, float* dbeta, int N, int HW, int C,
                    int act, void* ws, size_t ws_bytes, float* amax_state, int* amax_nslots, cg_stream_t stream);
/* The decoder's head for passes without a gradient tape, as ONE kernel: the three 1x1 Conv2dBlocks 64 -> 64 (ReLU) -> 64 (ReLU)
 * -> out_dim * nmask + nmask (tanh) of networks.py:393-395 and the mask / blend head of networks.py:398-407.  xs: the {hi, lo}
 * planes of the trunk output [npix][channels] (unscaled, as cg_instnorm_apply_split writes them); w7s / w8s / w9s: the layers'
 * split weights [cout][channels] with their power-of-two scale (w_scale x *w_scale_dev); im_in / im_out / mask: [npix][out_dim
 * | nmask] fp32.  Split-precision arithmetic as cg_conv2d_fwd_x3; a pixel's channels are read once and nothing but the image
 * and the mask is written.  `group`: npix covers the members' pixels back to back.  Built for channels = 64, out_dim = 3,
 * nmask = 3 (every shipped config); other shapes return CG_ERR_ARG and take the layer-by-layer path. */
int cg_decoder_head_fwd_x3(const void* xs, size_t x_lo_elems, const void* w7s, const void* w8s, const void* w9s,
                           size_t w_lo_elems, float w_scale, const float* w_scale_dev, const float* b7, const float* b8,
                           const float* b9, const cg_group* group, const float* im_in, float* im_out, float* mask,
                           long long npix, int channels, int out_dim, int nmask, cg_stream_t stream);

/* The same backward with dx handed over as the {hi, lo} fp16 planes of scale * dx (interleaved layout, dx_lo_elems =
 * CG_X3_LO_ELEMS) that the split-precision data- / weight-gradient kernels of the convolution IN FRONT of the norm read
 * (networks.py:515-518: conv -> norm): no fp32 round trip and no separate split pass.  The power-of-two scale is chosen
 * from an upper bound of max |dx| that the reduction pass yields before dx exists (per (sample, channel): rstd |gamma|
 * (max|dz| + |S1/HW| + max|xhat| |S2/HW|)) and left in state[1] (state: CG_SPLIT_STATE_FLOATS floats, as
 * cg_split_f16_dynamic); `dx` (optional) additionally receives the fp32 tensor.  Needs C % 32 == 0;
 * cg_instnorm_bwd_split_workspace returns 0 for shapes it does not take. */
size_t cg_instnorm_bwd_split_workspace(int N, int HW, int C);
int cg_instnorm_bwd_split(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                          const float* beta, int gstride, void* dx_split, size_t dx_lo_elems, float* state, float* dx,
                          float* dgamma, float* dbeta, int N, int HW, int C, int act, void* ws, size_t ws_bytes,
                          cg_stream_t stream);
/* amax_state / amax_nslots (optional): the apply pass leaves per-block max |dx| in amax_state[2..] and their count in
 * *amax_nslots (0 if it could not) for cg_split_f16_dynamic */

/* ---- LayerNorm (networks.py:659-686): per-sample mean / unbiased std over C*H*W, x/(std+eps),
 *      per-channel gamma/beta --------------------------------------------------------------- */
size_t cg_layernorm_workspace(int N, int HW, int C);
int cg_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                     float* std, int N, int HW, int C, float eps, void* ws, size_t ws_bytes, cg_stream_t stream);
int cg_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* std,
                     float* dx, float* dgamma, float* dbeta, int N, int HW, int C, float eps, void* ws,
                     size_t ws_bytes, cg_stream_t stream);

/* ---- resampling --------------------------------------------------------------------------- */
/* nn.AvgPool2d(3, stride=2, padding=1, count_include_pad=False), networks.py:32,129 */
int cg_avgpool3s2_fwd(const float* x, float* y, int N, int H, int W, int C, cg_stream_t stream);
int cg_avgpool3s2_bwd(const float* dy, float* dx, int N, int H, int W, int C, cg_stream_t stream);
/* nn.Upsample(scale_factor=2) nearest, networks.py:385 (forward is normally fused into the conv gather) */
int cg_upsample2x_fwd(const float* x, float* y, int N, int H, int W, int C, cg_stream_t stream);
int cg_upsample2x_bwd(const float* dy, float* dx, int N, int H, int W, int C, cg_stream_t stream);
/* nn.AdaptiveAvgPool2d(1), networks.py:347 (style encoder; forward only on the hot path) */
int cg_global_avgpool_fwd(const float* x, float* y, int N, int HW, int C, cg_stream_t stream);

/* ---- mask / blend head, Decoder_V2_atten.forward networks.py:398-407 ------------------------
 * new_x [N,HW,od*k+k] (already tanh'ed by the last conv), im_in [N,HW,od].
 * mask_j = (tanh(10*new_x[od*k+j])+1)/2;  im <- (1-mask_j)*im + mask_j*new_x[od*j : od*(j+1)] */
int cg_mask_blend_fwd(const float* new_x, const float* im_in, float* im_out, float* mask, size_t npix, int od,
                      int k, cg_stream_t stream);
/* d_im_out [npix,od], d_mask [npix,k] (may be NULL) -> d_new_x [npix, od*k+k] */
int cg_mask_blend_bwd(const float* new_x, const float* im_in, const float* d_im_out, const float* d_mask,
                      float* d_new_x, size_t npix, int od, int k, cg_stream_t stream);

/* ---- losses ------------------------------------------------------------------------------- */
/* LSGAN (networks.py:64,90,166,194) over a batch of patch maps out[nb][hw]: sample s has target
 * tgt[s] and weight wt[s]:  loss (+)= sum_s wt[s] * sum_hw (o - tgt[s])^2 / (group * hw)
 * (`group` = samples per torch.mean, i.e. the reference batch size).  tgt / wt are device arrays. */
int cg_lsgan_fwd(const float* out, const float* tgt, const float* wt, int nb, int hw, int group, float* loss,
                 int accumulate, cg_stream_t stream);
/* d_out = gscale[0] * wt[s] * 2*(o - tgt[s]) / (group*hw);  gscale is a device scalar */
int cg_lsgan_bwd(const float* out, const float* tgt, const float* wt, const float* gscale, int nb, int hw,
                 int group, float* d_out, cg_stream_t stream);

/* Focus-loss criteria (trainer_council.py:230-250) on mask [npix_total = N*H*W][k]:
 * sums[0] = sum 1/(|m-center|+eps), sums[1] = sum m, sums[2] = sum|dh| + sum|dw| (TV) */
size_t cg_focus_workspace(void);
int cg_focus_sums(const float* mask, int N, int H, int W, int k, float center, float eps, float* sums, void* ws,
                  size_t ws_bytes, cg_stream_t stream);
/* out[0] = w_zo*zero_one + w_total*mask_small + w_tv*tv, out[1..3] = the three unweighted criteria */
int cg_focus_total(const float* sums, size_t numel, float w_zo, float w_total, float w_tv, int use_abs,
                   int use_square, float* out, cg_stream_t stream);
/* d_mask = gscale[0] * ( w_zo * d(zero_one) + w_total * d(mask_small) + w_tv * d(TV) ) / numel
 * use_abs / use_square select mask_small's form (trainer_council.py:233-246); sums from cg_focus_sums */
int cg_focus_bwd(const float* mask, const float* sums, const float* gscale, int N, int H, int W, int k,
                 float center, float eps, float w_zo, float w_total, float w_tv, int use_abs, int use_square,
                 float* d_mask, cg_stream_t stream);
/* mean |a - b| (recon_criterion trainer_council.py:207-208 / council_basic_criterion :227-228) */
int cg_l1_mean_fwd(const float* a, const float* b, size_t n, float* loss, cg_stream_t stream);
int cg_l1_mean_bwd(const float* a, const float* b, const float* gscale, size_t n, float* da, cg_stream_t stream);

/* ---- optimizer: torch.optim.Adam with L2 weight decay (trainer_council.py:170-179) ----------- */
int cg_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                 float eps, float weight_decay, int step, cg_stream_t stream);

/* ---- opt-in per-kernel timing (bench.py's roofline leg; off by default, no cost when off) -------
 * While enabled, every MFMA conv launch (forward/dgrad kernel and wgrad kernel) is bracketed by
 * HIP events recorded on the launch stream.  cg_prof_collect() synchronises them and returns, per
 * kernel slot (slot = kernel family * 20 + tile shape * 2 + fast-path flag; see cg_prof_slot_name),
 * launch count, total milliseconds and total algorithmic FLOPs (2*M*N*K of each launch). */
#define CG_PROF_SLOTS 200
int cg_prof_enable(int on);
int cg_prof_collect(int64_t* counts, double* ms, double* flops);
const char* cg_prof_slot_name(int slot);
/* text table of the last cg_prof_collect(): one line per (kernel family, tile, layer shape) */
const char* cg_prof_report(void);

/* Timing probe (development aid): after a cg_conv2d_fwd_tile() launch with tile_cfg 31, 16 words per block --
 * {shader clock, 100 MHz wall clock} at kernel entry / loop start / loop end / exit, hardware id, block id. */
int cg_debug_fetch(long long* host, int nwords);

/* ---- small utilities ------------------------------------------------------------------------ */
int cg_fill(float* p, size_t n, float value, cg_stream_t stream);
int cg_add(const float* a, const float* b, float* out, size_t n, cg_stream_t stream);     /* out = a + b */
int cg_axpby(float alpha, const float* a, float beta, float* b, size_t n, cg_stream_t stream); /* b = alpha*a + beta*b */
/* gather rows of a stacked tensor: out[i] = src[idx[i]] (colleague pick after the all-gather) */
int cg_gather_rows(const float* src, const int32_t* idx_dev, float* out, int nidx, size_t row_elems,
                   cg_stream_t stream);
/* scalar helper for the on-device loss matching (trainer_council.py:518-524,576-586):
 * ring[pos % n] = value[0]; out[0] = mean(ring_a)/mean(ring_b) */
int cg_loss_match(float* ring_gan, float* ring_council, int n, int pos, const float* council_loss,
                  float* w_out, cg_stream_t stream);
int cg_ring_push(float* ring, int n, int pos, const float* value, cg_stream_t stream);

/* ---- member-batched forms of the per-member reductions (the reference's member loops, trainer_council.py:328,558,747,
 *      826,858, as ONE launch): sample / mask blocks of the members follow each other; per-member outputs -------------
 * cg_lsgan_*_g: nb = samples of ALL members, loss[nmember], gscale[nmember].
 * cg_focus_*_g: N = samples of ALL members, sums[nmember][3], out[nmember][4], gscale[nmember]; ws nmember x
 *               cg_focus_workspace().   cg_adam_step_g: the same run of `nmember` pool slices `mstride` elements apart.
 * cg_ring_push_g / cg_loss_match_g: rings [nmember][n], value / w_out [nmember].
 * A member's result never depends on which other members share the launch. */
int cg_lsgan_fwd_g(const float* out, const float* tgt, const float* wt, int nb, int hw, int group, int nmember,
                   float* loss, int accumulate, cg_stream_t stream);
int cg_lsgan_bwd_g(const float* out, const float* tgt, const float* wt, const float* gscale, int nb, int hw, int group,
                   int nmember, float* d_out, cg_stream_t stream);
int cg_focus_sums_g(const float* mask, int N, int H, int W, int k, int nmember, float center, float eps, float* sums,
                    void* ws, size_t ws_bytes, cg_stream_t stream);
int cg_focus_total_g(const float* sums, size_t numel_per_member, int nmember, float w_zo, float w_total, float w_tv,
                     int use_abs, int use_square, float* out, cg_stream_t stream);
int cg_focus_bwd_g(const float* mask, const float* sums, const float* gscale, int N, int H, int W, int k, int nmember,
                   float center, float eps, float w_zo, float w_total, float w_tv, int use_abs, int use_square,
                   float* d_mask, cg_stream_t stream);
int cg_adam_step_g(float* p, const float* g, float* m, float* v, size_t n, int nmember, long long mstride, float lr,
                   float beta1, float beta2, float eps, float weight_decay, int step, cg_stream_t stream);
/* hipGraph-friendly forms: the scalars that change from iteration to iteration come from DEVICE memory, so a captured launch
 * stays valid when replayed.  cg_adam_hyper (host-only, no launch) computes the two per-step floats exactly as
 * cg_adam_step_g does; cg_adam_step_dev reads them from hyper_dev[0..1] -- bit-identical to cg_adam_step_g.
 * cg_ring_push_dev / cg_loss_match_dev read the ring write position from pos_dev[0]. */
int cg_adam_hyper(float lr, float beta1, float beta2, int step, float* out2_host);
int cg_adam_step_dev(float* p, const float* g, float* m, float* v, size_t n, int nmember, long long mstride, float beta1,
                     float beta2, float eps, float weight_decay, const float* hyper_dev, cg_stream_t stream);
int cg_ring_push_dev(float* ring, int n, const int32_t* pos_dev, const float* value, int nmember, cg_stream_t stream);
int cg_loss_match_dev(float* ring_gan, float* ring_council, int n, const int32_t* pos_dev, const float* council_loss,
                      float* w_out, int nmember, cg_stream_t stream);
int cg_ring_push_g(float* ring, int n, int pos, const float* value, int nmember, cg_stream_t stream);
int cg_loss_match_g(float* ring_gan, float* ring_council, int n, int pos, const float* council_loss, float* w_out,
                    int nmember, cg_stream_t stream);
/* out[i] = idx[i] >= 0 ? src_a[idx[i]] : src_b[-idx[i] - 1] (rows of row_elems floats, a multiple of 4): assembles the
 * discriminator batches [fake | real] (networks.py:56-82) / [own | colleagues] (trainer_council.py:872-874) of all members */
int cg_gather_rows2(const float* src_a, const float* src_b, const int32_t* idx_dev, float* out, int nidx,
                    size_t row_elems, cg_stream_t stream);
/* generator objective per member (trainer_council.py:447-451,529,588-624): council[m] = council_w * w_match[m] * lc[m],
 * total[m] = focus[4m] + gan_w * adv[m] + council[m], gcouncil[m] = council_w * w_match[m]; NULL terms are zero */
int cg_gen_total(const float* focus, const float* adv, const float* lc, const float* w_match, float gan_w,
                 float council_w, float* total, float* council, float* gcouncil, int nmember, cg_stream_t stream);


/* ---- collectives of the sharded path ------------------------------------
 * One process per GPU.  The reference is single-process: these replace the in-process reads of the other members'
 * images (trainer_council.py:853-856, 872-874) and, when a member is replicated over several ranks, average its
 * replicas' gradients.  RCCL (xGMI) is resolved at the first call (the copy already mapped into the process, else
 * librccl.so.1 from the loader path / the ROCm installation); it is not a link-time dependency of this library.
 *   cg_comm_unique_id   rank 0 of a group draws the 128-byte id; the host distributes it (any side channel)
 *   cg_comm_create      every rank of the group, on its own HIP device (collective: returns once all ranks called)
 *   cg_allgather_images recv[r * elems_per_rank ...] = rank r's send (the local members' comparison images,
 *                       [members_per_rank * B, H, W, 3] fp32 NHWC), enqueued on `stream`
 *   cg_allreduce_sum    in place over the flat gradient buffer of a pool (optim.ParamPool), enqueued on `stream` */
typedef struct cg_comm cg_comm;
#define CG_COMM_ID_BYTES 128
int cg_comm_unique_id(unsigned char* id /* CG_COMM_ID_BYTES */);
int cg_comm_create(const unsigned char* id, int rank, int nranks, cg_comm** comm);
int cg_comm_destroy(cg_comm* comm);
int cg_comm_info(const cg_comm* comm, int* rank, int* nranks);
int cg_allgather_images(cg_comm* comm, const float* send, float* recv, size_t elems_per_rank, cg_stream_t stream);
int cg_allreduce_sum(cg_comm* comm, float* buf, size_t elems, cg_stream_t stream);


/* ---- input pipeline tail on the device ---------------------------------------
 * The last three transforms of the reference's loader (utils.py:124-129,133-134): RandomCrop((H, W)) window,
 * RandomHorizontalFlip, ToTensor + Normalize(mean, std) -- applied to a batch of decoded uint8 images:
 *   dst[n][y][x][c] = ((float)src[n][top_n + y][left_n + (flip_n ? W-1-x : x)][c] / 255 - mean) / std
 * src: [N][Hs][Ws][C] uint8 (PIL's HWC layout), dst: [N][H][W][C] fp32 = the NHWC batch the trainer consumes.
 * crop_tl: N x {top, left} int32 on the device (NULL: top-left corner), flip: N bytes on the device (NULL: none).
 * The geometric / colour transforms that torchvision applies to PIL images before these stay with the host loader. */
int cg_u8_to_f32_nhwc(const uint8_t* src, int N, int Hs, int Ws, int C, const int32_t* crop_tl, const uint8_t* flip,
                      int H, int W, float mean, float std, float* dst, cg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
