/* somi_hip.h - C ABI of libsomi_hip.so, the MI355X (gfx950) hot path of YOLO-SOMI.
 *
 * Drop-in boundary.  The reference's only native boundary is the pybind
 * module `DCNv3` (models/ops_dcnv3/src/vision.cpp:14-17, src/dcnv3.h:20-59); everything else on
 * the path is reached through Python call signatures (models/yolo.py, utils/loss.py,
 * utils/general.py).  This header declares one plain-C entry point per device operation on
 * that path; the Python host layer in yolo-somi_amd/somi_amd/ mirrors the reference's
 * Python signatures on top of it (see INTEGRATION.md for the reference-side binding).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (HBM) unless the name ends in _host; the library never
 *    allocates, frees, copies or synchronises: outputs and workspaces are caller-owned;
 *  - tensors are fp32, activations are NHWC with an explicit channel stride (`*_cs`, in floats)
 *    and channel offset (`*_coff`) so that channel slices / concatenations need no copy;
 *    vectorised paths need `*_cs % 4 == 0`, `*_coff % 4 == 0` and 16-byte aligned bases;
 *  - `stream` is a hipStream_t passed as void*; kernels are enqueued on it and the call returns;
 *  - return value: 0 on success, a negative SOMI_E* code on a rejected argument (nothing was
 *    launched), or a positive hipError_t from the launch.  somi_last_error() gives the text.
 *    Argument errors mirror the reference's AT_ASSERTM checks (dcnv3_cuda.cu:29-53).
 */
#ifndef SOMI_HIP_H
#define SOMI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SOMI_ABI_VERSION 14

#define SOMI_EINVAL   (-1) /* bad shape / stride / alignment */
#define SOMI_ENOTIMPL (-2) /* configuration outside the SOMI path */
#define SOMI_EWORKSPACE (-3) /* workspace too small */

typedef void *somi_stream_t;

int somi_abi_version(void);
const char *somi_last_error(void);
/* sizeof of a descriptor struct as the library was compiled (0: somi_conv_desc, 1: somi_loss_desc): lets a binding check its mirror */
size_t somi_sizeof_desc(int which);

/* ------------------------------------------------------------------------------------------
 * Activations used in fused epilogues.
 */
enum somi_act { SOMI_ACT_NONE = 0, SOMI_ACT_SILU = 1, SOMI_ACT_GELU = 2, SOMI_ACT_RELU = 3, SOMI_ACT_SIGMOID = 4 };

/* ------------------------------------------------------------------------------------------
 * Dense convolution as NHWC implicit GEMM on MFMA with a fused epilogue.
 * Replaces: nn.Conv2d + BatchNorm2d(eval, folded) + SiLU inside `Conv.forward`
 * (models/common.py:64-70, utils/torch_utils.py:202-222), the grouped conv of ODConv2d_3rd
 * (models/common.py:4602-4605, per-sample weights), SEAM's 1x1 (+GELU+BN, models/common.py:8463-8465),
 * Decouple's b3/c3 1x1 (+bias, models/yolo.py:1057,1063) and nn.Linear projections of DCNv3
 * (models/ops_dcnv3/modules/dcnv3.py:324,330-331,377) which are 1x1 convs in NHWC.
 *
 *   y[b,ho,wo,n] = post( act( sum_{r,q,c} x'[b, ho*s-p+r*d, wo*s-p+q*d, c] * w[wset][n][(r*kw+q)*Cin+c] + bias[wset][n] ) )
 *   post(v) = v*post_scale[n] + post_shift[n]  (if given), then + residual[b,ho,wo,n] (if given)
 *   x' = x * a_chan_scale[b][c] * a_pix_scale[b][h][w]   (each factor optional; CBAM fused on the operand load)
 *   wset = b if per_sample_w else 0.
 */
typedef struct somi_conv_desc {
    const float *x;
    const float *w;
    const float *bias;        /* [n_wsets][Cout] or NULL */
    const float *post_scale;  /* [Cout] or NULL */
    const float *post_shift;  /* [Cout] or NULL (required iff post_scale) */
    const float *residual;    /* NHWC (B,Ho,Wo,*) or NULL */
    const float *a_chan_scale;/* [B][Cin] or NULL */
    const float *a_pix_scale; /* [B][H][W] or NULL */
    float *y;
    int32_t B, H, W, Cin, x_cs, x_coff;
    int32_t Ho, Wo, Cout, y_cs, y_coff;
    int32_t kh, kw, stride, pad, dil;
    int32_t res_cs, res_coff;
    int32_t act;              /* enum somi_act */
    int32_t per_sample_w;
    int32_t res2_cs, res2_coff; /* channel stride / offset of residual2 */
    void *workspace;          /* optional scratch (16 B aligned, somi_conv2d_workspace_bytes()); with it, layers whose tile
                               * count leaves workgroup slots idle in the last round are scheduled stream-K: the K range of
                               * some tiles is cut between workgroups and the pieces are added in a fixed order (results
                               * then differ from the unsplit sum by fp32 rounding only).  NULL: one workgroup per tile. */
    uint64_t workspace_bytes;
    const float *residual2;   /* optional second tensor added after the activation (NHWC (B,Ho,Wo,*)), or NULL */
    /* Optional per-channel statistics of the stored output for a following BatchNorm in training mode (the conv epilogue
     * has every value in registers: this saves the separate read of y).  stat_sum / stat_sumsq: [somi_conv2d_stat_rows(d)][Cout]
     * partial sums of (y - pivot) and (y - pivot)^2 - one row per (row tile, wave row), each written exactly once; stat_pivot:
     * [Cout] or NULL (= 0).  Reduce them with somi_bn_stats_partials_f32.  Needs Cout % 4 == 0 and per_sample_w == 0. */
    float *stat_sum;
    float *stat_sumsq;
    const float *stat_pivot;
    /* Opt-in reduced precision of the products (train.py:263 `amp.autocast`; the default 0 is the exact fp32 path every parity claim is
     * made on): 1 = operands rounded to bf16 (v_mfma_f32_32x32x16_bf16, fp32 accumulate - autocast's arithmetic), 2 = "bf16x3": each
     * operand split into two bf16 values and hi*hi + hi*lo + lo*hi accumulated in fp32 (~1e-5 relative error).  Tensors stay fp32 in
     * HBM either way.  Honoured by the plain channel-aligned launches (Cin % 32 == 0, shared weights, no operand modulation) of
     * somi_conv2d_nhwc_f32 / _dgrad_ / _wgrad_; every other launch computes in exact fp32. */
    int32_t prec;
} somi_conv_desc;

int somi_conv2d_nhwc_f32(const somi_conv_desc *d, somi_stream_t stream);
/* Scratch size that enables the stream-K schedule for every shape (64 MiB). */
size_t somi_conv2d_workspace_bytes(void);
/* Rows of the stat_sum / stat_sumsq partial arrays somi_conv2d_nhwc_f32 would write for this descriptor. */
int somi_conv2d_stat_rows(const somi_conv_desc *d);

/* Data gradient of the convolution whose FORWARD geometry `fwd` describes (x (B,H,W,Cin) -> y (B,Ho,Wo,Cout); the pointer
 * fields of `fwd` are ignored):  dx[b,h,w,ci] = sum_{r,q,co} dy[b,(h+p-r)/s,(w+p-q)/s,co] * W[co][ci][r][q]
 * over the taps for which the divisions are exact and in range (autograd of F.conv2d, train.py:270 `backward()`).
 * Runs the same MFMA implicit-GEMM kernel with rows = forward-input pixels.  `w_dgrad` is packed [Cin][kh*kw*Cout],
 * k = (r*kw+q)*Cout + co.  `accumulate` (optional, may alias dx) is added to the result (skip connections).
 * fwd->workspace / workspace_bytes are honoured as in somi_conv2d_nhwc_f32; fwd->residual2 (with res2_cs / res2_coff), if set,
 * is a second tensor of dx's shape added to the result (a shortcut branch's gradient).
 * per_sample_w in `fwd` selects per-image weight sets [B][Cin][kh*kw*Cout]. */
int somi_conv2d_dgrad_nhwc_f32(const somi_conv_desc *fwd, const float *dy, int dy_cs, int dy_coff, const float *w_dgrad,
                               float *dx, int dx_cs, int dx_coff, const float *accumulate, int acc_cs, int acc_coff,
                               somi_stream_t stream);

/* Weight gradient of the convolution whose FORWARD geometry `fwd` describes:
 *   dw[co][(r*kw+q)*Cin + ci] = sum_{b,ho,wo} dy[b,ho,wo,co] * x[b, ho*s-p+r, wo*s-p+q, ci]     (same packing as the forward weights)
 * MFMA GEMM with the reduction over pixels, split over workgroups; the split partials live in `workspace`
 * (somi_conv2d_wgrad_workspace_bytes) and are summed in a fixed order, on top of `accumulate` if given (may alias dw).
 * per_sample_w: one gradient per image, dw is [B][Cout][K]. Cin, Cout multiples of 4. */
size_t somi_conv2d_wgrad_workspace_bytes(const somi_conv_desc *fwd);
int somi_conv2d_wgrad_nhwc_f32(const somi_conv_desc *fwd, const float *x, int x_cs, int x_coff, const float *dy, int dy_cs,
                               int dy_coff, float *dw, const float *accumulate, void *workspace, size_t workspace_bytes,
                               somi_stream_t stream);

/* Name of the kernel instantiation somi_conv2d_nhwc_f32 would launch for this descriptor (for profiling: matches the
 * kernel name rocprofv3 reports), or NULL for an invalid descriptor. */
const char *somi_conv2d_kernel_name(const somi_conv_desc *d);

/* ------------------------------------------------------------------------------------------
 * DCNv3 operator.  Replaces `dcnv3_forward` / `dcnv3_backward` of the reference extension
 * (models/ops_dcnv3/src/dcnv3.h:20-59, src/cuda/dcnv3_cuda.cu:21-174, kernels
 * src/cuda/dcnv3_im2col_cuda.cuh:216-275 fwd and :278-839 bwd).
 * input (N,H,W,G*Gc); offset (N,Ho,Wo,G*K*2) x,y interleaved, points ordered kernel_w-outer /
 * kernel_h-inner; mask (N,Ho,Wo,G*K); output (N,Ho,Wo,G*Gc).  All contiguous.
 * Backward: grad_input must be ZEROED by the caller (the reference allocates it with at::zeros,
 * dcnv3_cuda.cu:126-133); grad_offset / grad_mask are fully overwritten.
 * workspace (optional, may be NULL / 0): somi_dcnv3_backward_workspace_bytes(...) bytes, 16-byte aligned, enable the windowed
 * form of the fp32 backward (grad_input summed per tile in LDS in exact arithmetic, combined in a fixed order: no float atomics,
 * run-to-run bit-identical, while every sampling tap stays within SOMI_DCN_SLACK = 2 pixels of the kernel footprint; taps
 * beyond go through fp32 atomics and are counted in the uint32 at workspace + size - 256).  Without it - or for group widths other than 8/16/32/64 (then the size is 0) - one kernel
 * scatters with fp32 atomics like the reference.  The _f16 / _f64 entries take the two arguments for a uniform signature and
 * ignore them.
 * im2col_step is validated like the reference (batch % min(batch, im2col_step) == 0) and otherwise
 * unused: the whole batch is one launch.
 */
int somi_dcnv3_forward_f32(const float *input, const float *offset, const float *mask, float *output,
                           int N, int H, int W, int G, int Gc, int kernel_h, int kernel_w, int stride_h,
                           int stride_w, int pad_h, int pad_w, int dilation_h, int dilation_w,
                           float offset_scale, int im2col_step, somi_stream_t stream);

int somi_dcnv3_backward_f32(const float *input, const float *offset, const float *mask, const float *grad_output,
                            float *grad_input, float *grad_offset, float *grad_mask,
                            int N, int H, int W, int G, int Gc, int kernel_h, int kernel_w, int stride_h,
                            int stride_w, int pad_h, int pad_w, int dilation_h, int dilation_w,
                            float offset_scale, int im2col_step, void *workspace, size_t workspace_bytes, somi_stream_t stream);
size_t somi_dcnv3_backward_workspace_bytes(int N, int H, int W, int G, int Gc, int kernel_h, int kernel_w, int stride_h, int stride_w,
                                           int pad_h, int pad_w, int dilation_h, int dilation_w, float offset_scale);

/* The same operator over offset / mask tensors that are column ranges of wider rows: the module (modules/dcnv3.py:330-334) computes both from
 * the same activations, so ONE 1x1 GEMM with the two Linear weights stacked produces [pixel][2*G*K offsets | G*K mask logits] and the operator
 * reads (and, backward, writes grad_offset / grad_mask) in place there.  offset_stride / mask_stride = floats between consecutive pixels
 * (0 = packed, i.e. the entries above); offset_stride must be even and offset 8-byte aligned.  grad_offset / grad_mask use the same
 * strides.  Arithmetic identical to the packed entries. */
int somi_dcnv3_forward_strided_f32(const float *input, const float *offset, const float *mask, long offset_stride, long mask_stride,
                                   float *output, int N, int H, int W, int G, int Gc, int kernel_h, int kernel_w, int stride_h,
                                   int stride_w, int pad_h, int pad_w, int dilation_h, int dilation_w, float offset_scale,
                                   int im2col_step, somi_stream_t stream);
int somi_dcnv3_backward_strided_f32(const float *input, const float *offset, const float *mask, long offset_stride, long mask_stride,
                                    const float *grad_output, float *grad_input, float *grad_offset, float *grad_mask, int N, int H,
                                    int W, int G, int Gc, int kernel_h, int kernel_w, int stride_h, int stride_w, int pad_h,
                                    int pad_w, int dilation_h, int dilation_w, float offset_scale, int im2col_step,
                                    void *workspace, size_t workspace_bytes, somi_stream_t stream);

/* The other two dtypes the reference extension dispatches (AT_DISPATCH_FLOATING_TYPES_AND_HALF, dcnv3_cuda.cu:69,136), same argument
 * meaning as the _f32 pair.  _f16: tensors are IEEE half (the AMP path of train.py:263), arithmetic fp32 (the reference's opmath_t),
 * and the three gradient outputs are FP32 buffers exactly like the reference's (dcnv3_cuda.cu:126-133: the caller casts them back,
 * :168-170).  _f64: everything double (the exact-parity mode of models/ops_dcnv3/test.py:55).  grad_input must be zeroed by the
 * caller (at::zeros in the reference); grad_offset / grad_mask are fully overwritten. */
int somi_dcnv3_forward_f16(const void *input, const void *offset, const void *mask, void *output, int N, int H, int W, int G, int Gc,
                           int kernel_h, int kernel_w, int stride_h, int stride_w, int pad_h, int pad_w, int dilation_h, int dilation_w,
                           float offset_scale, int im2col_step, somi_stream_t stream);
int somi_dcnv3_backward_f16(const void *input, const void *offset, const void *mask, const void *grad_output, float *grad_input,
                            float *grad_offset, float *grad_mask, int N, int H, int W, int G, int Gc, int kernel_h, int kernel_w,
                            int stride_h, int stride_w, int pad_h, int pad_w, int dilation_h, int dilation_w, float offset_scale,
                            int im2col_step, void *workspace, size_t workspace_bytes, somi_stream_t stream);
int somi_dcnv3_forward_f64(const double *input, const double *offset, const double *mask, double *output, int N, int H, int W, int G,
                           int Gc, int kernel_h, int kernel_w, int stride_h, int stride_w, int pad_h, int pad_w, int dilation_h,
                           int dilation_w, float offset_scale, int im2col_step, somi_stream_t stream);
int somi_dcnv3_backward_f64(const double *input, const double *offset, const double *mask, const double *grad_output, double *grad_input,
                            double *grad_offset, double *grad_mask, int N, int H, int W, int G, int Gc, int kernel_h, int kernel_w,
                            int stride_h, int stride_w, int pad_h, int pad_w, int dilation_h, int dilation_w, float offset_scale,
                            int im2col_step, void *workspace, size_t workspace_bytes, somi_stream_t stream);

/* Pieces of the DCNv3 nn.Module around the operator (models/ops_dcnv3/modules/dcnv3.py:283-291,334,370-376):
 *  LayerNorm over C (biased variance, eps inside the sqrt) + activation on contiguous NHWC;
 *  softmax over the K points of each (pixel, group): x, y are (n_groups, K) contiguous;
 *  centre-feature-scale blend y = x*(1-s) + xproj*s with s = sigmoid(logit[p*logit_cs + g]). */
int somi_layernorm_act_nhwc_f32(const float *x, const float *gamma, const float *beta, float eps, int act, float *y,
                                long npix, int C, somi_stream_t stream);
/* depthwise 3x3 conv (+bias) -> LayerNorm over C -> activation in ONE pass (the DCNv3 block's dw_conv chain, modules/dcnv3.py:283-291), C == 256:
 * u = the conv output (the LayerNorm backward reads it), y = act(LN(u)); bit-identical to somi_dwconv3x3_nhwc_f32 + somi_layernorm_act_nhwc_f32. */
int somi_dwconv3x3_ln_nhwc_f32(const float *x, const float *w, const float *bias, const float *gamma, const float *beta, float eps, int act,
                               float *u, float *y, int B, int H, int W, int C, somi_stream_t stream);
int somi_group_softmax_f32(const float *x, float *y, long n_groups, int K, somi_stream_t stream);
/* group g of pixel p at x + p*x_stride + g*K (y likewise); x == y allowed */
int somi_group_softmax_strided_f32(const float *x, long x_stride, float *y, long y_stride, long npix, int G, int K, somi_stream_t stream);
int somi_dcnv3_cfs_blend_f32(const float *x, const float *xproj, const float *logit, int logit_cs, float *y, long npix,
                             int G, int Gc, somi_stream_t stream);

/* Backward of those pieces (training through the DCNv3 module).  LayerNorm->GELU: z = gelu(LN(u)), du from dz; dgamma / dbeta
 * are ACCUMULATED; workspace: somi_layernorm_act_bwd_workspace_floats(npix, C) floats.  Softmax: dx = y*(dy - sum dy*y).
 * Blend: dx = dout*(1-s), dxproj = dout*s, dlogit[p][g] = s(1-s) * sum_c dout*(xproj - x)  (dlogit row stride dlogit_cs). */
size_t somi_layernorm_act_bwd_workspace_floats(long npix, int C);
int somi_layernorm_gelu_bwd_nhwc_f32(const float *u, const float *gamma, const float *beta, float eps, const float *dz, float *du,
                                     float *dgamma_accumulate, float *dbeta_accumulate, float *workspace, long npix, int C,
                                     somi_stream_t stream);
int somi_group_softmax_bwd_f32(const float *y, const float *dy, float *dx, long n_groups, int K, somi_stream_t stream);
/* y rows of y_stride floats, dy / dx rows of d_stride floats; dx == dy allowed */
int somi_group_softmax_bwd_strided_f32(const float *y, long y_stride, const float *dy, float *dx, long d_stride, long npix, int G, int K,
                                       somi_stream_t stream);
int somi_dcnv3_cfs_blend_bwd_f32(const float *x, const float *xproj, const float *logit, int logit_cs, const float *dout, float *dx,
                                 float *dxproj, float *dlogit, int dlogit_cs, long npix, int G, int Gc, somi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Bandwidth-bound layer kernels of Model._forward_once (models/yolo.py:1269-1290).
 */

/* uint8 NCHW image batch -> fp32 NHWC with 4 channels (4th = 0), scaled by 1/255
 * (train.py:249 `imgs.float()/255`, val.py:150-152).  Also accepts fp32 NCHW input via the _f32 form. */
int somi_image_u8_to_nhwc4(const uint8_t *img, float *y, int B, int C, int H, int W, somi_stream_t stream);
int somi_image_f32_to_nhwc4(const float *img, float *y, int B, int C, int H, int W, float scale, somi_stream_t stream);

/* Depthwise 3x3, stride 1, pad 1, + bias, then act, then per-channel affine, then optional residual.
 * Replaces SEAM's `Conv2d(groups=c) -> GELU -> BatchNorm2d` stages (+Residual) (models/common.py:8454-8466, 7183-7189).
 * w is [3][3][C] (tap-major), C % 4 == 0. */
int somi_dwconv3x3_nhwc_f32(const float *x, const float *w, const float *bias, const float *post_scale,
                            const float *post_shift, const float *residual, float *y, int B, int H, int W, int C,
                            int act, somi_stream_t stream);

/* SPPF pooling: from x = slice [x_coff, x_coff+C) of a (B,H,W,cs) tensor write the three chained 5x5/s1/p2
 * max-pools (== 5x5, 9x9, 13x13 windows) to the slices at y_coff + {1,2,3}*C of the same tensor
 * (models/common.py:1856-1861).  In-place on one concat buffer: the pooled slices never alias the source slice. */
int somi_sppf_pool_nhwc_f32(float *buf, int B, int H, int W, int C, int cs, int x_coff, somi_stream_t stream);
/* The same for a training forward, level by level (three launches), leaving for every (element, level) the position r*5 + q of the first maximum
 * (row-major) in its 5x5 window of the previous slice: `codes` = 3*B*H*W*C bytes, handed to somi_sppf_pool_bwd_nhwc_f32 as its workspace with buf = NULL. */
int somi_sppf_pool_codes_nhwc_f32(float *buf, void *codes, int B, int H, int W, int C, int cs, int x_coff, somi_stream_t stream);

/* BiFPN fusion (models/common.py:3695-3704) with the preceding nn.Upsample(2,'nearest') folded in:
 * y = sum_i wn[i] * src_i, where source i is read at (h>>up[i], w>>up[i]).  wn = w / (sum swish(w) + eps) is computed
 * inside the kernel from the raw parameter w_dev (n_in floats on the device; no host copy of it is needed).
 * n_in in {2,3}; all sources C channels, contiguous NHWC.  src_host / up_host are host arrays of n_in entries. */
int somi_bifpn_nhwc_f32(const float *const *src_host, const int *up_host, const float *w_dev, float eps, int n_in, float *y,
                        int B, int H, int W, int C, somi_stream_t stream);

/* Global average + max pool over H*W of a channel slice: out_avg[b][c], out_max[b][c]
 * (ChannelAttentionModule models/common.py:355-357; also GAP for ODConv :4558 and SEAM :8483).
 * partial workspace: 2*B*nchunk*C floats with nchunk = somi_pool_nchunk(H*W). out_max may be NULL. */
int somi_pool_nchunk(int HW);
int somi_global_pool_nhwc_f32(const float *x, int x_cs, int x_coff, int B, int HW, int C, float *out_avg,
                              float *out_max, float *workspace, somi_stream_t stream);
/* The average pool of act(x), optionally followed by a per-channel affine map: out_avg[b][c] = post_scale[c] * mean_p act(x[b,p,c]) + post_shift[c]
 * (post_scale / post_shift NULL: the plain mean).  SEAM's squeeze (models/common.py:8483) reads BatchNorm(GELU(u)) only through its global average,
 * which is that expression with the BatchNorm's scale / shift: the normalised tensor is never written.  Same workspace. */
int somi_global_pool_act_nhwc_f32(const float *x, int x_cs, int x_coff, int B, int HW, int C, int act, const float *post_scale,
                                  const float *post_shift, float *out_avg, float *workspace, somi_stream_t stream);
/* z = silu(x * scale + shift) (the Conv block's BatchNorm + SiLU, as somi_chan_affine_act_nhwc_f32 with act = SILU, order = 0) AND the global
 * average / max pools of z per (image, channel) in the same pass - what a channel attention right behind the block would otherwise re-read z for
 * (models/common.py:339-358).  C / 4 must divide 256 or be a multiple of it (somi_affine_silu_pool_rows returns 0 otherwise: use the two
 * separate entries).  workspace: 2 * B * somi_affine_silu_pool_rows(B, HW, C) * C floats. */
int somi_affine_silu_pool_rows(int B, int HW, int C);
int somi_affine_silu_pool_nhwc_f32(const float *x, int x_cs, int x_coff, const float *scale, const float *shift, float *z, int z_cs, int z_coff,
                                   int B, int HW, int C, float *out_avg, float *out_max, float *workspace, somi_stream_t stream);

/* Tiny per-sample MLP heads (host picks the recipe):
 *  mode 0 (CBAM channel attention, models/common.py:355-358):
 *        out[b] = sigmoid( W2 relu(W1 avg[b] + b1) + b2 + W2 relu(W1 max[b] + b1) + b2 )
 *  mode 1 (SEAM, models/common.py:8484-8487): out[b] = exp( sigmoid( W2 relu(W1 avg[b]) ) )   (no biases)
 * W1 [mid][C], W2 [C][mid]. */
int somi_attn_mlp_f32(int mode, const float *avg, const float *mx, const float *W1, const float *b1, const float *W2,
                      const float *b2, float *out, int B, int C, int mid, somi_stream_t stream);

/* Per-pixel channel statistics of x*ca: stats[b,h,w,0] = mean_c(x*ca), stats[...,1] = max_c(x*ca)
 * (SpatialAttentionModule models/common.py:400-402 applied to `channel_attention(x)*x`, :686-688). */
int somi_chan_stats_nhwc_f32(const float *x, int x_cs, int x_coff, const float *ca, float *stats, int B, int HW, int C,
                             somi_stream_t stream);

/* k x k conv 2->1 channels + bias + sigmoid on the stats map -> sa[b,h,w] (models/common.py:396,403). w is [k][k][2]. */
int somi_spatial_attn_f32(const float *stats, const float *w, const float *bias /* device, 1 value */, float *sa, int B, int H, int W, int k,
                          somi_stream_t stream);

/* CBAM apply in one pass: sa = sigmoid(conv_kxk(stats) + bias); y = x * ca[b][c] * sa[b,h,w] (models/common.py:686-688:
 * `out = channel_attention(x2) * x2; out = spatial_attention(out) * out`).  x / y are channel slices; in place allowed. */
int somi_cbam_apply_nhwc_f32(const float *x, int x_cs, int x_coff, const float *ca, const float *stats, const float *w,
                             const float *bias /* device, 1 value */, float *y, int y_cs, int y_coff, int B, int H, int W, int C, int k,
                             somi_stream_t stream);

/* y = x * s[b][c] (SEAM output x*exp(fc), models/common.py:8489-8490; also materialised CBAM scaling). In place allowed. */
int somi_scale_channels_nhwc_f32(const float *x, const float *s, const float *pix, float *y, int B, int HW, int C,
                                 somi_stream_t stream);

/* ODConv attention + per-sample weight synthesis (models/common.py:4557-4590):
 *  z = relu(bn(fc(gap)))  [bn folded into fc_w/fc_b by the host; skipped when B==1 as the reference does]
 *  a_f = sigmoid(Wf z + bf) [Cout], a_s = sigmoid(Ws z + bs) [kk], a_c = sigmoid(Wc z + bc) [Cin], a_w = softmax(Ww z + bw) [K]
 *  wout[b][n][(t)*Cin_pad + c] = a_f[n] a_s[t] a_c[c] sum_K a_w[K] Wk[K][n][t][c]   (zero in padded c)
 *  bout[b][n] = sum_K a_w[K] bias[K][n]
 * then conv BN (eval) is folded: wout *= bn_scale[n]; bout = bout*bn_scale[n] + bn_shift[n].
 * Wk is [K][Cout][kk][Cin_pad]. attn workspace: B*(hid + Cout + kk + Cin + K) floats. */
int somi_odconv_weights_f32(const float *gap, const float *fc_w, const float *fc_b, const float *Wf, const float *bf,
                            const float *Ws, const float *bs, const float *Wc, const float *bc, const float *Ww,
                            const float *bw, const float *Wk, const float *biask, const float *bn_scale,
                            const float *bn_shift, float *wout, float *bout, float *workspace, int B, int Cin,
                            int Cin_pad, int Cout, int kk, int K, int hid, somi_stream_t stream);

/* Detection decode (DecoupledDetect.forward eval branch, models/yolo.py:943-961 + Decouple interleave :1073):
 * box (B,ny,nx,box_cs>=na*5) and cls (B,ny,nx,cls_cs>=na*nc) head outputs ->
 *   raw (B,na,ny,nx,no)            [x[i] of the reference, always written if non-NULL]
 *   z   (B,total,no) rows [row_off, row_off+na*ny*nx): sigmoid, xy=(s*2-0.5+grid)*stride, wh=(s*2)^2*anchor_px. */
int somi_detect_decode_f32(const float *box, int box_cs, const float *cls, int cls_cs, const float *anchors_px_host,
                           float stride, float *raw, float *z, int B, int ny, int nx, int na, int nc, int total,
                           int row_off, somi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Training-mode normalisation / activation (nn.BatchNorm2d with batch statistics: models/common.py:64-66 under model.train(),
 * eps 1e-3 / momentum 0.03 from utils/torch_utils.py:165-174; SEAM's act-then-norm stages models/common.py:8454-8466).
 * Tensors are channel slices (cs, coff) of NHWC buffers with npix = B*H*W rows.  order 0: z = act(x*scale+shift) (norm then
 * act); order 1: z = act(x)*scale+shift (act then norm).  Workspaces: bn_stats 2*nchunk*C floats,
 * bn_act_backward 2*nchunk*C + 3*C(+pad to 4) floats, chan_sum 2*nchunk*C floats, nchunk = somi_red_nchunk(npix).
 */
int somi_red_nchunk(long npix);
/* batch mean / biased variance of x per channel -> mean, rstd = 1/sqrt(var+eps), scale = gamma*rstd, shift = beta - mean*scale;
 * running_mean / running_var (optional) updated with `momentum` and the unbiased variance like nn.BatchNorm2d. */
int somi_bn_stats_nhwc_f32(const float *x, int x_cs, int x_coff, long npix, int C, float eps, float momentum,
                           const float *gamma, const float *beta, float *mean, float *rstd, float *scale, float *shift,
                           float *running_mean, float *running_var, float *workspace, somi_stream_t stream);
/* The same statistics of act(x) (enum somi_act) without act(x) ever being stored: SEAM's act-then-norm stages (models/common.py:8455-8457)
 * then are two passes - this one, and somi_chan_affine_act_nhwc_f32 with order 1 - instead of act, statistics, affine. */
int somi_bn_stats_act_nhwc_f32(const float *x, int x_cs, int x_coff, int act, long npix, int C, float eps, float momentum,
                               const float *gamma, const float *beta, float *mean, float *rstd, float *scale, float *shift,
                               float *running_mean, float *running_var, float *workspace, somi_stream_t stream);
/* The same from the partial sums a convolution left behind (somi_conv_desc.stat_sum / stat_sumsq, `rows` rows, taken around
 * running_mean as the pivot when running_mean is given - pass the same pointer as stat_pivot).  workspace: 2*1024*C floats. */
int somi_bn_stats_partials_f32(const float *part_sum, const float *part_sumsq, int rows, long npix, int C, float eps, float momentum,
                               const float *gamma, const float *beta, float *mean, float *rstd, float *scale, float *shift,
                               float *running_mean, float *running_var, float *workspace, somi_stream_t stream);
/* SyncBatchNorm (reference train.py:165-167, torch.nn.SyncBatchNorm.convert_sync_batchnorm under --sync-bn): the statistics of a
 * BatchNorm layer over the batches of ALL ranks.  The library has no communicator: the caller exchanges one small record per layer.
 *   forward   somi_bn_local_sums_f64 -> record [2*C + 1] doubles {mean_r, M2_r = sum (x - mean_r)^2, pixel count} of this rank, from x
 *             or (part_sum != NULL) from a convolution's partial rows (taken around `pivot` there); the record itself is pivot-free,
 *             so ranks whose running means have drifted apart still combine correctly.
 *             all-gather the records ([nranks][2*C + 1]); somi_bn_stats_from_sums_f64 combines them in rank order (parallel-variance
 *             formula in double; identical results on every rank) -> mean / rstd / scale / shift and the running statistics
 *             (unbiased variance over the global count).
 *   backward  somi_bn_act_backward_sums_f64 -> record {sum d, sum d*(v - mean), count}; all-gather; ..._apply_sync_f32 builds the
 *             input gradient from the global sums and ACCUMULATES dgamma / dbeta from the LOCAL ones (the gradient exchange sums them).
 * workspace: max(2*1024*C, 2*somi_red_nchunk(npix)*C) floats (sums), 3*C floats (apply). */
int somi_bn_local_sums_f64(const float *x, int x_cs, int x_coff, long npix, int C, const float *pivot /* or NULL */, const float *part_sum,
                           const float *part_sumsq, int rows, double *sums, float *workspace, somi_stream_t stream);
int somi_bn_stats_from_sums_f64(const double *all_sums, int nranks, int C, float eps, float momentum, const float *gamma, const float *beta,
                                float *mean, float *rstd, float *scale, float *shift, float *running_mean, float *running_var,
                                somi_stream_t stream);
int somi_bn_act_backward_sums_f64(const float *dz, int dz_cs, int dz_coff, const float *x, int x_cs, int x_coff, const float *mean,
                                  const float *scale, const float *shift, int act, int order, long npix, int C, double *sums, float *workspace,
                                  somi_stream_t stream);
int somi_bn_act_backward_apply_sync_f32(const float *dz, int dz_cs, int dz_coff, const float *x, int x_cs, int x_coff, const float *mean,
                                        const float *rstd, const float *scale, const float *shift, int act, int order,
                                        const double *local_sums, const double *all_sums, int nranks, float *dx, int dx_cs, int dx_coff,
                                        float *dgamma, float *dbeta, long npix, int C, float *workspace, somi_stream_t stream);
int somi_chan_affine_act_nhwc_f32(const float *x, int x_cs, int x_coff, const float *scale, const float *shift, int act,
                                  int order, float *z, int z_cs, int z_coff, long npix, int C, const float *residual /* or NULL: added last */,
                                  int res_cs, int res_coff, somi_stream_t stream);
/* gradient of z = [order 0] act(norm(x)) / [order 1] norm(act(x)) w.r.t. x (written to dx, may alias dz), and dgamma / dbeta
 * ACCUMULATED into the given arrays (may be NULL).  batch_stats = 1: statistics were computed from this batch (train);
 * 0: frozen statistics (the norm is a per-channel affine map). */
int somi_bn_act_backward_nhwc_f32(const float *dz, int dz_cs, int dz_coff, const float *x, int x_cs, int x_coff, const float *mean,
                                  const float *rstd, const float *scale, const float *shift, int act, int order,
                                  int batch_stats, float *dx, int dx_cs, int dx_coff, float *dgamma, float *dbeta, long npix,
                                  int C, float *workspace, somi_stream_t stream);
/* The same backward (batch statistics) for the conv in front of a CBAM attention pair (models/common.py:339-358, 671-691), with the gradient of
 * the channel attention's global pools folded in instead of added by a pass of its own (somi_pool_bwd_add_nhwc_f32):
 *   dz_eff[b,p,c] = dz[b,p,c] + davg[b,c] / HW + [p == amaxp[b,c]] * dmax[b,c];   dz is only read.
 * davg, dmax (B,C) float, amaxp (B,C) int32 - the first pixel of each channel's spatial maximum.  dmax and amaxp may both be NULL (no max-pool:
 * SEAM's squeeze, models/common.py:8483-8490), and dz may be NULL when the pooled part is the WHOLE incoming gradient (the tensor is read by a
 * global average pool only: no zero tensor is materialised and read).  workspace: 2 * somi_bn_pooled_rows(B, HW) * C + 3 * round_up(C, 4) floats. */
int somi_bn_pooled_rows(int B, int HW);
int somi_bn_act_backward_pooled_nhwc_f32(const float *dz, int dz_cs, int dz_coff, const float *x, int x_cs, int x_coff, const float *mean,
                                         const float *rstd, const float *scale, const float *shift, int act, int order, const float *davg,
                                         const float *dmax, const int32_t *amaxp, float *dx, int dx_cs, int dx_coff, float *dgamma,
                                         float *dbeta, int B, int HW, int C, float *workspace, somi_stream_t stream);
/* out = a + b on channel slices (residual connections, gradient accumulation); out may alias a or b */
int somi_add_nhwc_f32(const float *a, int a_cs, int a_coff, const float *b, int b_cs, int b_coff, float *out, int o_cs,
                      int o_coff, long npix, int C, somi_stream_t stream);
/* out[c] += sum over pixels of x[p,c] (bias gradients) */
int somi_chan_sum_nhwc_f32(const float *x, int x_cs, int x_coff, long npix, int C, float *out_accumulate, float *workspace,
                           somi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Backward of the attention pieces (autograd of models/common.py:339-405, 671-691 CBAM; :8483-8490 SEAM squeeze).
 * See yolo-somi_amd/csrc/train_blocks.hip for the derivation.  Per-image reductions use nchunk = somi_img_nchunk(H*W).
 */
int somi_img_nchunk(int HW);
/* A: dlogit[p] = (sum_c dt2*t*ca) * sa*(1-sa), amaxc[p] = argmax_c(t*ca) */
int somi_cbam_bwd_pixel_f32(const float *dt2, int d_cs, int d_coff, const float *t, int t_cs, int t_coff, const float *ca,
                            const float *sa, float *dlogit, int32_t *amaxc, int B, int HW, int C, somi_stream_t stream);
/* A + D: as somi_cbam_bwd_pixel_f32, and amaxp[b,c] = first pixel index p with t[b,p,c] == t_max[b,c], where t_max (B,C) is the spatial maximum the
 * forward pooled from the same tensor (somi_global_pool_nhwc_f32's out_max) - the arg-max of the channel attention's max-pool without the pass
 * over t that somi_pool_argmax_nhwc_f32 takes (integer min over the matching pixels: order-independent). */
int somi_cbam_bwd_pixel_argmax_f32(const float *dt2, int d_cs, int d_coff, const float *t, int t_cs, int t_coff, const float *ca,
                                   const float *sa, const float *t_max, float *dlogit, int32_t *amaxc, int32_t *amaxp, int B, int HW, int C,
                                   somi_stream_t stream);
/* B: gradient of sigmoid's argument through the k x k conv: dstats (B,H,W,2); dw and dbias ACCUMULATED; dw_chw = 0: dw laid out [k][k][2] like
 * the forward's packed weight, 1: (2,k,k) - nn.Conv2d(2, 1, k).weight's own layout, so that the parameter's .grad can be the target.
 * workspace: ceil(B*H*W/512) * (2*k*k+1) floats */
int somi_spatial_attn_bwd_f32(const float *dlogit, const float *stats, const float *w, float *dstats, float *dw_accumulate,
                              float *dbias_accumulate, float *workspace, int B, int H, int W, int k, int dw_chw, somi_stream_t stream);
/* Step C INSIDE the BatchNorm + SiLU backward of the bottleneck's first conv (models/common.py:671-691: cv1 = Conv -> BN -> SiLU, then t*ca*sa):
 * d = gradient w.r.t. t*ca*sa, y = cv1's convolution output, t = silu(scale*y + shift) is rebuilt in registers, dt never exists in memory.
 *   reduce: dca[b,c] = sum_p dt1*t (as step C) and, per (image chunk, channel), the separable batch sums of dt*silu'(u) - kept in `workspace`;
 *   (the caller then runs somi_attn_mlp_bwd_f32 on dca -> davg, dmax)
 *   apply:  the BatchNorm backward (batch statistics) of dz = dt + davg/HW + [p == amaxp]*dmax: dx, dgamma / dbeta ACCUMULATED.
 * Same tensors and `workspace` (somi_cbam_bn_bwd_workspace_floats(B,HW,C) floats) for both calls.  amaxp (B,C): first pixel of each channel's spatial
 * maximum (somi_cbam_bwd_pixel_argmax_f32); amaxc, dstats: steps A / B. */
size_t somi_cbam_bn_bwd_workspace_floats(int B, int HW, int C);
int somi_cbam_bn_bwd_reduce_f32(const float *d, int d_cs, int d_coff, const float *y, int y_cs, int y_coff, const float *scale, const float *shift,
                                const float *mean, const float *ca, const float *sa, const float *dstats, const int32_t *amaxc,
                                const int32_t *amaxp, float *dca, float *workspace, int B, int HW, int C, somi_stream_t stream);
int somi_cbam_bn_bwd_apply_f32(const float *d, int d_cs, int d_coff, const float *y, int y_cs, int y_coff, const float *scale, const float *shift,
                               const float *mean, const float *rstd, const float *ca, const float *sa, const float *dstats, const int32_t *amaxc,
                               const int32_t *amaxp, const float *davg, const float *dmax, float *dx, int dx_cs, int dx_coff, float *dgamma,
                               float *dbeta, float *workspace, int B, int HW, int C, somi_stream_t stream);
/* C: dt1 = dt2*sa + dstats0/C + [c==amaxc]*dstats1; dca[b,c] = sum_p dt1*t; dt2 <- dt1*ca (in place).
 * workspace: B*nchunk*C floats */
int somi_cbam_bwd_chan_f32(float *dt2_inout, int d_cs, int d_coff, const float *t, int t_cs, int t_coff, const float *ca,
                           const float *sa, const float *dstats, const int32_t *amaxc, float *dca, float *workspace, int B, int HW,
                           int C, somi_stream_t stream);
/* D: amaxp[b,c] = first pixel index of max_p x[b,p,c]. workspace: 2*B*nchunk*C 4-byte words */
int somi_pool_argmax_nhwc_f32(const float *x, int x_cs, int x_coff, int B, int HW, int C, int32_t *amaxp, void *workspace,
                              somi_stream_t stream);
/* E: backward of somi_attn_mlp_f32 (same modes); dW1/db1/dW2/db2 ACCUMULATED (db* may be NULL), davg / dmax overwritten.
 *    The sum over samples runs in ascending order (no atomics).  workspace: somi_attn_mlp_bwd_workspace_floats(B, C, mid) floats. */
size_t somi_attn_mlp_bwd_workspace_floats(int B, int C, int mid);
int somi_attn_mlp_bwd_f32(int mode, const float *dout, const float *out, const float *avg, const float *mx, const float *W1,
                          const float *b1, const float *W2, float *dW1, float *db1, float *dW2, float *db2, float *davg,
                          float *dmax, float *workspace, int B, int C, int mid, somi_stream_t stream);
/* F: dt[p,c] += davg[b,c]/HW + [p==amaxp[b,c]]*dmax[b,c]  (dmax / amaxp may be NULL: average pool only) */
int somi_pool_bwd_add_nhwc_f32(float *dt_inout, int d_cs, int d_coff, const float *davg, const float *dmax, const int32_t *amaxp,
                               int B, int HW, int C, somi_stream_t stream);

/* Backward of the remaining layer kernels (train_blocks.hip).
 * detect: d raw (B,na,ny,nx,no) -> d box (B,ny,nx,box_cs), d cls (B,ny,nx,cls_cs) (inverse of the interleave; pads zeroed).
 * sppf:   dbuf slices 1..3 (the gradients of the three chained 5x5 pools y1 = m(x), y2 = m(y1), y3 = m(y2)) are routed back through the chain - each
 *         pool's gradient to the arg-max of its 5x5 window of the previous slice, first maximum in row-major order - and ADDED into slice 0 of dbuf
 *         (gather form, fixed summation order).  Slices 1 and 2 of dbuf hold the chain's intermediate gradients afterwards.  workspace: 3*B*H*W*C bytes;
 *         with buf = NULL it already holds the codes of somi_sppf_pool_codes_nhwc_f32 and no search runs.  SPP's parallel 5 / 9 / 13
 *         pools (models/common.py:1806-1826) give the same values and the same routing except at exact ties inside a window.
 * bifpn:  dsrc_i = wn_i*dout (2x2 sum for an upsampled source, dsrc_i low-res); dw ACCUMULATED incl. the normalisation's chain
 *         rule.  workspace: 3*2048 floats.
 * dwconv: dx (+dx_accumulate), dw [3][3][C] and dbias ACCUMULATED.  workspace: ceil(B*H*W/512)*10*C floats.
 * scale:  y = x*s[b][c]: dx = dout*s, ds[b,c] = sum_p dout*x.  workspace: B*nchunk*C floats. */
int somi_detect_raw_bwd_f32(const float *draw, float *dbox, int box_cs, float *dcls, int cls_cs, int B, int ny, int nx, int na,
                            int nc, somi_stream_t stream);
int somi_sppf_pool_bwd_nhwc_f32(const float *buf, float *dbuf, void *workspace, int B, int H, int W, int C, int cs, int x_coff,
                                somi_stream_t stream);   /* workspace: 3*B*H*W*C bytes (arg-max codes) */
int somi_bifpn_bwd_nhwc_f32(const float *const *src_host, float *const *dsrc_host, const int *up_host, const float *w_dev,
                            float eps, int n_in, const float *dout, float *dw_accumulate, float *workspace, int B, int H,
                            int W, int C, somi_stream_t stream);
size_t somi_dwconv3x3_bwd_workspace_floats(int B, int W, int C);   /* floats of `workspace` below; C/4 must divide 256 */
int somi_dwconv3x3_bwd_nhwc_f32(const float *dy, const float *x, const float *w, float *dx, const float *dx_accumulate,
                                float *dw_accumulate, float *dbias_accumulate, float *workspace, int B, int H, int W, int C,
                                somi_stream_t stream);
int somi_scale_channels_bwd_nhwc_f32(const float *dout, const float *x, const float *s, float *dx, float *ds, float *workspace, int B,
                                     int HW, int C, somi_stream_t stream);

/* Training path of ODConv (models/common.py:4495-4624), see train_odconv.hip.
 * linear:     y[b][y_off+o] = act(x[b] . W[o] + bias[o]); act = enum somi_act or 5 = softmax over the nout outputs (<= 64).
 * linear_bwd: (dy, y) are slices (ld, off) of one buffer pair; computes d(pre-activation) through `act`, then dW, db ACCUMULATED,
 *             dx (B,nin) written or accumulated.  workspace: B*nout floats.
 * synth:      per-sample weights / biases from an attention buffer laid out [a_f(Cout) | a_s(kk) | a_c(Cin) | a_w(K)] per sample.
 * synth_bwd:  from dW_b (B,Cout,kk*Cin_pad) and dbias_b (B,Cout): dWk, dbiask ACCUMULATED (float atomics), dattn (same layout) written. */
int somi_linear_f32(const float *x, int ldx, const float *W, const float *bias, int act, float *y, int ldy, int y_off, int B, int nin,
                    int nout, somi_stream_t stream);
int somi_linear_bwd_f32(const float *x, int ldx, const float *W, const float *dy, const float *y, int ld, int off, int act, float *dW,
                        float *db, float *dx, int ldx_out, int dx_accumulate, float *workspace, int B, int nin, int nout,
                        somi_stream_t stream);
int somi_odconv_synth_f32(const float *attn, const float *Wk, const float *biask, float *wout, float *bout, int B, int Cin, int Cin_pad,
                          int Cout, int kk, int K, somi_stream_t stream);
/* dWk / dbiask ACCUMULATED, dattn overwritten; fixed summation orders (no atomics).
 * workspace: somi_odconv_synth_bwd_workspace_floats(B, Cin, Cout, kk, K) floats. */
size_t somi_odconv_synth_bwd_workspace_floats(int B, int Cin, int Cout, int kk, int K);
int somi_odconv_synth_bwd_f32(const float *dWb, const float *attn, const float *Wk, const float *biask, const float *dbias_b, float *dWk,
                              float *dbiask, float *dattn, float *workspace, int B, int Cin, int Cin_pad, int Cout, int kk, int K,
                              somi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Optimizer: torch.optim.Adam step (train.py:134-140,271) fused with the ModelEMA update
 * (utils/torch_utils.py:335-345) over one flat, 16 B-aligned parameter segment:
 *   g' = grad + weight_decay*param; m = b1 m + (1-b1) g'; v = b2 v + (1-b2) g'^2;
 *   param -= lr/(1-b1^step) * m / (sqrt(v)/sqrt(1-b2^step) + eps);   ema = d*ema + (1-d)*param  (ema may be NULL)
 * axpby: y = a*y + b*x (EMA of the BN running statistics). */
int somi_adam_ema_step_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float *ema, long n, float lr, float beta1,
                           float beta2, float eps, float weight_decay, int step, float ema_decay, somi_stream_t stream);
/* The other optimizer branch of train.py:136-138: torch.optim.SGD(momentum, nesterov=True) + the same EMA update in one pass
 * (g' = g + weight_decay*p; buf = step == 1 ? g' : momentum*buf + g'; p -= lr*(g' + momentum*buf)).  momentum_buf plays exp_avg's role. */
int somi_sgd_ema_step_f32(float *param, const float *grad, float *momentum_buf, float *ema, long n, float lr, float momentum,
                          float weight_decay, int step, float ema_decay, somi_stream_t stream);
int somi_axpby_f32(float *y, const float *x, long n, float a, float b, somi_stream_t stream);
/* Forward-packed conv weights [Cout][taps][Cin] -> the packing somi_conv2d_dgrad_nhwc_f32 reads, [Cin][taps][Cout].  Run once
 * per optimizer step on the master weights (which the training path keeps in the forward packing). */
int somi_pack_dgrad_weights_f32(const float *w_packed, float *w_dgrad, int Cout, int taps, int Cin, somi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Stock YOLOv5 module set (north_star "CSP/Darknet backbone, PANet/FPN neck, anchor-based detection head"; BASELINE configs[0]).
 * Bottleneck / C3 / SPP (models/common.py:1494-1509,1541-1565,1806-1826) are compositions of the convolution entry points above
 * (C3's torch.cat never materialises: cv2 and the last bottleneck write the two halves of cv3's input; SPP's parallel 5/9/13
 * pools equal SPPF's chained 5x5 pools exactly and share somi_sppf_pool_nhwc_f32).  What needs kernels of its own:
 *
 * somi_resample_slice_nhwc_f32 - `Concat` (models/common.py:2085-2097) of inputs that cannot be produced in place, with
 *   `nn.Upsample(None, 2, 'nearest')` folded into the copy.
 *     reduce = 0:  dst[b, h, w, dst_coff + c] = src[b, h >> up, w >> up, src_coff + c]      src (B,Hs,Ws,src_cs) -> dst (B,Hs<<up,Ws<<up,dst_cs)
 *     reduce = 1:  dst[b, h, w, dst_coff + c] (+)= sum_{i,j < 2^up} src[b, (h<<up)+i, (w<<up)+j, src_coff + c]   (the adjoint; fixed order)
 *   C, the strides and the offsets are multiples of 4, bases 16-byte aligned.
 * somi_space_to_depth_nhwc_f32 - `Focus` (models/common.py:1996): y[b,h,w,q*C+c] = x[b, 2h+(q&1), 2w+(q>>1), c], q = 0..3;
 *   inverse = 1: x is the gradient in y's layout (B,Ho,Wo,x_cs), y receives the gradient in the image layout (B,2Ho,2Wo,y_cs).
 * somi_detect_plain_decode_f32 - `Detect.forward` (models/yolo.py:66-98): t (B,ny,nx,t_cs >= na*no) = the level's 1x1 conv output ->
 *   raw (B,na,ny,nx,no) and, if z != NULL, rows [row_off, row_off + na*ny*nx) of z (B,total,no):
 *   xy = (2*sigmoid - 0.5 + cell)*stride (that operation order), wh = (2*sigmoid)^2 * anchors_px, rest sigmoid.
 * somi_detect_plain_raw_bwd_f32 - the adjoint of the view/permute: d raw -> d t (pad channels zeroed). */
int somi_resample_slice_nhwc_f32(const float *src, int src_cs, int src_coff, float *dst, int dst_cs, int dst_coff, int B, int Hs,
                                 int Ws, int C, int up, int reduce, int accumulate, somi_stream_t stream);
int somi_space_to_depth_nhwc_f32(const float *x, int x_cs, int x_coff, float *y, int y_cs, int y_coff, int B, int Ho, int Wo, int C,
                                 int inverse, somi_stream_t stream);
int somi_detect_plain_decode_f32(const float *t, int t_cs, const float *anchors_px_host, float stride, float *raw, float *z, int B,
                                 int ny, int nx, int na, int nc, int total, int row_off, somi_stream_t stream);
int somi_detect_plain_raw_bwd_f32(const float *draw, float *dt, int t_cs, int B, int ny, int nx, int na, int nc, somi_stream_t stream);

/* Test-time augmentation, `Model._forward_augment` (models/yolo.py:1253-1267).
 * somi_tta_resample_nhwc4_f32: `scale_img(x.flip(3) if flip_lr else x, ratio, gs)` (utils/torch_utils.py:270-282) on the ingested image
 *   x (B,H,W,4) -> y (B,Hp,Wp,4): bilinear resize to (Hs,Ws) = (int(H*ratio), int(W*ratio)) with torch's align_corners=False arithmetic,
 *   the rest up to (Hp,Wp) = the next stride multiples filled with `pad` (0.447); channels >= `channels` stay zero.
 * somi_tta_descale_f32: `_descale_pred` (:1292-1308) in place on z (rows, no): xywh /= scale, x = img_w - x after a left-right flip. */
int somi_tta_resample_nhwc4_f32(const float *x, float *y, int B, int H, int W, int Hs, int Ws, int Hp, int Wp, int flip_lr, float pad,
                                int channels, somi_stream_t stream);
int somi_tta_descale_f32(float *z, long rows, int no, float scale, int flip_lr, float img_w, somi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Post-processing: batched NMS (utils/general.py:629-711 incl. the torchvision.ops.nms core at :694).
 * pred (B,n,5+nc) decoded.  Output: det (B,max_det,6) [x1,y1,x2,y2,conf,cls], count (B) int32.
 * Selection is bit-exact with the oracle: candidates in prediction order (row-major over (box, class) for
 * multi_label), stable sort by descending score, greedy suppression with IoU > iou_thres, class offset 4096.
 * classes_mask: HOST pointer to ceil(nc/64) 64-bit words, bit (c % 64) of word c/64 set = keep class c (the `classes` argument,
 * general.py:676-677); NULL = no filter.  nc <= 1024.
 * workspace bytes: somi_nms_workspace_bytes(B, n, nc, multi_label).
 */
size_t somi_nms_workspace_bytes(int B, int n, int nc, int multi_label);
int somi_nms_f32(const float *pred, int B, int n, int nc, float conf_thres, float iou_thres, int multi_label,
                 int agnostic, const uint64_t *classes_mask, int max_det, float *det, int32_t *count, void *workspace,
                 size_t workspace_bytes, somi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Loss: ComputeLoss.__call__ + build_targets (utils/loss.py:142-262) with CIoU (utils/metrics.py:476-518).
 * p[l] (B,na,ny_l,nx_l,no) for l < nl<=4; targets (nt,6) [img,cls,x,y,w,h]; anchors (nl,na,2) in grid units.
 * out[0..3] = total*bs, lbox, lobj, lcls (already multiplied by the hyp gains, like loss_items).
 * grad[l] (optional, may be NULL): d(out[0])/d p[l], fully overwritten.  Cells hit by several targets sum their entries in
 * ascending (level, anchor, target, offset) order: value and gradient are run-to-run bit-identical.
 * workspace bytes: somi_loss_workspace_bytes(B, na, nl, ny[], nx[], nt).
 */
typedef struct somi_loss_desc {
    const float *p[4];
    float *grad[4];
    int32_t ny[4], nx[4];
    int32_t nl, na, nc, B, nt;
    const float *targets;
    const float *anchors;
    float balance[4];
    float box_gain, obj_gain, cls_gain, cls_pw, obj_pw, anchor_t, cp, cn, gr;
    float fl_gamma;     /* hyp['fl_gamma']: > 0 wraps both BCE terms in FocalLoss(gamma, alpha 0.25) (utils/loss.py:125-127,35-60) */
    int32_t slide;      /* hyp['slide_ratio'] > 0: SlideLoss around them (:129-131,378-402), weighted by the level's mean IoU */
    float nwd_ratio;    /* 0, or iou_ratio = 0.5 when hyp['nwdloss'] > 0: box term (1-r)(1-CIoU) + r(1-NWD) (:148,162-169) */
    float nwd_constant; /* 12.8 = wasserstein_loss (utils/metrics.py:341); 2.5 = wasserstein under hyp['shapeloss'] > 0 (:373, its shape
                         * weights are 1 at the scale1 = 0 the loss calls it with) */
} somi_loss_desc;

size_t somi_loss_workspace_bytes(const somi_loss_desc *d);
/* out8: 8 device floats = {loss * B, lbox, lobj, lcls, obji[0..3]}; obji[l] = the level's mean objectness BCE before `balance`, the value
 * ComputeLoss(autobalance=True) updates its balance with after the call (utils/loss.py:197-201; host state, the kernel only reports). */
int somi_yolo_loss_f32(const somi_loss_desc *d, float *out8, void *workspace, size_t workspace_bytes,
                       somi_stream_t stream);

/* Repulsion loss, RepGT + RepBox (utils/RepulsionLoss.py:47-95; imported by utils/loss.py:8 but never called by ComputeLoss,
 * so it is an optional term and off by default).  pbox, gtbox: (B,A,4) xyxy; fg_mask: (B,A) bytes (non-zero = foreground).
 * Value only - the reference detaches both box sets.  out2 = {rep_gt, rep_box}, each the mean over the images that have
 * foreground anchors (0/0 = NaN when none has, like the reference). */
size_t somi_repulsion_workspace_bytes(int B, int A);
int somi_repulsion_loss_f32(const float *pbox, const float *gtbox, const uint8_t *fg_mask, int B, int A, float sigma_repgt,
                            float sigma_repbox, float pnms, float gtnms, float *out2, void *workspace, size_t workspace_bytes,
                            somi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Weighted boxes fusion (wbf.py:68 -> ensemble_boxes.weighted_boxes_fusion, conf_type 'avg').
 * boxes (n,4) xyxy in [0,1], scores (n), labels (n) int32, model (n) int32, already concatenated over models
 * in model order; out_* sized n; out_count 1 int32.  One launch per image; sequential per label by design.
 */
size_t somi_wbf_workspace_bytes(int n);
int somi_wbf_f32(const float *boxes, const float *scores, const int32_t *labels, const int32_t *model, int n,
                 int n_models, const float *weights_host, float iou_thr, float skip_box_thr, float *out_boxes,
                 float *out_scores, int32_t *out_labels, int32_t *out_count, void *workspace, size_t workspace_bytes,
                 somi_stream_t stream);
/* Batched form (BASELINE configs[4]: val.py's NMS followed by wbf.py over the detections of several models): one workgroup per image.
 * det[t] (B,max_det,6) [x1,y1,x2,y2,conf,cls] in pixels and count[t] (B) are somi_nms_f32's outputs for model t (host arrays of
 * n_models device pointers).  Image b's members are model 0's rows, then model 1's, ... (the order wbf.py:44-59 appends them in),
 * boxes divided by (img_w, img_h) and clipped to [0,1].  Outputs are strided per image by N = n_models * max_det:
 * out_boxes (B,N,4), out_scores (B,N), out_labels (B,N), out_count (B).  Same arithmetic as somi_wbf_f32, image by image.
 */
size_t somi_wbf_batch_workspace_bytes(int B, int max_det, int n_models);
int somi_wbf_batch_f32(const float *const *det, const int32_t *const *count, int B, int max_det, int n_models,
                       const float *weights_host, float img_w, float img_h, float iou_thr, float skip_box_thr, float *out_boxes,
                       float *out_scores, int32_t *out_labels, int32_t *out_count, void *workspace, size_t workspace_bytes,
                       somi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Validation metrics after NMS.
 * somi_val_match_f32 = val.py:50-71 `process_batch` for a batch of images: det (sum N_b, 6) x1,y1,x2,y2,conf,cls and
 * labels (sum M_b, 5) cls,x1,y1,x2,y2 are concatenated per image, det_off / lab_off (B+1 ints, device) delimit the images;
 * iouv: T <= 16 ascending IoU levels (device); correct: (sum N_b, T) bytes.  max_det / max_labels: largest per-image counts
 * (host values; they size the LDS tables).  Exact IoU ties are broken by lowest label index (the reference's sort is
 * unstable there).
 * somi_ap_per_class_f64 = utils/metrics.py:21-95 `ap_per_class`: tp (N,T) bytes, conf / pred_cls (N), target_cls (M) with
 * integer-valued class ids in [0, ncap).  Outputs for the classes present among the targets, ascending: out_classes,
 * *out_n, ap (n,T), and p / r / f1 at the confidence of best mean F1 - all in fp64 like numpy.  Equal confidences keep
 * their input order. */
int somi_val_match_f32(const float *det, const int *det_off, const float *labels, const int *lab_off, const float *iouv, int T,
                       int B, int max_det, int max_labels, uint8_t *correct, somi_stream_t stream);
/* somi_confusion_matrix_f32 = utils/metrics.py:98-142 `ConfusionMatrix.process_batch` (val.py:141,186) for a batch of images in
 * the layout of somi_val_match_f32: detections above `conf` are matched to labels regardless of class (best label per detection,
 * then best detection per label, IoU > iou_thres) and counted into matrix[(nc+1) x (nc+1)] int32, row = predicted class, column =
 * true class, index nc = background; counts are ADDED (zero the matrix once).  An image's unmatched detections count only if the
 * image has at least one match, as in the reference.  Classes outside [0, nc] are ignored. */
int somi_confusion_matrix_f32(const float *det, const int *det_off, const float *labels, const int *lab_off, int B, int max_det,
                              int max_labels, int nc, float conf, float iou_thres, int32_t *matrix, somi_stream_t stream);
size_t somi_ap_per_class_workspace_bytes(long N, int T, int ncap);
int somi_ap_per_class_f64(const uint8_t *tp, const float *conf, const float *pred_cls, const float *target_cls, long N, long M,
                          int T, int ncap, int *out_classes, int *out_n, double *out_ap, double *out_p, double *out_r,
                          double *out_f1, void *workspace, size_t workspace_bytes, somi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Input pipeline on device: what `LoadImagesAndLabels.__getitem__` does to the pixels of one
 * sample once the source images are cached at the training size - mosaic composition (utils/datasets.py:732-777), the
 * affine crop (`random_perspective` -> cv2.warpAffine, utils/augmentations.py:126-169), mixup (:305-310), HSV jitter
 * (`augment_hsv` :47-61), flips and the HWC BGR -> CHW RGB transpose (datasets.py:648-671) - fused into one pass per output
 * pixel: the 2s x 2s mosaic canvas is never materialised.  Random draws, matrices and label boxes stay with the caller
 * (they are a few hundred scalars per sample); it passes one `somi_aug_sample` per output image, in device memory.
 *
 * canvas(x, y) = the LAST source whose rectangle [x1,x2) x [y1,y2) holds (x, y), read at pixels[y - dy][x - dx], else `fill`.
 * warp == 0: out(x, y) = canvas(x, y) (letterbox: one source placed at (left, top)).
 * warp == 1: out(x, y) = bilinear sample of the canvas at minv * (x, y, 1), taps outside the canvas = `fill`, in the fixed
 *            point of OpenCV 4.9 warpAffine (coordinates to 1/32 pixel from round(m*1024) terms, 15-bit weights).
 * mix:  out = (uint8) trunc(canvas[0] * mix_r + canvas[1] * (1 - mix_r)) in fp64.
 * hsv:  BGR -> HSV (OpenCV 8-bit, hue range 180), channel-wise through lut[0..2], HSV -> BGR.
 * flips move the pixel, then it is stored as out[b][2 - c][y][x] (RGB planes).
 * Sources and output are at most 16384 px a side.  The caller guarantees that every rectangle maps inside its source
 * (0 <= x1 - dx, x2 - dx <= w, same for y): the records live in device memory, so the library cannot check them. */
typedef struct somi_aug_source {
    const uint8_t *pixels;       /* device, (h, w, 3) BGR, tightly packed */
    int32_t h, w;
    int32_t x1, y1, x2, y2;
    int32_t dx, dy;
} somi_aug_source;

typedef struct somi_aug_canvas {
    somi_aug_source src[4];
    int32_t nsrc, height, width, warp;
    double minv[6];              /* dst -> src, row-major 2x3 (the inverse of the matrix cv2.warpAffine is given) */
} somi_aug_canvas;

typedef struct somi_aug_sample {
    somi_aug_canvas canvas[2];
    int32_t mix, hsv, flipud, fliplr;
    double mix_r;
    uint8_t lut[3][256];
} somi_aug_sample;

int somi_augment_u8(const somi_aug_sample *samples, int B, int H, int W, int fill, uint8_t *out, somi_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SOMI_HIP_H */
