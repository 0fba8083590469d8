// DCNv3 forward / backward for the other two dtypes the reference extension dispatches (AT_DISPATCH_FLOATING_TYPES_AND_HALF,
// models/ops_dcnv3/src/cuda/dcnv3_cuda.cu:69,136): fp16 (the AMP path of train.py:263: storage half, arithmetic fp32 - the
// reference's `opmath_t` - and fp32 gradient buffers, dcnv3_cuda.cu:126-133) and fp64 (the reference test's exact-parity mode,
// models/ops_dcnv3/test.py:55).  fp32 - the arithmetic of the path - has the tiled kernels of dcnv3.hip; these two are completeness
// entries of the drop-in boundary, written the plain way: one lane per (output pixel, channel) with the channel fastest, so a
// pixel's reads and writes are contiguous; every lane redoes the bilinear set-up of its group's K sampling points (what the
// reference's own kernel does, dcnv3_im2col_cuda.cu:216-275), the backward adds into grad_input / grad_offset / grad_mask with
// atomics of the accumulation type (as :116-146, :211-213 do).
#include <hip/hip_fp16.h>
#include "common.h"

namespace somi {

struct DcnGeo {
    int N, H, W, G, Gc, C, Ho, Wo, kh, kw, K, sh, sw, ph, pw, dh, dw;
    double offset_scale;
};

template <typename T> struct Load;
template <> struct Load<__half> { static __device__ __forceinline__ float get(const __half *p, long i) { return __half2float(p[i]); } };
template <> struct Load<double> { static __device__ __forceinline__ double get(const double *p, long i) { return p[i]; } };
__device__ __forceinline__ void store_out(__half *p, long i, float v) { p[i] = __float2half_rn(v); }
__device__ __forceinline__ void store_out(double *p, long i, double v) { p[i] = v; }

// One sampling point: location, validity (dcnv3_im2col_cuda.cuh:249-263), the four taps' element offsets and validity bits.
template <typename A>
struct Point {
    A lh, lw;
    long off[4];
    int bits;
};
template <typename A>
__device__ __forceinline__ Point<A> make_point(const DcnGeo &g, int ho, int wo, int k, A ox, A oy) {
    const int i = k / g.kh, j = k % g.kh;                                // kernel_w outer, kernel_h inner (:253-254)
    const int half_w = (g.dw * (g.kw - 1)) >> 1, half_h = (g.dh * (g.kh - 1)) >> 1;
    const A osc = (A)g.offset_scale;
    const A p0w = (A)(half_w - g.pw + wo * g.sw) - (A)half_w * osc;
    const A p0h = (A)(half_h - g.ph + ho * g.sh) - (A)half_h * osc;
    const A loc_w = p0w + ((A)(i * g.dw) + ox) * osc;
    const A loc_h = p0h + ((A)(j * g.dh) + oy) * osc;
    Point<A> r;
    r.bits = 0;
    r.lh = r.lw = (A)0;
    for (int t = 0; t < 4; ++t) r.off[t] = 0;
    if (!(loc_h > (A)-1 && loc_w > (A)-1 && loc_h < (A)g.H && loc_w < (A)g.W)) return r;
    const A fh = floor(loc_h), fw = floor(loc_w);
    const int h0 = (int)fh, w0 = (int)fw;
    r.lh = loc_h - fh;
    r.lw = loc_w - fw;
    const bool h0ok = h0 >= 0, h1ok = h0 + 1 <= g.H - 1, w0ok = w0 >= 0, w1ok = w0 + 1 <= g.W - 1;
    const int hc0 = h0 < 0 ? 0 : h0, hc1 = h0 + 1 > g.H - 1 ? g.H - 1 : h0 + 1;
    const int wc0 = w0 < 0 ? 0 : w0, wc1 = w0 + 1 > g.W - 1 ? g.W - 1 : w0 + 1;
    r.off[0] = ((long)hc0 * g.W + wc0) * g.C;
    r.off[1] = ((long)hc0 * g.W + wc1) * g.C;
    r.off[2] = ((long)hc1 * g.W + wc0) * g.C;
    r.off[3] = ((long)hc1 * g.W + wc1) * g.C;
    r.bits = ((h0ok && w0ok) ? 1 : 0) | ((h0ok && w1ok) ? 2 : 0) | ((h1ok && w0ok) ? 4 : 0) | ((h1ok && w1ok) ? 8 : 0);
    return r;
}

template <typename T, typename A>
__global__ __launch_bounds__(256) void dcnv3_fwd_typed_kernel(const T *__restrict__ input, const T *__restrict__ offset, const T *__restrict__ mask,
                                                              T *__restrict__ output, const DcnGeo g) {
    const long items = (long)g.N * g.Ho * g.Wo * g.C, img = (long)g.H * g.W * g.C;
    for (long it = blockIdx.x * 256L + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
        const int c = (int)(it % g.C);
        const long pix = it / g.C;
        const int wo = (int)(pix % g.Wo), ho = (int)((pix / g.Wo) % g.Ho);
        const long n = pix / ((long)g.Wo * g.Ho);
        const int grp = c / g.Gc;
        const T *src = input + n * img + c;
        A acc = (A)0;
        for (int k = 0; k < g.K; ++k) {
            const long s = (pix * g.G + grp) * g.K + k;
            const Point<A> p = make_point<A>(g, ho, wo, k, Load<T>::get(offset, s * 2), Load<T>::get(offset, s * 2 + 1));
            if (!p.bits) continue;
            const A hh = (A)1 - p.lh, hw = (A)1 - p.lw;
            const A v1 = (p.bits & 1) ? Load<T>::get(src, p.off[0]) : (A)0, v2 = (p.bits & 2) ? Load<T>::get(src, p.off[1]) : (A)0;
            const A v3 = (p.bits & 4) ? Load<T>::get(src, p.off[2]) : (A)0, v4 = (p.bits & 8) ? Load<T>::get(src, p.off[3]) : (A)0;
            acc += Load<T>::get(mask, s) * (hh * hw * v1 + hh * p.lw * v2 + p.lh * hw * v3 + p.lh * p.lw * v4);
        }
        store_out(output, it, acc);
    }
}

// grad_input / grad_offset / grad_mask are of the accumulation type A (fp32 for half, like the reference's buffers) and zeroed.
template <typename T, typename A>
__global__ __launch_bounds__(256) void dcnv3_bwd_typed_kernel(const T *__restrict__ input, const T *__restrict__ offset, const T *__restrict__ mask,
                                                              const T *__restrict__ grad_output, A *grad_input, A *grad_offset, A *grad_mask,
                                                              const DcnGeo g) {
    const long items = (long)g.N * g.Ho * g.Wo * g.C, img = (long)g.H * g.W * g.C;
    for (long it = blockIdx.x * 256L + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
        const int c = (int)(it % g.C);
        const long pix = it / g.C;
        const int wo = (int)(pix % g.Wo), ho = (int)((pix / g.Wo) % g.Ho);
        const long n = pix / ((long)g.Wo * g.Ho);
        const int grp = c / g.Gc;
        const T *src = input + n * img + c;
        A *gin = grad_input + n * img + c;
        const A tg = Load<T>::get(grad_output, it);
        for (int k = 0; k < g.K; ++k) {
            const long s = (pix * g.G + grp) * g.K + k;
            const Point<A> p = make_point<A>(g, ho, wo, k, Load<T>::get(offset, s * 2), Load<T>::get(offset, s * 2 + 1));
            if (!p.bits) continue;
            const A m = Load<T>::get(mask, s), tm = tg * m;
            const A hh = (A)1 - p.lh, hw = (A)1 - p.lw;
            const A w1 = hh * hw, w2 = hh * p.lw, w3 = p.lh * hw, w4 = p.lh * p.lw;
            A v1 = (A)0, v2 = (A)0, v3 = (A)0, v4 = (A)0;
            if (p.bits & 1) { v1 = Load<T>::get(src, p.off[0]); atomicAdd(gin + p.off[0], w1 * tm); }
            if (p.bits & 2) { v2 = Load<T>::get(src, p.off[1]); atomicAdd(gin + p.off[1], w2 * tm); }
            if (p.bits & 4) { v3 = Load<T>::get(src, p.off[2]); atomicAdd(gin + p.off[2], w3 * tm); }
            if (p.bits & 8) { v4 = Load<T>::get(src, p.off[3]); atomicAdd(gin + p.off[3], w4 * tm); }
            // dcnv3_col2im_bilinear: grad_h_weight / grad_w_weight (dcnv3_im2col_cuda.cuh:112-141), scaled by offset_scale (:144-145)
            const A ghw = -hw * v1 - p.lw * v2 + hw * v3 + p.lw * v4;
            const A gww = -hh * v1 + hh * v2 - p.lh * v3 + p.lh * v4;
            atomicAdd(grad_mask + s, tg * (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4));
            atomicAdd(grad_offset + s * 2, (A)g.offset_scale * gww * tm);
            atomicAdd(grad_offset + s * 2 + 1, (A)g.offset_scale * ghw * tm);
        }
    }
}

static int fill_geo(DcnGeo &a, int N, int H, int W, int G, int Gc, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                    float offset_scale, int im2col_step) {
    SOMI_REQUIRE(N > 0 && H > 0 && W > 0 && G > 0 && Gc > 0 && kh > 0 && kw > 0 && sh > 0 && sw > 0 && dh > 0 && dw > 0 && ph >= 0 && pw >= 0,
                 SOMI_EINVAL, "dcnv3: bad geometry");
    const int step = N < im2col_step ? N : im2col_step;
    SOMI_REQUIRE(im2col_step > 0 && N % step == 0, SOMI_EINVAL, "batch(%d) must divide im2col_step(%d)", N, step);
    a.N = N; a.H = H; a.W = W; a.G = G; a.Gc = Gc; a.C = G * Gc;
    a.kh = kh; a.kw = kw; a.K = kh * kw; a.sh = sh; a.sw = sw; a.ph = ph; a.pw = pw; a.dh = dh; a.dw = dw;
    a.Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;
    a.Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
    SOMI_REQUIRE(a.Ho > 0 && a.Wo > 0, SOMI_EINVAL, "dcnv3: empty output");
    a.offset_scale = (double)offset_scale;
    return 0;
}
static inline int grid_for(long items) {
    long g = (items + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

template <typename T, typename A>
static int run_fwd(const void *input, const void *offset, const void *mask, void *output, const DcnGeo &g, hipStream_t s, const char *what) {
    hipLaunchKernelGGL((dcnv3_fwd_typed_kernel<T, A>), dim3(grid_for((long)g.N * g.Ho * g.Wo * g.C)), dim3(256), 0, s, static_cast<const T *>(input),
                       static_cast<const T *>(offset), static_cast<const T *>(mask), static_cast<T *>(output), g);
    return launch_status(what);
}
template <typename T, typename A>
static int run_bwd(const void *input, const void *offset, const void *mask, const void *grad_output, void *gi, void *go, void *gm, const DcnGeo &g,
                   hipStream_t s, const char *what) {
    const size_t npt = (size_t)g.N * g.Ho * g.Wo * g.G * g.K;
    (void)hipMemsetAsync(go, 0, npt * 2 * sizeof(A), s);
    (void)hipMemsetAsync(gm, 0, npt * sizeof(A), s);
    hipLaunchKernelGGL((dcnv3_bwd_typed_kernel<T, A>), dim3(grid_for((long)g.N * g.Ho * g.Wo * g.C)), dim3(256), 0, s, static_cast<const T *>(input),
                       static_cast<const T *>(offset), static_cast<const T *>(mask), static_cast<const T *>(grad_output), static_cast<A *>(gi),
                       static_cast<A *>(go), static_cast<A *>(gm), g);
    return launch_status(what);
}

}  // namespace somi

using namespace somi;

#define DCN_GEO_ARGS N, H, W, G, Gc, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, offset_scale, im2col_step

extern "C" int somi_dcnv3_forward_f16(const void *input, const void *offset, const void *mask, void *output, int N, int H, int W, int G, int Gc,
                                      int kernel_h, int kernel_w, int stride_h, int stride_w, int pad_h, int pad_w, int dilation_h, int dilation_w,
                                      float offset_scale, int im2col_step, somi_stream_t stream) {
    SOMI_REQUIRE(input && offset && mask && output, SOMI_EINVAL, "dcnv3 forward: null tensor");
    DcnGeo g{};
    if (int rc = fill_geo(g, DCN_GEO_ARGS)) return rc;
    return run_fwd<__half, float>(input, offset, mask, output, g, (hipStream_t)stream, "somi_dcnv3_forward_f16");
}

extern "C" int somi_dcnv3_forward_f64(const double *input, const double *offset, const double *mask, double *output, int N, int H, int W, int G,
                                      int Gc, int kernel_h, int kernel_w, int stride_h, int stride_w, int pad_h, int pad_w, int dilation_h,
                                      int dilation_w, float offset_scale, int im2col_step, somi_stream_t stream) {
    SOMI_REQUIRE(input && offset && mask && output, SOMI_EINVAL, "dcnv3 forward: null tensor");
    DcnGeo g{};
    if (int rc = fill_geo(g, DCN_GEO_ARGS)) return rc;
    return run_fwd<double, double>(input, offset, mask, output, g, (hipStream_t)stream, "somi_dcnv3_forward_f64");
}

extern "C" int somi_dcnv3_backward_f16(const void *input, const void *offset, const void *mask, const void *grad_output, float *grad_input,
                                       float *grad_offset, float *grad_mask, int N, int H, int W, int G, int Gc, int kernel_h, int kernel_w,
                                       int stride_h, int stride_w, int pad_h, int pad_w, int dilation_h, int dilation_w, float offset_scale,
                                       int im2col_step, void *workspace, size_t workspace_bytes, somi_stream_t stream) {
    (void)workspace; (void)workspace_bytes;                      // uniform signature with the _f32 entry
    SOMI_REQUIRE(input && offset && mask && grad_output && grad_input && grad_offset && grad_mask, SOMI_EINVAL, "dcnv3 backward: null tensor");
    DcnGeo g{};
    if (int rc = fill_geo(g, DCN_GEO_ARGS)) return rc;
    return run_bwd<__half, float>(input, offset, mask, grad_output, grad_input, grad_offset, grad_mask, g, (hipStream_t)stream, "somi_dcnv3_backward_f16");
}

extern "C" int somi_dcnv3_backward_f64(const double *input, const double *offset, const double *mask, const double *grad_output, double *grad_input,
                                       double *grad_offset, double *grad_mask, int N, int H, int W, int G, int Gc, int kernel_h, int kernel_w,
                                       int stride_h, int stride_w, int pad_h, int pad_w, int dilation_h, int dilation_w, float offset_scale,
                                       int im2col_step, void *workspace, size_t workspace_bytes, somi_stream_t stream) {
    (void)workspace; (void)workspace_bytes;                      // uniform signature with the _f32 entry
    SOMI_REQUIRE(input && offset && mask && grad_output && grad_input && grad_offset && grad_mask, SOMI_EINVAL, "dcnv3 backward: null tensor");
    DcnGeo g{};
    if (int rc = fill_geo(g, DCN_GEO_ARGS)) return rc;
    return run_bwd<double, double>(input, offset, mask, grad_output, grad_input, grad_offset, grad_mask, g, (hipStream_t)stream, "somi_dcnv3_backward_f64");
}
