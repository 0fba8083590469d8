// DCNv3 forward / backward for gfx950.
//
// Replaces the reference extension's kernels (models/ops_dcnv3/src/cuda/dcnv3_im2col_cuda.cuh:216-275 forward,
// :82-147 + :278-839 backward) and host launchers (src/cuda/dcnv3_cuda.cu:21-174).
//
// The reference runs one thread per output scalar and every one of the Gc channel-threads of a sampling site
// re-reads that site's 3*K offset/mask floats and redoes the bilinear set-up.  Here a workgroup owns a tile of TP
// consecutive output pixels:
//   phase 1  every (pixel, group, point) "sampling record" of the tile is built ONCE: offset (8 B/lane) and mask
//            (4 B/lane) are read with unit-stride coalesced loads, the bilinear set-up is done by one lane per
//            record, and the record (4 tap offsets + 4 tap coefficients already multiplied by the mask; or the
//            fractions + validity bits for backward) is parked in LDS (32 B per record);
//   phase 2  lanes sweep (pixel, channel-quad) items: each lane pulls its group's record from LDS (broadcast across
//            the Gc/4 lanes of a group), gathers the 4 taps as 16 B loads (a group's channels are contiguous in
//            NHWC, so a tap is one 4*Gc-byte run) and accumulates; outputs leave as coalesced 16 B stores.
// HBM-bound: algorithmic bytes per output pixel = 4*(2*C + 3*G*K) forward (input once + output + offset + mask).
// Backward adds the grad_output read, fp32 atomics into grad_input (as the reference does) and the in-wave
// reduction over a group's channels for grad_offset / grad_mask (DPP/shuffle butterflies instead of the
// reference's shared-memory tree).
#include "common.h"

namespace somi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct DcnArgs {
    const float *input, *offset, *mask, *grad_output;
    float *output, *grad_input, *grad_offset, *grad_mask;
    int N, H, W, G, Gc, C, Ho, Wo, kh, kw, K, sh, sw, ph, pw, dh, dw;
    float offset_scale;
    int TP;          // pixels per tile
    long npix;       // N*Ho*Wo
    int ntile;
};

struct Rec {         // 32 B, one per (pixel, group, point)
    i32x4 off;       // element offsets of the 4 taps inside the image (clamped into range)
    f32x4 f;         // fwd: tap coefficients * mask ; bwd: {lh, lw, mask, validity bits as int}
};

// Builds the record for (pix, g, k).  Point order inside K: i over kernel_w outer, j over kernel_h inner
// (dcnv3_im2col_cuda.cuh:253-254), offsets stored (x, y).
template <bool BWD>
__device__ __forceinline__ Rec make_record(const DcnArgs &a, long pix, int g, int k) {
    const int wo = (int)(pix % a.Wo), ho = (int)((pix / a.Wo) % a.Ho);
    const long s = (pix * a.G + g) * a.K + k;
    const float2 ofs = *reinterpret_cast<const float2 *>(a.offset + s * 2);
    const float m = a.mask[s];
    const int i = k / a.kh, j = k % a.kh;
    const int half_w = (a.dw * (a.kw - 1)) >> 1, half_h = (a.dh * (a.kh - 1)) >> 1;
    const float p0w = (float)(half_w - a.pw + wo * a.sw) - (float)half_w * a.offset_scale;
    const float p0h = (float)(half_h - a.ph + ho * a.sh) - (float)half_h * a.offset_scale;
    const float loc_w = p0w + ((float)(i * a.dw) + ofs.x) * a.offset_scale;
    const float loc_h = p0h + ((float)(j * a.dh) + ofs.y) * a.offset_scale;
    Rec r;
    const bool use = loc_h > -1.f && loc_w > -1.f && loc_h < (float)a.H && loc_w < (float)a.W;
    const float fh = floorf(loc_h), fw = floorf(loc_w);
    const int h0 = use ? (int)fh : 0, w0 = use ? (int)fw : 0;
    const float lh = loc_h - fh, lw = loc_w - fw;
    const bool h0ok = use && h0 >= 0, h1ok = use && h0 + 1 <= a.H - 1;
    const bool w0ok = w0 >= 0, w1ok = w0 + 1 <= a.W - 1;
    const int hc0 = h0 < 0 ? 0 : h0, hc1 = h0 + 1 > a.H - 1 ? a.H - 1 : h0 + 1;
    const int wc0 = w0 < 0 ? 0 : w0, wc1 = w0 + 1 > a.W - 1 ? a.W - 1 : w0 + 1;
    r.off[0] = (hc0 * a.W + wc0) * a.C;
    r.off[1] = (hc0 * a.W + wc1) * a.C;
    r.off[2] = (hc1 * a.W + wc0) * a.C;
    r.off[3] = (hc1 * a.W + wc1) * a.C;
    const bool t0 = h0ok && w0ok, t1 = h0ok && w1ok, t2 = h1ok && w0ok, t3 = h1ok && w1ok;
    if constexpr (!BWD) {
        const float hh = 1.f - lh, hw = 1.f - lw;
        r.f[0] = t0 ? hh * hw * m : 0.f;
        r.f[1] = t1 ? hh * lw * m : 0.f;
        r.f[2] = t2 ? lh * hw * m : 0.f;
        r.f[3] = t3 ? lh * lw * m : 0.f;
    } else {
        r.f[0] = lh;
        r.f[1] = lw;
        r.f[2] = m;
        r.f[3] = __int_as_float((t0 ? 1 : 0) | (t1 ? 2 : 0) | (t2 ? 4 : 0) | (t3 ? 8 : 0));
    }
    return r;
}

// ------------------------------------------------------------------------------------------------ forward
// VEC = 4: one lane = 4 consecutive channels of one group (needs Gc % 4 == 0); VEC = 1: one lane = one channel.
template <int VEC>
__global__ __launch_bounds__(256) void dcnv3_fwd_kernel(const DcnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Rec *recs = reinterpret_cast<Rec *>(smem);
    const int tile = xcd_remap(blockIdx.x, a.ntile);
    const long pix0 = (long)tile * a.TP;
    const int np = (int)min((long)a.TP, a.npix - pix0);
    const int GK = a.G * a.K;
    for (int i = threadIdx.x; i < np * GK; i += 256) {
        const int pl = i / GK, gk = i % GK;
        recs[i] = make_record<false>(a, pix0 + pl, gk / a.K, gk % a.K);
    }
    __syncthreads();
    const int CV = a.C / VEC;
    const long img = (long)a.H * a.W * a.C;
    for (int it = threadIdx.x; it < np * CV; it += 256) {
        const int pl = it / CV, c = (it % CV) * VEC;
        const long pix = pix0 + pl;
        const int n = (int)(pix / ((long)a.Ho * a.Wo));
        const int g = c / a.Gc;
        const float *src = a.input + n * img + c;
        const Rec *rr = recs + (pl * a.G + g) * a.K;
        if constexpr (VEC == 4) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            int k = 0;
            for (; k + 3 <= a.K; k += 3) {                                 // 3 points = 12 gathers in flight; summation order unchanged
                const i32x4 o0 = rr[k].off, o1 = rr[k + 1].off, o2 = rr[k + 2].off;
                const f32x4 f0 = rr[k].f, f1 = rr[k + 1].f, f2 = rr[k + 2].f;
                f32x4 u[12];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    u[t] = *reinterpret_cast<const f32x4 *>(src + o0[t]);
                    u[4 + t] = *reinterpret_cast<const f32x4 *>(src + o1[t]);
                    u[8 + t] = *reinterpret_cast<const f32x4 *>(src + o2[t]);
                }
                acc += f0[0] * u[0] + f0[1] * u[1] + f0[2] * u[2] + f0[3] * u[3];
                acc += f1[0] * u[4] + f1[1] * u[5] + f1[2] * u[6] + f1[3] * u[7];
                acc += f2[0] * u[8] + f2[1] * u[9] + f2[2] * u[10] + f2[3] * u[11];
            }
            for (; k < a.K; ++k) {
                const i32x4 o = rr[k].off;
                const f32x4 f = rr[k].f;
                const f32x4 v0 = *reinterpret_cast<const f32x4 *>(src + o[0]);
                const f32x4 v1 = *reinterpret_cast<const f32x4 *>(src + o[1]);
                const f32x4 v2 = *reinterpret_cast<const f32x4 *>(src + o[2]);
                const f32x4 v3 = *reinterpret_cast<const f32x4 *>(src + o[3]);
                acc += f[0] * v0 + f[1] * v1 + f[2] * v2 + f[3] * v3;
            }
            *reinterpret_cast<f32x4 *>(a.output + pix * a.C + c) = acc;
        } else {
            float acc = 0.f;
            for (int k = 0; k < a.K; ++k) {
                const i32x4 o = rr[k].off;
                const f32x4 f = rr[k].f;
                acc += f[0] * src[o[0]] + f[1] * src[o[1]] + f[2] * src[o[2]] + f[3] * src[o[3]];
            }
            a.output[pix * a.C + c] = acc;
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward
// LDS: records, then 3 float accumulators per record (grad_mask, grad_w, grad_h partial sums over channels).
template <int VEC>
__global__ __launch_bounds__(256) void dcnv3_bwd_kernel(const DcnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int GK = a.G * a.K;
    Rec *recs = reinterpret_cast<Rec *>(smem);
    float *accs = reinterpret_cast<float *>(smem + (size_t)a.TP * GK * sizeof(Rec));   // [np*GK][3]
    const int tile = xcd_remap(blockIdx.x, a.ntile);
    const long pix0 = (long)tile * a.TP;
    const int np = (int)min((long)a.TP, a.npix - pix0);
    for (int i = threadIdx.x; i < np * GK; i += 256) {
        const int pl = i / GK, gk = i % GK;
        recs[i] = make_record<true>(a, pix0 + pl, gk / a.K, gk % a.K);
        accs[i * 3] = 0.f;
        accs[i * 3 + 1] = 0.f;
        accs[i * 3 + 2] = 0.f;
    }
    __syncthreads();
    const int CV = a.C / VEC;
    const long img = (long)a.H * a.W * a.C;
    const int LG = a.Gc / VEC;                                 // lanes per group
    // groups are LG consecutive lanes, aligned because CV = G*LG: butterflies need LG to be a power of two <= 64
    const bool shuffle = LG <= 64 && (LG & (LG - 1)) == 0;
    const int lane = threadIdx.x & 63;
    const int nit = np * CV;
    const int nit_pad = (nit + 255) / 256 * 256;               // keep whole waves in the loop for the shuffles
    for (int it = threadIdx.x; it < nit_pad; it += 256) {
        const bool live = it < nit;
        const int itc = live ? it : nit - 1;
        const int pl = itc / CV, c = (itc % CV) * VEC;
        const long pix = pix0 + pl;
        const int n = (int)(pix / ((long)a.Ho * a.Wo));
        const int g = c / a.Gc;
        const float *src = a.input + n * img + c;
        float *gin = a.grad_input + n * img + c;
        const int rbase = (pl * a.G + g) * a.K;
        f32x4 tg = {0.f, 0.f, 0.f, 0.f};
        if (live) {
            if constexpr (VEC == 4) tg = *reinterpret_cast<const f32x4 *>(a.grad_output + pix * a.C + c);
            else tg[0] = a.grad_output[pix * a.C + c];
        }
        for (int k = 0; k < a.K; ++k) {
            const Rec r = recs[rbase + k];
            const float lh = r.f[0], lw = r.f[1], m = r.f[2];
            const int bits = __float_as_int(r.f[3]);
            const float hh = 1.f - lh, hw = 1.f - lw;
            float gm = 0.f, gw = 0.f, gh = 0.f;
            if (bits) {                                        // wave-divergent only where groups differ
                const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float t = tg[e], tm = t * m;
                    float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
                    if (bits & 1) { v1 = src[r.off[0] + e]; if (live) atomicAdd(gin + r.off[0] + e, w1 * tm); }
                    if (bits & 2) { v2 = src[r.off[1] + e]; if (live) atomicAdd(gin + r.off[1] + e, w2 * tm); }
                    if (bits & 4) { v3 = src[r.off[2] + e]; if (live) atomicAdd(gin + r.off[2] + e, w3 * tm); }
                    if (bits & 8) { v4 = src[r.off[3] + e]; if (live) atomicAdd(gin + r.off[3] + e, w4 * tm); }
                    // dcnv3_col2im_bilinear: grad_h_weight / grad_w_weight (dcnv3_im2col_cuda.cuh:112-141)
                    const float ghw = -hw * v1 - lw * v2 + hw * v3 + lw * v4;
                    const float gww = -hh * v1 + hh * v2 - lh * v3 + lh * v4;
                    gm += t * (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
                    gw += gww * tm;
                    gh += ghw * tm;
                }
            }
            if (shuffle) {
                for (int o = LG >> 1; o > 0; o >>= 1) {
                    gm += __shfl_xor(gm, o);
                    gw += __shfl_xor(gw, o);
                    gh += __shfl_xor(gh, o);
                }
                if (live && (lane & (LG - 1)) == 0) {          // one lane per group owns the record: plain store
                    float *ac = accs + (rbase + k) * 3;
                    ac[0] = gm;
                    ac[1] = gw;
                    ac[2] = gh;
                }
            } else if (live) {
                float *ac = accs + (rbase + k) * 3;
                atomicAdd(ac, gm);
                atomicAdd(ac + 1, gw);
                atomicAdd(ac + 2, gh);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < np * GK; i += 256) {
        const long s = (pix0 * GK) + i;
        a.grad_mask[s] = accs[i * 3];
        *reinterpret_cast<float2 *>(a.grad_offset + s * 2) =
            make_float2(a.offset_scale * accs[i * 3 + 1], a.offset_scale * accs[i * 3 + 2]);
    }
}

static int fill_args(DcnArgs &a, int N, int H, int W, int G, int Gc, int kh, int kw, int sh, int sw, int ph, int pw, int dh,
                     int dw, float offset_scale, int im2col_step, size_t rec_bytes) {
    SOMI_REQUIRE(N > 0 && H > 0 && W > 0 && G > 0 && Gc > 0 && kh > 0 && kw > 0 && sh > 0 && sw > 0 && dh > 0 && dw > 0 &&
                     ph >= 0 && pw >= 0, SOMI_EINVAL, "dcnv3: bad geometry");
    const int step = N < im2col_step ? N : im2col_step;
    SOMI_REQUIRE(im2col_step > 0 && N % step == 0, SOMI_EINVAL, "batch(%d) must divide im2col_step(%d)", N, step);
    a.N = N; a.H = H; a.W = W; a.G = G; a.Gc = Gc; a.C = G * Gc;
    a.kh = kh; a.kw = kw; a.K = kh * kw; a.sh = sh; a.sw = sw; a.ph = ph; a.pw = pw; a.dh = dh; a.dw = dw;
    a.Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;
    a.Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
    SOMI_REQUIRE(a.Ho > 0 && a.Wo > 0, SOMI_EINVAL, "dcnv3: empty output");
    SOMI_REQUIRE((long)H * W * a.C < (1L << 31), SOMI_EINVAL, "dcnv3: one image must stay below 2^31 elements");
    a.offset_scale = offset_scale;
    a.npix = (long)N * a.Ho * a.Wo;
    const size_t per_pix = (size_t)G * a.K * rec_bytes;
    SOMI_REQUIRE(per_pix <= 60 * 1024, SOMI_ENOTIMPL, "dcnv3: G*K = %d does not fit one LDS tile", G * a.K);
    int tp = (int)((48 * 1024) / per_pix);
    if (tp < 1) tp = 1;
    if (tp > 16) tp = 16;
    a.TP = tp;
    a.ntile = (int)((a.npix + tp - 1) / tp);
    return 0;
}

}  // namespace somi

using namespace somi;

extern "C" int somi_dcnv3_forward_f32(const float *input, const float *offset, const float *mask, float *output, int N, int H,
                                      int W, int G, int Gc, int kernel_h, int kernel_w, int stride_h, int stride_w, int pad_h,
                                      int pad_w, int dilation_h, int dilation_w, float offset_scale, int im2col_step,
                                      somi_stream_t stream) {
    SOMI_REQUIRE(input && offset && mask && output, SOMI_EINVAL, "dcnv3 forward: null tensor");
    DcnArgs a{};
    int rc = fill_args(a, N, H, W, G, Gc, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w,
                       offset_scale, im2col_step, sizeof(Rec));
    if (rc) return rc;
    a.input = input; a.offset = offset; a.mask = mask; a.output = output;
    SOMI_REQUIRE((reinterpret_cast<uintptr_t>(offset) & 7u) == 0, SOMI_EINVAL, "dcnv3: offset must be 8 B aligned");
    const size_t lds = (size_t)a.TP * G * a.K * sizeof(Rec);
    const bool vec = (Gc % 4 == 0) && aligned16(input) && aligned16(output);
    if (vec) hipLaunchKernelGGL(dcnv3_fwd_kernel<4>, dim3(a.ntile), dim3(256), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(dcnv3_fwd_kernel<1>, dim3(a.ntile), dim3(256), lds, (hipStream_t)stream, a);
    return launch_status("somi_dcnv3_forward_f32");
}

extern "C" int somi_dcnv3_backward_f32(const float *input, const float *offset, const float *mask, const float *grad_output,
                                       float *grad_input, float *grad_offset, float *grad_mask, int N, int H, int W, int G,
                                       int Gc, int kernel_h, int kernel_w, int stride_h, int stride_w, int pad_h, int pad_w,
                                       int dilation_h, int dilation_w, float offset_scale, int im2col_step,
                                       somi_stream_t stream) {
    SOMI_REQUIRE(input && offset && mask && grad_output && grad_input && grad_offset && grad_mask, SOMI_EINVAL,
                 "dcnv3 backward: null tensor");
    DcnArgs a{};
    int rc = fill_args(a, N, H, W, G, Gc, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w,
                       offset_scale, im2col_step, sizeof(Rec) + 3 * sizeof(float));
    if (rc) return rc;
    a.input = input; a.offset = offset; a.mask = mask; a.grad_output = grad_output;
    a.grad_input = grad_input; a.grad_offset = grad_offset; a.grad_mask = grad_mask;
    SOMI_REQUIRE((reinterpret_cast<uintptr_t>(offset) & 7u) == 0 && (reinterpret_cast<uintptr_t>(grad_offset) & 7u) == 0,
                 SOMI_EINVAL, "dcnv3: offset / grad_offset must be 8 B aligned");
    const size_t lds = (size_t)a.TP * G * a.K * (sizeof(Rec) + 3 * sizeof(float));
    // one channel per lane: a wave's fp32 atomic covers 256 contiguous bytes of grad_input (the shape that runs at the
    // chip-wide atomic rate, MI355X_MICROARCH.md "Global float atomics"); 4 channels per lane (16 B stride between lanes'
    // dwords) measured 4x slower.  The float4 form is kept for group widths where it avoids the LDS-atomic fallback.
    const bool vec = (Gc % 4 == 0) && aligned16(grad_output) && !((Gc & (Gc - 1)) == 0 && Gc <= 64);
    if (vec) hipLaunchKernelGGL(dcnv3_bwd_kernel<4>, dim3(a.ntile), dim3(256), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(dcnv3_bwd_kernel<1>, dim3(a.ntile), dim3(256), lds, (hipStream_t)stream, a);
    return launch_status("somi_dcnv3_backward_f32");
}
