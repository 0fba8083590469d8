// Bandwidth-bound layer kernels of the SOMI forward (everything that is not a dense conv):
// image ingest, depthwise 3x3, SPPF pooling, BiFPN fusion (+nearest upsample), global pooling,
// CBAM / SEAM attention heads, ODConv per-sample weight synthesis, detection decode.
// All NHWC fp32, 16 B per lane along channels, grid-stride over (pixel, channel-quad) items.
#include "common.h"

namespace somi {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int ew_grid(long items, int block = 256) {
    long g = (items + block - 1) / block;
    const long cap = 256L * 8;   // 8 workgroups per CU, grid-stride beyond that
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// ------------------------------------------------------------------------------------------------ image ingest
template <typename T, bool DIV>   // DIV: v / scale (bit-exact `imgs.float()/255`, train.py:249); else v * scale
__global__ __launch_bounds__(256) void image_to_nhwc4_kernel(const T *__restrict__ img, float *__restrict__ y, int B,
                                                             int C, int H, int W, float scale) {
    const long npix = (long)B * H * W;
    const long HW = (long)H * W;
    for (long p = blockIdx.x * 256L + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
        const long b = p / HW, hw = p % HW;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < C; ++c) {
            const float t = (float)img[(b * C + c) * HW + hw];
            v[c] = DIV ? t / scale : t * scale;
        }
        *reinterpret_cast<f32x4 *>(y + p * 4) = v;
    }
}

// ------------------------------------------------------------------------------------------------ depthwise 3x3
// One lane owns 4 channels of one image column and walks SEG rows down it with the 3x3 window (9 float4) and the 9 weight
// quads in registers: 3 loads per output instead of 18 (the one-output-per-lane form was bound by the L1 request rate at
// ~1.5 TB/s).  Lanes run over (column, channel quad), so a wave reads whole contiguous lines; the column neighbours' loads
// of the same pixels hit in L1.  FLIP walks the taps backwards: the data gradient of the same layer.
template <bool FLIP>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                        const float *__restrict__ bias, const float *__restrict__ ps,
                                                        const float *__restrict__ pt, const float *__restrict__ res,
                                                        float *__restrict__ y, int B, int H, int W, int C, int act, int SEG) {
    const int C4 = C >> 2, nseg = (H + SEG - 1) / SEG;
    const long items = (long)B * nseg * W * C4;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (long it = blockIdx.x * 256L + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
        const int c = (int)(it % C4) * 4;
        const int wv = (int)((it / C4) % W);
        const int seg = (int)((it / ((long)C4 * W)) % nseg);
        const long b = it / ((long)C4 * W * nseg);
        f32x4 wk[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) wk[k] = *reinterpret_cast<const f32x4 *>(w + (FLIP ? 8 - k : k) * C + c);
        const f32x4 bv = bias ? *reinterpret_cast<const f32x4 *>(bias + c) : zero;
        const bool wl = wv > 0, wr = wv + 1 < W;
        const float *xb = x + (b * H * W + wv) * C + c;              // (b, row 0, column wv)
        auto load_row = [&](int h, f32x4 (&row)[3]) {
            if ((unsigned)h < (unsigned)H) {
                const float *pr = xb + (long)h * W * C;
                row[0] = wl ? *reinterpret_cast<const f32x4 *>(pr - C) : zero;
                row[1] = *reinterpret_cast<const f32x4 *>(pr);
                row[2] = wr ? *reinterpret_cast<const f32x4 *>(pr + C) : zero;
            } else {
                row[0] = row[1] = row[2] = zero;
            }
        };
        const int h_lo = seg * SEG, h_hi = min(H, h_lo + SEG);
        f32x4 top[3], mid[3], bot[3];
        load_row(h_lo - 1, top);
        load_row(h_lo, mid);
        for (int h = h_lo; h < h_hi; ++h) {
            load_row(h + 1, bot);
            f32x4 acc = bv;
#pragma unroll
            for (int q = 0; q < 3; ++q) acc += top[q] * wk[q];
#pragma unroll
            for (int q = 0; q < 3; ++q) acc += mid[q] * wk[3 + q];
#pragma unroll
            for (int q = 0; q < 3; ++q) acc += bot[q] * wk[6 + q];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = apply_act_rt(acc[e], act);
            if (ps) acc = acc * *reinterpret_cast<const f32x4 *>(ps + c) + *reinterpret_cast<const f32x4 *>(pt + c);
            const long o = ((b * H + h) * W + wv) * C + c;
            if (res) acc += *reinterpret_cast<const f32x4 *>(res + o);
            *reinterpret_cast<f32x4 *>(y + o) = acc;
#pragma unroll
            for (int q = 0; q < 3; ++q) { top[q] = mid[q]; mid[q] = bot[q]; }
        }
    }
}

// The DCNv3 block's depthwise conv -> LayerNorm -> GELU chain (modules/dcnv3.py:283-291) in one pass, for C == 256: the lanes of a wave hold the 64
// channel quads of ONE image column, so each output row's LayerNorm statistics are two wave reductions of values already in registers.  Writes u
// (the conv output: the LayerNorm backward reads it) and y = act(LN(u)); the separate LayerNorm pass's read of u is gone.  Same arithmetic and
// summation order as dwconv3x3_kernel<false> + layernorm_act_kernel (bit-identical results).
__global__ __launch_bounds__(256) void dwconv3x3_ln_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                                                           const float *__restrict__ gamma, const float *__restrict__ beta, float eps, int act,
                                                           float *__restrict__ u, float *__restrict__ y, int B, int H, int W, int SEG) {
    constexpr int C = 256, C4 = 64;
    const int nseg = (H + SEG - 1) / SEG;
    const long items = (long)B * nseg * W * C4;                         // a multiple of 64: whole waves enter or leave the loop
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (long it = blockIdx.x * 256L + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
        const int c = (int)(it % C4) * 4;
        const int wv = (int)((it / C4) % W);
        const int seg = (int)((it / ((long)C4 * W)) % nseg);
        const long b = it / ((long)C4 * W * nseg);
        f32x4 wk[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) wk[k] = *reinterpret_cast<const f32x4 *>(w + k * C + c);
        const f32x4 bv = bias ? *reinterpret_cast<const f32x4 *>(bias + c) : zero;
        const f32x4 gm = *reinterpret_cast<const f32x4 *>(gamma + c), bt = *reinterpret_cast<const f32x4 *>(beta + c);
        const bool wl = wv > 0, wr = wv + 1 < W;
        const float *xb = x + (b * H * W + wv) * C + c;
        auto load_row = [&](int h, f32x4 (&row)[3]) {
            if ((unsigned)h < (unsigned)H) {
                const float *pr = xb + (long)h * W * C;
                row[0] = wl ? *reinterpret_cast<const f32x4 *>(pr - C) : zero;
                row[1] = *reinterpret_cast<const f32x4 *>(pr);
                row[2] = wr ? *reinterpret_cast<const f32x4 *>(pr + C) : zero;
            } else {
                row[0] = row[1] = row[2] = zero;
            }
        };
        const int h_lo = seg * SEG, h_hi = min(H, h_lo + SEG);
        f32x4 top[3], mid[3], bot[3];
        load_row(h_lo - 1, top);
        load_row(h_lo, mid);
        for (int h = h_lo; h < h_hi; ++h) {
            load_row(h + 1, bot);
            f32x4 acc = bv;
#pragma unroll
            for (int q = 0; q < 3; ++q) acc += top[q] * wk[q];
#pragma unroll
            for (int q = 0; q < 3; ++q) acc += mid[q] * wk[3 + q];
#pragma unroll
            for (int q = 0; q < 3; ++q) acc += bot[q] * wk[6 + q];
            const long o = ((b * H + h) * W + wv) * C + c;
            *reinterpret_cast<f32x4 *>(u + o) = acc;
            float s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
            for (int sh = 32; sh > 0; sh >>= 1) s += __shfl_xor(s, sh);
            const float mean = s / (float)C;
            const f32x4 d = acc - mean;
            float q2 = (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
            for (int sh = 32; sh > 0; sh >>= 1) q2 += __shfl_xor(q2, sh);
            const float rstd = rsqrtf(q2 / (float)C + eps);
            f32x4 v = d * rstd;
            v = v * gm + bt;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = apply_act_rt(v[e], act);
            *reinterpret_cast<f32x4 *>(y + o) = v;
#pragma unroll
            for (int q = 0; q < 3; ++q) { top[q] = mid[q]; mid[q] = bot[q]; }
        }
    }
}

void launch_dwconv3x3(bool flip, const float *x, const float *w, const float *bias, const float *ps, const float *pt, const float *res,
                      float *y, int B, int H, int W, int C, int act, hipStream_t s) {
    const int seg = H >= 64 ? 16 : 8;
    const dim3 grid(ew_grid((long)B * cdiv(H, seg) * W * (C / 4)));
    if (flip) hipLaunchKernelGGL(dwconv3x3_kernel<true>, grid, dim3(256), 0, s, x, w, bias, ps, pt, res, y, B, H, W, C, act, seg);
    else hipLaunchKernelGGL(dwconv3x3_kernel<false>, grid, dim3(256), 0, s, x, w, bias, ps, pt, res, y, B, H, W, C, act, seg);
}

// ------------------------------------------------------------------------------------------------ SPPF pooling
// chained 5x5/s1/p2 max-pools == max over 5x5, 9x9, 13x13 windows clipped to the image (-inf padding)
__global__ __launch_bounds__(256) void sppf_pool_kernel(float *__restrict__ buf, int B, int H, int W, int C, int cs,
                                                        int x_coff) {
    const int C4 = C >> 2;
    const long items = (long)B * H * W * C4;
    for (long it = blockIdx.x * 256L + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
        const int c = (int)(it % C4) * 4;
        const long pix = it / C4;
        const int wv = (int)(pix % W), hv = (int)((pix / W) % H);
        const long b = pix / ((long)W * H);
        const float ninf = -__builtin_huge_valf();
        f32x4 m5 = {ninf, ninf, ninf, ninf}, m9 = m5, m13 = m5;
        for (int dh = -6; dh <= 6; ++dh) {
            const int hi = hv + dh;
            if ((unsigned)hi >= (unsigned)H) continue;
            const int ah = dh < 0 ? -dh : dh;
            for (int dw = -6; dw <= 6; ++dw) {
                const int wi = wv + dw;
                if ((unsigned)wi >= (unsigned)W) continue;
                const int aw = dw < 0 ? -dw : dw;
                const int rad = ah > aw ? ah : aw;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(buf + ((b * H + hi) * W + wi) * cs + x_coff + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    m13[e] = fmaxf(m13[e], v[e]);
                    if (rad <= 4) m9[e] = fmaxf(m9[e], v[e]);
                    if (rad <= 2) m5[e] = fmaxf(m5[e], v[e]);
                }
            }
        }
        float *o = buf + pix * cs + x_coff + c;
        *reinterpret_cast<f32x4 *>(o + C) = m5;
        *reinterpret_cast<f32x4 *>(o + 2 * C) = m9;
        *reinterpret_cast<f32x4 *>(o + 3 * C) = m13;
    }
}

// ------------------------------------------------------------------------------------------------ BiFPN (+upsample)
struct BifpnArgs {
    const float *src[3];
    int up[3];
    const float *w;      // raw fusion parameter (device), normalised in the kernel: no host round trip per step
    float eps;
    int n_in;
};
__global__ __launch_bounds__(256) void bifpn_kernel(BifpnArgs a, float *__restrict__ y, int B, int H, int W, int C) {
    float wn[3];
    bifpn_norm(a.w, a.n_in, a.eps, wn);
    const int C4 = C >> 2;
    const long items = (long)B * H * W * C4;
    for (long it = blockIdx.x * 256L + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
        const int c = (int)(it % C4) * 4;
        const long pix = it / C4;
        const int wv = (int)(pix % W), hv = (int)((pix / W) % H);
        const long b = pix / ((long)W * H);
        // stack-then-sum of the reference == left-to-right fp32 sum starting from 0
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (i >= a.n_in) break;
            const int u = a.up[i];
            const long sp = (b * (H >> u) + (hv >> u)) * (W >> u) + (wv >> u);
            acc += wn[i] * *reinterpret_cast<const f32x4 *>(a.src[i] + sp * C + c);
        }
        *reinterpret_cast<f32x4 *>(y + pix * C + c) = acc;
    }
}

// ------------------------------------------------------------------------------------------------ global pooling
// stage 1: grid (nchunk, B): each workgroup reduces a chunk of pixels for all channels (lane = channel quad,
// coalesced rows); stage 2 folds the chunks.  Deterministic (no atomics).
constexpr int POOL_CHUNK = 256;   // pixels per chunk
template <bool ACT>                 // ACT: the pools of act(x) (SEAM's squeeze reads BatchNorm(GELU(u)) only through its average: that tensor is never written)
__global__ __launch_bounds__(256) void global_pool_stage1(const float *__restrict__ x, int x_cs, int x_coff, int HW, int C,
                                                          float *__restrict__ part_sum, float *__restrict__ part_max,
                                                          int nchunk, int act) {
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int C4 = C >> 2;
    const int p0 = chunk * POOL_CHUNK, p1 = min(p0 + POOL_CHUNK, HW);
    // threads: cq = tid % C4g lanes over channel quads, rows strided by 256 / C4g
    for (int cq0 = 0; cq0 < C4; cq0 += 256) {
        const int ncq = min(256, C4 - cq0);
        const int rows_par = 256 / ncq;                     // >= 1
        const int cq = threadIdx.x % ncq, rr = threadIdx.x / ncq;
        const float ninf = -__builtin_huge_valf();
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, m = {ninf, ninf, ninf, ninf};
        if (rr < rows_par) {
#pragma unroll 4
            for (int p = p0 + rr; p < p1; p += rows_par) {            // unrolled: four rows' loads in flight, sums in row order
                f32x4 v = *reinterpret_cast<const f32x4 *>(x + ((long)b * HW + p) * x_cs + x_coff + (cq0 + cq) * 4);
                if (ACT) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = apply_act_rt(v[e], act);
                }
                s += v;
#pragma unroll
                for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
            }
        }
        __shared__ f32x4 ls[256], lm[256];
        ls[threadIdx.x] = s;
        lm[threadIdx.x] = m;
        __syncthreads();
        if (threadIdx.x < ncq) {
            for (int r2 = 1; r2 < rows_par; ++r2) {
                s += ls[r2 * ncq + cq];
                const f32x4 o = lm[r2 * ncq + cq];
#pragma unroll
                for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], o[e]);
            }
            const long o = ((long)b * nchunk + chunk) * C + (cq0 + cq) * 4;
            *reinterpret_cast<f32x4 *>(part_sum + o) = s;
            *reinterpret_cast<f32x4 *>(part_max + o) = m;
        }
        __syncthreads();
    }
}
// 256 threads = 16 (sample, channel) columns x 16 chunk groups, four rows requested per trip (one thread per column walking all chunks was a
// chain of `nchunk` memory latencies); group sums combined in ascending group order
__global__ __launch_bounds__(256) void global_pool_stage2(const float *__restrict__ part_sum, const float *__restrict__ part_max,
                                                          int nchunk, int C, int B, float inv_hw, float *__restrict__ out_avg,
                                                          float *__restrict__ out_max, const float *__restrict__ post_scale,
                                                          const float *__restrict__ post_shift) {
    __shared__ double ls[256];
    __shared__ float lm[256];
    const int cl = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + cl;
    // the chunk partials are folded in double: at 1280x1280 a channel's mean is the sum of 1600 partials, and ODConv's squeeze
    // BatchNorm over a batch of two turns the DIFFERENCE of two such means into an O(1) signal (tools/layer_drift.py)
    double s = 0.0;
    float m = -__builtin_huge_valf();
    if (i < B * C) {
        const int b = i / C, c = i % C;
        const float *ps = part_sum + (long)b * nchunk * C + c, *pm = part_max + (long)b * nchunk * C + c;
        for (int k = grp; k < nchunk; k += 64) {
            float v[4], w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool ok = k + u * 16 < nchunk;
                v[u] = ok ? ps[(long)(k + u * 16) * C] : 0.f;
                w[u] = ok ? pm[(long)(k + u * 16) * C] : -__builtin_huge_valf();
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { s += (double)v[u]; m = fmaxf(m, w[u]); }
        }
    }
    ls[threadIdx.x] = s;
    lm[threadIdx.x] = m;
    __syncthreads();
    if (grp != 0 || i >= B * C) return;
#pragma unroll
    for (int g = 1; g < 16; ++g) { s += ls[g * 16 + cl]; m = fmaxf(m, lm[g * 16 + cl]); }
    float a = (float)(s * (double)inv_hw);
    if (post_scale) a = a * post_scale[i % C] + post_shift[i % C];     // the mean of an affine map of the pooled values (per-channel scale / shift)
    out_avg[i] = a;
    if (out_max) out_max[i] = m;
}

// ------------------------------------------------------------------------------------------------ affine + activation + global pools in one pass
// z = act(x * scale + shift)  AND  the per-(image, channel) sum and maximum of z (the channel attention's GAP / GMP, models/common.py:339-358)
// in the pass that writes z - global_pool_stage1's read of the tensor it would pool disappears (round 4).  Image-aligned: blockIdx.y = image,
// a thread keeps one channel quad of that image (gridDim.x * 256 is a multiple of C / 4, and C / 4 divides 256 or is a multiple of it: the host
// checks) and walks its pixels four at a time; the threads of a workgroup that share a quad are combined through LDS in a fixed order, one
// partial row per workgroup (or per group of workgroups covering all quads); global_pool_stage2 folds the rows.  SiLU after the norm only
// (the conv block's case); same arithmetic as chan_affine_act_kernel.
__global__ __launch_bounds__(256) void affine_silu_pool_kernel(const float *__restrict__ x, int x_cs, int x_coff, const float *__restrict__ scale,
                                                               const float *__restrict__ shift, float *__restrict__ z, int z_cs, int z_coff,
                                                               int HW, int C, float *__restrict__ part_sum, float *__restrict__ part_max,
                                                               int rows) {
    __shared__ f32x4 ls[256], lm[256];
    const unsigned C4 = (unsigned)C >> 2, nthreads = gridDim.x * 256u, t = blockIdx.x * 256u + threadIdx.x;
    const int c = (int)(t % C4) * 4, b = blockIdx.y, pstep = (int)(nthreads / C4);
    const f32x4 sc = *reinterpret_cast<const f32x4 *>(scale + c), sh = *reinterpret_cast<const f32x4 *>(shift + c);
    const float *xb = x + (long)b * HW * x_cs + x_coff + c;
    float *zb = z + (long)b * HW * z_cs + z_coff + c;
    const float ninf = -__builtin_huge_valf();
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, m = {ninf, ninf, ninf, ninf};
    auto one = [&](f32x4 v) {
        v = v * sc + sh;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act<SOMI_ACT_SILU>(v[e]);
        s += v;
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
        return v;
    };
    int p = (int)(t / C4);
    for (; p + 3 * pstep < HW; p += 4 * pstep) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4 *>(xb + (long)(p + u * pstep) * x_cs);
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<f32x4 *>(zb + (long)(p + u * pstep) * z_cs) = one(v[u]);
    }
    for (; p < HW; p += pstep) *reinterpret_cast<f32x4 *>(zb + (long)p * z_cs) = one(*reinterpret_cast<const f32x4 *>(xb + (long)p * x_cs));
    // combine the threads of this workgroup that hold the same quad (C4 < 256: threads tid, tid + C4, ...), ascending thread order
    int row = blockIdx.x;
    if (C4 < 256u) {
        ls[threadIdx.x] = s;
        lm[threadIdx.x] = m;
        __syncthreads();
        if (threadIdx.x >= C4) return;
        for (unsigned o = threadIdx.x + C4; o < 256u; o += C4) {
            s += ls[o];
            const f32x4 w = lm[o];
#pragma unroll
            for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], w[e]);
        }
    } else {
        row = (int)(blockIdx.x / (C4 / 256u));                         // C4 / 256 consecutive workgroups cover all quads once
    }
    const long o = ((long)b * rows + row) * C + c;
    *reinterpret_cast<f32x4 *>(part_sum + o) = s;
    *reinterpret_cast<f32x4 *>(part_max + o) = m;
}

// ------------------------------------------------------------------------------------------------ attention MLPs
// one workgroup per sample; C <= 1024, mid <= 64
__global__ __launch_bounds__(256) void attn_mlp_kernel(int mode, const float *__restrict__ avg, const float *__restrict__ mx,
                                                       const float *__restrict__ W1, const float *__restrict__ b1,
                                                       const float *__restrict__ W2, const float *__restrict__ b2,
                                                       float *__restrict__ out, int C, int mid) {
    __shared__ float h_avg[64], h_max[64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *va = avg + (long)b * C;
    const float *vm = mx ? mx + (long)b * C : nullptr;
    for (int j = wave; j < mid; j += 4) {          // one wave per hidden unit: dot over C
        float sa = 0.f, sm = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float w = W1[(long)j * C + c];
            sa += w * va[c];
            if (vm) sm += w * vm[c];
        }
        for (int o = 32; o > 0; o >>= 1) {
            sa += __shfl_down(sa, o);
            sm += __shfl_down(sm, o);
        }
        if (lane == 0) {
            const float bb = b1 ? b1[j] : 0.f;
            h_avg[j] = fmaxf(sa + bb, 0.f);
            h_max[j] = fmaxf(sm + bb, 0.f);
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float oa = b2 ? b2[c] : 0.f, om = oa;
        for (int j = 0; j < mid; ++j) {
            const float w = W2[(long)c * mid + j];
            oa += w * h_avg[j];
            om += w * h_max[j];
        }
        float r;
        if (mode == 0) r = 1.0f / (1.0f + expf(-(oa + om)));
        else r = expf(1.0f / (1.0f + expf(-oa)));
        out[(long)b * C + c] = r;
    }
}

// ------------------------------------------------------------------------------------------------ CBAM spatial stats
// one wave per pixel: lanes stride the channels (coalesced), wave-reduce mean and max of x*ca
__global__ __launch_bounds__(256) void chan_stats_kernel(const float *__restrict__ x, int x_cs, int x_coff,
                                                         const float *__restrict__ ca, float *__restrict__ stats, int B,
                                                         int HW, int C) {
    int LG = 64;                                                     // lanes per pixel: the smallest power of two covering C / 4 channel quads
    while (LG > 1 && (LG >> 1) * 4 >= C) LG >>= 1;
    const int lane = threadIdx.x & 63, sub = lane / LG, sl = lane % LG, ppw = 64 / LG;
    const long wave_id = (blockIdx.x * 256L + threadIdx.x) >> 6, nwave = (long)gridDim.x * 4;
    const long npix = (long)B * HW;
    const float inv_c = 1.0f / (float)C;
    for (long p = wave_id * ppw + sub; p < npix; p += nwave * ppw) {
        const long b = p / HW;
        const float *xr = x + p * x_cs + x_coff;
        const float *cr = ca + b * C;
        float s = 0.f, m = -__builtin_huge_valf();
        for (int c = sl * 4; c < C; c += LG * 4) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(xr + c) * *reinterpret_cast<const f32x4 *>(cr + c);
            s += (v[0] + v[1]) + (v[2] + v[3]);
            m = fmaxf(fmaxf(fmaxf(m, v[0]), fmaxf(v[1], v[2])), v[3]);
        }
        for (int o = LG >> 1; o > 0; o >>= 1) {
            s += __shfl_xor(s, o);
            m = fmaxf(m, __shfl_xor(m, o));
        }
        if (sl == 0) {
            stats[p * 2] = s * inv_c;
            stats[p * 2 + 1] = m;
        }
    }
}

__global__ __launch_bounds__(256) void spatial_attn_kernel(const float *__restrict__ stats, const float *__restrict__ w,
                                                           const float *__restrict__ bias_p, float *__restrict__ sa, int B, int H, int W, int k) {
    const float bias = bias_p[0];
    const long npix = (long)B * H * W;
    const int pad = k >> 1;
    for (long p = blockIdx.x * 256L + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
        const int wv = (int)(p % W), hv = (int)((p / W) % H);
        const long b = p / ((long)W * H);
        float acc = bias;
        for (int r = 0; r < k; ++r) {
            const int hi = hv + r - pad;
            if ((unsigned)hi >= (unsigned)H) continue;
            for (int q = 0; q < k; ++q) {
                const int wi = wv + q - pad;
                if ((unsigned)wi >= (unsigned)W) continue;
                const float2 st = *reinterpret_cast<const float2 *>(stats + ((b * H + hi) * W + wi) * 2);
                acc += st.x * w[(r * k + q) * 2] + st.y * w[(r * k + q) * 2 + 1];
            }
        }
        sa[p] = 1.0f / (1.0f + expf(-acc));
    }
}

// CBAM apply: sa = sigmoid(conv_kxk(stats) + bias) for a tile of 64 pixels (one lane each), then y = x * ca[b][c] * sa[p]
// for the tile's 64 x C elements (float4 per lane).  In place allowed.  Replaces materialising `spatial_attention(out) * out`
// with `out = channel_attention(x2) * x2` (models/common.py:686-688) in two passes.
__global__ __launch_bounds__(256) void cbam_apply_kernel(const float *__restrict__ x, int x_cs, int x_coff, const float *__restrict__ ca,
                                                         const float *__restrict__ stats, const float *__restrict__ w, const float *__restrict__ bias_p,
                                                         float *__restrict__ y, int y_cs, int y_coff, int B, int H, int W, int C, int k) {
    const float bias = bias_p[0];
    __shared__ float sa[64];
    const long npix = (long)B * H * W;
    const int pad = k >> 1, C4 = C >> 2;
    for (long p0 = blockIdx.x * 64L; p0 < npix; p0 += (long)gridDim.x * 64) {
        __syncthreads();
        if (threadIdx.x < 64) {
            const long p = p0 + threadIdx.x;
            float acc = bias;
            if (p < npix) {
                const int wv = (int)(p % W), hv = (int)((p / W) % H);
                const long b = p / ((long)W * H);
                for (int r = 0; r < k; ++r) {
                    const int hi = hv + r - pad;
                    if ((unsigned)hi >= (unsigned)H) continue;
                    for (int q = 0; q < k; ++q) {
                        const int wi = wv + q - pad;
                        if ((unsigned)wi >= (unsigned)W) continue;
                        const float2 st = *reinterpret_cast<const float2 *>(stats + ((b * H + hi) * W + wi) * 2);
                        acc += st.x * w[(r * k + q) * 2] + st.y * w[(r * k + q) * 2 + 1];
                    }
                }
            }
            sa[threadIdx.x] = 1.0f / (1.0f + expf(-acc));
        }
        __syncthreads();
        const int np = (int)min(64L, npix - p0);
        for (int it = threadIdx.x; it < np * C4; it += 256) {
            const int pl = it / C4, c = (it % C4) * 4;
            const long p = p0 + pl, b = p / ((long)W * H);
            f32x4 v = *reinterpret_cast<const f32x4 *>(x + p * x_cs + x_coff + c);
            v = v * *reinterpret_cast<const f32x4 *>(ca + b * C + c) * sa[pl];
            *reinterpret_cast<f32x4 *>(y + p * y_cs + y_coff + c) = v;
        }
    }
}

__global__ __launch_bounds__(256) void scale_channels_kernel(const float *__restrict__ x, const float *__restrict__ s,
                                                             const float *__restrict__ pix, float *__restrict__ y, int B,
                                                             int HW, int C) {
    const int C4 = C >> 2;
    const long items = (long)B * HW * C4;
    for (long it = blockIdx.x * 256L + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
        const int c = (int)(it % C4) * 4;
        const long p = it / C4, b = p / HW;
        f32x4 v = *reinterpret_cast<const f32x4 *>(x + p * C + c);
        if (s) v *= *reinterpret_cast<const f32x4 *>(s + b * C + c);
        if (pix) v *= pix[p];
        *reinterpret_cast<f32x4 *>(y + p * C + c) = v;
    }
}

// ------------------------------------------------------------------------------------------------ ODConv
// stage A: one workgroup per sample computes z and the four attention vectors into the workspace
__global__ __launch_bounds__(256) void odconv_attn_kernel(const float *__restrict__ gap, const float *__restrict__ fc_w,
                                                          const float *__restrict__ fc_b, const float *__restrict__ Wf,
                                                          const float *__restrict__ bf, const float *__restrict__ Ws,
                                                          const float *__restrict__ bs, const float *__restrict__ Wc,
                                                          const float *__restrict__ bc, const float *__restrict__ Ww,
                                                          const float *__restrict__ bw, float *__restrict__ ws, int Cin,
                                                          int Cout, int kk, int K, int hid) {
    __shared__ float z[64];
    __shared__ float lw[16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *g = gap + (long)b * Cin;
    for (int j = wave; j < hid; j += 4) {
        float s = 0.f;
        for (int c = lane; c < Cin; c += 64) s += fc_w[(long)j * Cin + c] * g[c];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
        if (lane == 0) z[j] = fmaxf(s + (fc_b ? fc_b[j] : 0.f), 0.f);
    }
    __syncthreads();
    float *o = ws + (long)b * (Cout + kk + Cin + K);
    auto head = [&](const float *Wm, const float *bm, int n, float *dst, bool sig) {
        for (int i = tid; i < n; i += 256) {
            float s = bm[i];
            for (int j = 0; j < hid; ++j) s += Wm[(long)i * hid + j] * z[j];
            dst[i] = sig ? 1.0f / (1.0f + expf(-s)) : s;
        }
    };
    head(Wf, bf, Cout, o, true);
    head(Ws, bs, kk, o + Cout, true);
    head(Wc, bc, Cin, o + Cout + kk, true);
    if (tid < K) {
        float s = bw[tid];
        for (int j = 0; j < hid; ++j) s += Ww[(long)tid * hid + j] * z[j];
        lw[tid] = s;
    }
    __syncthreads();
    if (tid < K) {                       // softmax over K (K <= 16)
        float mx = lw[0];
        for (int i = 1; i < K; ++i) mx = fmaxf(mx, lw[i]);
        float den = 0.f;
        for (int i = 0; i < K; ++i) den += expf(lw[i] - mx);
        o[Cout + kk + Cin + tid] = expf(lw[tid] - mx) / den;
    }
}
// stage B: wout[b][n][t][c]; one thread = 4 consecutive c; BN of the following layer folded in
__global__ __launch_bounds__(256) void odconv_synth_kernel(const float *__restrict__ ws, const float *__restrict__ Wk,
                                                           const float *__restrict__ biask, const float *__restrict__ bn_s,
                                                           const float *__restrict__ bn_t, float *__restrict__ wout,
                                                           float *__restrict__ bout, int B, int Cin, int Cin_pad, int Cout,
                                                           int kk, int K) {
    const int C4 = Cin_pad >> 2;
    const long per_b = (long)Cout * kk * C4;
    const long items = (long)B * per_b;
    const long set = (long)Cout * kk * Cin_pad;
    for (long it = blockIdx.x * 256L + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
        const long b = it / per_b, rem = it % per_b;
        const int c = (int)(rem % C4) * 4;
        const int t = (int)((rem / C4) % kk);
        const int n = (int)(rem / ((long)C4 * kk));
        const float *at = ws + b * (Cout + kk + Cin + K);
        const float *aw = at + Cout + kk + Cin;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        // the reference sums attn*W over K with the full product attn = a_f*a_s*a_c*a_w per term (common.py:4570-4580)
        const float fs = at[n] * at[Cout + t];
        f32x4 ac = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) ac[e] = (c + e < Cin) ? at[Cout + kk + c + e] : 0.f;
        for (int q = 0; q < K; ++q) {
            const f32x4 wv = *reinterpret_cast<const f32x4 *>(Wk + q * set + ((long)n * kk + t) * Cin_pad + c);
            acc += ((fs * ac) * aw[q]) * wv;
        }
        const float sc = bn_s ? bn_s[n] : 1.f;
        *reinterpret_cast<f32x4 *>(wout + b * set + ((long)n * kk + t) * Cin_pad + c) = acc * sc;
        if (t == 0 && c == 0) {
            float bv = 0.f;
            if (biask)
                for (int q = 0; q < K; ++q) bv += aw[q] * biask[q * Cout + n];
            bout[b * Cout + n] = bn_s ? bv * sc + bn_t[n] : bv;
        }
    }
}

// ------------------------------------------------------------------------------------------------ detect decode
struct DecodeArgs {
    float anchor_px[16];   // na*2, anchors * stride
};
__global__ __launch_bounds__(256) void detect_decode_kernel(const float *__restrict__ box, int box_cs,
                                                            const float *__restrict__ cls, int cls_cs, DecodeArgs da,
                                                            float stride, float *__restrict__ raw, float *__restrict__ z, int B,
                                                            int ny, int nx, int na, int nc, int total, int row_off) {
    const int no = nc + 5;
    const long items = (long)B * na * ny * nx * no;
    for (long it = blockIdx.x * 256L + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
        const int o = (int)(it % no);
        long r = it / no;
        const int xg = (int)(r % nx);
        r /= nx;
        const int yg = (int)(r % ny);
        r /= ny;
        const int an = (int)(r % na);
        const long b = r / na;
        const long pix = (b * ny + yg) * nx + xg;
        const float v = o < 5 ? box[pix * box_cs + an * 5 + o] : cls[pix * cls_cs + an * nc + (o - 5)];
        if (raw) raw[it] = v;
        if (z) {
            const float s = 1.0f / (1.0f + expf(-v));
            float out = s;
            if (o == 0) out = (s * 2.0f + ((float)xg - 0.5f)) * stride;
            else if (o == 1) out = (s * 2.0f + ((float)yg - 0.5f)) * stride;
            else if (o == 2) out = (s * 2.0f) * (s * 2.0f) * da.anchor_px[an * 2];
            else if (o == 3) out = (s * 2.0f) * (s * 2.0f) * da.anchor_px[an * 2 + 1];
            const long row = row_off + ((long)an * ny + yg) * nx + xg;
            z[(b * total + row) * no + o] = out;
        }
    }
}

// ------------------------------------------------------------------------------------------------ DCNv3 module helpers
// LayerNorm over C (eps inside the sqrt, biased variance) followed by an activation: one wave per pixel, two passes over a
// row that stays in L1 (models/ops_dcnv3/modules/dcnv3.py:283-291: LayerNorm(eps=1e-6) -> GELU).
__global__ __launch_bounds__(256) void layernorm_act_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, float eps, int act,
                                                            float *__restrict__ y, long npix, int C) {
    const int lane = threadIdx.x & 63;
    const long wave_id = (blockIdx.x * 256L + threadIdx.x) >> 6, nwave = (long)gridDim.x * 4;
    for (long p = wave_id; p < npix; p += nwave) {
        const float *xr = x + p * C;
        float s = 0.f;
        for (int c = lane * 4; c < C; c += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(xr + c);
            s += (v[0] + v[1]) + (v[2] + v[3]);
        }
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s / (float)C;
        float q = 0.f;
        for (int c = lane * 4; c < C; c += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(xr + c) - mean;
            q += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        }
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
        const float rstd = rsqrtf(q / (float)C + eps);
        for (int c = lane * 4; c < C; c += 256) {
            f32x4 v = (*reinterpret_cast<const f32x4 *>(xr + c) - mean) * rstd;
            v = v * *reinterpret_cast<const f32x4 *>(gamma + c) + *reinterpret_cast<const f32x4 *>(beta + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = apply_act_rt(v[e], act);
            *reinterpret_cast<f32x4 *>(y + p * C + c) = v;
        }
    }
}

// softmax over the K sampling points of every (pixel, group) (modules/dcnv3.py:334); one lane per (pixel, group).  Group g of pixel p
// starts at p*ps + g*K: packed rows (ps = G*K) or a column range of wider rows; x == y (in place) is allowed, hence no __restrict__.
__global__ __launch_bounds__(256) void group_softmax_kernel(const float *x, float *y, long n, int G, int K, long x_ps, long y_ps) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long p = i / G;
        const int g = (int)(i - p * G);
        const float *r = x + p * x_ps + g * K;
        float *o = y + p * y_ps + g * K;
        float m = r[0];
        for (int k = 1; k < K; ++k) m = fmaxf(m, r[k]);
        float den = 0.f;
        for (int k = 0; k < K; ++k) den += expf(r[k] - m);
        const float inv = 1.f / den;
        for (int k = 0; k < K; ++k) o[k] = expf(r[k] - m) * inv;
    }
}

// The same softmax for rows whose G*K floats are 16-byte aligned float4 runs: a workgroup stages SM_P pixels' rows in LDS with coalesced
// 16-byte loads (a lane per group reading K scalars with a K-float stride costs K cache transactions per line), every lane then owns one
// (pixel, group) in LDS (stride K words: conflict-free for odd K), and the rows go back as float4.  Arithmetic identical to the kernel above.
constexpr int SM_P = 32;
__global__ __launch_bounds__(256) void group_softmax_tile_kernel(const float *x, float *y, long npix, int G, int K, long x_ps, long y_ps) {
    extern __shared__ float sm_rows[];
    const int GK = G * K, Q = GK >> 2;
    for (long p0 = (long)blockIdx.x * SM_P; p0 < npix; p0 += (long)gridDim.x * SM_P) {
        const int np = (int)min((long)SM_P, npix - p0);
        for (int i = threadIdx.x; i < np * Q; i += 256) {
            const int pl = i / Q, q = i - pl * Q;
            reinterpret_cast<f32x4 *>(sm_rows)[i] = *reinterpret_cast<const f32x4 *>(x + (p0 + pl) * x_ps + q * 4);
        }
        __syncthreads();
        for (int gi = threadIdx.x; gi < np * G; gi += 256) {
            float *r = sm_rows + gi * K;
            float m = r[0];
            for (int k = 1; k < K; ++k) m = fmaxf(m, r[k]);
            float den = 0.f;
            for (int k = 0; k < K; ++k) den += expf(r[k] - m);
            const float inv = 1.f / den;
            for (int k = 0; k < K; ++k) r[k] = expf(r[k] - m) * inv;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < np * Q; i += 256) {
            const int pl = i / Q, q = i - pl * Q;
            *reinterpret_cast<f32x4 *>(y + (p0 + pl) * y_ps + q * 4) = reinterpret_cast<const f32x4 *>(sm_rows)[i];
        }
        __syncthreads();
    }
}

static void launch_group_softmax(const float *x, float *y, long npix, int G, int K, long x_ps, long y_ps, hipStream_t s) {
    const size_t lds = (size_t)SM_P * G * K * sizeof(float);
    if ((G * K) % 4 == 0 && x_ps % 4 == 0 && y_ps % 4 == 0 && aligned16(x) && aligned16(y) && lds <= 48 * 1024) {
        long g = (npix + SM_P - 1) / SM_P;
        hipLaunchKernelGGL(group_softmax_tile_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), lds, s, x, y, npix, G, K, x_ps, y_ps);
    } else {
        hipLaunchKernelGGL(group_softmax_kernel, dim3(ew_grid(npix * G)), dim3(256), 0, s, x, y, npix * G, G, K, x_ps, y_ps);
    }
}

// centre-feature-scale blend (modules/dcnv3.py:370-376): s = sigmoid(logit[p][g]); y = x*(1-s) + xproj*s
__global__ __launch_bounds__(256) void cfs_blend_kernel(const float *__restrict__ x, const float *__restrict__ xproj,
                                                        const float *__restrict__ logit, int logit_cs, float *__restrict__ y,
                                                        long npix, int G, int Gc) {
    const int C = G * Gc;
    const long items = npix * C;
    for (long it = blockIdx.x * 256L + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
        const long p = it / C;
        const int g = (int)(it % C) / Gc;
        const float sg = 1.0f / (1.0f + expf(-logit[p * logit_cs + g]));
        y[it] = x[it] * (1.f - sg) + xproj[it] * sg;
    }
}

}  // namespace somi

using namespace somi;

extern "C" int somi_layernorm_act_nhwc_f32(const float *x, const float *gamma, const float *beta, float eps, int act, float *y,
                                           long npix, int C, somi_stream_t stream) {
    SOMI_REQUIRE(x && gamma && beta && y && npix > 0 && C > 0 && C % 4 == 0 && aligned16(x) && aligned16(y) && aligned16(gamma) &&
                     aligned16(beta), SOMI_EINVAL, "layernorm: bad arguments (C %% 4, 16 B alignment)");
    hipLaunchKernelGGL(layernorm_act_kernel, dim3(ew_grid(npix * 64)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, eps, act, y,
                       npix, C);
    return launch_status("somi_layernorm_act_nhwc_f32");
}

extern "C" int somi_group_softmax_f32(const float *x, float *y, long n_groups, int K, somi_stream_t stream) {
    SOMI_REQUIRE(x && y && n_groups > 0 && K > 0, SOMI_EINVAL, "group softmax: bad arguments");
    hipLaunchKernelGGL(group_softmax_kernel, dim3(ew_grid(n_groups)), dim3(256), 0, (hipStream_t)stream, x, y, n_groups, 1, K, (long)K, (long)K);
    return launch_status("somi_group_softmax_f32");
}

extern "C" int somi_group_softmax_strided_f32(const float *x, long x_stride, float *y, long y_stride, long npix, int G, int K,
                                              somi_stream_t stream) {
    SOMI_REQUIRE(x && y && npix > 0 && G > 0 && K > 0 && x_stride >= (long)G * K && y_stride >= (long)G * K, SOMI_EINVAL,
                 "group softmax (strided): bad arguments");
    launch_group_softmax(x, y, npix, G, K, x_stride, y_stride, (hipStream_t)stream);
    return launch_status("somi_group_softmax_strided_f32");
}

extern "C" int somi_dcnv3_cfs_blend_f32(const float *x, const float *xproj, const float *logit, int logit_cs, float *y, long npix,
                                        int G, int Gc, somi_stream_t stream) {
    SOMI_REQUIRE(x && xproj && logit && y && npix > 0 && G > 0 && Gc > 0 && logit_cs >= G, SOMI_EINVAL, "cfs blend: bad arguments");
    hipLaunchKernelGGL(cfs_blend_kernel, dim3(ew_grid(npix * G * Gc)), dim3(256), 0, (hipStream_t)stream, x, xproj, logit, logit_cs, y,
                       npix, G, Gc);
    return launch_status("somi_dcnv3_cfs_blend_f32");
}

extern "C" int somi_image_u8_to_nhwc4(const uint8_t *img, float *y, int B, int C, int H, int W, somi_stream_t stream) {
    SOMI_REQUIRE(img && y && B > 0 && H > 0 && W > 0 && C >= 1 && C <= 4, SOMI_EINVAL, "image ingest: bad arguments");
    SOMI_REQUIRE(aligned16(y), SOMI_EINVAL, "image ingest: output must be 16 B aligned");
    hipLaunchKernelGGL((image_to_nhwc4_kernel<uint8_t, true>), dim3(ew_grid((long)B * H * W)), dim3(256), 0, (hipStream_t)stream,
                       img, y, B, C, H, W, 255.0f);
    return launch_status("somi_image_u8_to_nhwc4");
}

extern "C" int somi_image_f32_to_nhwc4(const float *img, float *y, int B, int C, int H, int W, float scale,
                                       somi_stream_t stream) {
    SOMI_REQUIRE(img && y && B > 0 && H > 0 && W > 0 && C >= 1 && C <= 4, SOMI_EINVAL, "image ingest: bad arguments");
    SOMI_REQUIRE(aligned16(y), SOMI_EINVAL, "image ingest: output must be 16 B aligned");
    hipLaunchKernelGGL((image_to_nhwc4_kernel<float, false>), dim3(ew_grid((long)B * H * W)), dim3(256), 0, (hipStream_t)stream,
                       img, y, B, C, H, W, scale);
    return launch_status("somi_image_f32_to_nhwc4");
}

extern "C" int somi_dwconv3x3_nhwc_f32(const float *x, const float *w, const float *bias, const float *post_scale,
                                       const float *post_shift, const float *residual, float *y, int B, int H, int W, int C,
                                       int act, somi_stream_t stream) {
    SOMI_REQUIRE(x && w && y && B > 0 && H > 0 && W > 0 && C > 0, SOMI_EINVAL, "dwconv: bad arguments");
    SOMI_REQUIRE(C % 4 == 0 && aligned16(x) && aligned16(w) && aligned16(y), SOMI_EINVAL, "dwconv: C %% 4 and 16 B alignment");
    SOMI_REQUIRE(!post_scale == !post_shift, SOMI_EINVAL, "dwconv: post_scale and post_shift go together");
    somi::launch_dwconv3x3(false, x, w, bias, post_scale, post_shift, residual, y, B, H, W, C, act, (hipStream_t)stream);
    return launch_status("somi_dwconv3x3_nhwc_f32");
}

extern "C" int somi_dwconv3x3_ln_nhwc_f32(const float *x, const float *w, const float *bias, const float *gamma, const float *beta, float eps,
                                          int act, float *u, float *y, int B, int H, int W, int C, somi_stream_t stream) {
    SOMI_REQUIRE(x && w && gamma && beta && u && y && B > 0 && H > 0 && W > 0, SOMI_EINVAL, "dwconv + layernorm: bad arguments");
    SOMI_REQUIRE(C == 256, SOMI_ENOTIMPL, "dwconv + layernorm: 256 channels only (one wave per pixel); run the two passes for other widths");
    SOMI_REQUIRE(aligned16(x) && aligned16(w) && aligned16(u) && aligned16(y) && aligned16(gamma) && aligned16(beta) && (!bias || aligned16(bias)),
                 SOMI_EINVAL, "dwconv + layernorm: 16 B alignment");
    const int seg = H >= 64 ? 16 : 8;
    hipLaunchKernelGGL(dwconv3x3_ln_kernel, dim3(ew_grid((long)B * cdiv(H, seg) * W * 64)), dim3(256), 0, (hipStream_t)stream, x, w, bias, gamma, beta,
                       eps, act, u, y, B, H, W, seg);
    return launch_status("somi_dwconv3x3_ln_nhwc_f32");
}

extern "C" int somi_sppf_pool_nhwc_f32(float *buf, int B, int H, int W, int C, int cs, int x_coff, somi_stream_t stream) {
    SOMI_REQUIRE(buf && B > 0 && H > 0 && W > 0 && C > 0, SOMI_EINVAL, "sppf: bad arguments");
    SOMI_REQUIRE(C % 4 == 0 && cs % 4 == 0 && x_coff % 4 == 0 && x_coff + 4 * C <= cs && aligned16(buf), SOMI_EINVAL,
                 "sppf: needs C,cs,x_coff %% 4 == 0 and room for 4*C channels");
    hipLaunchKernelGGL(sppf_pool_kernel, dim3(ew_grid((long)B * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, buf, B,
                       H, W, C, cs, x_coff);
    return launch_status("somi_sppf_pool_nhwc_f32");
}

extern "C" int somi_bifpn_nhwc_f32(const float *const *src_host, const int *up_host, const float *w_dev, float eps, int n_in, float *y,
                                   int B, int H, int W, int C, somi_stream_t stream) {
    SOMI_REQUIRE(src_host && up_host && w_dev && y && (n_in == 2 || n_in == 3), SOMI_EINVAL, "bifpn: bad arguments");
    SOMI_REQUIRE(C % 4 == 0 && aligned16(y), SOMI_EINVAL, "bifpn: C %% 4 and alignment");
    BifpnArgs a;
    for (int i = 0; i < 3; ++i) {
        a.src[i] = i < n_in ? src_host[i] : nullptr;
        a.up[i] = i < n_in ? up_host[i] : 0;
        if (i < n_in) {
            SOMI_REQUIRE(a.src[i] && aligned16(a.src[i]) && (a.up[i] == 0 || a.up[i] == 1), SOMI_EINVAL, "bifpn: bad source %d", i);
            SOMI_REQUIRE(a.up[i] == 0 || (H % 2 == 0 && W % 2 == 0), SOMI_EINVAL, "bifpn: upsampled source needs even H, W");
        }
    }
    a.n_in = n_in;
    a.w = w_dev;
    a.eps = eps;
    hipLaunchKernelGGL(bifpn_kernel, dim3(ew_grid((long)B * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, a, y, B, H, W, C);
    return launch_status("somi_bifpn_nhwc_f32");
}

extern "C" int somi_pool_nchunk(int HW) { return (HW + POOL_CHUNK - 1) / POOL_CHUNK; }

extern "C" int somi_global_pool_nhwc_f32(const float *x, int x_cs, int x_coff, int B, int HW, int C, float *out_avg,
                                         float *out_max, float *workspace, somi_stream_t stream) {
    SOMI_REQUIRE(x && out_avg && workspace && B > 0 && HW > 0 && C > 0, SOMI_EINVAL, "global pool: bad arguments");
    SOMI_REQUIRE(C % 4 == 0 && x_cs % 4 == 0 && x_coff % 4 == 0 && x_coff + C <= x_cs && aligned16(x) && aligned16(workspace),
                 SOMI_EINVAL, "global pool: C,x_cs,x_coff %% 4 and alignment");
    const int nchunk = somi_pool_nchunk(HW);
    float *ps = workspace, *pm = workspace + (size_t)B * nchunk * C;
    hipLaunchKernelGGL(global_pool_stage1<false>, dim3(nchunk, B), dim3(256), 0, (hipStream_t)stream, x, x_cs, x_coff, HW, C, ps, pm, nchunk, 0);
    hipLaunchKernelGGL(global_pool_stage2, dim3(cdiv((long)B * C, 16)), dim3(256), 0, (hipStream_t)stream, ps, pm, nchunk, C, B,
                       1.0f / (float)HW, out_avg, out_max, nullptr, nullptr);
    return launch_status("somi_global_pool_nhwc_f32");
}

extern "C" int somi_global_pool_act_nhwc_f32(const float *x, int x_cs, int x_coff, int B, int HW, int C, int act, const float *post_scale,
                                             const float *post_shift, float *out_avg, float *workspace, somi_stream_t stream) {
    SOMI_REQUIRE(x && out_avg && workspace && B > 0 && HW > 0 && C > 0 && !post_scale == !post_shift && act >= 0 && act <= SOMI_ACT_SIGMOID,
                 SOMI_EINVAL, "global pool (act): bad arguments");
    SOMI_REQUIRE(C % 4 == 0 && x_cs % 4 == 0 && x_coff % 4 == 0 && x_coff + C <= x_cs && aligned16(x) && aligned16(workspace),
                 SOMI_EINVAL, "global pool (act): C,x_cs,x_coff %% 4 and alignment");
    const int nchunk = somi_pool_nchunk(HW);
    float *ps = workspace, *pm = workspace + (size_t)B * nchunk * C;
    hipLaunchKernelGGL(global_pool_stage1<true>, dim3(nchunk, B), dim3(256), 0, (hipStream_t)stream, x, x_cs, x_coff, HW, C, ps, pm, nchunk, act);
    hipLaunchKernelGGL(global_pool_stage2, dim3(cdiv((long)B * C, 16)), dim3(256), 0, (hipStream_t)stream, ps, pm, nchunk, C, B,
                       1.0f / (float)HW, out_avg, nullptr, post_scale, post_shift);
    return launch_status("somi_global_pool_act_nhwc_f32");
}

// workgroups per image of affine_silu_pool_kernel (0: the shape is not covered - C / 4 must divide 256 or be a multiple of it) and its partial rows
static inline int asp_grid(int B, int HW, int C, int *rows) {
    const int C4 = C / 4;
    if (C % 4 || C4 <= 0 || !((256 % C4 == 0) || (C4 % 256 == 0))) return 0;
    const int mult = C4 <= 256 ? 1 : C4 / 256;                       // workgroups that cover every quad once
    long g = ((long)HW * C4 + 255) / 256, per = (5 * 256) / (B > 0 ? B : 1);   // ~ one resident round of the chip over the whole batch
    if (per < 1) per = 1;
    g = g < 1 ? 1 : (g > per ? per : g);
    g = (g + mult - 1) / mult * mult;
    *rows = (int)(g / mult);
    return (int)g;
}
extern "C" int somi_affine_silu_pool_rows(int B, int HW, int C) {
    int rows = 0;
    return asp_grid(B, HW, C, &rows) ? rows : 0;
}
extern "C" int somi_affine_silu_pool_nhwc_f32(const float *x, int x_cs, int x_coff, const float *scale, const float *shift, float *z, int z_cs,
                                              int z_coff, int B, int HW, int C, float *out_avg, float *out_max, float *workspace,
                                              somi_stream_t stream) {
    int rows = 0;
    const int gx = asp_grid(B, HW, C, &rows);
    SOMI_REQUIRE(x && z && scale && shift && out_avg && out_max && workspace && B > 0 && B <= 65535 && HW > 0 && gx > 0, SOMI_EINVAL,
                 "affine + silu + pool: bad arguments (C / 4 must divide 256 or be a multiple of it)");
    SOMI_REQUIRE(x_cs % 4 == 0 && x_coff % 4 == 0 && z_cs % 4 == 0 && z_coff % 4 == 0 && x_coff + C <= x_cs && z_coff + C <= z_cs && aligned16(x) &&
                     aligned16(z) && aligned16(scale) && aligned16(shift) && aligned16(workspace), SOMI_EINVAL,
                 "affine + silu + pool: strides / offsets %% 4 and 16 B alignment");
    float *ps = workspace, *pm = workspace + (size_t)B * rows * C;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(affine_silu_pool_kernel, dim3(gx, B), dim3(256), 0, s, x, x_cs, x_coff, scale, shift, z, z_cs, z_coff, HW, C, ps, pm, rows);
    hipLaunchKernelGGL(global_pool_stage2, dim3(cdiv((long)B * C, 16)), dim3(256), 0, s, ps, pm, rows, C, B, 1.0f / (float)HW, out_avg, out_max, nullptr, nullptr);
    return launch_status("somi_affine_silu_pool_nhwc_f32");
}

extern "C" int somi_attn_mlp_f32(int mode, const float *avg, const float *mx, const float *W1, const float *b1,
                                 const float *W2, const float *b2, float *out, int B, int C, int mid, somi_stream_t stream) {
    SOMI_REQUIRE(avg && W1 && W2 && out && B > 0 && C > 0 && mid > 0 && mid <= 64, SOMI_EINVAL, "attn mlp: bad arguments (mid <= 64)");
    SOMI_REQUIRE(mode == 0 || mode == 1, SOMI_EINVAL, "attn mlp: mode must be 0 or 1");
    SOMI_REQUIRE(mode == 1 || mx, SOMI_EINVAL, "attn mlp: mode 0 needs the max-pooled vector");
    hipLaunchKernelGGL(attn_mlp_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, mode, avg, mode == 0 ? mx : nullptr, W1, b1,
                       W2, b2, out, C, mid);
    return launch_status("somi_attn_mlp_f32");
}

extern "C" int somi_chan_stats_nhwc_f32(const float *x, int x_cs, int x_coff, const float *ca, float *stats, int B, int HW,
                                        int C, somi_stream_t stream) {
    SOMI_REQUIRE(x && ca && stats && B > 0 && HW > 0 && C > 0, SOMI_EINVAL, "chan stats: bad arguments");
    SOMI_REQUIRE(C % 4 == 0 && x_cs % 4 == 0 && x_coff % 4 == 0 && aligned16(x) && aligned16(ca), SOMI_EINVAL,
                 "chan stats: C,x_cs,x_coff %% 4 and alignment");
    hipLaunchKernelGGL(chan_stats_kernel, dim3(ew_grid((long)B * HW * 64)), dim3(256), 0, (hipStream_t)stream, x, x_cs, x_coff, ca,
                       stats, B, HW, C);
    return launch_status("somi_chan_stats_nhwc_f32");
}

extern "C" int somi_spatial_attn_f32(const float *stats, const float *w, const float *bias, float *sa, int B, int H, int W, int k,
                                     somi_stream_t stream) {
    SOMI_REQUIRE(stats && w && bias && sa && B > 0 && H > 0 && W > 0 && (k == 3 || k == 5 || k == 7), SOMI_EINVAL,
                 "spatial attn: bad arguments (k in 3,5,7)");
    hipLaunchKernelGGL(spatial_attn_kernel, dim3(ew_grid((long)B * H * W)), dim3(256), 0, (hipStream_t)stream, stats, w, bias, sa, B,
                       H, W, k);
    return launch_status("somi_spatial_attn_f32");
}

extern "C" int somi_cbam_apply_nhwc_f32(const float *x, int x_cs, int x_coff, const float *ca, const float *stats, const float *w,
                                        const float *bias, float *y, int y_cs, int y_coff, int B, int H, int W, int C, int k,
                                        somi_stream_t stream) {
    SOMI_REQUIRE(x && ca && stats && w && bias && y && B > 0 && H > 0 && W > 0 && C > 0 && (k == 3 || k == 5 || k == 7), SOMI_EINVAL,
                 "cbam apply: bad arguments (k in 3,5,7)");
    SOMI_REQUIRE(C % 4 == 0 && x_cs % 4 == 0 && x_coff % 4 == 0 && y_cs % 4 == 0 && y_coff % 4 == 0 && aligned16(x) && aligned16(y) &&
                     aligned16(ca), SOMI_EINVAL, "cbam apply: C, strides, offsets %% 4 and 16 B alignment");
    const long tiles = ((long)B * H * W + 63) / 64;
    hipLaunchKernelGGL(cbam_apply_kernel, dim3((unsigned)(tiles > 256L * 16 ? 256L * 16 : tiles)), dim3(256), 0, (hipStream_t)stream, x,
                       x_cs, x_coff, ca, stats, w, bias, y, y_cs, y_coff, B, H, W, C, k);
    return launch_status("somi_cbam_apply_nhwc_f32");
}

extern "C" int somi_scale_channels_nhwc_f32(const float *x, const float *s, const float *pix, float *y, int B, int HW, int C,
                                            somi_stream_t stream) {
    SOMI_REQUIRE(x && y && B > 0 && HW > 0 && C > 0 && C % 4 == 0 && aligned16(x) && aligned16(y), SOMI_EINVAL,
                 "scale channels: bad arguments");
    hipLaunchKernelGGL(scale_channels_kernel, dim3(ew_grid((long)B * HW * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, s, pix, y,
                       B, HW, C);
    return launch_status("somi_scale_channels_nhwc_f32");
}

extern "C" int somi_odconv_weights_f32(const float *gap, const float *fc_w, const float *fc_b, const float *Wf, const float *bf,
                                       const float *Ws, const float *bs, const float *Wc, const float *bc, const float *Ww,
                                       const float *bw, const float *Wk, const float *biask, const float *bn_scale,
                                       const float *bn_shift, float *wout, float *bout, float *workspace, int B, int Cin,
                                       int Cin_pad, int Cout, int kk, int K, int hid, somi_stream_t stream) {
    SOMI_REQUIRE(gap && fc_w && Wf && bf && Ws && bs && Wc && bc && Ww && bw && Wk && wout && bout && workspace, SOMI_EINVAL,
                 "odconv: null tensor");
    SOMI_REQUIRE(B > 0 && Cin > 0 && Cin_pad >= Cin && Cin_pad % 4 == 0 && Cout > 0 && kk > 0 && K > 0 && K <= 16 && hid > 0 && hid <= 64,
                 SOMI_EINVAL, "odconv: bad sizes (K <= 16, hid <= 64, Cin_pad %% 4 == 0)");
    SOMI_REQUIRE(!bn_scale == !bn_shift, SOMI_EINVAL, "odconv: bn_scale and bn_shift go together");
    SOMI_REQUIRE(aligned16(Wk) && aligned16(wout), SOMI_EINVAL, "odconv: Wk / wout must be 16 B aligned");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(odconv_attn_kernel, dim3(B), dim3(256), 0, s, gap, fc_w, fc_b, Wf, bf, Ws, bs, Wc, bc, Ww, bw, workspace, Cin,
                       Cout, kk, K, hid);
    hipLaunchKernelGGL(odconv_synth_kernel, dim3(ew_grid((long)B * Cout * kk * (Cin_pad / 4))), dim3(256), 0, s, workspace, Wk, biask,
                       bn_scale, bn_shift, wout, bout, B, Cin, Cin_pad, Cout, kk, K);
    return launch_status("somi_odconv_weights_f32");
}

extern "C" int somi_detect_decode_f32(const float *box, int box_cs, const float *cls, int cls_cs, const float *anchors_px_host,
                                      float stride, float *raw, float *z, int B, int ny, int nx, int na, int nc, int total,
                                      int row_off, somi_stream_t stream) {
    SOMI_REQUIRE(box && cls && anchors_px_host && (raw || z), SOMI_EINVAL, "detect decode: null tensor");
    SOMI_REQUIRE(B > 0 && ny > 0 && nx > 0 && na > 0 && na <= 8 && nc > 0 && box_cs >= na * 5 && cls_cs >= na * nc, SOMI_EINVAL,
                 "detect decode: bad sizes (na <= 8)");
    SOMI_REQUIRE(!z || (row_off >= 0 && row_off + na * ny * nx <= total), SOMI_EINVAL, "detect decode: rows out of range");
    DecodeArgs da;
    for (int i = 0; i < 16; ++i) da.anchor_px[i] = i < na * 2 ? anchors_px_host[i] : 0.f;
    hipLaunchKernelGGL(detect_decode_kernel, dim3(ew_grid((long)B * na * ny * nx * (nc + 5))), dim3(256), 0, (hipStream_t)stream, box,
                       box_cs, cls, cls_cs, da, stride, raw, z, B, ny, nx, na, nc, total, row_off);
    return launch_status("somi_detect_decode_f32");
}
