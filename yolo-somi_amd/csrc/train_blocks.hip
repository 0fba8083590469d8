// Backward kernels of the attention pieces of the SOMI blocks (CBAM: models/common.py:339-405, 671-691; SEAM squeeze:
// :8483-8490; global pools).  All are bandwidth-bound passes over activation-sized tensors or tiny per-sample MLPs.
//
// CBAM forward (per bottleneck):  ca = sigmoid(MLP(avg_hw t) + MLP(max_hw t));  t1 = t*ca;  stats = [mean_c t1, max_c t1];
//                                 sa = sigmoid(conv7x7(stats) + b);  t2 = t1*sa.
// Backward, given d t2:
//   A  per pixel:   dlogit = (sum_c dt2*t*ca) * sa*(1-sa);  amaxc = argmax_c (t*ca)
//   B  7x7 conv:    dstats = conv7x7^T(dlogit);  dW7 += sum_p dlogit * stats(shifted);  db7 += sum_p dlogit
//   C  per element: dt1 = dt2*sa + dstats[.,0]/C + [c==amaxc]*dstats[.,1];  dca[b,c] = sum_p dt1*t;  dt = dt1*ca
//   D  per (b,c):   amaxp = argmax_p t            (the max-pool's winner)
//   E  per sample:  MLP backward -> dW1,db1,dW2,db2 (accumulated), d avg, d max
//   F  per element: dt += davg/HW + [p==amaxp]*dmax
#include "common.h"

namespace somi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int IMG_CHUNK = 256;       // pixels per workgroup in the per-image reductions

static inline int ew_grid(long items) {
    long g = (items + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

// ------------------------------------------------------------------------------------------------ A
__global__ __launch_bounds__(256) void cbam_bwd_pixel_kernel(const float *__restrict__ dt2, int d_cs, int d_coff, const float *__restrict__ t,
                                                             int t_cs, int t_coff, const float *__restrict__ ca, const float *__restrict__ sa,
                                                             float *__restrict__ dlogit, int *__restrict__ amaxc, int B, int HW, int C,
                                                             const float *__restrict__ mxv, int *__restrict__ amaxp) {
    // LG lanes per pixel: the smallest power of two that covers the C / 4 channel quads (64 channels: 16 lanes, four pixels per wave and trip -
    // a whole wave per pixel left 48 lanes idle there and one 256-byte row in flight per wave)
    int LG = 64;
    while (LG > 1 && (LG >> 1) * 4 >= C) LG >>= 1;
    const int lane = threadIdx.x & 63, sub = lane / LG, sl = lane % LG, ppw = 64 / LG;
    const long wave_id = (blockIdx.x * 256L + threadIdx.x) >> 6, nwave = (long)gridDim.x * 4;
    const long npix = (long)B * HW;
    for (long p = wave_id * ppw + sub; p < npix; p += nwave * ppw) {
        const long b = p / HW;
        const float *tr = t + p * t_cs + t_coff, *dr = dt2 + p * d_cs + d_coff, *cr = ca + b * C;
        float s = 0.f, m = -__builtin_huge_valf();
        int mi = 0x7fffffff;
        for (int c = sl * 4; c < C; c += LG * 4) {
            const f32x4 tq = *reinterpret_cast<const f32x4 *>(tr + c);
            const f32x4 v = tq * *reinterpret_cast<const f32x4 *>(cr + c);
            const f32x4 g = *reinterpret_cast<const f32x4 *>(dr + c);
            if (mxv) {
                // step D without a pass of its own: mxv[b][c] is the channel's spatial maximum the forward pooled from these very values, so
                // the pixels that hold it are found by an exact compare; the FIRST of them (torch's rule) by an integer min - a handful
                // of atomics per (image, channel), order-independent.  (Tracking a running arg-max inside kernel C instead was measured:
                // 59 -> 90 us per launch for that bandwidth-bound kernel, more than the 25 us pass it replaced.)
                const f32x4 mq = *reinterpret_cast<const f32x4 *>(mxv + b * C + c);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (tq[e] == mq[e]) atomicMin(amaxp + b * C + c + e, (int)(p - b * HW));
            }
            s += (g[0] * v[0] + g[1] * v[1]) + (g[2] * v[2] + g[3] * v[3]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (v[e] > m) { m = v[e]; mi = c + e; }                  // first maximum inside the lane (ascending c)
        }
        for (int o = LG >> 1; o > 0; o >>= 1) {
            s += __shfl_xor(s, o);
            const float om = __shfl_xor(m, o);
            const int oi = __shfl_xor(mi, o);
            if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }   // lowest index on ties (torch.max)
        }
        if (sl == 0) {
            const float a = sa[p];
            dlogit[p] = s * a * (1.f - a);
            amaxc[p] = mi;
        }
    }
}

// ------------------------------------------------------------------------------------------------ B
__global__ __launch_bounds__(256) void spatial_attn_bwd_data_kernel(const float *__restrict__ dlogit, const float *__restrict__ w,
                                                                    float *__restrict__ dstats, int B, int H, int W, int k) {
    const long npix = (long)B * H * W;
    const int pad = k >> 1;
    for (long p = blockIdx.x * 256L + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
        const int wv = (int)(p % W), hv = (int)((p / W) % H);
        const long b = p / ((long)W * H);
        float a0 = 0.f, a1 = 0.f;
        for (int r = 0; r < k; ++r) {
            const int ho = hv - (r - pad);                              // logit[ho,wo] used stats[ho + r - pad, wo + q - pad]
            if ((unsigned)ho >= (unsigned)H) continue;
            for (int q = 0; q < k; ++q) {
                const int wo = wv - (q - pad);
                if ((unsigned)wo >= (unsigned)W) continue;
                const float g = dlogit[(b * H + ho) * W + wo];
                a0 += g * w[(r * k + q) * 2];
                a1 += g * w[(r * k + q) * 2 + 1];
            }
        }
        *reinterpret_cast<float2 *>(dstats + p * 2) = make_float2(a0, a1);
    }
}
// weight / bias gradient: workgroup (chunk, j) reduces weight j (j == k*k*2: the bias) over one chunk of pixels, lanes over pixels
__global__ __launch_bounds__(256) void spatial_attn_bwd_weight_kernel(const float *__restrict__ dlogit, const float *__restrict__ stats,
                                                                      float *__restrict__ part, int B, int H, int W, int k, int chunk) {
    __shared__ double red[4];
    const long npix = (long)B * H * W;
    const int nw = k * k * 2, pad = k >> 1;
    const int j = blockIdx.y;
    const long p0 = (long)blockIdx.x * chunk, p1 = min(p0 + chunk, npix);
    // double accumulators: these are sums of 10^4 ... 10^5 terms of both signs that cancel to a small fraction of their size (the 1280x1280
    // gradient check is sensitive to exactly this), and the kernel is tiny
    double acc = 0.0;
    if (j == nw) {
        for (long p = p0 + threadIdx.x; p < p1; p += 256) acc += (double)dlogit[p];
    } else {
        const int ch = j & 1, rq = j >> 1, r = rq / k, q = rq % k;
        for (long p = p0 + threadIdx.x; p < p1; p += 256) {
            const int wv = (int)(p % W), hv = (int)((p / W) % H);
            const int hi = hv + r - pad, wi = wv + q - pad;
            if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W)
                acc += (double)dlogit[p] * (double)stats[(p + (long)(r - pad) * W + (q - pad)) * 2 + ch];
        }
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[(long)blockIdx.x * (nw + 1) + j] = (float)((red[0] + red[1]) + (red[2] + red[3]));
}
// The same sums, one workgroup per SAW_CHUNK pixels for ALL weights: the chunk's dlogit, a per-pixel validity mask (bit r: row h + r - pad is inside
// the image, bit 8 + q: column w + q - pad is) and the stats rows it can reach (+- pad rows and columns) are staged in LDS once; thread (kernel
// position, slice) then walks every nsl-th pixel of the chunk for both input channels in fp64, and the slices are added in a fixed order.
// (Round 3: the form above - a workgroup per (chunk, weight), 64-bit division per pixel and term - took 121 us per call at 160x160 for 81 M
// multiply-adds.  Round 4: chunks of 2048 pixels made ONE workgroup's serial walk (410 pixels per thread, each a chain of dependent LDS reads behind a
// bounds branch) the launch's duration - 57 us on every map size, 25 workgroups on a 40x40 map; now 512 pixels per workgroup, the reads
// unconditional (an invalid tap multiplies by zero) and four pixels in flight.)
constexpr int SAW_CHUNK = 512;
__global__ __launch_bounds__(256) void spatial_attn_bwd_weight_tile_kernel(const float *__restrict__ dlogit, const float *__restrict__ stats,
                                                                           float *__restrict__ part, int B, int H, int W, int k) {
    extern __shared__ __attribute__((aligned(16))) float saw_lds[];
    const long npix = (long)B * H * W;
    const int nw = k * k * 2, kk = k * k, pad = k >> 1, halo = pad * W + pad, tid = threadIdx.x;
    const long p0 = (long)blockIdx.x * SAW_CHUNK;
    const int np = (int)min((long)SAW_CHUNK, npix - p0);
    float *sdl = saw_lds;                                              // [SAW_CHUNK] dlogit
    int *svm = reinterpret_cast<int *>(saw_lds + SAW_CHUNK);           // [SAW_CHUNK] validity bits of the k rows (0..) and k columns (8..)
    float *sst = saw_lds + 2 * SAW_CHUNK;                              // [SAW_CHUNK + 2 halo][2] stats
    for (int i = tid; i < SAW_CHUNK; i += 256) {
        const long p = p0 + i;
        float g = 0.f;
        int m = 0;
        if (i < np) {
            g = dlogit[p];
            const int hv = (int)((p / W) % H), wv = (int)(p % W);
            for (int r = 0; r < k; ++r) {
                m |= ((unsigned)(hv + r - pad) < (unsigned)H) << r;
                m |= ((unsigned)(wv + r - pad) < (unsigned)W) << (8 + r);
            }
        }
        sdl[i] = g;                                                    // pixels past the end: dlogit 0, no valid tap
        svm[i] = m;
    }
    for (int i = tid; i < SAW_CHUNK + 2 * halo; i += 256) {
        const long q = p0 - halo + i;
        const float2 v = (q >= 0 && q < npix) ? *reinterpret_cast<const float2 *>(stats + q * 2) : make_float2(0.f, 0.f);
        *reinterpret_cast<float2 *>(sst + i * 2) = v;
    }
    __syncthreads();
    const int nsl = 256 / kk;                                          // 5 / 10 / 28 slices for k = 7 / 5 / 3
    double a0 = 0.0, a1 = 0.0, ab = 0.0;
    if (tid < kk * nsl) {
        const int rq = tid % kk, sl = tid / kk, r = rq / k, q = rq % k;
        const int off = (r - pad) * W + (q - pad) + halo;
        const int sel = (1 << r) | (1 << (8 + q));
        const bool first = rq == 0;                                    // the bias: each slice's position-0 thread sums its pixels' dlogit
        int i = sl;
        for (; i + 3 * nsl < SAW_CHUNK; i += 4 * nsl) {               // the chunk is padded with zeros, so no tail test on np
            float g[4];
            int m[4];
            float2 st[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                g[u] = sdl[i + u * nsl];
                m[u] = svm[i + u * nsl];
                st[u] = *reinterpret_cast<const float2 *>(sst + (i + u * nsl + off) * 2);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (first) ab += (double)g[u];
                const double gd = (m[u] & sel) == sel ? (double)g[u] : 0.0;
                a0 += gd * (double)st[u].x;
                a1 += gd * (double)st[u].y;
            }
        }
        for (; i < SAW_CHUNK; i += nsl) {
            const float g = sdl[i];
            const float2 st = *reinterpret_cast<const float2 *>(sst + (i + off) * 2);
            if (first) ab += (double)g;
            const double gd = (svm[i] & sel) == sel ? (double)g : 0.0;
            a0 += gd * (double)st.x;
            a1 += gd * (double)st.y;
        }
    }
    __syncthreads();
    double *red = reinterpret_cast<double *>(saw_lds), *redb = red + 512;
    red[tid * 2] = a0;
    red[tid * 2 + 1] = a1;
    if (tid < kk * nsl && tid % kk == 0) redb[tid / kk] = ab;
    __syncthreads();
    if (tid < kk) {
        double s0 = 0.0, s1 = 0.0;
        for (int sl = 0; sl < nsl; ++sl) { s0 += red[(sl * kk + tid) * 2]; s1 += red[(sl * kk + tid) * 2 + 1]; }
        part[(long)blockIdx.x * (nw + 1) + tid * 2] = (float)s0;
        part[(long)blockIdx.x * (nw + 1) + tid * 2 + 1] = (float)s1;
    }
    if (tid == 255) {
        double sb = 0.0;
        for (int sl = 0; sl < nsl; ++sl) sb += redb[sl];
        part[(long)blockIdx.x * (nw + 1) + nw] = (float)sb;
    }
}
// one workgroup per weight: the chunk partials in fp64, fixed order (thread-strided sums, then waves, then the four wave sums)
__global__ __launch_bounds__(256) void spatial_attn_bwd_weight_final(const float *__restrict__ part, int nblk, int nw, float *dw, float *dbias,
                                                                     int chw) {
    __shared__ double red[4];
    const int j = blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) s += (double)part[(long)i * (nw + 1) + j];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x != 0) return;
    const float v = (float)((red[0] + red[1]) + (red[2] + red[3]));
    if (j == nw) *dbias += v;
    else dw[chw ? (j & 1) * (nw >> 1) + (j >> 1) : j] += v;      // [k][k][2] (the packed forward layout), or nn.Conv2d's (1,2,k,k)
}

// ------------------------------------------------------------------------------------------------ C
// grid (nchunk, B): lanes over channel quads, rows over the chunk's pixels; writes dt in place of dt2 and the partial dca sums
__global__ __launch_bounds__(256) void cbam_bwd_chan_kernel(float *__restrict__ dt2, int d_cs, int d_coff, const float *__restrict__ t, int t_cs,
                                                            int t_coff, const float *__restrict__ ca, const float *__restrict__ sa,
                                                            const float *__restrict__ dstats, const int *__restrict__ amaxc,
                                                            float *__restrict__ part, int HW, int C, int nchunk) {
    __shared__ f32x4 l1[256];
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int C4 = C >> 2;
    const int p0 = chunk * IMG_CHUNK, p1 = min(p0 + IMG_CHUNK, HW);
    const float inv_c = 1.f / (float)C;
    for (int cq0 = 0; cq0 < C4; cq0 += 256) {
        const int ncq = min(256, C4 - cq0);
        const int rows_par = 256 / ncq;
        const int cq = threadIdx.x % ncq, rr = threadIdx.x / ncq;
        const int c = (cq0 + cq) * 4;
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f};
        if (rr < rows_par) {
            const f32x4 cav = *reinterpret_cast<const f32x4 *>(ca + (long)b * C + c);
            auto one = [&](long p, const f32x4 tv, f32x4 g, const float2 ds, int am, float sav) {
                g = g * sav + ds.x * inv_c;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (c + e == am) g[e] += ds.y;
                s1 += g * tv;                                            // d ca
                *reinterpret_cast<f32x4 *>(dt2 + p * d_cs + d_coff + c) = g * cav;
            };
            int pl = p0 + rr;
            for (; pl + 3 * rows_par < p1; pl += 4 * rows_par) {          // four pixels' loads go out together; sums and stores in pixel order
                f32x4 tv[4], g[4];
                float2 ds[4];
                int am[4];
                float sav[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long p = (long)b * HW + pl + u * rows_par;
                    tv[u] = *reinterpret_cast<const f32x4 *>(t + p * t_cs + t_coff + c);
                    g[u] = *reinterpret_cast<const f32x4 *>(dt2 + p * d_cs + d_coff + c);
                    ds[u] = *reinterpret_cast<const float2 *>(dstats + p * 2);
                    am[u] = amaxc[p];
                    sav[u] = sa[p];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) one((long)b * HW + pl + u * rows_par, tv[u], g[u], ds[u], am[u], sav[u]);
            }
            for (; pl < p1; pl += rows_par) {
                const long p = (long)b * HW + pl;
                one(p, *reinterpret_cast<const f32x4 *>(t + p * t_cs + t_coff + c), *reinterpret_cast<const f32x4 *>(dt2 + p * d_cs + d_coff + c),
                    *reinterpret_cast<const float2 *>(dstats + p * 2), amaxc[p], sa[p]);
            }
        }
        l1[threadIdx.x] = s1;
        __syncthreads();
        if (threadIdx.x < ncq) {
            for (int r2 = 1; r2 < rows_par; ++r2) s1 += l1[r2 * ncq + cq];
            *reinterpret_cast<f32x4 *>(part + ((long)b * nchunk + chunk) * C + c) = s1;
        }
        __syncthreads();
    }
}
// 16 (sample, channel) columns x 16 chunk groups per workgroup, four rows per trip; fp64, groups combined in ascending order
__global__ __launch_bounds__(256) void img_partial_sum_kernel(const float *__restrict__ part, int nchunk, int C, int B, float *__restrict__ out) {
    __shared__ double ls[256];
    const int cl = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + cl;
    double s = 0.0;
    if (i < B * C) {
        const int b = i / C, c = i % C;
        const float *pp = part + (long)b * nchunk * C + c;
        for (int k = grp; k < nchunk; k += 64) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = k + u * 16 < nchunk ? pp[(long)(k + u * 16) * C] : 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) s += (double)v[u];
        }
    }
    ls[threadIdx.x] = s;
    __syncthreads();
    if (grp != 0 || i >= B * C) return;
#pragma unroll
    for (int g = 1; g < 16; ++g) s += ls[g * 16 + cl];
    out[i] = (float)s;
}

// ------------------------------------------------------------------------------------------------ D
__global__ __launch_bounds__(256) void pool_argmax_stage1(const float *__restrict__ x, int x_cs, int x_coff, int HW, int C,
                                                          float *__restrict__ pmax, int *__restrict__ pidx, int nchunk) {
    __shared__ float lm[256][4];
    __shared__ int li[256][4];
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int C4 = C >> 2;
    const int p0 = chunk * IMG_CHUNK, p1 = min(p0 + IMG_CHUNK, HW);
    for (int cq0 = 0; cq0 < C4; cq0 += 256) {
        const int ncq = min(256, C4 - cq0);
        const int rows_par = 256 / ncq;
        const int cq = threadIdx.x % ncq, rr = threadIdx.x / ncq;
        const int c = (cq0 + cq) * 4;
        float m[4];
        int mi[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { m[e] = -__builtin_huge_valf(); mi[e] = 0x7fffffff; }
        if (rr < rows_par) {
#pragma unroll 4
            for (int pl = p0 + rr; pl < p1; pl += rows_par) {            // unrolled: four rows' loads in flight
                const f32x4 v = *reinterpret_cast<const f32x4 *>(x + ((long)b * HW + pl) * x_cs + x_coff + c);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (v[e] > m[e]) { m[e] = v[e]; mi[e] = pl; }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { lm[threadIdx.x][e] = m[e]; li[threadIdx.x][e] = mi[e]; }
        __syncthreads();
        if (threadIdx.x < ncq) {
            for (int r2 = 1; r2 < rows_par; ++r2)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float om = lm[r2 * ncq + cq][e];
                    const int oi = li[r2 * ncq + cq][e];
                    if (om > m[e] || (om == m[e] && oi < mi[e])) { m[e] = om; mi[e] = oi; }
                }
            const long o = ((long)b * nchunk + chunk) * C + c;
#pragma unroll
            for (int e = 0; e < 4; ++e) { pmax[o + e] = m[e]; pidx[o + e] = mi[e]; }
        }
        __syncthreads();
    }
}
// 16 columns x 16 chunk groups; a group keeps its first maximum (chunks ascend in pixel order inside a group), the groups are merged by
// (value, lowest pixel index) - the first maximum in pixel order, as torch.max returns it
__global__ __launch_bounds__(256) void pool_argmax_stage2(const float *__restrict__ pmax, const int *__restrict__ pidx, int nchunk, int C, int B,
                                                          int *__restrict__ amaxp) {
    __shared__ float lm[256];
    __shared__ int li[256];
    const int cl = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + cl;
    float m = -__builtin_huge_valf();
    int mi = 0x7fffffff;
    if (i < B * C) {
        const int b = i / C, c = i % C;
        const long base = (long)b * nchunk * C + c;
        for (int k = grp; k < nchunk; k += 64) {
            float v[4];
            int x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool ok = k + u * 16 < nchunk;
                v[u] = ok ? pmax[base + (long)(k + u * 16) * C] : -__builtin_huge_valf();
                x[u] = ok ? pidx[base + (long)(k + u * 16) * C] : 0x7fffffff;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (v[u] > m || (v[u] == m && x[u] < mi)) { m = v[u]; mi = x[u]; }
        }
    }
    lm[threadIdx.x] = m;
    li[threadIdx.x] = mi;
    __syncthreads();
    if (grp != 0 || i >= B * C) return;
#pragma unroll
    for (int g = 1; g < 16; ++g) {
        const float om = lm[g * 16 + cl];
        const int oi = li[g * 16 + cl];
        if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
    }
    amaxp[i] = mi;
}

// ------------------------------------------------------------------------------------------------ E
// one workgroup per sample.  mode 0 (CBAM): out = sigmoid(o_avg + o_max), o_x = W2 relu(W1 x + b1) + b2
//                            mode 1 (SEAM): out = exp(sigmoid(W2 relu(W1 avg)))               (no biases)
// dout = gradient w.r.t. `out`.  Stage 1 (this kernel) produces the data gradients and leaves the per-sample factors of the
// weight gradients in `ws` = [B][C + 3*mid]: dov (C), ra (mid), dh_avg (mid), dh_max (mid); stage 2 sums the outer products
// over the samples in ascending order (no atomics: run-to-run bit-identical).
__global__ __launch_bounds__(256) void attn_mlp_bwd_kernel(int mode, const float *__restrict__ dout, const float *__restrict__ out,
                                                           const float *__restrict__ avg, const float *__restrict__ mx,
                                                           const float *__restrict__ W1, const float *__restrict__ b1,
                                                           const float *__restrict__ W2, float *__restrict__ ws,
                                                           float *__restrict__ davg, float *__restrict__ dmax, int C, int mid) {
    __shared__ float h_avg[64], h_max[64], dh_avg[64], dh_max[64];
    __shared__ float dov[1024];                                          // d(pre-activation of the second layer), C <= 1024
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *va = avg + (long)b * C, *vm = mx ? mx + (long)b * C : nullptr;
    float *wb = ws + (long)b * (C + 3 * mid);
    for (int j = wave; j < mid; j += 4) {
        float sa = 0.f, sm = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float w = W1[(long)j * C + c];
            sa += w * va[c];
            if (vm) sm += w * vm[c];
        }
        for (int o = 32; o > 0; o >>= 1) { sa += __shfl_down(sa, o); sm += __shfl_down(sm, o); }
        if (lane == 0) {
            const float bb = b1 ? b1[j] : 0.f;
            h_avg[j] = sa + bb;                                          // pre-ReLU
            h_max[j] = sm + bb;
            wb[C + j] = fmaxf(sa + bb, 0.f) + (mode == 0 ? fmaxf(sm + bb, 0.f) : 0.f);      // ra
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        const float o = out[(long)b * C + c], g = dout[(long)b * C + c];
        float d;
        if (mode == 0) d = g * o * (1.f - o);                            // sigmoid'
        else { const float sg = logf(o); d = g * o * sg * (1.f - sg); }  // out = exp(s), s = sigmoid(.): d/dpre = out * s(1-s)
        dov[c] = d;
        wb[c] = d;
    }
    __syncthreads();
    for (int j = wave; j < mid; j += 4) {                                // dh = W2^T dov, gated by the ReLU
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += W2[(long)c * mid + j] * dov[c];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
        if (lane == 0) {
            dh_avg[j] = h_avg[j] > 0.f ? s : 0.f;
            dh_max[j] = (mode == 0 && h_max[j] > 0.f) ? s : 0.f;
            wb[C + mid + j] = dh_avg[j];
            wb[C + 2 * mid + j] = dh_max[j];
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float da = 0.f, dm = 0.f;
        for (int j = 0; j < mid; ++j) {
            const float w = W1[(long)j * C + c];
            da += w * dh_avg[j];
            dm += w * dh_max[j];
        }
        davg[(long)b * C + c] = da;
        if (dmax) dmax[(long)b * C + c] = dm;
    }
}

// stage 2: one lane per (j, c) pair of the two weight matrices (+ the bias lanes); samples summed in ascending order, ADDED to the outputs
__global__ __launch_bounds__(256) void attn_mlp_bwd_weights_kernel(int mode, const float *__restrict__ ws, const float *__restrict__ avg,
                                                                   const float *__restrict__ mx, float *dW1, float *db1, float *dW2, float *db2,
                                                                   int B, int C, int mid) {
    const int stride = C + 3 * mid;
    const long n = (long)C * mid;
    for (long it = blockIdx.x * 256L + threadIdx.x; it < n; it += (long)gridDim.x * 256) {
        const int c = (int)(it % C), j = (int)(it / C);                  // consecutive lanes: consecutive channels
        float s1 = 0.f, s2 = 0.f;
#pragma unroll 8
        for (int b = 0; b < B; ++b) {                                    // unrolled: eight samples' loads in flight, sums in sample order
            const float *wb = ws + (long)b * stride;
            s2 += wb[c] * wb[C + j];
            s1 += wb[C + mid + j] * avg[(long)b * C + c] + (mx ? wb[C + 2 * mid + j] * mx[(long)b * C + c] : 0.f);
        }
        dW2[(long)c * mid + j] += s2;
        dW1[(long)j * C + c] += s1;
        if (j == 0 && db2) {
            float s = 0.f;
            for (int b = 0; b < B; ++b) s += ws[(long)b * stride + c];
            db2[c] += mode == 0 ? 2.f * s : s;
        }
        if (c == 0 && db1) {
            float s = 0.f;
            for (int b = 0; b < B; ++b) s += ws[(long)b * stride + C + mid + j] + ws[(long)b * stride + C + 2 * mid + j];
            db1[j] += s;
        }
    }
}

// ------------------------------------------------------------------------------------------------ F
// blockIdx.y = image; a thread keeps one channel quad of that image (its pooled gradients and argmax positions live in registers) and walks
// the image's pixels four at a time with the loads issued together - no division, no per-element index loads in the loop (round 3: the
// item-indexed form ran at 2.4 TB/s).  The host makes gridDim.x * 256 a multiple of C / 4.
__global__ __launch_bounds__(256) void pool_bwd_add_kernel(float *dt, int d_cs, int d_coff, const float *__restrict__ davg,
                                                           const float *__restrict__ dmax, const int *__restrict__ amaxp, int B, int HW, int C) {
    const unsigned C4 = (unsigned)C >> 2, nthreads = gridDim.x * 256u, t = blockIdx.x * 256u + threadIdx.x;
    const int c = (int)(t % C4) * 4, b = blockIdx.y;
    const int pstep = (int)(nthreads / C4);
    const f32x4 avg = *reinterpret_cast<const f32x4 *>(davg + (long)b * C + c) * (1.f / (float)HW);
    f32x4 mx = {0.f, 0.f, 0.f, 0.f};
    int am[4] = {-1, -1, -1, -1};
    if (dmax) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { mx[e] = dmax[(long)b * C + c + e]; am[e] = amaxp[(long)b * C + c + e]; }
    }
    float *base = dt + (long)b * HW * d_cs + d_coff + c;
    auto one = [&](f32x4 g, int pl) {
        g += avg;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (am[e] == pl) g[e] += mx[e];
        return g;
    };
    int p = (int)(t / C4);
    for (; p + 3 * pstep < HW; p += 4 * pstep) {
        f32x4 g[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) g[u] = *reinterpret_cast<const f32x4 *>(base + (long)(p + u * pstep) * d_cs);
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<f32x4 *>(base + (long)(p + u * pstep) * d_cs) = one(g[u], p + u * pstep);
    }
    for (; p < HW; p += pstep) *reinterpret_cast<f32x4 *>(base + (long)p * d_cs) = one(*reinterpret_cast<const f32x4 *>(base + (long)p * d_cs), p);
}

// workgroups per image for the kernel above: one item per thread up to `cap` workgroups over the whole batch, a multiple of C4 / gcd(C4, 256)
static inline int img_grid_c(int B, int HW, int C, int cap) {
    const int C4 = C / 4;
    int a = C4, b = 256;
    while (b) { const int t = a % b; a = b; b = t; }
    const long mult = C4 / a;
    long g = ((long)HW * C4 + 255) / 256, per = cap / (B > 0 ? B : 1);
    if (per < 1) per = 1;
    g = g < 1 ? 1 : (g > per ? per : g);
    return (int)((g + mult - 1) / mult * mult);
}

static inline bool sl_ok(const void *p, int cs, int coff, int C) { return p && cs % 4 == 0 && coff % 4 == 0 && coff + C <= cs && aligned16(p); }

}  // namespace somi

using namespace somi;

extern "C" int somi_img_nchunk(int HW) { return (HW + IMG_CHUNK - 1) / IMG_CHUNK; }

extern "C" int somi_cbam_bwd_pixel_f32(const float *dt2, int d_cs, int d_coff, const float *t, int t_cs, int t_coff, const float *ca,
                                       const float *sa, float *dlogit, int32_t *amaxc, int B, int HW, int C, somi_stream_t stream) {
    SOMI_REQUIRE(sl_ok(dt2, d_cs, d_coff, C) && sl_ok(t, t_cs, t_coff, C) && ca && sa && dlogit && amaxc && B > 0 && HW > 0 && C % 4 == 0 &&
                     aligned16(ca), SOMI_EINVAL, "cbam bwd pixel: bad arguments");
    hipLaunchKernelGGL(cbam_bwd_pixel_kernel, dim3(ew_grid((long)B * HW * 64)), dim3(256), 0, (hipStream_t)stream, dt2, d_cs, d_coff, t, t_cs,
                       t_coff, ca, sa, dlogit, amaxc, B, HW, C, nullptr, nullptr);
    return launch_status("somi_cbam_bwd_pixel_f32");
}

extern "C" int somi_cbam_bwd_pixel_argmax_f32(const float *dt2, int d_cs, int d_coff, const float *t, int t_cs, int t_coff, const float *ca,
                                              const float *sa, const float *t_max, float *dlogit, int32_t *amaxc, int32_t *amaxp, int B, int HW,
                                              int C, somi_stream_t stream) {
    SOMI_REQUIRE(sl_ok(dt2, d_cs, d_coff, C) && sl_ok(t, t_cs, t_coff, C) && ca && sa && dlogit && amaxc && t_max && amaxp && B > 0 && HW > 0 &&
                     C % 4 == 0 && aligned16(ca) && aligned16(t_max), SOMI_EINVAL, "cbam bwd pixel + argmax: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    (void)hipMemsetAsync(amaxp, 0x7f, (size_t)B * C * sizeof(int32_t), s);          // 0x7f7f7f7f: above every pixel index, the integer min's identity here
    hipLaunchKernelGGL(cbam_bwd_pixel_kernel, dim3(ew_grid((long)B * HW * 64)), dim3(256), 0, s, dt2, d_cs, d_coff, t, t_cs, t_coff, ca, sa, dlogit,
                       amaxc, B, HW, C, t_max, amaxp);
    return launch_status("somi_cbam_bwd_pixel_argmax_f32");
}

extern "C" int somi_spatial_attn_bwd_f32(const float *dlogit, const float *stats, const float *w, float *dstats, float *dw_accumulate,
                                         float *dbias_accumulate, float *workspace, int B, int H, int W, int k, int dw_chw,
                                         somi_stream_t stream) {
    SOMI_REQUIRE(dlogit && stats && w && dstats && dw_accumulate && dbias_accumulate && workspace && B > 0 && H > 0 && W > 0 &&
                     (k == 3 || k == 5 || k == 7), SOMI_EINVAL, "spatial attn bwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const long npix = (long)B * H * W;
    hipLaunchKernelGGL(spatial_attn_bwd_data_kernel, dim3(ew_grid(npix)), dim3(256), 0, s, dlogit, w, dstats, B, H, W, k);
    const int nw = k * k * 2, halo = (k >> 1) * (W + 1);
    const size_t lds = (size_t)(4 * SAW_CHUNK + 4 * halo) * sizeof(float);
    int nblk;
    if (lds <= 64 * 1024) {                                           // workspace: (npix / 512 rounded up) rows of nw + 1 floats
        nblk = (int)((npix + SAW_CHUNK - 1) / SAW_CHUNK);
        hipLaunchKernelGGL(spatial_attn_bwd_weight_tile_kernel, dim3(nblk), dim3(256), lds, s, dlogit, stats, workspace, B, H, W, k);
    } else {
        const int chunk = 1024 * 16;
        nblk = (int)((npix + chunk - 1) / chunk);
        hipLaunchKernelGGL(spatial_attn_bwd_weight_kernel, dim3(nblk, nw + 1), dim3(256), 0, s, dlogit, stats, workspace, B, H, W, k, chunk);
    }
    hipLaunchKernelGGL(spatial_attn_bwd_weight_final, dim3(nw + 1), dim3(256), 0, s, workspace, nblk, nw, dw_accumulate, dbias_accumulate,
                       dw_chw);
    return launch_status("somi_spatial_attn_bwd_f32");
}

extern "C" int somi_cbam_bwd_chan_f32(float *dt2_inout, int d_cs, int d_coff, const float *t, int t_cs, int t_coff, const float *ca,
                                      const float *sa, const float *dstats, const int32_t *amaxc, float *dca, float *workspace, int B, int HW,
                                      int C, somi_stream_t stream) {
    SOMI_REQUIRE(sl_ok(dt2_inout, d_cs, d_coff, C) && sl_ok(t, t_cs, t_coff, C) && ca && sa && dstats && amaxc && dca && workspace && B > 0 &&
                     HW > 0 && C % 4 == 0 && aligned16(ca) && aligned16(workspace), SOMI_EINVAL, "cbam bwd chan: bad arguments");
    const int nchunk = somi_img_nchunk(HW);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(cbam_bwd_chan_kernel, dim3(nchunk, B), dim3(256), 0, s, dt2_inout, d_cs, d_coff, t, t_cs, t_coff, ca, sa, dstats, amaxc,
                       workspace, HW, C, nchunk);
    hipLaunchKernelGGL(img_partial_sum_kernel, dim3(cdiv((long)B * C, 16)), dim3(256), 0, s, workspace, nchunk, C, B, dca);
    return launch_status("somi_cbam_bwd_chan_f32");
}

extern "C" int somi_pool_argmax_nhwc_f32(const float *x, int x_cs, int x_coff, int B, int HW, int C, int32_t *amaxp, void *workspace,
                                         somi_stream_t stream) {
    SOMI_REQUIRE(sl_ok(x, x_cs, x_coff, C) && amaxp && workspace && B > 0 && HW > 0 && C % 4 == 0, SOMI_EINVAL, "pool argmax: bad arguments");
    const int nchunk = somi_img_nchunk(HW);
    float *pm = static_cast<float *>(workspace);
    int *pi = reinterpret_cast<int *>(pm + (size_t)B * nchunk * C);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(pool_argmax_stage1, dim3(nchunk, B), dim3(256), 0, s, x, x_cs, x_coff, HW, C, pm, pi, nchunk);
    hipLaunchKernelGGL(pool_argmax_stage2, dim3(cdiv((long)B * C, 16)), dim3(256), 0, s, pm, pi, nchunk, C, B, amaxp);
    return launch_status("somi_pool_argmax_nhwc_f32");
}

extern "C" size_t somi_attn_mlp_bwd_workspace_floats(int B, int C, int mid) { return (size_t)B * (C + 3 * mid); }

extern "C" int somi_attn_mlp_bwd_f32(int mode, const float *dout, const float *out, const float *avg, const float *mx, const float *W1,
                                     const float *b1, const float *W2, float *dW1, float *db1, float *dW2, float *db2, float *davg,
                                     float *dmax, float *workspace, int B, int C, int mid, somi_stream_t stream) {
    SOMI_REQUIRE(dout && out && avg && W1 && W2 && dW1 && dW2 && davg && workspace && B > 0 && C > 0 && C <= 1024 && mid > 0 && mid <= 64,
                 SOMI_EINVAL, "attn mlp bwd: bad arguments (C <= 1024, mid <= 64)");
    SOMI_REQUIRE((mode == 0 && mx && dmax) || mode == 1, SOMI_EINVAL, "attn mlp bwd: mode 0 needs max inputs/outputs");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(attn_mlp_bwd_kernel, dim3(B), dim3(256), 0, s, mode, dout, out, avg, mode == 0 ? mx : nullptr, W1, b1, W2, workspace, davg,
                       mode == 0 ? dmax : nullptr, C, mid);
    hipLaunchKernelGGL(attn_mlp_bwd_weights_kernel, dim3(cdiv((long)C * mid, 256)), dim3(256), 0, s, mode, workspace, avg, mode == 0 ? mx : nullptr, dW1,
                       db1, dW2, db2, B, C, mid);
    return launch_status("somi_attn_mlp_bwd_f32");
}

extern "C" int somi_pool_bwd_add_nhwc_f32(float *dt_inout, int d_cs, int d_coff, const float *davg, const float *dmax, const int32_t *amaxp,
                                          int B, int HW, int C, somi_stream_t stream) {
    SOMI_REQUIRE(sl_ok(dt_inout, d_cs, d_coff, C) && davg && (!dmax || amaxp) && B > 0 && HW > 0 && C % 4 == 0 && aligned16(davg), SOMI_EINVAL,
                 "pool bwd add: bad arguments");
    SOMI_REQUIRE(B <= 65535, SOMI_EINVAL, "pool bwd add: batch beyond the grid's y range");
    hipLaunchKernelGGL(pool_bwd_add_kernel, dim3(img_grid_c(B, HW, C, 7 * 256), B), dim3(256), 0, (hipStream_t)stream, dt_inout, d_cs, d_coff, davg,
                       dmax, amaxp, B, HW, C);
    return launch_status("somi_pool_bwd_add_nhwc_f32");
}

// =================================================================================================================
// Backward of the remaining layer kernels: detect raw assembly, SPPF pooling, BiFPN fusion, depthwise 3x3, SEAM scaling.
namespace somi {

// ---- detect: d raw (B,na,ny,nx,no) -> d box (B,ny,nx,box_cs) / d cls (B,ny,nx,cls_cs); pad channels get zeros
__global__ __launch_bounds__(256) void detect_raw_bwd_kernel(const float *__restrict__ draw, float *__restrict__ dbox, int box_cs,
                                                             float *__restrict__ dcls, int cls_cs, int B, int ny, int nx, int na, int nc) {
    const int no = nc + 5;
    const long npix = (long)B * ny * nx;
    const long items = npix * (box_cs + cls_cs);
    for (long it = blockIdx.x * 256L + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
        const int j = (int)(it % (box_cs + cls_cs));
        const long pix = it / (box_cs + cls_cs);
        const int xg = (int)(pix % nx), yg = (int)((pix / nx) % ny);
        const long b = pix / ((long)nx * ny);
        if (j < box_cs) {
            float v = 0.f;
            if (j < na * 5) { const int an = j / 5, o = j % 5; v = draw[((((b * na + an) * ny + yg) * nx + xg)) * no + o]; }
            dbox[pix * box_cs + j] = v;
        } else {
            const int jc = j - box_cs;
            float v = 0.f;
            if (jc < na * nc) { const int an = jc / nc, o = jc % nc; v = draw[((((b * na + an) * ny + yg) * nx + xg)) * no + 5 + o]; }
            dcls[pix * cls_cs + jc] = v;
        }
    }
}

// ---- SPPF / SPP: the three pooled slices are CHAINED 5x5 max-pools (models/common.py:1846-1861: y1 = m(x), y2 = m(y1), y3 = m(y2)), and autograd
// routes each pool's gradient to the arg-max element of its 5x5 window of the PREVIOUS slice.  Gather form, no atomics (run-to-run bit-identical):
// pass 1 records, per (element, level), where in its 5x5 window the maximum of the previous slice sits - one byte r*5 + q, first maximum in row-major
// order like torch's max_pool2d; then level 3, 2, 1 in turn: every element of slice l-1 adds, in a fixed order, the (already complete) gradients of the
// slice-l outputs around it whose byte points back at it.  (Rounds 1-3 treated the slices as 5 / 9 / 13 windows of slice 0 - the same values, and the
// same routing except at exact ties - and paid 275 code checks per element in one 539 us launch, after a 276 us search over 169 taps; chained: 75 + 75.)
__global__ __launch_bounds__(256) void sppf_pool_argmax_kernel(const float *__restrict__ buf, uint8_t *__restrict__ arg, int B, int H, int W, int C,
                                                               int cs, int x_coff) {
    const int C4 = C >> 2;                                             // 4 channels per lane: one 16-byte load per tap, one 4-byte code store
    const long items = (long)B * H * W * C4, plane = (long)B * H * W;
    for (long it = blockIdx.x * 256L + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
        const int c = (int)(it % C4) * 4;
        const long pix = it / C4;
        const int wv = (int)(pix % W), hv = (int)((pix / W) % H);
        const long b = pix / ((long)W * H);
        const float ninf = -__builtin_huge_valf();
        for (int l = 0; l < 3; ++l) {                                   // level l + 1 pools slice l
            f32x4 m = {ninf, ninf, ninf, ninf};
            int mi[4] = {12, 12, 12, 12};                               // the centre; every window contains it
            for (int r = 0; r < 5; ++r) {
                const int hi = hv + r - 2;
                if ((unsigned)hi >= (unsigned)H) continue;
                for (int q = 0; q < 5; ++q) {
                    const int wi = wv + q - 2;
                    if ((unsigned)wi >= (unsigned)W) continue;
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(buf + ((b * H + hi) * W + wi) * cs + x_coff + l * C + c);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (v[e] > m[e]) { m[e] = v[e]; mi[e] = r * 5 + q; }
                }
            }
            *reinterpret_cast<uint32_t *>(arg + ((long)l * plane + pix) * C + c) =
                (uint32_t)mi[0] | ((uint32_t)mi[1] << 8) | ((uint32_t)mi[2] << 16) | ((uint32_t)mi[3] << 24);
        }
    }
}

// training forward: ONE level of the chain - slice l + 1 = 5x5 max-pool of slice l - with the level's codes written beside it, so that the backward
// pass needs no search of its own (three launches of 25 taps instead of a 169-tap pooling launch plus, in backward, a 75-tap search)
__global__ __launch_bounds__(256) void sppf_pool5_codes_kernel(float *__restrict__ buf, uint8_t *__restrict__ arg, int B, int H, int W, int C, int cs,
                                                               int x_coff, int l) {
    const int C4 = C >> 2;
    const long items = (long)B * H * W * C4, plane = (long)B * H * W;
    for (long it = blockIdx.x * 256L + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
        const int c = (int)(it % C4) * 4;
        const long pix = it / C4;
        const int wv = (int)(pix % W), hv = (int)((pix / W) % H);
        const long b = pix / ((long)W * H);
        const float ninf = -__builtin_huge_valf();
        f32x4 m = {ninf, ninf, ninf, ninf};
        int mi[4] = {12, 12, 12, 12};
        for (int r = 0; r < 5; ++r) {
            const int hi = hv + r - 2;
            if ((unsigned)hi >= (unsigned)H) continue;
            for (int q = 0; q < 5; ++q) {
                const int wi = wv + q - 2;
                if ((unsigned)wi >= (unsigned)W) continue;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(buf + ((b * H + hi) * W + wi) * cs + x_coff + l * C + c);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (v[e] > m[e]) { m[e] = v[e]; mi[e] = r * 5 + q; }
            }
        }
        *reinterpret_cast<f32x4 *>(buf + pix * cs + x_coff + (l + 1) * C + c) = m;
        *reinterpret_cast<uint32_t *>(arg + ((long)l * plane + pix) * C + c) =
            (uint32_t)mi[0] | ((uint32_t)mi[1] << 8) | ((uint32_t)mi[2] << 16) | ((uint32_t)mi[3] << 24);
    }
}

// level l (2, 1, 0 in turn): slice l of dbuf += the gradients of slice l + 1 routed by the level's codes
__global__ __launch_bounds__(256) void sppf_pool_bwd_gather_kernel(const uint8_t *__restrict__ arg, float *__restrict__ dbuf, int B, int H, int W, int C,
                                                                   int cs, int x_coff, int l) {
    const int C4 = C >> 2;
    const long items = (long)B * H * W * C4, plane = (long)B * H * W;
    const uint8_t *codes_l = arg + (long)l * plane * C;
    for (long it = blockIdx.x * 256L + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
        const int c = (int)(it % C4) * 4;
        const long pix = it / C4;
        const int wv = (int)(pix % W), hv = (int)((pix / W) % H);
        const long b = pix / ((long)W * H);
        float *mine = dbuf + pix * cs + x_coff + l * C + c;
        f32x4 acc = *reinterpret_cast<const f32x4 *>(mine);
        for (int dh = -2; dh <= 2; ++dh) {                              // o = p + (dh, dw) points back at p when its code is (2 - dh, 2 - dw)
            const int ho = hv + dh;
            if ((unsigned)ho >= (unsigned)H) continue;
            for (int dw = -2; dw <= 2; ++dw) {
                const int wo = wv + dw;
                if ((unsigned)wo >= (unsigned)W) continue;
                const long o = (b * H + ho) * W + wo;
                const uint32_t want = (uint32_t)((2 - dh) * 5 + (2 - dw));
                const uint32_t x = *reinterpret_cast<const uint32_t *>(codes_l + o * C + c) ^ (want * 0x01010101u);   // a zero byte = that channel's maximum sits at p
                if (((x - 0x01010101u) & ~x & 0x80808080u) == 0) continue;
                const f32x4 d = *reinterpret_cast<const f32x4 *>(dbuf + o * cs + x_coff + (l + 1) * C + c);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (((x >> (8 * e)) & 0xFFu) == 0) acc[e] += d[e];
            }
        }
        *reinterpret_cast<f32x4 *>(mine) = acc;
    }
}

// ---- BiFPN: dx_i = wn_i * dout (2x2 sum for an upsampled source); dwn_i = sum dout * src_i
struct BifpnBwdArgs {
    const float *src[3];
    float *dsrc[3];
    int up[3];
    const float *w;      // raw fusion parameter (device)
    float eps;
    int n_in;
};
__global__ __launch_bounds__(256) void bifpn_bwd_kernel(BifpnBwdArgs a, const float *__restrict__ dout, float *__restrict__ part, int B, int H,
                                                        int W, int C) {
    __shared__ float red[3][4];
    float wn[3];
    bifpn_norm(a.w, a.n_in, a.eps, wn);
    const int C4 = C >> 2;
    const long items = (long)B * H * W * C4;
    float acc[3] = {0.f, 0.f, 0.f};
    for (long it = blockIdx.x * 256L + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
        const int c = (int)(it % C4) * 4;
        const long pix = it / C4;
        const int wv = (int)(pix % W), hv = (int)((pix / W) % H);
        const long b = pix / ((long)W * H);
        const f32x4 g = *reinterpret_cast<const f32x4 *>(dout + pix * C + c);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (i >= a.n_in) break;
            const int u = a.up[i];
            const long sp = (b * (H >> u) + (hv >> u)) * (W >> u) + (wv >> u);
            const f32x4 sv = *reinterpret_cast<const f32x4 *>(a.src[i] + sp * C + c);
            acc[i] += (g[0] * sv[0] + g[1] * sv[1]) + (g[2] * sv[2] + g[3] * sv[3]);
            if (!u) *reinterpret_cast<f32x4 *>(a.dsrc[i] + pix * C + c) = g * wn[i];
        }
    }
    for (int i = 0; i < 3; ++i) {
        float v = acc[i];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        if ((threadIdx.x & 63) == 0) red[i][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 3) part[(long)blockIdx.x * 3 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}
__global__ __launch_bounds__(256) void bifpn_bwd_up_kernel(const float *__restrict__ dout, float *__restrict__ dsrc, const float *__restrict__ w,
                                                           int n_in, float eps, int which, int B, int Hl, int Wl, int C) {
    float wn3[3];
    bifpn_norm(w, n_in, eps, wn3);
    const float wn = wn3[which];
    const int C4 = C >> 2;
    const long items = (long)B * Hl * Wl * C4;
    for (long it = blockIdx.x * 256L + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
        const int c = (int)(it % C4) * 4;
        const long pix = it / C4;
        const int wv = (int)(pix % Wl), hv = (int)((pix / Wl) % Hl);
        const long b = pix / ((long)Wl * Hl);
        const long base = ((b * (Hl * 2) + hv * 2) * (Wl * 2) + wv * 2) * C + c;
        const f32x4 s = (*reinterpret_cast<const f32x4 *>(dout + base) + *reinterpret_cast<const f32x4 *>(dout + base + C)) +
                        (*reinterpret_cast<const f32x4 *>(dout + base + (long)Wl * 2 * C) +
                         *reinterpret_cast<const f32x4 *>(dout + base + (long)Wl * 2 * C + C));
        *reinterpret_cast<f32x4 *>(dsrc + pix * C + c) = s * wn;
    }
}
// dw_k += dwn_k / S - (sum_i dwn_i w_i / S^2) * swish'(w_k),  S = sum swish(w) + eps   (models/common.py:3696)
__global__ void bifpn_bwd_weight_kernel(const float *__restrict__ part, int nblk, const float *__restrict__ w, int n_in, float eps, float *dw) {
    double dwn[3] = {0, 0, 0};
    for (int i = threadIdx.x; i < nblk; i += 64)
        for (int k = 0; k < n_in; ++k) dwn[k] += part[(long)i * 3 + k];
    for (int k = 0; k < 3; ++k)
        for (int o = 32; o > 0; o >>= 1) dwn[k] += __shfl_xor(dwn[k], o);
    if (threadIdx.x != 0) return;
    double S = eps, T = 0.0;
    for (int k = 0; k < n_in; ++k) { const double sg = 1.0 / (1.0 + exp(-(double)w[k])); S += w[k] * sg; }
    for (int k = 0; k < n_in; ++k) T += dwn[k] * w[k];
    for (int k = 0; k < n_in; ++k) {
        const double sg = 1.0 / (1.0 + exp(-(double)w[k]));
        const double dsw = sg * (1.0 + w[k] * (1.0 - sg));
        dw[k] += (float)(dwn[k] / S - T / (S * S) * dsw);
    }
}

// ---- depthwise 3x3 backward.  Data gradient = the same sliding-window stencil with the taps walked backwards
// (launch_dwconv3x3(flip)).  Weight / bias gradient: per-channel reduction over pixels of g * x(shifted); one lane owns 4
// channels of one image column and walks it top to bottom with the 3x3 window of x in registers (4 loads per pixel instead
// of 10), the lanes of a workgroup that share a channel quad are combined through LDS, one [10][C] partial per workgroup.
__global__ __launch_bounds__(256) void dwconv3x3_bwd_weight_kernel(const float *__restrict__ dy, const float *__restrict__ x, float *__restrict__ part,
                                                                   int B, int H, int W, int C) {
    __shared__ f32x4 ls[256];
    const int C4 = C >> 2;
    const long items = (long)B * W * C4;
    const long it = blockIdx.x * 256L + threadIdx.x;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[k] = zero;
    const int cq = (int)(it % C4), c = cq * 4;
    if (it < items) {
        const int wv = (int)((it / C4) % W);
        const long b = it / ((long)C4 * W);
        const bool wl = wv > 0, wr = wv + 1 < W;
        const float *xb = x + (b * H * W + wv) * C + c;
        const float *gb = dy + (b * H * W + wv) * C + c;
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
        f32x4 top[3], mid[3], bot[3];
        load_row(-1, top);
        load_row(0, mid);
        for (int h = 0; h < H; ++h) {
            load_row(h + 1, bot);
            const f32x4 g = *reinterpret_cast<const f32x4 *>(gb + (long)h * W * C);
            acc[9] += g;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                acc[q] += g * top[q];
                acc[3 + q] += g * mid[q];
                acc[6 + q] += g * bot[q];
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) { top[q] = mid[q]; mid[q] = bot[q]; }
        }
    }
    // lanes t, t + C4, t + 2*C4, ... of the workgroup share a channel quad (256 % C4 == 0, checked by the launcher)
    const int rows_par = 256 / C4, cl = threadIdx.x % C4;
    for (int k = 0; k < 10; ++k) {
        ls[threadIdx.x] = acc[k];
        __syncthreads();
        if (threadIdx.x < C4) {
            f32x4 s = ls[cl];
            for (int r2 = 1; r2 < rows_par; ++r2) s += ls[r2 * C4 + cl];
            *reinterpret_cast<f32x4 *>(part + ((long)blockIdx.x * 10 + k) * C + c) = s;
        }
        __syncthreads();
    }
}
// 256 threads = 16 (tap, channel) columns x 16 groups of partial rows, combined in a fixed order
__global__ __launch_bounds__(256) void dwconv3x3_bwd_weight_final(const float *__restrict__ part, int nchunk, int C, float *dw, float *dbias) {
    __shared__ double l[256];
    const int col = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + col;
    double s0 = 0.0, s1 = 0.0;
    if (i < 10 * C) {
        int j = grp;
        for (; j + 16 < nchunk; j += 32) { s0 += part[(long)j * 10 * C + i]; s1 += part[(long)(j + 16) * 10 * C + i]; }
        if (j < nchunk) s0 += part[(long)j * 10 * C + i];
    }
    l[threadIdx.x] = s0 + s1;
    __syncthreads();
    if (grp != 0 || i >= 10 * C) return;
    double s = 0.0;
#pragma unroll
    for (int g = 0; g < 16; ++g) s += l[g * 16 + col];
    const int k = i / C, c = i % C;
    if (k == 9) { if (dbias) dbias[c] += (float)s; }
    else dw[k * C + c] += (float)s;
}

// ---- y = x * s[b][c] backward: dx = dout * s ; ds[b,c] = sum_p dout * x     (SEAM output, models/common.py:8489-8490)
__global__ __launch_bounds__(256) void scale_channels_bwd_kernel(const float *__restrict__ dout, const float *__restrict__ x, const float *__restrict__ s,
                                                                 float *__restrict__ dx, float *__restrict__ part, int HW, int C, int nchunk) {
    __shared__ f32x4 l1[256];
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int C4 = C >> 2;
    const int p0 = chunk * IMG_CHUNK, p1 = min(p0 + IMG_CHUNK, HW);
    for (int cq0 = 0; cq0 < C4; cq0 += 256) {
        const int ncq = min(256, C4 - cq0);
        const int rows_par = 256 / ncq;
        const int cq = threadIdx.x % ncq, rr = threadIdx.x / ncq;
        const int c = (cq0 + cq) * 4;
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f};
        if (rr < rows_par) {
            const f32x4 sv = *reinterpret_cast<const f32x4 *>(s + (long)b * C + c);
            for (int pl = p0 + rr; pl < p1; pl += rows_par) {
                const long p = (long)b * HW + pl;
                const f32x4 g = *reinterpret_cast<const f32x4 *>(dout + p * C + c);
                s1 += g * *reinterpret_cast<const f32x4 *>(x + p * C + c);
                *reinterpret_cast<f32x4 *>(dx + p * C + c) = g * sv;
            }
        }
        l1[threadIdx.x] = s1;
        __syncthreads();
        if (threadIdx.x < ncq) {
            for (int r2 = 1; r2 < rows_par; ++r2) s1 += l1[r2 * ncq + cq];
            *reinterpret_cast<f32x4 *>(part + ((long)b * nchunk + chunk) * C + c) = s1;
        }
        __syncthreads();
    }
}

}  // namespace somi

extern "C" int somi_detect_raw_bwd_f32(const float *draw, float *dbox, int box_cs, float *dcls, int cls_cs, int B, int ny, int nx, int na,
                                       int nc, somi_stream_t stream) {
    SOMI_REQUIRE(draw && dbox && dcls && B > 0 && ny > 0 && nx > 0 && na > 0 && nc > 0 && box_cs >= na * 5 && cls_cs >= na * nc, SOMI_EINVAL,
                 "detect raw bwd: bad arguments");
    hipLaunchKernelGGL(detect_raw_bwd_kernel, dim3(ew_grid((long)B * ny * nx * (box_cs + cls_cs))), dim3(256), 0, (hipStream_t)stream, draw, dbox,
                       box_cs, dcls, cls_cs, B, ny, nx, na, nc);
    return launch_status("somi_detect_raw_bwd_f32");
}

extern "C" int somi_sppf_pool_bwd_nhwc_f32(const float *buf, float *dbuf, void *workspace, int B, int H, int W, int C, int cs, int x_coff,
                                           somi_stream_t stream) {
    SOMI_REQUIRE(dbuf && workspace && B > 0 && H > 0 && W > 0 && C > 0 && x_coff + 4 * C <= cs, SOMI_EINVAL, "sppf bwd: bad arguments");
    SOMI_REQUIRE(C % 4 == 0 && cs % 4 == 0 && x_coff % 4 == 0 && aligned16(buf) && aligned16(dbuf) && (reinterpret_cast<uintptr_t>(workspace) & 3u) == 0,
                 SOMI_EINVAL, "sppf bwd: channels / strides must be multiples of 4, tensors 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    uint8_t *arg = static_cast<uint8_t *>(workspace);                    // 3*B*H*W*C bytes
    if (buf)                                                             // buf == NULL: the workspace holds the codes somi_sppf_pool_codes_nhwc_f32 left
        hipLaunchKernelGGL(sppf_pool_argmax_kernel, dim3(ew_grid((long)B * H * W * (C / 4))), dim3(256), 0, s, buf, arg, B, H, W, C, cs, x_coff);
    for (int l = 2; l >= 0; --l)                                         // slice 2 takes slice 3's gradients, then slice 1 takes slice 2's, then slice 0
        hipLaunchKernelGGL(sppf_pool_bwd_gather_kernel, dim3(ew_grid((long)B * H * W * (C / 4))), dim3(256), 0, s, arg, dbuf, B, H, W, C, cs, x_coff, l);
    return launch_status("somi_sppf_pool_bwd_nhwc_f32");
}

extern "C" int somi_sppf_pool_codes_nhwc_f32(float *buf, void *codes, int B, int H, int W, int C, int cs, int x_coff, somi_stream_t stream) {
    SOMI_REQUIRE(buf && codes && B > 0 && H > 0 && W > 0 && C > 0 && x_coff + 4 * C <= cs, SOMI_EINVAL, "sppf (codes): bad arguments");
    SOMI_REQUIRE(C % 4 == 0 && cs % 4 == 0 && x_coff % 4 == 0 && aligned16(buf) && (reinterpret_cast<uintptr_t>(codes) & 3u) == 0, SOMI_EINVAL,
                 "sppf (codes): channels / strides must be multiples of 4, the tensor 16-byte aligned");
    for (int l = 0; l < 3; ++l)
        hipLaunchKernelGGL(sppf_pool5_codes_kernel, dim3(ew_grid((long)B * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, buf,
                           static_cast<uint8_t *>(codes), B, H, W, C, cs, x_coff, l);
    return launch_status("somi_sppf_pool_codes_nhwc_f32");
}

extern "C" int somi_bifpn_bwd_nhwc_f32(const float *const *src_host, float *const *dsrc_host, const int *up_host, const float *w_dev,
                                       float eps, int n_in, const float *dout, float *dw_accumulate, float *workspace, int B, int H,
                                       int W, int C, somi_stream_t stream) {
    SOMI_REQUIRE(src_host && dsrc_host && up_host && w_dev && dout && dw_accumulate && workspace && (n_in == 2 || n_in == 3) &&
                     C % 4 == 0, SOMI_EINVAL, "bifpn bwd: bad arguments");
    BifpnBwdArgs a;
    for (int i = 0; i < 3; ++i) {
        a.src[i] = i < n_in ? src_host[i] : nullptr;
        a.dsrc[i] = i < n_in ? dsrc_host[i] : nullptr;
        a.up[i] = i < n_in ? up_host[i] : 0;
        SOMI_REQUIRE(i >= n_in || (a.src[i] && a.dsrc[i] && (a.up[i] == 0 || a.up[i] == 1)), SOMI_EINVAL, "bifpn bwd: bad source %d", i);
    }
    a.n_in = n_in;
    a.w = w_dev;
    a.eps = eps;
    hipStream_t s = (hipStream_t)stream;
    const int nblk = ew_grid((long)B * H * W * (C / 4));
    hipLaunchKernelGGL(bifpn_bwd_kernel, dim3(nblk), dim3(256), 0, s, a, dout, workspace, B, H, W, C);
    for (int i = 0; i < n_in; ++i)
        if (a.up[i])
            hipLaunchKernelGGL(bifpn_bwd_up_kernel, dim3(ew_grid((long)B * (H / 2) * (W / 2) * (C / 4))), dim3(256), 0, s, dout, a.dsrc[i], w_dev, n_in, eps, i, B,
                               H / 2, W / 2, C);
    hipLaunchKernelGGL(bifpn_bwd_weight_kernel, dim3(1), dim3(64), 0, s, workspace, nblk, w_dev, n_in, eps, dw_accumulate);
    return launch_status("somi_bifpn_bwd_nhwc_f32");
}

extern "C" size_t somi_dwconv3x3_bwd_workspace_floats(int B, int W, int C) {
    return (size_t)cdiv((long)B * W * (C / 4), 256) * 10 * C;
}

extern "C" int somi_dwconv3x3_bwd_nhwc_f32(const float *dy, const float *x, const float *w, float *dx, const float *dx_accumulate, float *dw_accumulate,
                                           float *dbias_accumulate, float *workspace, int B, int H, int W, int C, somi_stream_t stream) {
    SOMI_REQUIRE(dy && x && w && dx && dw_accumulate && workspace && B > 0 && H > 0 && W > 0 && C % 4 == 0 && aligned16(dy) && aligned16(x) &&
                     aligned16(dx) && aligned16(w), SOMI_EINVAL, "dwconv bwd: bad arguments");
    SOMI_REQUIRE(C / 4 <= 256 && 256 % (C / 4) == 0, SOMI_ENOTIMPL, "dwconv bwd: C/4 (%d) must divide 256", C / 4);
    hipStream_t s = (hipStream_t)stream;
    launch_dwconv3x3(true, dy, w, nullptr, nullptr, nullptr, dx_accumulate, dx, B, H, W, C, SOMI_ACT_NONE, s);
    const int nchunk = (int)cdiv((long)B * W * (C / 4), 256);
    hipLaunchKernelGGL(dwconv3x3_bwd_weight_kernel, dim3(nchunk), dim3(256), 0, s, dy, x, workspace, B, H, W, C);
    hipLaunchKernelGGL(dwconv3x3_bwd_weight_final, dim3(cdiv(10L * C, 16)), dim3(256), 0, s, workspace, nchunk, C, dw_accumulate, dbias_accumulate);
    return launch_status("somi_dwconv3x3_bwd_nhwc_f32");
}

extern "C" int somi_scale_channels_bwd_nhwc_f32(const float *dout, const float *x, const float *s, float *dx, float *ds, float *workspace, int B,
                                                int HW, int C, somi_stream_t stream) {
    SOMI_REQUIRE(dout && x && s && dx && ds && workspace && B > 0 && HW > 0 && C % 4 == 0 && aligned16(dout) && aligned16(x) && aligned16(dx) &&
                     aligned16(s), SOMI_EINVAL, "scale channels bwd: bad arguments");
    const int nchunk = somi_img_nchunk(HW);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(scale_channels_bwd_kernel, dim3(nchunk, B), dim3(256), 0, st, dout, x, s, dx, workspace, HW, C, nchunk);
    hipLaunchKernelGGL(img_partial_sum_kernel, dim3(cdiv((long)B * C, 16)), dim3(256), 0, st, workspace, nchunk, C, B, ds);
    return launch_status("somi_scale_channels_bwd_nhwc_f32");
}
