// Training-mode kernels around the convolutions: batch-norm statistics, the fused normalise+activation pass and its
// backward (reduce + apply), activation derivatives.  All NHWC fp32, channel slices via (cs, coff), 16 B per lane.
//
// Conv block in training mode (models/common.py:64-66, nn.BatchNorm2d with batch statistics, eps 1e-3, momentum 0.03 set by
// utils/torch_utils.py:165-174):      y = conv(x);  z = act(y * scale[c] + shift[c])   with scale = gamma*rstd, shift = beta - mean*scale
// SEAM stage (models/common.py:8454-8466): g = act(u); z = g * scale[c] + shift[c]      (activation BEFORE the norm)
// Backward of the norm with batch statistics over N = B*H*W samples per channel, for its input v and upstream gradient dv_out:
//   d v = scale * ( d - mean(d) - vhat * mean(d * vhat) ),   vhat = (v - mean) * rstd
// which is affine per channel in (d, v):  dv = A[c]*d + Bc[c]*v + Cc[c].  The reduce kernel produces S1 = sum d, S2 = sum d*v,
// a tiny finalize turns them into (A, Bc, Cc, dgamma, dbeta), the apply kernel writes dv (fused with the activation derivative).
#include "common.h"

namespace somi {
typedef int i32x4_ __attribute__((ext_vector_type(4)));

typedef float f32x4 __attribute__((ext_vector_type(4)));

// pixels per stage-1 workgroup: about 1024 chunks for large tensors (stage 2 walks the chunks), never fewer than 32 pixels (the 20x20 maps of
// a batch of 32 are 12800 pixels: 128-pixel chunks left 156 of the 256 CUs without a workgroup)
static inline int red_chunk(long npix) {
    long c = (npix + 1023) / 1024;
    c = (c + 31) / 32 * 32;
    return (int)(c < 32 ? 32 : (c > 4096 ? 4096 : c));
}

__device__ __forceinline__ float act_fwd(float u, int act) { return apply_act_rt(u, act); }
__device__ __forceinline__ float act_grad(float u, int act) {
    switch (act) {
        case SOMI_ACT_SILU: { const float s = 1.f / (1.f + expf(-u)); return s * (1.f + u * (1.f - s)); }
        case SOMI_ACT_GELU: return 0.5f * (1.f + erff(u * 0.70710678118654752440f)) + u * 0.39894228040143267794f * expf(-0.5f * u * u);
        case SOMI_ACT_RELU: return u > 0.f ? 1.f : 0.f;
        case SOMI_ACT_SIGMOID: { const float s = 1.f / (1.f + expf(-u)); return s * (1.f - s); }
        default: return 1.f;
    }
}

// ---- element-wise sweeps: a thread keeps ONE channel quad for the whole launch (the host makes the thread count a multiple of C/4), so
// the per-channel vectors are loaded once and there is no division in the loop; it walks its pixels EW_U at a time with all the loads of a
// group issued before the first use (~64-128 KiB of HBM reads in flight per CU: what it takes to stream at 5+ TB/s, MI355X guide).
constexpr int EW_U = 4;
struct EwMap { int c; long p, pstep; };
__device__ __forceinline__ EwMap ew_map(int C4) {
    const unsigned nthreads = gridDim.x * 256u, t = blockIdx.x * 256u + threadIdx.x;
    return {(int)(t % (unsigned)C4) * 4, (long)(t / (unsigned)C4), (long)(nthreads / (unsigned)C4)};
}

// ---- generic two-stage per-channel reduction over pixels: each workgroup reduces RED_CHUNK pixels for all channels.
// F(p, c4) returns up to two float4 terms for pixel p, channel quad c4.
template <typename F>
__device__ __forceinline__ void chunk_reduce2(long npix, int C, float *part1, float *part2, int RED_CHUNK, F f) {
    __shared__ f32x4 l1[256], l2[256];
    const int chunk = blockIdx.x;
    const int C4 = C >> 2;
    const long p0 = (long)chunk * RED_CHUNK, p1 = min(p0 + RED_CHUNK, npix);
    for (int cq0 = 0; cq0 < C4; cq0 += 256) {
        const int ncq = min(256, C4 - cq0);
        const int rows_par = 256 / ncq;
        const int cq = threadIdx.x % ncq, rr = threadIdx.x / ncq;
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
        if (rr < rows_par) {
            // four pixels per trip: the loads of all four go out before the first sum (the order of the sums is unchanged)
#pragma unroll 4
            for (long p = p0 + rr; p < p1; p += rows_par) f(p, (cq0 + cq) * 4, s1, s2);
        }
        l1[threadIdx.x] = s1;
        l2[threadIdx.x] = s2;
        __syncthreads();
        if (threadIdx.x < ncq) {
            for (int r2 = 1; r2 < rows_par; ++r2) { s1 += l1[r2 * ncq + cq]; s2 += l2[r2 * ncq + cq]; }
            const long o = (long)chunk * C + (cq0 + cq) * 4;
            *reinterpret_cast<f32x4 *>(part1 + o) = s1;
            *reinterpret_cast<f32x4 *>(part2 + o) = s2;
        }
        __syncthreads();
    }
}

// stage-2 helper: 256 threads = 4 channels x 64 chunk groups.  The walk over <= 1024 chunk partials is a chain of memory latencies (the rows were
// written by the kernel just before: every trip is an L2 miss), so it is made SHORT - a thread sees at most 16 rows - and each trip has 16 loads
// in flight; 16 channels x 16 groups took 17-18 us per launch, 282 launches per step.  Returns (for group 0 lanes) the sums over all chunks of
// p1 / p2, combined in a fixed order
constexpr int S2_CH = 4, S2_GRP = 256 / S2_CH;
__device__ __forceinline__ bool stage2_sums(const float *__restrict__ p1, const float *__restrict__ p2, int nchunk, int C, int &c, double &s1, double &s2) {
    __shared__ double l1[256], l2[256];
    const int cl = threadIdx.x % S2_CH, grp = threadIdx.x / S2_CH;
    c = blockIdx.x * S2_CH + cl;
    double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
    if (c < C) {
        constexpr int S2_DEEP = 8;                                     // rows requested together (rows past the end contribute +0.0: exact)
        for (int k = grp; k < nchunk; k += S2_DEEP * S2_GRP) {
            float v[S2_DEEP], w[S2_DEEP];
#pragma unroll
            for (int j = 0; j < S2_DEEP; ++j) {
                const int r = k + j * S2_GRP;
                const bool ok = r < nchunk;
                v[j] = ok ? p1[(long)r * C + c] : 0.f;
                w[j] = (ok && p2) ? p2[(long)r * C + c] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < S2_DEEP; j += 2) { a0 += v[j]; a1 += v[j + 1]; b0 += w[j]; b1 += w[j + 1]; }
        }
    }
    l1[threadIdx.x] = a0 + a1;
    l2[threadIdx.x] = b0 + b1;
    __syncthreads();
    if (grp != 0 || c >= C) return false;
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int g = 0; g < S2_GRP; ++g) { t1 += l1[g * S2_CH + cl]; t2 += l2[g * S2_CH + cl]; }
    s1 = t1;
    s2 = t2;
    return true;
}

// ------------------------------------------------------------------------------------------------ BN statistics
// Sums are taken around a per-channel pivot (the running mean before this step's update, when there is one): the variance
// sum_sq/n - mean^2 then does not cancel catastrophically for channels whose mean is large against their spread.
// act != NONE: the statistics of act(x) - the act-then-norm stages (SEAM) need no materialised act(x): the normalise pass applies act again
__global__ __launch_bounds__(256) void bn_stats_stage1(const float *__restrict__ x, int cs, int coff, long npix, int C,
                                                       const float *__restrict__ pivot, float *__restrict__ p1, float *__restrict__ p2,
                                                       int chunk, int act) {
    chunk_reduce2(npix, C, p1, p2, chunk, [&](long p, int c, f32x4 &s1, f32x4 &s2) {
        f32x4 v = *reinterpret_cast<const f32x4 *>(x + p * cs + coff + c);
        if (act != SOMI_ACT_NONE) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = act_fwd(v[e], act);
        }
        if (pivot) v -= f32x4{pivot[c], pivot[c + 1], pivot[c + 2], pivot[c + 3]};
        s1 += v;
        s2 += v * v;
    });
}
// mean, biased var; scale/shift for the normalise pass; running statistics update (unbiased var, momentum), like nn.BatchNorm2d
__global__ __launch_bounds__(256) void bn_stats_stage2(const float *__restrict__ p1, const float *__restrict__ p2, int nchunk, int C,
                                                       long npix, float eps, float momentum, const float *__restrict__ gamma,
                                                       const float *__restrict__ beta, float *__restrict__ mean, float *__restrict__ rstd,
                                                       float *__restrict__ scale, float *__restrict__ shift, float *running_mean,
                                                       float *running_var) {
    int c;
    double s, q;
    if (!stage2_sums(p1, p2, nchunk, C, c, s, q)) return;
    const double dm = s / (double)npix;                           // mean relative to the pivot (the old running mean)
    const double m = (running_mean ? (double)running_mean[c] : 0.0) + dm;
    double var = q / (double)npix - dm * dm;
    if (var < 0.0) var = 0.0;
    const float rs = (float)(1.0 / sqrt(var + (double)eps));
    mean[c] = (float)m;
    rstd[c] = rs;
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    scale[c] = g * rs;
    shift[c] = b - (float)m * g * rs;
    if (running_mean) {
        const double unb = npix > 1 ? var * (double)npix / (double)(npix - 1) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
}

// partial rows left by a convolution's epilogue (one per row tile and wave row, up to tens of thousands): fold them to <= 1024 rows
// so that stage 2 keeps its short chains.  grid (rows_out, ceil(C/256)); fixed summation order.
__global__ __launch_bounds__(256) void rows_fold_kernel(const float *__restrict__ p1, const float *__restrict__ p2, int rows, int C, int per,
                                                        float *__restrict__ o1, float *__restrict__ o2) {
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= C) return;
    const int r0 = blockIdx.x * per, r1 = min(r0 + per, rows);
    float a = 0.f, b = 0.f;
    for (int r = r0; r < r1; ++r) { a += p1[(size_t)r * C + c]; b += p2[(size_t)r * C + c]; }
    o1[(size_t)blockIdx.x * C + c] = a;
    o2[(size_t)blockIdx.x * C + c] = b;
}

// ---- SyncBatchNorm (train.py:165-167 --sync-bn): the same two reductions with the exchange between ranks in the middle.  A rank
// folds its partial sums to doubles [2][C] + its pixel count; the caller all-gathers those records; every rank then adds the
// records in rank order (the same order everywhere: identical statistics on every rank, bit for bit) and finishes as above.
// central: the forward statistics leave as a PIVOT-FREE record {mean_r, M2_r = sum (x - mean_r)^2, n_r} (the partial sums were taken around
// this rank's own pivot, its running mean: records of ranks whose pivots differ could not simply be added - ADVICE r2)
__global__ __launch_bounds__(256) void sums_fold_f64_kernel(const float *__restrict__ p1, const float *__restrict__ p2, int nchunk, int C, long npix,
                                                            double *__restrict__ sums, const float *__restrict__ pivot, int central) {
    int c;
    double s1, s2;
    if (blockIdx.x == 0 && threadIdx.x == 0) sums[2 * (size_t)C] = (double)npix;
    if (!stage2_sums(p1, p2, nchunk, C, c, s1, s2)) return;
    if (central) {
        const double dm = s1 / (double)npix;
        double m2 = s2 - s1 * dm;
        sums[c] = (pivot ? (double)pivot[c] : 0.0) + dm;
        sums[(size_t)C + c] = m2 < 0.0 ? 0.0 : m2;
    } else {
        sums[c] = s1;
        sums[(size_t)C + c] = s2;
    }
}
__global__ __launch_bounds__(256) void bn_stats_from_sums_kernel(const double *__restrict__ all, int nranks, int C, float eps, float momentum,
                                                                 const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                 float *__restrict__ mean, float *__restrict__ rstd, float *__restrict__ scale,
                                                                 float *__restrict__ shift, float *running_mean, float *running_var) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    // records {mean_r, M2_r, n_r} combined in rank order (Chan et al.): the same doubles in the same order on every rank
    const size_t rec = 2 * (size_t)C + 1;
    double cnt = 0.0, msum = 0.0;
    for (int r = 0; r < nranks; ++r) { const double nr = all[r * rec + 2 * (size_t)C]; cnt += nr; msum += nr * all[r * rec + c]; }
    const double m = msum / cnt;
    double m2 = 0.0;
    for (int r = 0; r < nranks; ++r) {
        const double nr = all[r * rec + 2 * (size_t)C], d = all[r * rec + c] - m;
        m2 += all[r * rec + C + c] + nr * d * d;
    }
    double var = m2 / cnt;
    if (var < 0.0) var = 0.0;
    const float rs = (float)(1.0 / sqrt(var + (double)eps));
    mean[c] = (float)m;
    rstd[c] = rs;
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    scale[c] = g * rs;
    shift[c] = b - (float)m * g * rs;
    if (running_mean) {
        const double unb = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
}
// coefficients of the input gradient from the sums of ALL ranks (torch.nn.SyncBatchNorm's backward: the two sums are all-reduced, the
// parameter gradients stay local and are summed by the gradient exchange like every other parameter)
__global__ __launch_bounds__(256) void bn_act_bwd_sync_stage2(const double *__restrict__ local, const double *__restrict__ all, int nranks, int C,
                                                              const float *__restrict__ mean, const float *__restrict__ rstd,
                                                              const float *__restrict__ scale, float *__restrict__ coefA,
                                                              float *__restrict__ coefB, float *__restrict__ coefC, float *dgamma, float *dbeta) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const size_t rec = 2 * (size_t)C + 1;
    double s1 = 0.0, s2 = 0.0, cnt = 0.0;
    for (int r = 0; r < nranks; ++r) { s1 += all[r * rec + c]; s2 += all[r * rec + C + c]; cnt += all[r * rec + 2 * (size_t)C]; }
    const double rs = rstd[c], sc = scale[c];
    const double D = rs * s2;
    coefA[c] = (float)sc;
    coefB[c] = (float)(-sc * rs * D / cnt);
    coefC[c] = (float)(-sc * s1 / cnt);
    if (dgamma) dgamma[c] += (float)(rs * local[(size_t)C + c]);
    if (dbeta) dbeta[c] += (float)local[c];
}

// ------------------------------------------------------------------------------------------------ affine + activation
// order 0: z = act(x*scale + shift) ; order 1: z = act(x)*scale + shift        (x, z: channel slices; in place allowed)
// SILU0: the conv block's case (SiLU after the norm) compiled on its own - the run-time activation switch costs registers (occupancy)
template <bool SILU0>
__global__ __launch_bounds__(256) void chan_affine_act_kernel(const float *x, int x_cs, int x_coff, const float *__restrict__ scale,
                                                              const float *__restrict__ shift, int act_rt, int order_rt, float *z, int z_cs,
                                                              int z_coff, long npix, int C, const float *res, int res_cs, int res_coff) {
    const int act = SILU0 ? (int)SOMI_ACT_SILU : act_rt, order = SILU0 ? 0 : order_rt;
    const EwMap m = ew_map(C >> 2);
    const int c = m.c;
    const f32x4 sc = *reinterpret_cast<const f32x4 *>(scale + c), sh = *reinterpret_cast<const f32x4 *>(shift + c);
    x += x_coff + c;
    z += z_coff + c;
    if (res) res += res_coff + c;
    auto one = [&](f32x4 v) {
        if (order == 0) {
            v = v * sc + sh;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = act_fwd(v[e], act);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = act_fwd(v[e], act);
            v = v * sc + sh;
        }
        return v;
    };
    long p = m.p;
    for (; p + (EW_U - 1) * m.pstep < npix; p += EW_U * m.pstep) {
        f32x4 v[EW_U], r[EW_U];
#pragma unroll
        for (int u = 0; u < EW_U; ++u) v[u] = *reinterpret_cast<const f32x4 *>(x + (p + u * m.pstep) * x_cs);
        if (res) {
#pragma unroll
            for (int u = 0; u < EW_U; ++u) r[u] = *reinterpret_cast<const f32x4 *>(res + (p + u * m.pstep) * res_cs);
        }
#pragma unroll
        for (int u = 0; u < EW_U; ++u) {
            f32x4 o = one(v[u]);
            if (res) o += r[u];
            *reinterpret_cast<f32x4 *>(z + (p + u * m.pstep) * z_cs) = o;
        }
    }
    for (; p < npix; p += m.pstep) {
        f32x4 o = one(*reinterpret_cast<const f32x4 *>(x + p * x_cs));
        if (res) o += *reinterpret_cast<const f32x4 *>(res + p * res_cs);
        *reinterpret_cast<f32x4 *>(z + p * z_cs) = o;
    }
}

// ------------------------------------------------------------------------------------------------ backward: reduce
// order 0 (norm then act): d = dz * act'(x*scale+shift), v = x.      S1 = sum d, S2 = sum d*v
// order 1 (act then norm): d = dz,                       v = act(x). S1 = sum d, S2 = sum d*v
template <bool SILU0>
__global__ __launch_bounds__(256) void bn_act_bwd_stage1(const float *__restrict__ dz, int dz_cs, int dz_coff, const float *__restrict__ x,
                                                         int x_cs, int x_coff, const float *__restrict__ scale,
                                                         const float *__restrict__ shift, const float *__restrict__ mean, int act_rt,
                                                         int order_rt, long npix, int C, float *__restrict__ p1, float *__restrict__ p2,
                                                         int chunk) {
    const int act = SILU0 ? (int)SOMI_ACT_SILU : act_rt, order = SILU0 ? 0 : order_rt;
    // S2 is accumulated as sum d*(v - mean) directly (not sum d*v minus mean * sum d afterwards: that difference cancels)
    chunk_reduce2(npix, C, p1, p2, chunk, [&](long p, int c, f32x4 &s1, f32x4 &s2) {
        const f32x4 g = *reinterpret_cast<const f32x4 *>(dz + p * dz_cs + dz_coff + c);
        const f32x4 v = *reinterpret_cast<const f32x4 *>(x + p * x_cs + x_coff + c);
        const f32x4 mu = {mean[c], mean[c + 1], mean[c + 2], mean[c + 3]};
        f32x4 d, w;
        if (order == 0) {
            const f32x4 u = v * *reinterpret_cast<const f32x4 *>(scale + c) + *reinterpret_cast<const f32x4 *>(shift + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = g[e] * act_grad(u[e], act);
            w = v;
        } else {
            d = g;
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = act_fwd(v[e], act);
        }
        s1 += d;
        s2 += d * (w - mu);
    });
}
// finalize: per channel coefficients of dv = A*d + Bc*(v - mean) + Cc, and the parameter gradients (accumulated into dgamma / dbeta)
//   batch statistics (train):  D = rstd*S2 = sum d*vhat  (S2 = sum d*(v - mean));  A = scale, Bc = -scale*rstd*D/N, Cc = -scale*S1/N
//   frozen statistics (eval):  A = scale, Bc = Cc = 0
// The centred form matters: expanded to Bc*v + (Cc + scale*rstd*D*mean/N) the two terms cancel to |v - mean| / |mean| of their size, and that
// rounding lands in exactly the property the layers above rely on (sum over pixels of dv == 0): the gradients of parameters that are
// sums over all pixels (the spatial-attention conv) came out 5-8x noisier than the fp32 CPU path's at 1280x1280.
__global__ __launch_bounds__(256) void bn_act_bwd_stage2(const float *__restrict__ p1, const float *__restrict__ p2, int nchunk, int C,
                                                         long npix, const float *__restrict__ mean, const float *__restrict__ rstd,
                                                         const float *__restrict__ scale, int batch_stats, float *__restrict__ coefA,
                                                         float *__restrict__ coefB, float *__restrict__ coefC, float *dgamma, float *dbeta) {
    int c;
    double s1, s2;
    if (!stage2_sums(p1, p2, nchunk, C, c, s1, s2)) return;
    const double rs = rstd[c], sc = scale[c];
    const double D = rs * s2;
    coefA[c] = (float)sc;
    if (batch_stats) {
        coefB[c] = (float)(-sc * rs * D / (double)npix);
        coefC[c] = (float)(-sc * s1 / (double)npix);
    } else {
        coefB[c] = 0.f;
        coefC[c] = 0.f;
    }
    if (dgamma) dgamma[c] += (float)D;
    if (dbeta) dbeta[c] += (float)s1;
}
// apply: order 0: dx = A*d + Bc*(x - mean) + Cc with d = dz*act'(x*scale+shift)
//        order 1: dx = (A*dz + Bc*(act(x) - mean) + Cc) * act'(x)
template <bool SILU0>
__global__ __launch_bounds__(256) void bn_act_bwd_apply(const float *dz, int dz_cs, int dz_coff, const float *x, int x_cs, int x_coff,
                                                        const float *__restrict__ scale, const float *__restrict__ shift,
                                                        const float *__restrict__ mean, const float *__restrict__ coefA,
                                                        const float *__restrict__ coefB, const float *__restrict__ coefC, int act_rt,
                                                        int order_rt, float *dx, int dx_cs, int dx_coff, long npix, int C) {
    const int act = SILU0 ? (int)SOMI_ACT_SILU : act_rt, order = SILU0 ? 0 : order_rt;
    const EwMap m = ew_map(C >> 2);
    const int c = m.c;
    const f32x4 A = *reinterpret_cast<const f32x4 *>(coefA + c), Bc = *reinterpret_cast<const f32x4 *>(coefB + c),
                Cc = *reinterpret_cast<const f32x4 *>(coefC + c), M = *reinterpret_cast<const f32x4 *>(mean + c),
                sc = *reinterpret_cast<const f32x4 *>(scale + c), sh = *reinterpret_cast<const f32x4 *>(shift + c);
    dz += dz_coff + c;
    x += x_coff + c;
    dx += dx_coff + c;
    auto one = [&](const f32x4 g, const f32x4 v) {
        f32x4 r;
        if (order == 0) {
            const f32x4 u = v * sc + sh;
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = A[e] * (g[e] * act_grad(u[e], act)) + Bc[e] * (v[e] - M[e]) + Cc[e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = (A[e] * g[e] + Bc[e] * (act_fwd(v[e], act) - M[e]) + Cc[e]) * act_grad(v[e], act);
        }
        return r;
    };
    long p = m.p;
    for (; p + (EW_U - 1) * m.pstep < npix; p += EW_U * m.pstep) {
        f32x4 g[EW_U], v[EW_U];
#pragma unroll
        for (int u = 0; u < EW_U; ++u) {
            g[u] = *reinterpret_cast<const f32x4 *>(dz + (p + u * m.pstep) * dz_cs);
            v[u] = *reinterpret_cast<const f32x4 *>(x + (p + u * m.pstep) * x_cs);
        }
#pragma unroll
        for (int u = 0; u < EW_U; ++u) *reinterpret_cast<f32x4 *>(dx + (p + u * m.pstep) * dx_cs) = one(g[u], v[u]);
    }
    for (; p < npix; p += m.pstep)
        *reinterpret_cast<f32x4 *>(dx + p * dx_cs) =
            one(*reinterpret_cast<const f32x4 *>(dz + p * dz_cs), *reinterpret_cast<const f32x4 *>(x + p * x_cs));
}

// ------------------------------------------------------------------------------------------------ backward with CBAM's pooled gradients folded in
// The conv in front of a CBAM attention pair receives  dz_eff[b,p,c] = dz[b,p,c] + davg[b,c] / HW + [p == amaxp[b,c]] * dmax[b,c]:  the gradient
// through the two multiplications (dz, written by cbam_bwd_chan) plus the global average / max pools' (models/common.py:339-358).  Round 3 added
// the pooled part in a pass of its own (pool_bwd_add_kernel: read + write of the whole tensor, 2.3 ms per step) and found that folding it into
// these kernels with their flat pixel walk cost more than the pass (a division and an index load per element).  Here both kernels are IMAGE-
// ALIGNED instead - blockIdx.y = image, a thread keeps one channel quad of that image - so the pooled terms are four registers loaded once and
// one compare per element; dz itself is only read.  Same sums as the plain kernels up to their order (chunks of one image instead of chunks
// of the flat pixel range): deterministic, run-to-run bit-identical.
template <bool SILU0>
__global__ __launch_bounds__(256) void bn_act_bwd_stage1_pooled(const float *__restrict__ dz, int dz_cs, int dz_coff, const float *__restrict__ x,
                                                                int x_cs, int x_coff, const float *__restrict__ scale,
                                                                const float *__restrict__ shift, const float *__restrict__ mean, int act_rt,
                                                                int order_rt, int HW, int C, float *__restrict__ p1, float *__restrict__ p2, int chunk,
                                                                const float *__restrict__ davg, const float *__restrict__ dmax,
                                                                const int *__restrict__ amaxp) {
    __shared__ f32x4 l1[256], l2[256];
    const int act = SILU0 ? (int)SOMI_ACT_SILU : act_rt, order = SILU0 ? 0 : order_rt;
    const int b = blockIdx.y, C4 = C >> 2;
    const int q0 = blockIdx.x * chunk, q1 = min(q0 + chunk, HW);
    const long row = (long)b * gridDim.x + blockIdx.x;
    const float inv_hw = 1.f / (float)HW;
    for (int cq0 = 0; cq0 < C4; cq0 += 256) {
        const int ncq = min(256, C4 - cq0);
        const int rows_par = 256 / ncq;
        const int cq = threadIdx.x % ncq, rr = threadIdx.x / ncq;
        const int c = (cq0 + cq) * 4;
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
        if (rr < rows_par) {
            // dz == nullptr: the incoming gradient is the pooled part alone (SEAM: only a global average pool reads the tensor); dmax ==
            // nullptr: no max-pool term
            const f32x4 kavg = *reinterpret_cast<const f32x4 *>(davg + (long)b * C + c) * inv_hw;
            const f32x4 kmax = dmax ? *reinterpret_cast<const f32x4 *>(dmax + (long)b * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
            const i32x4_ am = dmax ? *reinterpret_cast<const i32x4_ *>(amaxp + (long)b * C + c) : i32x4_{-1, -1, -1, -1};
            const f32x4 mu = *reinterpret_cast<const f32x4 *>(mean + c), sc = *reinterpret_cast<const f32x4 *>(scale + c),
                        sh = *reinterpret_cast<const f32x4 *>(shift + c);
#pragma unroll 4
            for (int pl = q0 + rr; pl < q1; pl += rows_par) {            // four pixels' loads go out before the first sum
                const long p = (long)b * HW + pl;
                f32x4 g = kavg;
                if (dz) g += *reinterpret_cast<const f32x4 *>(dz + p * dz_cs + dz_coff + c);
                const f32x4 v = *reinterpret_cast<const f32x4 *>(x + p * x_cs + x_coff + c);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (am[e] == pl) g[e] += kmax[e];
                f32x4 d, w;
                if (order == 0) {
                    const f32x4 u = v * sc + sh;
#pragma unroll
                    for (int e = 0; e < 4; ++e) d[e] = g[e] * act_grad(u[e], act);
                    w = v;
                } else {
                    d = g;
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = act_fwd(v[e], act);
                }
                s1 += d;
                s2 += d * (w - mu);
            }
        }
        l1[threadIdx.x] = s1;
        l2[threadIdx.x] = s2;
        __syncthreads();
        if (threadIdx.x < ncq) {
            for (int r2 = 1; r2 < rows_par; ++r2) { s1 += l1[r2 * ncq + cq]; s2 += l2[r2 * ncq + cq]; }
            *reinterpret_cast<f32x4 *>(p1 + row * C + c) = s1;
            *reinterpret_cast<f32x4 *>(p2 + row * C + c) = s2;
        }
        __syncthreads();
    }
}
// grid (gx, B), gx * 256 a multiple of C / 4 (ew_grid_img): a thread keeps one channel quad of image blockIdx.y and walks its pixels EW_U at a time
template <bool SILU0>
__global__ __launch_bounds__(256) void bn_act_bwd_apply_pooled(const float *dz, int dz_cs, int dz_coff, const float *x, int x_cs, int x_coff,
                                                               const float *__restrict__ scale, const float *__restrict__ shift,
                                                               const float *__restrict__ mean, const float *__restrict__ coefA,
                                                               const float *__restrict__ coefB, const float *__restrict__ coefC, int act_rt,
                                                               int order_rt, float *dx, int dx_cs, int dx_coff, int HW, int C,
                                                               const float *__restrict__ davg, const float *__restrict__ dmax,
                                                               const int *__restrict__ amaxp) {
    const int act = SILU0 ? (int)SOMI_ACT_SILU : act_rt, order = SILU0 ? 0 : order_rt;
    const unsigned C4 = (unsigned)C >> 2, nthreads = gridDim.x * 256u, t = blockIdx.x * 256u + threadIdx.x;
    const int c = (int)(t % C4) * 4, b = blockIdx.y, pstep = (int)(nthreads / C4);
    const f32x4 A = *reinterpret_cast<const f32x4 *>(coefA + c), Bc = *reinterpret_cast<const f32x4 *>(coefB + c),
                Cc = *reinterpret_cast<const f32x4 *>(coefC + c), M = *reinterpret_cast<const f32x4 *>(mean + c),
                sc = *reinterpret_cast<const f32x4 *>(scale + c), sh = *reinterpret_cast<const f32x4 *>(shift + c);
    const f32x4 kavg = *reinterpret_cast<const f32x4 *>(davg + (long)b * C + c) * (1.f / (float)HW);
    const f32x4 kmax = dmax ? *reinterpret_cast<const f32x4 *>(dmax + (long)b * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    const i32x4_ am = dmax ? *reinterpret_cast<const i32x4_ *>(amaxp + (long)b * C + c) : i32x4_{-1, -1, -1, -1};
    const bool has_dz = dz != nullptr;
    if (has_dz) dz += (long)b * HW * dz_cs + dz_coff + c;
    x += (long)b * HW * x_cs + x_coff + c;
    dx += (long)b * HW * dx_cs + dx_coff + c;
    auto one = [&](f32x4 g, const f32x4 v, int pl) {
        g += kavg;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (am[e] == pl) g[e] += kmax[e];
        f32x4 r;
        if (order == 0) {
            const f32x4 u = v * sc + sh;
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = A[e] * (g[e] * act_grad(u[e], act)) + Bc[e] * (v[e] - M[e]) + Cc[e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = (A[e] * g[e] + Bc[e] * (act_fwd(v[e], act) - M[e]) + Cc[e]) * act_grad(v[e], act);
        }
        return r;
    };
    int p = (int)(t / C4);
    for (; p + (EW_U - 1) * pstep < HW; p += EW_U * pstep) {
        f32x4 g[EW_U], v[EW_U];
#pragma unroll
        for (int u = 0; u < EW_U; ++u) {
            g[u] = has_dz ? *reinterpret_cast<const f32x4 *>(dz + (long)(p + u * pstep) * dz_cs) : f32x4{0.f, 0.f, 0.f, 0.f};
            v[u] = *reinterpret_cast<const f32x4 *>(x + (long)(p + u * pstep) * x_cs);
        }
#pragma unroll
        for (int u = 0; u < EW_U; ++u) *reinterpret_cast<f32x4 *>(dx + (long)(p + u * pstep) * dx_cs) = one(g[u], v[u], p + u * pstep);
    }
    for (; p < HW; p += pstep)
        *reinterpret_cast<f32x4 *>(dx + (long)p * dx_cs) =
            one(has_dz ? *reinterpret_cast<const f32x4 *>(dz + (long)p * dz_cs) : f32x4{0.f, 0.f, 0.f, 0.f}, *reinterpret_cast<const f32x4 *>(x + (long)p * x_cs), p);
}
// ---- CBAM bottleneck: step C of train_blocks.hip (the gradient through x*ca*sa and the spatial branch) INSIDE the first conv's BatchNorm + SiLU
// backward.  Round 4's sequence was C (read d(t*ca*sa), t; write dt) -> pooled stage 1 (read dt, y) -> pooled apply (read dt, y; write dy): 8 tensor
// passes.  dt is a per-element function of d, t and a few per-pixel / per-(image, channel) scalars, and t = silu(scale*y + shift) is a function of
// the y the BatchNorm backward reads anyway - so both BatchNorm kernels take d and y and rebuild dt in registers: 5 passes, no dt tensor, no read of t.
// The pooled gradients (davg, dmax) of the channel attention depend on dca = sum_p dt1*t, which only this reduction produces; the batch sums are
// therefore left SEPARABLE per (image chunk, channel): P = sum dt*f, Q = sum f, F = f at the arg-max pixel (f = silu'(u)), each also times (y - mean),
// and the finalize adds  P + davg/HW * Q + dmax * F  per row once the attention MLP's backward has run.
struct CbamBnArgs {
    const float *d;  int d_cs, d_coff;          // gradient w.r.t. t*ca*sa (cv2's input gradient)
    const float *y;  int y_cs, y_coff;          // cv1's convolution output (before BatchNorm)
    const float *scale, *shift, *mean;          // BatchNorm as an affine map + the batch mean
    const float *ca, *sa, *dstats;              // (B,C), (B,HW), (B,HW,2)
    const int *amaxc, *amaxp;                   // (B,HW) arg-max channel of ca*t per pixel; (B,C) first pixel of t's spatial maximum
    int HW, C;
};
struct CbamDt {                                 // what one element contributes
    f32x4 dt, f, t, g1;
};
__device__ __forceinline__ CbamDt cbam_dt(const f32x4 g, const f32x4 v, const f32x4 sc, const f32x4 sh, const f32x4 cav, float sav, float2 ds, int am,
                                          int c, float inv_c) {
    CbamDt r;
    const f32x4 u = v * sc + sh;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float den = 1.0f + expf(-u[e]), s = 1.0f / den;
        r.t[e] = u[e] / den;                                             // silu(u), as the forward pass wrote it
        r.f[e] = s * (1.f + u[e] * (1.f - s));                           // silu'(u)
    }
    r.g1 = g * sav + ds.x * inv_c;                                       // through *sa, plus the channel-mean branch of the spatial attention
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (c + e == am) r.g1[e] += ds.y;                                // ... and its channel-max branch
    r.dt = r.g1 * cav;                                                   // through *ca
    return r;
}
// grid (nimg, B): image-aligned chunks like bn_act_bwd_stage1_pooled; part = 7 planes of rows x C: dca | P1 | P2 | Q1 | Q2 | F1 | F2
__global__ __launch_bounds__(256) void cbam_bn_bwd_reduce_kernel(CbamBnArgs a, float *__restrict__ part, int chunk) {
    __shared__ f32x4 lr[256];
    const int b = blockIdx.y, C = a.C, HW = a.HW, C4 = C >> 2;
    const int q0 = blockIdx.x * chunk, q1 = min(q0 + chunk, HW);
    const long row = (long)b * gridDim.x + blockIdx.x, plane = (long)gridDim.x * gridDim.y * C;
    const float inv_c = 1.f / (float)C;
    for (int cq0 = 0; cq0 < C4; cq0 += 256) {
        const int ncq = min(256, C4 - cq0);
        const int rows_par = 256 / ncq;
        const int cq = threadIdx.x % ncq, rr = threadIdx.x / ncq;
        const int c = (cq0 + cq) * 4;
        f32x4 acc[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (rr < rows_par) {
            const f32x4 cav = *reinterpret_cast<const f32x4 *>(a.ca + (long)b * C + c), mu = *reinterpret_cast<const f32x4 *>(a.mean + c),
                        sc = *reinterpret_cast<const f32x4 *>(a.scale + c), sh = *reinterpret_cast<const f32x4 *>(a.shift + c);
            auto one = [&](int pl, const f32x4 g, const f32x4 v, float sav, float2 ds, int am) {
                const CbamDt r = cbam_dt(g, v, sc, sh, cav, sav, ds, am, c, inv_c);
                const f32x4 xc = v - mu, df = r.dt * r.f;
                acc[0] += r.g1 * r.t;
                acc[1] += df;
                acc[2] += df * xc;
                acc[3] += r.f;
                acc[4] += r.f * xc;
            };
            int pl = q0 + rr;
            for (; pl + 3 * rows_par < q1; pl += 4 * rows_par) {         // four pixels' loads go out together; sums in pixel order
                f32x4 g[4], v[4];
                float2 ds[4];
                int am[4];
                float sav[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long p = (long)b * HW + pl + u * rows_par;
                    g[u] = *reinterpret_cast<const f32x4 *>(a.d + p * a.d_cs + a.d_coff + c);
                    v[u] = *reinterpret_cast<const f32x4 *>(a.y + p * a.y_cs + a.y_coff + c);
                    ds[u] = *reinterpret_cast<const float2 *>(a.dstats + p * 2);
                    am[u] = a.amaxc[p];
                    sav[u] = a.sa[p];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) one(pl + u * rows_par, g[u], v[u], sav[u], ds[u], am[u]);
            }
            for (; pl < q1; pl += rows_par) {
                const long p = (long)b * HW + pl;
                one(pl, *reinterpret_cast<const f32x4 *>(a.d + p * a.d_cs + a.d_coff + c), *reinterpret_cast<const f32x4 *>(a.y + p * a.y_cs + a.y_coff + c),
                    a.sa[p], *reinterpret_cast<const float2 *>(a.dstats + p * 2), a.amaxc[p]);
            }
            if (rr == 0) {                                               // F: silu'(u) and silu'(u) (y - mean) at the arg-max pixel, by the chunk that holds it
                const i32x4_ amp = *reinterpret_cast<const i32x4_ *>(a.amaxp + (long)b * C + c);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (amp[e] >= q0 && amp[e] < q1) {
                        const float v = a.y[((long)b * HW + amp[e]) * a.y_cs + a.y_coff + c + e], u = v * sc[e] + sh[e];
                        const float sg = 1.0f / (1.0f + expf(-u)), f = sg * (1.f + u * (1.f - sg));
                        acc[5][e] = f;
                        acc[6][e] = f * (v - mu[e]);
                    }
            }
        }
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            lr[threadIdx.x] = acc[j];
            __syncthreads();
            if (threadIdx.x < ncq) {
                f32x4 t = acc[j];
                for (int r2 = 1; r2 < rows_par; ++r2) t += lr[r2 * ncq + cq];
                *reinterpret_cast<f32x4 *>(part + j * plane + row * C + c) = t;
            }
            __syncthreads();
        }
    }
}
// finalize: rows of image b get that image's pooled terms, then as bn_act_bwd_stage2 (batch statistics)
__global__ __launch_bounds__(256) void cbam_bn_bwd_stage2(const float *__restrict__ part, int rows, int nimg, int C, int HW, const float *__restrict__ davg,
                                                          const float *__restrict__ dmax, const float *__restrict__ rstd,
                                                          const float *__restrict__ scale, float *__restrict__ coefA, float *__restrict__ coefB,
                                                          float *__restrict__ coefC, float *dgamma, float *dbeta) {
    __shared__ double l1[256], l2[256];
    const int cl = threadIdx.x % S2_CH, grp = threadIdx.x / S2_CH;
    const int c = blockIdx.x * S2_CH + cl;
    const long plane = (long)rows * C;
    const float inv_hw = 1.f / (float)HW;
    double a1 = 0.0, a2 = 0.0;
    if (c < C) {
        for (int k = grp; k < rows; k += 4 * S2_GRP) {
            float v[4][6], ka[4], km[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = k + j * S2_GRP;
                const bool ok = r < rows;
                const int b = ok ? r / nimg : 0;
#pragma unroll
                for (int q = 0; q < 6; ++q) v[j][q] = ok ? part[(q + 1) * plane + (long)r * C + c] : 0.f;
                ka[j] = ok ? davg[(long)b * C + c] * inv_hw : 0.f;
                km[j] = ok ? dmax[(long)b * C + c] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a1 += (double)(v[j][0] + ka[j] * v[j][2] + km[j] * v[j][4]);
                a2 += (double)(v[j][1] + ka[j] * v[j][3] + km[j] * v[j][5]);
            }
        }
    }
    l1[threadIdx.x] = a1;
    l2[threadIdx.x] = a2;
    __syncthreads();
    if (grp != 0 || c >= C) return;
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int g = 0; g < S2_GRP; ++g) { s1 += l1[g * S2_CH + cl]; s2 += l2[g * S2_CH + cl]; }
    const double rs = rstd[c], sc = scale[c], n = (double)rows / nimg * HW;
    const double D = rs * s2;
    coefA[c] = (float)sc;
    coefB[c] = (float)(-sc * rs * D / n);
    coefC[c] = (float)(-sc * s1 / n);
    if (dgamma) dgamma[c] += (float)D;
    if (dbeta) dbeta[c] += (float)s1;
}
// grid (gx, B) as bn_act_bwd_apply_pooled: dy = A * (dt + davg/HW + [p == amaxp] dmax) * silu'(u) + Bc * (y - mean) + Cc, dt rebuilt from d
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void cbam_bn_bwd_apply_kernel(CbamBnArgs a, const float *__restrict__ coefA, const float *__restrict__ coefB,
                                                                const float *__restrict__ coefC, const float *__restrict__ davg,
                                                                const float *__restrict__ dmax, float *dx, int dx_cs, int dx_coff) {
    const int C = a.C, HW = a.HW;
    const unsigned C4 = (unsigned)C >> 2, nthreads = gridDim.x * 256u, t = blockIdx.x * 256u + threadIdx.x;
    const int c = (int)(t % C4) * 4, b = blockIdx.y, pstep = (int)(nthreads / C4);
    const f32x4 A = *reinterpret_cast<const f32x4 *>(coefA + c), Bc = *reinterpret_cast<const f32x4 *>(coefB + c),
                Cc = *reinterpret_cast<const f32x4 *>(coefC + c), M = *reinterpret_cast<const f32x4 *>(a.mean + c),
                sc = *reinterpret_cast<const f32x4 *>(a.scale + c), sh = *reinterpret_cast<const f32x4 *>(a.shift + c),
                cav = *reinterpret_cast<const f32x4 *>(a.ca + (long)b * C + c);
    const f32x4 kavg = *reinterpret_cast<const f32x4 *>(davg + (long)b * C + c) * (1.f / (float)HW);
    const f32x4 kmax = *reinterpret_cast<const f32x4 *>(dmax + (long)b * C + c);
    const i32x4_ amp = *reinterpret_cast<const i32x4_ *>(a.amaxp + (long)b * C + c);
    const float inv_c = 1.f / (float)C;
    const float *d = a.d + (long)b * HW * a.d_cs + a.d_coff + c, *y = a.y + (long)b * HW * a.y_cs + a.y_coff + c;
    const float *sa = a.sa + (long)b * HW, *dst = a.dstats + (long)b * HW * 2;
    const int *amc = a.amaxc + (long)b * HW;
    dx += (long)b * HW * dx_cs + dx_coff + c;
    auto one = [&](int pl, const f32x4 g, const f32x4 v, float sav, float2 ds, int am) {
        const CbamDt r = cbam_dt(g, v, sc, sh, cav, sav, ds, am, c, inv_c);
        f32x4 dz = r.dt + kavg;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (amp[e] == pl) dz[e] += kmax[e];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = A[e] * (dz[e] * r.f[e]) + Bc[e] * (v[e] - M[e]) + Cc[e];
        return o;
    };
    int p = (int)(t / C4);
    for (; p + (EW_U - 1) * pstep < HW; p += EW_U * pstep) {
        f32x4 g[EW_U], v[EW_U];
        float2 ds[EW_U];
        int am[EW_U];
        float sav[EW_U];
#pragma unroll
        for (int u = 0; u < EW_U; ++u) {
            const int q = p + u * pstep;
            g[u] = *reinterpret_cast<const f32x4 *>(d + (long)q * a.d_cs);
            v[u] = *reinterpret_cast<const f32x4 *>(y + (long)q * a.y_cs);
            ds[u] = *reinterpret_cast<const float2 *>(dst + (long)q * 2);
            am[u] = amc[q];
            sav[u] = sa[q];
        }
#pragma unroll
        for (int u = 0; u < EW_U; ++u) *reinterpret_cast<f32x4 *>(dx + (long)(p + u * pstep) * dx_cs) = one(p + u * pstep, g[u], v[u], sav[u], ds[u], am[u]);
    }
    for (; p < HW; p += pstep)
        *reinterpret_cast<f32x4 *>(dx + (long)p * dx_cs) = one(p, *reinterpret_cast<const f32x4 *>(d + (long)p * a.d_cs), *reinterpret_cast<const f32x4 *>(y + (long)p * a.y_cs),
                                                                sa[p], *reinterpret_cast<const float2 *>(dst + (long)p * 2), amc[p]);
}
// defined in train_blocks.hip: out[b,c] = sum over the nchunk partial rows of image b (fp64, fixed order)
__global__ __launch_bounds__(256) void img_partial_sum_kernel(const float *__restrict__ part, int nchunk, int C, int B, float *__restrict__ out);

// pixels of one image per stage-1 workgroup of the pooled form: the batch's partial rows stay at <= 1024 like red_chunk's
static inline int red_chunk_img(int B, int HW) {
    const int per = B >= 1024 ? 1 : 1024 / B;
    int c = (HW + per - 1) / per;
    c = (c + 31) / 32 * 32;
    return c < 32 ? 32 : c;
}
// workgroups per image of an image-aligned sweep: one item per thread up to `cap` workgroups over the whole batch, thread count a multiple of C/4
static inline int ew_grid_img(int B, int HW, int C, int cap) {
    const int C4 = C / 4;
    int a = C4, b = 256;
    while (b) { const int t = a % b; a = b; b = t; }
    const long mult = C4 / a;
    long g = ((long)HW * C4 + 255) / 256, per = cap / (B > 0 ? B : 1);
    if (per < 1) per = 1;
    g = g < 1 ? 1 : (g > per ? per : g);
    return (int)((g + mult - 1) / mult * mult);
}

// per-channel sum over pixels of a tensor (bias gradients): out[c] += sum_p x[p,c]
__global__ __launch_bounds__(256) void chan_sum_stage1(const float *__restrict__ x, int cs, int coff, long npix, int C, float *__restrict__ p1,
                                                       float *__restrict__ p2, int chunk) {
    chunk_reduce2(npix, C, p1, p2, chunk, [&](long p, int c, f32x4 &s1, f32x4 &s2) {
        s1 += *reinterpret_cast<const f32x4 *>(x + p * cs + coff + c);
    });
}
__global__ __launch_bounds__(256) void chan_sum_stage2(const float *__restrict__ p1, int nchunk, int C, float *out) {
    int c;
    double s, unused;
    if (!stage2_sums(p1, nullptr, nchunk, C, c, s, unused)) return;
    out[c] += (float)s;
}

// out = a + b on channel slices (residual adds / gradient accumulation); out may alias a or b
__global__ __launch_bounds__(256) void add_kernel(const float *a, int a_cs, int a_coff, const float *b, int b_cs, int b_coff, float *o,
                                                  int o_cs, int o_coff, long npix, int C) {
    const EwMap m = ew_map(C >> 2);
    a += a_coff + m.c;
    b += b_coff + m.c;
    o += o_coff + m.c;
    long p = m.p;
    for (; p + (EW_U - 1) * m.pstep < npix; p += EW_U * m.pstep) {
        f32x4 va[EW_U], vb[EW_U];
#pragma unroll
        for (int u = 0; u < EW_U; ++u) {
            va[u] = *reinterpret_cast<const f32x4 *>(a + (p + u * m.pstep) * a_cs);
            vb[u] = *reinterpret_cast<const f32x4 *>(b + (p + u * m.pstep) * b_cs);
        }
#pragma unroll
        for (int u = 0; u < EW_U; ++u) *reinterpret_cast<f32x4 *>(o + (p + u * m.pstep) * o_cs) = va[u] + vb[u];
    }
    for (; p < npix; p += m.pstep)
        *reinterpret_cast<f32x4 *>(o + p * o_cs) = *reinterpret_cast<const f32x4 *>(a + p * a_cs) + *reinterpret_cast<const f32x4 *>(b + p * b_cs);
}

// grid of an element-wise sweep in the ew_map layout: enough 256-thread workgroups for one item per thread, at most `EW_WG_MAX` = 256 CUs x the workgroups of that kernel
// one CU holds (one resident round of the chip - a second, partial round would idle most CUs), rounded up to a thread count that is a multiple of C/4
static inline int ew_grid_c(long npix, int C, int EW_WG_MAX) {
    const int C4 = C / 4;
    int a = C4, b = 256;
    while (b) { const int t = a % b; a = b; b = t; }              // a = gcd(C4, 256)
    const long mult = C4 / a;                                     // workgroups per whole number of pixels
    long g = (npix * C4 + 255) / 256;
    g = g < 1 ? 1 : (g > EW_WG_MAX ? EW_WG_MAX : g);
    g = (g + mult - 1) / mult * mult;
    return (int)g;
}
// workgroups per CU (from the kernels' register counts: 512 VGPRs per SIMD lane, one wave of a 256-thread workgroup per SIMD) x 256 CUs
constexpr int WG_AFFINE = 5 * 256, WG_AFFINE_RT = 5 * 256, WG_APPLY = 5 * 256, WG_APPLY_RT = 4 * 256, WG_ADD = 7 * 256;

static void launch_affine_act(const float *x, int x_cs, int x_coff, const float *scale, const float *shift, int act, int order, float *z, int z_cs,
                              int z_coff, long npix, int C, const float *res, int res_cs, int res_coff, hipStream_t s) {
    if (act == SOMI_ACT_SILU && order == 0)
        hipLaunchKernelGGL(chan_affine_act_kernel<true>, dim3(ew_grid_c(npix, C, WG_AFFINE)), dim3(256), 0, s, x, x_cs, x_coff, scale, shift, act, order,
                           z, z_cs, z_coff, npix, C, res, res_cs, res_coff);
    else
        hipLaunchKernelGGL(chan_affine_act_kernel<false>, dim3(ew_grid_c(npix, C, WG_AFFINE_RT)), dim3(256), 0, s, x, x_cs, x_coff, scale, shift, act,
                           order, z, z_cs, z_coff, npix, C, res, res_cs, res_coff);
}
static void launch_bwd_stage1(int nchunk, const float *dz, int dz_cs, int dz_coff, const float *x, int x_cs, int x_coff, const float *scale,
                              const float *shift, const float *mean, int act, int order, long npix, int C, float *p1, float *p2, hipStream_t s) {
    if (act == SOMI_ACT_SILU && order == 0)
        hipLaunchKernelGGL(bn_act_bwd_stage1<true>, dim3(nchunk), dim3(256), 0, s, dz, dz_cs, dz_coff, x, x_cs, x_coff, scale, shift, mean, act, order,
                           npix, C, p1, p2, red_chunk(npix));
    else
        hipLaunchKernelGGL(bn_act_bwd_stage1<false>, dim3(nchunk), dim3(256), 0, s, dz, dz_cs, dz_coff, x, x_cs, x_coff, scale, shift, mean, act, order,
                           npix, C, p1, p2, red_chunk(npix));
}
static void launch_bwd_apply(const float *dz, int dz_cs, int dz_coff, const float *x, int x_cs, int x_coff, const float *scale, const float *shift,
                             const float *mean, const float *cA, const float *cB, const float *cC, int act, int order, float *dx, int dx_cs,
                             int dx_coff, long npix, int C, hipStream_t s) {
    if (act == SOMI_ACT_SILU && order == 0)
        hipLaunchKernelGGL(bn_act_bwd_apply<true>, dim3(ew_grid_c(npix, C, WG_APPLY)), dim3(256), 0, s, dz, dz_cs, dz_coff, x, x_cs, x_coff, scale, shift,
                           mean, cA, cB, cC, act, order, dx, dx_cs, dx_coff, npix, C);
    else
        hipLaunchKernelGGL(bn_act_bwd_apply<false>, dim3(ew_grid_c(npix, C, WG_APPLY_RT)), dim3(256), 0, s, dz, dz_cs, dz_coff, x, x_cs, x_coff, scale,
                           shift, mean, cA, cB, cC, act, order, dx, dx_cs, dx_coff, npix, C);
}

static inline bool slice_ok(const void *p, int cs, int coff, int C) { return p && cs % 4 == 0 && coff % 4 == 0 && coff + C <= cs && aligned16(p); }

}  // namespace somi

using namespace somi;

extern "C" int somi_red_nchunk(long npix) { const int c = red_chunk(npix); return (int)((npix + c - 1) / c); }

extern "C" int somi_bn_stats_nhwc_f32(const float *x, int x_cs, int x_coff, long npix, int C, float eps, float momentum,
                                      const float *gamma, const float *beta, float *mean, float *rstd, float *scale, float *shift,
                                      float *running_mean, float *running_var, float *workspace, somi_stream_t stream) {
    return somi_bn_stats_act_nhwc_f32(x, x_cs, x_coff, SOMI_ACT_NONE, npix, C, eps, momentum, gamma, beta, mean, rstd, scale, shift, running_mean,
                                      running_var, workspace, stream);
}

extern "C" int somi_bn_stats_act_nhwc_f32(const float *x, int x_cs, int x_coff, int act, long npix, int C, float eps, float momentum,
                                          const float *gamma, const float *beta, float *mean, float *rstd, float *scale, float *shift,
                                          float *running_mean, float *running_var, float *workspace, somi_stream_t stream) {
    SOMI_REQUIRE(slice_ok(x, x_cs, x_coff, C) && npix > 0 && C > 0 && C % 4 == 0 && mean && rstd && scale && shift && workspace && act >= 0 &&
                     act <= 4, SOMI_EINVAL, "bn stats: bad arguments");
    SOMI_REQUIRE(!running_mean == !running_var, SOMI_EINVAL, "bn stats: running_mean and running_var go together");
    const int nchunk = somi_red_nchunk(npix);
    float *p1 = workspace, *p2 = workspace + (size_t)nchunk * C;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_stats_stage1, dim3(nchunk), dim3(256), 0, s, x, x_cs, x_coff, npix, C, (const float *)running_mean, p1, p2,
                       red_chunk(npix), act);
    hipLaunchKernelGGL(bn_stats_stage2, dim3(cdiv(C, S2_CH)), dim3(256), 0, s, p1, p2, nchunk, C, npix, eps, momentum, gamma, beta, mean, rstd,
                       scale, shift, running_mean, running_var);
    return launch_status("somi_bn_stats_act_nhwc_f32");
}

extern "C" int somi_bn_stats_partials_f32(const float *part_sum, const float *part_sumsq, int rows, long npix, int C, float eps, float momentum,
                                          const float *gamma, const float *beta, float *mean, float *rstd, float *scale, float *shift,
                                          float *running_mean, float *running_var, float *workspace, somi_stream_t stream) {
    SOMI_REQUIRE(part_sum && part_sumsq && rows > 0 && npix > 0 && C > 0 && mean && rstd && scale && shift && workspace, SOMI_EINVAL,
                 "bn stats from partials: bad arguments");
    SOMI_REQUIRE(!running_mean == !running_var, SOMI_EINVAL, "bn stats: running_mean and running_var go together");
    hipStream_t s = (hipStream_t)stream;
    const float *p1 = part_sum, *p2 = part_sumsq;
    int nchunk = rows;
    if (rows > 1024) {
        const int per = cdiv(rows, 1024);
        nchunk = cdiv(rows, per);
        float *o1 = workspace, *o2 = workspace + (size_t)1024 * C;
        hipLaunchKernelGGL(rows_fold_kernel, dim3(nchunk, cdiv(C, 256)), dim3(256), 0, s, part_sum, part_sumsq, rows, C, per, o1, o2);
        p1 = o1;
        p2 = o2;
    }
    hipLaunchKernelGGL(bn_stats_stage2, dim3(cdiv(C, S2_CH)), dim3(256), 0, s, p1, p2, nchunk, C, npix, eps, momentum, gamma, beta, mean, rstd,
                       scale, shift, running_mean, running_var);
    return launch_status("somi_bn_stats_partials_f32");
}

extern "C" int somi_chan_affine_act_nhwc_f32(const float *x, int x_cs, int x_coff, const float *scale, const float *shift, int act,
                                             int order, float *z, int z_cs, int z_coff, long npix, int C, const float *residual,
                                             int res_cs, int res_coff, somi_stream_t stream) {
    SOMI_REQUIRE(slice_ok(x, x_cs, x_coff, C) && slice_ok(z, z_cs, z_coff, C) && scale && shift && npix > 0 && C % 4 == 0 &&
                     (order == 0 || order == 1) && aligned16(scale) && aligned16(shift), SOMI_EINVAL, "chan affine act: bad arguments");
    SOMI_REQUIRE(!residual || slice_ok(residual, res_cs, res_coff, C), SOMI_EINVAL, "chan affine act: bad residual slice");
    launch_affine_act(x, x_cs, x_coff, scale, shift, act, order, z, z_cs, z_coff, npix, C, residual, res_cs, res_coff, (hipStream_t)stream);
    return launch_status("somi_chan_affine_act_nhwc_f32");
}

extern "C" int somi_bn_act_backward_nhwc_f32(const float *dz, int dz_cs, int dz_coff, const float *x, int x_cs, int x_coff, const float *mean,
                                             const float *rstd, const float *scale, const float *shift, int act, int order,
                                             int batch_stats, float *dx, int dx_cs, int dx_coff, float *dgamma, float *dbeta, long npix,
                                             int C, float *workspace, somi_stream_t stream) {
    SOMI_REQUIRE(slice_ok(dz, dz_cs, dz_coff, C) && slice_ok(x, x_cs, x_coff, C) && slice_ok(dx, dx_cs, dx_coff, C) && mean && rstd && scale &&
                     shift && workspace && npix > 0 && C % 4 == 0 && (order == 0 || order == 1), SOMI_EINVAL, "bn act backward: bad arguments");
    const int nchunk = somi_red_nchunk(npix);
    const size_t cpad = ((size_t)C + 3) / 4 * 4;
    float *p1 = workspace, *p2 = p1 + (size_t)nchunk * C, *cA = p2 + (size_t)nchunk * C, *cB = cA + cpad, *cC = cB + cpad;
    hipStream_t s = (hipStream_t)stream;
    launch_bwd_stage1(nchunk, dz, dz_cs, dz_coff, x, x_cs, x_coff, scale, shift, mean, act, order, npix, C, p1, p2, s);
    hipLaunchKernelGGL(bn_act_bwd_stage2, dim3(cdiv(C, S2_CH)), dim3(256), 0, s, p1, p2, nchunk, C, npix, mean, rstd, scale, batch_stats, cA, cB, cC,
                       dgamma, dbeta);
    launch_bwd_apply(dz, dz_cs, dz_coff, x, x_cs, x_coff, scale, shift, mean, cA, cB, cC, act, order, dx, dx_cs, dx_coff, npix, C, s);
    return launch_status("somi_bn_act_backward_nhwc_f32");
}

extern "C" int somi_bn_pooled_rows(int B, int HW) {
    if (B <= 0 || HW <= 0) return 0;
    const int chunk = red_chunk_img(B, HW);
    return B * ((HW + chunk - 1) / chunk);
}

extern "C" int somi_bn_act_backward_pooled_nhwc_f32(const float *dz, int dz_cs, int dz_coff, const float *x, int x_cs, int x_coff, const float *mean,
                                                    const float *rstd, const float *scale, const float *shift, int act, int order,
                                                    const float *davg, const float *dmax, const int32_t *amaxp, float *dx, int dx_cs, int dx_coff,
                                                    float *dgamma, float *dbeta, int B, int HW, int C, float *workspace, somi_stream_t stream) {
    SOMI_REQUIRE((!dz || slice_ok(dz, dz_cs, dz_coff, C)) && slice_ok(x, x_cs, x_coff, C) && slice_ok(dx, dx_cs, dx_coff, C) && mean && rstd && scale &&
                     shift && workspace && B > 0 && HW > 0 && C % 4 == 0 && (order == 0 || order == 1), SOMI_EINVAL,
                 "bn act backward (pooled): bad arguments");
    SOMI_REQUIRE(davg && !dmax == !amaxp && aligned16(davg) && (!dmax || (aligned16(dmax) && aligned16(amaxp))) && aligned16(mean) && aligned16(scale) &&
                     aligned16(shift),
                 SOMI_EINVAL, "bn act backward (pooled): davg (and dmax + amaxp, together or not at all) (B,C) and the per-channel vectors must be 16 B aligned");
    SOMI_REQUIRE(B <= 65535 && (long)B * HW < (1L << 31), SOMI_EINVAL, "bn act backward (pooled): batch beyond the grid's y range");
    const int chunk = red_chunk_img(B, HW), nimg = (HW + chunk - 1) / chunk, rows = B * nimg;
    const size_t cpad = ((size_t)C + 3) / 4 * 4;
    float *p1 = workspace, *p2 = p1 + (size_t)rows * C, *cA = p2 + (size_t)rows * C, *cB = cA + cpad, *cC = cB + cpad;
    hipStream_t s = (hipStream_t)stream;
    const bool silu0 = act == SOMI_ACT_SILU && order == 0;
    if (silu0)
        hipLaunchKernelGGL(bn_act_bwd_stage1_pooled<true>, dim3(nimg, B), dim3(256), 0, s, dz, dz_cs, dz_coff, x, x_cs, x_coff, scale, shift, mean, act,
                           order, HW, C, p1, p2, chunk, davg, dmax, amaxp);
    else
        hipLaunchKernelGGL(bn_act_bwd_stage1_pooled<false>, dim3(nimg, B), dim3(256), 0, s, dz, dz_cs, dz_coff, x, x_cs, x_coff, scale, shift, mean, act,
                           order, HW, C, p1, p2, chunk, davg, dmax, amaxp);
    hipLaunchKernelGGL(bn_act_bwd_stage2, dim3(cdiv(C, S2_CH)), dim3(256), 0, s, p1, p2, rows, C, (long)B * HW, mean, rstd, scale, 1, cA, cB, cC,
                       dgamma, dbeta);
    if (silu0)
        hipLaunchKernelGGL(bn_act_bwd_apply_pooled<true>, dim3(ew_grid_img(B, HW, C, WG_APPLY), B), dim3(256), 0, s, dz, dz_cs, dz_coff, x, x_cs, x_coff,
                           scale, shift, mean, cA, cB, cC, act, order, dx, dx_cs, dx_coff, HW, C, davg, dmax, amaxp);
    else
        hipLaunchKernelGGL(bn_act_bwd_apply_pooled<false>, dim3(ew_grid_img(B, HW, C, WG_APPLY_RT), B), dim3(256), 0, s, dz, dz_cs, dz_coff, x, x_cs,
                           x_coff, scale, shift, mean, cA, cB, cC, act, order, dx, dx_cs, dx_coff, HW, C, davg, dmax, amaxp);
    return launch_status("somi_bn_act_backward_pooled_nhwc_f32");
}

static bool cbam_bn_args(CbamBnArgs &a, const float *d, int d_cs, int d_coff, const float *y, int y_cs, int y_coff, const float *scale, const float *shift,
                         const float *mean, const float *ca, const float *sa, const float *dstats, const int32_t *amaxc, const int32_t *amaxp, int B, int HW,
                         int C) {
    if (!(slice_ok(d, d_cs, d_coff, C) && slice_ok(y, y_cs, y_coff, C) && scale && shift && mean && ca && sa && dstats && amaxc && amaxp && B > 0 &&
          B <= 65535 && HW > 0 && (long)B * HW < (1L << 31) && C % 4 == 0 && aligned16(scale) && aligned16(shift) && aligned16(mean) && aligned16(ca) &&
          aligned16(amaxp) && ((uintptr_t)dstats & 7) == 0))
        return false;
    a = CbamBnArgs{d, d_cs, d_coff, y, y_cs, y_coff, scale, shift, mean, ca, sa, dstats, amaxc, amaxp, HW, C};
    return true;
}

extern "C" size_t somi_cbam_bn_bwd_workspace_floats(int B, int HW, int C) {
    if (B <= 0 || HW <= 0 || C <= 0) return 0;
    return (size_t)7 * somi_bn_pooled_rows(B, HW) * C + 3 * (((size_t)C + 3) / 4 * 4);
}

extern "C" int somi_cbam_bn_bwd_reduce_f32(const float *d, int d_cs, int d_coff, const float *y, int y_cs, int y_coff, const float *scale,
                                           const float *shift, const float *mean, const float *ca, const float *sa, const float *dstats,
                                           const int32_t *amaxc, const int32_t *amaxp, float *dca, float *workspace, int B, int HW, int C,
                                           somi_stream_t stream) {
    CbamBnArgs a;
    SOMI_REQUIRE(cbam_bn_args(a, d, d_cs, d_coff, y, y_cs, y_coff, scale, shift, mean, ca, sa, dstats, amaxc, amaxp, B, HW, C) && dca && workspace,
                 SOMI_EINVAL, "cbam bn bwd reduce: bad arguments");
    const int chunk = red_chunk_img(B, HW), nimg = (HW + chunk - 1) / chunk;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(cbam_bn_bwd_reduce_kernel, dim3(nimg, B), dim3(256), 0, s, a, workspace, chunk);
    hipLaunchKernelGGL(img_partial_sum_kernel, dim3(cdiv((long)B * C, 16)), dim3(256), 0, s, workspace, nimg, C, B, dca);
    return launch_status("somi_cbam_bn_bwd_reduce_f32");
}

extern "C" int somi_cbam_bn_bwd_apply_f32(const float *d, int d_cs, int d_coff, const float *y, int y_cs, int y_coff, const float *scale,
                                          const float *shift, const float *mean, const float *rstd, const float *ca, const float *sa,
                                          const float *dstats, const int32_t *amaxc, const int32_t *amaxp, const float *davg, const float *dmax,
                                          float *dx, int dx_cs, int dx_coff, float *dgamma, float *dbeta, float *workspace, int B, int HW, int C,
                                          somi_stream_t stream) {
    CbamBnArgs a;
    SOMI_REQUIRE(cbam_bn_args(a, d, d_cs, d_coff, y, y_cs, y_coff, scale, shift, mean, ca, sa, dstats, amaxc, amaxp, B, HW, C) && rstd && davg && dmax &&
                     aligned16(davg) && aligned16(dmax) && slice_ok(dx, dx_cs, dx_coff, C) && workspace,
                 SOMI_EINVAL, "cbam bn bwd apply: bad arguments");
    const int chunk = red_chunk_img(B, HW), nimg = (HW + chunk - 1) / chunk, rows = B * nimg;
    const size_t cpad = ((size_t)C + 3) / 4 * 4;
    float *cA = workspace + (size_t)7 * rows * C, *cB = cA + cpad, *cC = cB + cpad;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(cbam_bn_bwd_stage2, dim3(cdiv(C, S2_CH)), dim3(256), 0, s, workspace, rows, nimg, C, HW, davg, dmax, rstd, scale, cA, cB, cC, dgamma,
                       dbeta);
    hipLaunchKernelGGL(cbam_bn_bwd_apply_kernel, dim3(ew_grid_img(B, HW, C, WG_APPLY_RT), B), dim3(256), 0, s, a, cA, cB, cC, davg, dmax, dx, dx_cs, dx_coff);
    return launch_status("somi_cbam_bn_bwd_apply_f32");
}

extern "C" int somi_bn_local_sums_f64(const float *x, int x_cs, int x_coff, long npix, int C, const float *pivot, const float *part_sum,
                                      const float *part_sumsq, int rows, double *sums, float *workspace, somi_stream_t stream) {
    SOMI_REQUIRE(npix > 0 && C > 0 && C % 4 == 0 && sums && workspace, SOMI_EINVAL, "bn local sums: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const float *p1, *p2;
    int nchunk;
    if (part_sum) {                                              // the partial rows a convolution's epilogue left (taken around `pivot` there)
        SOMI_REQUIRE(part_sumsq && rows > 0, SOMI_EINVAL, "bn local sums: bad partial rows");
        p1 = part_sum; p2 = part_sumsq; nchunk = rows;
        if (rows > 1024) {
            const int per = cdiv(rows, 1024);
            nchunk = cdiv(rows, per);
            float *o1 = workspace, *o2 = workspace + (size_t)1024 * C;
            hipLaunchKernelGGL(rows_fold_kernel, dim3(nchunk, cdiv(C, 256)), dim3(256), 0, s, part_sum, part_sumsq, rows, C, per, o1, o2);
            p1 = o1; p2 = o2;
        }
    } else {
        SOMI_REQUIRE(slice_ok(x, x_cs, x_coff, C), SOMI_EINVAL, "bn local sums: bad slice");
        nchunk = somi_red_nchunk(npix);
        float *o1 = workspace, *o2 = workspace + (size_t)nchunk * C;
        hipLaunchKernelGGL(bn_stats_stage1, dim3(nchunk), dim3(256), 0, s, x, x_cs, x_coff, npix, C, pivot, o1, o2, red_chunk(npix), (int)SOMI_ACT_NONE);
        p1 = o1; p2 = o2;
    }
    hipLaunchKernelGGL(sums_fold_f64_kernel, dim3(cdiv(C, S2_CH)), dim3(256), 0, s, p1, p2, nchunk, C, npix, sums, pivot, 1);
    return launch_status("somi_bn_local_sums_f64");
}

extern "C" int somi_bn_stats_from_sums_f64(const double *all_sums, int nranks, int C, float eps, float momentum, const float *gamma,
                                           const float *beta, float *mean, float *rstd, float *scale, float *shift, float *running_mean,
                                           float *running_var, somi_stream_t stream) {
    SOMI_REQUIRE(all_sums && nranks > 0 && C > 0 && mean && rstd && scale && shift, SOMI_EINVAL, "bn stats from sums: bad arguments");
    SOMI_REQUIRE(!running_mean == !running_var, SOMI_EINVAL, "bn stats: running_mean and running_var go together");
    hipLaunchKernelGGL(bn_stats_from_sums_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, all_sums, nranks, C, eps, momentum, gamma,
                       beta, mean, rstd, scale, shift, running_mean, running_var);
    return launch_status("somi_bn_stats_from_sums_f64");
}

extern "C" int somi_bn_act_backward_sums_f64(const float *dz, int dz_cs, int dz_coff, const float *x, int x_cs, int x_coff, const float *mean,
                                             const float *scale, const float *shift, int act, int order, long npix, int C, double *sums,
                                             float *workspace, somi_stream_t stream) {
    SOMI_REQUIRE(slice_ok(dz, dz_cs, dz_coff, C) && slice_ok(x, x_cs, x_coff, C) && mean && scale && shift && sums && workspace && npix > 0 &&
                     C % 4 == 0 && (order == 0 || order == 1), SOMI_EINVAL, "bn act backward sums: bad arguments");
    const int nchunk = somi_red_nchunk(npix);
    float *p1 = workspace, *p2 = p1 + (size_t)nchunk * C;
    hipStream_t s = (hipStream_t)stream;
    launch_bwd_stage1(nchunk, dz, dz_cs, dz_coff, x, x_cs, x_coff, scale, shift, mean, act, order, npix, C, p1, p2, s);
    hipLaunchKernelGGL(sums_fold_f64_kernel, dim3(cdiv(C, S2_CH)), dim3(256), 0, s, p1, p2, nchunk, C, npix, sums, nullptr, 0);
    return launch_status("somi_bn_act_backward_sums_f64");
}

extern "C" int somi_bn_act_backward_apply_sync_f32(const float *dz, int dz_cs, int dz_coff, const float *x, int x_cs, int x_coff,
                                                   const float *mean, const float *rstd, const float *scale, const float *shift, int act,
                                                   int order, const double *local_sums, const double *all_sums, int nranks, float *dx, int dx_cs,
                                                   int dx_coff, float *dgamma, float *dbeta, long npix, int C, float *workspace,
                                                   somi_stream_t stream) {
    SOMI_REQUIRE(slice_ok(dz, dz_cs, dz_coff, C) && slice_ok(x, x_cs, x_coff, C) && slice_ok(dx, dx_cs, dx_coff, C) && mean && rstd && scale &&
                     shift && local_sums && all_sums && nranks > 0 && workspace && npix > 0 && C % 4 == 0 && (order == 0 || order == 1),
                 SOMI_EINVAL, "bn act backward apply (sync): bad arguments");
    const size_t cpad = ((size_t)C + 3) / 4 * 4;
    float *cA = workspace, *cB = cA + cpad, *cC = cB + cpad;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_act_bwd_sync_stage2, dim3(cdiv(C, 256)), dim3(256), 0, s, local_sums, all_sums, nranks, C, mean, rstd, scale, cA, cB, cC,
                       dgamma, dbeta);
    launch_bwd_apply(dz, dz_cs, dz_coff, x, x_cs, x_coff, scale, shift, mean, cA, cB, cC, act, order, dx, dx_cs, dx_coff, npix, C, s);
    return launch_status("somi_bn_act_backward_apply_sync_f32");
}

extern "C" int somi_chan_sum_nhwc_f32(const float *x, int x_cs, int x_coff, long npix, int C, float *out_accumulate, float *workspace,
                                      somi_stream_t stream) {
    SOMI_REQUIRE(slice_ok(x, x_cs, x_coff, C) && out_accumulate && workspace && npix > 0 && C % 4 == 0, SOMI_EINVAL, "chan sum: bad arguments");
    const int nchunk = somi_red_nchunk(npix);
    float *p1 = workspace, *p2 = workspace + (size_t)nchunk * C;
    hipLaunchKernelGGL(chan_sum_stage1, dim3(nchunk), dim3(256), 0, (hipStream_t)stream, x, x_cs, x_coff, npix, C, p1, p2, red_chunk(npix));
    hipLaunchKernelGGL(chan_sum_stage2, dim3(cdiv(C, S2_CH)), dim3(256), 0, (hipStream_t)stream, p1, nchunk, C, out_accumulate);
    return launch_status("somi_chan_sum_nhwc_f32");
}

extern "C" int somi_add_nhwc_f32(const float *a, int a_cs, int a_coff, const float *b, int b_cs, int b_coff, float *out, int o_cs,
                                 int o_coff, long npix, int C, somi_stream_t stream) {
    SOMI_REQUIRE(slice_ok(a, a_cs, a_coff, C) && slice_ok(b, b_cs, b_coff, C) && slice_ok(out, o_cs, o_coff, C) && npix > 0 && C % 4 == 0,
                 SOMI_EINVAL, "add: bad arguments");
    hipLaunchKernelGGL(add_kernel, dim3(ew_grid_c(npix, C, WG_ADD)), dim3(256), 0, (hipStream_t)stream, a, a_cs, a_coff, b, b_cs, b_coff, out, o_cs,
                       o_coff, npix, C);
    return launch_status("somi_add_nhwc_f32");
}
