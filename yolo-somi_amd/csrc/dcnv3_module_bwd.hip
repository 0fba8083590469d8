// Backward of the layers around the DCNv3 operator inside the DCNv3 nn.Module (models/ops_dcnv3/modules/dcnv3.py:283-291, 334,
// 355-377): LayerNorm -> GELU after the depthwise conv, the softmax over the K sampling points of each (pixel, group), and
// the centre-feature-scale blend.  (The Linear layers are 1x1 convolutions: somi_conv2d_dgrad / wgrad_nhwc_f32; the depthwise
// conv: somi_dwconv3x3_bwd_nhwc_f32; the operator itself: somi_dcnv3_backward_f32.)
#include "common.h"

namespace somi {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float gelu_grad(float v) {               // d/dv [0.5 v (1 + erf(v/sqrt2))]
    return 0.5f * (1.f + erff(v * 0.70710678118654752440f)) + v * 0.39894228040143267794f * expf(-0.5f * v * v);
}

// z = gelu(LN(u)):  one wave per pixel row.  du = rstd * (g - mean(g) - xhat * mean(g * xhat)) with g = dz * gelu'(v) * gamma;
// per-row-block partial sums of dgamma = sum g0 * xhat, dbeta = sum g0 (g0 = dz * gelu'(v)) go to part[blk][2][C].
__global__ __launch_bounds__(256) void ln_gelu_bwd_kernel(const float *__restrict__ u, const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, float eps, const float *__restrict__ dz,
                                                          float *__restrict__ du, float *__restrict__ part, long npix, int C) {
    extern __shared__ float sm[];                                   // [4 waves][2][C]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *mg = sm + (size_t)wave * 2 * C, *mb = mg + C;
    for (int c = lane; c < C; c += 64) { mg[c] = 0.f; mb[c] = 0.f; }
    const long wave_id = blockIdx.x * 4L + wave, nwave = (long)gridDim.x * 4;
    for (long p = wave_id; p < npix; p += nwave) {
        const float *ur = u + p * C, *dr = dz + p * C;
        float s = 0.f;
        for (int c = lane * 4; c < C; c += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(ur + c);
            s += (v[0] + v[1]) + (v[2] + v[3]);
        }
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s / (float)C;
        float q = 0.f;
        for (int c = lane * 4; c < C; c += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(ur + c) - mean;
            q += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        }
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
        const float rstd = rsqrtf(q / (float)C + eps);
        float sg = 0.f, sgx = 0.f;
        for (int c = lane * 4; c < C; c += 256) {
            const f32x4 xh = (*reinterpret_cast<const f32x4 *>(ur + c) - mean) * rstd;
            const f32x4 gm = *reinterpret_cast<const f32x4 *>(gamma + c), bt = *reinterpret_cast<const f32x4 *>(beta + c);
            const f32x4 d = *reinterpret_cast<const f32x4 *>(dr + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float g0 = d[e] * gelu_grad(xh[e] * gm[e] + bt[e]);
                mg[c + e] += g0 * xh[e];                              // this lane owns columns c..c+3 of its wave's partial
                mb[c + e] += g0;
                const float g = g0 * gm[e];
                sg += g;
                sgx += g * xh[e];
            }
        }
        for (int o = 32; o > 0; o >>= 1) { sg += __shfl_xor(sg, o); sgx += __shfl_xor(sgx, o); }
        const float m1 = sg / (float)C, m2 = sgx / (float)C;
        for (int c = lane * 4; c < C; c += 256) {
            const f32x4 xh = (*reinterpret_cast<const f32x4 *>(ur + c) - mean) * rstd;
            const f32x4 gm = *reinterpret_cast<const f32x4 *>(gamma + c), bt = *reinterpret_cast<const f32x4 *>(beta + c);
            const f32x4 d = *reinterpret_cast<const f32x4 *>(dr + c);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = rstd * (d[e] * gelu_grad(xh[e] * gm[e] + bt[e]) * gm[e] - m1 - xh[e] * m2);
            *reinterpret_cast<f32x4 *>(du + p * C + c) = o;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * C; c += 256)                    // fixed order over the 4 waves
        part[(size_t)blockIdx.x * 2 * C + c] = (sm[c] + sm[2 * C + c]) + (sm[4 * C + c] + sm[6 * C + c]);
}
// The same for C = NV * 256 (the DCNv3 sites: 256): a lane keeps its NV float4 of the row, of dz and of gamma / beta in registers, so a row costs
// ONE round of loads instead of four dependent ones (the form above re-reads the row for the mean, the variance, the sums and the result), the
// next row's loads are issued before this row's three reductions, gelu' is evaluated once, and the lane's columns of the dgamma / dbeta
// partials stay in registers until the end.  Same expressions in the same order: bit-identical results.
template <int NV>
__global__ __launch_bounds__(256) void ln_gelu_bwd_reg_kernel(const float *__restrict__ u, const float *__restrict__ gamma,
                                                              const float *__restrict__ beta, float eps, const float *__restrict__ dz,
                                                              float *__restrict__ du, float *__restrict__ part, long npix) {
    constexpr int C = NV * 256;
    extern __shared__ float sm[];                                   // [4 waves][2][C]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *mg = sm + (size_t)wave * 2 * C, *mb = mg + C;
    f32x4 gm[NV], bt[NV], ag[NV], ab[NV], v[NV], d[NV], nv[NV], nd[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        gm[k] = *reinterpret_cast<const f32x4 *>(gamma + k * 256 + lane * 4);
        bt[k] = *reinterpret_cast<const f32x4 *>(beta + k * 256 + lane * 4);
        ag[k] = ab[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const long wave_id = blockIdx.x * 4L + wave, nwave = (long)gridDim.x * 4;
    long p = wave_id;
    if (p < npix) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            nv[k] = *reinterpret_cast<const f32x4 *>(u + p * C + k * 256 + lane * 4);
            nd[k] = *reinterpret_cast<const f32x4 *>(dz + p * C + k * 256 + lane * 4);
        }
    }
    for (; p < npix; p += nwave) {
#pragma unroll
        for (int k = 0; k < NV; ++k) { v[k] = nv[k]; d[k] = nd[k]; }
        if (p + nwave < npix) {
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                nv[k] = *reinterpret_cast<const f32x4 *>(u + (p + nwave) * C + k * 256 + lane * 4);
                nd[k] = *reinterpret_cast<const f32x4 *>(dz + (p + nwave) * C + k * 256 + lane * 4);
            }
        }
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s / (float)C;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const f32x4 w = v[k] - mean;
            q += (w[0] * w[0] + w[1] * w[1]) + (w[2] * w[2] + w[3] * w[3]);
        }
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
        const float rstd = rsqrtf(q / (float)C + eps);
        float sg = 0.f, sgx = 0.f;
        f32x4 xh[NV], g[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            xh[k] = (v[k] - mean) * rstd;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float g0 = d[k][e] * gelu_grad(xh[k][e] * gm[k][e] + bt[k][e]);
                ag[k][e] += g0 * xh[k][e];
                ab[k][e] += g0;
                g[k][e] = g0 * gm[k][e];
                sg += g[k][e];
                sgx += g[k][e] * xh[k][e];
            }
        }
        for (int o = 32; o > 0; o >>= 1) { sg += __shfl_xor(sg, o); sgx += __shfl_xor(sgx, o); }
        const float m1 = sg / (float)C, m2 = sgx / (float)C;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = rstd * (g[k][e] - m1 - xh[k][e] * m2);
            *reinterpret_cast<f32x4 *>(du + p * C + k * 256 + lane * 4) = o;
        }
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        *reinterpret_cast<f32x4 *>(mg + k * 256 + lane * 4) = ag[k];
        *reinterpret_cast<f32x4 *>(mb + k * 256 + lane * 4) = ab[k];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * C; c += 256)                    // fixed order over the 4 waves
        part[(size_t)blockIdx.x * 2 * C + c] = (sm[c] + sm[2 * C + c]) + (sm[4 * C + c] + sm[6 * C + c]);
}

// 256 threads = 16 columns x 16 row groups: a group walks every 16th partial row with 8 loads in flight (one thread per column walking all
// <= 1024 rows was a chain of 1024 memory latencies: 0.25 ms for a 512-column sum), the 16 group sums are added in a fixed order
__global__ __launch_bounds__(256) void ln_param_grad_kernel(const float *__restrict__ part, int nblk, int C, float *dgamma, float *dbeta) {
    __shared__ double l[256];
    const int cl = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double a = 0.0;
    if (c < 2 * C) {
        int b = grp;
        for (; b + 7 * 16 < nblk; b += 8 * 16) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = part[(size_t)(b + j * 16) * 2 * C + c];
#pragma unroll
            for (int j = 0; j < 8; ++j) a += v[j];
        }
        for (; b < nblk; b += 16) a += part[(size_t)b * 2 * C + c];
    }
    l[threadIdx.x] = a;
    __syncthreads();
    if (grp != 0 || c >= 2 * C) return;
    double s = 0.0;
#pragma unroll
    for (int g = 0; g < 16; ++g) s += l[g * 16 + cl];
    if (c < C) dgamma[c] += (float)s; else dbeta[c - C] += (float)s;
}

// y = softmax(x) over K: dx_k = y_k * (dy_k - sum_j dy_j y_j); one lane per (pixel, group).  Group g of pixel p starts at p*ps + g*K
// (packed rows or a column range of wider rows); dx == dy (in place) is allowed, hence no __restrict__ on them.
__global__ __launch_bounds__(256) void group_softmax_bwd_kernel(const float *__restrict__ y, const float *dy, float *dx, long n, int G, int K,
                                                                long y_ps, long d_ps) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long p = i / G;
        const int g = (int)(i - p * G);
        const float *yr = y + p * y_ps + g * K, *dr = dy + p * d_ps + g * K;
        float *o = dx + p * d_ps + g * K;
        float dot = 0.f;
        for (int k = 0; k < K; ++k) dot += dr[k] * yr[k];
        for (int k = 0; k < K; ++k) o[k] = yr[k] * (dr[k] - dot);
    }
}

// The same for 16-byte aligned rows: SMB_P pixels' y and dy rows staged in LDS by coalesced float4 loads, a lane per (pixel, group) there
// (see group_softmax_tile_kernel in layers.hip), dx back as float4.  Arithmetic identical to the kernel above.
constexpr int SMB_P = 32;
__global__ __launch_bounds__(256) void group_softmax_bwd_tile_kernel(const float *__restrict__ y, const float *dy, float *dx, long npix, int G, int K,
                                                                     long y_ps, long d_ps) {
    extern __shared__ float smb_rows[];
    const int GK = G * K, Q = GK >> 2;
    float *sy = smb_rows, *sd = smb_rows + SMB_P * GK;
    for (long p0 = (long)blockIdx.x * SMB_P; p0 < npix; p0 += (long)gridDim.x * SMB_P) {
        const int np = (int)min((long)SMB_P, npix - p0);
        for (int i = threadIdx.x; i < np * Q; i += 256) {
            const int pl = i / Q, q = i - pl * Q;
            reinterpret_cast<f32x4 *>(sy)[i] = *reinterpret_cast<const f32x4 *>(y + (p0 + pl) * y_ps + q * 4);
            reinterpret_cast<f32x4 *>(sd)[i] = *reinterpret_cast<const f32x4 *>(dy + (p0 + pl) * d_ps + q * 4);
        }
        __syncthreads();
        for (int gi = threadIdx.x; gi < np * G; gi += 256) {
            const float *yr = sy + gi * K;
            float *dr = sd + gi * K;
            float dot = 0.f;
            for (int k = 0; k < K; ++k) dot += dr[k] * yr[k];
            for (int k = 0; k < K; ++k) dr[k] = yr[k] * (dr[k] - dot);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < np * Q; i += 256) {
            const int pl = i / Q, q = i - pl * Q;
            *reinterpret_cast<f32x4 *>(dx + (p0 + pl) * d_ps + q * 4) = reinterpret_cast<const f32x4 *>(sd)[i];
        }
        __syncthreads();
    }
}

// out = x*(1-s) + xproj*s, s = sigmoid(logit[p][g]):  dx = dout*(1-s); dxproj = dout*s; dlogit[p][g] = s(1-s) * sum_c dout*(xproj - x).
// One wave per pixel, lanes over channels; the sum over a group's channels is an LDS-free segmented reduction per group.
__global__ __launch_bounds__(256) void cfs_blend_bwd_kernel(const float *__restrict__ x, const float *__restrict__ xproj,
                                                            const float *__restrict__ logit, int logit_cs, const float *__restrict__ dout,
                                                            float *__restrict__ dx, float *__restrict__ dxproj, float *__restrict__ dlogit,
                                                            int dlogit_cs, long npix, int G, int Gc) {
    const int C = G * Gc;
    const int lane = threadIdx.x & 63;
    const long wave_id = (blockIdx.x * 256L + threadIdx.x) >> 6, nwave = (long)gridDim.x * 4;
    for (long p = wave_id; p < npix; p += nwave) {
        for (int g = 0; g < G; ++g) {
            const float sg = 1.0f / (1.0f + expf(-logit[p * logit_cs + g]));
            float acc = 0.f;
            for (int c = lane; c < Gc; c += 64) {
                const long i = p * C + g * Gc + c;
                const float d = dout[i];
                dx[i] = d * (1.f - sg);
                dxproj[i] = d * sg;
                acc += d * (xproj[i] - x[i]);
            }
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
            if (lane == 0) dlogit[p * dlogit_cs + g] = acc * sg * (1.f - sg);
        }
    }
}

static inline int grid_for(long items, long cap = 2048) {
    long g = (items + 255) / 256;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace somi

using namespace somi;

extern "C" size_t somi_layernorm_act_bwd_workspace_floats(long npix, int C) {
    return (size_t)grid_for(npix * 64, 1024) * 2 * (size_t)C;
}

extern "C" int somi_layernorm_gelu_bwd_nhwc_f32(const float *u, const float *gamma, const float *beta, float eps, const float *dz, float *du,
                                                float *dgamma_accumulate, float *dbeta_accumulate, float *workspace, long npix, int C,
                                                somi_stream_t stream) {
    SOMI_REQUIRE(u && gamma && beta && dz && du && dgamma_accumulate && dbeta_accumulate && workspace && npix > 0 && C > 0 && C % 4 == 0 &&
                     aligned16(u) && aligned16(dz) && aligned16(du) && aligned16(gamma) && aligned16(beta), SOMI_EINVAL,
                 "layernorm+gelu backward: bad arguments (C %% 4, 16 B alignment)");
    SOMI_REQUIRE((size_t)C * 8 * sizeof(float) <= 64 * 1024, SOMI_ENOTIMPL, "layernorm+gelu backward: C up to 2048");
    const int nblk = grid_for(npix * 64, 1024);
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = (size_t)C * 8 * sizeof(float);
    if (C == 256) hipLaunchKernelGGL(ln_gelu_bwd_reg_kernel<1>, dim3(nblk), dim3(256), lds, s, u, gamma, beta, eps, dz, du, workspace, npix);
    else if (C == 512) hipLaunchKernelGGL(ln_gelu_bwd_reg_kernel<2>, dim3(nblk), dim3(256), lds, s, u, gamma, beta, eps, dz, du, workspace, npix);
    else hipLaunchKernelGGL(ln_gelu_bwd_kernel, dim3(nblk), dim3(256), lds, s, u, gamma, beta, eps, dz, du, workspace, npix, C);
    hipLaunchKernelGGL(ln_param_grad_kernel, dim3(cdiv(2L * C, 16)), dim3(256), 0, s, workspace, nblk, C, dgamma_accumulate, dbeta_accumulate);
    return launch_status("somi_layernorm_gelu_bwd_nhwc_f32");
}

extern "C" int somi_group_softmax_bwd_f32(const float *y, const float *dy, float *dx, long n_groups, int K, somi_stream_t stream) {
    SOMI_REQUIRE(y && dy && dx && n_groups > 0 && K > 0, SOMI_EINVAL, "group softmax backward: bad arguments");
    hipLaunchKernelGGL(group_softmax_bwd_kernel, dim3(grid_for(n_groups)), dim3(256), 0, (hipStream_t)stream, y, dy, dx, n_groups, 1, K, (long)K,
                       (long)K);
    return launch_status("somi_group_softmax_bwd_f32");
}

extern "C" int somi_group_softmax_bwd_strided_f32(const float *y, long y_stride, const float *dy, float *dx, long d_stride, long npix, int G, int K,
                                                  somi_stream_t stream) {
    SOMI_REQUIRE(y && dy && dx && npix > 0 && G > 0 && K > 0 && y_stride >= (long)G * K && d_stride >= (long)G * K, SOMI_EINVAL,
                 "group softmax backward (strided): bad arguments");
    const size_t lds = 2 * (size_t)SMB_P * G * K * sizeof(float);
    if ((G * K) % 4 == 0 && y_stride % 4 == 0 && d_stride % 4 == 0 && aligned16(y) && aligned16(dy) && aligned16(dx) && lds <= 48 * 1024) {
        const long g = (npix + SMB_P - 1) / SMB_P;
        hipLaunchKernelGGL(group_softmax_bwd_tile_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), lds, (hipStream_t)stream, y, dy, dx, npix, G,
                           K, y_stride, d_stride);
    } else {
        hipLaunchKernelGGL(group_softmax_bwd_kernel, dim3(grid_for(npix * G)), dim3(256), 0, (hipStream_t)stream, y, dy, dx, npix * G, G, K, y_stride,
                           d_stride);
    }
    return launch_status("somi_group_softmax_bwd_strided_f32");
}

extern "C" int somi_dcnv3_cfs_blend_bwd_f32(const float *x, const float *xproj, const float *logit, int logit_cs, const float *dout, float *dx,
                                            float *dxproj, float *dlogit, int dlogit_cs, long npix, int G, int Gc, somi_stream_t stream) {
    SOMI_REQUIRE(x && xproj && logit && dout && dx && dxproj && dlogit && npix > 0 && G > 0 && Gc > 0 && logit_cs >= G && dlogit_cs >= G,
                 SOMI_EINVAL, "cfs blend backward: bad arguments");
    hipLaunchKernelGGL(cfs_blend_bwd_kernel, dim3(grid_for(npix * 64)), dim3(256), 0, (hipStream_t)stream, x, xproj, logit, logit_cs, dout, dx,
                       dxproj, dlogit, dlogit_cs, npix, G, Gc);
    return launch_status("somi_dcnv3_cfs_blend_bwd_f32");
}
