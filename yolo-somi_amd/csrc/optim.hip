// Fused optimizer step: Adam (torch.optim.Adam semantics, L2 weight decay added to the gradient) + the model-EMA update of
// utils/torch_utils.py:335-345 in ONE pass over a flat parameter segment (train.py:271-277 runs them as separate sweeps of
// all 77.5 M parameters).  Bandwidth-bound: 5 reads + 4 writes of 4 B per parameter.
#include "common.h"

namespace somi {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void adam_ema_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v,
                                                       float *__restrict__ ema, long n, float lr, float b1, float b2, float eps, float wd,
                                                       float bc1, float bc2_sqrt, float ema_d) {
    const float step = lr / bc1;
    const long n4 = n >> 2;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4 pv = *reinterpret_cast<f32x4 *>(p + i * 4);
        f32x4 gv = *reinterpret_cast<const f32x4 *>(g + i * 4) + wd * pv;
        f32x4 mv = b1 * *reinterpret_cast<f32x4 *>(m + i * 4) + (1.f - b1) * gv;
        f32x4 vv = b2 * *reinterpret_cast<f32x4 *>(v + i * 4) + (1.f - b2) * gv * gv;
#pragma unroll
        for (int e = 0; e < 4; ++e) pv[e] -= step * mv[e] / (sqrtf(vv[e]) / bc2_sqrt + eps);
        *reinterpret_cast<f32x4 *>(p + i * 4) = pv;
        *reinterpret_cast<f32x4 *>(m + i * 4) = mv;
        *reinterpret_cast<f32x4 *>(v + i * 4) = vv;
        if (ema) *reinterpret_cast<f32x4 *>(ema + i * 4) = ema_d * *reinterpret_cast<f32x4 *>(ema + i * 4) + (1.f - ema_d) * pv;
    }
    for (long i = n4 * 4 + blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {      // tail
        float pv = p[i];
        const float gv = g[i] + wd * pv;
        const float mv = b1 * m[i] + (1.f - b1) * gv, vv = b2 * v[i] + (1.f - b2) * gv * gv;
        pv -= step * mv / (sqrtf(vv) / bc2_sqrt + eps);
        p[i] = pv; m[i] = mv; v[i] = vv;
        if (ema) ema[i] = ema_d * ema[i] + (1.f - ema_d) * pv;
    }
}

// torch.optim.SGD(momentum, nesterov=True, dampening 0) + the EMA update, one pass: g' = g + wd p; buf = first ? g' : mom buf + g';
// p -= lr (g' + mom buf)   (train.py:138: the branch the reference keeps behind `opt.adam = False`)
__global__ __launch_bounds__(256) void sgd_ema_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ buf, float *__restrict__ ema,
                                                      long n, float lr, float mom, float wd, int first, float ema_d) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float pv = p[i];
        const float gv = g[i] + wd * pv;
        const float bv = first ? gv : mom * buf[i] + gv;
        pv -= lr * (gv + mom * bv);
        p[i] = pv;
        buf[i] = bv;
        if (ema) ema[i] = ema_d * ema[i] + (1.f - ema_d) * pv;
    }
}

// y = a*y + b*x  (EMA of the non-parameter float state: BN running statistics)
__global__ __launch_bounds__(256) void axpby_kernel(float *__restrict__ y, const float *__restrict__ x, long n, float a, float b) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = a * y[i] + b * x[i];
}

static inline int grid_for(long n) {
    long g = (n / 4 + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

// forward packing [Cout][T][Cin] -> data-gradient packing [Cin][T][Cout], one 32x32 LDS-transposed tile per workgroup and tap
__global__ __launch_bounds__(256) void pack_dgrad_kernel(const float *__restrict__ w, float *__restrict__ wt, int Cout, int T, int Cin) {
    __shared__ float tile[32][33];
    const int t = blockIdx.z, co0 = blockIdx.y * 32, ci0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + tx;
        tile[r][tx] = (co < Cout && ci < Cin) ? w[((size_t)co * T + t) * Cin + ci] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + tx;
        if (ci < Cin && co < Cout) wt[((size_t)ci * T + t) * Cout + co] = tile[tx][r];
    }
}

}  // namespace somi

using namespace somi;

extern "C" int somi_adam_ema_step_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float *ema, long n, float lr, float beta1,
                                      float beta2, float eps, float weight_decay, int step, float ema_decay, somi_stream_t stream) {
    SOMI_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, SOMI_EINVAL, "adam: bad arguments");
    SOMI_REQUIRE(aligned16(param) && aligned16(grad) && aligned16(exp_avg) && aligned16(exp_avg_sq) && (!ema || aligned16(ema)), SOMI_EINVAL,
                 "adam: segments must be 16 B aligned");
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adam_ema_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, ema, n, lr, beta1,
                       beta2, eps, weight_decay, bc1, sqrtf(bc2), ema_decay);
    return launch_status("somi_adam_ema_step_f32");
}

extern "C" int somi_sgd_ema_step_f32(float *param, const float *grad, float *momentum_buf, float *ema, long n, float lr, float momentum,
                                     float weight_decay, int step, float ema_decay, somi_stream_t stream) {
    SOMI_REQUIRE(param && grad && momentum_buf && n > 0 && step >= 1, SOMI_EINVAL, "sgd: bad arguments");
    hipLaunchKernelGGL(sgd_ema_kernel, dim3(grid_for(n * 4)), dim3(256), 0, (hipStream_t)stream, param, grad, momentum_buf, ema, n, lr, momentum,
                       weight_decay, step == 1 ? 1 : 0, ema_decay);
    return launch_status("somi_sgd_ema_step_f32");
}

extern "C" int somi_axpby_f32(float *y, const float *x, long n, float a, float b, somi_stream_t stream) {
    SOMI_REQUIRE(y && x && n > 0, SOMI_EINVAL, "axpby: bad arguments");
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n * 4)), dim3(256), 0, (hipStream_t)stream, y, x, n, a, b);
    return launch_status("somi_axpby_f32");
}

extern "C" int somi_pack_dgrad_weights_f32(const float *w_packed, float *w_dgrad, int Cout, int taps, int Cin, somi_stream_t stream) {
    using namespace somi;
    SOMI_REQUIRE(w_packed && w_dgrad && Cout > 0 && taps > 0 && Cin > 0 && taps < 65536, SOMI_EINVAL, "pack dgrad: bad arguments");
    hipLaunchKernelGGL(pack_dgrad_kernel, dim3(cdiv(Cin, 32), cdiv(Cout, 32), taps), dim3(256), 0, (hipStream_t)stream, w_packed, w_dgrad,
                       Cout, taps, Cin);
    return launch_status("somi_pack_dgrad_weights_f32");
}
