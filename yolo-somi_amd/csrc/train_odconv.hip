// Training path of ODConv (models/common.py:4495-4624): small dense layers on (B, features) matrices, their backward, and the
// backward of the per-sample weight synthesis.  The heavy parts (per-sample conv forward / dgrad / wgrad, pooling, batch-norm)
// reuse the kernels of conv_igemm.hip / conv_wgrad.hip / train_ops.hip / layers.hip.
//
// Forward:  g = avgpool(x) (B,Cin);  z = relu(BN_batch(fc g))  [BN skipped when B == 1, :4562];  a_f = sig(Wf z + bf),
//           a_s = sig(Ws z + bs), a_c = sig(Wc z + bc), a_w = softmax(Ww z + bw);   W_b = a_f (x) a_s (x) a_c * sum_K a_w[K] W[K];
//           bias_b = a_w @ bias;  y = conv(x; W_b) + bias_b.
// Backward of the synthesis, with G = dW_b, P = a_f a_s a_c, M = sum_K a_w[K] W[K]:
//           dW[K] += sum_b G P a_w[b,K];  da_w[b,K] = sum G P W[K] + dbias_b . bias[K];  da_f[b,n] = sum_{t,c} G M a_s a_c  (etc.);
//           dbias[K] += sum_b a_w[b,K] dbias_b.
#include "common.h"

namespace somi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ACT_SOFTMAX = 5;

// y[b][y_off + o] = act(sum_i x[b][i] W[o][i] + bias[o]);  one workgroup per sample, one wave per output (strided)
__global__ __launch_bounds__(256) void linear_kernel(const float *__restrict__ x, int ldx, const float *__restrict__ W, const float *__restrict__ bias,
                                                     int act, float *__restrict__ y, int ldy, int y_off, int nin, int nout) {
    __shared__ float pre[1024];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *xr = x + (long)b * ldx;
    for (int o = wave; o < nout; o += 4) {
        float s = 0.f;
        for (int i = lane; i < nin; i += 64) s += W[(long)o * nin + i] * xr[i];
        for (int sft = 32; sft > 0; sft >>= 1) s += __shfl_down(s, sft);
        if (lane == 0) pre[o] = s + (bias ? bias[o] : 0.f);
    }
    __syncthreads();
    if (act == ACT_SOFTMAX) {
        if (threadIdx.x == 0) {
            float m = pre[0];
            for (int o = 1; o < nout; ++o) m = fmaxf(m, pre[o]);
            float den = 0.f;
            for (int o = 0; o < nout; ++o) den += expf(pre[o] - m);
            for (int o = 0; o < nout; ++o) y[(long)b * ldy + y_off + o] = expf(pre[o] - m) / den;
        }
    } else {
        for (int o = threadIdx.x; o < nout; o += 256) y[(long)b * ldy + y_off + o] = apply_act_rt(pre[o], act);
    }
}

// d(pre-activation) from d(output): sigmoid y(1-y), relu [y>0], softmax y(dy - sum dy y), none
__global__ __launch_bounds__(64) void act_bwd_rows_kernel(const float *__restrict__ dy, const float *__restrict__ y, int ld, int off, int act,
                                                          float *__restrict__ dpre, int ldp, int ncol) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float *yr = y + (long)b * ld + off, *dr = dy + (long)b * ld + off;
    float dot = 0.f;
    if (act == ACT_SOFTMAX) {
        for (int o = lane; o < ncol; o += 64) dot += dr[o] * yr[o];
        for (int s = 32; s > 0; s >>= 1) dot += __shfl_xor(dot, s);
    }
    for (int o = lane; o < ncol; o += 64) {
        const float v = yr[o], g = dr[o];
        float d = g;
        if (act == SOMI_ACT_SIGMOID) d = g * v * (1.f - v);
        else if (act == SOMI_ACT_RELU) d = v > 0.f ? g : 0.f;
        else if (act == ACT_SOFTMAX) d = v * (g - dot);
        dpre[(long)b * ldp + o] = d;
    }
}

// dW[o][i] += sum_b dpre[b][o] x[b][i];  db[o] += sum_b dpre[b][o]   (fixed order over b: deterministic)
__global__ __launch_bounds__(256) void linear_bwd_weight_kernel(const float *__restrict__ x, int ldx, const float *__restrict__ dpre, int ldp, float *dW,
                                                                float *db, int B, int nin, int nout) {
    const long n = (long)nout * nin;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < n + nout; e += (long)gridDim.x * 256) {
        if (e < n) {
            const int o = (int)(e / nin), i = (int)(e % nin);
            float s = 0.f;
            for (int b = 0; b < B; ++b) s += dpre[(long)b * ldp + o] * x[(long)b * ldx + i];
            dW[e] += s;
        } else if (db) {
            const int o = (int)(e - n);
            float s = 0.f;
            for (int b = 0; b < B; ++b) s += dpre[(long)b * ldp + o];
            db[o] += s;
        }
    }
}
// dx[b][i] (+)= sum_o dpre[b][o] W[o][i]
__global__ __launch_bounds__(256) void linear_bwd_data_kernel(const float *__restrict__ W, const float *__restrict__ dpre, int ldp, float *__restrict__ dx,
                                                              int ldx, int accumulate, int B, int nin, int nout) {
    const long n = (long)B * nin;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const int b = (int)(e / nin), i = (int)(e % nin);
        float s = accumulate ? dx[(long)b * ldx + i] : 0.f;
        for (int o = 0; o < nout; ++o) s += dpre[(long)b * ldp + o] * W[(long)o * nin + i];
        dx[(long)b * ldx + i] = s;
    }
}

// forward synthesis from a given attention buffer (B, Cout+kk+Cin+K), optional following-BN fold
__global__ __launch_bounds__(256) void odconv_synth_fwd_kernel(const float *__restrict__ ws, const float *__restrict__ Wk, const float *__restrict__ biask,
                                                               float *__restrict__ wout, float *__restrict__ bout, int B, int Cin, int Cin_pad, int Cout,
                                                               int kk, int K) {
    const int C4 = Cin_pad >> 2;
    const long per_b = (long)Cout * kk * C4, items = (long)B * per_b, set = (long)Cout * kk * Cin_pad;
    for (long it = blockIdx.x * 256L + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
        const long b = it / per_b, rem = it % per_b;
        const int c = (int)(rem % C4) * 4, t = (int)((rem / C4) % kk), n = (int)(rem / ((long)C4 * kk));
        const float *at = ws + b * (Cout + kk + Cin + K), *aw = at + Cout + kk + Cin;
        const float fs = at[n] * at[Cout + t];
        f32x4 ac = {0.f, 0.f, 0.f, 0.f}, acc = ac;
#pragma unroll
        for (int e = 0; e < 4; ++e) ac[e] = (c + e < Cin) ? at[Cout + kk + c + e] : 0.f;
        for (int q = 0; q < K; ++q) acc += ((fs * ac) * aw[q]) * *reinterpret_cast<const f32x4 *>(Wk + q * set + ((long)n * kk + t) * Cin_pad + c);
        *reinterpret_cast<f32x4 *>(wout + b * set + ((long)n * kk + t) * Cin_pad + c) = acc;
        if (t == 0 && c == 0) {
            float bv = 0.f;
            if (biask) for (int q = 0; q < K; ++q) bv += aw[q] * biask[q * Cout + n];
            bout[b * Cout + n] = bv;
        }
    }
}

// backward synthesis: one workgroup per (n, b); lanes sweep (t, c)
__global__ __launch_bounds__(256) void odconv_synth_bwd_kernel(const float *__restrict__ dWb, const float *__restrict__ ws, const float *__restrict__ Wk,
                                                               const float *__restrict__ biask, const float *__restrict__ dbias_b, float *dWk,
                                                               float *dbiask, float *__restrict__ da, int Cin, int Cin_pad, int Cout, int kk, int K) {
    __shared__ float red[4];
    __shared__ float dsk[64];                                            // da_s partials per tap (kk <= 49)
    __shared__ float dwk_s[16];
    const int n = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int na = Cout + kk + Cin + K;
    const float *at = ws + (long)b * na, *aw = at + Cout + kk + Cin;
    float *dat = da + (long)b * na;
    const long set = (long)Cout * kk * Cin_pad;
    const float af = at[n];
    if (tid < 64) dsk[tid] = 0.f;
    if (tid < 16) dwk_s[tid] = 0.f;
    __syncthreads();
    float d_af = 0.f;
    float d_aw[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int e = tid; e < kk * Cin; e += 256) {
        const int t = e / Cin, c = e % Cin;
        const long idx = ((long)n * kk + t) * Cin_pad + c;
        const float G = dWb[(long)b * set + idx];
        const float as = at[Cout + t], ac = at[Cout + kk + c];
        float M = 0.f;
        const float GP = G * af * as * ac;
        for (int q = 0; q < K; ++q) {
            const float wv = Wk[q * set + idx];
            M += aw[q] * wv;
            if (q < 8) d_aw[q] += GP * wv;
            atomicAdd(dWk + q * set + idx, GP * aw[q]);
        }
        const float GM = G * M;
        d_af += GM * as * ac;
        atomicAdd(&dsk[t], GM * af * ac);
        atomicAdd(dat + Cout + kk + c, GM * af * as);                     // da_c: summed over n by the atomics
    }
    // block reductions
    for (int o = 32; o > 0; o >>= 1) d_af += __shfl_down(d_af, o);
    if ((tid & 63) == 0) red[tid >> 6] = d_af;
    for (int q = 0; q < K && q < 8; ++q) {
        float v = d_aw[q];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        if ((tid & 63) == 0) atomicAdd(&dwk_s[q], v);
    }
    __syncthreads();
    if (tid == 0) {
        dat[n] = (red[0] + red[1]) + (red[2] + red[3]);                  // da_f[b,n]: this workgroup owns it
        const float dbb = dbias_b ? dbias_b[(long)b * Cout + n] : 0.f;
        for (int q = 0; q < K; ++q) {
            float v = dwk_s[q];
            if (biask) { v += dbb * biask[q * Cout + n]; atomicAdd(dbiask + q * Cout + n, aw[q] * dbb); }
            atomicAdd(dat + Cout + kk + Cin + q, v);                      // da_w: summed over n
        }
    }
    if (tid < kk) atomicAdd(dat + Cout + tid, dsk[tid]);                  // da_s: summed over n
}

static inline int ew_grid(long items) {
    long g = (items + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace somi

using namespace somi;

extern "C" int somi_linear_f32(const float *x, int ldx, const float *W, const float *bias, int act, float *y, int ldy, int y_off, int B, int nin,
                               int nout, somi_stream_t stream) {
    SOMI_REQUIRE(x && W && y && B > 0 && nin > 0 && nout > 0 && nout <= 1024 && ldx >= nin && ldy >= y_off + nout, SOMI_EINVAL,
                 "linear: bad arguments (nout <= 1024)");
    SOMI_REQUIRE((act >= 0 && act <= 4) || (act == ACT_SOFTMAX && nout <= 64), SOMI_EINVAL, "linear: bad activation");
    hipLaunchKernelGGL(linear_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, ldx, W, bias, act, y, ldy, y_off, nin, nout);
    return launch_status("somi_linear_f32");
}

extern "C" int somi_linear_bwd_f32(const float *x, int ldx, const float *W, const float *dy, const float *y, int ld, int off, int act, float *dW,
                                   float *db, float *dx, int ldx_out, int dx_accumulate, float *workspace, int B, int nin, int nout,
                                   somi_stream_t stream) {
    SOMI_REQUIRE(x && W && dy && y && dW && workspace && B > 0 && nin > 0 && nout > 0, SOMI_EINVAL, "linear bwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    float *dpre = workspace;                                              // (B, nout)
    hipLaunchKernelGGL(act_bwd_rows_kernel, dim3(B), dim3(64), 0, s, dy, y, ld, off, act, dpre, nout, nout);
    hipLaunchKernelGGL(linear_bwd_weight_kernel, dim3(ew_grid((long)nout * nin + nout)), dim3(256), 0, s, x, ldx, dpre, nout, dW, db, B, nin, nout);
    if (dx) hipLaunchKernelGGL(linear_bwd_data_kernel, dim3(ew_grid((long)B * nin)), dim3(256), 0, s, W, dpre, nout, dx, ldx_out, dx_accumulate, B, nin, nout);
    return launch_status("somi_linear_bwd_f32");
}

extern "C" int somi_odconv_synth_f32(const float *attn, const float *Wk, const float *biask, float *wout, float *bout, int B, int Cin, int Cin_pad,
                                     int Cout, int kk, int K, somi_stream_t stream) {
    SOMI_REQUIRE(attn && Wk && wout && bout && B > 0 && Cin > 0 && Cin_pad >= Cin && Cin_pad % 4 == 0 && Cout > 0 && kk > 0 && K > 0 && K <= 16 &&
                     aligned16(Wk) && aligned16(wout), SOMI_EINVAL, "odconv synth: bad arguments");
    hipLaunchKernelGGL(odconv_synth_fwd_kernel, dim3(ew_grid((long)B * Cout * kk * (Cin_pad / 4))), dim3(256), 0, (hipStream_t)stream, attn, Wk, biask,
                       wout, bout, B, Cin, Cin_pad, Cout, kk, K);
    return launch_status("somi_odconv_synth_f32");
}

extern "C" int somi_odconv_synth_bwd_f32(const float *dWb, const float *attn, const float *Wk, const float *biask, const float *dbias_b, float *dWk,
                                         float *dbiask, float *dattn, int B, int Cin, int Cin_pad, int Cout, int kk, int K, somi_stream_t stream) {
    SOMI_REQUIRE(dWb && attn && Wk && dWk && dattn && B > 0 && Cin > 0 && Cin_pad >= Cin && Cout > 0 && kk > 0 && kk <= 64 && K > 0 && K <= 8, SOMI_EINVAL,
                 "odconv synth bwd: bad arguments (K <= 8, kk <= 64)");
    SOMI_REQUIRE(!biask || (dbias_b && dbiask), SOMI_EINVAL, "odconv synth bwd: bias gradients missing");
    hipStream_t s = (hipStream_t)stream;
    (void)hipMemsetAsync(dattn, 0, (size_t)B * (Cout + kk + Cin + K) * 4, s);
    hipLaunchKernelGGL(odconv_synth_bwd_kernel, dim3(Cout, B), dim3(256), 0, s, dWb, attn, Wk, biask, dbias_b, dWk, dbiask, dattn, Cin, Cin_pad, Cout, kk, K);
    return launch_status("somi_odconv_synth_bwd_f32");
}
