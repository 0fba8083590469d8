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
    if (nin <= 32) {
        // the attention heads (hidden 16 -> up to 1024 outputs): a thread per output - a wave per output left 48 lanes idle and walked 256 outputs each
        for (int o = threadIdx.x; o < nout; o += 256) {
            float s = 0.f;
            for (int i = 0; i < nin; ++i) s += W[(long)o * nin + i] * xr[i];
            pre[o] = s + (bias ? bias[o] : 0.f);
        }
    } else {
        for (int o = wave; o < nout; o += 4) {
            float s = 0.f;
            for (int i = lane; i < nin; i += 64) s += W[(long)o * nin + i] * xr[i];
            for (int sft = 32; sft > 0; sft >>= 1) s += __shfl_down(s, sft);
            if (lane == 0) pre[o] = s + (bias ? bias[o] : 0.f);
        }
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
#pragma unroll 8
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
// dx[b][i] (+)= sum_o dpre[b][o] W[o][i]: one wave per output, lanes stride the (up to ~1000) inputs of the sum, butterfly reduction - a fixed
// order.  (One thread per output walked them one after the other: 77 us for 512 outputs.)
__global__ __launch_bounds__(256) void linear_bwd_data_kernel(const float *__restrict__ W, const float *__restrict__ dpre, int ldp, float *__restrict__ dx,
                                                              int ldx, int accumulate, int B, int nin, int nout) {
    const long n = (long)B * nin;
    const int lane = threadIdx.x & 63;
    const long wave = (blockIdx.x * 256L + threadIdx.x) >> 6, nwave = (long)gridDim.x * 4;
    for (long e = wave; e < n; e += nwave) {
        const int b = (int)(e / nin), i = (int)(e % nin);
        float s = 0.f;
        for (int o = lane; o < nout; o += 64) s += dpre[(long)b * ldp + o] * W[(long)o * nin + i];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) dx[(long)b * ldx + i] = (accumulate ? dx[(long)b * ldx + i] : 0.f) + s;
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

// Backward of the synthesis  W_b[n,t,c] = a_f[b,n] a_s[b,t] a_c[b,c] sum_q a_w[b,q] Wk[q,n,t,c]  without atomics (run-to-run
// bit-identical), in three kernels:
//   candidates  one lane per 4 weights (n,t,c..c+3): dWk[q] += sum_b G a_f a_s a_c a_w[q], samples in ascending order; the lanes of
//               (t=0,c=0) also own dbiask[q,n] += sum_b a_w[b,q] dbias_b[b,n]
//   partials    one workgroup per (n, b) sweeps the (t, c) plane of G = dW_b and M = sum_q a_w Wk: da_f[b,n] is final here;
//               P_s[b,n,t] = sum_c G M a_c, P_c[b,n,c] = sum_t G M a_s, P_w[b,n,q] = sum_{t,c} G a_s a_c Wk[q] go to the workspace
//               (each lane owns whole (t-column, channel) runs, block sums are fixed-shape trees)
//   finish      one workgroup per sample: da_s[t] = sum_n a_f P_s, da_c[c] = sum_n a_f P_c, da_w[q] = sum_n (a_f P_w + dbias_b biask[q]),
//               n ascending
__global__ __launch_bounds__(256) void odconv_synth_bwd_cand_kernel(const float *__restrict__ dWb, const float *__restrict__ ws, const float *__restrict__ biask,
                                                                    const float *__restrict__ dbias_b, float *dWk, float *dbiask, int B, int Cin,
                                                                    int Cin_pad, int Cout, int kk, int K) {
    const int C4 = Cin_pad >> 2, na = Cout + kk + Cin + K;
    const long per = (long)Cout * kk * C4, set = (long)Cout * kk * Cin_pad;
    for (long it = blockIdx.x * 256L + threadIdx.x; it < per; it += (long)gridDim.x * 256) {
        const int c = (int)(it % C4) * 4, t = (int)((it / C4) % kk), n = (int)(it / ((long)C4 * kk));
        const long idx = ((long)n * kk + t) * Cin_pad + c;
        f32x4 acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        float bacc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const bool bias_lane = biask && t == 0 && c == 0;
        for (int b = 0; b < B; ++b) {
            const float *at = ws + (long)b * na, *aw = at + Cout + kk + Cin;
            const float fs = at[n] * at[Cout + t];
            f32x4 ac;
#pragma unroll
            for (int e = 0; e < 4; ++e) ac[e] = (c + e < Cin) ? at[Cout + kk + c + e] : 0.f;
            const f32x4 gp = *reinterpret_cast<const f32x4 *>(dWb + (long)b * set + idx) * (fs * ac);
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (q < K) acc[q] += gp * aw[q];
            if (bias_lane) {
                const float dbb = dbias_b[(long)b * Cout + n];
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (q < K) bacc[q] += aw[q] * dbb;
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (q < K) {
                float *o = dWk + q * set + idx;
                *reinterpret_cast<f32x4 *>(o) = *reinterpret_cast<const f32x4 *>(o) + acc[q];
                if (bias_lane) dbiask[q * Cout + n] += bacc[q];
            }
        }
    }
}

__device__ __forceinline__ float block_sum_256(float v, float *red /* >= 4 floats */) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    __syncthreads();                                                     // red may still be read from the previous call
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void odconv_synth_bwd_part_kernel(const float *__restrict__ dWb, const float *__restrict__ ws, const float *__restrict__ Wk,
                                                                    float *__restrict__ da, float *__restrict__ part, int Cin, int Cin_pad, int Cout,
                                                                    int kk, int K) {
    __shared__ float red[4];
    const int n = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int na = Cout + kk + Cin + K, np = kk + Cin + K;
    const float *at = ws + (long)b * na, *aw = at + Cout + kk + Cin;
    float *pb = part + ((long)b * Cout + n) * np;                        // [P_s (kk) | P_c (Cin) | P_w (K)]
    const long set = (long)Cout * kk * Cin_pad;
    float d_af = 0.f, d_aw[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float pc[4] = {0.f, 0.f, 0.f, 0.f};                                  // a lane owns channels tid, tid+256, ... (Cin <= 1024): P_c needs no exchange
    for (int t = 0; t < kk; ++t) {
        const float as = at[Cout + t];
        float pst = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + 256 * i;
            if (c < Cin) {
                const float ac = at[Cout + kk + c];
                const long idx = ((long)n * kk + t) * Cin_pad + c;
                const float G = dWb[(long)b * set + idx];
                float M = 0.f;
                for (int q = 0; q < K; ++q) {
                    const float wv = Wk[q * set + idx];
                    M += aw[q] * wv;
                    d_aw[q] += G * as * ac * wv;
                }
                const float GM = G * M;
                d_af += GM * as * ac;
                pst += GM * ac;
                pc[i] += GM * as;
            }
        }
        const float v = block_sum_256(pst, red);
        if (tid == 0) pb[t] = v;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (tid + 256 * i < Cin) pb[kk + tid + 256 * i] = pc[i];
    const float f = block_sum_256(d_af, red);
    if (tid == 0) da[(long)b * na + n] = f;                              // da_f[b,n]: this workgroup owns it
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        if (q < K) {
            const float v = block_sum_256(d_aw[q], red);
            if (tid == 0) pb[kk + Cin + q] = v;
        }
    }
}

__global__ __launch_bounds__(256) void odconv_synth_bwd_finish_kernel(const float *__restrict__ ws, const float *__restrict__ part, const float *__restrict__ biask,
                                                                      const float *__restrict__ dbias_b, float *__restrict__ da, int Cin, int Cout, int kk,
                                                                      int K) {
    const int b = blockIdx.x, na = Cout + kk + Cin + K, np = kk + Cin + K;
    const float *at = ws + (long)b * na;
    for (int i = threadIdx.x; i < np; i += 256) {                        // i indexes [da_s | da_c | da_w] exactly like the partial rows
        float s = 0.f;
        const bool is_w = i >= kk + Cin;
#pragma unroll 8
        for (int n = 0; n < Cout; ++n) {                                 // unrolled: eight rows' loads in flight, sums in row order
            s += at[n] * part[((long)b * Cout + n) * np + i];
            if (is_w && biask) s += dbias_b[(long)b * Cout + n] * biask[(i - kk - Cin) * Cout + n];
        }
        da[(long)b * na + Cout + i] = s;
    }
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
    if (dx) hipLaunchKernelGGL(linear_bwd_data_kernel, dim3(ew_grid((long)B * nin * 64)), dim3(256), 0, s, W, dpre, nout, dx, ldx_out, dx_accumulate, B, nin, nout);
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

extern "C" size_t somi_odconv_synth_bwd_workspace_floats(int B, int Cin, int Cout, int kk, int K) { return (size_t)B * Cout * (kk + Cin + K); }

extern "C" int somi_odconv_synth_bwd_f32(const float *dWb, const float *attn, const float *Wk, const float *biask, const float *dbias_b, float *dWk,
                                         float *dbiask, float *dattn, float *workspace, int B, int Cin, int Cin_pad, int Cout, int kk, int K,
                                         somi_stream_t stream) {
    SOMI_REQUIRE(dWb && attn && Wk && dWk && dattn && workspace && B > 0 && Cin > 0 && Cin_pad >= Cin && Cin_pad % 4 == 0 && Cout > 0 && kk > 0 &&
                     kk <= 64 && K > 0 && K <= 8 && Cin <= 1024 && aligned16(dWb) && aligned16(dWk), SOMI_EINVAL,
                 "odconv synth bwd: bad arguments (K <= 8, kk <= 64, Cin <= 1024)");
    SOMI_REQUIRE(!biask || (dbias_b && dbiask), SOMI_EINVAL, "odconv synth bwd: bias gradients missing");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(odconv_synth_bwd_cand_kernel, dim3(ew_grid((long)Cout * kk * (Cin_pad / 4))), dim3(256), 0, s, dWb, attn, biask, dbias_b, dWk, dbiask,
                       B, Cin, Cin_pad, Cout, kk, K);
    hipLaunchKernelGGL(odconv_synth_bwd_part_kernel, dim3(Cout, B), dim3(256), 0, s, dWb, attn, Wk, dattn, workspace, Cin, Cin_pad, Cout, kk, K);
    hipLaunchKernelGGL(odconv_synth_bwd_finish_kernel, dim3(B), dim3(256), 0, s, attn, workspace, biask, dbias_b, dattn, Cin, Cout, kk, K);
    return launch_status("somi_odconv_synth_bwd_f32");
}
