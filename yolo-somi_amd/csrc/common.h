// Shared helpers for the libsomi_hip.so kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/somi_hip.h"

namespace somi {

void set_error(const char *fmt, ...);

#define SOMI_REQUIRE(cond, code, ...)            \
    do {                                         \
        if (!(cond)) {                           \
            ::somi::set_error(__VA_ARGS__);      \
            return (code);                       \
        }                                        \
    } while (0)

static inline int launch_status(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

constexpr int kNumXCD = 8;

// depthwise 3x3 stencil (layers.hip); flip = taps walked backwards (the data gradient)
void launch_dwconv3x3(bool flip, const float *x, const float *w, const float *bias, const float *ps, const float *pt, const float *res,
                      float *y, int B, int H, int W, int C, int act, hipStream_t s);

// Bijective XCD-aware remap of a 1-D block id: blocks with equal (id % 8) share an XCD (observed round-robin
// dispatch), so give each XCD one contiguous chunk of logical tile ids.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk / kNumXCD, r = nblk % kNumXCD;
    const int xcd = bid % kNumXCD, idx = bid / kNumXCD;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + idx;
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

// BiFPN fusion weights w_i / (sum_j swish(w_j) + eps) (models/common.py:3696, Swish :8210) from the raw parameter on the device
__device__ __forceinline__ void bifpn_norm(const float *__restrict__ w, int n_in, float eps, float (&wn)[3]) {
    float s = 0.f;
    for (int i = 0; i < n_in; ++i) s += w[i] * (1.0f / (1.0f + expf(-w[i])));
    for (int i = 0; i < 3; ++i) wn[i] = i < n_in ? w[i] / (s + eps) : 0.f;
}

template <int ACT>
__device__ __forceinline__ float apply_act(float v) {
    if constexpr (ACT == SOMI_ACT_SILU) return v / (1.0f + expf(-v));
    else if constexpr (ACT == SOMI_ACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    else if constexpr (ACT == SOMI_ACT_RELU) return fmaxf(v, 0.0f);
    else if constexpr (ACT == SOMI_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    else return v;
}

__device__ __forceinline__ float apply_act_rt(float v, int act) {
    switch (act) {
        case SOMI_ACT_SILU: return apply_act<SOMI_ACT_SILU>(v);
        case SOMI_ACT_GELU: return apply_act<SOMI_ACT_GELU>(v);
        case SOMI_ACT_RELU: return apply_act<SOMI_ACT_RELU>(v);
        case SOMI_ACT_SIGMOID: return apply_act<SOMI_ACT_SIGMOID>(v);
        default: return v;
    }
}

}  // namespace somi
