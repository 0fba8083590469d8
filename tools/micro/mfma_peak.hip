// Micro-benchmark: sustained v_mfma_f32_32x32x2_f32 rate on this MI355X (random operands), to price the fp32 conv kernel
// against what the chip actually sustains (DVFS) rather than against the 2.4 GHz datasheet number only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void mfma_loop(const float *in, float *out, int iters, unsigned long long *clk) {
    float a = in[threadIdx.x], b = in[threadIdx.x + 256];
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}
int main() {
    const int blocks = 256 * 2, iters = 20000;   // 2 workgroups per CU -> 2 waves per SIMD
    float *in, *out; unsigned long long *clk;
    hipMalloc(&in, 512 * 4); hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 16);
    std::vector<float> h(512); for (int i = 0; i < 512; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
    hipMemcpy(in, h.data(), 512 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, in, out, iters, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> c(blocks * 2); hipMemcpy(c.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
        double flops = (double)blocks * 4 * iters * 32.0 * 32 * 32 * 2 * 2 * 2;   // waves * iters * 32 mfma * flops(32x32x2)
        double ghz = (double)c[0] / ((double)c[1] / 100e6) / 1e9;
        printf("{\"rep\": %d, \"ms\": %.3f, \"TFLOPs\": %.1f, \"shader_GHz\": %.3f}\n", rep, ms, flops / ms / 1e9, ghz);
    }
    return 0;
}
