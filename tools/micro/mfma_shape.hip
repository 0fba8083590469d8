// Micro-benchmark: does the fp32 MFMA *shape* change what the chip sustains under load?  MI355X_MICROARCH.md (DVFS give-back, item 7)
// reports 1.12-1.15x the FLOP/s for the 16x16 bf16 shape over the 32x32 one at equal cycles per FLOP (the chip holds a higher clock).
// This loop is the conv kernel's inner loop without global traffic: 512-thread workgroups, 2 per CU, operands re-read from a
// [row][36] LDS image by ds_read_b128 (random data), wave tile 64 x 32, v_mfma_f32_32x32x2_f32 vs v_mfma_f32_16x16x4_f32.
// Prints TFLOP/s and the in-kernel shader clock for both.   build: hipcc --offload-arch=gfx950 -O3 -o mfma_shape mfma_shape.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int LD = 36, ROWS = 256, TILE = ROWS * LD;

template <int SHAPE>   // 32 or 16
__global__ __launch_bounds__(512, 4) void loop_kernel(const float *in, float *out, int iters, unsigned long long *clk) {
    __shared__ __attribute__((aligned(16))) float lds[2 * TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * TILE; i += 512) lds[i] = in[(i * 7 + blockIdx.x) % 65536];
    __syncthreads();
    const int wm = wave >> 2, wn = wave & 3;
    unsigned long long t0 = 0, r0 = 0;
    float s = 0.f;
    if constexpr (SHAPE == 32) {
        f32x16 acc[2] = {};
        const int frag = (lane & 31) * LD + (lane >> 5) * 4;
        const float *pa = lds + (wm * 64) * LD + frag, *pb = lds + (128 + wn * 32) * LD + frag;
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it) {
            const float *a = pa + (it & 1) * TILE, *b = pb + (it & 1) * TILE;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 fa0 = *reinterpret_cast<const f32x4 *>(a + j * 8), fa1 = *reinterpret_cast<const f32x4 *>(a + 32 * LD + j * 8);
                const f32x4 fb = *reinterpret_cast<const f32x4 *>(b + j * 8);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[t], fa0[t], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[t], fa1[t], acc[1], 0, 0, 0);
                }
            }
        }
        for (int i = 0; i < 2; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    } else {
        f32x4 acc[4][2] = {};
        // lane (row = lane & 15, quarter q = lane >> 4) reads the 16 B column 4*jj + q of its row; columns XOR-swizzled by (row >> 1) & 7
        const int row = lane & 15, q = lane >> 4;
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it) {
            const float *base = lds + (it & 1) * TILE;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                f32x4 fa[4], fb[2];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = wm * 64 + i * 16 + row;
                    fa[i] = *reinterpret_cast<const f32x4 *>(base + r * 32 + (((4 * jj + q) ^ ((r >> 1) & 7)) << 2));
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int r = 128 + wn * 32 + i * 16 + row;
                    fb[i] = *reinterpret_cast<const f32x4 *>(base + r * 32 + (((4 * jj + q) ^ ((r >> 1) & 7)) << 2));
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[j][t], fa[i][t], acc[i][j], 0, 0, 0);
            }
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 4; ++e) s += acc[i][j][e];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 512 + tid] = s;
    if (tid == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int SHAPE>
void run(const float *in, float *out, unsigned long long *clk, int blocks, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((loop_kernel<SHAPE>), dim3(blocks), dim3(512), 0, 0, in, out, iters, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> c(blocks * 2); hipMemcpy(c.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
        // per iteration and wave: 64 x 32 x 32 (k) x 2 FLOP
        const double flops = (double)blocks * 8 * iters * 64.0 * 32 * 32 * 2;
        std::vector<double> g;
        for (int b = 0; b < blocks; ++b) g.push_back((double)c[2 * b] / ((double)c[2 * b + 1] / 100e6) / 1e9);
        std::sort(g.begin(), g.end());
        printf("{\"mfma\": \"%s\", \"rep\": %d, \"ms\": %.3f, \"TFLOPs\": %.1f, \"shader_GHz_median\": %.3f}\n",
               SHAPE == 32 ? "32x32x2_f32" : "16x16x4_f32", rep, ms, flops / ms / 1e9, g[g.size() / 2]);
    }
}

int main() {
    const int blocks = 512, iters = 6000;
    float *in, *out; unsigned long long *clk;
    hipMalloc(&in, 65536 * 4); hipMalloc(&out, blocks * 512 * 4); hipMalloc(&clk, blocks * 16);
    std::vector<float> h(65536);
    unsigned x = 12345u;
    for (auto &v : h) { x = x * 1664525u + 1013904223u; v = (float)((x >> 8) & 0xFFFF) / 32768.f - 1.f; }
    hipMemcpy(in, h.data(), 65536 * 4, hipMemcpyHostToDevice);
    for (int round = 0; round < 2; ++round) {
        run<32>(in, out, clk, blocks, iters);
        run<16>(in, out, clk, blocks, iters);
    }
    return 0;
}
