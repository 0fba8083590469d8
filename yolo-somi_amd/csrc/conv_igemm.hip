// NHWC implicit-GEMM convolution on the fp32 matrix cores of gfx950, with a fused epilogue.
//
// GEMM view:  M = B*Ho*Wo output pixels, N = Cout, K = kh*kw*Cin with k = (r*kw+q)*Cin + c.
//   A[m][k] = x[b, ho*s-p+r*d, wo*s-p+q*d, c]  (0 outside the image)  - gathered on the fly, never materialised
//   B[k][n] = w[n][k]                                                     - weights pre-packed K-contiguous
// Arithmetic: v_mfma_f32_32x32x2_f32 - exact fp32 products, fp32 accumulate (bit-for-bit an fmaf chain), so the
// result matches the reference CPU path to fp32 rounding (MI355X_MICROARCH.md, FP32-input MFMA: 157 TFLOP/s peak).
//
// Workgroup = 256 threads = 4 waves (one per SIMD).  Block tile BM x BN, K-tile 32:
//   global -> registers (16 B per lane, coalesced along c / k) -> LDS image [row][36] (32 k + 4 pad floats:
//   16 B-aligned rows, conflict-free for both the ds_write_b128 staging store and the ds_read_b128 fragment load)
//   -> each lane reads 4 consecutive k of one row per ds_read_b128 and feeds them to 4 consecutive MFMA k-steps.
//   The k order inside a K-tile is permuted identically for A and B (lane half h of k-step t takes k = 8j+4h+t),
//   which a sum over k does not care about.
// The next K-tile's global loads are issued before the MFMAs of the current one and written to LDS after them
// (register-staged software pipeline), 2-3 workgroups per CU cover the two barriers per K-tile.
//
// Reference being replaced: F.conv2d + BatchNorm2d(eval) + SiLU in Conv.forward (models/common.py:64-70,
// folded as utils/torch_utils.py:202-222), ODConv2d_3rd's grouped per-sample conv (models/common.py:4602-4605).
#include "common.h"

namespace somi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;          // K-tile (floats)
constexpr int LDS_LD = BK + 4;  // LDS row stride in floats (144 B)

struct ConvArgs {
    somi_conv_desc d;
    int M;        // rows per weight set: B*Ho*Wo, or Ho*Wo when per_sample_w (grid.z = B)
    int K;        // kh*kw*Cin
    int tiles_m, tiles_n;
};

template <int BM, int BN, int WAVES_M, int WAVES_N, bool MODULATE>
__global__ __launch_bounds__(256) void conv_igemm_f32_kernel(const ConvArgs a) {
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;   // wave tile
    constexpr int TM = WM / 32, TN = WN / 32;             // 32x32 MFMA tiles per wave
    constexpr int A_ROWS = BM / 32, B_ROWS = BN / 32;     // float4 loads per thread per K-tile
    static_assert(WAVES_M * WAVES_N == 4 && TM >= 1 && TN >= 1, "bad tiling");

    __shared__ __attribute__((aligned(16))) float lds[(BM + BN) * LDS_LD];
    float *As = lds, *Bs = lds + BM * LDS_LD;

    const somi_conv_desc &d = a.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    const int ntile = a.tiles_m * a.tiles_n;
    const int tile = xcd_remap(blockIdx.x, ntile);
    const int tile_m = tile / a.tiles_n, tile_n = tile % a.tiles_n;   // n fastest: neighbours share the A rows
    const int bz = blockIdx.z;                                        // weight set / image (per_sample_w)
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const float *__restrict__ wbase = d.w + (size_t)bz * d.Cout * a.K;

    // ---- per-thread load geometry: float4 column kc (4 floats) of rows row0 + 32*i
    const int kc = (tid & 7) * 4, row0 = tid >> 3;
    const float *a_base[A_ROWS];
    int a_hi0[A_ROWS], a_wi0[A_ROWS];
    const float *a_pix[A_ROWS];
    const float *a_chan[A_ROWS];
    const int HoWo = d.Ho * d.Wo;
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
        const int m = m0 + row0 + 32 * i;
        if (m < a.M) {
            const int b = m / HoWo + bz, rem = m % HoWo;
            const int ho = rem / d.Wo, wo = rem % d.Wo;
            a_hi0[i] = ho * d.stride - d.pad;
            a_wi0[i] = wo * d.stride - d.pad;
            a_base[i] = d.x + (size_t)b * d.H * d.W * d.x_cs + d.x_coff;
            if constexpr (MODULATE) {
                a_pix[i] = d.a_pix_scale ? d.a_pix_scale + (size_t)b * d.H * d.W : nullptr;
                a_chan[i] = d.a_chan_scale ? d.a_chan_scale + (size_t)b * d.Cin : nullptr;
            }
        } else {
            a_hi0[i] = -(1 << 28);   // forces the bounds test to fail -> zeros
            a_wi0[i] = 0;
            a_base[i] = d.x;
            if constexpr (MODULATE) { a_pix[i] = nullptr; a_chan[i] = nullptr; }
        }
    }
    const float *b_ptr[B_ROWS];
    bool b_ok[B_ROWS];
#pragma unroll
    for (int i = 0; i < B_ROWS; ++i) {
        const int n = n0 + row0 + 32 * i;
        b_ok[i] = n < d.Cout;
        b_ptr[i] = wbase + (size_t)(b_ok[i] ? n : 0) * a.K + kc;
    }

    // k -> (tap r,q ; channel c) for this thread's float4 column, advanced incrementally per K-tile
    int k = kc, c = kc % d.Cin, tap = kc / d.Cin;
    int r = tap / d.kw, q = tap % d.kw;

    f32x4 ra[A_ROWS], rb[B_ROWS];
    auto load_tile = [&]() {
        const bool kin = k < a.K;
        const int dh = r * d.dil, dw = q * d.dil;
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) {
            const int hi = a_hi0[i] + dh, wi = a_wi0[i] + dw;
            const bool ok = kin && (unsigned)hi < (unsigned)d.H && (unsigned)wi < (unsigned)d.W;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) {
                const size_t pix = (size_t)hi * d.W + wi;
                v = *reinterpret_cast<const f32x4 *>(a_base[i] + pix * d.x_cs + c);
                if constexpr (MODULATE) {
                    if (a_chan[i]) v *= *reinterpret_cast<const f32x4 *>(a_chan[i] + c);
                    if (a_pix[i]) v *= a_pix[i][pix];
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (kin && b_ok[i]) v = *reinterpret_cast<const f32x4 *>(b_ptr[i]);
            rb[i] = v;
        }
    };
    auto advance_k = [&]() {
        k += BK;
        c += BK;
        while (c >= d.Cin) {
            c -= d.Cin;
            if (++q == d.kw) { q = 0; ++r; }
        }
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i) b_ptr[i] += BK;
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i)
            *reinterpret_cast<f32x4 *>(&As[(row0 + 32 * i) * LDS_LD + kc]) = ra[i];
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i)
            *reinterpret_cast<f32x4 *>(&Bs[(row0 + 32 * i) * LDS_LD + kc]) = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nkt = (a.K + BK - 1) / BK;
    const int frag_off = (lane & 31) * LDS_LD + (lane >> 5) * 4;
    const float *Aw = As + (wm * WM) * LDS_LD + frag_off;
    const float *Bw = Bs + (wn * WN) * LDS_LD + frag_off;

    load_tile();
    store_tile();
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = kt + 1 < nkt;
        if (more) {
            advance_k();
            load_tile();           // in flight while the MFMAs below run
        }
#pragma unroll
        for (int j = 0; j < BK / 8; ++j) {
            f32x4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4 *>(Aw + i * 32 * LDS_LD + j * 8);
#pragma unroll
            for (int i = 0; i < TN; ++i) fb[i] = *reinterpret_cast<const f32x4 *>(Bw + i * 32 * LDS_LD + j * 8);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int jn = 0; jn < TN; ++jn)
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][t], fb[jn][t], acc[i][jn], 0, 0, 0);
        }
        __syncthreads();           // every wave is done reading this K-tile
        if (more) {
            store_tile();
            __syncthreads();
        }
    }

    // ---- epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const float *bias = d.bias ? d.bias + (size_t)bz * d.Cout : nullptr;
    const size_t row_base = (size_t)bz * a.M;
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
        const int n = n0 + wn * WN + jn * 32 + (lane & 31);
        if (n >= d.Cout) continue;
        const float bv = bias ? bias[n] : 0.f;
        const float ps = d.post_scale ? d.post_scale[n] : 1.f;
        const float pt = d.post_scale ? d.post_shift[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (m >= a.M) continue;
                float v = acc[i][jn][e] + bv;
                v = apply_act_rt(v, d.act);
                v = v * ps + pt;
                const size_t row = row_base + m;
                if (d.residual) v += d.residual[row * d.res_cs + d.res_coff + n];
                d.y[row * d.y_cs + d.y_coff + n] = v;
            }
        }
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
static int launch(const ConvArgs &a, hipStream_t s) {
    ConvArgs args = a;
    args.tiles_m = cdiv(a.M, BM);
    args.tiles_n = cdiv(a.d.Cout, BN);
    const dim3 grid(args.tiles_m * args.tiles_n, 1, a.d.per_sample_w ? a.d.B : 1);
    const bool mod = a.d.a_chan_scale || a.d.a_pix_scale;
    if (mod)
        hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WAVES_M, WAVES_N, true>), grid, dim3(256), 0, s, args);
    else
        hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WAVES_M, WAVES_N, false>), grid, dim3(256), 0, s, args);
    return launch_status("somi_conv2d_nhwc_f32");
}

// tile choice: widest N tile that Cout fills reasonably; small-M problems take the 64-row tile to fill the chip
static int pick_tile(const somi_conv_desc &d, int M) {
    const long blocks128 = (long)cdiv(M, 128) * cdiv(d.Cout, 128) * (d.per_sample_w ? d.B : 1);
    if (d.Cout > 64) return (blocks128 >= 512 || M >= 128 * 256) ? 0 : 1;
    if (d.Cout > 32) return 2;
    return 3;
}

}  // namespace somi

extern "C" int somi_conv2d_nhwc_f32(const somi_conv_desc *dp, somi_stream_t stream) {
    using namespace somi;
    SOMI_REQUIRE(dp, SOMI_EINVAL, "conv: null descriptor");
    const somi_conv_desc &d = *dp;
    SOMI_REQUIRE(d.x && d.w && d.y, SOMI_EINVAL, "conv: null tensor");
    SOMI_REQUIRE(d.B > 0 && d.H > 0 && d.W > 0 && d.Cin > 0 && d.Cout > 0, SOMI_EINVAL, "conv: empty shape");
    SOMI_REQUIRE(d.kh > 0 && d.kw > 0 && d.stride > 0 && d.dil > 0 && d.pad >= 0, SOMI_EINVAL, "conv: bad geometry");
    SOMI_REQUIRE(d.Ho == (d.H + 2 * d.pad - (d.dil * (d.kh - 1) + 1)) / d.stride + 1 &&
                     d.Wo == (d.W + 2 * d.pad - (d.dil * (d.kw - 1) + 1)) / d.stride + 1,
                 SOMI_EINVAL, "conv: Ho/Wo (%d,%d) do not match the geometry", d.Ho, d.Wo);
    SOMI_REQUIRE(d.Cin % 4 == 0 && d.x_cs % 4 == 0 && d.x_coff % 4 == 0 && aligned16(d.x) && aligned16(d.w),
                 SOMI_EINVAL, "conv: Cin (%d), x_cs (%d), x_coff (%d) must be multiples of 4 and bases 16 B aligned",
                 d.Cin, d.x_cs, d.x_coff);
    SOMI_REQUIRE(d.x_coff + d.Cin <= d.x_cs && d.y_coff + d.Cout <= d.y_cs, SOMI_EINVAL, "conv: channel slice out of range");
    SOMI_REQUIRE(!d.post_scale == !d.post_shift, SOMI_EINVAL, "conv: post_scale and post_shift go together");
    SOMI_REQUIRE(!d.residual || d.res_coff + d.Cout <= d.res_cs, SOMI_EINVAL, "conv: residual slice out of range");
    SOMI_REQUIRE(d.act >= SOMI_ACT_NONE && d.act <= SOMI_ACT_SIGMOID, SOMI_EINVAL, "conv: unknown activation %d", d.act);
    SOMI_REQUIRE((long)d.B * d.Ho * d.Wo < (1L << 31), SOMI_EINVAL, "conv: too many output pixels");
    if (d.a_chan_scale) SOMI_REQUIRE(aligned16(d.a_chan_scale), SOMI_EINVAL, "conv: a_chan_scale must be 16 B aligned");

    ConvArgs a;
    a.d = d;
    a.K = d.kh * d.kw * d.Cin;
    a.M = d.per_sample_w ? d.Ho * d.Wo : d.B * d.Ho * d.Wo;
    a.tiles_m = a.tiles_n = 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (pick_tile(d, a.M)) {
        case 0: return launch<128, 128, 2, 2>(a, s);
        case 1: return launch<64, 128, 1, 4>(a, s);
        case 2: return launch<128, 64, 2, 2>(a, s);
        default: return launch<128, 32, 4, 1>(a, s);
    }
}

extern "C" const char *somi_conv2d_kernel_name(const somi_conv_desc *dp) {
    if (!dp || dp->Cout <= 0 || dp->Ho <= 0 || dp->Wo <= 0 || dp->B <= 0) return nullptr;
    const int M = dp->per_sample_w ? dp->Ho * dp->Wo : dp->B * dp->Ho * dp->Wo;
    const bool mod = dp->a_chan_scale || dp->a_pix_scale;
    static const char *names[4][2] = {
        {"conv_igemm_f32_kernel<128,128,2,2,false>", "conv_igemm_f32_kernel<128,128,2,2,true>"},
        {"conv_igemm_f32_kernel<64,128,1,4,false>", "conv_igemm_f32_kernel<64,128,1,4,true>"},
        {"conv_igemm_f32_kernel<128,64,2,2,false>", "conv_igemm_f32_kernel<128,64,2,2,true>"},
        {"conv_igemm_f32_kernel<128,32,4,1,false>", "conv_igemm_f32_kernel<128,32,4,1,true>"}};
    return names[somi::pick_tile(*dp, M)][mod ? 1 : 0];
}
