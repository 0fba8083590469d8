// Weight gradient of the NHWC convolution on the fp32 matrix cores of gfx950 (autograd of F.conv2d w.r.t. its weight,
// train.py:270 `scaler.scale(loss).backward()`):
//   dW[co][(r*kw+q)*Cin + ci] = sum over pixels (b,ho,wo) of dy[b,ho,wo,co] * x[b, ho*s-p+r, wo*s-p+q, ci]
// GEMM view: rows = co, columns = k = (tap, ci) flattened exactly as the packed weight, reduction = pixels.  A workgroup (256
// threads, 4 waves) owns a BM (co) x BN (k) tile and a contiguous slice of the pixel range (split-K); a column tile may span
// several taps (each thread's 16 B column quad has its own fixed tap), so narrow layers (Cin 4 / 64 / 192) still fill whole
// tiles.  Every 32-pixel K-tile it stages dy[32][BM] and the tap-shifted x[32][BN] rows (16 B buffer loads, padding by the
// hardware range check) into a double-buffered LDS image
// [pixel][channel] and feeds v_mfma_f32_32x32x2_f32 with one ds_read_b32 per operand value (lanes = consecutive channels,
// conflict-free).  The image is lane-linear (thread t owns floats [4t, 4t+4) of every pass), so it is filled by LDS-DMA
// (`buffer_load_dwordx4 ... lds`): no VGPR staging and no ds_write pass.  Partial tiles of the splits go to a workspace and are summed in a fixed order (deterministic), optionally
// on top of an existing gradient.
#include <stdlib.h>
#include "common.h"

namespace somi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short i16x4 __attribute__((ext_vector_type(4)));
typedef short i16x8 __attribute__((ext_vector_type(8)));

constexpr int WG_PIX = 32;                  // pixels per K-tile
constexpr unsigned W_OOB = 0xFFFFFFE0u;
constexpr unsigned W_MAX_BUF = 0xE0000000u;

struct WgradArgs {
    const float *x, *dy;
    float *out;                              // [splits][n_sets][Cout][K]  (workspace) or dW itself when splits == 1
    int B, H, W, Cin, x_cs, x_coff, Ho, Wo, Cout, dy_cs, dy_coff, kh, kw, stride, pad;
    int K;                                   // kh*kw*Cin
    int tiles_co, tiles_k, splits, pix_per_split, npix;    // npix = pixels per weight set (B*Ho*Wo, or Ho*Wo per sample)
    int bm, bn;                              // tile variant
    int per_sample;
    unsigned x_bytes, dy_bytes;
    int ns;                                  // 0: exact fp32; 1: bf16; 2: bf16x3 (128 x 128 tile, shared weights)
};

__device__ __forceinline__ f32x4 wbuf_load4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}

// LDS-DMA: 16 B per lane from a buffer descriptor straight into LDS at `lds_base` (wave-uniform) + 16 * lane; completion is tracked by
// the vector-memory counter.  The builtin and the wait only exist in the device pass (the host pass just needs the kernel's stub).
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t r, float *lds_base, unsigned voff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)lds_base, 16, voff, 0, 0, 0);
#endif
}
__device__ __forceinline__ void lds_dma_wait() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

// transposed LDS read (gfx950 ds_read_b64_tr_b16): per 16-lane group a block of 4 rows x 16 columns of 16-bit values, delivered
// column-major - lane i of the group receives column i of the 4 rows.  Device pass only (the host pass just needs the kernel's stub).
__device__ __forceinline__ i16x4 lds_read_tr16(const char *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4 *)p);
#else
    return i16x4{0, 0, 0, 0};
#endif
}

// NS (opt-in reduced precision, see conv_igemm.hip): 0 = exact fp32; 1 = bf16 operands; 2 = bf16x3 split.  The bf16 forms (128 x 128 tile,
// 8 waves) convert dy / x while they are staged into a [pixel][channel] bf16 image (256-byte rows, 16-byte chunks XOR-swizzled by
// ((row & 3) << 2) | ((row >> 2) & 3)) and read their MFMA operands - 8 consecutive PIXELS of one channel per lane - with the hardware
// transposing read.
template <int BM, int BN, int NW = 4, int NS = 0>   // co rows x k columns per tile: 128 x {128,64,32} or 64 x {128,64}; NW waves (8: 128 x 128 only)
__global__ __launch_bounds__(NW * 64, NW / 2) void conv_wgrad_f32_kernel(const WgradArgs a) {
    constexpr int NT = NW * 64;
    static_assert(NS == 0 || ((BM == 128 || BM == 64) && BN == 128 && NW == 8), "the bf16 forms exist for the 128 / 64 x 128 tiles with 8 waves");
    constexpr int WAVES_N = NW == 8 ? 4 : (BM == 64 ? 2 : (BN == 128 ? 2 : 1)), WAVES_M = NW / WAVES_N;
    static_assert(BM / WAVES_M >= 32 && BN / WAVES_N >= 32, "bad wgrad tiling");
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, TM = WM / 32, TN = WN / 32;
    constexpr int A_LD = BM, B_LD = BN;
    constexpr int TILE = WG_PIX * (A_LD + B_LD);
    // (the bf16 image always has 256-byte rows - a 64-channel dy tile leaves the upper half of its rows to the swizzle - and four 8 KB planes)
    constexpr int BUF = NS ? 8192 : TILE;                                // floats per buffer
    __shared__ __attribute__((aligned(16))) float lds[2 * BUF];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    // tile decode: logical id = split * tiles + tile_co * tiles_k + tile_k (blockIdx.z = weight set).  The XCD remap hands each
    // XCD a contiguous run of logical ids, so all tiles of one pixel split - which read the same dy and x rows - share an L2.
    const int ntile = a.tiles_co * a.tiles_k;
    const int lid = xcd_remap(blockIdx.x, ntile * a.splits);
    const int tile = lid % ntile, split = lid / ntile, set = blockIdx.z;
    const int tile_k = tile % a.tiles_k, tile_co = tile / a.tiles_k;
    const int co0 = tile_co * BM, k0 = tile_k * BN;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc((void *)a.dy, 0, a.dy_bytes, 0x00020000);

    // pixel range of this split inside the weight set
    const int p_lo = split * a.pix_per_split, p_hi = min(p_lo + a.pix_per_split, a.npix);
    const long set_pix0 = (long)set * a.npix;                            // first pixel of this set in the flattened (b,ho,wo) order

    // fetch geometry.  A (dy): 32 pixels x BM floats = BM/4 quads per pixel.  thread -> (pixel row, quad), rows strided.
    constexpr int A_QUADS = BM / 4, B_QUADS = BN / 4;
    constexpr int A_ROWS_PER_PASS = NT / A_QUADS, B_ROWS_PER_PASS = NT / B_QUADS;        // 8 ; 8,16,32 (4 waves)
    const int a_quad = tid % A_QUADS, a_row0 = tid / A_QUADS;
    const int b_quad = tid % B_QUADS, b_row0 = tid / B_QUADS;
    const int b_col = k0 + b_quad * 4;                                    // this thread's column quad: one tap, 4 channels
    const bool a_col_ok = co0 + a_quad * 4 < a.Cout, b_col_ok = b_col < a.K;
    const int b_tap = b_col / a.Cin, b_ci = b_col % a.Cin;
    const int r = b_tap / a.kw, q = b_tap % a.kw;
    constexpr int A_N = WG_PIX / A_ROWS_PER_PASS, B_N = WG_PIX / B_ROWS_PER_PASS;        // 4 ; 4,2,1

    // per B-row pixel coordinates (b, ho, wo) advanced incrementally by WG_PIX per K-tile
    int bb[B_N], bho[B_N], bwo[B_N];
    const int HoWo = a.Ho * a.Wo;
#pragma unroll
    for (int i = 0; i < B_N; ++i) {
        const long P = set_pix0 + p_lo + b_row0 + i * B_ROWS_PER_PASS;
        bb[i] = (int)(P / HoWo);
        const int rem = (int)(P % HoWo);
        bho[i] = rem / a.Wo;
        bwo[i] = rem % a.Wo;
    }
    // all offsets are 32-bit: both tensors fit their 4 GiB buffer descriptors (checked by the launcher)
    const unsigned set_p0 = (unsigned)(set_pix0 + p_lo);                  // first pixel of this split in the flattened order
    const unsigned a_col = (unsigned)(a.dy_coff + co0 + a_quad * 4), b_colo = (unsigned)(a.x_coff + b_ci);
    const int npl = p_hi - p_lo;
    auto advance = [&]() {
#pragma unroll
        for (int i = 0; i < B_N; ++i) {
            bwo[i] += WG_PIX;
            while (bwo[i] >= a.Wo) {
                bwo[i] -= a.Wo;
                if (++bho[i] == a.Ho) { bho[i] = 0; ++bb[i]; }
            }
        }
    };
    // LDS-DMA: thread t's 16 B of pass i land at float 4t of that pass - one wave instruction = 1 KiB at a wave-uniform base
    auto dma = [&](float *buf, int pt) {
        float *wbase = buf + wave * 256;
#pragma unroll
        for (int i = 0; i < A_N; ++i) {
            const int pl = pt + a_row0 + i * A_ROWS_PER_PASS;
            const bool ok = a_col_ok && pl < npl;
            const unsigned off = ((set_p0 + (unsigned)pl) * (unsigned)a.dy_cs + a_col) * 4u;
            lds_dma16(rdy, wbase + i * A_ROWS_PER_PASS * A_LD, ok ? off : W_OOB);
        }
#pragma unroll
        for (int i = 0; i < B_N; ++i) {
            const int pl = pt + b_row0 + i * B_ROWS_PER_PASS;
            const int hi = bho[i] * a.stride - a.pad + r, wi = bwo[i] * a.stride - a.pad + q;
            const bool ok = b_col_ok && pl < npl && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
            const unsigned off = ((unsigned)((bb[i] * a.H + hi) * a.W + wi) * (unsigned)a.x_cs + b_colo) * 4u;
            lds_dma16(rx, wbase + WG_PIX * A_LD + i * B_ROWS_PER_PASS * B_LD, ok ? off : W_OOB);
        }
    };

    // bf16 forms: staging through registers (fetch -> convert -> ds_write_b64)
    f32x4 ra[A_N], rb[B_N];
    auto fetch = [&](int pt) {
#pragma unroll
        for (int i = 0; i < A_N; ++i) {
            const int pl = pt + a_row0 + i * A_ROWS_PER_PASS;
            const bool ok = a_col_ok && pl < npl;
            const unsigned off = ((set_p0 + (unsigned)pl) * (unsigned)a.dy_cs + a_col) * 4u;
            ra[i] = wbuf_load4(rdy, ok ? off : W_OOB);
        }
#pragma unroll
        for (int i = 0; i < B_N; ++i) {
            const int pl = pt + b_row0 + i * B_ROWS_PER_PASS;
            const int hi = bho[i] * a.stride - a.pad + r, wi = bwo[i] * a.stride - a.pad + q;
            const bool ok = b_col_ok && pl < npl && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
            const unsigned off = ((unsigned)((bb[i] * a.H + hi) * a.W + wi) * (unsigned)a.x_cs + b_colo) * 4u;
            rb[i] = wbuf_load4(rx, ok ? off : W_OOB);
        }
    };
    // image of one buffer (bytes): dy hi [32 px][128] bf16 | dy lo | x hi | x lo, 8 KB each
    auto split_store = [&](char *img, int row, int quad, const f32x4 &v) {
        const int key = ((row & 3) << 2) | ((row >> 2) & 3);
        char *p = img + 256 * row + 16 * ((quad >> 1) ^ key) + 8 * (quad & 1);
        const __bf16 h0 = (__bf16)v[0], h1 = (__bf16)v[1], h2 = (__bf16)v[2], h3 = (__bf16)v[3];
        *reinterpret_cast<bf16x4 *>(p) = bf16x4{h0, h1, h2, h3};
        if constexpr (NS == 2)
            *reinterpret_cast<bf16x4 *>(p + 8192) = bf16x4{(__bf16)(v[0] - (float)h0), (__bf16)(v[1] - (float)h1), (__bf16)(v[2] - (float)h2),
                                                           (__bf16)(v[3] - (float)h3)};
    };
    auto store_bf16 = [&](float *buf) {
        char *img = reinterpret_cast<char *>(buf);
#pragma unroll
        for (int i = 0; i < A_N; ++i) split_store(img, a_row0 + i * A_ROWS_PER_PASS, a_quad, ra[i]);
#pragma unroll
        for (int i = 0; i < B_N; ++i) split_store(img + 16384, b_row0 + i * B_ROWS_PER_PASS, b_quad, rb[i]);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nkt = (p_hi - p_lo + WG_PIX - 1) / WG_PIX;
    const int a_frag = (lane >> 5) * A_LD + wm * WM + (lane & 31);
    const int b_frag = WG_PIX * A_LD + (lane >> 5) * B_LD + wn * WN + (lane & 31);
    auto mma_quarter = [&](const float *buf, int qtr) {                   // 4 of the 16 k-steps (8 pixels)
#pragma unroll
        for (int ks = qtr * 4; ks < qtr * 4 + 4; ++ks) {
            float fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = buf[a_frag + ks * 2 * A_LD + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = buf[b_frag + ks * 2 * B_LD + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    };

    if constexpr (NS != 0) {
        // lane geometry of the transposed reads: group half cg = channels +16, q = pixel row of the 4-row block, p = 8-byte piece
        const int h = lane >> 5, cg = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
        int lpart[2];                                                 // per 4-pixel block jj: row bytes + the low chunk bits + the piece
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int keylo = (2 * h + jj) & 3;                       // (row >> 2) & 3 of row 16 st + 8 h + 4 jj + q
            lpart[jj] = 256 * (8 * h + 4 * jj + q4) + 16 * ((2 * cg + (p4 >> 1)) ^ keylo) + 8 * (p4 & 1);
        }
        int ta[TM], tb[TN];                                           // the high chunk bits of a 32-channel tile, swizzled by (row & 3) << 2 = q << 2
#pragma unroll
        for (int i = 0; i < TM; ++i) ta[i] = 16 * ((((wm * WM + i * 32) >> 3)) ^ (q4 << 2));
#pragma unroll
        for (int j = 0; j < TN; ++j) tb[j] = 16384 + 16 * ((((wn * WN + j * 32) >> 3)) ^ (q4 << 2));
        auto frag = [&](const char *img, int off) -> bf16x8 {
            const i16x4 lo4 = lds_read_tr16(img + off + lpart[0]), hi4 = lds_read_tr16(img + off + lpart[1]);
            return __builtin_bit_cast(bf16x8, i16x8{lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]});
        };
        if (nkt > 0) {
            fetch(0);
            store_bf16(lds);
        }
        __syncthreads();
        for (int kt = 0; kt < nkt; ++kt) {
            const char *cur = reinterpret_cast<const char *>(lds + (kt & 1) * BUF);
            float *nxt = lds + ((kt + 1) & 1) * BUF;
            const bool more = kt + 1 < nkt;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ah[i] = frag(cur, 4096 * st + ta[i]);
                    if constexpr (NS == 2) al[i] = frag(cur, 8192 + 4096 * st + ta[i]);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    bh[j] = frag(cur, 4096 * st + tb[j]);
                    if constexpr (NS == 2) bl[j] = frag(cur, 8192 + 4096 * st + tb[j]);
                }
                if (st == 0 && more) {
                    advance();
                    fetch((kt + 1) * WG_PIX);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if constexpr (NS == 2) {                      // the small cross terms first
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                        }
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    }
            }
            if (more) store_bf16(nxt);
            __syncthreads();
        }
    }
    if (NS == 0 && nkt > 0) {
        dma(lds, 0);
        lds_dma_wait();               // the DMA's LDS writes are tracked by the vector-memory counter
    }
    __syncthreads();
    for (int kt = 0; NS == 0 && kt < nkt; ++kt) {
        const float *cur = lds + (kt & 1) * TILE;
        float *nxt = lds + ((kt + 1) & 1) * TILE;
        const bool more = kt + 1 < nkt;
        mma_quarter(cur, 0);
        if (more) {
            advance();
            dma(nxt, (kt + 1) * WG_PIX);                               // lands behind the rest of this K-tile's MFMAs
        }
        mma_quarter(cur, 1);
        mma_quarter(cur, 2);
        mma_quarter(cur, 3);
        lds_dma_wait();
        __syncthreads();
    }

    // D[co][k]: col = lane&31 -> k, row = (e&3) + 8*(e>>2) + 4*(lane>>5) -> co
    float *out = a.out + ((size_t)split * gridDim.z + set) * a.Cout * a.K;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int kcol = k0 + wn * WN + j * 32 + (lane & 31);
        if (kcol >= a.K) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = co0 + wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (co < a.Cout) out[(size_t)co * a.K + kcol] = acc[i][j][e];
            }
        }
    }
}

// dW = (accumulate ? accumulate : 0) + sum_s part[s]   (fixed order: per group two interleaved partial sums, the 16 groups combined 0..15).
// 16 float4 columns x 16 split groups per workgroup, four rows requested per trip: the walk over the ~100 split partials of a layer is a chain
// of memory latencies (4 groups with two rows per trip took 13.6 us per launch, 140 launches per step).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ part, int splits, long n, const float *accumulate,
                                                           float *__restrict__ dw) {
    __shared__ f32x4 sm[16][16];
    const long n4 = n >> 2;
    const int col = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (long i0 = blockIdx.x * 16L; i0 < n4; i0 += (long)gridDim.x * 16) {
        const long i = i0 + col;
        f32x4 s0 = zero, s1 = zero;
        if (i < n4) {
            for (int k = grp; k < splits; k += 64) {
                f32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    v[u] = k + u * 16 < splits ? *reinterpret_cast<const f32x4 *>(part + (size_t)(k + u * 16) * n + i * 4) : zero;
                s0 += v[0];
                s1 += v[1];
                s0 += v[2];
                s1 += v[3];
            }
        }
        sm[grp][col] = s0 + s1;
        __syncthreads();
        if (grp == 0 && i < n4) {
            f32x4 s = accumulate ? *reinterpret_cast<const f32x4 *>(accumulate + i * 4) : zero;
#pragma unroll
            for (int g = 0; g < 16; ++g) s += sm[g][col];
            *reinterpret_cast<f32x4 *>(dw + i * 4) = s;
        }
        __syncthreads();
    }
}

static int plan(const somi_conv_desc &f, WgradArgs &a) {
    SOMI_REQUIRE(f.B > 0 && f.H > 0 && f.W > 0 && f.Cin > 0 && f.Cout > 0 && f.kh > 0 && f.kw > 0 && f.stride > 0 && f.pad >= 0,
                 SOMI_EINVAL, "conv wgrad: bad geometry");
    SOMI_REQUIRE(f.dil == 1, SOMI_ENOTIMPL, "conv wgrad: dilation 1 only");
    SOMI_REQUIRE(f.Ho == (f.H + 2 * f.pad - f.kh) / f.stride + 1 && f.Wo == (f.W + 2 * f.pad - f.kw) / f.stride + 1, SOMI_EINVAL,
                 "conv wgrad: Ho/Wo do not match the forward geometry");
    SOMI_REQUIRE(f.Cin % 4 == 0 && f.Cout % 4 == 0, SOMI_EINVAL, "conv wgrad: Cin and Cout must be multiples of 4");
    a.B = f.B; a.H = f.H; a.W = f.W; a.Cin = f.Cin; a.Ho = f.Ho; a.Wo = f.Wo; a.Cout = f.Cout;
    a.kh = f.kh; a.kw = f.kw; a.stride = f.stride; a.pad = f.pad;
    a.K = f.kh * f.kw * f.Cin;
    a.per_sample = f.per_sample_w ? 1 : 0;
    a.npix = a.per_sample ? f.Ho * f.Wo : f.B * f.Ho * f.Wo;
    // 64 co rows when that wastes fewer rows of the last tile (Cout 64, 192, ...); widest column tile K fills
    a.bm = (cdiv(f.Cout, 64) * 64 < cdiv(f.Cout, 128) * 128) ? 64 : 128;
    a.bn = a.K > 64 ? 128 : (a.K > 32 || a.bm == 64 ? 64 : 32);
    a.tiles_co = cdiv(f.Cout, a.bm);
    a.tiles_k = cdiv(a.K, a.bn);
    const long tiles = (long)a.tiles_co * a.tiles_k * (a.per_sample ? f.B : 1);
    // enough splits to fill the chip a few times over, but at least 8 K-tiles of work per split
    static const long target = getenv("SOMI_WGRAD_WGS") ? atol(getenv("SOMI_WGRAD_WGS")) : 1024;
    long want = target / tiles;                                          // whole rounds of the 512 workgroup slots, never a bit more
    const long max_splits = (a.npix + WG_PIX * 8 - 1) / (WG_PIX * 8);
    if (want > max_splits) want = max_splits;
    if (want < 1) want = 1;
    a.pix_per_split = (int)(((a.npix + want - 1) / want + WG_PIX - 1) / WG_PIX * WG_PIX);
    a.splits = cdiv(a.npix, a.pix_per_split);
    return 0;
}

}  // namespace somi

using namespace somi;

extern "C" size_t somi_conv2d_wgrad_workspace_bytes(const somi_conv_desc *fwd) {
    WgradArgs a{};
    if (!fwd || plan(*fwd, a)) return 0;
    // also with a single split: accumulating on top of an existing gradient goes through the workspace + reduce pass
    return (size_t)(a.splits < 1 ? 1 : a.splits) * (a.per_sample ? a.B : 1) * a.Cout * a.K * 4 + 256;
}

extern "C" int somi_conv2d_wgrad_nhwc_f32(const somi_conv_desc *fwd, const float *x, int x_cs, int x_coff, const float *dy, int dy_cs,
                                          int dy_coff, float *dw, const float *accumulate, void *workspace, size_t workspace_bytes,
                                          somi_stream_t stream) {
    SOMI_REQUIRE(fwd && x && dy && dw && workspace, SOMI_EINVAL, "conv wgrad: null argument");
    WgradArgs a{};
    int rc = plan(*fwd, a);
    if (rc) return rc;
    SOMI_REQUIRE(x_cs % 4 == 0 && x_coff % 4 == 0 && dy_cs % 4 == 0 && dy_coff % 4 == 0 && aligned16(x) && aligned16(dy) && aligned16(dw),
                 SOMI_EINVAL, "conv wgrad: strides / offsets must be multiples of 4 and bases 16 B aligned");
    SOMI_REQUIRE(x_coff + a.Cin <= x_cs && dy_coff + a.Cout <= dy_cs, SOMI_EINVAL, "conv wgrad: channel slice out of range");
    SOMI_REQUIRE(workspace_bytes >= somi_conv2d_wgrad_workspace_bytes(fwd), SOMI_EWORKSPACE, "conv wgrad: workspace too small");
    const size_t xb = (size_t)a.B * a.H * a.W * x_cs * 4, yb = (size_t)a.B * a.Ho * a.Wo * dy_cs * 4;
    if (xb > W_MAX_BUF || yb > W_MAX_BUF) {
        // 32-bit buffer descriptors: reduce the batch in slices that fit, each accumulating on top of the previous ones
        SOMI_REQUIRE(!a.per_sample, SOMI_ENOTIMPL, "conv wgrad: per-sample gradients of tensors beyond the 4 GiB descriptor range");
        const size_t per_img = (xb > yb ? xb : yb) / a.B;
        SOMI_REQUIRE(per_img <= W_MAX_BUF, SOMI_ENOTIMPL, "conv wgrad: one image exceeds the 4 GiB descriptor range");
        const int bsub = (int)(W_MAX_BUF / per_img);
        for (int b0 = 0; b0 < a.B; b0 += bsub) {
            somi_conv_desc sub = *fwd;
            sub.B = a.B - b0 < bsub ? a.B - b0 : bsub;
            rc = somi_conv2d_wgrad_nhwc_f32(&sub, x + (size_t)b0 * a.H * a.W * x_cs, x_cs, x_coff, dy + (size_t)b0 * a.Ho * a.Wo * dy_cs, dy_cs,
                                            dy_coff, dw, b0 == 0 ? accumulate : dw, workspace, workspace_bytes, stream);
            if (rc) return rc;
        }
        return 0;
    }
    a.x = x; a.dy = dy; a.x_cs = x_cs; a.x_coff = x_coff; a.dy_cs = dy_cs; a.dy_coff = dy_coff;
    a.x_bytes = (unsigned)xb; a.dy_bytes = (unsigned)yb;
    const int sets = a.per_sample ? a.B : 1;
    const bool direct = a.splits == 1 && !accumulate;
    a.out = direct ? dw : static_cast<float *>(workspace);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid(a.tiles_co * a.tiles_k * a.splits, 1, sets);
    static const int eight = getenv("SOMI_WGRAD_8WAVE") ? atoi(getenv("SOMI_WGRAD_8WAVE")) : 2;
    a.ns = (fwd->prec == 1 || fwd->prec == 2) && !a.per_sample && a.bn == 128 ? fwd->prec : 0;
    if (a.ns == 1 && a.bm == 128) hipLaunchKernelGGL((conv_wgrad_f32_kernel<128, 128, 8, 1>), grid, dim3(512), 0, s, a);
    else if (a.ns == 2 && a.bm == 128) hipLaunchKernelGGL((conv_wgrad_f32_kernel<128, 128, 8, 2>), grid, dim3(512), 0, s, a);
    else if (a.ns == 1) hipLaunchKernelGGL((conv_wgrad_f32_kernel<64, 128, 8, 1>), grid, dim3(512), 0, s, a);
    else if (a.ns == 2) hipLaunchKernelGGL((conv_wgrad_f32_kernel<64, 128, 8, 2>), grid, dim3(512), 0, s, a);
    else if (a.bm == 128 && a.bn == 128 && eight) hipLaunchKernelGGL((conv_wgrad_f32_kernel<128, 128, 8>), grid, dim3(512), 0, s, a);
    else if (a.bm == 128 && a.bn == 128) hipLaunchKernelGGL((conv_wgrad_f32_kernel<128, 128>), grid, dim3(256), 0, s, a);
    else if (a.bm == 128 && a.bn == 64) hipLaunchKernelGGL((conv_wgrad_f32_kernel<128, 64>), grid, dim3(256), 0, s, a);
    else if (a.bm == 128) hipLaunchKernelGGL((conv_wgrad_f32_kernel<128, 32>), grid, dim3(256), 0, s, a);
    else if (a.bn == 128 && eight > 1) hipLaunchKernelGGL((conv_wgrad_f32_kernel<64, 128, 8>), grid, dim3(512), 0, s, a);
    else if (a.bn == 128) hipLaunchKernelGGL((conv_wgrad_f32_kernel<64, 128>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((conv_wgrad_f32_kernel<64, 64>), grid, dim3(256), 0, s, a);
    if (!direct) {
        const long n = (long)sets * a.Cout * a.K;
        long g = (n / 4 + 15) / 16;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(g > 4096 ? 4096 : (g < 1 ? 1 : g))), dim3(256), 0, s,
                           static_cast<const float *>(workspace), a.splits, n, accumulate, dw);
    }
    return launch_status("somi_conv2d_wgrad_nhwc_f32");
}
