// NHWC implicit-GEMM convolution on the fp32 matrix cores of gfx950, with a fused epilogue.
//
// GEMM view:  M = B*Ho*Wo output pixels, N = Cout, K = kh*kw*Cin with k = (r*kw+q)*Cin + c.
//   act[m][k] = x[b, ho*s-p+r*d, wo*s-p+q*d, c]  (0 outside the image) - gathered on the fly, never materialised
//   wgt[n][k] = w[n][k]                                                  - weights pre-packed K-contiguous
// Arithmetic: v_mfma_f32_32x32x2_f32 - exact fp32 products, fp32 accumulate (bit-for-bit an fmaf chain), so the
// result matches the reference CPU path to fp32 rounding (MI355X_MICROARCH.md, FP32-input MFMA: 157 TFLOP/s peak).
//
// Workgroup = 256 threads = 4 waves (one per SIMD), block tile BM x BN, K-tile 32, two workgroups per CU.
//  * operand fetch: `buffer_load_dwordx4` through wave-uniform descriptors; padding taps, the M / N / K tails are mapped
//    to an out-of-range offset so the hardware range check returns zeros - no branches, ~40 VALU per K-tile instead
//    of ~200 with pointer arithmetic (PMC: 4.1 VALU per MFMA before, which phase-locked the two waves of a SIMD into
//    issuing addresses together while the matrix pipe idled);
//  * LDS image [row][32] floats, the eight 16 B chunks of a row XOR-swizzled by (row >> 1) & 7 (conflict-free ds_read_b128 for the
//    32x32 fragment pattern without padding), DOUBLE buffered, one barrier per K-tile.  The image is lane-linear per wave instruction
//    (8 rows x 128 B), so the plain FAST path fills it by LDS-DMA (`buffer_load_dwordx4 ... lds`: no VGPR staging, no ds_write pass -
//    measured round 3: the register-staged write pass cost 5 % of the kernel) with the swizzle on the SOURCE address; the modulated and
//    generic paths stage through registers into the same image;
//  * each lane reads 4 consecutive k of one row per ds_read_b128 and feeds them to 4 consecutive MFMA k-steps (the k
//    order inside a K-tile is permuted identically for both operands: lane half h of k-step t takes k = 8j+4h+t);
//  * the weights are the MFMA A operand and the activations the B operand, so the accumulator holds D[n][m] with
//    m on the lane and 4 consecutive n per register quad: the epilogue (bias, activation, affine, residual) works on
//    float4 and leaves as 16 B stores.
//
//  * tile scheduling: one workgroup per tile when the tile count fills the 512 workgroup slots (256 CUs x 2) in nearly whole
//    rounds; otherwise STREAM-K: 512 persistent workgroups each take an equal contiguous run of (tile, K-tile) units, so
//    the last round is never a mostly idle one (1600 tiles = 3.125 rounds cost 4 before).  A tile whose K range is cut
//    between workgroups leaves raw partial accumulators in a workspace; a second small kernel adds them in ascending
//    workgroup order (deterministic) and runs the epilogue.
//
// Reference being replaced: F.conv2d + BatchNorm2d(eval) + SiLU in Conv.forward (models/common.py:64-70,
// folded as utils/torch_utils.py:202-222), ODConv2d_3rd's grouped per-sample conv (models/common.py:4602-4605).
#include <stdlib.h>
#include "common.h"

namespace somi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BK = 32;                    // K-tile (floats)
constexpr int LDS_LD = BK + 4;            // sizes the LDS array (the epilogue stages D tiles with this row stride)
constexpr int OP_LD = BK;                 // operand image row stride in floats (128 B, chunks XOR-swizzled)
constexpr unsigned OOB = 0xFFFFFFE0u;     // byte offset beyond every descriptor: the load returns 0
constexpr unsigned OOB_BASE = 0xF0000000u;  // + any K offset (< 2^27) still beyond every descriptor
constexpr unsigned MAX_BUF_BYTES = 0xE0000000u;

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

__device__ __forceinline__ float fast_silu(float v) { return v * __frcp_rn(1.0f + __expf(-v)); }

struct ConvArgs {
    somi_conv_desc d;
    int M;        // rows per weight set: B*Ho*Wo, or Ho*Wo when per_sample_w (grid.z = B)
    int K;        // kh*kw*Cin
    int tiles_m, tiles_n;
    unsigned x_bytes, w_bytes, chan_bytes, pix_bytes;
    int dgrad;    // 1: data-gradient geometry (rows = forward-input pixels, source = dy, taps walk backwards, stride parity)
    int cls;      // dgrad on the FAST path: grid.y = stride^2 parity classes, each walks only the taps that reach it
    int sk;       // stream-K: gridDim.x persistent workgroups share tiles_m*tiles_n*nkt units (FAST, no strided classes)
    int sk_whole; // hybrid: every workgroup first takes sk_whole WHOLE tiles (tile j * gridDim.x + wg), only the tiles behind them are streamed
    int sk_rem_g; // ... over this many workgroups (<= gridDim.x: pieces shorter than a few K-tiles are not worth a partial store)
    int sk_grid_main;   // gridDim.x of the main kernel (the fix-up kernel needs it to find the first streamed tile)
    float *ws;    // stream-K partial accumulators: [workgroup][2][BM*BN]
    int ns;       // 0: exact fp32; 1: bf16 operands; 2: bf16x3 split (plain FAST launches of the 8-wave tiles only)
};

constexpr int SK_GRID = 512;              // 256 CUs x 2 resident workgroups

// rows of the GEMM -> pixel rows of y (identity except for the strided dgrad classes)
struct RowMap {
    int Mrows, HoWo, Wc, cstep, h0, w0, bz;
    bool strided;
};

// LDS floats of a tile variant: the two operand buffers, or the epilogue's D staging if that is larger.  The staging is split into
// 32-column slabs when the whole D tile would not fit the (padded) operand area.
template <int BM, int BN, int WAVES_M, int WAVES_N>
struct TileLds {
    static constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, W = WAVES_M * WAVES_N;
    static constexpr bool SPLIT = W * WM * (WN + 4) > 2 * (BM + BN) * LDS_LD;
    static constexpr int SW = SPLIT ? 32 : WN;
    static constexpr int STAGE = W * WM * (SW + 4), OPS = 2 * (BM + BN) * OP_LD;
    static constexpr int FLOATS = STAGE > OPS ? STAGE : OPS;
};

// Epilogue.  C/D map of the 32x32 MFMA: col = lane&31 (-> m), row = (e&3) + 8*(e>>2) + 4*(lane>>5) (-> n).
// Each wave parks its D tile in its own LDS region as [m][n] (the operand buffers are free after the last barrier),
// then sweeps it with a compact loop: 16 lanes cover 64 consecutive n of one row -> bias / activation / affine /
// residual on float4 and full-line 16 B stores.
template <int BM, int BN, int WAVES_M, int WAVES_N>
__device__ __forceinline__ void conv_epilogue(f32x16 (&acc)[BN / WAVES_N / 32][BM / WAVES_M / 32], float *lds, const ConvArgs &a,
                                              const RowMap &rm, int m0, int n0) {
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, TM = WM / 32;
    // the whole D tile staged at once when it fits the operand buffers, else one 32-column slab of every wave tile per pass
    using TL = TileLds<BM, BN, WAVES_M, WAVES_N>;
    constexpr int SW = TL::SW, NPASS = WN / SW, SLD = SW + 4;
    static_assert(WAVES_M * WAVES_N * WM * SLD <= TL::FLOATS, "epilogue staging does not fit the LDS array");
    const somi_conv_desc &d = a.d;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    float *stage = lds + wave * (WM * SLD);
    const int h4 = (lane >> 5) * 4;
    const float *bias = d.bias ? d.bias + (size_t)rm.bz * d.Cout : nullptr;
    const size_t row_base = (size_t)rm.bz * a.M;
    constexpr int NQ = SW / 4;                                    // float4 per staged row
    const int mw = m0 + wm * WM;
    const bool stats = d.stat_sum != nullptr;
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        if (pass) __syncthreads();                                // the previous slab has been swept
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int js = 0; js < SW / 32; ++js)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int jn = pass * (SW / 32) + js;
                    const f32x4 v = {acc[jn][i][4 * g], acc[jn][i][4 * g + 1], acc[jn][i][4 * g + 2], acc[jn][i][4 * g + 3]};
                    *reinterpret_cast<f32x4 *>(&stage[(i * 32 + (lane & 31)) * SLD + js * 32 + 8 * g + h4]) = v;
                }
        __syncthreads();
        const int nw = n0 + wn * WN + pass * SW;
        // optional BatchNorm statistics: a lane always sweeps the same column quad (64 % NQ == 0), so it keeps that quad's sums
        f32x4 st1 = {0.f, 0.f, 0.f, 0.f}, st2 = st1, piv = st1;
        if (stats && d.stat_pivot && nw + (lane % NQ) * 4 < d.Cout) piv = *reinterpret_cast<const f32x4 *>(d.stat_pivot + nw + (lane % NQ) * 4);
        for (int idx = lane; idx < WM * NQ; idx += 64) {
            const int ml = idx / NQ, n = nw + (idx % NQ) * 4, m = mw + ml;
            if (m >= rm.Mrows || n >= d.Cout) continue;
            f32x4 v = *reinterpret_cast<const f32x4 *>(&stage[ml * SLD + (idx % NQ) * 4]);
            size_t row = row_base + m;
            if (rm.strided) {
                const int rem = m % rm.HoWo;
                row = ((size_t)(m / rm.HoWo + rm.bz) * d.Ho + rm.h0 + (rem / rm.Wc) * rm.cstep) * d.Wo + rm.w0 + (rem % rm.Wc) * rm.cstep;
            }
            float *yrow = d.y + row * d.y_cs + d.y_coff;
            const float *rrow = d.residual ? d.residual + row * d.res_cs + d.res_coff : nullptr;
            const float *rrow2 = d.residual2 ? d.residual2 + row * d.res2_cs + d.res2_coff : nullptr;
            if (n + 3 < d.Cout) {
                if (bias) v += *reinterpret_cast<const f32x4 *>(bias + n);
                if (d.act == SOMI_ACT_SILU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fast_silu(v[e]);
                } else if (d.act != SOMI_ACT_NONE) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = apply_act_rt(v[e], d.act);
                }
                if (d.post_scale)
                    v = v * *reinterpret_cast<const f32x4 *>(d.post_scale + n) + *reinterpret_cast<const f32x4 *>(d.post_shift + n);
                if (rrow) v += *reinterpret_cast<const f32x4 *>(rrow + n);
                if (rrow2) v += *reinterpret_cast<const f32x4 *>(rrow2 + n);
                *reinterpret_cast<f32x4 *>(yrow + n) = v;
                if (stats) {
                    const f32x4 t = v - piv;
                    st1 += t;
                    st2 += t * t;
                }
            } else {
                for (int e = 0; e < 4 && n + e < d.Cout; ++e) {          // ragged Cout tail
                    float t = v[e] + (bias ? bias[n + e] : 0.f);
                    t = apply_act_rt(t, d.act);
                    if (d.post_scale) t = t * d.post_scale[n + e] + d.post_shift[n + e];
                    if (rrow) t += rrow[n + e];
                    if (rrow2) t += rrow2[n + e];
                    yrow[n + e] = t;
                }
            }
        }
        if (stats) {                                              // fold the 64 / NQ lanes that share a column quad, then one row per (tile, wave row)
#pragma unroll
            for (int off = NQ; off < 64; off <<= 1)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    st1[e] += __shfl_xor(st1[e], off);
                    st2[e] += __shfl_xor(st2[e], off);
                }
            const int n = nw + lane * 4;
            if (lane < NQ && n < d.Cout) {
                const size_t prow = (size_t)(m0 / BM) * WAVES_M + wm;
                *reinterpret_cast<f32x4 *>(d.stat_sum + prow * d.Cout + n) = st1;
                *reinterpret_cast<f32x4 *>(d.stat_sumsq + prow * d.Cout + n) = st2;
            }
        }
    }
}

// unit range of stream-K workgroup g: [g*U/G, (g+1)*U/G)
__device__ __forceinline__ long sk_lo(int g, long U, int G) { return (long)g * U / G; }

__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
// FAST: Cin % 32 == 0 (every K-tile lies inside one filter tap), kh*kw <= 32: the tap walk is wave-uniform (SALU), each
// row's padding test is one bit of a mask built once, and a fetch costs 4 VALU per 16 B.
// NS (opt-in reduced precision, train.py:263 `amp.autocast`): 0 = exact fp32 products (v_mfma_f32_32x32x2_f32); 1 = operands rounded to
// bf16, fp32 accumulate (v_mfma_f32_32x32x16_bf16: autocast's arithmetic); 2 = "bf16x3": every operand split x = hi + lo into two
// bf16 values and hi*hi + hi*lo + lo*hi accumulated in fp32 - 16 mantissa bits per operand, ~1e-5 relative error, 3 MFMAs at 16x the
// fp32 rate.  The operands stay fp32 in HBM and are converted while they are staged into LDS ([row][32 bf16 hi | 32 bf16 lo] = the
// fp32 image's 128-byte rows), so nothing outside this kernel changes.
// Waves per SIMD the register budget is set for: 4 (two 8-wave workgroups per CU), 2 for the big experiment tile, and 6 for the 8-wave
// 128 x 64 tile's plain fp32 form - its 48 KB of LDS let THREE workgroups share a CU, which the 64-channel layers (K = 576: 18 K-tiles
// between a prologue and an epilogue) use to hide those phases behind two neighbours instead of one.
template <int BM, int BN, int WAVES, bool MODULATE, bool FAST, int NS>
constexpr int conv_waves_per_simd() {
    return BM + BN > 320 ? 2 : (WAVES == 8 && BM + BN <= 192 && !MODULATE && FAST && NS == 0 ? 6 : WAVES / 2);
}
template <int BM, int BN, int WAVES_M, int WAVES_N, bool MODULATE, bool FAST, int NS = 0>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64, (conv_waves_per_simd<BM, BN, WAVES_M * WAVES_N, MODULATE, FAST, NS>())) void conv_igemm_f32_kernel(const ConvArgs a) {
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;   // wave tile
    constexpr int TM = WM / 32, TN = WN / 32;             // 32x32 MFMA tiles per wave
    constexpr int NT = WAVES_M * WAVES_N * 64, RPP = NT / 8;   // threads; rows fetched per pass (8 threads x 16 B cover a row's K-tile)
    constexpr int A_ROWS = BM / RPP, B_ROWS = BN / RPP;   // 16 B loads per thread per K-tile (activations / weights)
    constexpr int TILE = (BM + BN) * OP_LD;               // one operand buffer
    // operands go global -> LDS directly (the DMA's LDS base travels in M0[15:0]: both buffers must lie below 64 KB)
    constexpr bool DMA = FAST && !MODULATE && NS == 0 && 2 * TILE * sizeof(float) <= 65536;
    static_assert(NS == 0 || (FAST && !MODULATE), "the bf16 forms exist for the plain FAST path only");
    static_assert((WAVES_M * WAVES_N == 4 || WAVES_M * WAVES_N == 8) && TM >= 1 && TN >= 1 && A_ROWS >= 1 && B_ROWS >= 1, "bad tiling");

    __shared__ __attribute__((aligned(16))) float lds[TileLds<BM, BN, WAVES_M, WAVES_N>::FLOATS];

    const somi_conv_desc &d = a.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int bz = blockIdx.z;                                        // weight set / image (per_sample_w)

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)d.x, 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)d.w, 0, a.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void *)(MODULATE && d.a_chan_scale ? d.a_chan_scale : d.x), 0,
                                                                        MODULATE && d.a_chan_scale ? a.chan_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void *)(MODULATE && d.a_pix_scale ? d.a_pix_scale : d.x), 0,
                                                                        MODULATE && d.a_pix_scale ? a.pix_bytes : 0, 0x00020000);
    const bool has_chan = MODULATE && d.a_chan_scale, has_pix = MODULATE && d.a_pix_scale;

    // dgrad parity class (FAST path): rows are the pixels with (h+pad)%s == ph, (w+pad)%s == pw, i.e. h = h0 + s*hc, and
    // only the taps r = ph + s*i, q = pw + s*j reach them - a stride-2 3x3 layer does 9/4 taps per pixel instead of 9.
    const bool cls = FAST && a.cls;
    const int ph = cls ? (int)blockIdx.y / d.stride : 0, pw = cls ? (int)blockIdx.y % d.stride : 0;
    const int h0 = cls ? ((ph - d.pad) % d.stride + d.stride) % d.stride : 0;
    const int w0 = cls ? ((pw - d.pad) % d.stride + d.stride) % d.stride : 0;
    const int cstep = cls ? d.stride : 1;
    const int Hc = cls ? (d.Ho - h0 + cstep - 1) / cstep : d.Ho, Wc = cls ? (d.Wo - w0 + cstep - 1) / cstep : d.Wo;
    const int nr_c = cls ? (d.kh - ph + cstep - 1) / cstep : d.kh, nq_c = cls ? (d.kw - pw + cstep - 1) / cstep : d.kw;
    const int HoWo = Hc * Wc;
    const int Mrows = cls ? (d.per_sample_w ? HoWo : d.B * HoWo) : a.M;
    const int nkt = cls ? nr_c * nq_c * (d.Cin / BK) : (a.K + BK - 1) / BK;      // 0 for a class no tap reaches (k=1, s=2)
    const RowMap rmap = {Mrows, HoWo, Wc, cstep, h0, w0, bz, cls && cstep > 1};

    // ---- schedule: one tile (all of its K-tiles), or a stream-K run of (tile, K-tile) units
    const int ntile = a.tiles_m * a.tiles_n;
    const bool sk = FAST && a.sk;
    const int wg = sk ? xcd_remap(blockIdx.x, gridDim.x) : 0;
    // hybrid stream-K (round 4, an experiment switch: sk_whole = 0 by default): whole rounds of tiles run like the plain schedule - tile
    // j * G + wg, no cut, no partial store - and only the tiles behind them (fewer than G) are streamed, over sk_rem_g workgroups.  The partials
    // and the fix-up shrink from one seam per workgroup (64 MiB written + read, 19.4 us x 193 launches per step) to the seams of the last partial
    // round - and the step does not get faster (launch<>).
    const int sk_whole = sk ? a.sk_whole : 0, tile_base = sk ? sk_whole * (int)gridDim.x : 0, Gr = sk ? a.sk_rem_g : 1;
    const long U = (long)(ntile - tile_base) * nkt;                  // streamed units
    long u = sk ? sk_lo(min(wg, Gr), U, Gr) : (long)xcd_remap(blockIdx.x, ntile) * (nkt > 0 ? nkt : 1);
    const long u_lo = u, u_hi = sk ? sk_lo(min(wg + 1, Gr), U, Gr) : u + (nkt > 0 ? nkt : 1);
    int whole_left = sk_whole;

    // A thread owns LDS chunk slot (tid & 7) of rows row0 + RPP*i; the slot holds k-chunk slot ^ ((row >> 1) & 7), and (row >> 1) & 7 is
    // the same for all of a thread's rows (RPP, BM are multiples of 16), so the swizzle is one XOR on the thread's fetch column.
    // (the bf16 forms fetch their natural column and swizzle the LDS address of the converted values instead: 8-byte pieces)
    const int kc = NS ? (tid & 7) * 4 : ((tid & 7) ^ ((tid >> 4) & 7)) * 4, row0 = tid >> 3;   // per-thread fetch column (floats) / first row
    const int lds_col = (tid & 7) * 4;                                // ... and where it lands in the row's image
    // bf16 image: 16-byte chunk c of a row (8 bf16: hi k 8c..8c+7 for c < 4, lo for c >= 4) sits at slot c ^ f(row),
    // f(row) = ((row >> 1) & 7) ^ ((row & 1) << 2): conflict-free ds_read_b128 fragments and ds_write_b64 pieces; lo = hi ^ 64 bytes
    const int f_st = ((row0 >> 1) & 7) ^ ((row0 & 1) << 2);           // same for all of a thread's rows (RPP, BM multiples of 16)
    const int st_off = ((((tid & 7) >> 1) ^ f_st) << 2) + ((tid & 1) << 1);   // float offset of this thread's 4 hi values in its row
    const int f_rd = ((lane >> 1) & 7) ^ ((lane & 1) << 2);
    int fo16[2];                                                      // float offset of hi chunk 2s + (lane >> 5) in the lane's row
#pragma unroll
    for (int st = 0; st < 2; ++st) fo16[st] = ((2 * st + (lane >> 5)) ^ f_rd) << 2;
    const int aw_off = (wm * WM + (lane & 31)) * OP_LD;               // activation rows of this wave (fragment row = lane & 31)
    const int bw_off = (BM + wn * WN + (lane & 31)) * OP_LD;          // weight rows of this wave
    int fo[4];                                                        // float offset of k-chunk 2j + (lane >> 5) inside the lane's row
#pragma unroll
    for (int j = 0; j < 4; ++j) fo[j] = ((2 * j + (lane >> 5)) ^ ((lane >> 1) & 7)) << 2;

    while (whole_left > 0 || u < u_hi) {
        int tile, kt0, kt1;
        const bool whole = whole_left > 0;
        if (whole) {                                                  // a whole tile of this workgroup's rounds
            tile = (sk_whole - whole_left) * (int)gridDim.x + wg;
            kt0 = 0;
            kt1 = nkt;
            --whole_left;
        } else {
            tile = tile_base + (int)(u / (nkt > 0 ? nkt : 1));
            kt0 = nkt > 0 ? (int)(u % nkt) : 0;
            kt1 = sk ? (int)min((long)nkt, kt0 + (u_hi - u)) : nkt;
            u += sk ? kt1 - kt0 : (nkt > 0 ? nkt : 1);
        }
        const int tile_m = tile / a.tiles_n, tile_n = tile % a.tiles_n;   // n fastest: neighbours share the activation rows
        const int m0 = tile_m * BM, n0 = tile_n * BN;
        if (m0 >= Mrows) return;                                          // smaller class than the grid was sized for (never stream-K)

        // ---- per-thread fetch geometry: 16 B column kc of rows row0 + 32*i
        int a_hi0[A_ROWS], a_wi0[A_ROWS], a_off[A_ROWS], a_pixi[A_ROWS], a_chn[A_ROWS], a_par[A_ROWS];
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) a_par[i] = 0;
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) {
            const int m = m0 + row0 + RPP * i;
            if (m < Mrows) {
                const int b = m / HoWo + bz, rem = m % HoWo;
                const int ho = h0 + (rem / Wc) * cstep, wo = w0 + (rem % Wc) * cstep;
                if (!a.dgrad) {
                    a_hi0[i] = ho * d.stride - d.pad;
                    a_wi0[i] = wo * d.stride - d.pad;
                } else {                                                      // source row of tap r: (ho+pad)/s - r/s, valid iff r%s == (ho+pad)%s
                    a_hi0[i] = (ho + d.pad) / d.stride;
                    a_wi0[i] = (wo + d.pad) / d.stride;
                    a_par[i] = ((ho + d.pad) % d.stride) | (((wo + d.pad) % d.stride) << 8);
                }
                a_pixi[i] = (b * d.H + a_hi0[i]) * d.W + a_wi0[i];           // pixel index of tap (0,0), may be "before" the image
                a_off[i] = a_pixi[i] * d.x_cs + d.x_coff;
                a_chn[i] = b * d.Cin;
            } else {
                a_hi0[i] = -(1 << 28);                                        // the bounds test fails for every tap -> zeros
                a_wi0[i] = 0; a_pixi[i] = 0; a_off[i] = 0; a_chn[i] = 0;
            }
        }
        unsigned b_off[B_ROWS];
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i) {
            const int n = n0 + row0 + RPP * i;
            b_off[i] = n < d.Cout ? (unsigned)(((size_t)bz * d.Cout + n) * a.K + kc) * 4u : (FAST ? OOB_BASE : OOB);
        }
        // FAST path state: per-row tap-validity masks and byte offsets with the thread's column folded in
        unsigned a_mask[A_ROWS], a_offb[A_ROWS], a_pixb[A_ROWS], a_chnb[A_ROWS];
        if constexpr (FAST) {
            const int ntap = nr_c * nq_c;
#pragma unroll
            for (int i = 0; i < A_ROWS; ++i) {
                unsigned mk = 0;
                for (int t = 0; t < ntap; ++t) {
                    const int tr = t / nq_c, tq = t % nq_c;                  // class-local tap index (dgrad: r = ph + s*tr)
                    const int hh = a_hi0[i] + (cls ? -tr : tr * d.dil);
                    const int ww = a_wi0[i] + (cls ? -tq : tq * d.dil);
                    mk |= ((unsigned)hh < (unsigned)d.H && (unsigned)ww < (unsigned)d.W) ? (1u << t) : 0u;
                }
                a_mask[i] = mk;
                a_offb[i] = (unsigned)(a_off[i] + kc) * 4u;
                a_pixb[i] = (unsigned)a_pixi[i] * 4u;
                a_chnb[i] = (unsigned)(a_chn[i] + kc) * 4u;
            }
        }
        // wave-uniform K walk (FAST), positioned on K-tile kt0.  Order: channel chunk OUTER, taps INNER - K-tile kt covers channels
        // [32 (kt / ntap), +32) of tap kt % ntap.  (Tap-major order kept 3 image rows x Cin channels live per workgroup: 64 workgroups
        // per XCD x 245 KB at 160x160x128 = 15 MB against a 4 MB L2, so every input row was fetched once per kernel row - PMC: 1079 MB
        // for a 419 MB input.  With the taps inner the live set is 3 rows x 32 channels = 61 KB per workgroup.)
        const int ntap_u = FAST ? nr_c * nq_c : 1;
        int tp_u = FAST ? kt0 % ntap_u : 0, c0_u = FAST ? (kt0 / ntap_u) * BK : 0;
        int r_u = tp_u / nq_c, q_u = tp_u % nq_c;

        // k -> (tap r,q ; channel c) for this thread's column, advanced incrementally per K-tile (generic path: kt0 == 0)
        int k = kc, c = kc % d.Cin, tap = kc / d.Cin;
        int r = tap / d.kw, q = tap % d.kw;

        f32x4 ra[A_ROWS], rb[B_ROWS];
        auto fetch_tile = [&](int kt_next) {
            if constexpr (FAST) {
                const int dpix = cls ? -(r_u * d.W + q_u) : (r_u * d.W + q_u) * d.dil;
                const unsigned sd = (unsigned)(dpix * d.x_cs + c0_u) * 4u, bit = 1u << tp_u;
#pragma unroll
                for (int i = 0; i < A_ROWS; ++i) {
                    const bool ok = (a_mask[i] & bit) != 0;
                    f32x4 v = buf_load4(rx, ok ? a_offb[i] + sd : OOB);
                    if constexpr (MODULATE) {
                        if (has_chan) v *= buf_load4(rc, ok ? a_chnb[i] + (unsigned)c0_u * 4u : OOB);
                        if (has_pix) v *= buf_load1(rp, ok ? a_pixb[i] + (unsigned)dpix * 4u : OOB);
                    }
                    ra[i] = v;
                }
                const unsigned ko = cls ? (unsigned)(((ph + cstep * r_u) * d.kw + pw + cstep * q_u) * d.Cin + c0_u) * 4u
                                        : (unsigned)(tp_u * d.Cin + c0_u) * 4u;
#pragma unroll
                for (int i = 0; i < B_ROWS; ++i) rb[i] = buf_load4(rw, b_off[i] + ko);
                return;
            }
            const bool kin = k < a.K;
            const int dh = a.dgrad ? -(r / d.stride) : r * d.dil, dw = a.dgrad ? -(q / d.stride) : q * d.dil;
            const int dpix = dh * d.W + dw;
            const int delta = dpix * d.x_cs + c;
            const int rpar = a.dgrad ? ((r % d.stride) | ((q % d.stride) << 8)) : 0;
#pragma unroll
            for (int i = 0; i < A_ROWS; ++i) {
                const bool ok = kin && a_par[i] == rpar && (unsigned)(a_hi0[i] + dh) < (unsigned)d.H &&
                                (unsigned)(a_wi0[i] + dw) < (unsigned)d.W;
                f32x4 v = buf_load4(rx, ok ? (unsigned)(a_off[i] + delta) * 4u : OOB);
                if constexpr (MODULATE) {
                    if (has_chan) v *= buf_load4(rc, ok ? (unsigned)(a_chn[i] + c) * 4u : OOB);
                    if (has_pix) v *= buf_load1(rp, ok ? (unsigned)(a_pixi[i] + dpix) * 4u : OOB);
                }
                ra[i] = v;
            }
#pragma unroll
            for (int i = 0; i < B_ROWS; ++i) rb[i] = buf_load4(rw, (kin && b_off[i] != OOB) ? b_off[i] : OOB);
        };
        auto advance_k = [&]() {
            if constexpr (FAST) {
                ++tp_u;
                if (++q_u == nq_c) { q_u = 0; ++r_u; }
                if (tp_u == ntap_u) {
                    tp_u = 0; r_u = 0; q_u = 0;
                    c0_u += BK;
                }
                return;
            }
            k += BK;
            c += BK;
            while (c >= d.Cin) {
                c -= d.Cin;
                if (++q == d.kw) { q = 0; ++r; }
            }
#pragma unroll
            for (int i = 0; i < B_ROWS; ++i) b_off[i] += (b_off[i] != OOB) ? BK * 4u : 0u;
        };
        auto store_split = [&](float *rowp, const f32x4 &v) {           // 4 floats -> 4 bf16 hi (8 B) [+ 4 bf16 lo in the chunk slot ^ 4]
            const __bf16 h0 = (__bf16)v[0], h1 = (__bf16)v[1], h2 = (__bf16)v[2], h3 = (__bf16)v[3];
            typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
            *reinterpret_cast<bf16x4 *>(rowp + st_off) = bf16x4{h0, h1, h2, h3};
            if constexpr (NS == 2) {
                const bf16x4 lo = {(__bf16)(v[0] - (float)h0), (__bf16)(v[1] - (float)h1), (__bf16)(v[2] - (float)h2), (__bf16)(v[3] - (float)h3)};
                *reinterpret_cast<bf16x4 *>(rowp + (st_off ^ 16)) = lo;
            }
        };
        auto store_tile = [&](float *buf) {
            if constexpr (NS != 0) {
#pragma unroll
                for (int i = 0; i < A_ROWS; ++i) store_split(&buf[(row0 + RPP * i) * OP_LD], ra[i]);
#pragma unroll
                for (int i = 0; i < B_ROWS; ++i) store_split(&buf[(BM + row0 + RPP * i) * OP_LD], rb[i]);
                return;
            }
#pragma unroll
            for (int i = 0; i < A_ROWS; ++i)
                *reinterpret_cast<f32x4 *>(&buf[(row0 + RPP * i) * OP_LD + lds_col]) = ra[i];
#pragma unroll
            for (int i = 0; i < B_ROWS; ++i)
                *reinterpret_cast<f32x4 *>(&buf[(BM + row0 + RPP * i) * OP_LD + lds_col]) = rb[i];
        };
        // the same fetch as LDS-DMA: each wave instruction lands 8 rows x 128 B at a wave-uniform LDS base + 16 B x lane
        auto dma_tile = [&](float *buf) {
            if constexpr (DMA) {
                const int dpix = cls ? -(r_u * d.W + q_u) : (r_u * d.W + q_u) * d.dil;
                const unsigned sd = (unsigned)(dpix * d.x_cs + c0_u) * 4u, bit = 1u << tp_u;
                float *wrow = buf + wave * 8 * OP_LD;
#pragma unroll
                for (int i = 0; i < A_ROWS; ++i)
                    lds_dma16(rx, wrow + RPP * i * OP_LD, (a_mask[i] & bit) ? a_offb[i] + sd : OOB);
                const unsigned ko = cls ? (unsigned)(((ph + cstep * r_u) * d.kw + pw + cstep * q_u) * d.Cin + c0_u) * 4u
                                        : (unsigned)(tp_u * d.Cin + c0_u) * 4u;
#pragma unroll
                for (int i = 0; i < B_ROWS; ++i)
                    lds_dma16(rw, wrow + (BM + RPP * i) * OP_LD, b_off[i] + ko);
            }
        };

        f32x16 acc[TN][TM];
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[jn][i][e] = 0.f;

        // Operand fragments are double buffered in registers: group j+1's three ds_read_b128 are issued BEFORE group j's MFMAs, so a
        // wave never sits on an LDS round trip between groups (round 2's loop read each group's fragments right before using them: four
        // exposed LDS latencies per K-tile per wave, and the two waves a workgroup puts on a SIMD hit them together).
        auto load_frag = [&](const float *buf, int j, f32x4 (&fa)[TM], f32x4 (&fb)[TN]) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4 *>(buf + aw_off + i * 32 * OP_LD + fo[j]);
#pragma unroll
            for (int i = 0; i < TN; ++i) fb[i] = *reinterpret_cast<const f32x4 *>(buf + bw_off + i * 32 * OP_LD + fo[j]);
        };
        auto mma_steps = [&](const f32x4 (&fa)[TM], const f32x4 (&fb)[TN], int t_lo, int t_hi) {
#pragma unroll
            for (int t = t_lo; t < t_hi; ++t)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        acc[jn][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[jn][t], fa[i][t], acc[jn][i], 0, 0, 0);
        };

        if (kt1 > kt0) {
            if constexpr (DMA) {
                dma_tile(lds);
                lds_dma_wait();       // the DMA's LDS writes are tracked by the vector-memory counter
            } else {
                fetch_tile(kt0);
                store_tile(lds);
            }
        }
        __syncthreads();
        if constexpr (NS != 0) {
            // bf16 forms: two 16-deep MFMA steps per K-tile; the next tile's global loads fly behind step 0, its conversion and LDS
            // stores follow step 1
            for (int kt = kt0; kt < kt1; ++kt) {
                const float *cur = lds + ((kt - kt0) & 1) * TILE;
                float *nxt = lds + ((kt - kt0 + 1) & 1) * TILE;
                const bool more = kt + 1 < kt1;
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    bf16x8 ah[TM], bh[TN], al[TM], bl[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const float *p = cur + aw_off + i * 32 * OP_LD;
                        ah[i] = *reinterpret_cast<const bf16x8 *>(p + fo16[st]);
                        if constexpr (NS == 2) al[i] = *reinterpret_cast<const bf16x8 *>(p + (fo16[st] ^ 16));
                    }
#pragma unroll
                    for (int i = 0; i < TN; ++i) {
                        const float *p = cur + bw_off + i * 32 * OP_LD;
                        bh[i] = *reinterpret_cast<const bf16x8 *>(p + fo16[st]);
                        if constexpr (NS == 2) bl[i] = *reinterpret_cast<const bf16x8 *>(p + (fo16[st] ^ 16));
                    }
                    if (st == 0 && more) {
                        advance_k();
                        fetch_tile(kt + 1);
                    }
#pragma unroll
                    for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                        for (int i = 0; i < TM; ++i) {
                            if constexpr (NS == 2) {                  // the small cross terms first
                                acc[jn][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[jn], ah[i], acc[jn][i], 0, 0, 0);
                                acc[jn][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[jn], al[i], acc[jn][i], 0, 0, 0);
                            }
                            acc[jn][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[jn], ah[i], acc[jn][i], 0, 0, 0);
                        }
                }
                if (more) store_tile(nxt);
                __syncthreads();
            }
        }
        f32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
        if (NS == 0 && kt1 > kt0) load_frag(lds, 0, fa0, fb0);
        for (int kt = kt0; NS == 0 && kt < kt1; ++kt) {
            const float *cur = lds + ((kt - kt0) & 1) * TILE;
            float *nxt = lds + ((kt - kt0 + 1) & 1) * TILE;
            const bool more = kt + 1 < kt1;
            load_frag(cur, 1, fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
            mma_steps(fa0, fb0, 0, 4);
            if (more) {
                advance_k();
                if constexpr (DMA) dma_tile(nxt);   // lands in the other buffer (last read before the previous barrier) behind the whole K-tile
                else fetch_tile(kt + 1);            // in flight behind the next two MFMA groups
            }
            __builtin_amdgcn_sched_barrier(0);
            load_frag(cur, 2, fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            mma_steps(fa1, fb1, 0, 4);
            __builtin_amdgcn_sched_barrier(0);
            load_frag(cur, 3, fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
            mma_steps(fa0, fb0, 0, 4);
            if constexpr (!DMA) {
                if (more) store_tile(nxt); // the other buffer: nobody reads it during this K-tile
            }
            __builtin_amdgcn_sched_barrier(0);
            mma_steps(fa1, fb1, 0, 1);
            if constexpr (DMA) lds_dma_wait();
            __syncthreads();           // every wave has read `cur` for the last time and its part of `nxt` has landed
            if (more) load_frag(nxt, 0, fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            mma_steps(fa1, fb1, 1, 4);
        }

        if (!sk || (kt0 == 0 && kt1 == nkt)) {
            conv_epilogue<BM, BN, WAVES_M, WAVES_N>(acc, lds, a, rmap, m0, n0);
        } else {
            // a cut tile: raw accumulators to this workgroup's slot (0: the run starts with this piece, 1: it ends with it),
            // one coalesced float per lane per register
            const bool first_piece = (long)(tile - tile_base) * nkt + kt0 == u_lo;
            float *slot = a.ws + ((size_t)wg * 2 + (first_piece ? 0 : 1)) * (BM * BN);
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) slot[((jn * TM + i) * 16 + e) * NT + tid] = acc[jn][i][e];
        }
        __syncthreads();                                              // the epilogue staging / LDS buffers are reused by the next piece
    }
}

// Stream-K fix-up: workgroup j looks at the boundary between the runs of workgroups j and j+1; if it cuts a tile and is the
// first cut inside that tile, it adds the tile's pieces in ascending workgroup order and runs the epilogue.
template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64, (BM + BN > 320 ? 2 : WAVES_M * WAVES_N / 2)) void conv_streamk_fixup_kernel(const ConvArgs a, int G) {
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, TM = WM / 32, TN = WN / 32, NT = WAVES_M * WAVES_N * 64;
    __shared__ __attribute__((aligned(16))) float lds[TileLds<BM, BN, WAVES_M, WAVES_N>::FLOATS];
    const int nkt = a.K / BK, ntile = a.tiles_m * a.tiles_n;
    const int tile_base = a.sk_whole * a.sk_grid_main;               // the streamed tiles sit behind the whole rounds; G = sk_rem_g
    const long U = (long)(ntile - tile_base) * nkt;
    const int g = blockIdx.x + 1;
    const long b = sk_lo(g, U, G);
    if (b % nkt == 0) return;                                         // the boundary coincides with a tile boundary
    const int tile = (int)(b / nkt);                                  // index among the streamed tiles
    const long t_lo = (long)tile * nkt, t_hi = t_lo + nkt;
    if (sk_lo(g - 1, U, G) > t_lo) return;                            // an earlier boundary inside this tile does the work
    const int tid = threadIdx.x;
    f32x16 acc[TN][TM];
#pragma unroll
    for (int jn = 0; jn < TN; ++jn)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[jn][i][e] = 0.f;
    // the pieces of this tile in ascending workgroup order; two are fetched at a time (the walk is latency bound: a tile of a small
    // layer is cut into a dozen pieces) and added in order
    auto slot_of = [&](int w) -> const float * {
        const long lo = sk_lo(w, U, G), hi = sk_lo(w + 1, U, G);
        if (w >= G || lo >= t_hi) return nullptr;                     // past the tile: stop
        if (hi <= t_lo || hi == lo) return a.ws;                      // contributes nothing (never dereferenced: see `live`)
        const long seg = lo > t_lo ? lo : t_lo;
        return a.ws + ((size_t)w * 2 + (seg == lo ? 0 : 1)) * (BM * BN);
    };
    auto live = [&](int w) {
        const long lo = sk_lo(w, U, G), hi = sk_lo(w + 1, U, G);
        return w < G && lo < t_hi && hi > t_lo && hi != lo;
    };
    for (int w = g - 1; w < G; w += 2) {
        const float *s0 = slot_of(w), *s1 = slot_of(w + 1);
        if (!s0) break;
        const bool l0 = live(w), l1 = s1 && live(w + 1);
        f32x16 t0[TN][TM], t1[TN][TM];
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int o = ((jn * TM + i) * 16 + e) * NT + tid;
                    t0[jn][i][e] = l0 ? s0[o] : 0.f;
                    t1[jn][i][e] = l1 ? s1[o] : 0.f;
                }
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    if (l0) acc[jn][i][e] += t0[jn][i][e];
                    if (l1) acc[jn][i][e] += t1[jn][i][e];
                }
        if (!s1) break;
    }
    const RowMap rmap = {a.M, a.d.Ho * a.d.Wo, a.d.Wo, 1, 0, 0, 0, false};
    const int gt = tile_base + tile;
    conv_epilogue<BM, BN, WAVES_M, WAVES_N>(acc, lds, a, rmap, (gt / a.tiles_n) * BM, (gt % a.tiles_n) * BN);
}

struct TilePlan {
    int variant;   // 0: 128x128, 1: 64x128, 2: 128x64, 3: 128x32, 4: 128x128 with 8 waves (2 per SIMD and workgroup), 5: 128x64 8 waves,
                   // 6: 256x128 (experiment), 7: 256x64 with 8 waves of 32x64 (64-channel layers on large maps)
    bool sk;       // stream-K schedule
};
static const int kTileBM[8] = {128, 64, 128, 128, 128, 128, 256, 256}, kTileBN[8] = {128, 128, 64, 32, 128, 64, 128, 64};
// resident workgroups chip-wide: two per CU; one for the big tile; three for the plain fp32 form of the 8-wave 128 x 64 tile
static inline bool three_per_cu(int variant, const somi_conv_desc &d) {
    static const bool on = [] { const char *e = getenv("SOMI_CONV_3WG"); return !(e && e[0] == '0'); }();
    return on && variant == 5 && !d.a_chan_scale && !d.a_pix_scale && d.prec != 1 && d.prec != 2;
}
static bool fast_path(const somi_conv_desc &d);
static inline int sk_slots(int variant, const somi_conv_desc &d) {
    return variant == 6 ? SK_GRID / 2 : (three_per_cu(variant, d) && fast_path(d) ? SK_GRID * 3 / 2 : SK_GRID);
}

static bool fast_path(const somi_conv_desc &d) {
    return d.Cin % BK == 0 && d.kh * d.kw <= 32 && (size_t)d.kh * d.kw * d.Cin * 4 < (1u << 27);
}

// Tile variant and schedule.  Widest N tile that Cout fills reasonably.  With a workspace the 128-row tile is kept for small
// problems too and the K-tiles are streamed over 512 workgroups whenever whole rounds of tiles would leave >10% of the
// slots idle; without one, small-M problems take the 64-row tile to fill the chip.
static TilePlan plan_tiles(const somi_conv_desc &d, int M, int dgrad) {
    const bool sk_ok = d.workspace && fast_path(d) && !d.per_sample_w && !(dgrad && d.stride > 1);
    TilePlan p{0, false};
    const long blocks128 = (long)cdiv(M, 128) * cdiv(d.Cout, 128) * (d.per_sample_w ? d.B : 1);
    if (d.Cout > 64) {
        // a 64-wide N tile when it wastes much less of the last tile (e.g. Cout 192: 3 x 64 instead of 2 x 128)
        if (cdiv(d.Cout, 64) * 64 * 5 <= cdiv(d.Cout, 128) * 128 * 4 && M >= 128 * 256) p.variant = 2;
        else p.variant = (sk_ok || blocks128 >= 512 || M >= 128 * 256) ? 0 : 1;
    } else {
        p.variant = d.Cout > 32 ? 2 : 3;
    }
    static const int eight = getenv("SOMI_CONV_8WAVE") ? atoi(getenv("SOMI_CONV_8WAVE")) : 2;
    if (eight && p.variant == 0) p.variant = 4;
    if (eight > 1 && p.variant == 2) p.variant = 5;
    // 64 output channels on a large map: a 256 x 64 tile of 8 waves x (32 x 64) issues as many MFMAs per barrier as the 128 x 128 form
    // (the 128 x 64 tile half of them) at 1.25x its operand bytes per FLOP; 80 KB of LDS, two per CU.  Measured round 3: no gain - the
    // 64 -> 64 3x3 layers at 160x160 ran 98-100 TFLOP/s with it against 101-103 with the 128 x 64 tile, so MFMAs per barrier are not what
    // holds those layers at 100 (their K = 576 is 18 K-tiles: prologue + epilogue weigh twice what they do at K = 1152).  Off by default.
    static const int tall = getenv("SOMI_CONV_TALL64") ? atoi(getenv("SOMI_CONV_TALL64")) : 0;
    if (tall && p.variant == 5 && d.Cout > 32 && fast_path(d) && !d.a_chan_scale && !d.a_pix_scale && M >= 256 * 512) p.variant = 7;
    // 256 x 128 tile, 8 waves of 64 x 64 (a third fewer LDS operand bytes per MFMA than the 64 x 32 wave tile).  Measured (round 2):
    // SLOWER - 109.7 vs 118.8 TFLOP/s on 128->128 3x3 at 160x160, 109 vs 114 at 80x80.  The premise was wrong: a 32x32x2 fp32 MFMA
    // occupies the pipe for 64 cycles, so the 128 x 128 form's operand reads + tile writes are ~31 B/clk per CU, a quarter of the LDS
    // port - LDS bandwidth is not what holds the kernel at 72-78 % MFMA busy; and the big tile's 110 KB of LDS leave one workgroup per CU
    // (two waves per SIMD instead of four), which costs latency hiding.  Kept behind SOMI_CONV_BIG=1 for experiments.
    static const int big = getenv("SOMI_CONV_BIG") ? atoi(getenv("SOMI_CONV_BIG")) : 0;
    if (big && p.variant == 4 && fast_path(d) && !d.per_sample_w && !d.a_chan_scale && !d.a_pix_scale && M >= 256 * 256) p.variant = 6;
    if (sk_ok) {
        const int bm = kTileBM[p.variant], bn = kTileBN[p.variant];
        const long ntile = (long)cdiv(M, bm) * cdiv(d.Cout, bn), nkt = (long)d.kh * d.kw * d.Cin / BK;
        const int slots = sk_slots(p.variant, d);
        const long rounds = (ntile + slots - 1) / slots;
        static const int sk_pct = getenv("SOMI_SK_PCT") ? atoi(getenv("SOMI_SK_PCT")) : 90;   // stream-K below this slot efficiency (%)
        p.sk = ntile * 100 < rounds * slots * sk_pct && ntile * nkt >= 32 &&       // (the grid shrinks to >= 8 K-tiles per workgroup)
               d.workspace_bytes >= (size_t)slots * 2 * bm * bn * sizeof(float);
    }
    return p;
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
static int launch(const ConvArgs &a, bool sk, hipStream_t s) {
    ConvArgs args = a;
    const bool mod = a.d.a_chan_scale || a.d.a_pix_scale;
    const bool fast = fast_path(a.d);
    args.cls = a.dgrad && fast;
    args.sk = sk ? 1 : 0;
    args.ws = static_cast<float *>(a.d.workspace);
    const int st = a.d.stride, ncls = args.cls ? st * st : 1;
    const int m_cls = args.cls ? (a.d.per_sample_w ? 1 : a.d.B) * cdiv(a.d.Ho, st) * cdiv(a.d.Wo, st) : a.M;   // largest class
    args.tiles_m = cdiv(m_cls, BM);
    args.tiles_n = cdiv(a.d.Cout, BN);
    // stream-K grid: all 512 slots, unless that would leave a workgroup fewer than min_kt K-tiles (prologue, partial store and
    // fix-up then cost more than the MFMA work of the piece)
    static const int min_kt = getenv("SOMI_SK_MIN_KT") ? atoi(getenv("SOMI_SK_MIN_KT")) : 8;
    int sk_grid = BM + BN > 320 ? SK_GRID / 2 : (WAVES_M * WAVES_N == 8 && BM + BN <= 192 && fast && three_per_cu(5, a.d) ? SK_GRID * 3 / 2 : SK_GRID);
    int fix_grid = 0;
    args.sk_whole = 0;
    args.sk_rem_g = args.sk_grid_main = 1;
    if (sk) {
        const long ntile = (long)args.tiles_m * args.tiles_n, nkt = a.K / BK, U = ntile * nkt;
        if (U / min_kt < sk_grid) sk_grid = (int)(U / min_kt);
        if (sk_grid < 2) sk_grid = 2;
        // hybrid (SOMI_SK_HYBRID=1): whole rounds first, only the remainder streamed.  Measured round 4 (interleaved 40-step runs on one box):
        // 330.1 - 330.8 ms per step against 329.6 - 329.9 with everything streamed - the fix-up shrinks, but the short remainder pieces and the
        // lost balance cost as much.  Off by default.
        static const int hybrid = getenv("SOMI_SK_HYBRID") ? atoi(getenv("SOMI_SK_HYBRID")) : 0;
        const int whole = hybrid ? (int)(ntile / sk_grid) : 0;
        const long u_rem = (ntile - (long)whole * sk_grid) * nkt;
        int gr = sk_grid;
        static const int rem_min_kt = getenv("SOMI_SK_REM_MIN_KT") ? atoi(getenv("SOMI_SK_REM_MIN_KT")) : 4;
        if (whole > 0 && u_rem / rem_min_kt < gr) gr = (int)(u_rem / rem_min_kt);   // no streamed piece shorter than this many K-tiles
        if (gr < 1) gr = 1;
        args.sk_whole = whole;
        args.sk_rem_g = gr;
        args.sk_grid_main = sk_grid;
        fix_grid = u_rem > 0 ? gr - 1 : 0;                              // one seam per boundary between streamed runs
    }
    const dim3 grid(sk ? sk_grid : args.tiles_m * args.tiles_n, ncls, a.d.per_sample_w ? a.d.B : 1);
    if constexpr (WAVES_M * WAVES_N == 8 && BM + BN <= 320) {
        if (a.ns && fast && !mod) {                                // reduced-precision forms (opt-in): same schedule, fix-up and epilogue
            if (a.ns == 1)
                hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WAVES_M, WAVES_N, false, true, 1>), grid, dim3(WAVES_M * WAVES_N * 64), 0, s, args);
            else
                hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WAVES_M, WAVES_N, false, true, 2>), grid, dim3(WAVES_M * WAVES_N * 64), 0, s, args);
            if (sk && fix_grid > 0)
                hipLaunchKernelGGL((conv_streamk_fixup_kernel<BM, BN, WAVES_M, WAVES_N>), dim3(fix_grid), dim3(WAVES_M * WAVES_N * 64), 0, s, args, args.sk_rem_g);
            return launch_status("somi_conv2d_nhwc_f32 (bf16)");
        }
    }
    if (mod && fast)
        hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WAVES_M, WAVES_N, true, true>), grid, dim3(WAVES_M * WAVES_N * 64), 0, s, args);
    else if (mod)
        hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WAVES_M, WAVES_N, true, false>), grid, dim3(WAVES_M * WAVES_N * 64), 0, s, args);
    else if (fast)
        hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WAVES_M, WAVES_N, false, true>), grid, dim3(WAVES_M * WAVES_N * 64), 0, s, args);
    else
        hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WAVES_M, WAVES_N, false, false>), grid, dim3(WAVES_M * WAVES_N * 64), 0, s, args);
    if (sk && fix_grid > 0)
        hipLaunchKernelGGL((conv_streamk_fixup_kernel<BM, BN, WAVES_M, WAVES_N>), dim3(fix_grid), dim3(WAVES_M * WAVES_N * 64), 0, s, args, args.sk_rem_g);
    return launch_status("somi_conv2d_nhwc_f32");
}

}  // namespace somi

namespace somi {
static int conv_launch(const somi_conv_desc *dp, somi_stream_t stream, int dgrad) {
    SOMI_REQUIRE(dp, SOMI_EINVAL, "conv: null descriptor");
    const somi_conv_desc &d = *dp;
    SOMI_REQUIRE(d.x && d.w && d.y, SOMI_EINVAL, "conv: null tensor");
    SOMI_REQUIRE(d.B > 0 && d.H > 0 && d.W > 0 && d.Cin > 0 && d.Cout > 0, SOMI_EINVAL, "conv: empty shape");
    SOMI_REQUIRE(d.kh > 0 && d.kw > 0 && d.stride > 0 && d.dil > 0 && d.pad >= 0, SOMI_EINVAL, "conv: bad geometry");
    SOMI_REQUIRE(dgrad || (d.Ho == (d.H + 2 * d.pad - (d.dil * (d.kh - 1) + 1)) / d.stride + 1 &&
                           d.Wo == (d.W + 2 * d.pad - (d.dil * (d.kw - 1) + 1)) / d.stride + 1),
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

    SOMI_REQUIRE(d.y_cs % 4 == 0 && d.y_coff % 4 == 0 && aligned16(d.y), SOMI_EINVAL,
                 "conv: y_cs (%d), y_coff (%d) must be multiples of 4 and y 16 B aligned", d.y_cs, d.y_coff);
    SOMI_REQUIRE(!d.residual || (d.res_cs % 4 == 0 && d.res_coff % 4 == 0 && aligned16(d.residual)), SOMI_EINVAL,
                 "conv: residual stride / offset must be multiples of 4 and 16 B aligned");
    SOMI_REQUIRE(!d.residual2 || (d.res2_cs % 4 == 0 && d.res2_coff % 4 == 0 && aligned16(d.residual2) && d.res2_coff + d.Cout <= d.res2_cs),
                 SOMI_EINVAL, "conv: residual2 slice must be 16 B aligned and inside its channel stride");
    SOMI_REQUIRE((!d.bias || aligned16(d.bias)) && (!d.post_scale || (aligned16(d.post_scale) && aligned16(d.post_shift))),
                 SOMI_EINVAL, "conv: bias / post_scale / post_shift must be 16 B aligned");
    ConvArgs a;
    a.d = d;
    a.K = d.kh * d.kw * d.Cin;
    a.M = d.per_sample_w ? d.Ho * d.Wo : d.B * d.Ho * d.Wo;
    a.tiles_m = a.tiles_n = 0;
    const size_t xb = (size_t)d.B * d.H * d.W * d.x_cs * 4, wb = (size_t)(d.per_sample_w ? d.B : 1) * d.Cout * a.K * 4;
    if (xb > MAX_BUF_BYTES || wb > MAX_BUF_BYTES) {
        // the operands are fetched through 32-bit buffer descriptors: run the batch in slices that fit (images are independent rows)
        const size_t per_img = (size_t)d.H * d.W * d.x_cs * 4, per_w = d.per_sample_w ? (size_t)d.Cout * a.K * 4 : 0;
        SOMI_REQUIRE(per_img <= MAX_BUF_BYTES && (size_t)d.Cout * a.K * 4 <= MAX_BUF_BYTES, SOMI_ENOTIMPL,
                     "conv: one image (%zu B) or one weight set exceeds the 4 GiB buffer-descriptor range", per_img);
        size_t bsub = MAX_BUF_BYTES / per_img;
        if (per_w && MAX_BUF_BYTES / per_w < bsub) bsub = MAX_BUF_BYTES / per_w;
        for (int b0 = 0; b0 < d.B; b0 += (int)bsub) {
            somi_conv_desc sub = d;
            sub.B = d.B - b0 < (int)bsub ? d.B - b0 : (int)bsub;
            sub.x = d.x + (size_t)b0 * d.H * d.W * d.x_cs;
            sub.y = d.y + (size_t)b0 * d.Ho * d.Wo * d.y_cs;
            if (d.residual) sub.residual = d.residual + (size_t)b0 * d.Ho * d.Wo * d.res_cs;
            if (d.residual2) sub.residual2 = d.residual2 + (size_t)b0 * d.Ho * d.Wo * d.res2_cs;
            SOMI_REQUIRE(!d.stat_sum, SOMI_ENOTIMPL, "conv: statistics of a tensor beyond the 4 GiB descriptor range");
            if (d.a_chan_scale) sub.a_chan_scale = d.a_chan_scale + (size_t)b0 * d.Cin;
            if (d.a_pix_scale) sub.a_pix_scale = d.a_pix_scale + (size_t)b0 * d.H * d.W;
            if (d.per_sample_w) {
                sub.w = d.w + (size_t)b0 * d.Cout * a.K;
                if (d.bias) sub.bias = d.bias + (size_t)b0 * d.Cout;
            }
            const int rc = conv_launch(&sub, stream, dgrad);
            if (rc) return rc;
        }
        return 0;
    }
    a.x_bytes = (unsigned)xb;
    a.w_bytes = (unsigned)wb;
    a.chan_bytes = (unsigned)((size_t)d.B * d.Cin * 4);
    a.pix_bytes = (unsigned)((size_t)d.B * d.H * d.W * 4);
    SOMI_REQUIRE(!d.workspace || aligned16(d.workspace), SOMI_EINVAL, "conv: workspace must be 16 B aligned");
    SOMI_REQUIRE(!d.stat_sum == !d.stat_sumsq, SOMI_EINVAL, "conv: stat_sum and stat_sumsq go together");
    SOMI_REQUIRE(!d.stat_sum || (d.Cout % 4 == 0 && !d.per_sample_w && !dgrad && aligned16(d.stat_sum) && aligned16(d.stat_sumsq) &&
                                 (!d.stat_pivot || aligned16(d.stat_pivot))),
                 SOMI_EINVAL, "conv: statistics need Cout %% 4 == 0, shared weights, 16 B aligned buffers");
    a.dgrad = dgrad;
    a.ns = (d.prec == 1 || d.prec == 2) ? d.prec : 0;
    a.cls = 0;
    a.sk = 0;
    a.ws = nullptr;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const TilePlan tp = plan_tiles(d, a.M, dgrad);
    switch (tp.variant) {
        case 0: return launch<128, 128, 2, 2>(a, tp.sk, s);
        case 1: return launch<64, 128, 1, 4>(a, tp.sk, s);
        case 2: return launch<128, 64, 2, 2>(a, tp.sk, s);
        case 4: return launch<128, 128, 2, 4>(a, tp.sk, s);
        case 5: return launch<128, 64, 4, 2>(a, tp.sk, s);
        case 6: return launch<256, 128, 4, 2>(a, tp.sk, s);
        case 7: return launch<256, 64, 8, 1>(a, tp.sk, s);
        default: return launch<128, 32, 4, 1>(a, tp.sk, s);
    }
}
}  // namespace somi

extern "C" int somi_conv2d_nhwc_f32(const somi_conv_desc *dp, somi_stream_t stream) { return somi::conv_launch(dp, stream, 0); }

// Data gradient of the convolution described by `f` (forward geometry: x (B,H,W,Cin) -> y (B,Ho,Wo,Cout)):
//   dx[b,h,w,ci] = sum_{r,q,co} dy[b, (h+p-r)/s, (w+p-q)/s, co] * W[co][ci][r][q]   over taps with (h+p-r) % s == 0 etc.
// The same implicit-GEMM kernel runs with rows = forward-input pixels and the reduction over (tap, co); `wt` is the
// dgrad packing [Cin][kh*kw*Cout] (k = (r*kw+q)*Cout + co), see somi_pack_dgrad_weights_f32.
extern "C" int somi_conv2d_dgrad_nhwc_f32(const somi_conv_desc *f, const float *dy, int dy_cs, int dy_coff, const float *wt,
                                          float *dx, int dx_cs, int dx_coff, const float *accumulate, int acc_cs, int acc_coff,
                                          somi_stream_t stream) {
    using namespace somi;
    SOMI_REQUIRE(f && dy && wt && dx, SOMI_EINVAL, "conv dgrad: null argument");
    SOMI_REQUIRE(f->dil == 1, SOMI_ENOTIMPL, "conv dgrad: dilation 1 only");
    SOMI_REQUIRE(f->Ho == (f->H + 2 * f->pad - f->kh) / f->stride + 1 && f->Wo == (f->W + 2 * f->pad - f->kw) / f->stride + 1,
                 SOMI_EINVAL, "conv dgrad: Ho/Wo do not match the forward geometry");
    somi_conv_desc g{};
    g.x = dy; g.w = wt; g.y = dx; g.residual = accumulate;
    g.B = f->B; g.H = f->Ho; g.W = f->Wo; g.Cin = f->Cout; g.x_cs = dy_cs; g.x_coff = dy_coff;      // source = dy
    g.Ho = f->H; g.Wo = f->W; g.Cout = f->Cin; g.y_cs = dx_cs; g.y_coff = dx_coff;                  // rows = forward-input pixels
    g.kh = f->kh; g.kw = f->kw; g.stride = f->stride; g.pad = f->pad; g.dil = 1;
    g.res_cs = acc_cs; g.res_coff = acc_coff; g.act = SOMI_ACT_NONE; g.per_sample_w = f->per_sample_w;
    g.workspace = f->workspace; g.workspace_bytes = f->workspace_bytes; g.prec = f->prec;
    g.residual2 = f->residual2; g.res2_cs = f->res2_cs; g.res2_coff = f->res2_coff;
    return conv_launch(&g, stream, 1);
}

extern "C" const char *somi_conv2d_kernel_name(const somi_conv_desc *dp) {
    if (!dp || dp->Cout <= 0 || dp->Ho <= 0 || dp->Wo <= 0 || dp->B <= 0) return nullptr;
    const int M = dp->per_sample_w ? dp->Ho * dp->Wo : dp->B * dp->Ho * dp->Wo;
    const int mod = (dp->a_chan_scale || dp->a_pix_scale) ? 1 : 0;
    const int fast = (dp->Cin % somi::BK == 0 && dp->kh * dp->kw <= 32) ? 1 : 0;
    static const char *tiles[8] = {"128,128,2,2", "64,128,1,4", "128,64,2,2", "128,32,4,1", "128,128,2,4", "128,64,4,2", "256,128,4,2", "256,64,8,1"};
    static thread_local char name[96];
    snprintf(name, sizeof(name), "conv_igemm_f32_kernel<%s,%s,%s>", tiles[somi::plan_tiles(*dp, M, 0).variant], mod ? "true" : "false",
             fast ? "true" : "false");
    return name;
}

extern "C" int somi_conv2d_stat_rows(const somi_conv_desc *dp) {
    if (!dp || dp->Cout <= 0 || dp->Ho <= 0 || dp->Wo <= 0 || dp->B <= 0 || dp->per_sample_w) return 0;
    static const int waves_m[8] = {2, 1, 2, 4, 2, 4, 4, 8};
    const int M = dp->B * dp->Ho * dp->Wo;
    const int v = somi::plan_tiles(*dp, M, 0).variant;
    return somi::cdiv(M, somi::kTileBM[v]) * waves_m[v];
}

extern "C" size_t somi_conv2d_workspace_bytes(void) { return (size_t)somi::SK_GRID * 2 * 128 * 128 * sizeof(float); }
