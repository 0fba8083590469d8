// DCNv3 forward / backward for gfx950.
//
// Replaces the reference extension's kernels (models/ops_dcnv3/src/cuda/dcnv3_im2col_cuda.cuh:216-275 forward,
// :82-147 + :278-839 backward) and host launchers (src/cuda/dcnv3_cuda.cu:21-174).
//
// The reference runs one thread per output scalar and every one of the Gc channel-threads of a sampling site
// re-reads that site's 3*K offset/mask floats and redoes the bilinear set-up.  Here a workgroup owns a tile of TP
// consecutive output pixels:
//   phase 1  every (pixel, group, point) "sampling record" of the tile is built ONCE: offset (8 B/lane) and mask
//            (4 B/lane) are read with unit-stride coalesced loads, the bilinear set-up is done by one lane per
//            record, and the record (4 tap offsets + 4 tap coefficients already multiplied by the mask; or the
//            fractions + validity bits for backward) is parked in LDS (32 B per record);
//   phase 2  lanes sweep (pixel, channel-quad) items: each lane pulls its group's record from LDS (broadcast across
//            the Gc/4 lanes of a group), gathers the 4 taps as 16 B loads (a group's channels are contiguous in
//            NHWC, so a tap is one 4*Gc-byte run) and accumulates; outputs leave as coalesced 16 B stores.
// HBM-bound: algorithmic bytes per output pixel = 4*(2*C + 3*G*K) forward (input once + output + offset + mask).
// Backward, two forms.
//   windowed (the fast path; needs a caller workspace, group widths 8/16/32/64):
//     A  dcnv3_win_kernel<Gc, 1> (dcnv3_bwd_om_kernel where the window does not fit): grad_offset / grad_mask - the forward's
//        gather plus the grad_output read and an in-wave reduction over a group's channels (DPP row operations / shuffle
//        butterflies instead of the reference's shared-memory tree); no atomics.
//     B  grad_input.  The reference scatters 4 taps x K points x C channels of fp32 atomics per output pixel (36.9 KB/px at C=256: the
//        kernel sat on the chip's 1.3 TB/s float-atomic rate, 2.6 % of the HBM roofline).  Here a workgroup owns an 8x8 tile of output
//        pixels of ONE group and a window of the input around it (tile + kernel reach + R pixels of offset slack); the window leaves
//        as plain stores into a staging slab [tile][cell][Gc].  Two forms, both without float atomics and run-to-run identical:
//        dcnv3_bwd_gin_mfma_kernel (32-wide groups, window up to ~230 cells): the window's gradient is the product S . go_tile of the (cells x 64 pixels)
//          coefficient matrix with the tile's grad_output; S is built densely in LDS by (pixel, corner) lanes - the four writers of a
//          column sit in one wave and never share a cell within an instruction, so plain read-add-write in program order is race-free
//          (ds_add_f32 atomics for the same adds cost 0.16 ms more) - and the product runs on the fp32 matrix cores: 0.48 ms at N32 80x80;
//        dcnv3_bwd_gin_kernel (the other widths, or when S does not fit): taps bucketed by window cell (integer counting sort in LDS),
//          every (cell, 4 channels) lane sums its list in EXACT fp64 (addends rounded onto a 2^-38 grid of the tile's largest
//          |grad_output|, so the sum does not depend on list order): 1.0 ms.  (Tried: LDS double atomics, 35 cycles per wave
//          instruction, 1.7 ms; lists sorted by tap id + fp32 sums in that order, 1.4 ms - the rank scans are dependent LDS reads.)
//     C  dcnv3_bwd_combine_kernel: every input pixel adds the (at most 2x2) windows that cover it in ascending tile order.
//     D  dcnv3_bwd_near_kernel: taps that leave their tile's window but land within two 8x8 input tiles of it ("near": offsets up to
//        ~10 px beyond the kernel footprint - everything a training run produces) are NOT scattered.  B only records, per (image, group,
//        tile), a 25-bit mask of the neighbouring destination tiles that receive such taps; D runs one workgroup per DESTINATION tile,
//        re-derives the sampling records of the flagged source tiles around it (same arithmetic as B), keeps the taps that left their
//        window and land in its own tile, and adds them in a fixed order (source tile ascending, record, corner) - plain
//        read-modify-write on pixels it alone owns.  No float atomics: grad_input is run-to-run bit-identical.
//     Only taps even farther out (the reference test's offsets of +-20 pixels; stride != 1) go to grad_input as fp32 atomics like the
//     reference's own; they are counted in the workspace's overflow word (0 <=> the launch was bit-reproducible).
//     B / C / D run over the batch in chunks of images through ONE staging slab capped at SOMI_DCN_SLAB_MB (default 1024 MB: a memory
//     bound, not a cache trick - slabs of 32 / 128 / 512 MB that would stay in the 256 MB Infinity Cache were measured in round 3 and
//     gain nothing, B is bound by its LDS build phase and C by its scattered reads; round 2 staged the whole batch, 2.95 GB at N32 160x160).
//   direct (no workspace, other group widths): one kernel, fp32 atomics into grad_input exactly as the reference does.
#include "common.h"

namespace somi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct DcnArgs {
    const float *input, *offset, *mask, *grad_output;
    float *output, *grad_input, *grad_offset, *grad_mask;
    int N, H, W, G, Gc, C, Ho, Wo, kh, kw, K, sh, sw, ph, pw, dh, dw;
    float offset_scale;
    int TP;          // pixels per tile
    long npix;       // N*Ho*Wo
    int ntile;
    // floats between consecutive pixels of offset / grad_offset (2*G*K when packed) and of mask / grad_mask (G*K when packed): a caller
    // that computes offset and mask logits with ONE 1x1 GEMM hands both as column ranges of the same [pixel][3*G*K] rows
    long off_ps, msk_ps;
    __device__ __forceinline__ long off_at(long pix, int gk) const { return pix * off_ps + gk * 2; }
    __device__ __forceinline__ long msk_at(long pix, int gk) const { return pix * msk_ps + gk; }
};

struct Rec {         // 32 B, one per (pixel, group, point)
    i32x4 off;       // element offsets of the 4 taps inside the image (clamped into range)
    f32x4 f;         // fwd: tap coefficients * mask ; bwd: {lh, lw, mask, validity bits as int}
};

// Builds the record for (pix, g, k).  Point order inside K: i over kernel_w outer, j over kernel_h inner
// (dcnv3_im2col_cuda.cuh:253-254), offsets stored (x, y).
template <bool BWD>
__device__ __forceinline__ Rec make_record(const DcnArgs &a, long pix, int g, int k) {
    const int wo = (int)(pix % a.Wo), ho = (int)((pix / a.Wo) % a.Ho);
    const float2 ofs = *reinterpret_cast<const float2 *>(a.offset + a.off_at(pix, g * a.K + k));
    const float m = a.mask[a.msk_at(pix, g * a.K + k)];
    const int i = k / a.kh, j = k % a.kh;
    const int half_w = (a.dw * (a.kw - 1)) >> 1, half_h = (a.dh * (a.kh - 1)) >> 1;
    const float p0w = (float)(half_w - a.pw + wo * a.sw) - (float)half_w * a.offset_scale;
    const float p0h = (float)(half_h - a.ph + ho * a.sh) - (float)half_h * a.offset_scale;
    const float loc_w = p0w + ((float)(i * a.dw) + ofs.x) * a.offset_scale;
    const float loc_h = p0h + ((float)(j * a.dh) + ofs.y) * a.offset_scale;
    Rec r;
    const bool use = loc_h > -1.f && loc_w > -1.f && loc_h < (float)a.H && loc_w < (float)a.W;
    const float fh = floorf(loc_h), fw = floorf(loc_w);
    const int h0 = use ? (int)fh : 0, w0 = use ? (int)fw : 0;
    const float lh = loc_h - fh, lw = loc_w - fw;
    const bool h0ok = use && h0 >= 0, h1ok = use && h0 + 1 <= a.H - 1;
    const bool w0ok = w0 >= 0, w1ok = w0 + 1 <= a.W - 1;
    const int hc0 = h0 < 0 ? 0 : h0, hc1 = h0 + 1 > a.H - 1 ? a.H - 1 : h0 + 1;
    const int wc0 = w0 < 0 ? 0 : w0, wc1 = w0 + 1 > a.W - 1 ? a.W - 1 : w0 + 1;
    r.off[0] = (hc0 * a.W + wc0) * a.C;
    r.off[1] = (hc0 * a.W + wc1) * a.C;
    r.off[2] = (hc1 * a.W + wc0) * a.C;
    r.off[3] = (hc1 * a.W + wc1) * a.C;
    const bool t0 = h0ok && w0ok, t1 = h0ok && w1ok, t2 = h1ok && w0ok, t3 = h1ok && w1ok;
    if constexpr (!BWD) {
        const float hh = 1.f - lh, hw = 1.f - lw;
        r.f[0] = t0 ? hh * hw * m : 0.f;
        r.f[1] = t1 ? hh * lw * m : 0.f;
        r.f[2] = t2 ? lh * hw * m : 0.f;
        r.f[3] = t3 ? lh * lw * m : 0.f;
    } else {
        r.f[0] = lh;
        r.f[1] = lw;
        r.f[2] = m;
        r.f[3] = __int_as_float((t0 ? 1 : 0) | (t1 ? 2 : 0) | (t2 ? 4 : 0) | (t3 ? 8 : 0));
    }
    return r;
}

// ------------------------------------------------------------------------------------------------ forward
// VEC = 4: one lane = 4 consecutive channels of one group (needs Gc % 4 == 0); VEC = 1: one lane = one channel.
template <int VEC>
__global__ __launch_bounds__(256) void dcnv3_fwd_kernel(const DcnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Rec *recs = reinterpret_cast<Rec *>(smem);
    const int tile = xcd_remap(blockIdx.x, a.ntile);
    const long pix0 = (long)tile * a.TP;
    const int np = (int)min((long)a.TP, a.npix - pix0);
    const int GK = a.G * a.K;
    for (int i = threadIdx.x; i < np * GK; i += 256) {
        const int pl = i / GK, gk = i % GK;
        recs[i] = make_record<false>(a, pix0 + pl, gk / a.K, gk % a.K);
    }
    __syncthreads();
    const int CV = a.C / VEC;
    const long img = (long)a.H * a.W * a.C;
    for (int it = threadIdx.x; it < np * CV; it += 256) {
        const int pl = it / CV, c = (it % CV) * VEC;
        const long pix = pix0 + pl;
        const int n = (int)(pix / ((long)a.Ho * a.Wo));
        const int g = c / a.Gc;
        const float *src = a.input + n * img + c;
        const Rec *rr = recs + (pl * a.G + g) * a.K;
        if constexpr (VEC == 4) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            int k = 0;
            for (; k + 3 <= a.K; k += 3) {                                 // 3 points = 12 gathers in flight; summation order unchanged
                const i32x4 o0 = rr[k].off, o1 = rr[k + 1].off, o2 = rr[k + 2].off;
                const f32x4 f0 = rr[k].f, f1 = rr[k + 1].f, f2 = rr[k + 2].f;
                f32x4 u[12];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    u[t] = *reinterpret_cast<const f32x4 *>(src + o0[t]);
                    u[4 + t] = *reinterpret_cast<const f32x4 *>(src + o1[t]);
                    u[8 + t] = *reinterpret_cast<const f32x4 *>(src + o2[t]);
                }
                acc += f0[0] * u[0] + f0[1] * u[1] + f0[2] * u[2] + f0[3] * u[3];
                acc += f1[0] * u[4] + f1[1] * u[5] + f1[2] * u[6] + f1[3] * u[7];
                acc += f2[0] * u[8] + f2[1] * u[9] + f2[2] * u[10] + f2[3] * u[11];
            }
            for (; k < a.K; ++k) {
                const i32x4 o = rr[k].off;
                const f32x4 f = rr[k].f;
                const f32x4 v0 = *reinterpret_cast<const f32x4 *>(src + o[0]);
                const f32x4 v1 = *reinterpret_cast<const f32x4 *>(src + o[1]);
                const f32x4 v2 = *reinterpret_cast<const f32x4 *>(src + o[2]);
                const f32x4 v3 = *reinterpret_cast<const f32x4 *>(src + o[3]);
                acc += f[0] * v0 + f[1] * v1 + f[2] * v2 + f[3] * v3;
            }
            *reinterpret_cast<f32x4 *>(a.output + pix * a.C + c) = acc;
        } else {
            float acc = 0.f;
            for (int k = 0; k < a.K; ++k) {
                const i32x4 o = rr[k].off;
                const f32x4 f = rr[k].f;
                acc += f[0] * src[o[0]] + f[1] * src[o[1]] + f[2] * src[o[2]] + f[3] * src[o[3]];
            }
            a.output[pix * a.C + c] = acc;
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward
// LDS: records, then 3 float accumulators per record (grad_mask, grad_w, grad_h partial sums over channels).
template <int VEC, bool SCATTER>
__global__ __launch_bounds__(256) void dcnv3_bwd_kernel(const DcnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int GK = a.G * a.K;
    Rec *recs = reinterpret_cast<Rec *>(smem);
    float *accs = reinterpret_cast<float *>(smem + (size_t)a.TP * GK * sizeof(Rec));   // [np*GK][3]
    const int tile = xcd_remap(blockIdx.x, a.ntile);
    const long pix0 = (long)tile * a.TP;
    const int np = (int)min((long)a.TP, a.npix - pix0);
    for (int i = threadIdx.x; i < np * GK; i += 256) {
        const int pl = i / GK, gk = i % GK;
        recs[i] = make_record<true>(a, pix0 + pl, gk / a.K, gk % a.K);
        accs[i * 3] = 0.f;
        accs[i * 3 + 1] = 0.f;
        accs[i * 3 + 2] = 0.f;
    }
    __syncthreads();
    const int CV = a.C / VEC;
    const long img = (long)a.H * a.W * a.C;
    const int LG = a.Gc / VEC;                                 // lanes per group
    // groups are LG consecutive lanes, aligned because CV = G*LG: butterflies need LG to be a power of two <= 64
    const bool shuffle = LG <= 64 && (LG & (LG - 1)) == 0;
    const int lane = threadIdx.x & 63;
    const int nit = np * CV;
    const int nit_pad = (nit + 255) / 256 * 256;               // keep whole waves in the loop for the shuffles
    for (int it = threadIdx.x; it < nit_pad; it += 256) {
        const bool live = it < nit;
        const int itc = live ? it : nit - 1;
        const int pl = itc / CV, c = (itc % CV) * VEC;
        const long pix = pix0 + pl;
        const int n = (int)(pix / ((long)a.Ho * a.Wo));
        const int g = c / a.Gc;
        const float *src = a.input + n * img + c;
        float *gin = a.grad_input + n * img + c;
        const int rbase = (pl * a.G + g) * a.K;
        f32x4 tg = {0.f, 0.f, 0.f, 0.f};
        if (live) {
            if constexpr (VEC == 4) tg = *reinterpret_cast<const f32x4 *>(a.grad_output + pix * a.C + c);
            else tg[0] = a.grad_output[pix * a.C + c];
        }
        for (int k = 0; k < a.K; ++k) {
            const Rec r = recs[rbase + k];
            const float lh = r.f[0], lw = r.f[1], m = r.f[2];
            const int bits = __float_as_int(r.f[3]);
            const float hh = 1.f - lh, hw = 1.f - lw;
            float gm = 0.f, gw = 0.f, gh = 0.f;
            if (bits) {                                        // wave-divergent only where groups differ
                const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
                f32x4 u1 = {0.f, 0.f, 0.f, 0.f}, u2 = u1, u3 = u1, u4 = u1;      // one 16-byte gather per tap, like the forward
                if constexpr (VEC == 4) {
                    if (bits & 1) u1 = *reinterpret_cast<const f32x4 *>(src + r.off[0]);
                    if (bits & 2) u2 = *reinterpret_cast<const f32x4 *>(src + r.off[1]);
                    if (bits & 4) u3 = *reinterpret_cast<const f32x4 *>(src + r.off[2]);
                    if (bits & 8) u4 = *reinterpret_cast<const f32x4 *>(src + r.off[3]);
                } else {
                    if (bits & 1) u1[0] = src[r.off[0]];
                    if (bits & 2) u2[0] = src[r.off[1]];
                    if (bits & 4) u3[0] = src[r.off[2]];
                    if (bits & 8) u4[0] = src[r.off[3]];
                }
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float t = tg[e], tm = t * m;
                    const float v1 = u1[e], v2 = u2[e], v3 = u3[e], v4 = u4[e];
                    if (SCATTER && live) {
                        if (bits & 1) atomicAdd(gin + r.off[0] + e, w1 * tm);
                        if (bits & 2) atomicAdd(gin + r.off[1] + e, w2 * tm);
                        if (bits & 4) atomicAdd(gin + r.off[2] + e, w3 * tm);
                        if (bits & 8) atomicAdd(gin + r.off[3] + e, w4 * tm);
                    }
                    // dcnv3_col2im_bilinear: grad_h_weight / grad_w_weight (dcnv3_im2col_cuda.cuh:112-141)
                    const float ghw = -hw * v1 - lw * v2 + hw * v3 + lw * v4;
                    const float gww = -hh * v1 + hh * v2 - lh * v3 + lh * v4;
                    gm += t * (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
                    gw += gww * tm;
                    gh += ghw * tm;
                }
            }
            if (shuffle) {
                for (int o = LG >> 1; o > 0; o >>= 1) {
                    gm += __shfl_xor(gm, o);
                    gw += __shfl_xor(gw, o);
                    gh += __shfl_xor(gh, o);
                }
                if (live && (lane & (LG - 1)) == 0) {          // one lane per group owns the record: plain store
                    float *ac = accs + (rbase + k) * 3;
                    ac[0] = gm;
                    ac[1] = gw;
                    ac[2] = gh;
                }
            } else if (live) {
                float *ac = accs + (rbase + k) * 3;
                atomicAdd(ac, gm);
                atomicAdd(ac + 1, gw);
                atomicAdd(ac + 2, gh);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < np * GK; i += 256) {
        const long pix = pix0 + i / GK;
        const int gk = i % GK;
        a.grad_mask[a.msk_at(pix, gk)] = accs[i * 3];
        *reinterpret_cast<float2 *>(a.grad_offset + a.off_at(pix, gk)) =
            make_float2(a.offset_scale * accs[i * 3 + 1], a.offset_scale * accs[i * 3 + 2]);
    }
}

// ------------------------------------------------------------------------------------------------ backward A: grad_offset / grad_mask
// The forward's structure (records in LDS, 3 sampling points = 12 sixteen-byte gathers in flight per lane) plus the grad_output read and a
// butterfly over the LG = Gc/4 lanes of a group; no atomics.  Needs Gc/4 a power of two <= 64.
__global__ __launch_bounds__(256) void dcnv3_bwd_om_kernel(const DcnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Rec *recs = reinterpret_cast<Rec *>(smem);
    const int tile = xcd_remap(blockIdx.x, a.ntile);
    const long pix0 = (long)tile * a.TP;
    const int np = (int)min((long)a.TP, a.npix - pix0);
    const int GK = a.G * a.K;
    for (int i = threadIdx.x; i < np * GK; i += 256) {
        const int pl = i / GK, gk = i % GK;
        recs[i] = make_record<true>(a, pix0 + pl, gk / a.K, gk % a.K);
    }
    __syncthreads();
    const int CV = a.C / 4, LG = a.Gc / 4, lane = threadIdx.x & 63;
    const long img = (long)a.H * a.W * a.C;
    const int nit = np * CV, nit_pad = (nit + 255) / 256 * 256;       // whole waves stay in the loop for the shuffles
    for (int it = threadIdx.x; it < nit_pad; it += 256) {
        const bool live = it < nit;
        const int itc = live ? it : nit - 1;
        const int pl = itc / CV, c = (itc % CV) * 4;
        const long pix = pix0 + pl;
        const int n = (int)(pix / ((long)a.Ho * a.Wo));
        const int g = c / a.Gc;
        const float *src = a.input + n * img + c;
        const Rec *rr = recs + (pl * a.G + g) * a.K;
        const f32x4 tg = live ? *reinterpret_cast<const f32x4 *>(a.grad_output + pix * a.C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        const int gk0 = g * a.K;
        for (int k0 = 0; k0 < a.K; k0 += 3) {
            const int nk = min(3, a.K - k0);
            f32x4 u[12];
            int bits[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const Rec &r = rr[k0 + (j < nk ? j : 0)];
                bits[j] = j < nk ? __float_as_int(r.f[3]) : 0;
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    u[j * 4 + t] = (bits[j] >> t) & 1 ? *reinterpret_cast<const f32x4 *>(src + r.off[t]) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (j >= nk) break;                                  // wave-uniform
                const Rec &r = rr[k0 + j];
                const float lh = r.f[0], lw = r.f[1], m = r.f[2], hh = 1.f - lh, hw = 1.f - lw;
                const f32x4 v1 = u[j * 4], v2 = u[j * 4 + 1], v3 = u[j * 4 + 2], v4 = u[j * 4 + 3];
                // dcnv3_col2im_bilinear: value, grad_h_weight, grad_w_weight (dcnv3_im2col_cuda.cuh:112-141)
                const f32x4 val = (hh * hw) * v1 + (hh * lw) * v2 + (lh * hw) * v3 + (lh * lw) * v4;
                const f32x4 ghw = hw * (v3 - v1) + lw * (v4 - v2);
                const f32x4 gww = hh * (v2 - v1) + lh * (v4 - v3);
                float gm = (tg[0] * val[0] + tg[1] * val[1]) + (tg[2] * val[2] + tg[3] * val[3]);
                float gw = ((tg[0] * gww[0] + tg[1] * gww[1]) + (tg[2] * gww[2] + tg[3] * gww[3])) * m;
                float gh = ((tg[0] * ghw[0] + tg[1] * ghw[1]) + (tg[2] * ghw[2] + tg[3] * ghw[3])) * m;
                for (int o = LG >> 1; o > 0; o >>= 1) {
                    gm += __shfl_xor(gm, o);
                    gw += __shfl_xor(gw, o);
                    gh += __shfl_xor(gh, o);
                }
                if (live && (lane & (LG - 1)) == 0) {                // one lane per group owns the point: plain stores
                    a.grad_mask[a.msk_at(pix, gk0 + k0 + j)] = gm;
                    *reinterpret_cast<float2 *>(a.grad_offset + a.off_at(pix, gk0 + k0 + j)) =
                        make_float2(a.offset_scale * gw, a.offset_scale * gh);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward B / C: grad_input
constexpr int GIN_TH = 8, GIN_TW = 8, GIN_TP = GIN_TH * GIN_TW;      // output pixels per workgroup (one group)
struct GinGeo {
    int R;                       // offset slack in input pixels around the kernel footprint
    int tiles_h, tiles_w;
    int WH, WW;                  // window extent (input pixels)
    int lo_h, lo_w;              // window origin = tile origin * stride + lo
    unsigned rkw, rWW;           // floor(65536/d) + 1 for d = kernel_w, WW: x / d == (x * r) >> 16 while x < 65536 / d (win_plan checks)
    float *staging;              // [N][G][tiles][WH*WW][Gc]
    unsigned *overflow;          // FAR taps (added with fp32 atomics): 0 <=> bit-reproducible
    unsigned *near;              // [N][G][tiles] masks of the destination tiles that receive a tile's NEAR taps; nullptr: no near pass (stride != 1)
    unsigned *near_any;          // one word per chunk of images: != 0 iff any tile of the chunk has near taps (the near pass exits on 0)
    int nch, ncw, col_h, col_w;  // COLOURED form of backward B: this launch takes the tiles (col_h + nch * i, col_w + ncw * j) - windows of one colour are disjoint
};

// A tap at input pixel (h, w) that fell outside its tile's window: its destination tile relative to the tile under the window's centre.
// Within +-2 tiles both ways -> `bit` in the tile's 5x5 mask and true (the near pass D adds it); else false (far: atomics).
__device__ __forceinline__ bool near_bit(const GinGeo &q, int h, int w, int win_h0, int win_w0, int &bit) {
    const int dth = (h >> 3) - ((win_h0 + (q.WH >> 1)) >> 3), dtw = (w >> 3) - ((win_w0 + (q.WW >> 1)) >> 3);
    bit = (dth + 2) * 5 + dtw + 2;
    return q.near != nullptr && (unsigned)(dth + 2) <= 4u && (unsigned)(dtw + 2) <= 4u;
}
struct RecG {                    // one sampling point of one output pixel: floor position and the four tap coefficients x mask
    int h0, w0;
    float cf[4];                 // 0 for a tap outside the image / an unused point
};

constexpr int GIN_OVF_CAP = 768;  // LDS list of taps beyond the window; further ones are added by the filing lane itself
struct OvfG {                    // a tap beyond the window: its image pixel instead of a cell
    int hw, px;
    float cf;
};

// LDS (dynamic): tgt [GIN_TP][GC] floats | recs [GIN_TP*K] | ent_cf [GIN_TP*K*4] floats | ovf [GIN_OVF_CAP] | cnt, start, cur [ncell+1] ints |
// ent_px [GIN_TP*K*4] bytes - 48 KB at K = 9, Gc = 32: three workgroups per CU
template <int GC>
__global__ __launch_bounds__(256) void dcnv3_bwd_gin_kernel(const DcnArgs a, const GinGeo q) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ncell = q.WH * q.WW, nrec = GIN_TP * a.K, ntap = nrec * 4;
    float *tgt = reinterpret_cast<float *>(smem);
    RecG *recs = reinterpret_cast<RecG *>(tgt + GIN_TP * GC);
    float *ent_cf = reinterpret_cast<float *>(recs + nrec);
    OvfG *ovf = reinterpret_cast<OvfG *>(ent_cf + ntap);
    int *cnt = reinterpret_cast<int *>(ovf + GIN_OVF_CAP), *start = cnt + ncell + 1, *cur = start + ncell + 1;
    uint8_t *ent_px = reinterpret_cast<uint8_t *>(cur + ncell + 1);
    __shared__ float red[4];
    __shared__ int wsum[4];
    __shared__ int novf;
    __shared__ unsigned nearmask;
    const int tile = blockIdx.x, n = blockIdx.y, g = blockIdx.z;
    const int th0 = (tile / q.tiles_w) * GIN_TH, tw0 = (tile % q.tiles_w) * GIN_TW;
    const int win_h0 = th0 * a.sh + q.lo_h, win_w0 = tw0 * a.sw + q.lo_w;
    for (int i = threadIdx.x; i <= ncell; i += 256) { cnt[i] = 0; cur[i] = 0; }
    if (threadIdx.x == 0) { novf = 0; nearmask = 0u; }
    // the tile's grad_output for this group, and its largest magnitude
    constexpr int SLOTS = 256 / GC, NPX = GIN_TP / SLOTS;
    const int c = threadIdx.x % GC, slot = threadIdx.x / GC;
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
        const int pl = slot + i * SLOTS;
        const int ho = th0 + pl / GIN_TW, wo = tw0 + pl % GIN_TW;
        const float t = (ho < a.Ho && wo < a.Wo) ? a.grad_output[(((long)n * a.Ho + ho) * a.Wo + wo) * a.C + g * GC + c] : 0.f;
        tgt[pl * GC + c] = t;
        mx = fmaxf(mx, fabsf(t));
    }
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    // 1. records (floor position + the four tap coefficients x mask) and the histogram of the taps over the window cells
    for (int i = threadIdx.x; i < nrec; i += 256) {
        const int pl = i / a.K, k = i % a.K;
        const int ho = th0 + pl / GIN_TW, wo = tw0 + pl % GIN_TW;
        RecG r;
        r.h0 = r.w0 = 0;
        r.cf[0] = r.cf[1] = r.cf[2] = r.cf[3] = 0.f;
        if (ho < a.Ho && wo < a.Wo) {
            const long pix = ((long)n * a.Ho + ho) * a.Wo + wo;
            const float2 ofs = *reinterpret_cast<const float2 *>(a.offset + a.off_at(pix, g * a.K + k));
            const float m = a.mask[a.msk_at(pix, g * a.K + k)];
            const int ii = k / a.kh, jj = k % a.kh;
            const int half_w = (a.dw * (a.kw - 1)) >> 1, half_h = (a.dh * (a.kh - 1)) >> 1;
            const float p0w = (float)(half_w - a.pw + wo * a.sw) - (float)half_w * a.offset_scale;
            const float p0h = (float)(half_h - a.ph + ho * a.sh) - (float)half_h * a.offset_scale;
            const float loc_w = p0w + ((float)(ii * a.dw) + ofs.x) * a.offset_scale;
            const float loc_h = p0h + ((float)(jj * a.dh) + ofs.y) * a.offset_scale;
            if (loc_h > -1.f && loc_w > -1.f && loc_h < (float)a.H && loc_w < (float)a.W) {
                const float fh = floorf(loc_h), fw = floorf(loc_w);
                const int h0 = (int)fh, w0 = (int)fw;
                const float lh = loc_h - fh, lw = loc_w - fw, hh = 1.f - lh, hw = 1.f - lw;
                const bool h0ok = h0 >= 0, h1ok = h0 + 1 <= a.H - 1, w0ok = w0 >= 0, w1ok = w0 + 1 <= a.W - 1;
                r.h0 = h0;
                r.w0 = w0;
                r.cf[0] = (h0ok && w0ok) ? hh * hw * m : 0.f;
                r.cf[1] = (h0ok && w1ok) ? hh * lw * m : 0.f;
                r.cf[2] = (h1ok && w0ok) ? lh * hw * m : 0.f;
                r.cf[3] = (h1ok && w1ok) ? lh * lw * m : 0.f;
            }
        }
        recs[i] = r;
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
            if (r.cf[tp] == 0.f) continue;
            const int wh = r.h0 + (tp >> 1) - win_h0, ww = r.w0 + (tp & 1) - win_w0;
            if ((unsigned)wh < (unsigned)q.WH && (unsigned)ww < (unsigned)q.WW) atomicAdd(&cnt[wh * q.WW + ww], 1);
        }
    }
    __syncthreads();
    // 2. exclusive scan of the histogram: every thread owns a run of bins, the run sums are scanned over the block
    const int bpt = (ncell + 255) / 256;
    int run = 0;
    for (int b = 0; b < bpt; ++b) { const int i = threadIdx.x * bpt + b; if (i < ncell) run += cnt[i]; }
    int inc = run;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if ((threadIdx.x & 63) >= o) inc += t; }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    int base = inc - run;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wsum[w];
    for (int b = 0; b < bpt; ++b) { const int i = threadIdx.x * bpt + b; if (i < ncell) { start[i] = base; base += cnt[i]; } }
    __syncthreads();
    // 3. file every tap under its cell (the order inside a cell's list is arbitrary: the sum below does not depend on it)
    for (int i = threadIdx.x; i < nrec; i += 256) {
        const RecG r = recs[i];
        const int pl = i / a.K;
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
            if (r.cf[tp] == 0.f) continue;
            const int h = r.h0 + (tp >> 1), w = r.w0 + (tp & 1);
            const int wh = h - win_h0, ww = w - win_w0;
            if ((unsigned)wh < (unsigned)q.WH && (unsigned)ww < (unsigned)q.WW) {
                const int cell = wh * q.WW + ww;
                const int at = start[cell] + atomicAdd(&cur[cell], 1);
                ent_cf[at] = r.cf[tp];
                ent_px[at] = (uint8_t)pl;
            } else if (int nb; near_bit(q, h, w, win_h0, win_w0, nb)) {
                atomicOr(&nearmask, 1u << nb);                     // left for the near pass (dcnv3_bwd_near_kernel)
            } else {
                const int at = atomicAdd(&novf, 1);
                if (at < GIN_OVF_CAP) {
                    ovf[at] = OvfG{h * a.W + w, pl, r.cf[tp]};
                } else {                                           // list full (offsets far beyond the slack everywhere): add it here
                    float *gp = a.grad_input + (((long)n * a.H + h) * a.W + w) * a.C + g * GC;
                    for (int cc = 0; cc < GC; ++cc) atomicAdd(gp + cc, tgt[pl * GC + cc] * r.cf[tp]);
                }
            }
        }
    }
    __syncthreads();
    // 4. one channel lane per (cell, channel): the cell's list summed in registers.  Addends are rounded onto the grid 2^(e-38) with
    //    2^e > mx: |addend| < 2^38 steps, at most 4*K*64 addends per cell -> every partial sum below 2^53 steps, i.e. exact in double
    int e2;
    (void)frexpf(mx, &e2);
    const double magic = ldexp(1.5, 52 + e2 - 38);               // (x + magic) - magic rounds x to a multiple of 2^(e2-38)
    const bool finite = mx > 0.f && mx < __builtin_huge_valf();
    float *dst = q.staging + (((long)n * a.G + g) * (q.tiles_h * q.tiles_w) + tile) * (long)ncell * GC;
    {
        constexpr int LG4 = GC / 4, SLOTS4 = 256 / LG4;          // four channels per lane: the list entry and its address are read once for them
        const int c4 = (threadIdx.x % LG4) * 4;
        for (int cell = threadIdx.x / LG4; cell < ncell; cell += SLOTS4) {
            double s0[4] = {0.0, 0.0, 0.0, 0.0}, s1[4] = {0.0, 0.0, 0.0, 0.0};   // exact sums: splitting the chain changes nothing but the latency
            if (finite) {
                int e = start[cell];
                const int e1 = e + cnt[cell];
                for (; e + 2 <= e1; e += 2) {
                    const float c0 = ent_cf[e], c1 = ent_cf[e + 1];
                    const f32x4 t0 = *reinterpret_cast<const f32x4 *>(tgt + ent_px[e] * GC + c4);
                    const f32x4 t1 = *reinterpret_cast<const f32x4 *>(tgt + ent_px[e + 1] * GC + c4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        s0[j] += ((double)(t0[j] * c0) + magic) - magic;
                        s1[j] += ((double)(t1[j] * c1) + magic) - magic;
                    }
                }
                if (e < e1) {
                    const float c0 = ent_cf[e];
                    const f32x4 t0 = *reinterpret_cast<const f32x4 *>(tgt + ent_px[e] * GC + c4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) s0[j] += ((double)(t0[j] * c0) + magic) - magic;
                }
            }
            f32x4 r;
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = (float)(s0[j] + s1[j]);
            *reinterpret_cast<f32x4 *>(dst + cell * GC + c4) = r;
        }
    }
    if (threadIdx.x == 0 && q.near) {
        q.near[((long)n * a.G + g) * (q.tiles_h * q.tiles_w) + tile] = nearmask;
        if (nearmask) atomicOr(q.near_any, 1u);
    }
    // 5. FAR taps: fp32 atomics into grad_input like the reference's own kernel (128 contiguous bytes per tap)
    if (novf) {
        float *gin = a.grad_input + (long)n * a.H * a.W * a.C + g * GC + c;
        const int nlist = novf < GIN_OVF_CAP ? novf : GIN_OVF_CAP;
        for (int i = slot; i < nlist; i += SLOTS) {
            const OvfG o = ovf[i];
            atomicAdd(gin + (long)o.hw * a.C, tgt[o.px * GC + c] * o.cf);
        }
        if (threadIdx.x == 0) atomicAdd(q.overflow, (unsigned)novf);
    }
}

// ---- backward B on the matrix cores.  The window's gradient is a product: G[cell][c] = sum_p S[cell][p] * go[p][c] with S the
// (cells x 64 tile pixels) matrix of tap coefficients - sparse (4*K of ~225 entries per column), but small enough to hold densely in
// LDS, and a 256 x 64 x 32 fp32 product is 256 MFMAs = 2 us per (tile, group) where the exact list sums above spend ~10.  S is
// built WITHOUT races or order dependence: lane (pixel p, corner t) walks the K points in order and adds its coefficient to
// S[cell][p].  Column p is written by the four corner lanes of pixel p only; they are neighbouring lanes of ONE wave (tid = 4p + t)
// and step through the points together: inside one LDS instruction they address four different cells (the corners of one point),
// and a wave's LDS instructions execute in program order, so a later point that lands on the same cell reads the earlier update.
// Hence plain read-add-write, no atomics (ds_add_f32 for the same adds made the kernel 35 % slower), and one fixed order of the adds.
// The MFMA sums over p in hardware order: fixed.  No exact-arithmetic tricks needed.
// LDS: S [ncell][68] floats | recs [64*K] (then, aliased, go^T [GC][68]) | ovf list: 76 KB at K = 9, Gc = 32, R = 2 (two per CU).
constexpr int GMM_LD = GIN_TP + 4;      // row stride of S and go^T in floats (272 B: a 16-lane group of ds_read_b128 covers all banks)
constexpr int GMM_OVF_CAP = 256;
struct RecM {                           // one sampling point: window coordinates of its floor position and the four coefficients x mask
    int wh, ww;                         // may lie outside [0, WH-2] x [0, WW-2]: those corners go to the overflow list
    float cf[4];
};
static_assert(sizeof(RecM) == 24, "step 2 reads the records as 6 dwords");

constexpr int GMM_MAX_CELLS = 320;    // window cells the coloured form tracks (15 x 15 = 225 at K = 9, R = 2)
constexpr int GMM_NT = 512;           // threads: LDS (S is 61 KB) allows two workgroups per CU, so each brings 8 waves for the non-MFMA phases

// COLOURED (round 4): no staging slab and no combine pass.  The windows of tiles that are `nch` x `ncw` tiles apart do not overlap (a window of
// an 8 x 8 tile with a 3 x 3 kernel and 2 px of slack is 15 x 15 input pixels, the tile pitch 8: every second tile both ways), so the tiles are
// run in nch * ncw launches of one COLOUR each, and a workgroup adds its window's product straight into grad_input with plain read-add-write -
// no other workgroup of the launch touches those pixels, and the launches of one stream run in order, so the adds to a pixel covered by several
// windows happen in the fixed order of the colours: deterministic, no atomics.  HBM traffic per call: the window's pixels read + written once
// per covering tile (2 x 0.74 GB at N32 80x80) instead of slab write + slab read + the combine's read-modify-write of grad_input (2.25 GB), and
// dcnv3_bwd_combine_kernel (0.19 / 0.27 ms per launch) is gone.  The values a lane will add to are requested BEFORE the matrix product.
template <int GC, bool COLOURED>
__global__ __launch_bounds__(GMM_NT) void dcnv3_bwd_gin_mfma_kernel(const DcnArgs a, const GinGeo q) {
    static_assert(GC == 32, "one 32-channel MFMA row block (64-wide groups: S + go^T exceed the LDS of two workgroups per CU, they take the list form)");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MB = GC / 32;
    const int ncell = q.WH * q.WW, nrec = GIN_TP * a.K;
    float *S = reinterpret_cast<float *>(smem);
    RecM *recs = reinterpret_cast<RecM *>(S + (size_t)ncell * GMM_LD);
    float *got = reinterpret_cast<float *>(recs);                 // go^T [GC][GMM_LD], written after the records are consumed
    const size_t mid = (size_t)nrec * sizeof(RecM) > (size_t)GC * GMM_LD * 4 ? (size_t)nrec * sizeof(RecM) : (size_t)GC * GMM_LD * 4;
    OvfG *ovf = reinterpret_cast<OvfG *>(reinterpret_cast<char *>(recs) + (mid + 15) / 16 * 16);
    __shared__ int novf;
    __shared__ unsigned nearmask;                                     // bits 0..24: destination tiles of NEAR taps; COLOURED: bit 31 = the tile has FAR taps
    __shared__ int touched[GMM_MAX_CELLS];                            // COLOURED: cell received a coefficient - the others are neither read nor written
    int tile = blockIdx.x;
    if constexpr (COLOURED) {                                          // the launch's colour class: tiles (col_h + nch * i, col_w + ncw * j)
        const int tcw = (q.tiles_w - q.col_w + q.ncw - 1) / q.ncw;
        tile = (q.col_h + q.nch * ((int)blockIdx.x / tcw)) * q.tiles_w + q.col_w + q.ncw * ((int)blockIdx.x % tcw);
    }
    const int n = blockIdx.y, g = blockIdx.z;
    const int th0 = (tile / q.tiles_w) * GIN_TH, tw0 = (tile % q.tiles_w) * GIN_TW;
    const int win_h0 = th0 * a.sh + q.lo_h, win_w0 = tw0 * a.sw + q.lo_w;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // 0. this thread's share of the tile's grad_output (kept in registers until the records are done with their LDS), zeros into S
    constexpr int NT = GMM_NT;
    constexpr int GQ = GC / 4, GPT = GIN_TP * GQ / NT;             // channel quads per pixel; float4 loads per thread
    static_assert(GPT >= 1 && GIN_TP * 4 <= NT, "thread count");
    f32x4 gr[GPT];
#pragma unroll
    for (int j = 0; j < GPT; ++j) {
        const int it = tid + j * NT, pl = it / GQ, cq = it % GQ;
        const int ho = th0 + pl / GIN_TW, wo = tw0 + pl % GIN_TW;
        gr[j] = (ho < a.Ho && wo < a.Wo) ? *reinterpret_cast<const f32x4 *>(a.grad_output + (((long)n * a.Ho + ho) * a.Wo + wo) * a.C + g * GC + cq * 4)
                                        : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // ... and the offsets / masks of this thread's first two sampling points: every global load of the workgroup is in flight before
    // anything waits (the chain load -> record -> next load was 3 latencies long: 15 us per workgroup with 8 waves per CU to hide it)
    constexpr int RB = 2;
    float2 ofs_r[RB];
    float m_r[RB];
#pragma unroll
    for (int j = 0; j < RB; ++j) {
        const int i = tid + j * NT;
        ofs_r[j] = make_float2(0.f, 0.f);
        m_r[j] = 0.f;
        if (i < nrec) {
            const int pl = i / a.K, k = i % a.K;
            const int ho = th0 + pl / GIN_TW, wo = tw0 + pl % GIN_TW;
            if (ho < a.Ho && wo < a.Wo) {
                const long pix = ((long)n * a.Ho + ho) * a.Wo + wo;
                ofs_r[j] = *reinterpret_cast<const float2 *>(a.offset + a.off_at(pix, g * a.K + k));
                m_r[j] = a.mask[a.msk_at(pix, g * a.K + k)];
            }
        }
    }
    for (int i = tid; i < ncell * GMM_LD / 4; i += NT) reinterpret_cast<f32x4 *>(S)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (COLOURED)
        for (int i = tid; i < GMM_MAX_CELLS; i += NT) touched[i] = 0;
    if (tid == 0) { novf = 0; nearmask = 0u; }
    // 1. records
    for (int i = tid, j = 0; i < nrec; i += NT, ++j) {
        const int pl = i / a.K, k = i % a.K;
        const int ho = th0 + pl / GIN_TW, wo = tw0 + pl % GIN_TW;
        RecM r;
        r.wh = r.ww = 0;
        r.cf[0] = r.cf[1] = r.cf[2] = r.cf[3] = 0.f;
        if (ho < a.Ho && wo < a.Wo) {
            float2 ofs;
            float m;
            if (j < RB) {
                ofs = j == 0 ? ofs_r[0] : ofs_r[1];
                m = j == 0 ? m_r[0] : m_r[1];
            } else {
                const long pix = ((long)n * a.Ho + ho) * a.Wo + wo;
                ofs = *reinterpret_cast<const float2 *>(a.offset + a.off_at(pix, g * a.K + k));
                m = a.mask[a.msk_at(pix, g * a.K + k)];
            }
            const int ii = k / a.kh, jj = k % a.kh;
            const int half_w = (a.dw * (a.kw - 1)) >> 1, half_h = (a.dh * (a.kh - 1)) >> 1;
            const float p0w = (float)(half_w - a.pw + wo * a.sw) - (float)half_w * a.offset_scale;
            const float p0h = (float)(half_h - a.ph + ho * a.sh) - (float)half_h * a.offset_scale;
            const float loc_w = p0w + ((float)(ii * a.dw) + ofs.x) * a.offset_scale;
            const float loc_h = p0h + ((float)(jj * a.dh) + ofs.y) * a.offset_scale;
            if (loc_h > -1.f && loc_w > -1.f && loc_h < (float)a.H && loc_w < (float)a.W) {
                const float fh = floorf(loc_h), fw = floorf(loc_w);
                const int h0 = (int)fh, w0 = (int)fw;
                const float lh = loc_h - fh, lw = loc_w - fw, hh = 1.f - lh, hw = 1.f - lw;
                const bool h0ok = h0 >= 0, h1ok = h0 + 1 <= a.H - 1, w0ok = w0 >= 0, w1ok = w0 + 1 <= a.W - 1;
                r.wh = h0 - win_h0;
                r.ww = w0 - win_w0;
                r.cf[0] = (h0ok && w0ok) ? hh * hw * m : 0.f;
                r.cf[1] = (h0ok && w1ok) ? hh * lw * m : 0.f;
                r.cf[2] = (h1ok && w0ok) ? lh * hw * m : 0.f;
                r.cf[3] = (h1ok && w1ok) ? lh * lw * m : 0.f;
            }
        }
        recs[i] = r;
    }
    __syncthreads();
    // 2. S: thread (pixel, corner) walks the points in order, three records' fields in flight (the first 256 threads)
    if (tid < GIN_TP * 4) {
        const int pl = tid >> 2, tp = tid & 3, dy = tp >> 1, dx = tp & 1;
        const int *rw = reinterpret_cast<const int *>(recs + pl * a.K);       // RecM = {wh, ww, cf[4]}: 6 dwords
        auto put = [&](int wh0, int ww0, float cf) {
            if (cf == 0.f) return;
            const int wh = wh0 + dy, ww = ww0 + dx;
            if ((unsigned)wh < (unsigned)q.WH && (unsigned)ww < (unsigned)q.WW) {
                S[(wh * q.WW + ww) * GMM_LD + pl] += cf;                      // plain read-add-write: see the note on column writers above
                if constexpr (COLOURED) touched[wh * q.WW + ww] = 1;          // (every writer stores the same value)
            } else if (int nb; near_bit(q, wh + win_h0, ww + win_w0, win_h0, win_w0, nb)) {
                atomicOr(&nearmask, 1u << nb);                                // left for the near pass (dcnv3_bwd_near_kernel)
            } else if constexpr (COLOURED) {
                atomicOr(&nearmask, 1u << 31);                                // FAR: dcnv3_bwd_far_kernel adds it after the last colour (an atomic
                                                                              // here could meet another workgroup's plain read-add-write)
            } else {
                const int at = atomicAdd(&novf, 1);
                if (at < GMM_OVF_CAP) ovf[at] = OvfG{(wh + win_h0) * a.W + ww + win_w0, pl, cf};   // a longer list: all of them again in step 5
            }
            // the four corner lanes of a pixel are neighbours in one wave and an LDS read-add-write pair of one put() must not be
            // interleaved with the next put()'s by the compiler: the wave's LDS instructions then execute in program order
            __builtin_amdgcn_wave_barrier();
        };
        int k = 0;
        for (; k + 3 <= a.K; k += 3) {
            int wh[3], ww[3];
            float cf[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                wh[j] = rw[(k + j) * 6];
                ww[j] = rw[(k + j) * 6 + 1];
                cf[j] = __int_as_float(rw[(k + j) * 6 + 2 + tp]);
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) put(wh[j], ww[j], cf[j]);
        }
        for (; k < a.K; ++k) put(rw[k * 6], rw[k * 6 + 1], __int_as_float(rw[k * 6 + 2 + tp]));
    }
    __syncthreads();
    // 3. go^T into the records' LDS: got[c][p]
#pragma unroll
    for (int j = 0; j < GPT; ++j) {
        const int it = tid + j * NT, pl = it / GQ, cq = it % GQ;
#pragma unroll
        for (int e = 0; e < 4; ++e) got[(cq * 4 + e) * GMM_LD + pl] = gr[j][e];
    }
    __syncthreads();
    // 4. G^T = go^T . S^T on the matrix cores: M = channels (32 per block), N = cells (32 per block), K = the 64 tile pixels.
    //    Operand fetch as in conv_igemm.hip: a lane reads 4 consecutive k of its row with one ds_read_b128, the two half-waves 4 apart,
    //    and MFMA step t pairs k = 8j + t (lanes 0-31) with k = 8j + 4 + t (lanes 32-63) - the same permutation on both operands.
    const int nblk = (ncell + 31) / 32;
    const int frag = (lane & 31) * GMM_LD + (lane >> 5) * 4;
    float *dst = COLOURED ? a.grad_input + (long)n * a.H * a.W * a.C + g * GC
                        : q.staging + (((long)n * a.G + g) * (q.tiles_h * q.tiles_w) + tile) * (long)ncell * GC;
    for (int cb = wave; cb < nblk; cb += NT / 64) {
        f32x16 acc[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mb][e] = 0.f;
        const int crow = cb * 32 + (lane & 31);
        const float *srow = S + (crow < ncell ? crow : ncell - 1) * GMM_LD + (lane >> 5) * 4;     // rows past the window: a valid row, never stored
        // COLOURED: this lane's cell as an image pixel, and what grad_input holds there now (earlier colours' adds) - requested before the product
        long gpix = -1;
        f32x4 cur[MB][4];
        if constexpr (COLOURED) {
            const int ch = crow / q.WW, h = win_h0 + ch, w = win_w0 + (crow - ch * q.WW);
            // a cell no sampling point reached (the slack ring around the kernel footprint, at small offsets: 40 % of the window) adds exact
            // zeros: skipping it saves its read AND its write
            if (crow < ncell && touched[crow] && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W) gpix = ((long)h * a.W + w) * a.C;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd)
                    cur[mb][qd] = gpix >= 0 ? *reinterpret_cast<const f32x4 *>(dst + gpix + mb * 32 + qd * 8 + (lane >> 5) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < GIN_TP / 8; ++j) {
            const f32x4 fs = *reinterpret_cast<const f32x4 *>(srow + j * 8);
            f32x4 fg[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) fg[mb] = *reinterpret_cast<const f32x4 *>(got + mb * 32 * GMM_LD + frag + j * 8);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fg[mb][t], fs[t], acc[mb], 0, 0, 0);
        }
        // D: lane holds cell n = lane % 32 and channels 8*(e/4) + 4*(lane/32) + e%4 of the block
        if constexpr (COLOURED) {
            if (gpix >= 0) {                                          // a cell outside the image received nothing (its coefficients are 0)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        const f32x4 v = {acc[mb][qd * 4], acc[mb][qd * 4 + 1], acc[mb][qd * 4 + 2], acc[mb][qd * 4 + 3]};
                        *reinterpret_cast<f32x4 *>(dst + gpix + mb * 32 + qd * 8 + (lane >> 5) * 4) = cur[mb][qd] + v;
                    }
            }
        } else if (crow < ncell) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const f32x4 v = {acc[mb][qd * 4], acc[mb][qd * 4 + 1], acc[mb][qd * 4 + 2], acc[mb][qd * 4 + 3]};
                    *reinterpret_cast<f32x4 *>(dst + (long)crow * GC + mb * 32 + qd * 8 + (lane >> 5) * 4) = v;
                }
        }
    }
    if (tid == 0 && q.near) {
        q.near[((long)n * a.G + g) * (q.tiles_h * q.tiles_w) + tile] = nearmask;
        if (nearmask) atomicOr(q.near_any, ((nearmask & 0x7fffffffu) ? 1u : 0u) | ((nearmask >> 31) ? 2u : 0u));
    }
    // 5. FAR taps: fp32 atomics into grad_input like the reference's own kernel (one channel per lane)
    if (!COLOURED && novf) {
        constexpr int SLOTS = NT / GC;
        const int c = tid % GC, slot = tid / GC;
        float *gin = a.grad_input + (long)n * a.H * a.W * a.C + g * GC + c;
        const int nlist = novf < GMM_OVF_CAP ? novf : GMM_OVF_CAP;
        const bool spilled = novf > GMM_OVF_CAP;
        if (!spilled) {
            for (int i = slot; i < nlist; i += SLOTS) {
                const OvfG o = ovf[i];
                atomicAdd(gin + (long)o.hw * a.C, got[c * GMM_LD + o.px] * o.cf);
            }
        } else {
            // the list was too short (offsets far beyond the slack everywhere): every out-of-window tap again, from the records' inputs
            for (int i = slot; i < nrec; i += SLOTS) {
                const int pl = i / a.K, k = i % a.K;
                const int ho = th0 + pl / GIN_TW, wo = tw0 + pl % GIN_TW;
                if (ho >= a.Ho || wo >= a.Wo) continue;
                const Rec r = make_record<true>(a, ((long)n * a.Ho + ho) * a.Wo + wo, g, k);
                const int bits = __float_as_int(r.f[3]);
                const float lh = r.f[0], lw = r.f[1], m = r.f[2], hh = 1.f - lh, hw = 1.f - lw;
                const float cf4[4] = {hh * hw * m, hh * lw * m, lh * hw * m, lh * lw * m};
                const float tv = got[c * GMM_LD + pl];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (!((bits >> t) & 1)) continue;
                    const int hwp = r.off[t] / a.C, h = hwp / a.W, w = hwp % a.W;
                    if ((unsigned)(h - win_h0) < (unsigned)q.WH && (unsigned)(w - win_w0) < (unsigned)q.WW) continue;   // went through S
                    if (int nb; near_bit(q, h, w, win_h0, win_w0, nb)) continue;                                          // the near pass adds it
                    atomicAdd(gin + (long)hwp * a.C, tv * cf4[t]);
                }
            }
        }
        if (tid == 0) atomicAdd(q.overflow, (unsigned)novf);
    }
}

// grad_input[n,h,w,c] += sum over the windows covering (h,w), ascending tile order (4 channels per lane)
__global__ __launch_bounds__(256) void dcnv3_bwd_combine_kernel(const DcnArgs a, const GinGeo q) {
    const int C4 = a.C >> 2;
    const long items = (long)a.N * a.H * a.W * C4;
    const int step_h = GIN_TH * a.sh, step_w = GIN_TW * a.sw, ntile = q.tiles_h * q.tiles_w;
    const long ncell = (long)q.WH * q.WW;
    for (long it = blockIdx.x * 256L + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
        const int c = (int)(it % C4) * 4;
        long p = it / C4;
        const int w = (int)(p % a.W);
        p /= a.W;
        const int h = (int)(p % a.H);
        const long n = p / a.H;
        const int g = c / a.Gc, cg = c % a.Gc;
        // tiles t with t*step + lo <= x <= t*step + lo + extent - 1
        const int xh = h - q.lo_h, xw = w - q.lo_w;
        int th_lo = xh - (q.WH - 1) <= 0 ? 0 : (xh - (q.WH - 1) + step_h - 1) / step_h, th_hi = xh < 0 ? -1 : xh / step_h;
        int tw_lo = xw - (q.WW - 1) <= 0 ? 0 : (xw - (q.WW - 1) + step_w - 1) / step_w, tw_hi = xw < 0 ? -1 : xw / step_w;
        if (th_hi > q.tiles_h - 1) th_hi = q.tiles_h - 1;
        if (tw_hi > q.tiles_w - 1) tw_hi = q.tiles_w - 1;
        float *o = a.grad_input + it * 4;
        f32x4 acc = *reinterpret_cast<const f32x4 *>(o);
        for (int th = th_lo; th <= th_hi; ++th)
            for (int tw = tw_lo; tw <= tw_hi; ++tw) {
                const long cell = (long)(xh - th * step_h) * q.WW + (xw - tw * step_w);
                acc += *reinterpret_cast<const f32x4 *>(q.staging + ((n * a.G + g) * ntile + th * q.tiles_w + tw) * ncell * a.Gc + cell * a.Gc + cg);
            }
        *reinterpret_cast<f32x4 *>(o) = acc;
    }
}

// ------------------------------------------------------------------------------------------------ backward B, coloured form: the far taps
// Runs after the last colour.  Persistent grid over (image, group, tile) that returns at once unless some tile flagged FAR taps (bit 31 of its
// mask word; offsets of +-20 pixels as in the reference's test - no training run produces them).  A flagged tile's sampling records are rebuilt
// with the arithmetic of make_record and every corner that left the window AND the near pass's reach goes to grad_input as
// an fp32 atomic, one channel per lane, exactly like the reference's own kernel (dcnv3_im2col_cuda.cuh:116-140); counted in q.overflow.
template <int GC>
__global__ __launch_bounds__(256) void dcnv3_bwd_far_kernel(const DcnArgs a, const GinGeo q) {
    if ((*q.near_any & 2u) == 0u) return;
    __shared__ unsigned cnt;
    const int ntile = q.tiles_h * q.tiles_w, nrec = GIN_TP * a.K, tid = threadIdx.x;
    constexpr int SLOTS = 256 / GC;
    const int c = tid % GC, slot = tid / GC;
    for (long item = blockIdx.x; item < (long)a.N * a.G * ntile; item += gridDim.x) {     // the index of the tile's mask word
        if (!(q.near[item] >> 31)) continue;                                             // workgroup-uniform
        const int tile = (int)(item % ntile), g = (int)((item / ntile) % a.G), n = (int)(item / ((long)ntile * a.G));
        const int th0 = (tile / q.tiles_w) * GIN_TH, tw0 = (tile % q.tiles_w) * GIN_TW;
        const int win_h0 = th0 * a.sh + q.lo_h, win_w0 = tw0 * a.sw + q.lo_w;
        float *gin = a.grad_input + (long)n * a.H * a.W * a.C + g * GC + c;
        if (tid == 0) cnt = 0u;
        __syncthreads();
        unsigned mine = 0;
        for (int i = slot; i < nrec; i += SLOTS) {
            const int pl = i / a.K, k = i % a.K;
            const int ho = th0 + pl / GIN_TW, wo = tw0 + pl % GIN_TW;
            if (ho >= a.Ho || wo >= a.Wo) continue;
            const long pix = ((long)n * a.Ho + ho) * a.Wo + wo;
            const Rec r = make_record<true>(a, pix, g, k);
            const int bits = __float_as_int(r.f[3]);
            const float lh = r.f[0], lw = r.f[1], m = r.f[2], hh = 1.f - lh, hw = 1.f - lw;
            const float cf4[4] = {hh * hw * m, hh * lw * m, lh * hw * m, lh * lw * m};
            const float tv = a.grad_output[pix * a.C + g * GC + c];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (!((bits >> t) & 1) || cf4[t] == 0.f) continue;
                const int hwp = r.off[t] / a.C, h = hwp / a.W, w = hwp % a.W;
                if ((unsigned)(h - win_h0) < (unsigned)q.WH && (unsigned)(w - win_w0) < (unsigned)q.WW) continue;   // went through S
                if (int nb; near_bit(q, h, w, win_h0, win_w0, nb)) continue;                                          // the near pass adds it
                atomicAdd(gin + (long)hwp * a.C, tv * cf4[t]);
                if (c == 0) ++mine;
            }
        }
        if (mine) atomicAdd(&cnt, mine);
        __syncthreads();
        if (tid == 0 && cnt) atomicAdd(q.overflow, cnt);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ backward D: the near taps
// One workgroup per (destination tile of 8x8 input pixels, image, group).  For every source tile within +-2 tiles whose mask names
// this destination: rebuild the source tile's sampling records (the arithmetic of backward B, so the same floor positions and
// coefficients), keep the corners that left the source's window and land in this tile, compact them in a fixed order
// (round j of records, thread, corner) and add grad_output[source pixel][c] * coefficient into an LDS image of the tile, one entry
// after the other (a cell row is only ever touched by the one thread group that owns its residue).  Finally the image is added to
// grad_input - pixels this workgroup alone owns at this point of the stream.  Stride 1 only (the launcher enables q.near for it).
constexpr int NEAR_CAP = 1024;                                        // candidates of one round: 256 threads x 4 corners
template <int GC>
__global__ __launch_bounds__(256) void dcnv3_bwd_near_kernel(const DcnArgs a, const GinGeo q) {
    __shared__ float accL[GIN_TP * GC];
    __shared__ int ent_key[NEAR_CAP];                                 // cell << 8 | source pixel of the tile
    __shared__ float ent_cf[NEAR_CAP];
    __shared__ int wsum[4];
    constexpr int NG = 256 / GC;                                      // thread groups; group j owns the cells with cell % NG == j
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = tid % GC, grp = tid / GC;
    if ((*q.near_any & 1u) == 0u) return;                             // the usual case: no tile of this chunk has a near tap (bit 1: far taps)
    const int dt_w = (a.W + 7) >> 3, dt_n = ((a.H + 7) >> 3) * dt_w;
    const int ntile = q.tiles_h * q.tiles_w;
    const int cb_h = (q.lo_h + (q.WH >> 1)) >> 3, cb_w = (q.lo_w + (q.WW >> 1)) >> 3;    // source tile t sits in destination tile t + cb
    const int nrec = GIN_TP * a.K;
    const int half_w = (a.dw * (a.kw - 1)) >> 1, half_h = (a.dh * (a.kh - 1)) >> 1;
    for (long item = blockIdx.x; item < (long)dt_n * a.N * a.G; item += gridDim.x) {    // (destination tile, image, group)
    const int dtile = (int)(item % dt_n), n = (int)((item / dt_n) % a.N), g = (int)(item / ((long)dt_n * a.N));
    const int ty = dtile / dt_w, tx = dtile % dt_w;
    const unsigned *masks = q.near + ((long)n * a.G + g) * ntile;
    // which of the 25 neighbours are flagged for this destination: lane s of every wave looks at neighbour s (one load latency per item
    // instead of 25 dependent ones: the idle pass took 0.18 ms at N32 80x80 with the sequential form), the ballot is wave-uniform
    bool mine = false;
    if (lane < 25) {
        const int th = ty + lane / 5 - 2 - cb_h, tw = tx + lane % 5 - 2 - cb_w;
        if ((unsigned)th < (unsigned)q.tiles_h && (unsigned)tw < (unsigned)q.tiles_w) {
            const int bit = (ty - (th + cb_h) + 2) * 5 + (tx - (tw + cb_w) + 2);
            mine = (masks[th * q.tiles_w + tw] >> bit) & 1u;
        }
    }
    const unsigned todo = (unsigned)__ballot(mine);
    if (!todo) continue;
    __syncthreads();                                                  // the previous item's image has been flushed
    for (int i = tid; i < GIN_TP * GC; i += 256) accL[i] = 0.f;
    for (int s = 0; s < 25; ++s) {                                    // ascending (th, tw): the fixed order of the sums
        if (!((todo >> s) & 1u)) continue;
        const int th = ty + s / 5 - 2 - cb_h, tw = tx + s % 5 - 2 - cb_w;
        const int th0 = th * GIN_TH, tw0 = tw * GIN_TW;
        const int win_h0 = th0 * a.sh + q.lo_h, win_w0 = tw0 * a.sw + q.lo_w;
        for (int i0 = 0; i0 < nrec; i0 += 256) {
            const int i = i0 + tid;
            int key[4];
            float cf[4];
            int cnt = 0;
            if (i < nrec) {
                const int pl = i / a.K, k = i % a.K;
                const int ho = th0 + pl / GIN_TW, wo = tw0 + pl % GIN_TW;
                if (ho < a.Ho && wo < a.Wo) {
                    const long pix = ((long)n * a.Ho + ho) * a.Wo + wo;
                    const float2 ofs = *reinterpret_cast<const float2 *>(a.offset + a.off_at(pix, g * a.K + k));
                    const float m = a.mask[a.msk_at(pix, g * a.K + k)];
                    const int ii = k / a.kh, jj = k % a.kh;
                    const float p0w = (float)(half_w - a.pw + wo * a.sw) - (float)half_w * a.offset_scale;
                    const float p0h = (float)(half_h - a.ph + ho * a.sh) - (float)half_h * a.offset_scale;
                    const float loc_w = p0w + ((float)(ii * a.dw) + ofs.x) * a.offset_scale;
                    const float loc_h = p0h + ((float)(jj * a.dh) + ofs.y) * a.offset_scale;
                    if (loc_h > -1.f && loc_w > -1.f && loc_h < (float)a.H && loc_w < (float)a.W) {
                        const float fh = floorf(loc_h), fw = floorf(loc_w);
                        const int h0 = (int)fh, w0 = (int)fw;
                        const float lh = loc_h - fh, lw = loc_w - fw, hh = 1.f - lh, hw = 1.f - lw;
                        const bool h0ok = h0 >= 0, h1ok = h0 + 1 <= a.H - 1, w0ok = w0 >= 0, w1ok = w0 + 1 <= a.W - 1;
                        const float c4[4] = {(h0ok && w0ok) ? hh * hw * m : 0.f, (h0ok && w1ok) ? hh * lw * m : 0.f,
                                             (h1ok && w0ok) ? lh * hw * m : 0.f, (h1ok && w1ok) ? lh * lw * m : 0.f};
#pragma unroll
                        for (int tp = 0; tp < 4; ++tp) {
                            const int h = h0 + (tp >> 1), w = w0 + (tp & 1);
                            const bool in_win = (unsigned)(h - win_h0) < (unsigned)q.WH && (unsigned)(w - win_w0) < (unsigned)q.WW;
                            if (c4[tp] != 0.f && !in_win && (h >> 3) == ty && (w >> 3) == tx) {
                                key[cnt] = (((h & 7) * 8 + (w & 7)) << 8) | pl;
                                cf[cnt] = c4[tp];
                                ++cnt;
                            }
                        }
                    }
                }
            }
            // exclusive scan of cnt over the block -> slots in (thread, corner) order
            int inc = cnt;
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
            __syncthreads();                                          // the previous round's entries have been consumed
            if (lane == 63) wsum[wave] = inc;
            __syncthreads();
            int base = inc - cnt;
            for (int w = 0; w < wave; ++w) base += wsum[w];
            const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
            for (int e = 0; e < cnt; ++e) { ent_key[base + e] = key[e]; ent_cf[base + e] = cf[e]; }
            __syncthreads();
            for (int e = 0; e < total; ++e) {                         // one entry after the other: the order of the adds is the list's
                const int ky = ent_key[e], cell = ky >> 8, pl = ky & 255;
                if (cell % NG != grp) continue;
                const int ho = th0 + pl / GIN_TW, wo = tw0 + pl % GIN_TW;
                accL[cell * GC + c] += a.grad_output[(((long)n * a.Ho + ho) * a.Wo + wo) * a.C + g * GC + c] * ent_cf[e];
            }
        }
    }
    __syncthreads();
    for (int cell = grp; cell < GIN_TP; cell += NG) {
        const int h = ty * 8 + (cell >> 3), w = tx * 8 + (cell & 7);
        if (h < a.H && w < a.W) {
            float *o = a.grad_input + (((long)n * a.H + h) * a.W + w) * a.C + g * GC + c;
            *o += accL[cell * GC + c];
        }
    }
    }
}

// ------------------------------------------------------------------------------------------------ windowed gathers: forward, backward A
// The tiled kernels above fetch every bilinear tap from L2: 4*K taps of 4*Gc bytes per (pixel, group), 59 M cache-line requests at
// N32 80x80 C256 - they run at the L2 request rate (1.4-1.7 TB/s of algorithmic bytes), although neighbouring pixels sample the same
// input pixels ~4*K times over.  Here a workgroup owns the 8x8 output tile of ONE group (the geometry of backward B) and first copies
// the input window around it - tile + kernel reach + R pixels of offset slack, zero outside the image - into LDS with coalesced
// 4*Gc-byte runs (225 runs instead of 2304 tap fetches at K = 9, R = 2; 4.8x fewer L2 requests measured).  The taps then come from
// LDS as 16-byte reads, Gc/4 lanes per pixel.  With the memory system out of the way the kernel is bound by its vector instructions
// (PMC: 900 per wave, 144 of them the multiply-adds), so everything per lane is 32-bit off uniform bases, divisions are reciprocal
// multiplies, and a record carries the ready LDS offset of its footprint: an unused point aims at a pad of zero cells behind the
// window, so the loop over the points has no selects.  A pixel with a footprint beyond the window (flag bit in the record) is redone
// with the tiled kernel's gather from global memory.
// MODE 0: output.  MODE 1: grad_offset / grad_mask (butterfly over the Gc/4 lanes of a pixel, no atomics).
// LDS (dynamic): win [WH*WW + WW + 2][GC] floats | rf [64*K] float4 | code [64*K] ints - 42 KB at K = 9, Gc = 32, R = 2.
constexpr int WIN_FAR = 1 << 30;        // record flag: footprint leaves the window
constexpr int WIN_OFF = WIN_FAR - 1;
constexpr int WIN_PF = 3;               // sampling points whose offset / mask loads are issued ahead of the window loads

// sum over an aligned group of LG (2, 4, 8, 16) neighbouring lanes, the same value in every lane: DPP row operations on the VALU
// (__shfl_xor compiles to ds_bpermute_b32, i.e. LDS traffic: 9 per sampling point in backward A)
template <int CTRL>
__device__ __forceinline__ float dpp_get(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int LG>
__device__ __forceinline__ float group_sum(float v) {
    if constexpr (LG >= 2) v += dpp_get<0xB1>(v);                 // quad_perm [1,0,3,2]
    if constexpr (LG >= 4) v += dpp_get<0x4E>(v);                 // quad_perm [2,3,0,1]
    if constexpr (LG >= 8) v += dpp_get<0x141>(v);                // row_half_mirror: the other quad of the 8
    if constexpr (LG >= 16) v += dpp_get<0x140>(v);               // row_mirror: the other half of the 16
    return v;
}

template <int GC, int MODE>
__global__ __launch_bounds__(256) void dcnv3_win_kernel(const DcnArgs a, const GinGeo q) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LG = GC / 4, PPW = 256 / LG;                      // lanes per pixel, pixels per pass of the block
    const int ncell = q.WH * q.WW, nrec = GIN_TP * a.K, ntile = q.tiles_h * q.tiles_w;
    float *win = reinterpret_cast<float *>(smem);
    f32x4 *rf = reinterpret_cast<f32x4 *>(win + (size_t)(ncell + q.WW + 2) * GC);
    int *code = reinterpret_cast<int *>(rf + nrec);
    float *ob = reinterpret_cast<float *>(code + nrec);             // MODE 1: the tile's [pixel][K] mask and [pixel][2K] offset gradients, stored together at the end
    // (image, tile, group) with the group fastest: the G workgroups of a tile read neighbouring bytes, and an XCD's L2 sees one
    // contiguous run of tiles (their halos overlap)
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int g = id % a.G, tile = (id / a.G) % ntile, n = id / (a.G * ntile);
    const int th0 = (tile / q.tiles_w) * GIN_TH, tw0 = (tile % q.tiles_w) * GIN_TW;
    const int win_h0 = th0 * a.sh + q.lo_h, win_w0 = tw0 * a.sw + q.lo_w;
    // uniform 64-bit bases; everything per lane is 32-bit from here (win_plan checks the extents)
    const float *img = a.input + (long)n * a.H * a.W * a.C + g * GC;
    const long opix0 = (long)n * a.Ho * a.Wo;
    const float *ofs_b = a.offset + opix0 * a.off_ps + g * a.K * 2;
    const float *msk_b = a.mask + opix0 * a.msk_ps + g * a.K;
    const int ops = (int)a.off_ps, mps = (int)a.msk_ps;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    // 1. loads first, all in flight together.  A thread builds the records of one (pixel, kernel column): kernel_h consecutive points
    //    (offsets and masks contiguous), the pixel decode and the base position shared.  Their first WIN_PF loads go out here ...
    const int ncol = GIN_TP * a.kw;
    float2 ofs_r[WIN_PF];
    float m_r[WIN_PF];
    {
        const int t = threadIdx.x;
        const int pl = (int)(((unsigned)t * q.rkw) >> 16), ii = t - pl * a.kw;
        const int ho = th0 + pl / GIN_TW, wo = tw0 + pl % GIN_TW;
        const bool on = t < ncol && ho < a.Ho && wo < a.Wo;
        const int so = (ho * a.Wo + wo) * ops + ii * a.kh * 2, sm = (ho * a.Wo + wo) * mps + ii * a.kh;
#pragma unroll
        for (int j = 0; j < WIN_PF; ++j) {
            ofs_r[j] = make_float2(0.f, 0.f);
            m_r[j] = 0.f;
            if (on && j < a.kh) {
                ofs_r[j] = *reinterpret_cast<const float2 *>(ofs_b + so + j * 2);
                m_r[j] = msk_b[sm + j];
            }
        }
    }
    //    ... then the window in batches of four 16-byte runs
    const int nrun = ncell * LG;
    for (int i0 = threadIdx.x; i0 < nrun; i0 += 4 * 256) {
        f32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = i0 + j * 256;
            const int cell = i / LG, part = i % LG;
            const int hr = (int)(((unsigned)cell * q.rWW) >> 16);
            const int h = win_h0 + hr, w = win_w0 + (cell - hr * q.WW);
            v[j] = zero;
            if (i < nrun && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W)
                v[j] = *reinterpret_cast<const f32x4 *>(img + (h * a.W + w) * a.C + part * 4);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = i0 + j * 256;
            if (i < nrun) *reinterpret_cast<f32x4 *>(win + (size_t)i * 4) = v[j];      // (cell, part) -> cell*GC + part*4 == i*4
        }
    }
    for (int i = threadIdx.x; i < (q.WW + 2) * LG; i += 256) *reinterpret_cast<f32x4 *>(win + (size_t)(nrun + i) * 4) = zero;   // the zero pad
    // 2. the sampling records: byte offset of the footprint's first cell in `win` (the zero pad for an unused point and, flagged
    //    WIN_FAR, for a footprint beyond the window) and, MODE 0, the four tap coefficients x mask / MODE 1, {lh, lw, mask, -} -
    //    the arithmetic of make_record
    const int pad_off = ncell * GC * 4;
    const int half_w = (a.dw * (a.kw - 1)) >> 1, half_h = (a.dh * (a.kh - 1)) >> 1;
    for (int t = threadIdx.x; t < ncol; t += 256) {
        const int pl = (int)(((unsigned)t * q.rkw) >> 16), ii = t - pl * a.kw;
        const int ho = th0 + pl / GIN_TW, wo = tw0 + pl % GIN_TW;
        const bool on = ho < a.Ho && wo < a.Wo;
        const int so = (ho * a.Wo + wo) * ops + ii * a.kh * 2, sm = (ho * a.Wo + wo) * mps + ii * a.kh;
        const float p0w = (float)(half_w - a.pw + wo * a.sw) - (float)half_w * a.offset_scale;
        const float p0h = (float)(half_h - a.ph + ho * a.sh) - (float)half_h * a.offset_scale;
        const float iw = (float)(ii * a.dw);
        const int r0 = pl * a.K + ii * a.kh;
        for (int jj = 0; jj < a.kh; ++jj) {
            int cd = pad_off;
            f32x4 f = zero;
            if (on) {
                float2 ofs;
                float m;
                if (t < 256 && jj < WIN_PF) {
                    ofs = jj == 0 ? ofs_r[0] : (jj == 1 ? ofs_r[1] : ofs_r[2]);
                    m = jj == 0 ? m_r[0] : (jj == 1 ? m_r[1] : m_r[2]);
                } else {
                    ofs = *reinterpret_cast<const float2 *>(ofs_b + so + jj * 2);
                    m = msk_b[sm + jj];
                }
                const float loc_w = p0w + (iw + ofs.x) * a.offset_scale;
                const float loc_h = p0h + ((float)(jj * a.dh) + ofs.y) * a.offset_scale;
                const float fh = floorf(loc_h), fw = floorf(loc_w);
                const float lh = loc_h - fh, lw = loc_w - fw;
                if (loc_h > -1.f && loc_w > -1.f && loc_h < (float)a.H && loc_w < (float)a.W) {
                    const int wh = (int)fh - win_h0, ww = (int)fw - win_w0;
                    const bool inside = wh >= 0 && wh + 1 < q.WH && ww >= 0 && ww + 1 < q.WW;
                    cd = inside ? (wh * q.WW + ww) * (GC * 4) : (pad_off | WIN_FAR);
                    if constexpr (MODE == 0) {                       // taps outside the image read zeros of the window: no validity tests
                        const float hh = 1.f - lh, hw = 1.f - lw;
                        f[0] = hh * hw * m;
                        f[1] = hh * lw * m;
                        f[2] = lh * hw * m;
                        f[3] = lh * lw * m;
                    }
                }
                if constexpr (MODE == 1) { f[0] = lh; f[1] = lw; f[2] = m; }
            }
            code[r0 + jj] = cd;
            rf[r0 + jj] = f;
        }
    }
    __syncthreads();
    // 3. Gc/4 lanes per pixel, three points = 12 LDS reads in flight
    const int part = threadIdx.x % LG;
    const int rowB = q.WW * GC * 4;
    const char *wbase = reinterpret_cast<const char *>(win + part * 4);
    for (int p0 = 0; p0 < GIN_TP; p0 += PPW) {
        const int pl = p0 + threadIdx.x / LG;
        const int ho = th0 + pl / GIN_TW, wo = tw0 + pl % GIN_TW;
        const bool live = pl < GIN_TP && ho < a.Ho && wo < a.Wo;
        if (MODE == 0 && !live) continue;
        const int plc = live ? pl : 0;
        const int lpix = live ? ho * a.Wo + wo : 0;                  // pixel inside the image
        const long pix = opix0 + lpix;
        const float *src = img + part * 4;
        const int *cr = code + plc * a.K;
        const f32x4 *fr = rf + plc * a.K;
        if constexpr (MODE == 0) {
            f32x4 acc = zero;
            int flags = 0;
            int k = 0;
            for (; k + 3 <= a.K; k += 3) {
                int cd[3];
                f32x4 f[3], u[12];
#pragma unroll
                for (int j = 0; j < 3; ++j) { cd[j] = cr[k + j]; f[j] = fr[k + j]; }
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    flags |= cd[j];
                    const char *b = wbase + (cd[j] & WIN_OFF);
                    u[j * 4] = *reinterpret_cast<const f32x4 *>(b);
                    u[j * 4 + 1] = *reinterpret_cast<const f32x4 *>(b + GC * 4);
                    u[j * 4 + 2] = *reinterpret_cast<const f32x4 *>(b + rowB);
                    u[j * 4 + 3] = *reinterpret_cast<const f32x4 *>(b + rowB + GC * 4);
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) acc += f[j][0] * u[j * 4] + f[j][1] * u[j * 4 + 1] + f[j][2] * u[j * 4 + 2] + f[j][3] * u[j * 4 + 3];
            }
            for (; k < a.K; ++k) {
                const int cd = cr[k];
                flags |= cd;
                const f32x4 f = fr[k];
                const char *b = wbase + (cd & WIN_OFF);
                acc += f[0] * *reinterpret_cast<const f32x4 *>(b) + f[1] * *reinterpret_cast<const f32x4 *>(b + GC * 4) +
                       f[2] * *reinterpret_cast<const f32x4 *>(b + rowB) + f[3] * *reinterpret_cast<const f32x4 *>(b + rowB + GC * 4);
            }
            if (flags & WIN_FAR) {                                 // rare: the far points (they added zeros above) by the tiled kernel's gather
                for (k = 0; k < a.K; ++k) {
                    if (!(cr[k] & WIN_FAR)) continue;
                    const Rec r = make_record<false>(a, pix, g, k);
                    acc += r.f[0] * *reinterpret_cast<const f32x4 *>(src + r.off[0]) + r.f[1] * *reinterpret_cast<const f32x4 *>(src + r.off[1]) +
                           r.f[2] * *reinterpret_cast<const f32x4 *>(src + r.off[2]) + r.f[3] * *reinterpret_cast<const f32x4 *>(src + r.off[3]);
                }
            }
            *reinterpret_cast<f32x4 *>(a.output + opix0 * a.C + g * GC + (lpix * a.C + part * 4)) = acc;
        } else {
            const f32x4 tg = live ? *reinterpret_cast<const f32x4 *>(a.grad_output + opix0 * a.C + g * GC + (lpix * a.C + part * 4)) : zero;
            for (int k0 = 0; k0 < a.K; k0 += 3) {
                const int nk = min(3, a.K - k0);
                int cd[3];
                f32x4 f[3], u[12];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int kk = k0 + (j < nk ? j : 0);
                    cd[j] = live ? cr[kk] : pad_off;
                    f[j] = fr[kk];
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const char *b = wbase + (cd[j] & WIN_OFF);
                    u[j * 4] = *reinterpret_cast<const f32x4 *>(b);
                    u[j * 4 + 1] = *reinterpret_cast<const f32x4 *>(b + GC * 4);
                    u[j * 4 + 2] = *reinterpret_cast<const f32x4 *>(b + rowB);
                    u[j * 4 + 3] = *reinterpret_cast<const f32x4 *>(b + rowB + GC * 4);
                }
                if ((cd[0] | cd[1] | cd[2]) & WIN_FAR) {              // rare: the far points by the tiled kernel's gather
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        if (!(cd[j] & WIN_FAR) || j >= nk) continue;
                        const Rec r = make_record<true>(a, pix, g, k0 + j);
                        const int bits = __float_as_int(r.f[3]);
#pragma unroll
                        for (int t = 0; t < 4; ++t) u[j * 4 + t] = (bits >> t) & 1 ? *reinterpret_cast<const f32x4 *>(src + r.off[t]) : zero;
                    }
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if (j >= nk) break;                                  // wave-uniform
                    const f32x4 v1 = u[j * 4], v2 = u[j * 4 + 1], v3 = u[j * 4 + 2], v4 = u[j * 4 + 3];
                    const float lh = f[j][0], lw = f[j][1], m = f[j][2], hh = 1.f - lh, hw = 1.f - lw;
                    // dcnv3_col2im_bilinear: value, grad_h_weight, grad_w_weight (dcnv3_im2col_cuda.cuh:112-141) with the sums over the
                    // channels taken FIRST: the bilinear coefficients are the same for every channel of the group, so
                    //   sum_c tg * val = hh (hw d1 + lw d2) + lh (hw d3 + lw d4),  sum_c tg * ghw = (hw d3 + lw d4) - (hw d1 + lw d2),
                    //   sum_c tg * gww = hh (d2 - d1) + lh (d4 - d3)            with d_i = sum_c tg[c] * v_i[c] over this lane's 4 channels
                    // - 16 multiply-adds and a dozen scalar operations per point instead of three blended 4-vectors and their dot products
                    // (round 3: 1.9e8 vector instructions per launch, 2.5x the forward's, for the same LDS reads)
                    const float d1 = (tg[0] * v1[0] + tg[1] * v1[1]) + (tg[2] * v1[2] + tg[3] * v1[3]);
                    const float d2 = (tg[0] * v2[0] + tg[1] * v2[1]) + (tg[2] * v2[2] + tg[3] * v2[3]);
                    const float d3 = (tg[0] * v3[0] + tg[1] * v3[1]) + (tg[2] * v3[2] + tg[3] * v3[3]);
                    const float d4 = (tg[0] * v4[0] + tg[1] * v4[1]) + (tg[2] * v4[2] + tg[3] * v4[3]);
                    const float top = hw * d1 + lw * d2, bot = hw * d3 + lw * d4;
                    const float gm = group_sum<LG>(hh * top + lh * bot);
                    const float gw = group_sum<LG>((hh * (d2 - d1) + lh * (d4 - d3)) * m);
                    const float gh = group_sum<LG>((bot - top) * m);
                    if (live && part == 0) {                              // staged: one lane of eight would otherwise issue 27 four- and eight-byte stores per pixel
                        ob[plc * a.K + k0 + j] = gm;
                        *reinterpret_cast<float2 *>(ob + GIN_TP * a.K + (plc * a.K + k0 + j) * 2) = make_float2(a.offset_scale * gw, a.offset_scale * gh);
                    }
                }
            }
        }
    }
    if constexpr (MODE == 1) {
        // 4. the staged gradients leave as runs: consecutive lanes write consecutive floats of a pixel's K masks / 2K offsets (36 / 72 contiguous
        //    bytes at K = 9; round 3 wrote them from one lane of eight as they were produced: 3.7e6 store instructions of 8 scattered lanes each
        //    per launch at N32 80x80 - the store path, not the arithmetic, was what held this kernel at 2.5x the forward's time)
        __syncthreads();
        float *gm_t = a.grad_mask + opix0 * a.msk_ps + g * a.K, *go_t = a.grad_offset + opix0 * a.off_ps + g * a.K * 2;
        for (int i = threadIdx.x; i < GIN_TP * a.K; i += 256) {
            const int pl = i / a.K, k = i - pl * a.K;
            const int ho = th0 + pl / GIN_TW, wo = tw0 + pl % GIN_TW;
            if (ho < a.Ho && wo < a.Wo) gm_t[(ho * a.Wo + wo) * mps + k] = ob[i];
        }
        for (int i = threadIdx.x; i < GIN_TP * a.K * 2; i += 256) {
            const int pl = i / (a.K * 2), k2 = i - pl * a.K * 2;
            const int ho = th0 + pl / GIN_TW, wo = tw0 + pl % GIN_TW;
            if (ho < a.Ho && wo < a.Wo) go_t[(ho * a.Wo + wo) * ops + k2] = ob[GIN_TP * a.K + i];
        }
    }
}

template <int MODE>
static void launch_win(const DcnArgs &a, const GinGeo &q, size_t lds, hipStream_t s) {
    const dim3 grid((unsigned)(a.N * a.G * q.tiles_h * q.tiles_w));
    switch (a.Gc) {
    case 8: hipLaunchKernelGGL((dcnv3_win_kernel<8, MODE>), grid, dim3(256), lds, s, a, q); break;
    case 16: hipLaunchKernelGGL((dcnv3_win_kernel<16, MODE>), grid, dim3(256), lds, s, a, q); break;
    case 32: hipLaunchKernelGGL((dcnv3_win_kernel<32, MODE>), grid, dim3(256), lds, s, a, q); break;
    default: hipLaunchKernelGGL((dcnv3_win_kernel<64, MODE>), grid, dim3(256), lds, s, a, q); break;
    }
}

// geometry of the windowed forms (tile + kernel reach + R pixels of offset slack); false for other group widths
static bool win_geo(const DcnArgs &a, GinGeo &q) {
    if (!(a.Gc == 8 || a.Gc == 16 || a.Gc == 32 || a.Gc == 64)) return false;
    static const int slack = [] { const char *e = getenv("SOMI_DCN_SLACK"); const int v = e ? atoi(e) : 2; return v < 0 ? 0 : (v > 8 ? 8 : v); }();
    q.R = slack;
    const int half_h = (a.dh * (a.kh - 1)) >> 1, half_w = (a.dw * (a.kw - 1)) >> 1;
    const int lo_rh = (int)ceilf((float)half_h * a.offset_scale), hi_rh = (int)ceilf((float)((a.kh - 1) * a.dh - half_h) * a.offset_scale);
    const int lo_rw = (int)ceilf((float)half_w * a.offset_scale), hi_rw = (int)ceilf((float)((a.kw - 1) * a.dw - half_w) * a.offset_scale);
    q.lo_h = half_h - a.ph - lo_rh - q.R;
    q.lo_w = half_w - a.pw - lo_rw - q.R;
    q.WH = (GIN_TH - 1) * a.sh + lo_rh + hi_rh + 2 * q.R + 2;
    q.WW = (GIN_TW - 1) * a.sw + lo_rw + hi_rw + 2 * q.R + 2;
    q.tiles_h = (a.Ho + GIN_TH - 1) / GIN_TH;
    q.tiles_w = (a.Wo + GIN_TW - 1) / GIN_TW;
    return true;
}

// the gathering kernels (forward, backward A): LDS bytes, or 0 when the window does not fit / the grid does not index
static size_t win_plan(const DcnArgs &a, GinGeo &q, int mode = 0) {
    if (!win_geo(a, q)) return 0;
    const char *e = getenv("SOMI_DCN_DIRECT");                    // read per call: tools flip it to time the two forms side by side
    if (e && e[0] == '1') return 0;
    const size_t lds = ((size_t)q.WH * q.WW + q.WW + 2) * a.Gc * sizeof(float) + (size_t)GIN_TP * a.K * (sizeof(f32x4) + sizeof(int)) +
                       (mode == 1 ? (size_t)GIN_TP * a.K * 3 * sizeof(float) : 0);      // backward A stages its outputs
    if (lds > 64 * 1024 || (long)a.N * a.G * q.tiles_h * q.tiles_w >= (1L << 31)) return 0;
    if ((long)a.Ho * a.Wo * a.off_ps >= (1L << 31) || (long)a.Ho * a.Wo * a.msk_ps >= (1L << 31) || (long)a.Ho * a.Wo * a.C >= (1L << 31)) return 0;      // 32-bit indices inside one image
    if (GIN_TP * a.kw + 256 >= 65536 / a.kw || q.WH * q.WW >= 65536 / q.WW) return 0;                             // the reciprocal divisions
    q.rkw = 65536u / a.kw + 1;
    q.rWW = 65536u / q.WW + 1;
    return lds;
}

// backward B/C/D; false when it does not apply (group width, LDS budget).  `chunk`: images per pass of B / C / D - the staging slab
// holds one chunk and is reused by the next; it is capped at SOMI_DCN_SLAB_MB (default 1024) purely as a memory bound: running the chunks
// through a slab small enough to stay in the 256 MB Infinity Cache was measured and gains nothing.
// Workspace layout: slab | near masks (one word per (image of the chunk, group, tile)) | near_any (one word per chunk) | 256 B whose
// first word is the far-tap counter - every region a multiple of 256 B, all of it sized HERE, before anything is launched.
struct GinPlan {
    size_t lds, mlds, slab_bytes, near_bytes, any_bytes, workspace_bytes;
    int chunk, nchunks;
    bool mfma, coloured;
    int nch, ncw;
};
static bool gin_plan(const DcnArgs &a, GinGeo &q, GinPlan &pl) {
    if (!win_geo(a, q)) return false;
    pl.lds = (size_t)GIN_TP * a.Gc * sizeof(float) + (size_t)GIN_TP * a.K * (sizeof(RecG) + 4 * sizeof(float) + 4) + GIN_OVF_CAP * sizeof(OvfG) +
             3 * ((size_t)q.WH * q.WW + 1) * sizeof(int);
    if (pl.lds > 150 * 1024 || (long)q.tiles_h * q.tiles_w > 65535L * 32) return false;
    // B on the matrix cores for 32-wide groups when its LDS image fits (SOMI_DCN_GIN=exact keeps the list sums for comparisons) ...
    const char *gsel = getenv("SOMI_DCN_GIN");
    const size_t ncell_ = (size_t)q.WH * q.WW, recb = (size_t)GIN_TP * a.K * sizeof(RecM), gotb = (size_t)a.Gc * GMM_LD * sizeof(float);
    pl.mlds = ncell_ * GMM_LD * sizeof(float) + ((recb > gotb ? recb : gotb) + 15) / 16 * 16 + GMM_OVF_CAP * sizeof(OvfG);
    pl.mfma = a.Gc == 32 && pl.mlds <= 78 * 1024 + 512 && !(gsel && gsel[0] == 'e');
    // ... and, stride 1 with the near pass on, in its COLOURED form: tiles whose windows cannot overlap run together and add straight into
    // grad_input - no staging slab, no combine pass.  Up to 2 x 2 colours (a 3 x 3 kernel); SOMI_DCN_SLAB=1 keeps the slab form (A/B runs).
    static const bool near_on = [] { const char *e = getenv("SOMI_DCN_NEAR"); return !(e && e[0] == '0'); }();
    const char *slab_env = getenv("SOMI_DCN_SLAB");
    pl.nch = (q.WH + GIN_TH * a.sh - 1) / (GIN_TH * a.sh);
    pl.ncw = (q.WW + GIN_TW * a.sw - 1) / (GIN_TW * a.sw);
    pl.coloured = pl.mfma && near_on && a.sh == 1 && a.sw == 1 && pl.nch * pl.ncw <= 4 && a.N <= 65535 && q.WH * q.WW <= GMM_MAX_CELLS &&
                  !(slab_env && slab_env[0] == '1');
    static const long cap_mb = [] { const char *e = getenv("SOMI_DCN_SLAB_MB"); const long v = e ? atol(e) : 1024; return v < 1 ? 1 : v; }();
    const size_t per_img = (size_t)a.G * q.tiles_h * q.tiles_w * q.WH * q.WW * a.Gc * sizeof(float);
    long c = (long)(((size_t)cap_mb << 20) / per_img);
    pl.chunk = pl.coloured ? a.N : (c < 1 ? 1 : (c > a.N ? a.N : (int)c));    // the coloured form has no slab: the whole batch per launch
    pl.nchunks = (a.N + pl.chunk - 1) / pl.chunk;
    pl.slab_bytes = pl.coloured ? 0 : ((size_t)pl.chunk * per_img + 255) / 256 * 256;
    pl.near_bytes = ((size_t)pl.chunk * a.G * q.tiles_h * q.tiles_w * sizeof(unsigned) + 255) / 256 * 256;
    pl.any_bytes = ((size_t)pl.nchunks * sizeof(unsigned) + 255) / 256 * 256;
    pl.workspace_bytes = pl.slab_bytes + pl.near_bytes + pl.any_bytes + 256;
    return true;
}

// the list-form B kernel: more than 64 KB of dynamic LDS needs the attribute, whatever the group width
template <int GC>
static void launch_gin(const dim3 &grid, size_t glds, hipStream_t s, const DcnArgs &c, const GinGeo &q) {
    if (glds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&dcnv3_bwd_gin_kernel<GC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)glds);
    hipLaunchKernelGGL((dcnv3_bwd_gin_kernel<GC>), grid, dim3(256), glds, s, c, q);
}

static int fill_args(DcnArgs &a, int N, int H, int W, int G, int Gc, int kh, int kw, int sh, int sw, int ph, int pw, int dh,
                     int dw, float offset_scale, int im2col_step, size_t rec_bytes) {
    SOMI_REQUIRE(N > 0 && H > 0 && W > 0 && G > 0 && Gc > 0 && kh > 0 && kw > 0 && sh > 0 && sw > 0 && dh > 0 && dw > 0 &&
                     ph >= 0 && pw >= 0, SOMI_EINVAL, "dcnv3: bad geometry");
    const int step = N < im2col_step ? N : im2col_step;
    SOMI_REQUIRE(im2col_step > 0 && N % step == 0, SOMI_EINVAL, "batch(%d) must divide im2col_step(%d)", N, step);
    a.N = N; a.H = H; a.W = W; a.G = G; a.Gc = Gc; a.C = G * Gc;
    a.kh = kh; a.kw = kw; a.K = kh * kw; a.sh = sh; a.sw = sw; a.ph = ph; a.pw = pw; a.dh = dh; a.dw = dw;
    a.Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;
    a.Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
    SOMI_REQUIRE(a.Ho > 0 && a.Wo > 0, SOMI_EINVAL, "dcnv3: empty output");
    SOMI_REQUIRE((long)H * W * a.C < (1L << 31), SOMI_EINVAL, "dcnv3: one image must stay below 2^31 elements");
    a.offset_scale = offset_scale;
    a.npix = (long)N * a.Ho * a.Wo;
    const size_t per_pix = (size_t)G * a.K * rec_bytes;
    SOMI_REQUIRE(per_pix <= 60 * 1024, SOMI_ENOTIMPL, "dcnv3: G*K = %d does not fit one LDS tile", G * a.K);
    int tp = (int)((48 * 1024) / per_pix);
    if (tp < 1) tp = 1;
    if (tp > 16) tp = 16;
    a.TP = tp;
    a.ntile = (int)((a.npix + tp - 1) / tp);
    return 0;
}

}  // namespace somi

using namespace somi;

// pixel strides of offset / mask: 0 = packed rows; otherwise both live inside wider rows (one GEMM's [pixel][>= 3*G*K] output)
static int set_strides(DcnArgs &a, long offset_stride, long mask_stride) {
    const long gk = (long)a.G * a.K;
    a.off_ps = offset_stride ? offset_stride : 2 * gk;
    a.msk_ps = mask_stride ? mask_stride : gk;
    SOMI_REQUIRE(a.off_ps >= 2 * gk && a.msk_ps >= gk && a.off_ps % 2 == 0, SOMI_EINVAL,
                 "dcnv3: offset / mask pixel strides (%ld, %ld) below the row lengths (%ld, %ld) or odd", a.off_ps, a.msk_ps, 2 * gk, gk);
    return 0;
}

extern "C" int somi_dcnv3_forward_strided_f32(const float *input, const float *offset, const float *mask, long offset_stride, long mask_stride,
                                              float *output, int N, int H, int W, int G, int Gc, int kernel_h, int kernel_w, int stride_h,
                                              int stride_w, int pad_h, int pad_w, int dilation_h, int dilation_w, float offset_scale,
                                              int im2col_step, somi_stream_t stream) {
    SOMI_REQUIRE(input && offset && mask && output, SOMI_EINVAL, "dcnv3 forward: null tensor");
    DcnArgs a{};
    int rc = fill_args(a, N, H, W, G, Gc, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w,
                       offset_scale, im2col_step, sizeof(Rec));
    if (rc) return rc;
    if ((rc = set_strides(a, offset_stride, mask_stride))) return rc;
    a.input = input; a.offset = offset; a.mask = mask; a.output = output;
    SOMI_REQUIRE((reinterpret_cast<uintptr_t>(offset) & 7u) == 0, SOMI_EINVAL, "dcnv3: offset must be 8 B aligned");
    const size_t lds = (size_t)a.TP * G * a.K * sizeof(Rec);
    const bool vec = (Gc % 4 == 0) && aligned16(input) && aligned16(output);
    GinGeo q{};
    const size_t wlds = vec ? win_plan(a, q) : 0;
    if (wlds) {
        launch_win<0>(a, q, wlds, (hipStream_t)stream);
        return launch_status("somi_dcnv3_forward_f32 (windowed)");
    }
    if (vec) hipLaunchKernelGGL(dcnv3_fwd_kernel<4>, dim3(a.ntile), dim3(256), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(dcnv3_fwd_kernel<1>, dim3(a.ntile), dim3(256), lds, (hipStream_t)stream, a);
    return launch_status("somi_dcnv3_forward_f32");
}

extern "C" int somi_dcnv3_forward_f32(const float *input, const float *offset, const float *mask, float *output, int N, int H,
                                      int W, int G, int Gc, int kernel_h, int kernel_w, int stride_h, int stride_w, int pad_h,
                                      int pad_w, int dilation_h, int dilation_w, float offset_scale, int im2col_step,
                                      somi_stream_t stream) {
    return somi_dcnv3_forward_strided_f32(input, offset, mask, 0, 0, output, N, H, W, G, Gc, kernel_h, kernel_w, stride_h, stride_w, pad_h,
                                          pad_w, dilation_h, dilation_w, offset_scale, im2col_step, stream);
}

extern "C" size_t somi_dcnv3_backward_workspace_bytes(int N, int H, int W, int G, int Gc, int kernel_h, int kernel_w, int stride_h, int stride_w,
                                                      int pad_h, int pad_w, int dilation_h, int dilation_w, float offset_scale) {
    DcnArgs a{};
    if (fill_args(a, N, H, W, G, Gc, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, offset_scale, N > 0 ? N : 1,
                  sizeof(Rec) + 3 * sizeof(float)))
        return 0;
    GinGeo q{};
    GinPlan pl{};
    return gin_plan(a, q, pl) ? pl.workspace_bytes : 0;
}

extern "C" int somi_dcnv3_backward_strided_f32(const float *input, const float *offset, const float *mask, long offset_stride, long mask_stride,
                                               const float *grad_output, float *grad_input, float *grad_offset, float *grad_mask, int N, int H,
                                               int W, int G, int Gc, int kernel_h, int kernel_w, int stride_h, int stride_w, int pad_h,
                                               int pad_w, int dilation_h, int dilation_w, float offset_scale, int im2col_step,
                                               void *workspace, size_t workspace_bytes, somi_stream_t stream) {
    SOMI_REQUIRE(input && offset && mask && grad_output && grad_input && grad_offset && grad_mask, SOMI_EINVAL,
                 "dcnv3 backward: null tensor");
    DcnArgs a{};
    int rc = fill_args(a, N, H, W, G, Gc, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w,
                       offset_scale, im2col_step, sizeof(Rec) + 3 * sizeof(float));
    if (rc) return rc;
    if ((rc = set_strides(a, offset_stride, mask_stride))) return rc;      // grad_offset / grad_mask use the same pixel strides
    a.input = input; a.offset = offset; a.mask = mask; a.grad_output = grad_output;
    a.grad_input = grad_input; a.grad_offset = grad_offset; a.grad_mask = grad_mask;
    SOMI_REQUIRE((reinterpret_cast<uintptr_t>(offset) & 7u) == 0 && (reinterpret_cast<uintptr_t>(grad_offset) & 7u) == 0,
                 SOMI_EINVAL, "dcnv3: offset / grad_offset must be 8 B aligned");
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = (size_t)a.TP * G * a.K * (sizeof(Rec) + 3 * sizeof(float));
    GinGeo q{};
    GinPlan pl{};
    const bool windowed = workspace && gin_plan(a, q, pl) && workspace_bytes >= pl.workspace_bytes && aligned16(workspace) &&
                          aligned16(grad_input) && aligned16(grad_output);
    if (windowed) {
        const size_t glds = pl.lds, slab = pl.slab_bytes;
        const int chunk = pl.chunk;
        // A: grad_offset / grad_mask (float4 gathers, no atomics), the whole batch
        GinGeo qa{};
        const size_t wlds = aligned16(input) ? win_plan(a, qa, 1) : 0;
        if (wlds) launch_win<1>(a, qa, wlds, s);
        else hipLaunchKernelGGL(dcnv3_bwd_om_kernel, dim3(a.ntile), dim3(256), (size_t)a.TP * G * a.K * sizeof(Rec), s, a);
        // B: grad_input windows, C: combine, D: the near taps - chunk of images after chunk through one staging slab
        char *wsb = static_cast<char *>(workspace);
        q.staging = reinterpret_cast<float *>(wsb);
        unsigned *const any_words = reinterpret_cast<unsigned *>(wsb + slab + pl.near_bytes);          // one near_any word per chunk
        q.overflow = reinterpret_cast<unsigned *>(wsb + pl.workspace_bytes - 256);
        // the near pass needs stride 1 (source tile t then sits over destination tile t + const) and 8-aligned tiles of the INPUT image
        static const bool near_on = [] { const char *e = getenv("SOMI_DCN_NEAR"); return !(e && e[0] == '0'); }();
        q.near = (near_on && stride_h == 1 && stride_w == 1) ? reinterpret_cast<unsigned *>(wsb + slab) : nullptr;
        (void)hipMemsetAsync(any_words, 0, pl.any_bytes + 256, s);    // the per-chunk near_any words and the far-tap counter behind them
        const size_t mlds = pl.mlds;
        const bool mfma = pl.mfma;
        if (pl.coloured) {
            // B coloured: nch x ncw launches of disjoint windows adding straight into grad_input, then the far taps (atomics, a no-op unless a
            // tile flagged some), then D (the near taps) - no slab, no combine, the whole batch per launch
            q.staging = nullptr;
            q.near_any = any_words;
            q.nch = pl.nch;
            q.ncw = pl.ncw;
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&dcnv3_bwd_gin_mfma_kernel<32, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)mlds);
            for (int chh = 0; chh < pl.nch; ++chh)
                for (int cww = 0; cww < pl.ncw; ++cww) {
                    const int th = (q.tiles_h - chh + pl.nch - 1) / pl.nch, tw = (q.tiles_w - cww + pl.ncw - 1) / pl.ncw;
                    if (th <= 0 || tw <= 0) continue;
                    q.col_h = chh;
                    q.col_w = cww;
                    hipLaunchKernelGGL((dcnv3_bwd_gin_mfma_kernel<32, true>), dim3(th * tw, N, G), dim3(GMM_NT), mlds, s, a, q);
                }
            const long items = (long)q.tiles_h * q.tiles_w * N * G, ditems = (long)((H + 7) / 8) * ((W + 7) / 8) * N * G;
            hipLaunchKernelGGL((dcnv3_bwd_far_kernel<32>), dim3((unsigned)(items > 4096 ? 4096 : items)), dim3(256), 0, s, a, q);
            hipLaunchKernelGGL((dcnv3_bwd_near_kernel<32>), dim3((unsigned)(ditems > 4096 ? 4096 : ditems)), dim3(256), 0, s, a, q);
            return launch_status("somi_dcnv3_backward_f32 (windowed, coloured)");
        }
#define SOMI_GMM_LAUNCH(GC)                                                                                                      \
    do {                                                                                                                         \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&dcnv3_bwd_gin_mfma_kernel<GC, false>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  (int)mlds);                                                                               \
        hipLaunchKernelGGL((dcnv3_bwd_gin_mfma_kernel<GC, false>), grid, dim3(GMM_NT), mlds, s, c, q);                             \
    } while (0)
        const size_t in_img = (size_t)H * W * a.C, out_img = (size_t)a.Ho * a.Wo * a.C, opix_img = (size_t)a.Ho * a.Wo;
        for (int n0 = 0; n0 < N; n0 += chunk) {
            DcnArgs c = a;                                         // the chunk as a batch of its own
            c.N = N - n0 < chunk ? N - n0 : chunk;
            c.input = a.input + n0 * in_img;
            c.grad_input = a.grad_input + n0 * in_img;
            c.grad_output = a.grad_output + n0 * out_img;
            c.offset = a.offset + n0 * opix_img * a.off_ps;
            c.mask = a.mask + n0 * opix_img * a.msk_ps;
            c.npix = (long)c.N * a.Ho * a.Wo;
            const dim3 grid(q.tiles_h * q.tiles_w, c.N, G);
            long ditems = (long)((H + 7) / 8) * ((W + 7) / 8) * c.N * G;
            const dim3 dgrid((unsigned)(ditems > 4096 ? 4096 : ditems));   // persistent: exits at once unless a tile of the chunk flagged near taps
            q.near_any = any_words + n0 / chunk;
            long blocks = ((long)c.N * H * W * (a.C / 4) + 255) / 256;
            const dim3 cgrid((unsigned)(blocks > 16384 ? 16384 : blocks));
            // D follows C (it adds to pixels C has just combined); the list-form macro launches B only, C and D come below
            if (mfma) SOMI_GMM_LAUNCH(32);
            else if (Gc == 8) launch_gin<8>(grid, glds, s, c, q);
            else if (Gc == 16) launch_gin<16>(grid, glds, s, c, q);
            else if (Gc == 32) launch_gin<32>(grid, glds, s, c, q);
            else launch_gin<64>(grid, glds, s, c, q);
            hipLaunchKernelGGL(dcnv3_bwd_combine_kernel, cgrid, dim3(256), 0, s, c, q);
            if (q.near) {
                switch (Gc) {
                case 8: hipLaunchKernelGGL((dcnv3_bwd_near_kernel<8>), dgrid, dim3(256), 0, s, c, q); break;
                case 16: hipLaunchKernelGGL((dcnv3_bwd_near_kernel<16>), dgrid, dim3(256), 0, s, c, q); break;
                case 32: hipLaunchKernelGGL((dcnv3_bwd_near_kernel<32>), dgrid, dim3(256), 0, s, c, q); break;
                default: hipLaunchKernelGGL((dcnv3_bwd_near_kernel<64>), dgrid, dim3(256), 0, s, c, q); break;
                }
            }
        }
#undef SOMI_GMM_LAUNCH
        return launch_status("somi_dcnv3_backward_f32 (windowed)");
    }
    // direct form.  One channel per lane: a wave's fp32 atomic covers 256 contiguous bytes of grad_input (the shape that runs at the
    // chip-wide atomic rate, MI355X_MICROARCH.md "Global float atomics"); 4 channels per lane (16 B stride between lanes'
    // dwords) measured 4x slower.  The float4 form is kept for group widths where it avoids the LDS-atomic fallback.
    const bool vec = (Gc % 4 == 0) && aligned16(grad_output) && !((Gc & (Gc - 1)) == 0 && Gc <= 64);
    if (vec) hipLaunchKernelGGL((dcnv3_bwd_kernel<4, true>), dim3(a.ntile), dim3(256), lds, s, a);
    else hipLaunchKernelGGL((dcnv3_bwd_kernel<1, true>), dim3(a.ntile), dim3(256), lds, s, a);
    return launch_status("somi_dcnv3_backward_f32");
}

extern "C" int somi_dcnv3_backward_f32(const float *input, const float *offset, const float *mask, const float *grad_output,
                                       float *grad_input, float *grad_offset, float *grad_mask, int N, int H, int W, int G,
                                       int Gc, int kernel_h, int kernel_w, int stride_h, int stride_w, int pad_h, int pad_w,
                                       int dilation_h, int dilation_w, float offset_scale, int im2col_step, void *workspace,
                                       size_t workspace_bytes, somi_stream_t stream) {
    return somi_dcnv3_backward_strided_f32(input, offset, mask, 0, 0, grad_output, grad_input, grad_offset, grad_mask, N, H, W, G, Gc, kernel_h,
                                           kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, offset_scale, im2col_step,
                                           workspace, workspace_bytes, stream);
}
