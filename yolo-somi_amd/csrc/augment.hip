// Input pipeline on gfx950: mosaic composition + affine crop + mixup + HSV jitter + flips + BGR->RGB planes, one pass.
// (utils/datasets.py:590-673, 732-798; utils/augmentations.py:47-61, 126-169, 305-310; pixel arithmetic of OpenCV 4.9.0:
// warpAffine INTER_LINEAR / BORDER_CONSTANT fixed point, RGB2HSV_b, HSV2RGB_b.)
//
// Every output pixel is independent, so one thread owns four consecutive pixels of a row of the s x s crop and walks the
// whole chain in registers: 4 (or 8, with mixup) canvas taps per pixel, fetched as two unaligned 8-byte loads per bilinear
// footprint -> fixed-point blend -> fp64 mixup -> integer HSV -> table lookups -> float HSV2BGR -> three 32-bit stores at
// the flipped position (one per colour plane).  The canvas is virtual: a tap resolves against the (at most
// four) placed rectangles of the mosaic and reads the cached source image directly, so the 2s x 2s canvas the reference
// fills and then crops is never written.  HBM traffic = the source pixels under the crop (read once, neighbouring lanes
// share cache lines) + 3 bytes per output pixel.  The sample record is uniform per workgroup (scalar loads); the two
// reciprocal tables of the HSV conversion and the three jitter tables sit in LDS.
//
// Bit-exactness: the results have to equal step-by-step CPU arithmetic, so floating-point contraction is switched off
// for this file (hipcc's default would fuse a*r + b*(1-r) into an fma and move a mixup result across an integer boundary).
// Arithmetic is written with plain operators: the *_rn intrinsics inline library code that still allows contraction.
#include "common.h"
#include <stdlib.h>

#pragma clang fp contract(off)

static_assert(sizeof(somi_aug_source) == 40 && sizeof(somi_aug_canvas) == 224 && sizeof(somi_aug_sample) == 1240,
              "somi_aug_* layout is part of the ABI (ctypes mirror in somi_amd/_lib.py)");

namespace somi {

// RGB2HSV_b reciprocal tables, built at compile time: sdiv[i] = round((255 << 12) / i), hdiv[i] = round((180 << 12) / (6 i))
// (round half to even, as cvRound does; the quotients are positive).
struct HsvTables { int sdiv[256], hdiv[256]; };
constexpr int round_half_even(double v) {
    const long long f = (long long)v;
    const double frac = v - (double)f;
    return (int)(frac > 0.5 ? f + 1 : (frac < 0.5 ? f : f + (f & 1)));
}
constexpr HsvTables make_hsv_tables() {
    HsvTables t{};
    for (int i = 1; i < 256; ++i) {
        t.sdiv[i] = round_half_even(1044480.0 / (double)i);
        t.hdiv[i] = round_half_even(737280.0 / (6.0 * (double)i));
    }
    return t;
}
__device__ const HsvTables kHsvTables = make_hsv_tables();
static_assert(make_hsv_tables().sdiv[255] == 4096 && make_hsv_tables().sdiv[3] == 348160 && make_hsv_tables().hdiv[1] == 122880 &&
              make_hsv_tables().hdiv[7] == 17554 && make_hsv_tables().sdiv[0] == 0, "HSV reciprocal tables");

struct Bgr { int b, g, r; };

// pointers read out of the sample record carry no address space: say that they are global memory (not flat)
typedef const __attribute__((address_space(1))) uint8_t *global_u8;
typedef const __attribute__((address_space(1), aligned(1))) uint64_t *global_unaligned_u64;

// The sample record, re-packed once per workgroup into LDS.  Reading its ~120 fields straight from the record costs a chain
// of dependent scalar loads per tap (measured: the kernel was bound by exactly that latency); from LDS the four rectangles
// are four 16-byte reads held in registers for all pixels of the thread, and the matching source is one indexed read.
struct CanvasLds {
    int4 rect[4];            // x1, y1, width, height, clipped to the canvas (zero size for unused slots)
    long long base[4];       // pixels - (dy * w + dx) * 3: address of canvas position (0, 0) in the source's own pitch
    int pitch[4];            // source width in pixels
    double minv[6];
    int warp, pad_;
};

struct Rects { int4 r[4]; };         // x1, y1, width, height - clipped to the canvas when staged, so a hit is always readable

__device__ __forceinline__ global_u8 source_address(const CanvasLds &L, int hit, int x, int y) {
    return (global_u8)(L.base[hit] + (long long)((y * L.pitch[hit] + x) * 3));      // |offset| < 2^31: sides <= 16384
}

__device__ __forceinline__ Bgr load_tap(const CanvasLds &L, int hit, int x, int y, int fill) {
    if (hit < 0) return Bgr{fill, fill, fill};
    const global_u8 p = source_address(L, hit, x, y);
    return Bgr{p[0], p[1], p[2]};
}

__device__ __forceinline__ Bgr canvas_tap(const CanvasLds &L, const Rects &R, int x, int y, int fill) {
    int hit = -1;
#pragma unroll
    for (int i = 0; i < 4; ++i)                       // the LAST matching source wins
        if ((unsigned)(x - R.r[i].x) < (unsigned)R.r[i].z && (unsigned)(y - R.r[i].y) < (unsigned)R.r[i].w) hit = i;
    return load_tap(L, hit, x, y, fill);
}

// The 2x2 bilinear footprint (x..x+1, y..y+1), resolved against the four rectangles in ONE pass (the compares are what
// this kernel spends its time on).  A row whose two taps fall into the same source, with a pixel x + 2 in that row as
// well, is fetched as one unaligned 8-byte load (the two spare bytes are that next pixel's, so nothing outside the row is
// read); rectangle edges and empty canvas take the per-tap path with the hits already known.
__device__ __forceinline__ void canvas_footprint(const CanvasLds &L, const Rects &R, int x, int y, int fill, Bgr (&t)[2][2]) {
    int hit[2][2] = {{-1, -1}, {-1, -1}};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned w = (unsigned)R.r[i].z, h = (unsigned)R.r[i].w;
        const bool c0 = (unsigned)(x - R.r[i].x) < w, c1 = (unsigned)(x + 1 - R.r[i].x) < w;
        const bool r0 = (unsigned)(y - R.r[i].y) < h, r1 = (unsigned)(y + 1 - R.r[i].y) < h;
        hit[0][0] = r0 && c0 ? i : hit[0][0]; hit[0][1] = r0 && c1 ? i : hit[0][1];
        hit[1][0] = r1 && c0 ? i : hit[1][0]; hit[1][1] = r1 && c1 ? i : hit[1][1];
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int h0 = hit[r][0];
        bool fast = h0 >= 0 && h0 == hit[r][1];
        if (fast) {
            int qx = R.r[0].x, qw = R.r[0].z;
            qx = h0 == 1 ? R.r[1].x : qx; qw = h0 == 1 ? R.r[1].z : qw;
            qx = h0 == 2 ? R.r[2].x : qx; qw = h0 == 2 ? R.r[2].z : qw;
            qx = h0 == 3 ? R.r[3].x : qx; qw = h0 == 3 ? R.r[3].z : qw;
            fast = (unsigned)(x + 2 - qx) < (unsigned)qw;
        }
        if (fast) {
            const uint64_t v = *(global_unaligned_u64)source_address(L, h0, x, y + r);
            t[r][0] = Bgr{(int)(v & 255), (int)((v >> 8) & 255), (int)((v >> 16) & 255)};
            t[r][1] = Bgr{(int)((v >> 24) & 255), (int)((v >> 32) & 255), (int)((v >> 40) & 255)};
        } else {
            t[r][0] = load_tap(L, h0, x, y + r, fill);
            t[r][1] = load_tap(L, hit[r][1], x + 1, y + r, fill);
        }
    }
}

// PX consecutive pixels of row y from one canvas.  cv2.warpAffine coordinates: adelta = round(m0*x*1024), X0 =
// round((m1*y + m2)*1024) + 16, position in 1/32 px = (X0 + adelta) >> 5 - 32-bit like cv2's own saturate_cast<int>.
template <int PX>
__device__ __forceinline__ void canvas_pixels(const CanvasLds &L, int x0, int y, int fill, Bgr (&out)[PX]) {
    Rects R;
#pragma unroll
    for (int i = 0; i < 4; ++i) R.r[i] = L.rect[i];
    if (!L.warp) {
#pragma unroll
        for (int i = 0; i < PX; ++i) out[i] = canvas_tap(L, R, x0 + i, y, fill);
        return;
    }
    const double fy = (double)y;
    const int X0 = __double2int_rn((L.minv[1] * fy + L.minv[2]) * 1024.0) + 16;
    const int Y0 = __double2int_rn((L.minv[4] * fy + L.minv[5]) * 1024.0) + 16;
    const double m0 = L.minv[0], m3 = L.minv[3];
#pragma unroll
    for (int i = 0; i < PX; ++i) {
        const double fx = (double)(x0 + i);
        const int X = (X0 + __double2int_rn(m0 * fx * 1024.0)) >> 5, Y = (Y0 + __double2int_rn(m3 * fx * 1024.0)) >> 5;
        const int sx = min(max(X >> 5, -32768), 32767), sy = min(max(Y >> 5, -32768), 32767);      // saturate_cast<short>
        const int wx1 = X & 31, wy1 = Y & 31, wx0 = 32 - wx1, wy0 = 32 - wy1;
        Bgr t[2][2];
        canvas_footprint(L, R, sx, sy, fill, t);
        // 15-bit weights 32*(32-fx)*(32-fy) ... sum to 32768 exactly
        out[i].b = (32 * (wy0 * (wx0 * t[0][0].b + wx1 * t[0][1].b) + wy1 * (wx0 * t[1][0].b + wx1 * t[1][1].b)) + 16384) >> 15;
        out[i].g = (32 * (wy0 * (wx0 * t[0][0].g + wx1 * t[0][1].g) + wy1 * (wx0 * t[1][0].g + wx1 * t[1][1].g)) + 16384) >> 15;
        out[i].r = (32 * (wy0 * (wx0 * t[0][0].r + wx1 * t[0][1].r) + wy1 * (wx0 * t[1][0].r + wx1 * t[1][1].r)) + 16384) >> 15;
    }
}

__device__ __forceinline__ int mix_u8(int a, int b, double r, double r1) {
    const double ar = (double)a * r, br = (double)b * r1;
    return (int)(ar + br);          // astype(uint8): truncation
}

__device__ __forceinline__ int to_u8(float v) {
    const int i = __float2int_rn(v * 255.f);
    return i < 0 ? 0 : (i > 255 ? 255 : i);
}

// BGR -> HSV (8-bit integer path) -> jitter tables -> HSV -> BGR (float path)
__device__ __forceinline__ Bgr hsv_jitter(Bgr in, const int *sdiv, const int *hdiv, const uint8_t (*lut)[256]) {
    const int b = in.b, g = in.g, r = in.r;
    const int v = max(max(b, g), r), vmin = min(min(b, g), r), diff = v - vmin;
    const int s = (diff * sdiv[v] + 2048) >> 12;
    int h = v == r ? g - b : (v == g ? b - r + 2 * diff : r - g + 4 * diff);
    h = (h * hdiv[diff] + 2048) >> 12;
    if (h < 0) h += 180;
    h = h > 255 ? 255 : h;
    const int hh = lut[0][h], ss = lut[1][s], vv = lut[2][v];
    float fh = (float)hh;
    const float fs = (float)ss * (1.0f / 255.0f), fv = (float)vv * (1.0f / 255.0f);
    float fb, fg, fr;
    if (fs == 0.f) {
        fb = fg = fr = fv;
    } else {
        fh = fh * (6.0f / 180.0f);
        while (fh >= 6.f) fh = fh - 6.f;                             // fmod(h, 6): exact for these magnitudes
        int sector = (int)floorf(fh);
        fh = fh - (float)sector;
        if ((unsigned)sector >= 6u) { sector = 0; fh = 0.f; }
        const float t0 = fv;
        const float t1 = fv * (1.f - fs);
        const float sh = fs * fh, t2 = fv * (1.f - sh);
        const float s1h = fs * (1.f - fh), t3 = fv * (1.f - s1h);
        switch (sector) {                                            // (b, g, r) picks of HSV2RGB_native's sector table
            case 0: fb = t1; fg = t3; fr = t0; break;
            case 1: fb = t1; fg = t0; fr = t2; break;
            case 2: fb = t3; fg = t0; fr = t1; break;
            case 3: fb = t0; fg = t2; fr = t1; break;
            case 4: fb = t0; fg = t1; fr = t3; break;
            default: fb = t2; fg = t1; fr = t0; break;
        }
    }
    return Bgr{to_u8(fb), to_u8(fg), to_u8(fr)};
}

// PX consecutive pixels of one row per thread (PX = 4 needs W % 4 == 0: three aligned 32-bit stores per thread).
template <int PX>
__global__ __launch_bounds__(256) void augment_kernel(const somi_aug_sample *__restrict__ samples, int H, int W, int fill,
                                                      uint8_t *__restrict__ out) {
    __shared__ int sdiv[256], hdiv[256];
    __shared__ uint8_t lut[3][256];
    __shared__ CanvasLds canvas[2];
    const somi_aug_sample &S = samples[blockIdx.y];
    const int tid = threadIdx.x;
    const bool hsv = S.hsv != 0, mix = S.mix != 0;
    if (hsv) {
        sdiv[tid] = kHsvTables.sdiv[tid];
        hdiv[tid] = kHsvTables.hdiv[tid];
        lut[0][tid] = S.lut[0][tid]; lut[1][tid] = S.lut[1][tid]; lut[2][tid] = S.lut[2][tid];
    }
    if (tid < 8) {                                     // one lane per (canvas, source slot)
        const int k = tid >> 2, i = tid & 3;
        const somi_aug_canvas &c = S.canvas[k];
        CanvasLds &L = canvas[k];
        if (i < c.nsrc && (k == 0 || mix)) {
            const somi_aug_source &s = c.src[i];
            const int x1 = max(s.x1, 0), y1 = max(s.y1, 0), x2 = min(s.x2, c.width), y2 = min(s.y2, c.height);
            L.rect[i] = make_int4(x1, y1, max(x2 - x1, 0), max(y2 - y1, 0));
            L.base[i] = (long long)reinterpret_cast<uintptr_t>(s.pixels) - (long long)((s.dy * s.w + s.dx) * 3);
            L.pitch[i] = s.w;
        } else {
            L.rect[i] = make_int4(0, 0, 0, 0);
            L.base[i] = 0; L.pitch[i] = 0;
        }
        if (i == 0) {
            L.warp = c.warp;
#pragma unroll
            for (int j = 0; j < 6; ++j) L.minv[j] = c.minv[j];
        }
    }
    __syncthreads();
    const int p = (blockIdx.x * 256 + tid) * PX;
    if (p >= H * W) return;
    const int y = p / W, x0 = p - y * W;
    Bgr px[PX];
    canvas_pixels<PX>(canvas[0], x0, y, fill, px);
    if (mix) {
        Bgr q[PX];
        canvas_pixels<PX>(canvas[1], x0, y, fill, q);
        const double m = S.mix_r, m1 = 1.0 - m;
#pragma unroll
        for (int i = 0; i < PX; ++i) px[i] = Bgr{mix_u8(px[i].b, q[i].b, m, m1), mix_u8(px[i].g, q[i].g, m, m1), mix_u8(px[i].r, q[i].r, m, m1)};
    }
    const bool fliplr = S.fliplr != 0;
    uint32_t pr = 0, pg = 0, pb = 0;
#pragma unroll
    for (int i = 0; i < PX; ++i) {
        if (hsv) px[i] = hsv_jitter(px[i], sdiv, hdiv, lut);
        const int sh = 8 * (fliplr ? PX - 1 - i : i);               // byte position inside the thread's output word
        pr |= (uint32_t)px[i].r << sh; pg |= (uint32_t)px[i].g << sh; pb |= (uint32_t)px[i].b << sh;
    }
    const int yo = S.flipud ? H - 1 - y : y, xo = fliplr ? W - PX - x0 : x0;
    const size_t plane = (size_t)H * W;
    uint8_t *o = out + (size_t)blockIdx.y * 3 * plane + (size_t)yo * W + xo;
    if (PX == 4) {
        *reinterpret_cast<uint32_t *>(o) = pr;
        *reinterpret_cast<uint32_t *>(o + plane) = pg;
        *reinterpret_cast<uint32_t *>(o + 2 * plane) = pb;
    } else {
        o[0] = (uint8_t)pr; o[plane] = (uint8_t)pg; o[2 * plane] = (uint8_t)pb;
    }
}

}  // namespace somi

using namespace somi;

extern "C" int somi_augment_u8(const somi_aug_sample *samples, int B, int H, int W, int fill, uint8_t *out, somi_stream_t stream) {
    SOMI_REQUIRE(B >= 0 && H > 0 && W > 0 && (long)H * W < (1L << 31), SOMI_EINVAL, "augment: bad sizes");
    SOMI_REQUIRE(fill >= 0 && fill <= 255, SOMI_EINVAL, "augment: fill must be a byte value");
    if (B == 0) return 0;
    SOMI_REQUIRE(samples && out, SOMI_EINVAL, "augment: null argument");
    SOMI_REQUIRE(B <= 65535, SOMI_EINVAL, "augment: at most 65535 samples per launch");
    SOMI_REQUIRE(H <= 16384 && W <= 16384, SOMI_EINVAL, "augment: output larger than 16384 px");
    hipStream_t st = static_cast<hipStream_t>(stream);
    static const int px_env = getenv("SOMI_AUG_PX") ? atoi(getenv("SOMI_AUG_PX")) : 4;
    if (px_env == 4 && W % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 3u) == 0)
        hipLaunchKernelGGL(augment_kernel<4>, dim3(cdiv((long)H * W / 4, 256), B), dim3(256), 0, st, samples, H, W, fill, out);
    else
        hipLaunchKernelGGL(augment_kernel<1>, dim3(cdiv((long)H * W, 256), B), dim3(256), 0, st, samples, H, W, fill, out);
    return launch_status("somi_augment_u8");
}
