// Data-movement kernels of the stock YOLOv5 module set (BASELINE configs[0]): Concat, nn.Upsample feeding a Concat, Focus and
// the plain Detect head.  Everything here is HBM-bound copy / gather work on NHWC fp32 channel slices: one float4 per lane,
// consecutive lanes on consecutive channels of a pixel (full 256 B wave segments whenever C >= 64).
//
//   Concat      models/common.py:2085-2097   torch.cat(x, 1): producers that can write a slice of the concat buffer do so (no copy);
//                                            the others (a 2x nearest-upsampled map, a map that already lives elsewhere) are copied
//                                            by resample_copy_kernel, which folds the upsample into the copy
//   Focus       models/common.py:1973-1997   space-to-depth: (row parity, col parity) = (0,0), (1,0), (0,1), (1,1)
//   Detect      models/yolo.py:46-109        view(bs,na,no,ny,nx).permute(0,1,3,4,2) + eval decode
#include "common.h"

namespace somi {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int ew_grid(long items) {
    long g = (items + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}
static inline bool slice_ok(const void *p, int cs, int coff, int C) {
    return p && cs % 4 == 0 && coff % 4 == 0 && C % 4 == 0 && coff + C <= cs && aligned16(p);
}

// dst[b, h, w, dst_coff + c] = src[b, h >> up, w >> up, src_coff + c]   (dst is (B, Hs << up, Ws << up, dst_cs))
__global__ __launch_bounds__(256) void resample_copy_kernel(const float *__restrict__ src, int src_cs, int src_coff, float *__restrict__ dst,
                                                            int dst_cs, int dst_coff, int B, int Hd, int Wd, int C, int up) {
    const int C4 = C >> 2, Ws = Wd >> up, Hs = Hd >> up;
    const long items = (long)B * Hd * Wd * C4;
    for (long it = blockIdx.x * 256L + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
        const int c = (int)(it % C4) * 4;
        long p = it / C4;
        const int w = (int)(p % Wd);
        p /= Wd;
        const int h = (int)(p % Hd);
        const long b = p / Hd;
        const long sp = (b * Hs + (h >> up)) * Ws + (w >> up);
        *reinterpret_cast<f32x4 *>(dst + ((b * Hd + h) * Wd + w) * dst_cs + dst_coff + c) =
            *reinterpret_cast<const f32x4 *>(src + sp * src_cs + src_coff + c);
    }
}

// The adjoint: dsrc[b, h, w, c] (+)= sum over the 2^up x 2^up block of ddst[b, (h << up) + i, (w << up) + j, ddst_coff + c],
// summed in a fixed (i, j) order - deterministic.
__global__ __launch_bounds__(256) void resample_reduce_kernel(const float *__restrict__ ddst, int ddst_cs, int ddst_coff, float *__restrict__ dsrc,
                                                              int dsrc_cs, int dsrc_coff, int B, int Hs, int Ws, int C, int up, int accumulate) {
    const int C4 = C >> 2, Wd = Ws << up, Hd = Hs << up, n = 1 << up;
    const long items = (long)B * Hs * Ws * C4;
    for (long it = blockIdx.x * 256L + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
        const int c = (int)(it % C4) * 4;
        long p = it / C4;
        const int w = (int)(p % Ws);
        p /= Ws;
        const int h = (int)(p % Hs);
        const long b = p / Hs;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j)
                acc += *reinterpret_cast<const f32x4 *>(ddst + ((b * Hd + (h << up) + i) * Wd + (w << up) + j) * ddst_cs + ddst_coff + c);
        float *o = dsrc + ((b * Hs + h) * Ws + w) * dsrc_cs + dsrc_coff + c;
        if (accumulate) acc += *reinterpret_cast<const f32x4 *>(o);
        *reinterpret_cast<f32x4 *>(o) = acc;
    }
}

// Focus.  forward (inverse = 0): y[b, h, w, q*C + c] = x[b, 2h + (q & 1), 2w + (q >> 1), c], q = 0..3 - the cat order
// [::2,::2], [1::2,::2], [::2,1::2], [1::2,1::2] of models/common.py:1996.  inverse = 1 scatters y's layout back (the gradient).
// Scalar per element: C is 3 for the image stem, the tensors are small next to the conv that follows.
__global__ __launch_bounds__(256) void space_to_depth_kernel(const float *__restrict__ x, int x_cs, int x_coff, float *__restrict__ y, int y_cs,
                                                             int y_coff, int B, int Ho, int Wo, int C, int inverse) {
    const long items = (long)B * Ho * Wo * 4 * C;
    for (long it = blockIdx.x * 256L + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
        const int j = (int)(it % (4 * C));
        long p = it / (4 * C);
        const int w = (int)(p % Wo);
        p /= Wo;
        const int h = (int)(p % Ho);
        const long b = p / Ho;
        const int q = j / C, c = j % C;
        const long xi = ((b * 2 * Ho + 2 * h + (q & 1)) * (2L * Wo) + 2 * w + (q >> 1)) * x_cs + x_coff + c;
        const long yi = ((b * Ho + h) * (long)Wo + w) * y_cs + y_coff + j;
        if (inverse) y[xi] = x[yi];          // x = gradient in the deep layout, y = gradient in the image layout (strides swapped by the host)
        else y[yi] = x[xi];
    }
}

struct PlainDecodeArgs {
    float anchor_px[16];   // na*2, anchors * stride
};
// t (B,ny,nx,t_cs) holds na*(5+nc) conv outputs per pixel, anchor-major -> raw (B,na,ny,nx,no) and, in eval, the decoded rows of z:
// xy = (2*sigmoid - 0.5 + cell) * stride in exactly that order (models/yolo.py:92,96), wh = (2*sigmoid)^2 * anchor_px.
__global__ __launch_bounds__(256) void detect_plain_decode_kernel(const float *__restrict__ t, int t_cs, PlainDecodeArgs da, float stride,
                                                                  float *__restrict__ raw, float *__restrict__ z, int B, int ny, int nx, int na,
                                                                  int nc, int total, int row_off) {
    const int no = nc + 5;
    const long items = (long)B * na * ny * nx * no;
    for (long it = blockIdx.x * 256L + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
        const int o = (int)(it % no);
        long r = it / no;
        const int xg = (int)(r % nx);
        r /= nx;
        const int yg = (int)(r % ny);
        r /= ny;
        const int an = (int)(r % na);
        const long b = r / na;
        const float v = t[((b * ny + yg) * nx + xg) * t_cs + an * no + o];
        if (raw) raw[it] = v;
        if (z) {
            const float s = 1.0f / (1.0f + expf(-v));
            float out = s;
            if (o == 0) out = __fmul_rn(__fadd_rn(__fsub_rn(__fmul_rn(s, 2.0f), 0.5f), (float)xg), stride);
            else if (o == 1) out = __fmul_rn(__fadd_rn(__fsub_rn(__fmul_rn(s, 2.0f), 0.5f), (float)yg), stride);
            else if (o == 2) out = (s * 2.0f) * (s * 2.0f) * da.anchor_px[an * 2];
            else if (o == 3) out = (s * 2.0f) * (s * 2.0f) * da.anchor_px[an * 2 + 1];
            const long row = row_off + ((long)an * ny + yg) * nx + xg;
            z[(b * total + row) * no + o] = out;
        }
    }
}

// d raw (B,na,ny,nx,no) -> d t (B,ny,nx,t_cs); pad channels get zeros
__global__ __launch_bounds__(256) void detect_plain_raw_bwd_kernel(const float *__restrict__ draw, float *__restrict__ dt, int t_cs, int B, int ny,
                                                                   int nx, int na, int nc) {
    const int no = nc + 5;
    const long items = (long)B * ny * nx * t_cs;
    for (long it = blockIdx.x * 256L + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
        const int j = (int)(it % t_cs);
        const long pix = it / t_cs;
        const int xg = (int)(pix % nx), yg = (int)((pix / nx) % ny);
        const long b = pix / ((long)nx * ny);
        float v = 0.f;
        if (j < na * no) { const int an = j / no, o = j % no; v = draw[(((b * na + an) * ny + yg) * nx + xg) * no + o]; }
        dt[it] = v;
    }
}

// Test-time augmentation input (models/yolo.py:1260 `scale_img(x.flip(fi) if fi else x, si, gs)`, utils/torch_utils.py:270-282): NHWC4 image
// -> optionally flipped left-right, resized to (Hs, Ws) with torch's bilinear / align_corners=False arithmetic
// (src = (dst + 0.5) * in/out - 0.5 clamped at 0, neighbour clamped at the border), padded with `pad` up to (Hp, Wp).
__global__ __launch_bounds__(256) void tta_resample_kernel(const float *__restrict__ x, float *__restrict__ y, int B, int H, int W, int Hs, int Ws,
                                                           int Hp, int Wp, int flip_lr, float pad, int nch) {
    const long items = (long)B * Hp * Wp;
    const float sh = (float)H / (float)Hs, sw = (float)W / (float)Ws;
    for (long it = blockIdx.x * 256L + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
        const int w = (int)(it % Wp), h = (int)((it / Wp) % Hp);
        const long b = it / ((long)Wp * Hp);
        f32x4 v = {pad, pad, pad, pad};
        if (h < Hs && w < Ws) {
            float fh = sh * ((float)h + 0.5f) - 0.5f, fw = sw * ((float)w + 0.5f) - 0.5f;
            fh = fh < 0.f ? 0.f : fh;
            fw = fw < 0.f ? 0.f : fw;
            const int h0 = (int)fh, w0 = (int)fw;
            const int h1 = h0 + (h0 < H - 1 ? 1 : 0), w1 = w0 + (w0 < W - 1 ? 1 : 0);
            const float lh = fh - (float)h0, lw = fw - (float)w0;
            const int a0 = flip_lr ? W - 1 - w0 : w0, a1 = flip_lr ? W - 1 - w1 : w1;      // the flip happens before the resize
            const float *r0 = x + ((b * H + h0) * W) * 4, *r1 = x + ((b * H + h1) * W) * 4;
            const f32x4 p00 = *reinterpret_cast<const f32x4 *>(r0 + a0 * 4), p01 = *reinterpret_cast<const f32x4 *>(r0 + a1 * 4);
            const f32x4 p10 = *reinterpret_cast<const f32x4 *>(r1 + a0 * 4), p11 = *reinterpret_cast<const f32x4 *>(r1 + a1 * 4);
            v = (1.f - lh) * ((1.f - lw) * p00 + lw * p01) + lh * ((1.f - lw) * p10 + lw * p11);
        }
        for (int c = nch; c < 4; ++c) v[c] = 0.f;                                          // pad channels stay zero
        *reinterpret_cast<f32x4 *>(y + it * 4) = v;
    }
}

// `_descale_pred` (models/yolo.py:1292-1308) on rows of z (rows, no): xywh /= scale; x = img_w - x after a left-right flip
__global__ __launch_bounds__(256) void tta_descale_kernel(float *__restrict__ z, long rows, int no, float scale, int flip_lr, float img_w) {
    for (long r = blockIdx.x * 256L + threadIdx.x; r < rows; r += (long)gridDim.x * 256) {
        float *p = z + r * no;
        float px = p[0] / scale;
        if (flip_lr) px = img_w - px;
        p[0] = px;
        p[1] = p[1] / scale;
        p[2] = p[2] / scale;
        p[3] = p[3] / scale;
    }
}

}  // namespace somi

using namespace somi;

extern "C" int somi_tta_resample_nhwc4_f32(const float *x, float *y, int B, int H, int W, int Hs, int Ws, int Hp, int Wp, int flip_lr, float pad,
                                           int channels, somi_stream_t stream) {
    SOMI_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && Hs > 0 && Ws > 0 && Hp >= Hs && Wp >= Ws && channels >= 1 && channels <= 4 && aligned16(x) &&
                     aligned16(y), SOMI_EINVAL, "tta resample: bad arguments");
    hipLaunchKernelGGL(tta_resample_kernel, dim3(ew_grid((long)B * Hp * Wp)), dim3(256), 0, (hipStream_t)stream, x, y, B, H, W, Hs, Ws, Hp, Wp, flip_lr,
                       pad, channels);
    return launch_status("somi_tta_resample_nhwc4_f32");
}

extern "C" int somi_tta_descale_f32(float *z, long rows, int no, float scale, int flip_lr, float img_w, somi_stream_t stream) {
    SOMI_REQUIRE(z && rows > 0 && no >= 4 && scale > 0.f, SOMI_EINVAL, "tta descale: bad arguments");
    hipLaunchKernelGGL(tta_descale_kernel, dim3(ew_grid(rows)), dim3(256), 0, (hipStream_t)stream, z, rows, no, scale, flip_lr, img_w);
    return launch_status("somi_tta_descale_f32");
}

extern "C" int somi_resample_slice_nhwc_f32(const float *src, int src_cs, int src_coff, float *dst, int dst_cs, int dst_coff, int B, int Hs,
                                            int Ws, int C, int up, int reduce, int accumulate, somi_stream_t stream) {
    SOMI_REQUIRE(B > 0 && Hs > 0 && Ws > 0 && C > 0 && up >= 0 && up <= 3, SOMI_EINVAL, "resample slice: bad sizes (up <= 3)");
    SOMI_REQUIRE(slice_ok(src, src_cs, src_coff, C) && slice_ok(dst, dst_cs, dst_coff, C), SOMI_EINVAL,
                 "resample slice: channel slices have to be 16-byte aligned multiples of 4");
    hipStream_t s = (hipStream_t)stream;
    if (!reduce) {      // src (B,Hs,Ws,.) -> dst (B,Hs<<up,Ws<<up,.)
        SOMI_REQUIRE(!accumulate, SOMI_EINVAL, "resample slice: accumulate is for the reducing direction only");
        hipLaunchKernelGGL(resample_copy_kernel, dim3(ew_grid((long)B * (Hs << up) * (Ws << up) * (C / 4))), dim3(256), 0, s, src, src_cs, src_coff, dst,
                           dst_cs, dst_coff, B, Hs << up, Ws << up, C, up);
    } else {            // src (B,Hs<<up,Ws<<up,.) -> dst (B,Hs,Ws,.)
        hipLaunchKernelGGL(resample_reduce_kernel, dim3(ew_grid((long)B * Hs * Ws * (C / 4))), dim3(256), 0, s, src, src_cs, src_coff, dst, dst_cs,
                           dst_coff, B, Hs, Ws, C, up, accumulate);
    }
    return launch_status("somi_resample_slice_nhwc_f32");
}

extern "C" int somi_space_to_depth_nhwc_f32(const float *x, int x_cs, int x_coff, float *y, int y_cs, int y_coff, int B, int Ho, int Wo, int C,
                                            int inverse, somi_stream_t stream) {
    SOMI_REQUIRE(x && y && B > 0 && Ho > 0 && Wo > 0 && C > 0, SOMI_EINVAL, "space to depth: bad arguments");
    if (!inverse) {     // x (B,2Ho,2Wo,x_cs) -> y (B,Ho,Wo,y_cs)
        SOMI_REQUIRE(x_coff + C <= x_cs && y_coff + 4 * C <= y_cs, SOMI_EINVAL, "space to depth: slices out of range");
        hipLaunchKernelGGL(space_to_depth_kernel, dim3(ew_grid((long)B * Ho * Wo * 4 * C)), dim3(256), 0, (hipStream_t)stream, x, x_cs, x_coff, y, y_cs,
                           y_coff, B, Ho, Wo, C, 0);
    } else {            // x = deep-layout gradient (B,Ho,Wo,x_cs) -> y = image-layout gradient (B,2Ho,2Wo,y_cs): every element written once
        SOMI_REQUIRE(x_coff + 4 * C <= x_cs && y_coff + C <= y_cs, SOMI_EINVAL, "space to depth: slices out of range");
        hipLaunchKernelGGL(space_to_depth_kernel, dim3(ew_grid((long)B * Ho * Wo * 4 * C)), dim3(256), 0, (hipStream_t)stream, x, y_cs, y_coff, y, x_cs,
                           x_coff, B, Ho, Wo, C, 1);
    }
    return launch_status("somi_space_to_depth_nhwc_f32");
}

extern "C" int somi_detect_plain_decode_f32(const float *t, int t_cs, const float *anchors_px_host, float stride, float *raw, float *z, int B,
                                            int ny, int nx, int na, int nc, int total, int row_off, somi_stream_t stream) {
    SOMI_REQUIRE(t && anchors_px_host && (raw || z), SOMI_EINVAL, "detect decode: null tensor");
    SOMI_REQUIRE(B > 0 && ny > 0 && nx > 0 && na > 0 && na <= 8 && nc > 0 && t_cs >= na * (nc + 5), SOMI_EINVAL, "detect decode: bad sizes (na <= 8)");
    SOMI_REQUIRE(!z || (row_off >= 0 && row_off + na * ny * nx <= total), SOMI_EINVAL, "detect decode: rows out of range");
    PlainDecodeArgs da;
    for (int i = 0; i < 16; ++i) da.anchor_px[i] = i < na * 2 ? anchors_px_host[i] : 0.f;
    hipLaunchKernelGGL(detect_plain_decode_kernel, dim3(ew_grid((long)B * na * ny * nx * (nc + 5))), dim3(256), 0, (hipStream_t)stream, t, t_cs, da, stride,
                       raw, z, B, ny, nx, na, nc, total, row_off);
    return launch_status("somi_detect_plain_decode_f32");
}

extern "C" int somi_detect_plain_raw_bwd_f32(const float *draw, float *dt, int t_cs, int B, int ny, int nx, int na, int nc, somi_stream_t stream) {
    SOMI_REQUIRE(draw && dt && B > 0 && ny > 0 && nx > 0 && na > 0 && nc > 0 && t_cs >= na * (nc + 5), SOMI_EINVAL, "detect raw bwd: bad arguments");
    hipLaunchKernelGGL(detect_plain_raw_bwd_kernel, dim3(ew_grid((long)B * ny * nx * t_cs)), dim3(256), 0, (hipStream_t)stream, draw, dt, t_cs, B, ny, nx,
                       na, nc);
    return launch_status("somi_detect_plain_raw_bwd_f32");
}
