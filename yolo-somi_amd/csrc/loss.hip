// YOLO-SOMI loss on gfx950: ComputeLoss.__call__ + build_targets (utils/loss.py:142-262) with the CIoU of
// utils/metrics.py:476-518, forward value and (optionally) the gradient w.r.t. every prediction tensor.
//
// The reference issues ~40 small torch ops per level plus boolean-mask syncs; here four launches do everything:
//   1. match    one lane per (level, anchor, target): the anchor-ratio test and the <= 5 neighbour cells of
//               build_targets; for every emitted entry it gathers the 5+nc logits, decodes the box, computes CIoU
//               (forward-mode duals give d(1-CIoU)/d logits), the class BCE, and
//                 - stores 1-CIoU and the class-BCE sum in the entry's slot (deterministic reduction later),
//                 - atomicMax's clamp(CIoU,0,1) into the objectness target map (the reference sorts by IoU and lets
//                   the last write win, utils/loss.py:174-178: the surviving value is the per-cell maximum),
//                 - atomically adds the unscaled box / class gradients into grad (cells can be hit more than once).
//   2. count    per-level number of entries (needed to scale the means) - folded into the dense pass below.
//   3. dense    one streaming pass over every prediction element: objectness BCE against the target map (partial
//               sums per workgroup) and, when gradients are wanted, the objectness gradient plus the 1/n scaling of
//               the box / class gradients written in step 1.
//   4. finish   fixed-order reduction of slots and partials -> out[0..3] = (lbox+lobj+lcls)*bs, lbox, lobj, lcls.
// Branches hyp.VisDrone.yaml switches off (focal, slide, NWD, autobalance) are rejected by the host layer.
#include "common.h"

namespace somi {

constexpr float CIOU_EPS = 1e-7f;
constexpr int NOFF = 5;

struct LossArgs {
    somi_loss_desc d;
    int no;
    long cells[4];          // B*na*ny*nx per level
    long cell_off[4];       // prefix of cells (offset into the tobj workspace)
    int nblk[4];            // dense workgroups per level
    int blk_off[4];
    unsigned *tobj;         // [sum cells] float bits (>= 0), zeroed
    float *slots;           // [nl][na*nt*NOFF][2] = {1-ciou or -1 (invalid), cls bce sum}
    int *nent;              // [4] entries per level, zeroed
    float *partial;         // [sum nblk] obj-BCE partial sums
};

// ---- forward-mode dual numbers over the 4 decoded box coordinates (px, py, pw, ph)
struct Dual {
    float v, g[4];
};
__device__ __forceinline__ Dual dconst(float v) { return {v, {0.f, 0.f, 0.f, 0.f}}; }
__device__ __forceinline__ Dual dvar(float v, int i) { Dual r = dconst(v); r.g[i] = 1.f; return r; }
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { Dual r; r.v = a.v + b.v; for (int i = 0; i < 4; ++i) r.g[i] = a.g[i] + b.g[i]; return r; }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { Dual r; r.v = a.v - b.v; for (int i = 0; i < 4; ++i) r.g[i] = a.g[i] - b.g[i]; return r; }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { Dual r; r.v = a.v * b.v; for (int i = 0; i < 4; ++i) r.g[i] = a.g[i] * b.v + a.v * b.g[i]; return r; }
__device__ __forceinline__ Dual operator/(Dual a, Dual b) {
    Dual r; r.v = a.v / b.v; const float inv = 1.f / b.v;
    for (int i = 0; i < 4; ++i) r.g[i] = (a.g[i] - r.v * b.g[i]) * inv;
    return r;
}
__device__ __forceinline__ Dual operator*(Dual a, float s) { Dual r; r.v = a.v * s; for (int i = 0; i < 4; ++i) r.g[i] = a.g[i] * s; return r; }
__device__ __forceinline__ Dual operator+(Dual a, float s) { a.v += s; return a; }
__device__ __forceinline__ Dual dmin(Dual a, Dual b) { return a.v <= b.v ? a : b; }      // torch.minimum: ties -> grad split; measure-zero
__device__ __forceinline__ Dual dmax(Dual a, Dual b) { return a.v >= b.v ? a : b; }
__device__ __forceinline__ Dual dclamp0(Dual a) { return a.v > 0.f ? a : dconst(0.f); }   // clamp(0): zero grad at/below 0
__device__ __forceinline__ Dual datan(Dual a) { Dual r; r.v = atanf(a.v); const float s = 1.f / (1.f + a.v * a.v); for (int i = 0; i < 4; ++i) r.g[i] = a.g[i] * s; return r; }

// CIoU of predicted (px,py,pw,ph) vs target (tx,ty,tw,th), xywh (utils/metrics.py:476-518 with alpha=1)
__device__ __forceinline__ Dual ciou_xywh(Dual px, Dual py, Dual pw, Dual ph, float tx, float ty, float tw, float th) {
    const Dual b1x1 = px - pw * 0.5f, b1x2 = px + pw * 0.5f, b1y1 = py - ph * 0.5f, b1y2 = py + ph * 0.5f;
    const Dual b2x1 = dconst(tx - tw / 2), b2x2 = dconst(tx + tw / 2), b2y1 = dconst(ty - th / 2), b2y2 = dconst(ty + th / 2);
    const Dual inter = dclamp0(dmin(b1x2, b2x2) - dmax(b1x1, b2x1)) * dclamp0(dmin(b1y2, b2y2) - dmax(b1y1, b2y1));
    const Dual w1 = b1x2 - b1x1, h1 = (b1y2 - b1y1) + CIOU_EPS;
    const Dual w2 = b2x2 - b2x1, h2 = (b2y2 - b2y1) + CIOU_EPS;
    const Dual uni = (w1 * h1 + w2 * h2 - inter) + CIOU_EPS;
    const Dual iou = inter / (uni + CIOU_EPS);
    const Dual cw = dmax(b1x2, b2x2) - dmin(b1x1, b2x1), ch = dmax(b1y2, b2y2) - dmin(b1y1, b2y1);
    const Dual c2 = (cw * cw + ch * ch) + CIOU_EPS;
    const Dual dx = (b2x1 + b2x2) - b1x1 - b1x2, dy = (b2y1 + b2y2) - b1y1 - b1y2;
    const Dual rho2 = (dx * dx + dy * dy) * 0.25f;
    const Dual da = datan(w2 / h2) - datan(w1 / h1);
    const Dual v = (da * da) * (4.f / (3.14159265358979323846f * 3.14159265358979323846f));
    const float alpha = v.v / (v.v - iou.v + (1.f + CIOU_EPS));          // computed under no_grad
    return iou - (rho2 / c2 + (v * alpha + CIOU_EPS));
}

__device__ __forceinline__ float softplus_neg(float x) { return fmaxf(-x, 0.f) + log1pf(expf(-fabsf(x))); }
// BCEWithLogits(pos_weight=pw): (1-y)*x + (1+(pw-1)*y)*softplus(-x)
__device__ __forceinline__ float bce_logits(float x, float y, float pw) { return (1.f - y) * x + (1.f + (pw - 1.f) * y) * softplus_neg(x); }
__device__ __forceinline__ float bce_logits_grad(float x, float y, float pw) {
    const float s = 1.f / (1.f + expf(-x));
    return (1.f - y) - (1.f + (pw - 1.f) * y) * (1.f - s);
}

// ------------------------------------------------------------------------------------------------ 1. match
__global__ __launch_bounds__(256) void loss_match_kernel(const LossArgs a) {
    const somi_loss_desc &d = a.d;
    const int l = blockIdx.y;
    const int per_level = d.na * d.nt;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= per_level) return;
    const int an = idx / d.nt, t = idx % d.nt;                 // anchor-major like targets.repeat(na,1,1)
    const int nx = d.nx[l], ny = d.ny[l];
    float *slot = a.slots + ((size_t)l * per_level + idx) * NOFF * 2;
    for (int k = 0; k < NOFF; ++k) slot[k * 2] = -1.f;         // invalid
    const float *tg = d.targets + (size_t)t * 6;
    const int b = (int)tg[0], cls = (int)tg[1];
    const float gx = tg[2] * (float)nx, gy = tg[3] * (float)ny, gw = tg[4] * (float)nx, gh = tg[5] * (float)ny;
    const float aw = d.anchors[(l * d.na + an) * 2], ah = d.anchors[(l * d.na + an) * 2 + 1];
    const float rw = gw / aw, rh = gh / ah;
    if (!(fmaxf(fmaxf(rw, 1.f / rw), fmaxf(rh, 1.f / rh)) < d.anchor_t)) return;          // utils/loss.py:233-235
    const float g = 0.5f;
    const float gxi = (float)nx - gx, gyi = (float)ny - gy;
    const bool fj = (fmodf(gx, 1.f) < g) && gx > 1.f, fk = (fmodf(gy, 1.f) < g) && gy > 1.f;
    const bool fl = (fmodf(gxi, 1.f) < g) && gxi > 1.f, fm = (fmodf(gyi, 1.f) < g) && gyi > 1.f;
    const bool use[NOFF] = {true, fj, fk, fl, fm};
    const float offx[NOFF] = {0.f, g, 0.f, -g, 0.f}, offy[NOFF] = {0.f, 0.f, g, 0.f, -g};
    const float *pl = d.p[l];
    float *gl = d.grad[l];
    int emitted = 0;
    for (int k = 0; k < NOFF; ++k) {
        if (!use[k]) continue;
        int gi = (int)(gx - offx[k]), gj = (int)(gy - offy[k]);                           // .long(): truncation
        gi = min(max(gi, 0), nx - 1);                                                     // clamp_ (also feeds tbox)
        gj = min(max(gj, 0), ny - 1);
        const float tbx = gx - (float)gi, tby = gy - (float)gj;
        const size_t cell = (((size_t)b * d.na + an) * ny + gj) * nx + gi;
        const float *ps = pl + cell * a.no;
        const float s0 = 1.f / (1.f + expf(-ps[0])), s1 = 1.f / (1.f + expf(-ps[1]));
        const float s2 = 1.f / (1.f + expf(-ps[2])), s3 = 1.f / (1.f + expf(-ps[3]));
        const float pxv = s0 * 2.f - 0.5f, pyv = s1 * 2.f - 0.5f;
        const float pwv = (s2 * 2.f) * (s2 * 2.f) * aw, phv = (s3 * 2.f) * (s3 * 2.f) * ah;
        const Dual c = ciou_xywh(dvar(pxv, 0), dvar(pyv, 1), dvar(pwv, 2), dvar(phv, 3), tbx, tby, gw, gh);
        slot[k * 2] = 1.f - c.v;
        const float iou01 = fminf(fmaxf(c.v, 0.f), 1.f);
        atomicMax(a.tobj + a.cell_off[l] + cell, __float_as_uint((1.f - d.gr) + d.gr * iou01));
        float csum = 0.f;
        if (d.nc > 1) {
            for (int j = 0; j < d.nc; ++j) {
                const float y = j == cls ? d.cp : d.cn;
                csum += bce_logits(ps[5 + j], y, d.cls_pw);
                if (gl) atomicAdd(gl + cell * a.no + 5 + j, bce_logits_grad(ps[5 + j], y, d.cls_pw));
            }
        }
        slot[k * 2 + 1] = csum;
        if (gl) {                                                                          // d(1-ciou)/d logits, unscaled
            atomicAdd(gl + cell * a.no + 0, -c.g[0] * 2.f * s0 * (1.f - s0));
            atomicAdd(gl + cell * a.no + 1, -c.g[1] * 2.f * s1 * (1.f - s1));
            atomicAdd(gl + cell * a.no + 2, -c.g[2] * 8.f * s2 * s2 * (1.f - s2) * aw);
            atomicAdd(gl + cell * a.no + 3, -c.g[3] * 8.f * s3 * s3 * (1.f - s3) * ah);
        }
        ++emitted;
    }
    if (emitted) atomicAdd(a.nent + l, emitted);
}

// ------------------------------------------------------------------------------------------------ 3. dense pass
__global__ __launch_bounds__(256) void loss_dense_kernel(const LossArgs a, int l) {
    __shared__ float red[4];
    const somi_loss_desc &d = a.d;
    const float *pl = d.p[l];
    float *gl = d.grad[l];
    const long cells = a.cells[l];
    const int n = a.nent[l];
    const float bs = (float)d.B;
    const float obj_scale = d.balance[l] * d.obj_gain * bs / (float)cells;
    const float box_scale = n ? d.box_gain * bs / (float)n : 0.f;
    const float cls_scale = n ? d.cls_gain * bs / ((float)n * (float)d.nc) : 0.f;
    float part = 0.f;
    if (gl) {
        const long items = cells * a.no;
        for (long it = blockIdx.x * 256L + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
            const int ch = (int)(it % a.no);
            if (ch == 4) {
                const float x = pl[it], y = __uint_as_float(a.tobj[a.cell_off[l] + it / a.no]);
                part += bce_logits(x, y, d.obj_pw);
                gl[it] = bce_logits_grad(x, y, d.obj_pw) * obj_scale;
            } else {
                const float gv = gl[it];
                if (gv != 0.f) gl[it] = gv * (ch < 4 ? box_scale : cls_scale);
            }
        }
    } else {
        for (long cidx = blockIdx.x * 256L + threadIdx.x; cidx < cells; cidx += (long)gridDim.x * 256)
            part += bce_logits(pl[cidx * a.no + 4], __uint_as_float(a.tobj[a.cell_off[l] + cidx]), d.obj_pw);
    }
    for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) a.partial[a.blk_off[l] + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ------------------------------------------------------------------------------------------------ 4. finish
__global__ __launch_bounds__(256) void loss_finish_kernel(const LossArgs a, float *out4) {
    __shared__ double red[256];
    const somi_loss_desc &d = a.d;
    double lbox = 0.0, lobj = 0.0, lcls = 0.0;
    const int per_level = d.na * d.nt * NOFF;
    for (int l = 0; l < d.nl; ++l) {
        // fixed-order reductions: thread-strided partial sums, then a tree over the 256 lanes
        double sb = 0.0, sc = 0.0, so = 0.0;
        const float *sl = a.slots + (size_t)l * per_level * 2;
        for (int i = threadIdx.x; i < per_level; i += 256)
            if (sl[i * 2] >= 0.f) { sb += sl[i * 2]; sc += sl[i * 2 + 1]; }
        for (int i = threadIdx.x; i < a.nblk[l]; i += 256) so += a.partial[a.blk_off[l] + i];
        double vals[3] = {sb, sc, so};
        for (int q = 0; q < 3; ++q) {
            red[threadIdx.x] = vals[q];
            __syncthreads();
            for (int s = 128; s > 0; s >>= 1) {
                if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
                __syncthreads();
            }
            vals[q] = red[0];
            __syncthreads();
        }
        const int n = a.nent[l];
        if (n) {
            lbox += vals[0] / n;
            if (d.nc > 1) lcls += vals[1] / ((double)n * d.nc);
        }
        lobj += vals[2] / (double)a.cells[l] * d.balance[l];
    }
    if (threadIdx.x == 0) {
        const float fb = (float)lbox * d.box_gain, fo = (float)lobj * d.obj_gain, fc = (float)lcls * d.cls_gain;
        out4[0] = (fb + fo + fc) * (float)d.B;
        out4[1] = fb;
        out4[2] = fo;
        out4[3] = fc;
    }
}

static size_t align_up(size_t v, size_t al) { return (v + al - 1) / al * al; }

static int plan(const somi_loss_desc &d, LossArgs &a) {
    SOMI_REQUIRE(d.nl >= 1 && d.nl <= 4 && d.na >= 1 && d.nc >= 1 && d.B >= 1 && d.nt >= 0, SOMI_EINVAL, "loss: bad sizes");
    a.d = d;
    a.no = d.nc + 5;
    long off = 0;
    int boff = 0;
    for (int l = 0; l < 4; ++l) {
        a.cells[l] = a.cell_off[l] = 0;
        a.nblk[l] = a.blk_off[l] = 0;
        if (l >= d.nl) continue;
        SOMI_REQUIRE(d.ny[l] > 0 && d.nx[l] > 0 && d.p[l], SOMI_EINVAL, "loss: level %d is empty", l);
        a.cells[l] = (long)d.B * d.na * d.ny[l] * d.nx[l];
        a.cell_off[l] = off;
        off += a.cells[l];
        const long items = d.grad[l] ? a.cells[l] * a.no : a.cells[l];
        long nb = (items + 255) / 256;
        a.nblk[l] = (int)(nb > 2048 ? 2048 : nb);
        a.blk_off[l] = boff;
        boff += a.nblk[l];
    }
    return 0;
}

}  // namespace somi

using namespace somi;

extern "C" size_t somi_loss_workspace_bytes(const somi_loss_desc *d) {
    LossArgs a;
    if (!d || plan(*d, a)) return 0;
    const size_t tobj = align_up((size_t)(a.cell_off[d->nl - 1] + a.cells[d->nl - 1]) * 4, 256);
    const size_t slots = align_up((size_t)d->nl * d->na * (d->nt > 0 ? d->nt : 1) * NOFF * 2 * 4, 256);
    const size_t part = align_up((size_t)(a.blk_off[d->nl - 1] + a.nblk[d->nl - 1]) * 4, 256);
    return tobj + 256 + slots + part;
}

extern "C" int somi_yolo_loss_f32(const somi_loss_desc *dp, float *out4, void *workspace, size_t workspace_bytes,
                                  somi_stream_t stream) {
    SOMI_REQUIRE(dp && out4 && workspace, SOMI_EINVAL, "loss: null argument");
    LossArgs a;
    int rc = plan(*dp, a);
    if (rc) return rc;
    const somi_loss_desc &d = *dp;
    SOMI_REQUIRE(d.nt == 0 || (d.targets && d.anchors), SOMI_EINVAL, "loss: targets / anchors missing");
    SOMI_REQUIRE(workspace_bytes >= somi_loss_workspace_bytes(dp), SOMI_EWORKSPACE, "loss: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    char *w = static_cast<char *>(workspace);
    const size_t tobj_b = align_up((size_t)(a.cell_off[d.nl - 1] + a.cells[d.nl - 1]) * 4, 256);
    a.tobj = reinterpret_cast<unsigned *>(w); w += tobj_b;
    a.nent = reinterpret_cast<int *>(w); w += 256;
    a.slots = reinterpret_cast<float *>(w); w += align_up((size_t)d.nl * d.na * (d.nt > 0 ? d.nt : 1) * NOFF * 2 * 4, 256);
    a.partial = reinterpret_cast<float *>(w);
    (void)hipMemsetAsync(a.tobj, 0, tobj_b + 256, s);                       // target maps + entry counters
    for (int l = 0; l < d.nl; ++l)
        if (d.grad[l]) (void)hipMemsetAsync(d.grad[l], 0, (size_t)a.cells[l] * a.no * 4, s);
    if (d.nt > 0)
        hipLaunchKernelGGL(loss_match_kernel, dim3(cdiv((long)d.na * d.nt, 256), d.nl), dim3(256), 0, s, a);
    for (int l = 0; l < d.nl; ++l) hipLaunchKernelGGL(loss_dense_kernel, dim3(a.nblk[l]), dim3(256), 0, s, a, l);
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(256), 0, s, a, out4);
    return launch_status("somi_yolo_loss_f32");
}
