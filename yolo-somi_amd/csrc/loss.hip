// YOLO-SOMI loss on gfx950: ComputeLoss.__call__ + build_targets (utils/loss.py:142-262) with the CIoU of
// utils/metrics.py:476-518, forward value and (optionally) the gradient w.r.t. every prediction tensor.
//
// The reference issues ~40 small torch ops per level plus boolean-mask syncs; here four launches do everything:
//   1. match    one lane per (level, anchor, target): the anchor-ratio test and the <= 5 neighbour cells of
//               build_targets; for every emitted entry it gathers the 5+nc logits, decodes the box, computes CIoU
//               (forward-mode duals give d(1-CIoU)/d logits), the class BCE, and
//                 - stores 1-CIoU and the class-BCE sum in the entry's slot (deterministic reduction later),
//                 - atomicMax's clamp(CIoU,0,1) into the objectness target map (the reference sorts by IoU and lets
//                   the last write win, utils/loss.py:174-178: the surviving value is the per-cell maximum; an integer
//                   max, so the result does not depend on the order),
//                 - pushes the entry onto its cell's list (an integer exchange on a per-cell head index).
//   1b. scatter one lane per entry; the lane whose entry heads a cell's list walks the list in ASCENDING entry order
//               (cells can be hit more than once: duplicate / clamped targets) and writes the summed unscaled box / class
//               gradients with plain stores - a fixed summation order, so the gradient is run-to-run bit-identical.
//   2. count    per-level number of entries (needed to scale the means) - folded into the dense pass below.
//   3. dense    one streaming pass over every prediction element: objectness BCE against the target map (partial
//               sums per workgroup) and, when gradients are wanted, the objectness gradient plus the 1/n scaling of
//               the box / class gradients written in step 1.
//   4. finish   fixed-order reduction of slots and partials -> out[0..3] = (lbox+lobj+lcls)*bs, lbox, lobj, lcls.
// The branches hyp.VisDrone.yaml leaves off are here too: FocalLoss (utils/loss.py:35-60) and SlideLoss (:378-402) as per-element
// weights of both BCE terms (stacked like :125-131), the NWD box term (:162-169, utils/metrics.py:341-354) blended into the box loss,
// its gradient and the objectness target.  SlideLoss needs the level's mean IoU before any BCE can be weighted, so the class BCE of
// the matched entries is evaluated in pass 1b (after a one-workgroup-per-level mean, 1a).  autobalance is host state: the per-level objectness means it feeds on leave in out[4..7].
#include "common.h"

namespace somi {

constexpr float CIOU_EPS = 1e-7f;
constexpr int NOFF = 5;
constexpr int SLOT = 3;
constexpr float NWD_EPS = 1e-7f, FOCAL_ALPHA = 0.25f;

constexpr int LOSS_FOLD = 64;   // slot segments per level: one workgroup each, so that the finish kernel does not walk ~34000 slots per level alone
struct LossArgs {
    somi_loss_desc d;
    int no;
    long cells[4];          // B*na*ny*nx per level
    long cell_off[4];       // prefix of cells (offset into the tobj workspace)
    int nblk[4];            // dense workgroups per level
    int blk_off[4];
    unsigned *tobj;         // [sum cells] float bits (>= 0), zeroed
    float *slots;           // [nl][na*nt*NOFF][SLOT] = {box loss or -1 (invalid), cls bce sum, clamp(iou,0,1)}
    float *auto_iou;        // [4] per level: mean of the entries' clamped IoU (SlideLoss), 0.5 when the level has none
    int *nent;              // [4] entries per level, zeroed
    int *head;              // [sum cells] last entry pushed onto the cell's list, -1 = none (only with gradients)
    int *next;              // [nl][na*nt*NOFF] list links
    float *partial;         // [sum nblk] obj-BCE partial sums
    double *fold;           // [nl][LOSS_FOLD][2] box / cls sums of a level's slot segments (finish adds them in segment order)
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

__device__ __forceinline__ Dual dsqrt(Dual a) { Dual r; r.v = sqrtf(a.v); const float s = 0.5f / r.v; for (int i = 0; i < 4; ++i) r.g[i] = a.g[i] * s; return r; }
__device__ __forceinline__ Dual dexp(Dual a) { Dual r; r.v = expf(a.v); for (int i = 0; i < 4; ++i) r.g[i] = a.g[i] * r.v; return r; }

// wasserstein_loss (utils/metrics.py:341-354): written for x1y1x2y2 boxes but fed (x, y, w, h) by ComputeLoss (utils/loss.py:166) - the
// columns are used exactly as the reference uses them
__device__ __forceinline__ Dual nwd_xywh(Dual px, Dual py, Dual pw, Dual ph, float tx, float ty, float tw, float th, float constant) {
    const Dual w1 = pw - px, h1 = (ph - py) + NWD_EPS;
    const float w2 = tw - tx, h2 = (th - ty) + NWD_EPS;
    const Dual dcx = (px + pw) * 0.5f + (-(tx + tw) / 2), dcy = (py + ph) * 0.5f + (-(ty + th) / 2);
    const Dual cd = (dcx * dcx + dcy * dcy) + NWD_EPS;
    const Dual dw = w1 + (-w2), dh = h1 + (-h2);
    const Dual whd = (dw * dw + dh * dh) * 0.25f;
    return dexp(dsqrt(cd + whd) * (-1.f / constant));
}

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

// SlideLoss weight of a target value (utils/loss.py:386-395); auto_iou is floored at 0.2 (:387-388)
__device__ __forceinline__ float slide_weight(float y, float auto_iou) {
    const float ai = auto_iou < 0.2f ? 0.2f : auto_iou;
    if (y <= ai - 0.1f) return 1.f;
    if (y < ai) return expf(1.f - ai);
    return expf(-(y - 1.f));
}
// One element of [SlideLoss(][FocalLoss(]BCEWithLogits(pos_weight)[)][)] with reduction 'none' (utils/loss.py:47-53,385-396):
// value and d value / d logit.  gamma <= 0: no focal factor; slide == 0: no slide factor.
__device__ __forceinline__ void weighted_bce(float x, float y, float pw, float gamma, int slide, float auto_iou, float &val, float &grad) {
    float l = bce_logits(x, y, pw), dl = bce_logits_grad(x, y, pw);
    if (gamma > 0.f) {
        const float s = 1.f / (1.f + expf(-x));
        const float q = 1.f - (y * s + (1.f - y) * (1.f - s));                              // 1 - p_t
        const float af = y * FOCAL_ALPHA + (1.f - y) * (1.f - FOCAL_ALPHA);
        const float m = q > 0.f ? powf(q, gamma) : 0.f;
        const float dm = q > 0.f ? gamma * powf(q, gamma - 1.f) * (-(2.f * y - 1.f) * s * (1.f - s)) : 0.f;
        dl = af * (dl * m + l * dm);
        l = l * af * m;
    }
    if (slide) {
        const float w = slide_weight(y, auto_iou);
        l *= w;
        dl *= w;
    }
    val = l;
    grad = dl;
}

// ------------------------------------------------------------------------------------------------ 1. match
// One (level, anchor, target) triple: the anchor-ratio test and the neighbour-cell offsets of build_targets (utils/loss.py:233-245)
struct Triple {
    bool ok;
    int b, cls, nx, ny;
    float gx, gy, gw, gh, aw, ah;
    bool use[NOFF];
};
__device__ __forceinline__ Triple load_triple(const somi_loss_desc &d, int l, int an, int t) {
    Triple r;
    r.nx = d.nx[l];
    r.ny = d.ny[l];
    const float *tg = d.targets + (size_t)t * 6;
    r.b = (int)tg[0];
    r.cls = (int)tg[1];
    r.gx = tg[2] * (float)r.nx; r.gy = tg[3] * (float)r.ny; r.gw = tg[4] * (float)r.nx; r.gh = tg[5] * (float)r.ny;
    r.aw = d.anchors[(l * d.na + an) * 2];
    r.ah = d.anchors[(l * d.na + an) * 2 + 1];
    const float rw = r.gw / r.aw, rh = r.gh / r.ah;
    r.ok = fmaxf(fmaxf(rw, 1.f / rw), fmaxf(rh, 1.f / rh)) < d.anchor_t;                  // utils/loss.py:233-235
    const float g = 0.5f;
    const float gxi = (float)r.nx - r.gx, gyi = (float)r.ny - r.gy;
    r.use[0] = true;
    r.use[1] = (fmodf(r.gx, 1.f) < g) && r.gx > 1.f;
    r.use[2] = (fmodf(r.gy, 1.f) < g) && r.gy > 1.f;
    r.use[3] = (fmodf(gxi, 1.f) < g) && gxi > 1.f;
    r.use[4] = (fmodf(gyi, 1.f) < g) && gyi > 1.f;
    return r;
}
// One entry (triple, offset k): its cell, the decoded box, CIoU with the derivatives w.r.t. the four box logits
struct Entry {
    size_t cell;
    Dual c;                 // the box term's similarity: CIoU, or (1-r) CIoU + r NWD with the NWD branch on; box loss = 1 - c.v
    float iou01;            // what goes to the objectness target / SlideLoss mean: clamp(c.v, 0, 1)
    float s0, s1, s2, s3;
};
__device__ __forceinline__ Entry eval_entry(const somi_loss_desc &d, int no, const float *pl, const Triple &r, int an, int k) {
    const float offx[NOFF] = {0.f, 0.5f, 0.f, -0.5f, 0.f}, offy[NOFF] = {0.f, 0.f, 0.5f, 0.f, -0.5f};
    int gi = (int)(r.gx - offx[k]), gj = (int)(r.gy - offy[k]);                           // .long(): truncation
    gi = min(max(gi, 0), r.nx - 1);                                                       // clamp_ (also feeds tbox)
    gj = min(max(gj, 0), r.ny - 1);
    const float tbx = r.gx - (float)gi, tby = r.gy - (float)gj;
    Entry e;
    e.cell = (((size_t)r.b * d.na + an) * r.ny + gj) * r.nx + gi;
    const float *ps = pl + e.cell * no;
    e.s0 = 1.f / (1.f + expf(-ps[0])); e.s1 = 1.f / (1.f + expf(-ps[1]));
    e.s2 = 1.f / (1.f + expf(-ps[2])); e.s3 = 1.f / (1.f + expf(-ps[3]));
    const float pxv = e.s0 * 2.f - 0.5f, pyv = e.s1 * 2.f - 0.5f;
    const float pwv = (e.s2 * 2.f) * (e.s2 * 2.f) * r.aw, phv = (e.s3 * 2.f) * (e.s3 * 2.f) * r.ah;
    e.c = ciou_xywh(dvar(pxv, 0), dvar(pyv, 1), dvar(pwv, 2), dvar(phv, 3), tbx, tby, r.gw, r.gh);
    if (d.nwd_ratio > 0.f) {                                                              // utils/loss.py:162-169
        const Dual w = nwd_xywh(dvar(pxv, 0), dvar(pyv, 1), dvar(pwv, 2), dvar(phv, 3), tbx, tby, r.gw, r.gh, d.nwd_constant);
        e.c = e.c * (1.f - d.nwd_ratio) + w * d.nwd_ratio;
    }
    e.iou01 = fminf(fmaxf(e.c.v, 0.f), 1.f);
    return e;
}

__global__ __launch_bounds__(256) void loss_match_kernel(const LossArgs a) {
    const somi_loss_desc &d = a.d;
    const int l = blockIdx.y;
    const int per_level = d.na * d.nt;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= per_level) return;
    const int an = idx / d.nt, t = idx % d.nt;                 // anchor-major like targets.repeat(na,1,1)
    float *slot = a.slots + ((size_t)l * per_level + idx) * NOFF * SLOT;
    for (int k = 0; k < NOFF; ++k) slot[k * SLOT] = -1.f;      // invalid
    const Triple r = load_triple(d, l, an, t);
    if (!r.ok) return;
    const float *pl = d.p[l];
    int emitted = 0;
    for (int k = 0; k < NOFF; ++k) {
        if (!r.use[k]) continue;
        const Entry e = eval_entry(d, a.no, pl, r, an, k);
        slot[k * SLOT] = 1.f - e.c.v;
        slot[k * SLOT + 1] = 0.f;
        slot[k * SLOT + 2] = e.iou01;
        atomicMax(a.tobj + a.cell_off[l] + e.cell, __float_as_uint((1.f - d.gr) + d.gr * e.iou01));
        if (d.grad[l]) {                                        // push onto the cell's list; 1b sums the list in a fixed order
            const int ent = ((int)l * per_level + idx) * NOFF + k;
            a.next[ent] = atomicExch(a.head + a.cell_off[l] + e.cell, ent);
        }
        ++emitted;
    }
    if (emitted) atomicAdd(a.nent + l, emitted);
}

// ------------------------------------------------------------------------------------------------ 1a. per-level mean IoU (SlideLoss)
// auto_iou = iou.mean() over the level's entries (utils/loss.py:180); 0.5 - SlideLoss's default - for a level without entries (:191-194)
__global__ __launch_bounds__(256) void loss_level_mean_kernel(const LossArgs a) {
    __shared__ double red[256];
    const somi_loss_desc &d = a.d;
    const int l = blockIdx.x;
    const int per_level = d.na * d.nt * NOFF;
    const float *sl = a.slots + (size_t)l * per_level * SLOT;
    double s = 0.0;
    for (int i = threadIdx.x; i < per_level; i += 256)
        if (sl[i * SLOT] >= 0.f) s += sl[i * SLOT + 2];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) a.auto_iou[l] = a.nent[l] ? (float)(red[0] / a.nent[l]) : 0.5f;
}

// ------------------------------------------------------------------------------------------------ 1b. gradient of the matched entries
__global__ __launch_bounds__(256) void loss_scatter_kernel(const LossArgs a) {
    const somi_loss_desc &d = a.d;
    const int l = blockIdx.y;
    const int per_level = d.na * d.nt;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    float *gl = d.grad[l];
    if (idx >= per_level) return;
    const int an = idx / d.nt, t = idx % d.nt;
    const Triple r = load_triple(d, l, an, t);
    if (!r.ok) return;
    const float *pl = d.p[l];
    const int base = (int)l * per_level * NOFF;
    const float ai = d.slide ? a.auto_iou[l] : 0.5f;
    float *slot = a.slots + ((size_t)l * per_level + idx) * NOFF * SLOT;
    for (int k = 0; k < NOFF; ++k) {
        if (!r.use[k]) continue;
        const Entry own = eval_entry(d, a.no, pl, r, an, k);
        const float *ps = pl + own.cell * a.no;
        if (d.nc > 1) {                                          // this entry's class BCE (utils/loss.py:182-189)
            float csum = 0.f;
            for (int j = 0; j < d.nc; ++j) {
                float v, g_;
                weighted_bce(ps[5 + j], j == r.cls ? d.cp : d.cn, d.cls_pw, d.fl_gamma, d.slide, ai, v, g_);
                csum += v;
            }
            slot[k * SLOT + 1] = csum;
        }
        if (!gl) continue;
        const int ent = base + idx * NOFF + k;
        const int *head = a.head + a.cell_off[l] + own.cell;
        if (*head != ent) continue;                              // exactly one entry per hit cell heads its list
        float *g = gl + own.cell * a.no;
        float gb[4] = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < d.nc && d.nc > 1; ++j) g[5 + j] = 0.f;
        // ascending entry order: repeatedly take the smallest list member above the last one (lists hold 1-3 entries)
        for (int last = -1;;) {
            int m = 0x7fffffff;
            for (int q = *head; q >= 0; q = a.next[q])
                if (q > last && q < m) m = q;
            if (m == 0x7fffffff) break;
            last = m;
            const int rel = m - base, k2 = rel % NOFF, idx2 = rel / NOFF;
            const int an2 = idx2 / d.nt, t2 = idx2 % d.nt;       // same level, same anchor (the cell index contains it)
            const Triple r2 = load_triple(d, l, an2, t2);
            const Entry e = eval_entry(d, a.no, pl, r2, an2, k2);
            gb[0] += -e.c.g[0] * 2.f * e.s0 * (1.f - e.s0);     // d(1 - similarity)/d logits, unscaled
            gb[1] += -e.c.g[1] * 2.f * e.s1 * (1.f - e.s1);
            gb[2] += -e.c.g[2] * 8.f * e.s2 * e.s2 * (1.f - e.s2) * r2.aw;
            gb[3] += -e.c.g[3] * 8.f * e.s3 * e.s3 * (1.f - e.s3) * r2.ah;
            if (d.nc > 1) {
                for (int j = 0; j < d.nc; ++j) {
                    float v, g_;
                    weighted_bce(ps[5 + j], j == r2.cls ? d.cp : d.cn, d.cls_pw, d.fl_gamma, d.slide, ai, v, g_);
                    g[5 + j] += g_;
                }
            }
        }
        g[0] = gb[0]; g[1] = gb[1]; g[2] = gb[2]; g[3] = gb[3];
    }
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
    const float ai = d.slide ? a.auto_iou[l] : 0.5f;               // 0.5 (SlideLoss's default) for a level without entries
    float part = 0.f;
    if (gl) {
        const long items = cells * a.no;
        for (long it = blockIdx.x * 256L + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
            const int ch = (int)(it % a.no);
            if (ch == 4) {
                const float x = pl[it], y = __uint_as_float(a.tobj[a.cell_off[l] + it / a.no]);
                float v, g_;
                weighted_bce(x, y, d.obj_pw, d.fl_gamma, d.slide, ai, v, g_);
                part += v;
                gl[it] = g_ * obj_scale;
            } else {
                const float gv = gl[it];
                if (gv != 0.f) gl[it] = gv * (ch < 4 ? box_scale : cls_scale);
            }
        }
    } else {
        for (long cidx = blockIdx.x * 256L + threadIdx.x; cidx < cells; cidx += (long)gridDim.x * 256) {
            float v, g_;
            weighted_bce(pl[cidx * a.no + 4], __uint_as_float(a.tobj[a.cell_off[l] + cidx]), d.obj_pw, d.fl_gamma, d.slide, ai, v, g_);
            part += v;
        }
    }
    for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) a.partial[a.blk_off[l] + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ------------------------------------------------------------------------------------------------ 4. finish
// grid (LOSS_FOLD, nl): box / cls sums of one segment of a level's slots - thread-strided partial sums in double, then a tree over the 256 lanes
// (fixed order).  Round 4: the finish kernel did this walk alone, one workgroup for every level: 227 us per step.
__global__ __launch_bounds__(256) void loss_slots_fold_kernel(const LossArgs a) {
    __shared__ double red[2][256];
    const somi_loss_desc &d = a.d;
    const int l = blockIdx.y, per_level = d.na * d.nt * NOFF;
    const int per = (per_level + LOSS_FOLD - 1) / LOSS_FOLD, i0 = blockIdx.x * per, i1 = min(i0 + per, per_level);
    const float *sl = a.slots + (size_t)l * per_level * SLOT;
    double sb = 0.0, sc = 0.0;
    for (int i = i0 + threadIdx.x; i < i1; i += 4 * 256) {
        float vb[4], vc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = i + u * 256;
            vb[u] = j < i1 ? sl[(size_t)j * SLOT] : -1.f;
            vc[u] = j < i1 ? sl[(size_t)j * SLOT + 1] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (vb[u] >= 0.f) { sb += vb[u]; sc += vc[u]; }
    }
    red[0][threadIdx.x] = sb;
    red[1][threadIdx.x] = sc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) { red[0][threadIdx.x] += red[0][threadIdx.x + s]; red[1][threadIdx.x] += red[1][threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        a.fold[((size_t)l * LOSS_FOLD + blockIdx.x) * 2] = red[0][0];
        a.fold[((size_t)l * LOSS_FOLD + blockIdx.x) * 2 + 1] = red[1][0];
    }
}
__global__ __launch_bounds__(256) void loss_finish_kernel(const LossArgs a, float *out4) {
    __shared__ double red[256];
    const somi_loss_desc &d = a.d;
    double lbox = 0.0, lobj = 0.0, lcls = 0.0;
    for (int l = 0; l < d.nl; ++l) {
        double so = 0.0;
        for (int i = threadIdx.x; i < a.nblk[l]; i += 256) so += a.partial[a.blk_off[l] + i];
        red[threadIdx.x] = so;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        double vals[3] = {0.0, 0.0, red[0]};
        __syncthreads();
        if (d.nt > 0)
            for (int g = 0; g < LOSS_FOLD; ++g) {                         // the segments in order; every thread computes the same sums
                vals[0] += a.fold[((size_t)l * LOSS_FOLD + g) * 2];
                vals[1] += a.fold[((size_t)l * LOSS_FOLD + g) * 2 + 1];
            }
        const int n = a.nent[l];
        if (n) {
            lbox += vals[0] / n;
            if (d.nc > 1) lcls += vals[1] / ((double)n * d.nc);
        }
        lobj += vals[2] / (double)a.cells[l] * d.balance[l];
        if (threadIdx.x == 0) out4[4 + l] = (float)(vals[2] / (double)a.cells[l]);   // obji of the level: what autobalance feeds on
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
    const size_t slots = align_up((size_t)d->nl * d->na * (d->nt > 0 ? d->nt : 1) * NOFF * SLOT * 4, 256);
    const size_t part = align_up((size_t)(a.blk_off[d->nl - 1] + a.nblk[d->nl - 1]) * 4, 256);
    const size_t next = align_up((size_t)d->nl * d->na * (d->nt > 0 ? d->nt : 1) * NOFF * 4, 256);
    return tobj + 256 + slots + part + tobj + next + 4 * LOSS_FOLD * 2 * sizeof(double);      // + the per-cell list heads, the entry links of the gradient pass, the slot-segment sums
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
    a.nent = reinterpret_cast<int *>(w);
    a.auto_iou = reinterpret_cast<float *>(w + 64); w += 256;          // same zeroed 256-byte block: counters, then the level means
    a.slots = reinterpret_cast<float *>(w); w += align_up((size_t)d.nl * d.na * (d.nt > 0 ? d.nt : 1) * NOFF * SLOT * 4, 256);
    a.partial = reinterpret_cast<float *>(w); w += align_up((size_t)(a.blk_off[d.nl - 1] + a.nblk[d.nl - 1]) * 4, 256);
    a.head = reinterpret_cast<int *>(w); w += tobj_b;
    a.next = reinterpret_cast<int *>(w); w += align_up((size_t)d.nl * d.na * (d.nt > 0 ? d.nt : 1) * NOFF * 4, 256);
    a.fold = reinterpret_cast<double *>(w);
    bool any_grad = false;
    for (int l = 0; l < d.nl; ++l) any_grad = any_grad || d.grad[l];
    (void)hipMemsetAsync(a.tobj, 0, tobj_b + 256, s);                       // target maps + entry counters
    if (any_grad && d.nt > 0) (void)hipMemsetAsync(a.head, 0xFF, tobj_b, s);   // -1: empty lists
    for (int l = 0; l < d.nl; ++l)
        if (d.grad[l]) (void)hipMemsetAsync(d.grad[l], 0, (size_t)a.cells[l] * a.no * 4, s);
    SOMI_REQUIRE(d.fl_gamma >= 0.f && d.nwd_ratio >= 0.f && d.nwd_ratio <= 1.f && (d.nwd_ratio == 0.f || d.nwd_constant > 0.f), SOMI_EINVAL,
                 "loss: bad focal gamma / NWD ratio / NWD constant");
    if (d.nt > 0) {
        hipLaunchKernelGGL(loss_match_kernel, dim3(cdiv((long)d.na * d.nt, 256), d.nl), dim3(256), 0, s, a);
        if (d.slide) hipLaunchKernelGGL(loss_level_mean_kernel, dim3(d.nl), dim3(256), 0, s, a);
        if (any_grad || d.nc > 1) hipLaunchKernelGGL(loss_scatter_kernel, dim3(cdiv((long)d.na * d.nt, 256), d.nl), dim3(256), 0, s, a);
    } else if (d.slide) {
        hipLaunchKernelGGL(loss_level_mean_kernel, dim3(d.nl), dim3(256), 0, s, a);     // no entries anywhere: every level gets the default
    }
    for (int l = 0; l < d.nl; ++l) hipLaunchKernelGGL(loss_dense_kernel, dim3(a.nblk[l]), dim3(256), 0, s, a, l);
    if (d.nt > 0) hipLaunchKernelGGL(loss_slots_fold_kernel, dim3(LOSS_FOLD, d.nl), dim3(256), 0, s, a);
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(256), 0, s, a, out4);
    return launch_status("somi_yolo_loss_f32");
}
