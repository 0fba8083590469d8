// Weighted boxes fusion for one image on gfx950 (wbf.py:68 -> ensemble_boxes.weighted_boxes_fusion, conf_type 'avg',
// allows_overflow False).  The published algorithm is sequential per label (every box is matched against the clusters
// built so far), so one workgroup owns an image: lanes parallelise the per-box work (prefilter, ranks, cluster scan),
// waves take labels in turn.  Arithmetic follows the package bit for bit: member boxes and scores in float64, the
// running fused box in a float32 accumulator re-rounded after every member (its accumulator is a float32 numpy array),
// IoU in float64, ties in both sorts resolved as `argsort(stable)[::-1]` does (later element first).
#include "common.h"

namespace somi {

constexpr int WBF_MAX_LABELS = 64;

struct WbfArgs {
    const float *boxes, *scores;
    const int32_t *labels, *model;
    int n, n_models;
    float weights[16];
    float iou_thr, skip_thr;
    float *out_boxes, *out_scores;
    int32_t *out_labels, *out_count;
    // workspace (n entries each)
    double *mx1, *my1, *mx2, *my2, *msc, *mwt;     // prefiltered members
    int *mlab, *mvalid, *order;                    // order: member indices grouped by label, each group sorted
    // clusters (<= n)
    double *cx1, *cy1, *cx2, *cy2, *cscore, *cconf, *cwsum;
    float *cacc;                                   // [n][4] float32 accumulators of sum(score * coord)
    int *ccnt, *clab;
    int *final_rank;
};

__device__ __forceinline__ double clip01(double v) { return v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v); }

__device__ __forceinline__ void wbf_image(const WbfArgs &a) {
    __shared__ int first_idx[WBF_MAX_LABELS];      // first appearance (member index) of each label, or INT_MAX
    __shared__ int lab_cnt[WBF_MAX_LABELS];
    __shared__ int lab_seq[WBF_MAX_LABELS];        // labels in first-appearance order
    __shared__ int lab_start[WBF_MAX_LABELS];      // start of each label's group in `order` (indexed by label)
    __shared__ int clu_start[WBF_MAX_LABELS];      // start of each label's clusters (indexed by label) == lab_start
    __shared__ int clu_cnt[WBF_MAX_LABELS];
    __shared__ int nlab_sh, ntot_sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = a.n;
    double wsum_all = 0.0;
    for (int t = 0; t < a.n_models; ++t) wsum_all += (double)a.weights[t];
    for (int i = tid; i < WBF_MAX_LABELS; i += 256) { first_idx[i] = 0x7FFFFFFF; lab_cnt[i] = 0; clu_cnt[i] = 0; }
    __syncthreads();
    // ---- A. prefilter (ensemble_boxes: score < thr dropped, corners ordered and clipped, zero area dropped)
    for (int i = tid; i < n; i += 256) {
        const float sc = a.scores[i];
        int ok = !(sc < a.skip_thr);
        double x1 = a.boxes[i * 4], y1 = a.boxes[i * 4 + 1], x2 = a.boxes[i * 4 + 2], y2 = a.boxes[i * 4 + 3];
        if (x2 < x1) { const double t = x1; x1 = x2; x2 = t; }
        if (y2 < y1) { const double t = y1; y1 = y2; y2 = t; }
        x1 = clip01(x1); y1 = clip01(y1); x2 = clip01(x2); y2 = clip01(y2);
        if ((x2 - x1) * (y2 - y1) == 0.0) ok = 0;
        const int lab = a.labels[i];
        if (lab < 0 || lab >= WBF_MAX_LABELS) ok = 0;
        const double w = (double)a.weights[a.model[i]];
        a.mx1[i] = x1; a.my1[i] = y1; a.mx2[i] = x2; a.my2[i] = y2;
        a.msc[i] = (double)sc * w; a.mwt[i] = w; a.mlab[i] = lab; a.mvalid[i] = ok;
        if (ok) { atomicMin(&first_idx[lab], i); atomicAdd(&lab_cnt[lab], 1); }
    }
    __syncthreads();
    // ---- B. label sequence in first-appearance order (dict insertion order of the package), group offsets
    if (tid == 0) {
        int nl = 0;
        for (int l = 0; l < WBF_MAX_LABELS; ++l) if (lab_cnt[l]) lab_seq[nl++] = l;
        for (int i = 1; i < nl; ++i) {               // insertion sort by first index (<= 64 labels)
            const int v = lab_seq[i];
            int j = i - 1;
            while (j >= 0 && first_idx[lab_seq[j]] > first_idx[v]) { lab_seq[j + 1] = lab_seq[j]; --j; }
            lab_seq[j + 1] = v;
        }
        int run = 0;
        for (int i = 0; i < nl; ++i) { lab_start[lab_seq[i]] = run; clu_start[lab_seq[i]] = run; run += lab_cnt[lab_seq[i]]; }
        nlab_sh = nl;
    }
    __syncthreads();
    // ---- C. rank inside the label: descending score, ties -> later member first (argsort(stable)[::-1])
    for (int i = tid; i < n; i += 256) {
        if (!a.mvalid[i]) continue;
        const int lab = a.mlab[i];
        const double s = a.msc[i];
        int rank = 0;
        for (int j = 0; j < n; ++j)
            if (a.mvalid[j] && a.mlab[j] == lab && (a.msc[j] > s || (a.msc[j] == s && j > i))) ++rank;
        a.order[lab_start[lab] + rank] = i;
    }
    __syncthreads();
    // ---- D. clustering: one wave per label, lanes scan that label's clusters
    const int nlab = nlab_sh;
    for (int li = wave; li < nlab; li += 4) {
        const int lab = lab_seq[li];
        const int g0 = lab_start[lab], gn = lab_cnt[lab], c0 = clu_start[lab];
        int nc = 0;
        for (int q = 0; q < gn; ++q) {
            const int m = a.order[g0 + q];
            const double bx1 = a.mx1[m], by1 = a.my1[m], bx2 = a.mx2[m], by2 = a.my2[m], bs = a.msc[m];
            const double area_b = (bx2 - bx1) * (by2 - by1);
            // best matching fused box: highest IoU, first index on ties (np.argmax)
            double best = -1.0;
            int bidx = -1;
            for (int c = lane; c < nc; c += 64) {
                const int k = c0 + c;
                const double xa = fmax(a.cx1[k], bx1), ya = fmax(a.cy1[k], by1);
                const double xb = fmin(a.cx2[k], bx2), yb = fmin(a.cy2[k], by2);
                const double inter = fmax(xb - xa, 0.0) * fmax(yb - ya, 0.0);
                const double area_a = (a.cx2[k] - a.cx1[k]) * (a.cy2[k] - a.cy1[k]);
                const double iou = inter / (area_a + area_b - inter);
                if (iou > best) { best = iou; bidx = c; }       // strided scan keeps the lowest c per lane on ties
            }
            for (int o = 32; o > 0; o >>= 1) {
                const double ob = __shfl_xor(best, o);
                const int oi = __shfl_xor(bidx, o);
                if (oi >= 0 && (bidx < 0 || ob > best || (ob == best && oi < bidx))) { best = ob; bidx = oi; }
            }
            const bool match = bidx >= 0 && best > (double)a.iou_thr;
            if (lane == 0) {
                if (match) {
                    const int k = c0 + bidx;
                    float *acc = a.cacc + (size_t)k * 4;
                    acc[0] = (float)((double)acc[0] + bs * bx1);
                    acc[1] = (float)((double)acc[1] + bs * by1);
                    acc[2] = (float)((double)acc[2] + bs * bx2);
                    acc[3] = (float)((double)acc[3] + bs * by2);
                    const double conf = a.cconf[k] + bs;
                    const int cnt = a.ccnt[k] + 1;
                    a.cconf[k] = conf; a.ccnt[k] = cnt; a.cwsum[k] += a.mwt[m];
                    a.cscore[k] = (double)(float)(conf / cnt);
                    a.cx1[k] = (double)(float)((double)acc[0] / conf);
                    a.cy1[k] = (double)(float)((double)acc[1] / conf);
                    a.cx2[k] = (double)(float)((double)acc[2] / conf);
                    a.cy2[k] = (double)(float)((double)acc[3] / conf);
                } else {
                    const int k = c0 + nc;
                    a.clab[k] = lab;
                    a.cx1[k] = bx1; a.cy1[k] = by1; a.cx2[k] = bx2; a.cy2[k] = by2;   // a lone member is kept exactly
                    a.cscore[k] = bs; a.cconf[k] = bs; a.ccnt[k] = 1; a.cwsum[k] = a.mwt[m];
                    float *acc = a.cacc + (size_t)k * 4;
                    acc[0] = (float)(bs * bx1); acc[1] = (float)(bs * by1); acc[2] = (float)(bs * bx2); acc[3] = (float)(bs * by2);
                }
            }
            if (!match) ++nc;
            __threadfence_block();                              // lane 0's update is visible to the wave's next scan
        }
        if (lane == 0) clu_cnt[lab] = nc;
    }
    __syncthreads();
    // ---- E. rescale scores, concatenate labels in sequence order, final descending sort (ties: later first)
    if (tid == 0) {
        int run = 0;
        for (int i = 0; i < nlab; ++i) { const int l = lab_seq[i]; lab_start[l] = run; run += clu_cnt[l]; }   // reuse as output offsets
        ntot_sh = run;
    }
    __syncthreads();
    const int ntot = ntot_sh;
    // flat index f of cluster (label l, local c) = lab_start[l] + c; its storage index = clu_start[l] + c
    for (int li = 0; li < nlab; ++li) {
        const int l = lab_seq[li];
        for (int c = tid; c < clu_cnt[l]; c += 256) {
            const int k = clu_start[l] + c;
            const double cnt = (double)a.ccnt[k];
            a.cscore[k] = a.cscore[k] * fmin(cnt, wsum_all) / wsum_all;
            a.final_rank[lab_start[l] + c] = k;                 // flat position -> storage index
        }
    }
    __syncthreads();
    for (int f = tid; f < ntot; f += 256) {
        const int k = a.final_rank[f];
        const double s = a.cscore[k];
        int rank = 0;
        for (int g = 0; g < ntot; ++g) {
            const double sg = a.cscore[a.final_rank[g]];
            if (sg > s || (sg == s && g > f)) ++rank;
        }
        a.out_boxes[rank * 4] = (float)a.cx1[k];
        a.out_boxes[rank * 4 + 1] = (float)a.cy1[k];
        a.out_boxes[rank * 4 + 2] = (float)a.cx2[k];
        a.out_boxes[rank * 4 + 3] = (float)a.cy2[k];
        a.out_scores[rank] = (float)s;
        a.out_labels[rank] = a.clab[k];
    }
    if (tid == 0) *a.out_count = ntot;
}

__global__ __launch_bounds__(256) void wbf_kernel(const WbfArgs a) { wbf_image(a); }

// Batched form for val.py-style pipelines (BASELINE configs[4]: NMS + WBF over the detections of several models): one workgroup per
// image.  The image's members are the NMS rows of model 0, then model 1, ... (the order wbf.py:44-59 appends them in), boxes scaled
// to [0,1] by the image size and clipped; everything after that is wbf_image() on the image's own slice of the workspace.
struct WbfBatchArgs {
    const float *det[16];        // per model (B, max_det, 6) [x1,y1,x2,y2,conf,cls] in pixels
    const int32_t *count[16];    // per model (B)
    int B, max_det, n_models;
    float img_w, img_h;          // true division like the caller's `boxes / size` (a reciprocal multiply differs in the last bit)
    WbfArgs img;                 // pointers of image 0; image b uses the same arrays at offset b * N (N = n_models * max_det)
    float *in_boxes, *in_scores; // workspace: the gathered members
    int32_t *in_labels, *in_model;
};

__global__ __launch_bounds__(256) void wbf_batch_kernel(const WbfBatchArgs g) {
    const int b = blockIdx.x, N = g.n_models * g.max_det;
    const size_t o = (size_t)b * N;
    float *boxes = g.in_boxes + o * 4, *scores = g.in_scores + o;
    int32_t *labels = g.in_labels + o, *model = g.in_model + o;
    int n = 0;
    for (int t = 0; t < g.n_models; ++t) {
        int c = g.count[t][b];
        c = c < 0 ? 0 : (c > g.max_det ? g.max_det : c);
        const float *d = g.det[t] + (size_t)b * g.max_det * 6;
        for (int i = threadIdx.x; i < c; i += 256) {
            boxes[(n + i) * 4 + 0] = fminf(fmaxf(d[i * 6 + 0] / g.img_w, 0.f), 1.f);
            boxes[(n + i) * 4 + 1] = fminf(fmaxf(d[i * 6 + 1] / g.img_h, 0.f), 1.f);
            boxes[(n + i) * 4 + 2] = fminf(fmaxf(d[i * 6 + 2] / g.img_w, 0.f), 1.f);
            boxes[(n + i) * 4 + 3] = fminf(fmaxf(d[i * 6 + 3] / g.img_h, 0.f), 1.f);
            scores[n + i] = d[i * 6 + 4];
            labels[n + i] = (int32_t)d[i * 6 + 5];
            model[n + i] = t;
        }
        n += c;
    }
    __syncthreads();
    WbfArgs a = g.img;
    a.boxes = boxes; a.scores = scores; a.labels = labels; a.model = model; a.n = n;
    a.out_boxes += o * 4; a.out_scores += o; a.out_labels += o; a.out_count += b;
    a.mx1 += o; a.my1 += o; a.mx2 += o; a.my2 += o; a.msc += o; a.mwt += o;
    a.mlab += o; a.mvalid += o; a.order += o;
    a.cx1 += o; a.cy1 += o; a.cx2 += o; a.cy2 += o; a.cscore += o; a.cconf += o; a.cwsum += o;
    a.cacc += o * 4; a.ccnt += o; a.clab += o; a.final_rank += o;
    wbf_image(a);
}

static size_t al8(size_t v) { return (v + 255) / 256 * 256; }

// carve the per-member / per-cluster arrays (N entries each) out of `w`
static char *carve(WbfArgs &a, char *w, size_t N) {
    auto take = [&](size_t bytes) { char *p = w; w += al8(bytes); return p; };
    double **d64[] = {&a.mx1, &a.my1, &a.mx2, &a.my2, &a.msc, &a.mwt, &a.cx1, &a.cy1, &a.cx2, &a.cy2, &a.cscore, &a.cconf, &a.cwsum};
    for (auto pp : d64) *pp = reinterpret_cast<double *>(take(N * 8));
    int **i32[] = {&a.mlab, &a.mvalid, &a.order, &a.ccnt, &a.clab, &a.final_rank};
    for (auto pp : i32) *pp = reinterpret_cast<int *>(take(N * 4));
    a.cacc = reinterpret_cast<float *>(take(N * 16));
    return w;
}

}  // namespace somi

using namespace somi;

extern "C" size_t somi_wbf_workspace_bytes(int n) {
    if (n <= 0) return 256;
    const size_t N = (size_t)n;
    return al8(N * 8) * 13 + al8(N * 4) * 6 + al8(N * 16) + 256;
}

extern "C" int somi_wbf_f32(const float *boxes, const float *scores, const int32_t *labels, const int32_t *model, int n,
                            int n_models, const float *weights_host, float iou_thr, float skip_box_thr, float *out_boxes,
                            float *out_scores, int32_t *out_labels, int32_t *out_count, void *workspace, size_t workspace_bytes,
                            somi_stream_t stream) {
    SOMI_REQUIRE(out_count && workspace, SOMI_EINVAL, "wbf: null argument");
    SOMI_REQUIRE(n >= 0 && n_models >= 1 && n_models <= 16, SOMI_EINVAL, "wbf: bad sizes (n_models <= 16)");
    SOMI_REQUIRE(n == 0 || (boxes && scores && labels && model && out_boxes && out_scores && out_labels), SOMI_EINVAL,
                 "wbf: null tensor");
    SOMI_REQUIRE(workspace_bytes >= somi_wbf_workspace_bytes(n), SOMI_EWORKSPACE, "wbf: workspace too small");
    WbfArgs a{};
    a.boxes = boxes; a.scores = scores; a.labels = labels; a.model = model; a.n = n; a.n_models = n_models;
    for (int i = 0; i < 16; ++i) a.weights[i] = i < n_models ? (weights_host ? weights_host[i] : 1.f) : 0.f;
    a.iou_thr = iou_thr; a.skip_thr = skip_box_thr;
    a.out_boxes = out_boxes; a.out_scores = out_scores; a.out_labels = out_labels; a.out_count = out_count;
    carve(a, static_cast<char *>(workspace), (size_t)(n > 0 ? n : 1));
    hipLaunchKernelGGL(wbf_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return launch_status("somi_wbf_f32");
}

extern "C" size_t somi_wbf_batch_workspace_bytes(int B, int max_det, int n_models) {
    if (B <= 0 || max_det <= 0 || n_models <= 0) return 256;
    const size_t T = (size_t)B * max_det * n_models;
    return somi_wbf_workspace_bytes((int)T) + al8(T * 16) + 3 * al8(T * 4);
}

extern "C" int somi_wbf_batch_f32(const float *const *det, const int32_t *const *count, int B, int max_det, int n_models,
                                  const float *weights_host, float img_w, float img_h, float iou_thr, float skip_box_thr,
                                  float *out_boxes, float *out_scores, int32_t *out_labels, int32_t *out_count, void *workspace,
                                  size_t workspace_bytes, somi_stream_t stream) {
    SOMI_REQUIRE(det && count && out_boxes && out_scores && out_labels && out_count && workspace, SOMI_EINVAL, "wbf batch: null argument");
    SOMI_REQUIRE(B > 0 && max_det > 0 && n_models >= 1 && n_models <= 16 && img_w > 0.f && img_h > 0.f, SOMI_EINVAL,
                 "wbf batch: bad sizes (n_models <= 16)");
    SOMI_REQUIRE((size_t)B * max_det * n_models < (1u << 30), SOMI_EINVAL, "wbf batch: too many boxes");
    SOMI_REQUIRE(workspace_bytes >= somi_wbf_batch_workspace_bytes(B, max_det, n_models), SOMI_EWORKSPACE, "wbf batch: workspace too small");
    WbfBatchArgs g{};
    for (int t = 0; t < n_models; ++t) {
        SOMI_REQUIRE(det[t] && count[t], SOMI_EINVAL, "wbf batch: null tensor of model %d", t);
        g.det[t] = det[t];
        g.count[t] = count[t];
    }
    g.B = B; g.max_det = max_det; g.n_models = n_models;
    g.img_w = img_w; g.img_h = img_h;
    const size_t T = (size_t)B * max_det * n_models;
    char *w = carve(g.img, static_cast<char *>(workspace), T);
    g.in_boxes = reinterpret_cast<float *>(w); w += al8(T * 16);
    g.in_scores = reinterpret_cast<float *>(w); w += al8(T * 4);
    g.in_labels = reinterpret_cast<int32_t *>(w); w += al8(T * 4);
    g.in_model = reinterpret_cast<int32_t *>(w);
    g.img.n_models = n_models;
    for (int i = 0; i < 16; ++i) g.img.weights[i] = i < n_models ? (weights_host ? weights_host[i] : 1.f) : 0.f;
    g.img.iou_thr = iou_thr; g.img.skip_thr = skip_box_thr;
    g.img.out_boxes = out_boxes; g.img.out_scores = out_scores; g.img.out_labels = out_labels; g.img.out_count = out_count;
    hipLaunchKernelGGL(wbf_batch_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream), g);
    return launch_status("somi_wbf_batch_f32");
}
