// Validation metrics after NMS, on the device: val.py:50-71 `process_batch` (which detections are
// correct at each IoU level) and utils/metrics.py:21-95 `ap_per_class` / `compute_ap` (precision / recall curves, 101-point
// interpolated AP, the best-F1 operating point).  The reference does both on the CPU with numpy after a device->host copy per
// image (val.py:189); here the detections never leave HBM.
//
// Arithmetic: IoU in fp32 in torch's order (bit-identical decisions against the IoU levels); curves and AP in fp64 like numpy.
// Ties: the reference sorts with unstable numpy sorts; exact ties (equal IoU of one detection with two labels, equal
// confidences) are broken here by lowest index - see oracle/somi_ref/metrics.py.
#include "common.h"

namespace somi {

// ------------------------------------------------------------------------------------------------ process_batch
// one workgroup per image.  LDS: labels [M][5], best label per (level, detection) [T][Ncap], lowest detection per (level, label) [T][Mcap]
__global__ __launch_bounds__(256) void val_match_kernel(const float *__restrict__ det, const int *__restrict__ det_off,
                                                        const float *__restrict__ lab, const int *__restrict__ lab_off,
                                                        const float *__restrict__ iouv, int T, int Ncap, int Mcap, uint8_t *__restrict__ correct) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *s_lab = reinterpret_cast<float *>(smem);                       // [Mcap][5]
    int *s_best = reinterpret_cast<int *>(s_lab + (size_t)Mcap * 5);      // [T][Ncap]
    int *s_min = s_best + (size_t)T * Ncap;                               // [T][Mcap]
    const int b = blockIdx.x, tid = threadIdx.x;
    const int d0 = det_off[b], N = det_off[b + 1] - d0, l0 = lab_off[b], M = lab_off[b + 1] - l0;
    if (N <= 0) return;
    if (M <= 0) {                                                         // val.py:188: nothing can be correct
        for (int i = tid; i < N * T; i += 256) correct[(size_t)d0 * T + i] = 0;
        return;
    }
    for (int i = tid; i < M * 5; i += 256) s_lab[i] = lab[(size_t)l0 * 5 + i];
    for (int i = tid; i < T * M; i += 256) s_min[(i / M) * Mcap + i % M] = 0x7fffffff;
    __syncthreads();
    constexpr int TMAX = 16;
    for (int d = tid; d < N; d += 256) {
        const float *dp = det + (size_t)(d0 + d) * 6;
        const float x1 = dp[0], y1 = dp[1], x2 = dp[2], y2 = dp[3], cls = dp[5];
        const float area2 = __fmul_rn(x2 - x1, y2 - y1);
        float bi[TMAX];
        int bl[TMAX];
#pragma unroll
        for (int i = 0; i < TMAX; ++i) { bi[i] = -1.f; bl[i] = -1; }
        for (int l = 0; l < M; ++l) {
            const float *lp = s_lab + l * 5;
            if (lp[0] != cls) continue;
            // utils/metrics.py:232-235 (box1 = labels, box2 = detections)
            const float iw = fmaxf(fminf(lp[3], x2) - fmaxf(lp[1], x1), 0.f), ih = fmaxf(fminf(lp[4], y2) - fmaxf(lp[2], y1), 0.f);
            const float inter = __fmul_rn(iw, ih);
            const float area1 = __fmul_rn(lp[3] - lp[1], lp[4] - lp[2]);
            const float iou = __fdiv_rn(inter, (area1 + area2) - inter);
#pragma unroll
            for (int i = 0; i < TMAX; ++i)
                if (i < T && iou >= iouv[i] && iou > bi[i]) { bi[i] = iou; bl[i] = l; }
        }
#pragma unroll
        for (int i = 0; i < TMAX; ++i)
            if (i < T) {
                s_best[i * Ncap + d] = bl[i];
                if (bl[i] >= 0) atomicMin(&s_min[i * Mcap + bl[i]], d);
            }
    }
    __syncthreads();
    for (int e = tid; e < N * T; e += 256) {
        const int d = e / T, i = e % T;
        const int l = s_best[i * Ncap + d];
        correct[(size_t)(d0 + d) * T + i] = (l >= 0 && s_min[i * Mcap + l] == d) ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------------------ confusion matrix
// utils/metrics.py:98-142 for a batch of images, one workgroup per image.  Class-agnostic matching: every detection above `conf`
// takes the label of highest IoU (> iou_thres; ties: lowest label), every label keeps the detection of highest IoU among those
// that took it (ties: lowest detection).  LDS: labels [Mcap][5], the label each detection took [Ncap], and per label a 64-bit
// key (IoU bits << 32 | ~detection) maximised atomically.  Counts go to the global int32 matrix [pred][true] with integer atomics.
__global__ __launch_bounds__(256) void confusion_kernel(const float *__restrict__ det, const int *__restrict__ det_off,
                                                        const float *__restrict__ lab, const int *__restrict__ lab_off, int Ncap, int Mcap,
                                                        int nc, float conf, float iou_thres, int *__restrict__ matrix) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long *s_key = reinterpret_cast<unsigned long long *>(smem);          // [Mcap]
    float *s_lab = reinterpret_cast<float *>(s_key + Mcap);                            // [Mcap][5]
    int *s_took = reinterpret_cast<int *>(s_lab + (size_t)Mcap * 5);                   // [Ncap]
    __shared__ int s_any;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int d0 = det_off[b], N = det_off[b + 1] - d0, l0 = lab_off[b], M = lab_off[b + 1] - l0;
    if (M <= 0) return;                                                               // no label: nothing is counted (:132,138)
    const int W = nc + 1;
    for (int i = tid; i < M * 5; i += 256) s_lab[i] = lab[(size_t)l0 * 5 + i];
    for (int i = tid; i < M; i += 256) s_key[i] = 0ull;
    if (tid == 0) s_any = 0;
    __syncthreads();
    for (int d = tid; d < N; d += 256) {
        const float *dp = det + (size_t)(d0 + d) * 6;
        int took = -2;                                                                // -2: below the confidence threshold (dropped, :111)
        if (dp[4] > conf) {
            const float x1 = dp[0], y1 = dp[1], x2 = dp[2], y2 = dp[3];
            const float area2 = __fmul_rn(x2 - x1, y2 - y1);
            float best = -1.f;
            took = -1;
            for (int l = 0; l < M; ++l) {
                const float *lp = s_lab + l * 5;
                const float iw = fmaxf(fminf(lp[3], x2) - fmaxf(lp[1], x1), 0.f), ih = fmaxf(fminf(lp[4], y2) - fmaxf(lp[2], y1), 0.f);
                const float inter = __fmul_rn(iw, ih);
                const float area1 = __fmul_rn(lp[3] - lp[1], lp[4] - lp[2]);
                const float iou = __fdiv_rn(inter, (area1 + area2) - inter);
                if (iou > iou_thres && iou > best) { best = iou; took = l; }
            }
            if (took >= 0)                                                            // IoU > 0: its float bits order like the value
                atomicMax(&s_key[took], ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)(0x7fffffff - d));
        }
        s_took[d] = took;
    }
    __syncthreads();
    for (int l = tid; l < M; l += 256) {
        const int gc = (int)s_lab[l * 5];
        if (gc < 0 || gc > nc) continue;
        const unsigned long long key = s_key[l];
        if (key) {
            const int d = 0x7fffffff - (int)(unsigned)(key & 0xffffffffu);
            const int dc = (int)det[(size_t)(d0 + d) * 6 + 5];
            if (dc >= 0 && dc <= nc) atomicAdd(&matrix[dc * W + gc], 1);
            s_any = 1;
        } else {
            atomicAdd(&matrix[nc * W + gc], 1);                                       // missed label: (background, class)
        }
    }
    __syncthreads();
    if (!s_any) return;                                                               // `if n:` (:138): no match in the image, no false positives counted
    for (int d = tid; d < N; d += 256) {
        const int took = s_took[d];
        if (took == -2) continue;
        const bool matched = took >= 0 && (0x7fffffff - (int)(unsigned)(s_key[took] & 0xffffffffu)) == d;
        if (!matched) {
            const int dc = (int)det[(size_t)(d0 + d) * 6 + 5];
            if (dc >= 0 && dc <= nc) atomicAdd(&matrix[dc * W + nc], 1);             // unmatched detection: (class, background)
        }
    }
}

// ------------------------------------------------------------------------------------------------ ap_per_class
struct ApArgs {
    const uint8_t *tp;        // [N][T]
    const float *conf, *pred_cls, *target_cls;
    int N, M, T, ncap;
    // workspace
    int *hist_t, *hist_p, *base, *cls_list, *n_present;   // [ncap] x4, [1]
    int *idx, *sorted;        // [N] each, grouped by class at base[c]
    double *rec, *pre;        // [T][N + 2*ncap] recall / precision (then envelope) per class segment, with the 2 sentinels
    double *pcurve, *rcurve;  // [ncap][1000]
    // outputs
    int *out_classes, *out_n;
    double *out_ap, *out_p, *out_r, *out_f1;
};

__global__ __launch_bounds__(256) void ap_hist_kernel(const ApArgs a) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < a.M; i += (long)gridDim.x * 256) {
        const int c = (int)a.target_cls[i];
        if (c >= 0 && c < a.ncap) atomicAdd(&a.hist_t[c], 1);
    }
    for (long i = blockIdx.x * 256L + threadIdx.x; i < a.N; i += (long)gridDim.x * 256) {
        const int c = (int)a.pred_cls[i];
        if (c >= 0 && c < a.ncap) atomicAdd(&a.hist_p[c], 1);
    }
}
// classes present among the targets, ascending (np.unique); segment bases of the per-class detection lists
__global__ void ap_classes_kernel(const ApArgs a) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int n = 0, acc = 0;
    for (int c = 0; c < a.ncap; ++c) {
        a.base[c] = acc;
        acc += a.hist_p[c];
        if (a.hist_t[c] > 0) a.cls_list[n++] = c;
    }
    *a.n_present = n;
}
// block c: ordered compaction of the detections of class c
__global__ __launch_bounds__(256) void ap_gather_kernel(const ApArgs a) {
    __shared__ int s_cnt[256];
    __shared__ int s_base;
    const int c = blockIdx.x, tid = threadIdx.x;
    if (a.hist_t[c] == 0 || a.hist_p[c] == 0) return;
    int *out = a.idx + a.base[c];
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int c0 = 0; c0 < a.N; c0 += 256) {
        const int i = c0 + tid;
        const int f = (i < a.N && (int)a.pred_cls[i] == c && a.pred_cls[i] >= 0.f) ? 1 : 0;
        s_cnt[tid] = f;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const int v = tid >= off ? s_cnt[tid - off] : 0;
            __syncthreads();
            s_cnt[tid] += v;
            __syncthreads();
        }
        if (f) out[s_base + s_cnt[tid] - 1] = i;
        __syncthreads();
        if (tid == 0) s_base += s_cnt[255];
        __syncthreads();
    }
}
// stable descending rank by confidence inside each class: grid (chunks, ncap)
__global__ __launch_bounds__(256) void ap_rank_kernel(const ApArgs a) {
    const int c = blockIdx.y;
    const int n = a.hist_p[c];
    if (a.hist_t[c] == 0 || n == 0) return;
    const int *idx = a.idx + a.base[c];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int me = idx[i];
        const float ci = a.conf[me];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const float cj = a.conf[idx[j]];
            rank += (cj > ci || (cj == ci && j < i)) ? 1 : 0;
        }
        a.sorted[a.base[c] + rank] = me;
    }
}

// np.interp for an increasing xp: largest j with xp[j] <= x by bisection (duplicates resolve to the last one, like numpy)
__device__ __forceinline__ double interp1(double x, const double *xp, const double *fp, int n, double left, double right) {
    if (x < xp[0]) return left;
    if (x > xp[n - 1]) return right;
    int lo = 0, hi = n - 1;                                                // invariant: xp[lo] <= x
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (xp[mid] <= x) lo = mid; else hi = mid;
    }
    if (xp[hi] <= x) lo = hi;
    if (lo == n - 1 || xp[lo] == x) return fp[lo];
    const double slope = __ddiv_rn(fp[lo + 1] - fp[lo], xp[lo + 1] - xp[lo]);
    return __dadd_rn(__dmul_rn(slope, x - xp[lo]), fp[lo]);
}

// grid (T, ncap): recall / precision of class c at IoU level j over its confidence-sorted detections, the precision envelope and
// the 101-point AP (utils/metrics.py:76-95).  One lane walks the running sums (a scan of a few thousand entries).
__global__ __launch_bounds__(64) void ap_curve_kernel(const ApArgs a) {
    const int j = blockIdx.x, ci = blockIdx.y;
    if (ci >= *a.n_present) return;
    const int c = a.cls_list[ci];
    const int n = a.hist_p[c], n_l = a.hist_t[c];
    double *ap_out = a.out_ap + (size_t)ci * a.T + j;
    if (threadIdx.x != 0) return;
    if (n == 0) { *ap_out = 0.0; return; }
    const size_t seg = (size_t)j * ((size_t)a.N + 2 * (size_t)a.ncap) + a.base[c] + 2 * (size_t)c;   // n + 2 entries
    double *mrec = a.rec + seg, *mpre = a.pre + seg;
    const int *srt = a.sorted + a.base[c];
    double tpc = 0.0, fpc = 0.0;
    mrec[0] = 0.0;
    mpre[0] = 1.0;
    for (int k = 0; k < n; ++k) {
        const double t = a.tp[(size_t)srt[k] * a.T + j] ? 1.0 : 0.0;
        tpc += t;
        fpc += 1.0 - t;
        mrec[k + 1] = tpc / ((double)n_l + 1e-16);
        mpre[k + 1] = tpc / (tpc + fpc);
    }
    mrec[n + 1] = 1.0;
    mpre[n + 1] = 0.0;
    for (int k = n; k >= 0; --k) mpre[k] = fmax(mpre[k], mpre[k + 1]);      // np.flip(np.maximum.accumulate(np.flip(mpre)))
    double sum = 0.0, prev = 0.0;
    for (int k = 0; k <= 100; ++k) {
        const double x = k == 100 ? 1.0 : (double)k * 0.01;                 // np.linspace(0, 1, 101)
        const double y = interp1(x, mrec, mpre, n + 2, mpre[0], mpre[n + 1]);
        if (k > 0) {
            const double xprev = (double)(k - 1) * 0.01;
            sum += (x - xprev) * (y + prev) / 2.0;                           // np.trapz
        }
        prev = y;
    }
    *ap_out = sum;
}

// grid (ncap): precision / recall of class ci at IoU level 0 as functions of confidence on the 1000-point grid
// (utils/metrics.py:49-57).  Recomputes the raw level-0 curves (the envelope pass overwrote precision) into LDS-free scratch.
__global__ __launch_bounds__(256) void ap_pr_kernel(const ApArgs a, double *xneg, double *rraw, double *praw) {
    const int ci = blockIdx.x;
    if (ci >= *a.n_present) return;
    const int c = a.cls_list[ci];
    const int n = a.hist_p[c], n_l = a.hist_t[c];
    double *pc = a.pcurve + (size_t)ci * 1000, *rc = a.rcurve + (size_t)ci * 1000;
    if (n == 0) {
        for (int k = threadIdx.x; k < 1000; k += 256) { pc[k] = 0.0; rc[k] = 0.0; }
        return;
    }
    const int *srt = a.sorted + a.base[c];
    double *xs = xneg + a.base[c], *rr = rraw + a.base[c], *pp = praw + a.base[c];
    if (threadIdx.x == 0) {
        double tpc = 0.0, fpc = 0.0;
        for (int k = 0; k < n; ++k) {
            const double t = a.tp[(size_t)srt[k] * a.T] ? 1.0 : 0.0;
            tpc += t;
            fpc += 1.0 - t;
            xs[k] = -(double)a.conf[srt[k]];
            rr[k] = tpc / ((double)n_l + 1e-16);
            pp[k] = tpc / (tpc + fpc);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 1000; k += 256) {
        const double px = k == 999 ? 1.0 : (double)k * (1.0 / 999.0);       // np.linspace(0, 1, 1000)
        rc[k] = interp1(-px, xs, rr, n, 0.0, rr[n - 1]);
        pc[k] = interp1(-px, xs, pp, n, 1.0, pp[n - 1]);
    }
}

// f1 = 2pr/(p+r+1e-16); operating point = argmax of the class-mean f1 (first maximum); outputs at that point
__global__ __launch_bounds__(256) void ap_final_kernel(const ApArgs a) {
    __shared__ double s_val[256];
    __shared__ int s_idx[256];
    const int nc = *a.n_present, tid = threadIdx.x;
    double best = -1.0;
    int besti = 0;
    for (int k = tid; k < 1000; k += 256) {
        double m = 0.0;
        for (int ci = 0; ci < nc; ++ci) {
            const double p = a.pcurve[(size_t)ci * 1000 + k], r = a.rcurve[(size_t)ci * 1000 + k];
            m += 2.0 * p * r / (p + r + 1e-16);
        }
        m = nc > 0 ? m / (double)nc : 0.0;
        if (m > best) { best = m; besti = k; }
    }
    s_val[tid] = best;
    s_idx[tid] = besti;
    __syncthreads();
    if (tid == 0) {
        for (int t = 1; t < 256; ++t)
            if (s_val[t] > s_val[0] || (s_val[t] == s_val[0] && s_idx[t] < s_idx[0])) { s_val[0] = s_val[t]; s_idx[0] = s_idx[t]; }
        const int k = s_idx[0];
        *a.out_n = nc;
        for (int ci = 0; ci < nc; ++ci) {
            const double p = a.pcurve[(size_t)ci * 1000 + k], r = a.rcurve[(size_t)ci * 1000 + k];
            a.out_classes[ci] = a.cls_list[ci];
            a.out_p[ci] = p;
            a.out_r[ci] = r;
            a.out_f1[ci] = 2.0 * p * r / (p + r + 1e-16);
        }
    }
}

static size_t al16(size_t v) { return (v + 15) / 16 * 16; }

}  // namespace somi

using namespace somi;

extern "C" int somi_val_match_f32(const float *det, const int *det_off, const float *labels, const int *lab_off, const float *iouv, int T,
                                  int B, int max_det, int max_labels, uint8_t *correct, somi_stream_t stream) {
    SOMI_REQUIRE(det_off && lab_off && iouv && correct && B > 0 && T > 0 && T <= 16 && max_det >= 0 && max_labels >= 0, SOMI_EINVAL,
                 "val match: bad arguments (at most 16 IoU levels)");
    if (max_det == 0) return 0;
    SOMI_REQUIRE(det && (labels || max_labels == 0), SOMI_EINVAL, "val match: null boxes");
    const int ncap = max_det, mcap = max_labels > 0 ? max_labels : 1;
    const size_t lds = (size_t)mcap * 5 * 4 + (size_t)T * ncap * 4 + (size_t)T * mcap * 4;
    SOMI_REQUIRE(lds <= 150 * 1024, SOMI_ENOTIMPL, "val match: %d detections x %d labels per image do not fit LDS", max_det, max_labels);
    if (lds > 64 * 1024) {
        static size_t set_to = 0;
        if (lds > set_to) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(val_match_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            set_to = 150 * 1024;
        }
    }
    hipLaunchKernelGGL(val_match_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, det, det_off, labels, lab_off, iouv, T, ncap, mcap, correct);
    return launch_status("somi_val_match_f32");
}

extern "C" int somi_confusion_matrix_f32(const float *det, const int *det_off, const float *labels, const int *lab_off, int B, int max_det,
                                         int max_labels, int nc, float conf, float iou_thres, int32_t *matrix, somi_stream_t stream) {
    SOMI_REQUIRE(det_off && lab_off && matrix && B > 0 && max_det >= 0 && max_labels >= 0 && nc > 0, SOMI_EINVAL,
                 "confusion matrix: bad arguments");
    SOMI_REQUIRE((det || max_det == 0) && (labels || max_labels == 0), SOMI_EINVAL, "confusion matrix: null boxes");
    const int ncap = max_det > 0 ? max_det : 1, mcap = max_labels > 0 ? max_labels : 1;
    const size_t lds = (size_t)mcap * 8 + (size_t)mcap * 5 * 4 + (size_t)ncap * 4;
    SOMI_REQUIRE(lds <= 60 * 1024, SOMI_ENOTIMPL, "confusion matrix: %d detections x %d labels per image do not fit LDS", max_det, max_labels);
    hipLaunchKernelGGL(confusion_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, det, det_off, labels, lab_off, ncap, mcap, nc, conf,
                       iou_thres, matrix);
    return launch_status("somi_confusion_matrix_f32");
}

extern "C" size_t somi_ap_per_class_workspace_bytes(long N, int T, int ncap) {
    if (N < 0 || T <= 0 || ncap <= 0) return 0;
    const size_t n = (size_t)N, curve = (size_t)T * (n + 2 * (size_t)ncap);
    return al16((size_t)ncap * 4) * 4 + 16 + al16(n * 4) * 2 + al16(curve * 8) * 2 + al16((size_t)ncap * 1000 * 8) * 2 + al16(n * 8) * 3 + 256;
}

extern "C" int somi_ap_per_class_f64(const uint8_t *tp, const float *conf, const float *pred_cls, const float *target_cls, long N, long M,
                                     int T, int ncap, int *out_classes, int *out_n, double *out_ap, double *out_p, double *out_r,
                                     double *out_f1, void *workspace, size_t workspace_bytes, somi_stream_t stream) {
    SOMI_REQUIRE(out_classes && out_n && out_ap && out_p && out_r && out_f1 && workspace && T > 0 && ncap > 0 && N >= 0 && M >= 0 &&
                     N < (1L << 30) && M < (1L << 30), SOMI_EINVAL, "ap_per_class: bad arguments");
    SOMI_REQUIRE((N == 0 || (tp && conf && pred_cls)) && (M == 0 || target_cls), SOMI_EINVAL, "ap_per_class: null input");
    SOMI_REQUIRE(workspace_bytes >= somi_ap_per_class_workspace_bytes(N, T, ncap) && aligned16(workspace), SOMI_EWORKSPACE,
                 "ap_per_class: workspace too small or unaligned");
    hipStream_t s = (hipStream_t)stream;
    ApArgs a{};
    a.tp = tp; a.conf = conf; a.pred_cls = pred_cls; a.target_cls = target_cls;
    a.N = (int)N; a.M = (int)M; a.T = T; a.ncap = ncap;
    char *w = static_cast<char *>(workspace);
    auto take = [&](size_t bytes) { char *p = w; w += al16(bytes); return p; };
    a.hist_t = (int *)take((size_t)ncap * 4); a.hist_p = (int *)take((size_t)ncap * 4);
    a.base = (int *)take((size_t)ncap * 4); a.cls_list = (int *)take((size_t)ncap * 4);
    a.n_present = (int *)take(16);
    const size_t zero_bytes = (size_t)(w - static_cast<char *>(workspace));
    a.idx = (int *)take((size_t)N * 4); a.sorted = (int *)take((size_t)N * 4);
    const size_t curve = (size_t)T * ((size_t)N + 2 * (size_t)ncap);
    a.rec = (double *)take(curve * 8); a.pre = (double *)take(curve * 8);
    a.pcurve = (double *)take((size_t)ncap * 1000 * 8); a.rcurve = (double *)take((size_t)ncap * 1000 * 8);
    double *xneg = (double *)take((size_t)N * 8), *rraw = (double *)take((size_t)N * 8), *praw = (double *)take((size_t)N * 8);
    a.out_classes = out_classes; a.out_n = out_n; a.out_ap = out_ap; a.out_p = out_p; a.out_r = out_r; a.out_f1 = out_f1;
    if (hipMemsetAsync(workspace, 0, zero_bytes, s) != hipSuccess) { set_error("ap_per_class: memset failed"); return SOMI_EINVAL; }
    hipLaunchKernelGGL(ap_hist_kernel, dim3(256), dim3(256), 0, s, a);
    hipLaunchKernelGGL(ap_classes_kernel, dim3(1), dim3(64), 0, s, a);
    if (N > 0) {
        hipLaunchKernelGGL(ap_gather_kernel, dim3(ncap), dim3(256), 0, s, a);
        hipLaunchKernelGGL(ap_rank_kernel, dim3(64, ncap), dim3(256), 0, s, a);
    }
    hipLaunchKernelGGL(ap_curve_kernel, dim3(T, ncap), dim3(64), 0, s, a);
    hipLaunchKernelGGL(ap_pr_kernel, dim3(ncap), dim3(256), 0, s, a, xneg, rraw, praw);
    hipLaunchKernelGGL(ap_final_kernel, dim3(1), dim3(256), 0, s, a);
    return launch_status("somi_ap_per_class_f64");
}
