// Repulsion loss (RepGT + RepBox) - utils/RepulsionLoss.py:47-95.  The reference imports it (utils/loss.py:8) but never calls
// it; it is provided as an optional, value-only term (the reference detaches both box sets, so there is no gradient).
//
// One workgroup per image: (1) ordered compaction of the foreground anchors, (2) every thread owns rows i of the n x n
// pred-vs-gt and pred-vs-pred IoU matrices and walks the columns (boxes come through L1: n is a few hundred), keeping the
// row's best gt match (first maximum, like torch.max) and the strictly-lower-triangle RepBox sum, (3) workgroup reduction
// in double.  A last one-workgroup kernel adds the images in order and divides by the number of images with positives.
#include "common.h"

namespace somi {

struct RepArgs {
    const float *pbox, *gtbox;       // (B, A, 4) xyxy
    const uint8_t *fg;               // (B, A)
    int B, A;
    float sigma_gt, sigma_box, pnms, gtnms, log1m_gt, log1m_box;
    int *idx;                        // [B][A] compacted foreground indices
    double *per_img;                 // [B][3]: rep_gt term, rep_box term, used
};

__device__ __forceinline__ float rep_smooth_ln(float x, float sigma, float log1m_sigma) {
    return x <= sigma ? -logf(1.f - x) : (x - sigma) / (1.f - sigma) - log1m_sigma;     // utils/RepulsionLoss.py:39-44
}

// IoU of RepulsionLoss.py:5-24 ('xyxy'): zero unless the boxes strictly overlap on both axes
__device__ __forceinline__ float rep_iou(const float4 a, const float4 b) {
    const float ltx = fmaxf(a.x, b.x), lty = fmaxf(a.y, b.y), rbx = fminf(a.z, b.z), rby = fminf(a.w, b.w);
    const float area1 = (a.z - a.x) * (a.w - a.y), area2 = (b.z - b.x) * (b.w - b.y);
    const float valid = (ltx < rbx && lty < rby) ? 1.f : 0.f;
    const float inter = (rbx - ltx) * (rby - lty) * valid;
    return inter / (area1 + area2 - inter);
}

__global__ __launch_bounds__(256) void repulsion_image_kernel(const RepArgs a) {
    __shared__ int s_cnt[256];
    __shared__ int s_base;
    __shared__ double r_gt[256], r_box[256];
    __shared__ int r_hit[256], r_any[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const uint8_t *fg = a.fg + (size_t)b * a.A;
    int *idx = a.idx + (size_t)b * a.A;
    // ---- ordered compaction, 256 anchors at a time
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int c0 = 0; c0 < a.A; c0 += 256) {
        const int i = c0 + tid;
        const int f = (i < a.A && fg[i]) ? 1 : 0;
        s_cnt[tid] = f;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {                     // inclusive scan
            const int v = tid >= off ? s_cnt[tid - off] : 0;
            __syncthreads();
            s_cnt[tid] += v;
            __syncthreads();
        }
        if (f) idx[s_base + s_cnt[tid] - 1] = i;
        __syncthreads();
        if (tid == 0) s_base += s_cnt[255];
        __syncthreads();
    }
    const int n = s_base;
    __threadfence_block();
    __syncthreads();
    const float4 *pb = reinterpret_cast<const float4 *>(a.pbox) + (size_t)b * a.A;
    const float4 *gb = reinterpret_cast<const float4 *>(a.gtbox) + (size_t)b * a.A;
    double sum_gt = 0.0, sum_box = 0.0;
    int hits = 0, any = 0;
    for (int i = tid; i < n; i += 256) {
        const float4 pi = pb[idx[i]], gi = gb[idx[i]];
        float best = -__builtin_huge_valf();
        int arg = 0;
        for (int j = 0; j < n; ++j) {
            const float4 pj = pb[idx[j]], gj = gb[idx[j]];
            const bool same = gi.x == gj.x && gi.y == gj.y && gi.z == gj.z && gi.w == gj.w;
            const float pg = same ? 0.f : rep_iou(pi, gj);
            if (pg > best) { best = pg; arg = j; }
            if (j < i && !same) {                                      // strict lower triangle, different gt
                const float pp = rep_iou(pi, pj);
                any |= pp > a.pnms;
                sum_box += (double)rep_smooth_ln(pp, a.sigma_box, a.log1m_box);
            }
        }
        if (best > a.gtnms) {
            const float4 g = gb[idx[arg]];
            const float iw = fmaxf(fminf(g.z, pi.z) - fmaxf(g.x, pi.x), 0.f), ih = fmaxf(fminf(g.w, pi.w) - fmaxf(g.y, pi.y), 0.f);
            const float garea = fmaxf((g.z - g.x) * (g.w - g.y), 1e-6f);
            sum_gt += (double)rep_smooth_ln(iw * ih / garea, a.sigma_gt, a.log1m_gt);   // IoG, RepulsionLoss.py:27-36
            ++hits;
        }
    }
    r_gt[tid] = sum_gt; r_box[tid] = sum_box; r_hit[tid] = hits; r_any[tid] = any;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) {
            r_gt[tid] += r_gt[tid + off]; r_box[tid] += r_box[tid + off];
            r_hit[tid] += r_hit[tid + off]; r_any[tid] |= r_any[tid + off];
        }
        __syncthreads();
    }
    if (tid == 0) {
        // the masked entries of the n x n pred-pred matrix contribute smooth_ln(0, sigma) = 0 to the mean for every sigma >= 0
        const double box_total = r_box[0];
        a.per_img[b * 3 + 0] = r_hit[0] > 0 ? r_gt[0] / (double)r_hit[0] : 0.0;
        a.per_img[b * 3 + 1] = (n > 0 && r_any[0]) ? box_total / ((double)n * (double)n) : 0.0;
        a.per_img[b * 3 + 2] = n > 0 ? 1.0 : 0.0;
    }
}

__global__ void repulsion_finish_kernel(const double *per_img, int B, float *out2) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float gt = 0.f, box = 0.f;                                         // fp32 running sums over images like the reference
    float used = 0.f;
    for (int b = 0; b < B; ++b) {
        if (per_img[b * 3 + 2] == 0.0) continue;
        gt += (float)per_img[b * 3 + 0];
        box += (float)per_img[b * 3 + 1];
        used += 1.f;
    }
    out2[0] = gt / used;                                               // no image with positives: 0/0 like the reference
    out2[1] = box / used;
}

}  // namespace somi

using namespace somi;

extern "C" size_t somi_repulsion_workspace_bytes(int B, int A) {
    if (B <= 0 || A <= 0) return 0;
    return (((size_t)B * A * sizeof(int) + 15) / 16) * 16 + (size_t)B * 3 * sizeof(double);
}

extern "C" int somi_repulsion_loss_f32(const float *pbox, const float *gtbox, const uint8_t *fg_mask, int B, int A, float sigma_repgt,
                                       float sigma_repbox, float pnms, float gtnms, float *out2, void *workspace,
                                       size_t workspace_bytes, somi_stream_t stream) {
    SOMI_REQUIRE(pbox && gtbox && fg_mask && out2 && workspace && B > 0 && A > 0, SOMI_EINVAL, "repulsion: bad arguments");
    SOMI_REQUIRE(aligned16(pbox) && aligned16(gtbox) && aligned16(workspace), SOMI_EINVAL, "repulsion: boxes / workspace must be 16 B aligned");
    SOMI_REQUIRE(sigma_repgt >= 0.f && sigma_repgt < 1.f && sigma_repbox >= 0.f && sigma_repbox < 1.f, SOMI_EINVAL,
                 "repulsion: sigma must be in [0,1)");
    SOMI_REQUIRE(workspace_bytes >= somi_repulsion_workspace_bytes(B, A), SOMI_EWORKSPACE, "repulsion: workspace too small");
    RepArgs a;
    a.pbox = pbox; a.gtbox = gtbox; a.fg = fg_mask; a.B = B; a.A = A;
    a.sigma_gt = sigma_repgt; a.sigma_box = sigma_repbox; a.pnms = pnms; a.gtnms = gtnms;
    a.log1m_gt = (float)log(1.0 - (double)sigma_repgt);
    a.log1m_box = (float)log(1.0 - (double)sigma_repbox);
    a.idx = static_cast<int *>(workspace);
    a.per_img = reinterpret_cast<double *>(static_cast<char *>(workspace) + (((size_t)B * A * sizeof(int) + 15) / 16) * 16);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(repulsion_image_kernel, dim3(B), dim3(256), 0, s, a);
    hipLaunchKernelGGL(repulsion_finish_kernel, dim3(1), dim3(64), 0, s, a.per_img, B, out2);
    return launch_status("somi_repulsion_loss_f32");
}
