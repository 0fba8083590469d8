// Batched NMS on gfx950: utils/general.py:629-711 including the torchvision.ops.nms core (:694).
//
// Per image, entirely on the device (the reference loops over images in Python with nonzero() syncs):
//   1. count   candidates per 1024-row chunk           (obj > thr, then obj*cls > thr per class / best class)
//   2. scan    chunk counts -> offsets                  (compaction order == prediction order: box-major, class-minor)
//   3. emit    key = ~bits(score), val = row*nc + cls   (8 B per candidate; boxes are re-derived from `pred` when needed)
//   4. sort    stable LSD radix sort, 4-bit digits, one workgroup per image (ties keep prediction order - the oracle's
//              stable descending sort); only the first max_nms = 30000 sorted entries are used (:688-689)
//   5. greedy  512 candidates per round: test against the kept list (LDS), build the round's 512x512 suppression bit
//              matrix, resolve it serially in one wave, append survivors; stop at max_det kept (:696-697).
// IoU arithmetic is written with explicit round-to-nearest intrinsics (no FMA contraction) in the oracle's operation
// order, so the kept index lists are bit-exact: area=(x2-x1)*(y2-y1) on class-offset boxes (offset 4096*cls, :692-693),
// inter=max(0,min(x2)-max(x1))*max(0,min(y2)-max(y1)), iou=inter/((area_i+area_j)-inter), suppress iff iou > thr.
#include "common.h"

namespace somi {

constexpr int NMS_CHUNK = 1024;      // rows per count/emit workgroup (256 threads x 4 consecutive rows)
constexpr int MAX_NMS = 30000;       // utils/general.py:641
constexpr float MAX_WH = 4096.f;     // utils/general.py:640
constexpr int ROUND = 512;           // candidates per greedy round
constexpr int MAX_DET_CAP = 1024;

struct NmsArgs {
    const float *pred;
    int B, n, nc, no;
    float conf_thres, iou_thres;
    int multi_label, agnostic, max_det;
    uint64_t classes_mask;
    int nchunk, cap;                 // cap = candidate capacity per image
    int *chunk_cnt;                  // [B][nchunk]
    int *chunk_off;                  // [B][nchunk]
    int *total;                      // [B]
    uint32_t *keyA, *valA, *keyB, *valB;   // [B][cap]
    float *det;                      // [B][max_det][6]
    int32_t *count;                  // [B]
};

__device__ __forceinline__ bool class_ok(uint64_t mask, int c) { return c >= 64 || ((mask >> c) & 1ull); }

// number of entries row r contributes; optionally writes them (keys/vals) starting at dst
template <bool EMIT>
__device__ __forceinline__ int row_entries(const NmsArgs &a, const float *row, int r, uint32_t *keys, uint32_t *vals, int dst) {
    const float obj = row[4];
    if (!(obj > a.conf_thres)) return 0;
    int cnt = 0;
    if (a.multi_label) {
        for (int j = 0; j < a.nc; ++j) {
            const float conf = __fmul_rn(row[5 + j], obj);
            if (conf > a.conf_thres && class_ok(a.classes_mask, j)) {
                if (EMIT) {
                    keys[dst + cnt] = ~__float_as_uint(conf);
                    vals[dst + cnt] = (uint32_t)r * (uint32_t)a.nc + (uint32_t)j;
                }
                ++cnt;
            }
        }
    } else {
        float best = __fmul_rn(row[5], obj);
        int bj = 0;
        for (int j = 1; j < a.nc; ++j) {
            const float conf = __fmul_rn(row[5 + j], obj);
            if (conf > best) { best = conf; bj = j; }          // first maximum wins, like torch.max
        }
        if (best > a.conf_thres && class_ok(a.classes_mask, bj)) {
            if (EMIT) {
                keys[dst] = ~__float_as_uint(best);
                vals[dst] = (uint32_t)r * (uint32_t)a.nc + (uint32_t)bj;
            }
            cnt = 1;
        }
    }
    return cnt;
}

// block-wide exclusive scan of one int per thread (256 threads); returns the exclusive prefix, total in *tot
__device__ __forceinline__ int block_exscan_256(int v, int *tot, int *lds /* >= 4 ints */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += lds[w];
    *tot = lds[0] + lds[1] + lds[2] + lds[3];
    __syncthreads();
    return base + inc - v;
}

template <bool EMIT>
__global__ __launch_bounds__(256) void nms_count_emit_kernel(const NmsArgs a) {
    __shared__ int lds[4];
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int r0 = chunk * NMS_CHUNK + threadIdx.x * 4;
    const float *img = a.pred + (size_t)b * a.n * a.no;
    int cnt[4], mine = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + i;
        cnt[i] = r < a.n ? row_entries<false>(a, img + (size_t)r * a.no, r, nullptr, nullptr, 0) : 0;
        mine += cnt[i];
    }
    int tot;
    const int pre = block_exscan_256(mine, &tot, lds);
    if (!EMIT) {
        if (threadIdx.x == 0) a.chunk_cnt[b * a.nchunk + chunk] = tot;
        return;
    }
    int dst = a.chunk_off[b * a.nchunk + chunk] + pre;
    uint32_t *keys = a.keyA + (size_t)b * a.cap, *vals = a.valA + (size_t)b * a.cap;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + i;
        if (cnt[i]) row_entries<true>(a, img + (size_t)r * a.no, r, keys, vals, dst);
        dst += cnt[i];
    }
}

__global__ __launch_bounds__(256) void nms_scan_kernel(const NmsArgs a) {
    __shared__ int lds[4];
    __shared__ int carry;
    const int b = blockIdx.x;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int c0 = 0; c0 < a.nchunk; c0 += 256) {
        const int c = c0 + threadIdx.x;
        const int v = c < a.nchunk ? a.chunk_cnt[b * a.nchunk + c] : 0;
        int tot;
        const int pre = block_exscan_256(v, &tot, lds);
        if (c < a.nchunk) a.chunk_off[b * a.nchunk + c] = carry + pre;
        __syncthreads();
        if (threadIdx.x == 0) carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) a.total[b] = carry;
}

// ------------------------------------------------------------------------------------------------ radix sort
// One workgroup (1024 threads = 16 waves) per image; 8 passes of 4 bits; stable.
__global__ __launch_bounds__(1024) void nms_sort_kernel(const NmsArgs a) {
    __shared__ int hist[16];            // digit totals -> exclusive bases (running, advanced tile by tile)
    __shared__ int wcnt[16][16];        // [wave][digit] counts of the current tile
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = a.total[b];
    if (n <= 1) return;
    uint32_t *k0 = a.keyA + (size_t)b * a.cap, *v0 = a.valA + (size_t)b * a.cap;
    uint32_t *k1 = a.keyB + (size_t)b * a.cap, *v1 = a.valB + (size_t)b * a.cap;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    for (int pass = 0; pass < 8; ++pass) {
        const int shift = pass * 4;
        if (tid < 16) hist[tid] = 0;
        __syncthreads();
        // (a) digit histogram of the whole segment: per-wave ballot counts, one LDS atomic per wave and digit
        for (int i0 = 0; i0 < n; i0 += 1024) {
            const int i = i0 + tid;
            const bool ok = i < n;
            const int dgt = ok ? (int)((k0[i] >> shift) & 15u) : 16;
            for (int dd = 0; dd < 16; ++dd) {
                const unsigned long long m = __ballot(dgt == dd);
                if (lane == 0 && m) atomicAdd(&hist[dd], __popcll(m));
            }
        }
        __syncthreads();
        if (tid == 0) {
            int run = 0;
            for (int dd = 0; dd < 16; ++dd) { const int c = hist[dd]; hist[dd] = run; run += c; }
        }
        __syncthreads();
        // (b) stable scatter, tile by tile in order
        for (int i0 = 0; i0 < n; i0 += 1024) {
            const int i = i0 + tid;
            const bool ok = i < n;
            uint32_t key = 0, val = 0;
            if (ok) { key = k0[i]; val = v0[i]; }
            const int dgt = ok ? (int)((key >> shift) & 15u) : 16;
            // lanes of this wave with my digit
            unsigned long long peers = 0;
            for (int dd = 0; dd < 16; ++dd) {
                const unsigned long long m = __ballot(dgt == dd);
                if (dgt == dd) peers = m;
                if (lane == 0) wcnt[wave][dd] = __popcll(m);
            }
            __syncthreads();
            int before = 0;                                  // same digit in earlier waves of this tile
            if (ok) {
                for (int w = 0; w < wave; ++w) before += wcnt[w][dgt];
                const int pos = hist[dgt] + before + __popcll(peers & lt_mask);
                k1[pos] = key;
                v1[pos] = val;
            }
            __syncthreads();
            if (tid < 16) {                                  // advance the running bases past this tile
                int s = 0;
                for (int w = 0; w < 16; ++w) s += wcnt[w][tid];
                hist[tid] += s;
            }
            __syncthreads();
        }
        uint32_t *t = k0; k0 = k1; k1 = t;
        t = v0; v0 = v1; v1 = t;
        __threadfence_block();
        __syncthreads();
    }
    // 8 passes: the sorted data is back in keyA / valA
}

// ------------------------------------------------------------------------------------------------ greedy NMS
struct BoxO { float x1, y1, x2, y2, area; };   // class-offset corners + area

__device__ __forceinline__ bool iou_gt(const BoxO &p, const BoxO &q, float thr) {
    const float xx1 = fmaxf(p.x1, q.x1), yy1 = fmaxf(p.y1, q.y1);
    const float xx2 = fminf(p.x2, q.x2), yy2 = fminf(p.y2, q.y2);
    const float iw = fmaxf(0.f, __fsub_rn(xx2, xx1)), ih = fmaxf(0.f, __fsub_rn(yy2, yy1));
    const float inter = __fmul_rn(iw, ih);
    const float iou = __fdiv_rn(inter, __fsub_rn(__fadd_rn(p.area, q.area), inter));
    return iou > thr;
}

__global__ __launch_bounds__(ROUND) void nms_greedy_kernel(const NmsArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    BoxO *kept = reinterpret_cast<BoxO *>(smem);                         // [max_det]
    BoxO *cand = kept + a.max_det;                                       // [ROUND]
    uint32_t *mat = reinterpret_cast<uint32_t *>(cand + ROUND);          // [ROUND][ROUND/32] suppression bits (j > i)
    uint32_t *alive = mat + ROUND * (ROUND / 32);                        // [ROUND/32]
    __shared__ int nk_sh, nsel_sh;
    __shared__ int sel[ROUND];                                           // indices (within the round) kept this round
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    int n = a.total[b];
    if (n > MAX_NMS) n = MAX_NMS;
    const uint32_t *vals = a.valA + (size_t)b * a.cap;
    const float *img = a.pred + (size_t)b * a.n * a.no;
    float *det = a.det + (size_t)b * a.max_det * 6;
    if (tid == 0) nk_sh = 0;
    __syncthreads();
    for (int r0 = 0; r0 < n; r0 += ROUND) {
        const int nk = nk_sh;
        if (nk >= a.max_det) break;
        const int i = r0 + tid;
        const bool ok = i < n;
        BoxO me = {0.f, 0.f, 0.f, 0.f, 0.f};
        float out6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (ok) {
            const uint32_t v = vals[i];
            const int row = (int)(v / (uint32_t)a.nc), cls = (int)(v % (uint32_t)a.nc);
            const float *p = img + (size_t)row * a.no;
            const float cx = p[0], cy = p[1], hw = __fdiv_rn(p[2], 2.f), hh = __fdiv_rn(p[3], 2.f);
            out6[0] = __fsub_rn(cx, hw);
            out6[1] = __fsub_rn(cy, hh);
            out6[2] = __fadd_rn(cx, hw);
            out6[3] = __fadd_rn(cy, hh);
            out6[4] = __fmul_rn(p[5 + cls], p[4]);
            out6[5] = (float)cls;
            const float c = a.agnostic ? 0.f : __fmul_rn((float)cls, MAX_WH);
            me.x1 = __fadd_rn(out6[0], c);
            me.y1 = __fadd_rn(out6[1], c);
            me.x2 = __fadd_rn(out6[2], c);
            me.y2 = __fadd_rn(out6[3], c);
            me.area = __fmul_rn(__fsub_rn(me.x2, me.x1), __fsub_rn(me.y2, me.y1));
        }
        cand[tid] = me;
        // phase 1: suppressed by a box kept in an earlier round?
        bool live = ok;
        for (int k = 0; k < nk && live; ++k) live = !iou_gt(kept[k], me, a.iou_thres);
        const unsigned long long bal = __ballot(live);
        if (lane == 0) {
            alive[(tid >> 5)] = (uint32_t)bal;
            alive[(tid >> 5) + 1] = (uint32_t)(bal >> 32);
        }
        __syncthreads();
        // phase 2: this round's suppression matrix, row tid = bits j > tid with IoU > thr (only if both live)
        for (int w = 0; w < ROUND / 32; ++w) {
            uint32_t bits = 0;
            if (live && (w * 32 + 31) > tid) {
                const uint32_t al = alive[w];
                for (int jj = 0; jj < 32; ++jj) {
                    const int j = w * 32 + jj;
                    if (j > tid && ((al >> jj) & 1u) && iou_gt(me, cand[j], a.iou_thres)) bits |= 1u << jj;
                }
            }
            mat[tid * (ROUND / 32) + w] = bits;
        }
        __syncthreads();
        // phase 3: serial resolve in wave 0: lanes 0..15 each own one 32-bit word of the `removed` set
        if (tid < 64) {
            uint32_t removed = 0;                       // word `lane` (lanes >= 16 idle)
            int nsel = 0, room = a.max_det - nk;
            for (int i2 = 0; i2 < ROUND && nsel < room; ++i2) {
                const int w = i2 >> 5;
                const uint32_t aw = alive[w];
                const uint32_t rw = __shfl(removed, w);
                if (((aw >> (i2 & 31)) & 1u) && !((rw >> (i2 & 31)) & 1u)) {
                    if (lane == 0) sel[nsel] = i2;
                    ++nsel;
                    if (lane < ROUND / 32) removed |= mat[i2 * (ROUND / 32) + lane];
                }
            }
            if (lane == 0) nsel_sh = nsel;
        }
        __syncthreads();
        // phase 4: append survivors (in order) to the kept list and the output
        const int nsel = nsel_sh;
        // each thread checks whether it was selected: position = rank in sel[]
        if (ok) {
            // binary search of tid in sel[0..nsel)
            int lo = 0, hi = nsel;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (sel[mid] < tid) lo = mid + 1; else hi = mid;
            }
            if (lo < nsel && sel[lo] == tid) {
                const int slot = nk + lo;
                kept[slot] = me;
#pragma unroll
                for (int e = 0; e < 6; ++e) det[slot * 6 + e] = out6[e];
            }
        }
        __syncthreads();
        if (tid == 0) nk_sh = nk + nsel;
        __syncthreads();
    }
    if (tid == 0) a.count[b] = nk_sh;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace somi

using namespace somi;

extern "C" size_t somi_nms_workspace_bytes(int B, int n, int nc, int multi_label) {
    if (B <= 0 || n <= 0 || nc <= 0) return 0;
    const size_t nchunk = (size_t)(n + NMS_CHUNK - 1) / NMS_CHUNK;
    const size_t cap = (size_t)n * (size_t)((multi_label && nc > 1) ? nc : 1);
    return align_up((size_t)B * nchunk * 4, 256) * 2 + align_up((size_t)B * 4, 256) + align_up((size_t)B * cap * 4, 256) * 4;
}

extern "C" int somi_nms_f32(const float *pred, int B, int n, int nc, float conf_thres, float iou_thres, int multi_label,
                            int agnostic, uint64_t classes_mask, int max_det, float *det, int32_t *count, void *workspace,
                            size_t workspace_bytes, somi_stream_t stream) {
    SOMI_REQUIRE(pred && det && count && workspace, SOMI_EINVAL, "nms: null tensor");
    SOMI_REQUIRE(B > 0 && n > 0 && nc > 0 && nc <= 64, SOMI_EINVAL, "nms: bad sizes (nc <= 64)");
    SOMI_REQUIRE(conf_thres >= 0.f && conf_thres <= 1.f, SOMI_EINVAL,
                 "Invalid Confidence threshold %g, valid values are between 0.0 and 1.0", conf_thres);   // general.py:635
    SOMI_REQUIRE(iou_thres >= 0.f && iou_thres <= 1.f, SOMI_EINVAL, "Invalid IoU %g, valid values are between 0.0 and 1.0",
                 iou_thres);                                                                             // general.py:636
    SOMI_REQUIRE(max_det > 0 && max_det <= MAX_DET_CAP, SOMI_EINVAL, "nms: max_det must be in [1, %d]", MAX_DET_CAP);
    SOMI_REQUIRE((size_t)n * nc < (1ull << 32), SOMI_EINVAL, "nms: n*nc must fit 32 bits");
    multi_label = (multi_label && nc > 1) ? 1 : 0;                                                       // general.py:643
    SOMI_REQUIRE(workspace_bytes >= somi_nms_workspace_bytes(B, n, nc, multi_label), SOMI_EWORKSPACE, "nms: workspace too small");
    NmsArgs a;
    a.pred = pred; a.B = B; a.n = n; a.nc = nc; a.no = nc + 5;
    a.conf_thres = conf_thres; a.iou_thres = iou_thres; a.multi_label = multi_label; a.agnostic = agnostic;
    a.max_det = max_det; a.classes_mask = classes_mask;
    a.nchunk = (n + NMS_CHUNK - 1) / NMS_CHUNK;
    a.cap = n * (multi_label ? nc : 1);
    char *w = static_cast<char *>(workspace);
    const size_t s_chunk = align_up((size_t)B * a.nchunk * 4, 256), s_tot = align_up((size_t)B * 4, 256);
    const size_t s_buf = align_up((size_t)B * a.cap * 4, 256);
    a.chunk_cnt = reinterpret_cast<int *>(w); w += s_chunk;
    a.chunk_off = reinterpret_cast<int *>(w); w += s_chunk;
    a.total = reinterpret_cast<int *>(w); w += s_tot;
    a.keyA = reinterpret_cast<uint32_t *>(w); w += s_buf;
    a.valA = reinterpret_cast<uint32_t *>(w); w += s_buf;
    a.keyB = reinterpret_cast<uint32_t *>(w); w += s_buf;
    a.valB = reinterpret_cast<uint32_t *>(w);
    a.det = det; a.count = count;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(nms_count_emit_kernel<false>, dim3(a.nchunk, B), dim3(256), 0, s, a);
    hipLaunchKernelGGL(nms_scan_kernel, dim3(B), dim3(256), 0, s, a);
    hipLaunchKernelGGL(nms_count_emit_kernel<true>, dim3(a.nchunk, B), dim3(256), 0, s, a);
    hipLaunchKernelGGL(nms_sort_kernel, dim3(B), dim3(1024), 0, s, a);
    const size_t lds = (size_t)(max_det + ROUND) * sizeof(BoxO) + (size_t)ROUND * (ROUND / 32) * 4 + (ROUND / 32 + 2) * 4;
    hipLaunchKernelGGL(nms_greedy_kernel, dim3(B), dim3(ROUND), lds, s, a);
    return launch_status("somi_nms_f32");
}
