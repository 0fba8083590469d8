// Batched NMS on gfx950: utils/general.py:629-711 including the torchvision.ops.nms core (:694).
//
// Per image, entirely on the device (the reference loops over images in Python with nonzero() syncs):
//   1. count   candidates per 1024-row chunk           (obj > thr, then obj*cls > thr per class / best class)
//   2. scan    chunk counts -> offsets                  (compaction order == prediction order: box-major, class-minor)
//   3. emit    key = ~bits(score), val = row*nc + cls   (8 B per candidate; boxes are re-derived from `pred` when needed)
//   4. select  three-level radix select of the exact top max_nms = 30000 (:688-689) + order-preserving compaction
//      sort    parallel rank sort of the <= 30000 survivors, stable (ties keep prediction order = the oracle's
//              stable descending sort)
//   5. greedy  512 candidates per round: test against the kept list (LDS), build the round's 512x512 suppression bit
//              matrix, resolve it serially in one wave, append survivors; stop at max_det kept (:696-697).
// IoU arithmetic is written with explicit round-to-nearest intrinsics (no FMA contraction) in the oracle's operation
// order, so the kept index lists are bit-exact: area=(x2-x1)*(y2-y1) on class-offset boxes (offset 4096*cls, :692-693),
// inter=max(0,min(x2)-max(x1))*max(0,min(y2)-max(y1)), iou=inter/((area_i+area_j)-inter), suppress iff iou > thr.
#include "common.h"

namespace somi {

constexpr int NMS_CHUNK = 512;       // rows per count/emit workgroup (256 threads x 2 consecutive rows)
constexpr int MAX_NMS = 30000;       // utils/general.py:641
constexpr float MAX_WH = 4096.f;     // utils/general.py:640
constexpr int ROUND = 512;           // candidates per greedy round
constexpr int MAX_DET_CAP = 1024;
constexpr int NMS_MASK_WORDS = 16;   // class filter bit array: nc <= 1024

struct NmsArgs {
    const float *pred;
    int B, n, nc, no;
    float conf_thres, iou_thres;
    int multi_label, agnostic, max_det;
    uint64_t classes_mask[NMS_MASK_WORDS];   // bit c of word c/64 set = keep class c
    int nchunk, cap;                 // cap = candidate capacity per image
    int *chunk_cnt;                  // [B][nchunk]
    int *chunk_off;                  // [B][nchunk]
    int *total;                      // [B]
    uint32_t *keyA, *valA, *keyB, *valB;   // [B][cap]
    float *det;                      // [B][max_det][6]
    int32_t *count;                  // [B]
};

__device__ __forceinline__ bool class_ok(const uint64_t *mask, int c) { return (mask[c >> 6] >> (c & 63)) & 1ull; }

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

// count (EMIT=false) / emit (EMIT=true) over one chunk of NMS_CHUNK prediction rows.  The chunk's rows are staged in LDS
// with coalesced 16 B loads (a row is 4*(5+nc) bytes, so per-thread row reads straight from HBM touch every line many
// times: PMC showed 9.1 GB fetched for 0.26 GB of predictions); emitted (key, val) pairs are compacted in LDS and leave
// as contiguous stores.
constexpr int EMIT_STAGE = 4096;     // (key,val) pairs staged per chunk before falling back to direct stores
__global__ void nms_scan_kernel(const NmsArgs a);

// RPT = rows per thread (chunk = 256 * RPT rows); STAGE = false reads the rows straight from HBM (heads too wide for the LDS
// staging: nc > 110) and keeps only the emit staging in LDS.
template <bool EMIT, int RPT, bool STAGE>
__global__ __launch_bounds__(256) void nms_count_emit_kernel(const NmsArgs a) {
    extern __shared__ __attribute__((aligned(16))) float rows[];     // [256 * RPT][no] (STAGE), then the emit staging
    __shared__ int lds[4];
    constexpr int CHUNK = 256 * RPT;
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int rbase = chunk * CHUNK;
    const int nrow = min(CHUNK, a.n - rbase);
    const float *src = a.pred + ((size_t)b * a.n + rbase) * a.no;
    if (STAGE) {
        const int nfl = nrow * a.no;
        if (((reinterpret_cast<uintptr_t>(src) & 15u) == 0)) {
            for (int i = threadIdx.x * 4; i + 3 < nfl; i += 1024)
                *reinterpret_cast<float4 *>(rows + i) = *reinterpret_cast<const float4 *>(src + i);
            for (int i = (nfl & ~3) + threadIdx.x; i < nfl; i += 256) rows[i] = src[i];
        } else {
            for (int i = threadIdx.x; i < nfl; i += 256) rows[i] = src[i];
        }
        __syncthreads();
    }
    const float *rsrc = STAGE ? rows : src;
    const int rl0 = threadIdx.x * RPT;
    int cnt[RPT], mine = 0;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int rl = rl0 + i;
        cnt[i] = rl < nrow ? row_entries<false>(a, rsrc + (size_t)rl * a.no, rbase + rl, nullptr, nullptr, 0) : 0;
        mine += cnt[i];
    }
    int tot;
    const int pre = block_exscan_256(mine, &tot, lds);
    if (!EMIT) {
        if (threadIdx.x == 0) a.chunk_cnt[b * a.nchunk + chunk] = tot;
        return;
    }
    const int gbase = a.chunk_off[b * a.nchunk + chunk];
    uint32_t *keys = a.keyA + (size_t)b * a.cap, *vals = a.valA + (size_t)b * a.cap;
    if (tot <= EMIT_STAGE) {
        uint32_t *skey = reinterpret_cast<uint32_t *>(rows + (STAGE ? CHUNK * a.no : 0)), *sval = skey + EMIT_STAGE;
        int dst = pre;
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int rl = rl0 + i;
            if (cnt[i]) row_entries<true>(a, rsrc + (size_t)rl * a.no, rbase + rl, skey, sval, dst);
            dst += cnt[i];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < tot; i += 256) {
            keys[gbase + i] = skey[i];
            vals[gbase + i] = sval[i];
        }
    } else {
        int dst = gbase + pre;
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int rl = rl0 + i;
            if (cnt[i]) row_entries<true>(a, rsrc + (size_t)rl * a.no, rbase + rl, keys, vals, dst);
            dst += cnt[i];
        }
    }
}

// rows per count/emit workgroup for a head of nc classes: the largest of 512 / 256 whose LDS image (+ emit staging) fits
static int nms_chunk_rows(int nc) { return ((size_t)NMS_CHUNK * (nc + 5) * 4 + (size_t)EMIT_STAGE * 8 <= 150 * 1024) ? NMS_CHUNK : 256; }
static bool nms_rows_staged(int nc) { return (size_t)nms_chunk_rows(nc) * (nc + 5) * 4 + (size_t)EMIT_STAGE * 8 <= 150 * 1024; }

template <int RPT, bool STAGE>
static void launch_count_emit(const NmsArgs &a, hipStream_t s) {
    const size_t rows_lds = STAGE ? (size_t)256 * RPT * a.no * 4 : 0, emit_lds = rows_lds + (size_t)EMIT_STAGE * 8;
    if (emit_lds > 64 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&nms_count_emit_kernel<false, RPT, STAGE>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)rows_lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&nms_count_emit_kernel<true, RPT, STAGE>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)emit_lds);
    }
    hipLaunchKernelGGL((nms_count_emit_kernel<false, RPT, STAGE>), dim3(a.nchunk, a.B), dim3(256), rows_lds, s, a);
    hipLaunchKernelGGL(nms_scan_kernel, dim3(a.B), dim3(256), 0, s, a);
    hipLaunchKernelGGL((nms_count_emit_kernel<true, RPT, STAGE>), dim3(a.nchunk, a.B), dim3(256), emit_lds, s, a);
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

// ------------------------------------------------------------------------------------------------ top-k select
// Only the first max_nms = 30000 entries of the (score desc, prediction order) ranking are used (general.py:688-689).
// A three-level radix select (12 + 12 + 8 key bits, histograms in LDS then merged with global atomics) finds the
// exact 32-bit threshold key T of the 30000-th entry; an order-preserving compaction then keeps every entry with
// key < T plus the first r entries with key == T.  Images with <= 30000 candidates skip all of this.
constexpr int SEL_CHUNK = 4096;      // candidates per select workgroup (256 threads x 16)

struct SelState {                    // per image, in global memory (zeroed every call)
    int hist1[4096];
    int hist2[4096];
    int hist3[256];
};

// exclusive prefix search over a histogram in global memory by one workgroup of 256 threads:
// returns the first bin whose inclusive cumulative count exceeds `need` (0-based rank), and the count before it.
__device__ __forceinline__ void find_bin(const int *hist, int nbins, int need, int *bin_out, int *before_out, int *lds) {
    // lds: >= 260 ints
    const int tid = threadIdx.x;
    const int per = nbins / 256;                   // 16 or 1
    int s = 0;
    for (int i = 0; i < per; ++i) s += hist[tid * per + i];
    int tot;
    const int pre = block_exscan_256(s, &tot, lds);
    __shared__ int res[2];
    if (need >= pre && need < pre + s) {           // exactly one thread
        int run = pre;
        for (int i = 0; i < per; ++i) {
            const int c = hist[tid * per + i];
            if (need < run + c) { res[0] = tid * per + i; res[1] = run; break; }
            run += c;
        }
    }
    __syncthreads();
    *bin_out = res[0];
    *before_out = res[1];
    __syncthreads();
}

// level 1/2/3 histogram kernels: grid (nsel_chunks, B)
template <int LEVEL>
__global__ __launch_bounds__(256) void nms_select_hist_kernel(const NmsArgs a, SelState *st) {
    __shared__ int lh[4096];
    __shared__ int lds[264];
    const int b = blockIdx.y;
    const int n = a.total[b];
    if (n <= MAX_NMS) return;
    const int i0 = blockIdx.x * SEL_CHUNK;
    if (i0 >= n) return;
    SelState &S = st[b];
    uint32_t prefix = 0;
    if (LEVEL >= 2) {
        int b1, bef1;
        find_bin(S.hist1, 4096, MAX_NMS - 1, &b1, &bef1, lds);
        prefix = (uint32_t)b1 << 20;
        if (LEVEL == 3) {
            int b2, bef2;
            find_bin(S.hist2, 4096, MAX_NMS - 1 - bef1, &b2, &bef2, lds);
            prefix |= (uint32_t)b2 << 8;
        }
    }
    constexpr int NB = LEVEL == 3 ? 256 : 4096;
    for (int i = threadIdx.x; i < NB; i += 256) lh[i] = 0;
    __syncthreads();
    const uint32_t *keys = a.keyA + (size_t)b * a.cap;
    const int i1 = min(i0 + SEL_CHUNK, n);
    for (int i = i0 + threadIdx.x; i < i1; i += 256) {
        const uint32_t k = keys[i];
        if (LEVEL == 1) atomicAdd(&lh[k >> 20], 1);
        else if (LEVEL == 2) { if ((k >> 20) == (prefix >> 20)) atomicAdd(&lh[(k >> 8) & 0xFFFu], 1); }
        else { if ((k >> 8) == (prefix >> 8)) atomicAdd(&lh[k & 0xFFu], 1); }
    }
    __syncthreads();
    int *gh = LEVEL == 1 ? S.hist1 : (LEVEL == 2 ? S.hist2 : S.hist3);
    for (int i = threadIdx.x; i < NB; i += 256)
        if (lh[i]) atomicAdd(&gh[i], lh[i]);
}

// threshold key T and number of ties r to take, recomputed by whoever needs them
__device__ __forceinline__ void select_threshold(const SelState &S, uint32_t *T, int *r, int *lds) {
    int b1, bef1, b2, bef2, b3, bef3;
    find_bin(S.hist1, 4096, MAX_NMS - 1, &b1, &bef1, lds);
    find_bin(S.hist2, 4096, MAX_NMS - 1 - bef1, &b2, &bef2, lds);
    find_bin(S.hist3, 256, MAX_NMS - 1 - bef1 - bef2, &b3, &bef3, lds);
    *T = ((uint32_t)b1 << 20) | ((uint32_t)b2 << 8) | (uint32_t)b3;
    *r = MAX_NMS - (bef1 + bef2 + bef3);           // entries with key == T to keep (>= 1)
}

// order-preserving compaction in three steps (count -> scan -> write) over SEL_CHUNK-sized chunks.
// sel_cnt[b][chunk] = {#key<T, #key==T}; after the scan kernel: exclusive prefixes.
template <bool WRITE>
__global__ __launch_bounds__(256) void nms_select_compact_kernel(const NmsArgs a, const SelState *st, int2 *sel_cnt,
                                                                 int nsel_chunk) {
    __shared__ int lds[264];
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int n = a.total[b];
    if (n <= MAX_NMS) return;
    const int i0 = chunk * SEL_CHUNK + threadIdx.x * 16;       // 16 consecutive candidates per thread
    uint32_t T;
    int r;
    select_threshold(st[b], &T, &r, lds);
    const uint32_t *keys = a.keyA + (size_t)b * a.cap;
    const uint32_t *vals = a.valA + (size_t)b * a.cap;
    uint32_t k[16];
    int lt = 0, eq = 0;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int i = i0 + e;
        k[e] = i < n ? keys[i] : 0xFFFFFFFFu;
        lt += (i < n && k[e] < T) ? 1 : 0;
        eq += (i < n && k[e] == T) ? 1 : 0;
    }
    int tot_lt, tot_eq;
    const int pre_lt = block_exscan_256(lt, &tot_lt, lds);
    const int pre_eq = block_exscan_256(eq, &tot_eq, lds);
    if (!WRITE) {
        if (threadIdx.x == 0) sel_cnt[b * nsel_chunk + chunk] = make_int2(tot_lt, tot_eq);
        return;
    }
    const int2 base = sel_cnt[b * nsel_chunk + chunk];          // exclusive prefixes over chunks
    int eq_rank = base.y + pre_eq;                               // rank among ties, in prediction order
    // destination: entries keep their relative order: position = (#selected before me)
    // selected-before = lt_before + min(eq_before, r)
    int lt_rank = base.x + pre_lt;
    uint32_t *ok_ = a.keyB + (size_t)b * a.cap, *ov = a.valB + (size_t)b * a.cap;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int i = i0 + e;
        if (i >= n) break;
        const bool is_lt = k[e] < T, is_eq = k[e] == T;
        if (is_lt || (is_eq && eq_rank < r)) {
            const int pos = lt_rank + min(eq_rank, r);
            ok_[pos] = k[e];
            ov[pos] = vals[i];
        }
        lt_rank += is_lt ? 1 : 0;
        eq_rank += is_eq ? 1 : 0;
    }
}

__global__ __launch_bounds__(256) void nms_select_scan_kernel(const NmsArgs a, int2 *sel_cnt, int nsel_chunk) {
    __shared__ int lds[8];
    __shared__ int carry[2];
    const int b = blockIdx.x;
    if (a.total[b] <= MAX_NMS) return;
    if (threadIdx.x == 0) carry[0] = carry[1] = 0;
    __syncthreads();
    const int used = (a.total[b] + SEL_CHUNK - 1) / SEL_CHUNK;
    for (int c0 = 0; c0 < used; c0 += 256) {
        const int c = c0 + threadIdx.x;
        const int2 v = c < used ? sel_cnt[b * nsel_chunk + c] : make_int2(0, 0);
        int tx, ty;
        const int px = block_exscan_256(v.x, &tx, lds);
        const int py = block_exscan_256(v.y, &ty, lds);
        if (c < used) sel_cnt[b * nsel_chunk + c] = make_int2(carry[0] + px, carry[1] + py);
        __syncthreads();
        if (threadIdx.x == 0) { carry[0] += tx; carry[1] += ty; }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ rank sort
// Stable sort of the (<= 30000) selected entries by (key, position): every entry counts the entries that precede it.
// grid (ceil(30000/256), B); the candidate keys stream through LDS in tiles of 2048.  O(n^2) compares spread over the
// whole chip (n = 30000: 9e8 compares per image), deterministic, no atomics.  Output: sorted vals -> sortV.
__global__ __launch_bounds__(256) void nms_rank_sort_kernel(const NmsArgs a, uint32_t *sortV) {
    __shared__ uint32_t tile[2048];
    const int b = blockIdx.y;
    const int tot = a.total[b];
    const bool sel = tot > MAX_NMS;
    const int n = sel ? MAX_NMS : tot;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 >= n) return;
    const uint32_t *keys = (sel ? a.keyB : a.keyA) + (size_t)b * a.cap;
    const uint32_t *vals = (sel ? a.valB : a.valA) + (size_t)b * a.cap;
    const uint32_t mine = i < n ? keys[i] : 0xFFFFFFFFu;
    int rank = 0;
    for (int j0 = 0; j0 < n; j0 += 2048) {
        const int m = min(2048, n - j0);
        __syncthreads();
        for (int t = threadIdx.x; t < m; t += 256) tile[t] = keys[j0 + t];
        __syncthreads();
        // entries before me in position with key <= mine, entries after me with key < mine
        const int split = min(max(i - j0, 0), m);          // tile positions [0,split) precede me
        int cnt = 0;
        for (int t = 0; t < split; ++t) cnt += tile[t] <= mine ? 1 : 0;
        for (int t = split; t < m; ++t) cnt += tile[t] < mine ? 1 : 0;
        rank += cnt;
    }
    if (i < n) sortV[(size_t)b * MAX_NMS + rank] = vals[i];
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

__global__ __launch_bounds__(ROUND) void nms_greedy_kernel(const NmsArgs a, const uint32_t *sortV) {
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
    const uint32_t *vals = sortV + (size_t)b * MAX_NMS;
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
    const size_t nchunk = (size_t)(n + nms_chunk_rows(nc) - 1) / nms_chunk_rows(nc);
    const size_t cap = (size_t)n * (size_t)((multi_label && nc > 1) ? nc : 1);
    const size_t nsel_chunk = (cap + SEL_CHUNK - 1) / SEL_CHUNK;
    return align_up((size_t)B * nchunk * 4, 256) * 2 + align_up((size_t)B * 4, 256) + align_up((size_t)B * cap * 4, 256) * 4 +
           align_up((size_t)B * sizeof(SelState), 256) + align_up((size_t)B * nsel_chunk * sizeof(int2), 256) +
           align_up((size_t)B * MAX_NMS * 4, 256);
}

extern "C" int somi_nms_f32(const float *pred, int B, int n, int nc, float conf_thres, float iou_thres, int multi_label,
                            int agnostic, const uint64_t *classes_mask, int max_det, float *det, int32_t *count, void *workspace,
                            size_t workspace_bytes, somi_stream_t stream) {
    SOMI_REQUIRE(pred && det && count && workspace, SOMI_EINVAL, "nms: null tensor");
    SOMI_REQUIRE(B > 0 && n > 0 && nc > 0 && nc <= 64 * NMS_MASK_WORDS, SOMI_EINVAL, "nms: bad sizes (nc <= %d)", 64 * NMS_MASK_WORDS);
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
    a.max_det = max_det;
    for (int i = 0; i < NMS_MASK_WORDS; ++i)                     // host words, NULL = keep every class (general.py:676-677)
        a.classes_mask[i] = classes_mask ? (i < (nc + 63) / 64 ? classes_mask[i] : 0ull) : ~0ull;
    a.nchunk = (n + nms_chunk_rows(nc) - 1) / nms_chunk_rows(nc);
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
    a.valB = reinterpret_cast<uint32_t *>(w); w += s_buf;
    const int nsel_chunk = (a.cap + SEL_CHUNK - 1) / SEL_CHUNK;
    SelState *st = reinterpret_cast<SelState *>(w); w += align_up((size_t)B * sizeof(SelState), 256);
    int2 *sel_cnt = reinterpret_cast<int2 *>(w); w += align_up((size_t)B * nsel_chunk * sizeof(int2), 256);
    uint32_t *sortV = reinterpret_cast<uint32_t *>(w);
    a.det = det; a.count = count;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (nms_chunk_rows(nc) == NMS_CHUNK) launch_count_emit<NMS_CHUNK / 256, true>(a, s);
    else if (nms_rows_staged(nc)) launch_count_emit<1, true>(a, s);
    else launch_count_emit<1, false>(a, s);
    // top-30000 select (every kernel exits at once for images with <= 30000 candidates)
    (void)hipMemsetAsync(st, 0, (size_t)B * sizeof(SelState), s);
    hipLaunchKernelGGL(nms_select_hist_kernel<1>, dim3(nsel_chunk, B), dim3(256), 0, s, a, st);
    hipLaunchKernelGGL(nms_select_hist_kernel<2>, dim3(nsel_chunk, B), dim3(256), 0, s, a, st);
    hipLaunchKernelGGL(nms_select_hist_kernel<3>, dim3(nsel_chunk, B), dim3(256), 0, s, a, st);
    hipLaunchKernelGGL(nms_select_compact_kernel<false>, dim3(nsel_chunk, B), dim3(256), 0, s, a, st, sel_cnt, nsel_chunk);
    hipLaunchKernelGGL(nms_select_scan_kernel, dim3(B), dim3(256), 0, s, a, sel_cnt, nsel_chunk);
    hipLaunchKernelGGL(nms_select_compact_kernel<true>, dim3(nsel_chunk, B), dim3(256), 0, s, a, st, sel_cnt, nsel_chunk);
    hipLaunchKernelGGL(nms_rank_sort_kernel, dim3((MAX_NMS + 255) / 256, B), dim3(256), 0, s, a, sortV);
    const size_t lds = (size_t)(max_det + ROUND) * sizeof(BoxO) + (size_t)ROUND * (ROUND / 32) * 4 + (ROUND / 32 + 2) * 4;
    hipLaunchKernelGGL(nms_greedy_kernel, dim3(B), dim3(ROUND), lds, s, a, sortV);
    return launch_status("somi_nms_f32");
}
