// FCOS post-head kernels: candidate selection + box decode (ordered stream compaction), stable radix sort by score,
// greedy batched NMS with early exit at top-k.  Everything takes its element counts from device memory, so the whole
// detection tail is a fixed launch sequence with no host round trip.
//
// Reference call sites: fcos_outputs.py:396-466 (forward_for_single_feature_map), :372-394 (level concat),
// :468-495 (select_over_all_levels), layers/ml_nms.py:65-98 -> detectron2 batched_nms -> torchvision nms
// (sources absent; algorithm restated in oracle/oracle_ops.c and oracle/centermask_oracle.py:batched_nms).
#include "cmk_common.hpp"

namespace cmk {

constexpr int SEL_CHUNK = 4096;   // flat (location*C + class) elements per block = 4 waves x 16 iterations x 64 lanes
constexpr int MAX_LEVELS = 8;

struct SelLevels {
    const float* logits[MAX_LEVELS];
    const float* regctr[MAX_LEVELS];
    int W[MAX_LEVELS];
    int stride[MAX_LEVELS];
    int elems[MAX_LEVELS];        // H*W*C
    int blk_begin[MAX_LEVELS + 1];  // first block of each level within an image
    int num_levels;
};

__device__ inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ inline int find_level(const SelLevels& L, int b) {
    int l = 0;
#pragma unroll
    for (int i = 1; i < MAX_LEVELS; ++i)
        if (i < L.num_levels && b >= L.blk_begin[i]) l = i;
    return l;
}

// K1: per-block candidate counts.  grid = (blocks_per_image, N)
// candidate test of fcos_outputs.py:410-414: sigmoid(cls) > thr, or with THRESH_WITH_CTR sigmoid(cls) * sigmoid(ctr) > thr
__device__ inline bool is_candidate(const float* lg, const float* rc, int e, int ne, int C, float thr, int with_ctr, float& p) {
    if (e >= ne) { p = 0.f; return false; }
    p = sigmoidf_(lg[e]);
    if (!with_ctr) return p > thr;
    return p * sigmoidf_(rc[(long)(e / C) * 5 + 4]) > thr;
}

__global__ __launch_bounds__(256) void sel_count_kernel(const SelLevels L, int C, float thr, int with_ctr, int32_t* __restrict__ block_counts) {
    const int b = blockIdx.x, n = blockIdx.y;
    const int l = find_level(L, b);
    const int e0 = (b - L.blk_begin[l]) * SEL_CHUNK;
    const int ne = L.elems[l];
    const float* lg = L.logits[l] + (long)n * ne;
    const float* rc = L.regctr[l] + (long)n * (ne / C) * 5;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int cnt = 0;
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
        int e = e0 + wave * 1024 + it * 64 + lane;
        float p;
        bool c = is_candidate(lg, rc, e, ne, C, thr, with_ctr, p);
        cnt += __popcll(__ballot(c));
    }
    __shared__ int wc[4];
    if (lane == 0) wc[wave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[(long)n * gridDim.x + b] = wc[0] + wc[1] + wc[2] + wc[3];
}

// K2: exclusive scan of the block counts of one image (in place) + total.  grid = N, block = 256
__global__ __launch_bounds__(256) void sel_scan_kernel(int32_t* __restrict__ block_counts, int blocks_per_image, int32_t* __restrict__ counts) {
    __shared__ int part[256];
    const int n = blockIdx.x;
    int32_t* bc = block_counts + (long)n * blocks_per_image;
    const int per = cdiv(blocks_per_image, 256);
    const int b0 = threadIdx.x * per, b1 = min(blocks_per_image, b0 + per);
    int s = 0;
    for (int b = b0; b < b1; ++b) s += bc[b];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < 256; ++i) { int v = part[i]; part[i] = run; run += v; }
        counts[n] = run;
    }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int b = b0; b < b1; ++b) { int v = bc[b]; bc[b] = run; run += v; }
}

// K3: ordered write of the candidates.  Same geometry as K1.
__global__ __launch_bounds__(256) void sel_write_kernel(const SelLevels L, int C, float thr, int with_ctr, const int32_t* __restrict__ block_offsets,
                                                       float* __restrict__ cand_box, float* __restrict__ cand_score,
                                                       int32_t* __restrict__ cand_cls, float* __restrict__ cand_loc, int cap) {
    const int b = blockIdx.x, n = blockIdx.y;
    const int l = find_level(L, b);
    const int e0 = (b - L.blk_begin[l]) * SEL_CHUNK;
    const int ne = L.elems[l];
    const float* lg = L.logits[l] + (long)n * ne;
    const float* rc = L.regctr[l] + (long)n * (ne / C) * 5;
    const int Wl = L.W[l];
    const float fstride = (float)L.stride[l];
    const float half = (float)(L.stride[l] / 2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

    // pass 1: this wave's count, to order the four waves of the block
    int cnt = 0;
    for (int it = 0; it < 16; ++it) {
        int e = e0 + wave * 1024 + it * 64 + lane;
        float p;
        bool c = is_candidate(lg, rc, e, ne, C, thr, with_ctr, p);
        cnt += __popcll(__ballot(c));
    }
    __shared__ int wc[4];
    if (lane == 0) wc[wave] = cnt;
    __syncthreads();
    int base = block_offsets[(long)n * gridDim.x + b];
    for (int w = 0; w < wave; ++w) base += wc[w];

    // pass 2: decode and write in flat order
    for (int it = 0; it < 16; ++it) {
        int e = e0 + wave * 1024 + it * 64 + lane;
        float p;
        const bool c = is_candidate(lg, rc, e, ne, C, thr, with_ctr, p);
        unsigned long long m = __ballot(c);
        int dst = base + __popcll(m & lt);
        base += __popcll(m);
        if (c && dst < cap) {
            int loc = e / C, cls = e - loc * C;
            const float* r = rc + (long)loc * 5;
            float ctr = sigmoidf_(r[4]);
            float score = sqrtf(p * ctr);                    // fcos_outputs.py:419-420,460
            float lx = (float)((loc % Wl) * L.stride[l]) + half;   // fcos.py:131-144
            float ly = (float)((loc / Wl) * L.stride[l]) + half;
            float r0 = r[0] * fstride, r1 = r[1] * fstride, r2 = r[2] * fstride, r3 = r[3] * fstride;  // :384
            long o = (long)n * cap + dst;
            cand_box[o * 4 + 0] = lx - r0;                   // :451-456
            cand_box[o * 4 + 1] = ly - r1;
            cand_box[o * 4 + 2] = lx + r2;
            cand_box[o * 4 + 3] = ly + r3;
            cand_score[o] = score;
            cand_cls[o] = cls;
            cand_loc[o * 2 + 0] = lx;
            cand_loc[o * 2 + 1] = ly;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Per-image stable LSD radix sort (descending score) + greedy NMS with early exit.  One 1024-thread block per image.
// ---------------------------------------------------------------------------------------------------------------
constexpr int NT = 1024;
constexpr int NW = NT / 64;
constexpr int MAXK = 1024;      // POST_NMS_TOPK_TEST the keep list (LDS) is sized for

__device__ inline float iou_f(float ax1, float ay1, float ax2, float ay2, float aarea, float bx1, float by1, float bx2, float by2,
                              float barea) {
    float xx1 = fmaxf(ax1, bx1), yy1 = fmaxf(ay1, by1);
    float xx2 = fminf(ax2, bx2), yy2 = fminf(ay2, by2);
    float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
    float inter = w * h;
    return inter / (aarea + barea - inter);
}

__global__ __launch_bounds__(NT) void nms_topk_kernel(const float* __restrict__ cand_box, const float* __restrict__ cand_score,
                                                     const int32_t* __restrict__ cand_cls, const float* __restrict__ cand_loc,
                                                     const int32_t* __restrict__ counts, int cap, float thr, int topk,
                                                     float* __restrict__ out_box, float* __restrict__ out_score,
                                                     int64_t* __restrict__ out_cls, float* __restrict__ out_loc,
                                                     int32_t* __restrict__ out_idx, int32_t* __restrict__ out_count,
                                                     uint32_t* __restrict__ sort_ws) {
    __shared__ int hist[256];
    __shared__ int base[256];
    __shared__ int wcnt[NW][256];
    __shared__ float red[NW];
    __shared__ float s_maxc;
    __shared__ float kx1[MAXK], ky1[MAXK], kx2[MAXK], ky2[MAXK], karea[MAXK];
    __shared__ int kcls[MAXK];
    __shared__ int s_nkept;

    const int n = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cnt = min(counts[n], cap);
    const float* box = cand_box + (long)n * cap * 4;
    const float* score = cand_score + (long)n * cap;
    const int32_t* cls = cand_cls + (long)n * cap;
    uint32_t* k0 = sort_ws + ((long)n * 4 + 0) * cap;
    uint32_t* v0 = sort_ws + ((long)n * 4 + 1) * cap;
    uint32_t* k1 = sort_ws + ((long)n * 4 + 2) * cap;
    uint32_t* v1 = sort_ws + ((long)n * 4 + 3) * cap;

    // keys: descending score == ascending ~bits (scores are non-negative floats); NaN scores sort first like torch
    float lmax = -INFINITY;
    for (int i = tid; i < cnt; i += NT) {
        k0[i] = ~__float_as_uint(score[i]);
        v0[i] = (uint32_t)i;
        const float* b = box + (long)i * 4;
        lmax = fmaxf(lmax, fmaxf(fmaxf(b[0], b[1]), fmaxf(b[2], b[3])));
    }
    lmax = wave_max(lmax);
    if (lane == 0) red[wave] = lmax;
    __syncthreads();
    if (tid == 0) {
        float m = red[0];
        for (int w = 1; w < NW; ++w) m = fmaxf(m, red[w]);
        s_maxc = m;            // boxes.max() of batched_nms's coordinate trick
        s_nkept = 0;
    }
    __syncthreads();

    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    // The greedy scan below stops after `topk` survivors, so it only ever consumes a prefix of the descending-score order.  With many
    // candidates (the fork applies no pre-NMS top-k, fcos_outputs.py:444-449) sorting all of them is most of this kernel: attempt 0
    // sorts only the candidates whose key's top 12 bits are below the bin where the running count first reaches NEED — a superset of
    // the NEED best, compacted IN ORDER so that the radix sort stays stable on the candidate index — and attempt 1 (everything) runs
    // only if that prefix is exhausted before `topk` boxes are kept.
    const int NEED = max(2048, 16 * topk);
    int sorted_cnt = cnt;
    for (int attempt = (cnt > 2 * NEED ? 0 : 1); attempt < 2; ++attempt) {
    uint32_t *kin = k0, *vin = v0, *kout = k1, *vout = v1;
    sorted_cnt = cnt;
    if (attempt == 1 && cnt > 2 * NEED) {          // the subset pass overwrote (k0, v0)
        for (int i = tid; i < cnt; i += NT) { k0[i] = ~__float_as_uint(score[i]); v0[i] = (uint32_t)i; }
        __threadfence_block();
        __syncthreads();
    }
    if (attempt == 0) {
        int* hist12 = &wcnt[0][0];                  // 4096 bins over the key's top 12 bits
        for (int j = tid; j < 4096; j += NT) hist12[j] = 0;
        __syncthreads();
        for (int i = tid; i < cnt; i += NT) atomicAdd(&hist12[k0[i] >> 20], 1);
        __syncthreads();
        // thread t owns bins 4t..4t+3; block-wide exclusive prefix of the per-thread sums
        const int h0 = hist12[4 * tid], h1 = hist12[4 * tid + 1], h2 = hist12[4 * tid + 2], h3 = hist12[4 * tid + 3];
        const int mine = h0 + h1 + h2 + h3;
        int incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
        if (lane == 63) base[wave] = incl;
        __syncthreads();
        int pre = incl - mine;
        for (int w = 0; w < wave; ++w) pre += base[w];
        if (pre < NEED && pre + mine >= NEED) {     // exactly one thread: the bin where the count reaches NEED
            int c = pre + h0, d = 4 * tid;
            if (c < NEED) { c += h1; ++d; }
            if (c < NEED) { c += h2; ++d; }
            if (c < NEED) { c += h3; ++d; }
            hist[0] = d;
            hist[1] = c;
        }
        __syncthreads();
        const uint32_t dmax = (uint32_t)hist[0];
        sorted_cnt = hist[1];
        __syncthreads();
        // ordered compaction of the candidates of bins <= dmax into (k1, v1)
        int run = 0;
        for (int t0 = 0; t0 < cnt; t0 += NT) {
            const int i = t0 + tid;
            const bool take = i < cnt && (k0[i] >> 20) <= dmax;
            const unsigned long long bm = __ballot(take);
            if (lane == 0) base[wave] = __popcll(bm);
            __syncthreads();
            int pos = run + __popcll(bm & lt);
            for (int w = 0; w < wave; ++w) pos += base[w];
            if (take) { k1[pos] = k0[i]; v1[pos] = (uint32_t)i; }
            int tot = 0;
            for (int w = 0; w < NW; ++w) tot += base[w];
            run += tot;
            __syncthreads();
        }
        __threadfence_block();
        __syncthreads();
        kin = k1; vin = v1; kout = k0; vout = v0;
    }
    const int scnt = sorted_cnt;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = pass * 8;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        for (int i = tid; i < scnt; i += NT) atomicAdd(&hist[(kin[i] >> shift) & 255], 1);
        __syncthreads();
        if (tid == 0) {
            int run = 0;
            for (int d = 0; d < 256; ++d) { base[d] = run; run += hist[d]; }
        }
        __syncthreads();
        const bool trivial = false;
        (void)trivial;
        for (int t0 = 0; t0 < scnt; t0 += NT) {
            for (int j = tid; j < NW * 256; j += NT) (&wcnt[0][0])[j] = 0;
            __syncthreads();
            const int i = t0 + tid;
            const bool valid = i < scnt;
            uint32_t key = valid ? kin[i] : 0u, val = valid ? vin[i] : 0u;
            int digit = valid ? (int)((key >> shift) & 255) : 256;
            // lanes of this wave with the same digit
            unsigned long long peers = __ballot(valid);
#pragma unroll
            for (int bit = 0; bit < 8; ++bit) {
                unsigned long long bm = __ballot((digit >> bit) & 1);
                peers &= ((digit >> bit) & 1) ? bm : ~bm;
            }
            int rank = __popcll(peers & lt);
            if (valid && rank == 0) wcnt[wave][digit] = __popcll(peers);
            __syncthreads();
            if (valid) {
                int pre = base[digit];
                for (int w = 0; w < wave; ++w) pre += wcnt[w][digit];
                kout[pre + rank] = key;
                vout[pre + rank] = val;
            }
            __syncthreads();
            if (tid < 256) {
                int s = 0;
                for (int w = 0; w < NW; ++w) s += wcnt[w][tid];
                base[tid] += s;
            }
            __syncthreads();
        }
        uint32_t* t;
        t = kin; kin = kout; kout = t;
        t = vin; vin = vout; vout = t;
        __threadfence_block();
        __syncthreads();
    }
    // after 4 passes the sorted order is back in the buffer the passes started from
    const uint32_t* order = vin;

    // greedy suppression by the first wave; stop once topk boxes are kept
    if (wave == 0) {
        const bool per_class = cnt >= 40000;                 // detectron2 batched_nms switches strategy there
        const float off_unit = per_class ? 0.0f : (s_maxc + 1.0f);
        int nkept = 0;
        for (int c0 = 0; c0 < scnt && nkept < topk; c0 += 64) {
            const int pos = c0 + lane;
            bool alive = pos < scnt;
            int idx = alive ? (int)order[pos] : 0;
            float x1 = 0, y1 = 0, x2 = 0, y2 = 0, area = 0;
            int mycls = -1;
            if (alive) {
                const float* b = box + (long)idx * 4;
                mycls = cls[idx];
                float off = (float)mycls * off_unit;
                x1 = b[0] + off; y1 = b[1] + off; x2 = b[2] + off; y2 = b[3] + off;
                area = (x2 - x1) * (y2 - y1);
                for (int k = 0; k < nkept; ++k) {
                    if (per_class && kcls[k] != mycls) continue;
                    if (iou_f(kx1[k], ky1[k], kx2[k], ky2[k], karea[k], x1, y1, x2, y2, area) > thr) { alive = false; break; }
                }
            }
            for (int t = 0; t < 64 && nkept < topk; ++t) {
                unsigned long long am = __ballot(alive);
                if (!((am >> t) & 1ull)) continue;
                // lane t is kept: record it, suppress later lanes
                float tx1 = __shfl(x1, t, 64), ty1 = __shfl(y1, t, 64), tx2 = __shfl(x2, t, 64), ty2 = __shfl(y2, t, 64);
                float tarea = __shfl(area, t, 64);
                int tcls = __shfl(mycls, t, 64);
                int tidx = __shfl(idx, t, 64);
                if (lane == 0) {
                    kx1[nkept] = tx1; ky1[nkept] = ty1; kx2[nkept] = tx2; ky2[nkept] = ty2; karea[nkept] = tarea; kcls[nkept] = tcls;
                    long o = (long)n * topk + nkept;
                    const float* b = box + (long)tidx * 4;
                    out_box[o * 4 + 0] = b[0]; out_box[o * 4 + 1] = b[1]; out_box[o * 4 + 2] = b[2]; out_box[o * 4 + 3] = b[3];
                    out_score[o] = score[tidx];
                    out_cls[o] = (int64_t)tcls;
                    out_loc[o * 2 + 0] = cand_loc[((long)n * cap + tidx) * 2 + 0];
                    out_loc[o * 2 + 1] = cand_loc[((long)n * cap + tidx) * 2 + 1];
                    out_idx[o] = tidx;
                }
                ++nkept;
                if (lane > t && alive && (!per_class || tcls == mycls) &&
                    iou_f(tx1, ty1, tx2, ty2, tarea, x1, y1, x2, y2, area) > thr)
                    alive = false;
                if (lane == t) alive = false;
            }
        }
        if (lane == 0) {
            out_count[n] = nkept;
            s_nkept = nkept;
        }
    }
    __syncthreads();
    if (s_nkept >= topk || sorted_cnt >= cnt) break;      // enough survivors, or everything was sorted
    __syncthreads();
    }
    __syncthreads();
    // zero the unused tail so padded consumers read finite values
    for (int k = s_nkept + tid; k < topk; k += NT) {
        long o = (long)n * topk + k;
        out_box[o * 4 + 0] = out_box[o * 4 + 1] = out_box[o * 4 + 2] = out_box[o * 4 + 3] = 0.f;
        out_score[o] = 0.f;
        out_cls[o] = 0;
        out_loc[o * 2] = out_loc[o * 2 + 1] = 0.f;
        out_idx[o] = -1;
    }
}

static int fill_levels(const cmk_fcos_level* levels, int num_levels, int C, SelLevels& L) {
    if (num_levels < 1 || num_levels > MAX_LEVELS) return -1;
    int blk = 0;
    L.num_levels = num_levels;
    for (int l = 0; l < MAX_LEVELS; ++l) {
        if (l < num_levels) {
            long e = (long)levels[l].H * levels[l].W * C;
            if (e <= 0 || e > 0x7fffffffL) return -1;
            L.logits[l] = levels[l].logits; L.regctr[l] = levels[l].regctr;
            L.W[l] = levels[l].W; L.stride[l] = levels[l].stride; L.elems[l] = (int)e;
            L.blk_begin[l] = blk;
            blk += (int)((e + SEL_CHUNK - 1) / SEL_CHUNK);
        } else {
            L.logits[l] = nullptr; L.regctr[l] = nullptr; L.W[l] = 1; L.stride[l] = 1; L.elems[l] = 0; L.blk_begin[l] = blk;
        }
    }
    L.blk_begin[MAX_LEVELS] = blk;
    return blk;
}

}  // namespace cmk

using namespace cmk;

extern "C" int64_t cmk_fcos_select_ws_len(const cmk_fcos_level* levels, int num_levels, int N, int C) {
    SelLevels L;
    int blk = fill_levels(levels, num_levels, C, L);
    return blk < 0 ? -1 : (int64_t)blk * N;
}

extern "C" int cmk_fcos_select(const cmk_fcos_level* levels, int num_levels, int N, int C, float pre_nms_thresh, int thresh_with_ctr, float* cand_box,
                               float* cand_score, int32_t* cand_cls, float* cand_loc, int32_t* counts, int32_t* block_counts,
                               int64_t block_counts_len, int cap, void* stream) {
    if (!levels || !cand_box || !cand_score || !cand_cls || !cand_loc || !counts || !block_counts)
        return fail(CMK_EINVAL, "fcos_select: null pointer%s", "");
    if (N < 1 || C < 1 || cap < 1) return fail(CMK_EINVAL, "fcos_select: bad sizes%s", "");
    SelLevels L;
    int blk = fill_levels(levels, num_levels, C, L);
    if (blk < 0) return fail(CMK_EINVAL, "fcos_select: bad level table%s", "");
    for (int l = 0; l < num_levels; ++l)
        if (!levels[l].logits || !levels[l].regctr) return fail(CMK_EINVAL, "fcos_select: null level pointer%s", "");
    if (block_counts_len < (int64_t)blk * N) return fail(CMK_EINVAL, "fcos_select: workspace too small%s", "");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sel_count_kernel, dim3(blk, N), dim3(256), 0, st, L, C, pre_nms_thresh, thresh_with_ctr, block_counts);
    int rc = check_launch("sel_count");
    if (rc) return rc;
    hipLaunchKernelGGL(sel_scan_kernel, dim3(N), dim3(256), 0, st, block_counts, blk, counts);
    rc = check_launch("sel_scan");
    if (rc) return rc;
    hipLaunchKernelGGL(sel_write_kernel, dim3(blk, N), dim3(256), 0, st, L, C, pre_nms_thresh, thresh_with_ctr, block_counts, cand_box, cand_score,
                       cand_cls, cand_loc, cap);
    return check_launch("sel_write");
}

extern "C" int cmk_nms_topk(const float* cand_box, const float* cand_score, const int32_t* cand_cls, const float* cand_loc,
                            const int32_t* counts, int N, int cap, float iou_thr, int topk, float* out_box, float* out_score,
                            int64_t* out_cls, float* out_loc, int32_t* out_idx, int32_t* out_count, uint32_t* sort_ws, void* stream) {
    if (!cand_box || !cand_score || !cand_cls || !cand_loc || !counts || !out_box || !out_score || !out_cls || !out_loc || !out_idx ||
        !out_count || !sort_ws)
        return fail(CMK_EINVAL, "nms_topk: null pointer%s", "");
    if (N < 1 || cap < 1 || topk < 1 || topk > MAXK) return fail(CMK_EINVAL, "nms_topk: need 1 <= topk <= %s%ld", "", (long)MAXK);
    hipLaunchKernelGGL(nms_topk_kernel, dim3(N), dim3(NT), 0, (hipStream_t)stream, cand_box, cand_score, cand_cls, cand_loc, counts, cap,
                       iou_thr, topk, out_box, out_score, out_cls, out_loc, out_idx, out_count, sort_ws);
    return check_launch("nms_topk");
}
